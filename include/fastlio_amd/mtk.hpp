// mtk.hpp -- the on-manifold primitives of IKFoM's MTK that the measurement update touches:
// vect<3>, SO3 (unit quaternion, Eigen coeff order x,y,z,w) and S2 (gravity direction).
// Mirrors include/IKFoM_toolkit/mtk/{types/vect.hpp,types/SOn.hpp,types/S2.hpp,src/mtkmath.hpp} of
// the reference, including its quirks (noted inline).  Host only, fp64.
#pragma once
#include <cfloat>
#include <cmath>
#include <utility>

#include "smallmat.hpp"

namespace MTK {
using fastlio_amd::hat;
using fastlio_amd::M3;
using fastlio_amd::Mat;
using fastlio_amd::V3;
using fastlio_amd::Vec;

inline double tolerance() { return 1e-11; }  // mtkmath.hpp:122

// mtkmath.hpp:142-174
inline std::pair<double, double> cos_sinc_sqrt(double x2) {
    static const double taylor_0_bound = DBL_EPSILON;
    static const double taylor_2_bound = std::sqrt(taylor_0_bound);
    static const double taylor_n_bound = std::sqrt(taylor_2_bound);
    if (x2 >= taylor_n_bound) {
        const double x = std::sqrt(x2);
        return std::make_pair(std::cos(x), std::sin(x) / x);
    }
    static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
    double cosi = 1., sinc = 1;
    double term = -1 / 2. * x2;
    for (int i = 0; i < 3; ++i) {
        cosi += term;
        term *= inv[2 * i];
        sinc += term;
        term *= -inv[2 * i + 1] * x2;
    }
    return std::make_pair(cosi, sinc);
}

// mtkmath.hpp:235-247
inline M3 A_matrix(const V3& v) {
    const double squaredNorm = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const double norm = std::sqrt(squaredNorm);
    if (norm < tolerance()) return M3::Identity();
    const M3 H = hat(v);
    return M3::Identity() + H * ((1 - std::cos(norm)) / squaredNorm) + (H * H) * ((1 - std::sin(norm) / norm) / squaredNorm);
}

// Eigen::Quaternion<double>, restated: product, conjugate, rotate, toRotationMatrix
struct Quat {
    double x = 0, y = 0, z = 0, w = 1;
    Quat() {}
    Quat(double w_, double x_, double y_, double z_) : x(x_), y(y_), z(z_), w(w_) {}
    Quat conjugate() const { return Quat(w, -x, -y, -z); }
    Quat operator*(const Quat& b) const {
        return Quat(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                    w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x);
    }
    V3 operator*(const V3& v) const {  // _transformVector
        double uvx = y * v[2] - z * v[1], uvy = z * v[0] - x * v[2], uvz = x * v[1] - y * v[0];
        uvx += uvx; uvy += uvy; uvz += uvz;
        const double cx = y * uvz - z * uvy, cy = z * uvx - x * uvz, cz = x * uvy - y * uvx;
        V3 r;
        r[0] = (v[0] + w * uvx) + cx;
        r[1] = (v[1] + w * uvy) + cy;
        r[2] = (v[2] + w * uvz) + cz;
        return r;
    }
    M3 toRotationMatrix() const {
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
        const double twx = tx * w, twy = ty * w, twz = tz * w;
        const double txx = tx * x, txy = ty * x, txz = tz * x;
        const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
        M3 R;
        R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
        R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
        return R;
    }
};

// MTK::exp (mtkmath.hpp:249-256): vec part = sinc(scale|v|) * scale * v, returns cos(scale|v|)
inline double exp3(V3& res, const V3& v, double scale) {
    const double norm2 = v.squaredNorm();
    const std::pair<double, double> cs = cos_sinc_sqrt(scale * scale * norm2);
    const double mult = cs.second * scale;
    res = v * mult;
    return cs.first;
}

// vect<3> (types/vect.hpp:117-126)
struct vect3 : public V3 {
    enum { DOF = 3, DIM = 3, TYP = 0 };
    vect3() {}
    vect3(const V3& v) : V3(v) {}
    vect3(double a0, double a1, double a2) { a[0] = a0; a[1] = a1; a[2] = a2; }
    void boxplus(const V3& d, double scale = 1) { for (int i = 0; i < 3; ++i) a[i] += scale * d[i]; }
    void oplus(const V3& d, double scale = 1) { boxplus(d, scale); }
    void boxminus(V3& res, const vect3& other) const { for (int i = 0; i < 3; ++i) res[i] = a[i] - other.a[i]; }
};

// SO3 (types/SOn.hpp:180-297)
struct SO3 : public Quat {
    enum { DOF = 3, DIM = 3, TYP = 2 };
    SO3() {}
    SO3(const Quat& q) : Quat(q) {}
    static SO3 exp(const V3& dvec, double scale = 1) {  // :284-288
        SO3 res;
        V3 vec;
        res.w = exp3(vec, dvec, scale / 2);
        res.x = vec[0]; res.y = vec[1]; res.z = vec[2];
        return res;
    }
    static V3 log(const SO3& q) {  // :293-297 -> MTK::log(.., scale 2, plus_minus_periodicity = true)
        double nv = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
        if (nv < tolerance()) nv = tolerance();
        const double s = 2.0 / nv * std::atan(nv / q.w);
        V3 r;
        r[0] = s * q.x; r[1] = s * q.y; r[2] = s * q.z;
        return r;
    }
    void boxplus(const V3& v, double scale = 1) { *this = SO3(static_cast<const Quat&>(*this) * exp(v, scale)); }  // :233-236
    void oplus(const V3& v, double scale = 1) { boxplus(v, scale); }
    void boxminus(V3& res, const SO3& other) const { res = log(SO3(other.conjugate() * *this)); }  // :237-239
};

// S2<double, 98090, 10000, 1> (types/S2.hpp): |g| = 9.809, x-axis chart (S2_typ == 1)
struct S2 {
    enum { DOF = 2, DIM = 3, TYP = 1 };
    static constexpr double length = 98090.0 / 10000.0;
    V3 vec;
    S2() { vec[0] = length; }  // S2.hpp:114-118
    S2(double x, double y, double z) {
        vec[0] = x; vec[1] = y; vec[2] = z;
        const double n = vec.norm();
        vec = (vec / n) * length;
    }
    double operator[](int i) const { return vec[i]; }
    void S2_Bx(Mat<3, 2>& res) const {  // :215-231
        if (vec[0] + length > tolerance()) {
            res(0, 0) = -vec[1]; res(0, 1) = -vec[2];
            res(1, 0) = length - vec[1] * vec[1] / (length + vec[0]); res(1, 1) = -vec[2] * vec[1] / (length + vec[0]);
            res(2, 0) = -vec[2] * vec[1] / (length + vec[0]); res(2, 1) = length - vec[2] * vec[2] / (length + vec[0]);
            res = res / length;
        } else {
            res = Mat<3, 2>::Zero();
            res(1, 1) = -1;
            res(2, 0) = 1;
        }
    }
    void S2_hat(M3& res) const { res = hat(vec); }
    void S2_Nx_yy(Mat<2, 3>& res) const {  // :259-264
        Mat<3, 2> Bx;
        S2_Bx(Bx);
        res = (Bx.transpose() * (1 / length / length)) * hat(vec);
    }
    void S2_Mx(Mat<3, 2>& res, const Vec<2>& delta) const {  // :266-280
        Mat<3, 2> Bx;
        S2_Bx(Bx);
        if (delta.norm() < tolerance()) {
            res = (-hat(vec)) * Bx;
        } else {
            const V3 Bu = Bx * delta;
            // exp_delta = exp(Bu, scalar(1/2)): integer division, scale 0 -> identity rotation (:277)
            res = ((-hat(vec)) * A_matrix(Bu).transpose()) * Bx;
        }
    }
    void boxplus(const Vec<2>& delta, double scale = 1) {  // :136-142
        Mat<3, 2> Bx;
        S2_Bx(Bx);
        const V3 Bu = Bx * delta;
        V3 v;
        Quat q;
        q.w = exp3(v, Bu, scale / 2);
        q.x = v[0]; q.y = v[1]; q.z = v[2];
        vec = q.toRotationMatrix() * vec;
    }
    void oplus(const V3& delta, double scale = 1) {  // :129-134
        V3 v;
        Quat q;
        q.w = exp3(v, delta, scale / 2);
        q.x = v[0]; q.y = v[1]; q.z = v[2];
        vec = q.toRotationMatrix() * vec;
    }
    void boxminus(Vec<2>& res, const S2& other) const {  // :144-167
        const double v_sin = (hat(vec) * other.vec).norm();
        const double v_cos = vec[0] * other.vec[0] + vec[1] * other.vec[1] + vec[2] * other.vec[2];
        const double theta = std::atan2(v_sin, v_cos);
        if (v_sin < tolerance()) {
            if (std::fabs(theta) > tolerance()) { res[0] = 3.1415926; res[1] = 0; }
            else { res[0] = 0; res[1] = 0; }
        } else {
            Mat<3, 2> Bx;
            other.S2_Bx(Bx);
            res = ((Bx.transpose() * (theta / v_sin)) * hat(other.vec)) * vec;
        }
    }
};

}  // namespace MTK
