// use-ikfom.hpp -- mirror of the reference's include/use-ikfom.hpp: the 23-DOF / 24-DIM state manifold
// and the IMU process model, with the same type and member names so code written against the
// reference reads the same.  (MTK_BUILD_MANIFOLD's preprocessor machinery is replaced by a plain struct.)
#pragma once
#include <utility>
#include <vector>

#include "mtk.hpp"

typedef MTK::vect3 vect3;
typedef MTK::SO3 SO3;
typedef MTK::S2 S2;

// MTK_BUILD_MANIFOLD(state_ikfom, pos rot offset_R_L_I offset_T_L_I vel bg ba grav) -- use-ikfom.hpp:12-21
struct state_ikfom {
    enum { DOF = 23, DIM = 24 };
    typedef double scalar;
    vect3 pos;
    SO3 rot;
    SO3 offset_R_L_I;
    vect3 offset_T_L_I;
    vect3 vel;
    vect3 bg;
    vect3 ba;
    S2 grav;
    // (DOF idx, DIM idx) tables, as build_S2_state/build_SO3_state/build_vect_state produce them
    // (build_manifold.hpp:201-212)
    std::vector<std::pair<int, int> > S2_state;
    std::vector<std::pair<int, int> > SO3_state;
    std::vector<std::pair<std::pair<int, int>, int> > vect_state;
    void build_S2_state() { S2_state.assign(1, std::make_pair(21, 21)); }
    void build_SO3_state() { SO3_state = {std::make_pair(3, 3), std::make_pair(6, 6)}; }
    void build_vect_state() {
        vect_state.clear();
        const int idx[5] = {0, 9, 12, 15, 18};
        for (int i = 0; i < 5; ++i) vect_state.push_back(std::make_pair(std::make_pair(idx[i], idx[i]), 3));
    }
    static fastlio_amd::V3 seg3(const fastlio_amd::Vec<23>& v, int i) { return v.block<3, 1>(i, 0); }
    void boxplus(const fastlio_amd::Vec<23>& d, double scale = 1) {  // build_manifold.hpp:192-194
        pos.boxplus(seg3(d, 0), scale);
        rot.boxplus(seg3(d, 3), scale);
        offset_R_L_I.boxplus(seg3(d, 6), scale);
        offset_T_L_I.boxplus(seg3(d, 9), scale);
        vel.boxplus(seg3(d, 12), scale);
        bg.boxplus(seg3(d, 15), scale);
        ba.boxplus(seg3(d, 18), scale);
        grav.boxplus(d.block<2, 1>(21, 0), scale);
    }
    void oplus(const fastlio_amd::Vec<24>& d, double scale = 1) {  // :195-197
        pos.oplus(d.block<3, 1>(0, 0), scale);
        rot.oplus(d.block<3, 1>(3, 0), scale);
        offset_R_L_I.oplus(d.block<3, 1>(6, 0), scale);
        offset_T_L_I.oplus(d.block<3, 1>(9, 0), scale);
        vel.oplus(d.block<3, 1>(12, 0), scale);
        bg.oplus(d.block<3, 1>(15, 0), scale);
        ba.oplus(d.block<3, 1>(18, 0), scale);
        grav.oplus(d.block<3, 1>(21, 0), scale);
    }
    void boxminus(fastlio_amd::Vec<23>& res, const state_ikfom& o) const {  // :198-200
        fastlio_amd::V3 t;
        pos.boxminus(t, o.pos); res.set_block(0, 0, t);
        rot.boxminus(t, o.rot); res.set_block(3, 0, t);
        offset_R_L_I.boxminus(t, o.offset_R_L_I); res.set_block(6, 0, t);
        offset_T_L_I.boxminus(t, o.offset_T_L_I); res.set_block(9, 0, t);
        vel.boxminus(t, o.vel); res.set_block(12, 0, t);
        bg.boxminus(t, o.bg); res.set_block(15, 0, t);
        ba.boxminus(t, o.ba); res.set_block(18, 0, t);
        fastlio_amd::Vec<2> t2;
        grav.boxminus(t2, o.grav); res.set_block(21, 0, t2);
    }
    void S2_hat(fastlio_amd::M3& res, int idx) { if (idx == 21) grav.S2_hat(res); }
    void S2_Nx_yy(fastlio_amd::Mat<2, 3>& res, int idx) { if (idx == 21) grav.S2_Nx_yy(res); }
    void S2_Mx(fastlio_amd::Mat<3, 2>& res, const fastlio_amd::Vec<2>& dx, int idx) { if (idx == 21) grav.S2_Mx(res, dx); }

    // flat 26-double layout of the C ABI (include/fastlio_hip.h)
    void to_flat(double x[26]) const {
        for (int i = 0; i < 3; ++i) { x[i] = pos[i]; x[11 + i] = offset_T_L_I[i]; x[14 + i] = vel[i]; x[17 + i] = bg[i]; x[20 + i] = ba[i]; x[23 + i] = grav.vec[i]; }
        x[3] = rot.x; x[4] = rot.y; x[5] = rot.z; x[6] = rot.w;
        x[7] = offset_R_L_I.x; x[8] = offset_R_L_I.y; x[9] = offset_R_L_I.z; x[10] = offset_R_L_I.w;
    }
    void from_flat(const double x[26]) {
        for (int i = 0; i < 3; ++i) { pos[i] = x[i]; offset_T_L_I[i] = x[11 + i]; vel[i] = x[14 + i]; bg[i] = x[17 + i]; ba[i] = x[20 + i]; grav.vec[i] = x[23 + i]; }
        rot.x = x[3]; rot.y = x[4]; rot.z = x[5]; rot.w = x[6];
        offset_R_L_I.x = x[7]; offset_R_L_I.y = x[8]; offset_R_L_I.z = x[9]; offset_R_L_I.w = x[10];
    }
};

struct input_ikfom {  // use-ikfom.hpp:23-26
    enum { DOF = 6, DIM = 6 };
    vect3 acc;
    vect3 gyro;
};

// use-ikfom.hpp:35-43
inline fastlio_amd::Mat<12, 12> process_noise_cov() {
    fastlio_amd::Mat<12, 12> cov;
    for (int i = 0; i < 3; ++i) {
        cov(0 + i, 0 + i) = 0.0001;
        cov(3 + i, 3 + i) = 0.0001;
        cov(6 + i, 6 + i) = 0.00001;
        cov(9 + i, 9 + i) = 0.00001;
    }
    return cov;
}

// use-ikfom.hpp:47-59
inline fastlio_amd::Vec<24> get_f(state_ikfom& s, const input_ikfom& in) {
    fastlio_amd::Vec<24> res;
    fastlio_amd::V3 omega;
    in.gyro.boxminus(omega, s.bg);
    fastlio_amd::V3 am;
    in.acc.boxminus(am, s.ba);
    const fastlio_amd::V3 a_inertial = s.rot * am;
    for (int i = 0; i < 3; ++i) {
        res[i] = s.vel[i];
        res[i + 3] = omega[i];
        res[i + 12] = a_inertial[i] + s.grav[i];
    }
    return res;
}

// use-ikfom.hpp:61-77
inline fastlio_amd::Mat<24, 23> df_dx(state_ikfom& s, const input_ikfom& in) {
    using namespace fastlio_amd;
    Mat<24, 23> cov;
    cov.set_block(0, 12, M3::Identity());
    V3 acc_;
    in.acc.boxminus(acc_, s.ba);
    const M3 R = s.rot.toRotationMatrix();
    cov.set_block(12, 3, -(R * hat(acc_)));
    cov.set_block(12, 18, -R);
    Mat<3, 2> grav_matrix;
    s.S2_Mx(grav_matrix, Vec<2>::Zero(), 21);
    cov.set_block(12, 21, grav_matrix);
    cov.set_block(3, 15, -M3::Identity());
    return cov;
}

// use-ikfom.hpp:80-88
inline fastlio_amd::Mat<24, 12> df_dw(state_ikfom& s, const input_ikfom& in) {
    using namespace fastlio_amd;
    (void)in;
    Mat<24, 12> cov;
    cov.set_block(12, 3, -s.rot.toRotationMatrix());
    cov.set_block(3, 0, -M3::Identity());
    cov.set_block(15, 6, M3::Identity());
    cov.set_block(18, 9, M3::Identity());
    return cov;
}
