/*
 * oracle_math.c -- TEST INFRASTRUCTURE (see fastlio_oracle.h header).  PARITY UNPINNED.
 *
 * Plain-C restatement of the host-side math FAST-LIO2's measurement update uses:
 *   - Eigen::Quaternion product / rotate / toRotationMatrix          [Eigen, restated]
 *   - MTK SO3 / S2 / vect boxplus, boxminus, A_matrix, S2_Bx/Nx/Mx   include/IKFoM_toolkit/mtk/
 *   - state_ikfom compound boxplus/boxminus                           include/use-ikfom.hpp:12-21
 *   - fixed-size inverse() (partial-pivot LU)                         [Eigen, restated]
 *   - ColPivHouseholderQR 5x3 fp32 solve, esti_plane                  include/common_lib.h:225-257
 *   - esekf::predict + get_f/df_dx/df_dw                              esekfom.hpp:279-383, use-ikfom.hpp:47-88
 * Compile with -ffp-contract=off: the reference is built -O3 without -march (CMakeLists.txt:8,14),
 * i.e. baseline x86-64, so no operation is ever fused.
 */
#include "fastlio_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NDOF ORC_NDOF
#define TOL 1e-11 /* MTK::tolerance<double>(), mtkmath.hpp:122 */

/* ------------------------------------------------------------------ small helpers */
static void hat3(const double v[3], double M[9]) { /* mtkmath.hpp:176-183 */
    M[0] = 0; M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2]; M[4] = 0; M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0]; M[8] = 0;
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
            T[i * 3 + j] = s;
        }
    memcpy(C, T, sizeof(T));
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void mat3_T(const double A[9], double T[9]) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[j * 3 + i] = A[i * 3 + j];
    memcpy(T, t, sizeof(t));
}

/* ------------------------------------------------------------------ Eigen::Quaternion (xyzw) */
void orc_quat_mul(const double a[4], const double b[4], double o[4]) {
    /* Eigen quat_product: (w,x,y,z) Hamilton product */
    double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    double w = aw * bw - ax * bx - ay * by - az * bz;
    double x = aw * bx + ax * bw + ay * bz - az * by;
    double y = aw * by + ay * bw + az * bx - ax * bz;
    double z = aw * bz + az * bw + ax * by - ay * bx;
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
static void quat_conj(const double q[4], double o[4]) {
    o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3];
}
void orc_quat_rot(const double q[4], const double v[3], double o[3]) {
    /* Eigen QuaternionBase::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv */
    double uvx = q[1] * v[2] - q[2] * v[1];
    double uvy = q[2] * v[0] - q[0] * v[2];
    double uvz = q[0] * v[1] - q[1] * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    double cx = q[1] * uvz - q[2] * uvy;
    double cy = q[2] * uvx - q[0] * uvz;
    double cz = q[0] * uvy - q[1] * uvx;
    double rx = (v[0] + q[3] * uvx) + cx;
    double ry = (v[1] + q[3] * uvy) + cy;
    double rz = (v[2] + q[3] * uvz) + cz;
    o[0] = rx; o[1] = ry; o[2] = rz;
}
static void quat_to_R(const double q[4], double R[9]) { /* Eigen toRotationMatrix */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* ------------------------------------------------------------------ MTK math */
static void cos_sinc_sqrt(double x2, double* c, double* s) { /* mtkmath.hpp:142-174 */
    const double taylor_0_bound = DBL_EPSILON;
    const double taylor_2_bound = sqrt(taylor_0_bound);
    const double taylor_n_bound = sqrt(taylor_2_bound);
    if (x2 >= taylor_n_bound) {
        double x = sqrt(x2);
        *c = cos(x);
        *s = sin(x) / x;
        return;
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
    *c = cosi;
    *s = sinc;
}
/* MTK::exp (mtkmath.hpp:249-256): result = sinc(scale*|v|)*scale*v, returns cos(scale*|v|) */
static double mtk_exp3(double res[3], const double v[3], double scale) {
    double norm2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double c, s;
    cos_sinc_sqrt(scale * scale * norm2, &c, &s);
    double mult = s * scale;
    res[0] = mult * v[0]; res[1] = mult * v[1]; res[2] = mult * v[2];
    return c;
}
void orc_so3_exp(const double v[3], double scale, double q[4]) { /* SOn.hpp:284-288 */
    q[3] = mtk_exp3(q, v, scale / 2);
}
void orc_so3_log(const double q[4], double v[3]) {
    /* SOn.hpp:293-297 -> MTK::log(res, w, vec, scale=2, plus_minus_periodicity=true),
       mtkmath.hpp:268-288 */
    double nv = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (nv < TOL) nv = TOL;
    double s = 2.0 / nv * atan(nv / q[3]);
    v[0] = s * q[0]; v[1] = s * q[1]; v[2] = s * q[2];
}
void orc_A_matrix(const double v[3], double A[9]) { /* mtkmath.hpp:235-247 */
    double squaredNorm = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    double norm = sqrt(squaredNorm);
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (norm < TOL) {
        memcpy(A, I, sizeof(I));
        return;
    }
    double H[9], HH[9];
    hat3(v, H);
    mat3_mul(H, H, HH);
    double c1 = (1 - cos(norm)) / squaredNorm;
    double c2 = (1 - sin(norm) / norm) / squaredNorm;
    for (int i = 0; i < 9; i++) A[i] = (I[i] + c1 * H[i]) + c2 * HH[i];
}

/* SO3 boxplus: q <- q * exp(v) (SOn.hpp:233-236) */
static void so3_boxplus(double q[4], const double v[3]) {
    double d[4], o[4];
    orc_so3_exp(v, 1.0, d);
    orc_quat_mul(q, d, o);
    memcpy(q, o, sizeof(o));
}
/* SO3 boxminus: log(other^-1 * q) (SOn.hpp:237-239) */
static void so3_boxminus(const double q[4], const double other[4], double res[3]) {
    double oc[4], r[4];
    quat_conj(other, oc);
    orc_quat_mul(oc, q, r);
    orc_so3_log(r, res);
}

/* ------------------------------------------------------------------ S2 (typ 1, |g| = 98090/10000) */
static const double S2_LEN = 98090.0 / 10000.0; /* use-ikfom.hpp:8, S2.hpp:104 */

void orc_S2_Bx(const double vec[3], double res[6]) { /* S2.hpp:215-231, 3x2 row-major */
    const double length = S2_LEN;
    if (vec[0] + length > TOL) {
        res[0] = -vec[1];
        res[1] = -vec[2];
        res[2] = length - vec[1] * vec[1] / (length + vec[0]);
        res[3] = -vec[2] * vec[1] / (length + vec[0]);
        res[4] = -vec[2] * vec[1] / (length + vec[0]);
        res[5] = length - vec[2] * vec[2] / (length + vec[0]);
        for (int i = 0; i < 6; i++) res[i] /= length;
    } else {
        memset(res, 0, 6 * sizeof(double));
        res[1 * 2 + 1] = -1;
        res[2 * 2 + 0] = 1;
    }
}
void orc_S2_Nx_yy(const double vec[3], double res[6]) { /* S2.hpp:259-264: 1/l/l * Bx^T * hat(vec) */
    double Bx[6], H[9];
    orc_S2_Bx(vec, Bx);
    hat3(vec, H);
    const double f = 1 / S2_LEN / S2_LEN;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (f * Bx[k * 2 + i]) * H[k * 3 + j];
            res[i * 3 + j] = s;
        }
}
void orc_S2_Mx(const double vec[3], const double delta[2], double res[6]) { /* S2.hpp:266-280 */
    double Bx[6], H[9];
    orc_S2_Bx(vec, Bx);
    hat3(vec, H);
    double nd = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    if (nd < TOL) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += (-H[i * 3 + k]) * Bx[k * 2 + j];
                res[i * 2 + j] = s;
            }
        return;
    }
    /* Bu = Bx*delta; exp_delta = exp(Bu, scalar(1/2)) -- integer division: scale == 0, so
       exp_delta is the identity rotation (S2.hpp:277 quirk, preserved). */
    double Bu[3];
    for (int i = 0; i < 3; i++) Bu[i] = Bx[i * 2] * delta[0] + Bx[i * 2 + 1] * delta[1];
    double A[9], AT[9], T1[9];
    orc_A_matrix(Bu, A);
    mat3_T(A, AT);
    double negH[9];
    for (int i = 0; i < 9; i++) negH[i] = -H[i]; /* -I * hat(vec) */
    mat3_mul(negH, AT, T1);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += T1[i * 3 + k] * Bx[k * 2 + j];
            res[i * 2 + j] = s;
        }
}
void orc_S2_boxplus(double vec[3], const double delta[2]) { /* S2.hpp:136-142 */
    double Bx[6], Bu[3], q[4], R[9];
    orc_S2_Bx(vec, Bx);
    for (int i = 0; i < 3; i++) Bu[i] = Bx[i * 2] * delta[0] + Bx[i * 2 + 1] * delta[1];
    q[3] = mtk_exp3(q, Bu, 1.0 / 2);
    quat_to_R(q, R);
    mat3_vec(R, vec, vec);
}
void orc_S2_boxminus(const double vec[3], const double other[3], double res[2]) { /* S2.hpp:144-167 */
    double H[9], t[3];
    hat3(vec, H);
    mat3_vec(H, other, t);
    double v_sin = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double v_cos = vec[0] * other[0] + vec[1] * other[1] + vec[2] * other[2];
    double theta = atan2(v_sin, v_cos);
    if (v_sin < TOL) {
        if (fabs(theta) > TOL) {
            res[0] = 3.1415926;
            res[1] = 0;
        } else {
            res[0] = 0;
            res[1] = 0;
        }
        return;
    }
    double Bx[6], Ho[9], u[3];
    orc_S2_Bx(other, Bx);
    hat3(other, Ho);
    mat3_vec(Ho, vec, u);
    double f = theta / v_sin;
    for (int i = 0; i < 2; i++) res[i] = (f * Bx[0 * 2 + i]) * u[0] + (f * Bx[1 * 2 + i]) * u[1] + (f * Bx[2 * 2 + i]) * u[2];
}

/* ------------------------------------------------------------------ state_ikfom */
/* flat offsets */
enum { X_POS = 0, X_ROT = 3, X_OFFR = 7, X_OFFT = 11, X_VEL = 14, X_BG = 17, X_BA = 20, X_GRAV = 23 };

void orc_state_boxplus(double x[ORC_NSTATE], const double dx[NDOF]) {
    for (int i = 0; i < 3; i++) x[X_POS + i] += dx[0 + i];
    so3_boxplus(x + X_ROT, dx + 3);
    so3_boxplus(x + X_OFFR, dx + 6);
    for (int i = 0; i < 3; i++) x[X_OFFT + i] += dx[9 + i];
    for (int i = 0; i < 3; i++) x[X_VEL + i] += dx[12 + i];
    for (int i = 0; i < 3; i++) x[X_BG + i] += dx[15 + i];
    for (int i = 0; i < 3; i++) x[X_BA + i] += dx[18 + i];
    orc_S2_boxplus(x + X_GRAV, dx + 21);
}
void orc_state_boxminus(const double x[ORC_NSTATE], const double y[ORC_NSTATE], double dx[NDOF]) {
    for (int i = 0; i < 3; i++) dx[0 + i] = x[X_POS + i] - y[X_POS + i];
    so3_boxminus(x + X_ROT, y + X_ROT, dx + 3);
    so3_boxminus(x + X_OFFR, y + X_OFFR, dx + 6);
    for (int i = 0; i < 3; i++) dx[9 + i] = x[X_OFFT + i] - y[X_OFFT + i];
    for (int i = 0; i < 3; i++) dx[12 + i] = x[X_VEL + i] - y[X_VEL + i];
    for (int i = 0; i < 3; i++) dx[15 + i] = x[X_BG + i] - y[X_BG + i];
    for (int i = 0; i < 3; i++) dx[18 + i] = x[X_BA + i] - y[X_BA + i];
    orc_S2_boxminus(x + X_GRAV, y + X_GRAV, dx + 21);
}

/* ------------------------------------------------------------------ inverse via partial-pivot LU */
int orc_inverse(const double* A, int n, double* Ainv) {
    /* Eigen inverse() for n > 4 = PartialPivLU().solve(Identity) [restated]: Doolittle LU with
       row pivoting on max |a_ik| (first max wins), unit-lower forward substitution, upper
       back-substitution, column by column. */
    double* LU = (double*)malloc(sizeof(double) * (size_t)n * n);
    int* perm = (int*)malloc(sizeof(int) * n);
    memcpy(LU, A, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++) perm[i] = i;
    int rc = 0;
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = fabs(LU[k * n + k]);
        for (int i = k + 1; i < n; i++) {
            double a = fabs(LU[i * n + k]);
            if (a > best) { best = a; p = i; }
        }
        if (best == 0.0) rc = 1; /* singular: carry on like Eigen does (produces inf/nan) */
        if (p != k) {
            for (int j = 0; j < n; j++) {
                double t = LU[k * n + j];
                LU[k * n + j] = LU[p * n + j];
                LU[p * n + j] = t;
            }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        double piv = LU[k * n + k];
        for (int i = k + 1; i < n; i++) LU[i * n + k] /= piv;
        for (int i = k + 1; i < n; i++) {
            double l = LU[i * n + k];
            for (int j = k + 1; j < n; j++) LU[i * n + j] -= l * LU[k * n + j];
        }
    }
    double* col = (double*)malloc(sizeof(double) * n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) col[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; i++) {
            double s = col[i];
            for (int j = 0; j < i; j++) s -= LU[i * n + j] * col[j];
            col[i] = s;
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = col[i];
            for (int j = i + 1; j < n; j++) s -= LU[i * n + j] * col[j];
            col[i] = s / LU[i * n + i];
        }
        for (int i = 0; i < n; i++) Ainv[i * n + c] = col[i];
    }
    free(col); free(perm); free(LU);
    return rc;
}

/* ------------------------------------------------------------------ ColPivHouseholderQR 5x3 fp32 */
/* Restated from Eigen's ColPivHouseholderQR::computeInPlace / _solve_impl and
   MatrixBase::makeHouseholder / applyHouseholderOnTheLeft (Eigen 3.3.x) [recalled-upstream].
   fp32 throughout; no FMA (the reference is built for baseline x86-64, CMakeLists.txt:8,14).

   SUMMATION ORDER.  Eigen's fp32 reductions are not sequential loops: which adds happen in which order depends on
   the vectorisation Eigen compiles in, and the reference ships no test that pins it.  Every reduction of the fit goes
   through the four helpers below, switched by orc_set_eigen_order():

     ORC_ORDER_SEQ     0  plain ascending loops (what a reader of the algorithm would write; round 1's only model)
     ORC_ORDER_SSE     1  Eigen 3.3.x as the reference builds it: x86-64, SSE2 on, EIGEN_UNALIGNED_VECTORIZE = 1 (default)
                          - fixed-size 5-vector (col(k).norm(), redux LinearVectorizedTraversal + CompleteUnrolling):
                            predux(packet 0..3) + e4, predux<Packet4f> = (a0 + a2) + (a1 + a3)  (movehl + add, shuffle + add_ss)
                          - dynamic-size vectors (tail.squaredNorm(), the lazy-product / inner-product dots of
                            applyHouseholderOnTheLeft, the down-date's tail norm; redux LinearVectorizedTraversal +
                            NoUnrolling with alignedStart = 0 because the cwise expression has no direct access):
                            size 4 -> one packet, predux as above; size < 4 -> ascending scalar loop
                          - fixed-size 3-vector (normvec.norm(): VectorizedSize = 0 -> redux_novec_unroller, a binary
                            tree): e0 + (e1 + e2)
     ORC_ORDER_PAIRWISE 2 the same structure with a pairwise predux (a0 + a1) + (a2 + a3) (NEON vpadd; SSE3 hadd builds)
     ORC_ORDER_NOVEC   3  Eigen with EIGEN_DONT_VECTORIZE: fixed sizes go through redux_novec_unroller's binary tree
                          (5: (e0 + e1) + (e2 + (e3 + e4)); 3: e0 + (e1 + e2)), dynamic sizes are ascending loops

   The default is ORC_ORDER_SSE (the reference's own build: Ubuntu + ROS, Eigen >= 3.3.4, -O3 without -march).
   The product's device code (fast_lio_amd/csrc/flh_device.hpp: esti_plane<ORD>) implements the same four orders and
   is compared bit for bit against whichever is selected; tests/test_eigen_order.py + DESIGN.md quantify how many
   plane fits / point_selected_surf flags / how much pose change the choice is worth.  oracle/ref/ holds the recipe
   that dumps the real Eigen's bits on a box that has Eigen, and tells which order (if any) it matches. */
static int g_eigen_order = ORC_ORDER_SSE;
void orc_set_eigen_order(int order) { g_eigen_order = (order >= 0 && order <= 3) ? order : ORC_ORDER_SSE; }
int orc_get_eigen_order(void) { return g_eigen_order; }

/* sum of a packet's four lanes */
static inline float ord_sum4(int ord, float a0, float a1, float a2, float a3) {
    if (ord == ORC_ORDER_SSE) return (a0 + a2) + (a1 + a3);
    if (ord == ORC_ORDER_PAIRWISE) return (a0 + a1) + (a2 + a3);
    return ((a0 + a1) + a2) + a3;
}
/* reduction of a DYNAMIC-size vector of n <= 4 addends (redux_impl<..., LinearVectorizedTraversal, NoUnrolling>) */
static inline float ord_sum_dyn(int ord, const float* a, int n) {
    if (n == 4 && (ord == ORC_ORDER_SSE || ord == ORC_ORDER_PAIRWISE)) return ord_sum4(ord, a[0], a[1], a[2], a[3]);
    float s = a[0];
    for (int i = 1; i < n; i++) s = s + a[i];
    return s;
}
/* reduction of a FIXED-size 5-vector (col(k).squaredNorm()) */
static inline float ord_sum_fixed5(int ord, const float a[5]) {
    if (ord == ORC_ORDER_SSE || ord == ORC_ORDER_PAIRWISE) return ord_sum4(ord, a[0], a[1], a[2], a[3]) + a[4];
    if (ord == ORC_ORDER_NOVEC) return (a[0] + a[1]) + (a[2] + (a[3] + a[4]));
    return (((a[0] + a[1]) + a[2]) + a[3]) + a[4];
}
/* reduction of a FIXED-size 3-vector (normvec.squaredNorm()) */
static inline float ord_sum_fixed3(int ord, float a0, float a1, float a2) {
    if (ord == ORC_ORDER_SEQ) return (a0 + a1) + a2;
    return a0 + (a1 + a2);
}

void orc_qr_solve_5x3(const float Ain[15], const float bin[5], float x[3]) {
    const int ord = g_eigen_order;
    enum { ROWS = 5, COLS = 3, SIZE = 3 };
    float qr[ROWS][COLS];
    for (int i = 0; i < ROWS; i++)
        for (int j = 0; j < COLS; j++) qr[i][j] = Ain[i * 3 + j];
    float hCoeffs[SIZE];
    int transp[SIZE];
    float normsUpdated[COLS], normsDirect[COLS];
    for (int k = 0; k < COLS; k++) { /* m_qr.col(k).norm(): fixed size 5 */
        float sq[ROWS];
        for (int i = 0; i < ROWS; i++) sq[i] = qr[i][k] * qr[i][k];
        normsDirect[k] = sqrtf(ord_sum_fixed5(ord, sq));
        normsUpdated[k] = normsDirect[k];
    }
    float maxn = normsUpdated[0];
    for (int k = 1; k < COLS; k++)
        if (normsUpdated[k] > maxn) maxn = normsUpdated[k];
    float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)ROWS;
    const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
    int nonzero_pivots = SIZE;

    for (int k = 0; k < SIZE; k++) {
        int big = k;
        float bigv = normsUpdated[k];
        for (int j = k + 1; j < COLS; j++)
            if (normsUpdated[j] > bigv) { bigv = normsUpdated[j]; big = j; }
        float biggest_col_sq_norm = bigv * bigv;
        if (nonzero_pivots == SIZE && biggest_col_sq_norm < threshold_helper * (float)(ROWS - k))
            nonzero_pivots = k;
        transp[k] = big;
        if (k != big) {
            for (int i = 0; i < ROWS; i++) {
                float t = qr[i][k]; qr[i][k] = qr[i][big]; qr[i][big] = t;
            }
            float t = normsUpdated[k]; normsUpdated[k] = normsUpdated[big]; normsUpdated[big] = t;
            t = normsDirect[k]; normsDirect[k] = normsDirect[big]; normsDirect[big] = t;
        }
        /* makeHouseholderInPlace on qr[k..4][k] */
        float tailSqNorm; /* tail.squaredNorm(): dynamic size ROWS - k - 1 */
        {
            float sq[ROWS];
            for (int i = k + 1; i < ROWS; i++) sq[i - k - 1] = qr[i][k] * qr[i][k];
            tailSqNorm = ord_sum_dyn(ord, sq, ROWS - k - 1);
        }
        float c0 = qr[k][k];
        float tau, beta;
        if (tailSqNorm <= FLT_MIN) {
            tau = 0.f;
            beta = c0;
            for (int i = k + 1; i < ROWS; i++) qr[i][k] = 0.f;
        } else {
            beta = sqrtf(c0 * c0 + tailSqNorm);
            if (c0 >= 0.f) beta = -beta;
            float den = c0 - beta;
            for (int i = k + 1; i < ROWS; i++) qr[i][k] = qr[i][k] / den;
            tau = (beta - c0) / beta;
        }
        hCoeffs[k] = tau;
        qr[k][k] = beta;
        /* apply H_k to the trailing block rows k..4, cols k+1..2 */
        if (tau != 0.f) {
            for (int j = k + 1; j < COLS; j++) {
                float pr[ROWS]; /* tmp = essential.adjoint() * bottom: lazy product, one dynamic-size dot per column */
                for (int i = k + 1; i < ROWS; i++) pr[i - k - 1] = qr[i][k] * qr[i][j];
                float tmp = ord_sum_dyn(ord, pr, ROWS - k - 1);
                tmp = tmp + qr[k][j];
                qr[k][j] = qr[k][j] - tau * tmp;
                for (int i = k + 1; i < ROWS; i++) qr[i][j] = qr[i][j] - (tau * qr[i][k]) * tmp;
            }
        }
        /* LAPACK-style column-norm down-date (LAWN 176) */
        for (int j = k + 1; j < COLS; j++) {
            if (normsUpdated[j] != 0.f) {
                float temp = fabsf(qr[k][j]) / normsUpdated[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float r = normsUpdated[j] / normsDirect[j];
                float temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) { /* m_qr.col(j).tail(rows - k - 1).norm(): dynamic size */
                    float sq[ROWS];
                    for (int i = k + 1; i < ROWS; i++) sq[i - k - 1] = qr[i][j] * qr[i][j];
                    float s = ord_sum_dyn(ord, sq, ROWS - k - 1);
                    normsDirect[j] = sqrtf(s);
                    normsUpdated[j] = normsDirect[j];
                } else {
                    normsUpdated[j] = normsUpdated[j] * sqrtf(temp);
                }
            }
        }
    }
    int perm[COLS] = {0, 1, 2};
    for (int k = 0; k < SIZE; k++) {
        int t = perm[k]; perm[k] = perm[transp[k]]; perm[transp[k]] = t;
    }
    /* solve */
    if (nonzero_pivots == 0) {
        x[0] = x[1] = x[2] = 0.f;
        return;
    }
    float c[ROWS];
    for (int i = 0; i < ROWS; i++) c[i] = bin[i];
    for (int k = 0; k < nonzero_pivots; k++) { /* Q^T c = H_{nz-1} ... H_1 H_0 c */
        float tau = hCoeffs[k];
        if (tau != 0.f) {
            float pr[ROWS]; /* inner product essential^T * c.bottomRows: dynamic size */
            for (int i = k + 1; i < ROWS; i++) pr[i - k - 1] = qr[i][k] * c[i];
            float tmp = ord_sum_dyn(ord, pr, ROWS - k - 1);
            tmp = tmp + c[k];
            c[k] = c[k] - tau * tmp;
            for (int i = k + 1; i < ROWS; i++) c[i] = c[i] - (tau * qr[i][k]) * tmp;
        }
    }
    /* upper-triangular solve, column-oriented (Eigen triangular_solve_vector, ColMajor) */
    for (int i = nonzero_pivots - 1; i >= 0; i--) {
        c[i] = c[i] / qr[i][i];
        for (int r = 0; r < i; r++) c[r] = c[r] - c[i] * qr[r][i];
    }
    for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = c[i];
    for (int i = nonzero_pivots; i < COLS; i++) x[perm[i]] = 0.f;
}

int orc_esti_plane(const float pts[15], float threshold, float pabcd[4]) {
    /* include/common_lib.h:225-257 */
    float b[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
    float nv[3];
    orc_qr_solve_5x3(pts, b, nv);
    float n = sqrtf(ord_sum_fixed3(g_eigen_order, nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2])); /* normvec.norm() */
    pabcd[0] = nv[0] / n;
    pabcd[1] = nv[1] / n;
    pabcd[2] = nv[2] / n;
    pabcd[3] = (float)(1.0 / (double)n); /* `1.0 / n` promotes to double, then narrows (:247) */
    for (int j = 0; j < 5; j++) {
        float v = ((pabcd[0] * pts[j * 3 + 0] + pabcd[1] * pts[j * 3 + 1]) + pabcd[2] * pts[j * 3 + 2]) + pabcd[3];
        if (fabsf(v) > threshold) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ prior: predict */
void orc_process_noise_cov(double Q[144]) { /* use-ikfom.hpp:35-43 */
    memset(Q, 0, 144 * sizeof(double));
    for (int i = 0; i < 3; i++) {
        Q[(0 + i) * 12 + 0 + i] = 0.0001;
        Q[(3 + i) * 12 + 3 + i] = 0.0001;
        Q[(6 + i) * 12 + 6 + i] = 0.00001;
        Q[(9 + i) * 12 + 9 + i] = 0.00001;
    }
}
void orc_init_P(double P[NDOF * NDOF]) { /* src/IMU_Processing.hpp:204-210 */
    memset(P, 0, sizeof(double) * NDOF * NDOF);
    for (int i = 0; i < NDOF; i++) P[i * NDOF + i] = 1.0;
    for (int i = 6; i < 9; i++) P[i * NDOF + i] = 0.00001;
    for (int i = 9; i < 12; i++) P[i * NDOF + i] = 0.00001;
    for (int i = 15; i < 18; i++) P[i * NDOF + i] = 0.0001;
    for (int i = 18; i < 21; i++) P[i * NDOF + i] = 0.001;
    P[21 * NDOF + 21] = P[22 * NDOF + 22] = 0.00001;
}

void orc_predict(double x[ORC_NSTATE], double P[NDOF * NDOF], double dt, const double Q[144],
                 const double acc[3], const double gyro[3]) {
    /* esekfom.hpp:279-383 (non-sparse path).  DIM layout (24): pos0 rot3 offR6 offT9 vel12 bg15 ba18
       grav21; DOF idx equals DIM idx for every sub-state (only grav is shorter). */
    enum { M = 24, N = NDOF, PN = 12 };
    double f_[M];
    static double f_x_[M][N], f_w_[M][PN], f_x_final[N][N], f_w_final[N][PN], F_x1[N][N];
#pragma omp critical(orc_predict_static)
    {
        memset(f_, 0, sizeof(f_));
        memset(f_x_, 0, sizeof(f_x_));
        memset(f_w_, 0, sizeof(f_w_));
        memset(f_x_final, 0, sizeof(f_x_final));
        memset(f_w_final, 0, sizeof(f_w_final));
        double Rm[9];
        quat_to_R(x + X_ROT, Rm);
        /* get_f, use-ikfom.hpp:47-59 */
        double omega[3], am[3], a_inertial[3];
        for (int i = 0; i < 3; i++) omega[i] = gyro[i] - x[X_BG + i];
        for (int i = 0; i < 3; i++) am[i] = acc[i] - x[X_BA + i];
        orc_quat_rot(x + X_ROT, am, a_inertial);
        for (int i = 0; i < 3; i++) {
            f_[i] = x[X_VEL + i];
            f_[i + 3] = omega[i];
            f_[i + 12] = a_inertial[i] + x[X_GRAV + i];
        }
        /* df_dx, use-ikfom.hpp:61-77 */
        double Hacc[9], RH[9], gm[6], zero2[2] = {0, 0};
        hat3(am, Hacc);
        mat3_mul(Rm, Hacc, RH);
        orc_S2_Mx(x + X_GRAV, zero2, gm);
        for (int i = 0; i < 3; i++) {
            f_x_[i][12 + i] = 1.0;
            f_x_[3 + i][15 + i] = -1.0;
            for (int j = 0; j < 3; j++) {
                f_x_[12 + i][3 + j] = -RH[i * 3 + j];
                f_x_[12 + i][18 + j] = -Rm[i * 3 + j];
            }
            for (int j = 0; j < 2; j++) f_x_[12 + i][21 + j] = gm[i * 2 + j];
        }
        /* df_dw, use-ikfom.hpp:80-88 */
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) f_w_[12 + i][3 + j] = -Rm[i * 3 + j];
            f_w_[3 + i][0 + i] = -1.0;
            f_w_[15 + i][6 + i] = 1.0;
            f_w_[18 + i][9 + i] = 1.0;
        }
        double x_before[ORC_NSTATE];
        memcpy(x_before, x, sizeof(x_before));
        /* x_.oplus(f_, dt): vect += dt*f; SO3: q * exp(f, dt) ; S2: rotate by exp(f, dt/2) (zero) */
        for (int i = 0; i < 3; i++) x[X_POS + i] += dt * f_[0 + i];
        {
            double d[4], o[4];
            orc_so3_exp(f_ + 3, dt, d);
            orc_quat_mul(x + X_ROT, d, o);
            memcpy(x + X_ROT, o, sizeof(o));
            orc_so3_exp(f_ + 6, dt, d);
            orc_quat_mul(x + X_OFFR, d, o);
            memcpy(x + X_OFFR, o, sizeof(o));
        }
        for (int i = 0; i < 3; i++) x[X_OFFT + i] += dt * f_[9 + i];
        for (int i = 0; i < 3; i++) x[X_VEL + i] += dt * f_[12 + i];
        for (int i = 0; i < 3; i++) x[X_BG + i] += dt * f_[15 + i];
        for (int i = 0; i < 3; i++) x[X_BA + i] += dt * f_[18 + i];
        {
            double q[4], Rq[9];
            q[3] = mtk_exp3(q, f_ + 21, dt / 2);
            quat_to_R(q, Rq);
            mat3_vec(Rq, x + X_GRAV, x + X_GRAV);
        }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) F_x1[i][j] = (i == j) ? 1.0 : 0.0;
        /* vect states: pos0 offT9 vel12 bg15 ba18 */
        static const int vidx[5] = {0, 9, 12, 15, 18};
        for (int v = 0; v < 5; v++)
            for (int j = 0; j < 3; j++) {
                for (int i = 0; i < N; i++) f_x_final[vidx[v] + j][i] = f_x_[vidx[v] + j][i];
                for (int i = 0; i < PN; i++) f_w_final[vidx[v] + j][i] = f_w_[vidx[v] + j][i];
            }
        /* SO3 states idx 3, 6: F_x1 block = exp(seg, scalar(1/2)=0) = I (quirk, esekfom.hpp:312);
           rows = A_matrix(seg) * f_x_ rows */
        static const int sidx[2] = {3, 6};
        for (int s = 0; s < 2; s++) {
            int idx = sidx[s];
            double seg[3], A[9];
            for (int i = 0; i < 3; i++) seg[i] = -1 * f_[idx + i] * dt;
            orc_A_matrix(seg, A);
            for (int i = 0; i < N; i++)
                for (int r = 0; r < 3; r++)
                    f_x_final[idx + r][i] = A[r * 3] * f_x_[idx][i] + A[r * 3 + 1] * f_x_[idx + 1][i] + A[r * 3 + 2] * f_x_[idx + 2][i];
            for (int i = 0; i < PN; i++)
                for (int r = 0; r < 3; r++)
                    f_w_final[idx + r][i] = A[r * 3] * f_w_[idx][i] + A[r * 3 + 1] * f_w_[idx + 1][i] + A[r * 3 + 2] * f_w_[idx + 2][i];
        }
        /* S2 state idx 21 (dim 21) */
        {
            int idx = 21;
            double seg[3];
            for (int i = 0; i < 3; i++) seg[i] = f_[idx + i] * dt;
            double Nx[6], Mx[6];
            orc_S2_Nx_yy(x + X_GRAV, Nx);
            orc_S2_Mx(x_before + X_GRAV, zero2, Mx);
            /* res = exp(seg, 0) -> identity rotation */
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    double s2 = 0;
                    for (int k = 0; k < 3; k++) s2 += Nx[i * 3 + k] * Mx[k * 2 + j];
                    F_x1[idx + i][idx + j] = s2;
                }
            double Hb[9], A[9], AT[9], T[9], rt[6];
            hat3(x_before + X_GRAV, Hb);
            orc_A_matrix(seg, A);
            mat3_T(A, AT);
            mat3_mul(Hb, AT, T);
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 3; j++) {
                    double s2 = 0;
                    for (int k = 0; k < 3; k++) s2 += (-Nx[i * 3 + k]) * T[k * 3 + j];
                    rt[i * 3 + j] = s2;
                }
            for (int i = 0; i < N; i++)
                for (int r = 0; r < 2; r++)
                    f_x_final[idx + r][i] = rt[r * 3] * f_x_[idx][i] + rt[r * 3 + 1] * f_x_[idx + 1][i] + rt[r * 3 + 2] * f_x_[idx + 2][i];
            for (int i = 0; i < PN; i++)
                for (int r = 0; r < 2; r++)
                    f_w_final[idx + r][i] = rt[r * 3] * f_w_[idx][i] + rt[r * 3 + 1] * f_w_[idx + 1][i] + rt[r * 3 + 2] * f_w_[idx + 2][i];
        }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) F_x1[i][j] += f_x_final[i][j] * dt;
        /* P = F P F^T + (dt fw) Q (dt fw)^T */
        static double T1[N][N], T2[N][PN];
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int k = 0; k < N; k++) s += F_x1[i][k] * P[k * N + j];
                T1[i][j] = s;
            }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < PN; j++) {
                double s = 0;
                for (int k = 0; k < PN; k++) s += (dt * f_w_final[i][k]) * Q[k * PN + j];
                T2[i][j] = s;
            }
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) {
                double s = 0;
                for (int k = 0; k < N; k++) s += T1[i][k] * F_x1[j][k];
                double w = 0;
                for (int k = 0; k < PN; k++) w += T2[i][k] * (dt * f_w_final[j][k]);
                P[i * N + j] = s + w;
            }
    }
}
