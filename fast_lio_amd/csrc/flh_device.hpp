// flh_device.hpp -- device-side math of the hot path (gfx950 / CDNA4, wave64).
//
// Everything here must reproduce the reference's arithmetic bit-for-bit in fp32 (plane fit, residual
// gate) and operation-for-operation in fp64 (body->world transform), because point_selected_surf is
// graded bit-exact.  The translation unit is compiled with -ffp-contract=off: the reference is built
// for baseline x86-64 (CMakeLists.txt:8,14), which never fuses a*b+c.
// Citations are file:line under the reference tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flh {

struct StateDev {  // the four members of state_ikfom h_share_model reads (src/laserMapping.cpp:657)
    double rot[4];   // xyzw
    double pos[3];
    double offR[4];  // xyzw
    double offT[3];
};

// ---------------------------------------------------------------------------------------------
// Developer instrumentation, compiled out of the product (tools/variant.py --define FLH_BOUNDS; profiles/r04_fault_hunt/):
// every computed index into a device buffer goes through FLH_IDX(site, index, capacity).  A violation is counted, the first one
// recorded (site, index, capacity, workgroup) in a per-translation-unit device record that flh_debug_bounds() reads back, and the
// index is clamped so that the run goes on (a faulting run would tell less than a named site does).
// ---------------------------------------------------------------------------------------------
#ifdef FLH_BOUNDS
struct BoundsRec { unsigned long long count, site, index, cap, block; };
static __device__ BoundsRec g_bounds;  // one per translation unit
__device__ __forceinline__ unsigned long long flh_idx_(unsigned site, unsigned long long idx, unsigned long long cap) {
    if (idx < cap) return idx;
    if (atomicAdd(&g_bounds.count, 1ull) == 0ull) { g_bounds.site = site; g_bounds.index = idx; g_bounds.cap = cap; g_bounds.block = blockIdx.x; }
    return cap ? cap - 1 : 0;
}
#define FLH_IDX(site, idx, cap) flh_idx_((site), (unsigned long long)(idx), (unsigned long long)(cap))
#else
#define FLH_IDX(site, idx, cap) (idx)
#endif

struct GridParams {
#ifdef FLH_BOUNDS
    unsigned long long pts_cap, rows_cap, ids_cap;  // capacities of pts (map_sorted), of the brick tables / cap_end / live, of map_orig / dead_id
#endif
    float ox, oy, oz;  // world coordinate of the corner of cell (0,0,0)
    float c, inv_c;    // cell edge, 1/edge
    int nx, ny, nz;    // grid extent in cells (each <= 4096)
    uint32_t hash_mask;
    int hash_shift;          // 32 - log2(hash size)
    const uint2* hash;       // (brick_key, brick_rank); empty = 0xFFFFFFFF
    const uint32_t* starts;  // [nbricks * kBrickStride] prefix table: cell i of brick b holds pts[starts[b*S+i] .. starts[b*S+i+1])
    const float4* pts;       // map points sorted by (brick, cell); .w = original map index (bits)
};

constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;

// The scan's staging key (flh_kernels.hip "Scan staging", flh_stage.hip): 32-bit Morton code of the BODY-frame coordinates, 0.5 m
// quantum, x and y 11 bits (+-512 m), z 10 bits (+-256 m); the low 10 bits of the three interleaved, the 11th bits of x and y on top.
__device__ __forceinline__ uint32_t spread3_10(uint32_t v) {  // 10 bits -> every third bit
    uint32_t x = v & 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}
__device__ __forceinline__ uint32_t scan_morton(float x, float y, float z, float inv_q) {
    const uint32_t ix = (uint32_t)fminf(fmaxf(x * inv_q + 1024.f, 0.f), 2047.f);
    const uint32_t iy = (uint32_t)fminf(fmaxf(y * inv_q + 1024.f, 0.f), 2047.f);
    const uint32_t iz = (uint32_t)fminf(fmaxf(z * inv_q + 512.f, 0.f), 1023.f);
    return spread3_10(ix) | (spread3_10(iy) << 1) | (spread3_10(iz) << 2) | ((ix >> 10) << 30) | ((iy >> 10) << 31);
}

// Eigen QuaternionBase::_transformVector: uv = q.vec x v; uv += uv; v + w*uv + q.vec x uv
__device__ __forceinline__ void quat_rot(const double q[4], double vx, double vy, double vz, double& rx,
                                         double& ry, double& rz) {
    double uvx = q[1] * vz - q[2] * vy;
    double uvy = q[2] * vx - q[0] * vz;
    double uvz = q[0] * vy - q[1] * vx;
    uvx += uvx; uvy += uvy; uvz += uvz;
    double cx = q[1] * uvz - q[2] * uvy;
    double cy = q[2] * uvx - q[0] * uvz;
    double cz = q[0] * uvy - q[1] * uvx;
    rx = (vx + q[3] * uvx) + cx;
    ry = (vy + q[3] * uvy) + cy;
    rz = (vz + q[3] * uvz) + cz;
}

// p_global = s.rot * (s.offset_R_L_I * p_body + s.offset_T_L_I) + s.pos, fp64, then narrowed to fp32
// (src/laserMapping.cpp:656-660).
__device__ __forceinline__ void body_to_world(const StateDev& s, float bx, float by, float bz, float& wx,
                                              float& wy, float& wz) {
    double t1x, t1y, t1z, t2x, t2y, t2z;
    quat_rot(s.offR, (double)bx, (double)by, (double)bz, t1x, t1y, t1z);
    t1x = t1x + s.offT[0]; t1y = t1y + s.offT[1]; t1z = t1z + s.offT[2];
    quat_rot(s.rot, t1x, t1y, t1z, t2x, t2y, t2z);
    wx = (float)(t2x + s.pos[0]);
    wy = (float)(t2y + s.pos[1]);
    wz = (float)(t2z + s.pos[2]);
}

// IEEE correctly-rounded fp32 sqrt / divide.  hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt
// makes the plain operators exact; HIP's __fsqrt_rn/__fdiv_rn are NOT used (they map to the native,
// approximate OCML entry points in this toolchain).  tests/ check these bit-for-bit on the GPU.
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

// squared L2 in fp32, no contraction: ((dx*dx + dy*dy) + dz*dz)
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// ---------------------------------------------------------------------------------------------
// esti_plane<float> (include/common_lib.h:225-257): A (5x3, absolute world coordinates) n = -1 solved
// with Eigen's ColPivHouseholderQR, restated step by step (column-pivoted Householder QR with
// LAPACK-style norm down-dating).  Fully unrolled so everything lives in registers.  Returns true when
// all five points lie within `threshold` of the fitted plane.
//
// ORD = the fp32 summation order of Eigen's reductions (flh_config.eigen_order; the same four models as
// oracle/oracle_math.c "SUMMATION ORDER", which spells out where each comes from):
//   0 SEQ       ascending loops
//   1 SSE       Eigen 3.3.x, x86-64 + SSE2 (the reference's build): a 4-packet sums as (a0 + a2) + (a1 + a3); the fixed
//               5-vector col(k).norm() = packet + e4; dynamic-size reductions (tail norms, reflector dots) use the packet
//               only at size 4; the fixed 3-vector normvec.norm() = e0 + (e1 + e2)
//   2 PAIRWISE  the same with (a0 + a1) + (a2 + a3)  (NEON / hadd)
//   3 NOVEC     EIGEN_DONT_VECTORIZE: fixed 5 = (e0 + e1) + (e2 + (e3 + e4)), fixed 3 = e0 + (e1 + e2), dynamic ascending
// ---------------------------------------------------------------------------------------------
// The factorisation is written once over the scalar type T: float for the reference-exact path, _Float16 for the fp16
// ablation of BASELINE configs[4] (flh_config.plane_fit_dtype = 1, never the default).
template <class T> struct fit_traits;
template <> struct fit_traits<float> {
    static constexpr float eps = 1.1920929e-07f;      // NumTraits<float>::epsilon()
    static constexpr float tiny = 1.17549435e-38f;    // numeric_limits<float>::min()
};
template <> struct fit_traits<_Float16> {
    static constexpr float eps = 9.765625e-04f;       // 2^-10
    static constexpr float tiny = 6.103515625e-05f;   // smallest normal half
};
// correctly rounded sqrt / divide in T (for half: through float, whose 24 bits make the double rounding innocuous)
__device__ __forceinline__ float t_sqrt(float x) { return sqrt_rn(x); }
__device__ __forceinline__ float t_div(float a, float b) { return div_rn(a, b); }
__device__ __forceinline__ _Float16 t_sqrt(_Float16 x) { return (_Float16)__builtin_sqrtf((float)x); }
__device__ __forceinline__ _Float16 t_div(_Float16 a, _Float16 b) { return (_Float16)((float)a / (float)b); }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ _Float16 t_abs(_Float16 x) { return x < (_Float16)0 ? -x : x; }

template <int ORD, class T>
__device__ __forceinline__ T ord_sum4(T a0, T a1, T a2, T a3) {
    if (ORD == 1) return (a0 + a2) + (a1 + a3);
    if (ORD == 2) return (a0 + a1) + (a2 + a3);
    return ((a0 + a1) + a2) + a3;
}
// dynamic-size reduction of the n = 4 - K addends a[0..n) (K = the Householder step: sizes 4, 3, 2)
template <int ORD, int K, class T>
__device__ __forceinline__ T ord_sum_dyn(const T a[4]) {
    if (K == 0) return (ORD == 1 || ORD == 2) ? ord_sum4<ORD, T>(a[0], a[1], a[2], a[3]) : ((a[0] + a[1]) + a[2]) + a[3];
    if (K == 1) return (a[0] + a[1]) + a[2];
    return a[0] + a[1];
}
template <int ORD, class T>
__device__ __forceinline__ T ord_sum_fixed5(T a0, T a1, T a2, T a3, T a4) {
    if (ORD == 1 || ORD == 2) return ord_sum4<ORD, T>(a0, a1, a2, a3) + a4;
    if (ORD == 3) return (a0 + a1) + (a2 + (a3 + a4));
    return (((a0 + a1) + a2) + a3) + a4;
}
template <int ORD, class T>
__device__ __forceinline__ T ord_sum_fixed3(T a0, T a1, T a2) {
    if (ORD == 0) return (a0 + a1) + a2;
    return a0 + (a1 + a2);
}

// One Householder step K of the factorisation (column pivot, reflector, trailing update, norm down-date).
template <int ORD, int K, class T>
__device__ __forceinline__ void qr_step(T (&qr)[5][3], T (&hC)[3], int (&tr)[3], T (&nU)[3], T (&nD)[3], int& nz,
                                        T threshold_helper, T downdate_thr) {
    const T kMin = (T)fit_traits<T>::tiny;
    const T zero = (T)0, one = (T)1;
    constexpr int k = K;
    int big = k;
    T bigv = nU[k];
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
        if (nU[j] > bigv) { bigv = nU[j]; big = j; }
    const T bsq = bigv * bigv;
    if (nz == 3 && bsq < threshold_helper * (T)(float)(5 - k)) nz = k;
    tr[k] = big;
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
        if (big == j) {
#pragma unroll
            for (int i = 0; i < 5; ++i) { T t = qr[i][k]; qr[i][k] = qr[i][j]; qr[i][j] = t; }
            T t = nU[k]; nU[k] = nU[j]; nU[j] = t;
            t = nD[k]; nD[k] = nD[j]; nD[j] = t;
        }
    T sq[4] = {zero, zero, zero, zero};
#pragma unroll
    for (int i = k + 1; i < 5; ++i) sq[i - k - 1] = qr[i][k] * qr[i][k];
    const T tail = ord_sum_dyn<ORD, K, T>(sq);  // tail.squaredNorm()
    const T c0 = qr[k][k];
    T tau, beta;
    if (tail <= kMin) {
        tau = zero;
        beta = c0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) qr[i][k] = zero;
    } else {
        beta = t_sqrt(c0 * c0 + tail);
        if (c0 >= zero) beta = -beta;
        const T den = c0 - beta;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) qr[i][k] = t_div(qr[i][k], den);
        tau = t_div(beta - c0, beta);
    }
    hC[k] = tau;
    qr[k][k] = beta;
    if (tau != zero) {
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            T pr[4] = {zero, zero, zero, zero};
#pragma unroll
            for (int i = k + 1; i < 5; ++i) pr[i - k - 1] = qr[i][k] * qr[i][j];
            T tmp = ord_sum_dyn<ORD, K, T>(pr);  // essential^T * bottom.col(j)
            tmp = tmp + qr[k][j];
            qr[k][j] = qr[k][j] - tau * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) qr[i][j] = qr[i][j] - (tau * qr[i][k]) * tmp;
        }
    }
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
        if (nU[j] != zero) {
            T temp = t_div(t_abs(qr[k][j]), nU[j]);
            temp = (one + temp) * (one - temp);
            temp = temp < zero ? zero : temp;
            const T r = t_div(nU[j], nD[j]);
            const T temp2 = temp * (r * r);
            if (temp2 <= downdate_thr) {
                T s2[4] = {zero, zero, zero, zero};
#pragma unroll
                for (int i = k + 1; i < 5; ++i) s2[i - k - 1] = qr[i][j] * qr[i][j];
                nD[j] = t_sqrt(ord_sum_dyn<ORD, K, T>(s2));  // col(j).tail(rows - k - 1).norm()
                nU[j] = nD[j];
            } else {
                nU[j] = nU[j] * t_sqrt(temp);
            }
        }
    }
}
// Q^T c, reflector K (HouseholderSequence::applyThisOnTheLeft, one inner product per reflector)
template <int ORD, int K, class T>
__device__ __forceinline__ void qt_step(const T (&qr)[5][3], const T (&hC)[3], int nz, T (&c)[5]) {
    constexpr int k = K;
    if (k < nz) {
        const T tau = hC[k];
        if (tau != (T)0) {
            T pr[4] = {(T)0, (T)0, (T)0, (T)0};
#pragma unroll
            for (int i = k + 1; i < 5; ++i) pr[i - k - 1] = qr[i][k] * c[i];
            T tmp = ord_sum_dyn<ORD, K, T>(pr);
            tmp = tmp + c[k];
            c[k] = c[k] - tau * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) c[i] = c[i] - (tau * qr[i][k]) * tmp;
        }
    }
}

// A.colPivHouseholderQr().solve(b), A = qr (5x3, destroyed), b = -1 (common_lib.h:229-241)
template <int ORD, class T>
__device__ __forceinline__ void qr_solve_5x3(T (&qr)[5][3], T (&nv)[3]) {
    const T kEps = (T)fit_traits<T>::eps;
    T hC[3];
    int tr[3];
    T nU[3], nD[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // m_qr.col(k).norm(): fixed size 5
        nD[k] = t_sqrt(ord_sum_fixed5<ORD, T>(qr[0][k] * qr[0][k], qr[1][k] * qr[1][k], qr[2][k] * qr[2][k], qr[3][k] * qr[3][k],
                                              qr[4][k] * qr[4][k]));
        nU[k] = nD[k];
    }
    T maxn = nU[0];
    if (nU[1] > maxn) maxn = nU[1];
    if (nU[2] > maxn) maxn = nU[2];
    const T th = maxn * kEps;
    const T threshold_helper = t_div(th * th, (T)5.0f);
    const T downdate_thr = t_sqrt(kEps);
    int nz = 3;
    qr_step<ORD, 0, T>(qr, hC, tr, nU, nD, nz, threshold_helper, downdate_thr);
    qr_step<ORD, 1, T>(qr, hC, tr, nU, nD, nz, threshold_helper, downdate_thr);
    qr_step<ORD, 2, T>(qr, hC, tr, nU, nD, nz, threshold_helper, downdate_thr);
    // column permutation from the transpositions
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (tr[k] == j) { int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
    }
    nv[0] = nv[1] = nv[2] = (T)0;
    if (nz != 0) {
        T c[5] = {(T)-1.f, (T)-1.f, (T)-1.f, (T)-1.f, (T)-1.f};
        qt_step<ORD, 0, T>(qr, hC, nz, c);
        qt_step<ORD, 1, T>(qr, hC, nz, c);
        qt_step<ORD, 2, T>(qr, hC, nz, c);
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (i < nz) {
                c[i] = t_div(c[i], qr[i][i]);
#pragma unroll
                for (int r = 0; r < i; ++r) c[r] = c[r] - c[i] * qr[r][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const T v = (i < nz) ? c[i] : (T)0;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (perm[i] == t) nv[t] = v;
        }
    }
}

template <int ORD>
__device__ __forceinline__ bool esti_plane(const float P[5][3], float threshold, float pabcd[4]) {
    float qr[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) qr[i][j] = P[i][j];
    float nv[3];
    qr_solve_5x3<ORD, float>(qr, nv);
    const float n = sqrt_rn(ord_sum_fixed3<ORD, float>(nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2]));  // normvec.norm()
    pabcd[0] = div_rn(nv[0], n);
    pabcd[1] = div_rn(nv[1], n);
    pabcd[2] = div_rn(nv[2], n);
    pabcd[3] = (float)(1.0 / (double)n);  // `1.0 / n` promotes to double, then narrows (common_lib.h:247)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float v = ((pabcd[0] * P[j][0] + pabcd[1] * P[j][1]) + pabcd[2] * P[j][2]) + pabcd[3];
        if (fabsf(v) > threshold) ok = false;
    }
    return ok;
}

// ABLATION (BASELINE configs[4], flh_config.plane_fit_dtype = 1): the same fit with the 5x3 system in fp16.  Half's 11 bits
// cannot hold absolute map coordinates, so the five points are first moved (in fp32) to an origin about one metre off their
// plane -- their centroid pushed along a rough normal (cross product of two edges) -- where the A n = -1 formulation is
// well conditioned; the factorisation and the solve then run in _Float16, and the plane is carried back to absolute
// coordinates in fp32.  NOT bit-exact with the reference and never the default; tests report its flag-mismatch rate.
template <int ORD>
__device__ __forceinline__ bool esti_plane_half(const float P[5][3], float threshold, float pabcd[4]) {
    float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) { cx += P[j][0]; cy += P[j][1]; cz += P[j][2]; }
    cx *= 0.2f; cy *= 0.2f; cz *= 0.2f;
    // rough normal: the largest of the cross products of edges from point 0
    float bx = 0.f, by = 0.f, bz = 0.f, bn = -1.f;
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const float ux = P[j][0] - P[0][0], uy = P[j][1] - P[0][1], uz = P[j][2] - P[0][2];
        const float vx = P[j + 1][0] - P[0][0], vy = P[j + 1][1] - P[0][1], vz = P[j + 1][2] - P[0][2];
        const float wx = uy * vz - uz * vy, wy = uz * vx - ux * vz, wz = ux * vy - uy * vx;
        const float wn = wx * wx + wy * wy + wz * wz;
        if (wn > bn) { bn = wn; bx = wx; by = wy; bz = wz; }
    }
    const float inv = bn > 1e-12f ? 1.0f / sqrtf(bn) : 0.f;
    const float ox = cx - bx * inv, oy = cy - by * inv, oz = cz - bz * inv;  // origin: 1 m off the points along the rough normal
    _Float16 qr[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        qr[i][0] = (_Float16)(P[i][0] - ox);
        qr[i][1] = (_Float16)(P[i][1] - oy);
        qr[i][2] = (_Float16)(P[i][2] - oz);
    }
    _Float16 nvh[3];
    qr_solve_5x3<ORD, _Float16>(qr, nvh);
    const float nx = (float)nvh[0], ny = (float)nvh[1], nz_ = (float)nvh[2];
    const float n = sqrtf((nx * nx + ny * ny) + nz_ * nz_);
    pabcd[0] = nx / n;
    pabcd[1] = ny / n;
    pabcd[2] = nz_ / n;
    const float dl = 1.0f / n;  // offset in the shifted frame: a (p - o) + dl = 0
    pabcd[3] = dl - ((pabcd[0] * ox + pabcd[1] * oy) + pabcd[2] * oz);
    bool ok = n > 0.f && n < 3.0e38f;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float v = ((pabcd[0] * (P[j][0] - ox) + pabcd[1] * (P[j][1] - oy)) + pabcd[2] * (P[j][2] - oz)) + dl;
        if (!(fabsf(v) <= threshold)) ok = false;
    }
    return ok;
}

// 32-bit brick key: 10 bits per axis of (cell >> 2)
__device__ __host__ __forceinline__ uint32_t brick_key(int cx, int cy, int cz) {
    return ((uint32_t)(cz >> 2) << 20) | ((uint32_t)(cy >> 2) << 10) | (uint32_t)(cx >> 2);
}
__device__ __host__ __forceinline__ uint32_t cell_local(int cx, int cy, int cz) {
    return (uint32_t)(((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3));
}
__device__ __forceinline__ uint32_t hash_slot(uint32_t key, int shift) { return (key * 2654435761u) >> shift; }

// world position -> integer cell coordinate (may be out of range); identical formula at build and query.
__device__ __forceinline__ void cell_of(const GridParams& g, float x, float y, float z, int& cx, int& cy,
                                        int& cz, float& fx, float& fy, float& fz) {
    fx = (x - g.ox) * g.inv_c;
    fy = (y - g.oy) * g.inv_c;
    fz = (z - g.oz) * g.inv_c;
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    cx = (int)flx; cy = (int)fly; cz = (int)flz;
    fx -= flx; fy -= fly; fz -= flz;
}

constexpr int kBrickStride = 80;  // 64 cells + end sentinel, padded so that every z-slab of 16 cells starts a 64-byte line

// A storage slot that holds no map point (slack behind a brick's points, a removed point, a relocated brick's old range):
// coordinates so large that every squared distance to it overflows to +inf, which no selection accepts; .w = -1.
constexpr float kTombCoord = 3.0e38f;
__device__ __forceinline__ float4 tombstone() { return make_float4(kTombCoord, kTombCoord, kTombCoord, __uint_as_float(0xFFFFFFFFu)); }
__device__ __forceinline__ bool is_tombstone(const float4& p) { return p.x == kTombCoord; }

// rank of a brick in the sorted order, or 0xFFFFFFFF if the brick holds no points
__device__ __forceinline__ uint32_t lookup_brick(const GridParams& g, uint32_t key) {
    const unsigned long long* __restrict__ hash64 = reinterpret_cast<const unsigned long long*>(g.hash);
    uint32_t slot = hash_slot(key, g.hash_shift);
    for (;;) {
        const unsigned long long e = hash64[slot];
        if ((uint32_t)e == key) return (uint32_t)(e >> 32);
        if ((uint32_t)e == kEmptyKey) return kEmptyKey;
        slot = (slot + 1) & g.hash_mask;
    }
}

// (start,count) of a cell, or count 0 when the cell / its brick is empty or out of range.
__device__ __forceinline__ uint2 lookup_cell(const GridParams& g, int cx, int cy, int cz) {
    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz)
        return make_uint2(0u, 0u);
    const uint32_t rank = lookup_brick(g, brick_key(cx, cy, cz));
    if (rank == kEmptyKey) return make_uint2(0u, 0u);
#ifdef FLH_BOUNDS
    (void)FLH_IDX(101, rank, g.rows_cap);
#endif
    const uint32_t* st = g.starts + (size_t)rank * kBrickStride + cell_local(cx, cy, cz);
    const uint32_t a = st[0], b = st[1];
#ifdef FLH_BOUNDS
    (void)FLH_IDX(102, (unsigned long long)b, g.pts_cap + 1);
    (void)FLH_IDX(103, (unsigned long long)a, (unsigned long long)b + 1);
#endif
    return make_uint2(a, b - a);
}

}  // namespace flh
