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

struct GridParams {
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
template <int ORD>
__device__ __forceinline__ float ord_sum4(float a0, float a1, float a2, float a3) {
    if (ORD == 1) return (a0 + a2) + (a1 + a3);
    if (ORD == 2) return (a0 + a1) + (a2 + a3);
    return ((a0 + a1) + a2) + a3;
}
// dynamic-size reduction of the n = 4 - K addends a[0..n) (K = the Householder step: sizes 4, 3, 2)
template <int ORD, int K>
__device__ __forceinline__ float ord_sum_dyn(const float a[4]) {
    if (K == 0) return (ORD == 1 || ORD == 2) ? ord_sum4<ORD>(a[0], a[1], a[2], a[3]) : ((a[0] + a[1]) + a[2]) + a[3];
    if (K == 1) return (a[0] + a[1]) + a[2];
    return a[0] + a[1];
}
template <int ORD>
__device__ __forceinline__ float ord_sum_fixed5(float a0, float a1, float a2, float a3, float a4) {
    if (ORD == 1 || ORD == 2) return ord_sum4<ORD>(a0, a1, a2, a3) + a4;
    if (ORD == 3) return (a0 + a1) + (a2 + (a3 + a4));
    return (((a0 + a1) + a2) + a3) + a4;
}
template <int ORD>
__device__ __forceinline__ float ord_sum_fixed3(float a0, float a1, float a2) {
    if (ORD == 0) return (a0 + a1) + a2;
    return a0 + (a1 + a2);
}

// One Householder step K of the factorisation (column pivot, reflector, trailing update, norm down-date).
template <int ORD, int K>
__device__ __forceinline__ void qr_step(float (&qr)[5][3], float (&hC)[3], int (&tr)[3], float (&nU)[3], float (&nD)[3],
                                        int& nz, float threshold_helper, float downdate_thr) {
    constexpr float kMin = 1.17549435e-38f;  // numeric_limits<float>::min()
    constexpr int k = K;
    int big = k;
    float bigv = nU[k];
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
        if (nU[j] > bigv) { bigv = nU[j]; big = j; }
    const float bsq = bigv * bigv;
    if (nz == 3 && bsq < threshold_helper * (float)(5 - k)) nz = k;
    tr[k] = big;
#pragma unroll
    for (int j = k + 1; j < 3; ++j)
        if (big == j) {
#pragma unroll
            for (int i = 0; i < 5; ++i) { float t = qr[i][k]; qr[i][k] = qr[i][j]; qr[i][j] = t; }
            float t = nU[k]; nU[k] = nU[j]; nU[j] = t;
            t = nD[k]; nD[k] = nD[j]; nD[j] = t;
        }
    float sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = k + 1; i < 5; ++i) sq[i - k - 1] = qr[i][k] * qr[i][k];
    const float tail = ord_sum_dyn<ORD, K>(sq);  // tail.squaredNorm()
    const float c0 = qr[k][k];
    float tau, beta;
    if (tail <= kMin) {
        tau = 0.f;
        beta = c0;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) qr[i][k] = 0.f;
    } else {
        beta = sqrt_rn(c0 * c0 + tail);
        if (c0 >= 0.f) beta = -beta;
        const float den = c0 - beta;
#pragma unroll
        for (int i = k + 1; i < 5; ++i) qr[i][k] = div_rn(qr[i][k], den);
        tau = div_rn(beta - c0, beta);
    }
    hC[k] = tau;
    qr[k][k] = beta;
    if (tau != 0.f) {
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            float pr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = k + 1; i < 5; ++i) pr[i - k - 1] = qr[i][k] * qr[i][j];
            float tmp = ord_sum_dyn<ORD, K>(pr);  // essential^T * bottom.col(j)
            tmp = tmp + qr[k][j];
            qr[k][j] = qr[k][j] - tau * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) qr[i][j] = qr[i][j] - (tau * qr[i][k]) * tmp;
        }
    }
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
        if (nU[j] != 0.f) {
            float temp = div_rn(fabsf(qr[k][j]), nU[j]);
            temp = (1.f + temp) * (1.f - temp);
            temp = temp < 0.f ? 0.f : temp;
            const float r = div_rn(nU[j], nD[j]);
            const float temp2 = temp * (r * r);
            if (temp2 <= downdate_thr) {
                float s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = k + 1; i < 5; ++i) s2[i - k - 1] = qr[i][j] * qr[i][j];
                nD[j] = sqrt_rn(ord_sum_dyn<ORD, K>(s2));  // col(j).tail(rows - k - 1).norm()
                nU[j] = nD[j];
            } else {
                nU[j] = nU[j] * sqrt_rn(temp);
            }
        }
    }
}
// Q^T c, reflector K (HouseholderSequence::applyThisOnTheLeft, one inner product per reflector)
template <int ORD, int K>
__device__ __forceinline__ void qt_step(const float (&qr)[5][3], const float (&hC)[3], int nz, float (&c)[5]) {
    constexpr int k = K;
    if (k < nz) {
        const float tau = hC[k];
        if (tau != 0.f) {
            float pr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = k + 1; i < 5; ++i) pr[i - k - 1] = qr[i][k] * c[i];
            float tmp = ord_sum_dyn<ORD, K>(pr);
            tmp = tmp + c[k];
            c[k] = c[k] - tau * tmp;
#pragma unroll
            for (int i = k + 1; i < 5; ++i) c[i] = c[i] - (tau * qr[i][k]) * tmp;
        }
    }
}

template <int ORD>
__device__ __forceinline__ bool esti_plane(const float P[5][3], float threshold, float pabcd[4]) {
    constexpr float kEps = 1.1920929e-07f;       // NumTraits<float>::epsilon()
    float qr[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) qr[i][j] = P[i][j];
    float hC[3];
    int tr[3];
    float nU[3], nD[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // m_qr.col(k).norm(): fixed size 5
        nD[k] = sqrt_rn(ord_sum_fixed5<ORD>(qr[0][k] * qr[0][k], qr[1][k] * qr[1][k], qr[2][k] * qr[2][k], qr[3][k] * qr[3][k],
                                            qr[4][k] * qr[4][k]));
        nU[k] = nD[k];
    }
    float maxn = nU[0];
    if (nU[1] > maxn) maxn = nU[1];
    if (nU[2] > maxn) maxn = nU[2];
    const float th = maxn * kEps;
    const float threshold_helper = (th * th) / 5.0f;
    const float downdate_thr = sqrt_rn(kEps);
    int nz = 3;
    qr_step<ORD, 0>(qr, hC, tr, nU, nD, nz, threshold_helper, downdate_thr);
    qr_step<ORD, 1>(qr, hC, tr, nU, nD, nz, threshold_helper, downdate_thr);
    qr_step<ORD, 2>(qr, hC, tr, nU, nD, nz, threshold_helper, downdate_thr);
    // column permutation from the transpositions
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = k + 1; j < 3; ++j)
            if (tr[k] == j) { int t = perm[k]; perm[k] = perm[j]; perm[j] = t; }
    }
    float nv[3] = {0.f, 0.f, 0.f};
    if (nz != 0) {
        float c[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
        qt_step<ORD, 0>(qr, hC, nz, c);
        qt_step<ORD, 1>(qr, hC, nz, c);
        qt_step<ORD, 2>(qr, hC, nz, c);
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            if (i < nz) {
                c[i] = div_rn(c[i], qr[i][i]);
#pragma unroll
                for (int r = 0; r < i; ++r) c[r] = c[r] - c[i] * qr[r][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float v = (i < nz) ? c[i] : 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (perm[i] == t) nv[t] = v;
        }
    }
    const float n = sqrt_rn(ord_sum_fixed3<ORD>(nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2]));  // normvec.norm()
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
    const uint32_t* st = g.starts + (size_t)rank * kBrickStride + cell_local(cx, cy, cz);
    const uint32_t a = st[0], b = st[1];
    return make_uint2(a, b - a);
}

}  // namespace flh
