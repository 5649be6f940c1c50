// flh_scanprep.hip -- SURVEY.md 8(f) row 2: the scan's voxel-grid down-sampling on the device.
//
// pcl::VoxelGrid<PointType>::applyFilter as FAST-LIO2 uses it (downSizeFilterSurf.filter, src/laserMapping.cpp:904-905,
// leaf = filter_size_surf_min narrowed to float at :813): one float centroid per occupied leaf, output in ascending
// voxel-index order.  The per-voxel float sum runs in ascending input index (a stable radix sort keeps that order),
// the order the oracle pins (oracle_path.c: orc_voxel_grid) -- PCL's own order is implementation-defined.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "flh_kernels.hpp"

namespace flh {

typedef unsigned long long u64;
static inline int cdiv3(long long a, long long b) { return (int)((a + b - 1) / b); }

__global__ void __launch_bounds__(256) k_vg_keys(const float4* __restrict__ raw, uint32_t n, float inv, float mbx, float mby,
                                                 float mbz, int mul1, int mul2, u64* __restrict__ keys,
                                                 uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = raw[i];
    // ijk = static_cast<int>(std::floor(x * inverse_leaf_size) - static_cast<float>(min_b))
    const int i0 = (int)(floorf(p.x * inv) - mbx), i1 = (int)(floorf(p.y * inv) - mby), i2 = (int)(floorf(p.z * inv) - mbz);
    const int idx = i0 + i1 * mul1 + i2 * mul2;
    keys[i] = (u64)(uint32_t)idx;
    vals[i] = i;
}

__global__ void __launch_bounds__(256) k_vg_heads(const u64* __restrict__ keys_sorted, uint32_t n, uint32_t* __restrict__ flags) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    flags[j] = (j == 0 || keys_sorted[j] != keys_sorted[j - 1]) ? 1u : 0u;
}

// one thread per occupied voxel: running float sum over the voxel's points in sorted (= input) order, then sum / n
__global__ void __launch_bounds__(256) k_vg_reduce(const float4* __restrict__ raw, const u64* __restrict__ keys_sorted,
                                                   const uint32_t* __restrict__ vals_sorted, const uint32_t* __restrict__ flags,
                                                   const uint32_t* __restrict__ incl, uint32_t n, float4* __restrict__ out) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n || !flags[j]) return;
    const u64 key = keys_sorted[j];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    uint32_t e = j;
    for (; e < n && keys_sorted[e] == key; ++e) {
        const float4 p = raw[FLH_IDX(501, vals_sorted[e], n)];
        sx = sx + p.x; sy = sy + p.y; sz = sz + p.z;
    }
    const float cnt = (float)(e - j);
    const uint32_t o = (uint32_t)FLH_IDX(502, incl[j] - 1, n);
    out[o] = make_float4(sx / cnt, sy / cnt, sz / cnt, 0.f);
}

// ------------------------------------------------------------------------------------------------
// SURVEY.md 8(f) row 3 -- ImuProcess::UndistortPcl, per-point half (src/IMU_Processing.hpp:307-349): every point is
// carried from its own sampling time to the scan-end frame.  raw.w = the point's time offset in ms (PointType::curvature).
// pose rows: 22 doubles each = {offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]} (msg/Pose6D.msg).
// Segment of a point = the last k <= n_pose-2 with offset_time[k] < t (what the reference's back-to-front sweep over the
// time-sorted cloud amounts to; no ordering of the offset_times is assumed); a point no segment claims stays as it is.
// The EARLIEST point of the cloud (lowest index among equal times) is the exception the reference's loop makes (:345, `if
// (it_pcl == begin) break` leaves without stepping past it): it is carried by EVERY segment k = n_pose-2 .. 0 whose
// offset_time is below its time, one after the other on its moved float coordinates.  k_undistort leaves each block's
// (time, index) minimum behind; k_undistort_first finds the cloud's and redoes that one point (flh_config.undistort_first_point
// = 0 skips it: every point once).  Exp() = so3_math.h:36-58.
// ------------------------------------------------------------------------------------------------
// the inner loop's body (:326-343) for a point at time t with segment k; x, y, z are the point's float members
__device__ __forceinline__ void und_apply(const StateDev& s_end, const double* __restrict__ poses, int k, double t, float& x, float& y,
                                          float& z) {
    const double* head = poses + 22 * k;
    const double* tail = head + 22;
    const double dt = t - head[0];
    const double* gyr = tail + 4;
    const double* acc = tail + 1;
    const double* vel = head + 7;
    const double* pos = head + 10;
    const double* Rh = head + 13;
    double E[9] = {1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0};
    const double nw = sqrt((gyr[0] * gyr[0] + gyr[1] * gyr[1]) + gyr[2] * gyr[2]);
    if (nw > 0.0000001) {
        const double a0 = gyr[0] / nw, a1 = gyr[1] / nw, a2 = gyr[2] / nw;
        const double Kx[9] = {0.0, -a2, a1, a2, 0.0, -a0, -a1, a0, 0.0};
        const double ang = nw * dt, sn = sin(ang), c1 = 1.0 - cos(ang);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double kk = 0.0;
#pragma unroll
                for (int m = 0; m < 3; ++m) kk = kk + (c1 * Kx[3 * r + m]) * Kx[3 * m + c];
                E[3 * r + c] = (E[3 * r + c] + sn * Kx[3 * r + c]) + kk;
            }
    }
    double Ri[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double a = 0.0;
#pragma unroll
            for (int m = 0; m < 3; ++m) a = a + Rh[3 * r + m] * E[3 * m + c];
            Ri[3 * r + c] = a;
        }
    double Tei[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) Tei[d] = ((pos[d] + vel[d] * dt) + ((0.5 * acc[d]) * dt) * dt) - s_end.pos[d];
    double q1x, q1y, q1z;
    quat_rot(s_end.offR, (double)x, (double)y, (double)z, q1x, q1y, q1z);
    q1x = q1x + s_end.offT[0]; q1y = q1y + s_end.offT[1]; q1z = q1z + s_end.offT[2];
    const double q2x = ((Ri[0] * q1x + Ri[1] * q1y) + Ri[2] * q1z) + Tei[0];
    const double q2y = ((Ri[3] * q1x + Ri[4] * q1y) + Ri[5] * q1z) + Tei[1];
    const double q2z = ((Ri[6] * q1x + Ri[7] * q1y) + Ri[8] * q1z) + Tei[2];
    const double rotc[4] = {-s_end.rot[0], -s_end.rot[1], -s_end.rot[2], s_end.rot[3]};
    const double offRc[4] = {-s_end.offR[0], -s_end.offR[1], -s_end.offR[2], s_end.offR[3]};
    double q3x, q3y, q3z, q4x, q4y, q4z;
    quat_rot(rotc, q2x, q2y, q2z, q3x, q3y, q3z);
    q3x = q3x - s_end.offT[0]; q3y = q3y - s_end.offT[1]; q3z = q3z - s_end.offT[2];
    quat_rot(offRc, q3x, q3y, q3z, q4x, q4y, q4z);
    x = (float)q4x; y = (float)q4y; z = (float)q4z;
}
// (time, index) as one ordered 64-bit key: smaller time first, then smaller index; a NaN time sorts last
__device__ __forceinline__ u64 time_key(float t, uint32_t i) {
    const uint32_t u = __float_as_uint(t);
    const uint32_t o = (t != t) ? 0xFFFFFFFFu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    return ((u64)o << 32) | (u64)i;
}
__global__ void __launch_bounds__(256) k_undistort(StateDev s_end, const double* __restrict__ poses, int n_pose,
                                                   const float4* __restrict__ raw, uint32_t n, float4* __restrict__ out,
                                                   u64* __restrict__ block_min) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    u64 key = ~0ull;
    if (i < n) {
        const float4 p = raw[i];
        float4 o = p;
        const double t = (double)p.w / (double)1000;
        int k = -1;
        for (int j = n_pose - 2; j >= 0; --j)
            if (t > poses[22 * j]) { k = j; break; }
        if (k >= 0) und_apply(s_end, poses, k, t, o.x, o.y, o.z);
        out[i] = o;
        key = time_key(p.w, i);
    }
    if (block_min) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(key, o, 64);
            key = other < key ? other : key;
        }
        __shared__ u64 wm[4];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 m = wm[0];
            for (int w = 1; w < 4; ++w) m = wm[w] < m ? wm[w] : m;
            block_min[blockIdx.x] = m;
        }
    }
}
// one block: the cloud's earliest point from the blocks' minima, then that point once more from its raw coordinates through
// every segment that is older than it (descending k)
__global__ void __launch_bounds__(256) k_undistort_first(StateDev s_end, const double* __restrict__ poses, int n_pose,
                                                         const float4* __restrict__ raw, uint32_t n, float4* __restrict__ out,
                                                         const u64* __restrict__ block_min, uint32_t nblk) {
    u64 key = ~0ull;
    for (uint32_t b = threadIdx.x; b < nblk; b += 256) {
        const u64 k = block_min[b];
        key = k < key ? k : key;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u64 other = __shfl_xor(key, o, 64);
        key = other < key ? other : key;
    }
    __shared__ u64 wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x != 0) return;
    u64 m = wm[0];
    for (int w = 1; w < 4; ++w) m = wm[w] < m ? wm[w] : m;
    if (m == ~0ull) return;
    const uint32_t i = (uint32_t)m;
    if (i >= n) return;
    const float4 p = raw[i];
    float4 o = p;
    const double t = (double)p.w / (double)1000;
    for (int j = n_pose - 2; j >= 0; --j)
        if (t > poses[22 * j]) und_apply(s_end, poses, j, t, o.x, o.y, o.z);
    out[i] = o;
}

// ------------------------------------------------------------------------------------------------
// SURVEY.md 8(f) row 4 -- publish_frame_world's loop (src/laserMapping.cpp:478-530): RGBpointBodyToWorld (:200-211) over a
// whole cloud (feats_undistort when dense_pub_en, else feats_down_body): the same fp64 transform as the hot path's first
// step, narrowed to float.  in.w is carried through (intensity / original index).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cloud_body_to_world(StateDev s, const float4* __restrict__ in, uint32_t n,
                                                             float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = in[i];
    float4 o;
    body_to_world(s, p.x, p.y, p.z, o.x, o.y, o.z);
    o.w = p.w;
    out[i] = o;
}
hipError_t launch_cloud_body_to_world(const StateDev& s, const float4* in, uint32_t n, float4* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_cloud_body_to_world, dim3(cdiv3(n, 256)), dim3(256), 0, st, s, in, n, out);
    return hipGetLastError();
}

uint32_t undistort_blocks(uint32_t n) { return (uint32_t)cdiv3(n, 256); }
hipError_t launch_undistort(const StateDev& s_end, const double* poses, int n_pose, const float4* raw, uint32_t n, float4* out,
                            unsigned long long* block_min /* undistort_blocks(n) words, or nullptr: every point once */, hipStream_t st) {
    if (n == 0) return hipSuccess;
    const uint32_t nblk = undistort_blocks(n);
    hipLaunchKernelGGL(k_undistort, dim3(nblk), dim3(256), 0, st, s_end, poses, n_pose, raw, n, out, block_min);
    if (block_min)
        hipLaunchKernelGGL(k_undistort_first, dim3(1), dim3(256), 0, st, s_end, poses, n_pose, raw, n, out, (const u64*)block_min, nblk);
    return hipGetLastError();
}

hipError_t launch_vg_keys(const float4* raw, uint32_t n, float inv, const int min_b[3], int mul1, int mul2, u64* keys,
                          uint32_t* vals, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vg_keys, dim3(cdiv3(n, 256)), dim3(256), 0, st, raw, n, inv, (float)min_b[0], (float)min_b[1],
                       (float)min_b[2], mul1, mul2, keys, vals);
    return hipGetLastError();
}
hipError_t sort_vg_pairs(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, const uint32_t* vin, uint32_t* vout,
                         uint32_t n, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)n, 0, 32, st);
}
hipError_t launch_vg_heads(const u64* keys_sorted, uint32_t n, uint32_t* flags, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vg_heads, dim3(cdiv3(n, 256)), dim3(256), 0, st, keys_sorted, n, flags);
    return hipGetLastError();
}
hipError_t launch_vg_reduce(const float4* raw, const u64* keys_sorted, const uint32_t* vals_sorted, const uint32_t* flags,
                            const uint32_t* incl, uint32_t n, float4* out, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_vg_reduce, dim3(cdiv3(n, 256)), dim3(256), 0, st, raw, keys_sorted, vals_sorted, flags, incl, n, out);
    return hipGetLastError();
}

#ifdef FLH_BOUNDS
void bounds_read_scanprep(unsigned long long out[5]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(BoundsRec)); }
#endif

}  // namespace flh
