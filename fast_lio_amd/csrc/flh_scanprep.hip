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
        const float4 p = raw[vals_sorted[e]];
        sx = sx + p.x; sy = sy + p.y; sz = sz + p.z;
    }
    const float cnt = (float)(e - j);
    const uint32_t o = incl[j] - 1;
    out[o] = make_float4(sx / cnt, sy / cnt, sz / cnt, 0.f);
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

}  // namespace flh
