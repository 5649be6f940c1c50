// flh_mapinc.hip -- SURVEY.md 8(f) row 1: the incremental map around the hot path.
//
//   k_mi_classify     map_incremental's add/skip decision per scan point (src/laserMapping.cpp:427-474), fed
//                     from the device-resident neighbour cache the last search left behind
//   k_add_keys/...    ikdtree.Add_Points(points, downsample) (:470-471): per filter_size_map voxel the point
//                     nearest to the voxel centre survives [ikd-Tree semantics, recalled-upstream; the oracle
//                     (oracle_path.c: orc_map_add) states them]
//   k_delete_boxes    ikdtree.Delete_Point_Boxes (:275)
//
// The map index is rebuilt from the compacted point array after every change (O(M) per scan; an in-place merge
// of the sorted arrays is the obvious next step).  Built with -ffp-contract=off like the rest.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "flh_device.hpp"
#include "flh_kernels.hpp"

namespace flh {

typedef unsigned long long u64;

static inline int cdiv2(long long a, long long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------------------
// map_incremental: classes 0 = skip, 1 = PointToAdd (down-sampled insert), 2 = PointNoNeedDownsample
// ------------------------------------------------------------------------------------------------
// The search that filled the cache is radius-bounded: entries with d2 <= max_sqdist are the true nearest neighbours in
// order, entries beyond it are merely some map points (whatever the visited cells held).  The reference's search is
// unbounded.  The two still agree on every decision below: a neighbour that can veto the insert lies within
// sqrt(3)*fsm of the point (well inside the bound), and the one case that needs a neighbour outside the bound --
// points_near[0] of a point with NO map point inside it -- is served by k_far_nearest first.
__global__ void __launch_bounds__(256)
k_far_nearest(GridParams g, StateDev s_search, const float4* __restrict__ body, const uint8_t* __restrict__ nn_cnt,
              const float* __restrict__ nn_d2, float max_sqdist, int N, uint32_t hash_size, float4* __restrict__ nn_pts) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= N) return;
    if (nn_cnt[q] != 0 && nn_d2[q] <= max_sqdist) return;  // wave-uniform: the cached nearest is the true one
    const float4 b = body[q];
    float wx, wy, wz;
    body_to_world(s_search, b.x, b.y, b.z, wx, wy, wz);  // the world position the last search used
    const unsigned long long* __restrict__ hash64 = reinterpret_cast<const unsigned long long*>(g.hash);
    const float bw = 4.0f * g.c;
    auto box = [&](uint32_t key, float& lb2, float& ub2) {
        const float bx = (float)(key & 1023u), by = (float)((key >> 10) & 1023u), bz = (float)(key >> 20);
        const float lo[3] = {g.ox + bx * bw, g.oy + by * bw, g.oz + bz * bw};
        const float w[3] = {wx, wy, wz};
        lb2 = 0.f; ub2 = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float m = 1e-3f * g.c + 1e-5f * fabsf(w[d]);
            const float l = lo[d] - m, h = lo[d] + bw + m;
            const float dmin = fmaxf(0.f, fmaxf(l - w[d], w[d] - h));
            const float dmax = fmaxf(fabsf(w[d] - l), fabsf(w[d] - h));
            lb2 += dmin * dmin;
            ub2 += dmax * dmax;
        }
    };
    // pass 1: every brick holds at least one point, so min over bricks of the far-corner distance bounds the answer
    float best_ub = INFINITY;
    for (uint32_t slot = lane; slot < hash_size; slot += 64) {
        const unsigned long long e = hash64[slot];
        if ((uint32_t)e == kEmptyKey) continue;
        float lb2, ub2;
        box((uint32_t)e, lb2, ub2);
        best_ub = fminf(best_ub, ub2);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best_ub = fminf(best_ub, __shfl_xor(best_ub, o, 64));
    best_ub *= 1.0001f;
    // pass 2: exact distances in the bricks that can hold it; key = (d2 bits, original index) so ties go to the lower index
    u64 best = ~0ull;
    float4 best_p = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t slot = lane; slot < hash_size; slot += 64) {
        const unsigned long long e = hash64[slot];
        if ((uint32_t)e == kEmptyKey) continue;
        float lb2, ub2;
        box((uint32_t)e, lb2, ub2);
        if (lb2 * 0.9999f > best_ub) continue;
        const uint32_t* stt = g.starts + (size_t)(uint32_t)(e >> 32) * kBrickStride;
        const uint32_t i0 = stt[0], i1 = stt[64];
        for (uint32_t i = i0; i < i1; ++i) {
            const float4 p = g.pts[i];
            const float d = dist2(p.x, p.y, p.z, wx, wy, wz);
            const u64 k = ((u64)__float_as_uint(d) << 32) | (u64)__float_as_uint(p.w);
            if (k < best) { best = k; best_p = p; }
            best_ub = fminf(best_ub, d * 1.0001f);
        }
    }
    u64 gbest = best;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u64 other = __shfl_xor(gbest, o, 64);
        gbest = other < gbest ? other : gbest;
    }
    if (gbest != ~0ull && best == gbest) nn_pts[q] = best_p;  // unique: the key carries the index
}

// Outputs are in ORIGINAL scan order (the body buffer is Morton-ordered; .w carries the original index).
__global__ void __launch_bounds__(256)
k_mi_classify(StateDev s, const float4* __restrict__ body, const float4* __restrict__ nn_pts,
              const uint8_t* __restrict__ nn_cnt, const float* __restrict__ nn_d2, float max_sqdist, int N, uint32_t map_points,
              double fsm, int ekf_inited, float4* __restrict__ world_out, uint8_t* __restrict__ cls) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 b = body[i];
    const uint32_t o = __float_as_uint(b.w);
    float wx, wy, wz;
    body_to_world(s, b.x, b.y, b.z, wx, wy, wz);  // pointBodyToWorld with the POSTERIOR state (:436)
    world_out[o] = make_float4(wx, wy, wz, 0.f);
    const int cnt = nn_cnt[i];                              // found inside the bound, ascending
    const int true_cnt = map_points < 5u ? (int)map_points : 5;  // what the unbounded search returns
    uint8_t c = 1;  // Nearest_Points[i].empty() || !flg_EKF_inited -> PointToAdd (:463-466)
    if (true_cnt > 0 && ekf_inited) {
        // mid_point members are float: each is a double expression narrowed to float (:443-445)
        const float mx = (float)(floor((double)wx / fsm) * fsm + 0.5 * fsm);
        const float my = (float)(floor((double)wy / fsm) * fsm + 0.5 * fsm);
        const float mz = (float)(floor((double)wz / fsm) * fsm + 0.5 * fsm);
        const float dist = dist2(wx, wy, wz, mx, my, mz);  // calc_dist (:446)
        const float4 n0 = nn_pts[i];                       // points_near[0] (from the search, or k_far_nearest)
        if (fabs((double)(n0.x - mx)) > 0.5 * fsm && fabs((double)(n0.y - my)) > 0.5 * fsm &&
            fabs((double)(n0.z - mz)) > 0.5 * fsm) {       // :447
            c = 2;
        } else {
            bool need_add = true;
            if (true_cnt >= 5) {  // points_near.size() < NUM_MATCH_POINTS -> break (:454)
                for (int r = 0; r < cnt; ++r) {
                    if (!(nn_d2[(size_t)r * N + i] <= max_sqdist)) break;  // beyond the bound: not a vetted neighbour, and too far to veto
                    const float4 pn = nn_pts[(size_t)r * N + i];
                    if (dist2(pn.x, pn.y, pn.z, mx, my, mz) < dist) need_add = false;  // :455-459
                }
            }
            c = need_add ? 1 : 0;
        }
    }
    cls[o] = c;
}

// class 1 (down-sampled insert) then class 2 (plain insert), each in original scan order
__global__ void __launch_bounds__(256) k_cls_flags(const uint8_t* __restrict__ cls, int N, uint32_t* __restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint8_t c = cls[i];
    flags[i] = c == 1 ? 1u : 0u;
    flags[N + i] = c == 2 ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_cls_compact(const float4* __restrict__ world, const uint8_t* __restrict__ cls,
                                                     const uint32_t* __restrict__ incl, int N, float4* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint8_t c = cls[i];
    if (c == 1) out[incl[i] - 1] = world[i];
    else if (c == 2) out[incl[N + i] - 1] = world[i];
}

// exact AABB of a point array: ordered-uint encoding of floats + atomics
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void __launch_bounds__(256) k_aabb(const float4* __restrict__ pts, uint32_t M, uint32_t* __restrict__ out6) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
        const float4 p = pts[i];
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o, 64));
        }
    // one atomic per block and axis: same-address atomics serialise at ~11 ns each
    __shared__ float smn[4][3], smx[4][3];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { smn[w][d] = mn[d]; smx[w][d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const float a = fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d]));
        const float b = fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]));
        atomicMin(out6 + d, f2ord(a));
        atomicMax(out6 + 3 + d, f2ord(b));
    }
}

// ------------------------------------------------------------------------------------------------
// Add_Points with down-sampling
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void vox_of(float x, float y, float z, double ds, long long& kx, long long& ky, long long& kz) {
    kx = (long long)floor((double)x / ds);
    ky = (long long)floor((double)y / ds);
    kz = (long long)floor((double)z / ds);
}
__device__ __forceinline__ float dist_to_center(float x, float y, float z, long long kx, long long ky, long long kz, double ds) {
    const float mx = (float)((double)kx * ds + 0.5 * ds), my = (float)((double)ky * ds + 0.5 * ds),
                mz = (float)((double)kz * ds + 0.5 * ds);
    return dist2(x, y, z, mx, my, mz);
}
__device__ __forceinline__ u64 pack_vox(long long kx, long long ky, long long kz) {
    return (((u64)(kx + (1ll << 20)) & 0x1FFFFFull) << 42) | (((u64)(ky + (1ll << 20)) & 0x1FFFFFull) << 21) |
           ((u64)(kz + (1ll << 20)) & 0x1FFFFFull);
}

__global__ void __launch_bounds__(256) k_add_keys(const float4* __restrict__ add, uint32_t n, double ds,
                                                  u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 p = add[i];
    long long kx, ky, kz;
    vox_of(p.x, p.y, p.z, ds, kx, ky, kz);
    keys[i] = pack_vox(kx, ky, kz);
    vals[i] = i;
}

// One thread per run of new points sharing a voxel (the radix sort is stable, so a run lists them in input
// order).  Final state of the voxel = the single point nearest to its centre among {points already in the map}
// U {new points}; ties: a new point beats an existing one, a later new point beats an earlier one; a voxel whose
// single existing point stays nearest is left untouched.
__global__ void __launch_bounds__(256)
k_add_resolve(GridParams g, const float4* __restrict__ add, const u64* __restrict__ keys_sorted,
              const uint32_t* __restrict__ vals_sorted, uint32_t n, double ds, uint8_t* __restrict__ dead_old,
              uint8_t* __restrict__ alive_new) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const u64 key = keys_sorted[j];
    if (j > 0 && keys_sorted[j - 1] == key) return;  // not the head of its run
    const float4 p0 = add[vals_sorted[j]];
    long long kx, ky, kz;
    vox_of(p0.x, p0.y, p0.z, ds, kx, ky, kz);
    // best new point of the run
    uint32_t best_new = vals_sorted[j];
    float best_d = dist_to_center(p0.x, p0.y, p0.z, kx, ky, kz, ds);
    uint32_t e = j + 1;
    for (; e < n && keys_sorted[e] == key; ++e) {
        const uint32_t id = vals_sorted[e];
        const float4 p = add[id];
        const float d = dist_to_center(p.x, p.y, p.z, kx, ky, kz, ds);
        if (d <= best_d) { best_d = d; best_new = id; }  // later wins a tie
    }
    // existing points in the voxel: every search cell the voxel box overlaps, exact voxel test per point
    const float bx0 = (float)((double)kx * ds), by0 = (float)((double)ky * ds), bz0 = (float)((double)kz * ds);
    const float bx1 = (float)((double)(kx + 1) * ds), by1 = (float)((double)(ky + 1) * ds), bz1 = (float)((double)(kz + 1) * ds);
    int c0x, c0y, c0z, c1x, c1y, c1z;
    float f0, f1, f2;
    // the corners were rounded to float: widen by 4 ulp so the (monotone) cell map covers every point of the voxel
    cell_of(g, bx0 - (fabsf(bx0) * 5e-7f + 1e-30f), by0 - (fabsf(by0) * 5e-7f + 1e-30f), bz0 - (fabsf(bz0) * 5e-7f + 1e-30f),
            c0x, c0y, c0z, f0, f1, f2);
    cell_of(g, bx1 + (fabsf(bx1) * 5e-7f + 1e-30f), by1 + (fabsf(by1) * 5e-7f + 1e-30f), bz1 + (fabsf(bz1) * 5e-7f + 1e-30f),
            c1x, c1y, c1z, f0, f1, f2);
    int n_exist = 0;
    float best_ed = INFINITY;
    uint32_t best_e = 0xFFFFFFFFu;
    for (int z = c0z; z <= c1z; ++z)
        for (int y = c0y; y <= c1y; ++y)
            for (int x = c0x; x <= c1x; ++x) {
                const uint2 ce = lookup_cell(g, x, y, z);
                for (uint32_t i = ce.x; i < ce.x + ce.y; ++i) {
                    const float4 q = g.pts[i];
                    long long qx, qy, qz;
                    vox_of(q.x, q.y, q.z, ds, qx, qy, qz);
                    if (qx != kx || qy != ky || qz != kz) continue;
                    ++n_exist;
                    const float d = dist_to_center(q.x, q.y, q.z, kx, ky, kz, ds);
                    const uint32_t id = __float_as_uint(q.w);
                    if (d < best_ed || (d == best_ed && id < best_e)) { best_ed = d; best_e = id; }  // tie: lower index
                }
            }
    const bool new_wins = !(best_ed < best_d);  // an existing point displaces only when strictly nearer
    if (n_exist == 1 && !new_wins) return;      // the single existing point stays; every new point is dropped
    // otherwise the voxel is emptied except for the winner
    if (n_exist > 0) {
        for (int z = c0z; z <= c1z; ++z)
            for (int y = c0y; y <= c1y; ++y)
                for (int x = c0x; x <= c1x; ++x) {
                    const uint2 ce = lookup_cell(g, x, y, z);
                    for (uint32_t i = ce.x; i < ce.x + ce.y; ++i) {
                        const float4 q = g.pts[i];
                        long long qx, qy, qz;
                        vox_of(q.x, q.y, q.z, ds, qx, qy, qz);
                        if (qx != kx || qy != ky || qz != kz) continue;
                        const uint32_t id = __float_as_uint(q.w);
                        if (new_wins || id != best_e) dead_old[id] = 1;
                    }
                }
    }
    if (new_wins) alive_new[best_new] = 1;
}

__global__ void __launch_bounds__(256) k_delete_boxes(const float4* __restrict__ pts, uint32_t M,
                                                      const float* __restrict__ boxes, int nb,
                                                      uint8_t* __restrict__ dead) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float4 p = pts[i];
    bool del = false;
    for (int b = 0; b < nb; ++b) {
        const float* bx = boxes + 6 * b;
        if (p.x >= bx[0] && p.x < bx[3] && p.y >= bx[1] && p.y < bx[4] && p.z >= bx[2] && p.z < bx[5]) del = true;
    }
    if (del) dead[i] = 1;
}

// alive flags (1 - dead for old points, alive_new for new ones) -> 0/1 words for the prefix sum
__global__ void __launch_bounds__(256) k_alive_flags(const uint8_t* __restrict__ dead_old, uint32_t M,
                                                     const uint8_t* __restrict__ alive_new, uint32_t n,
                                                     uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) flags[i] = dead_old[i] ? 0u : 1u;
    else if (i < M + n) flags[i] = alive_new[i - M] ? 1u : 0u;
}
// survivors keep their relative order: old points first, then the new ones
__global__ void __launch_bounds__(256) k_compact(const float4* __restrict__ old_pts, uint32_t M,
                                                 const float4* __restrict__ new_pts, uint32_t n,
                                                 const uint32_t* __restrict__ flags, const uint32_t* __restrict__ incl,
                                                 float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M + n || !flags[i]) return;
    float4 p = i < M ? old_pts[i] : new_pts[i - M];
    p.w = 0.f;
    out[incl[i] - 1] = p;
}

// ------------------------------------------------------------------------------------------------
hipError_t launch_mi_classify(const GridParams& g, uint32_t hash_size, uint32_t map_points, const StateDev& s_search,
                              const StateDev& s_post, const float4* body, float4* nn_pts, const uint8_t* nn_cnt,
                              const float* nn_d2, float max_sqdist, int N, double fsm, int ekf_inited, float4* world_out,
                              uint8_t* cls, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    if (map_points > 0 && ekf_inited)
        hipLaunchKernelGGL(k_far_nearest, dim3(cdiv2(N, 4)), dim3(256), 0, st, g, s_search, body, nn_cnt, nn_d2, max_sqdist, N,
                           hash_size, nn_pts);
    hipLaunchKernelGGL(k_mi_classify, dim3(cdiv2(N, 256)), dim3(256), 0, st, s_post, body, nn_pts, nn_cnt, nn_d2, max_sqdist, N,
                       map_points, fsm, ekf_inited, world_out, cls);
    return hipGetLastError();
}
hipError_t launch_cls_flags(const uint8_t* cls, int N, uint32_t* flags, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cls_flags, dim3(cdiv2(N, 256)), dim3(256), 0, st, cls, N, flags);
    return hipGetLastError();
}
hipError_t launch_cls_compact(const float4* world, const uint8_t* cls, const uint32_t* incl, int N, float4* out,
                              hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cls_compact, dim3(cdiv2(N, 256)), dim3(256), 0, st, world, cls, incl, N, out);
    return hipGetLastError();
}
hipError_t launch_aabb(const float4* pts, uint32_t M, uint32_t* out6, hipStream_t st) {
    if (M == 0) return hipSuccess;
    int blocks = cdiv2(M, 256 * 8);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_aabb, dim3(blocks), dim3(256), 0, st, pts, M, out6);
    return hipGetLastError();
}
hipError_t launch_add_keys(const float4* add, uint32_t n, double ds, u64* keys, uint32_t* vals, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_keys, dim3(cdiv2(n, 256)), dim3(256), 0, st, add, n, ds, keys, vals);
    return hipGetLastError();
}
hipError_t sort_vox_pairs(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, const uint32_t* vin, uint32_t* vout,
                          uint32_t n, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)n, 0, 63, st);
}
hipError_t launch_add_resolve(const GridParams& g, const float4* add, const u64* ks, const uint32_t* vs, uint32_t n,
                              double ds, uint8_t* dead_old, uint8_t* alive_new, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_resolve, dim3(cdiv2(n, 256)), dim3(256), 0, st, g, add, ks, vs, n, ds, dead_old, alive_new);
    return hipGetLastError();
}
hipError_t launch_delete_boxes(const float4* pts, uint32_t M, const float* boxes, int nb, uint8_t* dead, hipStream_t st) {
    if (M == 0 || nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_delete_boxes, dim3(cdiv2(M, 256)), dim3(256), 0, st, pts, M, boxes, nb, dead);
    return hipGetLastError();
}
hipError_t launch_alive_flags(const uint8_t* dead_old, uint32_t M, const uint8_t* alive_new, uint32_t n, uint32_t* flags,
                              hipStream_t st) {
    if (M + n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_alive_flags, dim3(cdiv2((long long)M + n, 256)), dim3(256), 0, st, dead_old, M, alive_new, n, flags);
    return hipGetLastError();
}
hipError_t launch_compact(const float4* old_pts, uint32_t M, const float4* new_pts, uint32_t n, const uint32_t* flags,
                          const uint32_t* incl, float4* out, hipStream_t st) {
    if (M + n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compact, dim3(cdiv2((long long)M + n, 256)), dim3(256), 0, st, old_pts, M, new_pts, n, flags, incl,
                       out);
    return hipGetLastError();
}

}  // namespace flh
