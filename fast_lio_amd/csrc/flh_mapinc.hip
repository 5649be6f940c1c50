// flh_mapinc.hip -- SURVEY.md 8(f) row 1: the incremental map around the hot path.
//
//   k_mi_classify     map_incremental's add/skip decision per scan point (src/laserMapping.cpp:427-474), fed
//                     from the device-resident neighbour cache the last search left behind
//   k_far_search      points_near[0] of the few scan points with no map point inside the search bound (shell search over bricks),
//                     and their decision (k_mi_classify lists them)
//   k_cls_compact     the two lists (PointToAdd, PointNoNeedDownsample) in original scan order, their lengths to host and device;
//                     on the path without the host's wait also k_add_insert's step
//   k_add_insert / k_add_resolve   ikdtree.Add_Points(points, downsample) (:470-471): per filter_size_map voxel the point
//                     nearest to the voxel centre survives [ikd-Tree semantics, recalled-upstream; the oracle
//                     (oracle_path.c: orc_map_add) states them]; the new points are grouped by voxel in a hash table (no sort)
//   k_delete_boxes    ikdtree.Delete_Point_Boxes (:275)
//
//   k_ins_sort_small + k_brick_rewrite_heads (a scan's worth of points) / k_ins_prepare + library sort + k_brick_rewrite +
//   k_map_publish (larger changes)   the surviving new points enter the brick storage: only the bricks that receive points are
//                     rewritten (LDS counting sort, in place while they fit their slack, else relocated); the whole index is
//                     rebuilt only when something no longer fits (flh_api.cpp: apply_map_changes); the change's counters go to
//                     the host as granules (the last workgroup of k_brick_rewrite_heads, or k_map_publish)
// Built with -ffp-contract=off like the rest.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

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
// points_near[0] of a point with NO map point inside it -- is served by k_far_search.
// One wave per query.  Bricks are visited in cubic shells around the query's brick (shell r = the bricks at Chebyshev distance
// r): after shell r everything within the cube of (2r+1)^3 bricks is known, so the search ends as soon as the best distance
// found is smaller than the distance from the query to that cube's faces -- a few hundred directory probes when the nearest
// point is some tens of metres away.  Only when the shells grow past the directory itself (a query very far from everything)
// the remaining work is done by scanning the whole directory twice (bound, then exact).
// The neighbour cache as map_incremental reads it: coordinates (nn_pts) or -- what a one-launch searching pass leaves,
// flh_config.index_cache -- map indices, whose coordinates are read from the id-ordered array on the spot (round 5 gathered them
// into nn_pts with a kernel of its own first: 7.5 us and a launch per scan of the config-3 stream).  Row r of query i: at = r N + i.
struct NnSrc {
    float4* pts;
    uint32_t* idx;           // nullptr: pts holds the coordinates
    const float4* map_orig;
    uint32_t n_ids;
};
__device__ __forceinline__ float4 nn_get(const NnSrc& s, size_t at) {
    if (!s.idx) return s.pts[at];
    const uint32_t id = s.idx[at];
    if (id >= s.n_ids) return make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));  // an empty row (k_nn_gather's rule)
    float4 v = s.map_orig[FLH_IDX(232, id, s.n_ids)];
    v.w = __uint_as_float(id);
    return v;
}

// Rows FIRST..4 of query i, all requested before the first is looked at: one round trip for the indices, one for the coordinates
// (map_incremental's decision used to make them one after the other as it went down the rows -- up to ten dependent trips).  Rows
// the search did not fill hold whatever the buffer held: the caller looks at rows < nn_cnt only.
template <int FIRST>
__device__ __forceinline__ void nn_rows(const NnSrc& s, int i, int N, float4 (&near)[5]) {
    if (s.idx) {
        uint32_t id[5];
#pragma unroll
        for (int r = FIRST; r < 5; ++r) id[r] = s.idx[(size_t)r * N + i];
#pragma unroll
        for (int r = FIRST; r < 5; ++r) {
            const bool ok = id[r] < s.n_ids;  // else: an empty row (k_nn_gather's rule)
            float4 v = s.map_orig[FLH_IDX(232, ok ? id[r] : 0u, s.n_ids)];
            v.w = __uint_as_float(id[r]);
            near[r] = ok ? v : make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
        }
    } else {
#pragma unroll
        for (int r = FIRST; r < 5; ++r) near[r] = s.pts[(size_t)r * N + i];
    }
}

// the shell search of ONE query by one wave (see above); wx, wy, wz = the world position the last search used
// Returns the nearest point to every lane (.w = its id; 0xFFFFFFFF: the map holds no live point).
__device__ __forceinline__ float4 far_search(const GridParams& g, int q, int lane, float wx, float wy, float wz, uint32_t hash_size,
                                             const uint32_t* __restrict__ live, const NnSrc& nn) {
    const unsigned long long* __restrict__ hash64 = reinterpret_cast<const unsigned long long*>(g.hash);
    const float bw = 4.0f * g.c;
    const float w[3] = {wx, wy, wz};
    const float org[3] = {g.ox, g.oy, g.oz};
    float marg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) marg[d] = 1e-3f * g.c + 1e-5f * fabsf(w[d]);
    auto box = [&](uint32_t key, float& lb2, float& ub2) {
        const float bc[3] = {(float)(key & 1023u), (float)((key >> 10) & 1023u), (float)(key >> 20)};
        lb2 = 0.f; ub2 = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float l = org[d] + bc[d] * bw - marg[d], h = org[d] + bc[d] * bw + bw + marg[d];
            const float dmin = fmaxf(0.f, fmaxf(l - w[d], w[d] - h));
            const float dmax = fmaxf(fabsf(w[d] - l), fabsf(w[d] - h));
            lb2 += dmin * dmin;
            ub2 += dmax * dmax;
        }
    };
    // exact distances over a brick's points; key = (d2 bits, original index) so ties go to the lower index
    u64 best = ~0ull;
    float4 best_p = make_float4(0.f, 0.f, 0.f, 0.f);
    float best_ub = INFINITY;  // >= the squared distance of the nearest point, with a rounding margin
    auto scan_brick = [&](uint32_t rank) {
        const uint32_t* stt = g.starts + (size_t)FLH_IDX(201, rank, g.rows_cap) * kBrickStride;
        const uint32_t i0 = stt[0], i1 = stt[64];
        for (uint32_t i = i0; i < i1; ++i) {
            const float4 p = g.pts[FLH_IDX(202, i, g.pts_cap)];
            if (is_tombstone(p)) continue;
            const float d = dist2(p.x, p.y, p.z, wx, wy, wz);
            const u64 k = ((u64)__float_as_uint(d) << 32) | (u64)__float_as_uint(p.w);
            if (k < best) { best = k; best_p = p; }
            best_ub = fminf(best_ub, d * 1.0001f);
        }
    };
    // the query's brick (it may lie outside the grid) and the grid's extent in bricks
    int bq[3], nb[3] = {(g.nx + 3) >> 2, (g.ny + 3) >> 2, (g.nz + 3) >> 2};
    int r0 = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float f = floorf((w[d] - org[d]) / bw);
        bq[d] = (int)fminf(fmaxf(f, -1.0e6f), 1.0e6f);
        r0 = max(r0, max(-bq[d], bq[d] - (nb[d] - 1)));  // shells nearer than this hold no brick of the grid
    }
    bool finished = false;
    for (int r = r0;; ++r) {
        const long long side = 2ll * r + 1;
        if (side * side * side > (long long)hash_size) break;  // block-uniform: the directory itself is the shorter list
        const uint32_t face = (uint32_t)(side * side);
        const uint32_t total = r == 0 ? 1u : 2u * face + (uint32_t)(2 * r - 1) * (uint32_t)(8 * r);
        for (uint32_t t = lane; t < total; t += 64) {
            int dx, dy, dz;
            if (r == 0) { dx = dy = dz = 0; }
            else if (t < 2u * face) {
                const uint32_t rem = t < face ? t : t - face;
                dz = t < face ? -r : r;
                dy = (int)(rem / (uint32_t)side) - r;
                dx = (int)(rem % (uint32_t)side) - r;
            } else {
                const uint32_t t2 = t - 2u * face;
                dz = (int)(t2 / (uint32_t)(8 * r)) - (r - 1);
                const uint32_t u = t2 % (uint32_t)(8 * r), e = u / (uint32_t)(2 * r);
                const int k = (int)(u % (uint32_t)(2 * r));
                dx = e == 0 ? -r + k : (e == 1 ? r : (e == 2 ? r - k : -r));
                dy = e == 0 ? -r : (e == 1 ? -r + k : (e == 2 ? r : r - k));
            }
            const int x = bq[0] + dx, y = bq[1] + dy, z = bq[2] + dz;
            if ((unsigned)x >= (unsigned)nb[0] || (unsigned)y >= (unsigned)nb[1] || (unsigned)z >= (unsigned)nb[2]) continue;
            const uint32_t key = ((uint32_t)z << 20) | ((uint32_t)y << 10) | (uint32_t)x;
            const uint32_t rank = lookup_brick(g, key);
            if (rank == kEmptyKey || live[FLH_IDX(203, rank, g.rows_cap)] == 0u) continue;
            float lb2, ub2;
            box(key, lb2, ub2);
            if (lb2 * 0.9999f > best_ub) continue;
            scan_brick(rank);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) best_ub = fminf(best_ub, __shfl_xor(best_ub, o, 64));
        // every brick inside the cube [bq - r, bq + r]^3 has been seen: what lies outside is at least this far away
        float dout = INFINITY;
        bool all = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float lo = org[d] + (float)(bq[d] - r) * bw, hi = org[d] + (float)(bq[d] + r + 1) * bw;
            dout = fminf(dout, fminf(w[d] - lo, hi - w[d]) - marg[d]);
            all = all && bq[d] - r <= 0 && bq[d] + r >= nb[d] - 1;
        }
        dout = fmaxf(dout, 0.f);
        if (all || best_ub < dout * dout * 0.9999f) { finished = true; break; }
    }
    if (!finished) {
        // pass 1: a brick with a live point holds one within its far corner, so the min over such bricks bounds the answer
        for (uint32_t slot = lane; slot < hash_size; slot += 64) {
            const unsigned long long e = hash64[slot];
            if ((uint32_t)e == kEmptyKey || live[FLH_IDX(204, (uint32_t)(e >> 32), g.rows_cap)] == 0u) continue;
            float lb2, ub2;
            box((uint32_t)e, lb2, ub2);
            best_ub = fminf(best_ub, ub2 * 1.0001f);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) best_ub = fminf(best_ub, __shfl_xor(best_ub, o, 64));
        // pass 2: exact distances in the bricks that can hold it (a brick seen by a shell before is simply seen again)
        for (uint32_t slot = lane; slot < hash_size; slot += 64) {
            const unsigned long long e = hash64[slot];
            if ((uint32_t)e == kEmptyKey) continue;
            float lb2, ub2;
            box((uint32_t)e, lb2, ub2);
            if (lb2 * 0.9999f > best_ub) continue;
            scan_brick((uint32_t)(e >> 32));
        }
    }
    u64 gbest = best;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u64 other = __shfl_xor(gbest, o, 64);
        gbest = other < gbest ? other : gbest;
    }
    if (gbest != ~0ull && best == gbest) {  // unique: the key carries the index
        if (nn.idx) nn.idx[q] = __float_as_uint(best_p.w);
        else nn.pts[q] = best_p;
    }
    if (gbest == ~0ull) return make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
    const int wl = __ffsll((long long)__ballot(best == gbest)) - 1;  // the lane that holds it
    return make_float4(__shfl(best_p.x, wl, 64), __shfl(best_p.y, wl, 64), __shfl(best_p.z, wl, 64), __shfl(best_p.w, wl, 64));
}

// map_incremental's decision for ONE scan point (src/laserMapping.cpp:441-466) given points_near = near[0 .. cnt): 0 = skip,
// 1 = PointToAdd, 2 = PointNoNeedDownsample.  (wx, wy, wz): the point at the POSTERIOR state (:436); (sx, sy, sz): where the last
// search saw it -- pointSearchSqDis is measured from there, with the search's own expression (and bits).
__device__ __forceinline__ uint8_t mi_decide(int cnt, int true_cnt, int ekf_inited, double fsm, float max_sqdist,
                                             float wx, float wy, float wz, float sx, float sy, float sz, const float4 (&near)[5]) {
    uint8_t c = 1;  // Nearest_Points[i].empty() || !flg_EKF_inited -> PointToAdd (:463-466)
    if (true_cnt > 0 && ekf_inited) {
        // mid_point members are float: each is a double expression narrowed to float (:443-445)
        const float mx = (float)(floor((double)wx / fsm) * fsm + 0.5 * fsm);
        const float my = (float)(floor((double)wy / fsm) * fsm + 0.5 * fsm);
        const float mz = (float)(floor((double)wz / fsm) * fsm + 0.5 * fsm);
        const float dist = dist2(wx, wy, wz, mx, my, mz);  // calc_dist (:446)
        if (fabs((double)(near[0].x - mx)) > 0.5 * fsm && fabs((double)(near[0].y - my)) > 0.5 * fsm &&
            fabs((double)(near[0].z - mz)) > 0.5 * fsm) {       // :447
            c = 2;
        } else {
            bool need_add = true;
            if (true_cnt >= 5) {  // points_near.size() < NUM_MATCH_POINTS -> break (:454)
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    if (r >= cnt) break;
                    const float4 pn = near[r];
                    // pointSearchSqDis[r], the expression (and bits) the search compared (k_fill_d2 writes the same on demand)
                    const float d2r = (__float_as_uint(pn.w) == 0xFFFFFFFFu) ? INFINITY : dist2(sx, sy, sz, pn.x, pn.y, pn.z);
                    if (!(d2r <= max_sqdist)) break;  // beyond the bound: not a vetted neighbour, and too far to veto
                    if (dist2(pn.x, pn.y, pn.z, mx, my, mz) < dist) need_add = false;  // :455-459
                }
            }
            c = need_add ? 1 : 0;
        }
    }
    return c;
}
// the class of original index o goes on record: cls[o], and the two lists' members per block of 256 ORIGINAL indices (integer
// atomics: order-independent); the lists are in original scan order: class 1 (down-sampled insert), then class 2
__device__ __forceinline__ void mi_record(uint8_t c, uint32_t o, int N, uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cnt) {
    cls[FLH_IDX(206, o, N)] = c;
    if (blk_cnt && c) {
        const uint32_t nb = ((uint32_t)N + 255u) >> 8;
        atomicAdd(blk_cnt + (c == 1 ? 0u : nb) + (o >> 8), 1u);
    }
}

// k_mi_classify: one THREAD per query (Morton order; outputs in ORIGINAL scan order: .w of the body point carries the index).  A
// query whose cached nearest neighbour lies inside the search bound -- all but a handful of a scan's points -- is decided on the
// spot.  The others (no map point inside the bound: points_near[0] of the reference's UNBOUNDED search lies beyond it) are LISTED
// (far[0] = their number, far[1..] = the queries; one atomic per wave) when defer_far is set, and k_far_search -- one WAVE per
// listed query (grid-stride), so that a cluster of far points is searched side by side -- finds their nearest map point by the
// shell search and decides them.  (Round 5: a kernel of its own listed them before the classification; k_cls_compact re-arms the
// counter.)
__global__ void __launch_bounds__(256)
k_mi_classify(StateDev s, StateDev s_search, const float4* __restrict__ body, NnSrc nn,
              const uint8_t* __restrict__ nn_cnt, float max_sqdist, int N, uint32_t map_points,
              double fsm, int ekf_inited, float4* __restrict__ world_out, uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cnt,
              uint32_t* __restrict__ far, int defer_far, u64* __restrict__ tab_fill, uint32_t tab_words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // the voxel table of the Add_Points that is enqueued behind this call's kernels arrives empty (0xFF: empty keys, maximal
    // values) without a fill launch of its own; k_cls_compact, two kernels on, starts filling it
    for (uint32_t j = (uint32_t)i; j < tab_words; j += gridDim.x * 256u) tab_fill[j] = ~0ull;
    bool need = false;
    if (i < N) {
        const float4 b = body[i];
        const uint32_t o = __float_as_uint(b.w);
        float wx, wy, wz;
        body_to_world(s, b.x, b.y, b.z, wx, wy, wz);  // pointBodyToWorld with the POSTERIOR state (:436)
        float sx, sy, sz;
        body_to_world(s_search, b.x, b.y, b.z, sx, sy, sz);  // the world position the last search used
        world_out[FLH_IDX(205, o, N)] = make_float4(wx, wy, wz, 0.f);
        const int cnt = nn_cnt[i];                              // found inside the bound, ascending
        const int true_cnt = map_points < 5u ? (int)map_points : 5;  // what the unbounded search returns
        float4 near[5];                                         // points_near as the bounded search left them
        nn_rows<0>(nn, i, N, near);
        const float4 n0 = near[0];
        if (defer_far) {
            need = true;
            if (cnt != 0 && __float_as_uint(n0.w) != 0xFFFFFFFFu && dist2(sx, sy, sz, n0.x, n0.y, n0.z) <= max_sqdist) need = false;
        }
        if (!need) mi_record(mi_decide(cnt, true_cnt, ekf_inited, fsm, max_sqdist, wx, wy, wz, sx, sy, sz, near), o, N, cls, blk_cnt);
    }
    const unsigned long long bal = __ballot(need);
    if (bal == 0ull) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(far, (uint32_t)__popcll(bal));
    base = (uint32_t)__shfl((int)base, 0, 64);
    if (need) far[1 + FLH_IDX(231, base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)), N)] = (uint32_t)i;
}
__global__ void __launch_bounds__(256)
k_far_search(GridParams g, StateDev s, StateDev s_search, const float4* __restrict__ body, uint32_t hash_size, const uint32_t* __restrict__ live,
             NnSrc nn, const uint8_t* __restrict__ nn_cnt, float max_sqdist, uint32_t map_points, double fsm, int ekf_inited,
             uint8_t* __restrict__ cls, uint32_t* __restrict__ blk_cnt, const uint32_t* __restrict__ far, int N) {
    const uint32_t n = min(far[0], (uint32_t)N);
    const int lane = threadIdx.x & 63;
    for (uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n; k += gridDim.x * 4) {  // wave-uniform
        const int q = (int)far[1 + k];
        const float4 b = body[q];
        float sx, sy, sz;
        body_to_world(s_search, b.x, b.y, b.z, sx, sy, sz);
        float4 near[5];
        near[0] = far_search(g, q, lane, sx, sy, sz, hash_size, live, nn);  // (also left in row 0 of the neighbour cache)
        if (lane == 0) {
            float wx, wy, wz;
            body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
            const int true_cnt = map_points < 5u ? (int)map_points : 5;
            nn_rows<1>(nn, q, N, near);
            mi_record(mi_decide((int)nn_cnt[q], true_cnt, ekf_inited, fsm, max_sqdist, wx, wy, wz, sx, sy, sz, near),
                      __float_as_uint(b.w), N, cls, blk_cnt);
        }
    }
}

// One 16-byte system-scope store: {a, b, c, sequence} lands in pinned host memory as one granule (the host polls the sequence word
// and then trusts the other three -- the hand-off k_fit's group reducers use, MI355X_MICROARCH.md "granule").
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void publish_granule(uint32_t* dst, uint32_t a, uint32_t b, uint32_t c, uint32_t seq) {
    const u32x4g v = {a, b, c, seq};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
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
// A map change enqueued before the host knows how many points it holds (flh_map_incremental without the wait in the middle):
// the launches are sized for `cap` points, the kernels read the true lengths {n1, n} from device memory (k_cls_compact left them
// there) and -- when the change turns out larger than the launches allow -- do NOTHING at all; k_map_publish then tells the host,
// which replays the change with launch sizes that fit (flh_api.cpp: map_settle).  p == nullptr: the host's n1 / n hold.
struct MiCounts {
    const uint32_t* p;
    uint32_t cap;
};
__device__ __forceinline__ bool mi_counts(const MiCounts& mc, uint32_t& n1, uint32_t& n) {
    if (!mc.p) return true;
    n1 = mc.p[0];
    n = mc.p[1];
    return n <= mc.cap;
}
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

// First kernel of a map change.  The n1 points inserted WITH down-sampling are grouped by voxel in a hash table (open addressing,
// 2 x u64 per slot: the packed voxel key, and the best new point of the voxel so far as (fp32 distance to the voxel centre << 32)
// | ~index -- an unsigned minimum over it picks the nearest point and, among equally near ones, the LATEST, which is the order
// Add_Points processes them in).  Order-independent, so the outcome does not depend on which thread gets where first.
// alive_new = 0 for these points (k_add_resolve decides), 1 for the n - n1 points inserted as they are; the change's counters
// (re-index flags, removed points) start at zero.  The table arrives filled with 0xFF (empty keys, maximal values).
__device__ __forceinline__ uint32_t vox_slot(u64 key, int shift) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift); }
__device__ __forceinline__ void vox_insert(u64* __restrict__ tab, uint32_t mask, int shift, const float4& p, uint32_t i, double ds) {
    long long kx, ky, kz;
    vox_of(p.x, p.y, p.z, ds, kx, ky, kz);
    const u64 key = pack_vox(kx, ky, kz);
    const u64 val = ((u64)__float_as_uint(dist_to_center(p.x, p.y, p.z, kx, ky, kz, ds)) << 32) | (u64)(~i);
    uint32_t slot = vox_slot(key, shift);
    for (;;) {
        const u64 prev = atomicCAS(tab + 2 * (size_t)FLH_IDX(209, slot, (u64)mask + 1), ~0ull, key);
        if (prev == ~0ull || prev == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(tab + 2 * (size_t)slot + 1, val);
}
__global__ void __launch_bounds__(256) k_add_insert(const float4* __restrict__ add, uint32_t n1, uint32_t n, double ds,
                                                    u64* __restrict__ tab, uint32_t mask, int shift, uint8_t* __restrict__ alive_new,
                                                    uint32_t* __restrict__ ctr, MiCounts mc) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { ctr[2] = 0u; ctr[3] = 0u; }
    const uint32_t launched = n;  // the points this launch was sized for (with device-side lengths: an upper bound)
    const bool applies = mi_counts(mc, n1, n);
    if (!applies || i >= n) {
        // device-side lengths: the general path's scan and sort run over the whole bound, so what lies behind the change's true
        // end (or all of it, when the change is larger than the launches and is not applied) must read "no point"
        if (mc.p && i < launched) alive_new[i] = 0;
        return;
    }
    alive_new[i] = i < n1 ? 0 : 1;
    if (i >= n1) return;
    vox_insert(tab, mask, shift, add[i], i, ds);
}

// The two lists of map_incremental (src/laserMapping.cpp:463-466: PointToAdd, PointNoNeedDownsample) compacted in ORIGINAL scan
// order, class 1 first -- ONE kernel (round 5: two library scan kernels over 2 N flags + this one).  k_mi_classify left the
// lists' members per block of 256 original indices (blk_cnt[0 .. nb): class 1, [nb .. 2 nb): class 2); every workgroup adds up
// the blocks before its own (and all of them: the class-1 total is where class 2 starts) -- at most a few words per thread -- and
// places its own 256 entries by ballot.  Also hands the two list lengths on: to the host as a granule {PointToAdd, PointToAdd +
// PointNoNeedDownsample, 0, seq} (it sizes the launches of Add_Points with them; a copy + stream synchronisation would cost more
// than this whole kernel), and to dev_counts in device memory, where the kernels of an Add_Points enqueued WITHOUT waiting for the
// granule read them (MiCounts).  cnt_next[0 .. next_words): the NEXT call's counters (the other half of a double buffer) as
// their last use left them, zeroed here.
// ins.tab != nullptr: that Add_Points is enqueued right behind this kernel with launches sized for ins.bound points, and this
// kernel does its first step -- what k_add_insert does -- on the way: a point of list 1 enters the voxel table (emptied by
// k_mi_classify) the moment its place in the list is known.  A change larger than the bound is not applied: no insert, alive_new
// reads "no point" over the whole bound (k_add_insert's rule).
struct AddIns {
    u64* tab;
    uint32_t mask;
    int shift;
    double ds;
    uint8_t* alive_new;
    uint32_t* ctr;
    uint32_t bound;
};
__global__ void __launch_bounds__(256) k_cls_compact(const float4* __restrict__ world, const uint8_t* __restrict__ cls,
                                                     const uint32_t* __restrict__ blk_cnt, uint32_t* __restrict__ cnt_next,
                                                     uint32_t next_words, int N, float4* __restrict__ out,
                                                     uint32_t* __restrict__ host_counts, uint32_t seq, uint32_t* __restrict__ dev_counts,
                                                     uint32_t* __restrict__ far, AddIns ins) {
    __shared__ uint32_t s_red[4][4], s_wave[4][2];
    const uint32_t nb = ((uint32_t)N + 255u) >> 8, b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t v[4] = {0u, 0u, 0u, 0u};  // class 1 before this block, class 1 in all, class 2 before, class 2 in all
    for (uint32_t j = (uint32_t)tid; j < nb; j += 256u) {
        const uint32_t c1 = blk_cnt[j], c2 = blk_cnt[nb + j];
        v[1] += c1; v[3] += c2;
        if (j < b) { v[0] += c1; v[2] += c2; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        if (lane == 0) s_red[wave][k] = v[k];
    }
    const int i = (int)(b * 256u) + tid;
    const uint8_t c = i < N ? cls[i] : (uint8_t)0;
    const unsigned long long m1 = __ballot(c == 1), m2 = __ballot(c == 2), lt = (1ull << lane) - 1ull;
    if (lane == 0) { s_wave[wave][0] = (uint32_t)__popcll(m1); s_wave[wave][1] = (uint32_t)__popcll(m2); }
    __syncthreads();
    uint32_t before1 = 0, total1 = 0, before2 = 0, total2 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        before1 += s_red[w][0]; total1 += s_red[w][1]; before2 += s_red[w][2]; total2 += s_red[w][3];
        if (w < wave) { w1 += s_wave[w][0]; w2 += s_wave[w][1]; }
    }
    const uint32_t total = total1 + total2;
    const bool applies = ins.tab != nullptr && total <= ins.bound;
    if (b == 0 && tid == 0) {
        if (far) far[0] = 0u;  // k_far_search (before this kernel on the stream) has read the list: empty for the next call
        if (host_counts) publish_granule(host_counts, total1, total, 0u, seq);
        if (dev_counts) { dev_counts[0] = total1; dev_counts[1] = total; }
        if (ins.tab) { ins.ctr[2] = 0u; ins.ctr[3] = 0u; }  // the change's counters (re-index flags, removed points) start at zero
    }
    for (uint32_t j = b * 256u + (uint32_t)tid; j < next_words; j += gridDim.x * 256u) cnt_next[j] = 0u;  // (what its last use left)
    if (ins.tab)  // behind the change's true end (the general path's scan and sort run over the whole bound): "no point"
        for (uint32_t j = (applies ? total : 0u) + b * 256u + (uint32_t)tid; j < ins.bound; j += gridDim.x * 256u) ins.alive_new[j] = 0;
    if (c == 1) {
        const uint32_t at = before1 + w1 + (uint32_t)__popcll(m1 & lt);
        const float4 p = world[i];
        out[FLH_IDX(207, at, N)] = p;
        if (applies) {
            ins.alive_new[at] = 0;  // (k_add_resolve decides)
            vox_insert(ins.tab, ins.mask, ins.shift, p, at, ins.ds);
        }
    } else if (c == 2) {
        const uint32_t at = total1 + before2 + w2 + (uint32_t)__popcll(m2 & lt);
        out[FLH_IDX(208, at, N)] = world[i];
        if (applies) ins.alive_new[at] = 1;
    }
}

// (start, count) of a cell and the rank of its brick
__device__ __forceinline__ uint2 lookup_cell_rank(const GridParams& g, int cx, int cy, int cz, uint32_t& rank) {
    rank = kEmptyKey;
    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz)
        return make_uint2(0u, 0u);
    rank = lookup_brick(g, brick_key(cx, cy, cz));
    if (rank == kEmptyKey) return make_uint2(0u, 0u);
    const uint32_t* st = g.starts + (size_t)rank * kBrickStride + cell_local(cx, cy, cz);
    const uint32_t a = st[0], b = st[1];
    return make_uint2(a, b - a);
}

// Eight lanes per new point; only the voxel's best new point (k_add_insert's table) goes on.  Final state of the voxel = the
// single point nearest to its centre among {points already in the map} U {new points}; ties: a new point beats an existing
// one, a later new point beats an earlier one; a voxel whose single existing point stays nearest is left untouched.
// A displaced map point is removed on the spot: its identity is marked dead (dead_id), its storage slot becomes a
// tombstone (no search will ever select it), its brick's live count drops.  Threads of other voxels may read that slot
// while it changes -- either value lies outside THEIR voxel, so it does not matter which they see.
__global__ void __launch_bounds__(256)
k_add_resolve(GridParams g, float4* pts_rw /* = g.pts */, const float4* __restrict__ add, const u64* __restrict__ tab, uint32_t mask,
              int shift, uint32_t n, double ds, uint8_t* __restrict__ dead_id, uint32_t* __restrict__ live, uint32_t* __restrict__ ctr,
              uint8_t* __restrict__ alive_new, MiCounts mc) {
    {
        uint32_t n_all = n;
        if (!mi_counts(mc, n, n_all)) return;  // n = the points inserted with down-sampling
    }
    // eight lanes per point: the cells the voxel box overlaps are dealt to the lanes, so the directory probe -> prefix table
    // -> points chain of each cell runs side by side instead of one after the other (it was 106 us with one thread per voxel)
    constexpr int L = 8;
    const uint32_t j = (blockIdx.x * 256 + threadIdx.x) / L;
    const int lane = threadIdx.x & (L - 1);
    if (j >= n) return;
    const float4 p0 = add[j];
    long long kx, ky, kz;
    vox_of(p0.x, p0.y, p0.z, ds, kx, ky, kz);
    const u64 key = pack_vox(kx, ky, kz);
    uint32_t slot = vox_slot(key, shift);
    while (tab[2 * (size_t)slot] != key) slot = (slot + 1) & mask;  // present: k_add_insert put it there
    const u64 best = tab[2 * (size_t)slot + 1];
    const uint32_t best_new = ~(uint32_t)best;
    if (best_new != j) return;  // group-uniform: another new point of this voxel is nearer to the centre (or as near and later)
    const float best_d = __uint_as_float((uint32_t)(best >> 32));
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
    const int sx = c1x - c0x + 1, sy = c1y - c0y + 1, sz = c1z - c0z + 1;
    const int ncell = sx * sy * sz;
    int n_exist = 0;
    float best_ed = INFINITY;
    uint32_t best_e = 0xFFFFFFFFu;
    // ---- the usual case (round 6): the voxel box overlaps at most eight search cells, one per lane.  The lane resolves its cell
    // ONCE (directory probe -> prefix table), loads the cell's points four at a time, drops those outside a float box around the
    // voxel (a superset of it: the corners rounded to float and widened) before the exact fp64 voxel test, and REMEMBERS the few
    // points of the voxel it finds (a map down-sampled to this voxel size holds one or two) -- so that emptying the voxel below
    // needs no second walk.  Round 5 walked every cell twice, one dependent load per point: ~25 dependent round trips, 21 us.
    // Other voxels' groups may change slots of this cell meanwhile, never a point of THIS voxel: what was remembered is what a
    // second walk would find.
    constexpr int kKeep = 4;
    uint32_t mslot[kKeep], mid[kKeep];
    int nm = 0;
    uint32_t my_rank = kEmptyKey, my_start = 0, my_cnt = 0;
    const bool one_cell_per_lane = ncell <= L;  // (group-uniform)
    if (one_cell_per_lane) {
        if (lane < ncell) {
            const int c = lane;
            const int z = c0z + c / (sx * sy), y = c0y + (c / sx) % sy, x = c0x + c % sx;
            const uint2 ce = lookup_cell_rank(g, x, y, z, my_rank);
            my_start = ce.x;
            my_cnt = ce.y;
        }
        const float ex = fmaxf(fabsf(bx0), fabsf(bx1)) * 1e-6f + 1e-30f, ey = fmaxf(fabsf(by0), fabsf(by1)) * 1e-6f + 1e-30f,
                    ez = fmaxf(fabsf(bz0), fabsf(bz1)) * 1e-6f + 1e-30f;
        for (uint32_t i0 = 0; i0 < my_cnt; i0 += 4u) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = pts_rw[FLH_IDX(210, my_start + min(i0 + (uint32_t)u, my_cnt - 1u), g.pts_cap)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + (uint32_t)u >= my_cnt || is_tombstone(q[u])) continue;
                if (!(q[u].x >= bx0 - ex && q[u].x <= bx1 + ex && q[u].y >= by0 - ey && q[u].y <= by1 + ey && q[u].z >= bz0 - ez &&
                      q[u].z <= bz1 + ez))
                    continue;
                long long qx, qy, qz;
                vox_of(q[u].x, q[u].y, q[u].z, ds, qx, qy, qz);
                if (qx != kx || qy != ky || qz != kz) continue;
                const uint32_t id = __float_as_uint(q[u].w);
                if (nm < kKeep) { mslot[nm] = my_start + i0 + (uint32_t)u; mid[nm] = id; }
                ++nm;
                ++n_exist;
                const float d = dist_to_center(q[u].x, q[u].y, q[u].z, kx, ky, kz, ds);
                if (d < best_ed || (d == best_ed && id < best_e)) { best_ed = d; best_e = id; }  // tie: lower index
            }
        }
    } else {
        for (int c = lane; c < ncell; c += L) {
            const int z = c0z + c / (sx * sy), y = c0y + (c / sx) % sy, x = c0x + c % sx;
            const uint2 ce = lookup_cell(g, x, y, z);
            for (uint32_t i = ce.x; i < ce.x + ce.y; ++i) {
                const float4 q = pts_rw[FLH_IDX(210, i, g.pts_cap)];
                if (is_tombstone(q)) continue;
                long long qx, qy, qz;
                vox_of(q.x, q.y, q.z, ds, qx, qy, qz);
                if (qx != kx || qy != ky || qz != kz) continue;
                ++n_exist;
                const float d = dist_to_center(q.x, q.y, q.z, kx, ky, kz, ds);
                const uint32_t id = __float_as_uint(q.w);
                if (d < best_ed || (d == best_ed && id < best_e)) { best_ed = d; best_e = id; }  // tie: lower index
            }
        }
    }
#pragma unroll
    for (int o = L / 2; o >= 1; o >>= 1) {  // the group's count and its (distance, index) minimum
        n_exist += __shfl_xor(n_exist, o, L);
        const float od = __shfl_xor(best_ed, o, L);
        const uint32_t oe = (uint32_t)__shfl_xor((int)best_e, o, L);
        if (od < best_ed || (od == best_ed && oe < best_e)) { best_ed = od; best_e = oe; }
    }
    const bool new_wins = !(best_ed < best_d);  // an existing point displaces only when strictly nearer
    if (n_exist == 1 && !new_wins) return;      // the single existing point stays; every new point is dropped
    // otherwise the voxel is emptied except for the winner
    if (n_exist > 0) {
        if (one_cell_per_lane && nm <= kKeep) {
            for (int m = 0; m < nm; ++m) {
                if (new_wins || mid[m] != best_e) {
                    dead_id[FLH_IDX(211, mid[m], g.ids_cap)] = 1;
                    pts_rw[FLH_IDX(212, mslot[m], g.pts_cap)] = tombstone();
                    atomicSub(live + FLH_IDX(213, my_rank, g.rows_cap), 1u);
                    atomicAdd(ctr + 3, 1u);
                }
            }
        } else {
            // (more points of the voxel in one cell than a lane remembers, or a voxel box over more than eight cells: walk again)
            for (int c = lane; c < ncell; c += L) {
                const int z = c0z + c / (sx * sy), y = c0y + (c / sx) % sy, x = c0x + c % sx;
                uint32_t rank;
                const uint2 ce = lookup_cell_rank(g, x, y, z, rank);
                for (uint32_t i = ce.x; i < ce.x + ce.y; ++i) {
                    const float4 q = pts_rw[i];
                    if (is_tombstone(q)) continue;
                    long long qx, qy, qz;
                    vox_of(q.x, q.y, q.z, ds, qx, qy, qz);
                    if (qx != kx || qy != ky || qz != kz) continue;
                    const uint32_t id = __float_as_uint(q.w);
                    if (new_wins || id != best_e) {
                        dead_id[FLH_IDX(211, id, g.ids_cap)] = 1;
                        pts_rw[FLH_IDX(212, i, g.pts_cap)] = tombstone();
                        atomicSub(live + FLH_IDX(213, rank, g.rows_cap), 1u);
                        atomicAdd(ctr + 3, 1u);
                    }
                }
            }
        }
    }
    if (new_wins && lane == 0) alive_new[FLH_IDX(214, best_new, n)] = 1;
}

// Delete_Point_Boxes over the storage: every live slot inside a box becomes a tombstone
__global__ void __launch_bounds__(256) k_delete_boxes(GridParams g, float4* pts_rw /* = g.pts */, uint32_t n_slots,
                                                      const float* __restrict__ boxes, int nb, uint8_t* __restrict__ dead_id,
                                                      uint32_t* __restrict__ live, uint32_t* __restrict__ ctr) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_slots) return;
    const float4 p = pts_rw[i];
    if (is_tombstone(p)) return;
    bool del = false;
    for (int b = 0; b < nb; ++b) {
        const float* bx = boxes + 6 * b;
        if (p.x >= bx[0] && p.x < bx[3] && p.y >= bx[1] && p.y < bx[4] && p.z >= bx[2] && p.z < bx[5]) del = true;
    }
    if (!del) return;
    int cx, cy, cz;
    float fx, fy, fz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
    cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);  // as k_map_keys
    const uint32_t rank = lookup_brick(g, brick_key(cx, cy, cz));
    dead_id[FLH_IDX(215, __float_as_uint(p.w), g.ids_cap)] = 1;
    pts_rw[i] = tombstone();
    if (rank != kEmptyKey) atomicSub(live + FLH_IDX(216, rank, g.rows_cap), 1u);
    atomicAdd(ctr + 3, 1u);
}

// ------------------------------------------------------------------------------------------------
// Insertion of points into the slack-carrying brick storage.
// k_ins_prepare: surviving new points get their identity (n_ids + rank), are appended to the index-ordered array and
//   keyed by the brick they fall into; a point outside the grid raises flag 1 (the caller re-indexes from scratch).
// k_brick_rewrite: one block per brick that receives points: its live points + the new ones are counting-sorted by local
//   cell in LDS and written back -- in place while they fit the brick's capacity, else into a freshly bump-allocated range
//   (the old one becomes tombstones); a brick seen for the first time gets a table row and a directory entry.  Nothing
//   outside the brick moves.  Anything that does not fit (LDS tile, storage, table rows, directory) raises a flag and leaves
//   the brick as it was; the caller then rebuilds the index from the index-ordered array, which is always complete.
// ------------------------------------------------------------------------------------------------
constexpr int kTile = 2048;  // points of one brick that fit the LDS tile

__global__ void __launch_bounds__(256)
k_ins_prepare(GridParams g, const float4* __restrict__ add, const uint8_t* __restrict__ alive_new, const uint32_t* __restrict__ incl,
              uint32_t n, uint32_t n_ids, float4* __restrict__ map_orig, uint8_t* __restrict__ dead_id, float4* __restrict__ ins,
              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ ctr) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !alive_new[i]) return;
    const uint32_t r = (uint32_t)FLH_IDX(217, incl[i] - 1, n), id = (uint32_t)FLH_IDX(218, n_ids + r, g.ids_cap);
    float4 p = add[i];
    p.w = 0.f;
    map_orig[id] = p;
    dead_id[id] = 0;
    int cx, cy, cz;
    float fx, fy, fz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
    if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz) {
        atomicOr(ctr + 2, 1u);
        cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
    }
    p.w = __uint_as_float(id);
    ins[r] = p;
    keys[r] = brick_key(cx, cy, cz);
    vals[r] = r;
}

// the run of new points that starts at sorted position j enters its brick (HEAD_KNOWN: j is known to be the head of a run)
template <bool HEAD_KNOWN>
__device__ __forceinline__ void brick_rewrite_one(const GridParams& g, float4* pts, uint32_t* starts, uint2* hash, uint32_t* __restrict__ cap_end,
                                                  uint32_t* __restrict__ live, uint32_t* __restrict__ ctr, const float4* __restrict__ ins,
                                                  const uint32_t* __restrict__ ks, const uint32_t* __restrict__ perm, uint32_t n,
                                                  uint32_t pts_cap, uint32_t rows_cap, uint32_t j) {
    __shared__ float4 buf[kTile];
    __shared__ uint32_t hist[64], offs[64];
    __shared__ uint32_t s_cnt, s_base, s_rank, s_cap_end, s_ok;
    const int tid = threadIdx.x;
    const uint32_t key = ks[j];
    if (!HEAD_KNOWN) {
        if (key == kEmptyKey) return;           // beyond the surviving points (n is the host's upper bound of their number)
        if (j > 0 && ks[j - 1] == key) return;  // block-uniform: not the head of its brick's run
    }
    uint32_t e = j + 1;
    while (e < n && ks[e] == key) ++e;
    const uint32_t run = e - j;
    const uint32_t rank = lookup_brick(g, key);
    uint32_t old_base = 0, old_end = 0, old_cap_end = 0;
    if (rank != kEmptyKey) {
        old_base = starts[(size_t)FLH_IDX(219, rank, rows_cap) * kBrickStride];
        old_end = starts[(size_t)rank * kBrickStride + 64];
        old_cap_end = cap_end[rank];
#ifdef FLH_BOUNDS
        (void)FLH_IDX(220, old_end, (u64)old_cap_end + 1);
        (void)FLH_IDX(221, old_cap_end, (u64)pts_cap + 1);
        (void)FLH_IDX(222, old_base, (u64)old_end + 1);
#endif
    }
    if (tid == 0) { s_cnt = 0; s_ok = 1; }
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = old_base + tid; i < old_end; i += 128) {
        const float4 p = pts[i];
        if (!is_tombstone(p)) {
            const uint32_t k = atomicAdd(&s_cnt, 1u);
            if (k < (uint32_t)kTile) buf[k] = p;
        }
    }
    __syncthreads();
    const uint32_t nold = s_cnt, total = nold + run;
    if (total > (uint32_t)kTile) {  // block-uniform
        if (tid == 0) atomicOr(ctr + 2, 2u);
        return;
    }
    for (uint32_t i = tid; i < run; i += 128) buf[nold + i] = ins[FLH_IDX(223, perm[j + i], n)];
    if (tid == 0) {
        if (rank != kEmptyKey && old_base + total <= old_cap_end) {
            s_base = old_base; s_rank = rank; s_cap_end = old_cap_end;
        } else {
            const uint32_t newcap = total + max(8u, total >> 2);
            const uint32_t nb = atomicAdd(ctr + 0, newcap);
            uint32_t r = rank;
            if (nb + newcap > pts_cap || nb + newcap < nb) { atomicOr(ctr + 2, 4u); s_ok = 0; }
            else if (rank == kEmptyKey) {
                r = atomicAdd(ctr + 1, 1u);
                if (r >= rows_cap) { atomicOr(ctr + 2, 8u); s_ok = 0; }
                else {
                    const uint32_t k32 = key;
                    uint32_t slot = hash_slot(k32, g.hash_shift);
                    bool placed = false;
                    for (uint32_t tries = 0; tries <= g.hash_mask; ++tries) {
                        const uint32_t prev = atomicCAS(&hash[slot].x, kEmptyKey, k32);
                        if (prev == kEmptyKey) { hash[slot].y = r; placed = true; break; }
                        slot = (slot + 1) & g.hash_mask;
                    }
                    if (!placed) { atomicOr(ctr + 2, 16u); s_ok = 0; }
                }
            }
            s_base = nb; s_rank = r; s_cap_end = nb + newcap;
        }
    }
    __syncthreads();
    if (!s_ok) return;
    const uint32_t base = s_base, r = s_rank;
    // counting sort by local cell
    uint32_t cl[kTile / 128];
#pragma unroll
    for (int u = 0; u < kTile / 128; ++u) {
        const uint32_t i = tid + u * 128;
        cl[u] = 0;
        if (i < total) {
            const float4 p = buf[i];
            int cx, cy, cz;
            float fx, fy, fz;
            cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
            cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
            cl[u] = cell_local(cx, cy, cz);
            atomicAdd(&hist[cl[u]], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int c = 0; c < 64; ++c) {
            offs[c] = acc;
            starts[(size_t)FLH_IDX(226, r, rows_cap) * kBrickStride + c] = base + acc;
            acc += hist[c];
        }
        starts[(size_t)r * kBrickStride + 64] = base + acc;
        cap_end[r] = s_cap_end;
        live[r] = total;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTile / 128; ++u) {
        const uint32_t i = tid + u * 128;
        if (i < total) pts[FLH_IDX(224, base + atomicAdd(&offs[cl[u]], 1u), pts_cap)] = buf[i];
    }
    // what the brick no longer uses becomes tombstones: the tail of the old range (in place), or all of it (relocated)
    if (rank != kEmptyKey) {
        const uint32_t a = (base == old_base) ? old_base + total : old_base;
        const uint32_t b = (base == old_base) ? old_end : old_cap_end;
        for (uint32_t i = a + tid; i < b; i += 128) pts[FLH_IDX(225, i, pts_cap)] = tombstone();
    }
}

// the counters of a map change and the number of points it inserted, as two granules, each carrying the sequence word (system-scope
// stores need not reach the host in order): {storage top, bricks, re-index flags, seq} {removed, inserted, points of the change, seq}.
// A change whose kernels found it larger than their launches (MiCounts) did nothing: flag kMapChangeNotApplied tells the host so.
struct MapPublish {
    const uint32_t* n_alive;
    uint32_t* host_out;
    uint32_t seq;
};
__device__ __forceinline__ void map_publish(const uint32_t* ctr, const MapPublish& pb, const MiCounts& mc) {
    const uint32_t c0 = __hip_atomic_load(ctr + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   c1 = __hip_atomic_load(ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   c3 = __hip_atomic_load(ctr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t c2 = __hip_atomic_load(ctr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t na = pb.n_alive ? *pb.n_alive : 0u;
    uint32_t n1 = 0, n = 0xFFFFFFFFu;  // the size of the change is not this kernel's to know unless the lengths live on the device
    if (!mi_counts(mc, n1, n)) { c2 |= kMapChangeNotApplied; na = 0u; }
    publish_granule(pb.host_out, c0, c1, c2, pb.seq);
    publish_granule(pb.host_out + 4, c3, na, n, pb.seq);
}

// General path: one workgroup per sorted position, the heads of the runs go on.
__global__ void __launch_bounds__(128)
k_brick_rewrite(GridParams g, float4* pts /* = g.pts */, uint32_t* starts /* = g.starts */, uint2* hash /* = g.hash */,
                uint32_t* __restrict__ cap_end, uint32_t* __restrict__ live, uint32_t* __restrict__ ctr,
                const float4* __restrict__ ins, const uint32_t* __restrict__ ks, const uint32_t* __restrict__ perm, uint32_t n,
                uint32_t pts_cap, uint32_t rows_cap, MiCounts mc) {
    {
        uint32_t n1_unused = 0;
        if (!mi_counts(mc, n1_unused, n)) return;
    }
    if (blockIdx.x >= n) return;
    brick_rewrite_one<false>(g, pts, starts, hash, cap_end, live, ctr, ins, ks, perm, n, pts_cap, rows_cap, blockIdx.x);
}
// Small changes: k_ins_sort_small has listed the heads of the runs (heads[0 .. ctr[6])), the workgroups share them out -- a scan of
// a running odometry touches a few hundred bricks, the launch is sized for 8 192 points -- and the LAST workgroup to finish hands
// the change's counters to the host: the publication needs no launch of its own, and the host, which waits for it before the next
// scan's first search, sees it a launch earlier.  "Last" by tickets in two levels (tick[32 (1 + g)]: the 32 workgroups of group g,
// tick[0]: the groups; 128 bytes apart): atomics on ONE address serialise at ~11 ns each -- a thousand workgroups on one ticket
// were 8 us of this kernel (call 18) -- so only the workgroups that have a brick take one, at most 32 + 32 deep.
constexpr uint32_t kTickWords = 32u * 33u;
uint32_t brick_ticket_words() { return kTickWords; }
__global__ void __launch_bounds__(128)
k_brick_rewrite_heads(GridParams g, float4* pts /* = g.pts */, uint32_t* starts /* = g.starts */, uint2* hash /* = g.hash */,
                      uint32_t* __restrict__ cap_end, uint32_t* __restrict__ live, uint32_t* __restrict__ ctr,
                      const float4* __restrict__ ins, const uint32_t* __restrict__ ks, const uint32_t* __restrict__ perm, uint32_t n,
                      uint32_t pts_cap, uint32_t rows_cap, MiCounts mc, const uint32_t* __restrict__ heads, uint32_t* __restrict__ tick,
                      MapPublish pb) {
    uint32_t n1_unused = 0;
    const uint32_t nheads = mi_counts(mc, n1_unused, n) ? min(ctr[6], n) : 0u;  // (block-uniform)
    const uint32_t nw = min(nheads, gridDim.x);                                   // the workgroups that have a brick
    if (nw == 0u) {  // nothing survived (or the change is larger than its launches): the counters as the earlier kernels left them
        if (blockIdx.x == 0 && threadIdx.x == 0) map_publish(ctr, pb, mc);
        return;
    }
    if (blockIdx.x >= nw) return;
    for (uint32_t k = blockIdx.x; k < nheads; k += gridDim.x) {
        brick_rewrite_one<true>(g, pts, starts, hash, cap_end, live, ctr, ins, ks, perm, n, pts_cap, rows_cap, heads[FLH_IDX(233, k, n)]);
        __syncthreads();  // the tile and its counters are free for the next brick
    }
    if (threadIdx.x == 0) {
        // The counters the publication reads are thread 0's own device-scope atomics: drained (vmcnt(0)) before the ticket is
        // taken, read back with agent-scope loads by the last arriver.  No release / acquire fence: at agent scope either one
        // sweeps the L2 (write-back / invalidate), once per workgroup -- 29 us for this kernel when it was tried (call 17).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t grp = blockIdx.x >> 5, gsize = min(32u, nw - 32u * grp), ngroups = (nw + 31u) >> 5;
        uint32_t* const tk = tick + 32u * (1u + grp);
        if (__hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1u) {
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the next change's tickets)
            if (__hip_atomic_fetch_add(tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u) {
                __hip_atomic_store(tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                map_publish(ctr, pb, mc);
            }
        }
    }
}

// Bricks that only LOSE points (Delete_Point_Boxes of lasermap_fov_segment, src/laserMapping.cpp:231-277, for the life of the
// node): a removal tombstones its slot in place, so the brick's range keeps its length and every later search of its cells still
// loads and evaluates the tombstones.  After a removal, every brick whose live points have fallen below half of its range is
// compacted where it lies: the live points are re-sorted by local cell into the front of the range, the prefix table shrinks to
// them, the rest of the range becomes slack again.  One workgroup per table row; a brick that needs nothing costs three loads.
// The map's contents and index order do not change (ids stay), only what the search has to read.  ctr[4] counts the purged bricks.
__global__ void __launch_bounds__(128)
k_brick_purge(GridParams g, float4* pts /* = g.pts */, uint32_t* starts /* = g.starts */, const uint32_t* __restrict__ live,
              uint32_t* __restrict__ ctr, uint32_t nrows, uint32_t pts_cap) {
    __shared__ float4 buf[kTile];
    __shared__ uint32_t hist[64], offs[64];
    __shared__ uint32_t s_cnt;
    const uint32_t r = blockIdx.x;
    const int tid = threadIdx.x;
    if (r >= nrows) return;
    const uint32_t base = starts[(size_t)r * kBrickStride], end = starts[(size_t)r * kBrickStride + 64], lv = live[r];
    const uint32_t span = end - base;
    if (end < base || end > pts_cap || 2u * lv >= span || (span < 8u && lv != 0u)) return;  // block-uniform: nothing to gain
    if (tid == 0) s_cnt = 0;
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = base + tid; i < end; i += 128) {
        const float4 p = pts[i];
        if (!is_tombstone(p)) {
            const uint32_t k = atomicAdd(&s_cnt, 1u);
            if (k < (uint32_t)kTile) buf[k] = p;
        }
    }
    __syncthreads();
    const uint32_t total = s_cnt;
    if (total > (uint32_t)kTile || total > span) return;  // (cannot happen: a range never holds more than the tile; leave it alone)
    uint32_t cl[kTile / 128];
#pragma unroll
    for (int u = 0; u < kTile / 128; ++u) {
        const uint32_t i = tid + u * 128;
        cl[u] = 0;
        if (i < total) {
            const float4 p = buf[i];
            int cx, cy, cz;
            float fx, fy, fz;
            cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
            cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
            cl[u] = cell_local(cx, cy, cz);
            atomicAdd(&hist[cl[u]], 1u);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int c = 0; c < 64; ++c) {
            offs[c] = acc;
            starts[(size_t)r * kBrickStride + c] = base + acc;
            acc += hist[c];
        }
        starts[(size_t)r * kBrickStride + 64] = base + acc;
        atomicAdd(ctr + 4, 1u);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTile / 128; ++u) {
        const uint32_t i = tid + u * 128;
        if (i < total) pts[base + atomicAdd(&offs[cl[u]], 1u)] = buf[i];
    }
    for (uint32_t i = base + total + tid; i < end; i += 128) pts[i] = tombstone();
}

// index-ordered array -> contiguous array of the live points, in order (download, full re-index)
__global__ void __launch_bounds__(256) k_byte_flags(const uint8_t* __restrict__ in, uint32_t n, int invert, uint32_t* __restrict__ flags,
                                                    uint32_t* __restrict__ keys_sentinel) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    flags[i] = ((in[i] != 0) != (invert != 0)) ? 1u : 0u;
    if (keys_sentinel) keys_sentinel[i] = kEmptyKey;  // k_ins_prepare overwrites the first n_alive of them
}
__global__ void __launch_bounds__(256) k_live_compact(const float4* __restrict__ map_orig, const uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ incl, uint32_t n_ids, float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ids || !flags[i]) return;
    float4 p = map_orig[i];
    p.w = 0.f;
    out[FLH_IDX(227, incl[i] - 1, n_ids)] = p;
}

// ------------------------------------------------------------------------------------------------
// Small map changes (at most kSmallMax points -- every scan of a running odometry): ONE workgroup gives the surviving points
// their ids, keys them by brick and sorts the keys -- k_byte_flags + the prefix sum + k_ins_prepare + the device-wide sort of
// the general path (eight launches) in one.  The sort is rocprim's block radix sort over keys re-packed to the bits that
// actually differ (per axis: brick coordinate - block minimum, in as many bits as the block's range needs): two or three
// eight-bit passes.  Same outputs in the same buffers as the general path: stable, ascending key, sentinels behind.
// ------------------------------------------------------------------------------------------------
constexpr int kSmallThreads = 1024;
constexpr int kSmallItems = 8;
constexpr uint32_t kSmallMax = kSmallThreads * kSmallItems;
template <int ITEMS> using SmallSort = rocprim::block_radix_sort<uint32_t, kSmallThreads, ITEMS, uint32_t>;
using SmallScan = rocprim::block_scan<uint32_t, kSmallThreads>;
struct SmallShared {
    union {
        SmallSort<8>::storage_type sort8;
        SmallSort<4>::storage_type sort4;
        SmallSort<2>::storage_type sort2;
        SmallScan::storage_type scan;
    } st;
    uint32_t red[kSmallThreads / 64][6];
    uint32_t mn[3];
    int bits[3];
    uint32_t last[kSmallThreads];  // a thread's last sorted key (run heads)
};
uint32_t small_change_max() { return kSmallMax; }
template <int ITEMS> __device__ __forceinline__ typename SmallSort<ITEMS>::storage_type& small_sort_storage(SmallShared& sh);
template <> __device__ __forceinline__ SmallSort<8>::storage_type& small_sort_storage<8>(SmallShared& sh) { return sh.st.sort8; }
template <> __device__ __forceinline__ SmallSort<4>::storage_type& small_sort_storage<4>(SmallShared& sh) { return sh.st.sort4; }
template <> __device__ __forceinline__ SmallSort<2>::storage_type& small_sort_storage<2>(SmallShared& sh) { return sh.st.sort2; }

// the body for a change of at most ITEMS * 1024 points: the sort's cost goes with the items a thread holds, and the launch is
// sized for the largest change the path takes -- a scan of a running odometry inserts a fraction of that
template <int ITEMS>
__device__ __forceinline__ void ins_sort_small_body(SmallShared& sh, const GridParams& g, const float4* __restrict__ add,
                                                    const uint8_t* __restrict__ alive_new, uint32_t n, uint32_t n_ids,
                                                    float4* __restrict__ map_orig, uint8_t* __restrict__ dead_id, float4* __restrict__ ins,
                                                    uint32_t* __restrict__ keys_tmp, uint32_t* __restrict__ ks, uint32_t* __restrict__ perm,
                                                    uint32_t* __restrict__ ctr, uint32_t* __restrict__ n_alive_out,
                                                    uint32_t* __restrict__ heads) {
    const uint32_t tid = threadIdx.x;
    // items in blocked arrangement: thread t holds points ITEMS*t .. ITEMS*t + ITEMS-1, so ranks follow the input order
    bool valid[ITEMS];
    uint32_t cnt = 0;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const uint32_t i = tid * ITEMS + u;
        valid[u] = i < n && alive_new[i] != 0;
        cnt += valid[u] ? 1u : 0u;
    }
    uint32_t base = 0, total = 0;
    SmallScan().exclusive_scan(cnt, base, 0u, total, sh.st.scan);
    uint32_t key[ITEMS], val[ITEMS];
    uint32_t mn[3] = {1023u, 1023u, 1023u}, mx[3] = {0u, 0u, 0u};
    uint32_t r = base;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const uint32_t i = tid * ITEMS + u;
        key[u] = 0u;
        val[u] = 0u;
        if (valid[u]) {
            const uint32_t id = (uint32_t)FLH_IDX(228, n_ids + r, g.ids_cap);
            float4 p = add[i];
            p.w = 0.f;
            map_orig[id] = p;
            dead_id[id] = 0;
            int cx, cy, cz;
            float fx, fy, fz;
            cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
            if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz) {
                atomicOr(ctr + 2, 1u);
                cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
            }
            p.w = __uint_as_float(id);
            ins[FLH_IDX(229, r, n)] = p;
            key[u] = brick_key(cx, cy, cz);
            keys_tmp[r] = key[u];
            val[u] = r;
            ++r;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const uint32_t f = (key[u] >> (10 * (2 - d))) & 1023u;
                mn[d] = min(mn[d], f);
                mx[d] = max(mx[d], f);
            }
        }
    }
    // block-wide range of each of the key's three 10-bit fields
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            mn[d] = min(mn[d], (uint32_t)__shfl_xor((int)mn[d], o, 64));
            mx[d] = max(mx[d], (uint32_t)__shfl_xor((int)mx[d], o, 64));
        }
    if ((tid & 63u) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { sh.red[tid >> 6][d] = mn[d]; sh.red[tid >> 6][3 + d] = mx[d]; }
    }
    __syncthreads();  // also: the scan's storage is free for the sort
    if (tid < 3) {
        uint32_t a = 1023u, b = 0u;
        for (int k = 0; k < kSmallThreads / 64; ++k) { a = min(a, sh.red[k][tid]); b = max(b, sh.red[k][3 + tid]); }
        sh.mn[tid] = a;
        sh.bits[tid] = b > a ? 32 - __clz((int)(b - a)) : 0;  // no surviving point at all: a = 1023 > b = 0 -> 0 bits
    }
    __syncthreads();
    const int b1 = sh.bits[1], b2 = sh.bits[2], nbits = sh.bits[0] + b1 + b2;  // <= 30
    uint32_t pk[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const uint32_t f0 = ((key[u] >> 20) & 1023u) - sh.mn[0], f1 = ((key[u] >> 10) & 1023u) - sh.mn[1], f2 = (key[u] & 1023u) - sh.mn[2];
        pk[u] = valid[u] ? ((f0 << (b1 + b2)) | (f1 << b2) | f2) : (1u << nbits);  // monotone in (z, y, x) brick order; no point: behind all
    }
    SmallSort<ITEMS>().sort(pk, val, small_sort_storage<ITEMS>(sh), 0u, (unsigned)nbits + 1u);
    sh.last[tid] = pk[ITEMS - 1];
    __syncthreads();  // keys_tmp was written by other threads of this block; the sort's storage is free for the scan
    // the sorted positions at which a brick's run starts, listed in order for k_brick_rewrite_heads (the packed keys are one to one
    // with the brick keys)
    {
        uint32_t prev = tid ? sh.last[tid - 1] : 0u, hc = 0;
        bool hd[ITEMS];
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const uint32_t j = tid * ITEMS + u;
            hd[u] = j < total && (j == 0u || pk[u] != prev);
            hc += hd[u] ? 1u : 0u;
            prev = pk[u];
        }
        uint32_t hb = 0, nheads = 0;
        SmallScan().exclusive_scan(hc, hb, 0u, nheads, sh.st.scan);
#pragma unroll
        for (int u = 0; u < ITEMS; ++u)
            if (hd[u]) heads[FLH_IDX(234, hb++, n)] = tid * ITEMS + u;
        if (tid == 0) ctr[6] = nheads;
    }
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const uint32_t j = tid * ITEMS + u;
        if (j < n) {
            const bool has = j < total;
            perm[j] = has ? val[u] : 0u;
            ks[j] = has ? keys_tmp[FLH_IDX(230, val[u], n)] : kEmptyKey;
        }
    }
    if (tid == 0) *n_alive_out = total;
}

__global__ void __launch_bounds__(kSmallThreads)
k_ins_sort_small(GridParams g, const float4* __restrict__ add, const uint8_t* __restrict__ alive_new, uint32_t n, uint32_t n_ids,
                 float4* __restrict__ map_orig, uint8_t* __restrict__ dead_id, float4* __restrict__ ins, uint32_t* __restrict__ keys_tmp,
                 uint32_t* __restrict__ ks, uint32_t* __restrict__ perm, uint32_t* __restrict__ ctr, uint32_t* __restrict__ n_alive_out,
                 uint32_t* __restrict__ heads, MiCounts mc) {
    __shared__ SmallShared sh;
    {
        uint32_t n1_unused = 0;
        if (!mi_counts(mc, n1_unused, n)) {  // block-uniform: a change larger than this launch was sized for -- nothing is done
            if (threadIdx.x == 0) { *n_alive_out = 0u; ctr[6] = 0u; }
            return;
        }
    }
    // (block-uniform; the three bodies write the same outputs: a stable sort of the same keys)
    if (n <= 2u * kSmallThreads) ins_sort_small_body<2>(sh, g, add, alive_new, n, n_ids, map_orig, dead_id, ins, keys_tmp, ks, perm, ctr, n_alive_out, heads);
    else if (n <= 4u * kSmallThreads) ins_sort_small_body<4>(sh, g, add, alive_new, n, n_ids, map_orig, dead_id, ins, keys_tmp, ks, perm, ctr, n_alive_out, heads);
    else ins_sort_small_body<8>(sh, g, add, alive_new, n, n_ids, map_orig, dead_id, ins, keys_tmp, ks, perm, ctr, n_alive_out, heads);
}
hipError_t launch_ins_sort_small(const GridParams& g, const float4* add, const uint8_t* alive_new, uint32_t n, uint32_t n_ids,
                                 float4* map_orig, uint8_t* dead_id, float4* ins, uint32_t* keys_tmp, uint32_t* ks, uint32_t* perm,
                                 uint32_t* ctr, uint32_t* n_alive_out, uint32_t* heads, hipStream_t st, const uint32_t* dev_counts) {
    if (n == 0 || n > kSmallMax) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ins_sort_small, dim3(1), dim3(kSmallThreads), 0, st, g, add, alive_new, n, n_ids, map_orig, dead_id, ins, keys_tmp,
                       ks, perm, ctr, n_alive_out, heads, MiCounts{dev_counts, n});
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
uint32_t cls_block_words(int N) { return 2u * (((uint32_t)(N > 0 ? N : 1) + 255u) >> 8); }
hipError_t launch_mi_classify(const GridParams& g, uint32_t hash_size, uint32_t map_points, const StateDev& s_search,
                              const StateDev& s_post, const float4* body, float4* nn_pts, uint32_t* nn_idx, const float4* map_orig,
                              uint32_t n_ids, const uint8_t* nn_cnt, float max_sqdist, int N, double fsm, int ekf_inited,
                              const uint32_t* live, float4* world_out, uint8_t* cls, uint32_t* blk_cnt, uint32_t* far, hipStream_t st,
                              unsigned long long* tab_fill, uint32_t tab_words) {
    if (N <= 0) return tab_words ? hipErrorInvalidValue : hipSuccess;
    const NnSrc nn{nn_pts, nn_idx, map_orig, n_ids};
    const int defer_far = (map_points > 0 && ekf_inited) ? 1 : 0;  // else Nearest_Points[i] is empty / nobody looks at it (:463-466)
    hipLaunchKernelGGL(k_mi_classify, dim3(cdiv2(N, 256)), dim3(256), 0, st, s_post, s_search, body, nn, nn_cnt, max_sqdist, N,
                       map_points, fsm, ekf_inited, world_out, cls, blk_cnt, far, defer_far, tab_fill, tab_words);
    if (defer_far)
        hipLaunchKernelGGL(k_far_search, dim3(min(cdiv2(N, 4), 512)), dim3(256), 0, st, g, s_post, s_search, body, hash_size, live, nn, nn_cnt,
                           max_sqdist, map_points, fsm, ekf_inited, cls, blk_cnt, far, N);
    return hipGetLastError();
}
static int vox_shift(uint32_t cap) { return 64 - (31 - __builtin_clz(cap)); }
hipError_t launch_cls_compact(const float4* world, const uint8_t* cls, const uint32_t* blk_cnt, uint32_t* cnt_next, uint32_t next_words,
                              int N, float4* out, uint32_t* host_counts, uint32_t seq, hipStream_t st, uint32_t* dev_counts,
                              uint32_t* far, unsigned long long* ins_tab, uint32_t ins_cap, double ins_ds, uint8_t* ins_alive_new,
                              uint32_t* ins_ctr, uint32_t ins_bound) {
    if (N <= 0) return hipSuccess;
    const AddIns ins{ins_tab, ins_tab ? ins_cap - 1u : 0u, ins_tab ? vox_shift(ins_cap) : 0, ins_ds, ins_alive_new, ins_ctr, ins_bound};
    hipLaunchKernelGGL(k_cls_compact, dim3(cdiv2(N, 256)), dim3(256), 0, st, world, cls, blk_cnt, cnt_next, next_words, N, out,
                       host_counts, seq, dev_counts, far, ins);
    return hipGetLastError();
}
__global__ void k_map_publish(const uint32_t* __restrict__ ctr, const uint32_t* __restrict__ n_alive, uint32_t* __restrict__ host_out,
                              uint32_t seq, MiCounts mc) {
    if (threadIdx.x != 0) return;
    map_publish(ctr, MapPublish{n_alive, host_out, seq}, mc);
}
hipError_t launch_map_publish(const uint32_t* ctr, const uint32_t* n_alive, uint32_t* host_out, uint32_t seq, hipStream_t st,
                              const uint32_t* dev_counts, uint32_t cap) {
    hipLaunchKernelGGL(k_map_publish, dim3(1), dim3(64), 0, st, ctr, n_alive, host_out, seq, MiCounts{dev_counts, cap});
    return hipGetLastError();
}
hipError_t launch_aabb(const float4* pts, uint32_t M, uint32_t* out6, hipStream_t st) {
    if (M == 0) return hipSuccess;
    int blocks = cdiv2(M, 256 * 8);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(k_aabb, dim3(blocks), dim3(256), 0, st, pts, M, out6);
    return hipGetLastError();
}
// slots of the voxel table of a change that down-samples n1 points: a power of two, at most half full
uint32_t vox_table_slots(uint32_t n1) {
    uint32_t cap = 64;
    while (cap < 2u * n1 && cap < (1u << 31)) cap <<= 1;
    return cap;
}
hipError_t launch_add_insert(const float4* add, uint32_t n1, uint32_t n, double ds, u64* tab, uint32_t cap, uint8_t* alive_new,
                             uint32_t* ctr, hipStream_t st, const uint32_t* dev_counts) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_insert, dim3(cdiv2(n, 256)), dim3(256), 0, st, add, n1, n, ds, tab, cap - 1u, vox_shift(cap), alive_new, ctr,
                       MiCounts{dev_counts, n});
    return hipGetLastError();
}
// the surviving points by brick: 30-bit brick keys (sentinel 0xFFFFFFFF behind them), stable
hipError_t sort_brick_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                            uint32_t n, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)n, 0, 32, st);
}
hipError_t launch_add_resolve(const GridParams& g, float4* pts_rw, const float4* add, const u64* tab, uint32_t cap, uint32_t n,
                              double ds, uint8_t* dead_id, uint32_t* live, uint32_t* ctr, uint8_t* alive_new, hipStream_t st,
                              const uint32_t* dev_counts) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_resolve, dim3(cdiv2((long long)n * 8, 256)), dim3(256), 0, st, g, pts_rw, add, tab, cap - 1u, vox_shift(cap),
                       n, ds, dead_id, live, ctr, alive_new, MiCounts{dev_counts, n});
    return hipGetLastError();
}
hipError_t launch_delete_boxes(const GridParams& g, float4* pts_rw, uint32_t n_slots, const float* boxes, int nb, uint8_t* dead_id,
                               uint32_t* live, uint32_t* ctr, hipStream_t st) {
    if (n_slots == 0 || nb == 0) return hipSuccess;
    hipLaunchKernelGGL(k_delete_boxes, dim3(cdiv2(n_slots, 256)), dim3(256), 0, st, g, pts_rw, n_slots, boxes, nb, dead_id, live, ctr);
    return hipGetLastError();
}
hipError_t launch_brick_purge(const GridParams& g, float4* pts, uint32_t* starts, const uint32_t* live, uint32_t* ctr, uint32_t nrows,
                              uint32_t pts_cap, hipStream_t st) {
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_purge, dim3(nrows), dim3(128), 0, st, g, pts, starts, live, ctr, nrows, pts_cap);
    return hipGetLastError();
}
hipError_t launch_ins_prepare(const GridParams& g, const float4* add, const uint8_t* alive_new, const uint32_t* incl, uint32_t n,
                              uint32_t n_ids, float4* map_orig, uint8_t* dead_id, float4* ins, uint32_t* keys, uint32_t* vals,
                              uint32_t* ctr, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ins_prepare, dim3(cdiv2(n, 256)), dim3(256), 0, st, g, add, alive_new, incl, n, n_ids, map_orig, dead_id, ins,
                       keys, vals, ctr);
    return hipGetLastError();
}
hipError_t launch_brick_rewrite(const GridParams& g, float4* pts, uint32_t* starts, uint2* hash, uint32_t* cap_end, uint32_t* live,
                                uint32_t* ctr, const float4* ins, const uint32_t* ks, const uint32_t* perm, uint32_t n, uint32_t pts_cap,
                                uint32_t rows_cap, hipStream_t st, const uint32_t* dev_counts) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_rewrite, dim3(n), dim3(128), 0, st, g, pts, starts, hash, cap_end, live, ctr, ins, ks, perm, n, pts_cap,
                       rows_cap, MiCounts{dev_counts, n});
    return hipGetLastError();
}
hipError_t launch_brick_rewrite_heads(const GridParams& g, float4* pts, uint32_t* starts, uint2* hash, uint32_t* cap_end, uint32_t* live,
                                      uint32_t* ctr, const float4* ins, const uint32_t* ks, const uint32_t* perm, uint32_t n,
                                      uint32_t pts_cap, uint32_t rows_cap, hipStream_t st, const uint32_t* dev_counts, const uint32_t* heads,
                                      uint32_t* tick, const uint32_t* n_alive, uint32_t* host_out, uint32_t seq) {
    if (n == 0) return hipErrorInvalidValue;
    static_assert(kTickWords >= 32u * (1u + 1024u / 32u), "a ticket line per 32 workgroups of the launch");
    hipLaunchKernelGGL(k_brick_rewrite_heads, dim3(std::min(n, 1024u)), dim3(128), 0, st, g, pts, starts, hash, cap_end, live, ctr, ins, ks,
                       perm, n, pts_cap, rows_cap, MiCounts{dev_counts, n}, heads, tick, MapPublish{n_alive, host_out, seq});
    return hipGetLastError();
}
hipError_t launch_byte_flags(const uint8_t* in, uint32_t n, int invert, uint32_t* flags, hipStream_t st, uint32_t* keys_sentinel) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_byte_flags, dim3(cdiv2(n, 256)), dim3(256), 0, st, in, n, invert, flags, keys_sentinel);
    return hipGetLastError();
}
hipError_t launch_live_compact(const float4* map_orig, const uint32_t* flags, const uint32_t* incl, uint32_t n_ids, float4* out,
                               hipStream_t st) {
    if (n_ids == 0) return hipSuccess;
    hipLaunchKernelGGL(k_live_compact, dim3(cdiv2(n_ids, 256)), dim3(256), 0, st, map_orig, flags, incl, n_ids, out);
    return hipGetLastError();
}

#ifdef FLH_BOUNDS
void bounds_read_mapinc(unsigned long long out[5]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(BoundsRec)); }
#endif

}  // namespace flh
