// flh_search_dev.hpp -- device code of the exact 5-NN search shared by its kernels (flh_kernels.hip: k_search_ring, k_search_exact;
// flh_pass.hip: k_pass, the searching pass as ONE launch).  gfx950 / wave64 only.
//
// ring_query<LPQ, RING, ...>: ONE query by a group of LPQ lanes over the (2 RING + 1)^3 block of cells around the query's cell.
// Each (y,z) row of the block is an x-run of consecutive local cells, i.e. ONE contiguous range of the cell-sorted map (two if
// the run crosses a brick boundary).  The 2 (2 RING + 1)^2 segment slots are resolved in parallel by the group's lanes (one
// directory probe, then ONE 16-byte read of the brick's prefix table for a ring-1 run), parked in LDS and prefix-summed, so that
// the group's T candidates form one flat list dealt round-robin to its lanes, eight independent loads in flight per lane.  One
// pass over the candidates: exact fp32 d2 (the oracle's op order) packed into a 32-bit key (the distance with its low PB mantissa
// bits replaced by the candidate's flat index) and kept in a SORTED TOP-8 per lane -- K0' = min(K0,t), Kj' = med3(K(j-1),Kj,t):
// eight VALU ops per candidate, no payload registers; the lanes' lists are merged over DPP with a bitonic half-cleaner + three-stage
// bitonic merge.  The packed keys only decide WHICH candidates can be among the five nearest (those whose key does not exceed the
// 5th's above the packed bits: at hand among the eight unless four neighbours agree to 2^-15 relative); the group loads those
// points, exchanges their exact (d2, map index) and every lane places its points at their exact rank.  A query is settled when
// its 5th distance lies within the block's guaranteed radius (RING + distance to the nearest face of the centre cell) * c and the
// packed keys left the set closed.
// Reference lines replaced: src/laserMapping.cpp:670-671 (ikdtree.Nearest_Search + the kNN gate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flh_device.hpp"

namespace flh {

typedef unsigned long long u64;
constexpr u64 kInfKey = ~0ull;

struct Top5 {
    u64 k[5];
    uint32_t p[5];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int j = 0; j < 5; ++j) { k[j] = kInfKey; p[j] = 0; }
    }
    __device__ __forceinline__ void insert(u64 key, uint32_t pos) {
        if (key < k[4] && (uint32_t)(key >> 32) < 0x7F800000u) {  // +inf distance = an empty storage slot: never a neighbour
            k[4] = key;
            p[4] = pos;
#pragma unroll
            for (int j = 4; j > 0; --j) {
                if (k[j] < k[j - 1]) {
                    const u64 tk = k[j]; k[j] = k[j - 1]; k[j - 1] = tk;
                    const uint32_t tp = p[j]; p[j] = p[j - 1]; p[j - 1] = tp;
                }
            }
        }
    }
};

__device__ __forceinline__ u64 make_key(float d, float w) {
    return ((u64)__float_as_uint(d) << 32) | (u64)__float_as_uint(w);
}

// ---- sorted top-8 of PACKED keys: the fp32 squared distance with its low PB mantissa bits replaced by the candidate's flat
// index inside the group's candidate list.  Positive floats order like their bit patterns, so the whole selection is
// unsigned-integer min / med3 (8 VALU ops per candidate, nothing else rides along).  Truncation keeps the order of any two
// candidates whose distances differ above the PB-th bit; a query whose best eight contain neighbours that agree there at the
// boundary is not settled by the packed keys (64-bit (d2, map index) keys order them exactly), so the packing never decides a
// result.
constexpr uint32_t kEmptyPacked = 0x7F000000u;  // above every real squared distance, below inf/nan patterns
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr int kTop = 8;  // packed keys kept per lane / per group: the five wanted + up to three to see ties at the boundary
__device__ __forceinline__ void insK(uint32_t (&K)[kTop], uint32_t t) {
    const uint32_t n0 = min(K[0], t);
    const uint32_t n1 = umed3(K[0], K[1], t);
    const uint32_t n2 = umed3(K[1], K[2], t);
    const uint32_t n3 = umed3(K[2], K[3], t);
    const uint32_t n4 = umed3(K[3], K[4], t);
    const uint32_t n5 = umed3(K[4], K[5], t);
    const uint32_t n6 = umed3(K[5], K[6], t);
    const uint32_t n7 = umed3(K[6], K[7], t);
    K[0] = n0; K[1] = n1; K[2] = n2; K[3] = n3; K[4] = n4; K[5] = n5; K[6] = n6; K[7] = n7;
}
__device__ __forceinline__ void cex2(uint32_t& a, uint32_t& b) {  // a <= b after
    const uint32_t lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32z(uint32_t v) {  // lanes without a source read 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// lowest eight of (mine U partner's), sorted: min(mine[j], partner's[7 - j]) are the eight smallest of the sixteen and
// form a bitonic sequence, which the three-stage bitonic merge sorts (tools/check_networks.py verifies it exhaustively
// with the 0/1 principle)
template <int CTRL>
__device__ __forceinline__ void merge8(uint32_t (&K)[kTop]) {
    uint32_t B[kTop];
#pragma unroll
    for (int j = 0; j < kTop; ++j) B[j] = dpp_u32<CTRL>(K[j]);
#pragma unroll
    for (int j = 0; j < kTop; ++j) K[j] = min(K[j], B[kTop - 1 - j]);
    cex2(K[0], K[4]); cex2(K[1], K[5]); cex2(K[2], K[6]); cex2(K[3], K[7]);
    cex2(K[0], K[2]); cex2(K[1], K[3]); cex2(K[4], K[6]); cex2(K[5], K[7]);
    cex2(K[0], K[1]); cex2(K[2], K[3]); cex2(K[4], K[5]); cex2(K[6], K[7]);
}
// the same with the partner at lane ^ XOR reached through the LDS crossbar (groups wider than a 16-lane DPP row)
template <int XOR>
__device__ __forceinline__ void merge8_xor(uint32_t (&K)[kTop]) {
    uint32_t B[kTop];
#pragma unroll
    for (int j = 0; j < kTop; ++j) B[j] = (uint32_t)__shfl_xor((int)K[j], XOR, 64);
#pragma unroll
    for (int j = 0; j < kTop; ++j) K[j] = min(K[j], B[kTop - 1 - j]);
    cex2(K[0], K[4]); cex2(K[1], K[5]); cex2(K[2], K[6]); cex2(K[3], K[7]);
    cex2(K[0], K[2]); cex2(K[1], K[3]); cex2(K[4], K[6]); cex2(K[5], K[7]);
    cex2(K[0], K[1]); cex2(K[2], K[3]); cex2(K[4], K[5]); cex2(K[6], K[7]);
}
template <int LPQ>
__device__ __forceinline__ void merge_group8(uint32_t (&K)[kTop]) {
    if (LPQ >= 2) merge8<0xB1>(K);    // quad_perm [1,0,3,2]
    if (LPQ >= 4) merge8<0x4E>(K);    // quad_perm [2,3,0,1]
    if (LPQ >= 8) merge8<0x141>(K);   // row_half_mirror
    if (LPQ >= 16) merge8<0x140>(K);  // row_mirror
    if (LPQ >= 32) merge8_xor<16>(K);
    if (LPQ >= 64) merge8_xor<32>(K);
}

// value of lane `src` of the LPQ-lane query group (src is a compile-time constant at every call site after unrolling)
template <int LPQ>
__device__ __forceinline__ float group_bcast(float v, int src) {
    if (LPQ == 1) return v;
    if (LPQ == 2) {
        const int x = __float_as_int(v);
        const int a = __builtin_amdgcn_update_dpp(0, x, 0xA0, 0xF, 0xF, false);  // quad_perm [0,0,2,2]
        const int b = __builtin_amdgcn_update_dpp(0, x, 0xF5, 0xF, 0xF, false);  // quad_perm [1,1,3,3]
        return __int_as_float(src == 0 ? a : b);
    }
    if (LPQ == 4) {
        const int x = __float_as_int(v);
        const int a = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, false);  // quad_perm [0,0,0,0]
        const int b = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, false);  // [1,1,1,1]
        const int c = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, false);  // [2,2,2,2]
        const int d = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xF, 0xF, false);  // [3,3,3,3]
        return __int_as_float(src == 0 ? a : (src == 1 ? b : (src == 2 ? c : d)));
    }
    return __shfl(v, src, LPQ);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 load_pt(__amdgpu_buffer_rsrc_t rsrc, uint32_t idx) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(idx << 4), 0, 0);  // out of range -> zeros
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// c ? a : b as one v_cndmask (the compiler turns chains of ?: on loaded values into jump trees)
__device__ __forceinline__ uint32_t sel_u32(bool c, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(__builtin_amdgcn_ballot_w64(c)));
    return r;
}
__device__ __forceinline__ u32x3 load_xyz(__amdgpu_buffer_rsrc_t rsrc, uint32_t idx) {  // coordinates only (12 of the 16 B)
    return __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(idx << 4), 0, 0);
}

// A query group never spans a wave, so the LDS hand-offs inside ring_query only need wave-level ordering:
// LDS operations of one wave complete in order; this keeps the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Where a query's result goes besides the neighbour cache in HBM.  park != nullptr (k_pass): the five neighbours' coordinates
// are also left in LDS (park[3 r + {0,1,2}], r = rank) for the fit that follows in the same workgroup, and the verdict of the
// kNN gate (src/laserMapping.cpp:671) goes to park[kParkStatus] instead of selected[]: the fit writes the final flag.
constexpr int kParkWorld = 15, kParkBody = 18, kParkStatus = 21, kParkUb = 22, kParkStride = 25;  // floats per query (odd stride: conflict-free)
constexpr uint32_t kStIdle = 0, kStFit = 1, kStNoFit = 2, kStOpen = 3;  // not a query of this rank / passes the gate / fails it / not settled yet

// The directory, the brick tables and the points through buffer resources: SGPR base + 32-bit VGPR offset, no 64-bit address
// arithmetic; out-of-range point reads return zeros.  (rows * 320 bytes and points * 16 bytes stay below 2^32: flh_api.cpp checks
// both when the index is built or grows.)
struct RingRsrc {
    __amdgpu_buffer_rsrc_t pts, hash, tab;
    const u64* hash64;
    __device__ __forceinline__ RingRsrc(const GridParams& g, uint32_t map_points) {
        pts = __builtin_amdgcn_make_buffer_rsrc((void*)g.pts, 0, (int)(map_points * 16u), 0x00020000);
        hash = __builtin_amdgcn_make_buffer_rsrc((void*)g.hash, 0, (int)((g.hash_mask + 1u) * 8u), 0x00020000);
        tab = __builtin_amdgcn_make_buffer_rsrc((void*)g.starts, 0, -1, 0x00020000);
        hash64 = reinterpret_cast<const u64*>(g.hash);
    }
};

// One row of a query's result in the neighbour cache.  nn_idx != nullptr (flh_config.index_cache, the one-launch pass): only the
// neighbour's map INDEX is kept (4 B instead of 16; -1 = none) -- no pass reads the coordinates from HBM again (the fit that
// follows has them in LDS, later passes take the cached plane), whoever else wants them gathers them by index (k_nn_gather).
__device__ __forceinline__ void nn_store(float4* __restrict__ nn_pts, uint32_t* __restrict__ nn_idx, size_t at, const float4& v) {
    if (nn_idx) nn_idx[at] = __float_as_uint(v.w);
    else nn_pts[at] = v;
}

// Group-wide merge of the lanes' sorted (d2, map index) lists and the query's result rows: 5 x (min butterfly, ballot, pop).
template <int LPQ>
__device__ __forceinline__ void top5_finish(Top5& L, const GridParams& g, int q, int N, int lane, float max_sqdist,
                                            float4* __restrict__ nn_pts, uint8_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected,
                                            float* park, uint32_t* __restrict__ nn_idx = nullptr) {
    const int wl0 = (threadIdx.x & 63) & ~(LPQ - 1);
    const u64 gmask = (LPQ == 64 ? ~0ull : ((1ull << LPQ) - 1ull)) << wl0;
    u64 rk[5];
    uint32_t rp[5];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        u64 m = L.k[0];
#pragma unroll
        for (int off = LPQ / 2; off >= 1; off >>= 1) {
            const u64 o = __shfl_xor(m, off, LPQ);
            m = o < m ? o : m;
        }
        const bool win = (L.k[0] == m) && (m != kInfKey);
        const u64 bal = __ballot(win) & gmask;
        const int wl = bal ? (__ffsll((long long)bal) - 1) : wl0;
        const uint32_t wp = __shfl(L.p[0], wl, 64);
        rk[j] = m;
        rp[j] = wp;
        if (m != kInfKey) ++cnt;
        if (win) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { L.k[t] = L.k[t + 1]; L.p[t] = L.p[t + 1]; }
            L.k[4] = kInfKey;
        }
    }
    const float d5 = (cnt == 5) ? __uint_as_float((uint32_t)(rk[4] >> 32)) : INFINITY;
#pragma unroll
    for (int r = 0; r < (5 + LPQ - 1) / LPQ; ++r) {  // lane l writes ranks l, l + LPQ, ...
        const int jr = lane + r * LPQ;
        if (jr < 5) {
            uint32_t pp = rp[0];
#pragma unroll
            for (int j = 1; j < 5; ++j)
                if (jr == j) pp = rp[j];
            const bool has = jr < cnt;
            float4 v = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (has) v = g.pts[FLH_IDX(10, pp, g.pts_cap)];
            nn_store(nn_pts, nn_idx, FLH_IDX(11, (size_t)jr * N + q, (size_t)5 * N), v);
            if (park) { park[3 * jr] = v.x; park[3 * jr + 1] = v.y; park[3 * jr + 2] = v.z; }
        }
    }
    if (lane == 0) {
        nn_cnt[q] = (uint8_t)cnt;
        const bool gate = cnt == 5 && !(d5 > max_sqdist);  // laserMapping.cpp:671
        if (park) park[kParkStatus] = __uint_as_float(gate ? kStFit : kStNoFit);
        else selected[q] = gate ? 1 : 0;
    }
}

// The general exact search of ONE query by a group of LPQ lanes: one pass over the cells that intersect the ball of
// radius sqrt(ub) around the query (ub = an upper bound of its true 5th squared distance, capped by the gate max_sqdist
// of src/laserMapping.cpp:671 -- beyond the gate a result can never be selected).  64-bit keys
// (d2 bits << 32 | map index) give the oracle's (d2, index) order, ties included.  Writes the query's result rows.
template <int LPQ>
__device__ __forceinline__ uint32_t exact_query(const GridParams& g, int q, int N, float qx, float qy, float qz, int cx, int cy,
                                                int cz, float fx, float fy, float fz, float ub, int rmax, float max_sqdist,
                                                int lane, float4* __restrict__ nn_pts, uint8_t* __restrict__ nn_cnt,
                                                uint8_t* __restrict__ selected, float* park, uint32_t* __restrict__ nn_idx = nullptr) {
    // cells with |offset| <= R cover the ball; +1 absorbs the position inside the centre cell
    const int r = min(rmax, (int)(sqrtf(ub) * g.inv_c) + 1);
    const float ubp = ub * 1.0001f + 1e-6f;
    Top5 L;
    L.reset();
    uint32_t ncand = 0;
    const int side = 2 * r + 1;
    const int side2 = side * side;
    const int ncell = side2 * side;
    for (int t = lane; t < ncell; t += LPQ) {
        const int iz = t / side2;
        const int rem = t - iz * side2;
        const int iy = rem / side;
        const int dx = rem - iy * side - r, dy = iy - r, dz = iz - r;
        // lower bound of the distance from the query to this cell's box; skip cells outside the ball
        const float gx = dx > 0 ? (float)dx - fx : (dx < 0 ? fx - (float)(dx + 1) : 0.f);
        const float gy = dy > 0 ? (float)dy - fy : (dy < 0 ? fy - (float)(dy + 1) : 0.f);
        const float gz = dz > 0 ? (float)dz - fz : (dz < 0 ? fz - (float)(dz + 1) : 0.f);
        const float lb = ((gx * gx + gy * gy) + gz * gz) * (g.c * g.c) * 0.995f - 1e-5f;
        if (lb > ubp) continue;
        const uint2 e = lookup_cell(g, cx + dx, cy + dy, cz + dz);
        ncand += e.y;
        for (uint32_t i = e.x; i < e.x + e.y; ++i) {
            const float4 pv = g.pts[FLH_IDX(14, i, g.pts_cap)];
            L.insert(make_key(dist2(qx, qy, qz, pv.x, pv.y, pv.z), pv.w), i);
        }
    }
    top5_finish<LPQ>(L, g, q, N, lane, max_sqdist, nn_pts, nn_cnt, selected, park, nn_idx);
    return ncand;
}

// LDS a query group needs: its segment table (2 (2 RING + 1)^2 segments + sentinel + one slot the walk's look-ahead may touch)
template <int RING>
constexpr int ring_seg_slots() { return 2 * (2 * RING + 1) * (2 * RING + 1) + 2; }

// ONE query by the LPQ lanes of its group.  Returns true when the query's result rows were written (settled).
//   BOUNDED   the query comes with an upper bound ub_raw of its true 5th squared distance (found by a smaller ring); rows and row
//             ends that lie entirely outside that ball are not visited
//   FINAL     a query the packed keys cannot settle is finished on the spot by its group (64-bit keys over the same candidates;
//             with EXACT also the general exact search for a 5th neighbour beyond the block); otherwise the caller lists it for
//             the next stage with the bound ub_next
//   EXACT     false where the block provably covers the gate radius (ring 2 with cells >= sqrt(max_sqdist) / 1.998): the general
//             search is compiled out
//   WIN4      (RING == 2, BOUNDED) the rows of a 4x4 window instead of the 5x5 block's: the ball of a bounded query reaches two
//             cells below the centre cell or two above it on an axis, never both, as long as its radius stays below 1.5 cells
//             (cells of 1.5 m and the gate of sqrt(5) m: 1.4983; the caller checks) -- 32 segment slots instead of 50
//   park      see above (nullptr: results to HBM only)
template <int LPQ, int RING, bool BOUNDED, int PB, bool FINAL, bool EXACT, bool WIN4 = false, int UNR = 8>
__device__ __forceinline__ bool ring_query(const GridParams& g, const RingRsrc& rs, uint2* __restrict__ seg /* LDS, this group's */,
                                           int lane, int q, int N, bool live, float qx, float qy, float qz, float ub_raw,
                                           float max_sqdist, int rmax, float4* __restrict__ nn_pts, uint8_t* __restrict__ nn_cnt,
                                           uint8_t* __restrict__ selected, u64* __restrict__ cand_counter, float* park,
                                           float& ub_next, uint32_t* __restrict__ nn_idx = nullptr) {
    static_assert(!WIN4 || (RING == 2 && BOUNDED), "the 4x4 window is for bounded ring-2 queries");
    constexpr int W = WIN4 ? 4 : 2 * RING + 1;     // rows per axis
    constexpr int NR = W * W;                      // (y,z) rows: each an x-run of at most 2 RING + 1 consecutive cells
    constexpr int NSEG = NR * 2;                   // a run crosses at most one brick boundary -> two segments
    constexpr int SPL = (NSEG + LPQ - 1) / LPQ;    // segments resolved per lane
    // UNR: independent point loads in flight per lane
    int cx, cy, cz;
    float fx, fy, fz;
    cell_of(g, qx, qy, qz, cx, cy, cz, fx, fy, fz);
    const float minfrac = fminf(fminf(fminf(fx, 1.f - fx), fminf(fy, 1.f - fy)), fminf(fz, 1.f - fz));

    // ---- phase 1: directory probes of this lane's segments
    float ubq = INFINITY;
    if (BOUNDED) ubq = fminf(ub_raw, max_sqdist) * 1.0001f + 1e-6f;
    const float inv_c2 = g.inv_c * g.inv_c;
    // WIN4: the window starts two cells below the centre cell on an axis where the ball reaches that far (gap f + 1), else one
    const float rcell2 = ubq * inv_c2 * 1.01f + 1e-4f;  // squared radius of the ball in cells, with the margin the row test uses
    const int wy0 = WIN4 ? (((fy + 1.f) * (fy + 1.f) <= rcell2) ? 2 : 1) : RING;
    const int wz0 = WIN4 ? (((fz + 1.f) * (fz + 1.f) <= rcell2) ? 2 : 1) : RING;
    uint32_t key[SPL], i0[SPL], i1[SPL];
    u64 he[SPL];
#pragma unroll
    for (int u = 0; u < SPL; ++u) {
        const int sl = lane + u * LPQ;
        const int half = sl / NR, r = sl - half * NR;  // slots [0,NR): first segments, [NR,2NR): second (split rows)
        const int rz = r / W, ry = r - rz * W;
        const int dy = ry - wy0, dz = rz - wz0;
        const int y = cy + dy, z = cz + dz;
        int xlo = cx - RING, xhi = cx + RING;
        bool inball = true;
        if (BOUNDED) {
            // distance (in cells) from the query to the row's (y,z) slab; what is left of the ball bounds x
            const float gy = dy > 0 ? (float)dy - fy : (dy < 0 ? fy - (float)(dy + 1) : 0.f);
            const float gz = dz > 0 ? (float)dz - fz : (dz < 0 ? fz - (float)(dz + 1) : 0.f);
            const float rem = ubq * inv_c2 * 1.01f + 1e-4f - (gy * gy + gz * gz);
            inball = rem >= 0.f;
            const float gmax = sqrtf(fmaxf(rem, 0.f));
            xlo = max(xlo, cx - (int)(gmax - fx + 1.f));  // dx < 0: gap = fx - (dx + 1) <= gmax
            xhi = min(xhi, cx + (int)(gmax + fx));        // dx > 0: gap = dx - fx       <= gmax
        }
        const int x0 = max(xlo, 0), x1 = min(xhi, g.nx - 1);
        const bool split = (x0 >> 2) != (x1 >> 2);
        bool valid = (sl < NSEG) && inball && x0 <= x1 && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz;
        int xa, xb;
        if (half == 0) { xa = x0; xb = split ? (x0 | 3) : x1; }
        else { xa = x1 & ~3; xb = x1; valid = valid && split; }
        key[u] = brick_key(xa, y, z);
        i0[u] = cell_local(xa, y, z);
        i1[u] = cell_local(xb, y, z) + 1;
        {
            const u32x2 e_ = __builtin_amdgcn_raw_buffer_load_b64(rs.hash, (int)(valid ? hash_slot(key[u], g.hash_shift) * 8u : 0u), 0, 0);
            he[u] = valid ? (((u64)e_.y << 32) | (u64)e_.x) : (u64)kEmptyKey;
        }
    }
    // ---- phase 2: directory entries -> point ranges.  Collisions first (rare, one branch for all segments), then
    // every prefix-table read of the lane in one batch: a miss reads brick 0's table and is masked afterwards.
    {
        bool coll = false;
#pragma unroll
        for (int u = 0; u < SPL; ++u) coll = coll || ((uint32_t)he[u] != key[u] && (uint32_t)he[u] != kEmptyKey);
        if (coll) {
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                u64 e = he[u];
                uint32_t sl_ = hash_slot(key[u], g.hash_shift);
                while ((uint32_t)e != key[u] && (uint32_t)e != kEmptyKey) {  // linear probing
                    sl_ = (sl_ + 1) & g.hash_mask;
                    e = rs.hash64[sl_];
                }
                he[u] = e;
            }
        }
    }
    uint32_t la[SPL], nseg[SPL];
    if (RING == 1) {
        // a ring-1 run covers at most three cells of one brick row, so its start and its end sit within four consecutive
        // table entries: ONE 16-byte load per segment (the table rows are padded so that a z-slab's 16 entries share
        // a 64-byte line -- the three rows of one z that neighbouring lanes resolve hit the same line)
        u32x4 tb[SPL];
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const bool hit = (uint32_t)he[u] == key[u];
            const uint32_t row = hit ? (uint32_t)(he[u] >> 32) : 0u;
#ifdef FLH_BOUNDS
            (void)FLH_IDX(6, row, g.rows_cap);
            (void)FLH_IDX(7, i0[u] + 3u, (unsigned)kBrickStride);
#endif
            tb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs.tab, (int)((row * (uint32_t)kBrickStride + i0[u]) * 4u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const bool hit = (uint32_t)he[u] == key[u];
            const uint32_t len = i1[u] - i0[u];  // 1..3 cells
            la[u] = tb[u].x;
            uint32_t lb = tb[u].w;  // two selects (written as a chain of ?: the compiler lowers it to a jump tree)
            lb = sel_u32(len == 2u, tb[u].z, lb);
            lb = sel_u32(len == 1u, tb[u].y, lb);
            nseg[u] = hit ? min(lb - la[u], 1u << 18) : 0u;  // cap: keeps the packed sums below exact (> 2^PB is unsettled anyway)
        }
    } else {
        uint32_t lb[SPL];
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const bool hit = (uint32_t)he[u] == key[u];
            const uint32_t row = hit ? (uint32_t)(he[u] >> 32) : 0u;
#ifdef FLH_BOUNDS
            (void)FLH_IDX(8, row, g.rows_cap);
            (void)FLH_IDX(9, i1[u], 65u);
#endif
            la[u] = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (int)((row * (uint32_t)kBrickStride + i0[u]) * 4u), 0, 0);
            lb[u] = __builtin_amdgcn_raw_buffer_load_b32(rs.tab, (int)((row * (uint32_t)kBrickStride + i1[u]) * 4u), 0, 0);
        }
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const bool hit = (uint32_t)he[u] == key[u];
            nseg[u] = hit ? min(lb[u] - la[u], 1u << 18) : 0u;
        }
    }
    // ---- flat candidate list of the group: exclusive prefix over its segments, lane-major (lane 0's segments, then lane 1's,
    // ...; any fixed order will do, the flat index only names a candidate).  In registers: the lane's own running sum, one DPP
    // scan of the lanes' totals.  Empty segments are dropped, so the table holds (first point - flat start, flat end) of the
    // non-empty ones, then a sentinel.
    uint32_t T = 0;
    {
        uint32_t loc[SPL], own = 0;  // bits 0..23: candidates, 24..31: non-empty segments
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            loc[u] = own;
            own += nseg[u] | (nseg[u] ? (1u << 24) : 0u);
        }
        uint32_t inc = own;
        if (LPQ >= 2) { const uint32_t up = dpp_u32z<0x111>(inc); inc += (lane >= 1) ? up : 0u; }
        if (LPQ >= 4) { const uint32_t up = dpp_u32z<0x112>(inc); inc += (lane >= 2) ? up : 0u; }
        if (LPQ >= 8) { const uint32_t up = dpp_u32z<0x114>(inc); inc += (lane >= 4) ? up : 0u; }
        if (LPQ >= 16) { const uint32_t up = dpp_u32z<0x118>(inc); inc += (lane >= 8) ? up : 0u; }
        if (LPQ >= 32) {  // groups wider than a 16-lane DPP row: add the totals of the rows before this lane's
            uint32_t before = 0;
#pragma unroll
            for (int r = 1; r < LPQ / 16; ++r) {
                const uint32_t row_total = (uint32_t)__shfl((int)inc, 16 * r - 1, LPQ);  // inclusive value at the row's last lane
                before += (lane >= 16 * r) ? row_total : 0u;
            }
            inc += before;
        }
        uint32_t tot;
        if (LPQ == 1) tot = inc;
        else if (LPQ == 2) tot = dpp_u32<0xF5>(inc);   // quad_perm [1,1,3,3]
        else if (LPQ == 4) tot = dpp_u32<0xFF>(inc);   // quad_perm [3,3,3,3]
        else tot = (uint32_t)__shfl((int)inc, LPQ - 1, LPQ);
        const uint32_t base_ = inc - own;
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const uint32_t ex = base_ + loc[u];
            const uint32_t exT = ex & 0xFFFFFFu;
            if (nseg[u]) seg[FLH_IDX(15, ex >> 24, NSEG + 1)] = make_uint2(la[u] - exT, exT + nseg[u]);
        }
        T = tot & 0xFFFFFFu;
        if (lane == 0) seg[FLH_IDX(16, tot >> 24, NSEG + 1)] = make_uint2(0u, 0xFFFFFFFFu);  // sentinel: the walk never runs off the end
    }
    wave_sync();
    // ---- one pass over the candidates: the group's T candidates are dealt round-robin to its lanes
    constexpr uint32_t PMASK = (1u << PB) - 1u;
    uint32_t K[kTop];
#pragma unroll
    for (int j = 0; j < kTop; ++j) K[j] = kEmptyPacked;
    {
        int cur = 0;
        uint2 sg = seg[0];
        uint2 nx = seg[1];  // the entry after the current one is always in flight before it is needed
        for (uint32_t t0 = lane; t0 < T; t0 += LPQ * UNR) {
            u32x3 v[UNR];
#pragma unroll
            for (int w = 0; w < UNR; ++w) {
                const uint32_t t = t0 + (uint32_t)(w * LPQ);
                if (t >= sg.y) {  // one step is the common case; its look-ahead read is not waited for here
                    sg = nx; ++cur; nx = seg[cur + 1];
                    while (t >= sg.y) { sg = nx; ++cur; nx = seg[cur + 1]; }
                }
#ifdef FLH_BOUNDS
                (void)FLH_IDX(1, cur + 1, NSEG + 2);
                if (t < T) (void)FLH_IDX(2, sg.x + t, g.pts_cap);
#endif
                v[w] = load_xyz(rs.pts, (t < T) ? sg.x + t : 0xFFFFFFFu);  // past the end: out-of-range -> zeros, masked below
            }
#pragma unroll
            for (int w = 0; w < UNR; ++w) {
                const uint32_t t = t0 + (uint32_t)(w * LPQ);
                const float d = dist2(qx, qy, qz, __uint_as_float(v[w].x), __uint_as_float(v[w].y), __uint_as_float(v[w].z));
                const uint32_t key_ = (__float_as_uint(d) & ~PMASK) | (t & PMASK);
                insK(K, (t < T) ? key_ : kEmptyPacked);
            }
        }
    }
    merge_group8<LPQ>(K);
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) cnt += (K[j] < kEmptyPacked) ? 1 : 0;
    // The packed order decides WHICH candidates can be among the five nearest: with v = the 5th key above the packed
    // bits, every candidate whose key exceeds v there is farther than five others, and those at or below v are all
    // among the best eight as long as the 8th key is above v.  Their exact (d2, map index) -- known once their points
    // are loaded -- then picks and orders the five.  Only an 8th key at v (four neighbours within 2^-15 relative, or
    // equal distances) or a list longer than the packed index can name leaves the set open: next stage / 64-bit keys.
    const uint32_t v5 = K[4] >> PB;
    int m = cnt;  // candidates to load: the found ones among the first five + the ties of the fifth
#pragma unroll
    for (int j = 5; j < kTop - 1; ++j) m += (cnt == 5 && K[j] < kEmptyPacked && (K[j] >> PB) == v5) ? 1 : 0;
    const bool amb = T > PMASK + 1u || (cnt == 5 && K[kTop - 1] < kEmptyPacked && (K[kTop - 1] >> PB) == v5);
    const float d5hi = (cnt == 5) ? __uint_as_float(K[4] | PMASK) : INFINITY;  // >= the true 5th distance
    const float gr = ((float)RING + minfrac) * g.c - 2e-3f * g.c;  // guaranteed-complete radius (fp margin)
    const float gr2 = gr * gr;
    const bool covered = (cnt == 5 && d5hi <= gr2) || gr2 >= max_sqdist;
    const bool done = !amb && covered;
    if (cand_counter && live && lane == 0) atomicAdd(cand_counter, (u64)T);
#ifdef FLH_PASS_STAMPS
    if (park && lane == 0 && RING == 1) park[23] = (float)T;  // developer build: the length of the candidate list (tools/pass_stamps.py)
#endif
    // ---- results: lane l loads ranks l, l + LPQ, ... (< m): flat index -> map position (table walk) -> one point load
    // each, all issued before the first is consumed; the exact squared distance is recomputed from the point (same
    // formula, same bits as the scan saw before packing).  The group then exchanges the (d2, map index) pairs and every
    // lane places its points at their exact rank; ranks beyond the fifth are dropped.
    if (done && live) {
        constexpr int NL = kTop - 1;                 // at most seven candidates are loaded
        constexpr int RPL = (NL + LPQ - 1) / LPQ;    // per lane
        float4 pv[RPL];
        float dv[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int j = lane + r * LPQ;
            uint32_t kj = K[0];
#pragma unroll
            for (int jj = 1; jj < NL; ++jj) kj = (j == jj) ? K[jj] : kj;
            pv[r] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (j < m) {
                const uint32_t t = kj & PMASK;
                int c2 = 0;
                uint2 s2 = seg[0];
                while (t >= s2.y) s2 = seg[++c2];
#ifdef FLH_BOUNDS
                (void)FLH_IDX(3, c2, NSEG + 1);
                (void)FLH_IDX(4, s2.x + t, g.pts_cap);
#endif
                pv[r] = load_pt(rs.pts, s2.x + t);
            }
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r)
            dv[r] = (lane + r * LPQ < m) ? dist2(qx, qy, qz, pv[r].x, pv[r].y, pv[r].z) : INFINITY;
        // every lane sees all (d2, id): candidate i lives in slot i / LPQ of lane i % LPQ
        float da[NL];
        uint32_t ia[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            da[i] = group_bcast<LPQ>(dv[i / LPQ], i % LPQ);
            ia[i] = __float_as_uint(group_bcast<LPQ>(pv[i / LPQ].w, i % LPQ));
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int j = lane + r * LPQ;
            if (j < NL) {
                int e = j;  // rows past the found ones (j >= m, only when fewer than five were found) are written empty
                if (j < m) {  // exact rank among the loaded ones (ids are distinct: a strict total order)
                    const uint32_t myid = __float_as_uint(pv[r].w);
                    e = 0;
#pragma unroll
                    for (int i = 0; i < NL; ++i)
                        e += (i < m && (da[i] < dv[r] || (da[i] == dv[r] && ia[i] < myid))) ? 1 : 0;
                }
                if (e < 5) {
                    nn_store(nn_pts, nn_idx, FLH_IDX(5, (size_t)e * N + q, (size_t)5 * N), pv[r]);  // the squared distances are not stored: k_fill_d2 recomputes them on demand
                    if (park) { park[3 * e] = pv[r].x; park[3 * e + 1] = pv[r].y; park[3 * e + 2] = pv[r].z; }
                    if (e == 4) {
                        const bool gate = j < m && !(dv[r] > max_sqdist);  // laserMapping.cpp:671
                        if (park) park[kParkStatus] = __uint_as_float(gate ? kStFit : kStNoFit);
                        else selected[q] = gate ? 1 : 0;
                    }
                }
            }
        }
        if (lane == 0) nn_cnt[q] = (uint8_t)cnt;
    }
    bool done2 = done;
    if (FINAL) {
        // Last stage only.  The block provably holds the five nearest (5th distance inside the guaranteed radius) but
        // the packed keys left the set open (5th and 6th agree above the packed bits -- equal distances included -- or
        // the list is longer than the packed index can name): one more pass over the SAME candidates with 64-bit
        // (d2, map index) keys settles it.
        if (live && !done && covered) {
            Top5 L;
            L.reset();
            int c2 = 0;
            uint2 s2 = seg[0];
            constexpr int RU = EXACT ? 8 : 4;  // loads in flight per lane (the rarer path: fewer registers where the kernel is tight)
            for (uint32_t t0 = lane; t0 < T; t0 += LPQ * RU) {
                float4 pv[RU];
                uint32_t pos[RU];
#pragma unroll
                for (int w = 0; w < RU; ++w) {
                    const uint32_t t = t0 + (uint32_t)(w * LPQ);
                    while (t >= s2.y) s2 = seg[++c2];
#ifdef FLH_BOUNDS
                    (void)FLH_IDX(12, c2, NSEG + 1);
                    if (t < T) (void)FLH_IDX(13, s2.x + t, g.pts_cap);
#endif
                    pos[w] = (t < T) ? s2.x + t : 0xFFFFFFFu;
                    pv[w] = load_pt(rs.pts, pos[w]);
                }
#pragma unroll
                for (int w = 0; w < RU; ++w)
                    if (t0 + (uint32_t)(w * LPQ) < T)
                        L.insert(make_key(dist2(qx, qy, qz, pv[w].x, pv[w].y, pv[w].z), pv[w].w), pos[w]);
            }
            top5_finish<LPQ>(L, g, q, N, lane, max_sqdist, nn_pts, nn_cnt, selected, park, nn_idx);
            done2 = true;
        }
        if (EXACT && live && !done2) {
            const float ubx = fminf(BOUNDED ? fminf(d5hi, ub_raw) : d5hi, max_sqdist);
            const uint32_t nc = exact_query<LPQ>(g, q, N, qx, qy, qz, cx, cy, cz, fx, fy, fz, ubx, rmax, max_sqdist, lane, nn_pts,
                                                 nn_cnt, selected, park, nn_idx);
            if (cand_counter && lane == 0) atomicAdd(cand_counter, (u64)nc);
            done2 = true;
        }
    }
    ub_next = BOUNDED ? fminf(d5hi, ub_raw) : d5hi;  // the true 5th distance is <= either bound
    return done2;
}

}  // namespace flh
