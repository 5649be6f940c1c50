// flh_kernels.hip -- the HIP kernels of the hot path, written for gfx950 (CDNA4, wave64) only.
//
//   K0  map index build     keys -> (radix sort) -> bricks, per-brick cell tables, directory            [setup]
//   A1  k_search_ring<4,1>  body->world transform + 5-NN over the 3x3x3 cell block, 4 lanes per query    [search passes]
//   A2  k_search_ring<16,2> the queries A1 could not settle: 5x5x5 block inside A1's bound, 16 lanes per
//                           query, finishing leftovers itself with the general exact search (exact_query)
//   A3  k_search_exact      the general exact search as a kernel (lanes_per_query = 0, grids without ring 2)
//   B   k_fit<ORD,HALF>     plane fit + residual gate + 12-col Jacobian row + 16x16 Gram contraction on
//                           v_mfma_f64_16x16x4_f64 + deterministic cross-block sum                       [every pass]
//   S   k_scan_restride / k_scan_keys / k_scan_gather   scan staging (records -> float4 + Morton key -> order)
//
// Reference lines replaced: src/laserMapping.cpp:650-693 (A1-A3, B), :695-752 + esekfom.hpp:1784,1804 (B).
// Built with -ffp-contract=off (see flh_device.hpp).
#include "flh_kernels.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "flh_device.hpp"

namespace flh {

typedef unsigned long long u64;
constexpr u64 kInfKey = ~0ull;
constexpr int kStripes = 64;  // work-list stripes (one counter + one list segment each)

// ------------------------------------------------------------------------------------------------
// K0: map index: points keyed by (brick, local cell), sorted, laid out per brick with slack behind its points; per brick a
// prefix table of absolute storage positions; an open-addressing directory brick key -> brick rank.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_map_keys(GridParams g, const float4* __restrict__ pts, uint32_t M,
                                                  u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const float4 p = pts[i];
    int cx, cy, cz;
    float fx, fy, fz;
    cell_of(g, p.x, p.y, p.z, cx, cy, cz, fx, fy, fz);
    // the host sized the grid from the exact AABB with padding, so these clamps never fire for finite input
    cx = min(max(cx, 0), g.nx - 1);
    cy = min(max(cy, 0), g.ny - 1);
    cz = min(max(cz, 0), g.nz - 1);
    keys[i] = ((u64)brick_key(cx, cy, cz) << 6) | (u64)cell_local(cx, cy, cz);
    vals[i] = i;
}

// brick_start[rank] = first sorted position of the brick (brick_start[nbricks] = M is written by the host)
__global__ void __launch_bounds__(256) k_brick_starts(const uint32_t* __restrict__ brick_head,
                                                      const uint32_t* __restrict__ brick_rank_incl, uint32_t M,
                                                      uint32_t* __restrict__ brick_start) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    if (brick_head[i]) brick_start[brick_rank_incl[i] - 1] = i;
}

// One thread per (brick, local cell 0..64): starts[b*65 + l] = lower_bound of key (bkey<<6 | l) inside the
// brick's run of sorted keys, so that any run of consecutive local cells (an x-row segment) maps to ONE
// contiguous point range.  Thread l == 0 also inserts (bkey -> rank) into the open-addressing directory.
__global__ void __launch_bounds__(256) k_brick_tables(const u64* __restrict__ keys_sorted,
                                                      const uint32_t* __restrict__ brick_start, uint32_t nbricks,
                                                      const uint32_t* __restrict__ cap_incl, const uint32_t* __restrict__ cap,
                                                      uint32_t* __restrict__ starts, uint32_t* __restrict__ cap_end,
                                                      uint32_t* __restrict__ live, uint2* __restrict__ hash,
                                                      uint32_t hash_mask, int hash_shift) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nbricks * (uint32_t)kBrickStride) return;
    const uint32_t rank = t / kBrickStride, l = t - rank * kBrickStride;
    if (l > 64) return;  // padding of the row
    const uint32_t b0 = brick_start[rank], b1 = brick_start[rank + 1];
    const u64 bkey = keys_sorted[b0] >> 6;
    const u64 target = (bkey << 6) + l;  // l == 64 -> first key of the next brick value
    uint32_t lo = b0, hi = b1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < target) lo = mid + 1; else hi = mid;
    }
    // the brick's points live at [base, base + count) of the storage, followed by its slack up to cap_end
    const uint32_t base = cap_incl[rank] - cap[rank];
    starts[t] = base + (lo - b0);
    if (l == 0) {
        cap_end[rank] = base + cap[rank];
        live[rank] = b1 - b0;
        const uint32_t k32 = (uint32_t)bkey;
        uint32_t slot = hash_slot(k32, hash_shift);
        for (;;) {
            const uint32_t prev = atomicCAS(&hash[slot].x, kEmptyKey, k32);
            if (prev == kEmptyKey) {
                hash[slot].y = rank;
                break;
            }
            slot = (slot + 1) & hash_mask;
        }
    }
}

// Storage layout with slack: brick r owns [base_r, base_r + cap_r), cap_r = count_r + max(8, count_r / 4), so that points can
// be inserted into a brick (flh_mapinc.hip: k_brick_rewrite) without moving any other brick.
__global__ void __launch_bounds__(256) k_brick_heads(const u64* __restrict__ keys_sorted, uint32_t M, uint32_t* __restrict__ bh) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    bh[i] = (i == 0 || (keys_sorted[i - 1] >> 6) != (keys_sorted[i] >> 6)) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_brick_caps(const uint32_t* __restrict__ brick_start, uint32_t nbricks,
                                                    uint32_t* __restrict__ cap) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nbricks) return;
    const uint32_t cnt = brick_start[r + 1] - brick_start[r];
    cap[r] = cnt + max(8u, cnt >> 2);
}
__global__ void __launch_bounds__(256) k_fill_tomb(float4* __restrict__ pts, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pts[i] = tombstone();
}
__global__ void __launch_bounds__(256) k_map_place(const float4* __restrict__ pts, const uint32_t* __restrict__ vals_sorted,
                                                   const uint32_t* __restrict__ brick_rank_incl, const uint32_t* __restrict__ brick_start,
                                                   const uint32_t* __restrict__ cap_incl, const uint32_t* __restrict__ cap, uint32_t M,
                                                   float4* __restrict__ out) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const uint32_t src = vals_sorted[j];
    float4 p = pts[src];
    p.w = __uint_as_float(src);  // identity of the point = its position in the index-ordered array
    const uint32_t r = brick_rank_incl[j] - 1;
    out[(cap_incl[r] - cap[r]) + (j - brick_start[r])] = p;
}

// ------------------------------------------------------------------------------------------------
// A: exact 5-NN: a first stage over every query, a second stage over the queries the first could not settle.
//
// A1 k_search_ring<4,1>     every query, the 3x3x3 cells around its cell: four lanes per query gather their candidates through
//                            18 row segments; settles every query whose 5th neighbour is provably inside that block and free
//                            of near-ties.  (An LDS-tile variant -- a block, later a wave, of Morton-neighbouring queries sharing
//                            one tile of the map -- was built, validated bit for bit and measured 1.6-2.2x SLOWER in three
//                            versions; see profiles/r03_wtile_experiment/ and DESIGN.md for why.  It is not in the product.)
// A2 k_search_ring<16,2>     the queries A1 listed: 5x5x5 cells clipped to the ball of A1's 5th distance; whatever it cannot
//                            settle either (a true tie, a list longer than the packed index can name) it finishes itself with
//                            64-bit (d2, map index) keys, and a 5th neighbour beyond its block with the general search
// A3 k_search_exact          the general exact search as a kernel: lanes_per_query = 0 (the tests' cross-check), grids whose
//                            cells are as large as the gate radius
//
// k_search_ring: LPQ lanes per query.  Each (y,z) row of the (2R+1)^3 block is an x-run of consecutive local cells, i.e. ONE
// contiguous range of the cell-sorted map (two if the run crosses a brick boundary).  The 2(2R+1)^2 segment slots are resolved in
// parallel by the group's lanes (one directory probe, then ONE 16-byte read of the brick's prefix table for a ring-1 run), parked
// in LDS and prefix-summed, so that the group's T candidates form one flat list dealt round-robin to its lanes, eight independent
// loads in flight per lane.  One pass over the candidates: exact fp32 d2 (the oracle's op order) packed into a 32-bit key (the
// distance with its low PB mantissa bits replaced by the candidate's flat index) and kept in a SORTED TOP-8 per lane --
// K0' = min(K0,t), Kj' = med3(K(j-1),Kj,t): eight VALU ops per candidate, no payload registers; the lanes' lists are merged over
// DPP with a bitonic half-cleaner + three-stage bitonic merge.  The packed keys only decide WHICH candidates can be among the
// five nearest (those whose key does not exceed the 5th's above the packed bits: at hand among the eight unless four neighbours
// agree to 2^-15 relative); the group loads those points, exchanges their exact (d2, map index) and every lane places its points
// at their exact rank.  A query is settled when its 5th distance lies within the block's guaranteed radius
// (R + distance to the nearest face of the centre cell) * c and the packed keys left the set closed; otherwise it goes to the
// next list.
// ------------------------------------------------------------------------------------------------
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

// ---- sorted top-6 of PACKED keys: the fp32 squared distance with its low PB mantissa bits replaced by the
// candidate's flat index inside the group's candidate list.  Positive floats order like their bit patterns, so the
// whole selection is unsigned-integer min / med3 (6 VALU ops per candidate, nothing else rides along):
// K0' = min(K0,t), Kj' = med3(K(j-1),Kj,t).  Truncation keeps the order of any two candidates whose distances differ
// above the PB-th bit; a query whose best six contain two neighbours that agree there is not settled by this kernel
// (the general path orders them by (d2, map index) exactly), so the packing never decides a result.
constexpr uint32_t kEmptyPacked = 0x7F000000u;  // above every real squared distance, below inf/nan patterns
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
constexpr int kTop = 8;  // packed keys kept per lane / per group: the five wanted + up to three to see ties at the boundary
// KT = how many of them the candidate loop maintains (the rest stay empty): 8, or 6 where a tie at the boundary may simply go to
// the next stage (the first stage: a 6th neighbour within 2^-15 relative of the 5th is a rarity, two VALU ops per candidate are not)
template <int KT>
__device__ __forceinline__ void insK(uint32_t (&K)[kTop], uint32_t t) {
    static_assert(KT == 6 || KT == 8, "KT");
    const uint32_t n0 = min(K[0], t);
    const uint32_t n1 = umed3(K[0], K[1], t);
    const uint32_t n2 = umed3(K[1], K[2], t);
    const uint32_t n3 = umed3(K[2], K[3], t);
    const uint32_t n4 = umed3(K[3], K[4], t);
    const uint32_t n5 = umed3(K[4], K[5], t);
    if (KT == 8) {
        const uint32_t n6 = umed3(K[5], K[6], t);
        const uint32_t n7 = umed3(K[6], K[7], t);
        K[6] = n6; K[7] = n7;
    }
    K[0] = n0; K[1] = n1; K[2] = n2; K[3] = n3; K[4] = n4; K[5] = n5;
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
__device__ __forceinline__ float4 load_pt(__amdgpu_buffer_rsrc_t rsrc, uint32_t idx) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(idx << 4), 0, 0);  // out of range -> zeros
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// c ? a : b as one v_cndmask (the compiler turns chains of ?: on loaded values into jump trees)
__device__ __forceinline__ uint32_t sel_u32(bool c, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(__builtin_amdgcn_ballot_w64(c)));
    return r;
}
__device__ __forceinline__ u32x3 load_xyz(__amdgpu_buffer_rsrc_t rsrc, uint32_t idx) {  // coordinates only (12 of the 16 B)
    return __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)(idx << 4), 0, 0);
}

// A query group never spans a wave, so the LDS hand-offs inside k_search_ring only need wave-level ordering:
// LDS operations of one wave complete in order; this keeps the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Developer instrumentation (tools/phases.py builds a separate library with -DFLH_PHASES): per-wave cycle counts
// of the phases of the first trip, summed into cand_counter[base + i]; compiled out of the product.
#ifdef FLH_PHASES
#define PH_DECL u64 ph_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; bool ph_first = true; const u64 ph_r0 = __builtin_amdgcn_s_memrealtime();
#define PH_MARK(i)                                                      \
    do {                                                                \
        if (ph_first) {                                                 \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
            ph_t[i] = __builtin_readcyclecounter();                     \
        }                                                               \
    } while (0)
#define PH_NEXT_TRIP() ph_first = false
#define PH_DUMP(base)                                                                                          \
    do {                                                                                                       \
        if (cand_counter && (threadIdx.x & 63) == 0) {                                                         \
            u64* o_ = cand_counter + 128 + ((size_t)(base) * 1024 * 4 + (size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 12; \
            for (int i_ = 0; i_ < 10; ++i_) o_[i_] = ph_t[i_];                                                 \
            o_[10] = ph_r0;                                                                                    \
            o_[11] = __builtin_amdgcn_s_memrealtime();                                                         \
        }                                                                                                      \
    } while (0)
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_NEXT_TRIP()
#define PH_DUMP(base)
#endif

template <int LPQ>
__device__ __forceinline__ uint32_t exact_query(const GridParams& g, int q, int N, float qx, float qy, float qz, int cx, int cy,
                                                int cz, float fx, float fy, float fz, float ub, int rmax, float max_sqdist,
                                                int lane, float4* __restrict__ nn_pts, float* __restrict__ nn_d2,
                                                uint8_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected);

template <int LPQ>
__device__ __forceinline__ void top5_finish(Top5& L, const GridParams& g, int q, int N, int lane, float max_sqdist,
                                            float4* __restrict__ nn_pts, float* __restrict__ nn_d2, uint8_t* __restrict__ nn_cnt,
                                            uint8_t* __restrict__ selected);

template <int LPQ, int RING, bool BOUNDED, int PB, bool FINAL, bool OCT = false>
__global__ void __launch_bounds__(256)
k_search_ring(GridParams g, StateDev s, const float4* __restrict__ body, int N, uint32_t map_points, float max_sqdist,
              float4* __restrict__ nn_pts, float* __restrict__ nn_d2, uint8_t* __restrict__ nn_cnt,
              uint8_t* __restrict__ selected, const uint32_t* __restrict__ in_list, const uint32_t* __restrict__ in_count,
              uint32_t* __restrict__ out_list, uint32_t* __restrict__ out_count, uint32_t stripe_cap,
              const float* ub_in, float* ub_out /* may alias ub_in */, int rmax, u64* __restrict__ cand_counter, int own_axis,
              float own_lo, float own_hi) {
#include "flh_ring_body.inc"
}

// Group-wide merge of the lanes' sorted (d2, map index) lists and the query's result rows: 5 x (min butterfly, ballot, pop).
template <int LPQ>
__device__ __forceinline__ void top5_finish(Top5& L, const GridParams& g, int q, int N, int lane, float max_sqdist,
                                            float4* __restrict__ nn_pts, float* __restrict__ nn_d2, uint8_t* __restrict__ nn_cnt,
                                            uint8_t* __restrict__ selected) {
    const int wl0 = (threadIdx.x & 63) & ~(LPQ - 1);
    const u64 gmask = (LPQ == 64 ? ~0ull : ((1ull << LPQ) - 1ull)) << wl0;
    // ---- group merge: 5 x (min butterfly, ballot, pop)
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
            u64 kk = rk[0];
            uint32_t pp = rp[0];
#pragma unroll
            for (int j = 1; j < 5; ++j)
                if (jr == j) { kk = rk[j]; pp = rp[j]; }
            const bool has = jr < cnt;
            float4 v = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (has) v = g.pts[pp];
            nn_pts[(size_t)jr * N + q] = v;
            (void)kk;
        }
    }
    if (lane == 0) {
        nn_cnt[q] = (uint8_t)cnt;
        selected[q] = (cnt == 5 && !(d5 > max_sqdist)) ? 1 : 0;  // laserMapping.cpp:671
    }
}

// The general exact search of ONE query by a group of LPQ lanes: one pass over the cells that intersect the ball of
// radius sqrt(ub) around the query (ub = an upper bound of its true 5th squared distance, capped by the gate max_sqdist
// of src/laserMapping.cpp:671 -- beyond the gate a result can never be selected).  64-bit keys
// (d2 bits << 32 | map index) give the oracle's (d2, index) order, ties included.  Writes the query's result rows.
template <int LPQ>
__device__ __forceinline__ uint32_t exact_query(const GridParams& g, int q, int N, float qx, float qy, float qz, int cx, int cy,
                                                int cz, float fx, float fy, float fz, float ub, int rmax, float max_sqdist,
                                                int lane, float4* __restrict__ nn_pts, float* __restrict__ nn_d2,
                                                uint8_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected) {
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
            const float4 pv = g.pts[i];
            L.insert(make_key(dist2(qx, qy, qz, pv.x, pv.y, pv.z), pv.w), i);
        }
    }
    top5_finish<LPQ>(L, g, q, N, lane, max_sqdist, nn_pts, nn_d2, nn_cnt, selected);
    return ncand;
}

// A3 as a kernel of its own: every query (lanes_per_query == 0, the tests' cross-check) or the queries of a work list
// (grids without a ring-2 stage); 32 lanes per query, 8 queries per block.
__global__ void __launch_bounds__(256)
k_search_exact(GridParams g, StateDev s, const float4* __restrict__ body, int N, float max_sqdist, int rmax,
               float4* __restrict__ nn_pts, float* __restrict__ nn_d2, uint8_t* __restrict__ nn_cnt,
               uint8_t* __restrict__ selected, const uint32_t* __restrict__ slow_list,
               const uint32_t* __restrict__ slow_count, uint32_t stripe_cap, const float* __restrict__ ub_in,
               int all_queries, u64* __restrict__ cand_counter, int own_axis, float own_lo, float own_hi) {
    constexpr int LPQ = 32;
    const int lane = threadIdx.x & (LPQ - 1);
    const int grp = threadIdx.x / LPQ;
    const uint32_t stripe = blockIdx.x & (kStripes - 1);
    const uint32_t sub = blockIdx.x / kStripes, nsub = gridDim.x / kStripes;
    if (!all_queries) slow_list += (size_t)stripe * stripe_cap;
    const uint32_t total = all_queries ? (uint32_t)N : slow_count[stripe];
    const uint32_t gi0 = all_queries ? blockIdx.x * 8 + grp : sub * 8 + grp;
    const uint32_t gstep = all_queries ? gridDim.x * 8 : nsub * 8;
    for (uint32_t gi = gi0; gi < total; gi += gstep) {
        const int q = all_queries ? (int)gi : (int)slow_list[gi];
        const float4 b = body[q];
        float qx, qy, qz;
        body_to_world(s, b.x, b.y, b.z, qx, qy, qz);
        if (all_queries && own_axis >= 0) {
            const float oc = own_axis == 0 ? qx : (own_axis == 1 ? qy : qz);
            if (!(oc >= own_lo && oc < own_hi)) {
                if (lane == 0) { selected[q] = 0; nn_cnt[q] = 0; }
                continue;
            }
        }
        int cx, cy, cz;
        float fx, fy, fz;
        cell_of(g, qx, qy, qz, cx, cy, cz, fx, fy, fz);
        const float ub = all_queries ? max_sqdist : fminf(ub_in[q], max_sqdist);
        const uint32_t ncand = exact_query<LPQ>(g, q, N, qx, qy, qz, cx, cy, cz, fx, fy, fz, ub, rmax, max_sqdist, lane, nn_pts,
                                                nn_d2, nn_cnt, selected);
        if (cand_counter && ncand) atomicAdd(cand_counter, (u64)ncand);
    }
}

// ------------------------------------------------------------------------------------------------
// B: one thread per scan point: plane fit, residual gate, Jacobian row; then the wave's 64 rows are
// contracted into a 16x16 Gram block on the f64 matrix core.  v = [row(12) | h=-pd2 | 1 | |pd2| | 0]:
//   G[i][j] (i,j<12) = HTH,  G[i][12] = HTh,  G[13][13] = n_eff,  G[14][13] = total_residual.
// v_mfma_f64_16x16x4_f64 takes A[i][k] in lane (i + 16k) and B[k][j] in lane (j + 16k): with A = B^T
// = the same register, one LDS transpose ([point][16] -> lane (col, point%4)) feeds both operands.
// ------------------------------------------------------------------------------------------------
#ifdef FLH_PHASES
__device__ u64 g_fit_ph[4096 * 4 * 12];
#define FPH(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); fph[i] = __builtin_readcyclecounter(); } while (0)
#define FPH_DUMP() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) { u64* o_ = g_fit_ph + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 12; \
    for (int i_ = 0; i_ < 10; ++i_) o_[i_] = fph[i_]; o_[10] = fph_r0; o_[11] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define FPH(i)
#define FPH_DUMP()
#endif
typedef double v4f64 __attribute__((ext_vector_type(4)));
// Compact layout of the entries of the 16x16 Gram block the filter reads: the upper triangle of the leading ncol x ncol block
// (ncol = 12 with extrinsic estimation, else 6: the last six columns are structurally zero, laserMapping.cpp:745), then the
// ncol entries of column 12 (H^T h), then n_eff (G[13][13]) and total_residual (G[14][13]).  -1 = not transmitted.
__host__ __device__ inline int gram_nslots(int ncol) { return ncol * (ncol + 1) / 2 + ncol + 2; }
__host__ __device__ inline int gram_slot(int r, int c, int ncol) {
    const int tri = ncol * (ncol + 1) / 2;
    if (c < ncol && r <= c) return r * ncol - r * (r - 1) / 2 + (c - r);
    if (c == 12 && r < ncol) return tri + r;
    if (r == 13 && c == 13) return tri + ncol;
    if (r == 14 && c == 13) return tri + ncol + 1;
    return -1;
}
int gram_slots_host(int ncol) { return gram_nslots(ncol); }
int gram_slot_host(int r, int c, int ncol) { return gram_slot(r, c, ncol); }
constexpr int kRed1 = 16;   // blocks per first-level reduction group
constexpr int kRed2 = 32;   // group sums added per unrolled batch at the top level
constexpr int kTileStride = 17;  // doubles per row: 16 + 1 pad (conflict-free ds_write_b64 / ds_read_b64)

template <int ORD, bool HALF, int PM = 0>
__global__ void __launch_bounds__(256)
k_fit(StateDev s, const float4* __restrict__ body, const float4* __restrict__ nn_pts, int N, int ext, float thr,
      uint8_t* __restrict__ selected, float4* __restrict__ normvec, float4* __restrict__ world,
      double* __restrict__ partials, double* __restrict__ part2, double* __restrict__ out256, double seq,
      uint32_t* __restrict__ tickets, uint32_t* __restrict__ slow_count, double* __restrict__ gran, int red1, int ncol,
      int store_aux, float4* __restrict__ plane_cache) {
    // PM (flh_config.plane_cache, on by default; 0 = off and the fetch path): a plane depends on the five neighbours only, not on the state, and a
    // point that enters a no-search pass with its flag set was fitted successfully on the pass before, from the very same
    // neighbours.  1 (searching pass): fit as always and keep (a, b, c, d) per point; 2 (no-search pass): take the plane from there
    // -- same bits -- instead of re-reading 80 B of neighbours and repeating the QR.  The gate and the Jacobian row are
    // recomputed either way: they depend on the state.
    // store_aux: also write feats_down_world and normvec (16 B per point each).  The filter never reads them, so the passes of an
    // update leave them out; a fetch (flh_fetch_world / _normvec / _rows) re-runs this kernel once with the flag set -- the
    // arithmetic is deterministic, so that run reproduces the pass bit for bit.
    __shared__ double lds[4 * 64 * kTileStride];
    __shared__ uint32_t s_ticket;
#ifdef FLH_PHASES
    u64 fph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const u64 fph_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    FPH(0);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;

    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    float wx = 0.f, wy = 0.f, wz = 0.f;
    // every load of the thread is issued up front (the flag, the point, its five cached neighbours): one memory round
    // trip instead of three dependent ones; the ~40 % of neighbour rows fetched for unselected points are cheap next to it
    const int ic = i < N ? i : (N > 0 ? N - 1 : 0);
    const uint8_t sel_in = selected[ic];
    float4 nn[5];
    b = body[ic];
    float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PM == 2) {
        pc = plane_cache[ic];
#pragma unroll
        for (int j = 0; j < 5; ++j) nn[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) nn[j] = nn_pts[(size_t)j * N + ic];
    }
    if (i < N) {  // feats_down_world is rewritten for every point on every pass (laserMapping.cpp:656-661)
        body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
        if (store_aux) world[i] = make_float4(wx, wy, wz, 0.f);
    }
    FPH(1);  // point loaded + transformed
    if (i < N && sel_in) {  // laserMapping.cpp:674
        float P[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { P[j][0] = nn[j].x; P[j][1] = nn[j].y; P[j][2] = nn[j].z; }
        float pabcd[4];
        FPH(2);  // neighbours loaded
        bool ok;
        if (PM == 2) {  // compile-time: this instantiation has no fit in it
            pabcd[0] = pc.x; pabcd[1] = pc.y; pabcd[2] = pc.z; pabcd[3] = pc.w;
            ok = true;
        } else {
            ok = HALF ? esti_plane_half<ORD>(P, thr, pabcd) : esti_plane<ORD>(P, thr, pabcd);  // :678
            if (PM == 1 && ok) plane_cache[i] = make_float4(pabcd[0], pabcd[1], pabcd[2], pabcd[3]);
        }
        FPH(3);  // plane fit
        bool sel = false;
        float pd2 = 0.f;
        if (ok) {
            pd2 = ((pabcd[0] * wx + pabcd[1] * wy) + pabcd[2] * wz) + pabcd[3];  // :680
            const double bx = (double)b.x, by = (double)b.y, bz = (double)b.z;
            const double nb = sqrt((bx * bx + by * by) + bz * bz);
            const float sg = (float)(1 - 0.9 * (double)fabsf(pd2) / sqrt(nb));  // :681
            sel = (double)sg > 0.9;                                             // :683
        }
        selected[i] = sel ? 1 : 0;
        if (sel) {
            if (store_aux) normvec[i] = make_float4(pabcd[0], pabcd[1], pabcd[2], pd2);  // :686-689
            // Jacobian row, fp64 (:723-752)
            const double bx = (double)b.x, by = (double)b.y, bz = (double)b.z;
            double px, py, pz;
            quat_rot(s.offR, bx, by, bz, px, py, pz);
            px = px + s.offT[0]; py = py + s.offT[1]; pz = pz + s.offT[2];
            const double rotc[4] = {-s.rot[0], -s.rot[1], -s.rot[2], s.rot[3]};
            const double nx = (double)pabcd[0], ny = (double)pabcd[1], nz = (double)pabcd[2];
            double Cx, Cy, Cz;
            quat_rot(rotc, nx, ny, nz, Cx, Cy, Cz);  // C = R^T n
            v[0] = nx; v[1] = ny; v[2] = nz;
            v[3] = (-pz) * Cy + py * Cz;             // A = hat(p_I) C
            v[4] = pz * Cx + (-px) * Cz;
            v[5] = (-py) * Cx + px * Cy;
            if (ext) {
                const double offRc[4] = {-s.offR[0], -s.offR[1], -s.offR[2], s.offR[3]};
                double Dx, Dy, Dz;
                quat_rot(offRc, Cx, Cy, Cz, Dx, Dy, Dz);
                v[6] = (-bz) * Dy + by * Dz;         // B = hat(p_b) R_LI^T C
                v[7] = bz * Dx + (-bx) * Dz;
                v[8] = (-by) * Dx + bx * Dy;
                v[9] = Cx; v[10] = Cy; v[11] = Cz;
            }
            v[12] = -(double)pd2;          // h(i) = -norm_p.intensity (:750)
            v[13] = 1.0;                   // effct_feat_num
            v[14] = (double)fabsf(pd2);    // res_last -> total_residual (:702)
        }
    }
    FPH(4);  // gate + Jacobian row
    double* T = lds + wave * 64 * kTileStride;
#pragma unroll
    for (int c = 0; c < 16; ++c) T[lane * kTileStride + c] = v[c];
    __syncthreads();
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const int col = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const double a = T[(4 * m + kq) * kTileStride + col];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    __syncthreads();
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* Rb = lds;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rb[wave * 256 + (kq + 4 * r) * 16 + col] = acc[r];
    __syncthreads();
    FPH(5);  // Gram block of the wave in LDS
    const int t = threadIdx.x;
    // ---- R: deterministic two-level cross-block sum inside this launch (no reduce kernels, no extra
    // boundaries).  Blocks are grouped kRed1 at a time; the LAST block of a group to finish sums the group's
    // partials in block order, the LAST group to finish sums the group sums in group order and writes the
    // result (out256 is pinned host memory on the flh_eval path).  Fixed summation order -> run-to-run
    // identical bits regardless of which block happens to arrive last.
    // Hand-off: partials are stored write-through (8-byte agent-scope stores = sc1), every storing wave
    // drains, one lane takes a relaxed agent-scope ticket; the reducer reads them back with agent-scope
    // (L1-bypassing) loads.  No release/acquire fence, hence no L2 write-back sweep per block.
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble* gpart = (gdouble*)partials;
    gdouble* gpart2 = (gdouble*)part2;
    __hip_atomic_store(gpart + (size_t)blockIdx.x * 256 + t, (Rb[t] + Rb[256 + t]) + (Rb[512 + t] + Rb[768 + t]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nblk = gridDim.x;
    const int red = gran ? red1 : kRed1;
    const int ngroups_gran = gran ? (nblk + red1 - 1) / red1 : 0;
    const int group = blockIdx.x / red;
    const int ngroups = (nblk + kRed1 - 1) / kRed1;
    const int gsize = min(red, nblk - group * red);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_ticket = __hip_atomic_fetch_add(&tickets[1 + group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    FPH(6);  // partial stored, ticket taken
    if (s_ticket != (uint32_t)(gsize - 1)) { FPH_DUMP(); return; }  // block-uniform
    if (gran) {
        // ---- flh_eval's path: ONE level on the device.  The last block of a group sums the group's partials in block
        // order and hands the entries the host needs (upper triangle of the leading ncol x ncol block, the Hth column,
        // n_eff, total_residual -- 29 values without extrinsic estimation, 92 with) straight to pinned host memory as
        // 16-byte {value, sequence} granules: no drain, no flag, no second ticket, no final block.  The host checks every
        // granule's tag and adds the groups up in group order (gram_slot() below is shared with it).
        const int b0 = group * red;
        double s0 = 0.0;
        for (int j0 = 0; j0 < gsize; j0 += 16) {
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[j] = (j0 + j < gsize) ? __hip_atomic_load(gpart + (size_t)(b0 + j0 + j) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : 0.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) s0 += v[j];
        }
        const int slot = gram_slot(t >> 4, t & 15, ncol);
        if (slot >= 0) {
            typedef double v2f64 __attribute__((ext_vector_type(2)));
            const v2f64 g2 = {s0, seq};
            double* dst = gran + ((size_t)group * gram_nslots(ncol) + slot) * 2;
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(g2) : "memory");  // one 16-byte system-scope store
        }
        if (t == 0) tickets[1 + group] = 0;                         // re-arm this group's ticket for the next launch
        if (group == 0) {
            // the number of queries the first search stage could not settle in this pass (its work-list counters): handed to the
            // host as one more granule behind the last group's -- it picks the launch plan of the next search pass from it --
            // then the counters are re-armed (the second stage has retired)
            if (t >= 192) {
                uint32_t c = slow_count[t - 192];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
                if (t == 255) {
                    typedef double v2f64 __attribute__((ext_vector_type(2)));
                    const v2f64 g2 = {(double)c, seq};
                    double* dst = gran + (size_t)ngroups_gran * gram_nslots(ncol) * 2;
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(g2) : "memory");
                }
            }
            __syncthreads();
            if (t < 2 * kStripes) slow_count[t] = 0;
        }
        FPH(7);
        FPH_DUMP();
        return;
    }
    {
        const int b0 = group * kRed1;
        double v[kRed1];
#pragma unroll
        for (int j = 0; j < kRed1; ++j)
            v[j] = (j < gsize) ? __hip_atomic_load(gpart + (size_t)(b0 + j) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : 0.0;
        double s0 = 0.0;
#pragma unroll
        for (int j = 0; j < kRed1; ++j) s0 += v[j];
        __hip_atomic_store(gpart2 + (size_t)group * 256 + t, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_ticket = __hip_atomic_fetch_add(&tickets[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    FPH(7);  // group summed, second ticket taken
    if (s_ticket != (uint32_t)(ngroups - 1)) { FPH_DUMP(); return; }
    {
        double sum = 0.0;
        for (int b0 = 0; b0 < ngroups; b0 += kRed2) {
            double v[kRed2];
#pragma unroll
            for (int j = 0; j < kRed2; ++j)
                v[j] = (b0 + j < ngroups)
                           ? __hip_atomic_load(gpart2 + (size_t)(b0 + j) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : 0.0;
#pragma unroll
            for (int j = 0; j < kRed2; ++j) sum += v[j];
        }
        // out256 is pinned host memory on the flh_eval path: system-scope write-through stores, then (below) a
        // sequence word in the unused G[15][15] slot that the host polls -- it need not wait for the kernel to retire.
        if (t != 255) __hip_atomic_store(out256 + t, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // the stores above are write-through to system scope: draining them (vmcnt) orders them before the flag -- no
    // release fence, whose L2 write-back sweep costs more than the whole publish (MI355X_MICROARCH.md, publish rows)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(out256 + 255, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    FPH(8);  // result published
    FPH_DUMP();
    // re-arm: tickets for the next launch, and the A1 -> A2 work-list counters for the next search pass
    for (int i = t; i < ngroups + 1; i += 256) tickets[i] = 0;
    if (t < 2 * kStripes) slow_count[t] = 0;
}

// ------------------------------------------------------------------------------------------------
// Scan staging: Morton-order the scan in the BODY frame (a rigid transform keeps neighbours together,
// so one sort per scan serves every IEKF pass).  Neighbouring lanes then walk the same bricks, cells
// and map points, which is what turns the search from line-traffic-bound into cache-resident.
// ------------------------------------------------------------------------------------------------
// 32-bit key: 0.5 m quantum (the leaf of the down-sampling that produced the scan: finer would order nothing), x and y 11 bits
// (+-512 m), z 10 bits (+-256 m); the low 10 bits of the three interleaved, the 11th bits of x and y on top.  32-bit keys
// halve what the sort moves per pass next to the 42-bit keys of round 2 (rocPRIM merge-sorts arrays of this size: block
// sort + 7 merge passes, all on the copy stream beside the previous scan's update).
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
__global__ void __launch_bounds__(256) k_scan_keys(const float4* __restrict__ raw, uint32_t N, float inv_q,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 p = raw[i];
    keys[i] = scan_morton(p.x, p.y, p.z, inv_q);
    vals[i] = i;
}
// The caller's point records exactly as they came over PCIe (xyz first, any stride that is a multiple of 4 -- 12 packed,
// 16 float4, 48 pcl::PointXYZINormal): re-strided to float4 on the device (the host used to do this in a per-point loop),
// optionally with a fourth float picked from inside the record (the time offset the undistortion needs), and -- for the
// plain staging path -- the Morton key in the same pass.  bad (optional) counts non-finite input.
__global__ void __launch_bounds__(256) k_scan_restride(const uint32_t* __restrict__ words, uint32_t stride_words, uint32_t w_off,
                                                       int has_w, uint32_t N, float inv_q, float4* __restrict__ raw,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                       uint32_t* __restrict__ bad) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t* r = words + (size_t)i * stride_words;
    float4 p;
    p.x = __uint_as_float(r[0]);
    p.y = __uint_as_float(r[1]);
    p.z = __uint_as_float(r[2]);
    p.w = has_w ? __uint_as_float(r[w_off]) : 0.f;
    raw[i] = p;
    if (keys) {
        keys[i] = scan_morton(p.x, p.y, p.z, inv_q);
        vals[i] = i;
    }
    if (bad && !(isfinite(p.x) && isfinite(p.y) && isfinite(p.z) && isfinite(p.w))) atomicAdd(bad, 1u);
}
// body[i] = raw[perm[i]] with .w = original index; perm == nullptr -> identity
__global__ void __launch_bounds__(256) k_scan_gather(const float4* __restrict__ raw, const uint32_t* __restrict__ perm,
                                                     uint32_t N, float4* __restrict__ body) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t src = perm ? perm[i] : i;
    float4 p = raw[src];
    p.w = __uint_as_float(src);
    body[i] = p;
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
// A launch that may carry the timing events of a sampled evaluation: hipExtLaunchKernelGGL takes the kernel's own start / stop
// time stamps from its dispatch packet -- no extra barrier packets on the stream (three hipEventRecord per sampled evaluation
// cost ~10 us of it, a sixth of a pass) and the bracket is the kernels' time, what rocprofv3 reports.
#define FLH_LAUNCH_EV(kernel, grid, block, st, evA, evB, ...)                                           \
    do {                                                                                                \
        if ((evA) != nullptr || (evB) != nullptr)                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, st, evA, evB, 0, __VA_ARGS__);                \
        else                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                \
    } while (0)

hipError_t launch_map_keys(const GridParams& g, const float4* pts, uint32_t M, u64* keys, uint32_t* vals, hipStream_t st) {
    hipLaunchKernelGGL(k_map_keys, dim3(cdiv(M, 256)), dim3(256), 0, st, g, pts, M, keys, vals);
    return hipGetLastError();
}
hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const u64* kin, u64* kout, const uint32_t* vin, uint32_t* vout,
                      uint32_t M, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)M, 0, 36, st);
}
hipError_t inclusive_sum(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint32_t* out, uint32_t M, hipStream_t st) {
    return hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, in, out, (int)M, st);
}
hipError_t launch_brick_starts(const uint32_t* brick_head, const uint32_t* rank_incl, uint32_t M, uint32_t* brick_start,
                               hipStream_t st) {
    hipLaunchKernelGGL(k_brick_starts, dim3(cdiv(M, 256)), dim3(256), 0, st, brick_head, rank_incl, M, brick_start);
    return hipGetLastError();
}
hipError_t launch_brick_tables(const u64* ks, const uint32_t* brick_start, uint32_t nbricks, const uint32_t* cap_incl,
                               const uint32_t* cap, uint32_t* starts, uint32_t* cap_end, uint32_t* live, uint2* hash,
                               uint32_t hash_mask, int hash_shift, hipStream_t st) {
    if (nbricks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_tables, dim3(cdiv((long long)nbricks * kBrickStride, 256)), dim3(256), 0, st, ks,
                       brick_start, nbricks, cap_incl, cap, starts, cap_end, live, hash, hash_mask, hash_shift);
    return hipGetLastError();
}
hipError_t launch_brick_heads(const u64* ks, uint32_t M, uint32_t* bh, hipStream_t st) {
    if (M == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_heads, dim3(cdiv(M, 256)), dim3(256), 0, st, ks, M, bh);
    return hipGetLastError();
}
hipError_t launch_brick_caps(const uint32_t* brick_start, uint32_t nbricks, uint32_t* cap, hipStream_t st) {
    if (nbricks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_brick_caps, dim3(cdiv(nbricks, 256)), dim3(256), 0, st, brick_start, nbricks, cap);
    return hipGetLastError();
}
hipError_t launch_fill_tomb(float4* pts, uint32_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_tomb, dim3(cdiv(n, 256)), dim3(256), 0, st, pts, n);
    return hipGetLastError();
}
hipError_t launch_map_place(const float4* pts, const uint32_t* vs, const uint32_t* br_incl, const uint32_t* brick_start,
                            const uint32_t* cap_incl, const uint32_t* cap, uint32_t M, float4* out, hipStream_t st) {
    if (M == 0) return hipSuccess;
    hipLaunchKernelGGL(k_map_place, dim3(cdiv(M, 256)), dim3(256), 0, st, pts, vs, br_incl, brick_start, cap_incl, cap, M, out);
    return hipGetLastError();
}

hipError_t launch_scan_keys(const float4* raw, uint32_t N, float quantum, uint32_t* keys, uint32_t* vals, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_keys, dim3(cdiv(N, 256)), dim3(256), 0, st, raw, N, 1.0f / quantum, keys, vals);
    return hipGetLastError();
}
hipError_t launch_scan_restride(const void* bytes, uint32_t stride_bytes, uint32_t w_off_bytes, int has_w, uint32_t N, float quantum,
                                float4* raw, uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_restride, dim3(cdiv(N, 256)), dim3(256), 0, st, (const uint32_t*)bytes, stride_bytes / 4u,
                       w_off_bytes / 4u, has_w, N, 1.0f / quantum, raw, keys, vals, bad);
    return hipGetLastError();
}
hipError_t sort_scan_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                           uint32_t N, hipStream_t st) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, (int)N, 0, 32, st);
}
hipError_t launch_scan_gather(const float4* raw, const uint32_t* perm, uint32_t N, float4* body, hipStream_t st) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_gather, dim3(cdiv(N, 256)), dim3(256), 0, st, raw, perm, N, body);
    return hipGetLastError();
}

int list_stripes() { return kStripes; }
uint32_t list_stripe_cap(int N) { return (uint32_t)(cdiv(cdiv(N > 0 ? N : 1, 16), kStripes) + 1) * 128u; }

hipError_t launch_search(int lpq, int first_stage, const GridParams& g, const StateDev& s, const float4* body, int N, uint32_t map_points,
                         float max_sqdist, int rmax, float4* nn_pts, float* nn_d2, uint8_t* nn_cnt, uint8_t* selected,
                         uint32_t* list1, uint32_t* list2, float* ub, uint32_t* counts /* [2 * kStripes] */,
                         u64* cand_counter, int own_axis, float own_lo, float own_hi, hipStream_t st,
                         hipEvent_t ev_start, hipEvent_t ev_stop, int lpq2) {
    if (N <= 0) return hipSuccess;
    const dim3 blk(256);
    const uint32_t cap = list_stripe_cap(N);
    const hipEvent_t ev_none = nullptr;
    if (lpq == 0) {  // exact path for every query (validation / fallback)
        FLH_LAUNCH_EV(k_search_exact, dim3(std::min(cdiv(N, 8), 4096)), blk, st, ev_start, ev_stop, g, s, body, N, max_sqdist, rmax,
                      nn_pts, nn_d2, nn_cnt, selected, list1, counts, cap, ub, 1, cand_counter, own_axis, own_lo, own_hi);
        return hipGetLastError();
    }
    // A1: ring 1, every query
#define FLH_A1(L, O)                                                                                                     \
    FLH_LAUNCH_EV((k_search_ring<L, 1, false, 8, false, O>), dim3(cdiv(N, 256 / L)), blk, st, ev_start, ev_none, g, s, body, N, \
                  map_points, max_sqdist, nn_pts, nn_d2, nn_cnt, selected, (const uint32_t*)nullptr,                     \
                  (const uint32_t*)nullptr, list1, counts, cap, (const float*)nullptr, ub, rmax, cand_counter, own_axis,          \
                  own_lo, own_hi)
    if (first_stage == 2 && rmax >= 2) {
        switch (lpq) {
            case 1: FLH_A1(1, true); break;
            case 2: FLH_A1(2, true); break;
            default: FLH_A1(4, true); break;
        }
    } else {
        switch (lpq) {
            case 1: FLH_A1(1, false); break;
            case 2: FLH_A1(2, false); break;
            case 8: FLH_A1(8, false); break;
            case 16: FLH_A1(16, false); break;
            default: FLH_A1(4, false); break;
        }
    }
#undef FLH_A1
    if (rmax >= 2) {
        // A2: ring 2 over list 1, inside the ball A1's 5th distance defines; whatever it cannot settle (distance ties, a
        // 5th neighbour beyond the 5x5x5 block) it finishes itself with the general exact search
#define FLH_A2(L2, BPS)                                                                                                          \
        FLH_LAUNCH_EV((k_search_ring<L2, 2, true, 11, true>), dim3(kStripes * BPS), blk, st, ev_none, ev_stop, g, s, body, N, map_points, \
                      max_sqdist, nn_pts, nn_d2, nn_cnt, selected, (const uint32_t*)list1, (const uint32_t*)counts, list2,       \
                      counts + kStripes, cap, (const float*)ub, ub, rmax, cand_counter, -1, 0.f, 0.f)
        // lanes per query of the second stage (flh_config.second_stage_lanes) and blocks per stripe so that a stripe's share of
        // ~10 % of the queries is one trip of its blocks
        // measured on BASELINE configs[1] (rocprofv3 mean over first and later searches): 4 lanes 15.5 us, 8 lanes 12.3, 16 lanes 13.3,
        // 32 lanes 16.3, 64 lanes 24.2
        if (lpq2 == 32) FLH_A2(32, 32);
        else if (lpq2 == 16) FLH_A2(16, 16);
        else FLH_A2(8, 8);
#undef FLH_A2
    } else {
        // cells as large as the gate radius: the general search drains list 1 directly
        FLH_LAUNCH_EV(k_search_exact, dim3(kStripes * 8), blk, st, ev_none, ev_stop, g, s, body, N, max_sqdist, rmax, nn_pts, nn_d2,
                      nn_cnt, selected, (const uint32_t*)list1, (const uint32_t*)counts, cap, ub, 0, cand_counter, -1, 0.f, 0.f);
    }
    return hipGetLastError();
}

#ifdef FLH_PHASES
void dump_fit_phases() {
    static std::vector<u64> ph(4096 * 4 * 12);
    if (hipMemcpyFromSymbol(ph.data(), HIP_SYMBOL(g_fit_ph), ph.size() * sizeof(u64)) != hipSuccess) return;
    double sum[8] = {0}, cnt[8] = {0};
    u64 r0 = ~0ull, r1 = 0;
    double longest = 0;
    for (size_t w = 0; w < 4096 * 4; ++w) {
        const u64* o = ph.data() + w * 12;
        if (o[0] == 0) continue;
        r0 = std::min(r0, o[10]);
        r1 = std::max(r1, o[11]);
        longest = std::max(longest, (double)(o[11] - o[10]) / 100.0);
        for (int i = 0; i < 8; ++i)
            if (o[i + 1] && o[i]) { sum[i] += (double)(o[i + 1] - o[i]); cnt[i] += 1; }
    }
    std::fprintf(stderr, "[phases] k_fit mean cycles/phase (waves that reached it):");
    for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %.0f(%.0f)", cnt[i] ? sum[i] / cnt[i] : 0.0, cnt[i]);
    std::fprintf(stderr, " | longest wave %.2f us | first start -> last end %.2f us\n", longest, (double)(r1 - r0) / 100.0);
    std::vector<u64> z(ph.size(), 0);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fit_ph), z.data(), z.size() * sizeof(u64));
}
#endif

// After an all-reduce: the summed 16x16 block from device memory to pinned host memory, then the sequence word (the same
// publish protocol as k_fit's last block).
__global__ void __launch_bounds__(256) k_publish256(const double* __restrict__ src, double* __restrict__ out256, double seq) {
    const int t = threadIdx.x;
    const double v = src[t];
    if (t != 255) __hip_atomic_store(out256 + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(out256 + 255, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_publish256(const double* src, double* out256, double seq, hipStream_t st) {
    hipLaunchKernelGGL(k_publish256, dim3(1), dim3(256), 0, st, src, out256, seq);
    return hipGetLastError();
}

// pointSearchSqDis on demand: the squared distances of the cached neighbours to the query's world position at the state of
// the search that found them -- the very expression (and bits) the search compared; rows without a neighbour get +inf.
__global__ void __launch_bounds__(256) k_fill_d2(StateDev s_search, const float4* __restrict__ body, const float4* __restrict__ nn_pts,
                                                 int N, float* __restrict__ nn_d2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float4 b = body[i];
    float wx, wy, wz;
    body_to_world(s_search, b.x, b.y, b.z, wx, wy, wz);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float4 p = nn_pts[(size_t)j * N + i];
        nn_d2[(size_t)j * N + i] = (__float_as_uint(p.w) == 0xFFFFFFFFu) ? INFINITY : dist2(wx, wy, wz, p.x, p.y, p.z);
    }
}
hipError_t launch_fill_d2(const StateDev& s_search, const float4* body, const float4* nn_pts, int N, float* nn_d2, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_d2, dim3(cdiv(N, 256)), dim3(256), 0, st, s_search, body, nn_pts, N, nn_d2);
    return hipGetLastError();
}

int fit_blocks(int N) { return cdiv(N > 0 ? N : 1, 256); }
int reduce1_blocks(int nblk, int* per_out) {
    if (per_out) *per_out = kRed1;
    return cdiv(nblk, kRed1);
}

hipError_t launch_fit(int order, int half_fit, const StateDev& s, const float4* body, const float4* nn_pts, int N, int ext, float thr,
                      uint8_t* selected, float4* normvec, float4* world, double* partials, double* part2,
                      double* out256, double seq, uint32_t* tickets, uint32_t* slow_count, double* gran, int red1, int store_aux,
                      hipStream_t st, float4* plane_cache, int plane_mode, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int nblk = fit_blocks(N);
    const int ncol = ext ? 12 : 6;
    if (!plane_cache || half_fit || order != 1) plane_mode = 0;  // the experiment exists for the default summation order only
    if (plane_mode == 1 || plane_mode == 2) {
        if (plane_mode == 1)
            FLH_LAUNCH_EV((k_fit<1, false, 1>), dim3(nblk), dim3(256), st, ev_start, ev_stop, s, body, nn_pts, N, ext, thr, selected, normvec, world,
                          partials, part2, out256, seq, tickets, slow_count, gran, red1, ncol, store_aux, plane_cache);
        else
            FLH_LAUNCH_EV((k_fit<1, false, 2>), dim3(nblk), dim3(256), st, ev_start, ev_stop, s, body, nn_pts, N, ext, thr, selected, normvec, world,
                          partials, part2, out256, seq, tickets, slow_count, gran, red1, ncol, store_aux, plane_cache);
        return hipGetLastError();
    }
#define FLH_FIT(O)                                                                                                      \
    FLH_LAUNCH_EV((k_fit<O, false>), dim3(nblk), dim3(256), st, ev_start, ev_stop, s, body, nn_pts, N, ext, thr, selected, normvec, world, \
                  partials, part2, out256, seq, tickets, slow_count, gran, red1, ncol, store_aux, plane_cache)
    if (half_fit) {  // the fp16 ablation exists for the default summation order only
        FLH_LAUNCH_EV((k_fit<1, true>), dim3(nblk), dim3(256), st, ev_start, ev_stop, s, body, nn_pts, N, ext, thr, selected, normvec, world,
                      partials, part2, out256, seq, tickets, slow_count, gran, red1, ncol, store_aux, plane_cache);
        return hipGetLastError();
    }
    switch (order) {
        case 0: FLH_FIT(0); break;
        case 2: FLH_FIT(2); break;
        case 3: FLH_FIT(3); break;
        default: FLH_FIT(1); break;
    }
#undef FLH_FIT
    return hipGetLastError();
}

}  // namespace flh
