// flh_kernels.hip -- the HIP kernels of the hot path, written for gfx950 (CDNA4, wave64) only.
//
//   K0  map index build     keys -> (radix sort) -> bricks, per-brick cell tables, directory            [setup]
//   A1  k_search_ring<4,1>  body->world transform + 5-NN over the 3x3x3 cell block, 4 lanes per query    [three-launch search pass]
//   A2  k_search_ring<8,2>  the queries A1 could not settle: 5x5x5 block inside A1's bound, 8 lanes per
//                           query, finishing leftovers itself with the general exact search (exact_query)
//   A3  k_search_exact      the general exact search as a kernel (lanes_per_query = 0, grids without ring 2)
//   B   k_fit<ORD,HALF,PM>  plane fit + residual gate + 12-col Jacobian row + 16x16 Gram contraction on
//                           v_mfma_f64_16x16x4_f64 + deterministic cross-block sum        [no-search passes, fetches]
//   (the default searching pass -- search, fit and reduction in ONE launch -- is k_pass, flh_pass.hip)
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
#include "flh_search_dev.hpp"
#include "flh_fit_dev.hpp"
#include "flh_mail_dev.hpp"

namespace flh {

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
    out[(cap_incl[r] - cap[r]) + (j - brick_start[r])] = p;  // (capacity: the host sized the storage from the same prefix sums)
}

// ------------------------------------------------------------------------------------------------
// A: exact 5-NN as kernels of its own -- the three-launch searching pass (flh_config.pass_kernel = 0; also what the RCCL path,
// the non-default summation orders and grids with cells below sqrt(max_sqdist) / 2 run).  The default searching pass is ONE
// launch: k_pass, flh_pass.hip.  The search itself (ring_query, exact_query) lives in flh_search_dev.hpp.
//
// A1 k_search_ring<4,1>     every query, the 3x3x3 cells around its cell, four lanes per query; settles every query whose 5th
//                            neighbour is provably inside that block and free of near-ties, lists the others
// A2 k_search_ring<8,2>      the queries A1 listed: 5x5x5 cells clipped to the ball of A1's 5th distance, eight lanes per query;
//                            whatever it cannot settle either (a true tie, a list longer than the packed index can name) it
//                            finishes itself with 64-bit (d2, map index) keys, a 5th neighbour beyond its block with the general search
// A3 k_search_exact          the general exact search as a kernel: lanes_per_query = 0 (the tests' cross-check), grids whose
//                            cells are as large as the gate radius
// (Tried on the MI355X and NOT in the product: an LDS tile of the map shared by a block / a wave of Morton-neighbouring queries,
// 1.6-2.2x slower in three versions, profiles/r03_wtile_experiment/; 1, 2, 8, 16 lanes per query and the 2x2x2 block in the first
// stage, 16 / 32 lanes in the second, profiles/r02_first_stage_sweep.log, profiles/r03_second_stage_lanes/.)
// ------------------------------------------------------------------------------------------------
template <int LPQ, int RING, bool BOUNDED, int PB, bool FINAL>
__global__ void __launch_bounds__(256)
k_search_ring(GridParams g, StateDev s, const float4* __restrict__ body, int N, uint32_t map_points, float max_sqdist,
              float4* __restrict__ nn_pts, uint8_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected,
              const uint32_t* __restrict__ in_list, const uint32_t* __restrict__ in_count, uint32_t* __restrict__ out_list,
              uint32_t* __restrict__ out_count, uint32_t stripe_cap, const float* ub_in, float* ub_out /* may alias ub_in */, int rmax,
              u64* __restrict__ cand_counter, int own_axis, float own_lo, float own_hi) {
    // own_axis >= 0 (map partitioned over ranks, flh_set_owned_interval): a query whose world coordinate lies outside
    // [own_lo, own_hi) belongs to another rank: its flag is cleared and it is not searched here.
    // Work lists are striped kStripes ways (stripe = blockIdx & (kStripes-1)) and appended to with one global atomic per WAVE:
    // thousands of returning atomics on a single word serialise at ~11 ns each and were the whole runtime of an earlier version
    // of this kernel.  No block-level barrier anywhere: waves run free.
    constexpr int GPB = 256 / LPQ;  // query groups per block
    __shared__ uint2 seg[GPB][ring_seg_slots<RING>()];
    const int grp = threadIdx.x / LPQ;
    const int lane = threadIdx.x & (LPQ - 1);
    const uint32_t stripe = blockIdx.x & (kStripes - 1);
    const uint32_t sub = blockIdx.x / kStripes, nsub = gridDim.x / kStripes;  // position among the stripe's blocks
    if (in_list) in_list += (size_t)stripe * stripe_cap;
    out_list += (size_t)stripe * stripe_cap;
    const uint32_t total = in_list ? in_count[stripe] : (uint32_t)N;
    const uint32_t first_base = in_list ? sub * GPB : blockIdx.x * GPB;
    const uint32_t step_base = in_list ? nsub * GPB : gridDim.x * GPB;
    const RingRsrc rs(g, map_points);
    // the first stage (no input list) launches one block per GPB queries and makes a single trip: the state and grid scalars
    // die after the transform instead of staying pinned in SGPRs around a loop
    for (uint32_t base = first_base; base < total; base += step_base) {
        const uint32_t gi = base + grp;
        bool live = gi < total;
        const int q = in_list ? (int)in_list[live ? gi : total - 1] : (int)(live ? gi : total - 1);
        const float4 b = body[q];
        float qx, qy, qz;
        body_to_world(s, b.x, b.y, b.z, qx, qy, qz);
        if (!in_list && own_axis >= 0) {
            const float oc = own_axis == 0 ? qx : (own_axis == 1 ? qy : qz);
            if (live && !(oc >= own_lo && oc < own_hi)) {
                if (lane == 0) { selected[q] = 0; nn_cnt[q] = 0; }
                live = false;
            }
        }
        float ub_next;
        const bool done = ring_query<LPQ, RING, BOUNDED, PB, FINAL, true>(g, rs, seg[grp], lane, q, N, live, qx, qy, qz,
                                                                               BOUNDED ? ub_in[q] : INFINITY, max_sqdist, rmax, nn_pts,
                                                                               nn_cnt, selected, cand_counter, nullptr, ub_next);
        // ---- unsettled queries go to the next stage's list: one global atomic per wave, 64 striped counters
        const bool append = !FINAL && live && !done && lane == 0;
        const u64 bal = __ballot(append);
        if (bal) {
            const int wlane = threadIdx.x & 63;
            const int leader = __ffsll((long long)bal) - 1;
            uint32_t wbase = 0;
            if (wlane == leader) wbase = atomicAdd(out_count + stripe, (uint32_t)__popcll(bal));
            wbase = __shfl(wbase, leader, 64);
            if (append) {
                out_list[wbase + (uint32_t)__popcll(bal & ((1ull << wlane) - 1ull))] = (uint32_t)q;
                ub_out[q] = ub_next;
            }
        }
        if (!BOUNDED) break;  // no input list: single trip (see above)
        wave_sync();  // seg[] is rewritten by the next trip
    }
}


// A3 as a kernel of its own: every query (lanes_per_query == 0, the tests' cross-check) or the queries of a work list
// (grids without a ring-2 stage); 32 lanes per query, 8 queries per block.
__global__ void __launch_bounds__(256)
k_search_exact(GridParams g, StateDev s, const float4* __restrict__ body, int N, float max_sqdist, int rmax,
               float4* __restrict__ nn_pts, uint8_t* __restrict__ nn_cnt,
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
                                                nn_cnt, selected, nullptr);
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
int gram_slots_host(int ncol) { return gram_nslots(ncol); }
int gram_slot_host(int r, int c, int ncol) { return gram_slot(r, c, ncol); }
constexpr int kRed1 = 16;   // blocks per first-level reduction group
constexpr int kRed2 = 32;   // group sums added per unrolled batch at the top level

template <int ORD, bool HALF, int PM = 0>
__global__ void __launch_bounds__(256)
k_fit(StateDev s, const float4* __restrict__ body, const float4* __restrict__ nn_pts, int N, int ext, float thr,
      uint8_t* __restrict__ selected, float4* __restrict__ normvec, float4* __restrict__ world,
      double* __restrict__ partials, double* __restrict__ part2, double* __restrict__ out256, double seq,
      uint32_t* __restrict__ tickets, uint32_t* __restrict__ slow_count, GranOut gout, int red1, int ncol,
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
        for (int j = 0; j < 5; ++j) nn[j] = nn_pts[FLH_IDX(301, (size_t)j * N + ic, (size_t)5 * (N > 0 ? N : 1))];
    }
    if (i < N) {  // feats_down_world is rewritten for every point on every pass (laserMapping.cpp:656-661)
        body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
        if (store_aux) world[i] = make_float4(wx, wy, wz, 0.f);
    }
    if (i < N && sel_in) {  // laserMapping.cpp:674
        float P[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { P[j][0] = nn[j].x; P[j][1] = nn[j].y; P[j][2] = nn[j].z; }
        float pabcd[4], pd2;
        bool ok;
        const bool sel = fit_point<ORD, HALF, PM>(s, b.x, b.y, b.z, wx, wy, wz, P, pc, ext, thr, pabcd, ok, pd2, v);
        if (PM == 1 && ok) plane_cache[i] = make_float4(pabcd[0], pabcd[1], pabcd[2], pabcd[3]);
        selected[i] = sel ? 1 : 0;
        if (sel && store_aux) normvec[i] = make_float4(pabcd[0], pabcd[1], pabcd[2], pd2);  // :686-689
    }
    tile_store(lds + wave * 64 * kTileStride, lane, v);
    __syncthreads();
    const v4f64 acc = tile_gram(lds + wave * 64 * kTileStride, lane);
    const int col = lane & 15, kq = lane >> 4;
    const int t = threadIdx.x;
    const int nblk = gridDim.x;
    if (red1 > 0) {
        // ---- flh_eval's path: every WAVE is a unit of the cross-workgroup sum k_pass uses (64 points each, the same points in
        // the same lanes) and this block is a QUAD of that sum's tree (flh_fit_dev.hpp): its four waves are added in LDS,
        // ((w0 + w1) + w2) + w3, and ONE record goes to memory -- a no-search pass produces the bits a searching pass would at
        // the same state.  red1 = units per group (a multiple of 4).  The statistic slot carries, through unit 0, the number of
        // queries the first search stage listed for the second in this pass (three-launch searching pass; the second stage has
        // retired), after which the work-list counters are re-armed.
        const int nsl = gran_section_slots(ncol);
        const int nunits = (N + 63) / 64 > 0 ? (N + 63) / 64 : 1;
        __syncthreads();  // every wave has read its tile: the same LDS now holds the waves' blocks
        double* Rq = lds;
#pragma unroll
        for (int r = 0; r < 4; ++r) Rq[wave * 256 + (kq + 4 * r) * 16 + col] = acc[r];
        __syncthreads();
        {
            const int slot = gram_slot(t >> 4, t & 15, ncol);  // t = row * 16 + col of the 16x16 block
            if (slot >= 0)
                __hip_atomic_store((gdouble*)partials + (size_t)blockIdx.x * nsl + slot, ((Rq[t] + Rq[256 + t]) + Rq[512 + t]) + Rq[768 + t],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave == 0) {  // the statistic (block 0 only; the other quads carry +0.0)
            double stat = 0.0;
            if (blockIdx.x == 0) {
                uint32_t c = slow_count[lane];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
                stat = (double)c;
                slow_count[lane] = 0;
                slow_count[kStripes + lane] = 0;
            }
            if (lane == 0) __hip_atomic_store((gdouble*)partials + (size_t)blockIdx.x * nsl + (nsl - 1), stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int bpg = red1 / 4;  // blocks (quads) per group
        const int group = blockIdx.x / bpg;
        const int gblocks = min(bpg, nblk - group * bpg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) s_ticket = __hip_atomic_fetch_add(&tickets[1 + group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (uint32_t)(gblocks - 1)) return;  // block-uniform
        if (wave == 0) {
            // (gout.n_dst == 0 -- an RCCL communicator is attached -- : the group's totals stay in device memory, out256[group][slot],
            // as the one-launch searching pass leaves them: both kinds of pass keep one tree)
            group_sum_publish<true>(partials, group, min(red1, nunits - group * red1), red1, nsl, (nunits + red1 - 1) / red1, gout, seq, lane, out256);
            if (lane == 0) tickets[1 + group] = 0;  // re-arm this group's ticket for the next launch
        }
        return;
    }
    __syncthreads();
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    double* Rb = lds;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rb[wave * 256 + (kq + 4 * r) * 16 + col] = acc[r];
    __syncthreads();
    // ---- R (evaluations whose block stays on the device: the RCCL path, flh_eval_device, the fetches): deterministic two-level
    // cross-block sum inside this launch (no reduce kernels, no extra boundaries).  Blocks are grouped kRed1 at a time; the LAST
    // block of a group to finish sums the group's partials in block order, the LAST group to finish sums the group sums in group
    // order and writes the result.  Fixed summation order -> run-to-run identical bits regardless of which block happens to
    // arrive last.  Hand-off as in flh_fit_dev.hpp: write-through stores, vmcnt(0), a relaxed agent-scope ticket, agent-scope loads.
    gdouble* gpart = (gdouble*)partials;
    gdouble* gpart2 = (gdouble*)part2;
    __hip_atomic_store(gpart + (size_t)blockIdx.x * 256 + t, (Rb[t] + Rb[256 + t]) + (Rb[512 + t] + Rb[768 + t]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int group = blockIdx.x / kRed1;
    const int ngroups = (nblk + kRed1 - 1) / kRed1;
    const int gsize = min(kRed1, nblk - group * kRed1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_ticket = __hip_atomic_fetch_add(&tickets[1 + group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (uint32_t)(gsize - 1)) return;  // block-uniform
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
    if (s_ticket != (uint32_t)(ngroups - 1)) return;
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
// (spread3_10 / scan_morton: flh_device.hpp -- flh_stage.hip forms the same key)
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
    const uint32_t src = (uint32_t)FLH_IDX(302, perm ? perm[i] : i, N);
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

hipError_t launch_search(int lpq, const GridParams& g, const StateDev& s, const float4* body, int N, uint32_t map_points,
                         float max_sqdist, int rmax, float4* nn_pts, uint8_t* nn_cnt, uint8_t* selected,
                         uint32_t* list1, uint32_t* list2, float* ub, uint32_t* counts /* [2 * kStripes] */,
                         u64* cand_counter, int own_axis, float own_lo, float own_hi, hipStream_t st,
                         hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (N <= 0) return hipSuccess;
    const dim3 blk(256);
    const uint32_t cap = list_stripe_cap(N);
    const hipEvent_t ev_none = nullptr;
    if (lpq == 0) {  // exact path for every query (validation / fallback)
        FLH_LAUNCH_EV(k_search_exact, dim3(std::min(cdiv(N, 8), 4096)), blk, st, ev_start, ev_stop, g, s, body, N, max_sqdist, rmax,
                      nn_pts, nn_cnt, selected, list1, counts, cap, ub, 1, cand_counter, own_axis, own_lo, own_hi);
        return hipGetLastError();
    }
    // A1: ring 1, every query, four lanes per query
    FLH_LAUNCH_EV((k_search_ring<4, 1, false, 8, false>), dim3(cdiv(N, 64)), blk, st, ev_start, ev_none, g, s, body, N, map_points,
                  max_sqdist, nn_pts, nn_cnt, selected, (const uint32_t*)nullptr, (const uint32_t*)nullptr, list1, counts, cap,
                  (const float*)nullptr, ub, rmax, cand_counter, own_axis, own_lo, own_hi);
    if (rmax >= 2) {
        // A2: ring 2 over list 1, inside the ball A1's 5th distance defines, eight lanes per query (measured on BASELINE
        // configs[1]: 4 lanes 15.5 us, 8 lanes 12.3, 16 lanes 13.3, 32 lanes 16.3, 64 lanes 24.2); whatever it cannot settle
        // (distance ties, a 5th neighbour beyond the 5x5x5 block) it finishes itself; eight blocks per stripe so that a stripe's
        // share of ~10 % of the queries is one trip of its blocks
        FLH_LAUNCH_EV((k_search_ring<8, 2, true, 11, true>), dim3(kStripes * 8), blk, st, ev_none, ev_stop, g, s, body, N, map_points,
                      max_sqdist, nn_pts, nn_cnt, selected, (const uint32_t*)list1, (const uint32_t*)counts, list2, counts + kStripes,
                      cap, (const float*)ub, ub, rmax, cand_counter, -1, 0.f, 0.f);
    } else {
        // cells as large as the gate radius: the general search drains list 1 directly
        FLH_LAUNCH_EV(k_search_exact, dim3(kStripes * 8), blk, st, ev_none, ev_stop, g, s, body, N, max_sqdist, rmax, nn_pts,
                      nn_cnt, selected, (const uint32_t*)list1, (const uint32_t*)counts, cap, ub, 0, cand_counter, -1, 0.f, 0.f);
    }
    return hipGetLastError();
}


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
// The RCCL path of a pass summed in the granules' tree: `totals` holds [group][slot] sums (this rank's group totals, all-reduced over
// the ranks in place; rows of groups this rank does not have are zero).  Every thread owns one entry of the 16 x 16 block, adds
// its slot over ALL kGroups rows in group order -- the host's order over granules, so one rank reproduces flh_eval's bits -- and
// hands the block to the host like k_publish256.  Rows behind this rank's own groups are zeroed again for the next pass (the
// all-reduce left the other ranks' sums there).
__global__ void __launch_bounds__(256) k_publish_groups(double* __restrict__ totals, int ngroups_own, int ngroups_all, int nsl, int ncol,
                                                        double* __restrict__ out256, double seq) {
    const int t = threadIdx.x, r = t >> 4, c = t & 15;
    int slot = gram_slot(r, c, ncol);
    if (slot < 0 && c < 12 && r > c) slot = gram_slot(c, r, ncol);  // the block is symmetric bit for bit
    double sum = 0.0;
    if (slot >= 0) {
        for (int g0 = 0; g0 < ngroups_all; g0 += 32) {
            double v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = (g0 + j < ngroups_all) ? totals[(size_t)(g0 + j) * nsl + slot] : 0.0;
#pragma unroll
            for (int j = 0; j < 32; ++j)
                if (g0 + j < ngroups_all) sum += v[j];
        }
    }
    if (t != 255) __hip_atomic_store(out256 + t, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every entry has been read
    if (t == 0) __hip_atomic_store(out256 + 255, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int i = ngroups_own * nsl + t; i < ngroups_all * nsl; i += 256) totals[i] = 0.0;
}
hipError_t launch_publish_groups(double* totals, int ngroups_own, int ngroups_all, int nsl, int ncol, double* out256, double seq, hipStream_t st) {
    hipLaunchKernelGGL(k_publish_groups, dim3(1), dim3(256), 0, st, totals, ngroups_own, ngroups_all, nsl, ncol, out256, seq);
    return hipGetLastError();
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
// The neighbour cache of a one-launch searching pass holds map indices (flh_config.index_cache): whoever needs the coordinates
// (map_incremental, a fetch, a re-fit without the plane cache) has them gathered once, off the update's critical path.
__global__ void __launch_bounds__(256) k_nn_gather(const float4* __restrict__ map_orig, uint32_t n_ids, const uint32_t* __restrict__ nn_idx,
                                                   int total, float4* __restrict__ nn_pts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t id = nn_idx[i];
    float4 v = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (id < n_ids) {
        v = map_orig[id];
        v.w = __uint_as_float(id);
    }
    nn_pts[i] = v;
}
hipError_t launch_nn_gather(const float4* map_orig, uint32_t n_ids, const uint32_t* nn_idx, int N, float4* nn_pts, hipStream_t st) {
    if (N <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_nn_gather, dim3(cdiv(5ll * N, 256)), dim3(256), 0, st, map_orig, n_ids, nn_idx, 5 * N, nn_pts);
    return hipGetLastError();
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
                      double* out256, double seq, uint32_t* tickets, uint32_t* slow_count, const GranOut& gran, int red1, int store_aux,
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

#ifdef FLH_BOUNDS
void bounds_read_kernels(unsigned long long out[5]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(BoundsRec)); }
#endif

}  // namespace flh

// ------------------------------------------------------------------------------------------------
// k_fit_mb: the no-search pass of flh_eval (k_fit<1, false, 2> on its granule path: plane cache, default summation order, group
// sums as granules) as a kernel that is enqueued BEFORE its state is known (flh_mail_dev.hpp; flh_eval_expect_next).  What does
// not depend on the state -- a point's flag, its body-frame coordinates, its cached plane -- is loaded before the wait, so that
// when the state arrives only arithmetic is left.  From the state on the code is k_fit's, statement for statement: same units,
// same quads, same summation tree -> the bits of k_fit<1, false, 2> at the same state (tests/test_gpu_parity.py).
// ------------------------------------------------------------------------------------------------
namespace flh {

__global__ void __launch_bounds__(256)
k_fit_mb(MailArgs mail, const float4* __restrict__ body, int N, int ext, float thr, uint8_t* __restrict__ selected,
         double* __restrict__ partials, double seq, uint32_t* __restrict__ tickets, uint32_t* __restrict__ slow_count, GranOut gout,
         int red1, int ncol, const float4* __restrict__ plane_cache) {
    __shared__ double lds[4 * 64 * kTileStride];
    __shared__ uint32_t s_ticket;
    __shared__ uint32_t s_cmd;
    __shared__ double s_box[16];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ic = i < N ? i : (N > 0 ? N - 1 : 0);
    const uint8_t sel_in = selected[ic];
    const float4 b = body[ic];
    const float4 pc = plane_cache[ic];
    StateDev s;
    if (!mailbox_wait(mail, s, &s_cmd, s_box)) return;  // aborted by the host, or nobody came: nothing was written

    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
    float wx = 0.f, wy = 0.f, wz = 0.f;
    if (i < N) body_to_world(s, b.x, b.y, b.z, wx, wy, wz);
    if (i < N && sel_in) {  // laserMapping.cpp:674
        float P[5][3];
#pragma unroll
        for (int j = 0; j < 5; ++j) { P[j][0] = 0.f; P[j][1] = 0.f; P[j][2] = 0.f; }
        float pabcd[4], pd2;
        bool ok;
        const bool sel = fit_point<1, false, 2>(s, b.x, b.y, b.z, wx, wy, wz, P, pc, ext, thr, pabcd, ok, pd2, v);
        selected[i] = sel ? 1 : 0;
    }
    tile_store(lds + wave * 64 * kTileStride, lane, v);
    __syncthreads();
    const v4f64 acc = tile_gram(lds + wave * 64 * kTileStride, lane);
    const int col = lane & 15, kq = lane >> 4;
    const int t = threadIdx.x;
    const int nblk = gridDim.x;
    const int nsl = gran_section_slots(ncol);
    const int nunits = (N + 63) / 64 > 0 ? (N + 63) / 64 : 1;
    __syncthreads();  // every wave has read its tile: the same LDS now holds the waves' blocks
    double* Rq = lds;
#pragma unroll
    for (int r = 0; r < 4; ++r) Rq[wave * 256 + (kq + 4 * r) * 16 + col] = acc[r];
    __syncthreads();
    {
        const int slot = gram_slot(t >> 4, t & 15, ncol);
        if (slot >= 0)
            __hip_atomic_store((gdouble*)partials + (size_t)blockIdx.x * nsl + slot, ((Rq[t] + Rq[256 + t]) + Rq[512 + t]) + Rq[768 + t],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {  // the statistic slot, as k_fit keeps it
        double stat = 0.0;
        if (blockIdx.x == 0) {
            uint32_t c = slow_count[lane];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
            stat = (double)c;
            slow_count[lane] = 0;
            slow_count[kStripes + lane] = 0;
        }
        if (lane == 0) __hip_atomic_store((gdouble*)partials + (size_t)blockIdx.x * nsl + (nsl - 1), stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int bpg = red1 / 4;
    const int group = blockIdx.x / bpg;
    const int gblocks = min(bpg, nblk - group * bpg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_ticket = __hip_atomic_fetch_add(&tickets[1 + group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (uint32_t)(gblocks - 1)) return;
    if (wave == 0) {
        group_sum_publish<true>(partials, group, min(red1, nunits - group * red1), red1, nsl, (nunits + red1 - 1) / red1, gout, seq, lane);
        if (lane == 0) tickets[1 + group] = 0;
    }
}

hipError_t launch_fit_mb(const MailArgs& mail, const float4* body, int N, int ext, float thr, uint8_t* selected, double* partials,
                         double seq, uint32_t* tickets, uint32_t* slow_count, const GranOut& gran, int red1, const float4* plane_cache,
                         hipStream_t st) {
    if (N <= 0 || !plane_cache || gran.n_dst < 1 || red1 < 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_fit_mb, dim3(fit_blocks(N)), dim3(256), 0, st, mail, body, N, ext, thr, selected, partials, seq, tickets,
                       slow_count, gran, red1, ext ? 12 : 6, plane_cache);
    return hipGetLastError();
}

}  // namespace flh
