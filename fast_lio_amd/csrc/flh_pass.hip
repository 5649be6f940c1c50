// flh_pass.hip -- k_pass: a SEARCHING pass of the iterated update as ONE launch (gfx950 / CDNA4, wave64).
//
// The reference does everything a scan point needs in one loop body -- transform, Nearest_Search, the kNN gate, esti_plane, the
// residual gate (src/laserMapping.cpp:650-693) -- then builds the rows (:720-752) and the filter forms H^T H, H^T h
// (esekfom.hpp:1784,1804).  Rounds 1-3 ran that as three dependent launches (first search stage -> second stage -> fit) with
// the neighbours written to HBM and read back in between.  Here a workgroup owns 64 consecutive (Morton-ordered) scan points
// from the transform to their share of the normal equations:
//
//   phase A   ring_query<4,1>: four lanes per query over the 3x3x3 cell block, all 64 queries at once.  Results go to the neighbour cache in HBM --
//             as map indices by default (flh_config.index_cache), map_incremental and the fetches gather coordinates from them --
//             AND stay in LDS for the fit.
//   phase B   the queries phase A could not settle (5th neighbour not provably inside the block: at the prior, the far field of
//             the scan; a handful later) are searched again by the same workgroup, eight lanes per query, over a 4x4 window of
//             the 5x5x5 block's rows clipped to the ball of phase A's bound (ring_query<8,2,...,WIN4>, which also finishes
//             distance ties with 64-bit keys).
//             Workgroup-local on purpose: no global work list, no second launch, no spin-waits, and the order in which rows
//             enter the sums does not depend on timing.
//   fit       ONE wave takes the 64 queries' neighbours from LDS (lane = query): esti_plane, residual, gate, Jacobian row
//             (fit_point, the reference's expressions), keeps the plane for the no-search passes (plane cache), contracts the
//             64 rows into a 16x16 Gram block on v_mfma_f64_16x16x4_f64 and stores the entries the filter reads; the other
//             three waves have retired by then.
//   reduce    workgroups are grouped `red` at a time; the last one of a group to finish sums the group's partials in workgroup
//             order and hands them to the host as 16-byte {value, sequence} granules in pinned memory (to every rank's buffer
//             when the scan is sharded over GPUs); the host adds the groups in group order.  With an RCCL communicator the
//             group sums stay in device memory, RCCL adds the ranks' and the publish kernel the groups, in the same order.
//             Fixed order -> identical bits run to run.
//
// Requires cells >= sqrt(max_sqdist) / 1.499 (1.5 m for the default gate: phase B's 4x4 window of rows then covers the ball of a
// bounded query, so it settles everything it is given; flh_create checks -- pass_ok -- and otherwise runs the three-launch pass of
// flh_kernels.hip).
#include "flh_kernels.hpp"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "flh_device.hpp"
#include "flh_search_dev.hpp"
#include "flh_fit_dev.hpp"

namespace flh {

// Registers: the kernel is held to 72 VGPRs = 7 waves per SIMD (a scan of 100 000 points is 6 252 waves on 1 024 SIMDs: one
// round); FLH_PASS_WAVES selects another bound for A/B builds (tools/variant.py).
// phase B keeps four point loads in flight per lane (eight lanes per query: 32 per query, as phase A's 4 x 8); with eight the
// kernel does not fit 72 registers
#ifndef FLH_UNR_B
#define FLH_UNR_B 4
#endif
#ifndef FLH_PASS_WAVES
#define FLH_PASS_WAVES 7
#endif
// (the wide variants run where the GPU is under-filled: one 1024-thread workgroup per CU is four waves per SIMD, registers are free)
#define PASS_ATTR __attribute__((amdgpu_waves_per_eu(FLH_PASS_WAVES, FLH_PASS_WAVES)))
constexpr int kPassQueries = 64;                       // scan points per workgroup
constexpr int kSegA = ring_seg_slots<1>();             // 20 LDS table entries per phase-A group (64 groups)
constexpr int kSegB = 2 * 16 + 2;                      // 34 per phase-B group (32 groups of eight lanes, 4x4 window)
// Four lanes per query in phase A (256 threads per unit).  Round 5 built the kernel for 8 and 16 lanes per query as well (512 /
// 1024 threads per unit) for scans and shards that leave the GPU under-filled, and measured it (profiles/r05_call2/exchange_probe.txt,
// a rank's share of an 8-way shard, 12 500 points: 38.0 / 36.6 / 39.1 us per searching pass at 4 / 8 / 16 lanes; 25 000 points:
// 40.1 / 39.9 / 50.9): no gain -- a small pass is the latency chain of one workgroup (launch, dependent misses, fit, ticket, group
// sum, publish: 22 us for 64 points), not phase A's issue -- so the wide variants were deleted again.
// Also built, measured and deleted in round 5: the pre-launched form of this kernel (k_pass<1, true>: enqueued beside the pass before it,
// every workgroup waiting for its state in a mailbox, as k_fit_mb does for the no-search pass).  With 1 563 workgroups waiting the
// hand-over is slower than a launch: searching pass 47.6 -> 49.7 us, value 7 605 -> 7 389 scans/s, three alternating pairs
// (profiles/r05_call5/) -- what tools/mailbox_probe.cpp had predicted (profiles/r05_call1/mailbox_probe.txt).
constexpr int kSegWords = (64 * kSegA > 32 * kSegB) ? 64 * kSegA : 32 * kSegB;
static_assert(kSegWords * 8 >= 64 * kTileStride * 8, "the fit's transpose tile reuses the segment tables");
// Developer instrumentation (tools/variant.py --define FLH_PASS_STAMPS, tools/pass_stamps.py): per-wave 100 MHz time stamps of the
// phases of k_pass, read back with flh_debug_pass_stamps.  Compiled out of the product.
#ifdef FLH_PASS_STAMPS
constexpr int kStampWaves = 16384;
constexpr int kStampWords = 12;  // 0..7 time stamps; 8 HW_ID, 9 XCC_ID, 10 the longest candidate list of the wave's queries, 11 its open queries
__device__ unsigned long long g_pass_stamps[kStampWaves * kStampWords];
#define STAMP_VAL(i, val)                                                                                                \
    do {                                                                                                                 \
        if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < kStampWaves)                                \
            g_pass_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * kStampWords + (i)] = (val);                            \
    } while (0)
#define STAMP(i) STAMP_VAL(i, __builtin_amdgcn_s_memrealtime())
void pass_stamps_read(unsigned long long* out, size_t words) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pass_stamps), sizeof(unsigned long long) * (words < (size_t)kStampWaves * kStampWords ? words : (size_t)kStampWaves * kStampWords));
}
#else
#define STAMP(i)
#endif

template <int ORD>
__global__ void __launch_bounds__(256) PASS_ATTR
k_pass(GridParams g, StateDev s, const float4* __restrict__ body, int N, uint32_t map_points, float max_sqdist, float thr, int ext,
       int ncol, float4* __restrict__ nn_pts, uint8_t* __restrict__ nn_cnt, uint8_t* __restrict__ selected,
       float4* __restrict__ plane_cache, double* __restrict__ partials, uint32_t* __restrict__ tickets, GranOut gout, double seq,
       int red, u64* __restrict__ cand_counter, int own_axis, float own_lo, float own_hi, uint32_t* __restrict__ nn_idx,
       double* __restrict__ group_totals) {
    constexpr int LPQ = 4;           // lanes per query in phase A
    constexpr int NW = 4;            // waves of the workgroup
    constexpr int QPW = 64 / LPQ;    // queries per wave in phase A
    __shared__ uint2 segs[kSegWords];
    __shared__ float park[kPassQueries * kParkStride];
    __shared__ uint32_t s_list[64];  // per wave: the slots (0..63) of its queries that go to phase B
    __shared__ uint32_t s_wcnt[NW];
    __shared__ uint32_t s_last;      // (extrinsic columns) this workgroup is its group's last arriver: three of its waves take the group sum
    const int tid = threadIdx.x;
    // which 64 scan points (a unit of the summation tree) this workgroup takes: the LAST ones first.  Workgroups are dispatched in
    // blockIdx order over ~2 us, and the end of the scan's Morton order is its far field, whose waves are the slowest of the launch
    // (sparse queries: every dependent load of the search misses) -- they decide when the kernel ends, so they start first.
    // Measured on BASELINE configs[1], same box, two pairs: 41.9 -> 38.8 us per launch (profiles/r04_call8/).
#ifdef FLH_PASS_FORWARD  // (developer A/B builds)
    const int unit = (int)blockIdx.x;
#else
    const int unit = (int)gridDim.x - 1 - (int)blockIdx.x;
#endif
    const int q0 = unit * kPassQueries;
    const RingRsrc rs(g, map_points);
    STAMP(0);  // start

    // ---- phase A: every query of the workgroup, four lanes each
    {
        const int grp = tid / LPQ, lane = tid & (LPQ - 1), wave = tid >> 6, wl = tid & 63;
        bool live = q0 + grp < N;
        const int q = live ? q0 + grp : N - 1;
        const float4 b = body[q];
        float qx, qy, qz;
        body_to_world(s, b.x, b.y, b.z, qx, qy, qz);  // src/laserMapping.cpp:656-660
        if (own_axis >= 0) {  // map partitioned over ranks (flh_set_owned_interval): another rank's query
            const float oc = own_axis == 0 ? qx : (own_axis == 1 ? qy : qz);
            if (live && !(oc >= own_lo && oc < own_hi)) {
                if (lane == 0) nn_cnt[q] = 0;
                live = false;
            }
        }
        float* pk = park + grp * kParkStride;
        if (lane == 0) {
            pk[kParkWorld] = qx; pk[kParkWorld + 1] = qy; pk[kParkWorld + 2] = qz;
            pk[kParkBody] = b.x; pk[kParkBody + 1] = b.y; pk[kParkBody + 2] = b.z;
        }
        float ub_next;
        const bool done = ring_query<LPQ, 1, false, 8, false, false>(g, rs, segs + grp * kSegA, lane, q, N, live, qx, qy, qz, INFINITY,
                                                                     max_sqdist, 2, nn_pts, nn_cnt, selected, cand_counter, pk, ub_next, nn_idx);
        const bool open = live && !done;
        if (lane == 0 && !(live && done)) {  // a settled query's status was written with its fifth neighbour
            pk[kParkStatus] = __uint_as_float(open ? kStOpen : kStIdle);
            pk[kParkUb] = ub_next;
        }
        const u64 bal = __ballot(open && lane == 0);
        if (open && lane == 0) s_list[FLH_IDX(401, wave * QPW + __popcll(bal & ((1ull << wl) - 1ull)), 64)] = (uint32_t)grp;
        if (wl == 0) s_wcnt[wave] = (uint32_t)__popcll(bal);
#ifdef FLH_PASS_STAMPS
        {
            float tm = lane == 0 ? pk[23] : 0.f;  // ring_query left the length of the query's candidate list there
            for (int off = 32; off >= 1; off >>= 1) tm = fmaxf(tm, __shfl_xor(tm, off, 64));
            STAMP_VAL(8, (unsigned long long)__builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11)));
            STAMP_VAL(9, (unsigned long long)__builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11)));
            STAMP_VAL(10, (unsigned long long)tm);
            STAMP_VAL(11, (unsigned long long)__popcll(bal));
        }
#endif
    }
    STAMP(1);  // this wave's phase A done
    __syncthreads();
    STAMP(2);  // the workgroup's phase A done

    // ---- phase B: the workgroup's open queries, eight lanes each, 32 per trip
    uint32_t n_open = 0;  // workgroup-uniform
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) n_open += s_wcnt[w_];
#ifndef NO_PHASE_B
    if (n_open) {
        for (uint32_t k0 = 0; k0 < n_open; k0 += 8 * LPQ) {
            // (normally one trip: the lane-dependent set-up is kept inside it instead of being hoisted into registers that
            // would have to live across the whole loop)
            int tid_b = tid;
            asm volatile("" : "+v"(tid_b));
            const int grp = tid_b >> 3, lane = tid_b & 7;
            const bool live = k0 + grp < n_open;
            uint32_t idx = live ? k0 + grp : 0u, w = 0;
            for (uint32_t cw = s_wcnt[0]; idx >= cw && w + 1 < (uint32_t)NW; cw = s_wcnt[w]) { idx -= cw; ++w; }  // which wave's list, and where in it
            const uint32_t slot = s_list[FLH_IDX(402, w * QPW + idx, 64)] & 63u;
            float* pk = park + slot * kParkStride;
            const float qx = pk[kParkWorld], qy = pk[kParkWorld + 1], qz = pk[kParkWorld + 2];
            const float ub = pk[kParkUb];
            float ub_next;
            const bool done = ring_query<8, 2, true, 11, true, false, true, FLH_UNR_B>(g, rs, segs + grp * kSegB, lane, q0 + (int)slot, N, live, qx, qy, qz,
                                                                      ub, max_sqdist, 2, nn_pts, nn_cnt, selected, cand_counter, pk, ub_next, nn_idx);
            (void)done;  // always settled: the block covers the gate radius (see the head of this file); a status left open is not fitted
            wave_sync();  // the segment tables are rewritten by the next trip
        }
        __syncthreads();
    }
#endif
    STAMP(3);  // phase B done (= stamp 2 without open queries)

    // ---- fit: one wave, lane = query.  Which wave: spread over the SIMDs (a workgroup's waves land on the four SIMDs of a CU
    // in order, so a fixed choice would load one SIMD of every CU with all the fits)
    // (lane and wave are derived again rather than kept in registers across phase B)
    int tid_f = threadIdx.x;
    asm volatile("" : "+v"(tid_f));
    const int fitw = (int)(((unsigned)unit >> 3) & 3u);
    const int myw = __builtin_amdgcn_readfirstlane(tid_f >> 6);
    // With the extrinsic columns (93 slots = three 32-slot trips of the group sum) the other waves stay: if this workgroup turns out to
    // be its group's last arriver, two of them take a trip each beside the fit wave's (flh_fit_dev.hpp: group_sum_trip)
    const bool helpers = ncol == 12;
    if (myw != fitw) {
        if (!helpers) return;
        __syncthreads();  // (1) the fit wave's verdict
        if (!s_last) return;
        const int hidx = (myw - fitw - 1) & 3;  // 0, 1, 2
        if (hidx < 2) {
            const int nsl_h = gran_section_slots(ncol);
            const int group_h = unit / red;
            const int gsize_h = min(red, (int)gridDim.x - group_h * red);
            group_sum_trip(partials, group_h, gsize_h, red, nsl_h, gout, seq, tid_f & 63, 1 + hidx, group_totals);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();  // (2) the three trips are out: the header may follow
        return;
    }
    const int wl = tid_f & 63;
    const int qf = q0 + wl;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = 0.0;
    {
        const float* pk = park + wl * kParkStride;
        const uint32_t st = __float_as_uint(pk[kParkStatus]);
        if (qf < N) {
            bool sel = false;
            if (st == kStFit) {  // laserMapping.cpp:674
                float P[5][3];
#pragma unroll
                for (int j = 0; j < 5; ++j) { P[j][0] = pk[3 * j]; P[j][1] = pk[3 * j + 1]; P[j][2] = pk[3 * j + 2]; }
                float pabcd[4], pd2;
                bool ok;
                const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
                sel = fit_point<ORD, false, 1>(s, pk[kParkBody], pk[kParkBody + 1], pk[kParkBody + 2], pk[kParkWorld], pk[kParkWorld + 1],
                                               pk[kParkWorld + 2], P, none, ext, thr, pabcd, ok, pd2, v);
                if (ok && plane_cache) plane_cache[qf] = make_float4(pabcd[0], pabcd[1], pabcd[2], pabcd[3]);
            }
            selected[qf] = sel ? 1 : 0;
        }
    }
    double* T = reinterpret_cast<double*>(segs);
    wave_sync();
    tile_store(T, wl, v);
    wave_sync();
    const v4f64 acc = tile_gram(T, wl);
    STAMP(4);  // fit + Gram done

    // ---- this workgroup's share of the normal equations -> HBM, the group's ticket, and for the last arriver the group's sum
    const int nsl = gran_section_slots(ncol);  // the last one: the number of queries that needed phase B (a statistic the host reports)
    uint32_t n_b = 0;  // (read again from LDS rather than kept in a register)
#pragma unroll
    for (int w_ = 0; w_ < NW; ++w_) n_b += s_wcnt[w_];
    unit_partial_store(partials, unit, nsl, ncol, acc, wl, (double)n_b);
    const int nblk = gridDim.x;
    const int group = unit / red;
    const int gsize = min(red, nblk - group * red);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(5);  // partial stored and drained
    uint32_t tk = 0;
    if (wl == 0) tk = __hip_atomic_fetch_add(&tickets[1 + group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    STAMP(6);  // ticket taken
    if (helpers) {
        const bool last = tk == (uint32_t)(gsize - 1);
        if (wl == 0) s_last = last ? 1u : 0u;
        __syncthreads();  // (1)
        if (!last) return;
        group_sum_trip(partials, group, gsize, red, nsl, gout, seq, wl, 0, group_totals);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // (2)
        if (gout.n_dst > 0 && wl == 0 && group == 0) publish_granule(gout, 0, (double)(((nblk + red - 1) / red) * nsl), seq);  // the section's header: the pass's last granule
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(7);
        if (wl == 0) tickets[1 + group] = 0;  // re-arm this group's ticket for the next launch
        return;
    }
    if (tk != (uint32_t)(gsize - 1)) return;
    // (gout.n_dst == 0 -- an RCCL communicator is attached -- : the group's totals stay in device memory, group_totals[group][slot];
    // the ranks' totals are all-reduced and the publish kernel adds the groups in the host's order, flh_kernels.hip)
    group_sum_publish<false>(partials, group, gsize, red, nsl, (nblk + red - 1) / red, gout, seq, wl, group_totals);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(7);  // (the group's last arriver) group sum published
    if (wl == 0) tickets[1 + group] = 0;  // re-arm this group's ticket for the next launch
}

int pass_blocks(int N) { return ((N > 0 ? N : 1) + kPassQueries - 1) / kPassQueries; }
// workgroups per reduction group: 64, more when that would make more than max_groups groups
int pass_group_size(int N, int max_groups) {
    const int nblk = pass_blocks(N);
    int red = 64;
    while ((nblk + red - 1) / red > max_groups) red *= 2;
    return red;
}

hipError_t launch_pass(int order, const GridParams& g, const StateDev& s, const float4* body, int N, uint32_t map_points, float max_sqdist,
                       float thr, int ext, float4* nn_pts, uint8_t* nn_cnt, uint8_t* selected, float4* plane_cache, double* partials,
                       uint32_t* tickets, const GranOut& out, double seq, int red, unsigned long long* cand_counter,
                       int own_axis, float own_lo, float own_hi, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, uint32_t* nn_idx,
                       double* group_totals) {
    if (N <= 0 || out.n_dst < 0 || out.n_dst > kPeersMax || (out.n_dst == 0 && !group_totals)) return hipErrorInvalidValue;
    const int ncol = ext ? 12 : 6;
    const dim3 grid(pass_blocks(N)), blk(256);
#define FLH_PASS(O)                                                                                                                   \
    do {                                                                                                                              \
        if (ev_start != nullptr || ev_stop != nullptr)                                                                                \
            hipExtLaunchKernelGGL((k_pass<O>), grid, blk, 0, st, ev_start, ev_stop, 0, g, s, body, N, map_points, max_sqdist, thr, ext, \
                                  ncol, nn_pts, nn_cnt, selected, plane_cache, partials, tickets, out, seq, red, cand_counter, own_axis, \
                                  own_lo, own_hi, nn_idx, group_totals);                                                               \
        else                                                                                                                          \
            hipLaunchKernelGGL((k_pass<O>), grid, blk, 0, st, g, s, body, N, map_points, max_sqdist, thr, ext, ncol, nn_pts, nn_cnt, \
                               selected, plane_cache, partials, tickets, out, seq, red, cand_counter, own_axis, own_lo, own_hi, nn_idx, \
                               group_totals);                                                                                         \
    } while (0)
    switch (order) {
        case 0: FLH_PASS(0); break;
        case 2: FLH_PASS(2); break;
        case 3: FLH_PASS(3); break;
        default: FLH_PASS(1); break;
    }
#undef FLH_PASS
    return hipGetLastError();
}

#ifdef FLH_BOUNDS
void bounds_read_pass(unsigned long long out[5]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(BoundsRec)); }
#endif

}  // namespace flh
