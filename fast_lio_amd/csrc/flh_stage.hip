// flh_stage.hip -- the scan's staging as TWO launches of this library's own: records -> Morton key -> stable order -> Morton-ordered
// float4 scan.  It replaces k_scan_restride + hipcub::DeviceRadixSort::SortPairs (rocPRIM: block sort + 7 merge passes + 2
// transforms) + k_scan_gather -- 12 launches, 75-90 us of copy-stream kernels per scan, the only library code of the timed region
// (VERDICT r5 item 1).  The hand-over it serves is the node's  downSizeFilterSurf.filter(*feats_down_body)  ->
// feats_down_size  (src/laserMapping.cpp:904-905, 935-951): the order of feats_down_body is free as long as every per-point
// product goes back by its original index, which body[i].w carries.
//
//   k_stage_tile_sort   one workgroup (1 024 threads) per TILE = 4 096 (8 192 above 131 072 points) consecutive records: reads the
//                       records as they came over PCIe (xyz first, any stride that is a multiple of 4 bytes), forms the key, and
//                       sorts (key, position) in LDS by a stable LSD radix sort, 8 bits per pass.  Per wave and digit one LDS cell
//                       {lane mask, count}: the 64 lanes of a row OR their lane bit into their digit's mask (order-independent,
//                       hence deterministic), read it back -- the lanes below me with my digit are my rank inside the row, the
//                       count is what earlier rows left -- and the lowest lane of each digit adds the row and clears the mask.
//                       No ballots (an 8-bit match by ballots costs ~50 VALU instructions per key; round 6 measured 36 us for
//                       100 000 keys that way, profiles/r06_call1/), no order-dependent atomics: equal digits keep the order of
//                       their positions.  Then a digit-major exclusive scan over (digit, wave), scatter, read back.  Besides the
//                       sorted (key, original index) of its tile the kernel leaves every 32nd sorted key (the tile's SAMPLES).
//   k_stage_merge_rank  every element of a sorted tile finds its place in the whole: its position in its own tile + for every
//                       other tile the number of elements that precede it -- keys <= its own in EARLIER tiles, keys < its own in
//                       LATER tiles, which is exactly the stable order of the composite (key, original index) because tiles are
//                       consecutive index ranges.  The samples of all tiles sit in LDS: a branch-free binary search there narrows
//                       each tile to 32 elements, a second one of six loads in global memory finishes; eight tiles in flight per
//                       lane, fixed trip counts.  The element's record is gathered and written to its place as float4, .w =
//                       original index.
//
// Same key, same stable order as the radix sort it replaces => the same units, the same bits downstream (tests/test_gpu_staging.py;
// every staged-vs-uploaded test).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flh_device.hpp"
#include "flh_kernels.hpp"

namespace flh {

constexpr int kStageSample = 32;       // every 32nd sorted key of a tile is a sample
constexpr int kStageTilesMax = 32;     // tiles of a scan at most; the samples of all of them sit in LDS (32 x 128 or 32 x 256 words)
constexpr uint32_t kStageSmallMax = 32u * 4096u;  // up to here tiles of 4 096 records (E = 4), above tiles of 8 192 (E = 8)

uint32_t stage_sort_max() { return 32u * 8192u; }  // 262 144 points; above it the library sort
static inline uint32_t stage_tile(uint32_t N) { return N <= kStageSmallMax ? 4096u : 8192u; }
uint32_t stage_sample_words(uint32_t N) {
    const uint32_t tile = stage_tile(N);
    return ((N + tile - 1u) / tile) * (tile / kStageSample);
}

__device__ __forceinline__ void wave_fence() {  // LDS hand-offs inside one wave: LDS keeps a wave's order, the compiler must too
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct __attribute__((aligned(16))) RankCell {
    unsigned long long mask;  // lanes of the current row with this digit
    uint32_t cnt;             // keys with this digit in the wave's earlier rows; after the scan: the wave's first place inside the digit
    uint32_t pad;
};

template <int E>
__global__ void __launch_bounds__(1024)
k_stage_tile_sort(const uint32_t* __restrict__ words, uint32_t stride_words, uint32_t N, float inv_q,
                  uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_idx, uint32_t* __restrict__ samples) {
    constexpr int TILE = 1024 * E;
    __shared__ uint2 s_kv[TILE];
    __shared__ RankCell s_rank[16 * 256];  // [wave][digit]
    __shared__ uint32_t s_dbase[256];      // first place of a digit
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t base = blockIdx.x * (uint32_t)TILE;
    const uint32_t n_t = min((uint32_t)TILE, N - base);
    const unsigned long long my_bit = 1ull << lane, lt_mask = my_bit - 1ull;

    uint32_t key[E], val[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {  // wave-striped: a row of 64 consecutive records per wave and e
        const uint32_t pos = (uint32_t)(wave * 64 * E + e * 64 + lane);
        val[e] = pos;
        key[e] = 0xFFFFFFFFu;  // padding sorts behind every record (stable: it also comes last by position)
        if (pos < n_t) {
            const uint32_t* r = words + (size_t)(base + pos) * stride_words;
            key[e] = scan_morton(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), inv_q);
        }
    }
    RankCell* cells = s_rank + wave * 256;  // this wave's cells: touched by this wave only until the scan
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        cells[d * 64 + lane].mask = 0ull;
        cells[d * 64 + lane].cnt = 0u;
    }
    for (int shift = 0; shift < 32; shift += 8) {
        uint32_t lr[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {  // rows in order; LDS executes a wave's operations in order
            RankCell* c = cells + ((key[e] >> shift) & 255u);
            __hip_atomic_fetch_or(&c->mask, my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            const unsigned long long peers = __hip_atomic_load(&c->mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const uint32_t prior = __hip_atomic_load(&c->cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            lr[e] = prior + (uint32_t)__popcll(peers & lt_mask);
            __builtin_amdgcn_wave_barrier();
            if ((peers & lt_mask) == 0ull) {  // the digit's lowest lane: the row joins the count, the mask is clear for the next row
                __hip_atomic_store(&c->cnt, prior + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(&c->mask, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        uint32_t incl = 0, total = 0;
        if (tid < 256) {  // digit = tid: the waves' counts of the digit become exclusive offsets; then a scan over the digits
            uint32_t c[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) c[w] = s_rank[w * 256 + tid].cnt;
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                s_rank[w * 256 + tid].cnt = run;
                run += c[w];
            }
            total = run;
            incl = total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (lane == 63) s_wsum[wave] = incl;
            if (tid == 0) s_skip = 0u;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t pre = 0;
            for (int w = 0; w < wave; ++w) pre += s_wsum[w];
            s_dbase[tid] = pre + incl - total;
            if (total == (uint32_t)TILE) s_skip = 1u;  // every key has this digit: the pass would move nothing
        }
        __syncthreads();
        const bool skip = s_skip != 0u;  // (uniform)
        uint32_t dst[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t digit = (key[e] >> shift) & 255u;
            dst[e] = s_dbase[digit] + cells[digit].cnt + lr[e];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 0; d < 4; ++d) cells[d * 64 + lane].cnt = 0u;  // (this wave's cells, read above by this wave only) for the next pass
        if (skip) continue;  // (the next pass's first barrier orders its s_skip write behind this read)
#pragma unroll
        for (int e = 0; e < E; ++e) s_kv[FLH_IDX(310, dst[e], TILE)] = make_uint2(key[e], val[e]);
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint2 kv = s_kv[wave * 64 * E + e * 64 + lane];
            key[e] = kv.x;
            val[e] = kv.y;
        }
        // (no barrier here: three barriers lie between this read-back and the next scatter into s_kv)
    }
    uint32_t* smp = samples + (size_t)blockIdx.x * (TILE / kStageSample);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t pos = (uint32_t)(wave * 64 * E + e * 64 + lane);
        if (pos < n_t) {
            tile_keys[base + pos] = key[e];
            tile_idx[base + pos] = base + val[e];
        }
        if ((lane & (kStageSample - 1)) == 0) smp[pos / kStageSample] = key[e];  // (padding included: 0xFFFFFFFF)
    }
}

// rank of k among the first n (<= S = 2^m) entries of a sorted LDS array: how many are <= k (le) / < k -- branch-free, m + 1 reads
template <uint32_t S>
__device__ __forceinline__ uint32_t count_before(const uint32_t* a, uint32_t n, uint32_t k, bool le) {
    uint32_t c = 0;
#pragma unroll
    for (uint32_t h = S / 2; h >= 1u; h >>= 1) {
        const uint32_t i = c + h - 1u;
        const uint32_t v = a[i];
        c += (i < n && (le ? (v <= k) : (v < k))) ? h : 0u;
    }
    const uint32_t v = a[c < S ? c : S - 1u];
    c += (c < n && (le ? (v <= k) : (v < k))) ? 1u : 0u;
    return c;
}

template <int TILE>
__global__ void __launch_bounds__(256)
k_stage_merge_rank(const uint32_t* __restrict__ words, uint32_t stride_words, uint32_t N, const uint32_t* __restrict__ tile_keys,
                   const uint32_t* __restrict__ tile_idx, const uint32_t* __restrict__ samples, float4* __restrict__ body) {
    constexpr uint32_t SPT = TILE / kStageSample;  // samples per tile: 128 / 256
    constexpr int C = 8;                           // tiles per batch: their windows are loaded together, searched side by side
    constexpr uint32_t WCAP = 128;                 // window entries per tile and batch held in LDS
    __shared__ uint32_t s_smp[kStageTilesMax * SPT];  // 16 KB (32 KB with the large tile)
    __shared__ uint32_t s_lo[4][kStageTilesMax], s_hi[4][kStageTilesMax];
    __shared__ uint32_t s_win[4][C][WCAP];            // 16 KB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t t = gid / TILE, p = gid % TILE;
    const uint32_t T = (N + TILE - 1u) / TILE;
    const uint32_t n_own = min((uint32_t)TILE, N - t * TILE);
    if ((p & ~255u) >= n_own) return;  // (the whole workgroup: nothing of its tile left)
    for (uint32_t i = threadIdx.x; i < T * SPT; i += 256u) s_smp[i] = samples[i];
    __syncthreads();
    const bool valid = p < n_own;
    const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
    if (vm == 0ull) return;  // (no workgroup barrier below: a wave may leave)
    const uint32_t k = valid ? tile_keys[t * TILE + p] : 0xFFFFFFFFu;
    // ---- the wave's 64 elements are consecutive in their sorted tile: every other tile is bracketed between the counts of the
    // wave's smallest and largest key -- exactly: lane j < 32 counts tile j against the smallest, lane j >= 32 tile j - 32
    // against the largest (samples in LDS narrow the tile to 32 elements, six loads finish)
    {
        const uint32_t kmin = __shfl(k, 0, 64), kmax = __shfl(k, __popcll(vm) - 1, 64);
        const uint32_t tj = (uint32_t)lane & 31u;
        const bool upper = lane >= 32;
        const bool act = tj < T && tj != t;
        const uint32_t ts = act ? tj : t;
        const uint32_t kq = upper ? kmax : kmin;
        const bool le = tj < t;  // earlier tile: its equal keys come first
        const uint32_t n_tj = act ? min((uint32_t)TILE, N - tj * TILE) : 0u;
        const uint32_t c = count_before<SPT>(s_smp + ts * SPT, SPT, kq, le);
        // samples 0 .. c-1 precede kq: at least 32 (c - 1) + 1 elements do, at most 32 c
        const uint32_t hi = min(c * (uint32_t)kStageSample, n_tj);
        const uint32_t lo = min(c ? (c - 1u) * (uint32_t)kStageSample + 1u : 0u, hi);
        const uint32_t* tk = tile_keys + (size_t)ts * TILE;
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t h = kStageSample / 2; h >= 1u; h >>= 1) {
            const uint32_t i = lo + pos + h - 1u;
            const uint32_t v = tk[i < hi ? i : 0u];
            pos += (i < hi && (le ? (v <= kq) : (v < kq))) ? h : 0u;
        }
        {
            const uint32_t i = lo + pos;
            const uint32_t v = tk[i < hi ? i : 0u];
            pos += (i < hi && (le ? (v <= kq) : (v < kq))) ? 1u : 0u;
        }
        if (upper) s_hi[wv][tj] = lo + pos; else s_lo[wv][tj] = lo + pos;
    }
    wave_fence();
    uint32_t rank = p;
    for (uint32_t tb = 0; tb < T; tb += (uint32_t)C) {
        uint32_t lo[C], w[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {  // (wave-uniform: every lane reads the same words)
            const uint32_t tt = tb + (uint32_t)j;
            const bool act = tt < T && tt != t;
            lo[j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(act ? s_lo[wv][tt & 31u] : 0u));
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(act ? s_hi[wv][tt & 31u] : 0u));
            w[j] = hi - lo[j];
        }
        uint32_t v[C][2];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t tt = tb + (uint32_t)j;
            const uint32_t* tk = tile_keys + (size_t)(tt < T ? tt : t) * TILE + lo[j];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = (uint32_t)lane + 64u * (uint32_t)h;
                v[j][h] = tk[i < w[j] ? i : 0u];
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            s_win[wv][j][lane] = v[j][0];
            s_win[wv][j][lane + 64] = v[j][1];
        }
        wave_fence();
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool le = tb + (uint32_t)j < t;
            rank += lo[j] + count_before<WCAP>(s_win[wv][j], min(w[j], WCAP), k, le);
        }
        wave_fence();
#pragma unroll
        for (int j = 0; j < C; ++j) {  // a window wider than the LDS slot (skewed tiles): the rest in pieces of 128, one tile at a time
            if (w[j] > WCAP) {         // (uniform)
                const uint32_t tt = tb + (uint32_t)j;
                const bool le = tt < t;
                const uint32_t* tk = tile_keys + (size_t)tt * TILE + lo[j];
                for (uint32_t b = WCAP; b < w[j]; b += WCAP) {
                    const uint32_t i0 = b + (uint32_t)lane, i1 = i0 + 64u;
                    const uint32_t a0 = tk[i0 < w[j] ? i0 : 0u], a1 = tk[i1 < w[j] ? i1 : 0u];
                    s_win[wv][j][lane] = a0;
                    s_win[wv][j][lane + 64] = a1;
                    wave_fence();
                    rank += count_before<WCAP>(s_win[wv][j], min(w[j] - b, WCAP), k, le);
                    wave_fence();
                }
            }
        }
    }
    if (valid) {
        const uint32_t src = (uint32_t)FLH_IDX(311, tile_idx[t * TILE + p], N);
        const uint32_t* r = words + (size_t)src * stride_words;
        float4 o;
        o.x = __uint_as_float(r[0]);
        o.y = __uint_as_float(r[1]);
        o.z = __uint_as_float(r[2]);
        o.w = __uint_as_float(src);
        body[FLH_IDX(312, rank, N)] = o;
    }
}

// records (xyz first, stride_bytes a multiple of 4) -> body: the scan in the stable Morton order of its body-frame coordinates,
// .w = original index.  tile_keys / tile_idx: N words of scratch each; samples: stage_sample_words(N).  N <= stage_sort_max().
hipError_t launch_stage_sort(const void* records, uint32_t stride_bytes, uint32_t N, float quantum, uint32_t* tile_keys,
                             uint32_t* tile_idx, uint32_t* samples, float4* body, hipStream_t st) {
    if (N == 0) return hipSuccess;
    if (N > stage_sort_max()) return hipErrorInvalidValue;
    const uint32_t tile = stage_tile(N);
    const uint32_t T = (N + tile - 1) / tile;
    const uint32_t* w = (const uint32_t*)records;
    const uint32_t sw = stride_bytes / 4u;
    if (tile == 4096u) {
        hipLaunchKernelGGL((k_stage_tile_sort<4>), dim3(T), dim3(1024), 0, st, w, sw, N, 1.0f / quantum, tile_keys, tile_idx, samples);
        hipLaunchKernelGGL((k_stage_merge_rank<4096>), dim3(T * (4096 / 256)), dim3(256), 0, st, w, sw, N, tile_keys, tile_idx, samples, body);
    } else {
        hipLaunchKernelGGL((k_stage_tile_sort<8>), dim3(T), dim3(1024), 0, st, w, sw, N, 1.0f / quantum, tile_keys, tile_idx, samples);
        hipLaunchKernelGGL((k_stage_merge_rank<8192>), dim3(T * (8192 / 256)), dim3(256), 0, st, w, sw, N, tile_keys, tile_idx, samples, body);
    }
    return hipGetLastError();
}

}  // namespace flh
