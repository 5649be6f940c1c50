// flh_stage.hip -- the scan's staging as TWO launches of this library's own: records -> Morton key -> stable order -> Morton-ordered
// float4 scan.  It replaces k_scan_restride + hipcub::DeviceRadixSort::SortPairs (rocPRIM: block sort + 7 merge passes + 2
// transforms) + k_scan_gather -- 12 launches of copy-stream kernels per scan, the only library code of the timed region
// (VERDICT r5 item 1).  The hand-over it serves is the node's  downSizeFilterSurf.filter(*feats_down_body)  ->
// feats_down_size  (src/laserMapping.cpp:904-905, 935-951): the order of feats_down_body is free as long as every per-point
// product goes back by its original index, which body[i].w carries.
//
//   k_stage_tile_sort   one workgroup (1 024 threads) per TILE = 4 096 (8 192 above 131 072 points) consecutive records: reads the
//                       records as they came over PCIe (xyz first, any stride that is a multiple of 4 bytes), forms the key, and
//                       sorts (key, position) in LDS by a stable LSD radix sort, 8 bits per pass: the lanes of a row that share a
//                       digit find each other with eight ballots (one compare when the whole row has one digit, the rule in the
//                       upper passes) -- no atomics, so equal digits keep the order of their positions; per-wave digit counts, a
//                       digit-major exclusive scan over (digit, wave), scatter, read back.  Output: the tile's keys and original
//                       indices in sorted order, and every 32nd key (the tile's SAMPLES).
//   k_stage_merge_rank  every element finds its place in the whole: its position in its own tile + for every other tile the
//                       number of elements that precede it -- keys <= its own in EARLIER tiles, keys < its own in LATER tiles,
//                       which is exactly the stable order of (key, original index) because tiles are consecutive index ranges.
//                       A wave holds 64 consecutive elements of a sorted tile, so it first brackets every other tile between the
//                       counts of its smallest and its largest key -- one lane per (tile, end): the samples of all tiles sit in
//                       LDS and narrow a tile to 32 elements, six loads finish -- then takes the tiles eight at a time: windows of
//                       up to 128 elements are loaded into LDS (coalesced) and every lane counts inside them, eight branch-free
//                       binary searches side by side; a wider window (a sensor that delivers its points sector by sector: a
//                       wave's key range then holds thousands of elements of a few tiles and none of the others) is searched
//                       where it lies, in log2(width) loads.  The element's record is gathered and written to its place as
//                       float4, .w = original index.
//
// Versions measured in round 6 (profiles/r06_call{1..5}/): ballots + global searches; LDS lane masks + linear windows (a
// sector-ordered scan took 120 us); round-robin tiles + 64-bit composites (robust, 48 us alone, but 66 KB of LDS per workgroup
// beside the update cost the pipeline 4-6 %); this one.
//
// Same key, same stable order as the radix sort it replaces => the same units, the same bits downstream (tests/test_gpu_staging.py;
// every staged-vs-uploaded test).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flh_device.hpp"
#include "flh_kernels.hpp"

namespace flh {

constexpr int kStageSample = 32;       // every 32nd sorted key of a tile is a sample
constexpr int kStageTilesMax = 32;     // tiles of a scan at most; the samples of all of them sit in LDS (32 x 128 or 32 x 256 keys)
constexpr uint32_t kStageSmallMax = 32u * 4096u;  // up to here tiles of 4 096 records (E = 4), above tiles of 8 192 (E = 8)

uint32_t stage_sort_max() { return 32u * 8192u; }  // 262 144 points; above it the library sort
static inline uint32_t stage_tile(uint32_t N) { return N <= kStageSmallMax ? 4096u : 8192u; }
// scratch, in 32-bit words: sorted keys (N), their original indices (N), the tiles' samples (T x tile / 32)
uint32_t stage_scratch_words(uint32_t N) {
    const uint32_t tile = stage_tile(N), T = (N + tile - 1u) / tile;
    return 2u * N + T * (tile / kStageSample);
}

__device__ __forceinline__ void wave_fence() {  // LDS hand-offs inside one wave: LDS keeps a wave's order, the compiler must too
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint64_t match_digit8(uint32_t digit) {  // the lanes of this wave whose digit equals mine
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)digit);
    if (__builtin_amdgcn_ballot_w64(digit == d0) == ~0ull) return ~0ull;  // (uniform) one digit in the whole row
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const uint64_t m = __builtin_amdgcn_ballot_w64(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

template <int E>
__global__ void __launch_bounds__(1024)
k_stage_tile_sort(const uint32_t* __restrict__ words, uint32_t stride_words, uint32_t N, float inv_q,
                  uint32_t* __restrict__ tile_keys, uint32_t* __restrict__ tile_idx, uint32_t* __restrict__ samples) {
    constexpr int TILE = 1024 * E;
    __shared__ uint2 s_kv[TILE];
    __shared__ uint32_t s_cnt[16 * 256];  // [wave][digit]: counts, then the wave's first place inside the digit
    __shared__ uint32_t s_dbase[256];     // first place of a digit
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tiles of EQUAL size: S = ceil(N / T) consecutive records each (a short last tile is sparse in key space: the 64 elements of
    // one of its waves span thousands of elements of every other tile -- measured: 50 us instead of 20, profiles/r06_call7/)
    const uint32_t S = (N + gridDim.x - 1u) / gridDim.x;
    const uint32_t base = blockIdx.x * S;
    const uint32_t n_t = min(S, N - base);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

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
    volatile uint32_t* cnt = s_cnt + wave * 256;  // this wave's counters: touched by this wave only until the scan
#pragma unroll
    for (int d = 0; d < 4; ++d) cnt[d * 64 + lane] = 0u;
    for (int shift = 0; shift < 32; shift += 8) {
        uint32_t lr[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {  // rows in order; LDS executes a wave's operations in order
            const uint32_t digit = (key[e] >> shift) & 255u;
            const uint64_t peers = match_digit8(digit);
            const uint32_t prior = cnt[digit];
            lr[e] = prior + (uint32_t)__popcll(peers & lt_mask);
            __builtin_amdgcn_wave_barrier();
            if ((peers & lt_mask) == 0ull) cnt[digit] = prior + (uint32_t)__popcll(peers);  // the digit's lowest lane
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        uint32_t incl = 0, total = 0;
        if (tid < 256) {  // digit = tid: the waves' counts of the digit become exclusive offsets; then a scan over the digits
            uint32_t c[16];
#pragma unroll
            for (int w = 0; w < 16; ++w) c[w] = s_cnt[w * 256 + tid];
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                s_cnt[w * 256 + tid] = run;
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
            dst[e] = s_dbase[digit] + cnt[digit] + lr[e];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 0; d < 4; ++d) cnt[d * 64 + lane] = 0u;  // (this wave's counters, read above by this wave only) for the next pass
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

// does v precede k?  le: v lives in an EARLIER tile (lower original indices): its equal keys come first
// (bitwise on purpose: the searches below are straight-line code, no branch per step)
__device__ __forceinline__ bool stage_before(uint32_t v, uint32_t k, bool le) { return (v < k) | ((v == k) & le); }

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
    const uint32_t S = (N + T - 1u) / T;  // records per tile (k_stage_tile_sort): tile tt holds [tt S, min(N, (tt + 1) S)), stored there
    const uint32_t n_own = min(S, N - t * S);
    if ((p & ~255u) >= n_own) return;  // (the whole workgroup: nothing of its tile left)
    for (uint32_t i = threadIdx.x; i < T * SPT; i += 256u) s_smp[i] = samples[i];
    __syncthreads();
    const bool valid = p < n_own;
    const uint64_t vm = __builtin_amdgcn_ballot_w64(valid);
    if (vm == 0ull) return;  // (no workgroup barrier below: a wave may leave)
    const uint32_t k = valid ? tile_keys[t * S + p] : 0xFFFFFFFFu;
    // ---- brackets: lane j < 32 counts tile j against the wave's smallest key, lane j >= 32 tile j - 32 against its largest
    {
        const uint32_t kmin = __shfl(k, 0, 64), kmax = __shfl(k, __popcll(vm) - 1, 64);
        const uint32_t tj = (uint32_t)lane & 31u;
        const bool upper = lane >= 32;
        const bool act = tj < T && tj != t;
        const uint32_t ts = act ? tj : t;
        const uint32_t kq = upper ? kmax : kmin;
        const bool le = tj < t;
        const uint32_t n_tj = act ? min(S, N - tj * S) : 0u;
        const uint32_t* sp = s_smp + ts * SPT;
        uint32_t c = 0;
#pragma unroll
        for (uint32_t h = SPT / 2; h >= 1u; h >>= 1) c += stage_before(sp[c + h - 1u], kq, le) ? h : 0u;
        c += stage_before(sp[c], kq, le) ? 1u : 0u;  // (c <= SPT - 1 here)
        // samples 0 .. c-1 precede kq: at least 32 (c - 1) + 1 elements do, at most 32 c
        const uint32_t hi = min(c * (uint32_t)kStageSample, n_tj);
        const uint32_t lo = min(c ? (c - 1u) * (uint32_t)kStageSample + 1u : 0u, hi);
        const uint32_t* tk = tile_keys + (size_t)ts * S;
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t h = kStageSample / 2; h >= 1u; h >>= 1) {
            const uint32_t i = lo + pos + h - 1u;
            const uint32_t v = tk[i < hi ? i : 0u];
            pos += ((i < hi) & stage_before(v, kq, le)) ? h : 0u;
        }
        {
            const uint32_t i = lo + pos;
            const uint32_t v = tk[i < hi ? i : 0u];
            pos += ((i < hi) & stage_before(v, kq, le)) ? 1u : 0u;
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
            const uint32_t* tk = tile_keys + (size_t)(tt < T ? tt : t) * S + lo[j];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t i = (uint32_t)lane + 64u * (uint32_t)h;
                v[j][h] = tk[(i < w[j] && w[j] <= WCAP) ? i : 0u];
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            s_win[wv][j][lane] = v[j][0];
            s_win[wv][j][lane + 64] = v[j][1];
        }
        wave_fence();
        {   // the eight counts side by side: one step of every search before the next step of any
            uint32_t c[C], n[C];
#pragma unroll
            for (int j = 0; j < C; ++j) { c[j] = 0u; n[j] = w[j] <= WCAP ? w[j] : 0u; }
#pragma unroll
            for (uint32_t h = WCAP / 2; h >= 1u; h >>= 1) {
                uint32_t a[C];
#pragma unroll
                for (int j = 0; j < C; ++j) a[j] = s_win[wv][j][c[j] + h - 1u];
#pragma unroll
                for (int j = 0; j < C; ++j) c[j] += ((c[j] + h - 1u < n[j]) & stage_before(a[j], k, tb + (uint32_t)j < t)) ? h : 0u;
            }
            uint32_t a[C];
#pragma unroll
            for (int j = 0; j < C; ++j) a[j] = s_win[wv][j][c[j] < WCAP ? c[j] : WCAP - 1u];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                c[j] += ((c[j] < n[j]) & stage_before(a[j], k, tb + (uint32_t)j < t)) ? 1u : 0u;
                rank += lo[j] + c[j];
            }
        }
        wave_fence();
#pragma unroll
        for (int j = 0; j < C; ++j) {
            if (w[j] > WCAP) {  // (uniform) a window wider than the LDS slot: every lane searches it where it lies, log2(width) loads
                const uint32_t tt = tb + (uint32_t)j;
                const bool le = tt < t;
                const uint32_t* tk = tile_keys + (size_t)tt * S + lo[j];
                uint32_t a = 0, b = w[j];  // the count lies in [a, b]
                while (__builtin_amdgcn_ballot_w64(a < b) != 0ull) {
                    const uint32_t mid = (a + b) >> 1;
                    const uint32_t vv = tk[mid < w[j] ? mid : 0u];
                    if (a < b) {
                        if (stage_before(vv, k, le)) a = mid + 1u; else b = mid;
                    }
                }
                rank += a;
            }
        }
    }
    if (valid) {
        const uint32_t src = (uint32_t)FLH_IDX(311, tile_idx[t * S + p], N);
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
// .w = original index.  scratch: stage_scratch_words(N) 32-bit words.  N <= stage_sort_max().
hipError_t launch_stage_sort(const void* records, uint32_t stride_bytes, uint32_t N, float quantum, uint32_t* scratch,
                             float4* body, hipStream_t st) {
    if (N == 0) return hipSuccess;
    if (N > stage_sort_max()) return hipErrorInvalidValue;
    const uint32_t tile = stage_tile(N);
    const uint32_t T = (N + tile - 1) / tile;
    const uint32_t* w = (const uint32_t*)records;
    const uint32_t sw = stride_bytes / 4u;
    uint32_t* tile_keys = scratch;
    uint32_t* tile_idx = scratch + N;
    uint32_t* samples = scratch + 2 * (size_t)N;
    if (tile == 4096u) {
        hipLaunchKernelGGL((k_stage_tile_sort<4>), dim3(T), dim3(1024), 0, st, w, sw, N, 1.0f / quantum, tile_keys, tile_idx, samples);
        hipLaunchKernelGGL((k_stage_merge_rank<4096>), dim3(T * (4096 / 256)), dim3(256), 0, st, w, sw, N, tile_keys, tile_idx, samples, body);
    } else {
        hipLaunchKernelGGL((k_stage_tile_sort<8>), dim3(T), dim3(1024), 0, st, w, sw, N, 1.0f / quantum, tile_keys, tile_idx, samples);
        hipLaunchKernelGGL((k_stage_merge_rank<8192>), dim3(T * (8192 / 256)), dim3(256), 0, st, w, sw, N, tile_keys, tile_idx, samples, body);
    }
    return hipGetLastError();
}

#ifdef FLH_BOUNDS
void bounds_read_stage(unsigned long long out[5]) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds), sizeof(BoundsRec)); }
#endif

}  // namespace flh
