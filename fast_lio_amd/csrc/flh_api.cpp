// flh_api.cpp -- layer 1 of the C ABI (include/fastlio_hip.h): device memory, map index build, scan
// upload, one h_share_model evaluation per call, lazy fetches.  Host side only; kernels live in
// flh_kernels.hip.  There is NO CPU fallback: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fastlio_hip.h"
#include "flh_kernels.hpp"

using flh::GridParams;
using flh::StateDev;
typedef unsigned long long u64;

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return -1;
}
#define HIPC(expr)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct flh_handle {
    flh_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[4]{};  // start, after search, after fit, end
    // map
    size_t M = 0;
    GridParams grid{};
    DevBuf<float4> map_sorted;
    DevBuf<uint2> hash;
    DevBuf<uint32_t> starts;
    uint32_t nbricks = 0;
    int rmax = 3;
    // scan
    size_t N = 0;
    DevBuf<float4> world, nn_pts, normvec;
    DevBuf<float> nn_d2;
    DevBuf<uint8_t> nn_cnt, selected;
    DevBuf<double> partials, part2, gram;
    DevBuf<u64> counter;
    DevBuf<uint32_t> slow_list, slow_list2, slow_count;  // A1 -> A2 -> A3 work lists (striped) and their counters
    DevBuf<float> slow_ub;
    DevBuf<uint32_t> tickets;                // arrival tickets of the in-kernel reduction (self re-arming)                   // per-query search radius^2 handed from A1 to A2
    double* h_gram = nullptr;  // pinned 256 doubles
    u64* h_counter = nullptr;  // pinned
    // last evaluation
    StateDev last_state{};
    int last_ext = 0;
    bool have_eval = false;
    bool stats = false;
    flh_timing timing{};
    bool searched_once = false;
    int timing_stride = 1;   // record HIP events on every n-th evaluation (0 = never)
    uint64_t eval_no = 0;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    // staging ring
    struct Slot {
        DevBuf<float4> body;            // Morton-ordered (internal order); .w = original index
        std::vector<float> h_body;      // host copy, original order (flh_fetch_rows)
        std::vector<uint32_t> h_perm;   // internal index -> original index
        size_t N = 0;
        hipEvent_t ready = nullptr;
        bool used = false;
    };
    Slot slots[FLH_MAX_SLOTS + 1];      // [FLH_MAX_SLOTS] backs flh_scan_upload
    hipStream_t copy_stream = nullptr;
    const float4* cur_body = nullptr;   // the active slot's buffer
    const Slot* cur = nullptr;
    // staging scratch (copy stream)
    DevBuf<float4> st_raw;
    DevBuf<u64> st_k0, st_k1;
    DevBuf<uint32_t> st_v0, st_v1;
    DevBuf<unsigned char> st_tmp;
};

extern "C" {

const char* flh_last_error(void) { return g_err.c_str(); }

int flh_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n > 0 ? 1 : 0;
}

void flh_default_config(flh_config* c) {
    if (!c) return;
    c->device = -1;
    c->cell_size = 1.5f;
    c->plane_threshold = 0.1f;
    c->max_sqdist = 5.0f;
    c->stream = nullptr;
    c->lanes_per_query = 4;
    c->sort_queries = -1;
}

int flh_create(const flh_config* cfg_in, flh_handle** out) {
    if (!out) return fail("flh_create: out == NULL");
    *out = nullptr;
    if (!flh_device_available())
        return fail("flh_create: no HIP device visible -- libfastlio_hip has no CPU fallback");
    flh_config cfg;
    flh_default_config(&cfg);
    if (cfg_in) cfg = *cfg_in;
    if (cfg.cell_size <= 0) cfg.cell_size = 1.5f;
    if (cfg.plane_threshold <= 0) cfg.plane_threshold = 0.1f;
    if (cfg.max_sqdist <= 0) cfg.max_sqdist = 5.0f;
    if (cfg.sort_queries < 0) cfg.sort_queries = 1;
    {
        const int l = cfg.lanes_per_query;  // 0 = exact kernel for every query
        if (l != 0 && l != 2 && l != 8 && l != 16) cfg.lanes_per_query = 4;
    }
    flh_handle* h = new flh_handle();
    h->cfg = cfg;
    if (cfg.device >= 0) {
        hipError_t e = hipSetDevice(cfg.device);
        if (e != hipSuccess) {
            delete h;
            return fail(std::string("hipSetDevice: ") + hipGetErrorString(e));
        }
    }
    (void)hipGetDevice(&h->device);
    if (cfg.stream) {
        h->stream = (hipStream_t)cfg.stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete h;
            return fail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        h->own_stream = true;
    }
    for (auto& e : h->ev) (void)hipEventCreate(&e);
    if (hipHostMalloc((void**)&h->h_gram, 256 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_counter, sizeof(u64), hipHostMallocDefault) != hipSuccess) {
        flh_destroy(h);
        return fail("hipHostMalloc failed");
    }
    if (h->gram.reserve(256) != hipSuccess || h->counter.reserve(1) != hipSuccess || h->slow_count.reserve(2 * flh::list_stripes()) != hipSuccess ||
        hipMemset(h->slow_count.p, 0, 2 * flh::list_stripes() * sizeof(uint32_t)) != hipSuccess) {
        flh_destroy(h);
        return fail("hipMalloc failed");
    }
    if (const char* e = std::getenv("FLH_TIMING_STRIDE")) h->timing_stride = std::max(0, std::atoi(e));
    h->rmax = (int)std::ceil((std::sqrt((double)cfg.max_sqdist) + 2e-3 * cfg.cell_size) / cfg.cell_size);
    if (h->rmax < 1) h->rmax = 1;
    *out = h;
    return 0;
}

void flh_destroy(flh_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->map_sorted.release(); h->hash.release(); h->starts.release(); h->slow_list.release(); h->slow_list2.release(); h->slow_ub.release(); h->slow_count.release(); h->tickets.release();
    h->world.release(); h->nn_pts.release(); h->normvec.release();
    h->nn_d2.release(); h->nn_cnt.release(); h->selected.release();
    h->partials.release(); h->part2.release(); h->gram.release(); h->counter.release();
    for (auto& sl : h->slots) {
        sl.body.release();
        if (sl.ready) (void)hipEventDestroy(sl.ready);
    }
    h->st_raw.release(); h->st_k0.release(); h->st_k1.release(); h->st_v0.release(); h->st_v1.release(); h->st_tmp.release();
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->h_gram) (void)hipHostFree(h->h_gram);
    if (h->h_counter) (void)hipHostFree(h->h_counter);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

size_t flh_map_size(const flh_handle* h) { return h ? h->M : 0; }
size_t flh_scan_size(const flh_handle* h) { return h ? h->N : 0; }

// ---------------------------------------------------------------------------------------------
// ikdtree.Build -- src/laserMapping.cpp:919
// ---------------------------------------------------------------------------------------------
int flh_map_build(flh_handle* h, const void* xyz, size_t stride_bytes, size_t M) {
    if (!h) return fail("flh_map_build: null handle");
    if (M > 0 && !xyz) return fail("flh_map_build: null points");
    if (stride_bytes < 12) return fail("flh_map_build: stride_bytes < 12");
    if (M >= (1ull << 31)) return fail("flh_map_build: M too large");
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    h->M = 0;
    h->searched_once = false;
    const float c = h->cfg.cell_size;
    // host pass: re-stride to float4 and take the exact AABB
    std::vector<float4> hp(M ? M : 1);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const unsigned char* src = (const unsigned char*)xyz;
    for (size_t i = 0; i < M; ++i) {
        float p[3];
        std::memcpy(p, src + i * stride_bytes, 12);
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]))
            return fail("flh_map_build: non-finite map point at index " + std::to_string(i));
        hp[i] = make_float4(p[0], p[1], p[2], 0.f);
        for (int d = 0; d < 3; ++d) {
            mn[d] = std::min(mn[d], p[d]);
            mx[d] = std::max(mx[d], p[d]);
        }
    }
    if (M == 0) mn[0] = mn[1] = mn[2] = mx[0] = mx[1] = mx[2] = 0.f;
    GridParams g{};
    g.c = c;
    g.inv_c = 1.0f / c;
    const int PAD = 4;  // cells of slack on every side, so near-outside queries keep non-negative cells
    float o[3];
    int dims[3];
    for (int d = 0; d < 3; ++d) {
        o[d] = (std::floor(mn[d] / c) - PAD) * c;
        dims[d] = (int)std::floor((mx[d] - o[d]) / c) + 1 + PAD;
        if (dims[d] > 4096)
            return fail("flh_map_build: map extent exceeds 4096 cells along an axis; raise flh_config.cell_size");
    }
    g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
    g.nx = dims[0]; g.ny = dims[1]; g.nz = dims[2];

    // device scratch
    DevBuf<float4> d_in;
    DevBuf<u64> k0, k1;
    DevBuf<uint32_t> v0, v1, bh, br;
    DevBuf<unsigned char> tmp;
    const uint32_t Mu = (uint32_t)M;
    auto cleanup = [&]() { d_in.release(); k0.release(); k1.release(); v0.release(); v1.release(); bh.release(); br.release(); tmp.release(); };
#define HIPC_CL(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            cleanup();                                                                       \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
        }                                                                                    \
    } while (0)
    HIPC_CL(h->map_sorted.reserve(M ? M : 1));
    uint32_t nbricks = 0;
    if (M > 0) {
        HIPC_CL(d_in.reserve(M)); HIPC_CL(k0.reserve(M)); HIPC_CL(k1.reserve(M));
        HIPC_CL(v0.reserve(M)); HIPC_CL(v1.reserve(M)); HIPC_CL(bh.reserve(M)); HIPC_CL(br.reserve(M));
        HIPC_CL(hipMemcpyAsync(d_in.p, hp.data(), M * sizeof(float4), hipMemcpyHostToDevice, st));
        GridParams gk = g;  // keys only need origin/extent
        HIPC_CL(flh::launch_map_keys(gk, d_in.p, Mu, k0.p, v0.p, st));
        size_t tb1 = 0, tb2 = 0;
        HIPC_CL(flh::sort_pairs(nullptr, tb1, k0.p, k1.p, v0.p, v1.p, Mu, st));
        HIPC_CL(flh::inclusive_sum(nullptr, tb2, bh.p, br.p, Mu, st));
        HIPC_CL(tmp.reserve(std::max(tb1, tb2)));
        size_t tb = tmp.cap;
        HIPC_CL(flh::sort_pairs(tmp.p, tb, k0.p, k1.p, v0.p, v1.p, Mu, st));
        HIPC_CL(flh::launch_map_gather(d_in.p, k1.p, v1.p, Mu, h->map_sorted.p, bh.p, st));
        tb = tmp.cap;
        HIPC_CL(flh::inclusive_sum(tmp.p, tb, bh.p, br.p, Mu, st));
        HIPC_CL(hipMemcpyAsync(&nbricks, br.p + (M - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPC_CL(hipStreamSynchronize(st));
    }
    // directory + per-brick prefix tables
    uint32_t hs = 1024;
    while (hs < 2 * (nbricks + 1)) hs <<= 1;
    int log2hs = 0;
    while ((1u << log2hs) < hs) ++log2hs;
    DevBuf<uint32_t> bstart;
    HIPC_CL(h->hash.reserve(hs));
    HIPC_CL(h->starts.reserve((size_t)(nbricks ? nbricks : 1) * flh::kBrickStride));
    HIPC_CL(hipMemsetAsync(h->hash.p, 0xFF, (size_t)hs * sizeof(uint2), st));
    if (M > 0) {
        hipError_t e1 = bstart.reserve((size_t)nbricks + 1);
        if (e1 != hipSuccess) { cleanup(); return fail(std::string("hipMalloc: ") + hipGetErrorString(e1)); }
        hipError_t e2 = flh::launch_brick_starts(bh.p, br.p, Mu, bstart.p, st);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(bstart.p + nbricks, &Mu, sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e2 == hipSuccess)
            e2 = flh::launch_brick_tables(k1.p, bstart.p, nbricks, h->starts.p, h->hash.p, hs - 1, 32 - log2hs, st);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(st);
        bstart.release();
        if (e2 != hipSuccess) { cleanup(); return fail(std::string("map tables: ") + hipGetErrorString(e2)); }
    }
    HIPC_CL(hipStreamSynchronize(st));
    cleanup();
#undef HIPC_CL
    g.hash_mask = hs - 1;
    g.hash_shift = 32 - log2hs;
    g.hash = h->hash.p;
    g.starts = h->starts.p;
    g.pts = h->map_sorted.p;
    h->grid = g;
    h->nbricks = nbricks;
    h->M = M;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// per-scan work buffers sized for N points; resets the per-scan state the reference keeps in globals
static int prepare_scan_buffers(flh_handle* h, size_t N, bool full_clear) {
    hipStream_t st = h->stream;
    const size_t n1 = N ? N : 1;
    HIPC(h->world.reserve(n1)); HIPC(h->nn_pts.reserve(5 * n1)); HIPC(h->normvec.reserve(n1));
    HIPC(h->nn_d2.reserve(5 * n1)); HIPC(h->nn_cnt.reserve(n1)); HIPC(h->selected.reserve(n1));
    {
        const size_t ln = (size_t)flh::list_stripes() * flh::list_stripe_cap((int)N);
        HIPC(h->slow_list.reserve(ln)); HIPC(h->slow_list2.reserve(ln)); HIPC(h->slow_ub.reserve(n1));
    }
    const int nblk = flh::fit_blocks((int)N);
    HIPC(h->partials.reserve((size_t)nblk * 256));
    const int ngroups = flh::reduce1_blocks(nblk, nullptr);
    HIPC(h->part2.reserve((size_t)ngroups * 256));
    {
        const uint32_t* before = h->tickets.p;
        HIPC(h->tickets.reserve((size_t)ngroups + 1));
        if (h->tickets.p != before)  // fresh allocation: tickets must start at zero
            HIPC(hipMemsetAsync(h->tickets.p, 0, h->tickets.cap * sizeof(uint32_t), st));
    }
    // Fast path (flh_scan_activate): no memsets at all.  point_selected_surf needs no reset because the first
    // evaluation of a scan always searches (enforced in enqueue_eval) and the search rewrites every flag; the
    // reduction tickets and work-list counters are re-armed by k_fit at the end of every evaluation.
    if (full_clear) {
        HIPC(hipMemsetAsync(h->tickets.p, 0, ((size_t)ngroups + 1) * sizeof(uint32_t), st));
        HIPC(hipMemsetAsync(h->slow_count.p, 0, 2 * flh::list_stripes() * sizeof(uint32_t), st));
        HIPC(hipMemsetAsync(h->selected.p, 1, n1, st));  // memset(point_selected_surf, true, ...) :812
        HIPC(hipMemsetAsync(h->nn_cnt.p, 0, n1, st));
        HIPC(hipMemsetAsync(h->nn_pts.p, 0xFF, 5 * n1 * sizeof(float4), st));  // idx = -1
        HIPC(hipMemsetAsync(h->nn_d2.p, 0x7F, 5 * n1 * sizeof(float), st));    // large finite; rewritten by search
        HIPC(hipMemsetAsync(h->normvec.p, 0, n1 * sizeof(float4), st));
        HIPC(hipMemsetAsync(h->world.p, 0, n1 * sizeof(float4), st));
    }
    h->N = N;
    h->have_eval = false;
    h->searched_once = false;
    return 0;
}

// Copies a scan to the device in Morton order of its body-frame coordinates (sort_queries != 0) on the
// copy stream; returns when the host buffer may be reused.
static int stage_into(flh_handle* h, flh_handle::Slot& sl, const void* pts, size_t stride_bytes, size_t N) {
    if (N > 0 && !pts) return fail("scan staging: null points");
    if (stride_bytes < 12 && N > 0) return fail("scan staging: stride_bytes < 12");
    if (N >= (1ull << 26)) return fail("scan staging: N too large");
    HIPC(hipSetDevice(h->device));
    if (!h->copy_stream) HIPC(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    hipStream_t cs = h->copy_stream;
    if (!sl.ready) HIPC(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    const size_t n1 = N ? N : 1;
    HIPC(sl.body.reserve(n1));
    HIPC(h->st_raw.reserve(n1));
    sl.h_body.resize(3 * n1);
    sl.h_perm.resize(n1);
    std::vector<float4> hb(n1);
    const unsigned char* src = (const unsigned char*)pts;
    for (size_t i = 0; i < N; ++i) {
        float p[3];
        std::memcpy(p, src + i * stride_bytes, 12);
        hb[i] = make_float4(p[0], p[1], p[2], 0.f);
        sl.h_body[3 * i] = p[0]; sl.h_body[3 * i + 1] = p[1]; sl.h_body[3 * i + 2] = p[2];
    }
    const bool do_sort = h->cfg.sort_queries != 0 && N > 1;
    if (N > 0) HIPC(hipMemcpyAsync(h->st_raw.p, hb.data(), N * sizeof(float4), hipMemcpyHostToDevice, cs));
    if (do_sort) {
        const uint32_t Nu = (uint32_t)N;
        HIPC(h->st_k0.reserve(N)); HIPC(h->st_k1.reserve(N)); HIPC(h->st_v0.reserve(N)); HIPC(h->st_v1.reserve(N));
        HIPC(flh::launch_scan_keys(h->st_raw.p, Nu, 0.25f, h->st_k0.p, h->st_v0.p, cs));
        size_t tb = 0;
        HIPC(flh::sort_scan_pairs(nullptr, tb, h->st_k0.p, h->st_k1.p, h->st_v0.p, h->st_v1.p, Nu, cs));
        HIPC(h->st_tmp.reserve(tb));
        tb = h->st_tmp.cap;
        HIPC(flh::sort_scan_pairs(h->st_tmp.p, tb, h->st_k0.p, h->st_k1.p, h->st_v0.p, h->st_v1.p, Nu, cs));
        HIPC(flh::launch_scan_gather(h->st_raw.p, h->st_v1.p, Nu, sl.body.p, cs));
        HIPC(hipMemcpyAsync(sl.h_perm.data(), h->st_v1.p, N * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    } else {
        HIPC(flh::launch_scan_gather(h->st_raw.p, nullptr, (uint32_t)N, sl.body.p, cs));
        for (size_t i = 0; i < N; ++i) sl.h_perm[i] = (uint32_t)i;
    }
    HIPC(hipEventRecord(sl.ready, cs));
    HIPC(hipStreamSynchronize(cs));  // hb is pageable and dies here; h_perm must be complete
    sl.N = N;
    sl.used = true;
    return 0;
}

static int activate(flh_handle* h, flh_handle::Slot& sl, bool full_clear) {
    HIPC(hipSetDevice(h->device));
    HIPC(hipStreamWaitEvent(h->stream, sl.ready, 0));
    if (prepare_scan_buffers(h, sl.N, full_clear) != 0) return -1;
    h->cur_body = sl.body.p;
    h->cur = &sl;
    return 0;
}

int flh_scan_upload(flh_handle* h, const void* pts, size_t stride_bytes, size_t N) {
    if (!h) return fail("flh_scan_upload: null handle");
    flh_handle::Slot& sl = h->slots[FLH_MAX_SLOTS];
    if (stage_into(h, sl, pts, stride_bytes, N) != 0) return -1;
    if (activate(h, sl, true) != 0) return -1;
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

int flh_scan_stage(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t N) {
    if (!h) return fail("flh_scan_stage: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_stage: bad slot");
    return stage_into(h, h->slots[slot], pts, stride_bytes, N);
}

int flh_scan_activate(flh_handle* h, int slot) {
    if (!h) return fail("flh_scan_activate: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS || !h->slots[slot].used) return fail("flh_scan_activate: slot not staged");
    return activate(h, h->slots[slot], false);
}

static StateDev make_state(const double rot[4], const double pos[3], const double offR[4], const double offT[3]) {
    StateDev s;
    for (int i = 0; i < 4; ++i) { s.rot[i] = rot[i]; s.offR[i] = offR[i]; }
    for (int i = 0; i < 3; ++i) { s.pos[i] = pos[i]; s.offT[i] = offT[i]; }
    return s;
}

static int enqueue_eval(flh_handle* h, const StateDev& s, int do_search, int ext, double* d_out, bool timed) {
    hipStream_t st = h->stream;
    if (!do_search && !h->searched_once && h->N > 0)
        return fail("flh_eval: do_search == 0 before any search on this scan (the reference always searches on the first pass)");
    if (timed) HIPC(hipEventRecord(h->ev[0], st));
    if (do_search) {
        if (h->stats) HIPC(hipMemsetAsync(h->counter.p, 0, sizeof(u64), st));
        HIPC(flh::launch_search(h->cfg.lanes_per_query, h->grid, s, h->cur_body, (int)h->N, (uint32_t)h->M,
                                h->cfg.max_sqdist, h->rmax, h->nn_pts.p, h->nn_d2.p, h->nn_cnt.p, h->selected.p,
                                h->slow_list.p, h->slow_list2.p, h->slow_ub.p, h->slow_count.p, h->stats ? h->counter.p : nullptr, st));
        h->searched_once = true;
    }
    if (timed) HIPC(hipEventRecord(h->ev[1], st));
    HIPC(flh::launch_fit(s, h->cur_body, h->nn_pts.p, (int)h->N, ext, h->cfg.plane_threshold, h->selected.p, h->normvec.p,
                         h->world.p, h->partials.p, h->part2.p, d_out, h->tickets.p, h->slow_count.p, st));
    if (timed) HIPC(hipEventRecord(h->ev[2], st));
    h->last_state = s;
    h->last_ext = ext;
    h->have_eval = true;
    return 0;
}

void flh_unpack_gram(const double G[256], double HTH[144], double HTh[12], int64_t* n_eff, double* total_residual) {
    for (int i = 0; i < 12; ++i) {
        for (int j = 0; j < 12; ++j) HTH[i * 12 + j] = G[i * 16 + j];
        HTh[i] = G[i * 16 + 12];
    }
    if (n_eff) *n_eff = (int64_t)std::llround(G[13 * 16 + 13]);
    if (total_residual) *total_residual = G[14 * 16 + 13];
}

int flh_eval(flh_handle* h, const double rot[4], const double pos[3], const double offR[4], const double offT[3],
             int do_search, int ext, double HTH[144], double HTh[12], int64_t* n_eff, double* total_residual) {
    if (!h) return fail("flh_eval: null handle");
    if (!rot || !pos || !offR || !offT || !HTH || !HTh) return fail("flh_eval: null argument");
    HIPC(hipSetDevice(h->device));
    const StateDev s = make_state(rot, pos, offR, offT);
    // the final reduce kernel writes the 16x16 block straight into pinned, device-mapped host memory:
    // no copy kernel, no extra boundary -- the stream sync below is the only wait
    const bool timed = h->timing_stride > 0 && (h->eval_no++ % (uint64_t)h->timing_stride) == 0;
    if (enqueue_eval(h, s, do_search, ext, h->h_gram, timed) != 0) return -1;
    hipStream_t st = h->stream;
    if (h->stats && do_search) HIPC(hipMemcpyAsync(h->h_counter, h->counter.p, sizeof(u64), hipMemcpyDeviceToHost, st));
    if (timed) HIPC(hipEventRecord(h->ev[3], st));
    HIPC(hipStreamSynchronize(st));
    flh_unpack_gram(h->h_gram, HTH, HTh, n_eff, total_residual);
    float a = 0, b = 0, c = 0;
    if (timed) {
        (void)hipEventElapsedTime(&a, h->ev[0], h->ev[1]);
        (void)hipEventElapsedTime(&b, h->ev[1], h->ev[2]);
        (void)hipEventElapsedTime(&c, h->ev[0], h->ev[3]);
    }
    h->timing.search_ms = do_search ? a : 0.f;
    h->timing.fit_ms = b;
    h->timing.total_ms = c;
    h->timing.candidates = (h->stats && do_search) ? (int64_t)*h->h_counter : 0;
    if (timed) {
        if (do_search) { h->acc[0] += a; h->acc[1] += 1; }
        h->acc[2] += b; h->acc[3] += 1;
        h->acc[4] += c; h->acc[5] += 1;
    }
    return 0;
}

int flh_get_counters(flh_handle* h, double out[6], int reset) {
    if (!h || !out) return fail("flh_get_counters: null argument");
    for (int i = 0; i < 6; ++i) out[i] = h->acc[i];
    if (reset)
        for (int i = 0; i < 6; ++i) h->acc[i] = 0;
    return 0;
}

int flh_eval_device(flh_handle* h, const double x[FLH_NSTATE], int do_search, int ext, double* d_gram256) {
    if (!h || !x || !d_gram256) return fail("flh_eval_device: null argument");
    HIPC(hipSetDevice(h->device));
    const StateDev s = make_state(x + 3, x + 0, x + 7, x + 11);
    return enqueue_eval(h, s, do_search, ext, d_gram256, false);
}

int flh_last_timing(flh_handle* h, flh_timing* t) {
    if (!h || !t) return fail("flh_last_timing: null argument");
    *t = h->timing;
    return 0;
}
int flh_set_timing_stride(flh_handle* h, int every_n) {
    if (!h) return fail("flh_set_timing_stride: null handle");
    h->timing_stride = every_n < 0 ? 0 : every_n;
    h->eval_no = 0;
    return 0;
}
int flh_enable_stats(flh_handle* h, int on) {
    if (!h) return fail("flh_enable_stats: null handle");
    h->stats = on != 0;
    return 0;
}

int flh_time_kernel(flh_handle* h, int which, const double x[FLH_NSTATE], int ext, int iters, float* mean_ms) {
    if (!h || !x || !mean_ms) return fail("flh_time_kernel: null argument");
    if (iters < 1) iters = 1;
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const StateDev s = make_state(x + 3, x + 0, x + 7, x + 11);
    if (which != 0 && !h->searched_once) return fail("flh_time_kernel: fit kernel timed before any search");
    HIPC(hipEventRecord(h->ev[0], st));
    for (int it = 0; it < iters; ++it) {
        if (which == 0) {
            HIPC(flh::launch_search(h->cfg.lanes_per_query, h->grid, s, h->cur_body, (int)h->N, (uint32_t)h->M,
                                    h->cfg.max_sqdist, h->rmax, h->nn_pts.p, h->nn_d2.p, h->nn_cnt.p, h->selected.p,
                                    h->slow_list.p, h->slow_list2.p, h->slow_ub.p, h->slow_count.p, nullptr, st));
            HIPC(hipMemsetAsync(h->slow_count.p, 0, 2 * flh::list_stripes() * sizeof(uint32_t), st));
            h->searched_once = true;
        } else {
            HIPC(flh::launch_fit(s, h->cur_body, h->nn_pts.p, (int)h->N, ext, h->cfg.plane_threshold, h->selected.p,
                                 h->normvec.p, h->world.p, h->partials.p, h->part2.p, h->gram.p, h->tickets.p, h->slow_count.p, st));
        }
    }
    HIPC(hipEventRecord(h->ev[3], st));
    HIPC(hipStreamSynchronize(st));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, h->ev[0], h->ev[3]));
    *mean_ms = ms / (float)iters;
    h->last_state = s;
    h->last_ext = ext;
    h->have_eval = true;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// lazy fetches
// ---------------------------------------------------------------------------------------------
int flh_fetch_selected(flh_handle* h, uint8_t* flags) {
    if (!h || !flags) return fail("flh_fetch_selected: null argument");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    std::vector<uint8_t> tmp(N);
    HIPC(hipMemcpyAsync(tmp.data(), h->selected.p, N, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) flags[perm[i]] = tmp[i];
    return 0;
}

int flh_fetch_neighbors(flh_handle* h, int32_t* idx, float* d2, uint8_t* cnt) {
    if (!h || !idx || !d2) return fail("flh_fetch_neighbors: null argument");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    std::vector<float4> pts(5 * N);
    std::vector<float> dd(5 * N);
    std::vector<uint8_t> cc(N);
    HIPC(hipMemcpyAsync(pts.data(), h->nn_pts.p, 5 * N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(dd.data(), h->nn_d2.p, 5 * N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(cc.data(), h->nn_cnt.p, N, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) {
        const size_t o = perm[i];
        for (int j = 0; j < 5; ++j) {
            int32_t id;
            std::memcpy(&id, &pts[(size_t)j * N + i].w, 4);
            idx[o * 5 + j] = id;
            d2[o * 5 + j] = id < 0 ? INFINITY : dd[(size_t)j * N + i];
        }
        if (cnt) cnt[o] = cc[i];
    }
    return 0;
}

int flh_fetch_world(flh_handle* h, float* xyz) {
    if (!h || !xyz) return fail("flh_fetch_world: null argument");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    std::vector<float4> w(N);
    HIPC(hipMemcpyAsync(w.data(), h->world.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) {
        const size_t o = perm[i];
        xyz[3 * o] = w[i].x; xyz[3 * o + 1] = w[i].y; xyz[3 * o + 2] = w[i].z;
    }
    return 0;
}

int flh_fetch_normvec(flh_handle* h, float* out) {
    if (!h || !out) return fail("flh_fetch_normvec: null argument");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    std::vector<float4> nv(N);
    HIPC(hipMemcpyAsync(nv.data(), h->normvec.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) std::memcpy(out + 4 * (size_t)perm[i], &nv[i], 16);
    return 0;
}

static void host_quat_rot(const double q[4], const double v[3], double o[3]) {
    double uvx = q[1] * v[2] - q[2] * v[1], uvy = q[2] * v[0] - q[0] * v[2], uvz = q[0] * v[1] - q[1] * v[0];
    uvx += uvx; uvy += uvy; uvz += uvz;
    const double cx = q[1] * uvz - q[2] * uvy, cy = q[2] * uvx - q[0] * uvz, cz = q[0] * uvy - q[1] * uvx;
    o[0] = (v[0] + q[3] * uvx) + cx;
    o[1] = (v[1] + q[3] * uvy) + cy;
    o[2] = (v[2] + q[3] * uvz) + cz;
}

// ekfom_data.h_x / ekfom_data.h in original order (src/laserMapping.cpp:720-752), rebuilt on the host from
// the device-resident planes.  Only the n_eff < 23 gain-form branch (esekfom.hpp:1715) and debugging use
// this; the hot path never materialises rows.
int flh_fetch_rows(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows) {
    if (!h || !n_rows) return fail("flh_fetch_rows: null argument");
    if (!h->have_eval) return fail("flh_fetch_rows: no evaluation yet");
    const size_t N = h->N;
    std::vector<uint8_t> sel(N ? N : 1);
    std::vector<float4> nv(N ? N : 1);
    if (N) {
        HIPC(hipSetDevice(h->device));
        HIPC(hipMemcpyAsync(sel.data(), h->selected.p, N, hipMemcpyDeviceToHost, h->stream));
        HIPC(hipMemcpyAsync(nv.data(), h->normvec.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
        HIPC(hipStreamSynchronize(h->stream));
    }
    int64_t n = 0;
    for (size_t i = 0; i < N; ++i) n += sel[i] ? 1 : 0;
    *n_rows = n;
    if (!hx || !hv) return 0;
    if (cap < n) return fail("flh_fetch_rows: buffers too small");
    std::vector<uint32_t> inv(N ? N : 1);
    for (size_t i = 0; i < N; ++i) inv[h->cur->h_perm[i]] = (uint32_t)i;
    const std::vector<float>& hbody = h->cur->h_body;
    const StateDev& s = h->last_state;
    const double rotc[4] = {-s.rot[0], -s.rot[1], -s.rot[2], s.rot[3]};
    const double offRc[4] = {-s.offR[0], -s.offR[1], -s.offR[2], s.offR[3]};
    int64_t k = 0;
    for (size_t o = 0; o < N; ++o) {  // original scan order, like the compaction loop at :697-706
        const size_t i = inv[o];
        if (!sel[i]) continue;
        const double pb[3] = {hbody[3 * o], hbody[3 * o + 1], hbody[3 * o + 2]};
        double pt[3], C[3];
        host_quat_rot(s.offR, pb, pt);
        for (int d = 0; d < 3; ++d) pt[d] += s.offT[d];
        const double nvec[3] = {nv[i].x, nv[i].y, nv[i].z};
        host_quat_rot(rotc, nvec, C);
        double row[12] = {nvec[0], nvec[1], nvec[2],
                          (-pt[2]) * C[1] + pt[1] * C[2], pt[2] * C[0] + (-pt[0]) * C[2], (-pt[1]) * C[0] + pt[0] * C[1],
                          0, 0, 0, 0, 0, 0};
        if (h->last_ext) {
            double D[3];
            host_quat_rot(offRc, C, D);
            row[6] = (-pb[2]) * D[1] + pb[1] * D[2];
            row[7] = pb[2] * D[0] + (-pb[0]) * D[2];
            row[8] = (-pb[1]) * D[0] + pb[0] * D[1];
            row[9] = C[0]; row[10] = C[1]; row[11] = C[2];
        }
        for (int c = 0; c < 12; ++c) hx[(size_t)c * n + k] = row[c];
        hv[k] = -(double)nv[i].w;
        ++k;
    }
    return 0;
}

}  // extern "C"
