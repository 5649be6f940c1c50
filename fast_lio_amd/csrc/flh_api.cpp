// flh_api.cpp -- layer 1 of the C ABI (include/fastlio_hip.h): device memory, map index build, scan
// upload, one h_share_model evaluation per call, lazy fetches.  Host side only; kernels live in
// flh_kernels.hip.  There is NO CPU fallback: without a HIP device every entry point fails loudly.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <emmintrin.h>
#include <immintrin.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <random>
#include <set>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fastlio_hip.h"
#include "../../include/fastlio_hip_dev.h"
#include "../../include/fastlio_amd/local_map.hpp"
#include "flh_kernels.hpp"
#include "flh_mail_dev.hpp"

using flh::GridParams;
using flh::StateDev;
// developer builds only (tools/variant.py): readers of the instrumentation records, one per kernel translation unit
#ifdef FLH_BOUNDS
namespace flh {
void bounds_read_kernels(unsigned long long out[5]);
void bounds_read_pass(unsigned long long out[5]);
void bounds_read_mapinc(unsigned long long out[5]);
void bounds_read_scanprep(unsigned long long out[5]);
void bounds_read_stage(unsigned long long out[5]);
}
#endif
#ifdef FLH_PASS_STAMPS
namespace flh { void pass_stamps_read(unsigned long long* out, size_t words); }
#endif
typedef unsigned long long u64;
#define FLH_COUNTER_WORDS 1

constexpr int kStagerPollUs = 400;  // how long the staging thread polls for its next job before it sleeps (stager_main)
constexpr int kGranGroups = 64;   // at most this many first-level groups go the granule way (else the in-kernel two-level sum)
constexpr int kGranSlots = 93;    // gran_section_slots(12): the Gram entries the filter reads + one statistic
constexpr size_t kGranSect = 1 + (size_t)kGranGroups * kGranSlots;  // granules of one rank's section: header + [group][slot]
static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return -1;
}
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
#define HIPC(expr)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

// The peers' shared granule segment: [dst rank][parity][src rank][kGranSect granules], then the row records of the gain-form
// branch [src rank][kPeerRec doubles].  One mapping per process (several handles of one process share it: refs).
struct PeerSeg {
    void* host = nullptr;     // mapped (shm) or hipHostMalloc'ed (one process) memory
    void* dev = nullptr;      // the device-side address of the same bytes
    size_t bytes = 0;
    bool shm = false, creator = false;
    std::string name;
    int refs = 0;
    int n = 0;
    struct flh_handle* members[8] = {};  // flh_peer_init_all (one process): the attached handles by rank
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        const size_t old = cap;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        // a buffer that has to grow grows by half at least: a run of slightly larger requests (the per-scan map changes of a stream)
        // costs one reallocation -- hipFree waits for the device -- not one per request
        size_t want = std::max(n + n / 8 + 64, old + old / 2);
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    // like reserve, but the first `keep` elements survive a reallocation
    hipError_t grow(size_t n, size_t keep, hipStream_t st) {
        if (n <= cap) return hipSuccess;
        T* np = nullptr;
        const size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void**)&np, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (p && keep) e = hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
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
    hipEvent_t ev[4]{};  // a timed evaluation: search start, search end, fit start, fit end (time stamps of the kernels themselves)
    // map
    size_t M = 0;
    GridParams grid{};
    DevBuf<float4> map_sorted;             // the storage the search reads: per brick its points (cell-sorted, .w = id) + slack
    DevBuf<float4> map_orig, map_next;     // the points by id (append-only between re-indexings), and a build buffer
    DevBuf<uint8_t> dead_id;               // 1 = the point with this id has been removed
    size_t n_ids = 0;                      // ids handed out since the last re-indexing (M = the live ones among them)
    DevBuf<uint32_t> cap_end, live;        // per brick rank: end of its storage range, live points
    DevBuf<uint32_t> mb_cap, mb_capincl;
    DevBuf<uint32_t> ctr;                  // device counters: [0] storage top, [1] bricks, [2] re-index flags, [3] removed
    uint32_t* h_ctr = nullptr;             // pinned mirror
    // A map change (flh_map_incremental with apply, flh_map_add) is ENQUEUED and returns; its counters come back as granules in
    // pinned memory (h_mi: [0..3] list lengths of map_incremental, [4..7] storage top / bricks / re-index flags / removed,
    // [8..11] inserted + sequence) and are folded into the host's bookkeeping by map_settle() when somebody needs them -- by
    // then the device has long finished, so nothing waits.
    uint32_t* h_mi = nullptr;
    uint32_t mi_seq = 0;
    // map_incremental without the wait in the middle (a change of a scan's usual size): the two list lengths stay on the device
    // (mi_cnt: [0] n1, [1] n, [2] survivors), Add_Points is enqueued with launches sized for small_change_max() points; a change
    // that turns out larger is not applied by those launches and replayed by map_settle()
    DevBuf<uint32_t> mi_cnt;
    DevBuf<uint32_t> mi_tick;          // k_brick_rewrite_heads' tickets (zero between launches)
    uint64_t n_search_redone = 0;      // searching passes run again because the map change in front of them asked for a re-index / replay
    DevBuf<uint32_t> mi_far;           // [0] the number of scan points whose nearest map point lies outside the search bound, then the points
    uint32_t mi_pred_n = 0xFFFFFFFFu;  // points of the previous change (the prediction for the next one); unknown at first
    bool mi_deferred = false;          // the pending change was enqueued that way
    uint32_t mi_cls_seq = 0;           // sequence word of the list-length granule of that change (h_mi[0..3])
    double mi_pending_ds = 0.0;        // its down-sampling length
    uint64_t n_mi_deferred = 0, n_mi_replayed = 0;
    bool map_pending = false;
    uint32_t map_pending_seq = 0;
    size_t pts_cap = 0, rows_cap = 0, alloc_top = 0;
    DevBuf<float4> ins;                    // points being inserted, with their ids
    std::vector<uint32_t> id_pos;          // id -> position among the live points (flh_fetch_neighbors), built on demand
    bool id_pos_valid = false;
    uint64_t n_reindex = 0, n_inplace = 0; // full re-indexings / changes applied brick-wise since creation
    uint64_t n_purged = 0;                 // bricks compacted in place after removals (k_brick_purge)
    DevBuf<u64> mb_k0, mb_k1;              // sort scratch of the index build and of the map updates (kept allocated)
    DevBuf<u64> vox_tab;                   // voxel hash table of a map change (key, best new point) x slots
    DevBuf<uint32_t> mb_v0, mb_v1, mb_bh, mb_br, mb_bstart, mb_aabb;
    DevBuf<unsigned char> mb_tmp;
    DevBuf<float4> mu_add, mi_world;       // incremental update: points to insert; map_incremental's world points
    DevBuf<uint8_t> mu_alive, mi_cls;
    DevBuf<uint32_t> mu_flags, mu_incl;
    DevBuf<uint32_t> mi_blk[2];        // map_incremental: the two lists' members per block of 256 original indices (double buffer)
    uint32_t mi_blk_dirty[2] = {0, 0}; // words of each half its last use wrote (zeroed by the other half's compaction)
    int mi_par = 0;                    // the half the next call's classification counts into
    DevBuf<float> mu_boxes;
    size_t mi_valid_N = (size_t)-1;        // N of the scan the last classification belongs to
    StateDev search_state{};               // state of the last do_search evaluation (Nearest_Points refer to it)
    DevBuf<uint2> hash;
    DevBuf<uint32_t> starts;
    uint32_t nbricks = 0;
    int rmax = 3;
    // scan
    size_t N = 0;
    DevBuf<float4> world, nn_pts, normvec;
    DevBuf<uint32_t> nn_idx;               // the neighbour cache as map indices (flh_config.index_cache): what a one-launch pass writes
    DevBuf<double> gsum;                   // RCCL path: [kGranGroups][slots] group totals of a pass, all-reduced in place (rows behind this rank's groups: zero)
    DevBuf<float4> plane;     // flh_config.plane_cache: (a, b, c, d) of the last searching pass's fits, reused by no-search passes
    bool plane_cache = false;
    bool planes_valid = false;  // `plane` holds the fits of the CURRENT neighbour cache (written by the fit that followed the last search)
    DevBuf<float> nn_d2;
    DevBuf<uint8_t> nn_cnt, selected;
    DevBuf<double> partials, part2, gram, gather_buf;
    DevBuf<u64> counter;
    DevBuf<uint32_t> slow_list, slow_list2, slow_count;  // work lists between the search stages (striped) and their counters
    DevBuf<float> slow_ub;                   // per-query bound on the 5th squared distance handed from A1 to A2
    DevBuf<uint32_t> tickets;                // arrival tickets of k_fit's in-kernel reduction (self re-arming)
    // multi-GPU (flh_rccl_*): the communicator this handle's rank belongs to; flh_eval all-reduces the Gram block over it
    ncclComm_t comm = nullptr;
    int comm_size = 1, comm_rank = 0;
    // map partitioned over the ranks: only queries whose world coordinate own_axis lies in [own_lo, own_hi) are searched here
    // launch plan of the search: number of queries the first stage left unsettled in the most recent LATER search of a scan
    // (-1 = unknown): when it was tiny, the next later search lets the first stage finish them itself and skips the second
    int64_t later_unsettled = -1;
    bool last_search_was_later = false;
    int own_axis = -1;
    float own_lo = -INFINITY, own_hi = INFINITY;
    // Granule buffers (pinned host memory, {value, sequence} pairs written by the passes' group reducers): two parities (a
    // rank may be one pass ahead of a peer that has not read the last one yet) x one section per rank.  h_gran is this
    // rank's own buffer; gran_dst lists every buffer this rank's kernels write (itself only, or all peers: flh_peer_*).
    double* h_gran = nullptr;
    bool gran_owned = true;        // h_gran came from hipHostMalloc (else: a window of the peers' shared segment)
    double* gran_dst[flh::kPeersMax] = {};
    int peer_n = 1, peer_rank = 0;
    int sect_ng[flh::kPeersMax] = {};   // groups in rank r's granule section for the ACTIVE scan (0 = not seen yet): learnt from a
                                        // section's header, the same for every pass of a scan (collect_granules)
    std::vector<double> gran_val;       // collect_granules' buffer when it takes a pass's granules in the order they arrive
    void* peer_map = nullptr;      // the mapped (and registered) shared segment
    size_t peer_map_bytes = 0;
    std::string peer_name;
    bool pass_ok = false;          // the one-launch searching pass may run (flh_config.pass_kernel and its requirements)
    uint64_t n_second_stage = 0;   // queries that needed the second search, summed over the searching passes since creation
    uint64_t n_search_pass = 0, n_one_launch = 0, n_nosearch_pass = 0;
    struct PendingEval {           // between flh_eval_begin and flh_eval_end
        bool active = false, do_search = false, granules = false, one_launch = false, timed = false, deferred = false;
        bool behind_map_change = false;  // enqueued behind a map change whose counters the host has not seen yet (flh_eval_end)
        uint64_t reindexed = 0, replayed = 0;  // n_reindex / n_mi_replayed at that moment
        int ext = 0;
        double seq = 0;
    } pend;
    double* h_gran_own = nullptr;  // the handle's own pinned buffer (h_gran points into the peers' segment while attached)
    struct PeerSeg* peer_seg = nullptr;
    double* h_gram = nullptr;  // pinned 256 doubles
    u64* h_counter = nullptr;  // pinned
    // last evaluation
    StateDev last_state{};
    int last_ext = 0;
    bool have_eval = false;
    bool stats = false;
    flh_timing timing{};
    bool searched_once = false;
    bool d2_valid = false;    // nn_d2 holds the distances of the current neighbour cache (filled on demand, ensure_d2)
    bool aux_valid = false;   // world / normvec hold the last evaluation's (filled on demand, ensure_aux)
    bool nn_pts_valid = true; // nn_pts holds the coordinates of the current neighbour cache (false: only nn_idx does; ensure_nn_pts gathers)
    int timing_stride = 1;   // record HIP events on every n-th evaluation (0 = never)
    bool timing_search_only = false;  // count (and time) SEARCHING evaluations only
    uint64_t eval_no = 0;
    // timing_stride >= 2 (sampling inside a running stream, bench.py): a timed evaluation records into the next free event
    // triple and does NOT wait for it -- waiting on an event costs the host tens of microseconds, more than a whole pass; the
    // elapsed times are read when somebody asks for them (drain_events).  timing_stride == 1 keeps the synchronous reading.
    static constexpr int kEvPool = 64;
    hipEvent_t evp[kEvPool][4]{};
    uint8_t evp_search[kEvPool]{};
    int evp_n = 0;            // triples recorded and not yet read
    bool evp_ready = false;   // the pool's events exist
    uint64_t seq = 0;        // sequence number of the last flh_eval (published by k_fit next to the result)
    double acc[6] = {0, 0, 0, 0, 0, 0};
    double acc_kind[4] = {0, 0, 0, 0};  // search-kernel ms / launches of a scan's FIRST search, of its LATER searches (bounded by the cache)
    // staging ring
    struct Slot {
        DevBuf<float4> body;            // Morton-ordered (internal order); .w = original index
        DevBuf<float4> dense;           // the cloud a raw-scan staging started from (feats_undistort), original order
        size_t n_dense = 0;
        std::vector<float> h_body;      // host copy, original order (lazy: ensure_host_copy)
        std::vector<uint32_t> h_perm;   // internal index -> original index (lazy)
        bool host_valid = false;
        size_t N = 0;
        hipEvent_t ready = nullptr, h2d_done = nullptr;
        hipEvent_t consumed = nullptr;  // recorded on the handle's stream behind the last ENQUEUED-AND-NOT-AWAITED reader of body
        bool consumed_set = false;      // (map_incremental's classification); the slot's next staging waits for it
        bool used = false;
        unsigned char* pin = nullptr;   // pinned staging buffer (pageable caller memory goes through it)
        size_t pin_cap = 0;
        bool pending = false;           // handed to the staging thread, not finished yet (guarded by st_mu)
        int async_rc = 0;
        std::string async_err;
        hipStream_t last_stream = nullptr;  // the stream the slot was staged on last (a staging on the other one waits for it)
    };
    struct StageJob { int slot; const void* pts; size_t stride; size_t N; };
    std::thread stager;                 // flh_scan_stage_async's worker
    std::mutex st_mu;
    std::mutex stage_mu;                // one staging at a time: the stager and the synchronous entry points share the scratch buffers below and the copy stream
    std::condition_variable st_cv, st_done;
    std::deque<StageJob> st_queue;
    bool st_quit = false;
    unsigned char* pin_in = nullptr;    // pinned scratch of the synchronous entry points
    size_t pin_in_cap = 0;
    unsigned char* pin_out = nullptr;
    size_t pin_out_cap = 0;
    uint32_t* h_small = nullptr;        // pinned: a few words read back by the staging paths
    DevBuf<unsigned char> st_bytes, fw_bytes;  // the caller's records as they came over PCIe
    DevBuf<float4> fw_in, fw_out;       // flh_frame_world / flh_points_body_to_world
    Slot slots[FLH_MAX_SLOTS + 1];      // [FLH_MAX_SLOTS] backs flh_scan_upload
    hipStream_t copy_stream = nullptr;
    // A second staging LANE (its own stream and its own scratch) for the plain staging of odd slots: the staging of a 100 000-point
    // scan occupies its stream for ~130 us (H2D 42 + re-stride, Morton sort, gather 80-90: profiles/r05_call4/) -- as long as the
    // update of a scan -- so with one lane the update waits for its scan whenever anything jitters.  With two, the scans after
    // next and after that are staged side by side (flh_esekf_run_scans keeps two in flight).
    hipStream_t copy_stream2 = nullptr;
    DevBuf<unsigned char> st2_bytes, st2_tmp;
    DevBuf<float4> st2_raw;
    DevBuf<uint32_t> st2_m0, st2_m1, st2_v0, st2_v1;
    const float4* cur_body = nullptr;   // the active slot's buffer
    Slot* cur = nullptr;
    // staging scratch (copy stream)
    DevBuf<float4> st_raw, ds_raw, ds_und; // ds_*: undistortion / voxel-grid down-sampling of a raw scan
    DevBuf<double> ds_poses;
    DevBuf<u64> ds_blockmin;               // k_undistort's per-block (time, index) minima
    DevBuf<uint32_t> ds_flags, ds_incl;
    DevBuf<u64> st_k0, st_k1;              // voxel keys of the scan's down-sampling
    DevBuf<uint32_t> st_m0, st_m1;          // Morton keys of the staging sort
    DevBuf<uint32_t> st_v0, st_v1;
    DevBuf<unsigned char> st_tmp;
    std::atomic<uint64_t> st_posted{0};  // jobs ever handed to the staging thread (what it polls before it sleeps)
    // developer counters (flh_debug_stage_stats): where a scan's staging and its activation spend their time.  [0..4] are written by
    // whoever stages (the staging thread in a running stream), [5..9] by the caller's thread; read after the stream has drained
    struct StageDiag {
        double n_jobs = 0, enq_us = 0, enq_max_us = 0, h2d_wait_us = 0, h2d_wait_max_us = 0;
        double n_act = 0, act_wait_us = 0, act_wait_max_us = 0, act_ev_not_ready = 0, act_slept = 0;
    } sdiag;
    // The pre-launched no-search pass (flh_eval_expect_next; device side: flh_mail_dev.hpp, k_fit_mb in flh_kernels.hip).
    //   expect     what the caller said the evaluation AFTER the next flh_eval_begin will be (consumed by that begin)
    //   armed      a k_fit_mb sits in the stream waiting for mailbox sequence `mseq`; it will publish with granule sequence `eval_seq`
    //   via_mail   the evaluation under way was started through the mailbox (collect_granules: if the kernel had already given up --
    //              the host came more than 20 ms late -- the pass is launched the usual way, pre_gone_relaunch)
    struct PreLaunch {
        bool off = false, armed = false, via_mail = false;
        int expect = 0;
        uint32_t mseq = 0;
        double eval_seq = 0;
        int ext = 0;
        size_t N = 0;
        const float4* body = nullptr;
        double* host_box = nullptr;
        unsigned long long* status = nullptr;
        double* dev_box = nullptr;
        uint64_t n_armed = 0, n_go = 0, n_abort = 0, n_gone = 0;
    } pre;
};

extern "C" {

static void release_build_scratch(flh_handle* h);
static void stop_stager(flh_handle* h);
static int rccl_allreduce_publish(flh_handle* h, double seq);
static void pre_release(flh_handle* h);
static void pre_cancel(flh_handle* h);
static StateDev make_state(const double rot[4], const double pos[3], const double offR[4], const double offT[3]);

const char* flh_last_error(void) { return g_err.c_str(); }

// a non-blocking stream at the highest (urgent) or the lowest priority the device offers
static hipError_t create_stream(hipStream_t* st, bool urgent) {
#ifndef FLH_NO_STREAM_PRIORITY  // (developer A/B builds only, tools/variant.py)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        if (hipStreamCreateWithPriority(st, hipStreamNonBlocking, urgent ? greatest : least) == hipSuccess) return hipSuccess;
    }
    (void)hipGetLastError();
#else
    (void)urgent;
#endif
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

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
    c->pass_kernel = -1;
    c->eigen_order = -1;
    c->plane_fit_dtype = 0;
    c->undistort_first_point = -1;
    c->plane_cache = -1;
    c->fused_small_changes = -1;
    c->prelaunch = -1;
    c->index_cache = -1;
    c->stage_sort = -1;
}

static std::mutex g_dev_mu;
static std::set<int> g_devices_used;  // devices a handle of this process has been created on (flh_host_free waits for these only)

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
    if (cfg.pass_kernel != 0) cfg.pass_kernel = 1;
    if (cfg.eigen_order < 0 || cfg.eigen_order > 3) cfg.eigen_order = FLH_ORDER_SSE;
    if (cfg.plane_fit_dtype != 1) cfg.plane_fit_dtype = 0;
    if (cfg.undistort_first_point != 0) cfg.undistort_first_point = 1;
    if (cfg.plane_cache != 0) cfg.plane_cache = 1;
    if (cfg.fused_small_changes != 0) cfg.fused_small_changes = 1;
    if (cfg.prelaunch != 0) cfg.prelaunch = 1;
    if (cfg.index_cache != 0) cfg.index_cache = 1;
    if (cfg.stage_sort != 0 && cfg.stage_sort != 2) cfg.stage_sort = 1;
    if (cfg.lanes_per_query != 0) cfg.lanes_per_query = 4;  // 0 = exact kernel for every query
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
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        g_devices_used.insert(h->device);
    }
    if (cfg.stream) {
        h->stream = (hipStream_t)cfg.stream;
    } else {
        // the update's kernels are the latency chain of a scan: its stream gets the device's highest priority, the copy stream
        // (staging of the NEXT scan, beside this one's update) the lowest, so that staging kernels take the wave slots the update
        // leaves and not the other way round
        hipError_t e = create_stream(&h->stream, true);
        if (e != hipSuccess) {
            delete h;
            return fail(std::string("hipStreamCreate: ") + hipGetErrorString(e));
        }
        h->own_stream = true;
    }
    for (auto& e : h->ev) (void)hipEventCreate(&e);
    if (hipHostMalloc((void**)&h->h_gram, 256 * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_counter, sizeof(u64), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_ctr, 8 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_mi, 16 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_small, 16 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&h->h_gran, 2 * kGranSect * 16, hipHostMallocDefault) != hipSuccess) {
        flh_destroy(h);
        return fail("hipHostMalloc failed");
    }
    std::memset(h->h_gram, 0, 256 * sizeof(double));
    std::memset(h->h_mi, 0, 16 * sizeof(uint32_t));
    std::memset(h->h_gran, 0, 2 * kGranSect * 16);
    h->gran_dst[0] = h->h_gran;
    if (h->gram.reserve(256) != hipSuccess || h->counter.reserve(FLH_COUNTER_WORDS) != hipSuccess || h->slow_count.reserve(2 * flh::list_stripes()) != hipSuccess ||
        hipMemset(h->slow_count.p, 0, 2 * flh::list_stripes() * sizeof(uint32_t)) != hipSuccess) {
        flh_destroy(h);
        return fail("hipMalloc failed");
    }
    // The plane cache (and with it the index-only neighbour cache) exists for the default summation order and the fp32 fit only:
    // launch_fit re-fits from the neighbours' COORDINATES for every other order, so those must stay in nn_pts (with the
    // cache left on, a no-search pass of eigen_order 0 / 2 / 3 read coordinates nobody had written).
    h->plane_cache = cfg.plane_cache != 0 && cfg.eigen_order == FLH_ORDER_SSE && cfg.plane_fit_dtype == 0;
    h->pre.off = cfg.prelaunch == 0;
    h->rmax = (int)std::ceil((std::sqrt((double)cfg.max_sqdist) + 2e-3 * cfg.cell_size) / cfg.cell_size);
    if (h->rmax < 1) h->rmax = 1;
    {
        // k_pass's second search visits a 4x4 window of rows: enough as long as the ball of a bounded query stays below 1.5
        // cells (the very expression ring_query clips its rows with, at the largest bound the gate allows)
        const float inv_c = 1.0f / cfg.cell_size;
        const float ubq = cfg.max_sqdist * 1.0001f + 1e-6f;
        const float rcell2 = ubq * (inv_c * inv_c) * 1.01f + 1e-4f;
        h->pass_ok = cfg.pass_kernel != 0 && cfg.lanes_per_query == 4 && cfg.plane_fit_dtype == 0 && h->rmax <= 2 && rcell2 < 1.499f * 1.499f;
    }
    *out = h;
    return 0;
}

void flh_destroy(flh_handle* h) {
    if (!h) return;
    stop_stager(h);
    (void)hipSetDevice(h->device);
    pre_cancel(h);
    flh_rccl_destroy(h);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->copy_stream2) (void)hipStreamSynchronize(h->copy_stream2);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    release_build_scratch(h);
    h->dead_id.release(); h->cap_end.release(); h->live.release(); h->mb_cap.release(); h->mb_capincl.release(); h->ctr.release();
    h->ins.release();
    if (h->h_ctr) (void)hipHostFree(h->h_ctr);
    if (h->h_mi) (void)hipHostFree(h->h_mi);
    h->map_orig.release(); h->map_next.release(); h->mb_aabb.release(); h->mu_add.release(); h->mi_world.release(); h->mi_cnt.release(); h->mi_far.release(); h->mi_tick.release();
    h->mu_alive.release(); h->mi_cls.release(); h->mu_flags.release(); h->mu_incl.release(); h->mi_blk[0].release(); h->mi_blk[1].release(); h->mu_boxes.release();
    h->map_sorted.release(); h->hash.release(); h->starts.release(); h->slow_list.release(); h->slow_list2.release(); h->slow_ub.release(); h->slow_count.release(); h->tickets.release();
    h->world.release(); h->nn_pts.release(); h->normvec.release(); h->plane.release(); h->nn_idx.release(); h->gsum.release();
    h->nn_d2.release(); h->nn_cnt.release(); h->selected.release(); h->vox_tab.release();
    h->partials.release(); h->part2.release(); h->gram.release(); h->gather_buf.release(); h->counter.release();
    for (auto& sl : h->slots) {
        sl.body.release();
        sl.dense.release();
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.h2d_done) (void)hipEventDestroy(sl.h2d_done);
        if (sl.consumed) (void)hipEventDestroy(sl.consumed);
        if (sl.pin) (void)hipHostFree(sl.pin);
    }
    if (h->pin_in) (void)hipHostFree(h->pin_in);
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    if (h->h_small) (void)hipHostFree(h->h_small);
    h->st_bytes.release(); h->fw_bytes.release(); h->fw_in.release(); h->fw_out.release();
    h->ds_raw.release(); h->ds_und.release(); h->ds_poses.release(); h->ds_blockmin.release(); h->ds_flags.release(); h->ds_incl.release();
    h->st_raw.release(); h->st_k0.release(); h->st_k1.release(); h->st_m0.release(); h->st_m1.release(); h->st_v0.release(); h->st_v1.release(); h->st_tmp.release();
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->copy_stream2) (void)hipStreamDestroy(h->copy_stream2);
    h->st2_bytes.release(); h->st2_tmp.release(); h->st2_raw.release(); h->st2_m0.release(); h->st2_m1.release(); h->st2_v0.release(); h->st2_v1.release();
    if (h->h_gram) (void)hipHostFree(h->h_gram);
    flh_peer_close(h);
    if (h->h_gran && h->gran_owned) (void)hipHostFree(h->h_gran);
    if (h->h_counter) (void)hipHostFree(h->h_counter);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& t3 : h->evp)
        for (auto& e : t3)
            if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    pre_release(h);
    delete h;
}

static int map_settle(flh_handle* h);
static int apply_map_changes(flh_handle* h, const float4* d_add, size_t n1, size_t n2, double ds, const uint32_t* d_cnt = nullptr,
                             bool inserted = false);
int flh_map_sync(flh_handle* h) {
    if (!h) return fail("flh_map_sync: null handle");
    return map_settle(h);
}
size_t flh_map_size(const flh_handle* h) {
    if (!h) return 0;
    (void)map_settle(const_cast<flh_handle*>(h));  // a map change still under way on the device decides the size
    return h->M;
}
size_t flh_scan_size(const flh_handle* h) { return h ? h->N : 0; }

// ---------------------------------------------------------------------------------------------
// Map index (re)build from a device array of points in INDEX order.  `pts` is h->map_orig or h->map_next; on
// success it becomes h->map_orig.  A failure before the tables are touched (extent / size checks) leaves the previous map
// in place; a later one (allocation, a launch) leaves the handle without a map.
static int rebuild_index_impl(flh_handle* h, DevBuf<float4>& pts, size_t M, bool& touched);
// Map index (re)build.  The new index is written over the old one's buffers, so a failure after that point leaves no
// usable map: the handle is then marked map-less (flh_eval refuses) instead of pointing at half-written tables.
static int ensure_nn_pts(flh_handle* h);
static int rebuild_index(flh_handle* h, DevBuf<float4>& pts, size_t M) {
    bool touched = false;
    if (ensure_nn_pts(h) != 0) return -1;  // a re-indexing renumbers the ids an index-only neighbour cache refers to: coordinates first
    const int rc = rebuild_index_impl(h, pts, M, touched);
    if (rc != 0 && touched) {
        h->M = 0;
        h->n_ids = 0;
        h->nbricks = 0;
        h->grid = GridParams{};
        h->searched_once = false;
        h->id_pos_valid = false;
    }
    return rc;
}
static int rebuild_index_impl(flh_handle* h, DevBuf<float4>& pts, size_t M, bool& touched) {
    hipStream_t st = h->stream;
    const float c = h->cfg.cell_size;
    const uint32_t Mu = (uint32_t)M;
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    if (M > 0) {  // exact AABB on the device
        HIPC(h->mb_aabb.reserve(6));
        const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
        uint32_t got[6];
        HIPC(hipMemcpyAsync(h->mb_aabb.p, init, sizeof(init), hipMemcpyHostToDevice, st));
        HIPC(flh::launch_aabb(pts.p, Mu, h->mb_aabb.p, st));
        HIPC(hipMemcpyAsync(got, h->mb_aabb.p, sizeof(got), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        for (int d = 0; d < 6; ++d) {
            const uint32_t u = got[d];
            const uint32_t bits = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
            float f;
            std::memcpy(&f, &bits, 4);
            (d < 3 ? mn[d] : mx[d - 3]) = f;
        }
    }
    GridParams g{};
    g.c = c;
    g.inv_c = 1.0f / c;
    const int PAD = 32;  // cells of slack on every side: room for the map to grow before the grid must be re-laid
    float o[3];
    int dims[3];
    for (int d = 0; d < 3; ++d) {
        o[d] = (std::floor(mn[d] / c) - PAD) * c;
        dims[d] = (int)std::floor((mx[d] - o[d]) / c) + 1 + PAD;
        if (dims[d] > 4096)
            return fail("map index: map extent exceeds 4096 cells along an axis; raise flh_config.cell_size");
    }
    g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
    g.nx = dims[0]; g.ny = dims[1]; g.nz = dims[2];

    // sort by (brick, local cell), find the bricks, give each its range of the storage with slack behind its points
    uint32_t nbricks = 0, used = 0;
    if (M > 0) {
        HIPC(h->mb_k0.reserve(M)); HIPC(h->mb_k1.reserve(M));
        HIPC(h->mb_v0.reserve(M)); HIPC(h->mb_v1.reserve(M)); HIPC(h->mb_bh.reserve(M)); HIPC(h->mb_br.reserve(M));
        HIPC(flh::launch_map_keys(g, pts.p, Mu, h->mb_k0.p, h->mb_v0.p, st));  // keys only need origin/extent
        size_t tb1 = 0, tb2 = 0;
        HIPC(flh::sort_pairs(nullptr, tb1, h->mb_k0.p, h->mb_k1.p, h->mb_v0.p, h->mb_v1.p, Mu, st));
        HIPC(flh::inclusive_sum(nullptr, tb2, h->mb_bh.p, h->mb_br.p, Mu, st));
        HIPC(h->mb_tmp.reserve(std::max(tb1, tb2)));
        size_t tb = h->mb_tmp.cap;
        HIPC(flh::sort_pairs(h->mb_tmp.p, tb, h->mb_k0.p, h->mb_k1.p, h->mb_v0.p, h->mb_v1.p, Mu, st));
        HIPC(flh::launch_brick_heads(h->mb_k1.p, Mu, h->mb_bh.p, st));
        tb = h->mb_tmp.cap;
        HIPC(flh::inclusive_sum(h->mb_tmp.p, tb, h->mb_bh.p, h->mb_br.p, Mu, st));
        HIPC(hipMemcpyAsync(&nbricks, h->mb_br.p + (M - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        HIPC(h->mb_bstart.reserve((size_t)nbricks + 1));
        HIPC(h->mb_cap.reserve(nbricks)); HIPC(h->mb_capincl.reserve(nbricks));
        HIPC(flh::launch_brick_starts(h->mb_bh.p, h->mb_br.p, Mu, h->mb_bstart.p, st));
        HIPC(hipMemcpyAsync(h->mb_bstart.p + nbricks, &Mu, sizeof(uint32_t), hipMemcpyHostToDevice, st));
        HIPC(flh::launch_brick_caps(h->mb_bstart.p, nbricks, h->mb_cap.p, st));
        tb = h->mb_tmp.cap;
        HIPC(flh::inclusive_sum(h->mb_tmp.p, tb, h->mb_cap.p, h->mb_capincl.p, nbricks, st));
        HIPC(hipMemcpyAsync(&used, h->mb_capincl.p + (nbricks - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    // room behind the bricks for relocated / new bricks, table rows and directory slots for new bricks
    const size_t pts_cap = (size_t)used + std::max<size_t>(M / 4, 65536);
    if (pts_cap >= (1ull << 27)) return fail("map index: too many points for one storage range");
    const size_t rows_cap = (size_t)2 * nbricks + 4096;
    // the search reads the prefix tables through a buffer resource with 32-bit byte offsets (flh_search_dev.hpp: RingRsrc), the
    // points likewise (pts_cap * 16 < 2^31, checked above)
    if (rows_cap * flh::kBrickStride * sizeof(uint32_t) >= (1ull << 32)) return fail("map index: too many bricks for 32-bit table offsets");
    uint32_t hs = 1024;
    while (hs < 2 * rows_cap) hs <<= 1;
    int log2hs = 0;
    while ((1u << log2hs) < hs) ++log2hs;
    touched = true;  // from here on the previous index is being overwritten
    HIPC(h->map_sorted.reserve(pts_cap));
    HIPC(h->hash.reserve(hs));
    HIPC(h->starts.reserve(rows_cap * flh::kBrickStride));
    HIPC(h->cap_end.reserve(rows_cap)); HIPC(h->live.reserve(rows_cap));
    HIPC(h->ctr.reserve(8));
    HIPC(hipMemsetAsync(h->hash.p, 0xFF, (size_t)hs * sizeof(uint2), st));
    HIPC(hipMemsetAsync(h->live.p, 0, rows_cap * sizeof(uint32_t), st));
    HIPC(flh::launch_fill_tomb(h->map_sorted.p, (uint32_t)pts_cap, st));
    if (M > 0) {
        HIPC(flh::launch_map_place(pts.p, h->mb_v1.p, h->mb_br.p, h->mb_bstart.p, h->mb_capincl.p, h->mb_cap.p, Mu, h->map_sorted.p, st));
        HIPC(flh::launch_brick_tables(h->mb_k1.p, h->mb_bstart.p, nbricks, h->mb_capincl.p, h->mb_cap.p, h->starts.p, h->cap_end.p,
                                      h->live.p, h->hash.p, hs - 1, 32 - log2hs, st));
    }
    const uint32_t ctr0[8] = {used, nbricks, 0u, 0u, 0u, 0u, 0u, 0u};
    HIPC(hipMemcpyAsync(h->ctr.p, ctr0, sizeof(ctr0), hipMemcpyHostToDevice, st));
    // identities restart at 0..M-1 = positions in the compacted array
    HIPC(h->dead_id.reserve(M + M / 4 + 65536));
    HIPC(hipMemsetAsync(h->dead_id.p, 0, h->dead_id.cap, st));
    HIPC(hipStreamSynchronize(st));
    g.hash_mask = hs - 1;
    g.hash_shift = 32 - log2hs;
    g.hash = h->hash.p;
    g.starts = h->starts.p;
    g.pts = h->map_sorted.p;
#ifdef FLH_BOUNDS
    g.pts_cap = pts_cap; g.rows_cap = rows_cap; g.ids_cap = std::min(pts.cap, h->dead_id.cap);  // (pts becomes map_orig below)
#endif
    h->grid = g;
    h->nbricks = nbricks;
    h->M = M;
    h->n_ids = M;
    h->pts_cap = pts_cap;
    h->rows_cap = rows_cap;
    h->alloc_top = used;
    ++h->n_reindex;
    h->id_pos_valid = false;
    if (&pts != &h->map_orig) std::swap(h->map_orig, pts);
    h->searched_once = false;  // cached neighbours refer to the previous map
    return 0;
}

static void release_build_scratch(flh_handle* h) {
    h->mb_k0.release(); h->mb_k1.release(); h->mb_v0.release(); h->mb_v1.release(); h->mb_bh.release(); h->mb_br.release();
    h->mb_bstart.release(); h->mb_tmp.release();
}

// ---------------------------------------------------------------------------------------------
// ikdtree.Build -- src/laserMapping.cpp:919
// ---------------------------------------------------------------------------------------------
static int upload_points(flh_handle* h, const char* who, const void* xyz, size_t stride_bytes, size_t n, DevBuf<float4>& dst) {
    if (n > 0 && !xyz) return fail(std::string(who) + ": null points");
    if (stride_bytes < 12) return fail(std::string(who) + ": stride_bytes < 12");
    std::vector<float4> hp(n ? n : 1);
    const unsigned char* src = (const unsigned char*)xyz;
    for (size_t i = 0; i < n; ++i) {
        float p[3];
        std::memcpy(p, src + i * stride_bytes, 12);
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2]))
            return fail(std::string(who) + ": non-finite point at index " + std::to_string(i));
        hp[i] = make_float4(p[0], p[1], p[2], 0.f);
    }
    HIPC(dst.reserve(n ? n : 1));
    if (n > 0) {
        HIPC(hipMemcpyAsync(dst.p, hp.data(), n * sizeof(float4), hipMemcpyHostToDevice, h->stream));
        HIPC(hipStreamSynchronize(h->stream));  // hp is pageable and dies here
    }
    return 0;
}

int flh_map_build(flh_handle* h, const void* xyz, size_t stride_bytes, size_t M) {
    if (!h) return fail("flh_map_build: null handle");
    if (M >= (1ull << 31)) return fail("flh_map_build: M too large");
    HIPC(hipSetDevice(h->device));
    if (h->map_pending) {  // a change of the map that is being replaced: let it finish, its counters no longer matter
        HIPC(hipStreamSynchronize(h->stream));
        h->map_pending = false;
    }
    if (upload_points(h, "flh_map_build", xyz, stride_bytes, M, h->map_next) != 0) return -1;
    const int rc = rebuild_index(h, h->map_next, M);
    if (rc != 0) {  // Build replaces the map: a failed build leaves none
        h->M = 0;
        h->searched_once = false;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------
// Incremental map -- SURVEY.md 8(f) row 1.
// The points live twice: by id in map_orig (append-only; dead_id marks removals) and in the brick storage the search
// reads.  A change touches only the bricks it concerns (k_add_resolve tombstones displaced points in place,
// k_brick_rewrite re-sorts the bricks that receive points, relocating one when it outgrows its slack); when something
// does not fit -- a point outside the grid, storage / table rows / directory full, a brick beyond the LDS tile -- the
// index is rebuilt from map_orig, which is always complete, and the ids are renumbered 0..M-1.
static int reindex_from_ids(flh_handle* h) {
    hipStream_t st = h->stream;
    pre_cancel(h);  // (a pass enqueued ahead of its state would sit in front of the synchronisations below)
    const size_t n_ids = h->n_ids;
    size_t total = 0;
    if (n_ids > 0) {
        HIPC(h->mu_flags.reserve(n_ids)); HIPC(h->mu_incl.reserve(n_ids));
        HIPC(flh::launch_byte_flags(h->dead_id.p, (uint32_t)n_ids, 1, h->mu_flags.p, st));
        size_t tb = 0;
        HIPC(flh::inclusive_sum(nullptr, tb, h->mu_flags.p, h->mu_incl.p, (uint32_t)n_ids, st));
        HIPC(h->mb_tmp.reserve(tb));
        tb = h->mb_tmp.cap;
        HIPC(flh::inclusive_sum(h->mb_tmp.p, tb, h->mu_flags.p, h->mu_incl.p, (uint32_t)n_ids, st));
        uint32_t t32 = 0;
        HIPC(hipMemcpyAsync(&t32, h->mu_incl.p + (n_ids - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
        total = t32;
        HIPC(h->map_next.reserve(total ? total : 1));
        HIPC(flh::launch_live_compact(h->map_orig.p, h->mu_flags.p, h->mu_incl.p, (uint32_t)n_ids, h->map_next.p, st));
    } else {
        HIPC(h->map_next.reserve(1));
    }
    return rebuild_index(h, h->map_next, total);
}

// Waits until the granule at h_mi[off .. off+3] carries `seq` in its last word (written by one 16-byte system-scope store of a
// kernel on the handle's stream); a stream that finished or failed without publishing is an error, not a hang.
static int wait_granule(flh_handle* h, int off, uint32_t seq, const char* who) {
    const volatile uint32_t* g = h->h_mi + off;
    uint64_t spins = 0;
    while (g[3] != seq) {
        cpu_relax();
        if ((++spins & 0xFFFFFu) == 0 && hipStreamQuery(h->stream) != hipErrorNotReady) {
            HIPC(hipStreamSynchronize(h->stream));
            if (g[3] != seq) return fail(std::string(who) + ": the device finished without publishing its counters");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}

// Folds the counters of the map change that is under way (if any) into the host's bookkeeping; re-indexes when the change
// said so.  Called by everything that reads the map's size / ids / tables or launches a search.
static int map_settle(flh_handle* h) {
    if (!h->map_pending) return 0;
    HIPC(hipSetDevice(h->device));
    // two granules, EACH with the sequence word: {top, bricks, flags, seq} {removed, inserted, 0, seq} (system-scope stores of one
    // kernel need not reach the host in order)
    if (wait_granule(h, 4, h->map_pending_seq, "map change") != 0 || wait_granule(h, 8, h->map_pending_seq, "map change") != 0) return -1;
    h->map_pending = false;
    const uint32_t top = h->h_mi[4], bricks = h->h_mi[5], flags = h->h_mi[6], removed = h->h_mi[8], n_alive = h->h_mi[9];
    if (h->h_mi[10] != 0xFFFFFFFFu) h->mi_pred_n = h->h_mi[10];  // the size of this change predicts the next one's
    if (h->mi_deferred) {
        h->mi_deferred = false;
        if (flags & flh::kMapChangeNotApplied) {
            // larger than the launches it was enqueued with: none of its kernels did anything.  The lists are still where
            // k_cls_compact put them (mu_add), their lengths arrived long ago: Add_Points again, with launches that fit.
            if (wait_granule(h, 0, h->mi_cls_seq, "map change (replay)") != 0) return -1;
            const uint32_t c1 = h->h_mi[0], c2 = h->h_mi[1] - h->h_mi[0];
            ++h->n_mi_replayed;
            pre_cancel(h);  // (a pass enqueued ahead of its state would sit in front of the replayed kernels)
            if (apply_map_changes(h, h->mu_add.p, c1, c2, h->mi_pending_ds) != 0) return -1;
            return map_settle(h);
        }
    }
    h->n_ids += n_alive;
    h->M = h->M + n_alive - removed;
    h->id_pos_valid = false;
    if (flags != 0) return reindex_from_ids(h);
    ++h->n_inplace;
    h->alloc_top = top;
    h->nbricks = bricks;
    // ids (and the id-ordered array) only grow between re-indexings: renumber once the removed ones outweigh half the map
    if (h->n_ids > h->M + h->M / 2 + (1u << 20)) return reindex_from_ids(h);
    return 0;
}

// d_add holds n1 points to insert WITH down-sampling followed by n2 points to insert as they are.  Everything is enqueued on
// the handle's stream with launch sizes the host knows (n1, n2; the number of points that survive the down-sampling stays on the
// device: n bounds it, and entries beyond it carry a sentinel key); the change's counters are collected by map_settle().
static int apply_map_changes(flh_handle* h, const float4* d_add, size_t n1, size_t n2, double ds, const uint32_t* d_cnt,
                             bool inserted) {  // inserted: the caller's kernels have done launch_add_insert's step for exactly this change
    hipStream_t st = h->stream;
    if (map_settle(h) != 0) return -1;
    // d_cnt: the true {n1, n} live on the device (mi_cnt); n1 = n2's sum is then only the bound the launches are sized for
    const size_t n = d_cnt ? n1 : n1 + n2;
    if (!d_cnt) h->mi_pred_n = (uint32_t)std::min<size_t>(n, 0xFFFFFFFEu);
    if (h->n_ids + n >= (1ull << 31)) return fail("map update: too many points");
    if (!h->grid.hash) {  // no map yet: index an empty one so there are tables to insert into
        if (rebuild_index(h, h->map_orig, 0) != 0) return -1;
    }
    if (n == 0) return 0;
    const uint32_t nu = (uint32_t)n;
    // (scratch sized for what the NEXT change's launches will be sized for -- this change's points + 50 % + 1024, see
    // flh_map_incremental -- so that a stream's second change does not reallocate everything its first one allocated)
    const size_t nr = d_cnt ? n : n + n / 2 + 1024;
    HIPC(h->mu_alive.reserve(nr));
    HIPC(h->mb_k0.reserve(nr)); HIPC(h->mb_k1.reserve(nr)); HIPC(h->mb_v0.reserve(nr)); HIPC(h->mb_v1.reserve(nr));
    HIPC(h->mu_flags.reserve(nr)); HIPC(h->mu_incl.reserve(nr));
    size_t tb_sort = 0, tb_scan = 0;
    uint32_t* const bk0 = reinterpret_cast<uint32_t*>(h->mb_k0.p);  // the brick keys of the surviving points are 32-bit
    uint32_t* const bk1 = reinterpret_cast<uint32_t*>(h->mb_k1.p);
    HIPC(flh::sort_brick_pairs(nullptr, tb_sort, bk0, bk1, h->mb_v0.p, h->mb_v1.p, (uint32_t)nr, st));
    HIPC(flh::inclusive_sum(nullptr, tb_scan, h->mu_flags.p, h->mu_incl.p, (uint32_t)nr, st));
    HIPC(h->mb_tmp.reserve(std::max(tb_sort, tb_scan)));
    // room for every point of the change (the survivors are at most n): allocated up front, nothing to wait for in between
    HIPC(h->map_orig.grow(h->n_ids + n, h->n_ids, st));  // (grow() allocates a quarter more than asked: a stream seldom reallocates)
    {
        const size_t before = h->dead_id.cap;
        HIPC(h->dead_id.grow(h->n_ids + n, h->n_ids, st));
        if (h->dead_id.cap != before) HIPC(hipMemsetAsync(h->dead_id.p + h->n_ids, 0, h->dead_id.cap - h->n_ids, st));
    }
    HIPC(h->ins.reserve(nr));
#ifdef FLH_BOUNDS
    h->grid.ids_cap = std::min(h->map_orig.cap, h->dead_id.cap);
#endif
    // the points inserted with down-sampling are grouped by voxel in a hash table (no sort): per voxel the best new point, which
    // then meets the points the map already holds there
    const uint32_t vcap = flh::vox_table_slots((uint32_t)n1);
    if (n1 > 0) {
        HIPC(h->vox_tab.reserve(2 * (size_t)flh::vox_table_slots((uint32_t)std::min<size_t>(nr, 0x7FFFFFFFu))));
        if (!inserted) HIPC(hipMemsetAsync(h->vox_tab.p, 0xFF, 2 * (size_t)vcap * sizeof(unsigned long long), st));
    }
    // where the number of surviving points goes (with device-side lengths the general path's scan runs over the whole bound: the
    // entries behind the change's true end count as "no point", k_add_insert)
    const bool small_path = h->cfg.fused_small_changes != 0 && nu <= flh::small_change_max();
    uint32_t* const d_alive_out = (d_cnt && small_path) ? h->mi_cnt.p + 2 : h->mu_incl.p + (n - 1);
    if (!inserted) HIPC(flh::launch_add_insert(d_add, (uint32_t)n1, nu, ds, h->vox_tab.p, vcap, h->mu_alive.p, h->ctr.p, st, d_cnt));
    if (n1 > 0)
        HIPC(flh::launch_add_resolve(h->grid, h->map_sorted.p, d_add, h->vox_tab.p, vcap, (uint32_t)n1, ds, h->dead_id.p, h->live.p,
                                     h->ctr.p, h->mu_alive.p, st, d_cnt));
    const uint32_t seq = ++h->mi_seq;
    if (small_path) {
        // a scan's worth of points: ids, brick keys and their sort in one workgroup (one launch instead of eight), which also lists
        // the bricks that receive points; the rewrite's last workgroup publishes the change's counters
        HIPC(flh::launch_ins_sort_small(h->grid, d_add, h->mu_alive.p, nu, (uint32_t)h->n_ids, h->map_orig.p, h->dead_id.p, h->ins.p, bk0,
                                        bk1, h->mb_v1.p, h->ctr.p, d_alive_out, h->mb_v0.p, st, d_cnt));
        if (h->mi_tick.cap < flh::brick_ticket_words()) {
            HIPC(h->mi_tick.reserve(flh::brick_ticket_words()));
            HIPC(hipMemsetAsync(h->mi_tick.p, 0, h->mi_tick.cap * sizeof(uint32_t), st));
        }
        HIPC(flh::launch_brick_rewrite_heads(h->grid, h->map_sorted.p, h->starts.p, h->hash.p, h->cap_end.p, h->live.p, h->ctr.p, h->ins.p,
                                             bk1, h->mb_v1.p, nu, (uint32_t)h->pts_cap, (uint32_t)h->rows_cap, st, d_cnt, h->mb_v0.p,
                                             h->mi_tick.p, d_alive_out, h->h_mi + 4, seq));
    } else {
        // ids of the survivors, in input order; the brick keys start as sentinels
        HIPC(flh::launch_byte_flags(h->mu_alive.p, nu, 0, h->mu_flags.p, st, bk0));
        {
            size_t tb = h->mb_tmp.cap;
            HIPC(flh::inclusive_sum(h->mb_tmp.p, tb, h->mu_flags.p, h->mu_incl.p, nu, st));
        }
        HIPC(flh::launch_ins_prepare(h->grid, d_add, h->mu_alive.p, h->mu_incl.p, nu, (uint32_t)h->n_ids, h->map_orig.p,
                                     h->dead_id.p, h->ins.p, bk0, h->mb_v0.p, h->ctr.p, st));
        {
            size_t tb = h->mb_tmp.cap;
            HIPC(flh::sort_brick_pairs(h->mb_tmp.p, tb, bk0, bk1, h->mb_v0.p, h->mb_v1.p, nu, st));
        }
        HIPC(flh::launch_brick_rewrite(h->grid, h->map_sorted.p, h->starts.p, h->hash.p, h->cap_end.p, h->live.p, h->ctr.p, h->ins.p,
                                       bk1, h->mb_v1.p, nu, (uint32_t)h->pts_cap, (uint32_t)h->rows_cap, st, d_cnt));
        HIPC(flh::launch_map_publish(h->ctr.p, d_alive_out, h->h_mi + 4, seq, st, d_cnt, nu));
    }
    h->mi_deferred = d_cnt != nullptr;
    h->mi_pending_ds = ds;
    h->map_pending = true;
    h->map_pending_seq = seq;
    h->id_pos_valid = false;
    h->searched_once = false;  // cached neighbours refer to the previous map
    return 0;
}

// ikdtree.Add_Points(points, downsample_on) -- src/laserMapping.cpp:470-471 (down-sampling length = the
// filter_size_map_min handed to ikdtree.set_downsample_param, :868)
int flh_map_add(flh_handle* h, const void* xyz, size_t stride_bytes, size_t n, int downsample, double downsample_size) {
    if (!h) return fail("flh_map_add: null handle");
    if (downsample && !(downsample_size > 0)) return fail("flh_map_add: downsample_size must be > 0");
    HIPC(hipSetDevice(h->device));
    if (map_settle(h) != 0) return -1;
    if (upload_points(h, "flh_map_add", xyz, stride_bytes, n, h->mu_add) != 0) return -1;
    return apply_map_changes(h, h->mu_add.p, downsample ? n : 0, downsample ? 0 : n, downsample_size);
}

// ikdtree.Delete_Point_Boxes(cub_needrm) -- src/laserMapping.cpp:275.  boxes: nb x {min xyz, max xyz}, a point is
// removed when min <= p < max on every axis: its storage slot becomes a tombstone, its id is marked dead.
int flh_map_delete_boxes(flh_handle* h, const float* boxes, size_t nb) {
    if (!h) return fail("flh_map_delete_boxes: null handle");
    if (nb > 0 && !boxes) return fail("flh_map_delete_boxes: null boxes");
    if (map_settle(h) != 0) return -1;
    if (nb == 0 || h->M == 0) return 0;
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    HIPC(h->mu_boxes.reserve(6 * nb));
    HIPC(hipMemcpyAsync(h->mu_boxes.p, boxes, 6 * nb * sizeof(float), hipMemcpyHostToDevice, st));
    HIPC(hipMemsetAsync(h->ctr.p + 3, 0, 2 * sizeof(uint32_t), st));
    HIPC(flh::launch_delete_boxes(h->grid, h->map_sorted.p, (uint32_t)h->alloc_top, h->mu_boxes.p, (int)nb, h->dead_id.p, h->live.p,
                                  h->ctr.p, st));
    // bricks that have lost more than half of their range to tombstones are compacted where they lie, so that the searches of
    // a long-running node do not go on reading what lasermap_fov_segment removed (a brick that RECEIVES points is rewritten anyway)
    HIPC(flh::launch_brick_purge(h->grid, h->map_sorted.p, h->starts.p, h->live.p, h->ctr.p, (uint32_t)h->nbricks, (uint32_t)h->pts_cap, st));
    HIPC(hipMemcpyAsync(h->h_ctr, h->ctr.p, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));  // boxes is the caller's
    h->M -= h->h_ctr[3];
    h->n_purged += h->h_ctr[4];
    h->id_pos_valid = false;
    h->searched_once = false;
    return 0;
}

// lasermap_fov_segment() -- src/laserMapping.cpp:230-280; the cube logic is include/fastlio_amd/local_map.hpp
int flh_fov_segment(flh_handle* h, flh_local_map* lm, const double pos_lid[3], double cube_len, float det_range,
                    float* boxes_out, int* n_boxes, int64_t* kdtree_delete_counter) {
    if (!h || !lm || !pos_lid) return fail("flh_fov_segment: null argument");
    fastlio_amd::LocalMap cube;
    cube.cube_len = cube_len;
    cube.DET_RANGE = det_range;
    cube.Localmap_Initialized = lm->initialized != 0;
    for (int a = 0; a < 3; ++a) {
        cube.LocalMap_Points.vertex_min[a] = lm->vertex_min[a];
        cube.LocalMap_Points.vertex_max[a] = lm->vertex_max[a];
    }
    const std::vector<fastlio_amd::BoxPointType> cub_needrm = cube.lasermap_fov_segment(pos_lid);
    lm->initialized = cube.Localmap_Initialized ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
        lm->vertex_min[a] = cube.LocalMap_Points.vertex_min[a];
        lm->vertex_max[a] = cube.LocalMap_Points.vertex_max[a];
    }
    float boxes[18];
    for (size_t b = 0; b < cub_needrm.size(); ++b)
        for (int a = 0; a < 3; ++a) {
            boxes[6 * b + a] = cub_needrm[b].vertex_min[a];
            boxes[6 * b + 3 + a] = cub_needrm[b].vertex_max[a];
        }
    if (boxes_out) std::memcpy(boxes_out, boxes, sizeof(float) * 6 * cub_needrm.size());
    if (n_boxes) *n_boxes = (int)cub_needrm.size();
    if (map_settle(h) != 0) return -1;
    const size_t before = h->M;
    if (!cub_needrm.empty() && flh_map_delete_boxes(h, boxes, cub_needrm.size()) != 0) return -1;
    if (kdtree_delete_counter) *kdtree_delete_counter = (int64_t)(before - h->M);
    return 0;
}

// Bookkeeping of the brick storage: {full re-indexings, changes applied brick-wise, storage slots in use, storage slots,
// ids handed out since the last re-indexing, bricks}.
int flh_map_stats(const flh_handle* h, uint64_t out[6]) {
    if (!h || !out) return fail("flh_map_stats: null argument");
    if (map_settle(const_cast<flh_handle*>(h)) != 0) return -1;
    out[0] = h->n_reindex; out[1] = h->n_inplace; out[2] = h->alloc_top; out[3] = h->pts_cap; out[4] = h->n_ids; out[5] = h->nbricks;
    return 0;
}

int flh_map_storage_stats(flh_handle* h, uint64_t out[4]) {
    if (!h || !out) return fail("flh_map_storage_stats: null argument");
    if (map_settle(h) != 0) return -1;
    out[0] = h->M; out[1] = 0; out[2] = h->n_purged; out[3] = h->nbricks;
    const size_t nb = h->nbricks;
    if (nb == 0 || !h->starts.p) return 0;
    HIPC(hipSetDevice(h->device));
    std::vector<uint32_t> a(nb), b(nb);
    const size_t pitch = (size_t)flh::kBrickStride * sizeof(uint32_t);
    HIPC(hipMemcpy2DAsync(a.data(), sizeof(uint32_t), h->starts.p, pitch, sizeof(uint32_t), nb, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpy2DAsync(b.data(), sizeof(uint32_t), h->starts.p + 64, pitch, sizeof(uint32_t), nb, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    uint64_t span = 0;
    for (size_t r = 0; r < nb; ++r) span += b[r] >= a[r] ? b[r] - a[r] : 0;
    out[1] = span;
    return 0;
}

int flh_map_change_stats(const flh_handle* h, uint64_t out[2]) {
    if (!h || !out) return fail("flh_map_change_stats: null argument");
    out[0] = h->n_mi_deferred; out[1] = h->n_mi_replayed;
    return 0;
}

// The map in index order (what PCL_Storage / flatten would hand back, src/laserMapping.cpp:406-411): the live points by id.
int flh_map_download(flh_handle* h, float* xyz, size_t capacity_points) {
    if (!h) return fail("flh_map_download: null handle");
    if (map_settle(h) != 0) return -1;
    if (capacity_points < h->M) return fail("flh_map_download: buffer too small");
    if (h->M == 0) return 0;
    if (!xyz) return fail("flh_map_download: null buffer");
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const float4* src = h->map_orig.p;
    if (h->n_ids != h->M) {  // compact the live ones first
        const size_t n_ids = h->n_ids;
        HIPC(h->mu_flags.reserve(n_ids)); HIPC(h->mu_incl.reserve(n_ids));
        HIPC(flh::launch_byte_flags(h->dead_id.p, (uint32_t)n_ids, 1, h->mu_flags.p, st));
        size_t tb = 0;
        HIPC(flh::inclusive_sum(nullptr, tb, h->mu_flags.p, h->mu_incl.p, (uint32_t)n_ids, st));
        HIPC(h->mb_tmp.reserve(tb));
        tb = h->mb_tmp.cap;
        HIPC(flh::inclusive_sum(h->mb_tmp.p, tb, h->mu_flags.p, h->mu_incl.p, (uint32_t)n_ids, st));
        HIPC(h->map_next.reserve(h->M));
        HIPC(flh::launch_live_compact(h->map_orig.p, h->mu_flags.p, h->mu_incl.p, (uint32_t)n_ids, h->map_next.p, st));
        src = h->map_next.p;
    }
    std::vector<float4> hp(h->M);
    HIPC(hipMemcpyAsync(hp.data(), src, h->M * sizeof(float4), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    for (size_t i = 0; i < h->M; ++i) { xyz[3 * i] = hp[i].x; xyz[3 * i + 1] = hp[i].y; xyz[3 * i + 2] = hp[i].z; }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// per-scan work buffers sized for N points; resets the per-scan state the reference keeps in globals
static int prepare_scan_buffers(flh_handle* h, size_t N, bool full_clear) {
    hipStream_t st = h->stream;
    const size_t n1 = N ? N : 1;
    HIPC(h->world.reserve(n1)); HIPC(h->nn_pts.reserve(5 * n1)); HIPC(h->normvec.reserve(n1));
    if (h->plane_cache) HIPC(h->plane.reserve(n1));
    if (h->plane_cache && h->cfg.index_cache) HIPC(h->nn_idx.reserve(5 * n1));
    HIPC(h->nn_d2.reserve(5 * n1)); HIPC(h->nn_cnt.reserve(n1)); HIPC(h->selected.reserve(n1));
    {
        const size_t ln = (size_t)flh::list_stripes() * flh::list_stripe_cap((int)N);
        HIPC(h->slow_list.reserve(ln)); HIPC(h->slow_list2.reserve(ln)); HIPC(h->slow_ub.reserve(n1));
    }
    const int nblk = flh::fit_blocks((int)N);
    HIPC(h->partials.reserve(std::max((size_t)nblk * 256, (size_t)flh::pass_blocks((int)N) * kGranSlots)));  // k_fit's blocks / k_pass's workgroups
    const int ngroups = std::max(flh::reduce1_blocks(nblk, nullptr), kGranGroups);
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
        if (h->nn_idx.p) HIPC(hipMemsetAsync(h->nn_idx.p, 0xFF, 5 * n1 * sizeof(uint32_t), st));
        HIPC(hipMemsetAsync(h->nn_d2.p, 0x7F, 5 * n1 * sizeof(float), st));    // large finite; rewritten by search
        HIPC(hipMemsetAsync(h->normvec.p, 0, n1 * sizeof(float4), st));
        HIPC(hipMemsetAsync(h->world.p, 0, n1 * sizeof(float4), st));
    }
    if (h->comm) {  // the all-reduced group totals: rows behind this scan's groups must read zero (k_publish_groups keeps them so)
        HIPC(h->gsum.reserve((size_t)kGranGroups * kGranSlots));
        HIPC(hipMemsetAsync(h->gsum.p, 0, (size_t)kGranGroups * kGranSlots * sizeof(double), st));
    }
    h->N = N;
    for (int& g : h->sect_ng) g = 0;  // the peers' shards change with the scan
    h->have_eval = false;
    h->searched_once = false;
    h->d2_valid = false;
    h->aux_valid = false;
    h->planes_valid = false;
    h->nn_pts_valid = true;  // (no search on this scan yet: nothing to gather)
    h->mi_valid_N = (size_t)-1;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Scan staging.  A scan travels: caller's buffer -> pinned host memory (skipped when the caller's buffer is itself
// pinned, flh_host_alloc) -> one H2D copy of the records as they are -> k_scan_restride (float4 + Morton key of the
// body-frame coordinates) -> radix sort -> gather into the slot.  Everything runs on the copy stream; nothing comes back
// to the host (the permutation and the host copy of feats_down_body are fetched lazily, flh_fetch_* / flh_fetch_rows
// need them, the hot path does not).  flh_scan_stage_async hands the whole sequence to the handle's staging thread so
// that the caller's thread goes straight on with the update of the previous scan.
// ---------------------------------------------------------------------------------------------
static int ensure_pinned(unsigned char*& p, size_t& cap, size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPC(hipHostMalloc((void**)&p, want, hipHostMallocDefault));
    cap = want;
    return 0;
}
// Page-locked buffers the DMA engine may read where they lie: ONLY the ones flh_host_alloc handed out (a registry of [base, end)
// ranges).  The runtime's pointer query is not asked about arbitrary caller memory any more: after a page-locked region has been
// freed, malloc may hand the same addresses out again, and a stale "this is pinned" answer would let the copy engine read
// pageable memory -- a device-side memory access fault at a host address, rarely and only after much host allocation churn
// (the shape of the two faults of rounds 2 and 3).  Everything else takes the slot's own page-locked bounce buffer.
static std::mutex g_pin_mu;
static std::vector<std::pair<uintptr_t, uintptr_t>> g_pin_ranges;
static bool is_pinned_host(const void* p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const auto& r : g_pin_ranges)
        if (a >= r.first && a + bytes <= r.second) return true;
    return false;
}

// A staging lane: a stream and the scratch of the plain staging on it.  Lane 0 is the copy stream every staging path uses; lane 1
// serves the plain staging (flh_scan_stage / _async) of odd slots, so that two scans can be staged side by side.
struct StageLane {
    hipStream_t cs;
    DevBuf<unsigned char>&bytes, &tmp;
    DevBuf<float4>& raw;
    DevBuf<uint32_t>&m0, &m1, &v0, &v1;
};
static StageLane stage_lane(flh_handle* h, int which) {
    if (which) return StageLane{h->copy_stream2, h->st2_bytes, h->st2_tmp, h->st2_raw, h->st2_m0, h->st2_m1, h->st2_v0, h->st2_v1};
    return StageLane{h->copy_stream, h->st_bytes, h->st_tmp, h->st_raw, h->st_m0, h->st_m1, h->st_v0, h->st_v1};
}
static int stage_prepare(flh_handle* h, flh_handle::Slot& sl, int lane = 0) {
    HIPC(hipSetDevice(h->device));
    if (!h->copy_stream) HIPC(create_stream(&h->copy_stream, false));
    if (lane && !h->copy_stream2) HIPC(create_stream(&h->copy_stream2, false));
    if (!sl.ready) HIPC(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
    if (!sl.h2d_done) HIPC(hipEventCreateWithFlags(&sl.h2d_done, hipEventDisableTiming));
    // the slot's buffer was last written on the other lane's stream: this staging's writes come after those
    hipStream_t cs = lane ? h->copy_stream2 : h->copy_stream;
    if (sl.used && sl.last_stream && sl.last_stream != cs) HIPC(hipStreamWaitEvent(cs, sl.ready, 0));
    // a kernel of the handle's stream that nobody waited for (map_incremental without the host's wait) may still be reading the
    // slot's scan: this staging's writes come after it
    if (sl.consumed_set) {
        HIPC(hipStreamWaitEvent(cs, sl.consumed, 0));
        sl.consumed_set = false;
    }
    sl.last_stream = cs;
    sl.host_valid = false;
    return 0;
}

// the library's own staging kernels (flh_stage.hip) take this scan?
// (stage_sort 1, the default: up to 28 tiles of 4 096 = 114 688 points.  The merge's work grows with the SQUARE of the scan: same-box
// pairs against the vendor sort, pipelined: 100 000 points +4.0 % (profiles/r06_call8/), 130 000 points -1.8 % (r06_call10/),
// 200 000 points -5 % (r06_call6/).  2: wherever the kernels can run, 262 144 points)
static bool own_stage_sort(const flh_handle* h, size_t N) {
    return h->cfg.stage_sort != 0 && N <= (h->cfg.stage_sort == 2 ? (size_t)flh::stage_sort_max() : (size_t)28 * 4096);
}

// Device side of the plain staging: st_raw (N float4, original order) + keys/vals are in place on the copy stream.
static int stage_sorted(flh_handle* h, flh_handle::Slot& sl, size_t N, bool have_keys, int lane = 0) {
    const StageLane L = stage_lane(h, lane);
    hipStream_t cs = L.cs;
    const size_t n1 = N ? N : 1;
    HIPC(sl.body.reserve(n1));
    const bool do_sort = h->cfg.sort_queries != 0 && N > 1;
    if (do_sort && !have_keys && own_stage_sort(h, N)) {  // float4 records in L.raw: the library's own two kernels
        HIPC(L.tmp.reserve((size_t)flh::stage_scratch_words((uint32_t)N) * 4u));
        HIPC(flh::launch_stage_sort(L.raw.p, 16u, (uint32_t)N, 0.5f, (uint32_t*)L.tmp.p, sl.body.p, cs));
    } else if (do_sort) {
        const uint32_t Nu = (uint32_t)N;
        HIPC(L.m0.reserve(N)); HIPC(L.m1.reserve(N)); HIPC(L.v0.reserve(N)); HIPC(L.v1.reserve(N));
        if (!have_keys) HIPC(flh::launch_scan_keys(L.raw.p, Nu, 0.5f, L.m0.p, L.v0.p, cs));
        size_t tb = 0;
        HIPC(flh::sort_scan_pairs(nullptr, tb, L.m0.p, L.m1.p, L.v0.p, L.v1.p, Nu, cs));
        HIPC(L.tmp.reserve(tb));
        tb = L.tmp.cap;
        HIPC(flh::sort_scan_pairs(L.tmp.p, tb, L.m0.p, L.m1.p, L.v0.p, L.v1.p, Nu, cs));
        HIPC(flh::launch_scan_gather(L.raw.p, L.v1.p, Nu, sl.body.p, cs));
    } else {
        HIPC(flh::launch_scan_gather(L.raw.p, nullptr, (uint32_t)N, sl.body.p, cs));
    }
    HIPC(hipEventRecord(sl.ready, cs));
    sl.N = N;
    sl.used = true;
    return 0;
}

// Copies a scan to the device (copy stream).  wait_reusable: return only when the caller's buffer may be overwritten.
static int stage_into(flh_handle* h, flh_handle::Slot& sl, const void* pts, size_t stride_bytes, size_t N, bool wait_reusable) {
    if (N > 0 && !pts) return fail("scan staging: null points");
    if ((stride_bytes < 12 || (stride_bytes & 3)) && N > 0) return fail("scan staging: stride_bytes must be a multiple of 4 and >= 12");
    if (N >= (1ull << 26)) return fail("scan staging: N too large");
    const int lane = (int)((&sl - h->slots) & 1);  // odd slots: the second lane (the ring's scans alternate)
    const auto t_job = std::chrono::steady_clock::now();
    if (stage_prepare(h, sl, lane) != 0) return -1;
    const StageLane L = stage_lane(h, lane);
    hipStream_t cs = L.cs;
    const size_t n1 = N ? N : 1;
    const size_t bytes = N * stride_bytes;
    HIPC(L.raw.reserve(n1));
    HIPC(L.bytes.reserve(bytes ? bytes : 4));
    bool direct = false;
    if (N > 0) {
        direct = is_pinned_host(pts, bytes);
        const void* src = pts;
        if (!direct) {  // pageable memory cannot be DMA'd: through the slot's pinned buffer (the caller's is free at once)
            if (ensure_pinned(sl.pin, sl.pin_cap, bytes) != 0) return -1;
            std::memcpy(sl.pin, pts, bytes);
            src = sl.pin;
        }
        HIPC(hipMemcpyAsync(L.bytes.p, src, bytes, hipMemcpyHostToDevice, cs));
        HIPC(hipEventRecord(sl.h2d_done, cs));
    }
    const bool do_sort = h->cfg.sort_queries != 0 && N > 1;
    if (do_sort) { HIPC(L.m0.reserve(N)); HIPC(L.v0.reserve(N)); }
    if (do_sort && own_stage_sort(h, N)) {
        // two launches straight from the records as they crossed PCIe: key + tile sort in LDS, then merge by rank + gather
        HIPC(sl.body.reserve(n1));
        HIPC(L.tmp.reserve((size_t)flh::stage_scratch_words((uint32_t)N) * 4u));
        HIPC(flh::launch_stage_sort(L.bytes.p, (uint32_t)stride_bytes, (uint32_t)N, 0.5f, (uint32_t*)L.tmp.p, sl.body.p, cs));
        HIPC(hipEventRecord(sl.ready, cs));
        sl.N = N;
        sl.used = true;
    } else {
        HIPC(flh::launch_scan_restride(L.bytes.p, (uint32_t)stride_bytes, 0, 0, (uint32_t)N, 0.5f, L.raw.p,
                                       do_sort ? L.m0.p : nullptr, do_sort ? L.v0.p : nullptr, nullptr, cs));
        if (stage_sorted(h, sl, N, do_sort, lane) != 0) return -1;
    }
    const auto t_enq = std::chrono::steady_clock::now();
    if (direct && wait_reusable && N > 0) HIPC(hipEventSynchronize(sl.h2d_done));
    {
        const double e = std::chrono::duration<double, std::micro>(t_enq - t_job).count();
        const double w = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq).count();
        flh_handle::StageDiag& d = h->sdiag;
        d.n_jobs += 1; d.enq_us += e; d.h2d_wait_us += w;
        if (e > d.enq_max_us) d.enq_max_us = e;
        if (w > d.h2d_wait_max_us) d.h2d_wait_max_us = w;
    }
    return 0;
}

// The raw-scan front end on the device: optional undistortion (UndistortPcl's per-point half, IMU_Processing.hpp:307-349),
// optional voxel-grid down-sampling (downSizeFilterSurf.filter, src/laserMapping.cpp:904-905), then staging.  The cloud the
// down-sampling starts from (feats_undistort) stays in the slot for flh_frame_world.
struct UndistortArgs {
    const flh_pose6d* poses = nullptr;
    int n_pose = 0;
    const double* x_end = nullptr;
    size_t time_offset_bytes = 0;
    float* undistorted_out = nullptr;
};
static int read_aabb(flh_handle* h, const float4* pts, uint32_t n, hipStream_t cs, float mn[3], float mx[3], uint32_t* bad_out) {
    HIPC(h->mb_aabb.reserve(8));
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    HIPC(hipMemcpyAsync(h->mb_aabb.p, init, sizeof(init), hipMemcpyHostToDevice, cs));
    HIPC(flh::launch_aabb(pts, n, h->mb_aabb.p, cs));
    HIPC(hipMemcpyAsync(h->h_small, h->mb_aabb.p, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    HIPC(hipStreamSynchronize(cs));
    for (int d = 0; d < 6; ++d) {
        const uint32_t u = h->h_small[d];
        const uint32_t bits = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        float f;
        std::memcpy(&f, &bits, 4);
        (d < 3 ? mn[d] : mx[d - 3]) = f;
    }
    if (bad_out) *bad_out = h->h_small[6];
    return 0;
}
static int stage_raw(flh_handle* h, flh_handle::Slot& sl, const char* who, const void* pts, size_t stride_bytes, size_t n,
                     const UndistortArgs* und, float leaf, size_t* n_out) {
    const std::string w(who);
    if (n > 0 && !pts) return fail(w + ": null points");
    if ((stride_bytes < 12 || (stride_bytes & 3)) && n > 0) return fail(w + ": stride_bytes must be a multiple of 4 and >= 12");
    if (n >= (1ull << 26)) return fail(w + ": n too large");
    if (und) {
        if (!und->poses || und->n_pose < 1 || !und->x_end) return fail(w + ": IMU poses / end state missing");
        if (n > 0 && (und->time_offset_bytes + 4 > stride_bytes || (und->time_offset_bytes & 3)))
            return fail(w + ": time_offset_bytes outside the point record (or not a multiple of 4)");
        for (int k = 0; k < und->n_pose; ++k)
            if (!std::isfinite(und->poses[k].offset_time)) return fail(w + ": non-finite IMU pose offset_time");
    }
    if (stage_prepare(h, sl) != 0) return -1;
    hipStream_t cs = h->copy_stream;
    const size_t n1 = n ? n : 1;
    HIPC(h->st_raw.reserve(n1));
    HIPC(sl.dense.reserve(n1));
    sl.n_dense = 0;
    size_t m = 0;
    if (n > 0) {
        const uint32_t nu = (uint32_t)n;
        const size_t bytes = n * stride_bytes;
        HIPC(h->st_bytes.reserve(bytes));
        const void* src = pts;
        const bool direct = is_pinned_host(pts, bytes);
        if (!direct) {
            if (ensure_pinned(sl.pin, sl.pin_cap, bytes) != 0) return -1;
            std::memcpy(sl.pin, pts, bytes);
            src = sl.pin;
        }
        HIPC(hipMemcpyAsync(h->st_bytes.p, src, bytes, hipMemcpyHostToDevice, cs));
        HIPC(h->mb_aabb.reserve(8));
        HIPC(hipMemsetAsync(h->mb_aabb.p + 6, 0, sizeof(uint32_t), cs));  // non-finite counter
        // records -> float4 (x, y, z, time offset); with undistortion into scratch, else straight into the slot's dense cloud
        if (und) HIPC(h->ds_raw.reserve(n1));
        float4* first = und ? h->ds_raw.p : sl.dense.p;
        HIPC(flh::launch_scan_restride(h->st_bytes.p, (uint32_t)stride_bytes, und ? (uint32_t)und->time_offset_bytes : 0u, und ? 1 : 0,
                                       nu, 0.25f, first, nullptr, nullptr, h->mb_aabb.p + 6, cs));
        if (und) {
            HIPC(h->ds_poses.reserve((size_t)22 * und->n_pose));
            static_assert(sizeof(flh_pose6d) == 22 * sizeof(double), "flh_pose6d must be 22 packed doubles");
            HIPC(hipMemcpyAsync(h->ds_poses.p, und->poses, sizeof(flh_pose6d) * und->n_pose, hipMemcpyHostToDevice, cs));
            const StateDev se = make_state(und->x_end + 3, und->x_end + 0, und->x_end + 7, und->x_end + 11);
            u64* bmin = nullptr;
            if (h->cfg.undistort_first_point) {  // the reference's repeated compensation of the earliest point (IMU_Processing.hpp:345)
                HIPC(h->ds_blockmin.reserve(flh::undistort_blocks(nu)));
                bmin = h->ds_blockmin.p;
            }
            HIPC(flh::launch_undistort(se, h->ds_poses.p, und->n_pose, h->ds_raw.p, nu, sl.dense.p, bmin, cs));
        }
        sl.n_dense = n;
        const float4* srcd = sl.dense.p;
        bool passthrough = !(leaf > 0.f);
        uint32_t bad = 0;
        if (!passthrough) {
            float mn[3], mx[3];  // getMinMax3D on the device (the points may just have been moved by the undistortion)
            if (read_aabb(h, srcd, nu, cs, mn, mx, &bad) != 0) return -1;
            if (bad) return fail(w + ": " + std::to_string(bad) + " non-finite point record(s)");
            const float inv = 1.0f / leaf;  // inverse_leaf_size_
            long long dxyz[3];
            for (int d = 0; d < 3; ++d) dxyz[d] = (long long)((mx[d] - mn[d]) * inv) + 1;
            if (dxyz[0] * dxyz[1] * dxyz[2] > (long long)INT32_MAX) {
                passthrough = true;  // "Leaf size is too small for the input dataset": PCL returns the input unchanged
            } else {
                int min_b[3], div_b[3];
                for (int d = 0; d < 3; ++d) {
                    min_b[d] = (int)std::floor(mn[d] * inv);
                    div_b[d] = (int)std::floor(mx[d] * inv) - min_b[d] + 1;
                }
                HIPC(h->st_k0.reserve(n)); HIPC(h->st_k1.reserve(n)); HIPC(h->st_v0.reserve(n)); HIPC(h->st_v1.reserve(n));
                HIPC(h->ds_flags.reserve(n)); HIPC(h->ds_incl.reserve(n));
                HIPC(flh::launch_vg_keys(srcd, nu, inv, min_b, div_b[0], div_b[0] * div_b[1], h->st_k0.p, h->st_v0.p, cs));
                size_t tb1 = 0, tb2 = 0;
                HIPC(flh::sort_vg_pairs(nullptr, tb1, h->st_k0.p, h->st_k1.p, h->st_v0.p, h->st_v1.p, nu, cs));
                HIPC(flh::inclusive_sum(nullptr, tb2, h->ds_flags.p, h->ds_incl.p, nu, cs));
                HIPC(h->st_tmp.reserve(std::max(tb1, tb2)));
                size_t tb = h->st_tmp.cap;
                HIPC(flh::sort_vg_pairs(h->st_tmp.p, tb, h->st_k0.p, h->st_k1.p, h->st_v0.p, h->st_v1.p, nu, cs));
                HIPC(flh::launch_vg_heads(h->st_k1.p, nu, h->ds_flags.p, cs));
                tb = h->st_tmp.cap;
                HIPC(flh::inclusive_sum(h->st_tmp.p, tb, h->ds_flags.p, h->ds_incl.p, nu, cs));
                HIPC(hipMemcpyAsync(h->h_small, h->ds_incl.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
                HIPC(flh::launch_vg_reduce(srcd, h->st_k1.p, h->st_v1.p, h->ds_flags.p, h->ds_incl.p, nu, h->st_raw.p, cs));
                HIPC(hipStreamSynchronize(cs));
                m = h->h_small[0];
            }
        }
        if (passthrough) {
            HIPC(hipMemcpyAsync(h->st_raw.p, srcd, n * sizeof(float4), hipMemcpyDeviceToDevice, cs));
            HIPC(hipMemcpyAsync(h->h_small, h->mb_aabb.p + 6, sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
            HIPC(hipStreamSynchronize(cs));
            if (h->h_small[0]) return fail(w + ": " + std::to_string(h->h_small[0]) + " non-finite point record(s)");
            m = n;
        }
        if (und && und->undistorted_out) {  // feats_undistort for the caller: only when asked for
            if (ensure_pinned(h->pin_out, h->pin_out_cap, n * sizeof(float4)) != 0) return -1;
            HIPC(hipMemcpyAsync(h->pin_out, sl.dense.p, n * sizeof(float4), hipMemcpyDeviceToHost, cs));
            HIPC(hipStreamSynchronize(cs));
            const float4* uo = (const float4*)h->pin_out;
            for (size_t i = 0; i < n; ++i) {
                und->undistorted_out[3 * i] = uo[i].x; und->undistorted_out[3 * i + 1] = uo[i].y; und->undistorted_out[3 * i + 2] = uo[i].z;
            }
        }
    }
    if (n_out) *n_out = m;
    return stage_sorted(h, sl, m, false);
}

// The slot's scan on the host (feats_down_body in the order it was staged + the permutation of the device order): one D2H
// of the Morton-ordered cloud, whose .w carries each point's original index.  Only the lazy fetches need it.
static int ensure_host_copy(flh_handle* h, flh_handle::Slot& sl) {
    if (sl.host_valid) return 0;
    const size_t N = sl.N;
    sl.h_perm.assign(N ? N : 1, 0u);
    sl.h_body.assign(3 * (N ? N : 1), 0.f);
    if (N > 0) {
        HIPC(hipSetDevice(h->device));
        if (ensure_pinned(h->pin_out, h->pin_out_cap, N * sizeof(float4)) != 0) return -1;
        HIPC(hipEventSynchronize(sl.ready));
        HIPC(hipMemcpy(h->pin_out, sl.body.p, N * sizeof(float4), hipMemcpyDeviceToHost));
        const float4* b = (const float4*)h->pin_out;
        for (size_t i = 0; i < N; ++i) {
            uint32_t o;
            std::memcpy(&o, &b[i].w, 4);
            if (o >= N) return fail("scan slot: corrupt permutation");
            sl.h_perm[i] = o;
            sl.h_body[3 * (size_t)o] = b[i].x; sl.h_body[3 * (size_t)o + 1] = b[i].y; sl.h_body[3 * (size_t)o + 2] = b[i].z;
        }
    }
    sl.host_valid = true;
    return 0;
}

// ---- the staging thread (flh_scan_stage_async) ----
static void stager_main(flh_handle* h) {
    (void)hipSetDevice(h->device);
    std::unique_lock<std::mutex> lk(h->st_mu);
    for (;;) {
        // In a running stream the next job arrives within a scan's time: poll for it for a while before sleeping, so that
        // flh_scan_stage_async's notify finds nobody to wake -- no futex call on the thread that is about to launch a scan's
        // first pass (measured without a device, tests/cpp/stager_check.cpp: the hand-over costs its caller 0.5 us instead
        // of 4-6).  A stream slower than the window pays the wake-up as before.
        if (h->st_queue.empty() && !h->st_quit) {
            const uint64_t seen = h->st_posted.load(std::memory_order_acquire);
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            while (h->st_posted.load(std::memory_order_acquire) == seen) {
                const auto waited = std::chrono::steady_clock::now() - t0;
                if (waited >= std::chrono::microseconds(kStagerPollUs)) break;
                if (waited < std::chrono::microseconds(50)) cpu_relax();
                else std::this_thread::yield();  // (gives the core away when somebody else wants it, returns at once when nobody does)
            }
            lk.lock();
        }
        h->st_cv.wait(lk, [&] { return h->st_quit || !h->st_queue.empty(); });
        if (h->st_queue.empty()) return;  // quit
        flh_handle::StageJob job = h->st_queue.front();
        h->st_queue.pop_front();
        lk.unlock();
        flh_handle::Slot& sl = h->slots[job.slot];
        g_err.clear();
        int rc;
        {
            std::lock_guard<std::mutex> sg(h->stage_mu);
            rc = stage_into(h, sl, job.pts, job.stride, job.N, true);
        }
        lk.lock();
        sl.async_rc = rc;
        sl.async_err = rc ? g_err : std::string();
        sl.pending = false;
        h->st_done.notify_all();
    }
}
static int wait_slot(flh_handle* h, flh_handle::Slot& sl, bool* was_pending = nullptr) {
    if (was_pending) *was_pending = false;
    if (!h->stager.joinable()) return 0;
    std::unique_lock<std::mutex> lk(h->st_mu);
    if (was_pending) *was_pending = sl.pending;
    h->st_done.wait(lk, [&] { return !sl.pending; });
    if (sl.async_rc != 0) {
        const std::string e = sl.async_err;
        sl.async_rc = 0;
        return fail("flh_scan_stage_async: " + e);
    }
    return 0;
}
static void stop_stager(flh_handle* h) {
    if (!h->stager.joinable()) return;
    {
        std::lock_guard<std::mutex> lk(h->st_mu);
        h->st_quit = true;
    }
    h->st_cv.notify_all();
    h->stager.join();
}

static int activate(flh_handle* h, flh_handle::Slot& sl, bool full_clear) {
    HIPC(hipSetDevice(h->device));
    pre_cancel(h);  // (a pass enqueued ahead for the previous scan that never got its state)
    if (wait_slot(h, sl) != 0) return -1;
    // In a running stream the staging of this scan finished while the previous scan was updated: then no barrier packet goes in
    // front of the scan's first pass, one look at the event instead (same box, two alternating pairs, profiles/r05_call1/:
    // 7 431 / 7 399 -> 7 496 / 7 476 scans/s)
    if (hipEventQuery(sl.ready) != hipSuccess) {
        h->sdiag.act_ev_not_ready += 1;
        (void)hipGetLastError();  // (not ready is not an error)
        HIPC(hipStreamWaitEvent(h->stream, sl.ready, 0));
    }
    if (prepare_scan_buffers(h, sl.N, full_clear) != 0) return -1;
    h->cur_body = sl.body.p;
    h->cur = &sl;
    return 0;
}

int flh_scan_upload(flh_handle* h, const void* pts, size_t stride_bytes, size_t N) {
    if (!h) return fail("flh_scan_upload: null handle");
    flh_handle::Slot& sl = h->slots[FLH_MAX_SLOTS];
    {
        std::lock_guard<std::mutex> sg(h->stage_mu);  // a staging the worker thread has under way finishes first (shared scratch)
        if (stage_into(h, sl, pts, stride_bytes, N, true) != 0) return -1;
    }
    if (activate(h, sl, true) != 0) return -1;
    HIPC(hipStreamSynchronize(h->stream));
    return 0;
}

int flh_scan_stage(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t N) {
    if (!h) return fail("flh_scan_stage: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_stage: bad slot");
    if (wait_slot(h, h->slots[slot]) != 0) return -1;
    if (h->cur == &h->slots[slot]) {  // re-staging the active scan's slot ends that scan
        h->cur = nullptr;
        h->cur_body = nullptr;
        h->have_eval = false;
    }
    std::lock_guard<std::mutex> sg(h->stage_mu);
    return stage_into(h, h->slots[slot], pts, stride_bytes, N, true);
}

int flh_scan_stage_async(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t N) {
    if (!h) return fail("flh_scan_stage_async: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_stage_async: bad slot");
    if (N > 0 && !pts) return fail("flh_scan_stage_async: null points");
    if (h->cur == &h->slots[slot]) {  // the active scan's slot is being re-used: that scan is over (no evaluation is in flight)
        h->cur = nullptr;
        h->cur_body = nullptr;
        h->have_eval = false;
    }
    std::lock_guard<std::mutex> lk(h->st_mu);
    if (h->slots[slot].pending) return fail("flh_scan_stage_async: the slot is still being staged");
    if (!h->stager.joinable()) h->stager = std::thread(stager_main, h);
    h->slots[slot].pending = true;
    h->st_queue.push_back({slot, pts, stride_bytes, N});
    h->st_posted.fetch_add(1, std::memory_order_release);
    h->st_cv.notify_one();
    return 0;
}

int flh_scan_wait(flh_handle* h, int slot) {
    if (!h) return fail("flh_scan_wait: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_wait: bad slot");
    return wait_slot(h, h->slots[slot]);
}

void* flh_host_alloc(size_t bytes) {
    void* p = nullptr;
    const size_t n = bytes ? bytes : 1;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) {
        (void)fail("flh_host_alloc: hipHostMalloc failed");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_ranges.emplace_back((uintptr_t)p, (uintptr_t)p + n);
    return p;
}
void flh_host_free(void* p) {
    if (!p) return;
    // The range leaves the registry FIRST: a staging job that only now reaches the front of its queue takes the bounce buffer instead
    // of a direct DMA from memory that is about to go.  Then every device THIS LIBRARY has a handle on in this process is
    // waited for -- a staging that reads the buffer where it lies may still be under way (flh_scan_stage_async), and the buffer is
    // not tied to a handle -- but no other device: in a one-rank-per-GPU process that would create contexts on the other ranks' GPUs.
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pin_ranges.size(); ++i)
            if (g_pin_ranges[i].first == (uintptr_t)p) { g_pin_ranges.erase(g_pin_ranges.begin() + (long)i); break; }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        devs.assign(g_devices_used.begin(), g_devices_used.end());
    }
    for (int d : devs)
        if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
    (void)hipSetDevice(cur);
    (void)hipGetLastError();
    (void)hipHostFree(p);
}

// downSizeFilterSurf.setInputCloud(feats_undistort); downSizeFilterSurf.filter(*feats_down_body) -- :904-905
int flh_scan_stage_downsampled(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t n, float leaf_size,
                               size_t* n_out) {
    if (!h) return fail("flh_scan_stage_downsampled: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_stage_downsampled: bad slot");
    if (!(leaf_size > 0.f)) return fail("flh_scan_stage_downsampled: leaf size must be > 0");
    if (wait_slot(h, h->slots[slot]) != 0) return -1;
    std::lock_guard<std::mutex> sg(h->stage_mu);
    return stage_raw(h, h->slots[slot], "flh_scan_stage_downsampled", pts, stride_bytes, n, nullptr, leaf_size, n_out);
}

// ImuProcess::UndistortPcl's per-point half (IMU_Processing.hpp:307-349), then :904-905, then staging
int flh_scan_stage_undistorted(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t time_offset_bytes, size_t n,
                               const flh_pose6d* imu_pose, int n_pose, const double x_end[FLH_NSTATE], float leaf_size,
                               float* undistorted_xyz, size_t* n_out) {
    if (!h) return fail("flh_scan_stage_undistorted: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_stage_undistorted: bad slot");
    if (wait_slot(h, h->slots[slot]) != 0) return -1;
    UndistortArgs u;
    u.poses = imu_pose;
    u.n_pose = n_pose;
    u.x_end = x_end;
    u.time_offset_bytes = time_offset_bytes;
    u.undistorted_out = undistorted_xyz;
    std::lock_guard<std::mutex> sg(h->stage_mu);
    return stage_raw(h, h->slots[slot], "flh_scan_stage_undistorted", pts, stride_bytes, n, &u, leaf_size, n_out);
}

// feats_down_body of the ACTIVE scan, original (staging) order
int flh_fetch_scan(flh_handle* h, float* xyz) {
    if (!h) return fail("flh_fetch_scan: null handle");
    if (!h->cur) return fail("flh_fetch_scan: no active scan");
    if (h->N > 0 && !xyz) return fail("flh_fetch_scan: null buffer");
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
    if (h->N > 0) std::memcpy(xyz, h->cur->h_body.data(), sizeof(float) * 3 * h->N);
    return 0;
}

int flh_debug_scan_order(flh_handle* h, uint32_t* order) {
    if (!h) return fail("flh_debug_scan_order: null handle");
    if (!h->cur) return fail("flh_debug_scan_order: no active scan");
    if (h->N > 0 && !order) return fail("flh_debug_scan_order: null buffer");
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
    if (h->N > 0) std::memcpy(order, h->cur->h_perm.data(), sizeof(uint32_t) * h->N);
    return 0;
}

// SURVEY.md 8(f) row 4 -- publish_frame_world (src/laserMapping.cpp:478-530): RGBpointBodyToWorld (:200-211) over
// feats_undistort (dense != 0: the cloud `slot` was down-sampled from, still on the device) or feats_down_body (dense == 0).
int flh_frame_world(flh_handle* h, int slot, const double x[FLH_NSTATE], int dense, float* world_xyz, size_t capacity_points,
                    size_t* n_points) {
    if (!h || !x) return fail("flh_frame_world: null argument");
    flh_handle::Slot* sl = nullptr;
    if (slot < 0) sl = h->cur;
    else if (slot <= FLH_MAX_SLOTS) sl = &h->slots[slot];
    if (!sl || !sl->used) return fail("flh_frame_world: slot not staged");
    if (wait_slot(h, *sl) != 0) return -1;
    if (dense && sl->n_dense == 0 && sl->N > 0)
        return fail("flh_frame_world: the slot was not staged from a raw scan (no feats_undistort on the device)");
    HIPC(hipSetDevice(h->device));
    const size_t n = dense ? sl->n_dense : sl->N;
    if (n_points) *n_points = n;
    if (n == 0 || (!world_xyz && capacity_points == 0)) return 0;  // size query
    if (!world_xyz) return fail("flh_frame_world: null buffer");
    if (capacity_points < n) return fail("flh_frame_world: buffer too small");
    hipStream_t st = h->stream;
    HIPC(hipStreamWaitEvent(st, sl->ready, 0));
    HIPC(h->fw_out.reserve(n));
    if (ensure_pinned(h->pin_out, h->pin_out_cap, n * sizeof(float4)) != 0) return -1;
    const StateDev s = make_state(x + 3, x + 0, x + 7, x + 11);
    HIPC(flh::launch_cloud_body_to_world(s, dense ? sl->dense.p : sl->body.p, (uint32_t)n, h->fw_out.p, st));
    HIPC(hipMemcpyAsync(h->pin_out, h->fw_out.p, n * sizeof(float4), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    const float4* o = (const float4*)h->pin_out;
    if (dense) {
        for (size_t i = 0; i < n; ++i) { world_xyz[3 * i] = o[i].x; world_xyz[3 * i + 1] = o[i].y; world_xyz[3 * i + 2] = o[i].z; }
    } else {  // the slot's cloud is Morton-ordered; .w = original index
        for (size_t i = 0; i < n; ++i) {
            uint32_t k;
            std::memcpy(&k, &o[i].w, 4);
            if (k >= n) return fail("flh_frame_world: corrupt permutation");
            world_xyz[3 * (size_t)k] = o[i].x; world_xyz[3 * (size_t)k + 1] = o[i].y; world_xyz[3 * (size_t)k + 2] = o[i].z;
        }
    }
    return 0;
}

// RGBpointBodyToWorld over any host cloud (e.g. pcl_wait_save's input, :505-512)
int flh_points_body_to_world(flh_handle* h, const double x[FLH_NSTATE], const void* pts, size_t stride_bytes, size_t n,
                             float* world_xyz) {
    if (!h || !x) return fail("flh_points_body_to_world: null argument");
    if (n == 0) return 0;
    if (!pts || !world_xyz) return fail("flh_points_body_to_world: null buffer");
    if (stride_bytes < 12 || (stride_bytes & 3)) return fail("flh_points_body_to_world: stride_bytes must be a multiple of 4 and >= 12");
    if (n >= (1ull << 28)) return fail("flh_points_body_to_world: n too large");
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const size_t bytes = n * stride_bytes;
    HIPC(h->fw_bytes.reserve(bytes)); HIPC(h->fw_in.reserve(n)); HIPC(h->fw_out.reserve(n));
    const void* src = pts;
    if (!is_pinned_host(pts, bytes)) {
        if (ensure_pinned(h->pin_in, h->pin_in_cap, bytes) != 0) return -1;
        std::memcpy(h->pin_in, pts, bytes);
        src = h->pin_in;
    }
    if (ensure_pinned(h->pin_out, h->pin_out_cap, n * sizeof(float4)) != 0) return -1;
    HIPC(hipMemcpyAsync(h->fw_bytes.p, src, bytes, hipMemcpyHostToDevice, st));
    HIPC(flh::launch_scan_restride(h->fw_bytes.p, (uint32_t)stride_bytes, 0, 0, (uint32_t)n, 0.25f, h->fw_in.p, nullptr, nullptr, nullptr, st));
    const StateDev s = make_state(x + 3, x + 0, x + 7, x + 11);
    HIPC(flh::launch_cloud_body_to_world(s, h->fw_in.p, (uint32_t)n, h->fw_out.p, st));
    HIPC(hipMemcpyAsync(h->pin_out, h->fw_out.p, n * sizeof(float4), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    const float4* o = (const float4*)h->pin_out;
    for (size_t i = 0; i < n; ++i) { world_xyz[3 * i] = o[i].x; world_xyz[3 * i + 1] = o[i].y; world_xyz[3 * i + 2] = o[i].z; }
    return 0;
}

int flh_scan_activate(flh_handle* h, int slot) {
    if (!h) return fail("flh_scan_activate: null handle");
    if (slot < 0 || slot >= FLH_MAX_SLOTS) return fail("flh_scan_activate: bad slot");
    const auto t_act = std::chrono::steady_clock::now();
    bool was_pending = false;
    if (wait_slot(h, h->slots[slot], &was_pending) != 0) return -1;  // an asynchronous staging of the slot finishes first
    {  // (developer counters: how long the update waited for the staging thread)
        const double w = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_act).count();
        flh_handle::StageDiag& d = h->sdiag;
        d.n_act += 1; d.act_wait_us += w;
        if (w > d.act_wait_max_us) d.act_wait_max_us = w;
        if (was_pending) d.act_slept += 1;
    }
    if (!h->slots[slot].used) return fail("flh_scan_activate: slot not staged");
    return activate(h, h->slots[slot], false);
}

static StateDev make_state(const double rot[4], const double pos[3], const double offR[4], const double offT[3]) {
    StateDev s;
    for (int i = 0; i < 4; ++i) { s.rot[i] = rot[i]; s.offR[i] = offR[i]; }
    for (int i = 0; i < 3; ++i) { s.pos[i] = pos[i]; s.offT[i] = offT[i]; }
    return s;
}

// Units (64 scan points each: a workgroup of k_pass, a wave of k_fit) per reduction group when the group sums go to the host as
// granules: 64, more when that would make more than kGranGroups groups; 0 = too many points for the granule path.
static int gran_group_size(size_t N) {
    const int red = flh::pass_group_size((int)N, kGranGroups);
    return red <= 1024 ? red : 0;
}

// Deferred event timing (see flh_handle::evp).
static bool ensure_event_pool(flh_handle* h) {
    if (h->evp_ready) return true;
    for (int k = 0; k < flh_handle::kEvPool; ++k)
        for (int j = 0; j < 4; ++j)
            if (hipEventCreate(&h->evp[k][j]) != hipSuccess) {
                (void)hipGetLastError();
                for (int k2 = 0; k2 <= k; ++k2)
                    for (int j2 = 0; j2 < 4; ++j2)
                        if (h->evp[k2][j2]) { (void)hipEventDestroy(h->evp[k2][j2]); h->evp[k2][j2] = nullptr; }
                return false;  // the caller falls back to the synchronous reading
            }
    h->evp_ready = true;
    return true;
}
static void drain_events(flh_handle* h) {
    if (h->evp_n == 0) return;
    (void)hipSetDevice(h->device);
    for (int k = 0; k < h->evp_n; ++k) {
        float a = 0, b = 0, c = 0;
        const int kind = h->evp_search[k] & 3;        // 0: no search, 1: a scan's first search, 2: a later one
        const bool one_launch = (h->evp_search[k] & 4) != 0;  // k_pass: one kernel, stamps in [0] and [3]
        const bool srch = kind != 0;
        hipError_t e = hipEventSynchronize(h->evp[k][3]);
        if (one_launch) {
            if (e == hipSuccess) e = hipEventElapsedTime(&c, h->evp[k][0], h->evp[k][3]);
            a = c;  // the search is not a kernel of its own: the pass kernel's time is reported as both
        } else {
            if (e == hipSuccess && srch) e = hipEventElapsedTime(&a, h->evp[k][0], h->evp[k][1]);
            if (e == hipSuccess) e = hipEventElapsedTime(&b, h->evp[k][2], h->evp[k][3]);
            if (e == hipSuccess) e = hipEventElapsedTime(&c, h->evp[k][srch ? 0 : 2], h->evp[k][3]);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            continue;  // a sample that cannot be read is dropped, not guessed
        }
        if (srch) { h->acc[0] += a; h->acc[1] += 1; h->acc_kind[kind == 2 ? 2 : 0] += a; h->acc_kind[kind == 2 ? 3 : 1] += 1; }
        if (!one_launch) { h->acc[2] += b; h->acc[3] += 1; }
        h->acc[4] += c; h->acc[5] += 1;
        h->timing.search_ms = srch ? a : 0.f;
        h->timing.fit_ms = b;
        h->timing.total_ms = c;
    }
    h->evp_n = 0;
}

// The granule buffers an evaluation with sequence number seq publishes to (parity: see flh_handle::h_gran).
static flh::GranOut gran_out(const flh_handle* h, double seq) {
    flh::GranOut o{};
    const size_t par = ((uint64_t)seq & 1u) * (size_t)h->peer_n * kGranSect * 2;
    for (int d = 0; d < h->peer_n; ++d) o.dst[d] = h->gran_dst[d] + par;
    o.n_dst = h->peer_n;
    o.sect_off = (int)((size_t)h->peer_rank * kGranSect);
    return o;
}
// With an RCCL communicator the pass's sums stay in device memory, but in the granules' tree: the group reducers leave their totals
// in gsum[group][slot], RCCL adds the ranks' totals in place, and the publish kernel adds the groups in the host's order
// (k_publish_groups) -- so the one-launch pass runs there too, both kinds of pass keep one tree, and one rank reproduces flh_eval's
// bits.  Every rank of the communicator must take this path or none (the collective's size differs): an EMPTY shard takes it
// (its totals are zeros), a scan beyond the granule limit (1.6 M points per rank) does not on any rank.
static bool device_tree(const flh_handle* h, bool host_granules) {
    return !host_granules && h->comm != nullptr && h->gsum.p != nullptr && (h->N == 0 || gran_group_size(h->N) > 0);
}
// does a searching evaluation of the active scan run as ONE launch?
static bool use_pass_kernel(const flh_handle* h, bool host_granules) {
    return h->pass_ok && (host_granules || device_tree(h, host_granules)) && h->N > 0;
}

// The coordinates of the current neighbour cache, for whoever reads nn_pts: gathered from the indices a one-launch pass left
// (flh_config.index_cache), once, on the handle's stream.  Ids are stable between re-indexings; rebuild_index calls this first.
static int ensure_nn_pts(flh_handle* h) {
    if (h->nn_pts_valid || h->N == 0 || !h->nn_idx.p) { h->nn_pts_valid = true; return 0; }
    HIPC(hipSetDevice(h->device));
    HIPC(flh::launch_nn_gather(h->map_orig.p, (uint32_t)h->n_ids, h->nn_idx.p, (int)h->N, h->nn_pts.p, h->stream));
    h->nn_pts_valid = true;
    return 0;
}

static int enqueue_eval(flh_handle* h, const StateDev& s, int do_search, int ext, double* d_out, double seq, hipEvent_t* ev3,
                        bool host_granules = false, bool rccl = false, bool behind_map_change = false) {
    const bool timed = ev3 != nullptr;  // four time stamps: first search kernel's start, last one's end, fit kernel's start and end
    hipStream_t st = h->stream;
    // a map change under way: its counters (and a re-index it asked for) first -- unless the caller settles it itself while the pass
    // waits behind the change on the stream (flh_eval_begin / flh_eval_end)
    if (h->map_pending && !behind_map_change && map_settle(h) != 0) return -1;
    if (!h->cur_body || !h->selected.p) return fail("flh_eval: no active scan (flh_scan_upload / flh_scan_activate first)");
    if (!h->grid.hash && h->N > 0) return fail("flh_eval: no map (flh_map_build / flh_map_add first)");
    if (!do_search && !h->searched_once && h->N > 0)
        return fail("flh_eval: do_search == 0 before any search on this scan (the reference always searches on the first pass)");
    flh::GranOut gout{};
    if (host_granules) gout = gran_out(h, seq);
    const bool tree = rccl && device_tree(h, host_granules);  // (rccl: called by flh_eval's / flh_eval_group's all-reduce path)
    if (tree) d_out = h->gsum.p;  // group totals instead of the 16x16 block (rccl_allreduce_publish)
    if (tree && h->have_eval && h->last_ext != ext) {
        // gsum is indexed [group][slots of this column count]: with another column count the rows behind this rank's groups sit
        // elsewhere, and what the previous all-reduce left there would be added again on every rank
        HIPC(hipMemsetAsync(h->gsum.p, 0, (size_t)kGranGroups * kGranSlots * sizeof(double), st));
    }
    if (tree && h->N == 0) {      // an empty shard: nothing to launch, its totals are the zeros the buffer holds
        h->last_state = s; h->last_ext = ext; h->have_eval = true; h->aux_valid = false;
        if (do_search) { h->last_search_was_later = h->searched_once; h->searched_once = true; h->d2_valid = false; h->search_state = s; }
        return 0;
    }
    if (do_search && h->stats) HIPC(hipMemsetAsync(h->counter.p, 0, FLH_COUNTER_WORDS * sizeof(u64), st));
    if (do_search && h->pass_ok && (host_granules || tree) && h->N > 0) {
        // the whole searching pass in one launch (flh_pass.hip); timed: the kernel's own start and end stamps in ev3[0] / ev3[3].
        // With the plane cache no later pass reads the neighbours' coordinates: the cache keeps their indices (flh_config.index_cache)
        uint32_t* idx = (h->plane_cache && h->cfg.index_cache) ? h->nn_idx.p : nullptr;
        HIPC(flh::launch_pass(h->cfg.eigen_order, h->grid, s, h->cur_body, (int)h->N, (uint32_t)h->pts_cap, h->cfg.max_sqdist,
                              h->cfg.plane_threshold, ext, h->nn_pts.p, h->nn_cnt.p, h->selected.p, h->plane_cache ? h->plane.p : nullptr,
                              h->partials.p, h->tickets.p, gout, seq, gran_group_size(h->N),
                              h->stats ? h->counter.p : nullptr, h->own_axis, h->own_lo, h->own_hi, st, timed ? ev3[0] : nullptr,
                              timed ? ev3[3] : nullptr, idx, tree ? h->gsum.p : nullptr));
        h->nn_pts_valid = idx == nullptr;
    } else {
        if (do_search) h->nn_pts_valid = true;  // (the search kernels write the coordinates)
        else if (!h->planes_valid && ensure_nn_pts(h) != 0) return -1;  // a re-fit from the cached neighbours
        if (do_search)
            HIPC(flh::launch_search(h->cfg.lanes_per_query, h->grid, s, h->cur_body, (int)h->N, (uint32_t)h->pts_cap, h->cfg.max_sqdist,
                                    h->rmax, h->nn_pts.p, h->nn_cnt.p, h->selected.p, h->slow_list.p, h->slow_list2.p, h->slow_ub.p,
                                    h->slow_count.p, h->stats ? h->counter.p : nullptr, h->own_axis, h->own_lo, h->own_hi, st,
                                    timed ? ev3[0] : nullptr, timed ? ev3[1] : nullptr));
        HIPC(flh::launch_fit(h->cfg.eigen_order, h->cfg.plane_fit_dtype, s, h->cur_body, h->nn_pts.p, (int)h->N, ext, h->cfg.plane_threshold, h->selected.p, h->normvec.p,
                             h->world.p, h->partials.p, h->part2.p, d_out, seq, h->tickets.p, h->slow_count.p, gout,
                             (host_granules || tree) ? gran_group_size(h->N) : 0, 0, st, h->plane_cache ? h->plane.p : nullptr,
                             (do_search || !h->planes_valid) ? 1 : 2, timed ? ev3[2] : nullptr, timed ? ev3[3] : nullptr));
    }
    if (do_search) {
        h->last_search_was_later = h->searched_once;
        h->searched_once = true;
        h->d2_valid = false;
        h->search_state = s;
    }
    h->planes_valid = h->plane_cache;
    h->aux_valid = false;
    h->last_state = s;
    h->last_ext = ext;
    h->have_eval = true;
    return 0;
}

// feats_down_world / normvec of the LAST evaluation, and pointSearchSqDis of the current neighbour cache, are produced on
// demand: the hot path neither writes nor reads them.
static int ensure_aux(flh_handle* h) {
    if (h->aux_valid || !h->have_eval || h->N == 0) return 0;
    if (ensure_nn_pts(h) != 0) return -1;
    HIPC(hipSetDevice(h->device));
    HIPC(flh::launch_fit(h->cfg.eigen_order, h->cfg.plane_fit_dtype, h->last_state, h->cur_body, h->nn_pts.p, (int)h->N, h->last_ext,
                         h->cfg.plane_threshold, h->selected.p, h->normvec.p, h->world.p, h->partials.p, h->part2.p, h->gram.p, 0.0,
                         h->tickets.p, h->slow_count.p, flh::GranOut{}, 0, 1, h->stream));
    h->aux_valid = true;
    return 0;
}
static int ensure_d2(flh_handle* h) {
    if (h->d2_valid || h->N == 0) return 0;
    if (!h->searched_once) return fail("no search on this scan yet");
    if (ensure_nn_pts(h) != 0) return -1;
    HIPC(hipSetDevice(h->device));
    HIPC(flh::launch_fill_d2(h->search_state, h->cur_body, h->nn_pts.p, (int)h->N, h->nn_d2.p, h->stream));
    h->d2_valid = true;
    return 0;
}

static int pre_gone_relaunch(flh_handle* h, double seq, int ext);
// The group reducers of every rank write {value, sequence} granules straight into this rank's pinned buffer: per rank a
// header (how many granules follow), then [group][slot].  Waits until every granule carries this evaluation's sequence
// number, adding the groups up in (rank, group) order as they are seen complete (fixed order -> identical bits run to
// run, and on every rank), and leaves the 16x16 block in h_gram (G[15][15] = seq).  No device-side final sum, no
// collective, no flag.
// slot of the compact Gram layout -> its place(s) in the 16 x 16 block, tabulated once per column count from gram_slot (the walk
// over all 256 cells with a call each used to be repeated on every pass)
struct GramMap {
    int n = 0;
    short pos[kGranSlots], mir[kGranSlots];
};
static GramMap build_gram_map(int ncol) {
    GramMap m;
    for (int k = 0; k < kGranSlots; ++k) { m.pos[k] = 0; m.mir[k] = -1; }
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            const int sl = flh::gram_slot_host(r, c, ncol);
            if (sl < 0 || sl >= kGranSlots) continue;
            m.pos[sl] = (short)(r * 16 + c);
            m.mir[sl] = (c < 12 && r < c) ? (short)(c * 16 + r) : (short)-1;
            if (sl + 1 > m.n) m.n = sl + 1;
        }
    return m;
}
static const GramMap& gram_map(int ncol) {
    static const GramMap m6 = build_gram_map(6), m12 = build_gram_map(12);
    return ncol == 12 ? m12 : m6;
}
// Four consecutive granules {value, sequence} (64 bytes) at once: true and their values in out4 when all four carry `seq`.  Two
// 32-byte loads, one compare: the pick-up of a pass is a loop over 25 x 31 (25 x 93 with the extrinsic columns) granules that all
// land within a few microseconds of the kernel's end, so what the host spends PER GRANULE is on the critical path of every pass.
// (Each granule is one 16-byte device store and is read here inside one aligned 16-byte half of a vector load, as before.)
static inline bool take4(const double* gp, double seq, double* out4) {
    const __m256d a = _mm256_loadu_pd(gp);       // v0 s0 v1 s1
    const __m256d b = _mm256_loadu_pd(gp + 4);   // v2 s2 v3 s3
    const __m256d ss = _mm256_unpackhi_pd(a, b);  // s0 s2 s1 s3
    if (_mm256_movemask_pd(_mm256_cmp_pd(ss, _mm256_set1_pd(seq), _CMP_EQ_OQ)) != 0xF) return false;
    _mm256_storeu_pd(out4, _mm256_permute4x64_pd(_mm256_unpacklo_pd(a, b), 0xD8));  // v0 v2 v1 v3 -> v0 v1 v2 v3
    return true;
}
static int collect_granules(flh_handle* h, double seq, int do_search, int ext) {
    hipStream_t st = h->stream;
    const int ncol = ext ? 12 : 6;
    const int nslots = flh::gram_slots_host(ncol), nsl = nslots + 1;
    double sum[kGranSlots];
    for (int k = 0; k < nsl; ++k) sum[k] = 0.0;
    const double* base = h->h_gran + ((uint64_t)seq & 1u) * (size_t)h->peer_n * kGranSect * 2;
    uint64_t spins = 0;
    bool retired = false;  // this rank's own kernel has been seen retired (from then on a peer's granules have 30 s)
    auto t_retired = std::chrono::steady_clock::now();
    auto wait_for = [&](const double* gp, double* value) -> int {
        for (;;) {
            const __m128d x = _mm_load_pd(gp);  // one 16-byte read: {value, sequence}
            if (_mm_cvtsd_f64(_mm_unpackhi_pd(x, x)) == seq) { *value = _mm_cvtsd_f64(x); return 0; }
            cpu_relax();
            if ((++spins & 0xFFFFFu) == 0) {
                if (pre_gone_relaunch(h, seq, ext) != 0) return -1;
                // this rank's OWN section is waited for as long as its stream is busy (an evaluation queued behind a long map build
                // or run under a profiler takes what it takes); only a PEER's section has a deadline, counted from the moment this
                // rank's own kernel has retired
                const bool own = gp >= base + (size_t)h->peer_rank * kGranSect * 2 && gp < base + (size_t)(h->peer_rank + 1) * kGranSect * 2;
                if (hipStreamQuery(st) != hipErrorNotReady) {  // this rank's kernel finished or failed
                    HIPC(hipStreamSynchronize(st));
                    const __m128d y = _mm_load_pd(gp);
                    if (own && _mm_cvtsd_f64(_mm_unpackhi_pd(y, y)) != seq) return fail("flh_eval: kernel retired without publishing its result");
                    if (!own) {
                        if (!retired) { retired = true; t_retired = std::chrono::steady_clock::now(); }
                        if (std::chrono::steady_clock::now() - t_retired > std::chrono::seconds(30))
                            return fail("flh_eval: timed out waiting for a peer's granules (ranks out of step?)");
                    }
                }
            }
        }
    };
    // The granules of a group are nsl x 16 bytes that a DEVICE has just written: the first read of every 64-byte line misses all
    // the way (~0.1 us), and the pick-up loop -- one load, one compare, one add per granule -- keeps only a handful of those misses
    // in flight.  The lines of the group that is read NEXT are requested while this one is read: if they have landed they are in
    // the cache when their turn comes; if not, the request costs nothing that the poll would not have cost.  With the extrinsic
    // columns a pass publishes 25 x 94 granules = 588 lines (194 without), and a no-search pass's groups land within 3 us of each
    // other: the host was the last to finish (profiles/r06_call33/: the columns cost k_fit 0.7 us and the pass 3 us).
    auto prefetch_group = [&](const double* gg) {
        const char* p = reinterpret_cast<const char*>(gg);
        for (int b = 0; b < nsl * 16; b += 64) _mm_prefetch(p + b, _MM_HINT_T0);
    };
    // A one-launch searching pass hands the END of the scan's order to the workgroups dispatched first, and workgroups finish in
    // dispatch order: its groups arrive LAST GROUP FIRST, and a section's header -- group 0's reducer
    // publishes it behind its sums -- is the very last granule of the pass.  Waiting for the header and only then reading 25 x 30
    // granules would put the whole pick-up behind the last arrival.  So the granules of such a pass are taken in the order they
    // come -- every rank's last group, then every rank's last but one, ... -- into a buffer, and added up in (rank, group) order
    // afterwards: the same additions in the same order, hence the same bits.  That needs the number of groups of every section
    // before its header is there: this rank's follows from its scan; a peer's is learnt from the header of an earlier pass of the
    // same scan (it depends on the shard's size only), until then the section is read header first, as a no-search pass's is
    // (k_fit's groups arrive in ascending order).
    const int own_red = gran_group_size(h->N);
    const int own_ng = own_red > 0 ? (flh::pass_blocks((int)h->N) + own_red - 1) / own_red : 0;  // what this rank's kernels publish
    bool arrival_order = do_search && use_pass_kernel(h, true);
    int ng_of[flh::kPeersMax] = {};
    for (int r = 0; r < h->peer_n; ++r) {
        ng_of[r] = (r == h->peer_rank) ? own_ng : h->sect_ng[r];
        if (ng_of[r] < 1 || ng_of[r] > kGranGroups) arrival_order = false;
    }
    bool taken[flh::kPeersMax] = {};  // sections whose sums are in val already (arrival order); the others are read header first
    size_t off[flh::kPeersMax + 1] = {};
    if (arrival_order) {
        int max_ng = 0;
        for (int r = 0; r < h->peer_n; ++r) {
            off[r + 1] = off[r] + (size_t)ng_of[r] * nsl;
            if (ng_of[r] > max_ng) max_ng = ng_of[r];
            taken[r] = true;
        }
        if (h->gran_val.size() < off[h->peer_n]) h->gran_val.resize(off[h->peer_n]);
        double* val = h->gran_val.data();
        // a peer's group count is a hint (its shard may have changed since the header it was learnt from): while waiting for a
        // hinted group the section's header is watched too -- it is the last granule a rank publishes, so a header without the
        // awaited group means the hint was wrong, and the section is read header first below
        auto wait_hinted = [&](const double* gp, double* value, const double* hdr) -> int {
            for (uint32_t n = 0;; ++n) {
                const __m128d x = _mm_load_pd(gp);
                if (_mm_cvtsd_f64(_mm_unpackhi_pd(x, x)) == seq) { *value = _mm_cvtsd_f64(x); return 0; }
                if ((n & 63u) == 63u) {
                    const __m128d hx = _mm_load_pd(hdr);
                    if (_mm_cvtsd_f64(_mm_unpackhi_pd(hx, hx)) == seq) {
                        const __m128d y = _mm_load_pd(gp);  // (the group may have landed between the two looks)
                        if (_mm_cvtsd_f64(_mm_unpackhi_pd(y, y)) == seq) { *value = _mm_cvtsd_f64(y); return 0; }
                        return 1;
                    }
                    double dummy;
                    if ((n & 0xFFFFu) == 0xFFFFu && wait_for(hdr, &dummy) != 0) return -1;  // the slow path's checks (it returns once the header is there)
                }
                cpu_relax();
            }
        };
        for (int step = 0; step < max_ng; ++step)
            for (int r = 0; r < h->peer_n; ++r) {
                const int gi = ng_of[r] - 1 - step;
                if (gi < 0 || !taken[r]) continue;
                const double* sect = base + (size_t)r * kGranSect * 2;
                const double* gg = sect + 2 * (1 + (size_t)gi * nsl);
                if (gi > 0) prefetch_group(gg - 2 * (size_t)nsl);  // (the groups arrive last first)
                if (gi > 1) prefetch_group(gg - 4 * (size_t)nsl);
                double* vdst = &val[off[r] + (size_t)gi * nsl];
                for (int k = 0; k < nsl; ++k) {
                    if (k + 4 <= nsl && take4(gg + 2 * k, seq, vdst + k)) { k += 3; continue; }  // four at once when they are there
                    const int rc = (r == h->peer_rank) ? wait_for(gg + 2 * k, vdst + k) : wait_hinted(gg + 2 * k, vdst + k, sect);
                    if (rc < 0) return -1;
                    if (rc > 0) { taken[r] = false; break; }
                }
            }
        for (int r = 0; r < h->peer_n; ++r) {
            if (!taken[r]) continue;
            double cnt_d = 0;
            if (wait_for(base + (size_t)r * kGranSect * 2, &cnt_d) != 0) return -1;
            if ((int)cnt_d != ng_of[r] * nsl) {
                if (r == h->peer_rank) return fail("flh_eval: malformed granule section");
                taken[r] = false;  // a peer with more groups than the hint said
            }
        }
    }
    for (int r = 0; r < h->peer_n; ++r) {  // the sums, in (rank, group) order
        if (taken[r]) {
            const double* val = h->gran_val.data();
            for (int gi = 0; gi < ng_of[r]; ++gi) {  // (behind the pass's LAST granule: four slots per step, each slot's additions in group order)
                const double* vg = val + off[r] + (size_t)gi * nsl;
                int k = 0;
                for (; k + 4 <= nsl; k += 4) _mm256_storeu_pd(sum + k, _mm256_add_pd(_mm256_loadu_pd(sum + k), _mm256_loadu_pd(vg + k)));
                for (; k < nsl; ++k) sum[k] += vg[k];
            }
            continue;
        }
        const double* sect = base + (size_t)r * kGranSect * 2;
        double cnt_d = 0;
        if (wait_for(sect, &cnt_d) != 0) return -1;
        const int cnt = (int)cnt_d;
        if (cnt <= 0 || cnt % nsl != 0 || cnt / nsl > kGranGroups) return fail("flh_eval: malformed granule section");
        h->sect_ng[r] = cnt / nsl;
        for (int gi = 0; gi < cnt / nsl; ++gi) {
            const double* gg = sect + 2 * (1 + (size_t)gi * nsl);
            if (gi + 1 < cnt / nsl) prefetch_group(gg + 2 * (size_t)nsl);
            if (gi + 2 < cnt / nsl) prefetch_group(gg + 4 * (size_t)nsl);
            for (int k = 0; k < nsl; ++k) {
                double v4[4];
                if (k + 4 <= nsl && take4(gg + 2 * k, seq, v4)) {  // four slots at once (each slot's sum keeps its order of additions)
                    _mm256_storeu_pd(sum + k, _mm256_add_pd(_mm256_loadu_pd(sum + k), _mm256_loadu_pd(v4)));
                    k += 3;
                    continue;
                }
                double v;
                if (wait_for(gg + 2 * k, &v) != 0) return -1;
                sum[k] += v;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (do_search) h->n_second_stage += (uint64_t)sum[nslots];
    double* G = h->h_gram;
    std::memset(G, 0, 256 * sizeof(double));
    const GramMap& gm = gram_map(ncol);
    for (int sl = 0; sl < gm.n; ++sl) {
        G[gm.pos[sl]] = sum[sl];
        if (gm.mir[sl] >= 0) G[gm.mir[sl]] = sum[sl];  // the block is symmetric bit for bit (same products, same order)
    }
    G[255] = seq;
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

// ---- the pre-launched no-search pass (flh_eval_expect_next) ----------------------------------------------------------------
// A filter that knows what its NEXT evaluation will probably be says so before it begins this one (the mirror esekf does:
// include/fastlio_amd/esekfom.hpp).  For FLH_NEXT_NOSEARCH, flh_eval_begin -- after it has started its own pass -- enqueues
// k_fit_mb for the next one: the launch call, the queue's processing and the dispatch ramp happen beside the pass that is running.
// When the next flh_eval_begin asks for exactly that (a no-search evaluation of the same scan, not timed, nothing enqueued in
// between) it writes the state into the mailbox instead of launching; anything else releases the waiting kernel with one store
// (it retires within a microsecond or two) and goes the usual way.  A wrong expectation costs time, never correctness; a kernel
// nobody comes for gives up after 20 ms (flh_mail_dev.hpp).  Measured (profiles/r05_call1/, same box, alternating): no-search
// pass 18.3 -> 14.7 us; the experiment armed after EVERY pass and paid 2.3 us per searching pass for the releases, which the
// filter's hints avoid (profiles/r05_call2/).
static int pre_init(flh_handle* h) {
    flh_handle::PreLaunch& p = h->pre;
    if (p.host_box) return 0;
    HIPC(hipHostMalloc((void**)&p.host_box, 16 * sizeof(double), hipHostMallocDefault));
    HIPC(hipHostMalloc((void**)&p.status, 64, hipHostMallocDefault));
    HIPC(hipMalloc((void**)&p.dev_box, 16 * sizeof(double)));
    std::memset(p.host_box, 0, 16 * sizeof(double));
    std::memset(p.status, 0, 64);
    HIPC(hipMemset(p.dev_box, 0, 16 * sizeof(double)));
    HIPC(hipDeviceSynchronize());
    return 0;
}
static void pre_release(flh_handle* h) {
    flh_handle::PreLaunch& p = h->pre;
    if (p.host_box) (void)hipHostFree(p.host_box);
    if (p.status) (void)hipHostFree(p.status);
    if (p.dev_box) (void)hipFree(p.dev_box);
    p.host_box = nullptr; p.status = nullptr; p.dev_box = nullptr;
}
// state (or nothing) + the {sequence, command} word, the word last
static void pre_post(flh_handle* h, const StateDev* s, uint32_t cmd) {
    flh_handle::PreLaunch& p = h->pre;
    static_assert(sizeof(StateDev) == 14 * sizeof(double), "the mailbox carries StateDev as 14 doubles");
    // two 64-byte lines, each seven doubles of the state + a word {checksum of the line, cmd, seq} written after them
    // (flh_mail_dev.hpp: the forwarder's poll is the read of the state)
    if (s) {
        const double* src = reinterpret_cast<const double*>(s);
        for (int i = 0; i < 7; ++i) { p.host_box[i] = src[i]; p.host_box[8 + i] = src[7 + i]; }
    }
    uint64_t ck[2] = {0, 0};
    for (int l = 0; l < 2; ++l)
        for (int i = 0; i < 7; ++i) {
            uint64_t bits;
            std::memcpy(&bits, p.host_box + 8 * l + i, 8);
            ck[l] ^= flh::mail_mix(bits, i);
        }
    for (int l = 0; l < 2; ++l) {
        const uint64_t w = ((uint64_t)flh::mail_fold(ck[l]) << 40) | ((uint64_t)(cmd & 0xFFu) << 32) | (uint64_t)p.mseq;
        __atomic_store_n(reinterpret_cast<uint64_t*>(p.host_box + 8 * l + 7), w, __ATOMIC_RELEASE);
    }
    p.armed = false;
}
static void pre_cancel(flh_handle* h) {
    if (!h->pre.armed) return;
    pre_post(h, nullptr, flh::kMailAbort);
    h->pre.n_abort++;
}
// flh_eval_begin: may this evaluation be handed to the kernel that is waiting?  true: it has been (the caller skips its launch).
static bool pre_try_go(flh_handle* h, const StateDev& s, const flh_handle::PendingEval& pe) {
    flh_handle::PreLaunch& p = h->pre;
    p.via_mail = false;
    if (!p.armed) return false;
    const bool fits = !pe.do_search && pe.granules && !pe.timed && pe.ext == p.ext && pe.seq == p.eval_seq && h->N == p.N &&
                      h->cur_body == p.body && !h->map_pending && h->planes_valid && h->searched_once && !h->comm && h->peer_n == 1;
    if (!fits) {
        pre_cancel(h);
        return false;
    }
    pre_post(h, &s, flh::kMailGo);
    p.via_mail = true;
    p.n_go++;
    // what enqueue_eval notes down for a no-search evaluation
    h->planes_valid = h->plane_cache;
    h->aux_valid = false;
    h->last_state = s;
    h->last_ext = pe.ext;
    h->have_eval = true;
    return true;
}
// flh_eval_begin, after this evaluation's pass is under way: enqueue the next one's kernel ahead of its state -- when the caller
// expects a no-search evaluation next (the expectation is consumed here, whatever it was)
static int pre_arm(flh_handle* h, const flh_handle::PendingEval& pe) {
    flh_handle::PreLaunch& p = h->pre;
    const int expect = p.expect;
    p.expect = FLH_NEXT_UNKNOWN;
    if (p.off || p.armed || expect != FLH_NEXT_NOSEARCH) return 0;
    const int red1 = gran_group_size(h->N);
    // (timing_stride 1 = every evaluation carries events: the next one would be refused anyway)
    const bool next_may_be_timed = h->timing_stride > 0 && !h->timing_search_only;
    if (!pe.granules || h->peer_n != 1 || h->comm || !h->plane_cache || h->cfg.eigen_order != FLH_ORDER_SSE || h->cfg.plane_fit_dtype != 0 ||
        h->N == 0 || red1 < 4 || next_may_be_timed || h->stats /* flh_eval_end waits for the stream then */ || !h->cur_body || !h->plane.p)
        return 0;
    if (pre_init(h) != 0) return -1;
    p.mseq++;
    p.eval_seq = (double)(h->seq + 1);
    p.ext = pe.ext;
    p.N = h->N;
    p.body = h->cur_body;
    flh::MailArgs m;
    m.host_box = p.host_box;
    m.dev_box = p.dev_box;
    m.status = p.status;
    m.seq = p.mseq;
    HIPC(flh::launch_fit_mb(m, h->cur_body, (int)h->N, pe.ext, h->cfg.plane_threshold, h->selected.p, h->partials.p, p.eval_seq, h->tickets.p,
                            h->slow_count.p, gran_out(h, p.eval_seq), red1, h->plane.p, h->stream));
    p.armed = true;
    p.n_armed++;
    return 0;
}
// collect_granules' slow path: the evaluation went to the mailbox, but the kernel had given up before the mail arrived (it has
// written nothing): launch the pass the usual way, once, and go on waiting for the same granules
static int pre_gone_relaunch(flh_handle* h, double seq, int ext) {
    flh_handle::PreLaunch& p = h->pre;
    if (!p.via_mail) return 0;
    const uint64_t st = __atomic_load_n(reinterpret_cast<uint64_t*>(p.status), __ATOMIC_ACQUIRE);
    if (st == (((uint64_t)flh::kMailLost << 32) | (uint64_t)p.mseq)) return fail("flh_eval: a workgroup of the pre-launched pass never saw its state (flh_mail_dev.hpp)");
    if (st != (((uint64_t)flh::kMailGone << 32) | (uint64_t)p.mseq)) return 0;
    p.via_mail = false;
    p.n_gone++;
    return enqueue_eval(h, h->last_state, 0, ext, h->h_gram, seq, nullptr, true);
}

int flh_eval_expect_next(flh_handle* h, int kind) {
    if (!h) return fail("flh_eval_expect_next: null handle");
    if (kind != FLH_NEXT_UNKNOWN && kind != FLH_NEXT_NOSEARCH && kind != FLH_NEXT_NONE) return fail("flh_eval_expect_next: unknown kind");
    h->pre.expect = h->pre.off ? FLH_NEXT_UNKNOWN : kind;
    if (kind == FLH_NEXT_NONE) pre_cancel(h);  // nothing follows: a kernel that is still waiting is released now
    return 0;
}
int flh_set_prelaunch(flh_handle* h, int on) {
    if (!h) return fail("flh_set_prelaunch: null handle");
    h->pre.off = on == 0;
    if (h->pre.off) { pre_cancel(h); h->pre.expect = FLH_NEXT_UNKNOWN; }
    return 0;
}
int flh_debug_stage_stats(flh_handle* h, double out[10], int reset) {
    if (!h || !out) return fail("flh_debug_stage_stats: null argument");
    std::lock_guard<std::mutex> sg(h->stage_mu);  // the staging thread writes its counters inside a staging, which holds this lock
    const flh_handle::StageDiag& d = h->sdiag;
    const double v[10] = {d.n_jobs, d.enq_us, d.enq_max_us, d.h2d_wait_us, d.h2d_wait_max_us, d.n_act, d.act_wait_us, d.act_wait_max_us, d.act_ev_not_ready, d.act_slept};
    for (int i = 0; i < 10; ++i) out[i] = v[i];
    if (reset) h->sdiag = flh_handle::StageDiag();
    return 0;
}
int flh_debug_search_redone(const flh_handle* h, uint64_t* out) {
    if (!h || !out) return fail("flh_debug_search_redone: null argument");
    *out = h->n_search_redone;
    return 0;
}
int flh_get_prelaunch_stats(const flh_handle* h, uint64_t out4[4]) {
    if (!h || !out4) return fail("flh_get_prelaunch_stats: null argument");
    out4[0] = h->pre.n_armed; out4[1] = h->pre.n_go; out4[2] = h->pre.n_abort; out4[3] = h->pre.n_gone;
    return 0;
}
// One h_share_model evaluation in two halves: flh_eval_begin enqueues the pass and returns; flh_eval_end waits for its normal
// equations.  Between the two the caller's thread is free for host work that does not depend on them (the mirror esekf projects
// the covariance and inverts P / R there: a third of the 23x23 algebra of a pass leaves the critical path).
int flh_eval_begin(flh_handle* h, const double rot[4], const double pos[3], const double offR[4], const double offT[3], int do_search,
                   int ext) {
    if (!h) return fail("flh_eval: null handle");
    if (!rot || !pos || !offR || !offT) return fail("flh_eval: null argument");
    if (h->pend.active) return fail("flh_eval_begin: the previous evaluation has not been collected (flh_eval_end)");
    HIPC(hipSetDevice(h->device));
    const StateDev s = make_state(rot, pos, offR, offT);
    flh_handle::PendingEval& pe = h->pend;
    pe = flh_handle::PendingEval();
    pe.do_search = do_search != 0;
    pe.ext = ext;
    pe.timed = h->timing_stride > 0 && (!h->timing_search_only || do_search) && (h->eval_no++ % (uint64_t)h->timing_stride) == 0;
    hipEvent_t* ev3 = nullptr;
    if (pe.timed) {
        pe.deferred = h->timing_stride >= 2 && ensure_event_pool(h);
        if (pe.deferred && h->evp_n == flh_handle::kEvPool) drain_events(h);
        ev3 = pe.deferred ? h->evp[h->evp_n] : h->ev;
    }
    pe.seq = (double)(++h->seq);
    hipStream_t st = h->stream;
    // group sums as granules in pinned memory (this rank's and, with peers, every rank's): not with an RCCL communicator (the
    // block is all-reduced on the device), not for an empty scan
    pe.granules = !h->comm && h->N > 0 && gran_group_size(h->N) > 0;
    pe.one_launch = pe.do_search && use_pass_kernel(h, pe.granules);  // (granules, or RCCL's group totals)
    if (h->peer_n > 1 && !pe.granules) return fail("flh_eval: a scan shard may not be empty when the ranks exchange granules (flh_peer_*)");
    if (pre_try_go(h, s, pe)) {
        // the kernel of this evaluation was enqueued beside the previous pass: the state went to its mailbox
    } else if (h->comm) {
        // this rank's partial block stays in device memory, RCCL sums the ranks' blocks in place (256 doubles: latency-bound,
        // xGMI bandwidth is irrelevant), then one small kernel publishes the sum + sequence word to pinned host memory
        if (enqueue_eval(h, s, do_search, ext, h->gram.p, 0.0, ev3, false, true) != 0) return -1;
        if (rccl_allreduce_publish(h, pe.seq) != 0) return -1;
    } else {
        // The first search of a scan follows the previous scan's map change (src/laserMapping.cpp:960 after :990 of the scan
        // before).  A change enqueued without the host's wait publishes its counters when its last kernel ends; waiting for them
        // HERE would leave the device idle from that moment until this pass's launch has travelled (granule to the host + launch
        // call + dispatch: ~7 us per scan of the config-3 stream).  The pass kernel needs nothing of those counters -- the index is
        // updated in place, behind the same pointers -- so it is enqueued behind the change now and the counters are folded in
        // flh_eval_end; in the rare case that they ask for a re-index or a replay, the pass is run again on the settled map.
        pe.behind_map_change = h->map_pending && h->mi_deferred && pe.one_launch && pe.granules && h->peer_n <= 1;
        pe.reindexed = h->n_reindex;
        pe.replayed = h->n_mi_replayed;
#ifdef FLH_SETTLE_FIRST  // developer builds (A/B, tools/r06_call24.sh): the change's counters first, as before
        pe.behind_map_change = false;
#endif
        if (enqueue_eval(h, s, do_search, ext, h->h_gram, pe.seq, ev3, pe.granules, false, pe.behind_map_change) != 0) return -1;
    }
    if (h->stats && do_search) HIPC(hipMemcpyAsync(h->h_counter, h->counter.p, sizeof(u64), hipMemcpyDeviceToHost, st));
    if (do_search) { h->n_search_pass++; if (pe.one_launch) h->n_one_launch++; } else h->n_nosearch_pass++;
    if (pre_arm(h, pe) != 0) return -1;
    pe.active = true;
    return 0;
}

int flh_eval_end(flh_handle* h, double HTH[144], double HTh[12], int64_t* n_eff, double* total_residual) {
    if (!h) return fail("flh_eval: null handle");
    if (!HTH || !HTh) return fail("flh_eval: null argument");
    if (!h->pend.active) return fail("flh_eval_end: no evaluation under way (flh_eval_begin)");
    const flh_handle::PendingEval pe = h->pend;
    h->pend.active = false;
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    double seq = pe.seq;
    const bool do_search = pe.do_search, one_launch = pe.one_launch, timed = pe.timed, deferred = pe.deferred;
    if (pe.behind_map_change) {
        // the map change this pass was enqueued behind: its counters now (they were published before the pass started; another
        // call may have collected them meanwhile)
        if (map_settle(h) != 0) return -1;
        if (h->n_reindex != pe.reindexed || h->n_mi_replayed != pe.replayed) {
            // the pass searched an index that was about to be rebuilt (or a map without the change): once more, on the settled map
            pre_cancel(h);
            ++h->n_search_redone;
            seq = (double)(++h->seq);
            if (enqueue_eval(h, h->last_state, 1, pe.ext, h->h_gram, seq, nullptr, true) != 0) return -1;
            if (h->stats) HIPC(hipMemcpyAsync(h->h_counter, h->counter.p, sizeof(u64), hipMemcpyDeviceToHost, st));
        }
    }
    if (pe.granules) {
        if (collect_granules(h, seq, do_search, pe.ext) != 0) return -1;
        if (h->stats) HIPC(hipStreamSynchronize(st));  // the candidate counter's copy
    } else if (h->stats) {
        HIPC(hipStreamSynchronize(st));
    } else {
        // k_fit publishes the block with system-scope stores and then the sequence word in G[15][15]: poll it instead of
        // waiting for the kernel to retire and the runtime to notice (saves the completion-signal round trip per pass).
        const volatile double* flag = h->h_gram + 255;
        uint64_t spins = 0;
        while (*flag != seq) {
            cpu_relax();
            if ((++spins & 0xFFFFFu) == 0 && hipStreamQuery(st) != hipErrorNotReady) {  // finished or failed without publishing
                HIPC(hipStreamSynchronize(st));
                if (*flag != seq) return fail("flh_eval: kernel retired without publishing its result");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    // timed evaluations: the kernels' own start / stop time stamps (four of them: the search's, the fit's; the one-launch pass:
    // the pass kernel's two); the last one completes when the last kernel retires, a moment after its granules
    if (deferred) {  // recorded, not awaited: read by drain_events
        h->evp_search[h->evp_n] = (uint8_t)(((do_search && h->N > 0) ? (h->last_search_was_later ? 2 : 1) : 0) | (one_launch ? 4 : 0));  // N == 0: no search kernel ran
        h->evp_n++;
    } else if (timed) {
        HIPC(hipEventSynchronize(h->ev[3]));
    }
    if (h->h_gram[255] != seq) return fail("flh_eval: result sequence mismatch");
    h->h_gram[255] = 0.0;  // G[15][15] is structurally zero
    flh_unpack_gram(h->h_gram, HTH, HTh, n_eff, total_residual);
    float a = 0, b = 0, c = 0;
    if (timed && !deferred) {
        if (one_launch) {
            (void)hipEventElapsedTime(&c, h->ev[0], h->ev[3]);
            a = c;
        } else {
            if (do_search && h->N > 0) (void)hipEventElapsedTime(&a, h->ev[0], h->ev[1]);
            (void)hipEventElapsedTime(&b, h->ev[2], h->ev[3]);
            (void)hipEventElapsedTime(&c, h->ev[(do_search && h->N > 0) ? 0 : 2], h->ev[3]);
        }
    }
    if (!deferred && h->evp_n == 0) {  // (with samples pending the last timing is whatever drain_events reads last)
        h->timing.search_ms = do_search ? a : 0.f;
        h->timing.fit_ms = b;
        h->timing.total_ms = c;
    }
    h->timing.candidates = (h->stats && do_search) ? (int64_t)*h->h_counter : 0;
    if (timed && !deferred) {
        if (do_search && h->N > 0) { h->acc[0] += a; h->acc[1] += 1; h->acc_kind[h->last_search_was_later ? 2 : 0] += a; h->acc_kind[h->last_search_was_later ? 3 : 1] += 1; }
        if (!one_launch) { h->acc[2] += b; h->acc[3] += 1; }
        h->acc[4] += c; h->acc[5] += 1;
    }
    return 0;
}

int flh_eval(flh_handle* h, const double rot[4], const double pos[3], const double offR[4], const double offT[3],
             int do_search, int ext, double HTH[144], double HTh[12], int64_t* n_eff, double* total_residual) {
    if (!HTH || !HTh) return fail("flh_eval: null argument");
    if (flh_eval_begin(h, rot, pos, offR, offT, do_search, ext) != 0) return -1;
    return flh_eval_end(h, HTH, HTh, n_eff, total_residual);
}

int flh_get_counters(flh_handle* h, double out[6], int reset) {
    if (!h || !out) return fail("flh_get_counters: null argument");
    drain_events(h);
    for (int i = 0; i < 6; ++i) out[i] = h->acc[i];
    if (reset) {
        for (int i = 0; i < 6; ++i) h->acc[i] = 0;
        for (int i = 0; i < 4; ++i) h->acc_kind[i] = 0;
    }
    return 0;
}
int flh_get_search_counters(flh_handle* h, double out[4]) {
    if (!h || !out) return fail("flh_get_search_counters: null argument");
    drain_events(h);
    for (int i = 0; i < 4; ++i) out[i] = h->acc_kind[i];
    return 0;
}

int flh_eval_device(flh_handle* h, const double x[FLH_NSTATE], int do_search, int ext, double* d_gram256) {
    if (!h || !x || !d_gram256) return fail("flh_eval_device: null argument");
    HIPC(hipSetDevice(h->device));
    const StateDev s = make_state(x + 3, x + 0, x + 7, x + 11);
    return enqueue_eval(h, s, do_search, ext, d_gram256, 0.0, nullptr);
}

// map_incremental() -- src/laserMapping.cpp:427-474, on the neighbour cache the scan's last search left on the
// device.  x = the POSTERIOR state (state_point after the update).  apply != 0 also performs the two Add_Points
// calls (:470-471) and re-indexes the map.
int flh_map_incremental(flh_handle* h, const double x[FLH_NSTATE], double filter_size_map, int flg_EKF_inited, int apply,
                        uint32_t* n_add, uint32_t* n_no_downsample) {
    if (!h || !x) return fail("flh_map_incremental: null argument");
    if (!(filter_size_map > 0)) return fail("flh_map_incremental: filter_size_map must be > 0");
    if (!h->cur_body) return fail("flh_map_incremental: no active scan");
    pre_cancel(h);
    if (map_settle(h) != 0) return -1;
    if (h->N > 0 && !h->searched_once)
        return fail("flh_map_incremental: the active scan has not been searched against the current map");
    HIPC(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const size_t N = h->N;
    const StateDev s_post = make_state(x + 3, x + 0, x + 7, x + 11);
    HIPC(h->mi_world.reserve(N ? N : 1)); HIPC(h->mi_cls.reserve(N ? N : 1));
    if (h->mi_far.cap < N + 1) {  // (re)allocated: the list's counter starts at zero; every call leaves it there
        HIPC(h->mi_far.reserve(N + 1));
        HIPC(hipMemsetAsync(h->mi_far.p, 0, sizeof(uint32_t), st));
    }
    // (pointSearchSqDis is recomputed inside the kernels from the neighbour cache -- the search's own expression -- instead of
    // being materialised by k_fill_d2 first.)  Nearest_Points' coordinates (laserMapping.cpp:436-466): where the last search left map
    // indices (flh_config.index_cache) the kernels read them from the id-ordered array on the spot -- round 5 gathered them into
    // nn_pts first, a kernel and 7.5 us per scan of the config-3 stream.
    uint32_t* const nn_idx = (!h->nn_pts_valid && h->nn_idx.p) ? h->nn_idx.p : nullptr;
    // the two lists' members per block of 256 original indices: a double buffer (the compaction of this call zeroes the other half
    // for the next one -- no memset launch); [0 .. words) of a half were written by its last use
    const uint32_t blk_words = flh::cls_block_words((int)N);
    for (int k = 0; k < 2; ++k)
        if (h->mi_blk[k].cap < blk_words) {
            HIPC(h->mi_blk[k].reserve((size_t)blk_words + blk_words / 2 + 64));
            HIPC(hipMemsetAsync(h->mi_blk[k].p, 0, h->mi_blk[k].cap * sizeof(uint32_t), st));
            h->mi_blk_dirty[k] = 0;
        }
    const int par = h->mi_par;  // (flips when the pair classify + compaction has run: N > 0)
    // A running odometry inserts about as many points with every scan.  When nobody asked for the list lengths, Add_Points is
    // enqueued right behind the compaction with the lengths read on the device: the host does not stand in the middle of the
    // call (a wait for the granule, then the launches, while the device idles).  A change that outgrows the launches is
    // replayed by map_settle().  The launches are sized from the PREVIOUS change (+ 50 %): up to small_change_max() points the
    // one-workgroup path, above it the general path (scan + device-wide sort over the bound, the entries behind the true end
    // reading "no point").  The voxel table of that Add_Points is emptied by the classification kernel and filled by the
    // compaction kernel on their way (no fill launch, no insert launch).
    const uint32_t cap = flh::small_change_max();
    const bool no_wait = N > 0 && apply && !n_add && !n_no_downsample && h->mi_pred_n != 0xFFFFFFFFu;
    size_t bound = 0;
    unsigned long long* tab_fill = nullptr;
    uint32_t tab_slots = 0;
    if (no_wait) {
        bound = (size_t)h->mi_pred_n + h->mi_pred_n / 2 + 1024;
        if (h->cfg.fused_small_changes != 0 && bound <= cap) bound = cap;
        bound = std::min<size_t>(bound, N);
        tab_slots = flh::vox_table_slots((uint32_t)bound);
        HIPC(h->vox_tab.reserve(2 * (size_t)tab_slots));
        HIPC(h->mu_alive.reserve(bound));
        tab_fill = h->vox_tab.p;
    }
    HIPC(flh::launch_mi_classify(h->grid, h->grid.hash_mask + 1, (uint32_t)h->M, h->search_state, s_post, h->cur_body,
                                 h->nn_pts.p, nn_idx, h->map_orig.p, (uint32_t)h->n_ids, h->nn_cnt.p, h->cfg.max_sqdist, (int)N,
                                 filter_size_map, flg_EKF_inited, h->live.p, h->mi_world.p, h->mi_cls.p,
                                 N > 0 ? h->mi_blk[par].p : nullptr, h->mi_far.p, st, tab_fill, 2u * tab_slots));
    if (h->cur && apply && !n_add && !n_no_downsample) {
        // the last readers of the scan's slot are enqueued; on the no-wait path below nobody waits for them before the ring
        // comes round to this slot again (flh_esekf_run_scans stages two scans ahead): the slot's next staging does
        if (!h->cur->consumed) HIPC(hipEventCreateWithFlags(&h->cur->consumed, hipEventDisableTiming));
        HIPC(hipEventRecord(h->cur->consumed, st));
        h->cur->consumed_set = true;
    }
    uint32_t c1 = 0, c2 = 0;
    if (N > 0) {
        // the two lists hold at most N points together: compacted without knowing their lengths (one kernel: the blocks' counts
        // added up by every workgroup, no library scan); the kernel hands the lengths to the host as a granule in pinned memory
        // (no copy, no stream synchronisation)
        HIPC(h->mu_add.reserve(N + 1));
        HIPC(h->mi_cnt.reserve(4));
        const uint32_t seq = ++h->mi_seq;
        HIPC(flh::launch_cls_compact(h->mi_world.p, h->mi_cls.p, h->mi_blk[par].p, h->mi_blk[par ^ 1].p, h->mi_blk_dirty[par ^ 1], (int)N,
                                     h->mu_add.p, h->h_mi, seq, st, h->mi_cnt.p, h->mi_far.p, tab_fill, tab_slots, filter_size_map,
                                     h->mu_alive.p, h->ctr.p, (uint32_t)bound));
        h->mi_blk_dirty[par ^ 1] = 0;
        h->mi_blk_dirty[par] = blk_words;
        h->mi_par ^= 1;
        if (no_wait) {
            h->mi_valid_N = N;
            h->mi_cls_seq = seq;
            ++h->n_mi_deferred;
            return apply_map_changes(h, h->mu_add.p, bound, 0, filter_size_map, h->mi_cnt.p, true);
        }
        if (wait_granule(h, 0, seq, "flh_map_incremental") != 0) return -1;
        c1 = h->h_mi[0];
        c2 = h->h_mi[1] - h->h_mi[0];
    }
    if (n_add) *n_add = c1;
    if (n_no_downsample) *n_no_downsample = c2;
    h->mi_valid_N = N;
    if (!apply) {
        HIPC(hipStreamSynchronize(st));
        return 0;
    }
    return apply_map_changes(h, h->mu_add.p, c1, c2, filter_size_map);
}

// Per scan point (original order): 0 = not inserted, 1 = PointToAdd, 2 = PointNoNeedDownsample; and the world
// points map_incremental computed (feats_down_world, :436).  Either pointer may be NULL.
int flh_fetch_map_incremental(flh_handle* h, uint8_t* cls, float* world_xyz) {
    if (!h) return fail("flh_fetch_map_incremental: null handle");
    if (h->mi_valid_N != h->N || !h->cur_body) return fail("flh_fetch_map_incremental: no classification for the active scan");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    if (cls) HIPC(hipMemcpyAsync(cls, h->mi_cls.p, N, hipMemcpyDeviceToHost, h->stream));
    std::vector<float4> w;
    if (world_xyz) {
        w.resize(N);
        HIPC(hipMemcpyAsync(w.data(), h->mi_world.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    }
    HIPC(hipStreamSynchronize(h->stream));
    if (world_xyz)
        for (size_t i = 0; i < N; ++i) { world_xyz[3 * i] = w[i].x; world_xyz[3 * i + 1] = w[i].y; world_xyz[3 * i + 2] = w[i].z; }
    return 0;
}

int flh_last_timing(flh_handle* h, flh_timing* t) {
    if (!h || !t) return fail("flh_last_timing: null argument");
    drain_events(h);
    *t = h->timing;
    return 0;
}
int flh_set_timing_stride(flh_handle* h, int every_n) {
    if (!h) return fail("flh_set_timing_stride: null handle");
    drain_events(h);
    h->timing_stride = every_n < 0 ? 0 : every_n;
    h->timing_search_only = false;
    h->eval_no = 0;
    if (h->timing_stride >= 2) {  // create the event pool here, not inside the caller's timed region
        (void)hipSetDevice(h->device);
        (void)ensure_event_pool(h);
    }
    return 0;
}
int flh_set_timing_sampling(flh_handle* h, int every_n, int search_only) {
    if (!h) return fail("flh_set_timing_sampling: null handle");
    if (flh_set_timing_stride(h, every_n) != 0) return -1;
    h->timing_search_only = search_only != 0;
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
    if (map_settle(h) != 0) return -1;
    if (which != 0 && !h->searched_once) return fail("flh_time_kernel: fit kernel timed before any search");
    if (which != 0 && ensure_nn_pts(h) != 0) return -1;
    HIPC(hipEventRecord(h->ev[0], st));
    for (int it = 0; it < iters; ++it) {
        if (which == 0) {
            HIPC(flh::launch_search(h->cfg.lanes_per_query, h->grid, s, h->cur_body, (int)h->N, (uint32_t)h->pts_cap,
                                    h->cfg.max_sqdist, h->rmax, h->nn_pts.p, h->nn_cnt.p, h->selected.p,
                                    h->slow_list.p, h->slow_list2.p, h->slow_ub.p, h->slow_count.p, nullptr, h->own_axis, h->own_lo, h->own_hi, st));
            HIPC(hipMemsetAsync(h->slow_count.p, 0, 2 * flh::list_stripes() * sizeof(uint32_t), st));
            h->searched_once = true;
            h->search_state = s;
            h->d2_valid = false;   // pointSearchSqDis on demand (ensure_d2) must be recomputed for the new neighbours
            h->planes_valid = false;
            h->nn_pts_valid = true;
        } else {
            HIPC(flh::launch_fit(h->cfg.eigen_order, h->cfg.plane_fit_dtype, s, h->cur_body, h->nn_pts.p, (int)h->N, ext, h->cfg.plane_threshold, h->selected.p,
                                 h->normvec.p, h->world.p, h->partials.p, h->part2.p, h->gram.p, 0.0, h->tickets.p, h->slow_count.p,
                                 flh::GranOut{}, 0, 0, st));
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
    if (!h->cur) return fail("flh_fetch_selected: no active scan");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    std::vector<uint8_t> tmp(N);
    HIPC(hipMemcpyAsync(tmp.data(), h->selected.p, N, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) flags[perm[i]] = tmp[i];
    return 0;
}

int flh_fetch_neighbors(flh_handle* h, int32_t* idx, float* d2, uint8_t* cnt) {
    if (!h || !idx || !d2) return fail("flh_fetch_neighbors: null argument");
    if (!h->cur) return fail("flh_fetch_neighbors: no active scan");
    if (map_settle(h) != 0) return -1;
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    if (ensure_nn_pts(h) != 0 || ensure_d2(h) != 0) return -1;
    std::vector<float4> pts(5 * N);
    std::vector<float> dd(5 * N);
    std::vector<uint8_t> cc(N);
    HIPC(hipMemcpyAsync(pts.data(), h->nn_pts.p, 5 * N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(dd.data(), h->nn_d2.p, 5 * N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(cc.data(), h->nn_cnt.p, N, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    // the cache carries point ids; the caller gets positions in the array flh_map_download returns (= rank among the live ids)
    if (h->n_ids != h->M && !h->id_pos_valid) {
        std::vector<uint8_t> dead(h->n_ids);
        HIPC(hipMemcpy(dead.data(), h->dead_id.p, h->n_ids, hipMemcpyDeviceToHost));
        h->id_pos.resize(h->n_ids);
        uint32_t r = 0;
        for (size_t i = 0; i < h->n_ids; ++i) { h->id_pos[i] = r; r += dead[i] ? 0u : 1u; }
        h->id_pos_valid = true;
    }
    const bool translate = h->n_ids != h->M;
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) {
        const size_t o = perm[i];
        for (int j = 0; j < 5; ++j) {
            int32_t id;
            std::memcpy(&id, &pts[(size_t)j * N + i].w, 4);
            if (translate && id >= 0 && (size_t)id < h->n_ids) id = (int32_t)h->id_pos[(size_t)id];
            idx[o * 5 + j] = id;
            d2[o * 5 + j] = id < 0 ? INFINITY : dd[(size_t)j * N + i];
        }
        if (cnt) cnt[o] = cc[i];
    }
    return 0;
}

int flh_fetch_world(flh_handle* h, float* xyz) {
    if (!h || !xyz) return fail("flh_fetch_world: null argument");
    if (!h->cur) return fail("flh_fetch_world: no active scan");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    if (ensure_aux(h) != 0) return -1;
    std::vector<float4> w(N);
    HIPC(hipMemcpyAsync(w.data(), h->world.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
    const uint32_t* perm = h->cur->h_perm.data();
    for (size_t i = 0; i < N; ++i) {
        const size_t o = perm[i];
        xyz[3 * o] = w[i].x; xyz[3 * o + 1] = w[i].y; xyz[3 * o + 2] = w[i].z;
    }
    return 0;
}

int flh_fetch_normvec(flh_handle* h, float* out) {
    if (!h || !out) return fail("flh_fetch_normvec: null argument");
    if (!h->cur) return fail("flh_fetch_normvec: no active scan");
    const size_t N = h->N;
    if (N == 0) return 0;
    HIPC(hipSetDevice(h->device));
    if (ensure_aux(h) != 0) return -1;
    std::vector<float4> nv(N);
    HIPC(hipMemcpyAsync(nv.data(), h->normvec.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
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
static int fetch_rows_local(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows);
static int fetch_rows_gathered(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows);
static int fetch_rows_peers(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows);
int flh_fetch_rows(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows) {
    if (!h || !n_rows) return fail("flh_fetch_rows: null argument");
    if (!h->have_eval) return fail("flh_fetch_rows: no evaluation yet");
    pre_cancel(h);  // (the gain-form branch fetches rows between two passes of an update)
    if (h->comm && h->comm_size > 1) return fetch_rows_gathered(h, hx, hv, cap, n_rows);
    if (h->peer_seg && h->peer_n > 1) return fetch_rows_peers(h, hx, hv, cap, n_rows);
    return fetch_rows_local(h, hx, hv, cap, n_rows);
}
static int fetch_rows_local(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows) {
    const size_t N = h->N;
    std::vector<uint8_t> sel(N ? N : 1);
    std::vector<float4> nv(N ? N : 1);
    if (N) {
        HIPC(hipSetDevice(h->device));
        if (ensure_aux(h) != 0) return -1;
        HIPC(hipMemcpyAsync(sel.data(), h->selected.p, N, hipMemcpyDeviceToHost, h->stream));
        HIPC(hipMemcpyAsync(nv.data(), h->normvec.p, N * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
        HIPC(hipStreamSynchronize(h->stream));
    }
    int64_t n = 0;
    for (size_t i = 0; i < N; ++i) n += sel[i] ? 1 : 0;
    *n_rows = n;
    if (!hx || !hv) return 0;
    if (cap < n) return fail("flh_fetch_rows: buffers too small");
    if (ensure_host_copy(h, *h->cur) != 0) return -1;
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


// ---------------------------------------------------------------------------------------------
// Multi-GPU: RCCL all-reduce of the normal equations (BASELINE north_star).  RCCL is loaded on first use
// (dlopen), so a single-GPU process never pays for it and the library has no link-time dependency on it.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return 0;
    void* l = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!l) l = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!l) return fail(std::string("RCCL not available: ") + dlerror());
#define RSYM(field, name)                                                         \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(l, name));      \
    if (!g_rccl.field) return fail(std::string("RCCL symbol missing: ") + name)
    RSYM(GetUniqueId, "ncclGetUniqueId");
    RSYM(CommInitRank, "ncclCommInitRank");
    RSYM(CommInitAll, "ncclCommInitAll");
    RSYM(CommDestroy, "ncclCommDestroy");
    RSYM(AllReduce, "ncclAllReduce");
    RSYM(AllGather, "ncclAllGather");
    RSYM(GroupStart, "ncclGroupStart");
    RSYM(GroupEnd, "ncclGroupEnd");
    RSYM(GetErrorString, "ncclGetErrorString");
#undef RSYM
    g_rccl.lib = l;
    return 0;
}
}  // namespace
#define NCCLC(expr)                                                                                          \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return fail(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));           \
    } while (0)

int flh_rccl_unique_id(char id[FLH_RCCL_ID_BYTES]) {
    if (!id) return fail("flh_rccl_unique_id: null argument");
    if (rccl_load() != 0) return -1;
    static_assert(FLH_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    NCCLC(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

// one process per GPU: every rank calls this with the id rank 0 generated (handed around by whatever launched the ranks)
int flh_rccl_init_rank(flh_handle* h, int nranks, const char id[FLH_RCCL_ID_BYTES], int rank) {
    if (!h || !id) return fail("flh_rccl_init_rank: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("flh_rccl_init_rank: bad rank / nranks");
    if (h->comm) return fail("flh_rccl_init_rank: the handle already has a communicator");
    if (h->peer_seg) return fail("flh_rccl_init_rank: the handle is attached to peers (flh_peer_*)");
    if (rccl_load() != 0) return -1;
    HIPC(hipSetDevice(h->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    NCCLC(g_rccl.CommInitRank(&h->comm, nranks, u, rank));
    h->comm_size = nranks;
    h->comm_rank = rank;
    return 0;
}

// one process, one handle per device (ncclCommInitAll): drive them with flh_eval_group
int flh_rccl_init_all(flh_handle* const* handles, int n) {
    if (!handles || n < 1) return fail("flh_rccl_init_all: bad arguments");
    if (rccl_load() != 0) return -1;
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n);
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail("flh_rccl_init_all: null handle");
        if (handles[i]->comm) return fail("flh_rccl_init_all: a handle already has a communicator");
        devs[i] = handles[i]->device;
    }
    NCCLC(g_rccl.CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        handles[i]->comm = comms[i];
        handles[i]->comm_size = n;
        handles[i]->comm_rank = i;
    }
    return 0;
}

void flh_rccl_destroy(flh_handle* h) {
    if (!h || !h->comm) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
    h->comm_size = 1;
    h->comm_rank = 0;
}
int flh_rccl_size(const flh_handle* h) { return h ? h->comm_size : 0; }
int flh_rccl_rank(const flh_handle* h) { return h ? h->comm_rank : -1; }

static int rccl_allreduce_publish(flh_handle* h, double seq) {
    if (device_tree(h, false)) {  // group totals (see device_tree): [kGranGroups][slots of this column count], summed over the ranks in place
        const int ncol = h->last_ext ? 12 : 6, nsl = flh::gram_slots_host(ncol) + 1;
        const int red = h->N ? gran_group_size(h->N) : 0;
        const int ng_own = red > 0 ? (flh::pass_blocks((int)h->N) + red - 1) / red : 0;
        NCCLC(g_rccl.AllReduce(h->gsum.p, h->gsum.p, (size_t)kGranGroups * nsl, ncclDouble, ncclSum, h->comm, h->stream));
        HIPC(flh::launch_publish_groups(h->gsum.p, ng_own, kGranGroups, nsl, ncol, h->h_gram, seq, h->stream));
        return 0;
    }
    NCCLC(g_rccl.AllReduce(h->gram.p, h->gram.p, 256, ncclDouble, ncclSum, h->comm, h->stream));
    HIPC(flh::launch_publish256(h->gram.p, h->h_gram, seq, h->stream));
    return 0;
}

// Single process, several devices: one h_share_model evaluation at `state` on every handle (each holds its shard of the
// scan / its slab of the map), the partial Gram blocks summed by RCCL, the result unpacked once.  Enqueues on all devices
// first (inside one RCCL group), then waits for rank 0's published block.
int flh_eval_group(flh_handle* const* handles, int n, const double state[FLH_NSTATE], int do_search, int ext, double HTH[144],
                   double HTh[12], int64_t* n_eff, double* total_residual) {
    if (!handles || n < 1 || !state || !HTH || !HTh) return fail("flh_eval_group: bad arguments");
    const StateDev s = make_state(state + 3, state + 0, state + 7, state + 11);
    if (handles[0] && handles[0]->peer_seg) {
        // peers (flh_peer_init_all): every handle's pass publishes to every handle's buffer; handle 0's host adds them up
        for (int i = 0; i < n; ++i)
            if (!handles[i] || handles[i]->peer_seg != handles[0]->peer_seg || handles[i]->peer_n != n || handles[i]->peer_rank != i)
                return fail("flh_eval_group: the handles are not the peers of one flh_peer_init_all, in its order");
        flh_handle* h0 = handles[0];
        const double seq = (double)(++h0->seq);
        for (int i = 0; i < n; ++i) {
            flh_handle* h = handles[i];
            h->seq = h0->seq;
            HIPC(hipSetDevice(h->device));
            if (h->N == 0) return fail("flh_eval_group: a scan shard may not be empty when the ranks exchange granules");
            if (enqueue_eval(h, s, do_search, ext, h->h_gram, seq, nullptr, true) != 0) return -1;
        }
        HIPC(hipSetDevice(h0->device));
        if (collect_granules(h0, seq, do_search, ext) != 0) return -1;
        h0->h_gram[255] = 0.0;
        flh_unpack_gram(h0->h_gram, HTH, HTh, n_eff, total_residual);
        return 0;
    }
    for (int i = 0; i < n; ++i)
        if (!handles[i] || !handles[i]->comm || handles[i]->comm_size != n) return fail("flh_eval_group: handles lack a common communicator (flh_rccl_init_all / flh_peer_init_all)");
    for (int i = 0; i < n; ++i) {
        flh_handle* h = handles[i];
        HIPC(hipSetDevice(h->device));
        if (enqueue_eval(h, s, do_search, ext, h->gram.p, 0.0, nullptr, false, true) != 0) return -1;
    }
    flh_handle* h0 = handles[0];
    const double seq = (double)(++h0->seq);
    bool tree = true;  // the granules' tree on every handle, or on none (the collective's size)
    for (int i = 0; i < n; ++i) tree = tree && device_tree(handles[i], false);
    const int ncol = ext ? 12 : 6, nsl = flh::gram_slots_host(ncol) + 1;
    NCCLC(g_rccl.GroupStart());
    for (int i = 0; i < n; ++i) {
        flh_handle* h = handles[i];
        if (tree) NCCLC(g_rccl.AllReduce(h->gsum.p, h->gsum.p, (size_t)kGranGroups * nsl, ncclDouble, ncclSum, h->comm, h->stream));
        else NCCLC(g_rccl.AllReduce(h->gram.p, h->gram.p, 256, ncclDouble, ncclSum, h->comm, h->stream));
    }
    NCCLC(g_rccl.GroupEnd());
    if (tree) {  // every handle's rows behind its own groups go back to zero (k_publish_groups); handle 0's block goes to the host
        for (int i = n - 1; i >= 0; --i) {
            flh_handle* h = handles[i];
            const int red = h->N ? gran_group_size(h->N) : 0;
            const int ng_own = red > 0 ? (flh::pass_blocks((int)h->N) + red - 1) / red : 0;
            HIPC(hipSetDevice(h->device));
            HIPC(flh::launch_publish_groups(h->gsum.p, ng_own, kGranGroups, nsl, ncol, h->h_gram, i == 0 ? seq : -1.0, h->stream));
        }
    } else {
        HIPC(hipSetDevice(h0->device));
        HIPC(flh::launch_publish256(h0->gram.p, h0->h_gram, seq, h0->stream));
    }
    HIPC(hipSetDevice(h0->device));
    const volatile double* flag = h0->h_gram + 255;
    uint64_t spins = 0;
    while (*flag != seq) {
        cpu_relax();
        if ((++spins & 0xFFFFFu) == 0 && hipStreamQuery(h0->stream) != hipErrorNotReady) {
            HIPC(hipStreamSynchronize(h0->stream));
            if (*flag != seq) return fail("flh_eval_group: result not published");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    h0->h_gram[255] = 0.0;
    flh_unpack_gram(h0->h_gram, HTH, HTh, n_eff, total_residual);
    return 0;
}

// flh_fetch_rows with a communicator: the rows of ALL ranks in rank order (the gain-form branch, esekfom.hpp:1715-1744, runs
// when the GLOBAL n_eff is below 23, so every rank holds at most 22 of them): one all-gather of a fixed-size record.
static int fetch_rows_gathered(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows) {
    constexpr int kMaxLocal = 64, kRec = 1 + kMaxLocal * 13;
    int64_t n_local = 0;
    if (fetch_rows_local(h, nullptr, nullptr, 0, &n_local) != 0) return -1;
    if (n_local > kMaxLocal) return fail("flh_fetch_rows: more rows on this rank than the gathered fetch carries (the information form needs none)");
    std::vector<double> lhx((size_t)std::max<int64_t>(n_local, 1) * 12), lhv((size_t)std::max<int64_t>(n_local, 1));
    if (n_local > 0 && fetch_rows_local(h, lhx.data(), lhv.data(), n_local, &n_local) != 0) return -1;
    const int G = h->comm_size;
    std::vector<double> rec((size_t)kRec, 0.0), all((size_t)kRec * G, 0.0);
    rec[0] = (double)n_local;
    for (int64_t k = 0; k < n_local; ++k) {
        for (int c = 0; c < 12; ++c) rec[1 + (size_t)k * 13 + c] = lhx[(size_t)c * n_local + k];
        rec[1 + (size_t)k * 13 + 12] = lhv[k];
    }
    HIPC(hipSetDevice(h->device));
    HIPC(h->gather_buf.reserve((size_t)kRec * (G + 1)));
    double* d_send = h->gather_buf.p;
    double* d_recv = h->gather_buf.p + kRec;
    HIPC(hipMemcpyAsync(d_send, rec.data(), sizeof(double) * kRec, hipMemcpyHostToDevice, h->stream));
    NCCLC(g_rccl.AllGather(d_send, d_recv, kRec, ncclDouble, h->comm, h->stream));
    HIPC(hipMemcpyAsync(all.data(), d_recv, sizeof(double) * kRec * G, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    int64_t n = 0;
    for (int r = 0; r < G; ++r) n += (int64_t)all[(size_t)r * kRec];
    *n_rows = n;
    if (!hx || !hv) return 0;
    if (cap < n) return fail("flh_fetch_rows: buffers too small");
    int64_t k = 0;
    for (int r = 0; r < G; ++r) {
        const double* a = all.data() + (size_t)r * kRec;
        const int64_t nr = (int64_t)a[0];
        for (int64_t j = 0; j < nr; ++j, ++k) {
            for (int c = 0; c < 12; ++c) hx[(size_t)c * n + k] = a[1 + (size_t)j * 13 + c];
            hv[k] = a[1 + (size_t)j * 13 + 12];
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU without a collective: peer-written granules (see include/fastlio_hip.h).  The sum the reference forms at
// esekfom.hpp:1784,1804 over all points is formed here by every host over (rank, group) granules.
// ---------------------------------------------------------------------------------------------
constexpr int kPeerRowsMax = 64, kPeerRec = 2 + kPeerRowsMax * 13;  // a rank's record of the gathered row fetch: {tag, n, rows}
static size_t peer_gran_bytes(int n) { return (size_t)n * 2 * (size_t)n * kGranSect * 16; }
// Behind the granule windows and the row records: the attach handshake of flh_peer_open.  A name can outlive a crashed run, and a
// rank > 0 that starts before rank 0 would map the stale segment (same size) while rank 0 unlinks it and creates another.  So an
// attach only counts once rank 0 -- which has just created THIS segment -- has echoed the fresh random token the rank left in it;
// a rank that gets no echo unmaps, opens the name again and retries.
struct PeerHello {
    std::atomic<uint64_t> hello[8];  // [rank]: a token of this attach attempt
    std::atomic<uint64_t> echo[8];   // [rank]: rank 0's copy of it
};
static size_t peer_rows_end(int n) { return peer_gran_bytes(n) + (size_t)n * kPeerRec * sizeof(double); }
static size_t peer_seg_bytes(int n) { return peer_rows_end(n) + sizeof(PeerHello); }
static uint64_t fresh_token() {
    std::random_device rd;
    uint64_t t = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 17);
    return t ? t : 1;
}
static double* peer_rows(const PeerSeg* sg, int src) { return reinterpret_cast<double*>((char*)sg->host + peer_gran_bytes(sg->n)) + (size_t)src * kPeerRec; }

static void peer_attach(flh_handle* h, PeerSeg* sg, int nranks, int rank) {
    const size_t window = 2 * (size_t)nranks * kGranSect * 2;  // doubles per destination rank: two parities x nranks sections
    h->peer_seg = sg;
    sg->refs++;
    h->peer_n = nranks;
    h->peer_rank = rank;
    for (int& g : h->sect_ng) g = 0;
    h->h_gran_own = h->h_gran;
    if (!sg->shm) sg->members[rank] = h;  // one process, several handles: fetch_rows_peers fills in every member's rows itself
    h->h_gran = reinterpret_cast<double*>(sg->host) + (size_t)rank * window;
    h->gran_owned = false;
    for (int d = 0; d < nranks; ++d) h->gran_dst[d] = reinterpret_cast<double*>(sg->dev) + (size_t)d * window;
    h->seq = 0;  // the ranks count their evaluations from the same origin
}

int flh_peer_open(flh_handle* h, const char* shm_name, int nranks, int rank) {
    if (!h || !shm_name) return fail("flh_peer_open: null argument");
    if (nranks < 1 || nranks > FLH_MAX_PEERS || rank < 0 || rank >= nranks) return fail("flh_peer_open: bad rank / nranks");
    if (h->comm) return fail("flh_peer_open: the handle has an RCCL communicator");
    if (h->peer_seg) return fail("flh_peer_open: the handle is already attached to peers");
    HIPC(hipSetDevice(h->device));
    const size_t bytes = peer_seg_bytes(nranks);
    const auto t0 = std::chrono::steady_clock::now();
    const auto deadline = std::chrono::seconds(30);
    void* m = MAP_FAILED;
    if (rank == 0) {
        (void)shm_unlink(shm_name);  // a stale segment of an earlier run
        int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fail(std::string("flh_peer_open: shm_open(create): ") + std::strerror(errno));
        if (ftruncate(fd, (off_t)bytes) != 0) {
            const std::string em = std::string("flh_peer_open: ftruncate: ") + std::strerror(errno);
            close(fd);
            (void)shm_unlink(shm_name);
            return fail(em);
        }
        m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { (void)shm_unlink(shm_name); return fail(std::string("flh_peer_open: mmap: ") + std::strerror(errno)); }
        // every other rank announces itself in THIS segment (ftruncate'd memory is zero-filled) and is answered
        PeerHello* hs = reinterpret_cast<PeerHello*>((char*)m + peer_rows_end(nranks));
        for (;;) {
            int answered = 0;
            for (int r = 1; r < nranks; ++r) {
                const uint64_t v = hs->hello[r].load(std::memory_order_acquire);
                if (v != 0) {
                    if (hs->echo[r].load(std::memory_order_relaxed) != v) hs->echo[r].store(v, std::memory_order_release);
                    ++answered;
                }
            }
            if (answered == nranks - 1) break;
            if (std::chrono::steady_clock::now() - t0 > deadline) {
                munmap(m, bytes);
                (void)shm_unlink(shm_name);
                return fail("flh_peer_open: not every rank attached to the segment within 30 s");
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    } else {
        for (;;) {  // rank 0's segment at its full size -- and rank 0 ALIVE in it: its echo of a token written just now
            int fd = shm_open(shm_name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                const bool sized = fstat(fd, &sb) == 0 && (size_t)sb.st_size == bytes;
                void* mm = sized ? mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
                close(fd);
                if (mm != MAP_FAILED) {
                    PeerHello* hs = reinterpret_cast<PeerHello*>((char*)mm + peer_rows_end(nranks));
                    const uint64_t tok = fresh_token();
                    hs->hello[rank].store(tok, std::memory_order_release);
                    const auto ta = std::chrono::steady_clock::now();
                    bool ok = false;
                    while (std::chrono::steady_clock::now() - ta < std::chrono::seconds(2)) {  // (a live rank 0 answers within a millisecond)
                        if (hs->echo[rank].load(std::memory_order_acquire) == tok) { ok = true; break; }
                        std::this_thread::sleep_for(std::chrono::microseconds(100));
                    }
                    if (ok) { m = mm; break; }
                    munmap(mm, bytes);  // nobody answered: a segment left behind by an earlier run; rank 0 will replace it
                }
            }
            if (std::chrono::steady_clock::now() - t0 > deadline) return fail("flh_peer_open: rank 0's segment did not appear");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    hipError_t e = hipHostRegister(m, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    void* dev = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, m, 0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        munmap(m, bytes);
        if (rank == 0) (void)shm_unlink(shm_name);
        return fail(std::string("flh_peer_open: hipHostRegister: ") + hipGetErrorString(e));
    }
    PeerSeg* sg = new PeerSeg();
    sg->host = m; sg->dev = dev; sg->bytes = bytes; sg->shm = true; sg->creator = rank == 0; sg->name = shm_name; sg->n = nranks;
    peer_attach(h, sg, nranks, rank);
    return 0;
}

int flh_peer_init_all(flh_handle* const* handles, int n) {
    if (!handles || n < 1 || n > FLH_MAX_PEERS) return fail("flh_peer_init_all: bad arguments");
    for (int i = 0; i < n; ++i) {
        if (!handles[i]) return fail("flh_peer_init_all: null handle");
        if (handles[i]->comm || handles[i]->peer_seg) return fail("flh_peer_init_all: a handle already has a communicator / peers");
    }
    HIPC(hipSetDevice(handles[0]->device));
    const size_t bytes = peer_seg_bytes(n);
    void* m = nullptr;
    HIPC(hipHostMalloc(&m, bytes, hipHostMallocPortable | hipHostMallocMapped));
    std::memset(m, 0, bytes);
    PeerSeg* sg = new PeerSeg();
    sg->host = m; sg->dev = m; sg->bytes = bytes; sg->n = n;
    for (int i = 0; i < n; ++i) peer_attach(handles[i], sg, n, i);
    return 0;
}

void flh_peer_close(flh_handle* h) {
    if (!h || !h->peer_seg) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    PeerSeg* sg = h->peer_seg;
    h->peer_seg = nullptr;
    h->h_gran = h->h_gran_own;
    h->gran_owned = true;
    h->gran_dst[0] = h->h_gran;
    h->peer_n = 1;
    h->peer_rank = 0;
    for (int& g : h->sect_ng) g = 0;
    // the handle's own buffer still holds granules tagged with sequence numbers of evaluations BEFORE the attach (which reset the
    // count): cleared, so that a later evaluation whose number happens to equal one of them cannot take them for its own
    std::memset(h->h_gran, 0, 2 * kGranSect * 16);
    for (int i = 0; i < sg->n && i < FLH_MAX_PEERS; ++i)
        if (sg->members[i] == h) sg->members[i] = nullptr;
    if (--sg->refs == 0) {
        if (sg->shm) {
            (void)hipHostUnregister(sg->host);
            munmap(sg->host, sg->bytes);
            if (sg->creator) (void)shm_unlink(sg->name.c_str());
        } else {
            (void)hipHostFree(sg->host);
        }
        delete sg;
    }
}
int flh_peer_size(const flh_handle* h) { return h ? h->peer_n : 0; }
int flh_peer_rank(const flh_handle* h) { return h ? h->peer_rank : -1; }
// Developer builds (-DFLH_BOUNDS, tools/variant.py): the bounds records of the four kernel translation units, 5 words each
// {violations, site of the first one, its index, the capacity, its workgroup}; all zero in the product (nothing is checked there).
int flh_debug_bounds(flh_handle* h, uint64_t out[20]) {
    if (!h || !out) return fail("flh_debug_bounds: null argument");
    for (int i = 0; i < 20; ++i) out[i] = 0;
#ifdef FLH_BOUNDS
    HIPC(hipSetDevice(h->device));
    HIPC(hipDeviceSynchronize());
    unsigned long long r[5];
    flh::bounds_read_kernels(r); for (int i = 0; i < 5; ++i) out[i] = r[i];
    flh::bounds_read_pass(r); for (int i = 0; i < 5; ++i) out[5 + i] = r[i];
    flh::bounds_read_mapinc(r); for (int i = 0; i < 5; ++i) out[10 + i] = r[i];
    flh::bounds_read_scanprep(r); for (int i = 0; i < 5; ++i) out[15 + i] = r[i];
    flh::bounds_read_stage(r);  // (the staging kernels, flh_stage.hip: reported with the scan front end's record)
    if (r[0]) {
        if (out[15] == 0) for (int i = 1; i < 5; ++i) out[15 + i] = r[i];
        out[15] += r[0];
    }
    return 1;
#else
    return 0;
#endif
}
// Developer builds (-DFLH_PASS_STAMPS): the per-wave phase stamps of the last k_pass launch, 12 words per wave (8 stamps in 100 MHz ticks,
// HW_ID, XCC_ID, longest candidate list, open queries).
int flh_debug_pass_stamps(flh_handle* h, uint64_t* out, size_t words) {
    if (!h || !out) return fail("flh_debug_pass_stamps: null argument");
#ifdef FLH_PASS_STAMPS
    HIPC(hipSetDevice(h->device));
    HIPC(hipDeviceSynchronize());
    flh::pass_stamps_read(reinterpret_cast<unsigned long long*>(out), words);
    return 1;
#else
    (void)words;
    return 0;
#endif
}
int flh_get_pass_stats(const flh_handle* h, uint64_t out[4]) {
    if (!h || !out) return fail("flh_get_pass_stats: null argument");
    out[0] = h->n_search_pass; out[1] = h->n_one_launch; out[2] = h->n_second_stage; out[3] = h->n_nosearch_pass;
    return 0;
}

// flh_fetch_rows among peers: every rank's host leaves its rows (at most kPeerRowsMax: the gain-form branch runs when the GLOBAL
// n_eff is below 23) in the shared segment, tagged with the sequence number of the evaluation they belong to, and reads the
// others'.  Plain host stores and loads; no device work.
// this rank's rows of the evaluation `tag` into its record of the shared segment
static int peer_post_rows(flh_handle* h, double tag) {
    int64_t n_local = 0;
    if (fetch_rows_local(h, nullptr, nullptr, 0, &n_local) != 0) return -1;
    if (n_local > kPeerRowsMax) return fail("flh_fetch_rows: more rows on this rank than the gathered fetch carries (the information form needs none)");
    std::vector<double> lhx((size_t)std::max<int64_t>(n_local, 1) * 12), lhv((size_t)std::max<int64_t>(n_local, 1));
    if (n_local > 0 && fetch_rows_local(h, lhx.data(), lhv.data(), n_local, &n_local) != 0) return -1;
    double* mine = peer_rows(h->peer_seg, h->peer_rank);
    if (mine[0] != tag) {  // (a second fetch after the same evaluation finds the record in place)
        mine[1] = (double)n_local;
        for (int64_t k = 0; k < n_local; ++k) {
            for (int c = 0; c < 12; ++c) mine[2 + (size_t)k * 13 + c] = lhx[(size_t)c * n_local + k];
            mine[2 + (size_t)k * 13 + 12] = lhv[k];
        }
        std::atomic_thread_fence(std::memory_order_release);
        reinterpret_cast<std::atomic<double>*>(mine)->store(tag, std::memory_order_release);
    }
    return 0;
}
static int fetch_rows_peers(flh_handle* h, double* hx, double* hv, int64_t cap, int64_t* n_rows) {
    const double tag = (double)h->seq;
    if (!h->peer_seg->shm) {
        // one process, several handles (flh_peer_init_all + flh_eval_group): nobody else will write the other handles' records --
        // this thread drives them all -- so they are filled in here, every member's rows of its own last evaluation
        for (int r = 0; r < h->peer_n; ++r) {
            flh_handle* hm = h->peer_seg->members[r];
            if (!hm || !hm->have_eval) return fail("flh_fetch_rows: a peer handle of this process has no evaluation to take rows from");
            if (peer_post_rows(hm, tag) != 0) return -1;
        }
    } else if (peer_post_rows(h, tag) != 0) {
        return -1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int64_t n = 0;
    for (int r = 0; r < h->peer_n; ++r) {
        const double* rec = peer_rows(h->peer_seg, r);
        while (reinterpret_cast<const std::atomic<double>*>(rec)->load(std::memory_order_acquire) != tag) {
            cpu_relax();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return fail("flh_fetch_rows: timed out waiting for a peer's rows");
        }
        n += (int64_t)rec[1];
    }
    *n_rows = n;
    if (!hx || !hv) return 0;
    if (cap < n) return fail("flh_fetch_rows: buffers too small");
    int64_t k = 0;
    for (int r = 0; r < h->peer_n; ++r) {
        const double* rec = peer_rows(h->peer_seg, r);
        const int64_t nr = (int64_t)rec[1];
        for (int64_t j = 0; j < nr; ++j, ++k) {
            for (int c = 0; c < 12; ++c) hx[(size_t)c * n + k] = rec[2 + (size_t)j * 13 + c];
            hv[k] = rec[2 + (size_t)j * 13 + 12];
        }
    }
    return 0;
}

// Map partitioned over the ranks (BASELINE configs[4]): this handle's map holds one slab of the world plus a halo of at least
// sqrt(max_sqdist) on either side, every rank holds the whole scan, and a query is searched -- and thereafter fitted -- only
// by the rank whose half-open interval [lo, hi) of world coordinate `axis` contains it; the others leave its flag at 0.
// Every query has exactly one owner when the ranks' intervals tile the axis.  axis < 0 switches the restriction off.
int flh_set_owned_interval(flh_handle* h, int axis, float lo, float hi) {
    if (!h) return fail("flh_set_owned_interval: null handle");
    if (axis > 2) return fail("flh_set_owned_interval: axis must be 0, 1, 2 or negative");
    if (axis >= 0 && !(lo < hi)) return fail("flh_set_owned_interval: empty interval");
    h->own_axis = axis;
    h->own_lo = axis < 0 ? -INFINITY : lo;
    h->own_hi = axis < 0 ? INFINITY : hi;
    h->searched_once = false;
    return 0;
}

}  // extern "C"
