// CPU check of flh_esekf_run_scans' staging schedule (fast_lio_amd/csrc/flh_esekf.cpp): the C ABI underneath is replaced by fakes
// that record what is asked of it.  Two scans are staged ahead of the one being updated when the ring has three or more slots
// (one with two), every scan is handed to the staging thread exactly once, in order, into slot index % ring, never past what the
// call may stage -- also across the two-call pattern bench.py uses (warm-up with FLH_RUN_STAGE_NEXT, measurement with
// FLH_RUN_FIRST_STAGED) -- and a scan is activated only after it has been staged.
#include <cstdio>
#include <vector>

#include "../../fast_lio_amd/csrc/flh_esekf.cpp"

struct flh_handle { int dummy; };
struct Ev { int kind; int slot; const void* pts; };  // kind 0: stage, 1: activate
static std::vector<Ev> g_ev;
extern "C" {
const char* flh_last_error(void) { return "fake"; }
int flh_scan_stage_async(flh_handle*, int slot, const void* pts, size_t, size_t) { g_ev.push_back({0, slot, pts}); return 0; }
int flh_scan_wait(flh_handle*, int) { return 0; }
int flh_scan_activate(flh_handle*, int slot) { g_ev.push_back({1, slot, nullptr}); return 0; }
int flh_eval_expect_next(flh_handle*, int) { return 0; }
int flh_eval_begin(flh_handle*, const double*, const double*, const double*, const double*, int, int) { return 0; }
int flh_eval_end(flh_handle*, double* HTH, double* HTh, int64_t* n_eff, double* tr) {
    for (int i = 0; i < 144; ++i) HTH[i] = 0;
    for (int i = 0; i < 12; ++i) HTh[i] = 0;
    *n_eff = 0;  // "No Effective Points": every pass invalid, the update returns after its passes -- the schedule is what is tested
    *tr = 0;
    return 0;
}
int flh_fetch_rows(flh_handle*, double*, double*, int64_t, int64_t* n) { *n = 0; return 0; }
int flh_map_incremental(flh_handle*, const double*, double, int, int, uint32_t*, uint32_t*) { return 0; }
}

static int check(int ring, int n_jobs, int64_t warm, int64_t steps) {
    flh_handle h{0};
    flh_esekf* e = flh_esekf_create(&h, 3, nullptr, 0);
    std::vector<float> pts((size_t)n_jobs * 3, 0.f);
    std::vector<double> x(26, 0.0), P(23 * 23, 0.0);
    x[6] = 1.0; x[10] = 1.0; x[25] = -9.81;
    for (int i = 0; i < 23; ++i) P[i * 23 + i] = 0.01;
    std::vector<flh_scan_job> jobs(n_jobs);
    for (int k = 0; k < n_jobs; ++k) jobs[k] = flh_scan_job{&pts[3 * k], 12, 1, x.data(), P.data(), -1};
    g_ev.clear();
    int bad = 0;
    // bench.py's pattern: the warm-up stages one scan past its end, the measurement starts from there
    if (flh_esekf_run_scans(e, jobs.data(), n_jobs, 0, warm, ring, 0.001, 0, 0.5, warm > 0 ? FLH_RUN_STAGE_NEXT : 0, nullptr, nullptr, nullptr) != 0) ++bad;
    const size_t split = g_ev.size();
    if (flh_esekf_run_scans(e, jobs.data(), n_jobs, warm, steps, ring, 0.001, 0, 0.5, warm > 0 ? FLH_RUN_FIRST_STAGED : 0, nullptr, nullptr, nullptr) != 0) ++bad;
    const int64_t total = warm + steps, ahead = ring >= 3 ? 2 : 1;
    int64_t staged = 0, activated = 0;  // scans [0, staged) staged, [0, activated) activated
    for (size_t k = 0; k < g_ev.size(); ++k) {
        const Ev& ev = g_ev[k];
        if (ev.kind == 0) {
            if (ev.slot != (int)(staged % ring) || ev.pts != jobs[staged % n_jobs].pts) { std::printf("  stage #%lld went to slot %d\n", (long long)staged, ev.slot); ++bad; }
            // never past the end of the call that stages it (the warm-up may stage exactly one scan of the next call)
            const int64_t limit = k < split ? warm + 1 : total;
            if (staged >= limit) { std::printf("  scan %lld staged by a call that may not\n", (long long)staged); ++bad; }
            if (staged > activated + ahead) { std::printf("  scan %lld staged more than %lld ahead\n", (long long)staged, (long long)ahead); ++bad; }
            ++staged;
        } else {
            if (ev.slot != (int)(activated % ring)) { std::printf("  activation #%lld of slot %d\n", (long long)activated, ev.slot); ++bad; }
            if (activated >= staged) { std::printf("  scan %lld activated before it was staged\n", (long long)activated); ++bad; }
            // in the steady stream the scans ahead are in flight when a scan is activated
            const int64_t call_end = k < split ? warm + 1 : total;
            const int64_t want = activated + 1 + ahead < call_end ? activated + 1 + ahead : call_end;
            if (staged < want) { std::printf("  scan %lld activated with only %lld staged (want %lld)\n", (long long)activated, (long long)staged, (long long)want); ++bad; }
            ++activated;
        }
    }
    if (staged != total || activated != total) { std::printf("  staged %lld activated %lld of %lld\n", (long long)staged, (long long)activated, (long long)total); ++bad; }
    flh_esekf_destroy(e);
    std::printf("ring %d, %d jobs, %lld + %lld scans: %s\n", ring, n_jobs, (long long)warm, (long long)steps, bad ? "FAILED" : "ok");
    return bad;
}

int main() {
    int bad = 0;
    bad += check(4, 128, 5, 20);   // the driver's command
    bad += check(4, 7, 30, 300);   // jobs cycled
    bad += check(3, 5, 2, 9);
    bad += check(2, 5, 3, 8);      // two slots: one ahead
    bad += check(4, 6, 0, 10);     // one call, nothing staged before
    bad += check(16, 3, 1, 1);
    std::printf(bad ? "FAILED\n" : "every scan staged once, in order, two ahead, never past the call's end\n");
    return bad ? 1 : 0;
}
