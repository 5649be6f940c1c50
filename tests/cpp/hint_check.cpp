// CPU check of the hints the mirrored filter gives its measurement model (include/fastlio_amd/esekfom.hpp: dyn_share_datastruct::next_pass,
// the finish hook): before every pass it says what the pass AFTER it will probably be -- a no-search pass unless nothing follows
// (the last pass) or a search is certain (esekfom.hpp:1829-1832: after the last but one pass when no step has converged) -- and when
// the update ends although a no-search pass had been announced, the model is told (a kernel enqueued ahead is released).  Driven
// through the schedules the update can take: never converging, converging at once, converging in the middle, an invalid pass.
#include <cstdio>
#include <cstring>
#include <vector>

#include "fastlio_amd/esekfom.hpp"
#include "fastlio_amd/use-ikfom.hpp"

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
struct Rec { int hint; bool search; };
static std::vector<Rec> g_pass;
static int g_finish = 0, g_call = 0;
static std::vector<double> g_scale;  // per pass: how large the residual is (0: the step converges); < 0: the pass is invalid

static void model(state_ikfom&, esekfom::dyn_share_datastruct<double>& d, void*) {
    const int k = g_call++;
    const double sc = k < (int)g_scale.size() ? g_scale[k] : 0.0;
    std::memset(d.HTH, 0, sizeof(d.HTH));
    std::memset(d.HTh, 0, sizeof(d.HTh));
    for (int i = 0; i < 6; ++i) { d.HTH[i * 12 + i] = 5000.0; d.HTh[i] = sc * 50.0 * (i + 1); }
    d.n_eff = 4000;
    d.total_residual = 1.0;
    d.has_normal_eq = true;
    d.valid = sc >= 0.0;
}
static void begin(state_ikfom&, esekfom::dyn_share_datastruct<double>& d, void*) { g_pass.push_back({d.next_pass, d.converge}); }
static void finish(void*) { ++g_finish; }

static int run(const char* name, std::vector<double> scale, int max_iter, std::vector<int> want_search, std::vector<int> want_hint, int want_finish) {
    g_pass.clear(); g_finish = 0; g_call = 0; g_scale = scale;
    kf_t kf;
    double epsi[23];
    for (double& e : epsi) e = 0.001;
    kf.init_dyn_share(get_f, df_dx, df_dw, static_cast<kf_t::measurementModel_dyn_share_ctx*>(model), max_iter, epsi, nullptr);
    kf.set_meas_begin(begin);
    kf.set_meas_finish(finish);
    kf_t::cov c = kf_t::cov::Identity();
    for (int i = 0; i < 23; ++i) c(i, i) = 0.01;
    kf.change_P(c);
    double st = 0;
    kf.update_iterated_dyn_share_modified(0.001, st);
    int bad = 0;
    if (g_pass.size() != want_search.size()) ++bad;
    for (size_t k = 0; k < g_pass.size() && k < want_search.size(); ++k)
        if ((int)g_pass[k].search != want_search[k] || g_pass[k].hint != want_hint[k]) ++bad;
    if (g_finish != want_finish) ++bad;
    // whatever the schedule: no "nothing follows" before a pass (that hint releases a waiting kernel: it belongs to the end), and an
    // announced no-search pass either comes, or a search comes instead, or the update ends and says so
    for (size_t k = 0; k < g_pass.size(); ++k) {
        if (g_pass[k].hint == esekfom::kNextNone) ++bad;
        if (g_pass[k].hint == esekfom::kNextNoSearch && k + 1 == g_pass.size() && g_finish != 1) ++bad;
    }
    std::printf("%-34s passes", name);
    for (const Rec& r : g_pass) std::printf(" %c%d", r.search ? 'S' : 'n', r.hint);
    std::printf("  finish %d  %s\n", g_finish, bad ? "FAILED" : "ok");
    return bad;
}

int main() {
    const int U = esekfom::kNextUnknown, N = esekfom::kNextNoSearch;
    int bad = 0;
    // max_iter 3 (the node's NUM_MAX_ITERATIONS): passes i = -1, 0, 1, 2
    bad += run("never converges (S n n S)", {1, 1, 1, 1}, 3, {1, 0, 0, 1}, {N, N, U, U}, 0);
    bad += run("converges at once, twice (S S)", {0, 0, 0, 0}, 3, {1, 1}, {N, N}, 1);
    bad += run("second step converges (S n S n)", {1, 0, 1, 1}, 3, {1, 0, 1, 0}, {N, N, N, U}, 0);
    bad += run("second and third converge (S n S)", {1, 0, 0, 1}, 3, {1, 0, 1}, {N, N, N}, 1);
    bad += run("first pass invalid", {-1, 1, 1, 1}, 3, {1, 1, 0, 1}, {N, N, U, U}, 0);
    bad += run("max_iter 1", {1, 1}, 1, {1, 1}, {U, U}, 0);
    std::printf(bad ? "FAILED\n" : "hints as specified\n");
    return bad ? 1 : 0;
}
