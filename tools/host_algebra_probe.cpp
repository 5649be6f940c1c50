// Developer tool (CPU only): what the 23x23 algebra of a pass costs the host.  The mirrored filter
// (include/fastlio_amd/esekfom.hpp) is driven by a measurement model that returns fixed normal equations at once, so an update's
// wall time IS its host algebra; run with the one-piece model (everything after the measurement) and with the two-halves model
// (the measurement-free part in front of the "wait": what flh_eval_begin / _end overlap with the device pass).
//   g++ -O3 -std=c++17 -mavx2 -ffp-contract=off -Iinclude tools/host_algebra_probe.cpp -o /tmp/host_algebra_probe && /tmp/host_algebra_probe
#include <chrono>
#include <cstdio>
#include <cstring>

#include "fastlio_amd/esekfom.hpp"
#include "fastlio_amd/use-ikfom.hpp"

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
using clk = std::chrono::steady_clock;
static double g_in_model_us = 0;  // time between the filter's call of the first half and its call of the second (split model only)
static clk::time_point g_begin_t;

static void fake_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {
    std::memset(d.HTH, 0, sizeof(d.HTH));
    std::memset(d.HTh, 0, sizeof(d.HTh));
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) d.HTH[i * 12 + j] = (i == j ? 4.0e6 + 1.0e5 * i : 3.0e4 / (1 + i + j));
        d.HTh[i] = 1.2e3 * (i + 1) - 9.0e5 * (s.pos[i % 3] - 0.3);
    }
    d.n_eff = 60000;
    d.total_residual = 17.0;
    d.has_normal_eq = true;
    d.valid = true;
}
static void fake_begin(state_ikfom&, esekfom::dyn_share_datastruct<double>&) { g_begin_t = clk::now(); }
static void fake_model_b(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {
    g_in_model_us += std::chrono::duration<double, std::micro>(clk::now() - g_begin_t).count();
    fake_model(s, d);
}

static double run(bool split, int reps, int* passes_out) {
    kf_t kf;
    double epsi[23];
    for (double& e : epsi) e = 0.001;
    kf.init_dyn_share(get_f, df_dx, df_dw, split ? fake_model_b : fake_model, 3, epsi);
    state_ikfom s0 = kf.get_x();
    s0.pos[0] = 0.3; s0.pos[1] = -0.2; s0.pos[2] = 1.0;
    s0.rot.x = 0.01; s0.rot.y = -0.02; s0.rot.z = 0.03; s0.rot.w = 0.99930;
    kf_t::cov c = kf_t::cov::Identity();
    for (int i = 0; i < 23; ++i) c(i, i) = 0.01 + 0.001 * i;
    c(0, 4) = c(4, 0) = 0.002;
    double st = 0;
    long passes = 0;
    const auto t0 = clk::now();
    for (int r = 0; r < reps; ++r) {
        state_ikfom s = s0;
        s.pos[0] += 1e-3 * (r % 7);
        kf.change_x(s);
        kf_t::cov cc = c;
        kf.change_P(cc);
        kf.update_iterated_dyn_share_modified(0.001, st);
        passes += kf.last_stats().passes;
    }
    const double us = std::chrono::duration<double, std::micro>(clk::now() - t0).count();
    *passes_out = (int)(passes / reps);
    return us / (double)passes;
}

int main() {
    esekfom::register_split_model(reinterpret_cast<void*>(&fake_model_b), reinterpret_cast<void*>(&fake_begin));
    int p = 0;
    (void)run(false, 2000, &p);
    const double one = run(false, 20000, &p);
    std::printf("one-piece model : %.2f us of host algebra per pass (%d passes per update)\n", one, p);
    g_in_model_us = 0;
    int p2 = 0;
    (void)run(true, 2000, &p2);
    g_in_model_us = 0;
    const double two = run(true, 20000, &p2);
    const double hidden = g_in_model_us / (20000.0 * p2);
    std::printf("two-halves model: %.2f us per pass, of which %.2f us lie between the two halves (beside the device pass) and %.2f us behind the "
                "measurement (on the critical path)\n", two, hidden, two - hidden);
    return 0;
}
