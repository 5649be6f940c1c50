// CPU check of the two-halves measurement model in the mirrored esekf (include/fastlio_amd/esekfom.hpp): a model whose first half
// is registered (esekfom::register_split_model / set_meas_begin) must give the very bits of the same model called in one piece --
// the filter only moves the covariance projection in front of the wait -- and an INVALID measurement on a pass
// must leave the covariance exactly as the one-piece flow leaves it.
#include <cstdio>
#include <cstring>

#include "fastlio_amd/esekfom.hpp"
#include "fastlio_amd/use-ikfom.hpp"

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
static int g_begins = 0, g_calls = 0, g_invalid_on = -1;

static void fake_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {
    // a fixed, well-conditioned information matrix and a residual that depends on the state, so that the passes differ
    const int call = g_calls++;
    std::memset(d.HTH, 0, sizeof(d.HTH));
    std::memset(d.HTh, 0, sizeof(d.HTh));
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) d.HTH[i * 12 + j] = (i == j ? 4000.0 + 100.0 * i : 30.0 / (1 + i + j));
        d.HTh[i] = 12.0 * (i + 1) - 900.0 * s.pos[i % 3] + 40.0 * s.rot.x * (i >= 3);
    }
    d.n_eff = 5000;
    d.total_residual = 17.0;
    d.has_normal_eq = true;
    d.valid = call != g_invalid_on;
}
static void fake_begin(state_ikfom&, esekfom::dyn_share_datastruct<double>&) { ++g_begins; }
static void fake_model_b(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) { fake_model(s, d); }  // a second address: registered split

static void run(bool split, int invalid_on, double x[26], double P[23 * 23], int* passes) {
    g_calls = 0;
    g_invalid_on = invalid_on;
    kf_t kf;
    double epsi[23];
    for (double& e : epsi) e = 0.001;
    kf.init_dyn_share(get_f, df_dx, df_dw, split ? fake_model_b : fake_model, 4, epsi);
    state_ikfom s = kf.get_x();
    s.pos[0] = 0.3; s.pos[1] = -0.2; s.pos[2] = 1.0;
    s.rot.x = 0.01; s.rot.y = -0.02; s.rot.z = 0.03; s.rot.w = 0.99930;
    kf.change_x(s);
    kf_t::cov c = kf_t::cov::Identity();
    for (int i = 0; i < 23; ++i) c(i, i) = 0.01 + 0.001 * i;
    c(0, 4) = c(4, 0) = 0.002;
    c(3, 21) = c(21, 3) = 0.0005;
    kf.change_P(c);
    double st = 0;
    kf.update_iterated_dyn_share_modified(0.001, st);
    kf.get_x().to_flat(x);
    std::memcpy(P, kf.get_P().a, sizeof(double) * 23 * 23);
    *passes = kf.last_stats().passes;
}

int main() {
    esekfom::register_split_model(reinterpret_cast<void*>(&fake_model_b), reinterpret_cast<void*>(&fake_begin));
    int bad = 0;
    for (int invalid_on : {-1, 0, 1, 3}) {
        double x0[26], P0[529], x1[26], P1[529];
        int p0 = 0, p1 = 0;
        g_begins = 0;
        run(false, invalid_on, x0, P0, &p0);
        if (g_begins != 0) { std::printf("one-piece model: begin called\n"); ++bad; }
        run(true, invalid_on, x1, P1, &p1);
        if (g_begins != p1) { std::printf("split model: %d begins for %d passes\n", g_begins, p1); ++bad; }
        if (p0 != p1 || std::memcmp(x0, x1, sizeof(x0)) != 0 || std::memcmp(P0, P1, sizeof(P0)) != 0) {
            std::printf("invalid_on=%d: split and one-piece flows differ (passes %d / %d)\n", invalid_on, p0, p1);
            ++bad;
        }
    }
    std::printf(bad ? "FAILED\n" : "split model: identical bits, %d checks\n", 4);
    return bad ? 1 : 0;
}
