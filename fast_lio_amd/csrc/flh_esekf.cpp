// flh_esekf.cpp -- layer 2 of the C ABI: a C binding of the C++ host filter in
// include/fastlio_amd/esekfom.hpp (the mirror of esekfom::esekf<state_ikfom,12,input_ikfom>), so that
// ctypes / C callers run exactly the code a C++ node would.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "../../include/fastlio_amd/esekfom.hpp"
#include "../../include/fastlio_amd/h_share_model.hpp"
#include "../../include/fastlio_amd/use-ikfom.hpp"
#include "../../include/fastlio_hip.h"

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;

struct flh_esekf {
    kf_t kf;
    fastlio_amd::HShareContext gpu_ctx;
    flh_meas_fn user_h = nullptr;
    void* user_ctx = nullptr;
    std::string err;
};

// adapter: user-supplied C measurement model -> dyn_share_datastruct
static void user_model_adapter(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d, void* ctx) {
    flh_esekf* e = static_cast<flh_esekf*>(ctx);
    double x[FLH_NSTATE];
    s.to_flat(x);
    flh_meas m;
    std::memset(&m, 0, sizeof(m));
    m.valid = 1;
    e->user_h(e->user_ctx, x, d.converge ? 1 : 0, &m);
    d.valid = m.valid != 0;
    d.n_eff = m.n_eff;
    d.total_residual = m.total_residual;
    d.has_normal_eq = m.has_normal_eq != 0;
    if (m.has_normal_eq) {
        std::memcpy(d.HTH, m.HTH, sizeof(m.HTH));
        std::memcpy(d.HTh, m.HTh, sizeof(m.HTh));
    }
    if (m.h_x && m.h && m.n_eff > 0) {
        d.h_x.assign(m.h_x, m.h_x + (size_t)m.n_eff * 12);
        d.h.assign(m.h, m.h + (size_t)m.n_eff);
        if (m.n_eff < state_ikfom::DOF) d.has_normal_eq = false;  // gain form works on rows
    } else {
        d.h_x.clear();
        d.h.clear();
        if (d.valid && m.n_eff < state_ikfom::DOF && m.n_eff > 0) d.valid = false;  // rows required but absent
    }
    if (m.n_eff < 1) d.valid = false;
}
static void gpu_model_adapter(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d, void* ctx) {
    flh_esekf* e = static_cast<flh_esekf*>(ctx);
    fastlio_amd::h_share_model(s, d, &e->gpu_ctx);
}
static void gpu_begin_adapter(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d, void* ctx) {
    flh_esekf* e = static_cast<flh_esekf*>(ctx);
    fastlio_amd::h_share_model_begin(s, d, &e->gpu_ctx);
}
static void gpu_finish_adapter(void* ctx) {
    flh_esekf* e = static_cast<flh_esekf*>(ctx);
    fastlio_amd::h_share_model_finish(&e->gpu_ctx);
}

extern "C" {

flh_esekf* flh_esekf_create(flh_handle* handle, int maximum_iter, const double limit[FLH_NDOF], int extrinsic_est_en) {
    flh_esekf* e = new flh_esekf();
    e->gpu_ctx.handle = handle;
    e->gpu_ctx.extrinsic_est_en = extrinsic_est_en != 0;
    double lim[FLH_NDOF];
    for (int i = 0; i < FLH_NDOF; ++i) lim[i] = limit ? limit[i] : 0.001;  // epsi, laserMapping.cpp:826-827
    e->kf.init_dyn_share(get_f, df_dx, df_dw, static_cast<kf_t::measurementModel_dyn_share_ctx*>(gpu_model_adapter), maximum_iter, lim, e);
    if (handle) {
        e->kf.set_meas_begin(gpu_begin_adapter);  // the pass runs on the device while the filter projects the covariance
        e->kf.set_meas_finish(gpu_finish_adapter);
    }
    return e;
}
void flh_esekf_destroy(flh_esekf* e) { delete e; }
void flh_esekf_set_meas_model(flh_esekf* e, flh_meas_fn h, void* ctx) {
    if (!e) return;
    e->user_h = h;
    e->user_ctx = ctx;
    e->kf.set_meas_model(static_cast<kf_t::measurementModel_dyn_share_ctx*>(h ? user_model_adapter : gpu_model_adapter), e);
    if (!h && e->gpu_ctx.handle) {
        e->kf.set_meas_begin(gpu_begin_adapter);
        e->kf.set_meas_finish(gpu_finish_adapter);
    }
}
void flh_esekf_change_x(flh_esekf* e, const double x[FLH_NSTATE]) {
    state_ikfom s = e->kf.get_x();
    s.from_flat(x);
    e->kf.change_x(s);
}
void flh_esekf_change_P(flh_esekf* e, const double P[FLH_NDOF * FLH_NDOF]) {
    kf_t::cov c;
    std::memcpy(c.a, P, sizeof(c.a));
    e->kf.change_P(c);
}
void flh_esekf_get_x(const flh_esekf* e, double x[FLH_NSTATE]) { e->kf.get_x().to_flat(x); }
void flh_esekf_get_P(const flh_esekf* e, double P[FLH_NDOF * FLH_NDOF]) { std::memcpy(P, e->kf.get_P().a, sizeof(double) * FLH_NDOF * FLH_NDOF); }
void flh_esekf_predict(flh_esekf* e, double dt, const double Q[144], const double acc[3], const double gyro[3]) {
    kf_t::processnoisecovariance q;
    std::memcpy(q.a, Q, sizeof(q.a));
    input_ikfom in;
    for (int i = 0; i < 3; ++i) { in.acc[i] = acc[i]; in.gyro[i] = gyro[i]; }
    e->kf.predict(dt, q, in);
}
int flh_esekf_update(flh_esekf* e, double R, flh_update_stats* st) {
    if (!e) return -1;
    e->err.clear();
    double solve_time = 0;
    try {
        e->kf.update_iterated_dyn_share_modified(R, solve_time);
    } catch (const std::exception& ex) {
        if (e->gpu_ctx.handle) (void)flh_eval_expect_next(e->gpu_ctx.handle, FLH_NEXT_NONE);  // (a pass enqueued ahead is released)
        e->err = ex.what();
        return -1;
    }
    if (st) {
        const kf_t::update_stats& s = e->kf.last_stats();
        st->passes = s.passes;
        st->searches = s.searches;
        st->returned_in_loop = s.returned_in_loop;
        for (int i = 0; i < 8; ++i) { st->n_eff[i] = s.n_eff[i]; st->pass_search[i] = s.pass_search[i]; st->pass_ms[i] = s.pass_ms[i]; }
        st->h_ms = s.h_ms;
        st->solve_ms = s.solve_ms;
    }
    return 0;
}

const char* flh_esekf_last_error(const flh_esekf* e) { return e ? e->err.c_str() : "null filter"; }

// One scan of the node's main loop in one call: feats_down_body := staged slot (slot < 0: keep the active scan), the
// propagated state and covariance from the IMU front end, then update_iterated_dyn_share_modified (:960).
int flh_esekf_update_scan(flh_esekf* e, int slot, const double x[FLH_NSTATE], const double P[FLH_NDOF * FLH_NDOF], double R,
                          flh_update_stats* st) {
    if (!e) return -1;
    if (slot >= 0) {
        if (!e->gpu_ctx.handle) {
            e->err = "flh_esekf_update_scan: filter has no device handle";
            return -1;
        }
        if (flh_scan_activate(e->gpu_ctx.handle, slot) != 0) {
            e->err = std::string("flh_scan_activate: ") + flh_last_error();
            return -1;
        }
    }
    if (x) flh_esekf_change_x(e, x);
    if (P) flh_esekf_change_P(e, P);
    return flh_esekf_update(e, R, st);
}


int flh_esekf_run_scans(flh_esekf* e, const flh_scan_job* jobs, int n_jobs, int64_t first, int64_t count, int ring, double R,
                        int with_map_incremental, double filter_size_map, int flags, flh_run_stats* out, double x_last[FLH_NSTATE],
                        double P_last[FLH_NDOF * FLH_NDOF]) {
    if (!e) return -1;
    e->err.clear();
    flh_handle* h = e->gpu_ctx.handle;
    if (!h) { e->err = "flh_esekf_run_scans: filter has no device handle"; return -1; }
    if (!jobs || n_jobs < 1 || count < 0 || first < 0) { e->err = "flh_esekf_run_scans: bad arguments"; return -1; }
    if (ring < 2) ring = 2;
    if (ring > FLH_MAX_SLOTS) ring = FLH_MAX_SLOTS;
    flh_run_stats rs;
    std::memset(&rs, 0, sizeof(rs));
    typedef std::chrono::steady_clock clk;
    auto stage = [&](int64_t i) -> int {
        const flh_scan_job& j = jobs[i % n_jobs];
        if (j.slot >= 0) return 0;
        if (flh_scan_stage_async(h, (int)(i % ring), j.pts, j.stride_bytes, j.N) != 0) {
            e->err = std::string("flh_scan_stage_async: ") + flh_last_error();
            return -1;
        }
        return 0;
    };
    const int64_t last = first + count;  // one past the last scan of this call; `last` itself is staged only with FLH_RUN_STAGE_NEXT
    auto may_stage = [&](int64_t k) { return k < last || (k == last && (flags & FLH_RUN_STAGE_NEXT)); };
    // Up to TWO scans are in flight on the staging side while one is updated (the library stages even and odd slots on two lanes):
    // a scan's staging occupies its stream for about as long as an update takes, so with one in flight the update waits for its
    // scan whenever anything jitters.  A ring of two slots can hold only one ahead.
    const int64_t ahead = ring >= 3 ? 2 : 1;
    int64_t staged_upto = (flags & FLH_RUN_FIRST_STAGED) ? first + 1 : first;  // scans [first, staged_upto) have been handed to the staging thread
    auto stage_ahead = [&](int64_t i) -> int {  // while scan i is updated, scans up to i + ahead are staged
        while (staged_upto <= i + ahead && may_stage(staged_upto)) {
            if (stage(staged_upto) != 0) return -1;
            ++staged_upto;
        }
        return 0;
    };
    // an error return must not leave the staging thread reading the caller's buffers: wait for every slot of the ring first
    auto bail = [&]() -> int {
        const std::string keep = e->err;
        for (int sidx = 0; sidx < ring; ++sidx) (void)flh_scan_wait(h, sidx);
        e->err = keep;
        return -1;
    };
    for (int64_t i = first; i < first + count; ++i) {
        const flh_scan_job& j = jobs[i % n_jobs];
        if (stage_ahead(i) != 0) return bail();  // scans i+1, i+2 cross PCIe while scan i updates (scan i itself first, if nobody has staged it)
        flh_update_stats st;
        if (flh_esekf_update_scan(e, j.slot >= 0 ? j.slot : (int)(i % ring), j.x, j.P, R, &st) != 0) return bail();
        rs.scans++;
        rs.passes += st.passes;
        rs.searches += st.searches;
        for (int k = 0; k < st.passes && k < 8; ++k) {
            if (st.pass_search[k]) { rs.ms_search_passes += st.pass_ms[k]; rs.n_search_passes++; }
            else { rs.ms_nosearch_passes += st.pass_ms[k]; rs.n_nosearch_passes++; }
        }
        if (with_map_incremental) {
            const auto t0 = clk::now();
            double x26[FLH_NSTATE];
            e->kf.get_x().to_flat(x26);
            if (flh_map_incremental(h, x26, filter_size_map, 1, 1, nullptr, nullptr) != 0) {
                e->err = std::string("flh_map_incremental: ") + flh_last_error();
                return bail();
            }
            rs.ms_map_incremental += std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        }
    }
    if (out) *out = rs;
    if (x_last) flh_esekf_get_x(e, x_last);
    if (P_last) flh_esekf_get_P(e, P_last);
    return 0;
}

}  // extern "C"
