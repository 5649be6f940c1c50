// h_share_model.hpp -- the measurement model of FAST-LIO2 (src/laserMapping.cpp:638-754) as a thin host
// wrapper over the HIP library: one flh_eval per call, normal equations straight into the extended
// dyn_share_datastruct.  Two forms:
//   void h_share_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&)          the reference's signature (:638): reads
//        its inputs from the global context fastlio_amd::g_hshare, as the reference's reads its globals (:69-114); with
//        `using fastlio_amd::h_share_model;` the node's line :828 registers it unchanged
//   void h_share_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&, void* ctx)  the same with an explicit context
#pragma once
#include <stdexcept>
#include <string>

#include "../fastlio_hip.h"
#include "esekfom.hpp"
#include "use-ikfom.hpp"

namespace fastlio_amd {

// What the reference keeps in globals (laserMapping.cpp:69-114) and h_share_model needs.
struct HShareContext {
    flh_handle* handle = nullptr;
    bool extrinsic_est_en = false;  // laserMapping.cpp:73
    // outputs mirrored from the reference's globals
    int effct_feat_num = 0;         // :93
    double total_residual = 0.0;    // :86
    double res_mean_last = 0.05;    // :86
    double match_ms = 0.0;          // wall time inside flh_eval (match_time + solve_time buckets, :640,716-717,753)
    bool begun = false;             // h_share_model_begin has enqueued this pass: h_share_model only waits for it
};

inline HShareContext g_hshare;  // the globals of laserMapping.cpp:69-114 that the two-argument form reads

inline void h_share_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data, void* ctx_);
inline void h_share_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data) {  // :638
    h_share_model(s, ekfom_data, &g_hshare);
}

// First half: the pass is enqueued on the device and the call returns (flh_eval_begin); the filter does the part of its algebra
// that needs no measurement, then calls h_share_model, which waits.  Registered for both forms (below; esekf::set_meas_begin).
inline void h_share_model_begin(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data, void* ctx_) {
    HShareContext* ctx = static_cast<HShareContext*>(ctx_);
    if (!ctx || !ctx->handle) throw std::runtime_error("fastlio_amd::h_share_model: no flh_handle bound");
    const double rot[4] = {s.rot.x, s.rot.y, s.rot.z, s.rot.w};
    const double offR[4] = {s.offset_R_L_I.x, s.offset_R_L_I.y, s.offset_R_L_I.z, s.offset_R_L_I.w};
    const double pos[3] = {s.pos[0], s.pos[1], s.pos[2]};
    const double offT[3] = {s.offset_T_L_I[0], s.offset_T_L_I[1], s.offset_T_L_I[2]};
    // what the filter expects after this evaluation (esekfom.hpp: next_pass; kNext* = FLH_NEXT_*): the library may enqueue an
    // expected no-search pass beside this one
    (void)flh_eval_expect_next(ctx->handle, ekfom_data.next_pass);
    if (flh_eval_begin(ctx->handle, rot, pos, offR, offT, ekfom_data.converge ? 1 : 0, ctx->extrinsic_est_en ? 1 : 0) != 0)
        throw std::runtime_error(std::string("flh_eval failed: ") + flh_last_error());
    ctx->begun = true;
}
inline void h_share_model_begin(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data) {
    h_share_model_begin(s, ekfom_data, &g_hshare);
}
// The update ended although the filter had announced another no-search pass: a kernel enqueued for it is released
inline void h_share_model_finish(void* ctx_) {
    HShareContext* ctx = static_cast<HShareContext*>(ctx_);
    if (ctx && ctx->handle) (void)flh_eval_expect_next(ctx->handle, FLH_NEXT_NONE);
}
inline void h_share_model_finish() { h_share_model_finish(&g_hshare); }
namespace detail {
inline const bool h_share_model_split_registered = esekfom::register_split_model(
    reinterpret_cast<void*>(static_cast<void (*)(state_ikfom&, esekfom::dyn_share_datastruct<double>&)>(&h_share_model)),
    reinterpret_cast<void*>(static_cast<void (*)(state_ikfom&, esekfom::dyn_share_datastruct<double>&)>(&h_share_model_begin)),
    reinterpret_cast<void*>(static_cast<void (*)()>(&h_share_model_finish)));
}

inline void h_share_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data, void* ctx_) {
    HShareContext* ctx = static_cast<HShareContext*>(ctx_);
    if (!ctx || !ctx->handle) throw std::runtime_error("fastlio_amd::h_share_model: no flh_handle bound");
    int64_t n_eff = 0;
    double total_res = 0;
    if (!ctx->begun) h_share_model_begin(s, ekfom_data, ctx_);  // called in one piece
    ctx->begun = false;
    if (flh_eval_end(ctx->handle, ekfom_data.HTH, ekfom_data.HTh, &n_eff, &total_res) != 0)
        throw std::runtime_error(std::string("flh_eval failed: ") + flh_last_error());
    ctx->effct_feat_num = (int)n_eff;
    ctx->total_residual = total_res;
    ekfom_data.n_eff = n_eff;
    ekfom_data.total_residual = total_res;
    ekfom_data.has_normal_eq = true;
    if (n_eff < 1) {  // laserMapping.cpp:708-713
        ekfom_data.valid = false;
        return;
    }
    ctx->res_mean_last = total_res / (double)n_eff;  // :715
    if (n_eff < state_ikfom::DOF) {
        // the gain-form branch (esekfom.hpp:1715-1744) needs explicit rows: fetch them (tiny)
        ekfom_data.h_x.assign((size_t)n_eff * 12, 0.0);
        ekfom_data.h.assign((size_t)n_eff, 0.0);
        int64_t rows = 0;
        if (flh_fetch_rows(ctx->handle, ekfom_data.h_x.data(), ekfom_data.h.data(), n_eff, &rows) != 0 || rows != n_eff)
            throw std::runtime_error(std::string("flh_fetch_rows failed: ") + flh_last_error());
        ekfom_data.has_normal_eq = false;
    }
}

}  // namespace fastlio_amd
