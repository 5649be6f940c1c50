// esekfom.hpp -- host-side iterated error-state Kalman filter with the call surface of the reference's
// esekfom::esekf<state, process_noise_dof, input> (include/IKFoM_toolkit/esekfom/esekfom.hpp), restricted
// to what FAST-LIO2's node uses: init_dyn_share, predict, update_iterated_dyn_share_modified, get_x/get_P,
// change_x/change_P.  The other seven update_* variants and the USE_sparse paths are dead code for that
// node and are not provided.
//
// One extension: dyn_share_datastruct carries the fused normal equations (HTH, HTh, n_eff) so that a
// measurement model running on the GPU never has to materialise the N_eff x 12 Jacobian; h_x/h are
// still honoured when a model fills them instead.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <vector>

#include "mtk.hpp"
#include "smallmat.hpp"

namespace esekfom {
using fastlio_amd::Mat;
using fastlio_amd::Vec;

// esekfom.hpp:79-89 (+ has_normal_eq/HTH/HTh/n_eff)
template <typename T>
struct dyn_share_datastruct {
    bool valid = true;
    bool converge = true;
    std::vector<T> z;
    std::vector<T> h;    // n_eff
    std::vector<T> h_v;
    std::vector<T> h_x;  // n_eff x 12, column-major (Eigen::MatrixXd layout)
    std::vector<T> R;
    // --- extension: fused normal equations
    bool has_normal_eq = false;
    int64_t n_eff = 0;
    T HTH[144];  // row-major 12x12 = h_x^T h_x
    T HTh[12];   // h_x^T h
    T total_residual = 0;
    // --- extension: what the filter expects the evaluation AFTER the one it is about to start to be (kNext*; set before every call
    // of the measurement model).  A model on a GPU may enqueue that evaluation's kernel ahead of its state (flh_eval_expect_next);
    // models that ignore it lose nothing.
    int next_pass = 0;
};
enum { kNextUnknown = 0, kNextNoSearch = 1, kNextNone = 2 };  // = FLH_NEXT_* (include/fastlio_hip.h)

// Measurement models of the reference's plain signature (esekfom.hpp:129) that come in two halves register their first half
// here, keyed by the address of the model; init_dyn_share / set_meas_model look it up.  fastlio_amd::h_share_model does
// (h_share_model.hpp), so the node's lines laserMapping.cpp:826-828 get the overlap without a change.
struct split_model_entry { void* model; void* begin; void* finish; };
inline std::vector<split_model_entry>& split_models() {
    static std::vector<split_model_entry> v;
    return v;
}
// finish (optional): called when an update returns after it had announced a no-search pass that then did not come (kNextNoSearch
// as the last hint): a model that acted on the hint takes it back (void(), reads the model's globals)
inline bool register_split_model(void* model, void* begin, void* finish = nullptr) {
    for (auto& e : split_models())
        if (e.model == model) { e.begin = begin; e.finish = finish; return true; }
    split_models().push_back({model, begin, finish});
    return true;
}
inline void* find_split_begin(void* model) {
    for (const auto& e : split_models())
        if (e.model == model) return e.begin;
    return nullptr;
}
inline void* find_split_finish(void* model) {
    for (const auto& e : split_models())
        if (e.model == model) return e.finish;
    return nullptr;
}

template <typename state, int process_noise_dof, typename input = state>
class esekf {
   public:
    enum { n = state::DOF, m = state::DIM };
    typedef double scalar_type;
    typedef Mat<n, n> cov;
    typedef Vec<n> vectorized_state;
    typedef Vec<m> flatted_state;
    typedef flatted_state processModel(state&, const input&);
    typedef Mat<m, n> processMatrix1(state&, const input&);
    typedef Mat<m, process_noise_dof> processMatrix2(state&, const input&);
    typedef Mat<process_noise_dof, process_noise_dof> processnoisecovariance;
    // measurementModel_dyn_share exactly as the reference declares it (esekfom.hpp:129): a plain function that reads
    // its inputs from globals.  The _ctx form carries an opaque context pointer instead (C callers, several filters in one
    // process); either can be registered.
    typedef void measurementModel_dyn_share(state&, dyn_share_datastruct<scalar_type>&);
    typedef void measurementModel_dyn_share_ctx(state&, dyn_share_datastruct<scalar_type>&, void* ctx);
    // Optional FIRST HALF of a measurement model that can be started before it is waited for (a GPU pass: flh_eval_begin).  When
    // one is known for the registered model -- set_meas_begin, or register_split_model below for the reference's plain signature --
    // the update starts the model, does the part of the pass's 23x23 algebra that does not depend on the measurement (x [-] x_prop, the
    // covariance projection) while the device works, and only then calls the model proper, which waits.
    // Same arithmetic on the same operands: same bits.
    typedef void measurementModel_begin(state&, dyn_share_datastruct<scalar_type>&);
    typedef void measurementModel_begin_ctx(state&, dyn_share_datastruct<scalar_type>&, void* ctx);
    // Optional: told when an update ends although its last hint (dyn_share_datastruct::next_pass) announced another no-search pass
    typedef void measurementModel_finish();
    typedef void measurementModel_finish_ctx(void* ctx);

    esekf(const state& x = state(), const cov& P = cov::Identity()) : x_(x), P_(P) {}

    // esekfom.hpp:238-254 -- the reference's own signature: laserMapping.cpp:828 compiles against it unchanged
    void init_dyn_share(processModel f_in, processMatrix1 f_x_in, processMatrix2 f_w_in,
                        measurementModel_dyn_share h_dyn_share_in, int maximum_iteration, scalar_type limit_vector[n]) {
        init_common(f_in, f_x_in, f_w_in, maximum_iteration, limit_vector);
        h_dyn_share = h_dyn_share_in;
        h_dyn_share_ctx = nullptr;
        h_ctx_ = nullptr;
        h_begin = reinterpret_cast<measurementModel_begin*>(find_split_begin(reinterpret_cast<void*>(h_dyn_share_in)));
        h_begin_ctx = nullptr;
        h_finish = reinterpret_cast<measurementModel_finish*>(find_split_finish(reinterpret_cast<void*>(h_dyn_share_in)));
        h_finish_ctx = nullptr;
    }
    // the same with a context pointer handed to the measurement model
    void init_dyn_share(processModel f_in, processMatrix1 f_x_in, processMatrix2 f_w_in,
                        measurementModel_dyn_share_ctx h_dyn_share_in, int maximum_iteration, const scalar_type limit_vector[n],
                        void* h_ctx) {
        init_common(f_in, f_x_in, f_w_in, maximum_iteration, limit_vector);
        h_dyn_share = nullptr;
        h_dyn_share_ctx = h_dyn_share_in;
        h_ctx_ = h_ctx;
        h_begin = nullptr;
        h_begin_ctx = nullptr;
        h_finish = nullptr;
        h_finish_ctx = nullptr;
    }
    void set_meas_model(measurementModel_dyn_share_ctx h, void* ctx) {
        h_dyn_share = nullptr; h_dyn_share_ctx = h; h_ctx_ = ctx; h_begin = nullptr; h_begin_ctx = nullptr; h_finish = nullptr; h_finish_ctx = nullptr;
    }
    void set_meas_model(measurementModel_dyn_share h) {
        h_dyn_share = h; h_dyn_share_ctx = nullptr; h_ctx_ = nullptr;
        h_begin = reinterpret_cast<measurementModel_begin*>(find_split_begin(reinterpret_cast<void*>(h)));
        h_begin_ctx = nullptr;
        h_finish = reinterpret_cast<measurementModel_finish*>(find_split_finish(reinterpret_cast<void*>(h)));
        h_finish_ctx = nullptr;
    }
    // first half of the _ctx model registered last (same context pointer); nullptr: the model is called in one piece
    void set_meas_begin(measurementModel_begin_ctx b) { h_begin_ctx = b; h_begin = nullptr; }
    void set_meas_finish(measurementModel_finish_ctx f) { h_finish_ctx = f; h_finish = nullptr; }

    // esekfom.hpp:279-383 (dense path)
    void predict(double& dt, processnoisecovariance& Q, const input& i_in) {
        flatted_state f_ = f(x_, i_in);
        Mat<m, n> f_x_ = f_x(x_, i_in);
        cov f_x_final;
        Mat<m, process_noise_dof> f_w_ = f_w(x_, i_in);
        Mat<n, process_noise_dof> f_w_final;
        state x_before = x_;
        x_.oplus(f_, dt);
        F_x1 = cov::Identity();
        for (auto it = x_.vect_state.begin(); it != x_.vect_state.end(); it++) {
            const int idx = it->first.first, dim = it->first.second, dof = it->second;
            for (int i = 0; i < n; i++)
                for (int j = 0; j < dof; j++) f_x_final(idx + j, i) = f_x_(dim + j, i);
            for (int i = 0; i < process_noise_dof; i++)
                for (int j = 0; j < dof; j++) f_w_final(idx + j, i) = f_w_(dim + j, i);
        }
        for (auto it = x_.SO3_state.begin(); it != x_.SO3_state.end(); it++) {
            const int idx = it->first, dim = it->second;
            fastlio_amd::V3 seg_SO3;
            for (int i = 0; i < 3; i++) seg_SO3[i] = -1 * f_[dim + i] * dt;
            // F_x1 block = exp(seg, scalar_type(1/2)).toRotationMatrix(): 1/2 is integer division -> the
            // identity (esekfom.hpp:312); F_x1 already holds it.
            const fastlio_amd::M3 res_temp_SO3 = MTK::A_matrix(seg_SO3);
            for (int i = 0; i < n; i++) f_x_final.template set_block<3, 1>(idx, i, res_temp_SO3 * f_x_.template block<3, 1>(dim, i));
            for (int i = 0; i < process_noise_dof; i++)
                f_w_final.template set_block<3, 1>(idx, i, res_temp_SO3 * f_w_.template block<3, 1>(dim, i));
        }
        for (auto it = x_.S2_state.begin(); it != x_.S2_state.end(); it++) {
            const int idx = it->first, dim = it->second;
            fastlio_amd::V3 seg_S2;
            for (int i = 0; i < 3; i++) seg_S2[i] = f_[dim + i] * dt;
            Mat<2, 3> Nx;
            Mat<3, 2> Mx;
            x_.S2_Nx_yy(Nx, idx);
            x_before.S2_Mx(Mx, Vec<2>::Zero(), idx);
            // res = exp(seg_S2, scalar_type(1/2)) -> identity rotation (same quirk, :344)
            F_x1.template set_block<2, 2>(idx, idx, Nx * Mx);
            fastlio_amd::M3 x_before_hat;
            x_before.S2_hat(x_before_hat, idx);
            const Mat<2, 3> res_temp_S2 = ((-Nx) * x_before_hat) * MTK::A_matrix(seg_S2).transpose();
            for (int i = 0; i < n; i++) f_x_final.template set_block<2, 1>(idx, i, res_temp_S2 * f_x_.template block<3, 1>(dim, i));
            for (int i = 0; i < process_noise_dof; i++)
                f_w_final.template set_block<2, 1>(idx, i, res_temp_S2 * f_w_.template block<3, 1>(dim, i));
        }
        F_x1 += f_x_final * dt;
        const Mat<n, process_noise_dof> fw = f_w_final * dt;
        P_ = (F_x1 * P_) * F_x1.transpose() + (fw * Q) * fw.transpose();
    }

    struct update_stats {
        int passes = 0, searches = 0, returned_in_loop = 0;
        int n_eff[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        int pass_search[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        double pass_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wall time of each pass: measurement model + host algebra
        double h_ms = 0, solve_ms = 0;
    };
    const update_stats& last_stats() const { return stats_; }

    // esekfom.hpp:1619-1931
    void update_iterated_dyn_share_modified(double R, double& solve_time) {
        typedef std::chrono::steady_clock clk;
        stats_ = update_stats();
        dyn_share_datastruct<scalar_type>& dyn_share = dyn_share_;
        dyn_share.valid = true;
        dyn_share.converge = true;
        int t = 0;
        state x_propagated = x_;
        cov P_propagated = P_;
        int dof_Measurement;
        Vec<n> K_h;
        cov K_x;
        vectorized_state dx_new = vectorized_state::Zero();
        vectorized_state dx_early, dx_new_early;
        cov P_early, P_temp_early;
        for (int i = -1; i < maximum_iter; i++) {
            dyn_share.valid = true;
            dyn_share.has_normal_eq = false;
            const auto t_h0 = clk::now();
            const bool searched = dyn_share.converge;
            // What follows this pass, as far as it can be known before it (esekfom.hpp:1823-1834): nothing after the last one; a
            // SEARCH after the last but one when no step has converged yet (:1829-1832 forces it); otherwise a no-search pass
            // unless this pass's step converges (then: a search if it is the first to converge, the end of the update if the second)
            // (kNextNone is for the END of the update -- finish_hint -- where a kernel enqueued ahead is released)
            dyn_share.next_pass = (i == maximum_iter - 1 || (!t && i == maximum_iter - 2)) ? kNextUnknown : kNextNoSearch;
            bool early = false;
            if ((h_dyn_share_ctx && h_begin_ctx) || (!h_dyn_share_ctx && h_begin)) {
                // the measurement model is under way on the device: what :1655-1699 and the first inverse of :1782 compute from the
                // state alone is computed now, into temporaries (an invalid measurement must leave P_ as the reference does)
                if (h_dyn_share_ctx) h_begin_ctx(x_, dyn_share, h_ctx_);
                else h_begin(x_, dyn_share);
                x_.boxminus(dx_early, x_propagated);
                dx_new_early = dx_early;
                P_early = P_propagated;
                project_cov(P_early, dx_new_early, dx_early, x_, x_propagated);
#ifdef FASTLIO_AMD_REFERENCE_ALGEBRA
                P_temp_early = fastlio_amd::inverse(P_early / R);
#endif
                early = true;
            }
            if (h_dyn_share_ctx) h_dyn_share_ctx(x_, dyn_share, h_ctx_);
            else h_dyn_share(x_, dyn_share);
            stats_.h_ms += std::chrono::duration<double, std::milli>(clk::now() - t_h0).count();
            const int pass_no = stats_.passes;
            if (stats_.passes < 8) {
                stats_.pass_search[stats_.passes] = searched ? 1 : 0;
                stats_.n_eff[stats_.passes] = dyn_share.valid ? (int)meas_rows(dyn_share) : 0;
            }
            stats_.passes++;
            stats_.searches += searched ? 1 : 0;
            if (!dyn_share.valid) {  // :1638-1641
                if (pass_no < 8) stats_.pass_ms[pass_no] = std::chrono::duration<double, std::milli>(clk::now() - t_h0).count();
                if (i == maximum_iter - 1) finish_hint(dyn_share);
                continue;
            }

            const auto solve_start = clk::now();
            dof_Measurement = (int)meas_rows(dyn_share);
            vectorized_state dx;
            if (early) {
                dx = dx_early;
                dx_new = dx_new_early;
                P_ = P_early;
            } else {
                x_.boxminus(dx, x_propagated);
                dx_new = dx;
                P_ = P_propagated;
                project_cov(P_, dx_new, dx, x_, x_propagated);  // :1659-1699
            }

            double HTH[144], HTh[12];
            bool kx_pending = false, six = true;  // (default algebra) K_x not formed yet: only the update's LAST pass reads it
            Vec<n> dx_;
            if (n > dof_Measurement) {  // :1715-1744 gain form on explicit rows
                gain_small(dyn_share, dof_Measurement, R, K_h, K_x);
                dx_ = K_h + (K_x - cov::Identity()) * dx_new;  // :1815
            } else {  // :1782-1809 information form
                normal_equations(dyn_share, dof_Measurement, HTH, HTh);
#ifdef FASTLIO_AMD_REFERENCE_ALGEBRA
                // the reference's own sequence (:1782,:1802): invert P / R, add HTH, invert again
                cov P_temp = early ? P_temp_early : fastlio_amd::inverse(P_ / R);
                for (int a = 0; a < 12; ++a)
                    for (int b = 0; b < 12; ++b) P_temp(a, b) += HTH[a * 12 + b];
                // only P_inv.block<n,12>(0,0) is read below (:1803-1806): the first 12 columns of the inverse, bit for bit
                const Mat<n, 12> P_inv = fastlio_amd::inverse_cols<n, 12>(P_temp);
                gain_from_cols<12>(P_inv, HTH, HTh, K_h, K_x);
                dx_ = K_h + (K_x - cov::Identity()) * dx_new;  // :1815
#else
                // The same 12 columns without either 23x23 inverse.  With A = (P / R)^-1 and U = [I12; 0] the reference forms
                // (A + U HTH U^T)^-1 and reads its first 12 columns, (A + U HTH U^T)^-1 U.  By the push-through identity that is
                //     A^-1 U (I + HTH U^T A^-1 U)^-1  =  B[:, :12] (I + HTH B[:12, :12])^-1,   B = P / R,
                // one 12 x 12 elimination with 23 right-hand sides; B needs no inverse at all.  Same quantity, fewer roundings
                // (the covariance is never inverted: tests/test_host_algebra.py measures what that is worth).  Without
                // extrinsic estimation rows and columns 6..11 of HTH are zero (laserMapping.cpp:745), the columns 6..11 of
                // P_inv then only ever multiply zeros, and the first six are B[:, :6] (I + HTH[:6, :6] B[:6, :6])^-1: a 6 x 6
                // elimination.  The columns that are not computed are left zero.
                for (int a = 0; a < 12 && six; ++a)
                    for (int b = (a < 6 ? 6 : 0); b < 12; ++b)
                        if (HTH[a * 12 + b] != 0.0) { six = false; break; }
                for (int a = 6; a < 12; ++a) six = six && HTh[a] == 0.0;
                // The step is on the critical path of the NEXT pass (its state is x_ [+] dx_); K_h and K_x are not: K_x is read
                // by the final covariance alone (:1836-1924).  With W = P_inv's columns, K_h = W HTh and K_x[:, :12] = W HTH, so
                //     dx_ = K_h + (K_x - I) dx_new  =  B[:, :NC] S^-1 (HTh + HTH dx_new[:NC]) - dx_new,   S = I + HTH B[:NC, :NC]:
                // ONE right-hand side instead of 23 and no 23 x NC x NC product per pass; K_x is formed once, for the pass that
                // ends the update (the reference's sequence: -DFASTLIO_AMD_REFERENCE_ALGEBRA).
                dx_ = six ? info_step<6>(P_, R, HTH, HTh, dx_new) : info_step<12>(P_, R, HTH, HTh, dx_new);
                kx_pending = true;
#endif
            }
            x_.boxplus(dx_);
            dyn_share.converge = true;
            for (int j = 0; j < n; j++) {
                if (std::fabs(dx_[j]) > limit[j]) {
                    dyn_share.converge = false;
                    break;
                }
            }
            if (dyn_share.converge) t++;
            if (!t && i == maximum_iter - 2) dyn_share.converge = true;  // :1829-1832

            if (t > 1 || i == maximum_iter - 1) {  // :1834-1928
                finish_hint(dyn_share);
                if (kx_pending) {
                    const Mat<n, 12> P_inv = six ? info_cols<6>(P_, R, HTH) : info_cols<12>(P_, R, HTH);
                    if (six) gain_from_cols<6>(P_inv, HTH, HTh, K_h, K_x);
                    else gain_from_cols<12>(P_inv, HTH, HTh, K_h, K_x);
                }
                if (kx_pending && six) final_cov<6>(P_, K_x, dx_, x_, x_propagated);
                else final_cov<12>(P_, K_x, dx_, x_, x_propagated);
                stats_.returned_in_loop = 1;
                const double ms = std::chrono::duration<double, std::milli>(clk::now() - solve_start).count();
                stats_.solve_ms += ms;
                solve_time += ms * 1e-3;
                if (pass_no < 8) stats_.pass_ms[pass_no] = std::chrono::duration<double, std::milli>(clk::now() - t_h0).count();
                return;
            }
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - solve_start).count();
            stats_.solve_ms += ms;
            solve_time += ms * 1e-3;
            if (pass_no < 8) stats_.pass_ms[pass_no] = std::chrono::duration<double, std::milli>(clk::now() - t_h0).count();
        }
    }

    void change_x(state& input_state) {  // :1933-1942
        x_ = input_state;
        if ((!x_.vect_state.size()) && (!x_.SO3_state.size()) && (!x_.S2_state.size())) {
            x_.build_S2_state();
            x_.build_SO3_state();
            x_.build_vect_state();
        }
    }
    void change_P(cov& input_cov) { P_ = input_cov; }
    const state& get_x() const { return x_; }
    const cov& get_P() const { return P_; }

   private:
    // the update ends: a model that was told to expect another no-search pass is told that none comes
    void finish_hint(dyn_share_datastruct<scalar_type>& d) {
        if (d.next_pass != kNextNoSearch) return;
        d.next_pass = kNextUnknown;
        if (h_finish_ctx) h_finish_ctx(h_ctx_);
        else if (h_finish) h_finish();
    }
    void init_common(processModel f_in, processMatrix1 f_x_in, processMatrix2 f_w_in, int maximum_iteration,
                     const scalar_type* limit_vector) {
        f = f_in;
        f_x = f_x_in;
        f_w = f_w_in;
        maximum_iter = maximum_iteration;
        for (int i = 0; i < n; i++) limit[i] = limit_vector[i];
        x_.build_S2_state();
        x_.build_SO3_state();
        x_.build_vect_state();
    }
    // K_h = P_inv HTh, K_x[:, :12] = P_inv HTH (:1803-1806)
    // (nc = 6: columns 6..11 of P_inv and of HTH are zero -- the terms left out are products with zero)
    template <int nc>
    static void gain_from_cols(const Mat<n, 12>& P_inv, const double HTH[144], const double HTh[12], Vec<n>& K_h, cov& K_x) {
        for (int r = 0; r < n; ++r) {
            double s = 0;
            for (int c = 0; c < nc; ++c) s += P_inv(r, c) * HTh[c];
            K_h[r] = s;
        }
        K_x = cov::Zero();
        for (int r = 0; r < n; ++r)
            for (int b = 0; b < nc; ++b) {
                double s = 0;
                for (int c = 0; c < nc; ++c) s += P_inv(r, c) * HTH[c * 12 + b];
                K_x(r, b) = s;
            }
    }
    // dx_ of :1815 from one NC x NC elimination with ONE right-hand side (see the call site)
    template <int NC>
    static Vec<n> info_step(const cov& P, double R, const double HTH[144], const double HTh[12], const vectorized_state& dx_new) {
        double S[NC * NC], B11[NC * NC], g[NC], v[NC];
        const double rinv = 1.0 / R;
        for (int a = 0; a < NC; ++a)
            for (int b = 0; b < NC; ++b) B11[a * NC + b] = P(a, b) * rinv;
        for (int a = 0; a < NC; ++a) {
            for (int b = 0; b < NC; ++b) {
                double s = 0;
                for (int c = 0; c < NC; ++c) s += HTH[a * 12 + c] * B11[c * NC + b];
                S[a * NC + b] = s + (a == b ? 1.0 : 0.0);
            }
            double s = HTh[a];
            for (int c = 0; c < NC; ++c) s += HTH[a * 12 + c] * dx_new[c];
            g[a] = s;
        }
        fastlio_amd::solve_lu_fixed<NC, 1>(S, g, v);
        Vec<n> dx;
        for (int r = 0; r < n; ++r) {
            double s = 0;
            for (int c = 0; c < NC; ++c) s += (P(r, c) * rinv) * v[c];
            dx[r] = s - dx_new[r];
        }
        return dx;
    }
    // The first NC columns of P_inv.block<n,12>(0,0) of :1802 as B[:, :NC] (I + HTH[:NC, :NC] B[:NC, :NC])^-1 with B = P / R
    // (see the call site); the other columns are zero
    template <int NC>
    static Mat<n, 12> info_cols(const cov& P, double R, const double HTH[144]) {
        double St[NC * NC], Bt[NC * n], Xt[NC * n];  // S^T, B[:, :NC]^T, the solution of S^T X = B[:, :NC]^T
        double B11[NC * NC];
        const double rinv = 1.0 / R;
        for (int a = 0; a < NC; ++a)
            for (int b = 0; b < NC; ++b) B11[a * NC + b] = P(a, b) * rinv;
        for (int a = 0; a < NC; ++a)
            for (int b = 0; b < NC; ++b) {
                double s = 0;
                for (int c = 0; c < NC; ++c) s += HTH[a * 12 + c] * B11[c * NC + b];
                St[b * NC + a] = s + (a == b ? 1.0 : 0.0);
            }
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < NC; ++c) Bt[c * n + r] = P(r, c) * rinv;
        fastlio_amd::solve_lu_fixed<NC, n>(St, Bt, Xt);
        Mat<n, 12> W;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < NC; ++c) W(r, c) = Xt[c * n + r];
        return W;
    }
    static int64_t meas_rows(const dyn_share_datastruct<scalar_type>& d) {
        return d.has_normal_eq ? d.n_eff : (int64_t)d.h.size();
    }
    static void normal_equations(const dyn_share_datastruct<scalar_type>& d, int rows, double HTH[144], double HTh[12]) {
        if (d.has_normal_eq) {
            for (int i = 0; i < 144; ++i) HTH[i] = d.HTH[i];
            for (int i = 0; i < 12; ++i) HTh[i] = d.HTh[i];
            return;
        }
        for (int a = 0; a < 12; ++a) {
            for (int b = 0; b < 12; ++b) {
                double s = 0;
                for (int k = 0; k < rows; ++k) s += d.h_x[(size_t)a * rows + k] * d.h_x[(size_t)b * rows + k];
                HTH[a * 12 + b] = s;
            }
            double s = 0;
            for (int k = 0; k < rows; ++k) s += d.h_x[(size_t)a * rows + k] * d.h[k];
            HTh[a] = s;
        }
    }
    // P <- J P J^T blockwise and dx_new <- J dx_new, J = A(dx_so3)^T (SO3) / Nx*Mx (S2): :1659-1699
    static void project_cov(cov& P, vectorized_state& dx_new, const vectorized_state& dx, state& x, state& x_prop) {
        for (auto it = x.SO3_state.begin(); it != x.SO3_state.end(); it++) {
            const int idx = it->first;
            fastlio_amd::V3 seg;
            for (int i = 0; i < 3; i++) seg[i] = dx[idx + i];
            const fastlio_amd::M3 J = MTK::A_matrix(seg).transpose();
            dx_new.template set_block<3, 1>(idx, 0, J * dx_new.template block<3, 1>(idx, 0));
            for (int i = 0; i < n; i++) P.template set_block<3, 1>(idx, i, J * P.template block<3, 1>(idx, i));
            for (int i = 0; i < n; i++) P.template set_block<1, 3>(i, idx, P.template block<1, 3>(i, idx) * J.transpose());
        }
        for (auto it = x.S2_state.begin(); it != x.S2_state.end(); it++) {
            const int idx = it->first;
            Vec<2> seg;
            for (int i = 0; i < 2; i++) seg[i] = dx[idx + i];
            Mat<2, 3> Nx;
            Mat<3, 2> Mx;
            x.S2_Nx_yy(Nx, idx);
            x_prop.S2_Mx(Mx, seg, idx);
            const Mat<2, 2> J = Nx * Mx;
            dx_new.template set_block<2, 1>(idx, 0, J * dx_new.template block<2, 1>(idx, 0));
            for (int i = 0; i < n; i++) P.template set_block<2, 1>(idx, i, J * P.template block<2, 1>(idx, i));
            for (int i = 0; i < n; i++) P.template set_block<1, 2>(i, idx, P.template block<1, 2>(i, idx) * J.transpose());
        }
    }
    // :1836-1924
    // (nc: the columns of K_x that are not structurally zero -- 6 without extrinsic estimation, else 12: what is left out are
    // products with zero)
    template <int nc>
    void final_cov(cov& P, cov& K_x, const Vec<n>& dx_, state& x, state& x_prop) {
        L_ = P;
        for (auto it = x.SO3_state.begin(); it != x.SO3_state.end(); it++) {
            const int idx = it->first;
            fastlio_amd::V3 seg;
            for (int i = 0; i < 3; i++) seg[i] = dx_[i + idx];
            const fastlio_amd::M3 J = MTK::A_matrix(seg).transpose();
            for (int i = 0; i < n; i++) L_.template set_block<3, 1>(idx, i, J * P.template block<3, 1>(idx, i));
            for (int i = 0; i < nc; i++) K_x.template set_block<3, 1>(idx, i, J * K_x.template block<3, 1>(idx, i));
            for (int i = 0; i < n; i++) {
                L_.template set_block<1, 3>(i, idx, L_.template block<1, 3>(i, idx) * J.transpose());
                P.template set_block<1, 3>(i, idx, P.template block<1, 3>(i, idx) * J.transpose());
            }
        }
        for (auto it = x.S2_state.begin(); it != x.S2_state.end(); it++) {
            const int idx = it->first;
            Vec<2> seg;
            for (int i = 0; i < 2; i++) seg[i] = dx_[i + idx];
            Mat<2, 3> Nx;
            Mat<3, 2> Mx;
            x.S2_Nx_yy(Nx, idx);
            x_prop.S2_Mx(Mx, seg, idx);
            const Mat<2, 2> J = Nx * Mx;
            for (int i = 0; i < n; i++) L_.template set_block<2, 1>(idx, i, J * P.template block<2, 1>(idx, i));
            for (int i = 0; i < nc; i++) K_x.template set_block<2, 1>(idx, i, J * K_x.template block<2, 1>(idx, i));
            for (int i = 0; i < n; i++) {
                L_.template set_block<1, 2>(i, idx, L_.template block<1, 2>(i, idx) * J.transpose());
                P.template set_block<1, 2>(i, idx, P.template block<1, 2>(i, idx) * J.transpose());
            }
        }
        cov Pn;
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                double s = 0;
                for (int k = 0; k < nc; ++k) s += K_x(r, k) * P(k, c);
                Pn(r, c) = L_(r, c) - s;
            }
        P = Pn;
    }
    // K = P H^T (H P H^T / R + I)^-1 / R with H zero-padded to rows x n (:1715-1744)
    void gain_small(const dyn_share_datastruct<scalar_type>& d, int rows, double R, Vec<n>& K_h, cov& K_x) {
        const int M = rows;
        std::vector<double> H((size_t)M * n, 0.0), PHt((size_t)n * M), S((size_t)M * M), Si((size_t)M * M), Kg((size_t)n * M);
        for (int k = 0; k < M; ++k)
            for (int c = 0; c < 12; ++c) H[(size_t)k * n + c] = d.h_x[(size_t)c * M + k];
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < M; ++k) {
                double s = 0;
                for (int j = 0; j < n; ++j) s += P_(i, j) * H[(size_t)k * n + j];
                PHt[(size_t)i * M + k] = s;
            }
        for (int a = 0; a < M; ++a)
            for (int b = 0; b < M; ++b) {
                double s = 0;
                for (int j = 0; j < n; ++j) s += H[(size_t)a * n + j] * PHt[(size_t)j * M + b];
                S[(size_t)a * M + b] = s / R + (a == b ? 1.0 : 0.0);
            }
        fastlio_amd::inverse_lu(S.data(), M, Si.data());
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < M; ++k) {
                double s = 0;
                for (int j = 0; j < M; ++j) s += PHt[(size_t)i * M + j] * Si[(size_t)j * M + k];
                Kg[(size_t)i * M + k] = s / R;
            }
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int k = 0; k < M; ++k) s += Kg[(size_t)i * M + k] * d.h[k];
            K_h[i] = s;
            for (int j = 0; j < n; ++j) {
                double w = 0;
                for (int k = 0; k < M; ++k) w += Kg[(size_t)i * M + k] * H[(size_t)k * n + j];
                K_x(i, j) = w;
            }
        }
    }

    state x_;
    cov P_;
    cov F_x1 = cov::Identity();
    cov L_ = cov::Identity();
    processModel* f = nullptr;
    processMatrix1* f_x = nullptr;
    processMatrix2* f_w = nullptr;
    measurementModel_dyn_share* h_dyn_share = nullptr;
    measurementModel_dyn_share_ctx* h_dyn_share_ctx = nullptr;
    void* h_ctx_ = nullptr;
    measurementModel_begin* h_begin = nullptr;
    measurementModel_begin_ctx* h_begin_ctx = nullptr;
    measurementModel_finish* h_finish = nullptr;
    measurementModel_finish_ctx* h_finish_ctx = nullptr;
    int maximum_iter = 0;
    scalar_type limit[n];
    dyn_share_datastruct<scalar_type> dyn_share_;
    update_stats stats_;
};

}  // namespace esekfom
