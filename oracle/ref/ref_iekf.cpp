// ref_iekf.cpp -- harness around the REFERENCE's esekfom::esekf<state_ikfom, 12, input_ikfom>::
// update_iterated_dyn_share_modified (include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1931), compiled from the reference tree
// (-I$REF/include).  TEST INFRASTRUCTURE.  Record-replay: the measurement model does not search a map (ikd-Tree is absent
// from the reference snapshot); it hands back, pass by pass, the rows h_x / h that the oracle's h_share_model produced for
// the same pass (oracle/ref/make_inputs.py recorded them), so the filter algebra and the MTK manifold operations of the
// real code run on exactly the oracle's inputs.
//   in : int32 n_cases; per case: double x0[26] (pos rot_xyzw offR_xyzw offT vel bg ba grav), double P0[23*23] row-major,
//        double R, int32 max_iter, int32 n_pass; per pass: int32 valid, int32 n_eff, n_eff x 12 doubles column-major, n_eff doubles
//   out: int32 n_cases; per case: int32 passes_used, double x[26], double P[23*23]
#include <cstdint>
#include <cstdio>
#include <vector>

#include <use-ikfom.hpp>

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
struct Pass { int valid, n_eff; std::vector<double> hx, h; };
static std::vector<Pass> g_passes;
static size_t g_next = 0;

static void replay_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& d) {  // measurementModel_dyn_share, :129
    (void)s;
    if (g_next >= g_passes.size()) { d.valid = false; return; }
    const Pass& p = g_passes[g_next++];
    if (!p.valid) { d.valid = false; return; }  // src/laserMapping.cpp:708-713
    d.h_x = Eigen::MatrixXd::Zero(p.n_eff, 12);  // :720
    d.h.resize(p.n_eff);                         // :721
    for (int c = 0; c < 12; ++c)
        for (int r = 0; r < p.n_eff; ++r) d.h_x(r, c) = p.hx[(size_t)c * p.n_eff + r];
    for (int r = 0; r < p.n_eff; ++r) d.h(r) = p.h[r];
}

static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    FILE* o = fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int32_t nc = 0;
    if (!rd(f, &nc, 4)) return 2;
    fwrite(&nc, 4, 1, o);
    for (int c = 0; c < nc; ++c) {
        double x0[26], P0[23 * 23], R;
        int32_t max_iter, n_pass;
        if (!rd(f, x0, sizeof(x0)) || !rd(f, P0, sizeof(P0)) || !rd(f, &R, 8) || !rd(f, &max_iter, 4) || !rd(f, &n_pass, 4)) return 2;
        g_passes.assign(n_pass, Pass());
        for (auto& p : g_passes) {
            if (!rd(f, &p.valid, 4) || !rd(f, &p.n_eff, 4)) return 2;
            p.hx.resize((size_t)p.n_eff * 12);
            p.h.resize(p.n_eff);
            if (p.n_eff && (!rd(f, p.hx.data(), p.hx.size() * 8) || !rd(f, p.h.data(), p.h.size() * 8))) return 2;
        }
        g_next = 0;
        kf_t kf;
        double epsi[23];
        for (double& e : epsi) e = 0.001;  // src/laserMapping.cpp:826-827
        kf.init_dyn_share(get_f, df_dx, df_dw, replay_model, max_iter, epsi);  // :828
        state_ikfom s = kf.get_x();
        s.pos = vect3(x0[0], x0[1], x0[2]);
        s.rot.coeffs() = Eigen::Vector4d(x0[3], x0[4], x0[5], x0[6]);
        s.offset_R_L_I.coeffs() = Eigen::Vector4d(x0[7], x0[8], x0[9], x0[10]);
        s.offset_T_L_I = vect3(x0[11], x0[12], x0[13]);
        s.vel = vect3(x0[14], x0[15], x0[16]);
        s.bg = vect3(x0[17], x0[18], x0[19]);
        s.ba = vect3(x0[20], x0[21], x0[22]);
        s.grav.vec = Eigen::Vector3d(x0[23], x0[24], x0[25]);
        kf.change_x(s);
        kf_t::cov P = kf.get_P();
        for (int i = 0; i < 23; ++i)
            for (int j = 0; j < 23; ++j) P(i, j) = P0[i * 23 + j];
        kf.change_P(P);
        double solve_time = 0;
        kf.update_iterated_dyn_share_modified(R, solve_time);  // :960
        const state_ikfom xs = kf.get_x();
        const kf_t::cov Ps = kf.get_P();
        double x[26] = {xs.pos[0], xs.pos[1], xs.pos[2], xs.rot.coeffs()[0], xs.rot.coeffs()[1], xs.rot.coeffs()[2], xs.rot.coeffs()[3],
                        xs.offset_R_L_I.coeffs()[0], xs.offset_R_L_I.coeffs()[1], xs.offset_R_L_I.coeffs()[2], xs.offset_R_L_I.coeffs()[3],
                        xs.offset_T_L_I[0], xs.offset_T_L_I[1], xs.offset_T_L_I[2], xs.vel[0], xs.vel[1], xs.vel[2], xs.bg[0], xs.bg[1],
                        xs.bg[2], xs.ba[0], xs.ba[1], xs.ba[2], xs.grav.vec[0], xs.grav.vec[1], xs.grav.vec[2]};
        double Pout[23 * 23];
        for (int i = 0; i < 23; ++i)
            for (int j = 0; j < 23; ++j) Pout[i * 23 + j] = Ps(i, j);
        const int32_t used = (int32_t)g_next;
        fwrite(&used, 4, 1, o);
        fwrite(x, 8, 26, o);
        fwrite(Pout, 8, 23 * 23, o);
    }
    fclose(f);
    fclose(o);
    printf("ref_iekf: %d cases\n", nc);
    return 0;
}
