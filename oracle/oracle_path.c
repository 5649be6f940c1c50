/*
 * oracle_path.c -- TEST INFRASTRUCTURE (see fastlio_oracle.h header).  PARITY UNPINNED.
 *
 * Plain-C restatement of
 *   h_share_model                               src/laserMapping.cpp:638-754
 *   esekf::update_iterated_dyn_share_modified   include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1931
 *   map_incremental (add/skip decision only)    src/laserMapping.cpp:427-474
 * and an exact 5-NN k-d tree standing in for ikd-Tree's Nearest_Search (source absent from the
 * snapshot: include/ikd-Tree is an un-vendored submodule, .gitmodules:1-4).  Semantics kept:
 * exact 5-NN, squared-L2 in fp32, ascending; ties broken by lower map index (our convention).
 *
 * The reference's static 100 000-point caps (laserMapping.cpp:76,94,112-114) are lifted: all per-
 * point arrays are sized by N.
 */
#include "fastlio_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NDOF ORC_NDOF
#define K ORC_K
enum { X_POS = 0, X_ROT = 3, X_OFFR = 7, X_OFFT = 11, X_VEL = 14, X_BG = 17, X_BA = 20, X_GRAV = 23 };

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

/* =================================================================== exact 5-NN k-d tree */
#define LEAF 12
typedef struct {
    float split;
    int32_t dim;   /* -1 = leaf */
    int32_t left;  /* child index, or first point for leaves */
    int32_t right; /* child index, or point count for leaves */
} kd_node;
struct orc_kdtree {
    size_t M;
    float* pts;     /* M x 3, tree order */
    int32_t* index; /* original indices, tree order */
    kd_node* nodes;
    int32_t nnodes, cap;
};

static inline float dist2f(const float* a, const float* b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}
/* (d2, idx) lexicographic "a before b" */
static inline int before(float d2a, int32_t ia, float d2b, int32_t ib) {
    return d2a < d2b || (d2a == d2b && ia < ib);
}
typedef struct {
    int n;
    int32_t idx[K];
    float d2[K];
} top5;
static inline void top5_insert(top5* t, float d2, int32_t idx) {
    if (t->n == K && !before(d2, idx, t->d2[K - 1], t->idx[K - 1])) return;
    int pos = t->n < K ? t->n : K - 1;
    while (pos > 0 && before(d2, idx, t->d2[pos - 1], t->idx[pos - 1])) {
        t->d2[pos] = t->d2[pos - 1];
        t->idx[pos] = t->idx[pos - 1];
        pos--;
    }
    t->d2[pos] = d2;
    t->idx[pos] = idx;
    if (t->n < K) t->n++;
}

static void swap_pt(orc_kdtree* t, size_t a, size_t b) {
    if (a == b) return;
    float tmp[3];
    memcpy(tmp, t->pts + 3 * a, sizeof(tmp));
    memcpy(t->pts + 3 * a, t->pts + 3 * b, sizeof(tmp));
    memcpy(t->pts + 3 * b, tmp, sizeof(tmp));
    int32_t ti = t->index[a];
    t->index[a] = t->index[b];
    t->index[b] = ti;
}
/* quickselect on dimension d so that element k is in sorted position within [lo,hi) */
static void nth_element(orc_kdtree* t, size_t lo, size_t hi, size_t k, int d) {
    while (hi - lo > 1) {
        size_t mid = lo + (hi - lo) / 2;
        /* median of three */
        float a = t->pts[3 * lo + d], b = t->pts[3 * mid + d], c = t->pts[3 * (hi - 1) + d];
        size_t pi = (a < b) ? ((b < c) ? mid : (a < c ? hi - 1 : lo)) : ((a < c) ? lo : (b < c ? hi - 1 : mid));
        float pv = t->pts[3 * pi + d];
        swap_pt(t, pi, hi - 1);
        size_t st = lo;
        for (size_t i = lo; i + 1 < hi; i++)
            if (t->pts[3 * i + d] < pv) swap_pt(t, i, st++);
        swap_pt(t, st, hi - 1);
        /* handle runs of equal keys to avoid quadratic behaviour on lattices */
        size_t eq = st + 1;
        for (size_t i = st + 1; i < hi; i++)
            if (t->pts[3 * i + d] == pv) swap_pt(t, i, eq++);
        if (k < st) hi = st;
        else if (k >= eq) lo = eq;
        else return;
    }
}
static int32_t new_node(orc_kdtree* t) {
    if (t->nnodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap);
    }
    return t->nnodes++;
}
static int32_t build_rec(orc_kdtree* t, size_t lo, size_t hi) {
    int32_t id = new_node(t);
    if (hi - lo <= LEAF) {
        t->nodes[id].dim = -1;
        t->nodes[id].left = (int32_t)lo;
        t->nodes[id].right = (int32_t)(hi - lo);
        t->nodes[id].split = 0;
        return id;
    }
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = lo; i < hi; i++)
        for (int d = 0; d < 3; d++) {
            float v = t->pts[3 * i + d];
            if (v < mn[d]) mn[d] = v;
            if (v > mx[d]) mx[d] = v;
        }
    int d = 0;
    if (mx[1] - mn[1] > mx[d] - mn[d]) d = 1;
    if (mx[2] - mn[2] > mx[d] - mn[d]) d = 2;
    if (mx[d] == mn[d]) { /* all points identical: oversized leaf */
        t->nodes[id].dim = -1;
        t->nodes[id].left = (int32_t)lo;
        t->nodes[id].right = (int32_t)(hi - lo);
        t->nodes[id].split = 0;
        return id;
    }
    size_t mid = lo + (hi - lo) / 2;
    nth_element(t, lo, hi, mid, d);
    float split = t->pts[3 * mid + d];
    /* left: [lo,mid) has values <= split; right: [mid,hi) has values >= split */
    int32_t l = build_rec(t, lo, mid);
    int32_t r = build_rec(t, mid, hi);
    t->nodes[id].dim = d;
    t->nodes[id].split = split;
    t->nodes[id].left = l;
    t->nodes[id].right = r;
    return id;
}
orc_kdtree* orc_kdtree_build(const float* xyz, size_t stride, size_t M) {
    orc_kdtree* t = (orc_kdtree*)calloc(1, sizeof(orc_kdtree));
    t->M = M;
    t->pts = (float*)malloc(sizeof(float) * 3 * (M ? M : 1));
    t->index = (int32_t*)malloc(sizeof(int32_t) * (M ? M : 1));
    for (size_t i = 0; i < M; i++) {
        t->pts[3 * i + 0] = xyz[i * stride + 0];
        t->pts[3 * i + 1] = xyz[i * stride + 1];
        t->pts[3 * i + 2] = xyz[i * stride + 2];
        t->index[i] = (int32_t)i;
    }
    if (M > 0) build_rec(t, 0, M);
    return t;
}
void orc_kdtree_free(orc_kdtree* t) {
    if (!t) return;
    free(t->pts); free(t->index); free(t->nodes); free(t);
}
size_t orc_kdtree_size(const orc_kdtree* t) { return t->M; }

static void search_rec(const orc_kdtree* t, int32_t id, const float* q, top5* best) {
    const kd_node* n = &t->nodes[id];
    if (n->dim < 0) {
        const float* p = t->pts + 3 * (size_t)n->left;
        for (int32_t i = 0; i < n->right; i++) top5_insert(best, dist2f(q, p + 3 * i), t->index[n->left + i]);
        return;
    }
    float diff = q[n->dim] - n->split;
    int32_t first = diff < 0 ? n->left : n->right;
    int32_t second = diff < 0 ? n->right : n->left;
    search_rec(t, first, q, best);
    /* fl(diff*diff) is a valid lower bound of the fp32 d2 of anything on the far side (rounding is
       monotone); do not prune on equality so equal-distance lower-index points are still found. */
    float pd = diff * diff;
    if (best->n < K || !(pd > best->d2[K - 1])) search_rec(t, second, q, best);
}
int orc_knn5(const orc_kdtree* t, const float q[3], int32_t idx[K], float d2[K]) {
    top5 b;
    b.n = 0;
    if (t->M > 0) search_rec(t, 0, q, &b);
    for (int i = 0; i < b.n; i++) { idx[i] = b.idx[i]; d2[i] = b.d2[i]; }
    for (int i = b.n; i < K; i++) { idx[i] = -1; d2[i] = INFINITY; }
    return b.n;
}
int orc_knn5_brute(const float* xyz, size_t stride, size_t M, const float q[3], int32_t idx[K], float d2[K]) {
    top5 b;
    b.n = 0;
    for (size_t i = 0; i < M; i++) top5_insert(&b, dist2f(q, xyz + i * stride), (int32_t)i);
    for (int i = 0; i < b.n; i++) { idx[i] = b.idx[i]; d2[i] = b.d2[i]; }
    for (int i = b.n; i < K; i++) { idx[i] = -1; d2[i] = INFINITY; }
    return b.n;
}
void orc_knn5_batch(const orc_kdtree* t, const float* q, size_t N, int32_t* idx, float* d2, uint8_t* cnt, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (long i = 0; i < (long)N; i++) cnt[i] = (uint8_t)orc_knn5(t, q + 3 * i, idx + K * i, d2 + K * i);
}

/* =================================================================== scan context */
orc_scan* orc_scan_create(const float* body, size_t stride, int N) {
    orc_scan* s = (orc_scan*)calloc(1, sizeof(orc_scan));
    size_t n = (size_t)(N > 0 ? N : 1);
    s->N = N;
    s->body = (float*)malloc(sizeof(float) * 3 * n);
    s->world = (float*)calloc(3 * n, sizeof(float));
    s->nn_idx = (int32_t*)malloc(sizeof(int32_t) * K * n);
    s->nn_d2 = (float*)malloc(sizeof(float) * K * n);
    s->nn_cnt = (uint8_t*)calloc(n, 1);
    s->selected = (uint8_t*)malloc(n);
    s->normvec = (float*)calloc(4 * n, sizeof(float));
    s->res_last = (float*)calloc(n, sizeof(float));
    for (int i = 0; i < N; i++) {
        s->body[3 * i + 0] = body[(size_t)i * stride + 0];
        s->body[3 * i + 1] = body[(size_t)i * stride + 1];
        s->body[3 * i + 2] = body[(size_t)i * stride + 2];
    }
    for (size_t i = 0; i < K * n; i++) { s->nn_idx[i] = -1; s->nn_d2[i] = INFINITY; }
    s->nthreads = 3; /* MP_PROC_NUM, CMakeLists.txt:21-24 */
    s->search_radius2 = 0.0;
    orc_scan_reset(s);
    return s;
}
void orc_scan_reset(orc_scan* s) {
    memset(s->selected, 1, (size_t)(s->N > 0 ? s->N : 1)); /* memset(point_selected_surf, true, ..), :812 */
    s->match_time = s->solve_time = 0;
    s->effct_feat_num = 0;
    s->total_residual = 0;
}
void orc_scan_free(orc_scan* s) {
    if (!s) return;
    free(s->body); free(s->world); free(s->nn_idx); free(s->nn_d2); free(s->nn_cnt);
    free(s->selected); free(s->normvec); free(s->res_last); free(s->h_x); free(s->h); free(s);
}

/* =================================================================== h_share_model */
/* RGBpointBodyToWorld (src/laserMapping.cpp:200-211) over a whole cloud, as publish_frame_world does for
   feats_undistort / feats_down_body (:478-530): p_global = rot * (offset_R_L_I * p_body + offset_T_L_I) + pos in double,
   stored as float. */
void orc_points_body_to_world(const double x[ORC_NSTATE], const float* pts, size_t stride_floats, size_t n, float* out_xyz) {
    const double* rot = x + X_ROT;
    const double* offR = x + X_OFFR;
    const double* offT = x + X_OFFT;
    const double* pos = x + X_POS;
    for (size_t i = 0; i < n; i++) {
        const float* pb = pts + stride_floats * i;
        double p_body[3] = {pb[0], pb[1], pb[2]};
        double t1[3], t2[3];
        orc_quat_rot(offR, p_body, t1);
        for (int d = 0; d < 3; d++) t1[d] = t1[d] + offT[d];
        orc_quat_rot(rot, t1, t2);
        for (int d = 0; d < 3; d++) out_xyz[3 * i + d] = (float)(t2[d] + pos[d]);
    }
}

int orc_h_share_model(orc_scan* sc, const orc_kdtree* map, const float* map_xyz, size_t mstride,
                      const double x[ORC_NSTATE], int converge, int extrinsic_est_en) {
    double match_start = now_s();
    const int N = sc->N;
    sc->total_residual = 0.0; /* :643 */
    const double* rot = x + X_ROT;
    const double* offR = x + X_OFFR;
    const double* offT = x + X_OFFT;
    const double* pos = x + X_POS;
#ifdef _OPENMP
    if (sc->nthreads > 0) omp_set_num_threads(sc->nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < N; i++) { /* :650-693 */
        const float* pb = sc->body + 3 * i;
        float* pw = sc->world + 3 * i;
        double p_body[3] = {pb[0], pb[1], pb[2]};
        double t1[3], t2[3], p_global[3];
        orc_quat_rot(offR, p_body, t1);
        for (int d = 0; d < 3; d++) t1[d] = t1[d] + offT[d];
        orc_quat_rot(rot, t1, t2);
        for (int d = 0; d < 3; d++) p_global[d] = t2[d] + pos[d];
        pw[0] = (float)p_global[0];
        pw[1] = (float)p_global[1];
        pw[2] = (float)p_global[2];

        int32_t* nidx = sc->nn_idx + K * i;
        float* nd2 = sc->nn_d2 + K * i;
        if (converge) { /* :667-672 */
            int cnt = orc_knn5(map, pw, nidx, nd2);
            if (sc->search_radius2 > 0.0) { /* radius-bounded variant (validation of SURVEY 8a note) */
                int c2 = 0;
                for (int j = 0; j < cnt; j++)
                    if (nd2[j] <= (float)sc->search_radius2) c2++;
                for (int j = c2; j < K; j++) { nidx[j] = -1; nd2[j] = INFINITY; }
                cnt = c2;
            }
            sc->nn_cnt[i] = (uint8_t)cnt;
            sc->selected[i] = cnt < K ? 0 : (nd2[K - 1] > 5 ? 0 : 1);
        }
        if (!sc->selected[i]) continue; /* :674 */

        float pts[15];
        for (int j = 0; j < K; j++) {
            const float* mp = map_xyz + (size_t)nidx[j] * mstride;
            pts[3 * j + 0] = mp[0]; pts[3 * j + 1] = mp[1]; pts[3 * j + 2] = mp[2];
        }
        float pabcd[4];
        sc->selected[i] = 0; /* :677 */
        if (orc_esti_plane(pts, 0.1f, pabcd)) {
            float pd2 = ((pabcd[0] * pw[0] + pabcd[1] * pw[1]) + pabcd[2] * pw[2]) + pabcd[3]; /* :680 */
            double nb = sqrt((p_body[0] * p_body[0] + p_body[1] * p_body[1]) + p_body[2] * p_body[2]);
            float s = (float)(1 - 0.9 * (double)fabsf(pd2) / sqrt(nb)); /* :681 */
            if ((double)s > 0.9) { /* :683: float compared with the double literal */
                sc->selected[i] = 1;
                sc->normvec[4 * i + 0] = pabcd[0];
                sc->normvec[4 * i + 1] = pabcd[1];
                sc->normvec[4 * i + 2] = pabcd[2];
                sc->normvec[4 * i + 3] = pd2;
                sc->res_last[i] = fabsf(pd2);
            }
        }
    }
    /* serial compaction :695-706 */
    int n_eff = 0;
    for (int i = 0; i < N; i++)
        if (sc->selected[i]) {
            sc->total_residual += sc->res_last[i];
            n_eff++;
        }
    sc->effct_feat_num = n_eff;
    if (n_eff < 1) { /* :708-713 */
        return 0;
    }
    sc->res_mean_last = sc->total_residual / n_eff; /* :715 */
    sc->match_time += now_s() - match_start;
    double solve_start = now_s();

    if (n_eff > sc->cap_rows) {
        free(sc->h_x); free(sc->h);
        sc->cap_rows = n_eff;
        sc->h_x = (double*)malloc(sizeof(double) * 12 * (size_t)n_eff);
        sc->h = (double*)malloc(sizeof(double) * (size_t)n_eff);
    }
    double rotc[4] = {-rot[0], -rot[1], -rot[2], rot[3]};
    double offRc[4] = {-offR[0], -offR[1], -offR[2], offR[3]};
    int k = 0;
    for (int i = 0; i < N; i++) { /* :723-752 (compaction fused: laserCloudOri[k] = body[i]) */
        if (!sc->selected[i]) continue;
        const float* pb = sc->body + 3 * i;
        const float* nv = sc->normvec + 4 * i;
        double pbe[3] = {pb[0], pb[1], pb[2]};
        double pthis[3];
        orc_quat_rot(offR, pbe, pthis);
        for (int d = 0; d < 3; d++) pthis[d] = pthis[d] + offT[d];
        double norm_vec[3] = {nv[0], nv[1], nv[2]};
        double C[3], A[3];
        orc_quat_rot(rotc, norm_vec, C);
        /* A = hat(point_this) * C */
        A[0] = (0.0 * C[0] + (-pthis[2]) * C[1]) + pthis[1] * C[2];
        A[1] = (pthis[2] * C[0] + 0.0 * C[1]) + (-pthis[0]) * C[2];
        A[2] = ((-pthis[1]) * C[0] + pthis[0] * C[1]) + 0.0 * C[2];
        double row[12];
        row[0] = nv[0]; row[1] = nv[1]; row[2] = nv[2];
        row[3] = A[0]; row[4] = A[1]; row[5] = A[2];
        if (extrinsic_est_en) {
            /* B = hat(point_be) * offset_R_L_I.conjugate() * C :740 -- Eigen evaluates
               (crossmat * R(q)) * C; restated as crossmat * (q^-1 rotates C) (rounding-level difference). */
            double D[3], B[3];
            orc_quat_rot(offRc, C, D);
            B[0] = (0.0 * D[0] + (-pbe[2]) * D[1]) + pbe[1] * D[2];
            B[1] = (pbe[2] * D[0] + 0.0 * D[1]) + (-pbe[0]) * D[2];
            B[2] = ((-pbe[1]) * D[0] + pbe[0] * D[1]) + 0.0 * D[2];
            row[6] = B[0]; row[7] = B[1]; row[8] = B[2];
            row[9] = C[0]; row[10] = C[1]; row[11] = C[2];
        } else {
            for (int c = 6; c < 12; c++) row[c] = 0.0; /* :745: six literal zeros */
        }
        for (int c = 0; c < 12; c++) sc->h_x[(size_t)c * n_eff + k] = row[c];
        sc->h[k] = -(double)nv[3]; /* :750 */
        k++;
    }
    sc->solve_time += now_s() - solve_start;
    return 1;
}

void orc_normal_equations(const orc_scan* sc, double HTH[144], double HTh[12]) {
    int n = sc->effct_feat_num;
    for (int a = 0; a < 12; a++) {
        for (int b = 0; b < 12; b++) {
            double s = 0;
            for (int k = 0; k < n; k++) s += sc->h_x[(size_t)a * n + k] * sc->h_x[(size_t)b * n + k];
            HTH[a * 12 + b] = s;
        }
        double s = 0;
        for (int k = 0; k < n; k++) s += sc->h_x[(size_t)a * n + k] * sc->h[k];
        HTh[a] = s;
    }
}

/* =================================================================== IEKF host algebra */
static void project_P(double* P, double* dx_new, const double x[ORC_NSTATE], const double x_prop[ORC_NSTATE],
                      const double dx[NDOF]) {
    /* esekfom.hpp:1659-1699: P <- J P J^T blockwise, dx_new <- J dx_new, J = A(dx_so3)^T / Nx*Mx */
    static const int so3_idx[2] = {3, 6};
    for (int s = 0; s < 2; s++) {
        int idx = so3_idx[s];
        double A[9], J[9];
        orc_A_matrix(dx + idx, A);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) J[i * 3 + j] = A[j * 3 + i];
        double t[3];
        for (int r = 0; r < 3; r++) t[r] = J[r * 3] * dx_new[idx] + J[r * 3 + 1] * dx_new[idx + 1] + J[r * 3 + 2] * dx_new[idx + 2];
        for (int r = 0; r < 3; r++) dx_new[idx + r] = t[r];
        for (int i = 0; i < NDOF; i++) { /* rows */
            double c0 = P[(idx)*NDOF + i], c1 = P[(idx + 1) * NDOF + i], c2 = P[(idx + 2) * NDOF + i];
            for (int r = 0; r < 3; r++) P[(idx + r) * NDOF + i] = J[r * 3] * c0 + J[r * 3 + 1] * c1 + J[r * 3 + 2] * c2;
        }
        for (int i = 0; i < NDOF; i++) { /* cols: P(i, idx:idx+3) = P(i, idx:idx+3) * J^T */
            double c0 = P[i * NDOF + idx], c1 = P[i * NDOF + idx + 1], c2 = P[i * NDOF + idx + 2];
            for (int r = 0; r < 3; r++) P[i * NDOF + idx + r] = c0 * J[r * 3] + c1 * J[r * 3 + 1] + c2 * J[r * 3 + 2];
        }
    }
    {
        int idx = 21;
        double Nx[6], Mx[6], J[4];
        orc_S2_Nx_yy(x + X_GRAV, Nx);
        orc_S2_Mx(x_prop + X_GRAV, dx + idx, Mx);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) J[i * 2 + j] = Nx[i * 3] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
        double t0 = J[0] * dx_new[idx] + J[1] * dx_new[idx + 1];
        double t1 = J[2] * dx_new[idx] + J[3] * dx_new[idx + 1];
        dx_new[idx] = t0; dx_new[idx + 1] = t1;
        for (int i = 0; i < NDOF; i++) {
            double c0 = P[idx * NDOF + i], c1 = P[(idx + 1) * NDOF + i];
            P[idx * NDOF + i] = J[0] * c0 + J[1] * c1;
            P[(idx + 1) * NDOF + i] = J[2] * c0 + J[3] * c1;
        }
        for (int i = 0; i < NDOF; i++) {
            double c0 = P[i * NDOF + idx], c1 = P[i * NDOF + idx + 1];
            P[i * NDOF + idx] = c0 * J[0] + c1 * J[1];
            P[i * NDOF + idx + 1] = c0 * J[2] + c1 * J[3];
        }
    }
}

/* The tail of one pass: dx_ = K_h + (K_x - I) dx_new ; x boxplus dx_ (esekfom.hpp:1815-1817) */
static void finish_pass(double x[ORC_NSTATE], const double K_h[NDOF], const double* K_x, const double dx_new[NDOF],
                        double dx_out[NDOF]) {
    for (int i = 0; i < NDOF; i++) {
        double s = 0;
        for (int j = 0; j < NDOF; j++) s += (K_x[i * NDOF + j] - (i == j ? 1.0 : 0.0)) * dx_new[j];
        dx_out[i] = K_h[i] + s;
    }
    orc_state_boxplus(x, dx_out);
}

static void gain_info(const double* P, double R, const double HTH[144], const double* h_x_cm, const double* h, int n_eff,
                      const double HTh_in[12], double K_h[NDOF], double* K_x) {
    /* esekfom.hpp:1782-1809 */
    double lPR[NDOF * NDOF], lPt[NDOF * NDOF], lPi[NDOF * NDOF];
    for (int i = 0; i < NDOF * NDOF; i++) lPR[i] = P[i] / R;
    orc_inverse(lPR, NDOF, lPt);
    for (int a = 0; a < 12; a++)
        for (int b = 0; b < 12; b++) lPt[a * NDOF + b] += HTH[a * 12 + b];
    orc_inverse(lPt, NDOF, lPi);
    if (h_x_cm) {
        /* K_h = P_inv[:, :12] * h_x^T * h, evaluated left to right like Eigen (n x N_eff temp) */
        double* T = (double*)malloc(sizeof(double) * (size_t)n_eff);
        for (int i = 0; i < NDOF; i++) {
            for (int k = 0; k < n_eff; k++) {
                double s = 0;
                for (int c = 0; c < 12; c++) s += lPi[i * NDOF + c] * h_x_cm[(size_t)c * n_eff + k];
                T[k] = s;
            }
            double s = 0;
            for (int k = 0; k < n_eff; k++) s += T[k] * h[k];
            K_h[i] = s;
        }
        free(T);
    } else {
        for (int i = 0; i < NDOF; i++) {
            double s = 0;
            for (int c = 0; c < 12; c++) s += lPi[i * NDOF + c] * HTh_in[c];
            K_h[i] = s;
        }
    }
    memset(K_x, 0, sizeof(double) * NDOF * NDOF);
    for (int i = 0; i < NDOF; i++)
        for (int b = 0; b < 12; b++) {
            double s = 0;
            for (int c = 0; c < 12; c++) s += lPi[i * NDOF + c] * HTH[c * 12 + b];
            K_x[i * NDOF + b] = s;
        }
}

static void gain_small(const double* P, double R, const double* h_x_cm, const double* h, int m, double K_h[NDOF], double* K_x) {
    /* esekfom.hpp:1715-1744: K = P H^T (H P H^T / R + I)^-1 / R, H zero-padded to m x 23 */
    double* H = (double*)calloc((size_t)m * NDOF, sizeof(double));
    for (int k = 0; k < m; k++)
        for (int c = 0; c < 12; c++) H[k * NDOF + c] = h_x_cm[(size_t)c * m + k];
    double* PHt = (double*)malloc(sizeof(double) * NDOF * (size_t)m);
    for (int i = 0; i < NDOF; i++)
        for (int k = 0; k < m; k++) {
            double s = 0;
            for (int j = 0; j < NDOF; j++) s += P[i * NDOF + j] * H[k * NDOF + j];
            PHt[i * m + k] = s;
        }
    double* S = (double*)malloc(sizeof(double) * (size_t)m * m);
    double* Si = (double*)malloc(sizeof(double) * (size_t)m * m);
    for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++) {
            double s = 0;
            for (int j = 0; j < NDOF; j++) s += H[a * NDOF + j] * PHt[j * m + b];
            S[a * m + b] = s / R + (a == b ? 1.0 : 0.0);
        }
    orc_inverse(S, m, Si);
    double* Kg = (double*)malloc(sizeof(double) * NDOF * (size_t)m);
    for (int i = 0; i < NDOF; i++)
        for (int k = 0; k < m; k++) {
            double s = 0;
            for (int j = 0; j < m; j++) s += PHt[i * m + j] * Si[j * m + k];
            Kg[i * m + k] = s / R;
        }
    for (int i = 0; i < NDOF; i++) {
        double s = 0;
        for (int k = 0; k < m; k++) s += Kg[i * m + k] * h[k];
        K_h[i] = s;
        for (int j = 0; j < NDOF; j++) {
            double w = 0;
            for (int k = 0; k < m; k++) w += Kg[i * m + k] * H[k * NDOF + j];
            K_x[i * NDOF + j] = w;
        }
    }
    free(H); free(PHt); free(S); free(Si); free(Kg);
}

static void final_cov(double* P, double* K_x, const double x[ORC_NSTATE], const double x_prop[ORC_NSTATE], const double dx_[NDOF]) {
    /* esekfom.hpp:1834-1924: L = P; project L, K_x, P by A(dx_)^T / Nx*Mx; P = L - K_x[:, :12] P[:12, :] */
    double L[NDOF * NDOF];
    memcpy(L, P, sizeof(L));
    static const int so3_idx[2] = {3, 6};
    for (int s = 0; s < 2; s++) {
        int idx = so3_idx[s];
        double A[9], J[9];
        orc_A_matrix(dx_ + idx, A);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) J[i * 3 + j] = A[j * 3 + i];
        for (int i = 0; i < NDOF; i++) { /* L rows from P rows */
            double c0 = P[idx * NDOF + i], c1 = P[(idx + 1) * NDOF + i], c2 = P[(idx + 2) * NDOF + i];
            for (int r = 0; r < 3; r++) L[(idx + r) * NDOF + i] = J[r * 3] * c0 + J[r * 3 + 1] * c1 + J[r * 3 + 2] * c2;
        }
        for (int i = 0; i < 12; i++) {
            double c0 = K_x[idx * NDOF + i], c1 = K_x[(idx + 1) * NDOF + i], c2 = K_x[(idx + 2) * NDOF + i];
            for (int r = 0; r < 3; r++) K_x[(idx + r) * NDOF + i] = J[r * 3] * c0 + J[r * 3 + 1] * c1 + J[r * 3 + 2] * c2;
        }
        for (int i = 0; i < NDOF; i++) {
            double c0 = L[i * NDOF + idx], c1 = L[i * NDOF + idx + 1], c2 = L[i * NDOF + idx + 2];
            for (int r = 0; r < 3; r++) L[i * NDOF + idx + r] = c0 * J[r * 3] + c1 * J[r * 3 + 1] + c2 * J[r * 3 + 2];
            c0 = P[i * NDOF + idx]; c1 = P[i * NDOF + idx + 1]; c2 = P[i * NDOF + idx + 2];
            for (int r = 0; r < 3; r++) P[i * NDOF + idx + r] = c0 * J[r * 3] + c1 * J[r * 3 + 1] + c2 * J[r * 3 + 2];
        }
    }
    {
        int idx = 21;
        double Nx[6], Mx[6], J[4];
        orc_S2_Nx_yy(x + X_GRAV, Nx);
        orc_S2_Mx(x_prop + X_GRAV, dx_ + idx, Mx);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) J[i * 2 + j] = Nx[i * 3] * Mx[0 * 2 + j] + Nx[i * 3 + 1] * Mx[1 * 2 + j] + Nx[i * 3 + 2] * Mx[2 * 2 + j];
        for (int i = 0; i < NDOF; i++) {
            double c0 = P[idx * NDOF + i], c1 = P[(idx + 1) * NDOF + i];
            L[idx * NDOF + i] = J[0] * c0 + J[1] * c1;
            L[(idx + 1) * NDOF + i] = J[2] * c0 + J[3] * c1;
        }
        for (int i = 0; i < 12; i++) {
            double c0 = K_x[idx * NDOF + i], c1 = K_x[(idx + 1) * NDOF + i];
            K_x[idx * NDOF + i] = J[0] * c0 + J[1] * c1;
            K_x[(idx + 1) * NDOF + i] = J[2] * c0 + J[3] * c1;
        }
        for (int i = 0; i < NDOF; i++) {
            double c0 = L[i * NDOF + idx], c1 = L[i * NDOF + idx + 1];
            L[i * NDOF + idx] = c0 * J[0] + c1 * J[1];
            L[i * NDOF + idx + 1] = c0 * J[2] + c1 * J[3];
            c0 = P[i * NDOF + idx]; c1 = P[i * NDOF + idx + 1];
            P[i * NDOF + idx] = c0 * J[0] + c1 * J[1];
            P[i * NDOF + idx + 1] = c0 * J[2] + c1 * J[3];
        }
    }
    double Pn[NDOF * NDOF];
    for (int i = 0; i < NDOF; i++)
        for (int j = 0; j < NDOF; j++) {
            double s = 0;
            for (int c = 0; c < 12; c++) s += K_x[i * NDOF + c] * P[c * NDOF + j];
            Pn[i * NDOF + j] = L[i * NDOF + j] - s;
        }
    memcpy(P, Pn, sizeof(Pn));
}

void orc_iekf_pass_info(double x[ORC_NSTATE], const double x_prop[ORC_NSTATE], const double P_prop[NDOF * NDOF], double R,
                        const double HTH[144], const double HTh[12], double P_out[NDOF * NDOF], double K_x_out[NDOF * NDOF],
                        double dx_out[NDOF]) {
    double dx[NDOF], dx_new[NDOF], K_h[NDOF];
    orc_state_boxminus(x, x_prop, dx);
    memcpy(dx_new, dx, sizeof(dx));
    memcpy(P_out, P_prop, sizeof(double) * NDOF * NDOF);
    project_P(P_out, dx_new, x, x_prop, dx);
    gain_info(P_out, R, HTH, NULL, NULL, 0, HTh, K_h, K_x_out);
    finish_pass(x, K_h, K_x_out, dx_new, dx_out);
}
void orc_iekf_pass_gain(double x[ORC_NSTATE], const double x_prop[ORC_NSTATE], const double P_prop[NDOF * NDOF], double R,
                        const double* h_x_cm, const double* h, int n_eff, double P_out[NDOF * NDOF], double K_x_out[NDOF * NDOF],
                        double dx_out[NDOF]) {
    double dx[NDOF], dx_new[NDOF], K_h[NDOF];
    orc_state_boxminus(x, x_prop, dx);
    memcpy(dx_new, dx, sizeof(dx));
    memcpy(P_out, P_prop, sizeof(double) * NDOF * NDOF);
    project_P(P_out, dx_new, x, x_prop, dx);
    gain_small(P_out, R, h_x_cm, h, n_eff, K_h, K_x_out);
    finish_pass(x, K_h, K_x_out, dx_new, dx_out);
}

/* Recorder for oracle/ref (record-replay against the real esekfom.hpp): called after every pass of orc_update_iterated
   with the rows h_share_model produced and the state the pass ended in. */
static orc_pass_recorder g_recorder = NULL;
static void* g_recorder_ctx = NULL;
void orc_set_pass_recorder(orc_pass_recorder cb, void* ctx) { g_recorder = cb; g_recorder_ctx = ctx; }

void orc_update_iterated(orc_scan* sc, const orc_kdtree* map, const float* map_xyz, size_t mstride, double x[ORC_NSTATE],
                         double P[NDOF * NDOF], double R, int maximum_iter, const double limit[NDOF], int extrinsic_est_en,
                         orc_update_stats* st) {
    /* esekfom.hpp:1619-1931 */
    orc_update_stats lst;
    if (!st) st = &lst;
    memset(st, 0, sizeof(*st));
    for (int i = 0; i < 8; i++) st->n_eff[i] = -1;
    int converge = 1;
    int t = 0;
    double x_prop[ORC_NSTATE], P_prop[NDOF * NDOF];
    memcpy(x_prop, x, sizeof(x_prop));
    memcpy(P_prop, P, sizeof(P_prop));
    double K_h[NDOF], K_x[NDOF * NDOF], dx_new[NDOF];
    memset(dx_new, 0, sizeof(dx_new));
    for (int i = -1; i < maximum_iter; i++) {
        double t0 = now_s();
        int valid = orc_h_share_model(sc, map, map_xyz, mstride, x, converge, extrinsic_est_en);
        st->h_time += now_s() - t0;
        if (st->passes < 8) {
            st->pass_search[st->passes] = converge;
            st->n_eff[st->passes] = sc->effct_feat_num;
        }
        st->passes++;
        st->searches += converge ? 1 : 0;
        if (!valid) {
            if (g_recorder) g_recorder(g_recorder_ctx, st->passes - 1, converge, 0, 0, NULL, NULL, x);
            continue; /* :1638-1641 */
        }
        const int rec_converge = converge;
        double solve_start = now_s();
        int dof_Measurement = sc->effct_feat_num;
        double dx[NDOF];
        orc_state_boxminus(x, x_prop, dx);
        memcpy(dx_new, dx, sizeof(dx));
        memcpy(P, P_prop, sizeof(P_prop));
        project_P(P, dx_new, x, x_prop, dx);
        if (NDOF > dof_Measurement) {
            gain_small(P, R, sc->h_x, sc->h, dof_Measurement, K_h, K_x);
        } else {
            double HTH[144], HTh[12];
            orc_normal_equations(sc, HTH, HTh);
            gain_info(P, R, HTH, sc->h_x, sc->h, dof_Measurement, HTh, K_h, K_x);
        }
        double dx_[NDOF];
        finish_pass(x, K_h, K_x, dx_new, dx_);
        converge = 1;
        for (int j = 0; j < NDOF; j++)
            if (fabs(dx_[j]) > limit[j]) { converge = 0; break; }
        if (converge) t++;
        if (!t && i == maximum_iter - 2) converge = 1; /* :1829-1832 */
        if (g_recorder) g_recorder(g_recorder_ctx, st->passes - 1, rec_converge, 1, sc->effct_feat_num, sc->h_x, sc->h, x);
        if (t > 1 || i == maximum_iter - 1) {          /* :1834 */
            final_cov(P, K_x, x, x_prop, dx_);
            st->returned_in_loop = 1;
            st->solve_time += now_s() - solve_start;
            return;
        }
        st->solve_time += now_s() - solve_start;
    }
}

/* =================================================================== map_incremental decision */
void orc_map_incremental_classify(const orc_scan* sc, const float* map_xyz, size_t mstride, const double x[ORC_NSTATE],
                                  double fsm, int flg_EKF_inited, float* world_out, uint8_t* cls) {
    /* src/laserMapping.cpp:427-474.  filter_size_map_min is a double global; mid_point members are
       float (PointType), so each mid coordinate is a double expression narrowed to float. */
    const int N = sc->N;
    for (int i = 0; i < N; i++) {
        const float* pb = sc->body + 3 * i;
        double p_body[3] = {pb[0], pb[1], pb[2]}, t1[3], t2[3];
        orc_quat_rot(x + X_OFFR, p_body, t1); /* pointBodyToWorld :166-176 */
        for (int d = 0; d < 3; d++) t1[d] = t1[d] + x[X_OFFT + d];
        orc_quat_rot(x + X_ROT, t1, t2);
        float pw[3];
        for (int d = 0; d < 3; d++) {
            pw[d] = (float)(t2[d] + x[X_POS + d]);
            world_out[3 * i + d] = pw[d];
        }
        int cnt = sc->nn_cnt[i];
        if (cnt > 0 && flg_EKF_inited) {
            float mid[3];
            for (int d = 0; d < 3; d++) mid[d] = (float)(floor((double)pw[d] / fsm) * fsm + 0.5 * fsm);
            float dist = ((pw[0] - mid[0]) * (pw[0] - mid[0]) + (pw[1] - mid[1]) * (pw[1] - mid[1])) + (pw[2] - mid[2]) * (pw[2] - mid[2]);
            const float* n0 = map_xyz + (size_t)sc->nn_idx[K * i] * mstride;
            if (fabs((double)(n0[0] - mid[0])) > 0.5 * fsm && fabs((double)(n0[1] - mid[1])) > 0.5 * fsm &&
                fabs((double)(n0[2] - mid[2])) > 0.5 * fsm) {
                cls[i] = 2;
                continue;
            }
            int need_add = 1;
            for (int r = 0; r < K; r++) {
                if (cnt < K) break;
                const float* pn = map_xyz + (size_t)sc->nn_idx[K * i + r] * mstride;
                float dn = ((pn[0] - mid[0]) * (pn[0] - mid[0]) + (pn[1] - mid[1]) * (pn[1] - mid[1])) + (pn[2] - mid[2]) * (pn[2] - mid[2]);
                if (dn < dist) { need_add = 0; break; }
            }
            cls[i] = need_add ? 1 : 0;
        } else {
            cls[i] = 1;
        }
    }
}

/* =================================================================== incremental map (ikd-Tree stand-in) */
typedef struct { long long kx, ky, kz; } vox_key;
static int vox_eq(vox_key a, vox_key b) { return a.kx == b.kx && a.ky == b.ky && a.kz == b.kz; }
static unsigned long long vox_hash(vox_key k) {
    unsigned long long h = (unsigned long long)k.kx * 0x9E3779B97F4A7C15ull;
    h ^= (unsigned long long)k.ky * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= (unsigned long long)k.kz * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    return h;
}
static vox_key vox_of(const float* p, double ds) {
    /* Box_of_Point.vertex_min = floor(p/ds)*ds in the reference's float arithmetic is evaluated on doubles here
       (filter_size_map_min is a double global); the integer voxel index is what matters. */
    vox_key k;
    k.kx = (long long)floor((double)p[0] / ds);
    k.ky = (long long)floor((double)p[1] / ds);
    k.kz = (long long)floor((double)p[2] / ds);
    return k;
}
static float dist_to_center(const float* p, vox_key k, double ds) {
    /* calc_dist(point, mid_point) with mid_point (float members) = min + 0.5*ds */
    float mx = (float)((double)k.kx * ds + 0.5 * ds), my = (float)((double)k.ky * ds + 0.5 * ds), mz = (float)((double)k.kz * ds + 0.5 * ds);
    float dx = p[0] - mx, dy = p[1] - my, dz = p[2] - mz;
    return (dx * dx + dy * dy) + dz * dz;
}

size_t orc_map_add(float* map, size_t M, const float* add, size_t n, int downsample, double ds) {
    if (!downsample) {
        memcpy(map + 3 * M, add, sizeof(float) * 3 * n);
        return M + n;
    }
    /* voxel hash over ALL points (old + new): chain heads per bucket */
    size_t tot = M + n, nb = 1;
    while (nb < 2 * tot + 16) nb <<= 1;
    long long* head = (long long*)malloc(sizeof(long long) * nb);
    long long* next = (long long*)malloc(sizeof(long long) * (tot ? tot : 1));
    unsigned char* dead = (unsigned char*)calloc(tot ? tot : 1, 1);
    for (size_t i = 0; i < nb; i++) head[i] = -1;
    memcpy(map + 3 * M, add, sizeof(float) * 3 * n);
    for (size_t i = 0; i < M; i++) {
        vox_key k = vox_of(map + 3 * i, ds);
        size_t b = (size_t)(vox_hash(k) & (nb - 1));
        next[i] = head[b];
        head[b] = (long long)i;
    }
    /* sequential semantics of the reference loop, one new point at a time */
    for (size_t a = 0; a < n; a++) {
        size_t i = M + a;
        const float* p = map + 3 * i;
        vox_key k = vox_of(p, ds);
        size_t b = (size_t)(vox_hash(k) & (nb - 1));
        float min_dist = dist_to_center(p, k, ds);
        long long best = (long long)i;
        int stored = 0;
        for (long long j = head[b]; j >= 0; j = next[j]) {
            if (dead[j] || !vox_eq(vox_of(map + 3 * j, ds), k)) continue;
            stored++;
            float d = dist_to_center(map + 3 * j, k, ds);
            /* strict '<': the new point keeps a tie.  Among EXISTING points at equal distance the reference keeps the
               first in tree order (arbitrary); pinned here to the lowest index. */
            if (d < min_dist || (d == min_dist && best != (long long)i && j < best)) { min_dist = d; best = j; }
        }
        if (stored > 1 || best == (long long)i) {
            /* Delete_by_range(box) + Add_by_point(downsample_result) */
            for (long long j = head[b]; j >= 0; j = next[j])
                if (!dead[j] && vox_eq(vox_of(map + 3 * j, ds), k) && j != best) dead[j] = 1;
            if (best == (long long)i) { /* the new point enters the map */
                next[i] = head[b];
                head[b] = (long long)i;
            } else {
                dead[i] = 1;
            }
        } else {
            dead[i] = 1; /* the single existing point stays; the new one is dropped */
        }
    }
    size_t w = 0;
    for (size_t i = 0; i < tot; i++)
        if (!dead[i]) {
            if (w != i) memmove(map + 3 * w, map + 3 * i, sizeof(float) * 3);
            w++;
        }
    free(head); free(next); free(dead);
    return w;
}

/* SENSITIVITY VARIANT of orc_map_add (downsample = true), not used by any parity test: the box arithmetic as ikd-Tree
   itself carries it [recalled-upstream: KD_TREE::Add_Points; source absent, .gitmodules:1-4] -- `float downsample_size`,
   Box_of_Point.vertex_min = floor(p / downsample_size) * downsample_size and vertex_max = vertex_min + downsample_size in
   FLOAT, membership by Search_by_range's  vertex_min <= p < vertex_max  on those float corners, mid_point =
   min + (max - min) / 2.0.  At downsample_size = 0.5 (every launch file but marsim) the float corners equal the double
   voxel grid of orc_map_add exactly; at marsim's 0.3 they do not, and a box can reach into a neighbouring voxel.
   tools/eigen_order_study.py reports how many surviving points differ between the two. */
size_t orc_map_add_floatbox(float* map, size_t M, const float* add, size_t n, float dsf) {
    const double ds = (double)dsf;
    size_t tot = M + n, nb = 1;
    while (nb < 2 * tot + 16) nb <<= 1;
    long long* head = (long long*)malloc(sizeof(long long) * nb);
    long long* next = (long long*)malloc(sizeof(long long) * (tot ? tot : 1));
    unsigned char* dead = (unsigned char*)calloc(tot ? tot : 1, 1);
    for (size_t i = 0; i < nb; i++) head[i] = -1;
    memcpy(map + 3 * M, add, sizeof(float) * 3 * n);
    for (size_t i = 0; i < M; i++) { /* coarse buckets (double grid): a float box overlaps at most 3 of them per axis */
        vox_key k = vox_of(map + 3 * i, ds);
        size_t b = (size_t)(vox_hash(k) & (nb - 1));
        next[i] = head[b];
        head[b] = (long long)i;
    }
    for (size_t a = 0; a < n; a++) {
        size_t i = M + a;
        const float* p = map + 3 * i;
        float bmin[3], bmax[3], mid[3];
        for (int d = 0; d < 3; d++) {
            bmin[d] = floorf(p[d] / dsf) * dsf;
            bmax[d] = bmin[d] + dsf;
            mid[d] = (float)((double)bmin[d] + (double)(bmax[d] - bmin[d]) / 2.0);
        }
        float dx = p[0] - mid[0], dy = p[1] - mid[1], dz = p[2] - mid[2];
        float min_dist = (dx * dx + dy * dy) + dz * dz;
        long long best = (long long)i;
        int stored = 0;
        vox_key kc = vox_of(p, ds);
        for (long long oz = -1; oz <= 1; oz++)
            for (long long oy = -1; oy <= 1; oy++)
                for (long long ox = -1; ox <= 1; ox++) {
                    vox_key k = {kc.kx + ox, kc.ky + oy, kc.kz + oz};
                    size_t b = (size_t)(vox_hash(k) & (nb - 1));
                    for (long long j = head[b]; j >= 0; j = next[j]) {
                        const float* q = map + 3 * j;
                        if (dead[j] || !vox_eq(vox_of(q, ds), k)) continue;
                        if (!(q[0] >= bmin[0] && q[0] < bmax[0] && q[1] >= bmin[1] && q[1] < bmax[1] && q[2] >= bmin[2] && q[2] < bmax[2])) continue;
                        stored++;
                        float ex = q[0] - mid[0], ey = q[1] - mid[1], ez = q[2] - mid[2];
                        float d = (ex * ex + ey * ey) + ez * ez;
                        if (d < min_dist || (d == min_dist && best != (long long)i && j < best)) { min_dist = d; best = j; }
                    }
                }
        if (stored > 1 || best == (long long)i) {
            for (long long oz = -1; oz <= 1; oz++)
                for (long long oy = -1; oy <= 1; oy++)
                    for (long long ox = -1; ox <= 1; ox++) {
                        vox_key k = {kc.kx + ox, kc.ky + oy, kc.kz + oz};
                        size_t b = (size_t)(vox_hash(k) & (nb - 1));
                        for (long long j = head[b]; j >= 0; j = next[j]) {
                            const float* q = map + 3 * j;
                            if (dead[j] || j == best || !vox_eq(vox_of(q, ds), k)) continue;
                            if (q[0] >= bmin[0] && q[0] < bmax[0] && q[1] >= bmin[1] && q[1] < bmax[1] && q[2] >= bmin[2] && q[2] < bmax[2]) dead[j] = 1;
                        }
                    }
            if (best == (long long)i) {
                size_t b = (size_t)(vox_hash(kc) & (nb - 1));
                next[i] = head[b];
                head[b] = (long long)i;
            } else {
                dead[i] = 1;
            }
        } else {
            dead[i] = 1;
        }
    }
    size_t w = 0;
    for (size_t i = 0; i < tot; i++)
        if (!dead[i]) {
            if (w != i) memmove(map + 3 * w, map + 3 * i, sizeof(float) * 3);
            w++;
        }
    free(head); free(next); free(dead);
    return w;
}

size_t orc_map_delete_boxes(float* map, size_t M, const float* boxes, size_t nb) {
    size_t w = 0;
    for (size_t i = 0; i < M; i++) {
        const float* p = map + 3 * i;
        int del = 0;
        for (size_t b = 0; b < nb && !del; b++) {
            const float* bx = boxes + 6 * b;
            if (p[0] >= bx[0] && p[0] < bx[3] && p[1] >= bx[1] && p[1] < bx[4] && p[2] >= bx[2] && p[2] < bx[5]) del = 1;
        }
        if (!del) {
            if (w != i) memmove(map + 3 * w, p, sizeof(float) * 3);
            w++;
        }
    }
    return w;
}

/* =================================================================== lasermap_fov_segment (src/laserMapping.cpp:230-280)
   The local-map cube bookkeeping: float box corners (BoxPointType), float distances, a float mov_dist narrowed from a
   double max() -- types as declared at :77-78,91.  Returns the number of boxes written to boxes_out (<= 3, each
   {min xyz, max xyz}) that the caller hands to Delete_Point_Boxes. */
int orc_fov_segment(orc_local_map* lm, const double pos_lid[3], double cube_len, float det_range, float* boxes_out) {
    const float MOV_THRESHOLD = 1.5f;
    if (!lm->initialized) { /* :238-245 */
        for (int i = 0; i < 3; i++) {
            lm->vertex_min[i] = (float)(pos_lid[i] - cube_len / 2.0);
            lm->vertex_max[i] = (float)(pos_lid[i] + cube_len / 2.0);
        }
        lm->initialized = 1;
        return 0;
    }
    float dist_to_map_edge[3][2];
    int need_move = 0;
    for (int i = 0; i < 3; i++) { /* :248-252 */
        dist_to_map_edge[i][0] = (float)fabs(pos_lid[i] - (double)lm->vertex_min[i]);
        dist_to_map_edge[i][1] = (float)fabs(pos_lid[i] - (double)lm->vertex_max[i]);
        if (dist_to_map_edge[i][0] <= MOV_THRESHOLD * det_range || dist_to_map_edge[i][1] <= MOV_THRESHOLD * det_range) need_move = 1;
    }
    if (!need_move) return 0;
    float nmin[3], nmax[3];
    for (int i = 0; i < 3; i++) { nmin[i] = lm->vertex_min[i]; nmax[i] = lm->vertex_max[i]; }
    const double m1 = (cube_len - 2.0 * MOV_THRESHOLD * det_range) * 0.5 * 0.9, m2 = (double)(det_range * (MOV_THRESHOLD - 1));
    const float mov_dist = (float)(m1 > m2 ? m1 : m2); /* :256 (std::max returns its first argument on ties) */
    int nb = 0;
    for (int i = 0; i < 3; i++) { /* :257-270 */
        float tmin[3], tmax[3];
        for (int d = 0; d < 3; d++) { tmin[d] = lm->vertex_min[d]; tmax[d] = lm->vertex_max[d]; }
        int push = 0;
        if (dist_to_map_edge[i][0] <= MOV_THRESHOLD * det_range) {
            nmax[i] -= mov_dist;
            nmin[i] -= mov_dist;
            tmin[i] = lm->vertex_max[i] - mov_dist;
            push = 1;
        } else if (dist_to_map_edge[i][1] <= MOV_THRESHOLD * det_range) {
            nmax[i] += mov_dist;
            nmin[i] += mov_dist;
            tmax[i] = lm->vertex_min[i] + mov_dist;
            push = 1;
        }
        if (push) {
            for (int d = 0; d < 3; d++) { boxes_out[6 * nb + d] = tmin[d]; boxes_out[6 * nb + 3 + d] = tmax[d]; }
            nb++;
        }
    }
    for (int i = 0; i < 3; i++) { lm->vertex_min[i] = nmin[i]; lm->vertex_max[i] = nmax[i]; }
    return nb;
}

/* =================================================================== pcl::VoxelGrid<PointType>::applyFilter
   (downSizeFilterSurf.filter, src/laserMapping.cpp:904-905; leaf = filter_size_surf_min narrowed to float by
   setLeafSize, :813).  PCL is an external dependency (README.md:69, PCL >= 1.8); restated from its published
   algorithm (filters/impl/voxel_grid.hpp): float min/max of the cloud, integer voxel coordinates
   floor(x * inverse_leaf) - min_b, linear index, sort by index, one centroid per occupied voxel in index order,
   xyz averaged in float (CentroidPoint / AccumulatorXYZ: running float sum, then sum / n).
   PCL sorts with an UNSTABLE sort that compares the voxel index only, so the order in which a voxel's points are
   summed is implementation-defined there; it is pinned here to ascending input index.  Centroids therefore agree
   with a given PCL build up to the rounding of a different float summation order, not bit for bit.
   Returns the number of output points; out_xyz needs room for n points.  If the grid would overflow int32 PCL
   warns and returns the input unchanged -- so does this. */
typedef struct { unsigned int idx; unsigned int pt; } vg_pair;
static int vg_cmp(const void* a, const void* b) {
    const vg_pair* x = (const vg_pair*)a; const vg_pair* y = (const vg_pair*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);
}
size_t orc_voxel_grid(const float* in, size_t stride, size_t n, float leaf, float* out_xyz) {
    if (n == 0) return 0;
    float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            float v = in[i * stride + d];
            if (v < min_p[d]) min_p[d] = v;
            if (v > max_p[d]) max_p[d] = v;
        }
    const float inv = 1.0f / leaf;
    long long dxyz[3];
    for (int d = 0; d < 3; d++) dxyz[d] = (long long)((max_p[d] - min_p[d]) * inv) + 1;
    if (dxyz[0] * dxyz[1] * dxyz[2] > (long long)INT_MAX) {
        for (size_t i = 0; i < n; i++)
            for (int d = 0; d < 3; d++) out_xyz[3 * i + d] = in[i * stride + d];
        return n;
    }
    int min_b[3], max_b[3], div_b[3], mul[3];
    for (int d = 0; d < 3; d++) {
        min_b[d] = (int)floorf(min_p[d] * inv);
        max_b[d] = (int)floorf(max_p[d] * inv);
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
    vg_pair* pr = (vg_pair*)malloc(sizeof(vg_pair) * n);
    for (size_t i = 0; i < n; i++) {
        int ijk[3];
        for (int d = 0; d < 3; d++) ijk[d] = (int)(floorf(in[i * stride + d] * inv) - (float)min_b[d]);
        int idx = ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2];
        pr[i].idx = (unsigned int)idx;
        pr[i].pt = (unsigned int)i;
    }
    qsort(pr, n, sizeof(vg_pair), vg_cmp);
    size_t m = 0, first = 0;
    while (first < n) {
        size_t last = first + 1;
        while (last < n && pr[last].idx == pr[first].idx) last++;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (size_t k = first; k < last; k++) {
            const float* p = in + (size_t)pr[k].pt * stride;
            sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2];
        }
        const float cnt = (float)(last - first);
        out_xyz[3 * m] = sx / cnt; out_xyz[3 * m + 1] = sy / cnt; out_xyz[3 * m + 2] = sz / cnt;
        m++;
        first = last;
    }
    free(pr);
    return m;
}

/* =================================================================== ImuProcess::UndistortPcl, per-point half
   (src/IMU_Processing.hpp:307-349): the backward sweep that moves every LiDAR point to the scan-end frame.  The
   forward half (:240-300, one kf_state.predict per IMU sample) is host work on orc_predict and produces `poses`
   (IMUpose, msg/Pose6D.msg) and the scan-end state `x_end`.
   The reference first sorts the cloud by curvature (= time offset in ms, :234) and sweeps segments from the back;
   since every point is handled independently the sweep reduces to: segment k = the LAST k <= n_pose-2 with
   poses[k].offset_time < t (a segment skipped for a later point has offset_time >= that point's time, hence >= every
   earlier point's too); a point no segment claims is left untouched, as the sweep never reaches it.
   THE FIRST POINT (:345): the reference's inner loop leaves through `if (it_pcl == begin) break` WITHOUT stepping past the
   first point, so the earliest point of the (time-sorted) cloud -- and only that one -- is compensated AGAIN by every earlier
   segment whose head is older than it, each time on its already-moved float coordinates: it ends up carried by every segment
   k = n_pose-2 .. 0 with poses[k].offset_time < t, in that order.  That happens only when the earliest point is younger than
   IMUpose[1] (> ~5 ms into the scan at 200 Hz); it is an artefact of the loop, but the contract is "identical to the
   reference on the same inputs", so it is reproduced by default (orc_set_undistort_first(0) switches it off: every point
   carried once).  "Earliest" = the smallest time offset; among equal ones the lowest input index (the reference's sort is
   std::sort, whose order of equal keys is unspecified).  No ordering of the offset_times is assumed: IMUpose[1] precedes
   IMUpose[0] = 0 whenever the first IMU sample is older than the first point.  The order of the output is the order of the
   input: the reference's sort order is not reproduced.
   Exp() is so3_math.h:36-58.  Double arithmetic in source order; Eigen's internal evaluation order of the 3x3
   products is not modelled (it moves results by ~1e-16 before the final narrowing to float). */
static void und_exp(const double w[3], double dt, double R[9]) {
    const double n = sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    if (n > 0.0000001) {
        const double a[3] = {w[0] / n, w[1] / n, w[2] / n};
        const double Kx[9] = {0.0, -a[2], a[1], a[2], 0.0, -a[0], -a[1], a[0], 0.0};
        const double ang = n * dt, s = sin(ang), c1 = 1.0 - cos(ang);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double kk = 0.0;
                for (int k = 0; k < 3; k++) kk = kk + (c1 * Kx[3 * i + k]) * Kx[3 * k + j];
                R[3 * i + j] = (R[3 * i + j] + s * Kx[3 * i + j]) + kk;
            }
    }
}
static int g_undistort_first = 1;
void orc_set_undistort_first(int on) { g_undistort_first = on ? 1 : 0; }
/* one pass of the inner loop's body (:326-343) for a point at time t with segment k; xyz are float members of the point */
static void und_apply(const orc_pose6d* poses, int k, const double x_end[ORC_NSTATE], double t, float xyz[3]) {
    const double* pos_e = x_end + X_POS;
    const double* rot_e = x_end + X_ROT;
    const double* offR = x_end + X_OFFR;
    const double* offT = x_end + X_OFFT;
    const double rot_c[4] = {-rot_e[0], -rot_e[1], -rot_e[2], rot_e[3]};
    const double offR_c[4] = {-offR[0], -offR[1], -offR[2], offR[3]};
    const orc_pose6d* head = poses + k;
    const orc_pose6d* tail = poses + k + 1;
    const double dt = t - head->offset_time;
    double E[9], R_i[9];
    und_exp(tail->gyr, dt, E); /* angvel_avr = tail->gyr */
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double a = 0.0;
            for (int m = 0; m < 3; m++) a = a + head->rot[3 * r + m] * E[3 * m + c];
            R_i[3 * r + c] = a;
        }
    const double P_i[3] = {xyz[0], xyz[1], xyz[2]};
    double T_ei[3], q1[3], q2[3], q3[3], q4[3];
    for (int d = 0; d < 3; d++)
        T_ei[d] = ((head->pos[d] + head->vel[d] * dt) + ((0.5 * tail->acc[d]) * dt) * dt) - pos_e[d];
    orc_quat_rot(offR, P_i, q1);
    for (int d = 0; d < 3; d++) q1[d] = q1[d] + offT[d];
    for (int r = 0; r < 3; r++) q2[r] = ((R_i[3 * r] * q1[0] + R_i[3 * r + 1] * q1[1]) + R_i[3 * r + 2] * q1[2]) + T_ei[r];
    orc_quat_rot(rot_c, q2, q3);
    for (int d = 0; d < 3; d++) q3[d] = q3[d] - offT[d];
    orc_quat_rot(offR_c, q3, q4);
    for (int d = 0; d < 3; d++) xyz[d] = (float)q4[d];
}
void orc_undistort(const orc_pose6d* poses, int n_pose, const double x_end[ORC_NSTATE], const float* pts, size_t stride,
                   size_t time_off, size_t n, float* out_xyz) {
    size_t first = 0; /* the earliest point (lowest index among equal times) */
    for (size_t i = 1; i < n; i++)
        if (pts[i * stride + time_off] < pts[first * stride + time_off]) first = i;
    for (size_t i = 0; i < n; i++) {
        const float* p = pts + i * stride;
        out_xyz[3 * i] = p[0]; out_xyz[3 * i + 1] = p[1]; out_xyz[3 * i + 2] = p[2];
        const double t = (double)p[time_off] / (double)1000; /* it_pcl->curvature / double(1000) */
        for (int j = n_pose - 2; j >= 0; j--)
            if (t > poses[j].offset_time) {
                und_apply(poses, j, x_end, t, out_xyz + 3 * i);
                if (!(g_undistort_first && i == first)) break; /* every point but the first leaves the sweep here (:326, :345) */
            }
    }
}

/* =================================================================== ImuProcess::UndistortPcl, forward half
   (src/IMU_Processing.hpp:217-300): one predict per pair of consecutive IMU samples; IMUpose records the state at every
   IMU sample; the filter ends at the scan end.  imu: n rows of {t, acc[3], gyr[3]} -- this scan's samples; the previous
   scan's last sample (st->last_imu) is prepended as the reference does (:220). */
static void q_to_R(const double q[4], double R[9]) { /* Eigen::Quaternion::toRotationMatrix */
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void imu_pose(orc_pose6d* kp, double t, const double a[3], const double g[3], const double x[ORC_NSTATE]) {
    kp->offset_time = t;
    for (int i = 0; i < 3; i++) { kp->acc[i] = a[i]; kp->gyr[i] = g[i]; kp->vel[i] = x[X_VEL + i]; kp->pos[i] = x[X_POS + i]; }
    q_to_R(x + X_ROT, kp->rot);
}
int orc_imu_forward(orc_imu_state* st, const double* imu, int n, double pcl_beg_time, double pcl_end_time,
                    double x[ORC_NSTATE], double P[ORC_NDOF * ORC_NDOF], orc_pose6d* poses_out) {
    const double G_m_s2 = 9.81;
    const int nv = n + 1;
    double* v = (double*)malloc(sizeof(double) * 7 * (size_t)nv);
    memcpy(v, st->last_imu, sizeof(double) * 7);
    memcpy(v + 7, imu, sizeof(double) * 7 * (size_t)n);
    const double imu_end_time = v[7 * (nv - 1)];
    double Q[144];
    orc_process_noise_cov(Q);
    int np = 0;
    imu_pose(poses_out + np++, 0.0, st->acc_s_last, st->angvel_last, x);
    double acc_avr[3] = {0, 0, 0}, angvel_avr[3] = {0, 0, 0}, dt = 0;
    const double mnorm = sqrt((st->mean_acc[0] * st->mean_acc[0] + st->mean_acc[1] * st->mean_acc[1]) + st->mean_acc[2] * st->mean_acc[2]);
    for (int k = 0; k + 1 < nv; k++) {
        const double* head = v + 7 * k;
        const double* tail = v + 7 * (k + 1);
        if (tail[0] < st->last_lidar_end_time) continue;
        for (int i = 0; i < 3; i++) {
            angvel_avr[i] = 0.5 * (head[4 + i] + tail[4 + i]);
            acc_avr[i] = 0.5 * (head[1 + i] + tail[1 + i]);
        }
        for (int i = 0; i < 3; i++) acc_avr[i] = acc_avr[i] * G_m_s2 / mnorm;
        if (head[0] < st->last_lidar_end_time) dt = tail[0] - st->last_lidar_end_time;
        else dt = tail[0] - head[0];
        for (int i = 0; i < 3; i++) {
            Q[(0 + i) * 12 + (0 + i)] = st->cov_gyr[i]; Q[(3 + i) * 12 + (3 + i)] = st->cov_acc[i];
            Q[(6 + i) * 12 + (6 + i)] = st->cov_bias_gyr[i]; Q[(9 + i) * 12 + (9 + i)] = st->cov_bias_acc[i];
        }
        orc_predict(x, P, dt, Q, acc_avr, angvel_avr);
        double unb[3], as[3];
        for (int i = 0; i < 3; i++) { st->angvel_last[i] = angvel_avr[i] - x[X_BG + i]; unb[i] = acc_avr[i] - x[X_BA + i]; }
        orc_quat_rot(x + X_ROT, unb, as);
        for (int i = 0; i < 3; i++) st->acc_s_last[i] = as[i] + x[X_GRAV + i];
        imu_pose(poses_out + np++, tail[0] - pcl_beg_time, st->acc_s_last, st->angvel_last, x);
    }
    const double note = pcl_end_time > imu_end_time ? 1.0 : -1.0;
    dt = note * (pcl_end_time - imu_end_time);
    orc_predict(x, P, dt, Q, acc_avr, angvel_avr);
    memcpy(st->last_imu, imu + 7 * (size_t)(n - 1), sizeof(double) * 7);
    st->last_lidar_end_time = pcl_end_time;
    free(v);
    return np;
}
