/*
 * fastlio_oracle.h -- CPU restatement (plain C) of FAST-LIO2's per-scan measurement update.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / timed CPU baseline.  The product path (fast_lio_amd/, include/) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path
 *  and cannot be compiled here (no Eigen/PCL/Boost/ROS; ikd-Tree is an empty
 * submodule -- SURVEY.md F1/F2).  This file restates the algorithm from the cited reference lines;
 * the Eigen routines it depends on (ColPivHouseholderQR, PartialPivLU inverse, Quaternion ops) are
 * restated from their published algorithms (Eigen >= 3.3.4, README.md:74 of the reference).
 *
 * All citations are file:line under /root/reference.
 */
#ifndef FASTLIO_ORACLE_H
#define FASTLIO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NDOF 23  /* state_ikfom::DOF, include/use-ikfom.hpp:12-21 */
#define ORC_NSTATE 26 /* flat storage: pos3 rot4(xyzw) offR4(xyzw) offT3 vel3 bg3 ba3 grav3 */
#define ORC_K 5      /* NUM_MATCH_POINTS, include/common_lib.h:26 */

/* Flat state layout (doubles):
 *   [0:3) pos | [3:7) rot xyzw | [7:11) offset_R_L_I xyzw | [11:14) offset_T_L_I | [14:17) vel |
 *   [17:20) bg | [20:23) ba | [23:26) grav (S2, |g| = 9.809)
 * DOF layout (include/use-ikfom.hpp:12-21): pos 0-2 | rot 3-5 | offR 6-8 | offT 9-11 | vel 12-14 |
 *   bg 15-17 | ba 18-20 | grav 21-22.  Covariances are 23x23 row-major. */

/* ---- exact 5-NN (stand-in for ikd-Tree Nearest_Search, src/laserMapping.cpp:670) ---- */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree* orc_kdtree_build(const float* xyz, size_t stride_floats, size_t M);
void orc_kdtree_free(orc_kdtree* t);
size_t orc_kdtree_size(const orc_kdtree* t);
/* Returns number found (min(5,M)); neighbours ascending by (d2, original index).
 * d2 = ((dx*dx + dy*dy) + dz*dz) in fp32, no contraction. */
int orc_knn5(const orc_kdtree* t, const float q[3], int32_t idx[ORC_K], float d2[ORC_K]);
int orc_knn5_brute(const float* xyz, size_t stride_floats, size_t M, const float q[3],
                   int32_t idx[ORC_K], float d2[ORC_K]);
/* Batch helpers (OpenMP over queries). */
void orc_knn5_batch(const orc_kdtree* t, const float* q_xyz, size_t N, int32_t* idx, float* d2,
                    uint8_t* cnt, int nthreads);

/* ---- esti_plane<float> (include/common_lib.h:225-257) ---- */
/* pts: 5x3 fp32 row-major.  Returns 1 if all 5 points are within `threshold` of the plane. */
int orc_esti_plane(const float pts[15], float threshold, float pabcd[4]);
/* Which fp32 summation order the restated Eigen reductions use (oracle_math.c, "SUMMATION ORDER"); process-wide, set it
 * before any fit.  Default ORC_ORDER_SSE = Eigen 3.3.x on x86-64 with SSE2, the reference's own build. */
enum { ORC_ORDER_SEQ = 0, ORC_ORDER_SSE = 1, ORC_ORDER_PAIRWISE = 2, ORC_ORDER_NOVEC = 3 };
void orc_set_eigen_order(int order);
int orc_get_eigen_order(void);
/* The restated Eigen ColPivHouseholderQR 5x3 solve of A x = b (b = -1), exposed for KATs. */
void orc_qr_solve_5x3(const float A[15], const float b[5], float x[3]);

/* ---- manifold helpers (IKFoM_toolkit/mtk), exposed for KATs ---- */
void orc_A_matrix(const double v[3], double A[9]);                 /* mtkmath.hpp:235-247 */
void orc_so3_exp(const double v[3], double scale, double q_xyzw[4]); /* SOn.hpp:284-288 */
void orc_so3_log(const double q_xyzw[4], double v[3]);              /* SOn.hpp:293-297 */
void orc_quat_mul(const double a[4], const double b[4], double out[4]);
void orc_quat_rot(const double q[4], const double v[3], double out[3]);
void orc_S2_Bx(const double g[3], double Bx[6]);                    /* S2.hpp:215-231 (typ 1), 3x2 row-major */
void orc_S2_Nx_yy(const double g[3], double Nx[6]);                 /* S2.hpp:259-264, 2x3 */
void orc_S2_Mx(const double g[3], const double delta[2], double Mx[6]); /* S2.hpp:266-280, 3x2 */
void orc_S2_boxplus(double g[3], const double delta[2]);            /* S2.hpp:136-142 */
void orc_S2_boxminus(const double g[3], const double other[3], double res[2]); /* S2.hpp:144-167 */
void orc_state_boxplus(double x[ORC_NSTATE], const double dx[ORC_NDOF]);   /* build_manifold.hpp:192-194 */
void orc_state_boxminus(const double x[ORC_NSTATE], const double y[ORC_NSTATE], double dx[ORC_NDOF]);
/* n x n inverse via partial-pivot LU (stand-in for Eigen inverse(), esekfom.hpp:1782,1802,1738).
 * Row-major.  Returns 0 on success. */
int orc_inverse(const double* A, int n, double* Ainv);

/* ---- esekf::predict (esekfom.hpp:279-383) with get_f/df_dx/df_dw (use-ikfom.hpp:47-88).
 * Only used to manufacture the prior (x^-, P^-) for tests/bench (SURVEY.md 8d "Prior"). ---- */
void orc_predict(double x[ORC_NSTATE], double P[ORC_NDOF * ORC_NDOF], double dt,
                 const double Q[12 * 12], const double acc[3], const double gyro[3]);
void orc_process_noise_cov(double Q[12 * 12]); /* use-ikfom.hpp:35-43 */
void orc_init_P(double P[ORC_NDOF * ORC_NDOF]); /* src/IMU_Processing.hpp:204-210 */

/* ---- scan context: the globals h_share_model reads and writes (src/laserMapping.cpp:76-114) ---- */
typedef struct orc_scan {
    int N;                /* feats_down_size */
    float* body;          /* N x 3, feats_down_body xyz */
    float* world;         /* N x 3, feats_down_world xyz (written by h) */
    int32_t* nn_idx;      /* N x 5, Nearest_Points (as map indices) */
    float* nn_d2;         /* N x 5, pointSearchSqDis */
    uint8_t* nn_cnt;      /* N,     Nearest_Points[i].size() */
    uint8_t* selected;    /* N,     point_selected_surf (initialised to 1, laserMapping.cpp:812) */
    float* normvec;       /* N x 4, (a,b,c,pd2) */
    float* res_last;      /* N */
    /* outputs of the last h call */
    int effct_feat_num;
    double total_residual;
    double res_mean_last;
    double* h_x;          /* n_eff x 12, column-major (ekfom_data.h_x) */
    double* h;            /* n_eff */
    int cap_rows;
    /* timing buckets (seconds, accumulated): match / build-H (laserMapping.cpp:640,716-717,753) */
    double match_time, solve_time;
    int nthreads;         /* OpenMP threads for the point loop (reference: MP_PROC_NUM=3) */
    double search_radius2; /* <=0: unbounded exact kNN; >0: discard neighbours with d2 > r2
                              (used to validate the radius-bounded equivalence, SURVEY 8a note) */
} orc_scan;

orc_scan* orc_scan_create(const float* body_xyz, size_t stride_floats, int N);
void orc_scan_free(orc_scan* s);
void orc_scan_reset(orc_scan* s); /* selected[] = 1, timers = 0 */

/* h_share_model (src/laserMapping.cpp:638-754).  Returns ekfom_data.valid (0/1). */
int orc_h_share_model(orc_scan* sc, const orc_kdtree* map, const float* map_xyz,
                      size_t map_stride_floats, const double x[ORC_NSTATE], int converge,
                      int extrinsic_est_en);
/* publish_frame_world's loop (src/laserMapping.cpp:478-530): RGBpointBodyToWorld (:200-211) over n points. */
void orc_points_body_to_world(const double x[ORC_NSTATE], const float* pts, size_t stride_floats, size_t n, float* out_xyz);
/* Convenience: HTH (12x12 row-major) and HTh (12) from the last h call's h_x/h. */
void orc_normal_equations(const orc_scan* sc, double HTH[144], double HTh[12]);

typedef struct orc_update_stats {
    int passes;            /* h evaluations performed */
    int searches;          /* of which with converge==true (kNN executed) */
    int returned_in_loop;  /* 1 if the final-covariance branch ran (esekfom.hpp:1834) */
    int n_eff[8];          /* effct_feat_num per pass (-1 = not run) */
    int pass_search[8];    /* converge flag given to each pass */
    double h_time, solve_time; /* seconds: total inside h / inside host algebra */
} orc_update_stats;

/* esekf::update_iterated_dyn_share_modified (esekfom.hpp:1619-1931).
 * x, P in/out.  R = LASER_POINT_COV.  limit[23] = epsi (laserMapping.cpp:826-828). */
void orc_update_iterated(orc_scan* sc, const orc_kdtree* map, const float* map_xyz,
                         size_t map_stride_floats, double x[ORC_NSTATE],
                         double P[ORC_NDOF * ORC_NDOF], double R, int maximum_iter,
                         const double limit[ORC_NDOF], int extrinsic_est_en,
                         orc_update_stats* stats);

/* Hook for oracle/ref (record-replay against the real esekfom.hpp): after every pass of orc_update_iterated the callback
 * receives (pass number, the converge flag the pass was given, ekfom_data.valid, n_eff, h_x n_eff x 12 column-major, h,
 * the state after the pass).  NULL removes it. */
typedef void (*orc_pass_recorder)(void* ctx, int pass, int converge, int valid, int n_eff, const double* h_x,
                                  const double* h, const double x_after[ORC_NSTATE]);
void orc_set_pass_recorder(orc_pass_recorder cb, void* ctx);

/* One IEKF pass's host algebra given precomputed normal equations (information form,
 * esekfom.hpp:1651-1817 with HTH/HTh substituted).  Used to check the product's host solver and the
 * info-form/gain-form equivalence KAT.  Returns dx_ (23). x is updated in place (x boxplus dx_). */
void orc_iekf_pass_info(double x[ORC_NSTATE], const double x_prop[ORC_NSTATE],
                        const double P_prop[ORC_NDOF * ORC_NDOF], double R, const double HTH[144],
                        const double HTh[12], double P_out[ORC_NDOF * ORC_NDOF],
                        double K_x_out[ORC_NDOF * ORC_NDOF], double dx_out[ORC_NDOF]);
/* Gain form (esekfom.hpp:1715-1744) from explicit rows; n_eff < 23 in the reference. */
void orc_iekf_pass_gain(double x[ORC_NSTATE], const double x_prop[ORC_NSTATE],
                        const double P_prop[ORC_NDOF * ORC_NDOF], double R, const double* h_x_colmajor,
                        const double* h, int n_eff, double P_out[ORC_NDOF * ORC_NDOF],
                        double K_x_out[ORC_NDOF * ORC_NDOF], double dx_out[ORC_NDOF]);

/* map_incremental decision (src/laserMapping.cpp:427-474): classifies each scan point after the
 * update into 0 = skip, 1 = PointToAdd (downsample insert), 2 = PointNoNeedDownsample.
 * world_out (N x 3) receives the posterior-transformed points. */
void orc_map_incremental_classify(const orc_scan* sc, const float* map_xyz, size_t map_stride_floats,
                                  const double x[ORC_NSTATE], double filter_size_map,
                                  int flg_EKF_inited, float* world_out, uint8_t* cls);

/* ---- incremental map (stand-in for ikd-Tree Add_Points / Delete_Point_Boxes; source absent, semantics
 * restated from the published ikd-Tree algorithm [recalled-upstream] as used at src/laserMapping.cpp:275,470-471).
 *
 * orc_map_add: inserts n points into the map array (cap >= M + n entries of xyz).
 *   downsample != 0 (Add_Points(PointToAdd, true)): each new point competes inside its filter_size_map
 *   voxel [floor(x/ds)*ds, +ds)^3: the voxel ends up holding exactly ONE point, the one nearest to the voxel
 *   centre among the points already there and the new ones (an existing point is displaced only by a strictly
 *   nearer one... of the NEW points the later wins a tie; a new point wins a tie against an existing one);
 *   a voxel with a single existing point that stays nearest is left untouched.
 *   downsample == 0 (Add_Points(PointNoNeedDownsample, false)): plain insert.
 * The result keeps surviving old points in their old order, followed by surviving new points in input order.
 * Returns the new size. */
size_t orc_map_add(float* map_xyz, size_t M, const float* add_xyz, size_t n, int downsample, double ds);
/* Sensitivity variant (oracle_path.c): the same insert with ikd-Tree's own FLOAT box arithmetic; equal to orc_map_add at
 * downsample_size 0.5, not at 0.3.  Used by tools/eigen_order_study.py only. */
size_t orc_map_add_floatbox(float* map_xyz, size_t M, const float* add_xyz, size_t n, float ds);
/* Delete_Point_Boxes: removes every point p with min <= p < max (per axis) for any of the nb boxes
 * (boxes: nb x 6 floats min xyz, max xyz).  Order preserved.  Returns the new size. */
size_t orc_map_delete_boxes(float* map_xyz, size_t M, const float* boxes, size_t nb);

/* lasermap_fov_segment (src/laserMapping.cpp:230-280): LocalMap_Points + Localmap_Initialized */
typedef struct { float vertex_min[3], vertex_max[3]; int initialized; } orc_local_map;
int orc_fov_segment(orc_local_map* lm, const double pos_lid[3], double cube_len, float det_range, float* boxes_out);

/* pcl::VoxelGrid (downSizeFilterSurf, src/laserMapping.cpp:904-905): one float centroid per occupied leaf, output in
 * ascending voxel-index order; summation order inside a voxel pinned to ascending input index (see oracle_path.c). */
size_t orc_voxel_grid(const float* in, size_t stride_floats, size_t n, float leaf, float* out_xyz);

/* ImuProcess::UndistortPcl, per-point half (src/IMU_Processing.hpp:307-349); poses = IMUpose (msg/Pose6D.msg). */
typedef struct { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; } orc_pose6d;
/* 1 (default): the reference's re-compensation of the earliest point (IMU_Processing.hpp:345) is reproduced; 0: every point once */
void orc_set_undistort_first(int on);
void orc_undistort(const orc_pose6d* poses, int n_pose, const double x_end[ORC_NSTATE], const float* pts, size_t stride_floats,
                   size_t time_off_floats, size_t n, float* out_xyz);

/* ImuProcess::UndistortPcl, forward half (src/IMU_Processing.hpp:217-300).  st = the members of ImuProcess that live
 * from scan to scan; imu = n rows {t, acc[3], gyr[3]}; poses_out needs room for n + 1 entries; returns their number;
 * x, P end at the scan-end state. */
typedef struct {
    double mean_acc[3], cov_acc[3], cov_gyr[3], cov_bias_gyr[3], cov_bias_acc[3], angvel_last[3], acc_s_last[3];
    double last_imu[7];
    double last_lidar_end_time;
} orc_imu_state;
int orc_imu_forward(orc_imu_state* st, const double* imu, int n, double pcl_beg_time, double pcl_end_time,
                    double x[ORC_NSTATE], double P[ORC_NDOF * ORC_NDOF], orc_pose6d* poses_out);

#ifdef __cplusplus
}
#endif
#endif
