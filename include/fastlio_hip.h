/*
 * fastlio_hip.h -- C ABI of libfastlio_hip.so: the MI355X (gfx950) implementation of FAST-LIO2's
 * per-scan measurement-update hot path.
 *
 * This is the drop-in boundary.  Everything above it (the IEKF loop, the ROS node)
 * stays host C++; everything below it is hand-written HIP.  Plain pointers and sizes only.
 * All citations are file:line under the reference tree (hku-mars/FAST_LIO @ 2024-08-07).
 *
 * Layer 1 (flh_*)        replaces the body of h_share_model (src/laserMapping.cpp:638-754) and the
 *                        ikd-Tree calls it makes (Build :919, Nearest_Search :670).
 * Layer 2 (flh_esekf_*)  is the host-side iterated ESKF, a C binding of the C++ mirror in
 *                        include/fastlio_amd/esekfom.hpp of
 *                        esekfom::esekf<state_ikfom,12,input_ikfom> (esekfom.hpp:108-2005), so that
 *                        non-C++ callers (the ctypes tests, bench.py) drive exactly the code the C++
 *                        node would.
 *
 * Conventions
 *   - state: 26 doubles  pos[3] rot_xyzw[4] offset_R_L_I_xyzw[4] offset_T_L_I[3] vel[3] bg[3] ba[3]
 *     grav[3]   (member order of MTK_BUILD_MANIFOLD(state_ikfom,...), include/use-ikfom.hpp:12-21;
 *     quaternions in Eigen coeffs() order x,y,z,w).
 *   - covariances: 23x23 doubles, row-major, DOF order pos rot offR offT vel bg ba grav(2).
 *   - points: fp32 xyz at a caller-given byte stride (16 for float4, 48 for pcl::PointXYZINormal,
 *     12 for packed xyz); only the first three floats of each record are read.
 *   - return value: 0 = ok, <0 = error (flh_last_error() gives the text).  "No effective points"
 *     (src/laserMapping.cpp:708-713) is NOT an error: n_eff == 0 and the caller sets valid=false.
 *   - one caller thread per handle; calls block until their result is on the host unless noted.
 */
#ifndef FASTLIO_HIP_H
#define FASTLIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLH_NSTATE 26
#define FLH_NDOF 23
#define FLH_K 5 /* NUM_MATCH_POINTS, include/common_lib.h:26 */

typedef struct flh_handle flh_handle;

typedef struct flh_config {
    int device;             /* HIP device ordinal; -1 = current device */
    float cell_size;        /* search-grid cell edge in metres; <=0 -> 1.5 (= 3 x filter_size_map 0.5).  The fast
                               kernel settles a query whose 5th neighbour is within ~cell_size; smaller cells
                               mean fewer candidates but more queries on the slower ring-expansion path */
    float plane_threshold;  /* esti_plane inlier threshold; <=0 -> 0.1f (src/laserMapping.cpp:678) */
    float max_sqdist;       /* kNN gate on the 5th neighbour; <=0 -> 5.0f (src/laserMapping.cpp:671) */
    void* stream;           /* hipStream_t to run on; NULL -> the handle creates its own */
    int lanes_per_query;    /* 4 (default, also for any other non-zero value): the ring search, four lanes per query in its first
                               stage; 0 = run the general exact kernel for every query (the tests' cross-check) */
    int sort_queries;       /* 1: Morton-sort scan points at upload for cache locality (default 1 if <0) */
    int pass_kernel;        /* 1 (default, also for < 0): a SEARCHING pass is ONE launch -- 5-NN, plane fit, residual gate, Jacobian
                               rows, H^T H / H^T h and their group sums in one kernel (k_pass); 0: the three-launch pass (two search
                               stages, then the fit).  Same results bit for bit.  The one-launch pass needs cells of at least
                               sqrt(max_sqdist) / 1.49 (1.5 m for the default gate) and lanes_per_query = 4; otherwise, and with
                               plane_fit_dtype = 1, the three-launch pass runs.  (With an RCCL communicator attached the one-launch
                               pass runs too: its group totals stay in device memory for the all-reduce) */
    int eigen_order;        /* fp32 summation order of esti_plane's reductions (include/common_lib.h:241 runs Eigen's
                               ColPivHouseholderQR, whose reduction order depends on how Eigen was vectorised):
                               FLH_ORDER_SEQ / _SSE / _PAIRWISE / _NOVEC; <0 -> FLH_ORDER_SSE (Eigen 3.3.x, x86-64 + SSE2:
                               the reference's own build).  See DESIGN.md "Eigen summation order" */
    int plane_fit_dtype;    /* 0 = fp32, exactly as the reference (esti_plane<float>); 1 = ABLATION ONLY: the plane fit in
                               fp16 on query-centred coordinates (BASELINE configs[4]); not bit-exact, never the default */
    int undistort_first_point; /* flh_scan_stage_undistorted: 1 (default, also for < 0) = as the reference, whose sweep compensates the
                               EARLIEST point of the cloud once per segment older than it (src/IMU_Processing.hpp:345); 0 = every
                               point once */
    int plane_cache;        /* 1 (default, also for < 0): a pass that does not search takes each point's plane from the fit of the last
                               searching pass instead of re-reading five neighbours and repeating the QR (a plane depends on the
                               neighbours only, not on the state: same bits); 0: re-fit on every pass */
    int fused_small_changes; /* 1 (default, also for < 0): a map change of at most 8192 points (flh_map_add, flh_map_incremental:
                               every scan of a running odometry) gives the surviving points their ids and sorts them by brick in
                               one workgroup instead of the general path's scan + device-wide sort (one launch instead of eight,
                               same results); 0: the general path for every size.  Performance only */
    int prelaunch;          /* 1 (default, also for < 0): flh_eval_expect_next is honoured -- the kernel of a no-search evaluation the
                               caller announces is enqueued beside the pass before it and takes its state from a mailbox in pinned
                               memory, so that its launch leaves the critical path (same bits); 0: hints are ignored.  Performance only */
    int index_cache;        /* 1 (default, also for < 0): the one-launch searching pass leaves the five neighbours of a query as map
                               INDICES (20 B per query); their coordinates are gathered on demand by whatever asks for them
                               (flh_map_incremental, flh_fetch_neighbors, a re-fit without the plane cache); 0: the search writes
                               the coordinates (80 B per query) itself.  Needs plane_cache (which exists for eigen_order =
                               FLH_ORDER_SSE and plane_fit_dtype = 0 only); same results.  Performance only */
    int stage_sort;         /* 1 (default, also for < 0): a scan of up to 114 688 points is staged by the library's own two kernels
                               (tile sort in LDS + merge by rank, flh_stage.hip); 2: up to 262 144 points; 0: k_scan_restride + the
                               vendor library's radix sort + gather (twelve launches; also what larger scans take).  Same order,
                               same bits.  Performance only */
} flh_config;
enum { FLH_ORDER_SEQ = 0, FLH_ORDER_SSE = 1, FLH_ORDER_PAIRWISE = 2, FLH_ORDER_NOVEC = 3 };

void flh_default_config(flh_config* cfg);
int flh_create(const flh_config* cfg, flh_handle** out);
void flh_destroy(flh_handle* h);
const char* flh_last_error(void);
/* 1 if a usable HIP device is visible to this process, else 0 (never throws, never aborts). */
int flh_device_available(void);

/* ikdtree.Build(feats_down_world->points) -- src/laserMapping.cpp:919.  Builds the device map and
 * its spatial index (radix-sorted cell grid) from M world-frame points.  Map indices reported by
 * flh_fetch_neighbors refer to positions in this array. */
int flh_map_build(flh_handle* h, const void* xyz, size_t stride_bytes, size_t M);
/* Map changes (flh_map_add, flh_map_delete_boxes, flh_map_incremental with apply) are ENQUEUED and return; their counters -- and
 * an error a change ran into on the device: a re-indexing that failed, a fault -- are collected by whichever call needs them
 * next.  flh_map_sync collects them now and returns the change's status (0 / -1 with flh_last_error); flh_map_size and
 * flh_map_stats collect them too (they may wait for the device and re-index; not to be called concurrently with other calls on
 * the handle) and report the state before the change if collecting failed. */
int flh_map_sync(flh_handle* h);
size_t flh_map_size(const flh_handle* h);

/* ---- the incremental map around the hot path --------------------------------------
 * The map keeps an INDEX ORDER: after every change the survivors keep their relative order, points that were
 * already in the map first, inserted points after them.  flh_fetch_neighbors' indices and the search's
 * equal-distance tie-break (lower index) refer to that order.  A change rewrites only the bricks of the device index it
 * touches (see flh_map_stats) and invalidates the active scan's neighbour cache (the next flh_eval must search, as the
 * reference's does). */

/* ikdtree.Add_Points(points, downsample_on) -- src/laserMapping.cpp:470-471, down-sampling length as set by
 * ikdtree.set_downsample_param(filter_size_map_min) (:868).  downsample != 0: per downsample_size voxel only the
 * point nearest to the voxel centre survives (an inserted point wins an exact tie against a map point, a later
 * inserted point against an earlier one); a voxel whose single map point stays nearest is left untouched. */
int flh_map_add(flh_handle* h, const void* xyz, size_t stride_bytes, size_t n, int downsample, double downsample_size);
/* ikdtree.Delete_Point_Boxes(cub_needrm) -- src/laserMapping.cpp:275 (lasermap_fov_segment).  boxes = nb x
 * {min x,y,z, max x,y,z}; a point with min <= p < max on every axis is removed. */
int flh_map_delete_boxes(flh_handle* h, const float* boxes, size_t nb);
/* Bookkeeping of the device map: out = {full re-indexings so far, changes applied brick-wise (no re-indexing), storage slots
 * in use, storage slots allocated, point ids handed out since the last re-indexing, bricks}.  The map's points sit in
 * per-brick storage ranges with slack; an insert rewrites only the bricks it touches and a removal tombstones its slot;
 * the whole index is rebuilt only when something no longer fits (a point outside the grid, storage or tables full). */
int flh_map_stats(const flh_handle* h, uint64_t out[6]);
/* What the searches have to read: out = {live map points, storage slots INSIDE the bricks' ranges (live points + the tombstones of
 * removed ones that a search of those cells still loads), bricks compacted in place so far, bricks}.  After a removal
 * (flh_map_delete_boxes, flh_fov_segment) every brick whose live points fell below half of its range is compacted where it lies,
 * so out[1] stays below about twice out[0] however long lasermap_fov_segment (src/laserMapping.cpp:231-277) keeps removing. */
int flh_map_storage_stats(flh_handle* h, uint64_t out[4]);
/* flh_map_incremental with apply and without the two count outputs: out = {calls whose Add_Points was enqueued right behind the
 * classification, the list lengths read on the device (no wait of the host in the middle of the call: taken from the second
 * change on, the launches sized for the previous change's points + 50 %), of those the ones that turned out larger than their
 * launches and were replayed when their counters were collected}. */
int flh_map_change_stats(const flh_handle* h, uint64_t out[2]);
/* The map in index order, 3 floats per point (what ikdtree.flatten / PCL_Storage hands back, :406-411). */
int flh_map_download(flh_handle* h, float* xyz, size_t capacity_points);
/* map_incremental() -- src/laserMapping.cpp:427-474, evaluated on the device from the neighbour cache the active
 * scan's last search left there (Nearest_Points).  x = the posterior state in the flat layout of flh_eval_device.
 * Classifies every scan point (skip / PointToAdd / PointNoNeedDownsample), and with apply != 0 performs the two
 * Add_Points calls.  n_add / n_no_downsample (optional) receive the two list lengths (add_point_size = n_add as
 * Add_Points counts it before down-sampling). */
int flh_map_incremental(flh_handle* h, const double x[FLH_NSTATE], double filter_size_map, int flg_EKF_inited, int apply,
                        uint32_t* n_add, uint32_t* n_no_downsample);
/* Results of the last flh_map_incremental in ORIGINAL scan order: cls[i] in {0,1,2}; world_xyz = feats_down_world
 * (:436).  Either may be NULL. */
int flh_fetch_map_incremental(flh_handle* h, uint8_t* cls, float* world_xyz);

/* lasermap_fov_segment() -- src/laserMapping.cpp:230-280: the local-map cube that follows the LiDAR.  The caller
 * owns the cube state (zero-initialise it); pos_lid = state.pos + state.rot * state.offset_T_L_I (:890).  When the
 * sensor comes within 1.5 * det_range of a face the cube shifts and the slabs that fall out are removed from the
 * device map (ikdtree.Delete_Point_Boxes, :275).  boxes_out (optional, room for 3 x 6 floats) receives the slabs,
 * n_boxes their number, kdtree_delete_counter (optional) the number of map points removed. */
typedef struct flh_local_map {
    float vertex_min[3], vertex_max[3]; /* LocalMap_Points */
    int initialized;                    /* Localmap_Initialized */
} flh_local_map;
int flh_fov_segment(flh_handle* h, flh_local_map* lm, const double pos_lid[3], double cube_len, float det_range,
                    float* boxes_out, int* n_boxes, int64_t* kdtree_delete_counter);

/* feats_down_body for the coming update -- src/laserMapping.cpp:904-905,935-951.  Resets
 * point_selected_surf to all-true (as memset at :812 leaves it for a fresh search) and clears the
 * neighbour cache. */
int flh_scan_upload(flh_handle* h, const void* pts, size_t stride_bytes, size_t N);
size_t flh_scan_size(const flh_handle* h);

/* Scan staging ring (double-buffering the H2D copy of scan k+1 behind the update of scan k).
 * flh_scan_stage copies a scan into device slot `slot` (0..FLH_MAX_SLOTS-1) on a separate copy stream and
 * returns once the host buffer may be reused; flh_scan_activate makes a staged scan the current one
 * (waits for its copy, resets point_selected_surf / the neighbour cache) without touching PCIe. */
#define FLH_MAX_SLOTS 64
int flh_scan_stage(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t N);
int flh_scan_activate(flh_handle* h, int slot);
/* The same staging, performed by the handle's staging thread: the call returns at once and the caller's thread goes on
 * with the update of the previous scan (the node's main loop, src/laserMapping.cpp:865-969, would hand over scan k+1
 * from its LiDAR callback).  `pts` must stay valid and unchanged until flh_scan_wait(slot) or flh_scan_activate(slot)
 * returns; a staging error is reported there.  Pageable memory is copied through a pinned buffer; memory from
 * flh_host_alloc is DMA'd from where it lies. */
int flh_scan_stage_async(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t N);
int flh_scan_wait(flh_handle* h, int slot);
/* Page-locked host memory for scan / cloud buffers handed to this library (saves the copy through the staging buffer).  ONLY
 * buffers from flh_host_alloc are read where they lie; any other memory -- pageable or page-locked by somebody else -- is copied
 * through the slot's own page-locked buffer (the library does not ask the runtime what a foreign pointer is).  flh_host_free
 * waits for the device first. */
void* flh_host_alloc(size_t bytes);
void flh_host_free(void* p);

/* SURVEY.md 8(f) row 2 -- downSizeFilterSurf.setInputCloud(feats_undistort); downSizeFilterSurf.filter(*feats_down_body)
 * (src/laserMapping.cpp:904-905; pcl::VoxelGrid with leaf = filter_size_surf_min, :813) on the device, and staging of the
 * result into `slot` exactly as flh_scan_stage does.  One float centroid per occupied leaf, in ascending voxel-index
 * order (PCL's output order); inside a leaf the float sum runs in ascending input index (PCL's order there is
 * implementation-defined).  n_out (optional) receives feats_down_size.  A grid that would overflow int32 returns
 * the input unchanged, as PCL does. */
int flh_scan_stage_downsampled(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t n, float leaf_size,
                               size_t* n_out);
/* SURVEY.md 8(f) row 3 -- ImuProcess::UndistortPcl's per-point half (src/IMU_Processing.hpp:307-349) on the device, in
 * front of the down-sampling above.  imu_pose = IMUpose (msg/Pose6D.msg; the forward half, :240-300, is one
 * esekf::predict per IMU sample on the host), x_end = the propagated state at the scan end (flat layout of
 * flh_eval_device), time_offset_bytes = where the float time offset in ms (PointType::curvature) sits in a point record.
 * A point at time t is carried to the scan-end frame with the LAST segment k <= n_pose-2 whose offset_time is < t
 * (what the reference's back-to-front sweep over the time-sorted cloud amounts to, also when the first IMU sample
 * precedes the first point and offset_time[1] < offset_time[0] = 0); a point no segment claims is left as it is.  The cloud
 * is NOT re-ordered by time (the reference's sort only serves its sweep).  The reference's loop (:326-346) breaks at begin()
 * without stepping past it, so the EARLIEST point of the cloud (lowest index among equal times here; the reference's
 * std::sort leaves that unspecified) is compensated again by every earlier segment older than it, each time on its moved
 * coordinates: reproduced by default, flh_config.undistort_first_point = 0 carries it once like every other point.
 * leaf_size <= 0 skips the down-sampling; undistorted_xyz (optional, 3*n floats) receives feats_undistort. */
typedef struct flh_pose6d {
    double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9];
} flh_pose6d;
int flh_scan_stage_undistorted(flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t time_offset_bytes, size_t n,
                               const flh_pose6d* imu_pose, int n_pose, const double x_end[FLH_NSTATE], float leaf_size,
                               float* undistorted_xyz, size_t* n_out);
/* feats_down_body of the ACTIVE scan (3 floats per point, the order it was staged in). */
int flh_fetch_scan(flh_handle* h, float* xyz);

/* SURVEY.md 8(f) row 4 -- publish_frame_world()'s loops (src/laserMapping.cpp:478-530): RGBpointBodyToWorld (:200-211), i.e.
 * p_world = rot * (offset_R_L_I * p_body + offset_T_L_I) + pos in double narrowed to float, over a whole cloud.
 * flh_frame_world: the cloud already on the device -- dense != 0: feats_undistort, the cloud `slot` was staged from by
 * flh_scan_stage_undistorted / _downsampled (dense_pub_en, and the pcd_save_en loop); dense == 0: the slot's feats_down_body.
 * slot < 0 = the active scan.  x = state_point (the posterior), flat layout.  Output in the cloud's original order;
 * world_xyz == NULL with capacity_points == 0 only reports the size in *n_points.
 * flh_points_body_to_world: the same for any host cloud. */
int flh_frame_world(flh_handle* h, int slot, const double x[FLH_NSTATE], int dense, float* world_xyz, size_t capacity_points,
                    size_t* n_points);
int flh_points_body_to_world(flh_handle* h, const double x[FLH_NSTATE], const void* pts, size_t stride_bytes, size_t n,
                             float* world_xyz);

/* One evaluation of h_share_model (src/laserMapping.cpp:638-754) at state s:
 *   transform :652-661, 5-NN + gate :667-672 (only if do_search = ekfom_data.converge), plane fit
 *   + residual gate :676-692, then -- instead of materialising h_x/h (:720-752) -- the normal
 *   equations the IEKF needs (esekfom.hpp:1784,1804):
 *     HTH[12*12] (row-major; symmetric) = h_x^T h_x,   HTh[12] = h_x^T h,
 *     n_eff = effct_feat_num, total_residual (laserMapping.cpp:702).
 * When extrinsic_est_en == 0 the last six columns are zero (laserMapping.cpp:745). */
int flh_eval(flh_handle* h, const double rot_xyzw[4], const double pos[3], const double offR_xyzw[4],
             const double offT[3], int do_search, int extrinsic_est_en, double HTH[144], double HTh[12],
             int64_t* n_eff, double* total_residual);

/* The same evaluation in two halves: flh_eval_begin enqueues the pass and returns, flh_eval_end waits for its normal equations.
 * Between the two the caller's thread may do host work that does not depend on them (the mirror esekf projects the covariance
 * there -- include/fastlio_amd/esekfom.hpp).  One evaluation under way per handle; an error in flh_eval_begin
 * leaves none under way.  A searching evaluation that follows a flh_map_incremental(apply = 1, no counts asked) is enqueued BEHIND
 * that map change without waiting for its counters; flh_eval_end folds them first and, when they ask for a re-index or a replay of
 * the change, runs the pass once more on the settled map (the results are those of the settled map either way). */
int flh_eval_begin(flh_handle* h, const double rot_xyzw[4], const double pos[3], const double offR_xyzw[4], const double offT[3],
                   int do_search, int extrinsic_est_en);
int flh_eval_end(flh_handle* h, double HTH[144], double HTh[12], int64_t* n_eff, double* total_residual);
/* What the caller expects the evaluation AFTER its next flh_eval_begin to be -- a hint, consumed by that flh_eval_begin; a wrong
 * hint costs time, never correctness.  The iterated update knows it before every pass (esekfom.hpp:1823-1834; the mirror filter
 * include/fastlio_amd/esekfom.hpp says so through dyn_share_datastruct::next_pass).
 *   FLH_NEXT_NOSEARCH  probably a no-search evaluation of the same scan, at a state not known yet: its kernel is enqueued beside
 *                      the pass flh_eval_begin starts and waits (bounded: 20 ms) for the state, which the following
 *                      flh_eval_begin then posts instead of launching.  Any other call that follows releases it.
 *   FLH_NEXT_UNKNOWN   no expectation (the default before every flh_eval_begin): nothing is enqueued ahead
 *   FLH_NEXT_NONE      no evaluation follows at all (the update has ended): a kernel that is still waiting is released NOW --
 *                      not to be said before a flh_eval_begin whose own kernel may be the one that is waiting */
#define FLH_NEXT_UNKNOWN 0
#define FLH_NEXT_NOSEARCH 1
#define FLH_NEXT_NONE 2
int flh_eval_expect_next(flh_handle* h, int kind);
/* flh_config.prelaunch at run time (0: hints are ignored from now on, a waiting kernel is released); and the counters since creation:
 * {kernels enqueued ahead, of them handed their state, released unused, given up before the host came (then launched the usual way)} */
int flh_set_prelaunch(flh_handle* h, int on);
int flh_get_prelaunch_stats(const flh_handle* h, uint64_t out[4]);

/* Same evaluation, but the reduced 16x16 Gram block is left in DEVICE memory at d_gram256 (256
 * doubles, row-major G = sum_k v_k v_k^T with v = [row(12) | h | 1 | |pd2| | 0]) and the call
 * returns after enqueueing on the handle's stream.  This is the hook for the multi-GPU path: the
 * caller all-reduces d_gram256 across ranks (RCCL) and then unpacks it with flh_unpack_gram. */
int flh_eval_device(flh_handle* h, const double state[FLH_NSTATE], int do_search, int extrinsic_est_en,
                    double* d_gram256);
void flh_unpack_gram(const double gram256[256], double HTH[144], double HTh[12], int64_t* n_eff,
                     double* total_residual);

/* ---- multi-GPU -------------------------------------------------------------------------------------------
 * The scan's points are sharded over the GPUs (map replicated), or the map is partitioned (flh_set_owned_interval); either
 * way a pass's only exchange is the sum of the ranks' normal equations (esekfom.hpp:1784,1804), and every rank runs the same
 * 23x23 solve on the same bits.  Two exchanges:
 *
 * PEER GRANULES (flh_peer_*, the default of bench.py --gpus N): no collective and no extra launch.  The last workgroup of every
 *   reduction group of a pass writes its {value, sequence} granules straight into EVERY rank's pinned granule buffer -- one
 *   POSIX shared-memory segment that each process maps and registers with its device -- and every host adds ranks x groups
 *   in (rank, group) order while it polls.  A pass on G GPUs then costs what it costs on one GPU with N/G points plus one
 *   PCIe write latency.  Every rank must call flh_eval the same number of times (they do: same normal equations, same
 *   decisions); a scan shard must not be empty.
 *     one process per GPU: every rank calls flh_peer_open(h, name, nranks, rank) with the same name ("/something", see
 *       shm_open(3)); rank 0 creates the segment, the others wait for it (30 s).  flh_peer_close (or flh_destroy) detaches;
 *       rank 0 unlinks the name.
 *     one process, several handles: flh_peer_init_all, then flh_eval_group.
 *
 * RCCL (flh_rccl_*): the pass kernels' group reducers leave their GROUP TOTALS in device memory (64 groups x 31 doubles, 94 with
 *   the extrinsic columns), ncclAllReduce adds the ranks' totals in place on the handle's stream, and one small kernel adds the
 *   groups in the host's order and publishes the 16x16 block -- the summation tree of the single-GPU path, so one rank reproduces
 *   its bits (scans beyond the granule limit of 1.6 M points per rank all-reduce the 256 doubles of the block instead).  Once a
 *   handle has a communicator, flh_eval all-reduces before it returns.  RCCL is loaded on first use.  A handle has at most one
 *   of the two.
 *   one process per GPU: rank 0 calls flh_rccl_unique_id, hands the 128 bytes to the others by any means (MPI, a file,
 *     torch.distributed), every rank calls flh_rccl_init_rank.
 *   one process, several GPUs: flh_rccl_init_all over one handle per device, then flh_eval_group (enqueues on every device
 *     inside one RCCL group, waits once) instead of flh_eval. */
#define FLH_RCCL_ID_BYTES 128
int flh_rccl_unique_id(char id[FLH_RCCL_ID_BYTES]);
int flh_rccl_init_rank(flh_handle* h, int nranks, const char id[FLH_RCCL_ID_BYTES], int rank);
int flh_rccl_init_all(flh_handle* const* handles, int n);
void flh_rccl_destroy(flh_handle* h);
int flh_rccl_size(const flh_handle* h);
int flh_rccl_rank(const flh_handle* h);
int flh_eval_group(flh_handle* const* handles, int n, const double state[FLH_NSTATE], int do_search, int extrinsic_est_en,
                   double HTH[144], double HTh[12], int64_t* n_eff, double* total_residual);
#define FLH_MAX_PEERS 8
int flh_peer_open(flh_handle* h, const char* shm_name, int nranks, int rank);
int flh_peer_init_all(flh_handle* const* handles, int n);
void flh_peer_close(flh_handle* h);
int flh_peer_size(const flh_handle* h);
int flh_peer_rank(const flh_handle* h);
/* Counters of the handle's evaluations since creation: {searching passes, of them as ONE launch (flh_config.pass_kernel),
 * queries that needed the second search (summed over those passes and, with peers, over the ranks), no-search passes}. */
int flh_get_pass_stats(const flh_handle* h, uint64_t out[4]);
/* Map partitioned over the ranks (BASELINE configs[4]): this handle's map is one slab of the world plus a halo of at least
 * sqrt(max_sqdist) on either side; every rank holds the whole scan; a query is searched (and then fitted) only by the
 * rank whose half-open interval [lo, hi) of world coordinate `axis` (0/1/2) contains it.  The ranks' intervals must tile
 * the axis.  axis < 0 removes the restriction. */
int flh_set_owned_interval(flh_handle* h, int axis, float lo, float hi);

/* Lazy D2H fetches of the globals later reference code reads (SURVEY.md 8b "Data passed implicitly"). */
int flh_fetch_selected(flh_handle* h, uint8_t* flags /* N: point_selected_surf */);
int flh_fetch_neighbors(flh_handle* h, int32_t* idx /* N x 5 map indices, -1 = none */,
                        float* d2 /* N x 5, ascending */, uint8_t* cnt /* N, may be NULL */);
int flh_fetch_world(flh_handle* h, float* xyz /* N x 3: feats_down_world */);
int flh_fetch_normvec(flh_handle* h, float* abcd_pd2 /* N x 4: normvec (a,b,c, intensity=pd2) */);
/* Materialise ekfom_data.h_x (n_eff x 12, column-major like Eigen::MatrixXd) and ekfom_data.h for
 * the LAST evaluated state, in original scan order (src/laserMapping.cpp:720-752).  Needed by the
 * n_eff < 23 gain-form branch (esekfom.hpp:1715-1744).  cap_rows = capacity of the caller buffers. */
int flh_fetch_rows(flh_handle* h, double* h_x_colmajor, double* hvec, int64_t cap_rows, int64_t* n_rows);

/* Per-call device timings of the last flh_eval, milliseconds (HIP events on the handle's stream). */
typedef struct flh_timing {
    float search_ms;  /* transform + 5-NN kernel (0 when do_search == 0) */
    float fit_ms;     /* plane fit + residual + Jacobian + Gram kernels */
    float total_ms;   /* first launch to result visible on the host */
    int64_t candidates; /* map points examined by the last search (0 unless stats are enabled) */
} flh_timing;
int flh_last_timing(flh_handle* h, flh_timing* t);
/* Accumulated device time (HIP events on the handle's stream) per kernel group since the last reset:
 * out[0] = sum of search-kernel ms, out[1] = number of search launches, out[2] = sum of fit(+reduce) ms,
 * out[3] = number of fit launches, out[4] = sum of first-launch-to-host-visible ms, out[5] = evaluations. */
int flh_get_counters(flh_handle* h, double out[6], int reset);
/* The search part of the same, split by kind: out[0], out[1] = ms and launches of a scan's FIRST search (at the prior: the most
 * queries for the second stage), out[2], out[3] = of its LATER searches.  The same kernels run either way; the split is a
 * measurement label.  With the one-launch pass (flh_config.pass_kernel) the figures are the pass kernel's.  Reset together with
 * flh_get_counters(reset != 0). */
int flh_get_search_counters(flh_handle* h, double out[4]);
/* The HIP events behind flh_last_timing / flh_get_counters: recorded on every n-th flh_eval (1 = always, the default;
 * 0 = never).  With every_n == 1 the evaluation waits for its last event and reads the three times at once (tens of
 * microseconds of host time per evaluation).  With every_n >= 2 -- sampling inside a running stream -- a sampled evaluation
 * only RECORDS its events; they are read when flh_get_counters / flh_last_timing / flh_set_timing_stride is called next
 * (or when 64 samples are pending).  flh_last_timing then reports the most recent sample. */
int flh_set_timing_stride(flh_handle* h, int every_n);
/* The same with search_only != 0: only SEARCHING evaluations are counted and timed (every every_n-th of them): a sampled
 * evaluation costs the host ~10 us (its launches carry events), so a short measurement samples the kernels it is after -- the
 * 5-NN search, and the fit kernel behind it -- and leaves the no-search evaluations alone. */
int flh_set_timing_sampling(flh_handle* h, int every_n, int search_only);
int flh_enable_stats(flh_handle* h, int on); /* count candidate points examined (slower) */
/* Run one kernel of the hot path `iters` times back-to-back on the handle's stream and return the
 * mean duration in ms, measured with HIP events on that stream (bench.py's roofline leg).
 * which: 0 = search (transform + 5-NN), 1 = fit (plane fit + Jacobian + Gram + final reduce). */
int flh_time_kernel(flh_handle* h, int which, const double state[FLH_NSTATE], int extrinsic_est_en,
                    int iters, float* mean_ms);

/* ------------------------------------------------------------------------------------------------
 * Layer 2: host-side iterated ESKF (C binding of include/fastlio_amd/esekfom.hpp).
 * ---------------------------------------------------------------------------------------------- */
typedef struct flh_esekf flh_esekf;

/* What a measurement model hands back per evaluation: the extended dyn_share_datastruct
 * (esekfom.hpp:79-89).  Either the fused normal equations (has_normal_eq) or explicit rows. */
typedef struct flh_meas {
    int valid;              /* ekfom_data.valid */
    int64_t n_eff;          /* rows of h_x */
    int has_normal_eq;      /* HTH/HTh filled */
    double HTH[144];
    double HTh[12];
    const double* h_x;      /* optional: n_eff x 12 column-major; NULL if not materialised */
    const double* h;        /* optional: n_eff */
    double total_residual;
} flh_meas;
/* measurementModel_dyn_share (esekfom.hpp:129): h(state&, dyn_share&).  `converge` is the input
 * flag ekfom_data.converge (re-run the kNN or reuse the cache). */
typedef void (*flh_meas_fn)(void* ctx, const double state[FLH_NSTATE], int converge, flh_meas* out);

typedef struct flh_update_stats {
    int passes;            /* h evaluations */
    int searches;          /* with converge == true */
    int returned_in_loop;  /* final-covariance branch taken (esekfom.hpp:1834) */
    int n_eff[8];
    int pass_search[8];
    double pass_ms[8];     /* wall time of each pass (measurement model + host algebra): "ms/IEKF-iter", search and
                              no-search passes apart (pass_search) */
    double h_ms;           /* wall time inside the measurement model */
    double solve_ms;       /* wall time of the host algebra (solve_H_time, esekfom.hpp:1649,1926) */
} flh_update_stats;

/* kf.init_dyn_share(get_f, df_dx, df_dw, h_share_model, NUM_MAX_ITERATIONS, epsi) -- laserMapping.cpp:828.
 * The process model is fixed to the reference's get_f/df_dx/df_dw (use-ikfom.hpp:47-88).  With
 * h == NULL the measurement model is the built-in GPU h_share_model bound to `handle`. */
flh_esekf* flh_esekf_create(flh_handle* handle, int maximum_iter, const double limit[FLH_NDOF],
                            int extrinsic_est_en);
void flh_esekf_destroy(flh_esekf* kf);
void flh_esekf_set_meas_model(flh_esekf* kf, flh_meas_fn h, void* ctx);
void flh_esekf_change_x(flh_esekf* kf, const double x[FLH_NSTATE]);           /* esekfom.hpp:1933 */
void flh_esekf_change_P(flh_esekf* kf, const double P[FLH_NDOF * FLH_NDOF]);  /* :1944 */
void flh_esekf_get_x(const flh_esekf* kf, double x[FLH_NSTATE]);              /* :1949 */
void flh_esekf_get_P(const flh_esekf* kf, double P[FLH_NDOF * FLH_NDOF]);     /* :1952 */
/* kf.predict(dt, Q, in) -- esekfom.hpp:279-383; acc/gyro = input_ikfom. Q is 12x12 row-major. */
void flh_esekf_predict(flh_esekf* kf, double dt, const double Q[144], const double acc[3], const double gyro[3]);
/* kf.update_iterated_dyn_share_modified(R, solve_time) -- esekfom.hpp:1619-1931. */
int flh_esekf_update(flh_esekf* kf, double R, flh_update_stats* stats);
/* The loop body of the node in one call: activate a staged scan (slot < 0: keep the active one), set the propagated
 * state / covariance handed over by the IMU front end (either may be NULL = keep), run the update (:960). */
int flh_esekf_update_scan(flh_esekf* e, int slot, const double x[FLH_NSTATE], const double P[FLH_NDOF * FLH_NDOF], double R,
                          flh_update_stats* st);
/* The node's main loop (src/laserMapping.cpp:865-969) over a sequence of scans, run natively: for each scan, hand the NEXT
 * one to the staging thread (flh_scan_stage_async, slots 0..ring-1 in turn), make this one the active scan, set the
 * propagated (x, P) the IMU front end produced, update (:960), optionally map_incremental (:923) with the posterior.
 * jobs are cycled: scan i uses jobs[i % n_jobs].  slot >= 0 in a job means that scan is already staged there (nothing crosses
 * PCIe); pts must stay valid during the call.  Statistics are summed over the scans run. */
typedef struct flh_scan_job {
    const void* pts;          /* feats_down_body as a host buffer (page-locked if from flh_host_alloc) */
    size_t stride_bytes, N;
    const double* x;          /* FLH_NSTATE: the propagated state for this scan */
    const double* P;          /* FLH_NDOF x FLH_NDOF */
    int slot;                 /* < 0: stage from pts; >= 0: already staged in this slot */
} flh_scan_job;
typedef struct flh_run_stats {
    int64_t scans, passes, searches;
    int64_t n_search_passes, n_nosearch_passes;
    double ms_search_passes, ms_nosearch_passes; /* sums of flh_update_stats.pass_ms by kind */
    double ms_map_incremental;
} flh_run_stats;
#define FLH_RUN_FIRST_STAGED 1 /* scan `first` was staged by the previous call (which had FLH_RUN_STAGE_NEXT) */
#define FLH_RUN_STAGE_NEXT 2   /* while the last scan updates, stage scan first + count for the next call: a continuous stream.
                                  The call may return while that staging is still under way: the buffer of jobs[(first + count) %
                                  n_jobs] must stay valid and unchanged until the next flh_esekf_run_scans call activates it or
                                  flh_scan_wait((first + count) % ring) returns.  On an error return no staging is left in flight. */
int flh_esekf_run_scans(flh_esekf* kf, const flh_scan_job* jobs, int n_jobs, int64_t first, int64_t count, int ring, double R,
                        int with_map_incremental, double filter_size_map, int flags, flh_run_stats* stats,
                        double x_last[FLH_NSTATE], double P_last[FLH_NDOF * FLH_NDOF]);
/* Text of the last error a flh_esekf_* call on this filter returned (failures inside the measurement model included). */
const char* flh_esekf_last_error(const flh_esekf* kf);

#ifdef __cplusplus
}
#endif
#endif /* FASTLIO_HIP_H */
