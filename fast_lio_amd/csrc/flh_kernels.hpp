// flh_kernels.hpp -- host-callable launch wrappers for the kernels in flh_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flh_device.hpp"
#include "flh_fit_dev.hpp"

namespace flh {

hipError_t launch_map_keys(const GridParams& g, const float4* pts, uint32_t M, unsigned long long* keys,
                           uint32_t* vals, hipStream_t st);
hipError_t sort_pairs(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                      const uint32_t* vin, uint32_t* vout, uint32_t M, hipStream_t st);
hipError_t inclusive_sum(void* tmp, size_t& tmp_bytes, const uint32_t* in, uint32_t* out, uint32_t M, hipStream_t st);
hipError_t launch_brick_heads(const unsigned long long* ks, uint32_t M, uint32_t* bh, hipStream_t st);
hipError_t launch_brick_starts(const uint32_t* brick_head, const uint32_t* rank_incl, uint32_t M, uint32_t* brick_start,
                               hipStream_t st);
hipError_t launch_brick_caps(const uint32_t* brick_start, uint32_t nbricks, uint32_t* cap, hipStream_t st);
hipError_t launch_fill_tomb(float4* pts, uint32_t n, hipStream_t st);
hipError_t launch_map_place(const float4* pts, const uint32_t* vs, const uint32_t* br_incl, const uint32_t* brick_start,
                            const uint32_t* cap_incl, const uint32_t* cap, uint32_t M, float4* out, hipStream_t st);
hipError_t launch_brick_tables(const unsigned long long* ks, const uint32_t* brick_start, uint32_t nbricks,
                               const uint32_t* cap_incl, const uint32_t* cap, uint32_t* starts, uint32_t* cap_end, uint32_t* live,
                               uint2* hash, uint32_t hash_mask, int hash_shift, hipStream_t st);

hipError_t launch_scan_keys(const float4* raw, uint32_t N, float quantum, uint32_t* keys, uint32_t* vals, hipStream_t st);
hipError_t launch_scan_restride(const void* bytes, uint32_t stride_bytes, uint32_t w_off_bytes, int has_w, uint32_t N, float quantum,
                                float4* raw, uint32_t* keys, uint32_t* vals, uint32_t* bad, hipStream_t st);
hipError_t sort_scan_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                           uint32_t N, hipStream_t st);
hipError_t launch_scan_gather(const float4* raw, const uint32_t* perm, uint32_t N, float4* body, hipStream_t st);

// ---- flh_stage.hip: the staging as two launches (records -> stable Morton order as float4, .w = original index) ----
uint32_t stage_sort_max();                 // the largest scan they take
uint32_t stage_scratch_words(uint32_t N);  // scratch: sorted keys + their original indices + the tiles' samples, in 32-bit words
hipError_t launch_stage_sort(const void* records, uint32_t stride_bytes, uint32_t N, float quantum, uint32_t* scratch, float4* body,
                             hipStream_t st);

int list_stripes();
uint32_t list_stripe_cap(int N);
// the three-launch searching pass's search: lpq = 4 (first stage, four lanes per query, + second stage) or 0 (the general exact
// kernel for every query: the tests' cross-check)
hipError_t launch_search(int lpq, const GridParams& g, const StateDev& s, const float4* body, int N, uint32_t map_points,
                         float max_sqdist, int rmax, float4* nn_pts, uint8_t* nn_cnt, uint8_t* selected,
                         uint32_t* list1, uint32_t* list2, float* ub, uint32_t* counts, unsigned long long* cand_counter,
                         int own_axis, float own_lo, float own_hi, hipStream_t st,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);  // optional: time stamps of the first kernel's start / the last one's end

// ---- flh_pass.hip: a searching pass as ONE launch (search, fit, Gram, group sums -> granules) ----
int pass_blocks(int N);                          // workgroups (64 scan points each)
int pass_group_size(int N, int max_groups);      // workgroups per reduction group
hipError_t launch_pass(int order, const GridParams& g, const StateDev& s, const float4* body, int N, uint32_t map_points, float max_sqdist,
                       float thr, int ext, float4* nn_pts, uint8_t* nn_cnt, uint8_t* selected, float4* plane_cache, double* partials,
                       uint32_t* tickets, const GranOut& gran, double seq, int red, unsigned long long* cand_counter,
                       int own_axis, float own_lo, float own_hi, hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr,
                       uint32_t* nn_idx = nullptr,       // nn_idx: the neighbour cache as map indices (flh_config.index_cache), else nn_pts
                       double* group_totals = nullptr);  // gran.n_dst == 0 (an RCCL communicator): the group sums go to group_totals[group][slot]
// the all-reduced group totals -> the 16x16 block in pinned host memory + the sequence word (the RCCL path's last kernel)
hipError_t launch_publish_groups(double* totals, int ngroups_own, int ngroups_all, int nsl, int ncol, double* out256, double seq, hipStream_t st);
// neighbour cache kept as indices -> coordinates (nn_pts[r * N + i] = {map_orig[id].xyz, id}; id == -1 or >= n_ids: an empty row)
hipError_t launch_nn_gather(const float4* map_orig, uint32_t n_ids, const uint32_t* nn_idx, int N, float4* nn_pts, hipStream_t st);
hipError_t launch_publish256(const double* src, double* out256, double seq, hipStream_t st);

int fit_blocks(int N);
int reduce1_blocks(int nblk, int* per_out);
hipError_t launch_fit(int order, int half_fit, const StateDev& s, const float4* body, const float4* nn_pts, int N, int ext, float thr,
                      uint8_t* selected, float4* normvec, float4* world, double* partials, double* part2,
                      double* out256, double seq, uint32_t* tickets, uint32_t* slow_count, const GranOut& gran, int red1, int store_aux,
                      hipStream_t st, float4* plane_cache = nullptr, int plane_mode = 0, hipEvent_t ev_start = nullptr,
                      hipEvent_t ev_stop = nullptr);
hipError_t launch_fill_d2(const StateDev& s_search, const float4* body, const float4* nn_pts, int N, float* nn_d2, hipStream_t st);
int gram_slots_host(int ncol);
int gram_slot_host(int r, int c, int ncol);
// the pre-launched no-search pass (flh_mail_dev.hpp)
struct MailArgs;
hipError_t launch_fit_mb(const MailArgs& mail, const float4* body, int N, int ext, float thr, uint8_t* selected, double* partials,
                         double seq, uint32_t* tickets, uint32_t* slow_count, const GranOut& gran, int red1, const float4* plane_cache,
                         hipStream_t st);

// ---- flh_mapinc.hip: map_incremental and the incremental map ----
// nn_idx != nullptr: the neighbour cache holds map indices (coordinates read from map_orig[0 .. n_ids) on the spot), else nn_pts.
// blk_cnt: cls_block_words(N) words, ZERO on entry: the two lists' members per block of 256 original indices (launch_cls_compact
// reads them and zeroes cnt_next[0 .. next_words), the other half of the caller's double buffer, for the next call)
uint32_t cls_block_words(int N);
hipError_t launch_mi_classify(const GridParams& g, uint32_t hash_size, uint32_t map_points, const StateDev& s_search,
                              const StateDev& s_post, const float4* body, float4* nn_pts, uint32_t* nn_idx, const float4* map_orig,
                              uint32_t n_ids, const uint8_t* nn_cnt, float max_sqdist, int N, double fsm, int ekf_inited,
                              const uint32_t* live, float4* world_out, uint8_t* cls, uint32_t* blk_cnt /* may be null */,
                              uint32_t* far /* N + 1 words, far[0] == 0 on entry, re-armed by launch_cls_compact */, hipStream_t st,
                              unsigned long long* tab_fill = nullptr, uint32_t tab_words = 0);  // tab_fill: a voxel table to empty (0xFF) on the way
hipError_t launch_cls_compact(const float4* world, const uint8_t* cls, const uint32_t* blk_cnt, uint32_t* cnt_next, uint32_t next_words,
                              int N, float4* out, uint32_t* host_counts, uint32_t seq, hipStream_t st, uint32_t* dev_counts = nullptr,
                              uint32_t* far = nullptr,  // launch_mi_classify's list, re-armed (far[0] = 0) for the next call
                              // the Add_Points enqueued behind this kernel (launches sized for ins_bound points; its voxel table of
                              // ins_cap slots was emptied by launch_mi_classify): the kernel performs launch_add_insert's step itself
                              unsigned long long* ins_tab = nullptr, uint32_t ins_cap = 0, double ins_ds = 0.0,
                              uint8_t* ins_alive_new = nullptr, uint32_t* ins_ctr = nullptr, uint32_t ins_bound = 0);
// dev_counts (the launches of a map change below): {n1, n} in device memory, read by the kernels instead of the host's values, which
// then only size the launches; a change larger than that does nothing and k_map_publish raises kMapChangeNotApplied
constexpr uint32_t kMapChangeNotApplied = 0x80000000u;
hipError_t launch_map_publish(const uint32_t* ctr, const uint32_t* n_alive, uint32_t* host_out, uint32_t seq, hipStream_t st,
                              const uint32_t* dev_counts = nullptr, uint32_t cap = 0);
hipError_t launch_aabb(const float4* pts, uint32_t M, uint32_t* out6, hipStream_t st);
uint32_t vox_table_slots(uint32_t n1);
hipError_t launch_add_insert(const float4* add, uint32_t n1, uint32_t n, double ds, unsigned long long* tab, uint32_t cap,
                             uint8_t* alive_new, uint32_t* ctr, hipStream_t st, const uint32_t* dev_counts = nullptr);
// map changes of at most small_change_max() points: ids, brick keys and their sort in one workgroup (flh_mapinc.hip)
uint32_t small_change_max();
hipError_t launch_ins_sort_small(const GridParams& g, const float4* add, const uint8_t* alive_new, uint32_t n, uint32_t n_ids,
                                 float4* map_orig, uint8_t* dead_id, float4* ins, uint32_t* keys_tmp, uint32_t* ks, uint32_t* perm,
                                 uint32_t* ctr, uint32_t* n_alive_out, uint32_t* heads, hipStream_t st,
                                 const uint32_t* dev_counts = nullptr);  // heads[0 .. ctr[6]): the sorted positions at which a brick's run starts
hipError_t sort_brick_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout,
                            uint32_t n, hipStream_t st);
hipError_t launch_add_resolve(const GridParams& g, float4* pts_rw, const float4* add, const unsigned long long* tab, uint32_t cap,
                              uint32_t n, double ds, uint8_t* dead_id, uint32_t* live, uint32_t* ctr, uint8_t* alive_new, hipStream_t st,
                              const uint32_t* dev_counts = nullptr);
hipError_t launch_delete_boxes(const GridParams& g, float4* pts_rw, uint32_t n_slots, const float* boxes, int nb, uint8_t* dead_id,
                               uint32_t* live, uint32_t* ctr, hipStream_t st);
// after a removal: bricks whose live points fell below half of their storage range are compacted in place (ctr[4] += purged bricks)
hipError_t launch_brick_purge(const GridParams& g, float4* pts, uint32_t* starts, const uint32_t* live, uint32_t* ctr, uint32_t nrows,
                              uint32_t pts_cap, hipStream_t st);
hipError_t launch_ins_prepare(const GridParams& g, const float4* add, const uint8_t* alive_new, const uint32_t* incl, uint32_t n,
                              uint32_t n_ids, float4* map_orig, uint8_t* dead_id, float4* ins, uint32_t* keys,
                              uint32_t* vals, uint32_t* ctr, hipStream_t st);
hipError_t launch_brick_rewrite(const GridParams& g, float4* pts, uint32_t* starts, uint2* hash, uint32_t* cap_end, uint32_t* live,
                                uint32_t* ctr, const float4* ins, const uint32_t* ks, const uint32_t* perm, uint32_t n,
                                uint32_t pts_cap, uint32_t rows_cap, hipStream_t st, const uint32_t* dev_counts = nullptr);
// behind launch_ins_sort_small: the workgroups share out the listed heads, the last one to finish publishes the change's counters
// (what launch_map_publish does for the general path)
hipError_t launch_brick_rewrite_heads(const GridParams& g, float4* pts, uint32_t* starts, uint2* hash, uint32_t* cap_end, uint32_t* live,
                                      uint32_t* ctr, const float4* ins, const uint32_t* ks, const uint32_t* perm, uint32_t n,
                                      uint32_t pts_cap, uint32_t rows_cap, hipStream_t st, const uint32_t* dev_counts,
                                      const uint32_t* heads, uint32_t* tick /* brick_ticket_words() words, zero on entry and on exit */,
                                      const uint32_t* n_alive, uint32_t* host_out, uint32_t seq);
uint32_t brick_ticket_words();
hipError_t launch_byte_flags(const uint8_t* in, uint32_t n, int invert, uint32_t* flags, hipStream_t st,
                             uint32_t* keys_sentinel = nullptr);
hipError_t launch_live_compact(const float4* map_orig, const uint32_t* flags, const uint32_t* incl, uint32_t n_ids, float4* out,
                               hipStream_t st);

// ---- flh_scanprep.hip: pcl::VoxelGrid of the scan ----
uint32_t undistort_blocks(uint32_t n);
hipError_t launch_undistort(const StateDev& s_end, const double* poses, int n_pose, const float4* raw, uint32_t n, float4* out,
                            unsigned long long* block_min, hipStream_t st);
hipError_t launch_cloud_body_to_world(const StateDev& s, const float4* in, uint32_t n, float4* out, hipStream_t st);
hipError_t launch_vg_keys(const float4* raw, uint32_t n, float inv, const int min_b[3], int mul1, int mul2,
                          unsigned long long* keys, uint32_t* vals, hipStream_t st);
hipError_t sort_vg_pairs(void* tmp, size_t& tmp_bytes, const unsigned long long* kin, unsigned long long* kout,
                         const uint32_t* vin, uint32_t* vout, uint32_t n, hipStream_t st);
hipError_t launch_vg_heads(const unsigned long long* keys_sorted, uint32_t n, uint32_t* flags, hipStream_t st);
hipError_t launch_vg_reduce(const float4* raw, const unsigned long long* keys_sorted, const uint32_t* vals_sorted,
                            const uint32_t* flags, const uint32_t* incl, uint32_t n, float4* out, hipStream_t st);

}  // namespace flh
