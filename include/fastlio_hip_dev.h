/* fastlio_hip_dev.h -- developer entry points of libfastlio_hip.so: instrumentation readers that only do something in variant builds
 * made by tools/variant.py (-DFLH_BOUNDS, -DFLH_PASS_STAMPS).  Not part of the drop-in boundary (include/fastlio_hip.h); the
 * product library exports them as stubs that return 0 so that the tools load against either build. */
#pragma once
#include "fastlio_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Developer builds only (-DFLH_BOUNDS: every computed device index checked against its buffer's capacity): returns 1 and the
 * violation records of the four kernel translation units, 5 words each {count, site, index, capacity, workgroup}; the product
 * library checks nothing, returns 0 and zeros. */
int flh_debug_bounds(flh_handle* h, uint64_t out[20]);
/* Developer builds only (-DFLH_PASS_STAMPS): 12 words per wave of the last one-launch pass (8 time stamps
 * at 100 MHz, HW_ID, XCC_ID, the longest candidate list among the wave's queries, its open queries); returns 1, else 0. */
int flh_debug_pass_stamps(flh_handle* h, uint64_t* out, size_t words);
/* The active scan's device order: order[i] = original index of the point at internal position i (what the staging's sort produced;
 * the tests compare the library's own staging kernels with the vendor sort through it). */
int flh_debug_scan_order(flh_handle* h, uint32_t* order);
/* Searching passes that were run a second time: a scan's first search is enqueued behind the previous scan's map change before the
 * host has seen the change's counters (flh_eval_begin); when they then ask for a re-index or a replay (flh_eval_end), the pass is
 * repeated on the settled map.  The tests provoke both cases and count them here. */
int flh_debug_search_redone(const flh_handle* h, uint64_t* out);
/* Where a scan's staging and its activation spend their time (counters since the last reset):
 * out = {stagings, enqueue us (sum), (max), wait for the H2D copy us (sum), (max),
 *        activations, wait for the slot's staging us (sum), (max), activations whose device-side event was not ready, activations
 *        that found the slot still with the staging thread}. */
int flh_debug_stage_stats(flh_handle* h, double out[10], int reset);
#ifdef __cplusplus
}
#endif
