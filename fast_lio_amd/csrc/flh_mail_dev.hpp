// flh_mail_dev.hpp -- the mailbox of the pre-launched no-search pass (k_fit_mb, flh_kernels.hip; host side: flh_api.cpp,
// flh_eval_expect_next).  Landed from round 4's experiment after its same-box A/B (profiles/r05_call1/).
// A kernel that is ENQUEUED BEFORE the host knows the state it is to be evaluated at: the no-search pass that follows a pass
// of the iterated update is launched while that pass still runs, becomes resident when it retires, and waits for a MAILBOX in
// pinned host memory.  When the host has the next state it writes 14 doubles and a sequence word -- no launch call, no queue
// processing, no dispatch on the critical path (tools/launch_probe.cpp: 6.05 us launch -> flag; tools/mailbox_probe.cpp times
// this hand-over in isolation).
//
//   host box   (pinned, 16 doubles)   d[0..13] StateDev, d[14] unused, d[15] = {seq, cmd} as ONE
//                                     64-bit word written last (x86 stores are not reordered with older stores)
//   device box (HBM, 16 doubles)      the same, forwarded by wave 0 of workgroup 0; every workgroup's thread 0 spins on d[15]
//   status     (pinned, 64 bits)      {code, seq}: kGone -- nobody came for kMailForwardTicks, the launch did nothing (the host falls
//                                     back to a plain launch); kLost -- a workgroup never saw the forwarded word (a bug)
//
// Hand-offs as everywhere in this library (flh_fit_dev.hpp): write-through stores at the scope of the reader, drained (vmcnt(0))
// before the word that announces them; readers poll that word and then read with cache-bypassing loads.  No fences, no L2 sweeps.
// Every spin is bounded: a launch can idle the GPU for kMailForwardTicks at most, it cannot hang it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flh_device.hpp"

namespace flh {

constexpr uint32_t kMailGo = 1u, kMailAbort = 2u;          // cmd
constexpr uint32_t kMailGone = 1u, kMailLost = 2u;         // status code
constexpr unsigned long long kMailForwardTicks = 2000000ull;   // 20 ms of the 100 MHz counter: the forwarder's patience
constexpr unsigned long long kMailSpinTicks = 10000000ull;     // 100 ms: everybody else's (the forwarder always answers first)

struct MailArgs {
    const double* host_box;
    double* dev_box;
    unsigned long long* status;
    uint32_t seq;  // the mailbox sequence number this launch waits for
};

__device__ __forceinline__ double lane_bcast_f64(double v, int src) {  // src: compile-time constant -> two v_readlane_b32, result in SGPRs
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Called by ALL threads of a workgroup (256) before anything else.  Returns false (workgroup-uniform) when the launch is to do
// nothing.  On success s holds the state.  s_cmd: one word of LDS.
__device__ __forceinline__ bool mailbox_wait(const MailArgs& m, StateDev& s, uint32_t* s_cmd) {
    typedef __attribute__((address_space(1))) const unsigned long long gcu64;
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    typedef __attribute__((address_space(1))) const double gcdouble;
    typedef __attribute__((address_space(1))) double gdouble_;
    const int t = threadIdx.x;
    if (blockIdx.x == 0 && t < 64) {  // ---- the forwarder: host box -> device box
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long w = 0;
        bool ok = true;
        for (;;) {
            w = __hip_atomic_load((gcu64*)(m.host_box + 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const int32_t ahead = (int32_t)((uint32_t)w - m.seq);
            if (ahead >= 0) {  // this launch's mail -- or already a later one's: this launch was passed over, i.e. aborted
                if (ahead > 0) w = ((unsigned long long)kMailAbort << 32) | m.seq;
                break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > kMailForwardTicks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        if (ok) {
            if (t < 14 && (uint32_t)(w >> 32) == kMailGo) {
                const double v = __hip_atomic_load((gcdouble*)(m.host_box + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store((gdouble_*)(m.dev_box + t), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            w = ((unsigned long long)kMailAbort << 32) | m.seq;
            if (t == 0) __hip_atomic_store((gu64*)m.status, ((unsigned long long)kMailGone << 32) | m.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the state has left this CU before the word that announces it
        if (t == 0) __hip_atomic_store((gu64*)(m.dev_box + 15), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 0) {  // ---- every workgroup: wait for the forwarded word
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long w = 0;
        for (;;) {
            w = __hip_atomic_load((gcu64*)(m.dev_box + 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)w == m.seq) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > kMailSpinTicks) {
                w = ((unsigned long long)kMailAbort << 32) | m.seq;
                __hip_atomic_store((gu64*)m.status, ((unsigned long long)kMailLost << 32) | m.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        *s_cmd = (uint32_t)(w >> 32);
    }
    __syncthreads();
    if (*s_cmd != kMailGo) return false;
    // the state: one cache-bypassing load per lane, then into SGPRs (a kernel argument's place)
    const int lane = t & 63;
    const double v = __hip_atomic_load((gcdouble*)(m.dev_box + (lane < 14 ? lane : 0)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s.rot[0] = lane_bcast_f64(v, 0); s.rot[1] = lane_bcast_f64(v, 1); s.rot[2] = lane_bcast_f64(v, 2); s.rot[3] = lane_bcast_f64(v, 3);
    s.pos[0] = lane_bcast_f64(v, 4); s.pos[1] = lane_bcast_f64(v, 5); s.pos[2] = lane_bcast_f64(v, 6);
    s.offR[0] = lane_bcast_f64(v, 7); s.offR[1] = lane_bcast_f64(v, 8); s.offR[2] = lane_bcast_f64(v, 9); s.offR[3] = lane_bcast_f64(v, 10);
    s.offT[0] = lane_bcast_f64(v, 11); s.offT[1] = lane_bcast_f64(v, 12); s.offT[2] = lane_bcast_f64(v, 13);
    return true;
}

}  // namespace flh
