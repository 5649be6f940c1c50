// flh_mail_dev.hpp -- the mailbox of the pre-launched no-search pass (k_fit_mb, flh_kernels.hip; host side: flh_api.cpp,
// flh_eval_expect_next).  Landed from round 4's experiment after its same-box A/B (profiles/r05_call1/).
// A kernel that is ENQUEUED BEFORE the host knows the state it is to be evaluated at: the no-search pass that follows a pass
// of the iterated update is launched while that pass still runs, becomes resident when it retires, and waits for a MAILBOX in
// pinned host memory.  When the host has the next state it writes 14 doubles and a sequence word -- no launch call, no queue
// processing, no dispatch on the critical path (tools/launch_probe.cpp: 6.05 us launch -> flag; tools/mailbox_probe.cpp times
// this hand-over in isolation).
//
//   host box   (pinned, 16 doubles = two 64-byte lines)   line 0: d[0..6] = state[0..6], d[7] = word; line 1: d[8..14] = state[7..13],
//                                     d[15] = word;  word = checksum of the line's seven doubles << 40 | cmd << 32 | seq, each
//                                     written after its line's doubles (x86 stores are not reordered with older stores)
//   device box (HBM, 16 doubles)      the same sixteen words, forwarded by wave 0 of workgroup 0; every workgroup's first sixteen
//                                     lanes poll them
//   status     (pinned, 64 bits)      {code, seq}: kGone -- nobody came for kMailForwardTicks, the launch did nothing (the host falls
//                                     back to a plain launch); kLost -- a workgroup never saw the forwarded words (a bug)
//
// A poll IS the read of the state (round 6; round 5 polled one word and then read the state: two PCIe round trips per pass for the
// forwarder, two L2 round trips for everybody else): sixteen lanes load the sixteen words in one instruction; when both words
// announce this launch and both checksums hold, the state is in the lanes already.  A read torn against the writer's stores fails
// a checksum and is simply repeated.  Same box, alternating (profiles/r06_call13/): no-search pass 15.6 -> 14.3 us, value +2.3 %
// at 300 steps, +4.5 % on the 20-step command.
//
// Hand-offs as everywhere in this library (flh_fit_dev.hpp): write-through stores at the scope of the reader, drained (vmcnt(0))
// before the words that announce them; cache-bypassing loads.  No fences, no L2 sweeps.
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

// the words' checksum: xor of the line's seven terms, folded to 24 bits (host: flh_api.cpp pre_post)
__host__ __device__ inline unsigned long long mail_mix(unsigned long long bits, int i) {
    const unsigned long long k = 0x9E3779B97F4A7C15ull * (unsigned long long)(2 * i + 1);
    return (bits ^ (bits >> 29)) * k;
}
__host__ __device__ inline uint32_t mail_fold(unsigned long long x) { return (uint32_t)((x ^ (x >> 24) ^ (x >> 48)) & 0xFFFFFFull); }

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

// One poll of a box by the first sixteen lanes of a wave (the others idle): true when both lines announce launch `seq` with one
// command and both checksums hold -- then u holds the box's sixteen words (lane l: word l) and cmd the command; passed_over: a LATER
// launch's mail is in the box (this launch was aborted).  SYSTEM: the pinned host box, else the device box.
template <bool SYSTEM>
__device__ __forceinline__ bool mail_poll(const double* box, uint32_t seq, int lane, unsigned long long& u, uint32_t& cmd, bool& passed_over) {
    typedef __attribute__((address_space(1))) const unsigned long long gcu64;
    u = 0ull;
    if (lane < 16) {
        if (SYSTEM) u = __hip_atomic_load((gcu64*)box + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else u = __hip_atomic_load((gcu64*)box + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t w0lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 7), w0hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 7);
    const uint32_t w1lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, 15), w1hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), 15);
    const int32_t a0 = (int32_t)(w0lo - seq), a1 = (int32_t)(w1lo - seq);
    passed_over = a0 > 0 || a1 > 0;
    cmd = w0hi & 0xFFu;
    if (passed_over || a0 != 0 || a1 != 0 || ((w0hi ^ w1hi) & 0xFFu) != 0u) return false;
    unsigned long long c = (lane < 16 && (lane & 7) != 7) ? mail_mix(u, lane & 7) : 0ull;
#pragma unroll
    for (int o = 1; o <= 4; o <<= 1) c ^= (unsigned long long)__shfl_xor((long long)c, o, 64);
    const uint32_t c0 = mail_fold((unsigned long long)__shfl((long long)c, 0, 64)), c1 = mail_fold((unsigned long long)__shfl((long long)c, 8, 64));
    return c0 == (w0hi >> 8) && c1 == (w1hi >> 8);
}

// Called by ALL threads of a workgroup (256) before anything else.  Returns false (workgroup-uniform) when the launch is to do
// nothing.  On success s holds the state.  s_cmd: one word of LDS; s_box: sixteen doubles of LDS.
__device__ __forceinline__ bool mailbox_wait(const MailArgs& m, StateDev& s, uint32_t* s_cmd, double* s_box) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const int t = threadIdx.x;
    if (t < 64) {  // ---- wave 0 of every workgroup
        const int lane = t;
        if (blockIdx.x == 0) {  // the forwarder: host box -> device box
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long u = 0;
            uint32_t cmd = 0;
            bool gone = false;
            for (;;) {
                bool over = false;
                if (mail_poll<true>(m.host_box, m.seq, lane, u, cmd, over)) break;
                if (over) { cmd = kMailAbort; break; }  // this launch was passed over, i.e. aborted
                if (__builtin_amdgcn_s_memrealtime() - t0 > kMailForwardTicks) { gone = true; cmd = kMailAbort; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
            if (gone && lane == 0)
                __hip_atomic_store((gu64*)m.status, ((unsigned long long)kMailGone << 32) | m.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (cmd != kMailGo) {  // nothing to hand on but the verdict: a box with zeros in its lines and the abort command
                u = 0ull;
                unsigned long long ck = 0;
                for (int i = 0; i < 7; ++i) ck ^= mail_mix(0ull, i);
                if ((lane & 7) == 7) u = ((unsigned long long)mail_fold(ck) << 40) | ((unsigned long long)kMailAbort << 32) | m.seq;
            }
            // the lines' doubles first, drained, then the two words that announce them
            if (lane < 16 && (lane & 7) != 7) __hip_atomic_store((gu64*)m.dev_box + lane, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < 16 && (lane & 7) == 7) __hip_atomic_store((gu64*)m.dev_box + lane, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every workgroup (the forwarder's too): wait for the forwarded box
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long u = 0;
        uint32_t cmd = 0;
        for (;;) {
            bool over = false;
            if (mail_poll<false>(m.dev_box, m.seq, lane, u, cmd, over)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > kMailSpinTicks) {
                cmd = kMailAbort;
                if (lane == 0) __hip_atomic_store((gu64*)m.status, ((unsigned long long)kMailLost << 32) | m.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane < 16) s_box[lane] = __longlong_as_double((long long)u);
        if (lane == 0) *s_cmd = cmd;
    }
    __syncthreads();
    if (*s_cmd != kMailGo) return false;
    // the state: one LDS read per lane, then into SGPRs (a kernel argument's place)
    const int lane = t & 63;
    const double v = s_box[lane < 16 ? lane : 0];
    s.rot[0] = lane_bcast_f64(v, 0); s.rot[1] = lane_bcast_f64(v, 1); s.rot[2] = lane_bcast_f64(v, 2); s.rot[3] = lane_bcast_f64(v, 3);
    s.pos[0] = lane_bcast_f64(v, 4); s.pos[1] = lane_bcast_f64(v, 5); s.pos[2] = lane_bcast_f64(v, 6);
    s.offR[0] = lane_bcast_f64(v, 8); s.offR[1] = lane_bcast_f64(v, 9); s.offR[2] = lane_bcast_f64(v, 10); s.offR[3] = lane_bcast_f64(v, 11);
    s.offT[0] = lane_bcast_f64(v, 12); s.offT[1] = lane_bcast_f64(v, 13); s.offT[2] = lane_bcast_f64(v, 14);
    return true;
}

}  // namespace flh
