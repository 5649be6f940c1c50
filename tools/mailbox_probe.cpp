// Developer tool (GPU box): can a pass get under the 6 us launch floor (tools/launch_probe.cpp) when its kernel is enqueued BEFORE
// the host knows the state it is to be evaluated at?  Three ways to hand a 256-byte state to a kernel that is already in the queue
// are timed against the plain launch, each for 1 / 64 / 1 563 workgroups (1 563 = the units of a 100 000-point scan):
//
//   launch     hipLaunchKernelGGL with the state as kernel arguments, every workgroup counts itself done, the last one stores a
//              sequence word to pinned host memory, the host polls it.                      (today's design: state known first)
//   spin       the kernel of pass i+1 is enqueued while pass i runs.  Its workgroup 0 polls a MAILBOX in pinned host memory
//              {state, sequence}; when the sequence arrives it copies the state to device memory and releases a device flag all
//              other workgroups spin on (s_sleep between polls, every spin bounded by a time-out).
//   wait       hipStreamWaitValue32 on a signal word, then the kernel; the host writes state + signal.  The workgroups read the
//              state from pinned host memory (wait-direct) or from device memory behind a one-workgroup forwarder kernel
//              (wait-forward).
//
//   gate       (round 6) a ONE-WAVE gate kernel spins on the mailbox and forwards the state to device memory; the
//              pass kernel is queued BEHIND it on the same stream and reads the state from device memory: no workgroup of the big
//              kernel ever waits, the price is one kernel boundary between the gate's exit and the pass's first wave.
//
// Reported per variant: host writes the mailbox -> host sees the last workgroup's flag (the critical path of a pass beside its
// work), with the host idle in between and with 3 us of host work between the flag and the next mailbox write (the 23x23
// algebra: it gives the queue time to bring the next kernel up).
// Nothing here touches the library.  Every device spin gives up after 20 ms and raises an error word; the host gives up after 1 s.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/mailbox_probe.cpp -o tools/mailbox_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); }    \
    } while (0)

constexpr int kState = 32;                      // doubles handed over per pass (the library's StateDev is 27)
constexpr unsigned long long kSpinTicks = 2000000ull;  // 20 ms of the 100 MHz counter

struct Mail {  // pinned host memory, written by the host, read by the device
    double state[kState];
    uint32_t seq;
    uint32_t pad[15];
};
struct State32 { double v[kState]; };

// what every variant ends with: the workgroup consumes the state, counts itself done, the last one tells the host
__device__ __forceinline__ void finish(const double* s_state, double* sink, uint32_t* done_cnt, uint32_t* host_flag, uint32_t seq) {
    if (threadIdx.x == 0) {
        double a = 0;
        for (int i = 0; i < kState; ++i) a += s_state[i];
        if (a == 12345.678) sink[blockIdx.x & 63] = a;  // keeps the reads alive
        const uint32_t t = __hip_atomic_fetch_add(done_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(done_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next kernel of the stream
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void __launch_bounds__(256) k_launch(State32 st, double* sink, uint32_t* done_cnt, uint32_t* host_flag, uint32_t seq) {
    __shared__ double s_state[kState];
    if (threadIdx.x < kState) s_state[threadIdx.x] = st.v[threadIdx.x];
    __syncthreads();
    finish(s_state, sink, done_cnt, host_flag, seq);
}

__global__ void __launch_bounds__(256) k_spin(const Mail* mail, double* dev_state, uint32_t* dev_flag, double* sink, uint32_t* done_cnt,
                                              uint32_t* host_flag, uint32_t* err, uint32_t seq) {
    __shared__ double s_state[kState];
    __shared__ uint32_t s_ok;
    if (blockIdx.x == 0 && threadIdx.x < 64) {  // the forwarder wave
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool ok = true;
        while (__hip_atomic_load(&mail->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (ok) {
            if (threadIdx.x < kState) {
                const double v = __hip_atomic_load(&mail->state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&dev_state[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (threadIdx.x == 0) __hip_atomic_store(dev_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (threadIdx.x == 0) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dev_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // let the others go
        }
    }
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t ok = 1;
        while (__hip_atomic_load(dev_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_ok = ok;
    }
    __syncthreads();
    if (threadIdx.x < kState) s_state[threadIdx.x] = __hip_atomic_load(&dev_state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    (void)s_ok;
    finish(s_state, sink, done_cnt, host_flag, seq);
}

// behind hipStreamWaitValue32: the state straight from pinned host memory (every workgroup crosses PCIe) ...
__global__ void __launch_bounds__(256) k_wait_direct(const Mail* mail, double* sink, uint32_t* done_cnt, uint32_t* host_flag, uint32_t seq) {
    __shared__ double s_state[kState];
    if (threadIdx.x < kState) s_state[threadIdx.x] = __hip_atomic_load(&mail->state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    finish(s_state, sink, done_cnt, host_flag, seq);
}
// ... or forwarded to device memory by one workgroup of its own launch
__global__ void k_forward(const Mail* mail, double* dev_state) {
    if (threadIdx.x < kState) {
        const double v = __hip_atomic_load(&mail->state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&dev_state[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void __launch_bounds__(256) k_from_dev(const double* dev_state, double* sink, uint32_t* done_cnt, uint32_t* host_flag, uint32_t seq) {
    __shared__ double s_state[kState];
    if (threadIdx.x < kState) s_state[threadIdx.x] = __hip_atomic_load(&dev_state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    finish(s_state, sink, done_cnt, host_flag, seq);
}

// the gate: one wave polls the mailbox, forwards the state to device memory, exits -- the kernel queued behind it then starts
__global__ void __launch_bounds__(64) k_gate(const Mail* mail, double* dev_state, uint32_t* err, uint32_t seq) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    while (__hip_atomic_load(&mail->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > kSpinTicks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) {
        if (threadIdx.x == 0) __hip_atomic_store(err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    if (threadIdx.x < kState) {
        const double v = __hip_atomic_load(&mail->state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&dev_state[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct Ctx {
    hipStream_t st;
    Mail* mail;            // pinned
    uint32_t* host_flag;   // pinned
    uint32_t* err;         // pinned
    uint32_t* sig;         // signal memory for hipStreamWaitValue32 (null: not available)
    double *dev_state, *sink;
    uint32_t *dev_flag, *done_cnt;
};

static void busy_us(double t) {
    const auto a = clk::now();
    while (us(a, clk::now()) < t) __builtin_ia32_pause();
}
static bool wait_flag(const Ctx& c, uint32_t seq) {
    const auto a = clk::now();
    while (__atomic_load_n(c.host_flag, __ATOMIC_ACQUIRE) != seq) {
        __builtin_ia32_pause();
        if (us(a, clk::now()) > 1.0e6) return false;
    }
    return true;
}
static void post_mail(const Ctx& c, uint32_t seq, bool signal) {
    for (int i = 0; i < kState; ++i) c.mail->state[i] = (double)seq + 0.001 * i;
    __atomic_store_n(&c.mail->seq, seq, __ATOMIC_RELEASE);
    if (signal && c.sig) __atomic_store_n(c.sig, seq, __ATOMIC_RELEASE);
}

enum Variant { LAUNCH, SPIN, WAIT_DIRECT, WAIT_FORWARD, GATE };
static const char* vname(Variant v) { return v == LAUNCH ? "launch      " : v == SPIN ? "spin        " : v == WAIT_DIRECT ? "wait-direct " : v == GATE ? "gate        " : "wait-forward"; }

static void enqueue(const Ctx& c, Variant v, int G, uint32_t seq) {
    switch (v) {
        case LAUNCH: break;  // launched when the state is known
        case SPIN:
            hipLaunchKernelGGL(k_spin, dim3(G), dim3(256), 0, c.st, c.mail, c.dev_state, c.dev_flag, c.sink, c.done_cnt, c.host_flag, c.err, seq);
            break;
        case WAIT_DIRECT:
            CK(hipStreamWaitValue32(c.st, c.sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_wait_direct, dim3(G), dim3(256), 0, c.st, c.mail, c.sink, c.done_cnt, c.host_flag, seq);
            break;
        case GATE:
            hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, c.st, c.mail, c.dev_state, c.err, seq);
            hipLaunchKernelGGL(k_from_dev, dim3(G), dim3(256), 0, c.st, c.dev_state, c.sink, c.done_cnt, c.host_flag, seq);
            break;
        case WAIT_FORWARD:
            CK(hipStreamWaitValue32(c.st, c.sig, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_forward, dim3(1), dim3(64), 0, c.st, c.mail, c.dev_state);
            hipLaunchKernelGGL(k_from_dev, dim3(G), dim3(256), 0, c.st, c.dev_state, c.sink, c.done_cnt, c.host_flag, seq);
            break;
    }
}

// returns the mean time mailbox written -> flag seen (us); < 0 on a time-out
static double run(const Ctx& c, Variant v, int G, double host_work_us, uint32_t& seq, int R, double* enqueue_us) {
    const int W = 200;
    double t_crit = 0, t_enq = 0;
    enqueue(c, v, G, seq + 1);  // the first pass's kernel, ahead of its state
    for (int i = 0; i < R + W; ++i) {
        ++seq;
        const auto e0 = clk::now();
        enqueue(c, v, G, seq + 1);  // the NEXT pass's kernel goes into the queue before this pass's state exists
        const auto e1 = clk::now();
        if (host_work_us > 0) busy_us(host_work_us);
        const auto t0 = clk::now();
        if (v == LAUNCH) {
            State32 s;
            for (int k = 0; k < kState; ++k) s.v[k] = (double)seq + 0.001 * k;
            hipLaunchKernelGGL(k_launch, dim3(G), dim3(256), 0, c.st, s, c.sink, c.done_cnt, c.host_flag, seq);
        } else {
            post_mail(c, seq, v != SPIN && v != GATE);
        }
        if (!wait_flag(c, seq)) {
            std::printf("%s G=%d: the host gave up waiting for pass %u (error word %u)\n", vname(v), G, seq, *c.err);
            post_mail(c, seq + 1, true);  // release what is queued
            (void)hipStreamSynchronize(c.st);
            ++seq;
            return -1.0;
        }
        const auto t1 = clk::now();
        if (i >= W) { t_crit += us(t0, t1); t_enq += us(e0, e1); }
    }
    ++seq;
    if (v != LAUNCH) {  // the kernel that was enqueued ahead of a pass that never comes: release it
        post_mail(c, seq, true);
        if (!wait_flag(c, seq)) std::printf("%s G=%d: the trailing kernel did not finish\n", vname(v), G);
    }
    CK(hipStreamSynchronize(c.st));
    if (*c.err) { std::printf("%s G=%d: a device spin timed out (error word %u)\n", vname(v), G, *c.err); *c.err = 0; return -1.0; }
    *enqueue_us = t_enq / R;
    return t_crit / R;
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::puts("no device"); return 3; }
    CK(hipSetDevice(0));
    Ctx c{};
    CK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    CK(hipHostMalloc((void**)&c.mail, sizeof(Mail), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&c.host_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&c.err, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c.mail, 0, sizeof(Mail));
    *c.host_flag = 0;
    *c.err = 0;
    CK(hipMalloc((void**)&c.dev_state, sizeof(double) * kState));
    CK(hipMalloc((void**)&c.sink, sizeof(double) * 64));
    CK(hipMalloc((void**)&c.dev_flag, 64));
    CK(hipMalloc((void**)&c.done_cnt, 64));
    CK(hipMemset(c.dev_flag, 0, 64));
    CK(hipMemset(c.done_cnt, 0, 64));
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    c.sig = nullptr;
    if (can_wait) {
        if (hipExtMallocWithFlags((void**)&c.sig, 64, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); c.sig = nullptr; }
        else *c.sig = 0;
    }
    std::printf("hipStreamWaitValue32 %s\n", c.sig ? "available" : "NOT available: the wait variants are skipped");
    uint32_t seq = 0;
    const int R = 3000;
    for (Variant v : {LAUNCH, SPIN, GATE, WAIT_DIRECT, WAIT_FORWARD}) {  // the wait variants last: an unsupported signal write must not cost the others
        if ((v == WAIT_DIRECT || v == WAIT_FORWARD) && !c.sig) continue;
        for (int G : {1, 64, 1563}) {
            for (double hw : {0.0, 3.0}) {
                double enq = 0;
                const double t = run(c, v, G, hw, seq, R, &enq);
                if (t < 0) break;
                std::printf("%s workgroups %4d  host work %.0f us: state posted -> host sees the last workgroup's flag %6.2f us  (enqueue of the next pass's kernel %.2f us, off the critical path)\n",
                            vname(v), G, hw, t, enq);
                std::fflush(stdout);
            }
        }
    }
    return 0;
}
