// Developer tool (GPU box): what a pass costs beside its kernel.  Times, through the C ABI only,
//   * a no-search and a searching evaluation of scans of 64 / 6 400 / 100 000 points (wall time per flh_eval, and the host time
//     of flh_eval_begin alone: the launch call),
//   * the floor of any launch-based design: an empty kernel that stores a sequence word to pinned host memory, the host polling it.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/launch_probe.cpp -Lfast_lio_amd/lib -lfastlio_hip -Wl,-rpath,'$ORIGIN/../fast_lio_amd/lib' -o tools/launch_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fastlio_hip.h"

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

__global__ void k_flag(volatile double* out, double seq) {
    if (threadIdx.x == 0) __hip_atomic_store((double*)out, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main() {
    if (!flh_device_available()) { std::puts("no device"); return 3; }
    // ---- the floor: launch -> kernel -> pinned flag -> host sees it
    {
        double* flag = nullptr;
        hipHostMalloc(&flag, 64, hipHostMallocMapped);
        *flag = 0;
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        double seq = 0, t_launch = 0, t_all = 0;
        const int R = 5000;
        for (int i = 0; i < R + 200; ++i) {
            seq += 1;
            const auto t0 = clk::now();
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, flag, seq);
            const auto t1 = clk::now();
            while (*(volatile double*)flag != seq) __builtin_ia32_pause();
            const auto t2 = clk::now();
            if (i >= 200) { t_launch += us(t0, t1); t_all += us(t0, t2); }
        }
        std::printf("floor: empty kernel -> pinned flag: launch call %.2f us, launch -> host sees the flag %.2f us\n", t_launch / R, t_all / R);
        // the same through hipModuleLaunchKernel on the kernel's hipFunction_t (no host-function lookup): with a kernelParams array,
        // and with the arguments as one prepared buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER) -- round-5 question: is the 2.8 us the
        // runtime's bookkeeping or the packet's way to the device?
        hipFunction_t fn = nullptr;
        if (hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(&k_flag)) == hipSuccess && fn) {
            for (int mode = 0; mode < 2; ++mode) {
                t_launch = 0; t_all = 0;
                for (int i = 0; i < R + 200; ++i) {
                    seq += 1;
                    double* fp = flag;
                    void* params[2] = {&fp, &seq};
                    struct { double* f; double s; } buf = {flag, seq};
                    size_t bytes = sizeof(buf);
                    void* extra[5] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
                    const auto t0 = clk::now();
                    const hipError_t e = mode == 0 ? hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, params, nullptr)
                                                   : hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, st, nullptr, extra);
                    const auto t1 = clk::now();
                    if (e != hipSuccess) { std::printf("hipModuleLaunchKernel (%s): %s\n", mode ? "buffer" : "params", hipGetErrorString(e)); break; }
                    while (*(volatile double*)flag != seq) __builtin_ia32_pause();
                    const auto t2 = clk::now();
                    if (i >= 200) { t_launch += us(t0, t1); t_all += us(t0, t2); }
                }
                std::printf("floor: hipModuleLaunchKernel (%s): launch call %.2f us, launch -> host sees the flag %.2f us\n",
                            mode ? "one argument buffer" : "kernelParams", t_launch / R, t_all / R);
            }
        } else {
            std::printf("hipGetFuncBySymbol not available: hipModuleLaunchKernel variants skipped\n");
            (void)hipGetLastError();
        }
        hipStreamDestroy(st);
        hipHostFree(flag);
    }
    // ---- the passes
    const int side = 400;  // map: a 200 m x 200 m plane lattice at 0.5 m with a little relief
    std::vector<float> map((size_t)side * side * 3);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.f; };
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j) {
            float* p = &map[((size_t)i * side + j) * 3];
            p[0] = (i - side / 2) * 0.5f + 0.1f * (rnd() - 0.5f); p[1] = (j - side / 2) * 0.5f + 0.1f * (rnd() - 0.5f); p[2] = 0.02f * (rnd() - 0.5f);
        }
    flh_config cfg;
    flh_default_config(&cfg);
    flh_handle* h = nullptr;
    if (flh_create(&cfg, &h) != 0) { std::printf("create: %s\n", flh_last_error()); return 1; }
    if (flh_map_build(h, map.data(), 12, map.size() / 3) != 0) { std::printf("map: %s\n", flh_last_error()); return 1; }
    const double rot[4] = {0, 0, 0, 1}, pos[3] = {0, 0, 1.0}, offR[4] = {0, 0, 0, 1}, offT[3] = {0, 0, 0};
    for (int N : {64, 6400, 100000}) {
        std::vector<float> scan((size_t)N * 3);
        for (int i = 0; i < N; ++i) { scan[3 * i] = 180.f * (rnd() - 0.5f); scan[3 * i + 1] = 180.f * (rnd() - 0.5f); scan[3 * i + 2] = -1.0f + 0.02f * (rnd() - 0.5f); }
        if (flh_scan_upload(h, scan.data(), 12, N) != 0) { std::printf("scan: %s\n", flh_last_error()); return 1; }
        flh_set_timing_stride(h, 0);
        double HTH[144], HTh[12], tr;
        int64_t ne = 0;
        for (int srch = 1; srch >= 0; --srch) {
            const int R = N > 10000 ? 1000 : 3000;
            double t_begin = 0, t_all = 0;
            for (int i = 0; i < R + 100; ++i) {
                const auto t0 = clk::now();
                if (flh_eval_begin(h, rot, pos, offR, offT, srch, 0) != 0) { std::printf("eval: %s\n", flh_last_error()); return 1; }
                const auto t1 = clk::now();
                if (flh_eval_end(h, HTH, HTh, &ne, &tr) != 0) { std::printf("eval: %s\n", flh_last_error()); return 1; }
                const auto t2 = clk::now();
                if (i >= 100) { t_begin += us(t0, t1); t_all += us(t0, t2); }
            }
            std::printf("N=%6d %s pass: flh_eval_begin (launch call) %.2f us, whole evaluation %.2f us, n_eff %lld\n", N, srch ? "searching" : "no-search",
                        t_begin / R, t_all / R, (long long)ne);
        }
    }
    flh_destroy(h);
    return 0;
}
