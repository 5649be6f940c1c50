// Developer tool (CPU only): how long after the LAST granule of a one-launch searching pass has been written does
// collect_granules return?  A thread plays the GPU (25 groups x 30 granules, last group first, ~0.7 us apart, the header last);
// measured with the old order of pick-up (wait for the header, then read everything: forced here through the three-launch
// setting) and with the order of arrival.  Cache-to-cache transfers between two cores stand in for lines the GPU wrote over
// PCIe; the library's host source is compiled into the program.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Xarch_host -mavx2 -x hip tools/pickup_probe.cpp \
//        -Lfast_lio_amd/lib -lfastlio_hip -Wl,-rpath,$PWD/fast_lio_amd/lib -o /tmp/pickup_probe && /tmp/pickup_probe
#include "../fast_lio_amd/csrc/flh_api.cpp"
#include <x86intrin.h>
static void put(double* g, double v, double seq) { _mm_store_pd(g, _mm_set_pd(seq, v)); }
int main() {
    flh_handle* h = new flh_handle();
    flh_default_config(&h->cfg);
    h->N = 100000; h->peer_n = 1; h->peer_rank = 0;
    h->h_gran = static_cast<double*>(aligned_alloc(64, 2 * kGranSect * 16)); std::memset(h->h_gran, 0, 2 * kGranSect * 16);
    h->gran_dst[0] = h->h_gran; h->h_gram = static_cast<double*>(aligned_alloc(64, 2048));
    const int nsl = 30, ng = 25;
    double seq = 0;
    for (int mode = 0; mode < 2; ++mode) {
        h->pass_ok = mode == 1;
        double tot = 0; int R = 2000;
        for (int rep = 0; rep < R + 100; ++rep) {
            ++seq;
            double* base = h->h_gran + ((uint64_t)seq & 1u) * kGranSect * 2;
            std::atomic<uint64_t> t_last{0};
            std::atomic<int> go{0};
            std::thread gpu([&] {
                while (!go.load()) {}
                for (int gi = ng - 1; gi >= 0; --gi) {
                    for (int k = 0; k < nsl; ++k) put(base + 2 * (1 + (size_t)gi * nsl + k), 1.0 + k, seq);
                    const uint64_t t0 = __rdtsc(); while (__rdtsc() - t0 < 1500) {}   // ~0.7 us between groups
                }
                put(base, (double)(ng * nsl), seq);
                t_last.store(__rdtsc());
            });
            go.store(1);
            collect_granules(h, seq, 1, 0);
            const uint64_t t1 = __rdtsc();
            gpu.join();
            if (rep >= 100) tot += (double)((int64_t)(t1 - t_last.load()));
        }
        std::printf("%s: return of collect_granules %.0f cycles (%.2f us at 2.1 GHz) after the header was written\n", mode ? "last group first" : "header first     ", tot / R, tot / R / 2100.0);
    }
    return 0;
}
