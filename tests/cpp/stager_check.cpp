// CPU check of the hand-over of staging jobs to a handle's staging thread (flh_scan_stage_async -> stager_main -> wait_slot in
// flh_api.cpp).  Without a device every job fails at once at hipSetDevice, which is all this check needs: no job may be lost, whatever the rhythm of the caller (jobs arriving inside and outside the thread's polling window), and the thread
// must stop when asked.  Also prints what the hand-over costs its caller.  The library's host source is compiled into the program.
#include "../../fast_lio_amd/csrc/flh_api.cpp"

#include <x86intrin.h>

int main() {
    flh_handle* h = new flh_handle();
    flh_default_config(&h->cfg);
    h->device = 0;
    float pts[12] = {0};
    int bad = 0;
    for (int gap_us : {0, 20, 140, 600}) {  // 600 us: longer than the polling window, the thread sleeps in between
        double tot = 0;
        const int R = gap_us >= 600 ? 300 : 3000;
        for (int i = 0; i < R; ++i) {
            const int slot = i % 4;
            const uint64_t t0 = __rdtsc();
            if (flh_scan_stage_async(h, slot, pts, 12, 1) != 0) { std::printf("stage_async: %s\n", flh_last_error()); return 1; }
            tot += (double)(__rdtsc() - t0);
            const auto w0 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - w0 < std::chrono::microseconds(gap_us)) {}
            if (wait_slot(h, h->slots[slot]) == 0) ++bad;      // every job comes back with the device error
            if (h->slots[slot].pending) ++bad;
        }
        std::printf("caller busy %3d us between jobs: flh_scan_stage_async costs its caller %.0f cycles\n", gap_us, tot / R);
    }
    stop_stager(h);
    std::printf("%s\n", bad ? "stager: LOST JOBS" : "stager: every job handed over and reported");
    return bad ? 1 : 0;
}
