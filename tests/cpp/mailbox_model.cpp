// CPU model of the mailbox protocol of the pre-launched no-search pass (fast_lio_amd/csrc/flh_mail_dev.hpp,
// flh_prelaunch_host.inc): the DECISIONS of the device side (forwarder wave, every workgroup's wait) and of the host side (post
// go / abort, the sequence numbers) restated with std::atomic and threads, driven through the cases the GPU code must survive:
//   go          the state arrives, every workgroup of the launch runs once with exactly that state
//   abort       the launch does nothing
//   passed over the host has already posted a LATER launch's mail when this launch's forwarder looks: treated as an abort
//   gone        nobody posts within the forwarder's patience: status = {gone, seq}, the launch does nothing, a later "go" for the
//               same sequence number finds nobody (the host falls back to a plain launch)
// Launches of one stream run one after the other (the model's "stream" joins a launch before it starts the next), exactly one
// mailbox, one device box.  It tests the logic, not the memory scopes: those are the GPU's to prove (tools/prelaunch_check.py).
// Build and run: g++ -O2 -std=c++17 -pthread tests/cpp/mailbox_model.cpp -o /tmp/mailbox_model && /tmp/mailbox_model
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <random>
#include <thread>
#include <vector>

using clk = std::chrono::steady_clock;
constexpr uint32_t kGo = 1, kAbort = 2, kGone = 1, kLost = 2;
constexpr auto kForwardPatience = std::chrono::milliseconds(20);
constexpr auto kSpinPatience = std::chrono::milliseconds(400);

struct Box { std::atomic<double> d[15]; std::atomic<uint64_t> word{0}; };
static Box host_box, dev_box;
static std::atomic<uint64_t> status{0};

struct LaunchResult { int ran = 0, skipped = 0; bool state_ok = true; };

// one workgroup of a launch waiting for sequence number seq; wg 0 is also the forwarder (flh_mail_dev.hpp: mailbox_wait)
static void workgroup(int wg, uint32_t seq, std::atomic<int>* ran, std::atomic<int>* skipped, std::atomic<int>* bad_state) {
    if (wg == 0) {
        const auto t0 = clk::now();
        uint64_t w = 0;
        bool ok = true;
        for (;;) {
            w = host_box.word.load(std::memory_order_acquire);
            const int32_t ahead = (int32_t)((uint32_t)w - seq);
            if (ahead >= 0) {
                if (ahead > 0) w = ((uint64_t)kAbort << 32) | seq;
                break;
            }
            if (clk::now() - t0 > kForwardPatience) { ok = false; break; }
            std::this_thread::yield();
        }
        if (ok) {
            if ((uint32_t)(w >> 32) == kGo)
                for (int i = 0; i < 14; ++i) dev_box.d[i].store(host_box.d[i].load(std::memory_order_relaxed), std::memory_order_relaxed);
        } else {
            w = ((uint64_t)kAbort << 32) | seq;
            status.store(((uint64_t)kGone << 32) | seq, std::memory_order_release);
        }
        dev_box.word.store(w, std::memory_order_release);
    }
    const auto t0 = clk::now();
    uint64_t w = 0;
    for (;;) {
        w = dev_box.word.load(std::memory_order_acquire);
        if ((uint32_t)w == seq) break;
        if (clk::now() - t0 > kSpinPatience) {
            w = ((uint64_t)kAbort << 32) | seq;
            status.store(((uint64_t)kLost << 32) | seq, std::memory_order_release);
            break;
        }
        std::this_thread::yield();
    }
    if ((uint32_t)(w >> 32) != kGo) { skipped->fetch_add(1); return; }
    for (int i = 0; i < 14; ++i)
        if (dev_box.d[i].load(std::memory_order_relaxed) != (double)seq + 0.01 * i) bad_state->fetch_add(1);
    ran->fetch_add(1);
}

static LaunchResult run_launch(uint32_t seq, int nwg) {  // a launch: all its workgroups, joined (the stream's order)
    std::atomic<int> ran{0}, skipped{0}, bad{0};
    std::vector<std::thread> t;
    for (int w = 0; w < nwg; ++w) t.emplace_back(workgroup, w, seq, &ran, &skipped, &bad);
    for (auto& x : t) x.join();
    LaunchResult r;
    r.ran = ran; r.skipped = skipped; r.state_ok = bad == 0;
    return r;
}

// the host's post (flh_prelaunch_host.inc: pre_post): state, then the {sequence, command} word
static void post(uint32_t seq, uint32_t cmd) {
    if (cmd == kGo)
        for (int i = 0; i < 14; ++i) host_box.d[i].store((double)seq + 0.01 * i, std::memory_order_relaxed);
    host_box.word.store(((uint64_t)cmd << 32) | seq, std::memory_order_release);
}

int main() {
    int fails = 0;
    auto expect = [&](bool c, const char* what, uint32_t seq) { if (!c) { std::printf("FAIL (%s) at sequence %u\n", what, seq); ++fails; } };
    std::mt19937 rng(12345);
    const int nwg = 12;
    uint32_t seq = 0;
    for (int round = 0; round < 300; ++round) {
        ++seq;
        const int kind = round % 5;  // 0,1: go  2: abort  3: passed over  4: (every 60th) gone
        const int delay_us = (int)(rng() % 300);
        if (kind == 4 && round % 60 != 4) {  // keep the slow case rare: a plain go instead
            std::thread host([&] { std::this_thread::sleep_for(std::chrono::microseconds(delay_us)); post(seq, kGo); });
            const LaunchResult r = run_launch(seq, nwg);
            host.join();
            expect(r.ran == nwg && r.skipped == 0 && r.state_ok, "go", seq);
            continue;
        }
        if (kind <= 1) {  // posted before or while the launch waits
            const bool before = (rng() & 1) != 0;
            if (before) post(seq, kGo);
            std::thread host([&] { if (!before) { std::this_thread::sleep_for(std::chrono::microseconds(delay_us)); post(seq, kGo); } });
            const LaunchResult r = run_launch(seq, nwg);
            host.join();
            expect(r.ran == nwg && r.skipped == 0 && r.state_ok, "go", seq);
        } else if (kind == 2) {
            std::thread host([&] { std::this_thread::sleep_for(std::chrono::microseconds(delay_us)); post(seq, kAbort); });
            const LaunchResult r = run_launch(seq, nwg);
            host.join();
            expect(r.ran == 0 && r.skipped == nwg, "abort", seq);
        } else if (kind == 3) {  // the host aborted this launch and has already posted the NEXT one's go when it gets to look
            post(seq, kAbort);
            post(seq + 1, kGo);
            const LaunchResult r = run_launch(seq, nwg);
            expect(r.ran == 0 && r.skipped == nwg, "passed over", seq);
            ++seq;
            const LaunchResult r2 = run_launch(seq, nwg);  // the next launch finds its own mail
            expect(r2.ran == nwg && r2.state_ok, "go after a passed-over launch", seq);
        } else {  // gone: nobody posts in time
            const LaunchResult r = run_launch(seq, nwg);
            expect(r.ran == 0 && r.skipped == nwg, "gone: nothing ran", seq);
            expect(status.load() == (((uint64_t)kGone << 32) | seq), "gone: status", seq);
            post(seq, kGo);  // the late host: its mail meets nobody; it reads the status and launches the usual way (not modelled)
        }
    }
    std::printf("%s (%u launches)\n", fails ? "FAILED" : "mailbox model: all cases as specified", seq);
    return fails ? 1 : 0;
}
