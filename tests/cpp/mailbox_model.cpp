// CPU model of the mailbox protocol of the pre-launched no-search pass (fast_lio_amd/csrc/flh_mail_dev.hpp,
// flh_prelaunch_host.inc): the DECISIONS of the device side (forwarder wave, every workgroup's wait) and of the host side (post
// go / abort, the sequence numbers) restated with std::atomic and threads, driven through the cases the GPU code must survive:
//   (round 6: a box is two 64-byte lines, each carrying its own {checksum, command, sequence} word, and a poll reads all sixteen
//   words at once -- possibly torn against the writer; the checksums catch that)
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
#include <cstring>
#include <random>
#include <thread>
#include <vector>

using clk = std::chrono::steady_clock;
constexpr uint32_t kGo = 1, kAbort = 2, kGone = 1, kLost = 2;
constexpr auto kForwardPatience = std::chrono::milliseconds(20);
constexpr auto kSpinPatience = std::chrono::milliseconds(400);

// sixteen 64-bit words = two 64-byte lines: words 0..6 / 8..14 the state's doubles, words 7 / 15 = checksum << 40 | cmd << 32 | seq
struct Box { std::atomic<uint64_t> w[16]; };
static Box host_box, dev_box;
static std::atomic<uint64_t> status{0};

static uint64_t mail_mix(uint64_t bits, int i) { return (bits ^ (bits >> 29)) * (0x9E3779B97F4A7C15ull * (uint64_t)(2 * i + 1)); }
static uint32_t mail_fold(uint64_t x) { return (uint32_t)((x ^ (x >> 24) ^ (x >> 48)) & 0xFFFFFFull); }
static uint64_t bits_of(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }
static double double_of(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }

struct LaunchResult { int ran = 0, skipped = 0; bool state_ok = true; };

// flh_mail_dev.hpp: mail_poll -- ONE read of the sixteen words (here: sixteen relaxed loads in any order, i.e. possibly torn against
// the writer); true when both lines announce `seq` with one command and both checksums hold
static bool poll(Box& box, uint32_t seq, uint64_t (&u)[16], uint32_t& cmd, bool& passed_over) {
    for (int i = 15; i >= 0; --i) u[i] = box.w[i].load(std::memory_order_relaxed);  // (words first: the worst order for tearing)
    const int32_t a0 = (int32_t)((uint32_t)u[7] - seq), a1 = (int32_t)((uint32_t)u[15] - seq);
    passed_over = a0 > 0 || a1 > 0;
    cmd = (uint32_t)(u[7] >> 32) & 0xFFu;
    if (passed_over || a0 != 0 || a1 != 0 || (((u[7] ^ u[15]) >> 32) & 0xFFu) != 0) return false;
    for (int l = 0; l < 2; ++l) {
        uint64_t ck = 0;
        for (int i = 0; i < 7; ++i) ck ^= mail_mix(u[8 * l + i], i);
        if (mail_fold(ck) != (uint32_t)(u[8 * l + 7] >> 40)) return false;
    }
    return true;
}
static void write_box(Box& box, const uint64_t (&u)[16]) {  // the lines' doubles, then the two words that announce them
    for (int i = 0; i < 16; ++i)
        if ((i & 7) != 7) box.w[i].store(u[i], std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_release);
    box.w[7].store(u[7], std::memory_order_release);
    box.w[15].store(u[15], std::memory_order_release);
}

// one workgroup of a launch waiting for sequence number seq; wg 0 is also the forwarder (flh_mail_dev.hpp: mailbox_wait)
static void workgroup(int wg, uint32_t seq, std::atomic<int>* ran, std::atomic<int>* skipped, std::atomic<int>* bad_state) {
    if (wg == 0) {
        const auto t0 = clk::now();
        uint64_t u[16];
        uint32_t cmd = 0;
        bool gone = false;
        for (;;) {
            bool over = false;
            if (poll(host_box, seq, u, cmd, over)) break;
            if (over) { cmd = kAbort; break; }
            if (clk::now() - t0 > kForwardPatience) { gone = true; cmd = kAbort; break; }
            std::this_thread::yield();
        }
        if (gone) status.store(((uint64_t)kGone << 32) | seq, std::memory_order_release);
        if (cmd != kGo) {
            uint64_t ck = 0;
            for (int i = 0; i < 7; ++i) ck ^= mail_mix(0, i);
            for (int i = 0; i < 16; ++i) u[i] = (i & 7) == 7 ? (((uint64_t)mail_fold(ck) << 40) | ((uint64_t)kAbort << 32) | seq) : 0;
        }
        write_box(dev_box, u);
    }
    const auto t0 = clk::now();
    uint64_t u[16];
    uint32_t cmd = 0;
    for (;;) {
        bool over = false;
        if (poll(dev_box, seq, u, cmd, over)) break;
        if (clk::now() - t0 > kSpinPatience) {
            cmd = kAbort;
            status.store(((uint64_t)kLost << 32) | seq, std::memory_order_release);
            break;
        }
        std::this_thread::yield();
    }
    if (cmd != kGo) { skipped->fetch_add(1); return; }
    for (int i = 0; i < 14; ++i)
        if (double_of(u[i < 7 ? i : i + 1]) != (double)seq + 0.01 * i) bad_state->fetch_add(1);
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

// the host's post (flh_api.cpp: pre_post): the lines' doubles, then each line's {checksum, command, sequence} word
static void post(uint32_t seq, uint32_t cmd) {
    uint64_t u[16];
    for (int i = 0; i < 16; ++i) u[i] = host_box.w[i].load(std::memory_order_relaxed);
    if (cmd == kGo)
        for (int i = 0; i < 14; ++i) u[i < 7 ? i : i + 1] = bits_of((double)seq + 0.01 * i);
    for (int l = 0; l < 2; ++l) {
        uint64_t ck = 0;
        for (int i = 0; i < 7; ++i) ck ^= mail_mix(u[8 * l + i], i);
        u[8 * l + 7] = ((uint64_t)mail_fold(ck) << 40) | ((uint64_t)cmd << 32) | seq;
    }
    write_box(host_box, u);
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
