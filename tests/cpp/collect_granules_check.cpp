// CPU check of flh_api.cpp's granule pick-up (collect_granules): a thread plays the GPU and writes {value, sequence} granules into
// the handle's buffer -- last group first and the header last, as a one-launch searching pass publishes them, or in ascending
// order, as k_fit does -- and the host must end with the groups added up in GROUP order, bit for bit, whatever order they came in.
// The library's source is compiled into this program (its internals are not part of the ABI); nothing here touches a device.
// Build: hipcc --offload-arch=gfx950 -O1 -std=c++17 -ffp-contract=off -x hip tests/cpp/collect_granules_check.cpp \
//        -Lfast_lio_amd/lib -lfastlio_hip -Wl,-rpath,$PWD/fast_lio_amd/lib -o /tmp/collect_granules_check
#include "../../fast_lio_amd/csrc/flh_api.cpp"

#include <random>

static void put(double* g, double v, double seq) { _mm_store_pd(g, _mm_set_pd(seq, v)); }  // one 16-byte store: {value, sequence}

static int scenario(flh_handle* h, double seq, int do_search, int ext, bool last_first, bool bad_header) {
    const int ncol = ext ? 12 : 6;
    const int nsl = flh::gram_slots_host(ncol) + 1;
    const int red = gran_group_size(h->N);
    const int ng = (flh::pass_blocks((int)h->N) + red - 1) / red;
    double* base = h->h_gran + ((uint64_t)seq & 1u) * (size_t)h->peer_n * kGranSect * 2;
    std::vector<double> val((size_t)ng * nsl);
    std::mt19937_64 rng((uint64_t)seq * 77 + ext);
    for (auto& v : val) v = std::ldexp((double)(rng() % 2000003) - 1.0e6, (int)(rng() % 40) - 20);
    for (int gi = 0; gi < ng; ++gi) val[(size_t)gi * nsl + nsl - 1] = (double)(rng() % 50);  // the statistic: a count
    std::thread gpu([&] {
        auto group = [&](int gi) {
            for (int k = 0; k < nsl; ++k) put(base + 2 * (1 + (size_t)gi * nsl + k), val[(size_t)gi * nsl + k], seq);
            std::this_thread::sleep_for(std::chrono::microseconds(30));
        };
        const double header = (double)(ng * nsl + (bad_header ? nsl : 0));
        if (last_first) {
            for (int gi = ng - 1; gi >= 0; --gi) group(gi);
            put(base, header, seq);
        } else {
            put(base, header, seq);
            for (int gi = 0; gi < ng; ++gi) group(gi);
        }
    });
    const uint64_t stat0 = h->n_second_stage;
    const int rc = collect_granules(h, seq, do_search, ext);
    gpu.join();
    if (bad_header) return rc != 0 ? 0 : 1;  // must be refused
    if (rc != 0) { std::printf("collect_granules failed: %s\n", flh_last_error()); return 1; }
    std::vector<double> want(nsl, 0.0);
    for (int gi = 0; gi < ng; ++gi)
        for (int k = 0; k < nsl; ++k) want[k] += val[(size_t)gi * nsl + k];
    int bad = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            const int sl = flh::gram_slot_host(r, c, ncol);
            double w = 0.0;
            if (sl >= 0) w = want[sl];
            else if (c < r && r < 12) { const int s2 = flh::gram_slot_host(c, r, ncol); if (s2 >= 0) w = want[s2]; }  // the mirrored half
            if (r == 15 && c == 15) w = seq;
            if (std::memcmp(&w, &h->h_gram[r * 16 + c], 8) != 0 && !(w == 0.0 && h->h_gram[r * 16 + c] == 0.0)) ++bad;
        }
    if (do_search && h->n_second_stage - stat0 != (uint64_t)want[nsl - 1]) ++bad;
    return bad;
}

int main() {
    flh_handle* h = new flh_handle();
    flh_default_config(&h->cfg);
    h->pass_ok = true;
    h->N = 100000;
    h->peer_n = 1;
    h->peer_rank = 0;
    h->h_gran = static_cast<double*>(aligned_alloc(64, 2 * kGranSect * 16));
    std::memset(h->h_gran, 0, 2 * kGranSect * 16);
    h->gran_dst[0] = h->h_gran;
    h->h_gram = static_cast<double*>(aligned_alloc(64, 256 * sizeof(double)));
    int bad = 0;
    double seq = 0;
    for (int rep = 0; rep < 6; ++rep) {
        for (int ext = 0; ext < 2; ++ext) {
            bad += scenario(h, ++seq, 1, ext, true, false);    // one-launch searching pass: last group first, header last
            bad += scenario(h, ++seq, 0, ext, false, false);   // no-search pass: ascending
            bad += scenario(h, ++seq, 1, ext, false, false);   // a searching pass whose groups happen to come in ascending order
        }
    }
    h->N = 777;  // a short scan: one partial group
    bad += scenario(h, ++seq, 1, 0, true, false);
    bad += scenario(h, ++seq, 0, 0, false, false);
    h->N = 100000;
    bad += scenario(h, ++seq, 1, 0, true, true);   // header that does not match the launch: refused
    h->pass_ok = false;                             // three-launch pass: k_fit publishes, ascending
    bad += scenario(h, ++seq, 1, 0, false, false);
    std::printf("%s\n", bad ? "collect_granules: MISMATCH" : "collect_granules: every order of arrival gives the group-order sums");
    return bad ? 1 : 0;
}
