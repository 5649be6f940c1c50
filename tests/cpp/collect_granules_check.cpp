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

// several ranks: every rank's section written by a thread of its own (last group first, header last); this rank's group count
// follows from h->N, the peers' are hints in h->sect_ng (0 = none yet, right, too large, too small)
static int scenario_peers(flh_handle* h, double seq, int nranks, int my_rank, const int* ng_true, const int* hint) {
    const int ncol = 6, nsl = flh::gram_slots_host(ncol) + 1;
    h->peer_n = nranks;
    h->peer_rank = my_rank;
    for (int r = 0; r < nranks; ++r) h->sect_ng[r] = hint[r];
    double* base = h->h_gran + ((uint64_t)seq & 1u) * (size_t)h->peer_n * kGranSect * 2;
    std::vector<std::vector<double>> val(nranks);
    std::mt19937_64 rng((uint64_t)seq * 131 + nranks);
    for (int r = 0; r < nranks; ++r) {
        val[r].resize((size_t)ng_true[r] * nsl);
        for (auto& v : val[r]) v = std::ldexp((double)(rng() % 2000003) - 1.0e6, (int)(rng() % 40) - 20);
        for (int gi = 0; gi < ng_true[r]; ++gi) val[r][(size_t)gi * nsl + nsl - 1] = (double)(rng() % 50);
    }
    std::vector<std::thread> gpus;
    for (int r = 0; r < nranks; ++r)
        gpus.emplace_back([&, r] {
            double* sect = base + (size_t)r * kGranSect * 2;
            std::this_thread::sleep_for(std::chrono::microseconds(17 * ((r * 7 + (int)seq) % 5)));
            for (int gi = ng_true[r] - 1; gi >= 0; --gi) {
                for (int k = 0; k < nsl; ++k) put(sect + 2 * (1 + (size_t)gi * nsl + k), val[r][(size_t)gi * nsl + k], seq);
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            put(sect, (double)(ng_true[r] * nsl), seq);
        });
    const int rc = collect_granules(h, seq, 1, 0);
    for (auto& t : gpus) t.join();
    if (rc != 0) { std::printf("collect_granules (peers) failed: %s\n", flh_last_error()); return 1; }
    std::vector<double> want(nsl, 0.0);
    for (int r = 0; r < nranks; ++r)
        for (int gi = 0; gi < ng_true[r]; ++gi)
            for (int k = 0; k < nsl; ++k) want[k] += val[r][(size_t)gi * nsl + k];
    int bad = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = r; c < 16; ++c) {
            const int sl = flh::gram_slot_host(r, c, ncol);
            if (sl >= 0 && std::memcmp(&want[sl], &h->h_gram[r * 16 + c], 8) != 0) ++bad;
        }
    { const int sl = flh::gram_slot_host(14, 13, ncol); if (std::memcmp(&want[sl], &h->h_gram[14 * 16 + 13], 8) != 0) ++bad; }
    for (int r = 0; r < nranks; ++r)
        if (r != my_rank && hint[r] != ng_true[r] && h->sect_ng[r] != ng_true[r]) ++bad;  // a wrong hint is replaced by the header's count
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
    // peers (a scan sharded over ranks): this rank's shard of 40 000 points publishes 10 groups
    h->pass_ok = true;
    h->N = 40000;
    std::free(h->h_gran);
    h->h_gran = static_cast<double*>(aligned_alloc(64, 2 * 3 * kGranSect * 16));
    std::memset(h->h_gran, 0, 2 * 3 * kGranSect * 16);
    {
        const int own = (flh::pass_blocks(40000) + gran_group_size(40000) - 1) / gran_group_size(40000);
        const int t2[2] = {own, 7}, t3[3] = {9, own, 12};
        const int none2[2] = {0, 0}, right2[2] = {own, 7}, big2[2] = {own, 11}, small2[2] = {own, 3};
        const int none3[3] = {0, 0, 0}, right3[3] = {9, own, 12}, mixed3[3] = {14, own, 5};
        for (int rep = 0; rep < 4; ++rep) {
            bad += scenario_peers(h, ++seq, 2, 0, t2, none2);    // first pass of a scan: nothing learnt yet (header first)
            bad += scenario_peers(h, ++seq, 2, 0, t2, right2);
            bad += scenario_peers(h, ++seq, 2, 0, t2, big2);     // the peer's shard shrank since the hint was learnt
            bad += scenario_peers(h, ++seq, 2, 0, t2, small2);   // ... or grew
            bad += scenario_peers(h, ++seq, 3, 1, t3, none3);
            bad += scenario_peers(h, ++seq, 3, 1, t3, right3);
            bad += scenario_peers(h, ++seq, 3, 1, t3, mixed3);
        }
    }
    std::printf("%s\n", bad ? "collect_granules: MISMATCH" : "collect_granules: every order of arrival gives the group-order sums");
    return bad ? 1 : 0;
}
