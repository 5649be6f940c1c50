// mapping_loop.cpp -- the body of FAST-LIO2's main loop (src/laserMapping.cpp:863-1000) on top of libfastlio_hip.so,
// written with the mirrored types so that it reads like the reference.  ROS, PCL and the IMU initialisation are replaced
// by a tiny synthetic source (a flat floor + a wall, a sensor gliding over it); everything per scan is the product path:
//
//   p_imu->Process(...)        -> ImuProcess::UndistortPcl : predict per IMU sample on the host, backward sweep +
//   downSizeFilterSurf.filter        voxel grid + staging on the device (flh_scan_stage_undistorted)
//   lasermap_fov_segment()     -> flh_fov_segment
//   ikdtree.Build (first scan) -> flh_map_build
//   kf.update_iterated_dyn_share_modified(LASER_POINT_COV, ...) with h_share_model -> flh_eval per pass
//   map_incremental()          -> flh_map_incremental
//
// Build (from the repo root):  g++ -O2 -std=c++17 -Iinclude examples/mapping_loop.cpp -Lfast_lio_amd/lib -lfastlio_hip \
//                                  -Wl,-rpath,$PWD/fast_lio_amd/lib -o mapping_loop
// Without a HIP device it stops at flh_create with the library's error message (there is no CPU fallback).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "fastlio_amd/esekfom.hpp"
#include "fastlio_amd/h_share_model.hpp"
#include "fastlio_amd/imu_processing.hpp"
#include "fastlio_amd/local_map.hpp"
#include "fastlio_amd/use-ikfom.hpp"
#include "fastlio_hip.h"

typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
using fastlio_amd::V3;

struct PointType {  // pcl::PointXYZINormal as FAST-LIO2 uses it: curvature = time offset in ms
    float x, y, z, intensity, normal_x, normal_y, normal_z, curvature;
};

// A scan of the synthetic scene seen from `pos` (sensor axes = world axes): floor z = 0, wall x = 20.
static std::vector<PointType> make_scan(const double pos[3], std::mt19937& rng, int n) {
    std::uniform_real_distribution<double> az(-M_PI, M_PI), el(-0.9, 0.05), tm(0.0, 100.0);
    std::vector<PointType> out;
    while ((int)out.size() < n) {
        const double a = az(rng), e = el(rng);
        const double d[3] = {std::cos(e) * std::cos(a), std::cos(e) * std::sin(a), std::sin(e)};
        double t = 1e9;
        if (d[2] < -1e-3) t = std::min(t, -pos[2] / d[2]);
        if (d[0] > 1e-3) t = std::min(t, (20.0 - pos[0]) / d[0]);
        if (t > 60.0) continue;
        PointType p{};
        p.x = (float)(t * d[0]); p.y = (float)(t * d[1]); p.z = (float)(t * d[2]);
        p.curvature = (float)tm(rng);
        out.push_back(p);
    }
    return out;
}

int main() {
    flh_config cfg;
    flh_default_config(&cfg);
    flh_handle* g_flh = nullptr;
    if (flh_create(&cfg, &g_flh) != 0) {
        std::printf("flh_create: %s\n", flh_last_error());
        return 2;
    }
    const double filter_size_surf_min = 0.5, filter_size_map_min = 0.5, LASER_POINT_COV = 0.001;
    const double cube_len = 200.0;
    const float DET_RANGE = 100.0f;

    fastlio_amd::HShareContext hctx;
    hctx.handle = g_flh;
    kf_t kf;
    double epsi[23];
    for (double& e : epsi) e = 0.001;
    kf.init_dyn_share(get_f, df_dx, df_dw, static_cast<kf_t::measurementModel_dyn_share_ctx*>(fastlio_amd::h_share_model), 3, epsi,
                      &hctx);  // laserMapping.cpp:828 with an explicit context (examples/node_lines.cpp: the line as it stands)

    state_ikfom st = kf.get_x();
    st.pos[0] = 0; st.pos[1] = 0; st.pos[2] = 2.0;
    st.grav = S2(0, 0, -9.81);
    st.vel[0] = 1.0;
    kf.change_x(st);
    kf_t::cov P = kf.get_P();
    for (int i = 0; i < 23; ++i) P(i, i) = i < 6 ? 1e-4 : 1e-5;
    kf.change_P(P);

    fastlio_amd::ImuProcess p_imu;
    p_imu.mean_acc[0] = 0; p_imu.mean_acc[1] = 0; p_imu.mean_acc[2] = 9.81;  // what IMU_init would have measured at rest
    fastlio_amd::LocalMap local_map;
    local_map.cube_len = cube_len;
    local_map.DET_RANGE = DET_RANGE;
    flh_local_map lm{};
    std::mt19937 rng(1);
    bool map_built = false;
    double t = 0.0;
    p_imu.last_imu_ = fastlio_amd::ImuSample{t, {0, 0, 9.81}, {0, 0, 0}};

    for (int scan = 0; scan < 5; ++scan) {
        // --- the measurements of this sweep: 20 IMU samples at 200 Hz, 20000 LiDAR points over 100 ms
        std::vector<fastlio_amd::ImuSample> imu;
        for (int k = 1; k <= 20; ++k) imu.push_back(fastlio_amd::ImuSample{t + 0.005 * k, {0, 0, 9.81}, {0, 0, 0}});
        const double true_pos[3] = {1.0 * (t + 0.1), 0.0, 2.0};
        std::vector<PointType> cloud = make_scan(true_pos, rng, 20000);

        // --- p_imu->Process + downSizeFilterSurf (laserMapping.cpp:880, 904-905)
        size_t feats_down_size = 0;
        std::vector<float> feats_undistort(3 * cloud.size());
        if (p_imu.UndistortPcl(imu, t, t + 0.1, kf, g_flh, /*slot=*/0, cloud.data(), sizeof(PointType), offsetof(PointType, curvature),
                               cloud.size(), (float)filter_size_surf_min, feats_undistort.data(), &feats_down_size) != 0 ||
            flh_scan_activate(g_flh, 0) != 0) {
            std::printf("scan front end: %s\n", flh_last_error());
            return 1;
        }
        state_ikfom state_point = kf.get_x();
        const V3 lid = state_point.rot * state_point.offset_T_L_I;
        const double pos_lid[3] = {state_point.pos[0] + lid[0], state_point.pos[1] + lid[1], state_point.pos[2] + lid[2]};  // :890
        int64_t kdtree_delete_counter = 0;
        flh_fov_segment(g_flh, &lm, pos_lid, cube_len, DET_RANGE, nullptr, nullptr, &kdtree_delete_counter);  // :886

        if (!map_built) {  // :909-921: the first scan becomes the map
            std::vector<float> body(3 * feats_down_size), world(3 * feats_down_size);
            flh_fetch_scan(g_flh, body.data());
            for (size_t i = 0; i < feats_down_size; ++i) {
                V3 pb; pb[0] = body[3 * i]; pb[1] = body[3 * i + 1]; pb[2] = body[3 * i + 2];
                V3 pi = state_point.offset_R_L_I * pb;
                for (int d = 0; d < 3; ++d) pi[d] += state_point.offset_T_L_I[d];
                const V3 pw = state_point.rot * pi;
                for (int d = 0; d < 3; ++d) world[3 * i + d] = (float)(pw[d] + state_point.pos[d]);
            }
            if (flh_map_build(g_flh, world.data(), 12, feats_down_size) != 0) { std::printf("%s\n", flh_last_error()); return 1; }
            map_built = true;
            std::printf("scan %d: map initialised with %zu points\n", scan, feats_down_size);
        } else {
            double solve_H_time = 0;
            kf.update_iterated_dyn_share_modified(LASER_POINT_COV, solve_H_time);  // :960
            state_point = kf.get_x();
            double x26[FLH_NSTATE];
            state_point.to_flat(x26);
            uint32_t add_point_size = 0, no_down = 0;
            if (flh_map_incremental(g_flh, x26, filter_size_map_min, 1, 1, &add_point_size, &no_down) != 0) {  // :923
                std::printf("map_incremental: %s\n", flh_last_error());
                return 1;
            }
            std::printf("scan %d: %zu points, effct_feat_num %d, res_mean %.4f, pos %.3f %.3f %.3f (true x %.3f), map %zu (+%u/%u)\n", scan,
                        feats_down_size, hctx.effct_feat_num, hctx.res_mean_last, state_point.pos[0], state_point.pos[1], state_point.pos[2],
                        true_pos[0], flh_map_size(g_flh), add_point_size, no_down);
        }
        t += 0.1;
    }
    flh_destroy(g_flh);
    return 0;
}
