// imu_processing.hpp -- the part of ImuProcess that runs per scan: UndistortPcl (reference: src/IMU_Processing.hpp:216-351).
//
// Forward half (:240-300): one esekf::predict per IMU sample, host work on the mirrored filter; records IMUpose
// (msg/Pose6D.msg) and leaves the filter at the scan-end state.  Backward half (:307-349): every LiDAR point is carried
// to the scan-end frame -- on the device (flh_scan_stage_undistorted), optionally followed by the voxel-grid
// down-sampling of src/laserMapping.cpp:904-905 and the staging of feats_down_body for the update.
// Not mirrored: IMU_init (:130-214, one-off), the MARSIM lidar type (:227-230, :310), ROS message types (an IMU sample
// is {t, acc, gyr} here).
#pragma once
#include <vector>

#include "../fastlio_hip.h"
#include "esekfom.hpp"
#include "use-ikfom.hpp"

namespace fastlio_amd {

struct ImuSample {  // the fields of sensor_msgs::Imu that UndistortPcl reads
    double t;       // header.stamp.toSec()
    double acc[3];  // linear_acceleration
    double gyr[3];  // angular_velocity
};

struct ImuProcess {
    typedef esekfom::esekf<state_ikfom, 12, input_ikfom> kf_t;
    static constexpr double G_m_s2 = 9.81;  // include/common_lib.h:22

    // members of the reference class that survive from scan to scan (:92-112)
    V3 mean_acc, cov_acc, cov_gyr, cov_bias_gyr, cov_bias_acc, angvel_last, acc_s_last;
    ImuSample last_imu_{};
    double last_lidar_end_time_ = 0.0;
    kf_t::processnoisecovariance Q = process_noise_cov();
    std::vector<flh_pose6d> IMUpose;

    ImuProcess() {
        mean_acc[0] = 0; mean_acc[1] = 0; mean_acc[2] = -1.0;  // :123
        for (int i = 0; i < 3; ++i) {
            cov_acc[i] = 0.1; cov_gyr[i] = 0.1; cov_bias_gyr[i] = 0.0001; cov_bias_acc[i] = 0.0001;  // :118-121
            angvel_last[i] = 0; acc_s_last[i] = 0;
        }
    }

    static flh_pose6d set_pose6d(double t, const V3& a, const V3& g, const V3& v, const V3& p, const M3& R) {  // common_lib.h:169-183
        flh_pose6d kp;
        kp.offset_time = t;
        for (int i = 0; i < 3; ++i) {
            kp.acc[i] = a[i]; kp.gyr[i] = g[i]; kp.vel[i] = v[i]; kp.pos[i] = p[i];
            for (int j = 0; j < 3; ++j) kp.rot[i * 3 + j] = R(i, j);
        }
        return kp;
    }

    // :217-300.  meas_imu = the IMU samples of this scan (the previous scan's last sample is prepended here, :220).
    void forward_propagate(const std::vector<ImuSample>& meas_imu, double pcl_beg_time, double pcl_end_time, kf_t& kf_state) {
        std::vector<ImuSample> v_imu;
        v_imu.reserve(meas_imu.size() + 1);
        v_imu.push_back(last_imu_);
        v_imu.insert(v_imu.end(), meas_imu.begin(), meas_imu.end());
        const double imu_end_time = v_imu.back().t;

        state_ikfom imu_state = kf_state.get_x();
        IMUpose.clear();
        IMUpose.push_back(set_pose6d(0.0, acc_s_last, angvel_last, imu_state.vel, imu_state.pos, imu_state.rot.toRotationMatrix()));

        V3 angvel_avr, acc_avr;
        double dt = 0;
        input_ikfom in;
        for (size_t k = 0; k + 1 < v_imu.size(); ++k) {
            const ImuSample& head = v_imu[k];
            const ImuSample& tail = v_imu[k + 1];
            if (tail.t < last_lidar_end_time_) continue;
            for (int i = 0; i < 3; ++i) {
                angvel_avr[i] = 0.5 * (head.gyr[i] + tail.gyr[i]);
                acc_avr[i] = 0.5 * (head.acc[i] + tail.acc[i]);
            }
            acc_avr = acc_avr * G_m_s2 / mean_acc.norm();  // :262
            if (head.t < last_lidar_end_time_) dt = tail.t - last_lidar_end_time_;
            else dt = tail.t - head.t;
            for (int i = 0; i < 3; ++i) { in.acc[i] = acc_avr[i]; in.gyro[i] = angvel_avr[i]; }
            for (int i = 0; i < 3; ++i) {  // :276-279
                Q(i, i) = cov_gyr[i]; Q(3 + i, 3 + i) = cov_acc[i]; Q(6 + i, 6 + i) = cov_bias_gyr[i]; Q(9 + i, 9 + i) = cov_bias_acc[i];
            }
            kf_state.predict(dt, Q, in);

            imu_state = kf_state.get_x();  // the pose at this IMU sample
            for (int i = 0; i < 3; ++i) angvel_last[i] = angvel_avr[i] - imu_state.bg[i];
            V3 unb;
            for (int i = 0; i < 3; ++i) unb[i] = acc_avr[i] - imu_state.ba[i];
            acc_s_last = imu_state.rot * unb;
            for (int i = 0; i < 3; ++i) acc_s_last[i] += imu_state.grav[i];
            const double offs_t = tail.t - pcl_beg_time;
            IMUpose.push_back(set_pose6d(offs_t, acc_s_last, angvel_last, imu_state.vel, imu_state.pos, imu_state.rot.toRotationMatrix()));
        }
        // pose at the frame end (:294-296)
        const double note = pcl_end_time > imu_end_time ? 1.0 : -1.0;
        dt = note * (pcl_end_time - imu_end_time);
        kf_state.predict(dt, Q, in);
        last_imu_ = meas_imu.back();
        last_lidar_end_time_ = pcl_end_time;
    }

    // The whole of UndistortPcl: forward half here, backward sweep (+ optional voxel grid, leaf_size > 0) on the device; the
    // result is staged in `slot` ready for flh_scan_activate.  pts: n records, stride_bytes apart, x y z as floats at offset 0
    // and the time offset in ms (PointType::curvature) at time_offset_bytes.
    int UndistortPcl(const std::vector<ImuSample>& meas_imu, double lidar_beg_time, double lidar_end_time, kf_t& kf_state,
                     flh_handle* h, int slot, const void* pts, size_t stride_bytes, size_t time_offset_bytes, size_t n,
                     float leaf_size, float* feats_undistort_xyz, size_t* feats_down_size) {
        forward_propagate(meas_imu, lidar_beg_time, lidar_end_time, kf_state);
        double x_end[FLH_NSTATE];
        kf_state.get_x().to_flat(x_end);
        return flh_scan_stage_undistorted(h, slot, pts, stride_bytes, time_offset_bytes, n, IMUpose.data(), (int)IMUpose.size(), x_end,
                                          leaf_size, feats_undistort_xyz, feats_down_size);
    }
};

}  // namespace fastlio_amd
