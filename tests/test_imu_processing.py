"""UndistortPcl's forward half (src/IMU_Processing.hpp:217-300): the product's host mirror
(include/fastlio_amd/imu_processing.hpp, compiled here with g++ against the mirrored esekf::predict) against the oracle's
restatement on the same IMU stream, over three consecutive scans (the carried members -- last IMU sample, last lidar end
time, acc_s_last / angvel_last -- matter from the second scan on)."""
import os
import struct
import subprocess

import numpy as np

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fastlio_amd/imu_processing.hpp"
// the device half is not linked here
extern "C" int flh_scan_stage_undistorted(flh_handle*, int, const void*, size_t, size_t, size_t, const flh_pose6d*, int, const double*, float, float*, size_t*) { return -1; }
static void no_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&, void*) {}
int main() {
    typedef fastlio_amd::ImuProcess::kf_t kf_t;
    kf_t kf;
    double lim[23]; for (auto& l : lim) l = 0.001;
    kf.init_dyn_share(get_f, df_dx, df_dw, no_model, 3, lim, nullptr);
    double x0[26], P0[23 * 23];
    if (fread(x0, sizeof(double), 26, stdin) != 26 || fread(P0, sizeof(double), 529, stdin) != 529) return 1;
    state_ikfom s = kf.get_x(); s.from_flat(x0); kf.change_x(s);
    kf_t::cov P; for (int i = 0; i < 529; ++i) P.a[i] = P0[i]; kf.change_P(P);
    fastlio_amd::ImuProcess imu;
    int nscan; if (fread(&nscan, sizeof(int), 1, stdin) != 1) return 1;
    for (int k = 0; k < nscan; ++k) {
        int n; double tb, te;
        if (fread(&n, sizeof(int), 1, stdin) != 1 || fread(&tb, 8, 1, stdin) != 1 || fread(&te, 8, 1, stdin) != 1) return 1;
        std::vector<fastlio_amd::ImuSample> v(n);
        for (auto& m : v) { double r[7]; if (fread(r, 8, 7, stdin) != 7) return 1; m.t = r[0]; for (int i = 0; i < 3; ++i) { m.acc[i] = r[1 + i]; m.gyr[i] = r[4 + i]; } }
        if (k == 0) imu.last_imu_ = v[0];   // the node seeds last_imu_ during IMU_init (:190)
        imu.forward_propagate(v, tb, te, kf);
        int np = (int)imu.IMUpose.size();
        fwrite(&np, sizeof(int), 1, stdout);
        fwrite(imu.IMUpose.data(), sizeof(flh_pose6d), np, stdout);
        double xe[26]; kf.get_x().to_flat(xe);
        fwrite(xe, 8, 26, stdout);
        fwrite(kf.get_P().a, 8, 529, stdout);
    }
    return 0;
}
"""


def test_forward_half_matches_oracle_over_three_scans(tmp_path):
    src, exe = tmp_path / "h.cpp", tmp_path / "h"
    src.write_text(HARNESS)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rng = np.random.default_rng(2)
    x0 = np.zeros(26)
    x0[0:3] = (1.0, -2.0, 0.5)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    x0[3:7] = q
    x0[7:11] = (0, 0, 0, 1)
    x0[11:14] = (0.04, 0.02, -0.03)
    x0[14:17] = (2.0, -1.0, 0.2)
    x0[17:20] = (0.01, -0.02, 0.005)
    x0[20:23] = (0.05, 0.03, -0.04)
    x0[23:26] = (0.3, -0.2, -9.8)
    x0[23:26] *= 9.809 / np.linalg.norm(x0[23:26])
    P0 = po.init_P()
    scans = []
    t = 100.0
    for k in range(3):
        n = 20
        ts = t + 0.005 * np.arange(n) + (0.001 if k else 0.0)
        imu = np.c_[ts, rng.normal([0.3, -0.2, 9.7], 0.3, (n, 3)), rng.normal([0.2, -0.1, 0.4], 0.05, (n, 3))]
        tb, te = t + 0.002, t + 0.1 + (0.003 if k == 1 else -0.002)   # lidar end after / before the last IMU sample
        scans.append((imu, tb, te))
        t += 0.1
    blob = x0.tobytes() + np.ascontiguousarray(P0, np.float64).tobytes() + struct.pack("i", len(scans))
    for imu, tb, te in scans:
        blob += struct.pack("i", len(imu)) + struct.pack("d", tb) + struct.pack("d", te)
        blob += np.ascontiguousarray(imu, np.float64).tobytes()
    out = subprocess.run([str(exe)], input=blob, stdout=subprocess.PIPE, check=True).stdout
    st = po.ImuState()
    st.last_imu[:] = list(scans[0][0][0])
    x, P = x0.copy(), np.array(P0, np.float64)
    off = 0
    for k, (imu, tb, te) in enumerate(scans):
        poses, x, P = po.imu_forward(st, imu, tb, te, x, P)
        (np_,) = struct.unpack_from("i", out, off)
        off += 4
        got = np.frombuffer(out, np.float64, 22 * np_, off).reshape(np_, 22)
        off += 8 * 22 * np_
        xe = np.frombuffer(out, np.float64, 26, off)
        off += 8 * 26
        Pe = np.frombuffer(out, np.float64, 529, off).reshape(23, 23)
        off += 8 * 529
        assert np_ == len(poses) and np_ >= 19
        want = np.array([[p.offset_time, *p.acc, *p.gyr, *p.vel, *p.pos, *p.rot] for p in poses])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, err_msg=f"IMUpose, scan {k}")
        np.testing.assert_allclose(xe, x, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(Pe, P, rtol=1e-10, atol=1e-14 + 1e-10 * np.abs(P).max())
        assert np.all(np.diff(want[1:, 0]) > 0)   # IMUpose[0] = 0 by construction; [1] may precede it (scan 0 here)
    assert off == len(out)
