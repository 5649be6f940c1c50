"""Known-answer tests for the oracle's restatement of UndistortPcl's per-point half (src/IMU_Processing.hpp:307-349)."""
import numpy as np

from oracle import pyoracle as po

I9 = np.eye(3).reshape(9)


def state(pos=(0, 0, 0), rot=(0, 0, 0, 1), offR=(0, 0, 0, 1), offT=(0, 0, 0)):
    x = np.zeros(26)
    x[0:3], x[3:7], x[7:11], x[11:14] = pos, rot, offR, offT
    x[23:26] = (0, 0, -9.81)
    return x


def rz(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def test_no_motion_leaves_points_alone():
    poses = po.make_poses([(0.0, (0,) * 3, (0,) * 3, (0,) * 3, (0,) * 3, I9), (0.05, (0,) * 3, (0,) * 3, (0,) * 3, (0,) * 3, I9),
                           (0.1, (0,) * 3, (0,) * 3, (0,) * 3, (0,) * 3, I9)])
    rng = np.random.default_rng(0)
    pts = np.c_[rng.uniform(-50, 50, (1000, 3)), rng.uniform(0, 100, 1000)].astype(np.float32)
    np.testing.assert_array_equal(po.undistort(poses, state(), pts), pts[:, :3])


def test_constant_yaw_rate_is_unwound():
    w, T = 0.8, 0.1      # rad/s, scan length
    ts = [0.0, 0.025, 0.05, 0.075, 0.1]
    rows = [(t, (0,) * 3, (0, 0, w), (0,) * 3, (0,) * 3, rz(w * t).reshape(9)) for t in ts]
    poses = po.make_poses(rows)
    q_end = (0, 0, np.sin(w * T / 2), np.cos(w * T / 2))
    rng = np.random.default_rng(1)
    tms = rng.uniform(0.5, 99.5, 2000)
    P = rng.uniform(-40, 40, (2000, 3))
    out = po.undistort(poses, state(rot=q_end), np.c_[P, tms].astype(np.float32))
    Pf = P.astype(np.float32).astype(np.float64)
    tf = tms.astype(np.float32).astype(np.float64) / 1000.0
    want = np.array([rz(w * T).T @ rz(w * t) @ p for p, t in zip(Pf, tf)])
    np.testing.assert_allclose(out, want, rtol=0, atol=2e-5)


def test_constant_velocity_with_lever_arm():
    v = np.array([3.0, -1.0, 0.5])
    ts = [0.0, 0.05, 0.1]
    rows = [(t, (0,) * 3, (0,) * 3, v, v * t, I9) for t in ts]
    poses = po.make_poses(rows)
    offT = np.array([0.1, -0.2, 0.3])
    x_end = state(pos=v * 0.1, offT=offT)
    pts = np.array([[10, 0, 0, 20.0], [0, 5, 1, 70.0], [1, 1, 1, 100.0]], np.float32)
    out = po.undistort(poses, x_end, pts)
    for (x, y, z, tm), o in zip(pts, out):
        t = tm / 1000.0
        want = np.array([x, y, z], np.float64) + v * t - v * 0.1      # identity rotations: lever arm cancels
        np.testing.assert_allclose(o, want, rtol=0, atol=1e-5)


def test_segment_selection_edges():
    # gyr differs per segment so the chosen segment is visible in the result
    rows = [(0.0, (0,) * 3, (0, 0, 0.0), (0,) * 3, (0,) * 3, I9), (0.05, (0,) * 3, (0, 0, 1.0), (0,) * 3, (0,) * 3, I9),
            (0.1, (0,) * 3, (0, 0, -2.0), (0,) * 3, (0,) * 3, I9)]
    poses = po.make_poses(rows)
    p = np.array([10.0, 0, 0])
    tms = np.array([0.0, -5.0, 50.0, 50.001, 100.0, 130.0], np.float32)
    out = po.undistort(poses, state(), np.c_[np.tile(p, (6, 1)), tms].astype(np.float32))
    # t <= offset_time[0]: untouched
    np.testing.assert_array_equal(out[0], p.astype(np.float32))
    np.testing.assert_array_equal(out[1], p.astype(np.float32))
    # t == offset_time[1] exactly still belongs to segment 0 (strict >), whose tail gyr is +1 rad/s over dt = 0.05
    np.testing.assert_allclose(out[2], rz(1.0 * 0.05) @ p, atol=2e-6)
    # just after: segment 1 (tail gyr -2 rad/s), dt tiny
    t3 = float(tms[3]) / 1000.0
    np.testing.assert_allclose(out[3], rz(-2.0 * (t3 - 0.05)) @ p, atol=2e-6)
    # later than the last IMU pose: the last segment extrapolates
    np.testing.assert_allclose(out[5], rz(-2.0 * (0.13 - 0.05)) @ p, atol=2e-6)
    np.testing.assert_allclose(out[4], rz(-2.0 * 0.05) @ p, atol=2e-6)


def test_first_imu_sample_older_than_the_first_point():
    # IMUpose = [0.0, -0.002, 0.003, ...]: points in (-0.002, 0.003] belong to segment 1, points <= -0.002 to none
    rows = [(0.0, (0,) * 3, (0, 0, 9.0), (0,) * 3, (0,) * 3, I9), (-0.002, (0,) * 3, (0, 0, 1.0), (0,) * 3, (0,) * 3, I9),
            (0.003, (0,) * 3, (0, 0, -2.0), (0,) * 3, (0,) * 3, I9), (0.008, (0,) * 3, (0, 0, 3.0), (0,) * 3, (0,) * 3, I9)]
    poses = po.make_poses(rows)
    p = np.array([10.0, 0, 0])
    tms = np.array([-3.0, -1.0, 0.0, 2.0, 5.0], np.float32)
    out = po.undistort(poses, state(), np.c_[np.tile(p, (5, 1)), tms].astype(np.float32))
    np.testing.assert_array_equal(out[0], p.astype(np.float32))                       # nobody claims t = -0.003
    for i in (1, 2, 3):                                                                # segment 1: tail gyr -2 rad/s
        np.testing.assert_allclose(out[i], rz(-2.0 * (tms[i] / 1000.0 + 0.002)) @ p, atol=2e-6)
    np.testing.assert_allclose(out[4], rz(3.0 * (0.005 - 0.003)) @ p, atol=2e-6)      # segment 2: tail gyr 3 rad/s


def test_the_earliest_point_is_carried_again_by_every_earlier_segment():
    # IMU_Processing.hpp:345: the sweep leaves its inner loop at begin() WITHOUT stepping past the first point, so the earliest
    # point -- when it is younger than IMUpose[1] -- is compensated once more per earlier segment, on its moved coordinates
    rows = [(0.0, (0,) * 3, (0, 0, 9.0), (0,) * 3, (0,) * 3, I9), (0.01, (0,) * 3, (0, 0, 1.0), (0,) * 3, (0,) * 3, I9),
            (0.02, (0,) * 3, (0, 0, -2.0), (0,) * 3, (0,) * 3, I9), (0.03, (0,) * 3, (0, 0, 3.0), (0,) * 3, (0,) * 3, I9)]
    poses = po.make_poses(rows)
    p = np.array([10.0, 0, 0])
    tms = np.array([25.0, 15.0, 35.0, 15.0], np.float32)   # the earliest time twice: the lower index is "the first point"
    pts = np.c_[np.tile(p, (4, 1)), tms].astype(np.float32)
    out = po.undistort(poses, state(), pts)
    once = po.undistort(poses, state(), pts, first_point=False)
    # segment 1 (tail gyr -2 rad/s, dt = 5 ms), then segment 0 (tail gyr +1 rad/s, dt = 15 ms) on the moved float coordinates
    s1 = (rz(-2.0 * 0.005) @ p).astype(np.float32)
    np.testing.assert_allclose(once[1], s1, atol=2e-6)
    np.testing.assert_allclose(out[1], rz(1.0 * 0.015) @ s1.astype(np.float64), atol=2e-6)
    assert np.abs(out[1] - once[1]).max() > 0.05
    np.testing.assert_array_equal(out[[0, 2, 3]], once[[0, 2, 3]])   # every other point (an equal time included) once
    np.testing.assert_allclose(out[3], s1, atol=2e-6)
