"""Host logic of flh_api.cpp without a device.  The granule pick-up: tests/cpp/collect_granules_check.cpp compiles the library's host source into a
program of its own, plays the GPU with a thread (granules in the order a one-launch searching pass publishes them -- last group
first, header last -- or ascending, as k_fit does) and checks that the host ends with the groups added in GROUP order, bit for bit."""
import os
import subprocess

from fast_lio_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _host_program(tmp_path, name):
    lib = _build.build()  # the launch wrappers the host source refers to come from the library
    exe = tmp_path / name
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC", "-O3")] + ["-O1"]
    subprocess.check_call([_build.hipcc()] + flags + ["-x", "hip", os.path.join(ROOT, "tests", "cpp", name + ".cpp"),
                           "-L", os.path.dirname(lib), "-lfastlio_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", str(exe)])
    return subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)


def test_collect_granules_any_order_of_arrival(tmp_path):
    r = _host_program(tmp_path, "collect_granules_check")
    assert r.returncode == 0 and "every order of arrival" in r.stdout.decode(), r.stdout.decode()


def test_staging_thread_hand_over(tmp_path):
    """flh_scan_stage_async -> the staging thread (which polls for its next job before it sleeps) -> wait_slot: no job is lost
    whatever the caller's rhythm, the thread stops when asked.  Without a device every job fails at hipSetDevice, which is all
    the hand-over needs (tests/cpp/stager_check.cpp)."""
    r = _host_program(tmp_path, "stager_check")
    assert r.returncode == 0 and "every job handed over and reported" in r.stdout.decode(), r.stdout.decode()


def test_run_scans_stages_two_scans_ahead_once_each_in_order(tmp_path):
    """flh_esekf_run_scans' staging schedule with the C ABI underneath replaced by recording fakes (tests/cpp/run_scans_check.cpp):
    two scans ahead with a ring of three or more slots (the library stages them on two lanes), one with two; every scan staged
    exactly once, in order, into slot index % ring; never past what a call may stage -- also across bench.py's two-call pattern
    (warm-up with FLH_RUN_STAGE_NEXT, measurement with FLH_RUN_FIRST_STAGED); activation only after staging."""
    exe = tmp_path / "run_scans_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "run_scans_check.cpp"),
                           "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and "every scan staged once" in r.stdout.decode(), r.stdout.decode()
