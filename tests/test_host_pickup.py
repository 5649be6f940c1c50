"""flh_api.cpp's granule pick-up without a device: tests/cpp/collect_granules_check.cpp compiles the library's host source into a
program of its own, plays the GPU with a thread (granules in the order a one-launch searching pass publishes them -- last group
first, header last -- or ascending, as k_fit does) and checks that the host ends with the groups added in GROUP order, bit for bit."""
import os
import subprocess

from fast_lio_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_collect_granules_any_order_of_arrival(tmp_path):
    lib = _build.build()  # the launch wrappers the host source refers to come from the library
    exe = tmp_path / "collect_granules_check"
    flags = [f for f in _build.FLAGS if f not in ("-shared", "-fPIC", "-O3")] + ["-O1"]
    subprocess.check_call([_build.hipcc()] + flags + ["-x", "hip", os.path.join(ROOT, "tests", "cpp", "collect_granules_check.cpp"),
                           "-L", os.path.dirname(lib), "-lfastlio_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and "every order of arrival" in r.stdout.decode(), r.stdout.decode()
