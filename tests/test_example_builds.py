"""examples/mapping_loop.cpp -- the reference's main-loop body written against the mirrored headers and the C ABI -- must
compile with a plain host compiler, link against libfastlio_hip.so, and, on a machine without a HIP device, stop at
flh_create with the library's own message (there is no CPU fallback to fall into)."""
import os
import subprocess

import pytest

from fast_lio_amd import _build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mapping_loop_example_links_and_fails_loudly_without_a_device(tmp_path):
    lib = _build.build()
    exe = tmp_path / "mapping_loop"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "mapping_loop.cpp"),
                           "-L", os.path.dirname(lib), "-lfastlio_hip", "-Wl,-rpath," + os.path.dirname(lib), "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = r.stdout.decode()
    if capi.device_available():
        assert r.returncode == 0 and "map initialised" in out and "effct_feat_num" in out, out
    else:
        assert r.returncode == 2 and "no HIP device visible" in out, out
