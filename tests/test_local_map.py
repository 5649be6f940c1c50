"""lasermap_fov_segment (src/laserMapping.cpp:230-280): the oracle's restatement against hand-worked answers, and the
product's host mirror (include/fastlio_amd/local_map.hpp, compiled here with g++) against the oracle, bit for bit,
along a random walk that keeps hitting the cube faces.  The GPU half (slabs actually deleted from the device map)
is tests/test_gpu_map.py::test_fov_segment_moves_cube_and_deletes_slabs."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <cstdio>
#include <cstring>
#include "fastlio_amd/local_map.hpp"
int main(int argc, char** argv) {
    fastlio_amd::LocalMap lm;
    lm.cube_len = atof(argv[1]);
    lm.DET_RANGE = (float)atof(argv[2]);
    double p[3];
    while (fread(p, sizeof(double), 3, stdin) == 3) {
        auto boxes = lm.lasermap_fov_segment(p);
        int nb = (int)boxes.size();
        fwrite(&nb, sizeof(int), 1, stdout);
        for (auto& b : boxes) { fwrite(b.vertex_min, sizeof(float), 3, stdout); fwrite(b.vertex_max, sizeof(float), 3, stdout); }
        fwrite(lm.LocalMap_Points.vertex_min, sizeof(float), 3, stdout);
        fwrite(lm.LocalMap_Points.vertex_max, sizeof(float), 3, stdout);
    }
    return 0;
}
"""


def test_first_call_centres_the_cube_and_deletes_nothing():
    lm = po.LocalMap()
    boxes = po.fov_segment(lm, [10.0, -20.0, 5.0], cube_len=1000.0, det_range=100.0)
    assert len(boxes) == 0 and lm.initialized == 1
    assert list(lm.vertex_min) == [-490.0, -520.0, -495.0] and list(lm.vertex_max) == [510.0, 480.0, 505.0]


def test_no_move_while_far_from_every_face():
    lm = po.LocalMap()
    po.fov_segment(lm, [0, 0, 0], 1000.0, 100.0)          # faces at +-500, trigger distance 1.5 * 100 = 150
    assert len(po.fov_segment(lm, [349.0, 0, 0], 1000.0, 100.0)) == 0
    assert list(lm.vertex_max) == [500.0, 500.0, 500.0]


def test_move_towards_the_high_face_drops_the_low_slab():
    lm = po.LocalMap()
    po.fov_segment(lm, [0, 0, 0], 1000.0, 100.0)
    boxes = po.fov_segment(lm, [350.0, 0, 0], 1000.0, 100.0)   # 150 from the +x face: exactly on the trigger
    # mov_dist = max((1000 - 300) * 0.45, 100 * 0.5) = 315
    np.testing.assert_array_equal(boxes, np.array([[-500, -500, -500, -185, 500, 500]], np.float32))
    assert list(lm.vertex_min) == [-185.0, -500.0, -500.0] and list(lm.vertex_max) == [815.0, 500.0, 500.0]


def test_low_face_wins_when_both_trigger_and_each_axis_is_independent():
    lm = po.LocalMap()
    po.fov_segment(lm, [0, 0, 0], 200.0, 300.0)            # the reference's defaults: every face is always "close"
    boxes = po.fov_segment(lm, [1.0, 2.0, 3.0], 200.0, 300.0)
    # mov_dist = max((200 - 900) * 0.45, 300 * 0.5) = 150; the low-face branch is tested first on every axis
    want = np.array([[-50, -100, -100, 100, 100, 100], [-100, -50, -100, 100, 100, 100], [-100, -100, -50, 100, 100, 100]], np.float32)
    np.testing.assert_array_equal(boxes, want)
    assert list(lm.vertex_min) == [-250.0, -250.0, -250.0] and list(lm.vertex_max) == [-50.0, -50.0, -50.0]


@pytest.mark.parametrize("cube_len,det_range", [(1000.0, 100.0), (200.0, 300.0), (600.0, 80.5)])
def test_host_mirror_matches_oracle_along_a_random_walk(tmp_path, cube_len, det_range):
    src = tmp_path / "h.cpp"
    exe = tmp_path / "h"
    src.write_text(HARNESS)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    rng = np.random.default_rng(5)
    pos = np.cumsum(rng.normal(0, 40.0, (400, 3)), axis=0) + rng.normal(0, 1e-3, (400, 3))
    out = subprocess.run([str(exe), repr(cube_len), repr(det_range)], input=pos.astype(np.float64).tobytes(), stdout=subprocess.PIPE, check=True).stdout
    lm = po.LocalMap()
    off = 0
    moved = 0
    for p in pos:
        want = po.fov_segment(lm, p, cube_len, det_range)
        (nb,) = struct.unpack_from("i", out, off)
        off += 4
        got = np.frombuffer(out, np.float32, 6 * nb, off).reshape(nb, 6)
        off += 24 * nb
        cube = np.frombuffer(out, np.float32, 6, off)
        off += 24
        assert nb == len(want)
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        np.testing.assert_array_equal(cube.view(np.uint32), np.r_[np.array(lm.vertex_min, np.float32), np.array(lm.vertex_max, np.float32)].view(np.uint32))
        moved += nb > 0
    assert off == len(out) and moved > 5
