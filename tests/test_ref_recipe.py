"""oracle/ref: the recipe that pins the oracle on the reference's own code where Eigen (and Boost) exist.  Here they do
not (parity unpinned): the build step must explain that and succeed; the comparison runs only if the binaries
are there (a box with Eigen), and then requires the default summation order to match the real Eigen bit for bit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def test_build_ref_explains_itself_and_succeeds():
    r = subprocess.run(["bash", os.path.join(ROOT, "oracle", "ref", "build_ref.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[oracle/ref]" in r.stdout


def test_make_inputs_runs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref", "make_inputs.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.getsize(os.path.join(REF, "planes_in.bin")) == 4 + 6000 * 60
    assert os.path.exists(os.path.join(REF, "iekf_in.bin"))


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_esti_plane")), reason="no Eigen on this box: oracle/_ref not built")
def test_reference_esti_plane_matches_the_default_order():
    subprocess.check_call([os.path.join(REF, "ref_esti_plane"), os.path.join(REF, "planes_in.bin"), os.path.join(REF, "planes_ref.bin")])
    if os.path.exists(os.path.join(REF, "ref_iekf")):
        subprocess.check_call([os.path.join(REF, "ref_iekf"), os.path.join(REF, "iekf_in.bin"), os.path.join(REF, "iekf_ref.bin")])
    subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "ref", "compare.py")])
