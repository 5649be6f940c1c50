"""CPU model of the pre-launched no-search pass's mailbox protocol (fast_lio_amd/csrc/flh_mail_dev.hpp; only a GPU runs the real one:
tests/test_gpu_parity.py)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mailbox_protocol_model(tmp_path):
    """The decisions of the forwarder wave, of every workgroup's wait and of the host's post -- go, abort, a launch that was passed
    over, a host that comes too late -- restated with threads and atomics (tests/cpp/mailbox_model.cpp)."""
    exe = tmp_path / "mailbox_model"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "cpp", "mailbox_model.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and "all cases as specified" in r.stdout.decode(), r.stdout.decode()


def test_product_library_exports_no_experiment():
    """Experiments land or are deleted: the product exports no flh_exp_* entry point."""
    from fast_lio_amd import _build

    if not os.path.exists(_build.LIB):
        import pytest

        pytest.skip("library not built")
    out = subprocess.run(["nm", "-D", "--defined-only", _build.LIB], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "flh_exp_" not in out


def test_filter_hints_for_every_schedule(tmp_path):
    """What the mirrored filter tells its measurement model before every pass (dyn_share_datastruct::next_pass -> flh_eval_expect_next)
    and at the end of an update (the finish hook), through the schedules an update can take (tests/cpp/hint_check.cpp): a no-search
    pass is announced unless nothing follows or a search is certain; "nothing follows" is never said BEFORE a pass (it releases a
    waiting kernel: round 5's first version said it there and released the kernel the pass was to use); an announced pass that does
    not come because the update ends is taken back."""
    exe = tmp_path / "hint_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "hint_check.cpp"),
                           "-o", str(exe)])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode == 0 and "hints as specified" in r.stdout.decode(), r.stdout.decode()
