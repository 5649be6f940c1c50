"""Builds libfastlio_hip.so (HIP kernels + C ABI + host IEKF) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfastlio_hip.so")
if os.environ.get("FLH_LIB"):  # developer tools only (tools/variant.py builds): another build of the same ABI; never rebuilt from here
    LIB = os.environ["FLH_LIB"]
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SOURCES = ["flh_kernels.hip", "flh_pass.hip", "flh_mapinc.hip", "flh_scanprep.hip", "flh_stage.hip", "flh_api.cpp", "flh_esekf.cpp"]
DEPS = SOURCES + ["flh_device.hpp", "flh_kernels.hpp", "flh_search_dev.hpp", "flh_fit_dev.hpp", "flh_mail_dev.hpp"]
HDRS = ["fastlio_hip.h", "fastlio_amd/esekfom.hpp", "fastlio_amd/mtk.hpp", "fastlio_amd/smallmat.hpp",
        "fastlio_amd/use-ikfom.hpp", "fastlio_amd/h_share_model.hpp", "fastlio_amd/local_map.hpp", "fastlio_amd/imu_processing.hpp"]

# -ffp-contract=off: the reference never fuses a*b+c (baseline x86-64 build); flags must match for
# bit-exact point_selected_surf.  -fhip-fp32-correctly-rounded-divide-sqrt is hipcc's default; stated.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-result", "-Werror=undefined-internal",
         "-Wl,--no-undefined",
         # host side (the 23x23 IEKF algebra between two launches): AVX2 without FMA contraction -- same results, wider
         "-Xarch_host", "-mavx2"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if os.environ.get("FLH_LIB"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in DEPS] + [os.path.join(INCLUDE, f) for f in HDRS] + [__file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc()] + FLAGS + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


REFALG = os.path.join(LIBDIR, "libfastlio_esekf_refalg.so")


def build_reference_algebra(force: bool = False) -> str:
    """The host filter alone (flh_esekf.cpp over include/fastlio_amd/esekfom.hpp) built with -DFASTLIO_AMD_REFERENCE_ALGEBRA -- the
    reference's own operation sequence in the information-form step (esekfom.hpp:1782-1802: two 23 x 23 inverses) instead of the
    product's 12 x 12 elimination -- as a second shared library whose flh_* calls resolve to libfastlio_hip.so.  g++, host only.
    tests/test_gpu_parity.py runs a GPU update through it at the tolerances the product's form had to loosen."""
    src = os.path.join(CSRC, "flh_esekf.cpp")
    deps = [src, LIB] + [os.path.join(INCLUDE, f) for f in HDRS]
    if not force and os.path.exists(REFALG) and all(os.path.getmtime(d) <= os.path.getmtime(REFALG) for d in deps if os.path.exists(d)):
        return REFALG
    cxx = shutil.which("g++") or hipcc()
    subprocess.check_call([cxx, "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mavx2",
                           "-DFASTLIO_AMD_REFERENCE_ALGEBRA", src, "-o", REFALG, "-L" + LIBDIR, "-lfastlio_hip", "-Wl,-Bsymbolic",
                           "-Wl,-rpath,$ORIGIN"])
    return REFALG


if __name__ == "__main__":
    print(build(force=True, verbose=True))
