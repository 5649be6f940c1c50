#!/bin/bash
# GPU box, round 5, call 1 (prepared at the end of round 4, when the GPU budget was spent -- nothing here has run on hardware yet):
#   a. the product's GPU suite on the box (sanity), b. tools/mailbox_probe (what handing a state to a kernel that is already in the
#   queue costs, against the 6 us launch floor), c. tools/prelaunch_check.py on the FLH_EXP_PRELAUNCH build (parity of the
#   pre-launched no-search pass, the late-host fall-back, scans/s on and off), d. bench A/B of the two libraries, alternating.
# Build here first (cross-compile; the files travel with the snapshot):
#   python tools/variant.py --name prelaunch --define FLH_EXP_PRELAUNCH --build-only
#   python tools/variant.py --name red8 --define FLH_EXP_RED8 --build-only
#   python tools/variant.py --name prelaunch_red8 --define FLH_EXP_PRELAUNCH --define FLH_EXP_RED8 --build-only
#   python tools/variant.py --name skipwait --define FLH_EXP_SKIPWAIT --build-only
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/mailbox_probe.cpp -o tools/mailbox_probe
#   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/launch_probe.cpp -Lfast_lio_amd/lib -lfastlio_hip -Wl,-rpath,'$ORIGIN/../fast_lio_amd/lib' -o tools/launch_probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r05_call1; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s); el() { echo "[t+$(( $(date +%s) - t0 ))s] $*"; }
cd $R
V=$R/fast_lio_amd/lib/libfastlio_hip_prelaunch.so
timeout 900 python -m pytest -q -m gpu -x tests 2>&1 | tail -15 > $O/gpu_tests.txt; tail -4 $O/gpu_tests.txt | cut -c1-200
el "product GPU suite"
if [ -x tools/launch_probe ]; then HIP_FORCE_DEV_KERNARG=0 timeout 120 tools/launch_probe 2>&1 | head -4 > $O/launch_probe_kernarg0.txt; HIP_FORCE_DEV_KERNARG=1 timeout 120 tools/launch_probe > $O/launch_probe.txt 2>&1; echo "launch probe (first lines: kernel arguments in host memory; then in device memory)"; cat $O/launch_probe_kernarg0.txt $O/launch_probe.txt; fi
el "launch probe"
if [ -x tools/mailbox_probe ]; then timeout 120 tools/mailbox_probe > $O/mailbox_probe.txt 2>&1; echo "mailbox probe rc=$?"; cat $O/mailbox_probe.txt; fi
el "mailbox probe"
if [ -f $V ]; then
  FLH_LIB=$V timeout 300 python tools/prelaunch_check.py --M 200000 --N 20000 --cfg 1 --steps 200 > $O/prelaunch_check_small.txt 2>&1; echo "prelaunch check (small) rc=$?"; tail -14 $O/prelaunch_check_small.txt
  FLH_LIB=$V timeout 600 python tools/prelaunch_check.py > $O/prelaunch_check_config2.txt 2>&1; echo "prelaunch check (config 2) rc=$?"; tail -16 $O/prelaunch_check_config2.txt
  el "prelaunch check"
  # A/B on one box, alternating: the product, the pre-launched no-search pass, k_fit's reducer with eight loads per trip
  # (-DFLH_EXP_RED8), both; no barrier packet in front of a scan's first pass when its staging has finished (-DFLH_EXP_SKIPWAIT)
  for rep in 1 2; do
    for v in base prelaunch red8 prelaunch_red8 skipwait; do
      L=$R/fast_lio_amd/lib/libfastlio_hip.so; [ $v != base ] && L=$R/fast_lio_amd/lib/libfastlio_hip_$v.so
      [ -f $L ] || continue
      FLH_LIB=$L timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err; echo "$v $rep rc=$?"; python tools/bench_line.py $O/bench_${v}_$rep.json
    done
  done
  el "bench A/B"
  FLH_LIB=$V timeout 900 python -m pytest -q -m gpu -x tests/test_gpu_z_fullsize.py tests/test_gpu_parity.py 2>&1 | tail -6 > $O/gpu_tests_variant.txt; tail -3 $O/gpu_tests_variant.txt | cut -c1-200
  el "variant library under the parity tests"
fi
exit 0
