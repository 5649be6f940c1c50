#!/bin/bash
# GPU box, round 4, call 3: the tests that failed in call 2 (full output), the phase stamps of k_pass, the bench with the host
# algebra overlapped, then the fault hunt.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call3; mkdir -p $O
cd $R
timeout 900 python -m pytest -q -m gpu --tb=long tests/test_gpu_parity.py::test_rccl_allreduce_path_single_rank tests/test_gpu_peers.py tests/test_gpu_zz_timing.py::test_one_launch_pass_equals_three_launch_pass tests/test_gpu_parity.py::test_peer_granules_one_and_two_handles > $O/failed_tests.txt 2>&1; tail -5 $O/failed_tests.txt
FLH_LIB=$R/fast_lio_amd/lib/libfastlio_hip_stamps.so timeout 300 python tools/pass_stamps.py > $O/pass_stamps.txt 2>&1; cat $O/pass_stamps.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/driver.err; echo "driver command rc=$?"; python tools/bench_line.py $O/bench_driver_cmd.json
timeout 300 python bench.py --steps 300 --warmup 30 --cpu-scans 0 --no-extra-legs > $O/bench_300.json 2> $O/bench_300.err; echo "300 steps rc=$?"; python tools/bench_line.py $O/bench_300.json
NA=${NA:-240} NB=${NB:-6} NC=${NC:-24} timeout 2400 bash tools/fault_hunt.sh 2>&1 | tail -12
exit 0
