#!/bin/bash
# GPU box, round 4, call 11: the headline config's full-size test through the common helper (normvec and d2 bit-compared), and the
# driver's command once more after bench.py began to keep stdout for its one line.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_call11; mkdir -p $O
cd $R
timeout 400 python -m pytest -q -m gpu tests/test_gpu_z_fullsize.py::test_config2_full_update_against_oracle 2>&1 | tail -5 | tee $O/fullsize_config2.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/driver.err; echo "driver command rc=$? stdout lines: $(wc -l < $O/bench_driver_cmd.json)"; python tools/bench_line.py $O/bench_driver_cmd.json
exit 0
