#!/bin/bash
# GPU visit: bench line (driver form), the reference-mode / reduction tests, the 2-rank dry run of bench.py's N > 1 path
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
S=gpurun_out/summary.log; : > $S
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench20 rc=$? $(tail -1 gpurun_out/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["value_reference_mode"], d["roofline"]["kernel_frac"], d["iter_roofline"]["speed_of_light_ms"], d["iter_roofline"]["frac_of_speed_of_light"])')" >> $S
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reduction_ops.py -m gpu -q -s -k "${K:-reference or reduce_shards}" ) > gpurun_out/pytest_sel.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_sel.log)" >> $S
( timeout 600 python tools/dryrun_2rank.py ) > gpurun_out/dryrun.log 2>&1; echo "dryrun rc=$?" >> $S
cat $S
grep -E "oracle chain fed|kappa|FAILED|Error" gpurun_out/pytest_sel.log | head -40
tail -3 gpurun_out/dryrun.log | cut -c1-2500
