#!/bin/bash
# the driver-form bench line ten times in one visit: how much a single 20-step run scatters
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/bench_spread.txt
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["value_reference_mode"], d["value_sh_sparsity"], d["host"]["gpu_event_ms_per_step"])' >> gpurun_out/bench_spread.txt
done
cat gpurun_out/bench_spread.txt
