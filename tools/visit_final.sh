#!/bin/bash
# final check of the tree: smoke, the driver-form bench line, the whole GPU suite (log kept as profiles/rNN_gpu_tests.txt)
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/smoke.log | cut -c1-160)"
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench20 rc=$?"
( timeout 300 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench50 rc=$?"
( timeout 1500 python -m pytest tests -m gpu -q -s ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_gpu.log)"
for f in bench.log bench_default.log; do tail -1 gpurun_out/$f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["value_reference_mode"], d["value_sh_sparsity"], d["roofline"]["frac"], d["roofline"]["kernel_frac"], d["roofline"]["traffic_source"][:30], d["roofline"]["traffic_collected_on_this_build"], d["iter_roofline"]["frac_of_8TBps"], d["iter_roofline"]["frac_counter_traffic"], d["iter_roofline"]["frac_of_speed_of_light"], d["ambiguous_profile_kernels"])'; done
