set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=1 TESTS=1 BENCH=1 PROF=1 STEPS=20 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
cat gpurun_out/summary.log; tail -3 gpurun_out/pytest_gpu.log
grep -o '"value": [0-9.]*' gpurun_out/bench.log | head -1
grep -o '"stages": {.*"stages_note"' gpurun_out/bench.log | cut -c1-900
head -4 gpurun_out/prof/r_kernel_stats.csv | cut -c1-150
