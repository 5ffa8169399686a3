set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=1 TESTS=1 BENCH=1 PROF=0 STEPS=20 T_TEST=600 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
cat gpurun_out/summary.log; tail -3 gpurun_out/pytest_gpu.log | cut -c1-300
grep -o '"value": [0-9.]*' gpurun_out/bench.log | head -1
grep -o '"stages": {.*"stages_note"' gpurun_out/bench.log | cut -c1-700
WORKLOADS="garden_like_2M_1600x1062 train_like_6M_1920x1080" bash tools/other_workloads.sh > gpurun_out/other.log 2>&1; tail -2 gpurun_out/other.log | cut -c1-400
