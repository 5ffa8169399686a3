set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=1 TESTS=1 BENCH=1 PROF=1 STEPS=20 T_TEST=600 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
cat gpurun_out/summary.log; tail -2 gpurun_out/pytest_gpu.log | cut -c1-200
grep -o '"value": [0-9.]*' gpurun_out/bench.log | head -1
grep -o '"stages": {.*"stages_note"' gpurun_out/bench.log | cut -c1-700
grep "emit_pairs" gpurun_out/prof/r_kernel_stats.csv | cut -c1-140
timeout 200 python bench.py --workload garden_like_2M_1600x1062 --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2M', d['value'], {k:v['avg_ms'] for k,v in d['stages'].items()})"
