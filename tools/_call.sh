set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd
export TMPDIR=/tmp
ROOT=$PWD
short() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["render_fps"], d["host"])'; }
echo "=== idle"; timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | short
echo "=== idle, graphs off"; R3DGS_GRAPH=0 timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | short
echo "=== idle, all-graph (no stage timer in the timed region)"; R3DGS_BENCH_PROFILE=off timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | short
echo "=== burner 64 (CFS throttled)"; python tools/cpu_burn.py 64 45 &
sleep 2
timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | short
timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | short
R3DGS_GRAPH=0 timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline | short
wait
echo "=== rocprof"
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o r -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-180
