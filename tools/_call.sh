set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=0 TESTS=0 BENCH=1 PROF=1 STEPS=20 PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS;FETCH_SIZE;WRITE_SIZE" bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
python tools/pmc_summary.py > gpurun_out/pmc_summary.log 2>&1
timeout 200 python bench.py > gpurun_out/bench_default.log 2>&1
echo "=== burner"; python tools/cpu_burn.py 64 40 &
sleep 2
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_a.log 2>&1
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_b.log 2>&1
wait
bash tools/other_workloads.sh > gpurun_out/other.log 2>&1
for f in bench.log bench_default.log bench_burner64_a.log bench_burner64_b.log; do tail -1 gpurun_out/$f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host"])'; done
tail -6 gpurun_out/other.log
