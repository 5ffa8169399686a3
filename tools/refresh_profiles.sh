#!/bin/bash
# One GPU-box visit that regenerates everything under profiles/rNN_* (run through gpurun; results land in gpurun_out/,
# tools/collect_profiles.py copies them into profiles/).
set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
rm -rf gpurun_out/pmc* gpurun_out/prof gpurun_out/prof_extra
SMOKE=1 TESTS=1 BENCH=1 PROF=1 STEPS=20 T_TEST=600 PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS;FETCH_SIZE;WRITE_SIZE" bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
python tools/pmc_summary.py > gpurun_out/pmc_summary.log 2>&1
timeout 200 python bench.py > gpurun_out/bench_default.log 2>&1
python tools/cpu_burn.py 64 40 &
sleep 2
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_a.log 2>&1
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_b.log 2>&1
wait
bash tools/other_workloads.sh > gpurun_out/other.log 2>&1
for wl in garden_like_2M_1600x1062 train_like_6M_1920x1080; do
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$wl -o r -- python $ROOT/bench.py --workload $wl --steps 10 --warmup 3 --cameras 4 --no-cpu-baseline ) > gpurun_out/prof_$wl.log 2>&1
done
cat gpurun_out/summary.log; tail -3 gpurun_out/pytest_gpu.log | cut -c1-200
for f in bench.log bench_default.log bench_burner64_a.log bench_burner64_b.log; do tail -1 gpurun_out/$f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host"]["host_ms_per_step_min_med_max"], d["host"]["cgroup"]["throttled_periods_in_timed_region"])'; done
tail -6 gpurun_out/other.log | cut -c1-400
ls gpurun_out/prof_garden_like_2M_1600x1062 gpurun_out/prof_train_like_6M_1920x1080 2>&1 | head
