set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
timeout 240 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1
timeout 240 python bench.py > gpurun_out/bench_default.log 2>&1
python tools/cpu_burn.py 64 40 &
sleep 2
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_a.log 2>&1
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_b.log 2>&1
wait
for f in bench.log bench_default.log bench_burner64_a.log bench_burner64_b.log; do tail -1 gpurun_out/$f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["host"]["host_ms_per_step_min_med_max"], d["host"]["cgroup"]["throttled_periods_in_timed_region"], d["render_fps"], d["config"]["clock_warmup_steps"])'; done
