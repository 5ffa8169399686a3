#!/bin/bash
# Round-3 GPU visit driver: runs the ';'-separated steps named in $STEPS, each under its own timeout, everything into
# gpurun_out/.  Steps: valu tests[:k-expr] bench[:tag[:ENV=V,...]] prof pmc:<ctrs> extra:<script.py>
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd
export TMPDIR=/tmp
ROOT=$PWD
S=gpurun_out/summary.log; : > $S
IFS=';' read -ra LIST <<< "${STEPS:-tests;bench}"
for step in "${LIST[@]}"; do
  IFS=':' read -ra F <<< "$step"
  case "${F[0]}" in
    valu)
      ( timeout 120 tools/valu_rate ${F[1]:-4096} ) > gpurun_out/valu_rate.txt 2>&1; echo "valu rc=$?" >> $S ;;
    smoke)
      ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/smoke.log)" >> $S ;;
    tests)
      if [ -n "${F[1]:-}" ]; then K=(-k "${F[1]}"); else K=(); fi
      ( timeout ${T_TEST:-600} python -m pytest tests -m gpu ${T_X--x} -q -s "${K[@]}" ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_gpu.log)" >> $S ;;
    bench)
      tag=${F[1]:-default}; envs=$(echo "${F[2]:-}" | tr ',' ' '); extra=$(echo "${F[3]:-}" | tr ',' ' ')
      ( env $envs timeout ${T_BENCH:-240} python bench.py --steps ${BSTEPS:-20} --warmup 5 ${extra:---no-cpu-baseline} ) > gpurun_out/bench_$tag.log 2>&1
      echo "bench[$tag] rc=$? $(tail -1 gpurun_out/bench_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k:v["avg_ms"] for k,v in d["stages"].items()}, "render_fps", d["render_fps"], d["config"].get("passes_in_timed_region"), "host", d["host"]["host_ms_per_step_min_med_max"])' 2>&1 | tail -1)" >> $S ;;
    prof)
      tag=${F[1]:-bench}; envs=$(echo "${F[2]:-}" | tr ',' ' '); extra=$(echo "${F[3]:-}" | tr ',' ' ')
      ( cd /tmp && env $envs timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$tag -o r -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra ) > gpurun_out/prof_$tag.log 2>&1; echo "prof[$tag] rc=$?" >> $S
      f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$tag.csv ;;
    pmc)
      tag=${F[1]}; ctrs=$(echo "${F[2]}" | tr ',' ' ')
      ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pmc_$tag -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmc_$tag.log 2>&1; echo "pmc[$tag] rc=$?" >> $S ;;
    extra)
      tag=$(basename ${F[1]} .py)
      ( timeout ${T_EXTRA:-300} python ${F[1]} ${F[2]:-} ) > gpurun_out/extra_$tag.log 2>&1; echo "extra[$tag] rc=$? $(tail -2 gpurun_out/extra_$tag.log | tr '\n' ' ' | cut -c1-400)" >> $S ;;
  esac
done
cat $S
echo "--- pytest tail"; tail -15 gpurun_out/pytest_gpu.log 2>/dev/null
for f in gpurun_out/kernel_stats_*.csv; do [ -f "$f" ] && { echo "--- $f"; head -24 "$f" | cut -c1-180; }; done
exit 0
