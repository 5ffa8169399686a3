#!/bin/bash
# One GPU-box visit: smoke, -m gpu parity tests, bench, rocprofv3 kernel stats (+ optional PMC pass and
# tuning-knob sweeps).  Every command is wrapped in its own `timeout` so a hang cannot eat the round's GPU
# budget.  Outputs land in gpurun_out/ (copy what should be judged into profiles/).
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd
export TMPDIR=/tmp
ROOT=$PWD
S=gpurun_out/summary.log; : > $S
if [ "${SMOKE:-1}" = "1" ]; then
  ( timeout ${T_SMOKE:-300} python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> $S
fi
if [ "${TESTS:-1}" = "1" ]; then
  ( timeout ${T_TEST:-420} python -m pytest ${PYTEST_TARGET:-tests} -m gpu -x -q ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $S
fi
if [ "${BENCH:-1}" = "1" ]; then
  ( timeout ${T_BENCH:-240} python bench.py --steps ${STEPS:-20} --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> $S
fi
if [ -n "${EXTRA:-}" ]; then   # one extra in-repo python script (e.g. tools/reduction_bench.py)
  ( timeout ${T_EXTRA:-240} python $EXTRA ) > gpurun_out/extra.log 2>&1; echo "extra rc=$?" >> $S
fi
# tuning sweeps: "NAME=VALUE NAME=VALUE;NAME=VALUE" -> one short bench per ';'-separated env set
if [ -n "${SWEEP:-}" ]; then
  IFS=';' read -ra SETS <<< "$SWEEP"
  for envset in "${SETS[@]}"; do
    tag=$(echo "$envset" | tr ' =' '__')
    ( env $envset timeout ${T_BENCH:-240} python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_$tag.log 2>&1
    echo "sweep [$envset] rc=$? $(tail -1 gpurun_out/bench_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], {k:v["avg_ms"] for k,v in d["stages"].items()}, d["render_fps"])' 2>/dev/null)" >> $S
  done
fi
if [ "${PROF:-1}" = "1" ]; then
  ( cd /tmp && timeout ${T_PROF:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o r -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $S
fi
if [ -n "${EXTRA_PROF:-}" ]; then   # kernel stats of an extra in-repo script
  ( cd /tmp && timeout ${T_PROF:-240} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_extra -o r -- python $ROOT/$EXTRA_PROF ) > gpurun_out/prof_extra.log 2>&1; echo "prof_extra rc=$?" >> $S
  f=$(find gpurun_out/prof_extra -name "*kernel_stats.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" gpurun_out/extra_kernel_stats.csv
fi
if [ "${LISTCTR:-0}" = "1" ]; then ( cd /tmp && timeout 60 rocprofv3 -L ) > gpurun_out/counters.txt 2>&1; fi
if [ -n "${PMC:-}" ]; then   # separate counter passes, kernel-trace only (never mixed with sys/hip traces)
  i=0
  IFS=';' read -ra PASSES <<< "$PMC"
  for ctrs in "${PASSES[@]}"; do
    ( cd /tmp && timeout ${T_PROF:-240} rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pmc$i -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmc$i.log 2>&1; echo "pmc$i [$ctrs] rc=$?" >> $S
    i=$((i+1))
  done
fi
cat $S; echo ---; tail -5 gpurun_out/smoke.log 2>/dev/null; echo ---; tail -30 gpurun_out/pytest_gpu.log 2>/dev/null; echo ---; tail -1 gpurun_out/bench.log 2>/dev/null; echo ---; tail -${EXTRA_TAIL:-30} gpurun_out/extra.log 2>/dev/null
f=$(find gpurun_out/prof -name "*kernel_stats.csv" 2>/dev/null | head -1); if [ -n "$f" ]; then echo "--- $f"; head -22 "$f" | cut -c1-200; fi
if [ -f gpurun_out/extra_kernel_stats.csv ] && [ -n "${EXTRA_PROF:-}" ]; then echo "--- extra kernel stats"; head -${EXTRA_STATS_LINES:-25} gpurun_out/extra_kernel_stats.csv | cut -c1-220; fi
exit 0
