#!/bin/bash
# One GPU-box visit: smoke, -m gpu parity tests, bench, rocprofv3 kernel stats.  Everything is wrapped in
# its own `timeout` so a hang cannot eat the round's GPU budget.  Outputs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd
export TMPDIR=/tmp
S=gpurun_out/summary.log; : > $S
( timeout ${T_SMOKE:-300} python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> $S
( timeout ${T_TEST:-420} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $S
( timeout ${T_BENCH:-240} python bench.py --steps ${STEPS:-20} --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> $S
if [ "${PROF:-1}" = "1" ]; then
  ( cd /tmp && timeout ${T_PROF:-240} rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $S
  find gpurun_out/prof -name "*kernel_stats*" | head -3 >> $S
fi
cat $S; echo ---; tail -25 gpurun_out/smoke.log; echo ---; tail -40 gpurun_out/pytest_gpu.log; echo ---; tail -3 gpurun_out/bench.log; echo ---; tail -5 gpurun_out/prof.log 2>/dev/null
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
