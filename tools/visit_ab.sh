#!/bin/bash
# One GPU visit: smoke, the driver-form bench line, this tree's library against libr3dgs_hip_old.so (the previous round's
# tree built with R3DGS_BUILD_TAG=old) alternating on the workloads named in $WLS, a kernel trace, then the GPU tests.
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
S=gpurun_out/summary.log; : > $S
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/smoke.log | cut -c1-200)" >> $S
( timeout 300 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench20 rc=$?" >> $S
: > gpurun_out/ab_rounds.txt
for wl in ${WLS:-metric_500k_1600x1062}; do
  for lib in ${LIBS:-old new old new}; do
    if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
    line=$(timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s", d["ms_per_step"], "ms;", " / ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in d["stages"].items()), "; ref-mode", d.get("value_reference_mode"))')
    echo "$wl [$lib] $line" >> gpurun_out/ab_rounds.txt
  done
done
unset R3DGS_LIB
cat gpurun_out/ab_rounds.txt >> $S
if [ "${PROF:-1}" = "1" ]; then
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o r -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ) > gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $S
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_metric.csv
fi
if [ "${TESTS:-1}" = "1" ]; then
  ( timeout ${T_TEST:-1200} python -m pytest tests -m gpu -q -s ${PYTEST_ARGS:-} ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_gpu.log)" >> $S
fi
cat $S
tail -1 gpurun_out/bench.log | cut -c1-3000
[ -f gpurun_out/kernel_stats_metric.csv ] && head -24 gpurun_out/kernel_stats_metric.csv | cut -c1-160
grep -E "FAILED|Error|assert" gpurun_out/pytest_gpu.log | head -40
exit 0
