#!/bin/bash
# What the unit order kernel's first pass and its eight register slots per thread cost (variants of blend.hip built from a
# patched copy: uo1 = no first pass -- valid whenever the shortest segments fit, as here; uo3 = also two slots per thread instead
# of eight): kernel traces of bench.py.
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/exp_unit_order.txt; : > $O
for lib in new uo1 uo3 new; do
  if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
  rm -rf gpurun_out/uox
  ( cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/uox -o r -- python $ROOT/bench.py --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline ) > gpurun_out/uox.log 2>&1
  v=$(grep '^{' gpurun_out/uox.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s; blend_bwd stage", d["stages"]["blend_bwd"]["avg_ms"])')
  k=$(python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/uox/r_kernel_stats.csv")):
    if "unit_order" in r["Name"] or "blend_bwd_kernel" in r["Name"]:
        print(r["Name"].split("(")[0].replace("void ", "").replace("r3::", ""), "%.2f us;" % (float(r["AverageNs"]) / 1e3), end=" ")
PY
)
  echo "[$lib] $v | $k" >> $O
done
unset R3DGS_LIB
cat $O
