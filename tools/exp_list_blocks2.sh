#!/bin/bash
# Follow-up of tools/exp_list_blocks.sh on the final tree: common.h kListBlock = 2 / 8 (variants built from a copy of csrc with that
# one constant changed: blend.hip and capi.hip recompiled) against the build's 4.
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/exp_list_blocks2.txt; : > $O
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s", d["ms_per_step"], "ms; blend_bwd stage", d["stages"]["blend_bwd"]["avg_ms"])'; }
for wl in metric_500k_1600x1062 clustered_500k_1600x1062; do
  for lib in new blk2 blk8; do
    if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
    echo "$wl [$lib] $(timeout 60 python bench.py --workload $wl --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | line)" >> $O
  done
done
for lib in blk8; do
  export R3DGS_LIB=$lib
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 60 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmcy_${lib}_$ctr -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmcy_${lib}_$ctr.log 2>&1
    f=$(find gpurun_out/pmcy_${lib}_$ctr -name "*counter_collection.csv" | head -1)
    python - "$f" "$lib" "$ctr" >> $O <<'PY'
import csv, sys, collections
f, lib, ctr = sys.argv[1:4]
s = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == ctr: s[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in s.items():
    if "blend_bwd" in k: print(f"{lib} {ctr} {k}: mean {sum(v)/len(v)/1024:.1f} MB over {len(v)} launches")
PY
  done
done
unset R3DGS_LIB
cat $O
