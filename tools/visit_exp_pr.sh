#!/bin/bash
# where pair_reduce's time goes: variants without its run-sum stores / without its slab-row loads (timing only)
set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
O=gpurun_out/exp_pair_reduce.txt; : > $O
for wl in metric_500k_1600x1062 garden_like_2M_1600x1062 train_like_6M_1920x1080; do
  for lib in ${LIBS:-new acc16 both16 new}; do
    if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
    rm -rf gpurun_out/prx
    ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prx -o r -- python $ROOT/bench.py --workload $wl --steps 10 --warmup 3 --cameras 4 --no-cpu-baseline --main-only ) > gpurun_out/prx.log 2>&1
    k=$(python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/prx/r_kernel_stats.csv")):
    if "pair_reduce" in r["Name"] or "preprocess_bwd_kernel" in r["Name"] or "blend_bwd_kernel" in r["Name"]:
        print(r["Name"].split("(")[0].replace("void ", "").replace("r3::", ""), "%.1f us;" % (float(r["AverageNs"]) / 1e3), end=" ")
PY
)
    echo "$wl [$lib] $k" >> $O
  done
done
unset R3DGS_LIB
cat $O
