#!/bin/bash
# One visit: pair_reduce_kernel with the Gaussian id of every pair loaded up front (pr1), the slab rows loaded without
# waiting for the flags (pr2), both (pr3), against the build: kernel traces of bench.py, alternating.  (Record of an experiment
# that changed nothing -- profiles/r05_exp_pair_reduce_latency.txt; the variants were patched copies of preprocess_bwd.hip built
# as tools/build_variant.py does and are not in the tree.)
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/exp_pair_reduce.txt; : > $O
for wl in metric_500k_1600x1062 garden_like_2M_1600x1062; do
for lib in new pr1 pr2 pr3 new pr3; do
  if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
  rm -rf gpurun_out/prx
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prx -o r -- python $ROOT/bench.py --workload $wl --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline ) > gpurun_out/prx.log 2>&1
  v=$(grep '^{' gpurun_out/prx.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s; blend_bwd stage", d["stages"]["blend_bwd"]["avg_ms"], "preprocess_bwd", d["stages"]["preprocess_bwd"]["avg_ms"])')
  k=$(python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/prx/r_kernel_stats.csv")):
    if "pair_reduce" in r["Name"] or "preprocess_bwd_kernel" in r["Name"] or "blend_bwd_kernel" in r["Name"]:
        print(r["Name"].split("(")[0].replace("void ", "").replace("r3::", ""), "%.1f us;" % (float(r["AverageNs"]) / 1e3), end=" ")
PY
)
  echo "$wl [$lib] $v | $k" >> $O
done
done
unset R3DGS_LIB
cat $O
