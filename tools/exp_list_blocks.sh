#!/bin/bash
# One visit: the backward's eight unit lists as interleaved tiles (the build) against lists of 2x2 / 4x4 tile blocks (variants
# libr3dgs_hip_blk2/blk4.so built from a patched unit_order_kernel): bench lines alternating, then FETCH_SIZE / WRITE_SIZE of
# the backward blend kernel for each.  (Record of the experiment behind common.h kListBlock = 4; the variants were the tree of
# that time with unit_order_kernel's tile_of() patched, built as tools/build_variant.py does.  To repeat it on today's tree: build
# libr3dgs_hip_blk<N>.so with kListBlock = N in a copy of common.h; N = 1 is the every-eighth-tile assignment.)
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/exp_list_blocks.txt; : > $O
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s", d["ms_per_step"], "ms;", " / ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in d["stages"].items()))'; }
for rep in 1 2; do
for wl in metric_500k_1600x1062 clustered_500k_1600x1062 garden_like_2M_1600x1062; do
  if [ $rep = 2 ] && [ $wl = garden_like_2M_1600x1062 ]; then continue; fi
  for lib in new blk2 blk4; do
    if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
    echo "$wl [$lib] $(timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | line)" >> $O
  done
done
done
for lib in new blk2 blk4; do
  if [ $lib = new ]; then unset R3DGS_LIB; else export R3DGS_LIB=$lib; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmcx_${lib}_$ctr -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmcx_${lib}_$ctr.log 2>&1
    f=$(find gpurun_out/pmcx_${lib}_$ctr -name "*counter_collection.csv" | head -1)
    python - "$f" "$lib" "$ctr" >> $O <<'PY'
import csv, sys, collections
f, lib, ctr = sys.argv[1:4]
s = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == ctr: s[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in s.items():
    if "blend_bwd" in k or "blend_fwd" in k or "pair_reduce" in k: print(f"{lib} {ctr} {k}: mean {sum(v)/len(v)/1024:.1f} MB over {len(v)} launches")
PY
  done
done
unset R3DGS_LIB
cat $O
