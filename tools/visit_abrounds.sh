#!/bin/bash
# the round-5 library against this round's, alternating, five workloads (the part of tools/refresh_profiles.sh that needs
# libr3dgs_hip_old.so)
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
: > gpurun_out/ab_rounds.txt
for wl in metric_500k_1600x1062 clustered_500k_1600x1062 garden_like_2M_1600x1062 train_like_6M_1920x1080 garden_clustered_2M; do
  for lib in old new old new; do
    if [ $lib = old ]; then export R3DGS_LIB=old; else unset R3DGS_LIB; fi
    line=$(timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s", d["ms_per_step"], "ms;", " / ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in d["stages"].items()))')
    echo "$wl [$lib] $line" >> gpurun_out/ab_rounds.txt
  done
done
unset R3DGS_LIB
cat gpurun_out/ab_rounds.txt
