#!/bin/bash
# A/B bench lines in ONE GPU visit: CONFIGS="tag[:ENV=V,ENV=V...] ..." (tag `new` = this tree's library, any other tag =
# libr3dgs_hip_<tag>.so via R3DGS_LIB), each run on every workload of $WLS, the whole list repeated $ROUNDS times (alternating:
# boxes drift within a visit).  Lines land in gpurun_out/ab.txt.
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
O=gpurun_out/ab.txt; : > $O
for round in $(seq 1 ${ROUNDS:-2}); do
  for wl in ${WLS:-metric_500k_1600x1062}; do
    for cfg in ${CONFIGS:-new}; do
      tag=${cfg%%:*}; envs=""; [ "$cfg" != "$tag" ] && envs=$(echo "${cfg#*:}" | tr ',' ' ')
      lib=${tag%%+*}
      if [ "$lib" = new ]; then libenv=""; else libenv="R3DGS_LIB=$lib"; fi
      line=$(env $libenv $envs timeout 300 python bench.py --workload $wl --steps ${BSTEPS:-20} --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s", d["ms_per_step"], "ms;", " / ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in d["stages"].items()), "; sparsity", d.get("value_sh_sparsity"), "; ref-mode", d.get("value_reference_mode"))' 2>&1 | tail -1)
      echo "$wl [$cfg] $line" >> $O
    done
  done
done
cat $O
