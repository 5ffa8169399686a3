#!/bin/bash
# One bench line per secondary workload (BASELINE.json configs[0..4] stand-ins) -> gpurun_out/other_workloads.jsonl
# (copy to profiles/rNN_other_workloads.jsonl).  Not the headline: bench.py's default workload is.
set -u
mkdir -p gpurun_out
: > gpurun_out/other_workloads.jsonl
for wl in ${WORKLOADS:-cfg0_10k_400 lego_like_300k_800 garden_like_2M_1600x1062 bicycle_like_5M_1600x1062 train_like_6M_1920x1080 clustered_500k_1600x1062 garden_clustered_2M}; do
  timeout 300 python bench.py --workload $wl --steps ${STEPS:-20} --warmup 5 --cameras 4 --no-cpu-baseline 2> gpurun_out/other_$wl.err | tail -1 >> gpurun_out/other_workloads.jsonl
  echo "$wl rc=$?"
done
python - <<'PY'
import json
for line in open("gpurun_out/other_workloads.jsonl"):
    try:
        d = json.loads(line)
    except Exception:
        continue
    print(d["config"]["workload"], d["value"], "it/s", d["ms_per_step"], "ms", "R", d["config"]["num_rendered_mean"], {k: v["avg_ms"] for k, v in d["stages"].items()}, "fps", d["render_fps"])
PY
