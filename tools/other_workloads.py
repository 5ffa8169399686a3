"""bench.py on the other synthetic workloads of synth_scene.WORKLOADS (secondary results; the metric workload is the
default of bench.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for w in ("cfg0_10k_400", "lego_like_300k_800", "garden_like_2M_1600x1062"):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "30", "--warmup", "5",
                          "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    print(line, flush=True)
