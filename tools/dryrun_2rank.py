"""Dry run of bench.py's N>1 path on ONE GPU: two ranks share cuda:0 and exchange over gloo
(R3DGS_BENCH_SINGLE_DEVICE / R3DGS_BENCH_BACKEND, see bench.py).  Functional check only -- the number it prints is
not a scaling result (both ranks time-share one GPU and the exchange goes through host memory)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, R3DGS_BENCH_SINGLE_DEVICE="1", R3DGS_BENCH_BACKEND="gloo")
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
       "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
       "--no-cpu-baseline"]
out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
line = next((l for l in reversed(out.stdout.splitlines()) if l.startswith("{")), None)
if line:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "dryrun_2rank_line.json"), "w").write(line + "\n")
print(out.stdout[-3000:])
print(out.stderr[-1500:], file=sys.stderr)
sys.exit(out.returncode)
