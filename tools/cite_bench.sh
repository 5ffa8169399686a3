set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 240 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench20 rc=$?"
( timeout 240 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench50 rc=$?"
bash tools/other_workloads.sh > gpurun_out/other.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-300
