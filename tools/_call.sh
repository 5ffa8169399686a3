set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=1 TESTS=0 BENCH=1 PROF=0 STEPS=20 SWEEP="R3DGS_LIB=occ6;R3DGS_LIB=occ7;R3DGS_LIB=occ6;R3DGS_LIB=occ7" bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
cat gpurun_out/summary.log
grep -o '"stages": {.*"stages_note"' gpurun_out/bench.log | cut -c1-700
