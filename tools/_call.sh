set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=0 TESTS=0 BENCH=1 PROF=0 STEPS=20 SWEEP="R3DGS_LIB=noilp;R3DGS_LIB=iter;R3DGS_LIB=o2;R3DGS_LIB=" bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
cat gpurun_out/summary.log
