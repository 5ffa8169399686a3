set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=0 TESTS=0 BENCH=1 PROF=0 STEPS=20 SWEEP="R3DGS_LIB=nx;R3DGS_LIB=;R3DGS_LIB=nx" bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
cat gpurun_out/summary.log
R3DGS_LIB=nx timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
