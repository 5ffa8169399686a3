#!/bin/bash
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sh_direction_derivatives or golden or repeated_backward or autograd_wrapper or empty_and_all" ) > gpurun_out/pytest_sel.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_sel.log)"
CONFIGS="new new:R3DGS_PREBWD_LEAN=0 occ3 occ5" WLS="metric_500k_1600x1062 garden_like_2M_1600x1062 train_like_6M_1920x1080" ROUNDS=2 bash tools/ab.sh
( timeout 600 python tools/dryrun_2rank.py ) > gpurun_out/dryrun.log 2>&1; echo "dryrun rc=$?"
grep -E "FAILED|Error" gpurun_out/pytest_sel.log | head
