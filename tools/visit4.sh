#!/bin/bash
# colour-stream split / grid sweep at the metric shape and 2 M (tools/ab.sh), then the whole GPU suite with its report
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
CONFIGS="new new:R3DGS_COLOR_GRID=256 new:R3DGS_COLOR_GRID=1024 new:R3DGS_COLOR_SPLIT0=10,R3DGS_COLOR_SPLIT1=30 new:R3DGS_COLOR_SPLIT0=30,R3DGS_COLOR_SPLIT1=35 new:R3DGS_COLOR_SPLIT0=25,R3DGS_COLOR_SPLIT1=45 new:R3DGS_COLOR_SPLIT0=15,R3DGS_COLOR_SPLIT1=25 new:R3DGS_DEPTH_BUCKET_LOAD=96 new:R3DGS_DEPTH_BUCKET_LOAD=192" WLS="metric_500k_1600x1062 garden_like_2M_1600x1062" ROUNDS=2 bash tools/ab.sh > gpurun_out/ab_console.txt 2>&1
cp gpurun_out/ab.txt gpurun_out/sweep_colour_split.txt
( timeout 1500 python -m pytest tests -m gpu -q -s ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_gpu.log)"
cut -c1-260 gpurun_out/sweep_colour_split.txt
grep -E "FAILED|Error" gpurun_out/pytest_gpu.log | head
grep -E "oracle chain fed" gpurun_out/pytest_gpu.log | head -12
