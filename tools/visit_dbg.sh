#!/bin/bash
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
mkdir -p gpurun_out
for lib in old new; do
  if [ $lib = old ]; then export R3DGS_LIB=old; else unset R3DGS_LIB; fi
  timeout 300 python tools/dbg_units.py garden_like_2M_1600x1062 2>&1 | tail -9
done
