#!/bin/bash
# the rocprofv3 parts of tools/refresh_profiles.sh alone: kernel traces and counter passes of `bench.py --main-only`
set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/pmc* gpurun_out/prof*
S=gpurun_out/summary_prof.log; : > $S
CL=clustered_500k_1600x1062; G2=garden_like_2M_1600x1062; T6=train_like_6M_1920x1080
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o r -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --main-only ) > gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $S
for wl in $G2 $T6 $CL garden_clustered_2M; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$wl -o r -- python $ROOT/bench.py --workload $wl --steps 10 --warmup 3 --cameras 4 --no-cpu-baseline --main-only ) > gpurun_out/prof_$wl.log 2>&1; echo "prof $wl rc=$?" >> $S
done
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" \
            "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pmc$i -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --main-only ) > gpurun_out/pmc$i.log 2>&1; echo "pmc$i rc=$?" >> $S
  i=$((i+1))
done
python tools/pmc_summary.py > gpurun_out/pmc_summary.log 2>&1
for pair in "2M:$G2" "6M:$T6" "clustered_500k:$CL"; do
  tag=${pair%%:*}; wl=${pair#*:}; i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE"; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $ROOT/gpurun_out/pmc${tag}_$i -o r -- python $ROOT/bench.py --workload $wl --steps 3 --warmup 1 --cameras 4 --no-cpu-baseline --main-only ) > gpurun_out/pmc${tag}_$i.log 2>&1; echo "pmc $tag $i rc=$?" >> $S
    i=$((i+1))
  done
  python tools/pmc_summary.py gpurun_out/pmc_summary_$tag.json pmc${tag}_ >> gpurun_out/pmc_summary.log 2>&1
done
( timeout 300 python tools/dryrun_2rank.py ) > gpurun_out/dryrun.log 2>&1; echo "dryrun rc=$?" >> $S
cat $S
head -20 gpurun_out/prof/r_kernel_stats.csv | cut -c1-150
