#!/bin/bash
# One GPU-box visit that regenerates everything under profiles/r06_* (run through gpurun; results land in gpurun_out/,
# tools/collect_profiles.py copies them into profiles/ and regenerates profiles/README.md).
set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
rm -rf gpurun_out/pmc* gpurun_out/prof*
S=gpurun_out/summary.log; : > $S
CL=clustered_500k_1600x1062; G2=garden_like_2M_1600x1062; T6=train_like_6M_1920x1080
# the bench lines first, on a box that has done nothing yet (as the driver's bench visit); the three-minute test suite and the
# instruction-rate loops leave the chip warm and the memory-bound stages ~20 % slower for a while
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 gpurun_out/smoke.log | cut -c1-300)" >> $S
( timeout 240 python bench.py --steps 20 --warmup 5 ) > gpurun_out/bench.log 2>&1; echo "bench20 rc=$?" >> $S
( timeout 240 python bench.py ) > gpurun_out/bench_default.log 2>&1; echo "bench50 rc=$?" >> $S
( R3DGS_STRICT=0 timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_nonstrict.log 2>&1; echo "bench nonstrict rc=$?" >> $S
( R3DGS_TIGHT_RECT=0 timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_refrects.log 2>&1; echo "bench reference rects rc=$?" >> $S
( R3DGS_BWD_SEG=0 timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_noseg.log 2>&1; echo "bench whole lists rc=$?" >> $S
( R3DGS_F64_CHAIN=0 timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > gpurun_out/bench_f32chain.log 2>&1; echo "bench fp32 chain rc=$?" >> $S
python tools/cpu_burn.py 64 40 &
sleep 2
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_a.log 2>&1
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_burner64_b.log 2>&1
wait
# kernel traces: the metric workload and the others
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -o r -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --main-only ) > gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $S
for wl in $G2 $T6 $CL garden_clustered_2M; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$wl -o r -- python $ROOT/bench.py --workload $wl --steps 10 --warmup 3 --cameras 4 --no-cpu-baseline --main-only ) > gpurun_out/prof_$wl.log 2>&1; echo "prof $wl rc=$?" >> $S
done
# counters: separate --pmc passes, kernel-trace only (never mixed with sys / hip traces)
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
# this round's library against the previous round's (R3DGS_LIB=old: tools/build_variant.py-style build of the round-4 tree),
# same visit, alternating: the only comparison across rounds that a box-to-box spread of 1.4x leaves standing
if [ -f reduced-3dgs_amd/libr3dgs_hip_old.so ]; then
  : > gpurun_out/ab_rounds.txt
  for wl in metric_500k_1600x1062 $CL $G2 $T6 garden_clustered_2M; do
    for lib in old new; do
      if [ $lib = old ]; then export R3DGS_LIB=old; else unset R3DGS_LIB; fi
      line=$(timeout 300 python bench.py --workload $wl --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "it/s", d["ms_per_step"], "ms;", " / ".join("%s %.4f" % (k, v["avg_ms"]) for k, v in d["stages"].items()))')
      echo "$wl [$lib] $line" >> gpurun_out/ab_rounds.txt
    done
  done
  unset R3DGS_LIB
  cat gpurun_out/ab_rounds.txt >> $S
fi
( timeout 300 python tools/host_bound_bench.py 500 ) > gpurun_out/host_bound.log 2>&1; echo "host_bound rc=$? $(tail -1 gpurun_out/host_bound.log)" >> $S
bash tools/other_workloads.sh > gpurun_out/other.log 2>&1
if [ -f reduced-3dgs_amd/libr3dgs_hip_tl.so ]; then
  ( timeout 200 python tools/bwd_timeline.py ) > gpurun_out/bwd_timeline.txt 2>&1; echo "timeline rc=$?" >> $S
  ( R3_TL_WORKLOAD=$CL timeout 200 python tools/bwd_timeline.py ) > gpurun_out/bwd_timeline_clustered.txt 2>&1; echo "timeline clustered rc=$?" >> $S
  ( R3_TL_WORKLOAD=$CL R3DGS_BWD_SEG=0 timeout 200 python tools/bwd_timeline.py ) > gpurun_out/bwd_timeline_clustered_whole_lists.txt 2>&1; echo "timeline clustered, whole lists rc=$?" >> $S
fi
( timeout 120 tools/valu_rate ) > gpurun_out/valu_rate.txt 2>&1; echo "valu rc=$?" >> $S
if [ "${TESTS:-1}" = "1" ]; then
  ( timeout 900 python -m pytest tests -m gpu -q -s ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/pytest_gpu.log)" >> $S
fi
cat $S
for f in bench.log bench_default.log bench_nonstrict.log bench_refrects.log bench_noseg.log bench_f32chain.log bench_burner64_a.log bench_burner64_b.log; do echo "$f: $(tail -1 gpurun_out/$f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], {k: v["avg_ms"] for k, v in d["stages"].items()}, d["host"]["host_ms_per_step_min_med_max"], d["host"]["cgroup"]["throttled_periods_in_timed_region"])' 2>&1 | tail -1)"; done
tail -8 gpurun_out/other.log | cut -c1-400
