set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
SMOKE=1 TESTS=1 BENCH=1 PROF=1 T_TEST=900 STEPS=20 PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES;SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS;FETCH_SIZE;WRITE_SIZE" bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
python tools/pmc_summary.py > gpurun_out/pmc_summary.log 2>&1
echo "=== dry run 2 ranks"; timeout 300 python tools/dryrun_2rank.py 2>&1 | tail -3 | cut -c1-1500
tail -30 gpurun_out/round.log | cut -c1-400
