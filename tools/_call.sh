set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
ROOT=$PWD
for wl in garden_like_2M_1600x1062 train_like_6M_1920x1080; do
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$wl -o r -- python $ROOT/bench.py --workload $wl --steps 6 --warmup 2 --cameras 2 --no-cpu-baseline ) > gpurun_out/prof_$wl.log 2>&1; echo "prof $wl rc=$?"
f=$(find gpurun_out/prof_$wl -name "*kernel_stats.csv" | head -1); head -24 "$f" | cut -d, -f1,2,4 | cut -c1-150
done
