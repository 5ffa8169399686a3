set -u
export PYTHONPATH=$PWD:$PWD/reduced-3dgs_amd TMPDIR=/tmp
for lib in "" rb2k rb4k; do
R3DGS_LIB=$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('500k','$lib', d['value'], {k:v['avg_ms'] for k,v in d['stages'].items()})"
R3DGS_LIB=$lib timeout 200 python bench.py --workload garden_like_2M_1600x1062 --steps 20 --warmup 5 --cameras 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2M','$lib', d['value'], {k:v['avg_ms'] for k,v in d['stages'].items()})"
done
R3DGS_LIB=rb2k timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or fuzz or repeated or elementwise" 2>&1 | tail -2
