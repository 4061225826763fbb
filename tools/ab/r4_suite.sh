set -u
OUT=$PWD/gpurun_out/r04a
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "suite rc=$?" >> $OUT/gpu_suite.log
tail -5 $OUT/gpu_suite.log
python bench.py --workload quadtank --steps 2 --warmup 1 --T 500 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('qt', '%.3e' % d['value'], d['ms_per_step'])"
