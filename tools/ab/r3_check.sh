#!/bin/bash
# round-3 quick check on the GPU box: RB / quad-tank parity tests, then the three bench lines that moved this round
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_rbfull.py tests/test_gpu_rbpf.py tests/test_independent_oracle.py tests/test_golden.py tests/test_user_models.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3/tests_rb.log
cat gpurun_out/r3/tests_rb.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quadtank or c3 or C3" 2>&1 | tail -5 | tee gpurun_out/r3/tests_qt.log
for w in rbpf_full quadtank; do
  python bench.py --workload $w --no-cpu-baseline --steps 2 > gpurun_out/r3/bench_$w.json 2> gpurun_out/r3/bench_$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/bench_$w.json").read().strip().splitlines()[-1])
print("$w", "%.3e" % d["value"], d.get("kernel_us"), d.get("roofline"))
PY
done
python bench.py --no-cpu-baseline --steps 5 > gpurun_out/r3/bench_c2.json 2> gpurun_out/r3/bench_c2.err
python -c "
import json
d=json.loads(open('gpurun_out/r3/bench_c2.json').read().strip().splitlines()[-1]); print('c2', d['ms_per_step'], d.get('roofline'))"
