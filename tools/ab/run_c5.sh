#!/bin/bash
# A/B timing of engine builds on the C5 workload: tools/ab/run_c5.sh libA.so libB.so ...  (each run twice, interleaved)
for rep in 1 2; do
for lib in "$@"; do
  LLPF_LIB=$PWD/$lib python bench.py --workload rbpf_full --steps 3 --warmup 1 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=[v for n,v in d['kernel_us'].items() if n.startswith('k_rbfull')][0]
print('$lib rep$rep value=%.3e us_per_timestep=%.1f k_rbfull_us=%.1f' % (d['value'], 1e3*d['ms_per_step']/d['config']['timesteps'], k))"
done
done
