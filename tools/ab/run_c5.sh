#!/bin/bash
# C5 (per-particle RBPF) particle-steps/s and k_rbfull us of several engine builds on one box: tools/ab/run_c5.sh libA.so ...
for rep in 1 2; do for lib in "$@"; do
  LLPF_LIB=$PWD/$lib python bench.py --workload rbpf_full --no-cpu-baseline --steps 2 --T 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep$rep', '%.3e' % d['value'], round(list(d['kernel_us'].values())[0],1))"
done; done
