#!/bin/bash
# quad-tank (C3) particle-steps/s and per-kernel us of several engine builds on one box: tools/ab/run_qt.sh libA.so ...
for rep in 1 2; do for lib in "$@"; do
  LLPF_LIB=$PWD/$lib python bench.py --workload quadtank --steps 2 --warmup 1 --T 500 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep$rep', '%.3e' % d['value'], {k.split('(')[0]: (round(v,2) if v else v) for k,v in d['kernel_us'].items()})"
done; done
