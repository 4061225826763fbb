#!/bin/bash
# C2 us per timestep of several engine builds on one box, one launch per timestep (LLPF_PERSIST=0): tools/ab/run_c2.sh libA.so ...
for rep in 1 2; do for lib in "$@"; do
  c2=$(LLPF_PERSIST=${PERSIST:-0} LLPF_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']))")
  echo "$lib rep$rep C2_us_per_step=$c2"
done; done
