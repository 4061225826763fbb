#!/bin/bash
# A/B timing of engine builds on one box: tools/ab/run_ab.sh libA.so libB.so ...  (each run twice, interleaved)
for rep in 1 2; do
for lib in "$@"; do
  c2=$(LLPF_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']))")
  bk=$(LLPF_LIB=$PWD/$lib python tools/bench_bank.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e' % d['particle_steps_per_s'])")
  echo "$lib rep$rep C2_us_per_step=$c2 bank=$bk"
done
done
