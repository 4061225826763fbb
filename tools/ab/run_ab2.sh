#!/bin/bash
# A/B timing of engine builds on one box, with and without the persistent launch:
#   tools/ab/run_ab2.sh libA.so libB.so ...      (each twice, interleaved; C2 us per timestep, bank particle-steps/s)
for rep in 1 2; do
for lib in "$@"; do
  for p in 1 0; do
  c2=$(LLPF_PERSIST=$p LLPF_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']))")
  echo "$lib rep$rep persist=$p C2_us_per_step=$c2"
  done
  bk=$(LLPF_LIB=$PWD/$lib python tools/bench_bank.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e' % d['particle_steps_per_s'])")
  qt=$(LLPF_LIB=$PWD/$lib python bench.py --workload quadtank --steps 1 --warmup 0 --T 300 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % (d['value']))")
  echo "$lib rep$rep bank=$bk quadtank=$qt"
done
done
