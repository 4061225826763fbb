#!/bin/bash
# A/B of engine builds over the bench workloads on one box: tools/ab/run_all.sh libA.so libB.so ...
# (C2 us per timestep; bank, quad-tank, RBPF and RBPF-full in particle-steps/s; each twice, interleaved)
val() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e' % d['value'])"; }
for rep in 1 2; do for lib in "$@"; do
  export LLPF_LIB=$PWD/$lib
  c2=$(python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']))")
  bk=$(python tools/bench_bank.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3e' % d['particle_steps_per_s'])")
  qt=$(python bench.py --workload quadtank --steps 1 --warmup 0 --T 300 --no-cpu-baseline 2>/dev/null | val)
  rb=$(python bench.py --workload rbpf --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | val)
  c5=$(python bench.py --workload rbpf_full --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | val)
  ax=$(python bench.py --workload aux --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | val)
  echo "$lib rep$rep C2_us=$c2 bank=$bk quadtank=$qt rbpf=$rb c5=$c5 aux=$ax"
done; done
