#!/bin/bash
# A/B of engine builds on the quad-tank workload (C3)
for rep in 1 2; do for lib in "$@"; do
  qt=$(LLPF_LIB=$PWD/$lib python bench.py --workload quadtank --steps 2 --warmup 1 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % (d['ms_per_step']/2))")
  echo "$lib rep$rep quadtank_us_per_timestep=$qt"
done; done
