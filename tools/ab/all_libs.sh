#!/bin/bash
# C2 / C3 / C4-share / C5 on one box for the product library and every library given (two rounds): timestep us per configuration
for rep in 1 2; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so "$@"; do
  for w in lg quadtank bank rbpf_full; do
    if [ $w = lg ]; then args="--steps 5"; elif [ $w = rbpf_full ]; then args="--steps 2 --T 300"; else args="--steps 2"; fi
    r=$(LLPF_LIB=$PWD/$lib python bench.py --workload $w --no-cpu-baseline --no-other-configs $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e %.2f' % (d['value'], d['roofline']['whole_timestep']['us']))")
    echo "$lib rep$rep $w $r"
  done
done; done
