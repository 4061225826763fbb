#!/bin/bash
# RBPF with shared covariance (bench.py --workload rbpf: the reference's own RBPF benchmark system) on one box for the product
# library and every library given, two rounds: particle-steps/s, us per timestep
for rep in 1 2; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so "$@"; do
  r=$(LLPF_LIB=$PWD/$lib python bench.py --workload rbpf --no-cpu-baseline --no-other-configs --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3e %.2f %s' % (d['value'], d['roofline']['whole_timestep']['us'], d['loglik']))")
  echo "$lib rep$rep rbpf $r"
done; done
