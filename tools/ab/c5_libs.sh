#!/bin/bash
# C5 on one box: k_rbfull us (events around the launches) for the product library and every library given; two rounds
for rep in 1 2; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so "$@"; do
  LLPF_LIB=$PWD/$lib python bench.py --workload rbpf_full --no-cpu-baseline --no-other-configs --steps 2 --T 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep$rep', '%.3e' % d['value'], [round(v,1) if v else None for v in list(d['kernel_us'].values())[:3]], round(d['roofline']['whole_timestep']['us'],1))"
done; done
