#!/bin/bash
# round 5: the fused kernel (C2) and the step kernel (C3) under the compiler's other instruction schedulers
O=gpurun_out/r05y; mkdir -p $O
R=$PWD
one() { echo "$2 $1 $(LLPF_LIB=$R/$1 timeout 300 python bench.py --workload $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e us/timestep %.2f' % (d['value'], d['roofline']['whole_timestep']['us']))")" >> $O/sched_ab2.txt; }
for rep in 1 2 3; do
  for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so lib_rp_ilp.so lib_rp_maxilp.so; do one $lib lg; done
  for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so lib_st_ilp.so; do one $lib quadtank; done
done
cat $O/sched_ab2.txt
