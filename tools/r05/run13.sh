#!/bin/bash
# round 5: k_rbfull under other instruction schedulers of the compiler (-mllvm -amdgpu-sched-strategy=...): parity, then C5 alternating
O=gpurun_out/r05y; mkdir -p $O
R=$PWD
LLPF_LIB=$R/lib_ilp.so timeout 400 python -m pytest tests/test_gpu_rbfull.py -x -q > $O/ilp_tests.log 2>&1; echo "rc=$?" >> $O/ilp_tests.log; tail -2 $O/ilp_tests.log
for rep in 1 2 3; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so lib_ilp.so lib_maxocc.so; do
  echo "$lib rep$rep $(LLPF_LIB=$R/$lib timeout 300 python bench.py --workload rbpf_full --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e us/timestep %.2f kernels %s' % (d['value'], d['roofline']['whole_timestep']['us'], {k: round(v, 2) for k, v in d['roofline'].get('kernel_us', {}).items()} if isinstance(d['roofline'].get('kernel_us'), dict) else d['roofline'].get('avg_launch_us')))")" >> $O/sched_ab.txt
done; done
cat $O/sched_ab.txt
