#!/bin/bash
# round 5: k_resample_fx with its argument block fetched in one batch (lib_fxpin.so) against the product
O=gpurun_out/r05x; mkdir -p $O
R=$PWD
for rep in 1 2 3; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so lib_fxpin.so; do
  echo "$lib rep$rep $(LLPF_LIB=$R/$lib timeout 300 python bench.py --workload quadtank --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4e us/timestep %.2f kernels %s' % (d['value'], d['roofline']['whole_timestep']['us'], {k: round(v, 2) for k, v in d['roofline'].get('kernel_us', {}).items()} if isinstance(d['roofline'].get('kernel_us'), dict) else d['roofline'].get('avg_launch_us')))")" >> $O/fxpin_ab.txt
done; done
LLPF_LIB=$R/lib_fxpin.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "quadtank or c3" > $O/fxpin_tests.log 2>&1; echo "rc=$?" >> $O/fxpin_tests.log
cat $O/fxpin_ab.txt; tail -2 $O/fxpin_tests.log
