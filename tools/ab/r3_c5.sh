#!/bin/bash
# C5 A/B on one box: RB parity tests with the product library, then k_rbfull us of every library given
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_rbfull.py tests/test_gpu_rbpf.py tests/test_independent_oracle.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so "$@"; do
  LLPF_LIB=$PWD/$lib python bench.py --workload rbpf_full --no-cpu-baseline --steps 2 --T 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep$rep', '%.3e' % d['value'], [round(v,1) if v else None for v in list(d['kernel_us'].values())[:3]], round(d['roofline']['whole_timestep']['us'],1))"
done; done
