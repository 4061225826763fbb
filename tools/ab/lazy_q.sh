#!/bin/bash
# Split schedule with the quanta formed in the fused kernel (the product) against the stored form (LLPF_LAZY_Q=0), same build, one box:
# C4 share and a single filter of 1.6e7 particles at thresholds 0.1 and 1.0, us per timestep
for rep in 1 2; do for lazy in 1 0; do
  export LLPF_LAZY_Q=$lazy
  echo "== lazy_q=$lazy rep $rep"
  echo "   bank thr 1.0: $(python tools/bench_bank.py --thr 1.0 | grep -E -o '"us_per_timestep": [0-9.]+|"ll_sum[a-z_]*": [-0-9.e+]+' | head -2 | tr '\n' ' ')"
  echo "   bank thr 0.1: $(python tools/bench_bank.py | grep -E -o '"us_per_timestep": [0-9.]+|"ll_sum[a-z_]*": [-0-9.e+]+' | head -2 | tr '\n' ' ')"
  for thr in 0.1 1.0; do
  echo "   single 1.6e7 thr $thr: $(python bench.py --particles 16000000 --T 100 --threshold $thr --no-cpu-baseline --no-other-configs --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['whole_timestep']['us'],1), d['config'].get('resamples_per_pass'), d['loglik'], {k[:6]: round(v,1) for k,v in d['kernel_us'].items() if v})")"
  done
done; done
