#!/bin/bash
# (needs tools/experiments/one_launch_large/one_launch.patch applied: the switch LLPF_ONE_LAUNCH exists only there)
# Working sets beyond the Infinity Cache, thresholds below 1: ONE launch per timestep (the fused kernel forms the exp-sums of the weights it
# produces) against k_norm + fused kernel (LLPF_ONE_LAUNCH=0, the product), same build, one box: us per timestep
for rep in 1 2; do for ol in 1 0; do
  export LLPF_ONE_LAUNCH=$ol
  echo "== one_launch=$ol rep $rep"
  echo "   bank thr 0.1: $(python tools/bench_bank.py | grep -E -o '"us_per_timestep": [0-9.]+' | head -1)"
  echo "   bank thr 0.5: $(python tools/bench_bank.py --thr 0.5 | grep -E -o '"us_per_timestep": [0-9.]+' | head -1)"
  echo "   single thr 0.1: $(python tools/bench_n.py --threshold 0.1 --sizes 2000000,4000000,16000000,64000000 --passes 2 | grep -E 'us_per_timestep"|loglik' | tr -d ' \n')"
done; done
