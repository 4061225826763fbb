#!/bin/bash
# A/B of the split-schedule output loop's prefetch depth (LLPF_RESPROP_PF, kernels/resprop.hpp) at particle counts beyond the Infinity
# Cache and on one GPU's share of C4: lib_pf1.so / lib_pf2.so are the same engine with k_resprop_split.o compiled at depth 1 / 2.
for rep in 1 2; do
for lib in pf1 pf2 default; do
  if [ $lib = default ]; then unset LLPF_LIB; else export LLPF_LIB=$PWD/lib_$lib.so; fi
  echo "== $lib rep $rep"
  python tools/bench_n.py --sizes 4000000,16000000,64000000 --passes 2 | grep -E '"particles"|us_per_timestep"|loglik'
  python tools/bench_bank.py | grep -E -o '"us_per_timestep": [0-9.]+|"value": [0-9.e+]+|"ll_sum[a-z_]*": [-0-9.e+]+' | head -4
done; done
