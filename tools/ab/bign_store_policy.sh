#!/bin/bash
# Store policy and prefetch depth of the split-schedule output loop (kernels/resprop.hpp: LLPF_RESPROP_ST 1 = write-through / 0 = plain /
# 2 = nontemporal, LLPF_RESPROP_PF = loop rounds whose sources are requested together) on one box: single filters beyond the Infinity
# Cache and one GPU's share of C4.  Libraries: tools/ab/build_variant.sh st_<policy>_pf<depth> k_resprop_split -DLLPF_RESPROP_ST=.. -DLLPF_RESPROP_PF=..
# ("default" = the product = plain stores, depth 2)
for rep in 1 2; do
for lib in st_wt_pf1 st_plain_pf1 st_nt_pf1 default st_plain_pf4; do
  if [ $lib = default ]; then unset LLPF_LIB; else [ -f lib_$lib.so ] || continue; export LLPF_LIB=$PWD/lib_$lib.so; fi
  echo "== $lib rep $rep: $(python tools/bench_n.py --sizes 4000000,16000000,64000000 --passes 2 | grep -E 'us_per_timestep"|loglik' | tr -d ' \n')"
  echo "   bank 128 x 1e5: $(python tools/bench_bank.py | grep -E -o '"us_per_timestep": [0-9.]+' | head -1)"
done; done
