#!/bin/bash
# same-box A/B of builds of the split-schedule fused kernel at particle counts beyond the Infinity Cache: tools/ab/bign_variants.sh lib1 lib2 ...
# (lib_<name>.so in the repo root; "default" = the product)
for rep in 1 2; do
for lib in "$@"; do
  if [ $lib = default ]; then unset LLPF_LIB; else export LLPF_LIB=$PWD/lib_$lib.so; fi
  echo "== $lib rep $rep: $(python tools/bench_n.py --sizes ${SIZES:-4000000,16000000} --passes 2 | grep -E 'us_per_timestep"|loglik' | tr -d ' \n')"
done; done
