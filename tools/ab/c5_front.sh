#!/bin/bash
# What the parts of a batch cost INSIDE k_rbfull (timing only, the variants do not compute the filter): lib_c5_nonoise.so (no generator),
# lib_c5_ldnoise.so (normals read from memory as if a launch before this one had written them), lib_c5_nodyn.so (no RK4), lib_c5_nofront.so
# (neither), lib_c5_nocorr.so (no measurement update: 700 of the recursion's 2600 instructions) — built with
#   tools/ab/build_variant.sh <name> k_rbfull -DLLPF_RBF_ABL_NOISE=1|2 / -DLLPF_RBF_ABL_DYN=1 / -DLLPF_RBF_ABL_CORR=1
for rep in 1 2; do for lib in lowlevelparticlefilters.jl_amd/libllpf_hip.so lib_c5_nonoise.so lib_c5_ldnoise.so lib_c5_nodyn.so lib_c5_nofront.so lib_c5_nocorr.so; do
  [ -f $lib ] || continue
  LLPF_LIB=$PWD/$lib python bench.py --workload rbpf_full --no-cpu-baseline --no-other-configs --steps 2 --T 300 2>/tmp/c5_err.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep$rep', '%.3e' % d['value'], [round(v,1) if v else None for v in list(d['kernel_us'].values())[:3]], round(d['roofline']['whole_timestep']['us'],1))
except Exception as e:
    print('$lib rep$rep failed:', open('/tmp/c5_err.txt').read()[-300:].replace(chr(10),' | '))"
done; done
