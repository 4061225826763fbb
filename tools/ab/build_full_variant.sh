#!/bin/bash
# A/B build of the WHOLE engine with extra defines (for switches that live in shared headers, e.g. -DLLPF_EXP_LDEXP=1): the sources are
# copied to a scratch directory, built there with the product Makefile, and the library lands at the repo root as lib_<name>.so
# (bench.py / tests pick it up through LLPF_LIB).      tools/ab/build_full_variant.sh <name> [-DFOO=1 ...]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/llpf_var_$NAME
rm -rf $W; mkdir -p $W/lowlevelparticlefilters.jl_amd $W/tools $W/include
cp -r $ROOT/lowlevelparticlefilters.jl_amd/csrc $W/lowlevelparticlefilters.jl_amd/csrc
cp $ROOT/tools/gen_jit_prelude.py $W/tools/
cp $ROOT/include/llpf.h $W/include/
rm -f $W/lowlevelparticlefilters.jl_amd/csrc/*.o $W/lowlevelparticlefilters.jl_amd/csrc/jit_prelude.inc
make -j8 -C $W/lowlevelparticlefilters.jl_amd/csrc EXTRA_DEFS="$*" > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/lowlevelparticlefilters.jl_amd/libllpf_hip.so $ROOT/lib_$NAME.so
echo built $ROOT/lib_$NAME.so
