#!/bin/bash
# A/B builds without touching the product library: recompile ONE translation unit with extra defines and link it with the
# other, already built objects into lib_<name>.so at the repo root (bench.py / tests pick it up through LLPF_LIB).
#   tools/ab/build_variant.sh <name> <unit: k_rbfull|k_step|k_resprop|k_resprop_split|k_quantile|kernels|capi> [-DFOO=1 ...]
set -e
NAME=$1; UNIT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
C=$ROOT/lowlevelparticlefilters.jl_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-pass-failed"
cd $C
if [ $UNIT = k_rbfull ] || [ $UNIT = k_step ]; then FLAGS="$FLAGS -mllvm -disable-machine-licm"; fi
/opt/rocm/bin/hipcc $FLAGS "$@" -c $UNIT.hip -o /tmp/${UNIT}_$NAME.o
OBJS=""
for u in kernels k_step k_resprop k_resprop_split k_rbfull k_quantile capi; do if [ $u = $UNIT ]; then OBJS="$OBJS /tmp/${UNIT}_$NAME.o"; else OBJS="$OBJS $u.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/lib_$NAME.so $OBJS -lhiprtc
echo built $ROOT/lib_$NAME.so
