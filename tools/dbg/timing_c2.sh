#!/bin/bash
# (needs an engine built with `make -C lowlevelparticlefilters.jl_amd/csrc DEVTOOLS=1`: the stamps are not in production builds)
# per-phase wall-clock stamps of ONE fused launch (timestep $1, default 500) of the default bench workload
K=${1:-500}
mkdir -p gpurun_out
LLPF_DEBUG_TIMING=$K python bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python tools/dbg/timing_report.py gpurun_out/llpf_timing.txt
python - <<'P'
import numpy as np
a=np.loadtxt('gpurun_out/llpf_timing.txt', dtype=np.uint64).astype(np.int64)
t0=a[:,0].min(); rel=(a[:,:5]-t0)/100.0
# blocks grouped by start order (dispatch round on their CU is unknown; use start-time quartiles)
order=np.argsort(rel[:,0]); q=len(a)//4
for i in range(4):
    idx=order[i*q:(i+1)*q]; d=np.diff(rel[idx],axis=1)
    print("start quartile %d: start %.2f head %.2f counts %.2f loop %.2f tail %.2f end %.2f"%(i, np.median(rel[idx,0]), *np.median(d,axis=0), np.median(rel[idx,4])))
P
