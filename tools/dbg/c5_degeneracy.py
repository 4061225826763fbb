"""How degenerate is the particle set of BASELINE config C5 (and C3) when it resamples?  Prints the number of resampling steps of a
run and the number of distinct ancestors of the last resampling: what sharing per-ancestor work could save."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench
from llpf_amd import _capi, _structs as S
for wl, N, T in (("rbpf_full", 200000, 300), ("quadtank", 1000000, 300)):
    model, U, Y, kind, thr, label = bench.build_workload(wl, N, T)
    pf = _capi.FilterHandle(S.make_config(model, N, kind, S.RESAMPLE_SYSTEMATIC, thr, 1000, 0))
    pf.reset()
    hist = []
    prev = 0
    for t in range(T):
        pf.run(U[t:t + 1], Y[t:t + 1], float(t), reset=False) if "reset" in pf.run.__code__.co_varnames else None
        break
    pf.reset()
    r = pf.run(U, Y, 1.0)
    a = pf.ancestors()
    print(wl, "N", N, "T", T, "resamples", pf.resample_count(), "distinct ancestors of the last resampling", len(np.unique(a)), "of", N)
