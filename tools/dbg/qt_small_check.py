"""Quick parity check of the quad-tank run loop against the device-order oracle (one-launch and two-launch forms)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bench
from llpf_amd import _capi, _structs as S
import oracle_binding as ob
for N, T in ((20000, 12), (300000, 6)):
    model, U, Y, kind, thr, label = bench.build_workload("quadtank", N, T)
    cfg = S.make_config(model, N, kind, S.RESAMPLE_SYSTEMATIC, thr, 7, 0)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    rg = g.run(U, Y, 1.0, ll_steps=True); ro = o.run(U, Y, 1.0, ll_steps=True)
    print(N, T, "ll equal:", np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64)),
          "anc equal:", np.array_equal(g.ancestors(), o.ancestors()),
          "x equal:", np.array_equal(g.particles().view(np.uint64), o.particles().view(np.uint64)), g.last_run_stats() if hasattr(g, "last_run_stats") else "")
