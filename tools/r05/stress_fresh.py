"""round 5: first run of a FRESH handle with history outputs at N = 1e6 (the configuration whose final particles came out wrong once in
fifteen such runs): what exactly differs.  usage: stress_fresh.py [handles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import fuzz_parity as FP
FP.BIG = True
S, _capi = FP.S, FP._capi
from stress_big import case_of
c = case_of(15, 20)
cfg = S.make_config(c["model"], c["N"], c["kind"], c["strat"], c["thr"], c["seed"], 0)
print("N=%d thr=%g strat=%d T=%d nx=%d ny=%d" % (c["N"], c["thr"], c["strat"], c["T"], c["model"].nx, c["model"].ny), flush=True)
ref = None
nbad = 0
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    h = _capi.FilterHandle(cfg)
    h.reset()
    r = h.run(c["U"], c["Y"], c["t0"], ll_steps=True, history=True)
    x = np.ascontiguousarray(h.particles())
    x2 = np.ascontiguousarray(h.particles())
    if ref is None:
        ref = x.copy(); continue
    if not np.array_equal(x.view(np.uint64), ref.view(np.uint64)):
        nbad += 1
        ne = x.view(np.uint64) != ref.view(np.uint64)
        idx = np.argwhere(ne)
        cols = np.unique(idx[:, -1]) if x.ndim == 2 else idx[:, 0]
        print("handle %d: %d entries differ (shape %s); second read equal to first: %s, second read equal to ref: %s" % (
            t, len(idx), x.shape, np.array_equal(x.view(np.uint64), x2.view(np.uint64)), np.array_equal(x2.view(np.uint64), ref.view(np.uint64))))
        print("   first %s last %s; particle indices min %d max %d, distinct %d; values engine %s ref %s" % (
            idx[0].tolist(), idx[-1].tolist(), cols.min(), cols.max(), len(cols), x[tuple(idx[0])], ref[tuple(idx[0])]))
        pi = np.unique(idx[:, -1] if x.shape[0] < x.shape[-1] else idx[:, 0])
        runs = np.split(pi, np.flatnonzero(np.diff(pi) != 1) + 1)
        print("   contiguous index runs: %s" % [(int(a[0]), int(a[-1])) for a in runs[:12]], len(runs))
print("bad: %d" % nbad)
