"""round 5: the history run at N = 1e6 whose FINAL particles came out different once in 600 runs — many more runs, and a close look at a
deviation when it happens (which particles, is the device state wrong or only the copy, what the wrong values are).
usage: stress_hist.py [runs] [cycle]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import fuzz_parity as FP
FP.BIG = True
from stress_big import case_of
S, _capi = FP.S, FP._capi
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
CYCLE = int(sys.argv[2]) if len(sys.argv) > 2 else 8          # (a handle draws new noise after every reset: only equal positions compare)
HIST = (sys.argv[3] != "nohist") if len(sys.argv) > 3 else True
c = case_of(15, 20)
cfg = S.make_config(c["model"], c["N"], c["kind"], c["strat"], c["thr"], c["seed"], 0)
print("N=%d thr=%g strat=%d T=%d nx=%d ny=%d, %d runs, a fresh handle every %d" % (c["N"], c["thr"], c["strat"], c["T"], c["model"].nx, c["model"].ny, RUNS, CYCLE), flush=True)
ref = {}
h = None
nbad = 0
for t in range(RUNS):
    pos = t % CYCLE
    if pos == 0:
        h = _capi.FilterHandle(cfg)
    h.reset()
    r = h.run(c["U"], c["Y"], c["t0"], ll_steps=True, history=HIST)
    x = np.ascontiguousarray(h.particles())
    if pos not in ref:
        ref[pos] = (x.copy(), r["x"][..., -1].copy() if False else None, r["ll_steps"].copy())
        continue
    rx = ref[pos][0]
    if not np.array_equal(x.view(np.uint64), rx.view(np.uint64)):
        nbad += 1
        x2 = np.ascontiguousarray(h.particles())
        j = np.ascontiguousarray(h.ancestors())
        ne = (x.view(np.uint64) != rx.view(np.uint64))
        idx = np.argwhere(ne)
        pax = 0 if x.shape[0] > x.shape[-1] else x.ndim - 1          # the particle axis
        pi = np.unique(idx[:, pax])
        runs = np.split(pi, np.flatnonzero(np.diff(pi) != 1) + 1)
        print("run %d (position %d): ll equal %s; %d entries of %s differ; read again: same as first read %s, same as reference %s" % (
            t, pos, np.array_equal(r["ll_steps"], ref[pos][2]), len(idx), x.shape, np.array_equal(x.view(np.uint64), x2.view(np.uint64)), np.array_equal(x2.view(np.uint64), rx.view(np.uint64))))
        print("   particles %d..%d, %d of them in %d contiguous runs: %s" % (pi.min(), pi.max(), len(pi), len(runs), [(int(a[0]), int(a[-1])) for a in runs[:10]]))
        raw = x.view(np.uint64)[ne]
        print("   raw values: %s" % " ".join("%x" % v for v in raw))
        print("   differences between consecutive raw values: %s" % np.diff(raw.astype(np.int64)).tolist())
        k = tuple(idx[0])
        print("   first differing entry %s: engine %r reference %r; ancestors there %s" % (list(k), x[k], rx[k], j[pi[:4]].tolist()))
        xs = np.take(x, pi[:3], axis=pax); rs = np.take(rx, pi[:3], axis=pax)
        print("   engine rows\n%s\n   reference rows\n%s" % (xs, rs))
print("bad: %d of %d" % (nbad, RUNS))
