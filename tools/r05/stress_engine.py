"""round 5: is the engine deterministic under repetition at N = 1e6?  The two configurations that failed once each in `fuzz_parity.py --big`
(wrong xmean row; wrong final particles after a history run) are run many hundred times — on one handle and on fresh ones, the history
run also without its history outputs — and every output is hashed.  usage: stress_engine.py [runs]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import fuzz_parity as FP
from stress_big import case_of, digest
FP.BIG = True
S, _capi = FP.S, FP._capi
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 400

for seed, index, history in ((14, 3, False), (15, 20, True), (15, 20, False)):
    c = case_of(seed, index)
    cfg = S.make_config(c["model"], c["N"], c["kind"], c["strat"], c["thr"], c["seed"], 0)
    print("seed %d case %d history=%s: N=%d thr=%g strat=%d T=%d nx=%d ny=%d" % (seed, index, history, c["N"], c["thr"], c["strat"], c["T"], c["model"].nx, c["model"].ny), flush=True)
    seen = {}
    h = None
    for t in range(RUNS):
        if t % 40 == 0:
            h = _capi.FilterHandle(cfg)
        h.reset()
        r = h.run(c["U"], c["Y"], c["t0"], ll_steps=True, xmean=not history, history=history)
        o = {"ll": digest(r["ll_steps"]), "x_final": digest(h.particles()), "w_final": digest(h.weights()), "j": digest(h.ancestors())}
        if history:
            o["hist_x"] = digest(r["x"]); o["hist_w"] = digest(r["w"])
        else:
            o["xmean"] = digest(r["xmean"])
        # a handle's k-th run after k resets draws its own noise: compare like with like
        key = (t % 40, tuple(sorted(o.items())))
        seen.setdefault(t % 40, {}).setdefault(key[1], []).append(t)
    bad = {k: v for k, v in seen.items() if len(v) > 1}
    print(" positions with more than one outcome: %d of %d" % (len(bad), len(seen)))
    for k, v in list(bad.items())[:5]:
        ref = max(v, key=lambda q: len(v[q]))
        for q, runs in v.items():
            if q != ref:
                print("  position %d, runs %s deviate in: %s" % (k, runs, ", ".join(a[0] for a, b2 in zip(q, ref) if a != b2)))
