"""round 5: the two intermittent failures of `fuzz_parity.py --big` (seed 14 case 3: xmean; seed 15 case 20: final particles) — who moves,
the engine or the (threaded) oracle?  Each configuration is run many times on fresh engine handles and on fresh oracle handles with 8 and
with 1 thread; every output is hashed."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import fuzz_parity as FP
FP.BIG = True
S, ob, _capi = FP.S, FP.ob, FP._capi


def case_of(seed, index):
    rng = np.random.default_rng(seed)
    for i in range(index + 1):
        assert not (rng.random() < 0.2 and not FP.BIG)
        c = FP.rand_case(rng)
    return c


def digest(*arrs):
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:10]


def outputs(h, c, history):
    h.reset()
    r = h.run(c["U"], c["Y"], c["t0"], ll_steps=True, xmean=not history, history=history)
    out = {"ll": digest(r["ll_steps"]), "x_final": digest(h.particles()), "w_final": digest(h.weights()), "j": digest(h.ancestors())}
    if history:
        out["hist_x"] = digest(r["x"]); out["hist_w"] = digest(r["w"])
    else:
        out["xmean"] = digest(r["xmean"])
    return out


if __name__ == "__main__":
    for seed, index, history in ((14, 3, False), (15, 20, True)):
        c = case_of(seed, index)
        cfg = S.make_config(c["model"], c["N"], c["kind"], c["strat"], c["thr"], c["seed"], 0)
        print("seed %d case %d: N=%d thr=%g strat=%d T=%d nx=%d ny=%d driver=%s" % (seed, index, c["N"], c["thr"], c["strat"], c["T"], c["model"].nx, c["model"].ny, c["driver"]), flush=True)
        seen = {}
        for who, n, threads in (("engine", 40, 0), ("oracle8", 12, 8), ("oracle1", 3, 1)):
            if threads:
                ob.set_threads(threads)
            for t in range(n):
                h = _capi.FilterHandle(cfg) if who == "engine" else ob.OracleFilter(cfg, ob.ORDER_DEVICE)
                o = outputs(h, c, history)
                key = tuple(sorted(o.items()))
                seen.setdefault(key, []).append("%s#%d" % (who, t))
                del h
        print(" distinct outcomes: %d" % len(seen))
        ref = max(seen, key=lambda k: len(seen[k]))
        for k, v in seen.items():
            tag = "majority" if k == ref else "DEVIANT: " + ", ".join("%s" % (a,) for a, b2 in zip(k, ref) if a != b2)
            print("  %3d runs (%s ...) %s" % (len(v), ", ".join(v[:4]), tag))
