"""round 5, after the store-hazard fix: do the BASELINE workloads reproduce themselves?  Each configuration (C2, C3, C5, the auxiliary
filter, the shared-covariance RBPF at their benchmark sizes, shortened runs; a 16-filter share of C4) is run many times — a fresh
handle every CYCLE runs, equal positions of the cycle compared — and ll per step, final particles, weights and ancestors are hashed.
usage: stress_configs.py [runs per configuration]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import bench
from llpf_amd import _capi, _structs as S

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
CYCLE = 5


def digest(*arrs):
    h = hashlib.sha1()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:10]


def stress(name, make, run):
    seen = {}
    h = None
    for t in range(RUNS):
        if t % CYCLE == 0:
            h = make()
        h.reset()
        out = run(h)
        seen.setdefault(t % CYCLE, {}).setdefault(out, []).append(t)
    bad = {k: v for k, v in seen.items() if len(v) > 1}
    print("%-28s %d runs: positions with more than one outcome: %d" % (name, RUNS, len(bad)), flush=True)
    for k, v in bad.items():
        ref = max(v, key=lambda q: len(v[q]))
        for q, runs in v.items():
            if q != ref:
                print("    position %d, runs %s deviate in: %s" % (k, runs[:6], ", ".join(a[0] for a, b2 in zip(q, ref) if a != b2)))


for wl, N, T in (("lg", 1000000, 40), ("quadtank", 1000000, 40), ("rbpf_full", 200000, 30), ("aux", 1000000, 30), ("rbpf", 1000000, 40)):
    model, U, Y, kind, thr, label = bench.build_workload(wl, N, T)
    cfg = S.make_config(model, N, kind, S.RESAMPLE_SYSTEMATIC, thr, 1000, 0)
    aux = wl == "aux"

    def run(h, aux=aux, U=U, Y=Y):
        r = h.run_aux(U, Y, 1, ll_steps=True) if aux else h.run(U, Y, 1.0, ll_steps=True)
        return tuple(sorted({"ll": digest(r["ll_steps"]), "x": digest(h.particles()), "w": digest(h.weights()), "j": digest(h.ancestors())}.items()))
    stress(wl + " N=%d T=%d" % (N, T), lambda cfg=cfg: _capi.FilterHandle(cfg), run)

# a share of C4: 16 filters x 1e5 (split schedule is chosen from 3 M particles on: 32 filters)
import models as M
model = M.lg_test_model()
_, U, Y = M.simulate_lg(model, 30, seed=1)
cfg = S.make_config(model, 100000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 1000, 0)
ms = []
for k in range(40):
    m = S.Model.from_buffer_copy(bytes(model))
    m.A[0] = m.A[0] * (1.0 - 0.002 * k)
    ms.append(m)


def run_bank(b):
    r = b.run(U, Y, 1.0, ll_steps=True)
    return tuple(sorted({"ll": digest(r["ll_steps"])}.items()))


def make_bank():
    return _capi.BankHandle(cfg, ms)


stress("bank 40 x 1e5 T=30", make_bank, run_bank)
