"""round 5: repeatability over a variety of configurations at N = 1e6 — resamplers, thresholds, state dimensions (precompiled and
compiled on demand), filter kinds, run outputs (weighted mean, covariance, history) — each run RUNS times, a fresh handle every 4.
usage: stress_variety.py [runs]"""
import hashlib, itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import models as M
from llpf_amd import _capi, _structs as S

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CYCLE = 4


def digest(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]


def lg(nx, ny, rng):
    Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
    return S.make_lg_model(Q @ np.diag(np.linspace(0.4, 0.97, nx)) @ Q.T, np.zeros((nx, 0)), rng.standard_normal((ny, nx)),
                           S.make_gaussian(np.zeros(nx), 0.05), S.make_gaussian(np.zeros(ny), 0.3), S.make_gaussian(np.zeros(nx), 2.0), 1.0)


rng = np.random.default_rng(9)
N, T = 1000000, 6
cases = []
for nx, strat, thr, kind, outs in ((2, S.RESAMPLE_SYSTEMATIC, 0.5, S.PARTICLE_FILTER, dict(xmean=True)),
                                   (2, S.RESAMPLE_STRATIFIED, 1.0, S.PARTICLE_FILTER, dict(xmean=True, xcov=True)),
                                   (2, S.RESAMPLE_RESIDUAL, 0.5, S.PARTICLE_FILTER, dict()),
                                   (3, S.RESAMPLE_RESIDUAL, 1.0, S.ADVANCED_PARTICLE_FILTER, dict(history=True)),
                                   (4, S.RESAMPLE_STRATIFIED, 0.3, S.PARTICLE_FILTER, dict(history=True, xcov=True)),
                                   (6, S.RESAMPLE_SYSTEMATIC, 0.5, S.ADVANCED_PARTICLE_FILTER, dict(xmean=True)),
                                   (12, S.RESAMPLE_SYSTEMATIC, 1.0, S.PARTICLE_FILTER, dict())):
    m = lg(nx, min(nx, 3), rng)
    _, U, Y = M.simulate_lg(m, T, seed=nx)
    cases.append(("lg nx=%d strat=%d thr=%g kind=%d %s" % (nx, strat, thr, kind, "+".join(outs) or "plain"), m, U, Y, kind, strat, thr, outs, 0.0))
m = M.quadtank_model()
U, Y = M.quadtank_data(T, seed=2)
cases.append(("quadtank strat=1 thr=0.5 history", m, U, Y, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_STRATIFIED, 0.5, dict(history=True), 1.0))
cases.append(("quadtank residual thr=0.1", m, U, Y, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_RESIDUAL, 0.1, dict(), 1.0))

for name, m, U, Y, kind, strat, thr, outs, t0 in cases:
    cfg = S.make_config(m, N, kind, strat, thr, 4242, 0)
    seen = {}
    h = None
    for t in range(RUNS):
        if t % CYCLE == 0:
            h = _capi.FilterHandle(cfg)
        h.reset()
        r = h.run(U, Y, t0, ll_steps=True, **outs)
        o = [("ll", digest(r["ll_steps"])), ("x", digest(h.particles())), ("w", digest(h.weights())), ("j", digest(h.ancestors()))]
        for k in ("xmean", "xcov", "x", "w"):
            if r.get(k) is not None and (k in outs or (k in ("x", "w") and outs.get("history"))):
                o.append(("run." + k, digest(r[k])))
        seen.setdefault(t % CYCLE, {}).setdefault(tuple(o), []).append(t)
    bad = {k: v for k, v in seen.items() if len(v) > 1}
    print("%-58s %d runs: positions with more than one outcome: %d" % (name, RUNS, len(bad)), flush=True)
    for k, v in bad.items():
        ref = max(v, key=lambda q: len(v[q]))
        for q, runs in v.items():
            if q != ref:
                print("    position %d, runs %s deviate in: %s" % (k, runs[:6], ", ".join(a[0] for a, b2 in zip(q, ref) if a != b2)))
