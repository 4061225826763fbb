"""Repeatability of the split schedule without stored quanta (two weight buffers, quanta formed in the fused kernel's scan) at sizes whose
blocks are NOT co-resident: the same run again and again on fresh handles and on one handle, every output hashed.  A race between blocks of
one launch shows up as differing hashes (the first version of this schedule read weights that other blocks were replacing: three different
log-likelihoods in three runs at N = 1.6e7, none below 4e6).    python tools/dbg/lazy_repeat.py [--reps 12]"""
import argparse, hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import models as M
from llpf_amd import _capi, _structs as S

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=12)
args = ap.parse_args()


def digest(h, r):
    m = hashlib.sha256()
    m.update(np.ascontiguousarray(r["ll_steps"]).tobytes())
    for a in (h.particles(), h.weights(), h.ancestors()):
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()[:16]


bad = 0
model = M.lg_test_model(0.1)
for N, T, thr in ((16_000_000, 24, 0.5), (4_200_000, 40, 0.3), (1_100_077 * 3, 30, 0.1)):
    _, U, Y = M.simulate_lg(model, T)
    cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 4242, 0)
    hs = set()
    one = _capi.FilterHandle(cfg)
    for rep in range(args.reps):
        h = _capi.FilterHandle(cfg) if rep % 2 == 0 else one
        h.seed(4242) if hasattr(h, "seed") else None
        h.reset()
        r = h.run(U, Y, 1.0, ll_steps=True)
        hs.add(digest(h, r))
    print("single filter N=%d T=%d thr=%.1f: %d runs, %d distinct digests, resamples %d" % (N, T, thr, args.reps, len(hs), one.resample_count()))
    bad += len(hs) != 1
F, N, T = 128, 100000, 60
models = [M.lg_test_model(0.03 + 0.002 * k) for k in range(F)]
_, U, Y = M.simulate_lg(models[40], T)
for thr in (0.1, 0.5):
    hs = set()
    for rep in range(args.reps):
        bank = _capi.BankHandle(S.make_config(models[0], N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 777, 0), models)
        bank.reset()
        r = bank.run(U, Y, 1.0, ll_steps=True)
        hs.add(hashlib.sha256(np.ascontiguousarray(r["ll_steps"]).tobytes()).hexdigest()[:16])
    print("bank %d x %d T=%d thr=%.1f: %d runs, %d distinct digests" % (F, N, T, thr, args.reps, len(hs)))
    bad += len(hs) != 1
print("OK" if bad == 0 else "NOT REPEATABLE: %d configurations" % bad)
sys.exit(1 if bad else 0)
