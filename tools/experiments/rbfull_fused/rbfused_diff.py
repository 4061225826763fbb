"""Where do the one-launch form of k_rbfull (kernels/rbfused.hpp) and the two-launch form differ?  Ancestors / particles after a short run."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import llpf_amd
from llpf_amd import _capi, _structs as S
import rbfull_models as RM

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2
model = RM.quadtank_case()
U, Y = RM.simulate_io(model, T, seed=3)
cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 9, 0)
out = {}
for v in ("0", "1"):
    os.environ["LLPF_RB_FUSED"] = v
    os.environ["LLPF_GRAPH"] = "0"
    g = _capi.FilterHandle(cfg)
    g.reset()
    r = g.run(U, Y, 0.0, ll_steps=True)
    out[v] = (r["ll_steps"].copy(), g.ancestors().copy(), g.particles().copy(), g.resample_count())
a0, a1 = out["0"][1], out["1"][1]
bad = np.nonzero(a0 != a1)[0]
print("N", N, "thr", thr, "T", T, "ll", out["0"][0], out["1"][0], "resamples", out["0"][3], out["1"][3])
print("ancestor mismatches:", bad.size)
if bad.size:
    print("first 40 outputs:", bad[:40], "batches (64):", np.unique(bad // 64)[:40], "lanes:", np.unique(bad % 64)[:64])
    print("two-launch:", a0[bad[:20]], "one-launch:", a1[bad[:20]])
    print("batch index mod 2048 >= ?: second-round batches among the bad:", np.sum(bad // 64 >= 2048), "of", bad.size)
x0, x1 = out["0"][2], out["1"][2]
print("particle rows differing:", int(np.sum(np.any(x0 != x1, axis=1))))
