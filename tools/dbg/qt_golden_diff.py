import sys, os, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_golden as tg
from llpf_amd import _capi, _structs as S
fn, mk, N, kind, thr, strat, t0 = tg.CASES['qt']
d = np.load(os.path.join(tg.G, fn))
g = _capi.FilterHandle(S.make_config(mk(), N, kind, strat, thr, 7, 0)); g.reset()
r = g.run(d["U"], d["Y"], t0, ll_steps=True, xmean=True)
x = g.particles(); xe = d["x_final_dev"]
bad = np.nonzero((x != xe).any(axis=1))[0]
print("N", N, "thr", thr, "strat", strat, "T", len(d["Y"]), "nbad", len(bad), bad[:20])
print("ll equal", np.array_equal(r["ll_steps"], d["ll_steps_dev"]))
if len(bad): print(x[bad[:3]], xe[bad[:3]])
