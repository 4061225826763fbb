"""State after short runs, one-launch form against k_norm + fused kernel (tools/experiments/one_launch_large/one_launch.patch applied): how the
shared mangled name of the two translation units' accumulating kernels was found (EXPERIMENTS 6.16)."""
import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
import numpy as np
import oracle_binding as ob, models as M
from llpf_amd import _capi, _structs as S
os.environ["LLPF_SCHEDULE"] = "split"
pass
model = M.lg_test_model(0.1)
N = 5000
_, U, Y = M.simulate_lg(model, 14)
for thr in (0.9, 0.001):
  cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 31, 0)
  for T in (1, 2, 3):
    st = {}
    for ol in ("1", "0"):
        os.environ["LLPF_ONE_LAUNCH"] = ol
        g = _capi.FilterHandle(cfg); g.reset()
        rg = g.run(U[:T], Y[:T], 1.0, ll_steps=True)
        st[ol] = (rg["ll_steps"].copy(), g.particles().copy(), g.weights().copy(), g.ancestors().copy(), g.resample_count(), g.ess())
    a, b = st["1"], st["0"]
    print("thr %.3f T=%d ll equal %s (first diff %s)  x equal %s  w equal %s  anc equal %s  resamples %d/%d ess %r %r" % (thr,
        T, np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64)), np.nonzero(a[0] != b[0])[0][:2], np.array_equal(a[1], b[1]),
        np.array_equal(a[2], b[2]), np.array_equal(a[3], b[3]), a[4], b[4], a[5], b[5]))
    if not np.array_equal(a[0], b[0]): print("   ll", a[0], b[0])
    if not np.array_equal(a[3], b[3]):
        d = np.nonzero(a[3] != b[3])[0]
        print("   anc differ at", d[:6], "count", len(d), "vals", a[3][d[:4]], b[3][d[:4]])
