"""Bit-for-bit check of the persistent k_rbfull (quad-tank RB model, per-particle 8x8 covariance) against the device-order oracle at
awkward sizes: below one batch, not a multiple of 64 or 1024, just below / above the number of resident waves (131072 particles),
several batches per wave; resampled and identity steps.  Developer aid: the same comparison at fixed sizes is tests/test_gpu_rbfull.py."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_binding as ob
import rbfull_models as M
from llpf_amd import _capi, _structs as S
ob.set_threads(8)
model = M.quadtank_case()
ok = True
for N, thr in ((65, 0.5), (1000, 0.9), (131071, 0.3), (131073, 0.9), (150001, 0.1), (262145, 0.6), (131072 + 64, 1.0)):
    T = 4
    U, Y = M.simulate_io(model, T, seed=N % 97)
    cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 7 + N, 0)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    rg = g.run(U, Y, 0.0, ll_steps=True); ro = o.run(U, Y, 0.0, ll_steps=True)
    same = np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    xs = np.array_equal(g.particles().view(np.uint64), o.particles().view(np.uint64))
    xl_g, R_g = g.rb_linear_state(); xl_o, R_o = o.rb_linear_state()
    rs = np.array_equal(R_g.view(np.uint64), R_o.view(np.uint64)) and np.array_equal(xl_g.view(np.uint64), xl_o.view(np.uint64))
    print(N, thr, "ll", same, "x", xs, "xl/R", rs, "resamples", g.resample_count(), o.resample_count())
    ok = ok and same and xs and rs
print("ALL OK" if ok else "MISMATCH")
