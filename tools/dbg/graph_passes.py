import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, models as M
from llpf_amd import _capi, _structs as S
model = M.lg_test_model(); _, U, Y = M.simulate_lg(model, 1000, seed=1)
cfg = S.make_config(model, 1000000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 1.0, 5, 0)
g = _capi.FilterHandle(cfg)
for p in range(6):
    g.reset()
    t0 = time.perf_counter(); r = g.run(U, Y, 1.0); dt = time.perf_counter() - t0
    print("pass", p, "wall ms %.2f" % (1e3 * dt), "device ms %.2f" % g.last_run_ms(), "ll %.6f" % r["ll"])
