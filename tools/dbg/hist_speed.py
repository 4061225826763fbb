import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, models as M
from llpf_amd import _capi, _structs as S
model = M.lg_test_model(); _, U, Y = M.simulate_lg(model, 200, seed=1)
for N in (10000, 100000, 1000000):
    cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 5, 0)
    g = _capi.FilterHandle(cfg)
    g.reset(); g.run(U[:5], Y[:5], 0.0, history=True)
    g.reset(); t0 = time.perf_counter(); r = g.run(U, Y, 0.0, history=True); dt = time.perf_counter() - t0
    g.reset(); t0 = time.perf_counter(); r2 = g.run(U, Y, 0.0); dt2 = time.perf_counter() - t0
    gb = N * 200 * 32 / 1e9
    print("N", N, "forward_trajectory with history: %.1f ms (%.0f us/step, %.2f GB/s of history)" % (1e3 * dt, 1e6 * dt / 200, gb / dt), "without: %.1f ms" % (1e3 * dt2))
