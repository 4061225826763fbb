import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, models as M, oracle_binding as ob
from llpf_amd import _capi, _structs as S
from gpu_common import cfg_of
for N in (4, 8192):
    cfg = cfg_of(M.lg_test_model(), N, thr=0.5, seed=41)
    x = np.random.default_rng(5).standard_normal((N, 2)); u = np.array([0.2])
    for support in (N // 2, N // 2 - 1, N // 2 + 1):
        w = np.full(N, -np.inf); w[:support] = -1.25
        g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
        for h in (g, o):
            h.set_particles(x); h.set_weights(w)
        print(N, support, "ess", g.ess(), o.ess(), "should", g.shouldresample(), o.shouldresample())
        for h in (g, o):
            h.predict(u, 0.0)
        print("   count", g.resample_count(), o.resample_count(), "last", g.last_resampled(), o.last_resampled(), "anc", g.ancestors()[:6], o.ancestors()[:6],
              "x eq", np.array_equal(g.particles(), o.particles()), "w", g.weights()[:4], o.weights()[:4])
