"""A likelihood that declares no bound (loglik without loglik_bound) against the same likelihood with its bound: ms per run.  The former
takes the exact-max form at every timestep — as launches of the run loop since round 4, as a host round trip per timestep before."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import models as M, user_models as UM
from llpf_amd import _capi, _structs as S
base = M.lg_test_model()
_, U, Y = M.simulate_lg(base, 500)
for name, src in (("bound", UM.LAPLACE_SRC), ("no bound", UM.LAPLACE_NO_BOUND_SRC)):
    m = S.Model.from_buffer_copy(bytes(base))
    m.model_id = _capi.model_compile(src, m.nx, m.ny)
    m.qt[0] = 0.8
    for N in (10000, 1000000):
        g = _capi.FilterHandle(S.make_config(m, N, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 3, 0))
        for _ in range(3):
            g.reset(); r = g.run(U, Y, 1.0)
        print(name, "N", N, "ms per run of 500 steps %.2f" % g.last_run_ms(), "ll %.6f" % r["ll"])
