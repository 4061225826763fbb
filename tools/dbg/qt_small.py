import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import models as M
from llpf_amd import _capi, _structs as S
model = M.quadtank_model()
for N in (500, 1000, 2000, 8000, 50000):
    T = 400
    U, Y = M.quadtank_data(T, seed=2)
    cfg = S.make_config(model, N, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 5, 0)
    h = _capi.FilterHandle(cfg)
    for rep in range(3):
        h.reset(); r = h.run(U, Y, 1.0)
    print("N=%6d  %.2f ms per run, %.1f us/step, ll=%.6f" % (N, h.last_run_ms(), 1e3 * h.last_run_ms() / T, r["ll"]))
