import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from llpf_amd import _capi, _structs as S
from gpu_common import cfg_of
rng = np.random.default_rng(0)
for nx, ny in ((1, 1), (2, 1), (2, 2), (3, 2), (4, 2), (4, 4)):
    A = 0.9 * np.eye(nx) + 0.05 * rng.standard_normal((nx, nx))
    B = rng.standard_normal((nx, 1)); Cm = rng.standard_normal((ny, nx))
    g = S.make_gaussian
    model = S.make_lg_model(A, B, Cm, g(np.zeros(nx), 0.04 * np.eye(nx) + 0.01), g(np.zeros(ny), np.full(ny, 0.5)), g(np.zeros(nx), 2.0))
    U = rng.standard_normal((30, 1)); Y = rng.standard_normal((30, ny))
    cfg = cfg_of(model, 7000, S.RESAMPLE_SYSTEMATIC, 0.4, seed=2)
    res = {}
    for p in ("1", "0"):
        os.environ["LLPF_PERSIST"] = p
        h = _capi.FilterHandle(cfg); h.reset()
        r = h.run(U, Y, 1.0, ll_steps=True)
        res[p] = (r["ll_steps"], h.particles(), h.last_run_stats(), h.resample_count())
    d = np.nonzero(res["1"][0] != res["0"][0])[0]
    print(nx, ny, "first differing step", d[:3], res["1"][2], res["1"][3], res["0"][3], "x equal", np.array_equal(res["1"][1], res["0"][1]))
