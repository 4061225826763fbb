import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rbfull_models as RM
from llpf_amd import _capi, _structs as S
for name, model in (("quadtank", RM.quadtank_case()), ("linear", RM.linear_case(4, 8, 2, seed=1)[0])):
    T = 300
    U, Y = RM.simulate_io(model, T, seed=3)
    h = _capi.FilterHandle(S.make_config(model, 200000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 5, 0))
    for rep in range(3):
        h.reset(); r = h.run(U, Y, 1.0)
    print("%s: %.1f us per timestep" % (name, 1e3 * h.last_run_ms() / T))
