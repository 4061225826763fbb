"""Shared helpers of the GPU parity tests."""
import numpy as np

from llpf_amd import _structs as S

TOL_LL_STEP = 1e-10      # per-step |ll_gpu - ll_ref_order|   
TOL_LL_SUM = 1e-8        # cumulative
TOL_WE_REL = 1e-12       # max relative error of exp-weights vs reference order


def cfg_of(model, N, strategy=S.RESAMPLE_SYSTEMATIC, thr=0.1, seed=7, kind=S.PARTICLE_FILTER):
    return S.make_config(model, N, kind, strategy, thr, seed, 0)


def compare_state(g, o, exact=True, we_rtol=0.0):
    xg, xo = g.particles(), o.particles()
    wg, wo = g.weights(), o.weights()
    eg, eo = g.expweights(), o.expweights()
    if exact:
        assert np.array_equal(xg.view(np.uint64), xo.view(np.uint64)), "particles differ"
        assert np.array_equal(wg.view(np.uint64), wo.view(np.uint64)), "log-weights differ"
        assert np.array_equal(eg.view(np.uint64), eo.view(np.uint64)), "exp-weights differ"
        assert np.array_equal(g.ancestors(), o.ancestors()), "ancestors differ"
    else:
        np.testing.assert_allclose(xg, xo, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(eg, eo, rtol=we_rtol, atol=1e-300)
