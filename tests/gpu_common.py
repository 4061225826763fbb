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


def teacher_forced_ancestor_mismatches(cfg, U, Y, steps, t_index0=1.0):
    """the engine against the REFERENCE-ORDER oracle (libm exp, pairwise sum, serial fp64 cumsum, two-pointer search:
    src/utils.jl:18-27, src/resample.jl:17-36) at full size.  A particle filter is chaotic in its ancestry, so the two are compared
    step by step from the SAME state: the reference-order state (particles, log-weights) is installed in the engine
    (llpf_set_particles / llpf_set_weights) before every correct! and again before every predict!; both sides then take that step
    with the same measurement / the same Philox draws, and the oracle alone carries the recursion on.  Compared: the log-likelihood
    increment and the normalised exp-weights of every correct! (the stated fp64 tolerance: |dll| <= 1e-10, rel <= 1e-12), the
    ancestor vectors of every resampling predict!, and the propagated particles of every output whose ancestor agrees.
    Returns dict(steps, resampling_steps, mismatches_total, mismatches_per_step_max, steps_with_mismatch,
    particles_equal_on_matching_ancestors, correct_steps, ll_abs_err_max, expweights_rel_err_max)."""
    import oracle_binding as ob
    from llpf_amd import _capi
    g = _capi.FilterHandle(cfg)
    r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    g.reset(); r.reset()
    Ts = cfg.model.Ts
    tot = worst = nsteps = nres = ncorr = 0
    same_x = True
    dll = dwe = 0.0
    for k in range(steps):
        t = (t_index0 + k) * Ts
        u = U[k] if U is not None and len(U) else None
        g.set_particles(r.particles())              # correct! from the same state on both sides
        g.set_weights(r.weights())
        ll_g = g.correct(u, Y[k], t)
        ll_r = r.correct(u, Y[k], t)
        if not np.any(np.isnan(Y[k])):
            ncorr += 1
            dll = max(dll, abs(ll_g - ll_r))
            eg, er = g.expweights(), r.expweights()
            nz = er > 1e-290
            dwe = max(dwe, float(np.max(np.abs(eg[nz] - er[nz]) / er[nz])))
        g.set_weights(r.weights())                  # predict! from the same state (particles are unchanged by correct!)
        g.predict(u, t)
        r.predict(u, t)
        if not r.last_resampled():
            assert not g.last_resampled()
            continue
        nres += 1
        jg, jr = g.ancestors(), r.ancestors()
        diff = jg != jr
        m = int(np.sum(diff))
        tot += m
        worst = max(worst, m)
        nsteps += 1 if m else 0
        xg, xr = g.particles(), r.particles()
        same_x = same_x and bool(np.array_equal(xg[~diff], xr[~diff]))
    return {"steps": int(steps), "resampling_steps": int(nres), "mismatches_total": int(tot), "mismatches_per_step_max": int(worst),
            "steps_with_mismatch": int(nsteps), "particles_equal_on_matching_ancestors": same_x,
            "correct_steps": int(ncorr), "ll_abs_err_max": float(dll), "expweights_rel_err_max": float(dwe)}
