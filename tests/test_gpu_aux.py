"""GPU parity of the AuxiliaryParticleFilter path (reference src/filtering.jl:170-217, 367-384; smoothing.jl:232-236)."""
import numpy as np
import pytest

import llpf_amd
from llpf_amd import _capi, _structs as S
import models as M
import oracle_binding as ob
from gpu_common import TOL_LL_STEP, cfg_of as _cfg, compare_state as _compare_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("strategy", [S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED, S.RESAMPLE_RESIDUAL])
def test_aux_filter_single_steps_bit_exact(strategy):
    """correct! / predict! of the auxiliary filter step by step: ll, particles, log-weights, lambda (the reference's
    `we` after predict!), ancestors — bit-identical to the device-order oracle; a missing look-ahead measurement."""
    model = M.lg_c1_model()
    _, U, Y = M.simulate_lg(model, 30)
    cfg = _cfg(model, 3000, strategy, 0.1, seed=31)
    g = _capi.FilterHandle(cfg)
    o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    for k in range(12):
        ll_g, ll_o = g.aux_correct(), o.aux_correct()
        assert np.float64(ll_g).view(np.uint64) == np.float64(ll_o).view(np.uint64), k
        _compare_state(g, o)
        y1 = None if k == 5 else Y[k + 1]
        g.aux_predict(U[k], y1, k * 1.0); o.aux_predict(U[k], y1, k * 1.0)
        _compare_state(g, o)                      # w = lambda - log N, expweights = lambda, j, x
        assert g.index() == o.index()
    # the wrapped filter's update! on the same state (the last step of loglik, src/smoothing.jl:235)
    ll_g, ll_o = g.update(U[12], Y[12], 12.0), o.update(U[12], Y[12], 12.0)
    assert ll_g == ll_o
    _compare_state(g, o)


@pytest.mark.parametrize("mode", [0, 1])
def test_aux_filter_trajectories(mode):
    """forward_trajectory (mode 0, with history) and loglik (mode 1) loops of the auxiliary filter: bit-identical to
    the device-order oracle, within tolerance of the reference-order one; includes an outlier (exact-max redo of
    both normalisations) and a missing measurement."""
    model = M.lg_test_model(0.1)
    _, U, Y = M.simulate_lg(model, 60, seed=5)
    Y = Y.copy()
    Y[23] += 11.0
    Y[40] = np.nan
    cfg = _cfg(model, 4000, S.RESAMPLE_SYSTEMATIC, 0.1, seed=33)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    for h in (g, o, r):
        h.reset()
    hist = mode == 0
    rg = g.run_aux(U, Y, mode, ll_steps=True, xmean=hist, history=hist)
    ro = o.run_aux(U, Y, mode, ll_steps=True, xmean=hist, history=hist)
    rr = r.run_aux(U, Y, mode, ll_steps=True)
    assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    assert rg["ll"] == ro["ll"]
    assert o.exact_steps() >= 1
    _compare_state(g, o)
    if hist:
        for key in ("x", "w", "we"):
            assert np.array_equal(rg[key].view(np.uint64), ro[key].view(np.uint64)), key
        np.testing.assert_allclose(rg["xmean"], ro["xmean"], rtol=1e-9, atol=1e-11)
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= TOL_LL_STEP
    assert g.resample_count() == o.resample_count()


def test_aux_filter_api_and_kalman():
    """The reference-shaped API (AuxiliaryParticleFilter(N, dynamics, ...), pfa(u, y, y1), loglik, forward_trajectory)
    and the reference's statistical check: |ll_KF - ll_APF| < 20 at N = 1000, T = 2000 (test/runtests.jl:446)."""
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]]); B = np.array([[0.1], [0.0]]); Cm = np.array([[0.0, 1.0]])
    df = llpf_amd.MvNormal(np.zeros(2), 0.1 ** 2); dg = llpf_amd.MvNormal(np.zeros(1), np.ones(1)); d0 = llpf_amd.MvNormal([0.3, -0.5], 4.0)
    pfa = llpf_amd.AuxiliaryParticleFilter(1000, llpf_amd.LinearDynamics(A, B), llpf_amd.LinearMeasurement(Cm), df, dg, d0, rng=5)
    assert not llpf_amd.shouldresample(pfa)
    model = M.lg_test_model(0.1)
    _, U, Y = M.simulate_lg(model, 2000, seed=3)
    ll = llpf_amd.loglik(pfa, U, Y)
    assert abs(ll - ob.kalman_loglik(model, U, Y)) < 20
    llpf_amd.reset(pfa)
    l0, _ = pfa(U[0], Y[0], Y[1])
    assert abs(l0) < 1e-12 and llpf_amd.index(pfa) == 2
    sol = llpf_amd.forward_trajectory(pfa, U[:20], Y[:20])
    assert sol.x.shape == (20, 1000, 2) and np.allclose(sol.we.sum(axis=1), 1.0)


def test_aux_filter_bank():
    """Bank of auxiliary filters (the ML sweep of test/runtests.jl:419-423) == the filters run one at a time."""
    models = [M.lg_test_model(s) for s in (0.05, 0.1, 0.2, 0.4)]
    _, U, Y = M.simulate_lg(models[1], 40)
    Y = Y.copy(); Y[17] += 9.0
    N = 3000
    bank = _capi.BankHandle(_cfg(models[0], N, thr=0.1, seed=950), models)
    bank.reset()
    rb = bank.run_aux(U, Y, mode=1, ll_steps=True)
    for k, mk in enumerate(models):
        o = ob.OracleFilter(_cfg(mk, N, thr=0.1, seed=950 + k), ob.ORDER_DEVICE)
        o.reset()
        ro = o.run_aux(U, Y, mode=1, ll_steps=True)
        assert np.array_equal(ro["ll_steps"].view(np.uint64), rb["ll_steps"][:, k].copy().view(np.uint64)), k


def test_aux_filter_async_run_with_several_outliers():
    """The asynchronous loglik / forward loop of the auxiliary filter (all launches enqueued, one poll): outliers that
    make the look-ahead normalisation AND the next correct! fall back to the exact-max form, several times per run,
    in a bank where only some filters are affected; T = 1 and T = 2 edge cases."""
    models = [M.lg_test_model(s) for s in (0.03, 0.1, 0.3)]
    _, U, Y = M.simulate_lg(models[1], 80, seed=8)
    Y = Y.copy()
    Y[10] += 12.0; Y[11] -= 9.0; Y[50] += 15.0; Y[51] = np.nan; Y[79] += 10.0
    N = 2500
    for mode in (0, 1):
        bank = _capi.BankHandle(_cfg(models[0], N, S.RESAMPLE_STRATIFIED, 0.1, seed=77), models)
        bank.reset()
        rb = bank.run_aux(U, Y, mode=mode, ll_steps=True)
        n_exact = 0
        for k, mk in enumerate(models):
            o = ob.OracleFilter(_cfg(mk, N, S.RESAMPLE_STRATIFIED, 0.1, seed=77 + k), ob.ORDER_DEVICE)
            o.reset()
            ro = o.run_aux(U, Y, mode=mode, ll_steps=True)
            n_exact += o.exact_steps()
            assert np.array_equal(ro["ll_steps"].view(np.uint64), rb["ll_steps"][:, k].copy().view(np.uint64)), (mode, k)
            assert ro["ll"] == rb["ll"][k]
        assert n_exact >= 4
    for T in (1, 2):
        for mode in (0, 1):
            cfg = _cfg(models[1], 1000, seed=5)
            g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
            g.reset(); o.reset()
            rg = g.run_aux(U[:T], Y[:T], mode, ll_steps=True); ro = o.run_aux(U[:T], Y[:T], mode, ll_steps=True)
            assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64)), (T, mode)
            _compare_state(g, o)


@pytest.mark.parametrize("which", ["quadtank", "lineargaussian"])
def test_aux_filter_over_an_advanced_filter(which):
    """AuxiliaryParticleFilter{AdvancedParticleFilter} (src/filtering.jl:219-234): the look-ahead weights only choose the ancestors,
    the particles are propagated again from xprev[j] with noise and the weights reset.  Single steps, forward_trajectory and loglik
    loops: bit-identical to the device-order oracle, within tolerance of the reference order; every auxiliary correct! returns ~0."""
    if which == "quadtank":
        model = M.quadtank_model()
        U, Y = M.quadtank_data(25, seed=2)
        t0 = 490.0
    else:
        model = M.lg_c1_model()
        _, U, Y = M.simulate_lg(model, 25)
        t0 = 0.0
    cfg = _cfg(model, 4000, S.RESAMPLE_SYSTEMATIC, 0.5, seed=13, kind=S.ADVANCED_PARTICLE_FILTER)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    for h in (g, o, r):
        h.reset()
    for k in range(10):
        ll_g, ll_o, ll_r = g.aux_correct(), o.aux_correct(), r.aux_correct()
        assert ll_g == ll_o and abs(ll_g - ll_r) < TOL_LL_STEP and abs(ll_g) < 1e-9            # logsumexp! of uniform weights
        _compare_state(g, o)
        y1 = None if k == 4 else Y[k + 1]
        t = t0 + k * model.Ts
        for h in (g, o, r):
            h.aux_predict(U[k], y1, t)
        _compare_state(g, o)
        assert g.last_resampled() and np.all(g.weights() == np.log(1.0 / 4000))                  # reset_weights!
        assert np.array_equal(g.ancestors(), r.ancestors())
        np.testing.assert_allclose(g.particles(), r.particles(), rtol=1e-12, atol=1e-12)
    # the run loops (driven synchronously for this kind)
    for mode in (0, 1):
        g2 = _capi.FilterHandle(cfg); o2 = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
        g2.reset(); o2.reset()
        rg = g2.run_aux(U, Y, mode, ll_steps=True, history=(mode == 0))
        ro = o2.run_aux(U, Y, mode, ll_steps=True, history=(mode == 0))
        assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
        if mode == 0:
            assert np.array_equal(rg["x"], ro["x"]) and np.array_equal(rg["we"], ro["we"])
        else:
            assert rg["ll_steps"][-1] != 0.0 and np.all(np.abs(rg["ll_steps"][:-1]) < 1e-9)       # only the wrapped filter's update! counts
        _compare_state(g2, o2)


@pytest.mark.parametrize("kind", [S.PARTICLE_FILTER, S.ADVANCED_PARTICLE_FILTER])
def test_aux_filter_with_residual_resampling(kind):
    """resample(ResampleResidual, ...) under the auxiliary filter (the reference's predict! resamples with whatever strategy the
    filter has, src/filtering.jl:206, src/resample.jl:63-117): residual ancestors are not sorted, so the second half runs in the
    balanced form (k_resample + k_step<NoModel, MODE_AUX2>; over an AdvancedParticleFilter: + the re-propagation).  Both run
    loops, with an outlier and a missing measurement: the device-order oracle's bits, the reference order within tolerance."""
    model = M.lg_test_model(0.1) if kind == S.PARTICLE_FILTER else M.quadtank_model()
    if kind == S.PARTICLE_FILTER:
        _, U, Y = M.simulate_lg(model, 40, seed=5)
        Y = Y.copy(); Y[17] += 11.0
    else:
        U, Y = M.quadtank_data(40, seed=2)
        Y = Y.copy()
    Y[25] = np.nan
    cfg = _cfg(model, 3000, S.RESAMPLE_RESIDUAL, 0.5, seed=35, kind=kind)
    for mode in (0, 1):
        g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
        for h in (g, o, r):
            h.reset()
        rg, ro, rr = (h.run_aux(U, Y, mode, ll_steps=True) for h in (g, o, r))
        assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64)), mode
        _compare_state(g, o)
        assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= TOL_LL_STEP
        assert g.resample_count() == o.resample_count() > 10
