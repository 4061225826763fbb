"""The reference's statistical bounds for this path, restated on the oracle (test/runtests.jl:108-143,
409-450, 553-599), at sizes the CPU suite can afford."""
import numpy as np
import pytest

import oracle_binding as ob
from llpf_amd import _structs as S
import models as M


@pytest.mark.parametrize("order", [ob.ORDER_REFERENCE, ob.ORDER_DEVICE])
@pytest.mark.parametrize("strategy", [S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED, S.RESAMPLE_RESIDUAL])
def test_resample_proportions(order, strategy):
    """test/runtests.jl:108-143: empirical index proportions over 10000 draws within 0.02 of we."""
    we = np.array([0.1, 0.5, 0.1, 0.15, 0.15])
    counts = np.zeros(5)
    for d in range(10000):
        U = ob.resample_uniforms(strategy, 5, 99, d)          # the Philox uniforms the filter path would use
        j, _ = ob.resample(strategy, we, U, order=order)
        counts += np.bincount(j, minlength=5)
    assert np.allclose(counts / counts.sum(), we, atol=0.02)


def test_pf_loglik_tracks_kalman_over_noise_sweep():
    """test/runtests.jl:409-450: 11 noise levels, argmax of the PF log-likelihood in 5..7 (1-based) like the
    Kalman filter's, and max |ll_KF - ll_PF| < 20  (the reference uses N=1000, T=2000; T=500 here)."""
    svec = 10.0 ** np.linspace(-2, 0, 11)
    true = M.lg_test_model(0.1)
    _, U, Y = M.simulate_lg(true, 500, seed=0)
    ll_pf, ll_kf = [], []
    for s in svec:
        m = M.lg_test_model(s)
        cfg = S.make_config(m, 1000, resample_threshold=0.1, seed=11)
        o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
        o.reset()
        ll_pf.append(o.run(U, Y, 1.0)["ll"])
        ll_kf.append(ob.kalman_loglik(m, U, Y))
    ll_pf, ll_kf = np.array(ll_pf), np.array(ll_kf)
    assert 4 <= int(np.argmax(ll_kf)) <= 6
    assert 4 <= int(np.argmax(ll_pf)) <= 6
    assert np.max(np.abs(ll_kf - ll_pf)) < 20


def test_pf_loglik_converges_to_kalman():
    """Monte-Carlo error of sum(ll) shrinks ~ 1/sqrt(N): calibrated from 8 seeds (SURVEY.md §4 take-away)."""
    model = M.lg_c1_model()
    _, U, Y = M.simulate_lg(model, 100)
    kf = ob.kalman_loglik(model, U, Y)
    errs = {}
    for N in (500, 8000):
        e = []
        for seed in range(8):
            cfg = S.make_config(model, N, resample_threshold=0.1, seed=seed)
            o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
            o.reset()
            e.append(o.run(U, Y, 0.0)["ll"] - kf)
        errs[N] = np.array(e)
    assert np.std(errs[8000]) < 0.6 * np.std(errs[500])
    assert abs(np.mean(errs[8000])) < 3 * np.std(errs[8000]) / np.sqrt(8) + 0.1


def test_advanced_filter_mean_error():
    """test/runtests.jl:553-599: 2-D linear system, N=500, T=200: mean state error small."""
    model = M.lg_c1_model()
    X, U, Y = M.simulate_lg(model, 200)
    cfg = S.make_config(model, 500, S.ADVANCED_PARTICLE_FILTER, resample_threshold=0.5, seed=3)
    o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    o.reset()
    r = o.run(U, Y, 0.0, xmean=True)
    assert np.linalg.norm(np.mean(X - r["xmean"], axis=0)) < 5
    assert np.mean((X - r["xmean"]) ** 2) < 2.0


def test_reference_and_device_order_agree():
    """The two arithmetic orders of the oracle on the same Philox inputs: fp within 1e-12, same ancestors."""
    model = M.quadtank_model()
    U, Y = M.quadtank_data(30)
    cfg = S.make_config(model, 2000, S.ADVANCED_PARTICLE_FILTER, resample_threshold=0.5, seed=1)
    od, orf = ob.OracleFilter(cfg, ob.ORDER_DEVICE), ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    od.reset(); orf.reset()
    rd = od.run(U, Y, 490.0, ll_steps=True, history=True)
    rr = orf.run(U, Y, 490.0, ll_steps=True, history=True)
    assert np.max(np.abs(rd["ll_steps"] - rr["ll_steps"])) < 1e-11
    assert np.array_equal(od.ancestors(), orf.ancestors())
    np.testing.assert_allclose(rd["we"], rr["we"], rtol=1e-11, atol=1e-300)
    np.testing.assert_allclose(rd["x"], rr["x"], rtol=1e-12, atol=1e-12)
    assert od.resample_count() == orf.resample_count() > 0


def test_aux_filter_loglik_tracks_kalman_over_noise_sweep():
    """test/runtests.jl:419-423, 441-446: loglik of AuxiliaryParticleFilters over the dynamics-noise sweep
    svec = exp10.(LinRange(-2, 0, 11)) at N = 1000, T = 2000: argmax in 5..7 and |ll_KF - ll_APF| < 20."""
    import ctypes as C
    base = M.lg_test_model(0.1)
    _, U, Y = M.simulate_lg(base, 2000, seed=3)
    svec = 10 ** np.linspace(-2, 0, 11)
    lls, kfs = [], []
    for k, s in enumerate(svec):
        m = M.lg_test_model(s)
        cfg = S.make_config(m, 1000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 11 + k, 0)
        o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
        o.reset()
        lls.append(o.run_aux(U, Y, mode=1)["ll"])
        kfs.append(ob.lib().orc_kalman_loglik(C.byref(m), ob.dptr(np.ascontiguousarray(U)), ob.dptr(np.ascontiguousarray(Y)), 2000))
    lls, kfs = np.array(lls), np.array(kfs)
    assert 5 <= np.argmax(lls) + 1 <= 7
    assert 5 <= np.argmax(kfs) + 1 <= 7
    assert np.max(np.abs(lls - kfs)) < 20


def test_aux_filter_semantics_and_orders():
    """correct! of the auxiliary filter ignores y (the first call after reset! returns logsumexp of uniform
    weights ~ 0, src/filtering.jl:170-174); predict! always resamples and leaves w = lambda - log N, we = lambda
    (:200-213); a fresh filter does not want to resample (test/runtests.jl:275); both arithmetic orders agree."""
    m = M.lg_test_model(0.1)
    _, U, Y = M.simulate_lg(m, 50, seed=4)
    N = 2000
    for strategy in (S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED):
        cfg = S.make_config(m, N, S.PARTICLE_FILTER, strategy, 0.1, 21, 0)
        r = []
        for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
            o = ob.OracleFilter(cfg, order)
            assert not o.shouldresample()
            o.reset()
            ll0 = o.aux_correct()
            assert abs(ll0) < 1e-12
            o.aux_predict(U[0], Y[1], 0.0)
            assert o.last_resampled() and o.index() == 2
            lam = o.expweights()
            np.testing.assert_allclose(o.weights(), lam - np.log(N), rtol=0, atol=1e-13)
            assert np.all(lam <= 0.5 * np.log(1 / (2 * np.pi)) + 1e-12)      # lambda = logpdf(N(0,1)) <= c0
            ll1 = o.aux_correct()
            assert abs(np.sum(o.expweights()) - 1.0) < 1e-12 and np.isfinite(ll1)
            o.reset()
            r.append(o.run_aux(U, Y, mode=0, ll_steps=True, history=True))
        assert np.max(np.abs(r[0]["ll_steps"] - r[1]["ll_steps"])) < 1e-10
        np.testing.assert_allclose(r[0]["x"], r[1]["x"], rtol=0, atol=1e-12)
        assert r[0]["ll_steps"][0] == r[1]["ll_steps"][0] or abs(r[0]["ll_steps"][0]) < 1e-12
        # missing look-ahead measurement: lambda = 0, weights stay uniform
        o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
        o.reset(); o.aux_correct()
        o.aux_predict(U[0], None, 0.0)
        assert np.all(o.expweights() == 0.0)
        np.testing.assert_allclose(o.weights(), -np.log(N), rtol=0, atol=1e-13)
        assert abs(o.aux_correct()) < 1e-12


def test_residual_resampling_structure():
    """resample(ResampleResidual, we) — src/resample.jl:63-117: floor(N we_i) deterministic copies in source order, then
    multinomial draws on the residuals; uniform weights give 1:N without any draw; both orders agree."""
    rng = np.random.default_rng(2)
    for n, m in ((5000, 5000), (777, 777), (1000, 400), (300, 1000)):
        w = rng.standard_normal(n) * 1.5
        we = np.exp(w - w.max()); we /= we.sum()
        U = ob.resample_uniforms(S.RESAMPLE_RESIDUAL, m, 7, 3)
        jr, _ = ob.resample(S.RESAMPLE_RESIDUAL, we, U, m, ob.ORDER_REFERENCE)
        jd, _ = ob.resample(S.RESAMPLE_RESIDUAL, we, U, m, ob.ORDER_DEVICE)
        assert np.sum(jr != jd) <= 1
        num = int(np.floor(we * m - 1e-9).clip(0).sum())          # at least this many deterministic copies, in source order
        assert np.all(np.diff(jd[:num]) >= 0)
        assert jd.min() >= 0 and jd.max() < n
        assert np.all(np.bincount(jd, minlength=n) >= np.floor(we * m - 1e-9).astype(int))
    we = np.full(64, 1 / 64)
    j, _ = ob.resample(S.RESAMPLE_RESIDUAL, we, np.zeros(64), order=ob.ORDER_DEVICE)
    assert np.array_equal(j, np.arange(64))


def test_residual_strategy_filter_tracks_kalman():
    """A ParticleFilter with resampling_strategy = ResampleResidual: log-likelihood within the reference's bound of
    the Kalman filter's (test/runtests.jl:445) and the two orders agree."""
    m = M.lg_test_model(0.1)
    _, U, Y = M.simulate_lg(m, 400, seed=6)
    cfg = S.make_config(m, 1000, S.PARTICLE_FILTER, S.RESAMPLE_RESIDUAL, 0.5, 3, 0)
    lls = []
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        o = ob.OracleFilter(cfg, order)
        o.reset()
        lls.append(o.run(U, Y, 1.0, ll_steps=True)["ll_steps"])
        assert o.resample_count() > 5
    assert np.max(np.abs(lls[0] - lls[1])) < 1e-10
    assert abs(lls[0].sum() - ob.kalman_loglik(m, U, Y)) < 20


def test_particle_smoother_and_draw_one_categorical():
    """smooth(pf, M, u, y) — test/runtests.jl:314-317: size (M, T) and mean(abs2, x - smoothed_mean) < 5;
    draw_one_categorical (src/resample.jl:128-152) draws index i with probability softmax(w)_i; both orders agree."""
    rng = np.random.default_rng(0)
    w = np.log(np.array([0.1, 0.5, 0.1, 0.15, 0.15])) + 3.0
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        counts = np.bincount([ob.draw_one_categorical(w, u, order) for u in rng.uniform(size=20000)], minlength=5)
        assert np.allclose(counts / counts.sum(), [0.1, 0.5, 0.1, 0.15, 0.15], atol=0.02)
    assert ob.draw_one_categorical(w, 1.0, ob.ORDER_DEVICE) == 4 and ob.draw_one_categorical(w, 0.0, ob.ORDER_REFERENCE) == 0
    model = M.lg_test_model(0.1)
    X, U, Y = M.simulate_lg(model, 200, seed=2)
    cfg = S.make_config(model, 1000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 3, 0)
    res = []
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        o = ob.OracleFilter(cfg, order)
        o.reset()
        r = o.run(U, Y, 0.0, history=True)
        xb, idx = o.smooth(100, U, r["x"], r["w"], r["we"])
        assert xb.shape == (200, 100, 2)
        assert np.mean((X - xb.mean(axis=1)) ** 2) < 5
        assert np.mean((X - xb.mean(axis=1)) ** 2) < np.mean((X - np.einsum("tnd,tn->td", r["x"], r["we"])) ** 2)
        assert np.array_equal(xb, r["x"][np.arange(200)[:, None], idx])
        res.append(idx)
    assert np.mean(res[0] != res[1]) < 0.01


def _rb_cases():
    g = S.make_gaussian
    A = np.array([[1, 0.1], [0, 1.0]]); B = np.array([[0.0], [1.0]]); Cm = np.array([[1.0, 0.0]])
    Ts = 0.1
    R1 = np.array([[Ts ** 4 / 4, Ts ** 3 / 2], [Ts ** 3 / 2, Ts ** 2]]) + 1e-6 * np.eye(2)
    R2 = np.array([[10.0]])
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal(2)
    T = 500
    U = rng.standard_normal((T, 1)); Y = np.zeros((T, 1)); x = x0 + np.sqrt(2) * rng.standard_normal(2)
    L1 = np.linalg.cholesky(R1)
    for t in range(T):
        Y[t] = Cm @ x + np.sqrt(10) * rng.standard_normal(1)
        x = A @ x + B @ U[t] + L1 @ rng.standard_normal(2)
    kfm = S.make_lg_model(A, B, Cm, g(np.zeros(2), R1), g(np.zeros(1), R2), g(x0, 2 * np.eye(2)))
    lin = S.make_rb_model([[1.0]], np.zeros((1, 1)), None, A, B, np.zeros((1, 1)), Cm, g(np.zeros(1), np.array([[1e-12]])), R1,
                          g(np.zeros(1), R2), g(np.zeros(1), np.array([[1e-12]])), g(x0, 2 * np.eye(2)))
    nonl = S.make_rb_model(A, B, None, [[0.0]], np.zeros((1, 1)), Cm, None, g(np.zeros(2), R1), [[1.0]], g(np.zeros(1), R2),
                           g(x0, 2 * np.eye(2)), g(np.zeros(1), np.array([[1.0]])))
    return kfm, U, Y, lin, nonl


def test_rbpf_matches_kalman_filter():
    """test/test_rbpf.jl:87-139: an RBPF whose whole state is linear reproduces the Kalman filter's log-likelihood
    (here to rounding: every particle carries the same Kalman filter), and one whose whole state is nonlinear matches
    it to rtol 1e-2; a mixed 1 + 1 system with An = 0.5 (:5-31) against the Kalman filter of the joint system."""
    kfm, U, Y, lin, nonl = _rb_cases()
    llkf = ob.kalman_loglik(kfm, U, Y)
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        o = ob.OracleFilter(S.make_config(lin, 500, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 3, 0), order)
        o.reset()
        assert abs(o.run(U, Y, 0.0)["ll"] - llkf) < 1e-8 * abs(llkf)
        o = ob.OracleFilter(S.make_config(nonl, 500, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 3, 0), order)
        o.reset()
        assert abs(o.run(U, Y, 0.0)["ll"] - llkf) < 1e-2 * abs(llkf)
    g = S.make_gaussian
    mixed = S.make_rb_model([[1.0]], np.zeros((1, 0)), [[0.5]], [[0.95]], np.zeros((1, 0)), [[1.0]], [[1.0]], g(np.zeros(1), np.array([[0.01]])),
                            [[0.01]], g(np.zeros(1), np.array([[0.1]])), g(np.array([1.0]), np.array([[0.01]])), g(np.array([1.0]), np.array([[1.0]])))
    rng = np.random.default_rng(1)
    xn, xl, T = 1.0, 1.0, 2000
    Y1 = np.zeros((T, 1))
    for t in range(T):
        Y1[t] = xn + xl + np.sqrt(0.1) * rng.standard_normal()
        xn, xl = xn + 0.5 * xl + 0.1 * rng.standard_normal(), 0.95 * xl + 0.1 * rng.standard_normal()
    joint = S.make_lg_model(np.array([[1, 0.5], [0, 0.95]]), np.zeros((2, 0)), np.array([[1.0, 1.0]]), g(np.zeros(2), np.diag([0.01, 0.01])),
                            g(np.zeros(1), np.array([[0.1]])), g(np.array([1.0, 1.0]), np.diag([0.01, 1.0])))
    llj = ob.kalman_loglik(joint, np.zeros((T, 0)), Y1)
    lls = []
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        o = ob.OracleFilter(S.make_config(mixed, 500, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 3, 0), order)
        o.reset()
        r = o.run(np.zeros((T, 0)), Y1, 0.0, ll_steps=True)
        assert abs(r["ll"] - llj) < 1e-2 * abs(llj)
        lls.append(r["ll_steps"])
        assert o.rb_R().shape == (1, 1) and 0 < o.rb_R()[0, 0] < 1
    assert np.max(np.abs(lls[0] - lls[1])) < 1e-10
