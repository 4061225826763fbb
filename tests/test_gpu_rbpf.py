"""GPU parity of the Rao-Blackwellized particle filter with constant matrices (reference src/rbpf.jl, test/test_rbpf.jl)."""
import numpy as np
import pytest

import llpf_amd
from llpf_amd import _capi, _structs as S
import oracle_binding as ob
from gpu_common import TOL_LL_STEP, cfg_of as _cfg, compare_state as _compare_state

pytestmark = pytest.mark.gpu


def _rb_models():
    g = S.make_gaussian
    A = np.array([[1, 0.1], [0, 1.0]]); B = np.array([[0.0], [1.0]]); Cm = np.array([[1.0, 0.0]])
    Ts = 0.1
    R1 = np.array([[Ts ** 4 / 4, Ts ** 3 / 2], [Ts ** 3 / 2, Ts ** 2]]) + 1e-6 * np.eye(2)
    R2 = np.array([[10.0]])
    rng = np.random.default_rng(0)
    x0 = rng.standard_normal(2)
    T = 300
    U = rng.standard_normal((T, 1)); Y = np.zeros((T, 1)); x = x0 + np.sqrt(2) * rng.standard_normal(2)
    L1 = np.linalg.cholesky(R1)
    for t in range(T):
        Y[t] = Cm @ x + np.sqrt(10) * rng.standard_normal(1)
        x = A @ x + B @ U[t] + L1 @ rng.standard_normal(2)
    kfm = S.make_lg_model(A, B, Cm, g(np.zeros(2), R1), g(np.zeros(1), R2), g(x0, 2 * np.eye(2)))
    # test/test_rbpf.jl:87-116: everything linear (a fake nonlinear state; its zero covariance replaced by 1e-12)
    lin = S.make_rb_model([[1.0]], np.zeros((1, 1)), None, A, B, np.zeros((1, 1)), Cm, g(np.zeros(1), np.array([[1e-12]])), R1,
                          g(np.zeros(1), R2), g(np.zeros(1), np.array([[1e-12]])), g(x0, 2 * np.eye(2)))
    # :118-139: everything nonlinear (a fake linear state, A0 = B0 = C0 = 0)
    nonl = S.make_rb_model(A, B, None, [[0.0]], np.zeros((1, 1)), Cm, None, g(np.zeros(2), R1), [[1.0]], g(np.zeros(1), R2),
                           g(x0, 2 * np.eye(2)), g(np.zeros(1), np.array([[1.0]])))
    # :5-31: mixed 1 + 1 with An = 0.5
    mixed = S.make_rb_model([[1.0]], np.zeros((1, 0)), [[0.5]], [[0.95]], np.zeros((1, 0)), [[1.0]], [[1.0]], g(np.zeros(1), np.array([[0.01]])),
                            [[0.01]], g(np.zeros(1), np.array([[0.1]])), g(np.array([1.0]), np.array([[0.01]])), g(np.array([1.0]), np.array([[1.0]])))
    rng = np.random.default_rng(1)
    xn, xl = 1.0, 1.0
    Y1 = np.zeros((T, 1))
    for t in range(T):
        Y1[t] = xn + xl + np.sqrt(0.1) * rng.standard_normal()
        xn, xl = xn + 0.5 * xl + 0.1 * rng.standard_normal(), 0.95 * xl + 0.1 * rng.standard_normal()
    return kfm, {"linear": (lin, U, Y), "nonlinear": (nonl, U, Y), "mixed": (mixed, np.zeros((T, 0)), Y1)}


@pytest.mark.parametrize("name", ["linear", "nonlinear", "mixed"])
def test_rbpf_bit_exact_and_kalman(name):
    """The three configurations of the reference's test/test_rbpf.jl: whole trajectories (history, per-step ll, the
    shared covariance) bit-identical to the device-order oracle, within tolerance of the reference-order one; single
    steps; and the reference's check ll_RBPF ~ ll_KF (rtol 1e-2) where the system is linear."""
    kfm, cases = _rb_models()
    model, U, Y = cases[name]
    cfg = _cfg(model, 500, S.RESAMPLE_SYSTEMATIC, 0.1, seed=3)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    _compare_state(g, o)
    for h in (g, o, r):
        h.reset()
    _compare_state(g, o)
    rg = g.run(U, Y, 0.0, ll_steps=True, history=True); ro = o.run(U, Y, 0.0, ll_steps=True, history=True)
    rr = r.run(U, Y, 0.0, ll_steps=True)
    assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    for key in ("x", "w", "we"):
        assert np.array_equal(rg[key].view(np.uint64), ro[key].view(np.uint64)), key
    _compare_state(g, o)
    assert np.array_equal(g.rb_covariance().view(np.uint64), o.rb_R().view(np.uint64))
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= TOL_LL_STEP
    if name != "mixed":
        assert abs(rg["ll"] - ob.kalman_loglik(kfm, U, Y)) <= 1e-2 * abs(rg["ll"])          # test/test_rbpf.jl:110,139
    # asynchronous loop and single steps
    g2 = _capi.FilterHandle(cfg); g2.reset()
    assert np.array_equal(g2.run(U, Y, 0.0, ll_steps=True)["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    g3 = _capi.FilterHandle(cfg); o3 = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g3.reset(); o3.reset()
    for k in range(12):
        assert g3.update(U[k], Y[k], k * 1.0) == o3.update(U[k], Y[k], k * 1.0)
        _compare_state(g3, o3)


@pytest.mark.parametrize("thr", [0.1, 1.0])
def test_rbpf_fused_kernel_with_the_split_known_at_compile_time(thr):
    """Round 6: the fused kernel of the reference's own RBPF benchmark system (test/test_rbpf.jl:5-31: 1 nonlinear + 1 linear state, one
    output) is instantiated with the split as a compile-time constant — RBLin<2, 1, 1>: no scratch memory, generator tables and owner
    table in LDS, four waves per SIMD, write-through stores (30.0 -> 22.7 us per timestep at N = 1e6) — while the step kernel keeps the
    run-time split.  Several tiles (the one-tile form is covered above), resampling seldom and at every step: the asynchronous run
    (fused kernel), the history run (k_resample + k_step) and the oracle give the same bits."""
    _, cases = _rb_models()
    model, U, Y = cases["mixed"]
    cfg = _cfg(model, 5000 + 13, S.RESAMPLE_SYSTEMATIC, thr, seed=8)
    g = _capi.FilterHandle(cfg); gh = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    for h in (g, gh, o):
        h.reset()
    rg = g.run(U[:80], Y[:80], 0.0, ll_steps=True)
    rh = gh.run(U[:80], Y[:80], 0.0, ll_steps=True, history=True)
    ro = o.run(U[:80], Y[:80], 0.0, ll_steps=True)
    assert g.last_run_stats()["fused_launches"] > 0, "the fused kernel did not run"
    assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    assert np.array_equal(rh["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    assert g.resample_count() == o.resample_count() >= (80 if thr == 1.0 else 1)
    _compare_state(g, o); _compare_state(gh, o)
    assert np.array_equal(g.ancestors(), o.ancestors())


def _random_rb_model(nn, nl, ny, seed):
    rng = np.random.default_rng(seed)
    g = S.make_gaussian

    def stable(n, rho):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        return Q @ np.diag(rho * rng.uniform(0.5, 1.0, n)) @ Q.T

    def spd(n, s):
        a = rng.standard_normal((n, n))
        return s * (a @ a.T + n * np.eye(n)) / n

    # the coupling An of the shared-covariance form needs a scalar Nt (host/densities.hpp: nn == 1); with more nonlinear states An = 0
    An = 0.4 * rng.standard_normal((nn, nl)) if nn == 1 else None
    return S.make_rb_model(stable(nn, 0.9), 0.3 * rng.standard_normal((nn, 1)), An, stable(nl, 0.95),
                           0.3 * rng.standard_normal((nl, 1)), rng.standard_normal((ny, nn)), rng.standard_normal((ny, nl)),
                           g(np.zeros(nn), spd(nn, 0.02)), spd(nl, 0.02), g(np.zeros(ny), spd(ny, 0.2)),
                           g(0.3 * rng.standard_normal(nn), spd(nn, 0.3)), g(0.3 * rng.standard_normal(nl), spd(nl, 0.5)))


@pytest.mark.parametrize("nn,nl,ny", [(1, 1, 2), (1, 2, 1), (2, 1, 2), (1, 3, 2), (2, 2, 1), (3, 1, 1), (2, 2, 3)])
def test_rbpf_fused_kernel_every_split(nn, nl, ny):
    """every nonlinear / linear split of up to four states through the fused kernel — with one or two outputs the instantiation that knows
    the split at compile time (k_resprop.hip: LLPF_RB_LEAN), with three the run-time split — against the device-order oracle: several
    tiles, a missing measurement, dense covariances everywhere, an input; the one-tile form and the history run (step kernel) as well."""
    model = _random_rb_model(nn, nl, ny, 100 * nn + 10 * nl + ny)
    rng = np.random.default_rng(7)
    T = 40
    U = rng.standard_normal((T, 1))
    Y = rng.standard_normal((T, ny))
    Y[9] = np.nan
    for N in (3000 + 7, 700):
        cfg = _cfg(model, N, S.RESAMPLE_SYSTEMATIC, 0.5, seed=13)
        g = _capi.FilterHandle(cfg); gh = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
        for h in (g, gh, o):
            h.reset()
        rg = g.run(U, Y, 0.0, ll_steps=True); rh = gh.run(U, Y, 0.0, ll_steps=True, history=True); ro = o.run(U, Y, 0.0, ll_steps=True)
        assert g.last_run_stats()["fused_launches"] > 0
        assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
        assert np.array_equal(rh["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
        assert o.resample_count() >= 1 and g.resample_count() == o.resample_count()
        _compare_state(g, o); _compare_state(gh, o)


def test_rbpf_api():
    """RBPF(N, kf, dynamics, nl_measurement_model, R1n, d0n; An, ...) through the reference-shaped API: loglik close to
    the Kalman filter's (test/test_rbpf.jl:110), forward_trajectory shapes, the shared covariance accessor."""
    A = np.array([[1, 0.1], [0, 1.0]]); B = np.array([[0.0], [1.0]]); Cm = np.array([[1.0, 0.0]])
    Ts = 0.1
    R1 = np.array([[Ts ** 4 / 4, Ts ** 3 / 2], [Ts ** 3 / 2, Ts ** 2]]) + 1e-6 * np.eye(2)
    R2 = np.array([[10.0]])
    kfm, cases = _rb_models()
    _, U, Y = cases["linear"]
    x0 = S.gaussian_mean(kfm.initial_density)
    kf = llpf_amd.KalmanFilter(A, B, Cm, 0, R1, R2, llpf_amd.MvNormal(x0, 2 * np.eye(2)))
    mm = llpf_amd.RBMeasurementModel(llpf_amd.LinearMeasurement(np.zeros((1, 1))), R2, 1)
    pf = llpf_amd.RBPF(500, kf, llpf_amd.LinearDynamics([[1.0]], np.zeros((1, 1))), mm, np.array([[1e-12]]),
                       llpf_amd.MvNormal(np.zeros(1), np.array([[1e-12]])), An=None, nu=1, rng=4)
    ll = llpf_amd.loglik(pf, U, Y)          # note: loglik runs at t = index*Ts, the model is time invariant
    assert abs(ll - ob.kalman_loglik(kfm, U, Y)) <= 1e-2 * abs(ll)
    sol = llpf_amd.forward_trajectory(pf, U[:30], Y[:30])
    assert sol.x.shape == (30, 500, 3) and pf.covariance.shape == (2, 2)
    assert np.allclose(sol.we.sum(axis=1), 1.0)
