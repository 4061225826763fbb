"""New parameters for an existing handle (llpf_set_model / llpf_bank_set_models) and the reference's parameter-estimation drivers on top
of it: log_likelihood_fun (src/smoothing.jl:266-283), metropolis (:311-330) and metropolis_threaded (:335-347; here one bank run per
iteration for all chains).  CPU part: the host logic of the drivers; -m gpu part: the handles."""
import numpy as np
import pytest

import llpf_amd
from llpf_amd import _capi, _structs as S, api
import models as M


class _Normal:
    def __init__(self, mu, sd):
        self.mu, self.sd = mu, sd

    def logpdf(self, x):
        return -0.5 * ((x - self.mu) / self.sd) ** 2 - np.log(self.sd * np.sqrt(2 * np.pi))


class _Positive:
    def logpdf(self, x):
        return 0.0 if x > 0 else -np.inf


def test_metropolis_is_the_reference_loop():
    """accept iff rand() < exp(ll' - ll); the chain starts at theta0; a rejected proposal repeats the previous state"""
    target = lambda th: -0.5 * float(np.sum((np.asarray(th) - 1.0) ** 2) / 0.25)
    rng = np.random.default_rng(0)
    params, lls = llpf_amd.metropolis(target, 4000, np.array([3.0, -2.0]), lambda th: th + 0.4 * rng.standard_normal(2), rng=rng)
    P = np.array(params)
    assert P.shape == (4000, 2) and np.array_equal(P[0], [3.0, -2.0]) and lls[0] == target([3.0, -2.0])
    same = np.all(P[1:] == P[:-1], axis=1)
    assert 0.2 < same.mean() < 0.9 and np.all(lls[1:][same] == lls[:-1][same])
    np.testing.assert_allclose(P[500:].mean(axis=0), [1.0, 1.0], atol=0.1)
    np.testing.assert_allclose(P[500:].std(axis=0), [0.5, 0.5], atol=0.1)
    # a proposal with a higher likelihood is always taken, one with -inf never
    always, _ = llpf_amd.metropolis(lambda th: float(th[0]), 5, np.array([0.5]), lambda th: th + 1.0, rng=np.random.default_rng(1))
    assert [float(a[0]) for a in always] == [0.5, 1.5, 2.5, 3.5, 4.5]
    never, _ = llpf_amd.metropolis(lambda th: 0.0 if th[0] == 0.5 else -np.inf, 5, np.array([0.5]), lambda th: th + 1.0, rng=np.random.default_rng(1))
    assert all(float(a[0]) == 0.5 for a in never)
    with pytest.raises(ValueError):
        llpf_amd.naive_sampler(np.array([1.0, 0.0]))


def test_log_likelihood_fun_host_logic(monkeypatch):
    calls = []

    def fake_loglik(pf, u, y):
        calls.append(pf)
        if pf["theta"][0] > 5:
            raise _capi.DegenerateWeights(4, "degenerate")
        return -float(np.sum(pf["theta"] ** 2))
    monkeypatch.setattr(api, "loglik", fake_loglik)

    def factory(theta, pf=None):
        return {"theta": np.array(theta), "prev": pf}
    priors = [_Normal(0.0, 1.0), _Positive()]
    ll = llpf_amd.log_likelihood_fun(factory, priors, None, None)
    v = ll([0.5, 2.0])
    assert abs(v - (priors[0].logpdf(0.5) - 4.25)) < 1e-14 and calls[-1]["prev"] is None
    ll([0.1, 1.0])
    assert calls[-1]["prev"] is calls[-2]                      # the previous filter is handed back (filter_from_parameters(theta, pf))
    n = len(calls)
    assert ll([0.5, -1.0]) == -np.inf and len(calls) == n      # outside the support: the filter is not even built
    assert ll([7.0, 1.0]) == -np.inf                           # a degenerate filter: -inf, as the reference's try / catch
    with pytest.raises(ValueError):
        ll([1.0])
    ll1 = llpf_amd.log_likelihood_fun(lambda theta: {"theta": np.array(theta)}, priors, None, None)      # a factory of theta alone
    assert np.isfinite(ll1([0.5, 2.0]))


# ---------------------------------------------------------------------------------------------------------------------------------
def _run_bits(h, U, Y):
    h.seed(77); h.reset()
    r = h.run(U, Y, 1.0, ll_steps=True)
    return r["ll_steps"].copy(), h.particles().copy(), h.ancestors().copy()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["lg", "lg_nx5", "quadtank", "user"])
def test_set_model_equals_a_fresh_handle(case):
    """A handle that has run (its run loop captured into a graph) is given new parameters; from the same seed it must produce the bits of
    a handle created with those parameters — and the old parameters' bits again after switching back."""
    N, T = 5000, 30
    if case == "lg":
        m1, m2 = M.lg_test_model(0.1), M.lg_test_model(0.25)
        m2.A[1] = -0.2; m2.initial_density = S.make_gaussian(np.array([0.0, 0.1]), 1.0)
        _, U, Y = M.simulate_lg(m1, T)
        kind = S.PARTICLE_FILTER
    elif case == "lg_nx5":
        rng = np.random.default_rng(2)

        def lg5(s):
            Q, _ = np.linalg.qr(np.random.default_rng(5).standard_normal((5, 5)))
            A = Q @ np.diag(np.linspace(0.5, 0.9, 5)) @ Q.T
            g = S.make_gaussian
            return S.make_lg_model(A * s, rng.standard_normal((5, 1)), np.eye(2, 5), g(np.zeros(5), 0.04 * s), g(np.zeros(2), 1.0), g(np.zeros(5), 2.0), 1.0)
        m1, m2 = lg5(1.0), lg5(0.9)
        _, U, Y = M.simulate_lg(m1, T)
        kind = S.PARTICLE_FILTER
    elif case == "quadtank":
        m1, m2 = M.quadtank_model(), M.quadtank_model()
        m2.qt[0] = 1.7; m2.measurement_density = S.make_gaussian(np.zeros(2), np.full(2, 4e-4))
        U, Y = M.quadtank_data(T)
        kind = S.ADVANCED_PARTICLE_FILTER
    else:
        import user_models as UM
        base = M.lg_test_model()
        _, U, Y = M.simulate_lg(base, T)

        def um(b):
            m = S.Model.from_buffer_copy(bytes(base))
            m.model_id = _capi.model_compile(UM.LAPLACE_SRC, m.nx, m.ny)
            m.qt[0] = b
            return m
        m1, m2 = um(0.8), um(1.3)
        kind = S.ADVANCED_PARTICLE_FILTER
    mk = lambda m: _capi.FilterHandle(S.make_config(m, N, kind, S.RESAMPLE_SYSTEMATIC, 0.5, 7, 0))
    a, f1, f2 = mk(m1), mk(m1), mk(m2)
    ref1, ref2 = _run_bits(f1, U, Y), _run_bits(f2, U, Y)
    assert not np.array_equal(ref1[0], ref2[0])
    for _ in range(2):                                           # the second run replays the captured graph
        got = _run_bits(a, U, Y)
    for k in range(3):
        assert np.array_equal(got[k].view(np.uint64) if got[k].dtype == np.float64 else got[k], ref1[k].view(np.uint64) if ref1[k].dtype == np.float64 else ref1[k])
    a.set_model(m2)
    got = _run_bits(a, U, Y)
    assert np.array_equal(got[0].view(np.uint64), ref2[0].view(np.uint64)) and np.array_equal(got[1].view(np.uint64), ref2[1].view(np.uint64)) and np.array_equal(got[2], ref2[2])
    a.set_model(m1)
    got = _run_bits(a, U, Y)
    assert np.array_equal(got[0].view(np.uint64), ref1[0].view(np.uint64)) and np.array_equal(got[1].view(np.uint64), ref1[1].view(np.uint64))
    # another family / other dimensions are refused
    with pytest.raises(_capi.LLPFError):
        a.set_model(M.lg_c1_model() if case != "lg" else M.quadtank_model())
    bad = S.Model.from_buffer_copy(bytes(m1))
    bad.dynamics_density = S.make_gaussian(np.zeros(m1.nx), -1.0)
    with pytest.raises(_capi.LLPFError):
        a.set_model(bad)


@pytest.mark.gpu
def test_bank_set_models_equals_a_fresh_bank():
    T, N = 25, 3000
    ms1 = [M.lg_test_model(s) for s in (0.05, 0.1, 0.2, 0.4)]
    ms2 = [M.lg_test_model(s) for s in (0.3, 0.07, 0.15, 0.5)]
    _, U, Y = M.simulate_lg(ms1[1], T)
    cfg = lambda ms: S.make_config(ms[0], N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 11, 0)
    a, fresh = _capi.BankHandle(cfg(ms1), ms1), _capi.BankHandle(cfg(ms2), ms2)

    def bits(b):
        b.seed(5); b.reset()
        return b.run(U, Y, 1.0, ll_steps=True)["ll_steps"].copy()
    bits(a); bits(a)
    a.set_models(ms2)
    assert np.array_equal(bits(a).view(np.uint64), bits(fresh).view(np.uint64))
    with pytest.raises(ValueError):
        a.set_models(ms2[:3])


@pytest.mark.gpu
def test_log_likelihood_fun_and_metropolis_on_the_engine():
    """the reference's PMMH example (src/smoothing.jl:296-308) in small: theta = log standard deviations of process and measurement noise"""
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]]); B = np.array([[0.1], [0.0]]); Cm = np.array([[0.0, 1.0]])
    d0 = llpf_amd.MvNormal(np.array([0.3, -0.5]), 4.0)
    dyn, meas = llpf_amd.LinearDynamics(A, B), llpf_amd.LinearMeasurement(Cm)
    truth = llpf_amd.ParticleFilter(2000, dyn, meas, llpf_amd.MvNormal(np.zeros(2), 0.01), llpf_amd.MvNormal(np.zeros(1), 1.0), d0, rng=1)
    _, u, y = llpf_amd.simulate(truth, 60, llpf_amd.MvNormal(np.zeros(1), 1.0), rng=np.random.default_rng(3))
    built = []

    def filter_from_parameters(theta, pf=None):
        df, dg = llpf_amd.MvNormal(np.zeros(2), float(np.exp(2 * theta[0]))), llpf_amd.MvNormal(np.zeros(1), float(np.exp(2 * theta[1])))
        if pf is None:
            built.append(1)
            return llpf_amd.ParticleFilter(2000, dyn, meas, df, dg, d0, rng=1)
        return pf.set_parameters(dynamics_density=df, measurement_density=dg)
    priors = [_Normal(np.log(0.1), 1.0), _Normal(0.0, 1.0)]
    ll = llpf_amd.log_likelihood_fun(filter_from_parameters, priors, u, y)
    at_truth, far = ll(np.log([0.1, 1.0])), ll(np.log([0.1, 0.05]))
    assert np.isfinite(at_truth) and at_truth > far + 50 and len(built) == 1
    # the same number as a filter built from scratch with those parameters (same key: loglik reset!s first)
    th = np.log([0.1, 1.0])      # (through exp(2 log .): the variance the factory computes, not the literal 0.01)
    fresh = llpf_amd.ParticleFilter(2000, dyn, meas, llpf_amd.MvNormal(np.zeros(2), float(np.exp(2 * th[0]))), llpf_amd.MvNormal(np.zeros(1), float(np.exp(2 * th[1]))), d0, rng=1)
    lp = priors[0].logpdf(np.log(0.1)) + priors[1].logpdf(0.0)
    fresh._h.seed(1); ll_fresh = llpf_amd.loglik(fresh, u, y)
    pf = filter_from_parameters(np.log([0.1, 1.0]), filter_from_parameters(np.log([0.3, 0.5])))
    pf._h.seed(1)
    assert llpf_amd.loglik(pf, u, y) == ll_fresh and np.isfinite(lp)
    rng = np.random.default_rng(4)
    params, lls = llpf_amd.metropolis(ll, 25, np.log([0.3, 0.5]), lambda th: th + 0.15 * rng.standard_normal(2), rng=rng)
    assert len(params) == 25 and np.all(np.isfinite(lls)) and lls.max() > lls[0] + 5 and len(built) == 2      # (the second was built by this test, above)
    # the chains of metropolis_threaded as one bank
    n_chains = 4
    spec = lambda th: (dyn, meas, llpf_amd.MvNormal(np.zeros(2), float(np.exp(2 * th[0]))), llpf_amd.MvNormal(np.zeros(1), float(np.exp(2 * th[1]))), d0)
    th0 = np.log([0.3, 0.5]) + 0.1 * rng.standard_normal((n_chains, 2))
    bank = llpf_amd.FilterBank(2000, [spec(t) for t in th0], resample_threshold=0.1, rng=1)
    out = llpf_amd.metropolis_bank(bank, spec, priors, u, y, 12, th0, draw=lambda th: th + 0.15 * rng.standard_normal(2), burnin=2, rng=rng)
    assert out.shape == (10 * n_chains, 3) and np.all(np.isfinite(out))
    # a chain's first retained state has the likelihood its own parameters give in a one-filter evaluation of the same bank slot
    assert np.all(out[:, 2] < 0)


@pytest.mark.gpu
def test_metropolis_bank_isolates_a_failing_chain():
    """The reference wraps every chain's loglik in try / catch and scores -Inf for that chain only (src/smoothing.jl:276-280), so
    metropolis_threaded keeps sampling when ONE proposal cannot be built (covariance not positive definite) or degenerates the filter
    (all weights -Inf).  One bank run serves all chains here, so the failing slot is retried on its chain's current parameters and the
    chain scores -inf alone (round-4 advisor finding: the exception used to abort every chain)."""
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]]); B = np.array([[0.1], [0.0]]); Cm = np.array([[0.0, 1.0]])
    d0 = llpf_amd.MvNormal(np.array([0.3, -0.5]), 4.0)
    dyn, meas = llpf_amd.LinearDynamics(A, B), llpf_amd.LinearMeasurement(Cm)
    truth = llpf_amd.ParticleFilter(1000, dyn, meas, llpf_amd.MvNormal(np.zeros(2), 0.01), llpf_amd.MvNormal(np.zeros(1), 1.0), d0, rng=1)
    _, u, y = llpf_amd.simulate(truth, 40, llpf_amd.MvNormal(np.zeros(1), 1.0), rng=np.random.default_rng(3))
    spec = lambda th: (dyn, meas, llpf_amd.MvNormal(np.zeros(2), float(np.exp(2 * th[0]))), llpf_amd.MvNormal(np.zeros(1), float(np.exp(2 * th[1]))), d0)
    priors = [_Normal(np.log(0.1), 1000.0), _Normal(0.0, 1000.0)]      # wide: the wild proposals below stay inside the support
    n_chains = 4
    th0 = np.tile(np.log([0.3, 0.5]), (n_chains, 1))
    bank = llpf_amd.FilterBank(1000, [spec(t) for t in th0], resample_threshold=0.1, rng=1)
    it = {"i": 0}

    def draw(th):
        k, i = it["i"] % n_chains, it["i"] // n_chains
        it["i"] += 1
        if i == 1 and k == 1:
            return np.array([th[0], -400.0])        # measurement variance exp(-800) = 0: the density cannot be built
        if i == 2 and k == 2:
            return np.array([th[0], 400.0])         # variance +Inf: every weight -Inf (or the density is refused): the filter degenerates
        return th + 0.05 * np.array([np.sin(it["i"]), np.cos(it["i"])])
    out = llpf_amd.metropolis_bank(bank, spec, priors, u, y, 6, th0, draw=draw, burnin=0, rng=np.random.default_rng(9))
    assert out.shape == (6 * n_chains, 3) and np.all(np.isfinite(out))       # the bad proposals were rejected, every chain went on
    ch = out.reshape(n_chains, 6, 3)
    assert np.all(np.abs(ch[1, :, 1]) < 10) and np.all(np.abs(ch[2, :, 1]) < 10)      # chains 1 and 2 never accepted the wild values
    assert len({tuple(r) for r in ch[0, :, :2]}) > 1                                   # and the healthy chains kept moving
