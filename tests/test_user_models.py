"""User-supplied models through the C ABI (llpf_model_compile, include/llpf.h): the replacement for the reference's arbitrary
`dynamics` / `measurement` callables (src/PFtypes.jl:59-63, 189-193, 226-289).  CPU: the snippets compile for gfx950 (hiprtc needs
no device).  GPU: the quad-tank written as a USER snippet reproduces the built-in quad-tank — and thereby the oracle — bit for bit;
a model with no built-in counterpart (a damped pendulum) filters sensibly, alone, in a bank, and through the auxiliary verbs."""
import os

import numpy as np
import pytest

import models as M
import oracle_binding as ob
import user_models as UM
from llpf_amd import _capi, _structs as S


def test_snippets_compile_for_gfx950_without_a_device():
    os.environ["LLPF_JIT_COMPILE_ONLY"] = "1"
    try:
        a = _capi.model_compile(UM.QUADTANK_SRC, 4, 2)
        b = _capi.model_compile(UM.PENDULUM_SRC, 2, 1)
        assert a >= 1000 and b == a + 1
        assert _capi.model_compile(UM.QUADTANK_SRC, 4, 2) == a                 # the same (source, nx, ny) again: the same id, no recompilation
        ids = [_capi.model_compile(src, 2, 1) for src in (UM.LAPLACE_SRC, UM.STUDENT_T_SRC, UM.LAPLACE_NO_BOUND_SRC)]   # likelihood hooks
        assert len(set(ids)) == 3 and "loglik_bound" not in UM.LAPLACE_NO_BOUND_SRC
        # what a snippet defines is reported without running anything (the template argument of a tag kernel's lowered name)
        T = _capi
        assert _capi.model_traits(a) == 0 and _capi.model_traits(ids[0]) == T.TRAIT_LOGLIK | T.TRAIT_LOGLIK_BOUND
        assert _capi.model_traits(ids[2]) == T.TRAIT_LOGLIK
        n1, n2 = _capi.model_compile(UM.MULT_NOISE_BOX_SRC, 2, 1), _capi.model_compile(UM.LAPLACE_NOISE_SRC, 2, 1)    # noise / initial hooks
        assert _capi.model_traits(n1) == T.TRAIT_NOISE | T.TRAIT_INITIAL and _capi.model_traits(n2) == T.TRAIT_NOISE
        with pytest.raises(_capi.LLPFError):
            _capi.model_traits(7)
        with pytest.raises(_capi.LLPFError) as ei:
            _capi.model_compile("struct UserModel { int broken }", 2, 1)
        assert "hiprtc" in str(ei.value)
        with pytest.raises(_capi.LLPFError):
            _capi.model_compile(UM.PENDULUM_SRC, 9, 1)
        with pytest.raises(_capi.LLPFError) as ei:                             # the kernels around the compiled one cover 1..4 dimensions
            _capi.model_compile(UM.PENDULUM_SRC, 5, 1)
        assert "1..4" in str(ei.value)
    finally:
        del os.environ["LLPF_JIT_COMPILE_ONLY"]


def _user_copy(model, model_id):
    m = S.Model.from_buffer_copy(bytes(model))
    m.model_id = model_id
    return m


@pytest.mark.gpu
def test_quadtank_as_a_user_snippet_is_the_builtin_model_bit_for_bit():
    mid = _capi.model_compile(UM.QUADTANK_SRC, 4, 2)
    qt = M.quadtank_model()
    U, Y = M.quadtank_data(60, seed=2)
    for strat, thr in ((S.RESAMPLE_SYSTEMATIC, 0.5), (S.RESAMPLE_STRATIFIED, 1.0)):
        cfg_b = S.make_config(qt, 6000, S.ADVANCED_PARTICLE_FILTER, strat, thr, 7, 0)
        cfg_u = S.make_config(_user_copy(qt, mid), 6000, S.ADVANCED_PARTICLE_FILTER, strat, thr, 7, 0)
        gb, gu = _capi.FilterHandle(cfg_b), _capi.FilterHandle(cfg_u)
        gb.reset(); gu.reset()
        rb = gb.run(U, Y, 470.0, ll_steps=True, xmean=True)        # crosses the t > 500 switch of the tank parameters
        ru = gu.run(U, Y, 470.0, ll_steps=True, xmean=True)
        assert np.array_equal(ru["ll_steps"].view(np.uint64), rb["ll_steps"].view(np.uint64))
        assert np.array_equal(gu.particles().view(np.uint64), gb.particles().view(np.uint64))
        assert np.array_equal(gu.ancestors(), gb.ancestors()) and gu.resample_count() == gb.resample_count() > 0
        np.testing.assert_allclose(ru["xmean"], rb["xmean"], rtol=1e-12, atol=1e-13)   # a plain fp64 sum of per-block partials (never fed back): the two forms may group them differently
        o = ob.OracleFilter(cfg_b, ob.ORDER_DEVICE)
        o.reset()
        ro = o.run(U, Y, 470.0, ll_steps=True)
        assert np.array_equal(ru["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    # step by step and with history too
    gu = _capi.FilterHandle(cfg_u); gb = _capi.FilterHandle(cfg_b)
    for g in (gu, gb):
        g.reset()
    for k in range(5):
        assert gu.update(U[k], Y[k], float(k)) == gb.update(U[k], Y[k], float(k))
    hu = gu.run(U[:10], Y[:10], 0.0, history=True); hb = gb.run(U[:10], Y[:10], 0.0, history=True)
    assert np.array_equal(hu["x"], hb["x"]) and np.array_equal(hu["we"], hb["we"])


@pytest.mark.gpu
def test_a_model_without_builtin_counterpart():
    mid = _capi.model_compile(UM.PENDULUM_SRC, 2, 1)
    g = S.make_gaussian
    m = S.make_lg_model(np.eye(2), np.zeros((2, 1)), np.array([[1.0, 0.0]]), g(np.zeros(2), np.array([1e-4, 4e-3])), g(np.zeros(1), 0.05 ** 2),
                        g(np.array([0.8, 0.0]), np.array([0.3, 0.3])), Ts=0.05)
    m.model_id = mid
    m.qt[0], m.qt[1] = 9.81, 0.05
    # simulate the same system on the host
    rng = np.random.default_rng(0)
    T = 200
    x = np.array([1.0, 0.0]); X = np.zeros((T, 2)); Y = np.zeros((T, 1)); U = 0.5 * np.sin(0.1 * np.arange(T)).reshape(T, 1)
    for k in range(T):
        X[k] = x
        Y[k] = np.sin(x[0]) + 0.05 * rng.standard_normal()
        x = np.array([x[0] + 0.05 * x[1], x[1] + 0.05 * (U[k, 0] - 9.81 * np.sin(x[0]) - 0.05 * x[1] ** 3)]) + np.sqrt([1e-4, 4e-3]) * rng.standard_normal(2)
    cfg = S.make_config(m, 20000, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 3, 0)
    f = _capi.FilterHandle(cfg)
    f.reset()
    r = f.run(U, Y, 0.0, ll_steps=True, xmean=True)
    assert np.all(np.isfinite(r["ll_steps"])) and f.resample_count() > 0
    assert np.sqrt(np.mean((r["xmean"][20:, 0] - X[20:, 0]) ** 2)) < 0.1          # tracks the angle
    # the same filter inside a bank (seed + k) and through the auxiliary verbs
    bank = _capi.BankHandle(cfg, None, n_filters=3)
    bank.reset()
    rb = bank.run(U, Y, 0.0, ll_steps=True)
    f2 = _capi.FilterHandle(cfg); f2.reset()
    assert np.array_equal(rb["ll_steps"][:, 0], f2.run(U, Y, 0.0, ll_steps=True)["ll_steps"])
    f3 = _capi.FilterHandle(cfg); f3.reset()
    assert np.isfinite(f3.run_aux(U, Y, 1)["ll"])
    # the smoother runs for such models too (k_smooth_fx compiled with the snippet): smoothed angle no worse than the filtered one
    f4 = _capi.FilterHandle(S.make_config(m, 2000, S.ADVANCED_PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.5, 11, 0))
    f4.reset()
    r4 = f4.run(U[:60], Y[:60], 0.0, history=True, xmean=True)
    xb, idx = f4.smooth(50, U[:60], r4["x"], r4["w"], r4["we"])
    assert xb.shape == (60, 50, 2) and np.all(np.isfinite(xb))
    assert np.sqrt(np.mean((xb.mean(axis=1)[10:, 0] - X[10:60, 0]) ** 2)) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pf_lg_laplace", "pf_lg_student_t"])
def test_user_measurement_likelihood(name):
    """A measurement likelihood of the user's own — the reference's measurement_likelihood(x, u, y, p, t) of an AdvancedParticleFilter
    (src/PFtypes.jl:226-239) / logpdf of a non-Gaussian measurement density (ext/LowLevelParticleFiltersDistributionsExt.jl:80) — as a
    `loglik` hook of a run-time compiled model with a declared upper bound: bit-identical to the device-order oracle (whose C
    counterpart of the same density is itself held to the independent numpy restatement, tests/test_independent_oracle.py), within
    1e-10 per step of the reference order; a bank of such filters, the auxiliary filter over it, and single steps."""
    import independent_cases as IC
    case = IC.cases()[name]
    U, Y = case["U"], case["Y"]
    g = IC.engine_of(case)
    od, orf = IC.oracle_of(ob, case, ob.ORDER_DEVICE), IC.oracle_of(ob, case, ob.ORDER_REFERENCE)
    for h in (g, od, orf):
        h.reset()
    rg, rd, rr = (h.run(U, Y, 0.0, ll_steps=True) for h in (g, od, orf))
    assert np.array_equal(rg["ll_steps"].view(np.uint64), rd["ll_steps"].view(np.uint64))
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64)) and np.array_equal(g.ancestors(), od.ancestors())
    assert np.array_equal(g.weights().view(np.uint64), od.weights().view(np.uint64))
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= 1e-10 and g.resample_count() == od.resample_count() > 3
    assert od.exact_steps() == 0                                  # the declared bound held: no step fell back to the exact form
    # step by step
    g.reset(); od.reset()
    for k in range(12):
        assert g.correct(U[k], Y[k], float(k)) == od.correct(U[k], Y[k], float(k))
        g.predict(U[k], float(k)); od.predict(U[k], float(k))
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64))
    # the auxiliary filter over it (look-ahead weights from the same likelihood)
    g.reset(); od.reset()
    ra, ro = g.run_aux(U, Y, 0, ll_steps=True), od.run_aux(U, Y, 0, ll_steps=True)
    assert np.array_equal(ra["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))


@pytest.mark.gpu
def test_user_likelihood_without_a_declared_bound_takes_the_exact_form():
    """no `loglik_bound` in the snippet: every step is normalised against the true maximum (the exact form: a k_norm launch in
    front of every head, no host round trip) — the results are the bounded model's to rounding, and the reference order's within tolerance"""
    import independent_cases as IC
    case = dict(IC.cases()["pf_lg_laplace"])
    g1 = IC.engine_of(case)
    case["user"] = case["user"][:2] + (UM.LAPLACE_NO_BOUND_SRC,) + case["user"][3:]
    g0 = IC.engine_of(case)
    orf = IC.oracle_of(ob, case, ob.ORDER_REFERENCE)
    for h in (g0, g1, orf):
        h.reset()
    r0, r1, rr = (h.run(case["U"], case["Y"], 0.0, ll_steps=True) for h in (g0, g1, orf))
    assert np.max(np.abs(r0["ll_steps"] - r1["ll_steps"])) <= 1e-12 and np.max(np.abs(r0["ll_steps"] - rr["ll_steps"])) <= 1e-10
    assert np.array_equal(g0.ancestors(), g1.ancestors())


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", [S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_RESIDUAL])
def test_auxiliary_filter_with_a_likelihood_without_a_declared_bound(strategy):
    """advisor finding of round 3: a likelihood without `loglik_bound` under the AuxiliaryParticleFilter — the "no bound" marker travels
    through the auxiliary second half (k_resprop<AUX> / k_step<MODE_AUX2>) and the next weighting as an offset before the exact form
    replaces it; the results must be the bounded model's to rounding and the reference order's within tolerance, with both resamplers"""
    import independent_cases as IC
    case = dict(IC.cases()["pf_lg_laplace"])
    case["strategy"] = strategy
    g1 = IC.engine_of(case)
    case0 = dict(case)
    case0["user"] = case["user"][:2] + (UM.LAPLACE_NO_BOUND_SRC,) + case["user"][3:]
    g0 = IC.engine_of(case0)
    orf = IC.oracle_of(ob, case, ob.ORDER_REFERENCE)
    for mode in (0, 1):
        for h in (g0, g1, orf):
            h.reset()
        r0, r1, rr = (h.run_aux(case["U"], case["Y"], mode, ll_steps=True) for h in (g0, g1, orf))
        assert np.all(np.isfinite(r0["ll_steps"]))
        assert np.max(np.abs(r0["ll_steps"] - r1["ll_steps"])) <= 1e-12 and np.max(np.abs(r0["ll_steps"] - rr["ll_steps"])) <= 1e-10
        assert np.array_equal(g0.ancestors(), g1.ancestors()) and np.max(np.abs(g0.particles() - g1.particles())) <= 1e-12


@pytest.mark.gpu
def test_user_likelihood_through_the_filter_objects():
    """the mirror of the reference's constructor: AdvancedParticleFilter(N, dynamics, measurement, measurement_likelihood, df, d0) with
    UserDynamics / UserMeasurement / UserLikelihood descriptors (lowlevelparticlefilters.jl_amd/api.py; julia/LLPFAmd.jl has the same)"""
    import llpf_amd
    import independent_cases as IC
    case = IC.cases()["pf_lg_laplace"]
    lg = case["model"]
    A = np.array(lg.A[:4]).reshape(2, 2); B = np.array(lg.B[:2]).reshape(2, 1); Cm = np.array(lg.C[:2]).reshape(1, 2)
    dyn = llpf_amd.UserDynamics(UM.LAPLACE_SRC, 2, 1, 1, A=A, B=B, C=Cm, qt=[0.8])
    df = llpf_amd.MvNormal(np.zeros(2), S.gaussian_cov_matrix(lg.dynamics_density))
    d0 = llpf_amd.MvNormal(np.array(lg.initial_density.mu[:2]), S.gaussian_cov_matrix(lg.initial_density))
    pf = llpf_amd.AdvancedParticleFilter(case["N"], dyn, llpf_amd.UserMeasurement(), llpf_amd.UserLikelihood(), df, d0,
                                         resample_threshold=case["thr"], rng=IC.SEED)
    o = IC.oracle_of(ob, case, ob.ORDER_REFERENCE)
    o.reset()
    ro = o.run(case["U"], case["Y"], 0.0, ll_steps=True)
    sol = llpf_amd.forward_trajectory(pf, case["U"], case["Y"])
    assert abs(sol.ll - ro["ll"]) <= 1e-9
    with pytest.raises(TypeError):
        llpf_amd.AdvancedParticleFilter(100, llpf_amd.LinearDynamics(A, B), llpf_amd.LinearMeasurement(Cm), llpf_amd.UserLikelihood(), df, d0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pf_lg_mult_noise_box", "pf_lg_laplace_noise"])
def test_user_process_noise_and_initial_density(name):
    """The random part of the step in the user's hands — the reference's AdvancedParticleFilter contract, dynamics(x, u, p, t, noise = true)
    adds its own noise (src/PFtypes.jl:242-259, test/runtests.jl:553-599), and a ParticleFilter with any dynamics_density / initial_density
    (rand!(rng, d, noise) src/PFtypes.jl:135, rand(rng, initial_density) src/filtering.jl:8) — as `noise` / `initial` members of a run-time
    compiled model: multiplicative (state-dependent) Gaussian noise with a uniform-box prior, and Laplace process noise.  Bit-identical to the
    device-order oracle (whose C counterparts are held to the independent numpy restatement), within 1e-10 per step of the reference order;
    the constructor's draw, whole runs, single steps, the auxiliary filter over it, and a bank."""
    import independent_cases as IC
    case = IC.cases()[name]
    U, Y = case["U"], case["Y"]
    g = IC.engine_of(case)
    od, orf = IC.oracle_of(ob, case, ob.ORDER_DEVICE), IC.oracle_of(ob, case, ob.ORDER_REFERENCE)
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64))      # the constructor draws from the model's own prior
    if case.get("initial"):
        lo, hi = np.array(case["initial"][1][:2]), np.array(case["initial"][1][2:])
        x0 = g.particles()
        assert np.all(x0 >= lo) and np.all(x0 < hi) and np.std(x0[:, 0]) > 0.5
    for h in (g, od, orf):
        h.reset()
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64))
    rg, rd, rr = (h.run(U, Y, 0.0, ll_steps=True) for h in (g, od, orf))
    assert np.array_equal(rg["ll_steps"].view(np.uint64), rd["ll_steps"].view(np.uint64))
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64)) and np.array_equal(g.ancestors(), od.ancestors())
    assert np.array_equal(g.weights().view(np.uint64), od.weights().view(np.uint64))
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= 1e-10 and g.resample_count() == od.resample_count() > 3
    # step by step
    g.reset(); od.reset()
    for k in range(12):
        assert g.correct(U[k], Y[k], float(k)) == od.correct(U[k], Y[k], float(k))
        g.predict(U[k], float(k)); od.predict(U[k], float(k))
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64))
    # the auxiliary filter over it: look-ahead from the noise-free prediction, then the propagate WITH the model's own noise (filtering.jl:219-234)
    g.reset(); od.reset()
    ra, ro = g.run_aux(U, Y, 0, ll_steps=True), od.run_aux(U, Y, 0, ll_steps=True)
    assert np.array_equal(ra["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    assert np.array_equal(g.particles().view(np.uint64), od.particles().view(np.uint64))
    # a bank: filter k is the single filter with seed + k
    kind, par, src, qt = case["user"]
    m = S.Model.from_buffer_copy(bytes(case["model"]))
    m.model_id = _capi.model_compile(src, m.nx, m.ny)
    for i, v in enumerate(qt):
        m.qt[i] = v
    cfg = S.make_config(m, case["N"], case["kind"], case["strategy"], case["thr"], IC.SEED, 0)
    bank = _capi.BankHandle(cfg, None, n_filters=3)
    bank.reset()
    rb = bank.run(U, Y, 0.0, ll_steps=True)
    g2 = _capi.FilterHandle(cfg)         # a fresh handle: the same reset! number as the bank's
    g2.reset()
    assert np.array_equal(rb["ll_steps"][:, 0], g2.run(U, Y, 0.0, ll_steps=True)["ll_steps"])
    assert not np.array_equal(rb["ll_steps"][:, 1], rb["ll_steps"][:, 0])


@pytest.mark.gpu
def test_user_noise_through_the_filter_objects():
    """UserNoise / UserInitial descriptors in the constructors that mirror the reference's (lowlevelparticlefilters.jl_amd/api.py; julia/LLPFAmd.jl
    has the same), and the pairing checks: a User* descriptor needs the member in the snippet, a member in the snippet needs the descriptor"""
    import llpf_amd
    import independent_cases as IC
    case = IC.cases()["pf_lg_mult_noise_box"]
    lg = case["model"]
    A = np.array(lg.A[:4]).reshape(2, 2); B = np.array(lg.B[:2]).reshape(2, 1); Cm = np.array(lg.C[:2]).reshape(1, 2)
    dyn = llpf_amd.UserDynamics(UM.MULT_NOISE_BOX_SRC, 2, 1, 1, A=A, B=B, C=Cm, qt=case["user"][3])
    dg = llpf_amd.MvNormal(np.zeros(1), S.gaussian_cov_matrix(lg.measurement_density))
    pf = llpf_amd.AdvancedParticleFilter(case["N"], dyn, llpf_amd.UserMeasurement(), llpf_amd.GaussianLikelihood(llpf_amd.UserMeasurement(), dg), llpf_amd.UserNoise(), llpf_amd.UserInitial(),
                                         resample_threshold=case["thr"], rng=IC.SEED)
    o = IC.oracle_of(ob, case, ob.ORDER_REFERENCE)
    o.reset()
    ro = o.run(case["U"], case["Y"], 0.0, ll_steps=True)
    sol = llpf_amd.forward_trajectory(pf, case["U"], case["Y"])
    assert abs(sol.ll - ro["ll"]) <= 1e-9
    df = llpf_amd.MvNormal(np.zeros(2), 0.1)
    with pytest.raises(TypeError):      # the snippet defines `noise`: a Gaussian dynamics_density would be silently overridden
        llpf_amd.AdvancedParticleFilter(100, dyn, llpf_amd.UserMeasurement(), llpf_amd.GaussianLikelihood(llpf_amd.UserMeasurement(), dg), df, llpf_amd.UserInitial())
    with pytest.raises(TypeError):      # UserNoise without a `noise` member in the snippet
        d2 = llpf_amd.UserDynamics(UM.LAPLACE_SRC, 2, 1, 1, A=A, B=B, C=Cm, qt=[0.8])
        llpf_amd.AdvancedParticleFilter(100, d2, llpf_amd.UserMeasurement(), llpf_amd.UserLikelihood(), llpf_amd.UserNoise(), df)
    with pytest.raises(TypeError):      # a snippet with `loglik` paired with a Gaussian likelihood (advisor finding, round 3)
        d2 = llpf_amd.UserDynamics(UM.LAPLACE_SRC, 2, 1, 1, A=A, B=B, C=Cm, qt=[0.8])
        llpf_amd.AdvancedParticleFilter(100, d2, llpf_amd.UserMeasurement(), llpf_amd.GaussianLikelihood(llpf_amd.UserMeasurement(), dg), df, df)


@pytest.mark.gpu
def test_reset_after_single_steps_sees_the_same_input_as_a_fresh_handle():
    """An initial density whose prepare() reads u (here: the box is shifted by u[0]).  reset! evaluates it with u = 0 — also after the
    single-step verbs have staged a non-zero u in the handle's scratch (round-4 advisor finding: llpf_reset used that scratch as if
    it were still zero-filled)."""
    import independent_cases as IC
    case = IC.cases()["pf_lg_mult_noise_box"]
    src = UM.MULT_NOISE_BOX_SRC.replace("out[d] = lo[d] + (hi[d] - lo[d]) * uu[d];", "out[d] = (lo[d] + ushift) + (hi[d] - lo[d]) * uu[d];") \
                               .replace("double s0, s1, lo[NXU], hi[NXU];", "double s0, s1, lo[NXU], hi[NXU], ushift;") \
                               .replace("s0 = m->qt[0]; s1 = m->qt[1];", "s0 = m->qt[0]; s1 = m->qt[1]; ushift = 100.0 * u[0];")
    assert "ushift" in src
    kind, par, _, qt = case["user"]
    m = S.Model.from_buffer_copy(bytes(case["model"]))
    m.model_id = _capi.model_compile(src, m.nx, m.ny)
    for i, v in enumerate(qt):
        m.qt[i] = v
    cfg = S.make_config(m, case["N"], case["kind"], case["strategy"], case["thr"], IC.SEED, 0)
    a, b = _capi.FilterHandle(cfg), _capi.FilterHandle(cfg)
    a.reset(); b.reset()
    x_fresh = b.particles()
    assert np.array_equal(a.particles().view(np.uint64), x_fresh.view(np.uint64)) and np.all(x_fresh < 10.0)      # u = 0: the unshifted box
    U, Y = case["U"], case["Y"]
    u = np.array(U[0], dtype=np.float64).copy(); u[0] = 0.7
    a.correct(u, Y[0], 0.0); a.predict(u, 0.0)
    a.reset(); b.reset()            # second reset! of both handles: same counters, and for `a` a scratch that held u = 0.7
    assert np.array_equal(a.particles().view(np.uint64), b.particles().view(np.uint64))
    assert np.all(a.particles() < 10.0)
