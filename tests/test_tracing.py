"""Callables without device source (round 6, lowlevelparticlefilters.jl_amd/tracing.py): the reference's constructors take closures
(src/PFtypes.jl:59-63, 189-193); here an ordinary Python function is run once on tracer numbers and emitted as the `UserModel` snippet.
CPU part: the emitted snippet is the callable's own operation sequence (evaluated with plain floats: same bits as the callable), what
cannot be traced is refused with a message, and hiprtc accepts the snippet (LLPF_JIT_COMPILE_ONLY: nothing runs).  GPU part: the
quad-tank of examples/example_quadtank.jl:8-27 written as an ordinary function + rk4 reproduces the built-in model bit for bit."""
import math
import os
import re
import struct

import numpy as np
import pytest

import llpf_amd
from llpf_amd import _capi, _structs as S, tracing as tr
import models as M

Q = dict(S.QUADTANK_DEFAULTS)


def quadtank(h, u, p, t):
    """examples/example_quadtank.jl:8-27 (the tank parameters in p), in the built-in model's expression order"""
    a1 = tr.ifelse(t > p["t_switch"], p["a1"] * p["a1_factor"], p["a1"])
    g2 = 2.0 * p["g"]
    ss = [tr.sqrt(tr.maximum(g2 * h[i], 0.0) + p["eps"]) for i in range(4)]
    c1a = tr.ifelse(t > p["t_switch"], (-(p["a1"] * p["a1_factor"])) / p["A1"], (-p["a1"]) / p["A1"])
    del a1
    return [c1a * ss[0] + (p["a3"] / p["A1"]) * ss[2] + ((p["gamma1"] * p["k1"]) / p["A1"]) * u[0],
            ((-p["a2"]) / p["A2"]) * ss[1] + (p["a4"] / p["A2"]) * ss[3] + ((p["gamma2"] * p["k2"]) / p["A2"]) * u[1],
            ((-p["a3"]) / p["A3"]) * ss[2] + (((1.0 - p["gamma2"]) * p["k2"]) / p["A3"]) * u[1],
            ((-p["a4"]) / p["A4"]) * ss[3] + (((1.0 - p["gamma1"]) * p["k1"]) / p["A4"]) * u[0]]


def levels(h, u, p, t):
    return [h[0], h[1]]


def _eval_snippet_body(src, member, x, u, t):
    """evaluate the emitted statements of one member with plain Python floats (the snippet is straight-line code)"""
    body = src[src.index("DEV void " + member):]
    body = body[body.index("{") + 1:body.index("\n    }")]
    env = {"x": list(x), "u_": list(u), "t_": t, "out": [0.0] * 16,
           "llpf_sqrt": math.sqrt, "llpf_fabs": abs, "llpf_exp": math.exp, "llpf_log": math.log, "llpf_log1p_nonneg": math.log1p,
           "llpf_u2d": lambda b: struct.unpack("<d", struct.pack("<Q", b))[0]}
    for line in body.strip().split("\n"):
        line = re.sub(r"/\*.*?\*/", "", line.strip()).rstrip(";").strip()
        line = re.sub(r"^const (double|bool) ", "", line)
        line = re.sub(r"0x([0-9a-f]{16})ULL", r"0x\1", line)
        line = re.sub(r"\(([^()?]+) \? ([^():]+) : ([^()]+)\)", r"(\2 if \1 else \3)", line)
        line = line.replace("&&", " and ").replace("||", " or ").replace("(!", "(not ")
        exec(line, env)
    return env["out"]


def test_the_emitted_snippet_is_the_callable_operation_for_operation():
    f = tr.rk4(quadtank, 1.0, 2)
    src = tr.emit_user_model(4, 2, 2, f, levels, p=Q)
    assert "struct UserModel" in src and "DEV void dynamics" in src and "out[3] =" in src and "loglik" not in src
    assert src.count("llpf_sqrt(") == 4 * 4 * 2                       # four stages x two sub-steps x four tanks: nothing duplicated, nothing lost
    rng = np.random.default_rng(0)
    for t in (3.0, 499.5, 500.0, 777.0):                              # both sides of the t > 500 switch, and the sub-step that crosses it
        x = list(2.0 + rng.random(4) * 3); u = [0.5, 0.25]
        host = f(x, u, Q, t)
        emitted = _eval_snippet_body(src, "dynamics", x, u, t)[:4]
        assert [struct.pack("<d", a) for a in host] == [struct.pack("<d", b) for b in emitted]
        # ... and it IS the quad-tank of the mirror (same formulas, written independently in api.QuadTankDynamics)
        np.testing.assert_allclose(host, llpf_amd.QuadTankDynamics(supersample=2)(x, u, None, t), rtol=1e-13)


def test_what_cannot_be_traced_is_refused_with_a_reason():
    with pytest.raises(tr.TraceError, match="truth value"):
        tr.emit_user_model(1, 0, 1, lambda x, u, p, t: [x[0] if x[0] > 0 else -x[0]])
    with pytest.raises(tr.TraceError, match="truth value"):
        tr.emit_user_model(1, 0, 1, lambda x, u, p, t: [max(x[0], 0.0)])
    with pytest.raises(tr.TraceError, match="float"):
        tr.emit_user_model(1, 0, 1, lambda x, u, p, t: [math.sin(x[0])])
    with pytest.raises(tr.TraceError, match="returned 1 values"):
        tr.emit_user_model(2, 0, 1, lambda x, u, p, t: [x[0]])
    with pytest.raises(tr.TraceError, match="state only"):
        tr.emit_user_model(1, 1, 1, lambda x, u, p, t: [x[0]], lambda x, u, p, t: [x[0] + u[0]])
    # what IS traced: numpy's ufuncs on the tracer numbers, integer powers, conditions combined, constants of any origin
    src = tr.emit_user_model(2, 1, 1, lambda x, u, p, t: [np.sqrt(abs(x[0])) + np.float64(0.1) * u[0], tr.ifelse((x[1] > 0.0) & (t < 5.0), x[1] ** 3, np.exp(x[1]))],
                             lambda x, u, p, t: [x[0] * x[1]],
                             loglik=lambda x, u, y, p, t: -0.5 * (y[0] - x[0]) ** 2 - t * 0.0, loglik_bound=0.0)
    assert "llpf_fabs" in src and "llpf_exp" in src and "&&" in src and "loglik_bound" in src and "const double t)" not in src


def test_hiprtc_accepts_the_snippet(monkeypatch):
    """compile only (no GPU needed: hiprtc cross-compiles for gfx950); the model id comes back and the traits say what the snippet defines"""
    monkeypatch.setenv("LLPF_JIT_COMPILE_ONLY", "1")
    src = tr.emit_user_model(4, 2, 2, tr.rk4(quadtank, 1.0, 2), levels, p=Q)
    mid = _capi.model_compile(src, 4, 2)
    assert mid >= 1000 and not (_capi.model_traits(mid) & _capi.TRAIT_LOGLIK)
    src2 = tr.emit_user_model(2, 0, 1, lambda x, u, p, t: [0.9 * x[0] + 0.1 * x[1], tr.ifelse(x[0] > 1.0, x[1], -x[1])], lambda x, u, p, t: [x[0]],
                              loglik=lambda x, u, y, p, t: -abs(y[0] - x[0]) / 0.8 - math.log(1.6), loglik_bound=-math.log(1.6))
    mid2 = _capi.model_compile(src2, 2, 1)
    tr2 = _capi.model_traits(mid2)
    assert (tr2 & _capi.TRAIT_LOGLIK) and (tr2 & _capi.TRAIT_LOGLIK_BOUND)


@pytest.mark.gpu
def test_the_quad_tank_as_an_ordinary_function_is_the_builtin_model_bit_for_bit():
    """AdvancedParticleFilter(N, dynamics, measurement, ...) with a Python FUNCTION for the dynamics (rk4 of the continuous-time tank
    equations, as examples/example_quadtank.jl builds it) against the built-in descriptor: per-step log-likelihoods, particles,
    ancestors — across the t > 500 switch, two resampling strategies."""
    df = llpf_amd.MvNormal(np.zeros(4), np.full(4, 0.1)); dg = llpf_amd.MvNormal(np.zeros(2), np.full(2, 1e-4))
    d0 = llpf_amd.MvNormal(np.array([2.0, 2.0, 3.0, 3.0]), np.full(4, 0.1))
    U, Y = M.quadtank_data(60, seed=2)
    for strat, thr in ((llpf_amd.ResampleSystematic, 0.5), (llpf_amd.ResampleStratified, 1.0)):
        built = llpf_amd.AdvancedParticleFilter(6000, llpf_amd.QuadTankDynamics(supersample=2), llpf_amd.QuadTankMeasurement(), llpf_amd.GaussianLikelihood(llpf_amd.QuadTankMeasurement(), dg), df, d0,
                                                resample_threshold=thr, resampling_strategy=strat, rng=7)
        traced = llpf_amd.AdvancedParticleFilter(6000, tr.rk4(quadtank, 1.0, 2), levels, llpf_amd.GaussianLikelihood(levels, dg), df, d0,
                                                 resample_threshold=thr, resampling_strategy=strat, rng=7, p=Q, nu=2)
        assert isinstance(traced.dynamics, llpf_amd.UserDynamics)
        for pf in (built, traced):
            pf._h.reset()
        rb = built._h.run(U, Y, 470.0, ll_steps=True)
        rt = traced._h.run(U, Y, 470.0, ll_steps=True)
        assert np.array_equal(rt["ll_steps"].view(np.uint64), rb["ll_steps"].view(np.uint64))
        assert np.array_equal(traced._h.particles().view(np.uint64), built._h.particles().view(np.uint64))
        assert np.array_equal(traced._h.ancestors(), built._h.ancestors()) and traced._h.resample_count() == built._h.resample_count() > 0


@pytest.mark.gpu
def test_a_traced_likelihood_against_the_handwritten_snippet():
    """ParticleFilter / AdvancedParticleFilter with closures for everything: a linear system with a Laplace measurement likelihood (the
    hand-written UM.LAPLACE_SRC of tests/user_models.py) — the same operations in the same order, so the same bits."""
    import user_models as UM
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]]); B = np.array([[0.1], [0.0]]); Cm = np.array([[0.0, 1.0]])
    b = 0.8
    # (the hand-written snippet forms its constant with the ENGINE's log, which is not libm's to the last bit: a constant a closure bakes in
    #  carries whatever bits the host gave it — here the same ones, through the device self-test of llpf_log)
    c = 1.0 * float(_capi.selftest_math(1, np.array([2.0 * b]))[0])

    def f(x, u, p, t):
        return [(A[r, 0] * x[0] + A[r, 1] * x[1]) + B[r, 0] * u[0] for r in range(2)]

    def g(x, u, p, t):
        return [Cm[0, 0] * x[0] + Cm[0, 1] * x[1]]

    def ll(x, u, y, p, t):
        return (-(abs(y[0] - g(x, u, p, t)[0]) / b)) - c

    df = llpf_amd.MvNormal(np.zeros(2), 0.01); d0 = llpf_amd.MvNormal(np.array([0.3, -0.5]), 4.0)
    hand = llpf_amd.AdvancedParticleFilter(4000, llpf_amd.UserDynamics(UM.LAPLACE_SRC, 2, 1, 1, A=A, B=B, C=Cm, qt=[b]), llpf_amd.UserMeasurement(),
                                           llpf_amd.UserLikelihood(), df, d0, rng=11)
    traced = llpf_amd.AdvancedParticleFilter(4000, f, g, ll, df, d0, rng=11, nu=1, ny=1, likelihood_bound=-c)
    _, U, Y = M.simulate_lg(M.lg_test_model(), 80, seed=4)
    la, lb = llpf_amd.loglik(hand, U, Y), llpf_amd.loglik(traced, U, Y)
    assert la == lb and np.isfinite(la)
    assert np.array_equal(hand._h.particles().view(np.uint64), traced._h.particles().view(np.uint64))


def test_the_julia_twin_emits_the_same_node_set():
    """julia/tracing.jl cannot run here (no Julia): held statically to the Python tracer — the same operations with the same device spellings,
    the same snippet frame, and the wrapper includes and exports it"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jl = open(os.path.join(root, "lowlevelparticlefilters.jl_amd", "julia", "tracing.jl"), encoding="utf-8").read()
    fmt = dict(re.findall(r':(\w+) => "([^"]*)"', jl[jl.index("const TRACE_FMT"):jl.index("const TRACE_BOOL")]))
    want = {k: v.replace("{0}", "\\$1").replace("{1}", "\\$2").replace("{2}", "\\$3") for k, v in tr._FMT.items() if k != "ne"}
    assert fmt == want
    assert set(re.findall(r":(\w+)", jl[jl.index("const TRACE_BOOL"):jl.index("const TRACE_INPUT")])) == tr._BOOL - {"ne"}
    for frame in ("static constexpr bool RB = false;", "DEV void prepare(const ModelD* m, const double* u, double t)", "DEV double loglik(const double* x, const double* y, double t) const",
                  "DEV double loglik_bound() const", "DEV void measurement(const double* x, double* out) const"):
        assert frame in jl and frame in open(tr.__file__).read()
    mod = open(os.path.join(root, "lowlevelparticlefilters.jl_amd", "julia", "LLPFAmd.jl"), encoding="utf-8").read()
    assert 'include(joinpath(@__DIR__, "tracing.jl"))' in mod and "trace_dynamics" in mod.split("const LIB")[0]
