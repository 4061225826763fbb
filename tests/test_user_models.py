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
        with pytest.raises(_capi.LLPFError) as ei:
            _capi.model_compile("struct UserModel { int broken }", 2, 1)
        assert "hiprtc" in str(ei.value)
        with pytest.raises(_capi.LLPFError):
            _capi.model_compile(UM.PENDULUM_SRC, 9, 1)
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
        assert np.array_equal(ru["xmean"], rb["xmean"])
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
    with pytest.raises(_capi.LLPFError):
        f3.smooth(10, U, np.zeros((T, 20000, 2)), np.zeros((T, 20000)), np.zeros((T, 20000)))
