"""GPU parity of the FFBS particle smoother (reference src/smoothing.jl:103-143)."""
import numpy as np
import pytest

import llpf_amd
from llpf_amd import _capi, _structs as S
import models as M
import oracle_binding as ob
from gpu_common import cfg_of as _cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("strategy", [S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED, S.RESAMPLE_RESIDUAL])
def test_smoother_bit_exact(strategy):
    """Backward simulation on the GPU == device-order oracle: every drawn index and every smoothed sample; ragged N
    (not a multiple of the 1024-particle chunk), M < N, nx = 2 and the quad-tank model (nx = 4)."""
    model = M.lg_test_model(0.1)
    X, U, Y = M.simulate_lg(model, 60, seed=2)
    cfg = _cfg(model, 2500, strategy, 0.1, seed=3)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    rg = g.run(U, Y, 0.0, history=True); ro = o.run(U, Y, 0.0, history=True)
    assert np.array_equal(rg["x"].view(np.uint64), ro["x"].view(np.uint64))
    xg, ig = g.smooth(300, U, rg["x"], rg["w"], rg["we"])
    xo, io = o.smooth(300, U, ro["x"], ro["w"], ro["we"])
    assert np.array_equal(ig, io)
    assert np.array_equal(xg.view(np.uint64), xo.view(np.uint64))
    assert np.mean((X - xg.mean(axis=1)) ** 2) < 5                      # test/runtests.jl:317
    if strategy == S.RESAMPLE_SYSTEMATIC:
        qm = M.quadtank_model(); Uq, Yq = M.quadtank_data(25)
        cq = _cfg(qm, 1500, strategy, 0.5, seed=9, kind=S.ADVANCED_PARTICLE_FILTER)
        g = _capi.FilterHandle(cq); o = ob.OracleFilter(cq, ob.ORDER_DEVICE)
        g.reset(); o.reset()
        rg = g.run(Uq, Yq, 0.0, history=True); ro = o.run(Uq, Yq, 0.0, history=True)
        xg, ig = g.smooth(64, Uq, rg["x"], rg["w"], rg["we"]); xo, io = o.smooth(64, Uq, ro["x"], ro["w"], ro["we"])
        assert np.array_equal(ig, io) and np.array_equal(xg.view(np.uint64), xo.view(np.uint64))


def test_smoother_api():
    """smooth(pf, M, u, y) / smoothed_mean / smoothed_cov / smoothed_trajs with the reference's checks
    (test/runtests.jl:314-330): size, mean error < 5, tr(cov) < 2."""
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]]); B = np.array([[0.1], [0.0]]); Cm = np.array([[0.0, 1.0]])
    df = llpf_amd.MvNormal(np.zeros(2), 0.1 ** 2); dg = llpf_amd.MvNormal(np.zeros(1), np.ones(1)); d0 = llpf_amd.MvNormal([0.3, -0.5], 4.0)
    pf = llpf_amd.ParticleFilter(1000, llpf_amd.LinearDynamics(A, B), llpf_amd.LinearMeasurement(Cm), df, dg, d0, rng=5)
    X, U, Y = M.simulate_lg(M.lg_test_model(0.1), 200, seed=2)
    xb, ll = llpf_amd.smooth(pf, 100, U, Y)
    assert xb.shape == (200, 100, 2) and np.isfinite(ll)
    xbm = llpf_amd.smoothed_mean(xb)
    assert xbm.shape == (2, 200) and np.mean((X.T - xbm) ** 2) < 5
    assert all(np.trace(Cv) < 2 for Cv in llpf_amd.smoothed_cov(xb))
    assert llpf_amd.smoothed_trajs(xb).shape == (2, 100, 200)


def test_smoother_of_models_compiled_at_run_time():
    """smooth for run-time compiled models (k_smooth_fx is compiled with the user's dynamics): the quad-tank written as a USER
    snippet gives the built-in quad-tank's — and the oracle's — indices and samples bit for bit; a linear-Gaussian model above the
    precompiled dimensions (nx = 5: compiled on demand) those of the oracle."""
    import user_models as UM
    qm = M.quadtank_model(); Uq, Yq = M.quadtank_data(25)
    qu = S.Model.from_buffer_copy(bytes(qm)); qu.model_id = _capi.model_compile(UM.QUADTANK_SRC, 4, 2)
    cb = _cfg(qm, 1500, S.RESAMPLE_SYSTEMATIC, 0.5, seed=9, kind=S.ADVANCED_PARTICLE_FILTER)
    cu = _cfg(qu, 1500, S.RESAMPLE_SYSTEMATIC, 0.5, seed=9, kind=S.ADVANCED_PARTICLE_FILTER)
    gb, gu, o = _capi.FilterHandle(cb), _capi.FilterHandle(cu), ob.OracleFilter(cb, ob.ORDER_DEVICE)
    for h in (gb, gu, o):
        h.reset()
    rb, ru, ro = (h.run(Uq, Yq, 0.0, history=True) for h in (gb, gu, o))
    xb, ib = gb.smooth(64, Uq, rb["x"], rb["w"], rb["we"]); xu, iu = gu.smooth(64, Uq, ru["x"], ru["w"], ru["we"])
    xo, io = o.smooth(64, Uq, ro["x"], ro["w"], ro["we"])
    assert np.array_equal(iu, ib) and np.array_equal(iu, io) and np.array_equal(xu.view(np.uint64), xo.view(np.uint64))
    rng = np.random.default_rng(3)
    A = 0.9 * np.linalg.qr(rng.standard_normal((5, 5)))[0]
    model = S.make_lg_model(A, 0.2 * rng.standard_normal((5, 1)), rng.standard_normal((2, 5)), S.make_gaussian(np.zeros(5), 0.05),
                            S.make_gaussian(np.zeros(2), 0.5), S.make_gaussian(np.zeros(5), 1.0))
    _, U, Y = M.simulate_lg(model, 30, seed=4)
    cfg = _cfg(model, 2000, S.RESAMPLE_SYSTEMATIC, 0.5, seed=3)
    g, o = _capi.FilterHandle(cfg), ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    rg, ro = g.run(U, Y, 0.0, history=True), o.run(U, Y, 0.0, history=True)
    xg, ig = g.smooth(100, U, rg["x"], rg["w"], rg["we"]); xo, io = o.smooth(100, U, ro["x"], ro["w"], ro["we"])
    assert np.array_equal(ig, io) and np.array_equal(xg.view(np.uint64), xo.view(np.uint64))
