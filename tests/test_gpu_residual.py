"""GPU parity of residual resampling (reference src/resample.jl:63-117)."""
import numpy as np
import pytest

from llpf_amd import _capi, _structs as S
import models as M
import oracle_binding as ob
from gpu_common import TOL_LL_STEP, cfg_of as _cfg, compare_state as _compare_state

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(10, 10), (9, 9), (1000, 1000), (5000, 5000), (4097, 300), (300, 4097), (100000, 100000)])
def test_standalone_residual_resample_matches_oracle(n, m):
    rng = np.random.default_rng(n + m)
    we = rng.exponential(size=n) ** 3
    we /= we.sum()
    U = rng.uniform(size=m)
    j0 = np.full(m, 5, dtype=np.int64)
    jg = _capi.resample(S.RESAMPLE_RESIDUAL, we, U, m, j0)
    jd, _ = ob.resample(S.RESAMPLE_RESIDUAL, we, U, m, ob.ORDER_DEVICE, j0)
    jr, _ = ob.resample(S.RESAMPLE_RESIDUAL, we, U, m, ob.ORDER_REFERENCE, j0)
    assert np.array_equal(jg, jd)
    assert np.sum(jg != jr) <= 1          # ulp-level ties only
    assert jg.min() >= 0 and jg.max() < n
    # uniform weights: every particle exactly once, no draw
    we = np.full(n, 1.0 / n)
    if m == n:
        assert np.array_equal(_capi.resample(S.RESAMPLE_RESIDUAL, we, U, m), np.arange(n))


@pytest.mark.parametrize("thr", [0.5, 1.0])
def test_residual_strategy_trajectory_bit_exact(thr):
    """ParticleFilter with resampling_strategy = ResampleResidual over whole trajectories (run loop, history, single
    steps) and in a bank: bit-identical to the device-order oracle, tolerance against the reference order."""
    model = M.lg_c1_model()
    _, U, Y = M.simulate_lg(model, 80)
    cfg = _cfg(model, 3000, S.RESAMPLE_RESIDUAL, thr, seed=41)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    for h in (g, o, r):
        h.reset()
    rg = g.run(U, Y, 0.0, ll_steps=True, history=True); ro = o.run(U, Y, 0.0, ll_steps=True, history=True)
    rr = r.run(U, Y, 0.0, ll_steps=True)
    assert np.array_equal(rg["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    for key in ("x", "w", "we"):
        assert np.array_equal(rg[key].view(np.uint64), ro[key].view(np.uint64)), key
    _compare_state(g, o)
    assert g.resample_count() == o.resample_count() > 3
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= TOL_LL_STEP
    # asynchronous loop (no history) and single steps
    g2 = _capi.FilterHandle(cfg); g2.reset()
    r2 = g2.run(U, Y, 0.0, ll_steps=True)
    assert np.array_equal(r2["ll_steps"].view(np.uint64), ro["ll_steps"].view(np.uint64))
    g3 = _capi.FilterHandle(cfg); o3 = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g3.reset(); o3.reset()
    for k in range(15):
        assert g3.update(U[k], Y[k], k * 1.0) == o3.update(U[k], Y[k], k * 1.0)
        assert np.array_equal(g3.ancestors(), o3.ancestors())
    _compare_state(g3, o3)
    # bank
    models = [M.lg_test_model(s) for s in (0.05, 0.2)]
    _, U2, Y2 = M.simulate_lg(models[0], 40)
    bank = _capi.BankHandle(_cfg(models[0], 5000, S.RESAMPLE_RESIDUAL, thr, seed=43), models)
    bank.reset()
    rb = bank.run(U2, Y2, 1.0, ll_steps=True)
    for k, mk in enumerate(models):
        ok = ob.OracleFilter(_cfg(mk, 5000, S.RESAMPLE_RESIDUAL, thr, seed=43 + k), ob.ORDER_DEVICE)
        ok.reset()
        assert np.array_equal(ok.run(U2, Y2, 1.0, ll_steps=True)["ll_steps"].view(np.uint64), rb["ll_steps"][:, k].copy().view(np.uint64))
