"""GPU parity of the Rao-Blackwellized particle filter with per-particle covariance (LLPF_MODEL_RB_BILINEAR: An a
function of the nonlinear state, reference src/rbpf.jl:163-283 with singleR off; BASELINE config C5)."""
import numpy as np
import pytest

import llpf_amd
from llpf_amd import _capi, _structs as S
import oracle_binding as ob
import rbfull_models as M
from gpu_common import TOL_LL_STEP, cfg_of as _cfg, compare_state as _compare_state

pytestmark = pytest.mark.gpu


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


def _compare_linear_state(g, o):
    xg, Rg = g.rb_linear_state()
    xo, Ro = o.rb_linear_state()
    assert _same_bits(xg, xo), "Kalman means differ"
    assert _same_bits(Rg, Ro), "Kalman covariances differ"


CASES = {
    "lin_1_2_1": lambda: M.linear_case(1, 2, 1, seed=1)[0],
    "lin_2_2_2": lambda: M.linear_case(2, 2, 2, seed=1)[0],
    "lin_4_8_2": lambda: M.linear_case(4, 8, 2, seed=1)[0],
    "quadtank_4_8_2": lambda: M.quadtank_case(),
    "lin_2_3_3": lambda: M.linear_case(2, 3, 3, seed=1)[0],          # three and four outputs (round 5): compiled on demand
    "lin_4_8_4": lambda: M.linear_case(4, 8, 4, seed=1)[0],
}


# every shape with systematic resampling, the other two strategies on two shapes
@pytest.mark.parametrize("name,strategy", [(n, st) for st in (S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED, S.RESAMPLE_RESIDUAL) for n in CASES
                                           if st == S.RESAMPLE_SYSTEMATIC or n in ("lin_2_2_2", "quadtank_4_8_2")])
def test_rbfull_bit_exact(name, strategy):
    """Whole trajectories (per-step ll, history of xn, the final xn / xl / R of every particle, ancestors) bit-identical
    to the device-order oracle and within tolerance of the reference-order one; then single steps."""
    model = CASES[name]()
    N, T = 3000, 40
    U, Y = M.simulate_io(model, T)
    Y[7] = np.nan                                              # a missing measurement: weights pass through
    cfg = _cfg(model, N, strategy, 0.5, seed=5)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    _compare_state(g, o); _compare_linear_state(g, o)
    for h in (g, o, r):
        h.reset()
    _compare_state(g, o); _compare_linear_state(g, o)
    rg = g.run(U, Y, 0.0, ll_steps=True, history=True); ro = o.run(U, Y, 0.0, ll_steps=True, history=True)
    rr = r.run(U, Y, 0.0, ll_steps=True)
    assert o.resample_count() > 3
    assert _same_bits(rg["ll_steps"], ro["ll_steps"])
    for key in ("x", "w", "we"):
        assert _same_bits(rg[key], ro[key]), key
    _compare_state(g, o); _compare_linear_state(g, o)
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= TOL_LL_STEP
    # the asynchronous loop (no history)
    g2 = _capi.FilterHandle(cfg); g2.reset()
    r2 = g2.run(U, Y, 0.0, ll_steps=True, xmean=True)
    assert _same_bits(r2["ll_steps"], ro["ll_steps"])
    _compare_linear_state(g2, o)
    # particles / history / means are [xn; xl]; the weighted mean is a block-tree sum (tolerance, never fed back)
    nn, nl = model.nx, model.rb.nxl
    assert rg["x"].shape == (T, N, nn + nl) and r2["xmean"].shape == (T, nn + nl)
    assert np.allclose(r2["xmean"], np.einsum("tnd,tn->td", ro["x"], ro["we"]), rtol=1e-11, atol=1e-13)
    assert _same_bits(g.particles()[:, nn:], g.rb_linear_state()[0])
    assert np.allclose(g.weighted_mean(), o.weighted_mean(), rtol=1e-11, atol=1e-13)
    # single steps
    g3 = _capi.FilterHandle(cfg); o3 = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g3.reset(); o3.reset()
    for k in range(10):
        assert g3.update(U[k], Y[k], k * 1.0) == o3.update(U[k], Y[k], k * 1.0)
        _compare_state(g3, o3); _compare_linear_state(g3, o3)


@pytest.mark.parametrize("N,T,thr", [(200_000, 12, 0.5), (500_000, 5, 0.1), (140_000, 6, 0.9)])
def test_rbfull_large_and_repeated_runs(N, T, thr):
    """N = 2e5 (the size of BASELINE config C5), quad-tank coupling: log-likelihood per step bit-identical to the
    device-order oracle; a second and third run of the same handle (captured graph) reproduce a fresh handle's results
    for their own noise.  The kernel is persistent (two waves per SIMD, each taking batches of 64 particles in turn, the
    next batch's covariance planes prefetched into LDS): 2e5 particles are two batches for about half of the waves,
    5e5 up to four for every wave, 1.4e5 one batch for most and two for a few."""
    model = M.quadtank_case()
    U, Y = M.simulate_io(model, T)
    cfg = _cfg(model, N, S.RESAMPLE_SYSTEMATIC, thr, seed=9)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    ob.set_threads(8)
    try:
        g.reset(); o.reset()
        rg = g.run(U, Y, 0.0, ll_steps=True); ro = o.run(U, Y, 0.0, ll_steps=True)
        assert _same_bits(rg["ll_steps"], ro["ll_steps"])
        _compare_linear_state(g, o)
        for _ in range(2):
            g.reset(); o.reset()
            rg = g.run(U, Y, 0.0, ll_steps=True); ro = o.run(U, Y, 0.0, ll_steps=True)
            assert _same_bits(rg["ll_steps"], ro["ll_steps"])
    finally:
        ob.set_threads(1)


@pytest.mark.parametrize("N,thr", [(65, 0.5), (1000, 0.9), (131071, 0.3), (131073, 0.9), (150001, 0.1), (131072 + 64, 1.0)])
def test_rbfull_awkward_sizes(N, thr):
    """The persistent kernel at sizes that are not its grain: below one 64-particle batch, not a multiple of 64 or of the 1024-particle
    tile, one particle below / above the 131072 the resident waves take in their first pass, a single batch beyond them; resampling at
    every step, sometimes, rarely.  Log-likelihood per step, particles, linear means and covariances bit-identical to the device-order oracle."""
    model = M.quadtank_case()
    T = 4
    U, Y = M.simulate_io(model, T, seed=N % 97)
    cfg = _cfg(model, N, S.RESAMPLE_SYSTEMATIC, thr, seed=7 + N)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    ob.set_threads(8)
    try:
        g.reset(); o.reset()
        rg = g.run(U, Y, 0.0, ll_steps=True); ro = o.run(U, Y, 0.0, ll_steps=True)
        assert _same_bits(rg["ll_steps"], ro["ll_steps"])
        assert g.resample_count() == o.resample_count()
        _compare_linear_state(g, o)
    finally:
        ob.set_threads(1)


def test_rbfull_api_and_errors():
    """RBPF(...; An = StateAffineCoupling(...)) through the reference-shaped API; unsupported combinations fail loudly."""
    model, mats = M.linear_case(2, 2, 2, seed=1)
    U, Y = M.simulate_io(model, 30)
    mv = lambda gs: llpf_amd.MvNormal(S.gaussian_mean(gs), S.gaussian_cov_matrix(gs))
    kf = llpf_amd.KalmanFilter(mats["Al"], mats["Bl"], mats["Cl"], 0, mats["R1l"], S.gaussian_cov_matrix(model.measurement_density),
                               mv(model.linear_initial))
    mm = llpf_amd.RBMeasurementModel(llpf_amd.LinearMeasurement(mats["Gn"]), S.gaussian_cov_matrix(model.measurement_density), 2)
    An = llpf_amd.StateAffineCoupling(mats["An"][0], mats["An"][1:])
    pf = llpf_amd.RBPF(2000, kf, llpf_amd.LinearDynamics(mats["Fn"], mats["Bn"]), mm, S.gaussian_cov_matrix(model.dynamics_density),
                       mv(model.initial_density), An=An, nu=1, rng=5, resample_threshold=0.5)
    o = ob.OracleFilter(_cfg(model, 2000, S.RESAMPLE_SYSTEMATIC, 0.5, seed=5), ob.ORDER_DEVICE)
    o.reset()
    sol = llpf_amd.forward_trajectory(pf, U, Y)
    assert sol.ll == o.run(U, Y, 0.0)["ll"]
    assert llpf_amd.particles(pf).shape == (2000, 4) and sol.x.shape == (30, 2000, 4)
    assert llpf_amd.mean_trajectory(sol).shape == (30, 4)        # weighted means of [xn; xl]: the linear-state estimate
    x0 = llpf_amd.particles(pf)
    pf._h.set_particles(x0 * 0.5)                                # both parts of the RBParticle are installed
    assert np.array_equal(llpf_amd.particles(pf), x0 * 0.5) and np.array_equal(pf.linear_state()[0], x0[:, 2:] * 0.5)
    xl, R = pf.linear_state()
    assert xl.shape == (2000, 2) and R.shape == (2000, 2, 2) and pf.covariance.shape == (2, 2)
    assert np.max(np.abs(R - R[0])) > 1e-6
    with pytest.raises(_capi.LLPFError):
        llpf_amd.smooth(pf, 10, U, Y)
    bad, _ = M.linear_case(2, 4, 2, seed=1)
    bad.rb.nxl = 9                                               # above the header's limit (LLPF_RBF_MAXL)
    with pytest.raises(_capi.LLPFError):
        _capi.FilterHandle(_cfg(bad, 1000))


@pytest.mark.parametrize("name,N,T", [("lin_2_2_2", 3000, 30), ("quadtank_4_8_2", 20000, 12), ("quadtank_4_8_2", 60000, 5)])
def test_bank_of_rb_filters(name, N, T):
    """A parameter sweep over Rao-Blackwellized filters — BASELINE config C4's pattern applied to C5's model (reference:
    `map(svec) do s ... loglik(pfs, u, y) end`, test/runtests.jl:412-417, over the filters of test/test_rbpf.jl:111-166): a bank of 8
    filters with different measurement-noise levels returns, filter by filter, the bits of 8 single filters (key seed + k) and of the
    device-order oracle, in both schedules (exp-sums inside k_rbfull / by k_norm), also sharded over two folded devices.  The
    persistent waves of the 8x8 kernel are shared among the bank's filters: at N = 6e4 every wave takes three or four batches."""
    import os
    base = CASES[name]()
    U, Y = M.simulate_io(base, T)
    Y[3] = np.nan
    models = []
    for k in range(8):
        m = type(base).from_buffer_copy(bytes(base))
        cov = np.asarray(S.gaussian_cov_matrix(base.measurement_density))
        m.measurement_density = S.make_gaussian(np.zeros(cov.shape[0]), cov * (0.5 + 0.25 * k))
        models.append(m)
    cfg = _cfg(base, N, S.RESAMPLE_SYSTEMATIC, 0.5, seed=40)
    singles = []
    for k in range(8):
        g = _capi.FilterHandle(_cfg(models[k], N, S.RESAMPLE_SYSTEMATIC, 0.5, seed=40 + k))
        g.reset()
        singles.append(g.run(U, Y, 0.0, ll_steps=True)["ll_steps"])
    for sched in ("merged", "split"):
        os.environ["LLPF_SCHEDULE"] = sched
        try:
            b = _capi.BankHandle(cfg, models)
            b.reset()
            rb = b.run(U, Y, 0.0, ll_steps=True)
        finally:
            del os.environ["LLPF_SCHEDULE"]
        for k in range(8):
            assert _same_bits(rb["ll_steps"][:, k].copy(), singles[k]), (sched, k)
    for k in (0, 7):
        o = ob.OracleFilter(_cfg(models[k], N, S.RESAMPLE_SYSTEMATIC, 0.5, seed=40 + k), ob.ORDER_DEVICE)
        o.reset()
        assert _same_bits(o.run(U, Y, 0.0, ll_steps=True)["ll_steps"], singles[k]), k
    mb = _capi.MBankHandle(cfg, models, devices=[0, 0])
    mb.reset()
    assert _same_bits(mb.run(U, Y, 0.0)["ll"], rb["ll"])


@pytest.mark.parametrize("shape", [(2, 4, 2), (3, 5, 1), (4, 3, 2)])
def test_rb_shapes_compiled_on_demand(shape):
    """(nxn, nxl, ny) beyond the precompiled (1,2,1), (2,2,2), (4,8,2): k_rbfull<LinGauss<nxn, ny>, nxn, nxl, ny> is compiled through
    hiprtc when the filter is built (kernels/jit.hpp, cached per shape) — the reference's RBPF is generic in its dimensions
    (src/rbpf.jl:63-144).  Trajectories, every particle's Kalman mean and covariance, single steps: the device-order oracle's bits."""
    model, _ = M.linear_case(*shape, seed=2)
    U, Y = M.simulate_io(model, 25)
    cfg = _cfg(model, 2000, S.RESAMPLE_SYSTEMATIC, 0.5, seed=6)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE); r = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    for h in (g, o, r):
        h.reset()
    _compare_state(g, o); _compare_linear_state(g, o)
    rg, ro, rr = (h.run(U, Y, 0.0, ll_steps=True) for h in (g, o, r))
    assert _same_bits(rg["ll_steps"], ro["ll_steps"]) and o.resample_count() > 1
    _compare_state(g, o); _compare_linear_state(g, o)
    assert np.max(np.abs(rg["ll_steps"] - rr["ll_steps"])) <= TOL_LL_STEP
    g.reset(); o.reset()
    for k in range(6):
        assert g.update(U[k], Y[k], float(k)) == o.update(U[k], Y[k], float(k))
    _compare_linear_state(g, o)

