"""The persistent multi-step launch (csrc/kernels/persist.hpp: all timesteps with a weighting phase in ONE cooperative launch,
a grid barrier between them, cross-block data through agent-scope coherent accesses) must be the one-launch-per-timestep
path bit for bit — and both are the device-order oracle's bits (reference loop: src/filtering.jl:343-365, 140-168)."""
import os

import numpy as np
import pytest

import models as M
import oracle_binding as ob
from gpu_common import cfg_of
from llpf_amd import _capi, _structs as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _needs_a_devtools_build():
    """The persistent form is an experiment that measured slower: the product library is built without it
    (csrc/Makefile, DEVTOOLS=1 compiles it in), and then LLPF_PERSIST=1 changes nothing."""
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, 8)
    r = _run(cfg_of(model, 2000, S.RESAMPLE_SYSTEMATIC, 1.0), U, Y, True)
    if r[0][1]["persistent_timesteps"] == 0:
        pytest.skip("library built without DEVTOOLS=1: the persistent multi-step kernel is not compiled in")


def _run(cfg, U, Y, persist, passes=1, t0=1.0, ll_steps=True):
    os.environ["LLPF_PERSIST"] = "1" if persist else "0"
    try:
        g = _capi.FilterHandle(cfg)
        out = []
        for _ in range(passes):
            g.reset()
            r = g.run(U, Y, t0, ll_steps=ll_steps)
            out.append((r, g.last_run_stats(), g.particles(), g.weights(), g.expweights(), g.ancestors(), g.resample_count()))
        return out
    finally:
        del os.environ["LLPF_PERSIST"]


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


@pytest.mark.parametrize("N,T,thr,strategy", [
    (5000, 60, 0.1, S.RESAMPLE_SYSTEMATIC), (5000, 60, 1.0, S.RESAMPLE_SYSTEMATIC), (20000, 40, 0.5, S.RESAMPLE_STRATIFIED),
    (1000, 30, 1.0, S.RESAMPLE_SYSTEMATIC), (300000, 25, 1.0, S.RESAMPLE_SYSTEMATIC), (1000000, 12, 1.0, S.RESAMPLE_SYSTEMATIC),
    (1000000, 12, 0.1, S.RESAMPLE_SYSTEMATIC)])
def test_persistent_run_is_the_per_launch_run(N, T, thr, strategy):
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, T, seed=4)
    cfg = cfg_of(model, N, strategy, thr, seed=11)
    p = _run(cfg, U, Y, True, passes=3)
    q = _run(cfg, U, Y, False, passes=3)
    for (rp, sp, xp, wp, ep, jp, cp), (rq, sq, xq, wq, eq, jq, cq) in zip(p, q):
        assert sp["persistent_timesteps"] == T - 1 and sp["fused_launches"] == 2, sp
        assert sq["persistent_timesteps"] == 0
        assert _same(rp["ll_steps"], rq["ll_steps"]) and rp["ll"] == rq["ll"]
        assert _same(xp, xq) and _same(wp, wq) and _same(ep, eq) and np.array_equal(jp, jq) and cp == cq
    if N <= 20000:      # and the oracle's bits (device order)
        o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
        o.reset()
        ro = o.run(U, Y, 1.0, ll_steps=True)
        assert _same(p[0][0]["ll_steps"], ro["ll_steps"])
        assert _same(p[0][2], o.particles()) and np.array_equal(p[0][5], o.ancestors())


def test_missing_measurements_and_forward_time_origin():
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, 40, seed=5)
    Y = Y.copy()
    Y[7] = np.nan
    Y[8] = np.nan
    Y[39] = np.nan
    cfg = cfg_of(model, 6000, S.RESAMPLE_SYSTEMATIC, 0.3, seed=3)
    p = _run(cfg, U, Y, True, t0=0.0)[0]
    q = _run(cfg, U, Y, False, t0=0.0)[0]
    assert p[1]["persistent_timesteps"] == 39
    assert _same(p[0]["ll_steps"], q[0]["ll_steps"]) and _same(p[2], q[2]) and _same(p[3], q[3])
    o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    o.reset()
    assert _same(p[0]["ll_steps"], o.run(U, Y, 0.0, ll_steps=True)["ll_steps"])


def test_failed_bound_test_leaves_the_persistent_launch_and_resumes():
    """an outlier measurement fails the bound test in the middle of the run: every block leaves at that step, the host redoes it
    in exact form with ordinary launches and starts a new persistent launch for the rest"""
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, 50, seed=6)
    Y = Y.copy()
    Y[20] += 40.0
    Y[33] -= 35.0
    cfg = cfg_of(model, 8000, S.RESAMPLE_SYSTEMATIC, 0.5, seed=9)
    p = _run(cfg, U, Y, True)[0]
    q = _run(cfg, U, Y, False)[0]
    # (timesteps are counted as enqueued: a launch that a failed test cuts short is enqueued for the whole rest of the run)
    assert p[1]["persistent_timesteps"] >= 49 and p[1]["fused_launches"] >= 4, p[1]
    assert _same(p[0]["ll_steps"], q[0]["ll_steps"]) and _same(p[2], q[2]) and np.array_equal(p[5], q[5])
    o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    o.reset()
    ro = o.run(U, Y, 1.0, ll_steps=True)
    assert _same(p[0]["ll_steps"], ro["ll_steps"]) and o.exact_steps() >= 2


def test_other_dimensions_and_covariance_kinds():
    rng = np.random.default_rng(0)
    for nx, ny in ((1, 1), (3, 2), (4, 4)):
        A = 0.9 * np.eye(nx) + 0.05 * rng.standard_normal((nx, nx))
        B = rng.standard_normal((nx, 1))
        Cm = rng.standard_normal((ny, nx))
        g = S.make_gaussian
        model = S.make_lg_model(A, B, Cm, g(np.zeros(nx), 0.04 * np.eye(nx) + 0.01), g(np.zeros(ny), np.full(ny, 0.5)), g(np.zeros(nx), 2.0))
        U = rng.standard_normal((30, 1))
        Y = rng.standard_normal((30, ny))
        cfg = cfg_of(model, 7000, S.RESAMPLE_SYSTEMATIC, 0.4, seed=2)
        p = _run(cfg, U, Y, True)[0]
        q = _run(cfg, U, Y, False)[0]
        assert p[1]["persistent_timesteps"] >= 29      # more when failed bound tests cut launches short (counted as enqueued)
        assert _same(p[0]["ll_steps"], q[0]["ll_steps"]) and _same(p[2], q[2])


def test_filters_beyond_the_resident_set_keep_the_per_launch_path():
    model = M.lg_test_model()
    _, U, Y = M.simulate_lg(model, 6, seed=4)
    os.environ["LLPF_PERSIST"] = "1"
    try:
        g = _capi.FilterHandle(cfg_of(model, 1200000, S.RESAMPLE_SYSTEMATIC, 1.0, seed=1))
        g.reset()
        r = g.run(U, Y, 1.0)
    finally:
        del os.environ["LLPF_PERSIST"]
    assert np.isfinite(r["ll"]) and g.last_run_stats()["persistent_timesteps"] == 0
