"""weighted_quantile (reference src/filtering.jl:583-595 = StatsBase.quantile(v, ProbabilityWeights(we), q)).

StatsBase is not vendored by the reference and Julia is not in the image: the oracle's restatement (oracle/llpf_oracle.c:
orc_weighted_quantile) is pinned here by (i) the property StatsBase's own tests assert — equal weights give the ordinary quantile
(Hyndman-Fan type 7 = numpy's default), (ii) a second, literal restatement of the published algorithm written independently below in plain
Python, (iii) hand-computed small cases.  The engine's device version is held to the oracle in the -m gpu part."""
import numpy as np
import pytest

import oracle_binding as ob
import models as M
from llpf_amd import _structs as S


def statsbase_quantile(v, w, p):
    """src/weights.jl `quantile(v, w::AbstractWeights, p)`, the branch for non-frequency weights, line by line"""
    v, w, p = list(map(float, v)), list(map(float, w)), list(map(float, p))
    wsum = float(np.sum(np.asarray(w)))
    vw = sorted((a, b) for a, b in zip(v, w) if b != 0.0)
    N = len(vw)
    order = sorted(range(len(p)), key=lambda i: p[i])
    out = [vw[-1][0]] * len(p)
    if any(a != a for a in v):
        return [float("nan")] * len(p)
    Sk = Skold = vk = vkold = 0.0
    k = 0
    w1 = vw[0][1]
    for i in order:
        h = p[i] * (wsum - w1) + w1
        while Sk <= h:
            k += 1
            if k > N:
                return out
            Skold, vkold = Sk, vk
            vk, wk = vw[k - 1]
            Sk += wk
        out[i] = vkold + (h - Skold) / (Sk - Skold) * (vk - vkold)
    return out


P = [0.0, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0]


@pytest.mark.parametrize("n", [1, 2, 5, 64, 1001])
def test_equal_weights_give_the_ordinary_quantile(n):
    rng = np.random.default_rng(n)
    v = rng.standard_normal(n)
    for w in (np.ones(n), np.full(n, 1.0 / n)):
        np.testing.assert_allclose(ob.weighted_quantile(v, w, P), np.quantile(v, P), rtol=1e-12, atol=1e-13)


def test_hand_computed_cases():
    # v = 1, 2, 3 with weights .2, .3, .5: h = .8 p + .2; S = .2, .5, 1
    np.testing.assert_allclose(ob.weighted_quantile([3.0, 1.0, 2.0], [0.5, 0.2, 0.3], [0.0, 0.25, 0.5, 1.0]),
                               [1.0, 1.0 + (0.4 - 0.2) / 0.3, 2.0 + (0.6 - 0.5) / 0.5, 3.0], rtol=1e-15)
    # a particle without weight is not there: same answer as without it, wherever its value lies
    a = ob.weighted_quantile([1.0, 2.0, 3.0], [0.2, 0.3, 0.5], P)
    for extra in (-7.0, 1.5, 2.0, 99.0):
        np.testing.assert_array_equal(ob.weighted_quantile([1.0, extra, 2.0, 3.0], [0.2, 0.0, 0.3, 0.5], P), a)
    # tied values are ordered by their weights (tuples sort lexicographically): .2 | 2 (.3) 2 (.5) vs the crossing inside the tie
    np.testing.assert_allclose(ob.weighted_quantile([1.0, 2.0, 2.0], [0.2, 0.5, 0.3], [0.25]), [1.0 + (0.4 - 0.2) / 0.3], rtol=1e-15)
    assert np.all(np.isnan(ob.weighted_quantile([1.0, np.nan, 2.0], [0.3, 0.0, 0.7], P)))
    with pytest.raises(ValueError):
        ob.weighted_quantile([1.0, 2.0], [0.0, 0.0], [0.5])


@pytest.mark.parametrize("seed", range(6))
def test_against_the_literal_restatement(seed):
    rng = np.random.default_rng(seed)
    n = [3, 17, 200, 200, 5000, 5000][seed]
    v = rng.standard_normal(n)
    w = rng.random(n) ** 4
    if seed % 2:
        w[rng.random(n) < 0.3] = 0.0                      # dropped particles
        v[rng.integers(0, n, n // 4)] = v[0]              # ties
    w /= w.sum()
    p = rng.random(7).tolist() + [0.0]
    # (the quantile is piecewise linear in h with slope gap / weight: the two sides' sums differ in their last bits — np.sum against the
    # pairwise sum — so they agree to 1e-10 of the spread, not to an ulp)
    np.testing.assert_allclose(ob.weighted_quantile(v, w, p), statsbase_quantile(v, w, p), rtol=1e-10, atol=1e-10)
    # p = 1: h = sum(w), reached by the running sum or missed by its rounding — the largest particle that carries weight, or a hair below it
    top = max(a for a, b in zip(v, w) if b != 0.0)
    for f in (ob.weighted_quantile, statsbase_quantile):
        assert abs(f(v, w, [1.0])[0] - top) <= 1e-7


def test_filter_accessor_of_the_oracle():
    model = M.lg_test_model()
    cfg = S.make_config(model, 500, resample_threshold=0.5, seed=3)
    o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
    o.reset()
    o.correct([0.1], [0.4], 0.0)
    q = o.weighted_quantile([0.1, 0.5, 0.9])
    x, we = o.particles(), o.expweights()
    for d in range(2):
        np.testing.assert_array_equal(q[:, d], ob.weighted_quantile(x[:, d], we, [0.1, 0.5, 0.9]))
    assert np.all(q[0] < q[1]) and np.all(q[1] < q[2])


ENDS = [0.0, 1e-300, 1e-30, 1e-12, 1.0]


def _wide_span_case(n, seed):
    """values and log-weights whose exp-weights span far more than 2^96, arranged so that StatsBase's own arithmetic is exact where the
    ends are decided: the two smallest values carry comparable (tiny) weights, the largest value carries real mass"""
    rng = np.random.default_rng(seed)
    v = np.sort(rng.standard_normal(n))
    k = rng.integers(0, 130, n).astype(np.float64)
    k[0], k[1], k[-1] = 120.0, 119.0, 2.0
    perm = rng.permutation(n)
    return v[perm], (-k * np.log(2.0))[perm]


def test_device_order_against_statsbase_order_on_the_cpu():
    """orc_weighted_quantile_dev (what the engine computes: integer crossing at a per-quantile scale) against the restated StatsBase:
    interior probabilities to the rounding of the running sums, the ends p in {0, 1e-300} EXACTLY on a weight vector spanning 2^120 —
    p -> 0 is the smallest value again (round 5's 2^-96-only sums returned a later one)."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 50, 3000):
        v = rng.standard_normal(n)
        w = rng.random(n) ** 4
        if n > 3:
            w[rng.random(n) < 0.3] = 0.0
            v[rng.integers(0, n, n // 4)] = v[0]
        if not w.any():
            w[0] = 1.0
        w /= w.sum()
        a, b = ob.weighted_quantile(v, w, P), ob.weighted_quantile_dev(v, w, P)
        assert np.all(np.abs(a - b) <= 1e-10 * (np.ptp(v) + 1.0)), (n, np.abs(a - b).max())
    v, lw = _wide_span_case(400, 9)
    w = np.exp(lw - lw.max()); w /= w.sum()
    assert w.max() / w[w > 0].min() > 2.0 ** 100
    a, b = ob.weighted_quantile(v, w, ENDS), ob.weighted_quantile_dev(v, w, ENDS)
    np.testing.assert_array_equal(b[[0, 1]], a[[0, 1]])
    # (p = 1: h = 1 (wsum - w1) + w1 is formed in fp64 on both sides and reaches the total or misses it by its last rounding: the
    #  largest value, or "a hair below" it)
    assert b[0] == v.min() and abs(b[4] - v.max()) <= 1e-12 and abs(a[4] - v.max()) <= 1e-12
    np.testing.assert_allclose(b, a, rtol=1e-12, atol=1e-12)
    assert np.all(np.isnan(ob.weighted_quantile_dev([1.0, np.nan], [0.5, 0.5], [0.5])))
    # tuple order inside a tie, a dropped particle, -0.0 below +0.0
    np.testing.assert_allclose(ob.weighted_quantile_dev([1.0, 2.0, 2.0, 7.0], [0.2, 0.5, 0.3, 0.0], [0.25]), [1.0 + (0.4 - 0.2) / 0.3], rtol=1e-15)
    # -0.0 sorts below +0.0 (isless): with the ties' weights in the other order the crossing would be elsewhere
    np.testing.assert_array_equal(ob.weighted_quantile_dev([0.0, -0.0, 1.0], [0.1, 0.5, 0.4], [0.3]), ob.weighted_quantile([0.0, -0.0, 1.0], [0.1, 0.5, 0.4], [0.3]))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 2, 777, 100000, 1_200_000])
def test_engine_against_the_oracle(N):
    """llpf_weighted_quantile (round 6: radix selection over weight histograms, no sort, no library) — bit for bit the device-order
    oracle (orc_weighted_quantile_dev) for every probability including the ends, and the restated StatsBase (sequential fp64 sums) to
    1e-10 of the spread of the particles."""
    from llpf_amd import _capi
    model = M.lg_test_model()
    cfg = S.make_config(model, N, resample_threshold=0.5, seed=21)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    ob.set_threads(16)
    for k in range(3):
        g.update([0.1], [0.4 + k], float(k)); o.update([0.1], [0.4 + k], float(k))
    g.correct([0.1], [0.2], 3.0); o.correct([0.1], [0.2], 3.0)
    ob.set_threads(1)
    pp = P + [1e-300, 1e-30, 1e-12, 0.3333, 0.999999]
    qg, qo = g.weighted_quantile(pp), o.weighted_quantile(pp)
    assert np.array_equal(qg.view(np.uint64), qo.view(np.uint64)), np.abs(qg - qo).max()
    x, we = o.particles(), o.expweights()
    spread = np.ptp(x, axis=0) + 1.0
    for d in range(2):
        ref = ob.weighted_quantile(x[:, d], we, P[:-1])
        assert np.all(np.abs(qg[:len(P) - 1, d] - ref) <= 1e-10 * spread[d])
    # p = 1: the largest value, or below it by the rounding of the fp64 h over the (possibly tiny) weight of the largest particle
    top = g.weighted_quantile([1.0])[0]
    assert np.all(top <= x[we > 0].max(axis=0) + 1e-12 * spread) and np.all(top >= qg[-1] - 1e-12 * spread)
    np.testing.assert_array_equal(g.weighted_quantile([0.0])[0], x[we > 0].min(axis=0))
    if N < 4:
        return
    # particles without weight and tied values: installed state
    rng = np.random.default_rng(N)
    x = rng.standard_normal((N, 2)); x[rng.integers(0, N, N // 3)] = x[0]
    w = rng.standard_normal(N) * 3; w[rng.random(N) < 0.4] = -np.inf
    for h in (g, o):
        h.set_particles(x); h.set_weights(w)
    qg, qo = g.weighted_quantile(pp), o.weighted_quantile(pp)
    assert np.array_equal(qg.view(np.uint64), qo.view(np.uint64)), np.abs(qg - qo).max()
    # exp-weights spanning 2^120: the ends are the smallest / largest particle that carries ANY weight, exactly, on both sides
    n2 = min(N, 5000)
    v, lw = _wide_span_case(n2, N)
    x2 = np.zeros((N, 2)); x2[:n2, 0] = v; x2[:n2, 1] = -v
    w2 = np.full(N, -np.inf); w2[:n2] = lw
    for h in (g, o):
        h.set_particles(x2); h.set_weights(w2)
    qg, qo = g.weighted_quantile(ENDS), o.weighted_quantile(ENDS)
    assert np.array_equal(qg.view(np.uint64), qo.view(np.uint64))
    assert qg[0, 0] == v.min() and abs(qg[-1, 0] - v.max()) <= 1e-12
    we2 = o.expweights()
    ref = ob.weighted_quantile(x2[:, 0], we2, ENDS)
    np.testing.assert_array_equal(qg[[0, 1], 0], ref[[0, 1]])
    np.testing.assert_allclose(qg[:, 0], ref, rtol=1e-12, atol=1e-12)
    x[N // 2, 1] = np.nan
    g.set_particles(x)
    qg = g.weighted_quantile([0.5])
    assert np.isnan(qg[0, 1]) and np.isfinite(qg[0, 0])
    with pytest.raises(Exception):
        g.weighted_quantile([1.5])


@pytest.mark.gpu
@pytest.mark.parametrize("N,nq", [(3000, 3), (40000, 6)])
def test_quantiles_from_inside_the_run_loop(N, nq):
    """llpf_run's xquant output (round 6): weighted_quantile(sol, q) of every timestep computed on the device inside the run loop — the
    state forward_trajectory records between correct! and predict! (src/filtering.jl:357-359, 583-595) — against the device-order
    oracle stepped by hand, bit for bit; the run with the output gives the same filter as the run without; the Python mirror serves
    weighted_quantile(sol, q) from it."""
    import llpf_amd
    from llpf_amd import _capi
    model = M.lg_test_model()
    T = 25
    _, U, Y = M.simulate_lg(model, T, seed=3)
    Y[7] = np.nan
    q = [0.05, 0.5, 0.95] if nq == 3 else [0.0, 0.01, 0.25, 0.5, 0.9, 1.0]
    cfg = S.make_config(model, N, resample_threshold=0.5, seed=33)
    g = _capi.FilterHandle(cfg); g2 = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    for h in (g, g2, o):
        h.reset()
    rq = g.run(U, Y, 0.0, ll_steps=True, quantiles=q)
    r0 = g2.run(U, Y, 0.0, ll_steps=True)
    assert np.array_equal(rq["ll_steps"].view(np.uint64), r0["ll_steps"].view(np.uint64))
    ref = np.zeros((T, 2, len(q)))
    for k in range(T):
        o.correct(U[k], Y[k], k * 1.0)
        ref[k] = o.weighted_quantile(q).T
        o.predict(U[k], k * 1.0)
    assert np.array_equal(rq["xquant"].view(np.uint64), ref.view(np.uint64)), np.abs(rq["xquant"] - ref).max()
    # the captured run loop (the shape seen a second and third time; a re-seeded handle restarts its counters, so these two runs draw the
    # same noise as each other) gives one and the same output, and the filter of a run without the output
    r23 = []
    for _ in range(2):
        g.seed(33); g.reset()
        r23.append(g.run(U, Y, 0.0, ll_steps=True, quantiles=q))
    assert np.array_equal(r23[0]["xquant"].view(np.uint64), r23[1]["xquant"].view(np.uint64))
    g2.seed(33); g2.reset()
    assert np.array_equal(g2.run(U, Y, 0.0, ll_steps=True)["ll_steps"].view(np.uint64), r23[1]["ll_steps"].view(np.uint64))
    # the mirror of the reference's API
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]]); B = np.array([[0.1], [0.0]]); Cm = np.array([[0.0, 1.0]])
    pf = llpf_amd.ParticleFilter(N, llpf_amd.LinearDynamics(A, B), llpf_amd.LinearMeasurement(Cm), llpf_amd.MvNormal(np.zeros(2), 0.01),
                                 llpf_amd.MvNormal(np.zeros(1), 1.0), llpf_amd.MvNormal(np.array([1.0, 1.0]), 4.0), rng=3, resample_threshold=0.5)
    sol = llpf_amd.forward_trajectory(pf, U, Y, quantiles=q)
    wq = llpf_amd.weighted_quantile(sol, q)
    assert len(wq) == T and wq[0].shape == (2, len(q))
    np.testing.assert_array_equal(np.stack(wq), sol.xquant)
    np.testing.assert_array_equal(np.stack(llpf_amd.weighted_quantile(sol, q[1])), sol.xquant[:, :, 1])
    host = llpf_amd.weighted_quantile(sol.x, sol.we, q)                  # the numpy restatement on the returned history
    keep = [i for i, v in enumerate(q) if v < 1.0]                          # (p = 1 is ill-conditioned in StatsBase's own formula: see above)
    assert np.all(np.abs(np.stack(host)[:, :, keep] - np.stack(wq)[:, :, keep]) <= 1e-9 * (np.ptp(sol.x) + 1.0))
    with pytest.raises(_capi.LLPFError):
        g.run(U, Y, 0.0, quantiles=[0.5, 1.5])


def test_host_function_of_the_python_mirror():
    """weighted_quantile(x, we, q) of the API mirror (numpy, for returned histories) against the oracle's restatement"""
    import llpf_amd
    rng = np.random.default_rng(8)
    T, N, nx = 3, 400, 2
    x = rng.standard_normal((T, N, nx))
    we = rng.random((T, N)) ** 3
    we[:, ::7] = 0.0
    we /= we.sum(axis=1, keepdims=True)
    q = [0.05, 0.5, 0.95]
    out = llpf_amd.weighted_quantile(x, we, q)
    assert len(out) == T and out[0].shape == (nx, 3)              # the reference's nesting: [t][state][q] (src/filtering.jl:592-594)
    for t in range(T):
        for d in range(nx):
            np.testing.assert_allclose(out[t][d], ob.weighted_quantile(x[t, :, d], we[t], q), rtol=1e-10, atol=1e-12)
    med = llpf_amd.weighted_quantile(x, we, 0.5)
    np.testing.assert_allclose(med[1], out[1][:, 1], rtol=0, atol=0)
