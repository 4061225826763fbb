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


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 2, 777, 100000])
def test_engine_against_the_oracle(N):
    """llpf_weighted_quantile: rocPRIM sorts + fixed-point running sums on the device against the oracle's sequential fp64 sums: 1e-10 of the
    spread of the particles (the quantile is piecewise linear in the running sum with slope gap / weight; the two sides' sums differ in their
    last bits: the device's are exact to 2^-96 per term, the sequential fp64 sum is not)."""
    from llpf_amd import _capi
    model = M.lg_test_model()
    cfg = S.make_config(model, N, resample_threshold=0.5, seed=21)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    for k in range(3):
        g.update([0.1], [0.4 + k], float(k)); o.update([0.1], [0.4 + k], float(k))
    g.correct([0.1], [0.2], 3.0); o.correct([0.1], [0.2], 3.0)
    qg, qo = g.weighted_quantile(P[:-1]), o.weighted_quantile(P[:-1])
    spread = np.ptp(o.particles(), axis=0) + 1.0
    assert np.all(np.abs(qg - qo) <= 1e-10 * spread), np.abs(qg - qo).max()
    we = o.expweights()
    np.testing.assert_array_equal(g.weighted_quantile([1.0])[0], o.particles()[we > 0].max(axis=0))      # h = the device's own total: the top, exactly
    # particles without weight and tied values: installed state
    rng = np.random.default_rng(N)
    x = rng.standard_normal((N, 2)); x[rng.integers(0, N, N // 3)] = x[0]
    w = rng.standard_normal(N) * 3; w[rng.random(N) < 0.4] = -np.inf
    if not np.isfinite(w).any():
        w[0] = 0.0
    for h in (g, o):
        h.set_particles(x); h.set_weights(w)
    qg, qo = g.weighted_quantile(P[:-1]), o.weighted_quantile(P[:-1])
    assert np.all(np.abs(qg - qo) <= 1e-10 * (np.ptp(x, axis=0) + 1.0)), np.abs(qg - qo).max()
    x[N // 2, 1] = np.nan
    g.set_particles(x)
    qg = g.weighted_quantile([0.5])
    assert np.isnan(qg[0, 1]) and np.isfinite(qg[0, 0])
    with pytest.raises(Exception):
        g.weighted_quantile([1.5])


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
