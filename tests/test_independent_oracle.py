"""The second, independent restatement of the path (oracle/independent.py: plain numpy, numpy.linalg, nothing shared with the engine
or with oracle/llpf_oracle.c but the Philox draws) against (1) its own frozen outputs tests/golden/indep_*.npz, (2) the C oracle in
both arithmetic orders, (3) the HIP engine.  This is what pins the code oracle and engine share — above all the per-particle
Riccati / Kalman recursion of the Rao-Blackwellized filter with a state-dependent coupling (reference src/rbpf.jl:206-221,
src/filtering.jl:100-128), csrc/shared/llpf_rbfull_body.h on both sides — against a second reading of the same reference lines.
Tolerances: the three implementations differ only in rounding (operation order inside numpy / BLAS vs explicit loops): 1e-11 per
step on log-likelihoods, 1e-10 on states; ancestors identical."""
import os

import numpy as np
import pytest

import independent_cases as IC
import oracle_binding as ob

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = IC.cases()
TOL_LL, TOL_X = 1e-11, 1e-10


def _golden(name):
    return np.load(os.path.join(G, "indep_%s.npz" % name))


@pytest.mark.parametrize("name", list(CASES))
def test_independent_restatement_reproduces_its_frozen_outputs(name):
    d, case = _golden(name), CASES[name]
    assert np.array_equal(d["U"], case["U"]) and np.array_equal(d["Y"], case["Y"], equal_nan=True)
    r = IC.run_independent(ob, case)
    assert np.max(np.abs(r["ll_steps"] - d["ll_steps"])) < 1e-12          # numpy / BLAS builds may round differently
    assert np.max(np.abs(r["x_final"] - d["x_final"])) < 1e-11
    assert np.array_equal(r["anc_final"], d["anc_final"]) and int(r["resamples"]) == int(d["resamples"]) > 0
    if case.get("rb"):
        assert np.max(np.abs(r["R_final"] - d["R_final"])) < 1e-13
        R = d["R_final"]
        assert np.max(np.abs(R - R[0])) > 1e-6          # the coupling is state dependent: covariances differ between particles


@pytest.mark.parametrize("order", [ob.ORDER_REFERENCE, ob.ORDER_DEVICE])
@pytest.mark.parametrize("name", list(CASES))
def test_c_oracle_agrees_with_the_independent_restatement(name, order):
    d, case = _golden(name), CASES[name]
    o = IC.oracle_of(ob, case, order)
    o.reset()
    r = o.run(case["U"], case["Y"], case["t0"], ll_steps=True)
    assert np.max(np.abs(r["ll_steps"] - d["ll_steps"])) < TOL_LL
    assert np.max(np.abs(o.particles() - d["x_final"])) < TOL_X
    assert np.array_equal(o.ancestors(), d["anc_final"]) and o.resample_count() == int(d["resamples"])
    if case.get("rb"):
        xl, R = o.rb_linear_state()
        assert np.max(np.abs(R - d["R_final"])) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_engine_agrees_with_the_independent_restatement(name):
    d, case = _golden(name), CASES[name]
    g = IC.engine_of(case)
    g.reset()
    r = g.run(case["U"], case["Y"], case["t0"], ll_steps=True)
    assert np.max(np.abs(r["ll_steps"] - d["ll_steps"])) < TOL_LL
    assert np.max(np.abs(g.particles() - d["x_final"])) < TOL_X
    assert np.array_equal(g.ancestors(), d["anc_final"]) and g.resample_count() == int(d["resamples"])
    if case.get("rb"):
        xl, R = g.rb_linear_state()
        assert np.max(np.abs(R - d["R_final"])) < 1e-12
        assert np.max(np.abs(xl - d["x_final"][:, case["model"].nx:])) < TOL_X
