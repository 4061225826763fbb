"""Oracle of the Rao-Blackwellized particle filter with per-particle covariance (LLPF_MODEL_RB_BILINEAR; reference
src/rbpf.jl:163-283 with An a function of the state, i.e. the !singleR branches).  Pinned the way the reference pins its
RBPF (test/test_rbpf.jl: ll_RBPF ~ ll_KF) plus a stronger identity: with a constant coupling the per-particle recursion
must reproduce the shared-covariance restatement (itself pinned against the Kalman filter) to rounding."""
import numpy as np

import oracle_binding as ob
import rbfull_models as M
from llpf_amd import _structs as S


def _cfg(m, N, thr=0.5, seed=3):
    return S.make_config(m, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, seed, 0)


def test_constant_coupling_equals_shared_covariance_recursion():
    m, mats = M.linear_case(1, 2, 1, seed=0, state_dependent=False)
    lin = S.make_rb_model(mats["Fn"], mats["Bn"], mats["An"][0], mats["Al"], mats["Bl"], mats["Gn"], mats["Cl"],
                          m.dynamics_density, mats["R1l"], m.measurement_density, m.initial_density, m.linear_initial)
    U, Y = M.simulate_io(m, 80)
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        a = ob.OracleFilter(_cfg(m, 400), order); b = ob.OracleFilter(_cfg(lin, 400), order)
        a.reset(); b.reset()
        ra = a.run(U, Y, 0.0, ll_steps=True); rb = b.run(U, Y, 0.0, ll_steps=True)
        assert a.resample_count() == b.resample_count() > 0
        assert np.max(np.abs(ra["ll_steps"] - rb["ll_steps"])) < 1e-12
        xl, R = a.rb_linear_state()
        assert np.max(np.abs(R - R[0])) == 0.0                       # data independent: every particle ends with the same R
        assert np.max(np.abs(R[0] - b.rb_R())) < 1e-14
        xb = b.particles()
        assert a.particles().shape == xb.shape == (400, 3)            # an RBParticle indexes like [xn; xl] in both forms
        assert np.max(np.abs(a.particles() - xb)) < 1e-12 and np.max(np.abs(xl - xb[:, 1:])) < 1e-12


def test_all_linear_matches_kalman_filter():
    """test/test_rbpf.jl:87-139 in the per-particle form: linear f_n / g and a linear substate that does not drive the
    nonlinear one (An = 0, run through the general formulas: Nt = R1n, L = 0) is a linear-Gaussian system; ll_RBPF ~
    ll_KF of the joint model, rtol 1e-2 as in the reference.  (With An != 0 the reference draws xn' with covariance R1n
    instead of An R An' + R1n, src/rbpf.jl:217, so its likelihood converges to a slightly different value than the
    Kalman filter's — 1-2 % on these systems; the restatement follows the reference, see the identity test above.)"""
    m, mats = M.linear_case(2, 2, 2, seed=2, state_dependent=False, an_scale=0.0)
    nn, nl = 2, 2
    A = np.block([[mats["Fn"], mats["An"][0]], [np.zeros((nl, nn)), mats["Al"]]])
    B = np.vstack([mats["Bn"], mats["Bl"]])
    Cm = np.hstack([mats["Gn"], mats["Cl"]])
    blk = lambda a, b: np.block([[a, np.zeros((a.shape[0], b.shape[1]))], [np.zeros((b.shape[0], a.shape[1])), b]])
    R1 = blk(S.gaussian_cov_matrix(m.dynamics_density), mats["R1l"])
    P0 = blk(S.gaussian_cov_matrix(m.initial_density), S.gaussian_cov_matrix(m.linear_initial))
    x0 = np.concatenate([S.gaussian_mean(m.initial_density), S.gaussian_mean(m.linear_initial)])
    joint = S.make_lg_model(A, B, Cm, S.make_gaussian(np.zeros(4), R1), m.measurement_density, S.make_gaussian(x0, P0))
    U, Y = M.simulate_io(m, 300)
    llkf = ob.kalman_loglik(joint, U, Y)
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        o = ob.OracleFilter(_cfg(m, 2000, thr=0.3), order)
        o.reset()
        assert abs(o.run(U, Y, 0.0)["ll"] - llkf) < 1e-2 * abs(llkf)


def test_state_dependent_coupling_orders_agree_and_covariances_differ():
    for shape in ((1, 2, 1), (2, 2, 2), (4, 8, 2), (2, 3, 3), (4, 8, 4)):        # ny = 3, 4: round 5 (LLPF_RBF_MAXY 2 -> 4)
        m, _ = M.linear_case(*shape, seed=1)
        U, Y = M.simulate_io(m, 40)
        lls, Rs = [], []
        for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
            o = ob.OracleFilter(_cfg(m, 600), order)
            o.reset()
            lls.append(o.run(U, Y, 0.0, ll_steps=True)["ll_steps"])
            xl, R = o.rb_linear_state()
            Rs.append(R)
            assert o.resample_count() > 0 and np.all(np.isfinite(lls[-1]))
            assert np.linalg.eigvalsh(R).min() > 0 and np.max(np.abs(R - np.transpose(R, (0, 2, 1)))) == 0.0
            assert np.max(np.abs(R - R[0])) > 1e-5                    # state-dependent An: one Riccati recursion per particle
        assert np.max(np.abs(lls[0] - lls[1])) < 1e-10
        assert np.max(np.abs(Rs[0] - Rs[1])) < 1e-10


def test_quadtank_coupled_case_runs_in_both_orders():
    """BASELINE config C5's model: quad-tank levels + 8 linear modes with a state-dependent coupling."""
    m = M.quadtank_case()
    U, Y = M.simulate_io(m, 25)
    lls = []
    for order in (ob.ORDER_REFERENCE, ob.ORDER_DEVICE):
        o = ob.OracleFilter(_cfg(m, 1500), order)
        o.reset()
        r = o.run(U, Y, 0.0, ll_steps=True)
        lls.append(r["ll_steps"])
        xl, R = o.rb_linear_state()
        assert xl.shape == (1500, 8) and R.shape == (1500, 8, 8) and np.linalg.eigvalsh(R).min() > 0
        assert np.max(np.abs(R - R[0])) > 1e-5
    assert np.all(np.isfinite(lls[0])) and np.max(np.abs(lls[0] - lls[1])) < 1e-10


def test_single_steps_and_missing_measurement():
    m, _ = M.linear_case(2, 2, 2, seed=4)
    U, Y = M.simulate_io(m, 12)
    Y[5] = np.nan
    a = ob.OracleFilter(_cfg(m, 300), ob.ORDER_DEVICE); b = ob.OracleFilter(_cfg(m, 300), ob.ORDER_DEVICE)
    a.reset(); b.reset()
    ll_steps = b.run(U, Y, 1.0, ll_steps=True)["ll_steps"]
    for k in range(12):
        assert a.update(U[k], Y[k], (1.0 + k) * 1.0) == ll_steps[k]
    assert np.array_equal(a.particles(), b.particles())
    assert np.array_equal(a.rb_linear_state()[1], b.rb_linear_state()[1])


def test_particles_history_and_means_are_xn_xl():
    """particles(pf), the x history of forward_trajectory and weighted_mean carry [xn; xl] (RBParticle, src/rbpf.jl:24-30);
    set_particles installs both parts."""
    m, _ = M.linear_case(2, 2, 2, seed=6)
    U, Y = M.simulate_io(m, 8)
    o = ob.OracleFilter(_cfg(m, 200), ob.ORDER_REFERENCE)
    o.reset()
    r = o.run(U, Y, 0.0, xmean=True, history=True)
    assert r["x"].shape == (8, 200, 4) and r["xmean"].shape == (8, 4)
    assert np.allclose(r["xmean"], np.einsum("tnd,tn->td", r["x"], r["we"]), rtol=1e-12, atol=1e-14)
    x = o.particles()
    assert np.array_equal(x[:, 2:], o.rb_linear_state()[0])
    o.set_particles(x + 1.0)
    assert np.array_equal(o.particles(), x + 1.0) and np.array_equal(o.rb_linear_state()[0], x[:, 2:] + 1.0)
    assert np.allclose(o.weighted_mean(), (o.particles() * o.expweights()[:, None]).sum(0))

