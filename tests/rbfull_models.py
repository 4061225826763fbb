"""Test models of the Rao-Blackwellized filter with per-particle covariance (LLPF_MODEL_RB_BILINEAR): the shapes the
engine instantiates, (nxn, nxl, ny) = (1,2,1), (2,2,2), (4,8,2) with linear f_n/g and (4,8,2) with the quad-tank."""
import numpy as np

from llpf_amd import _structs as S

g = S.make_gaussian


def _stable(n, rng, rho=0.9):
    A = rng.standard_normal((n, n))
    A = A * (rho / max(abs(np.linalg.eigvals(A))))
    return A


def _spd(n, rng, scale):
    M = rng.standard_normal((n, n))
    return scale * (M @ M.T / n + np.eye(n))


def linear_case(nn, nl, ny, seed=0, state_dependent=True, nu=1, an_scale=0.3):
    """random stable system; An(xn) = An0 + sum_k xn[k] An_k (An_k = 0 when not state_dependent)"""
    rng = np.random.default_rng(100 + seed)
    Fn = _stable(nn, rng, 0.8)
    Al = _stable(nl, rng, 0.9)
    Bn = 0.3 * rng.standard_normal((nn, nu))
    Bl = 0.3 * rng.standard_normal((nl, nu))
    Gn = rng.standard_normal((ny, nn))
    Cl = rng.standard_normal((ny, nl))
    An = np.zeros((nn + 1, nn, nl))
    An[0] = an_scale * rng.standard_normal((nn, nl))
    if state_dependent:
        An[1:] = 0.15 * rng.standard_normal((nn, nn, nl))
    R1n = g(np.zeros(nn), _spd(nn, rng, 0.02))
    R1l = _spd(nl, rng, 0.02)
    R2 = g(np.zeros(ny), _spd(ny, rng, 0.1))
    d0n = g(0.5 * rng.standard_normal(nn), _spd(nn, rng, 0.05))
    d0l = g(0.5 * rng.standard_normal(nl), _spd(nl, rng, 0.3))
    m = S.make_rb_bilinear_model(An, Al, Bl, Cl, R1n, R1l, R2, d0n, d0l, Fn=Fn, Bn=Bn, Gn=Gn)
    mats = dict(Fn=Fn, Al=Al, Bn=Bn, Bl=Bl, Gn=Gn, Cl=Cl, An=An, R1l=R1l)
    return m, mats


def quadtank_case(seed=0):
    """BASELINE config C5: quad-tank levels (4 nonlinear states, RK4 x 2) driven through a state-dependent coupling by 8
    linear states (slow disturbance / actuator modes); outputs = levels 1, 2 + a linear combination of the modes."""
    rng = np.random.default_rng(500 + seed)
    nn, nl, ny = 4, 8, 2
    Al = np.diag(np.linspace(0.80, 0.97, nl)) + 0.02 * rng.standard_normal((nl, nl))
    Bl = 0.05 * rng.standard_normal((nl, 2))
    Cl = 0.3 * rng.standard_normal((ny, nl))
    An = np.zeros((nn + 1, nn, nl))
    An[0] = 0.05 * rng.standard_normal((nn, nl))
    An[1:] = 0.004 * rng.standard_normal((nn, nn, nl))      # levels are O(10): the coupling varies by ~ its own size
    R1n = g(np.zeros(nn), np.diag([0.01, 0.01, 0.01, 0.01]))
    R1l = 0.01 * np.eye(nl) + 0.002 * np.ones((nl, nl))
    R2 = g(np.zeros(ny), np.diag([0.05, 0.05]))
    d0n = g(np.array([10.0, 10.0, 6.0, 6.0]), np.diag([1.0, 1.0, 1.0, 1.0]))
    d0l = g(np.zeros(nl), 0.5 * np.eye(nl))
    return S.make_rb_bilinear_model(An, Al, Bl, Cl, R1n, R1l, R2, d0n, d0l, quadtank={}, Ts=1.0, supersample=2)


def simulate_io(m, T, seed=1):
    """inputs and plausible measurements: a noisy trajectory of the model itself (numpy, test-side only)"""
    rng = np.random.default_rng(seed)
    nn, nl, ny, nu = m.nx, m.rb.nxl, m.ny, m.nu
    U = 0.5 + 0.2 * rng.standard_normal((T, nu)) if nu else np.zeros((T, 0))
    if m.rb.fn_kind == 1:
        U = np.abs(1.0 + 0.1 * rng.standard_normal((T, 2)))
    Al = np.array(m.rb.Al[:nl * nl]).reshape(nl, nl)
    Cl = np.array(m.rb.Cl[:ny * nl]).reshape(ny, nl)
    An = np.array([list(m.rb.An[k][:nn * nl]) for k in range(nn + 1)]).reshape(nn + 1, nn, nl)
    xn = S.gaussian_mean(m.initial_density).copy()
    xl = S.gaussian_mean(m.linear_initial).copy()
    Y = np.zeros((T, ny))
    for t in range(T):
        if m.rb.fn_kind == 1:
            yn = xn[:2]
        else:
            Gn = np.array(m.C[:ny * nn]).reshape(ny, nn)
            yn = Gn @ xn
        Y[t] = yn + Cl @ xl + 0.2 * rng.standard_normal(ny)
        A_t = An[0] + np.tensordot(xn, An[1:], axes=(0, 0))
        if m.rb.fn_kind == 1:
            fn = xn + 0.02 * (np.array([10.0, 10.0, 6.0, 6.0]) - xn)      # a slow pull towards the operating point
        else:
            Fn = np.array(m.A[:nn * nn]).reshape(nn, nn)
            Bn = np.array(m.B[:nn * nu]).reshape(nn, nu) if nu else np.zeros((nn, 0))
            fn = Fn @ xn + (Bn @ U[t] if nu else 0.0)
        xn = fn + A_t @ xl + 0.1 * rng.standard_normal(nn)
        xl = Al @ xl + 0.1 * rng.standard_normal(nl)
    return U, Y
