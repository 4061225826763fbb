"""Model / data builders shared by the tests, bench.py and __graft_entry__.smoke()."""
import numpy as np

from llpf_amd import _structs as S


def lg_c1_model(seed=0):
    """BASELINE config C1: examples/example_lineargaussian.jl:10-23 (nx=nu=ny=2; the reference draws
    Tr, B, C and mean(d0) unseeded, this fixes one draw)."""
    rng = np.random.default_rng(seed)
    nx = nu = ny = 2
    Tr = rng.standard_normal((nx, nx))
    A = Tr @ np.diag(np.linspace(0.5, 0.95, nx)) @ np.linalg.inv(Tr)
    B = rng.standard_normal((nx, nu))
    Cm = rng.standard_normal((ny, nx))
    df = S.make_gaussian(np.zeros(nx), np.ones(nx))            # MvNormal(Diagonal(ones(nx)))  -> PDiagMat
    dg = S.make_gaussian(np.zeros(ny), np.ones(ny))
    d0 = S.make_gaussian(rng.standard_normal(nx), 4.0)         # MvNormal(randn(nx), 2.0^2*I)  -> ScalMat
    return S.make_lg_model(A, B, Cm, df, dg, d0, 1.0)


def lg_test_model(sigma_f=0.1, kind_f=S.COV_SCAL):
    """BASELINE config C2 / the reference's end-to-end test system, test/runtests.jl:255-266
    (nx=2, nu=1, ny=1): df = N(0, 0.1^2 I), dg = N(0, 1), d0 = N(m0, 2^2 I)."""
    A = np.array([[0.97043, -0.097368], [0.09736, 0.970437]])
    B = np.array([[0.1], [0.0]])
    Cm = np.array([[0.0, 1.0]])
    df = S.make_gaussian(np.zeros(2), np.full(2, sigma_f ** 2) if kind_f == S.COV_DIAG else sigma_f ** 2, kind_f)
    dg = S.make_gaussian(np.zeros(1), np.ones(1))              # mvnormal(p, 1.0): Diagonal -> PDiagMat
    d0 = S.make_gaussian(np.array([0.3, -0.5]), 4.0)
    return S.make_lg_model(A, B, Cm, df, dg, d0, 1.0)


def quadtank_model():
    """BASELINE config C3: examples/example_quadtank.jl:8-44,128-133."""
    df = S.make_gaussian(np.zeros(4), np.full(4, 0.1))          # Diagonal([0.1,...])
    dg = S.make_gaussian(np.zeros(2), np.full(2, 1e-4))         # Diagonal((1e-2)^2 * ones(ny))
    d0 = S.make_gaussian(np.array([2.0, 2.0, 3.0, 3.0]), np.full(4, 0.1))
    return S.make_quadtank_model(df, dg, d0, 1.0, 2)


def simulate_lg(model, T, seed=1):
    """simulate(pf, T, du) semantics (src/filtering.jl:457-477): x1 = mean(d0), y_t = C x_t + e_t,
    x_{t+1} = A x_t + B u_t + w_t, u_t ~ N(0, I)."""
    rng = np.random.default_rng(seed)
    nx, nu, ny = model.nx, model.nu, model.ny
    A = np.array(model.A[:nx * nx]).reshape(nx, nx)
    B = np.array(model.B[:nx * nu]).reshape(nx, nu)
    Cm = np.array(model.C[:ny * nx]).reshape(ny, nx)
    Lf = np.linalg.cholesky(S.gaussian_cov_matrix(model.dynamics_density))
    Lg = np.linalg.cholesky(S.gaussian_cov_matrix(model.measurement_density))
    x = S.gaussian_mean(model.initial_density).copy()
    U = rng.standard_normal((T, nu))
    Y = np.zeros((T, ny))
    X = np.zeros((T, nx))
    for t in range(T):
        X[t] = x
        Y[t] = Cm @ x + Lg @ rng.standard_normal(ny)
        x = A @ x + B @ U[t] + Lf @ rng.standard_normal(nx)
    return X, U, Y


def quadtank_data(T, seed=2):
    """Input square wave and noisy rollout of examples/example_quadtank.jl:37-44 extended to T steps
    (rollout calls the dynamics with t = i*Ts, src/filtering.jl:527-533)."""
    import llpf_amd
    rng = np.random.default_rng(seed)
    dyn = llpf_amd.QuadTankDynamics(supersample=2)
    t = np.arange(T)
    u1 = 0.25 * np.sign(np.sin(2 * np.pi / 200 * t)) + 0.25
    U = np.stack([u1, u1], axis=1)
    x = np.array([2.0, 2.0, 3.0, 3.0])
    Y = np.zeros((T, 2))
    for i in range(T):
        Y[i] = x[:2] + 0.01 * rng.standard_normal(2)
        x = dyn(x, U[i], None, (i + 1) * 1.0, 1.0)
    return U, Y
