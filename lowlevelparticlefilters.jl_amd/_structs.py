"""ctypes mirrors of the plain-data structs declared in include/llpf.h (llpf_gaussian,
llpf_model, llpf_config, llpf_run_outputs) and builders from Python values."""
import ctypes as C

import numpy as np

MAX_DIM = 16         # states / outputs (LLPF_MAX_DIM)
MAX_INPUTS = 8       # inputs (LLPF_MAX_INPUTS)

COV_SCAL, COV_DIAG, COV_FULL = 0, 1, 2
MODEL_LINEAR_GAUSSIAN, MODEL_QUADTANK_RK4, MODEL_RB_LINEAR, MODEL_RB_BILINEAR = 0, 1, 2, 3
RESAMPLE_SYSTEMATIC, RESAMPLE_STRATIFIED, RESAMPLE_RESIDUAL = 0, 1, 2
PARTICLE_FILTER, ADVANCED_PARTICLE_FILTER = 0, 1

QT_NAMES = ["k1", "k2", "g", "A1", "A2", "A3", "A4", "a1", "a2", "a3", "a4",
            "gamma1", "gamma2", "t_switch", "a1_factor", "eps"]
QT_COUNT = len(QT_NAMES)


class Gaussian(C.Structure):
    _fields_ = [("dim", C.c_int32), ("kind", C.c_int32),
                ("mu", C.c_double * MAX_DIM), ("cov", C.c_double * (MAX_DIM * MAX_DIM))]


class RBCoupling(C.Structure):
    _fields_ = [("nxl", C.c_int32), ("fn_kind", C.c_int32),
                ("Al", C.c_double * 64), ("Bl", C.c_double * 64), ("Cl", C.c_double * 64),
                ("An", (C.c_double * 32) * 5)]


class Model(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("nx", C.c_int32), ("nu", C.c_int32), ("ny", C.c_int32),
                ("A", C.c_double * (MAX_DIM * MAX_DIM)), ("B", C.c_double * (MAX_DIM * MAX_INPUTS)), ("C", C.c_double * (MAX_DIM * MAX_DIM)),
                ("qt", C.c_double * QT_COUNT),
                ("supersample", C.c_int32), ("nxn", C.c_int32),
                ("Ts", C.c_double),
                ("dynamics_density", Gaussian), ("measurement_density", Gaussian),
                ("initial_density", Gaussian), ("linear_noise", Gaussian), ("linear_initial", Gaussian),
                ("rb", RBCoupling)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("filter_kind", C.c_int32),
                ("n_particles", C.c_int64),
                ("resampling_strategy", C.c_int32), ("device", C.c_int32),
                ("resample_threshold", C.c_double),
                ("seed", C.c_uint64),
                ("model", Model)]


class RunOutputs(C.Structure):
    _fields_ = [("ll_steps", C.POINTER(C.c_double)), ("xmean", C.POINTER(C.c_double)),
                ("x_hist", C.POINTER(C.c_double)), ("w_hist", C.POINTER(C.c_double)),
                ("we_hist", C.POINTER(C.c_double)), ("xcov", C.POINTER(C.c_double)),
                ("xquant", C.POINTER(C.c_double)), ("quant_p", C.POINTER(C.c_double)), ("nq", C.c_int32), ("pad", C.c_int32)]


class MBankInfo(C.Structure):
    _fields_ = [("n_filters", C.c_int32), ("n_shards", C.c_int32), ("n_local_shards", C.c_int32),
                ("first_local_shard", C.c_int32), ("n_local_filters", C.c_int32), ("collective", C.c_int32),
                ("last_run_ms", C.c_double), ("last_collective_ms", C.c_double), ("resample_count", C.c_int64)]


def make_gaussian(mu, cov, kind=None):
    """N(mu, cov).  cov: python float -> ScalMat(sigma^2 = cov); 1-D array -> PDiagMat(diag = cov);
    2-D array -> PDMat(cov).  `kind` forces a storage kind (e.g. a diagonal 2-D matrix as FULL)."""
    g = Gaussian()
    mu = np.atleast_1d(np.asarray(mu, dtype=np.float64))
    n = mu.size
    if n < 1 or n > MAX_DIM:
        raise ValueError("dimension must be in 1..%d" % MAX_DIM)
    g.dim = n
    for i in range(n):
        g.mu[i] = float(mu[i])
    cov_a = np.asarray(cov, dtype=np.float64)
    if kind is None:
        kind = {0: COV_SCAL, 1: COV_DIAG, 2: COV_FULL}[cov_a.ndim]
    g.kind = kind
    if kind == COV_SCAL:
        g.cov[0] = float(cov_a.reshape(-1)[0])
    elif kind == COV_DIAG:
        d = cov_a if cov_a.ndim == 1 else np.diag(cov_a)
        if d.size != n:
            raise ValueError("diag length mismatch")
        for i in range(n):
            g.cov[i] = float(d[i])
    else:
        S = cov_a if cov_a.ndim == 2 else np.diag(np.broadcast_to(cov_a, (n,)))
        if S.shape != (n, n):
            raise ValueError("cov shape mismatch")
        for i in range(n):
            for j in range(n):
                g.cov[i * n + j] = float(S[i, j])
    return g


def gaussian_cov_matrix(g):
    n = g.dim
    if g.kind == COV_SCAL:
        return np.eye(n) * g.cov[0]
    if g.kind == COV_DIAG:
        return np.diag([g.cov[i] for i in range(n)])
    return np.array([[g.cov[i * n + j] for j in range(n)] for i in range(n)])


def gaussian_mean(g):
    return np.array([g.mu[i] for i in range(g.dim)])


QUADTANK_DEFAULTS = dict(k1=1.6, k2=1.6, g=9.81, A1=4.9, A2=4.9, A3=4.9, A4=4.9,
                         a1=0.03, a2=0.03, a3=0.03, a4=0.03, gamma1=0.2, gamma2=0.2,
                         t_switch=500.0, a1_factor=2.0, eps=1e-3)


def make_lg_model(A, B, Cm, df, dg, d0, Ts=1.0):
    A = np.atleast_2d(np.asarray(A, dtype=np.float64))
    nx = A.shape[0]
    Cm = np.atleast_2d(np.asarray(Cm, dtype=np.float64))
    ny = Cm.shape[0]
    if B is None:
        B = np.zeros((nx, 0))
    B = np.asarray(B, dtype=np.float64).reshape(nx, -1)
    nu = B.shape[1]
    if A.shape != (nx, nx) or Cm.shape != (ny, nx):
        raise ValueError("A must be nx x nx and C ny x nx")
    if max(nx, ny) > MAX_DIM or nu > MAX_INPUTS:
        raise ValueError("more than %d states / outputs or %d inputs are not supported" % (MAX_DIM, MAX_INPUTS))
    m = Model()
    m.model_id = MODEL_LINEAR_GAUSSIAN
    m.nx, m.nu, m.ny = nx, nu, ny
    for r in range(nx):
        for c in range(nx):
            m.A[r * nx + c] = A[r, c]
        for c in range(nu):
            m.B[r * nu + c] = B[r, c]
    for r in range(ny):
        for c in range(nx):
            m.C[r * nx + c] = Cm[r, c]
    m.supersample = 1
    m.Ts = float(Ts)
    m.dynamics_density, m.measurement_density, m.initial_density = df, dg, d0
    if df.dim != nx or d0.dim != nx or dg.dim != ny:
        raise ValueError("density dimensions do not match the model")
    return m


def make_rb_model(Fn, Bn, An, Al, Bl, Gn, Cl, R1n, R1l, R2, d0n, d0l, Ts=1.0):
    """Rao-Blackwellized model with constant matrices (reference src/rbpf.jl:92-98):
    xn' = Fn xn + Bn u + An xl + wn, xl' = Al xl + Bl u + wl, y = Gn xn + Cl xl + e.
    An / Cl may be None (no coupling).  R1n, R2, d0n, d0l are Gaussian structs (make_gaussian); R1l is a matrix."""
    Fn = np.atleast_2d(np.asarray(Fn, dtype=np.float64)); Al = np.atleast_2d(np.asarray(Al, dtype=np.float64))
    nn, nl = Fn.shape[0], Al.shape[0]
    nx = nn + nl
    Gn = np.asarray(Gn, dtype=np.float64).reshape(-1, nn)
    ny = Gn.shape[0]
    An = np.zeros((nn, nl)) if An is None else np.asarray(An, dtype=np.float64).reshape(nn, nl)
    Cl = np.zeros((ny, nl)) if Cl is None else np.asarray(Cl, dtype=np.float64).reshape(ny, nl)
    Bn = np.zeros((nn, 0)) if Bn is None else np.asarray(Bn, dtype=np.float64).reshape(nn, -1)
    nu = Bn.shape[1]
    Bl = np.zeros((nl, nu)) if Bl is None else np.asarray(Bl, dtype=np.float64).reshape(nl, nu)
    if nx > 4 or ny > 4 or nn < 1 or nl < 1:
        raise ValueError("RB model: 1 <= nxn, 1 <= nxl, nxn + nxl <= 4, ny <= 4")
    A = np.block([[Fn, An], [np.zeros((nl, nn)), Al]])
    B = np.vstack([Bn, Bl])
    Cm = np.hstack([Gn, Cl])
    m = Model()
    m.model_id = MODEL_RB_LINEAR
    m.nx, m.nu, m.ny, m.nxn = nx, nu, ny, nn
    for r in range(nx):
        for c in range(nx):
            m.A[r * nx + c] = A[r, c]
        for c in range(nu):
            m.B[r * nu + c] = B[r, c]
    for r in range(ny):
        for c in range(nx):
            m.C[r * nx + c] = Cm[r, c]
    m.supersample = 1
    m.Ts = float(Ts)
    m.dynamics_density, m.measurement_density, m.initial_density = R1n, R2, d0n
    m.linear_noise = make_gaussian(np.zeros(nl), np.atleast_2d(np.asarray(R1l, dtype=np.float64)), COV_FULL)
    m.linear_initial = d0l
    if R1n.dim != nn or d0n.dim != nn or R2.dim != ny or d0l.dim != nl:
        raise ValueError("density dimensions do not match the RB model")
    return m


def make_rb_bilinear_model(An, Al, Bl, Cl, R1n, R1l, R2, d0n, d0l, Fn=None, Bn=None, Gn=None, quadtank=None,
                           Ts=1.0, supersample=2):
    """Rao-Blackwellized model whose coupling depends on the nonlinear state (reference src/rbpf.jl:92-98 with `An` a
    function of x): xn' = f_n(xn, u) + An(xn) xl + wn, An(xn) = An[0] + sum_k xn[k] An[1+k]; xl' = Al xl + Bl u + wl;
    y = g(xn) + Cl xl + e.  f_n / g: linear (Fn, Bn, Gn) or the quad-tank (quadtank = dict of constants, nxn = 4).
    An: array [1 + nxn, nxn, nxl].  R1n, R2, d0n, d0l are Gaussian structs; R1l a matrix."""
    An = np.asarray(An, dtype=np.float64)
    nn, nl = An.shape[1], An.shape[2]
    if An.shape[0] != nn + 1 or nn > 4 or nl > 8:
        raise ValueError("An must be [1 + nxn, nxn, nxl] with nxn <= 4, nxl <= 8")
    Al = np.asarray(Al, dtype=np.float64).reshape(nl, nl)
    Cl = np.asarray(Cl, dtype=np.float64).reshape(-1, nl)
    ny = Cl.shape[0]
    m = Model()
    m.model_id = MODEL_RB_BILINEAR
    if quadtank is not None:
        if nn != 4 or ny != 2:
            raise ValueError("the quad-tank nonlinear part has 4 states and 2 outputs")
        nu = 2
        vals = dict(QUADTANK_DEFAULTS)
        vals.update(quadtank)
        for i, name in enumerate(QT_NAMES):
            m.qt[i] = float(vals[name])
        m.rb.fn_kind = 1
    else:
        Fn = np.asarray(Fn, dtype=np.float64).reshape(nn, nn)
        Gn = np.asarray(Gn, dtype=np.float64).reshape(ny, nn)
        Bn = np.zeros((nn, 0)) if Bn is None else np.asarray(Bn, dtype=np.float64).reshape(nn, -1)
        nu = Bn.shape[1]
        for r in range(nn):
            for c in range(nn):
                m.A[r * nn + c] = Fn[r, c]
            for c in range(nu):
                m.B[r * nu + c] = Bn[r, c]
        for r in range(ny):
            for c in range(nn):
                m.C[r * nn + c] = Gn[r, c]
        m.rb.fn_kind = 0
    Bl = np.zeros((nl, nu)) if Bl is None else np.asarray(Bl, dtype=np.float64).reshape(nl, nu)
    m.nx, m.nu, m.ny, m.nxn = nn, nu, ny, nn
    m.rb.nxl = nl
    for r in range(nl):
        for c in range(nl):
            m.rb.Al[r * nl + c] = Al[r, c]
        for c in range(nu):
            m.rb.Bl[r * nu + c] = Bl[r, c]
    for r in range(ny):
        for c in range(nl):
            m.rb.Cl[r * nl + c] = Cl[r, c]
    for k in range(nn + 1):
        for r in range(nn):
            for c in range(nl):
                m.rb.An[k][r * nl + c] = An[k, r, c]
    m.supersample = int(supersample)
    m.Ts = float(Ts)
    m.dynamics_density, m.measurement_density, m.initial_density = R1n, R2, d0n
    m.linear_noise = make_gaussian(np.zeros(nl), np.atleast_2d(np.asarray(R1l, dtype=np.float64)), COV_FULL)
    m.linear_initial = d0l
    if R1n.dim != nn or d0n.dim != nn or R2.dim != ny or d0l.dim != nl:
        raise ValueError("density dimensions do not match the RB model")
    return m


def make_quadtank_model(df, dg, d0, Ts=1.0, supersample=2, **consts):
    m = Model()
    m.model_id = MODEL_QUADTANK_RK4
    m.nx, m.nu, m.ny = 4, 2, 2
    vals = dict(QUADTANK_DEFAULTS)
    vals.update(consts)
    for i, name in enumerate(QT_NAMES):
        m.qt[i] = float(vals[name])
    m.supersample = int(supersample)
    m.Ts = float(Ts)
    m.dynamics_density, m.measurement_density, m.initial_density = df, dg, d0
    if df.dim != 4 or d0.dim != 4 or dg.dim != 2:
        raise ValueError("quad-tank needs 4-D dynamics/initial and 2-D measurement densities")
    return m


def make_config(model, n_particles, filter_kind=PARTICLE_FILTER, strategy=RESAMPLE_SYSTEMATIC,
                resample_threshold=0.1, seed=0, device=0):
    c = Config()
    c.struct_size = C.sizeof(Config)
    c.filter_kind = filter_kind
    c.n_particles = int(n_particles)
    c.resampling_strategy = strategy
    c.device = device
    c.resample_threshold = float(resample_threshold)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c.model = model
    return c
