"""oracle/independent.py — a SECOND, independent CPU restatement of the hot path, in plain numpy.

TEST INFRASTRUCTURE ONLY (imported by tests/ and tests/golden/make_independent.py; never by the product).

Why it exists.  oracle/llpf_oracle.c shares headers with the engine (csrc/shared/: the deterministic exp / log, the fixed-point
sums, Philox, and — for the Rao-Blackwellized filter with per-particle covariance — the whole Kalman recursion body
llpf_rbfull_body.h).  A wrong sign in that shared Riccati update would be bit-identical on both sides.  This file shares
NOTHING with the engine except the random numbers it is fed: plain 2-D numpy arrays, numpy.linalg.cholesky / solve, libm-grade
numpy exp / log, Python loops over particles.  It follows the reference line by line:

  particle filter      reset!              src/filtering.jl:4-14
                       correct!            src/filtering.jl:164-168 = measurement_equation! src/PFtypes.jl:107-120 + logsumexp!
                                           src/utils.jl:18-27 (sum_all_but :66-71)
                       predict!            src/filtering.jl:140-153 = shouldresample src/resample.jl:5-10, resample
                                           (systematic :17-36, stratified :38-61), propagate_particles! src/PFtypes.jl:122-139,
                                           reset_weights! src/utils.jl:73-79
                       forward_trajectory  src/filtering.jl:343-365;  loglik src/smoothing.jl:227-230
                       Gaussian logpdf     src/utils.jl:252-257;  rk4 src/utils.jl:220-237;  quad-tank examples/example_quadtank.jl:8-35
  RBPF, !singleR       reset! src/rbpf.jl:146-160, predict! :163-232, correct! :235-283 with
                       correct!(kf, ...) src/filtering.jl:100-128

The reference's own random streams (Xoshiro randn, global rand()) are unpinned by its tests; the normals and uniforms are
INPUTS here (callables `normals(step, n, nd)` / `uniforms(step, n)`), supplied by the tests from the Philox generator that the
Random123 known-answer vectors pin (tests/test_detmath.py).  Rounding differs from the C oracle's reference order only through
operation order inside numpy (BLAS dot products, pairwise sums): agreement is asserted at 1e-10 per step, observed ~1e-13.
"""
import math

import numpy as np

LOG2PI = math.log(2.0 * math.pi)


# ---- Gaussian pieces (src/utils.jl:241-270) ---------------------------------------------------------------------------------
class Gaussian:
    def __init__(self, mu, Sigma):
        self.mu = np.asarray(mu, dtype=np.float64).reshape(-1)
        k = self.mu.size
        S = np.asarray(Sigma, dtype=np.float64)
        if S.ndim == 0:
            S = float(S) * np.eye(k)
        elif S.ndim == 1:
            S = np.diag(S)
        self.Sigma = S.reshape(k, k)
        self.L = np.linalg.cholesky(self.Sigma)

    def logpdf(self, x):
        """extended_logpdf(d, x) = mvnormal_c0(d) - invquad(Sigma, x - mu) / 2, src/utils.jl:252-257; x: (..., k)"""
        d = np.atleast_2d(x) - self.mu
        k = self.mu.size
        logdet = 2.0 * np.sum(np.log(np.diag(self.L)))
        z = np.linalg.solve(self.Sigma, d.T).T
        return -(k * LOG2PI + logdet) / 2.0 - np.sum(d * z, axis=1) / 2.0

    def sample(self, xi):
        """rand(rng, d) = mu + cholesky(Sigma).L * randn, src/utils.jl:260; xi: (n, k) standard normals"""
        return self.mu + np.atleast_2d(xi) @ self.L.T


class Laplace:
    """independent Laplace components of scale b: a measurement density other than a Gaussian (the reference weights with
    logpdf of any Distribution, ext/LowLevelParticleFiltersDistributionsExt.jl:80, or with a measurement_likelihood callable,
    src/PFtypes.jl:226-239)"""

    def __init__(self, b):
        self.b = float(b)

    def logpdf(self, v):
        v = np.atleast_2d(v)
        return -np.sum(np.abs(v), axis=1) / self.b - v.shape[1] * math.log(2.0 * self.b)


class StudentT:
    """independent Student-t components, nu degrees of freedom, scale sigma"""

    def __init__(self, nu, sigma):
        self.nu, self.sigma = float(nu), float(sigma)
        self.c1 = math.lgamma((self.nu + 1.0) / 2.0) - math.lgamma(self.nu / 2.0) - 0.5 * math.log(self.nu * math.pi) - math.log(self.sigma)

    def logpdf(self, v):
        v = np.atleast_2d(v)
        return np.sum(self.c1 - (self.nu + 1.0) / 2.0 * np.log1p((v / self.sigma) ** 2 / self.nu), axis=1)


# ---- process noise / initial densities other than a Gaussian ---------------------------------------------------------------
# The reference's AdvancedParticleFilter hands the noise to the user — x[i] = dynamics(xprev[j[i]], u, p, t, true) adds its own
# (src/PFtypes.jl:242-259, test/runtests.jl:553-599) — and its ParticleFilter draws from ANY dynamics_density / initial_density
# (rand!(rng, d, noise) src/PFtypes.jl:135, rand(rng, initial_density) src/filtering.jl:8).  The draws are inputs here too: per
# particle, nx standard normals xi and nx uniforms uu in [0, 1).
class MultiplicativeGaussianNoise:
    """x' = f(x) + (s0 + s1 |x|) .* xi : Gaussian noise whose standard deviation grows with the state it leaves"""

    def __init__(self, s0, s1):
        self.s0, self.s1 = float(s0), float(s1)

    def propagate(self, x, fx, xi, uu):
        return fx + (self.s0 + self.s1 * np.abs(x)) * xi


class LaplaceNoise:
    """x' = f(x) + Laplace(0, b) per component, drawn through the inverse CDF of one uniform each"""

    def __init__(self, b):
        self.b = float(b)

    def propagate(self, x, fx, xi, uu):
        v = 2.0 * uu - 1.0
        t = np.maximum(1.0 - np.abs(v), 2.0 ** -53)
        return fx + np.sign(v) * self.b * (-np.log(t))


class UniformBox:
    """initial density: independent uniforms on [lo_d, hi_d]"""

    def __init__(self, lo, hi):
        self.lo, self.hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)

    def sample_u(self, xi, uu):
        return self.lo + (self.hi - self.lo) * uu


# ---- models -------------------------------------------------------------------------------------------------------------------
class LinearModel:
    """dynamics A x + B u, measurement C x (examples/example_lineargaussian.jl:28-29)"""

    def __init__(self, A, B, C):
        self.A, self.C = np.atleast_2d(A), np.atleast_2d(C)
        self.B = np.zeros((self.A.shape[0], 0)) if B is None else np.asarray(B, dtype=np.float64).reshape(self.A.shape[0], -1)

    def f(self, x, u, t):
        out = x @ self.A.T
        if self.B.shape[1]:
            out = out + self.B @ np.asarray(u, dtype=np.float64).reshape(-1)
        return out

    def g(self, x, u, t):
        return x @ self.C.T


class QuadTank:
    """examples/example_quadtank.jl:8-35 through rk4(f, Ts; supersample) of src/utils.jl:220-237"""

    def __init__(self, k1=1.6, k2=1.6, g=9.81, A=(4.9, 4.9, 4.9, 4.9), a=(0.03, 0.03, 0.03, 0.03), gamma=(0.2, 0.2),
                 tswitch=500.0, a1factor=2.0, eps=1e-3, Ts=1.0, supersample=2):
        self.k1, self.k2, self.grav, self.A, self.a, self.gamma = k1, k2, g, A, a, gamma
        self.tswitch, self.a1factor, self.eps, self.Ts, self.ss = tswitch, a1factor, eps, Ts, supersample

    def rhs(self, h, u, t):
        A1, A2, A3, A4 = self.A
        a1, a2, a3, a4 = self.a
        if t > self.tswitch:
            a1 = a1 * self.a1factor
        g1, g2 = self.gamma
        s = np.sqrt(np.maximum(2.0 * self.grav * h, 0.0) + self.eps)          # ssqrt(x) = sqrt(max(x, 0) + 1e-3)
        return np.stack([
            -a1 / A1 * s[:, 0] + a3 / A1 * s[:, 2] + g1 * self.k1 / A1 * u[0],
            -a2 / A2 * s[:, 1] + a4 / A2 * s[:, 3] + g2 * self.k2 / A2 * u[1],
            -a3 / A3 * s[:, 2] + (1.0 - g2) * self.k2 / A3 * u[1],
            -a4 / A4 * s[:, 3] + (1.0 - g1) * self.k1 / A4 * u[0]], axis=1)

    def f(self, x, u, t):
        Tss = self.Ts / self.ss
        for _ in range(self.ss):
            f1 = self.rhs(x, u, t)
            f2 = self.rhs(x + Tss / 2.0 * f1, u, t + Tss / 2.0)
            f3 = self.rhs(x + Tss / 2.0 * f2, u, t + Tss / 2.0)
            f4 = self.rhs(x + Tss * f3, u, t + Tss)
            x = x + Tss / 6.0 * (f1 + 2.0 * f2 + 2.0 * f3 + f4)
            t = t + Tss
        return x

    def g(self, x, u, t):
        return x[:, :2]


# ---- logsumexp! and the resamplers --------------------------------------------------------------------------------------------
def logsumexp_inplace(w):
    """ll = logsumexp!(w, we), src/utils.jl:18-27: offset = maximum, s = sum of all but the maximum's exp(0) = 1"""
    imax = int(np.argmax(w))
    m = w[imax]
    w = w - m
    we = np.exp(w)
    we_wo = we.copy()
    we_wo[imax] -= 1.0
    s = float(np.sum(we_wo))
    we = we * (1.0 / (s + 1.0))
    w = w - math.log1p(s)
    return w, we, math.log1p(s) + m


def resample_systematic(we, jprev, U, M=None):
    """src/resample.jl:17-36: bins = cumsum(we); r = rand() * bins[end] / N; s_i = r + (i-1)/M; j[i] = first b with s_i < bins[b],
    entries whose threshold is never met keep their previous value"""
    N = we.size
    M = N if M is None else M
    bins = np.cumsum(we)
    r = U * bins[-1] / N
    s = r + np.arange(M) * (1.0 / M)
    j = np.searchsorted(bins, s, side="right")
    keep = j >= N
    j = np.where(keep, jprev[:M], j)
    return j.astype(np.int64), bins


def resample_stratified(we, jprev, Us, M=None):
    """src/resample.jl:38-61: u_i = (i - 1 + rand()) / M * bins[end]"""
    N = we.size
    M = N if M is None else M
    bins = np.cumsum(we)
    s = (np.arange(M) + Us[:M]) / M * bins[-1]
    j = np.searchsorted(bins, s, side="right")
    keep = j >= N
    j = np.where(keep, jprev[:M], j)
    return j.astype(np.int64), bins


# ---- the particle filter ----------------------------------------------------------------------------------------------------
class ParticleFilter:
    def __init__(self, N, model, df, dg, d0, resample_threshold=0.1, stratified=False, Ts=1.0):
        self.N, self.model, self.df, self.dg, self.d0 = N, model, df, dg, d0
        self.thr, self.stratified, self.Ts = resample_threshold, stratified, Ts
        self.j = np.arange(N)

    def reset(self, xi, uu=None):
        self.x = self.d0.sample_u(xi, uu) if hasattr(self.d0, "sample_u") else self.d0.sample(xi)      # x_i ~ d0
        self.w = np.full(self.N, -math.log(self.N))
        self.we = np.full(self.N, 1.0 / self.N)
        self.t = 1

    def correct(self, u, y, t):
        if y is not None and not np.any(np.isnan(y)):            # any(ismissing, y) && return, src/PFtypes.jl:109
            self.w = self.w + self.dg.logpdf(np.asarray(y, dtype=np.float64) - self.model.g(self.x, u, t))
        self.w, self.we, ll = logsumexp_inplace(self.w)
        return ll

    def predict(self, u, t, xi, U, uu=None):
        N = self.N
        ess = 1.0 / float(np.sum(self.we * self.we))
        self.resampled = self.thr == 1.0 or ess < N * self.thr
        if self.resampled:
            if self.stratified:
                self.j, self.bins = resample_stratified(self.we, self.j, U)
            else:
                self.j, self.bins = resample_systematic(self.we, self.j, float(U[0]))
            xprev = self.x[self.j]
            self.w = np.full(N, math.log(1.0 / N))
            self.we = np.full(N, 1.0 / N)
        else:
            self.j = np.arange(N)
            xprev = self.x
        if hasattr(self.df, "propagate"):                        # the model adds its own noise (PFtypes.jl:254 / any dynamics_density, :135)
            self.x = self.df.propagate(xprev, self.model.f(xprev, u, t), xi, uu)
        else:
            self.x = self.model.f(xprev, u, t) + self.df.sample(xi)
        self.t += 1

    def run(self, U, Y, t_index0, normals, uniforms, step0=0, user_uniforms=None):
        """T iterations of correct!(u_k, y_k, t_k); predict!(u_k, t_k), t_k = (t_index0 + k) Ts: forward_trajectory
        (t_index0 = 0) / loglik (t_index0 = 1).  normals(step, n, nd), uniforms(step, n): the draws of predict! number `step`."""
        T = len(Y)
        ll_steps = np.zeros(T)
        nres = 0
        for k in range(T):
            t = (t_index0 + k) * self.Ts
            u = U[k] if U is not None and len(U) else None
            ll_steps[k] = self.correct(u, Y[k], t)
            uu = user_uniforms(step0 + k, self.N, self.x.shape[1]) if user_uniforms is not None else None
            self.predict(u, t, normals(step0 + k, self.N, self.x.shape[1]), uniforms(step0 + k, self.N), uu)
            nres += int(self.resampled)
        return ll_steps, nres


# ---- Rao-Blackwellized filter, every particle its own Kalman filter ----------------------------------------------------------------
class RBPF:
    """xn' = f_n(xn, u) + An(xn) xl + wn,  xl' = Al xl + Bl u + wl,  y = g(xn) + Cl xl + e;  An(xn) = An[0] + sum_k xn[k] An[1 + k]"""

    def __init__(self, N, fn_model, An, Al, Bl, Cl, R1n, R1l, R2, d0n, d0l, resample_threshold=0.1, stratified=False, Ts=1.0):
        self.N, self.fn, self.An = N, fn_model, np.asarray(An, dtype=np.float64)
        self.Al, self.Cl = np.atleast_2d(Al), np.atleast_2d(Cl)
        self.Bl = np.zeros((self.Al.shape[0], 0)) if Bl is None else np.asarray(Bl, dtype=np.float64).reshape(self.Al.shape[0], -1)
        self.R1n, self.R1l, self.R2, self.d0n, self.d0l = R1n, np.atleast_2d(R1l), R2, d0n, d0l
        self.thr, self.stratified, self.Ts = resample_threshold, stratified, Ts
        self.j = np.arange(N)

    def coupling(self, xn):
        return self.An[0] + np.tensordot(xn, self.An[1:], axes=(0, 0))

    def reset(self, xi):
        N, nl = self.N, self.Al.shape[0]
        self.xn = self.d0n.sample(xi)
        self.xl = np.tile(self.d0l.mu, (N, 1))                   # xl = copy(pf.kf.d0.mu), R = copy(pf.kf.d0.Sigma), :151-152
        self.R = np.tile(self.d0l.Sigma, (N, 1, 1))
        self.w = np.full(N, -math.log(N))
        self.we = np.full(N, 1.0 / N)
        self.t = 1

    def correct(self, u, y, t):
        if y is not None and not np.any(np.isnan(y)):
            y = np.asarray(y, dtype=np.float64)
            yn = self.fn.g(self.xn, u, t)
            C = self.Cl
            I = np.eye(self.Al.shape[0])
            for i in range(self.N):                              # correct!(kf, u, y - yn), src/filtering.jl:100-128
                R = self.R[i]
                e = (y - yn[i]) - C @ self.xl[i]
                S = C @ R @ C.T
                S = (S + S.T) / 2.0 + self.R2.Sigma               # symmetrize(Ct R Ct') .+ R2
                Sc = np.linalg.cholesky(S)
                K = np.linalg.solve(S, (R @ C.T).T).T            # (R Ct') / S
                self.xl[i] = self.xl[i] + K @ e
                Rn = (I - K @ C) @ R
                self.R[i] = (Rn + Rn.T) / 2.0                    # symmetrize((I - K Ct) R)
                logdet = 2.0 * np.sum(np.log(np.diag(Sc)))
                self.w[i] += -(e.size * LOG2PI + logdet) / 2.0 - float(e @ np.linalg.solve(S, e)) / 2.0
        self.w, self.we, ll = logsumexp_inplace(self.w)
        return ll

    def predict(self, u, t, xi, U, uu=None):
        N = self.N
        ess = 1.0 / float(np.sum(self.we * self.we))
        self.resampled = self.thr == 1.0 or ess < N * self.thr
        if self.resampled:
            if self.stratified:
                self.j, self.bins = resample_stratified(self.we, self.j, U)
            else:
                self.j, self.bins = resample_systematic(self.we, self.j, float(U[0]))
            self.w = np.full(N, math.log(1.0 / N))
            self.we = np.full(N, 1.0 / N)
        else:
            self.j = np.arange(N)
        xn0, xl0, R0 = self.xn[self.j], self.xl[self.j], self.R[self.j]
        assert np.any(self.coupling(self.xn[0]) != 0.0), "zeroAn branch (src/rbpf.jl:175) is not restated here"
        fi = self.fn.f(xn0, u, t)
        wn = self.R1n.sample(xi)                                 # rand(pf.rng, pf.R1n)
        uu = np.zeros(0) if (u is None or self.Bl.shape[1] == 0) else np.asarray(u, dtype=np.float64).reshape(-1)
        Al = self.Al
        xn1, xl1, R1 = np.empty_like(xn0), np.empty_like(xl0), np.empty_like(R0)
        for i in range(N):                                       # src/rbpf.jl:206-221
            An = self.coupling(xn0[i])
            R = R0[i]
            Nt = An @ R @ An.T + self.R1n.Sigma
            L = np.linalg.solve(Nt.T, (Al @ R @ An.T).T).T       # Al R An' / Nt
            R1[i] = Al @ R @ Al.T + self.R1l - L @ Nt @ L.T
            Axl = An @ xl0[i]
            z = Axl + wn[i]
            xn1[i] = fi[i] + z
            xl1[i] = Al @ xl0[i] + (self.Bl @ uu if uu.size else 0.0) + L @ (z - Axl)
        self.xn, self.xl, self.R = xn1, xl1, R1
        self.t += 1

    def run(self, U, Y, t_index0, normals, uniforms, step0=0, user_uniforms=None):
        T = len(Y)
        ll_steps = np.zeros(T)
        nres = 0
        for k in range(T):
            t = (t_index0 + k) * self.Ts
            u = U[k] if U is not None and len(U) else None
            ll_steps[k] = self.correct(u, Y[k], t)
            self.predict(u, t, normals(step0 + k, self.N, self.xn.shape[1]), uniforms(step0 + k, self.N))
            nres += int(self.resampled)
        return ll_steps, nres
