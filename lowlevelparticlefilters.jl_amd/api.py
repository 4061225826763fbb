"""Host-side mirror of the reference's particle-filter API for the hot path.

Names follow LowLevelParticleFilters.jl (exports: reference src/LowLevelParticleFilters.jl:3-16) with the
trailing `!` dropped (`reset!` -> `reset`, `predict!` -> `predict`, `correct!` -> `correct`,
`update!` -> `update`, `logsumexp!` -> `logsumexp`).  Every verb is a thin call into the C ABI
(include/llpf.h); nothing here computes particle data on the host, and there is no fallback when
the GPU library is missing.

A closure cannot run on the GPU, so the `dynamics` / `measurement` /
`measurement_likelihood` arguments are *model descriptors*:

    LinearDynamics(A, B), LinearMeasurement(C)        x+ = A x + B u,  y = C x
    QuadTankDynamics(...), QuadTankMeasurement()      reference examples/example_quadtank.jl:8-35
    GaussianLikelihood(measurement, dg)               logpdf(dg, y - g(x)) for AdvancedParticleFilter
    UserDynamics(src, ...) + UserMeasurement / ...    HIP source of a model of one's own (llpf_model_compile)

— or, since round 6, ordinary Python callables with the reference's signatures (`dynamics(x, u, p, t)`, `measurement(x, u, p, t)`,
`measurement_likelihood(x, u, y, p, t)`; pass `nu=` / `ny=`): straight-line functions are traced once on tracer numbers and emitted as
the device snippet (tracing.py), operation for operation.
"""
import ctypes as C

import numpy as np

from . import _capi
from . import _structs as S

__all__ = [
    "StateAffineCoupling",
    "MvNormal", "ResampleSystematic", "ResampleStratified",
    "LinearDynamics", "LinearMeasurement", "QuadTankDynamics", "QuadTankMeasurement", "GaussianLikelihood",
    "ResampleResidual", "weighted_cov", "weighted_quantile", "log_likelihood_fun", "metropolis", "metropolis_bank", "naive_sampler", "mode_trajectory", "KalmanFilter", "RBMeasurementModel", "RBPF", "smooth", "smoothed_mean", "smoothed_cov", "smoothed_trajs", "ParticleFilter", "AdvancedParticleFilter", "AuxiliaryParticleFilter", "FilterBank", "ParticleFilteringSolution",
    "reset", "predict", "correct", "update", "forward_trajectory", "mean_trajectory", "loglik",
    "particles", "weights", "expweights", "state", "num_particles", "index", "effective_particles",
    "shouldresample", "resample", "weighted_mean", "logsumexp", "simulate", "parameters",
    "dynamics", "measurement", "measurement_likelihood", "dynamics_density", "measurement_density",
    "initial_density", "resample_threshold", "resampling_strategy", "UserDynamics", "UserMeasurement", "UserLikelihood", "UserNoise", "UserInitial",
]


# ---------------------------------------------------------------------------------------------------
# densities and model descriptors
# ---------------------------------------------------------------------------------------------------
class MvNormal:
    """N(mu, Sigma).  `cov`: float -> sigma^2 * I (PDMats.ScalMat), 1-D array -> diagonal (PDiagMat),
    2-D array -> full (PDMat).  Mirrors Distributions.MvNormal / SimpleMvNormal as used by the
    reference (src/utils.jl:241-270, ext/LowLevelParticleFiltersDistributionsExt.jl:16,80)."""

    def __init__(self, mu, cov=None, kind=None):
        if cov is None:            # MvNormal(Sigma): zero mean
            cov = mu
            c = np.asarray(cov, dtype=np.float64)
            if c.ndim == 0:
                raise ValueError("MvNormal(cov) needs an array covariance to infer the dimension")
            mu = np.zeros(c.shape[0])
        self._g = S.make_gaussian(mu, cov, kind)

    def __len__(self):
        return self._g.dim

    @property
    def mean(self):
        return S.gaussian_mean(self._g)

    @property
    def cov(self):
        return S.gaussian_cov_matrix(self._g)

    def rand(self, rng):
        return self.mean + np.linalg.cholesky(self.cov) @ rng.standard_normal(len(self))

    def struct(self):
        return self._g


class ResampleSystematic:      # reference src/LowLevelParticleFilters.jl:44
    code = S.RESAMPLE_SYSTEMATIC


class ResampleStratified:      # reference src/LowLevelParticleFilters.jl:45
    code = S.RESAMPLE_STRATIFIED


class ResampleResidual:        # reference src/LowLevelParticleFilters.jl:46, src/resample.jl:63-117
    code = S.RESAMPLE_RESIDUAL


class LinearDynamics:
    def __init__(self, A, B=None):
        self.A = np.atleast_2d(np.asarray(A, dtype=np.float64))
        self.B = None if B is None else np.asarray(B, dtype=np.float64).reshape(self.A.shape[0], -1)

    def __call__(self, x, u, p=None, t=0.0):
        x = np.asarray(x, dtype=np.float64)
        out = self.A @ x
        if self.B is not None and self.B.shape[1]:
            out = out + self.B @ np.asarray(u, dtype=np.float64).reshape(-1)
        return out


class LinearMeasurement:
    def __init__(self, Cm):
        self.C = np.atleast_2d(np.asarray(Cm, dtype=np.float64))

    def __call__(self, x, u=None, p=None, t=0.0):
        return self.C @ np.asarray(x, dtype=np.float64)


class QuadTankDynamics:
    """Quad-tank process discretised with rk4 (reference examples/example_quadtank.jl:8-35,
    src/utils.jl:220-237)."""

    def __init__(self, supersample=2, **consts):
        self.supersample = int(supersample)
        self.consts = dict(S.QUADTANK_DEFAULTS)
        self.consts.update(consts)

    def rhs(self, h, u, t):
        c = self.consts
        a1 = c["a1"] * (c["a1_factor"] if t > c["t_switch"] else 1.0)
        ss = lambda z: np.sqrt(max(z, 0.0) + c["eps"])
        g2 = 2.0 * c["g"]
        return np.array([
            -a1 / c["A1"] * ss(g2 * h[0]) + c["a3"] / c["A1"] * ss(g2 * h[2]) + c["gamma1"] * c["k1"] / c["A1"] * u[0],
            -c["a2"] / c["A2"] * ss(g2 * h[1]) + c["a4"] / c["A2"] * ss(g2 * h[3]) + c["gamma2"] * c["k2"] / c["A2"] * u[1],
            -c["a3"] / c["A3"] * ss(g2 * h[2]) + (1 - c["gamma2"]) * c["k2"] / c["A3"] * u[1],
            -c["a4"] / c["A4"] * ss(g2 * h[3]) + (1 - c["gamma1"]) * c["k1"] / c["A4"] * u[0]])

    def __call__(self, x, u, p=None, t=0.0, Ts=1.0):
        x = np.asarray(x, dtype=np.float64).copy()
        h = Ts / self.supersample
        for _ in range(self.supersample):
            f1 = self.rhs(x, u, t)
            f2 = self.rhs(x + h / 2 * f1, u, t + h / 2)
            f3 = self.rhs(x + h / 2 * f2, u, t + h / 2)
            f4 = self.rhs(x + h * f3, u, t + h)
            x = x + h / 6 * (f1 + 2 * f2 + 2 * f3 + f4)
            t += h
        return x


class QuadTankMeasurement:
    def __call__(self, x, u=None, p=None, t=0.0):
        return np.asarray(x, dtype=np.float64)[:2].copy()


class GaussianLikelihood:
    """measurement_likelihood(x,u,y,p,t) = logpdf(dg, y - measurement(x,u,p,t))."""

    def __init__(self, measurement, measurement_density):
        self.measurement = measurement
        self.measurement_density = measurement_density


class UserDynamics:
    """A model the engine has no built-in for: HIP device source defining `struct UserModel` — prepare / dynamics / measurement and,
    optionally, `loglik` + `loglik_bound` (a measurement likelihood of its own) — compiled for the GPU with hiprtc when the filter
    is built (include/llpf.h: llpf_model_compile).  A, B, C, qt fill the parameter block the snippet reads (m->A, ..., m->qt[16]);
    `host` (optional) is the same dynamics as a Python callable, used only by host-side simulate."""

    def __init__(self, src, nx, nu, ny, A=None, B=None, C=None, qt=(), supersample=1, host=None):
        self.src, self.nx, self.nu, self.ny = str(src), int(nx), int(nu), int(ny)
        self.A, self.B, self.C, self.qt, self.supersample, self.host = A, B, C, list(qt), int(supersample), host

    def __call__(self, x, u=None, p=None, t=0.0, noise=False):
        if self.host is None:
            raise TypeError("no host version of this device model was given")
        return self.host(x, u, p, t)


class UserMeasurement:
    """the measurement of a UserDynamics snippet (`UserModel::measurement`); `host`: the same as a Python callable, for simulate"""

    def __init__(self, host=None):
        self.host = host

    def __call__(self, x, u=None, p=None, t=0.0, noise=False):
        if self.host is None:
            raise TypeError("no host version of this device model was given")
        return self.host(x, u, p, t)


class UserLikelihood:
    """measurement_likelihood(x, u, y, p, t) of an AdvancedParticleFilter (reference src/PFtypes.jl:226-239) as the `loglik(x, y, t)`
    member of the paired UserDynamics snippet; its `loglik_bound()` member declares the upper bound the normalisation works
    against (without one every step is normalised against the true maximum: one more launch per step, ≈ 20 % slower)."""

    def __init__(self, host=None):
        self.host = host

    def __call__(self, x, u, y, p=None, t=0.0):
        if self.host is None:
            raise TypeError("no host version of this device likelihood was given")
        return self.host(x, u, y, p, t)


class UserNoise:
    """dynamics_density of a filter whose model adds its own process noise: the `noise(x, fx, xi, uu, out)` member of the paired
    UserDynamics snippet — the reference's AdvancedParticleFilter contract, dynamics(x, u, p, t, noise = true) (src/PFtypes.jl:242-259),
    or a ParticleFilter with a dynamics_density that is not Gaussian (rand!(rng, d, noise), :122-139).  `gaussian`: the MvNormal the
    FFBS smoother and the auxiliary filter's add_noise! keep using (default: standard normal)."""

    def __init__(self, gaussian=None, host=None):
        self.gaussian, self.host = gaussian, host


class UserInitial:
    """initial_density of a filter whose model draws its own initial particles: the `initial(xi, uu, out)` member of the paired
    UserDynamics snippet (x_i = rand(rng, initial_density), reference src/filtering.jl:4-14)."""

    def __init__(self, host=None):
        self.host = host


_DESCRIPTORS = ()      # filled below: the descriptor classes a constructor accepts as they are


def _is_plain_callable(f):
    return callable(f) and not isinstance(f, _DESCRIPTORS)


def _trace_callables(dyn, meas, mlik, nx, nu, ny, p, likelihood_bound=None):
    """The reference's constructors take closures (src/PFtypes.jl:59-63, 189-193).  Plain Python callables with the reference's signatures
    — dynamics(x, u, p, t), measurement(x, u, p, t), measurement_likelihood(x, u, y, p, t) — are traced once (tracing.py) and emitted
    as the device snippet; returns the (UserDynamics, UserMeasurement, UserLikelihood | None) descriptors that stand for them."""
    from . import tracing
    if not _is_plain_callable(meas):
        raise TypeError("a callable dynamics needs a callable measurement (both are traced into one device model)")
    if nu < 0:
        raise ValueError("pass nu= (the number of inputs) with callable dynamics: it cannot be read off a closure")
    d = tracing.traced_dynamics(dyn, nx, nu, p=p, measurement=meas, ny=ny, measurement_likelihood=mlik if _is_plain_callable(mlik) else None,
                                loglik_bound=likelihood_bound)
    return d, d.measurement_model, d.likelihood_model


def _build_model(dyn, meas, df, dg, d0, Ts, user_likelihood=False):
    if isinstance(dyn, UserDynamics):
        if not isinstance(meas, UserMeasurement):
            raise TypeError("pair UserDynamics with UserMeasurement (the snippet's own measurement)")
        m = S.Model()
        m.model_id = _capi.model_compile(dyn.src, dyn.nx, dyn.ny)
        # what the snippet defines must be what the filter was told to use: a missing member would silently fall back to the Gaussian
        # descriptor, a present one silently override it
        traits = _capi.model_traits(m.model_id)
        for what, wanted, bit, member in (("measurement likelihood", user_likelihood, _capi.TRAIT_LOGLIK, "loglik"),
                                          ("dynamics_density", isinstance(df, UserNoise), _capi.TRAIT_NOISE, "noise"),
                                          ("initial_density", isinstance(d0, UserInitial), _capi.TRAIT_INITIAL, "initial")):
            if wanted and not traits & bit:
                raise TypeError("the %s is a User* descriptor but the snippet defines no `%s` member" % (what, member))
            if not wanted and traits & bit:
                raise TypeError("the snippet defines `%s`, which would override the Gaussian %s: pass the matching User* descriptor" % (member, what))
        if isinstance(df, UserNoise):
            df = df.gaussian if df.gaussian is not None else MvNormal(np.zeros(dyn.nx), 1.0)
        if isinstance(d0, UserInitial):
            d0 = MvNormal(np.zeros(dyn.nx), 1.0)
        m.nx, m.nu, m.ny = dyn.nx, dyn.nu, dyn.ny
        for name, mat in (("A", dyn.A), ("B", dyn.B), ("C", dyn.C)):
            if mat is not None:
                for i, v in enumerate(np.asarray(mat, dtype=np.float64).reshape(-1)):
                    getattr(m, name)[i] = v
        for i, v in enumerate(dyn.qt):
            m.qt[i] = float(v)
        m.supersample, m.Ts = dyn.supersample, float(Ts)
        m.dynamics_density, m.measurement_density, m.initial_density = df.struct(), dg.struct(), d0.struct()
        return m
    if isinstance(dyn, LinearDynamics) and isinstance(meas, LinearMeasurement):
        return S.make_lg_model(dyn.A, dyn.B, meas.C, df.struct(), dg.struct(), d0.struct(), Ts)
    if isinstance(dyn, QuadTankDynamics) and isinstance(meas, QuadTankMeasurement):
        return S.make_quadtank_model(df.struct(), dg.struct(), d0.struct(), Ts, dyn.supersample, **dyn.consts)
    raise TypeError("dynamics / measurement must be model descriptors (LinearDynamics + LinearMeasurement, QuadTankDynamics + "
                    "QuadTankMeasurement, UserDynamics + UserMeasurement) or a pair of plain callables (traced: tracing.py)")


# ---------------------------------------------------------------------------------------------------
# filters
# ---------------------------------------------------------------------------------------------------
class _AbstractParticleFilter:
    kind = S.PARTICLE_FILTER

    def _setup(self, N, dyn, meas, df, dg, d0, resample_threshold, resampling_strategy, rng, p, threads, Ts, nu, ny, device):
        self.dynamics = dyn
        self.measurement = meas
        self.dynamics_density = df
        self.measurement_density = dg
        self.initial_density = d0
        self.resample_threshold = float(resample_threshold)
        self.resampling_strategy = resampling_strategy
        self.rng = 0 if rng is None else int(rng)       # the Philox key (the reference stores an Xoshiro, src/PFtypes.jl:30)
        self.p = p
        self.threads = threads                          # accepted for signature parity; the GPU is always parallel
        self.Ts = float(Ts)
        if _is_plain_callable(dyn):        # closures as the reference takes them: traced into a device model (tracing.py)
            dyn, meas, _ = _trace_callables(dyn, meas, None, len(d0), nu, len(dg) if ny < 0 else ny, p)
            self.dynamics, self.measurement = dyn, meas
            nu = ny = -1
        if not isinstance(dyn, UserDynamics) and (isinstance(df, UserNoise) or isinstance(d0, UserInitial)):
            raise TypeError("UserNoise / UserInitial are members of a UserDynamics snippet")
        self._model = _build_model(dyn, meas, df, dg, d0, Ts, getattr(self, "_user_likelihood", False))
        self.nx, self.nu, self.ny = self._model.nx, self._model.nu, self._model.ny
        if nu not in (-1, self.nu) or ny not in (-1, self.ny):
            raise ValueError("nu / ny do not match the model descriptor")
        self._cfg = S.make_config(self._model, N, self.kind, resampling_strategy.code, resample_threshold, self.rng, device)
        self._h = _capi.FilterHandle(self._cfg)
        self.N = int(N)

    def set_parameters(self, *, dynamics=None, measurement=None, dynamics_density=None, measurement_density=None, initial_density=None):
        """New parameters for THIS filter (same model family and dimensions): what the reference's `filter_from_parameters(theta, pf)` is
        handed the old filter for (src/smoothing.jl:266-283) — nothing is reallocated on the device (llpf_set_model).  Returns self."""
        dyn = self.dynamics if dynamics is None else dynamics
        meas = self.measurement if measurement is None else measurement
        df = self.dynamics_density if dynamics_density is None else dynamics_density
        dg = self.measurement_density if measurement_density is None else measurement_density
        d0 = self.initial_density if initial_density is None else initial_density
        model = _build_model(dyn, meas, df, dg, d0, self.Ts, getattr(self, "_user_likelihood", False))
        self._h.set_model(model)
        self.dynamics, self.measurement, self.dynamics_density, self.measurement_density, self.initial_density = dyn, meas, df, dg, d0
        self._model = model
        return self

    # pf(u, y, p, t): one update! step (reference src/filtering.jl:238,240)
    def __call__(self, u, y, p=None, t=None):
        return update(self, u, y, p, t)

    @property
    def state(self):
        return _StateView(self)


class ParticleFilter(_AbstractParticleFilter):
    """ParticleFilter(N, dynamics, measurement, dynamics_density, measurement_density, initial_density; ...)
    — reference src/PFtypes.jl:21-36,65-75 (defaults: resample_threshold = 0.1, ResampleSystematic, Ts = 1)."""
    kind = S.PARTICLE_FILTER

    def __init__(self, N, dynamics, measurement, dynamics_density, measurement_density, initial_density, *,
                 resample_threshold=0.1, resampling_strategy=ResampleSystematic, rng=None, p=None,
                 threads=False, Ts=1.0, nu=-1, ny=-1, device=0):
        self._setup(N, dynamics, measurement, dynamics_density, measurement_density, initial_density,
                    resample_threshold, resampling_strategy, rng, p, threads, Ts, nu, ny, device)
        self.measurement_likelihood = None


class AdvancedParticleFilter(_AbstractParticleFilter):
    """AdvancedParticleFilter(N, dynamics, measurement, measurement_likelihood, dynamics_density,
    initial_density; ...) — reference src/PFtypes.jl:162-210 (default resample_threshold = 0.5)."""
    kind = S.ADVANCED_PARTICLE_FILTER

    def __init__(self, N, dynamics, measurement, measurement_likelihood, dynamics_density, initial_density, *,
                 resample_threshold=0.5, resampling_strategy=ResampleSystematic, rng=None, p=None,
                 threads=False, Ts=1.0, nu=-1, ny=-1, device=0, likelihood_bound=None):
        if _is_plain_callable(measurement_likelihood):      # measurement_likelihood(x, u, y, p, t) as a closure (src/PFtypes.jl:226-239)
            if not _is_plain_callable(dynamics) or ny < 0:
                raise TypeError("a callable measurement_likelihood needs callable dynamics / measurement and ny=")
            dynamics, measurement, measurement_likelihood = _trace_callables(dynamics, measurement, measurement_likelihood, len(initial_density), nu, ny, p,
                                                                             likelihood_bound)
            nu = -1
        self._user_likelihood = isinstance(measurement_likelihood, UserLikelihood)
        if isinstance(measurement_likelihood, UserLikelihood):
            if not isinstance(dynamics, UserDynamics):
                raise TypeError("a UserLikelihood is the loglik member of a UserDynamics snippet")
            dgs = MvNormal(np.zeros(dynamics.ny), 1.0)       # the descriptor's Gaussian is not used by such a model (its bound replaces the peak)
        elif isinstance(measurement_likelihood, GaussianLikelihood):
            dgs = measurement_likelihood.measurement_density
        else:
            raise TypeError("measurement_likelihood must be a GaussianLikelihood or a UserLikelihood descriptor")
        self._setup(N, dynamics, measurement, dynamics_density, dgs,
                    initial_density, resample_threshold, resampling_strategy, rng, p, threads, Ts, nu, ny, device)
        self.measurement_likelihood = measurement_likelihood


_DESCRIPTORS = (LinearDynamics, LinearMeasurement, QuadTankDynamics, QuadTankMeasurement, GaussianLikelihood, UserDynamics, UserMeasurement,
                UserLikelihood)


class KalmanFilter:
    """KalmanFilter(A, B, C, D, R1, R2, d0) — descriptor of the inner filter of an RBPF (reference src/kalman.jl; only
    constant matrices and D = 0 are supported)."""

    def __init__(self, A, B, C, D, R1, R2, d0):
        if np.any(np.asarray(D) != 0):
            raise NotImplementedError("D != 0")
        self.A, self.B, self.C, self.D = np.atleast_2d(np.asarray(A, float)), B, C, D
        self.R1, self.R2, self.d0 = np.atleast_2d(np.asarray(R1, float)), np.atleast_2d(np.asarray(R2, float)), d0


class RBMeasurementModel:
    """RBMeasurementModel(measurement, R2, ny) — reference src/rbpf.jl:36-60; `measurement` is a LinearMeasurement
    descriptor of the nonlinear state's contribution y = Gn xn (+ C xl + e)."""

    def __init__(self, measurement, R2, ny):
        self.measurement, self.R2, self.ny = measurement, R2, int(ny)


class StateAffineCoupling:
    """An as a function of the state (reference src/rbpf.jl:108: "`An` may be a matrix or a function of x, u, p, t that
    returns a matrix"), in the descriptor form the GPU can run: An(xn) = An0 + sum_k xn[k] Ank[k]."""

    def __init__(self, An0, Ank):
        self.An0 = np.atleast_2d(np.asarray(An0, dtype=np.float64))
        self.Ank = np.asarray(Ank, dtype=np.float64).reshape((self.An0.shape[0],) + self.An0.shape)

    def __call__(self, x, u=None, p=None, t=0.0):
        xn = np.asarray(getattr(x, "xn", x), dtype=np.float64)[: self.An0.shape[0]]
        return self.An0 + np.tensordot(xn, self.Ank, axes=(0, 0))


class RBPF(_AbstractParticleFilter):
    """RBPF(N, kf, dynamics, nl_measurement_model, R1n, d0n; An, nu, Ts, rng, resample_threshold) — the
    Rao-Blackwellized ("marginalized") particle filter of reference src/rbpf.jl:63-144.  With constant matrices
    (dynamics = LinearDynamics(Fn, Bn) of the nonlinear substate, An a matrix or None) one covariance serves all
    particles; with An = StateAffineCoupling(...) every particle carries its own Kalman filter (the reference's
    !singleR branches, :176/:247), and dynamics / nl_measurement_model.measurement may also be the quad-tank
    descriptors."""
    kind = S.PARTICLE_FILTER

    def __init__(self, N, kf, dynamics, nl_measurement_model, R1n, d0n, *, An=None, nu=-1, Ts=1.0, p=None, rng=None,
                 resample_threshold=0.1, names=None, device=0):
        as_g = lambda d: d.struct() if isinstance(d, MvNormal) else d
        as_cov = lambda d, n: d if not isinstance(d, (np.ndarray, list)) else MvNormal(np.zeros(n), np.atleast_2d(np.asarray(d, float)))
        nn = 4 if isinstance(dynamics, QuadTankDynamics) else np.atleast_2d(np.asarray(dynamics.A, float)).shape[0]
        R1n = as_cov(R1n, nn)
        R2 = as_cov(nl_measurement_model.R2, nl_measurement_model.ny)
        mm = nl_measurement_model.measurement
        Gn = np.zeros((nl_measurement_model.ny, nn)) if mm is None else getattr(mm, "C", None)
        self.kf, self.dynamics, self.nl_measurement_model, self.An, self.R1n, self.d0n = kf, dynamics, nl_measurement_model, An, R1n, d0n
        self.resample_threshold = float(resample_threshold)
        self.resampling_strategy = ResampleSystematic            # resampling_strategy(pf::RBPF), src/rbpf.jl:306
        self.rng = 0 if rng is None else int(rng)
        self.p, self.Ts, self.names, self.threads = p, float(Ts), names, False
        self.measurement_likelihood = None
        self.per_particle_covariance = isinstance(An, StateAffineCoupling)
        if self.per_particle_covariance:
            Ast = np.concatenate([An.An0[None], An.Ank], axis=0)
            if isinstance(dynamics, QuadTankDynamics):
                self._model = S.make_rb_bilinear_model(Ast, kf.A, kf.B, kf.C, as_g(R1n), kf.R1, as_g(R2), as_g(d0n), as_g(kf.d0),
                                                       quadtank=dynamics.consts, Ts=Ts, supersample=dynamics.supersample)
            else:
                self._model = S.make_rb_bilinear_model(Ast, kf.A, kf.B, kf.C, as_g(R1n), kf.R1, as_g(R2), as_g(d0n), as_g(kf.d0),
                                                       Fn=dynamics.A, Bn=dynamics.B, Gn=Gn, Ts=Ts)
        else:
            self._model = S.make_rb_model(dynamics.A, dynamics.B, An, kf.A, kf.B, Gn, kf.C, as_g(R1n), kf.R1, as_g(R2), as_g(d0n), as_g(kf.d0), Ts)
        self.nx, self.nu, self.ny = self._model.nx, self._model.nu, self._model.ny
        if self.per_particle_covariance:
            self.nx = self._model.nx + self._model.rb.nxl      # particles are [xn; xl]
        self._cfg = S.make_config(self._model, N, self.kind, S.RESAMPLE_SYSTEMATIC, resample_threshold, self.rng, device)
        self._h = _capi.FilterHandle(self._cfg)
        self.N = int(N)

    @property
    def covariance(self):
        """x[1].R: the covariance of the linear substate (shared by all particles for constant matrices)."""
        if self.per_particle_covariance:
            return self._h.rb_linear_state()[1][0]
        return self._h.rb_covariance()

    def linear_state(self):
        """(xl [N, nxl], R [N, nxl, nxl]): the fields xl, R of every RBParticle (reference src/rbpf.jl:1-5)."""
        if not self.per_particle_covariance:
            nn = self._model.nxn
            xl = self._h.particles()[:, nn:]
            return xl, np.broadcast_to(self._h.rb_covariance(), (self.N,) + self._h.rb_covariance().shape).copy()
        return self._h.rb_linear_state()


class AuxiliaryParticleFilter:
    """AuxiliaryParticleFilter(args...; kwargs...) = AuxiliaryParticleFilter(ParticleFilter(args...; kwargs...)) or
    AuxiliaryParticleFilter(pf) — reference src/PFtypes.jl:38-49.  Everything but predict!/correct!/update!/
    forward_trajectory/loglik is forwarded to the wrapped filter (PFtypes.jl:101-105, 299)."""

    def __init__(self, *args, **kwargs):
        if len(args) == 1 and not kwargs and isinstance(args[0], _AbstractParticleFilter):
            pf = args[0]
        else:
            pf = ParticleFilter(*args, **kwargs)
        object.__setattr__(self, "pf", pf)

    def __getattr__(self, name):            # getproperty forwarding, src/PFtypes.jl:101-105
        return getattr(object.__getattribute__(self, "pf"), name)

    # pf(u, y, y1, p, t): update! (reference src/filtering.jl:239)
    def __call__(self, u, y, y1, p=None, t=None):
        return update(self, u, y, y1, p, t)


class _StateView:
    """Read-only view with the reference's PFstate field names (src/PFtypes.jl:8-17)."""

    def __init__(self, pf):
        self._pf = pf

    x = property(lambda s: s._pf._h.particles())
    xprev = property(lambda s: s._pf._h.particles())    # xprev == x outside predict! (copyto!, filtering.jl:151)
    w = property(lambda s: s._pf._h.weights())
    we = property(lambda s: s._pf._h.expweights())
    maxw = property(lambda s: s._pf._h.maxw())
    j = property(lambda s: s._pf._h.ancestors())
    bins = property(lambda s: s._pf._h.bins())
    t = property(lambda s: s._pf._h.index())


class ParticleFilteringSolution:
    """Fields f,u,y,x,w,we,ll,t of the reference's struct (src/solutions.jl:334-345).  x is [T, N, nx]
    (column t of the reference's N x T matrix of SVectors is x[t]); w, we are [T, N]."""

    def __init__(self, f, u, y, x, w, we, ll, quantile_p=None, xquant=None):
        self.f, self.u, self.y, self.x, self.w, self.we, self.ll = f, u, y, x, w, we, ll
        self.t = np.arange(x.shape[0]) * f.Ts
        # forward_trajectory(..., quantiles=q): weighted_quantile of every timestep, computed on the device inside the run loop
        self.quantile_p = None if quantile_p is None else np.atleast_1d(np.asarray(quantile_p, dtype=np.float64))
        self.xquant = xquant                      # [T, nx, len(quantile_p)]


# ---------------------------------------------------------------------------------------------------
# verbs
# ---------------------------------------------------------------------------------------------------
def reset(pf):
    """reset!(pf) — reference src/filtering.jl:4-14."""
    pf._h.reset()


def _t(pf, t):
    return pf._h.index() * pf.Ts if t is None else float(t)


def _pt(args, kw, skip):
    """(p, t) from the trailing positional / keyword arguments of a verb."""
    rest = list(args[skip:])
    p = rest[0] if len(rest) > 0 else kw.get("p")
    t = rest[1] if len(rest) > 1 else kw.get("t")
    return p, t


def predict(pf, u, *args, **kw):
    """predict!(pf, u, p, t = index(pf)*Ts) — reference src/filtering.jl:140-153;
    predict!(pf::AuxiliaryParticleFilter, u, y1, p, t) — :195-217 (y1 = the NEXT measurement)."""
    if isinstance(pf, AuxiliaryParticleFilter):
        y1 = args[0] if args else kw.get("y1")
        pf._h.aux_predict(u, y1, _t(pf, _pt(args, kw, 1)[1]))
        return
    pf._h.predict(u, _t(pf, _pt(args, kw, 0)[1]))


def correct(pf, u, y, p=None, t=None):
    """ll, 0 = correct!(pf, u, y, p, t) — reference src/filtering.jl:164-168.  y=None means missing.
    For an AuxiliaryParticleFilter (:170-174) only logsumexp! runs: the measurement update was done in predict!."""
    if isinstance(pf, AuxiliaryParticleFilter):
        return pf._h.aux_correct(), 0
    return pf._h.correct(u, y, _t(pf, t)), 0


def update(pf, u, y, *args, **kw):
    """ll, 0 = update!(pf, u, y, p, t) — reference src/filtering.jl:181-185;
    update!(pf::AuxiliaryParticleFilter, u, y, y1, p, t) — :187-191."""
    if isinstance(pf, AuxiliaryParticleFilter):
        y1 = args[0] if args else kw.get("y1")
        return pf._h.aux_update(u, y1, _t(pf, _pt(args, kw, 1)[1])), 0
    return pf._h.update(u, y, _t(pf, _pt(args, kw, 0)[1])), 0


def forward_trajectory(pf, u, y, p=None, quantiles=None):
    """sol = forward_trajectory(pf, u, y, p) — reference src/filtering.jl:343-365 (:367-384 for the auxiliary
    filter).  The whole T-step loop is enqueued on the device; callbacks of the reference signature are not
    supported (a fused on-device loop cannot call back into the host) — drive update() step by step if needed.
    quantiles=q (not for the auxiliary filter): weighted_quantile(sol, q) of every timestep is computed on the device inside the run loop
    (llpf_run's xquant output) and kept in the solution; weighted_quantile(sol, q') with q' among q is then served from it."""
    reset(pf)
    if isinstance(pf, AuxiliaryParticleFilter):
        if quantiles is not None:
            raise ValueError("forward_trajectory(quantiles=...) is not provided for the auxiliary filter: use weighted_quantile(sol, q)")
        r = pf._h.run_aux(u, y, mode=0, history=True)
    else:
        r = pf._h.run(u, y, t_index0=0.0, history=True, quantiles=quantiles)
    return ParticleFilteringSolution(pf, u, y, r["x"], r["w"], r["we"], r["ll"], quantiles, r.get("xquant"))


def loglik(pf, u, y, p=None):
    """loglik(pf, u, y, p) — reference src/smoothing.jl:227-230 (reset!, then sum of update! with
    t = index(pf)*Ts, i.e. the first step is at t = 1*Ts); :232-236 for the auxiliary filter (t = (k-1)*Ts, the
    last step is an update! of the wrapped filter)."""
    reset(pf)
    if isinstance(pf, AuxiliaryParticleFilter):
        return pf._h.run_aux(u, y, mode=1)["ll"]
    return pf._h.run(u, y, t_index0=1.0)["ll"]


def log_likelihood_fun(filter_from_parameters, priors, u, y):
    """ll = log_likelihood_fun(filter_from_parameters, priors, u, y) — reference src/smoothing.jl:266-283: theta -> log prior + loglik of
    the filter built from theta; -inf outside the priors' support or when the filter degenerates.  `filter_from_parameters(theta, pf)` is
    called with the previous filter (None the first time) so that it can return `pf.set_parameters(...)` instead of a new filter;
    a function of theta alone is accepted too.  `priors[i]` is anything with a `logpdf(x)` method (scipy.stats frozen distributions)."""
    import inspect
    try:
        two = len(inspect.signature(filter_from_parameters).parameters) >= 2
    except (TypeError, ValueError):
        two = False
    state = {"pf": None}

    def ll(theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.size != len(priors):
            raise ValueError("Input must have same length as priors")
        lp = float(sum(np.float64(priors[i].logpdf(theta[i])) for i in range(theta.size)))
        if not np.isfinite(lp):
            return -np.inf
        state["pf"] = filter_from_parameters(theta, state["pf"]) if two else filter_from_parameters(theta)
        try:
            return lp + loglik(state["pf"], u, y)
        except (_capi.LLPFError, ValueError, FloatingPointError):
            return -np.inf
    return ll


def naive_sampler(theta0, rng=None):
    """theta -> theta + N(0, diag(0.1 |theta0|)) — reference src/smoothing.jl:284-287"""
    theta0 = np.asarray(theta0, dtype=np.float64)
    if np.any(theta0 == 0):
        raise ValueError("Naive sampler does not work if initial parameter vector contains zeros")
    rng = np.random.default_rng() if rng is None else rng
    sd = np.sqrt(0.1 * np.abs(theta0))
    return lambda theta: np.asarray(theta) + sd * rng.standard_normal(theta0.size)


def metropolis(ll, R, theta0, draw=None, rng=None):
    """params, lls = metropolis(ll, R, theta0, draw) — reference src/smoothing.jl:311-330: marginal Metropolis with a symmetric proposal;
    `rng` (numpy Generator) supplies the acceptance uniforms (the reference uses the global rand())."""
    rng = np.random.default_rng() if rng is None else rng
    draw = naive_sampler(theta0, rng) if draw is None else draw
    params, lls = [np.asarray(theta0, dtype=np.float64)], [float(ll(theta0))]
    for _ in range(1, int(R)):
        theta = np.asarray(draw(params[-1]), dtype=np.float64)
        lli = float(ll(theta))
        if rng.random() < np.exp(lli - lls[-1]):
            params.append(theta); lls.append(lli)
        else:
            params.append(params[-1]); lls.append(lls[-1])
    return params, np.array(lls)


def _spec_builds(spec_from_parameters, theta):
    """does this candidate give a model the library accepts? (densities positive definite, dimensions as declared)"""
    try:
        dy, me, df, dg, d0 = spec_from_parameters(theta)
        _build_model(dy, me, df, dg, d0, 1.0)
        for d in (df, dg, d0):                       # what the library's gauss_prepare (csrc/host/densities.hpp:23) asks of a covariance
            c = np.asarray(d.cov, dtype=np.float64)
            if not np.all(np.isfinite(c)) or not np.all(np.diag(c) > 0.0):
                return False
            np.linalg.cholesky(c)
        return True
    except Exception:
        return False


def metropolis_bank(bank, spec_from_parameters, priors, u, y, R, theta0s, draw=None, burnin=0, rng=None):
    """The GPU form of the reference's metropolis_threaded (src/smoothing.jl:335-347: one independent chain per thread): the chains advance in
    lockstep and every iteration is ONE bank run — llpf_bank_set_models with the chains' candidates, then the bank's loglik.
    `bank`: a FilterBank with one filter per chain; `spec_from_parameters(theta)` -> (dynamics, measurement, df, dg, d0) of a candidate;
    `theta0s`: [n_chains, n_parameters].  Returns an array [(R - burnin) * n_chains, n_parameters + 1], log-likelihoods (with the log
    prior) in the last column, chain after chain — the layout metropolis_threaded returns."""
    rng = np.random.default_rng() if rng is None else rng
    theta0s = np.atleast_2d(np.asarray(theta0s, dtype=np.float64))
    n = theta0s.shape[0]
    if n != bank.n_filters:
        raise ValueError("one chain per filter of the bank")
    draw = (lambda th, d=naive_sampler(theta0s[0], rng): d(th)) if draw is None else draw

    import re

    def lls_of(thetas):
        lp = np.array([sum(np.float64(priors[i].logpdf(th[i])) for i in range(th.size)) for th in thetas])
        ok = np.isfinite(lp)
        # Chains whose candidate lies outside the priors' support keep a valid model in their slot; its likelihood is not used.  A chain
        # whose candidate cannot be built (covariance not positive definite) or degenerates the filter scores -inf ALONE — the reference
        # wraps each chain's loglik in try / catch (src/smoothing.jl:276-280), so one bad proposal must not stop the others: the failing
        # slot (the library's message names the filter) is given its chain's current parameters and the bank runs again.
        # Candidates that cannot be BUILT are found on the host first (cheap: no device work), all of them at once — what can still cost a
        # rerun is a filter that degenerates on the device or a density that only the library refuses (a variance that underflowed to
        # zero passes the host's shape checks); the library names the first such filter.  An error that names no filter (bad shapes of
        # u / y, a lost device) is nobody's candidate and propagates at once.
        for k in range(n):
            if ok[k] and not _spec_builds(spec_from_parameters, thetas[k]):
                ok[k] = False
        ll = None
        for _ in range(n + 1):
            try:
                bank.set_parameters([spec_from_parameters(thetas[k] if ok[k] else cur[k]) for k in range(n)])
                ll = bank.loglik(u, y)
                break
            except _capi.LLPFError as e:                 # DegenerateWeights is one
                m = re.search(r"in filter (\d+)", str(e))
                bad = int(m.group(1)) if m else None
                if bad is None or not ok[bad]:
                    raise
                ok[bad] = False
        if ll is None:
            raise RuntimeError("metropolis_bank: the bank fails with every chain on its current parameters")
        out = np.where(ok, lp + ll, -np.inf)
        return np.where(np.isnan(out), -np.inf, out)

    cur = theta0s.copy()
    ll_cur = lls_of(cur)
    chain = np.empty((int(R), n, theta0s.shape[1] + 1))
    chain[0, :, :-1], chain[0, :, -1] = cur, ll_cur
    for i in range(1, int(R)):
        cand = np.stack([np.asarray(draw(cur[k]), dtype=np.float64) for k in range(n)])
        ll_c = lls_of(cand)
        acc = rng.random(n) < np.exp(ll_c - ll_cur)
        cur = np.where(acc[:, None], cand, cur)
        ll_cur = np.where(acc, ll_c, ll_cur)
        chain[i, :, :-1], chain[i, :, -1] = cur, ll_cur
    return np.concatenate([chain[int(burnin):, k, :] for k in range(n)], axis=0)


def weighted_cov(x, we=None):
    """weighted_cov(x, we) / weighted_cov(sol) — reference src/filtering.jl:573-581: per time step the covariance of
    the particles under probability weights with StatsBase's correction n / ((n - 1) sum(w)), n = count(w != 0).
    weighted_cov(pf): the current particles and weights of a filter, computed on the device (llpf_weighted_cov)."""
    if isinstance(x, _AbstractParticleFilter):
        return x._h.weighted_cov()
    if isinstance(x, ParticleFilteringSolution):
        x, we = x.x, x.we
    x, we = np.asarray(x), np.asarray(we)
    out = []
    for t in range(x.shape[0]):
        w = we[t]
        s = w.sum()
        mu = (x[t] * w[:, None]).sum(axis=0) / s
        d = x[t] - mu
        n = np.count_nonzero(w)
        out.append((d * w[:, None]).T @ d * (n / ((n - 1) * s)))
    return out


def _statsbase_quantile(v, w, q):
    """StatsBase.quantile(v, ProbabilityWeights(w), q) (src/weights.jl, the branch for non-frequency weights), vectorised over q"""
    q = np.atleast_1d(np.asarray(q, dtype=np.float64))
    if np.isnan(v).any():
        return np.full(q.shape, np.nan)
    keep = w != 0
    order = np.lexsort((w[keep], v[keep]))           # tuples (v, w) sort lexicographically
    vs, ws = v[keep][order], w[keep][order]
    S = np.cumsum(ws)
    h = q * (w.sum() - ws[0]) + ws[0]
    k = np.searchsorted(S, h, side="right")          # first k with S_k > h
    past = k >= vs.size
    k = np.minimum(k, vs.size - 1)
    Skold = np.where(k > 0, S[np.maximum(k - 1, 0)], 0.0)
    vkold = np.where(k > 0, vs[np.maximum(k - 1, 0)], 0.0)
    return np.where(past, vs[-1], vkold + (h - Skold) / (S[k] - Skold) * (vs[k] - vkold))


def weighted_quantile(x, we=None, q=None):
    """weighted_quantile(x, we, q) / weighted_quantile(sol, q) — reference src/filtering.jl:583-595: per time step and state dimension the
    weighted quantile of the particles (StatsBase's definition), a list of length T of [nx, len(q)] arrays (of nx-vectors for a scalar q):
    the reference's nesting [t][state][q].
    weighted_quantile(pf, q): the current particles and weights of a filter, sorted and summed on the device (llpf_weighted_quantile);
    [nx, len(q)] like one time step of the above (the C ABI itself writes [nq][nx]; the Julia accessor returns nx x nq as well)."""
    if isinstance(x, _AbstractParticleFilter):
        qq = we if q is None else q
        out = x._h.weighted_quantile(qq)
        return out[0] if np.isscalar(qq) else np.ascontiguousarray(out.T)
    if isinstance(x, ParticleFilteringSolution):
        sol, q = x, (we if q is None else q)
        if sol.xquant is not None:       # computed by the run loop on the device: served from there when every q was asked for
            qq = np.atleast_1d(np.asarray(q, dtype=np.float64))
            idx = [np.flatnonzero(sol.quantile_p == v) for v in qq]
            if all(len(i) for i in idx):
                sel = sol.xquant[:, :, [int(i[0]) for i in idx]]
                return [sel[t][:, 0].copy() if np.isscalar(q) else sel[t].copy() for t in range(sel.shape[0])]
        x, we = sol.x, sol.we
    x, we = np.asarray(x), np.asarray(we)
    out = []
    for t in range(x.shape[0]):
        r = np.stack([_statsbase_quantile(x[t][:, d], we[t], q) for d in range(x.shape[2])], axis=0)      # [state][q]
        out.append(r[:, 0] if np.isscalar(q) else r)
    return out


def mode_trajectory(x, we=None):
    """mode_trajectory(sol) / (x, we) — reference src/filtering.jl:415,436: the particle with the largest weight, T x nx."""
    if isinstance(x, ParticleFilteringSolution):
        x, we = x.x, x.we
    x, we = np.asarray(x), np.asarray(we)
    return x[np.arange(x.shape[0]), np.argmax(we, axis=1)]


def smooth(pf, *args):
    """xb, ll = smooth(pf, M, u, y, p) / smooth(pf, xf, wf, wef, ll, M, u, y, p) — forward filtering, backward
    simulation (reference src/smoothing.jl:103-143).  xb is [T, M, nx]."""
    if len(args) >= 7:
        xf, wf, wef, ll, M, u, y = args[:7]
    else:
        M, u, y = args[:3]
        sol = forward_trajectory(pf, u, y)
        xf, wf, wef, ll = sol.x, sol.w, sol.we, sol.ll
    xb, _ = pf._h.smooth(M, u, xf, wf, wef)
    return xb, ll


def smoothed_mean(xb):
    """smoothed_mean(xb) — reference src/smoothing.jl:356-361; returns [nx, T] like the reference's hcat."""
    return np.asarray(xb).mean(axis=1).T


def smoothed_cov(xb):
    """smoothed_cov(xb) — reference src/smoothing.jl:368-372: list of T covariance matrices (normalised by M-1)."""
    xb = np.asarray(xb)
    return [np.atleast_2d(np.cov(xb[t].T)) for t in range(xb.shape[0])]


def smoothed_trajs(xb):
    """smoothed_trajs(xb) — reference src/smoothing.jl:379-383: array (nx, M, T)."""
    return np.ascontiguousarray(np.transpose(np.asarray(xb), (2, 1, 0)))


def mean_trajectory(pf, u=None, y=None, p=None):
    """x̂, ll = mean_trajectory(pf, u, y) — reference src/filtering.jl:393, 417-432 (including its quirk:
    correct!(u[1], y[1], t=0) is followed by pf(u[t-1], y[t]) for t = 2..T).
    mean_trajectory(sol) — reference :405: T x nx matrix of weighted means."""
    if isinstance(pf, ParticleFilteringSolution):
        return np.einsum("tnd,tn->td", pf.x, pf.we)
    reset(pf)
    Y = np.asarray(y, dtype=np.float64).reshape(-1, pf.ny)
    Um = np.asarray(u, dtype=np.float64).reshape(Y.shape[0], -1)
    T = Y.shape[0]
    xh = np.zeros((T, pf.nx))
    ll = pf._h.correct(Um[0], Y[0], 0.0)
    xh[0] = pf._h.weighted_mean()
    for t in range(1, T):
        ll += pf._h.update(Um[t - 1], Y[t], t * pf.Ts)
        xh[t] = pf._h.weighted_mean()
    return xh, ll


# accessors — reference src/PFtypes.jl:296-334
def particles(pf):
    return pf._h.particles()


def weights(pf):
    return pf._h.weights()


def expweights(pf):
    return pf._h.expweights()


def state(pf):
    return pf.state


def num_particles(pf):
    return pf.N


def index(pf):
    return pf._h.index()


def parameters(pf):
    return pf.p


def dynamics(pf):
    return pf.dynamics


def measurement(pf):
    return pf.measurement


def measurement_likelihood(pf):
    return pf.measurement_likelihood


def dynamics_density(pf):
    return pf.dynamics_density


def measurement_density(pf):
    return pf.measurement_density


def initial_density(pf):
    return pf.initial_density


def resample_threshold(pf):
    return pf.resample_threshold


def resampling_strategy(pf):
    return pf.resampling_strategy


def effective_particles(pf_or_we):
    """effective_particles(pf | we) — reference src/resample.jl:1-2."""
    if isinstance(pf_or_we, _AbstractParticleFilter):
        return pf_or_we._h.ess()
    we = np.asarray(pf_or_we, dtype=np.float64)
    # ESS of a plain vector goes through the same normalise kernel: we are exp-weights => log them
    h = _weights_handle(we)
    return h.ess()


def _weights_handle(we):
    """A scratch 1-D filter holding log(we) as its weights (weights-only operations on plain vectors)."""
    g = MvNormal(np.zeros(1), 1.0)
    m = S.make_lg_model(np.eye(1), None, np.eye(1), g.struct(), g.struct(), g.struct())
    cfg = S.make_config(m, we.size)
    h = _capi.FilterHandle(cfg)
    with np.errstate(divide="ignore"):
        h.set_weights(np.log(we))
    return h


def shouldresample(pf):
    """shouldresample(pf) — reference src/resample.jl:5-10."""
    return pf._h.shouldresample()


def resample(a, we=None, M=None, U=None, seed=0, step=0):
    """resample(pf) / resample(T, we[, M]) / resample(we) — reference src/resample.jl:12-15.  Returns 0-based
    ancestor indices.  The uniforms the reference draws from the global RNG come from Philox(seed, step)
    unless given explicitly in U."""
    if isinstance(a, _AbstractParticleFilter):
        strategy, wev = a.resampling_strategy, a._h.expweights()
    elif we is None:
        strategy, wev = ResampleSystematic, np.asarray(a, dtype=np.float64)
    else:
        strategy, wev = a, np.asarray(we, dtype=np.float64)
    M = wev.size if M is None else int(M)
    if U is None:
        U = _capi.resample_uniforms(strategy.code, M, seed, step)
    return _capi.resample(strategy.code, wev, U, M)


def weighted_mean(pf):
    """weighted_mean(pf) — reference src/filtering.jl:541-549,568."""
    return pf._h.weighted_mean()


def logsumexp(w):
    """ll, w, we = logsumexp!(w, we) — reference src/utils.jl:18-27 (returns the normalised copies)."""
    return _capi.logsumexp(w)


def simulate(pf, T_or_u, du=None, p=None, dynamics_noise=True, measurement_noise=True, sample_initial=False, rng=None):
    """x, u, y = simulate(pf, T, du) / simulate(pf, u) — reference src/filtering.jl:457-477.  Host-side data
    generation (not part of the hot path); rng is a numpy Generator."""
    rng = np.random.default_rng(0) if rng is None else rng
    if np.isscalar(T_or_u):
        T = int(T_or_u)
        u = np.stack([du.rand(rng) for _ in range(T)]) if pf.nu else np.zeros((T, 0))
    else:
        u = np.asarray(T_or_u, dtype=np.float64).reshape(len(T_or_u), -1)
        T = u.shape[0]
    d0, df, dg = pf.initial_density, pf.dynamics_density, pf.measurement_density
    x = np.zeros((T, pf.nx))
    y = np.zeros((T, pf.ny))
    x[0] = d0.rand(rng) if sample_initial else d0.mean
    for t in range(T):
        ti = t * pf.Ts
        g = pf.measurement(x[t], u[t], p, ti)
        y[t] = g + (dg.rand(rng) if measurement_noise else 0.0)
        if t + 1 < T:
            if isinstance(pf.dynamics, QuadTankDynamics):
                fx = pf.dynamics(x[t], u[t], p, ti, pf.Ts)
            else:
                fx = pf.dynamics(x[t], u[t], p, ti)
            x[t + 1] = fx + (df.rand(rng) if dynamics_noise else 0.0)
    return x, u, y


# ---------------------------------------------------------------------------------------------------
# banks of independent filters (parameter sweeps; reference test/runtests.jl:412-417)
# ---------------------------------------------------------------------------------------------------
class FilterBank:
    """n independent filters sharing N, model family and dimensions, batched into single launches.
    `filters_spec` is a list of (dynamics, measurement, df, dg, d0) tuples."""

    def __init__(self, N, filters_spec, *, resample_threshold=0.1, resampling_strategy=ResampleSystematic,
                 rng=None, Ts=1.0, device=0):
        models = [_build_model(dy, me, df, dg, d0, Ts) for (dy, me, df, dg, d0) in filters_spec]
        self.Ts = float(Ts)
        self.N = int(N)
        self.rng = 0 if rng is None else int(rng)
        cfg = S.make_config(models[0], N, S.PARTICLE_FILTER, resampling_strategy.code, resample_threshold, self.rng, device)
        self._h = _capi.BankHandle(cfg, models)
        self.n_filters = len(models)

    def set_parameters(self, filters_spec):
        """new (dynamics, measurement, df, dg, d0) for every filter of the bank, nothing reallocated (llpf_bank_set_models)"""
        self._h.set_models([_build_model(dy, me, df, dg, d0, self.Ts) for (dy, me, df, dg, d0) in filters_spec])

    def loglik(self, u, y):
        """[loglik(pf_k, u, y) for k] — reference src/smoothing.jl:227-230 applied to every filter."""
        self._h.reset()
        return self._h.run(u, y, t_index0=1.0)["ll"]

    def forward(self, u, y, ll_steps=False):
        self._h.reset()
        return self._h.run(u, y, t_index0=0.0, ll_steps=ll_steps)
