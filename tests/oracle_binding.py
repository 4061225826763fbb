"""ctypes binding of oracle/libllpf_oracle.so — test infrastructure only (see oracle/llpf_oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from llpf_amd import _structs as S

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libllpf_oracle.so")

ORDER_REFERENCE, ORDER_DEVICE = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def _build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def load():
    if not os.path.exists(_SO):
        _build()
    lib = C.CDLL(_SO)
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.POINTER(S.Config), C.c_int]
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_seed.argtypes = [C.c_void_p, C.c_uint64]
    lib.orc_set_user_loglik.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]
    lib.orc_set_user_noise.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]
    lib.orc_set_user_initial.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int]
    lib.orc_reset.argtypes = [C.c_void_p]
    lib.orc_reset_explicit.argtypes = [C.c_void_p, _dp]
    lib.orc_correct.restype = C.c_double
    lib.orc_correct.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
    lib.orc_predict.argtypes = [C.c_void_p, _dp, C.c_double]
    lib.orc_predict_explicit.argtypes = [C.c_void_p, _dp, C.c_double, _dp, _dp]
    lib.orc_update.restype = C.c_double
    lib.orc_update.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
    lib.orc_run.restype = C.c_double
    lib.orc_run.argtypes = [C.c_void_p, _dp, _dp, C.c_int64, C.c_double, _dp, _dp, _dp, _dp, _dp]
    lib.orc_aux_correct.restype = C.c_double
    lib.orc_aux_correct.argtypes = [C.c_void_p]
    lib.orc_aux_predict.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
    lib.orc_aux_update.restype = C.c_double
    lib.orc_aux_update.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
    lib.orc_run_aux.restype = C.c_double
    lib.orc_run_aux.argtypes = [C.c_void_p, _dp, _dp, C.c_int64, C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib.orc_draw_one_categorical.restype = C.c_int64
    lib.orc_draw_one_categorical.argtypes = [_dp, _dp, C.c_int64, C.c_double, C.c_int]
    lib.orc_smooth.argtypes = [C.c_void_p, C.c_int64, _dp, C.c_int64, _dp, _dp, _dp, _dp, _ip]
    lib.orc_rb_get_R.argtypes = [C.c_void_p, _dp]
    lib.orc_rb_get_linear_state.argtypes = [C.c_void_p, _dp, _dp]
    lib.orc_num_particles.restype = C.c_int64
    lib.orc_num_particles.argtypes = [C.c_void_p]
    lib.orc_index.restype = C.c_int64
    lib.orc_index.argtypes = [C.c_void_p]
    for nm in ("orc_get_particles", "orc_get_weights", "orc_get_expweights", "orc_get_bins",
               "orc_set_particles", "orc_set_weights", "orc_weighted_mean"):
        getattr(lib, nm).argtypes = [C.c_void_p, _dp]
    lib.orc_get_ancestors.argtypes = [C.c_void_p, _ip]
    lib.orc_weighted_quantile.argtypes = [_dp, _dp, C.c_int64, _dp, C.c_int, _dp]
    lib.orc_weighted_quantile_dev.argtypes = [_dp, _dp, C.c_int64, _dp, C.c_int, _dp]
    lib.orc_filter_weighted_quantile.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
    lib.orc_set_index.argtypes = [C.c_void_p, C.c_int64]
    lib.orc_filter_ess.restype = C.c_double
    lib.orc_filter_ess.argtypes = [C.c_void_p]
    lib.orc_shouldresample.argtypes = [C.c_void_p]
    lib.orc_last_resampled.argtypes = [C.c_void_p]
    lib.orc_maxw.restype = C.c_double
    lib.orc_maxw.argtypes = [C.c_void_p]
    lib.orc_resample_count.restype = C.c_int64
    lib.orc_resample_count.argtypes = [C.c_void_p]
    lib.orc_degenerate.argtypes = [C.c_void_p]
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_get_threads.restype = C.c_int
    lib.orc_exact_steps.restype = C.c_int64
    lib.orc_exact_steps.argtypes = [C.c_void_p]
    lib.orc_logsumexp.restype = C.c_double
    lib.orc_logsumexp.argtypes = [_dp, _dp, C.c_int64, C.c_int, _dp]
    lib.orc_expnormalize.argtypes = [_dp, _dp, C.c_int64]
    lib.orc_expnormalize_inplace.argtypes = [_dp, C.c_int64]
    lib.orc_effective_particles.restype = C.c_double
    lib.orc_effective_particles.argtypes = [_dp, C.c_int64]
    lib.orc_resample.argtypes = [C.c_int, _dp, C.c_int64, C.c_int64, _dp, _ip, _dp, C.c_int]
    lib.orc_resample_uniforms.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_uint32, _dp]
    lib.orc_gauss_logpdf.restype = C.c_double
    lib.orc_gauss_logpdf.argtypes = [C.POINTER(S.Gaussian), _dp]
    lib.orc_gauss_sample.argtypes = [C.POINTER(S.Gaussian), _dp, _dp]
    lib.orc_dynamics.argtypes = [C.POINTER(S.Model), _dp, _dp, C.c_double, _dp]
    lib.orc_measurement.argtypes = [C.POINTER(S.Model), _dp, _dp, C.c_double, _dp]
    lib.orc_rk4_scalar_decay.argtypes = [C.c_double, C.c_double, C.c_int, _dp]
    lib.orc_kalman_loglik.restype = C.c_double
    lib.orc_kalman_loglik.argtypes = [C.POINTER(S.Model), _dp, _dp, C.c_int64]
    lib.orc_pairwise_sum.restype = C.c_double
    lib.orc_pairwise_sum.argtypes = [_dp, C.c_int64]
    lib.orc_math_vec.argtypes = [C.c_int, _dp, _dp, C.c_int64]
    lib.orc_philox_block.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    lib.orc_philox_block_engine.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    lib.orc_philox_block_engine.restype = C.c_int
    lib.orc_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _dp, C.c_int64]
    lib.orc_uniforms_nd.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _dp, C.c_int64]
    lib.orc_fix96.argtypes = [C.c_double, C.POINTER(C.c_uint64)]
    lib.orc_q64.restype = C.c_uint64
    lib.orc_q64.argtypes = [C.c_double, C.c_int]
    lib.orc_fix96_unit.argtypes = [C.c_double, C.POINTER(C.c_uint64)]
    lib.orc_q64_unit.restype = C.c_uint64
    lib.orc_q64_unit.argtypes = [C.c_double, C.c_int]
    lib.orc_u128_to_double.restype = C.c_double
    lib.orc_u128_to_double.argtypes = [C.c_uint64, C.c_uint64]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def set_threads(n):
    """OpenMP threads of the oracle's per-particle loops (results do not depend on it)."""
    lib().orc_set_threads(int(n))


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleFilter:
    """Thin object wrapper over orc_* (same verbs as the product's filter objects)."""

    def __init__(self, cfg, order=ORDER_REFERENCE):
        self.L = lib()
        self.cfg = cfg
        self.order = order
        self.h = self.L.orc_create(C.byref(cfg), order)
        if not self.h:
            raise ValueError("orc_create failed (bad covariance?)")
        self.N = cfg.n_particles
        self.nx, self.nu, self.ny = cfg.model.nx, cfg.model.nu, cfg.model.ny
        if cfg.model.model_id == S.MODEL_RB_BILINEAR:      # particles, history and means are [xn; xl] (RBParticle, reference src/rbpf.jl:24-30)
            self.nx = cfg.model.nx + cfg.model.rb.nxl

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def seed(self, s):
        self.L.orc_seed(self.h, s)

    def set_user_noise(self, kind, par):
        """process noise of the model's own (oracle/llpf_oracle.c: apply_noise): 1 multiplicative Gaussian (s0, s1), 2 Laplace (b)"""
        a = np.ascontiguousarray(par, dtype=np.float64)
        if self.L.orc_set_user_noise(self.h, int(kind), dptr(a), int(a.size)) != 0:
            raise ValueError("orc_set_user_noise")

    def set_user_initial(self, kind, par):
        """initial density of the model's own: 1 uniform box (lo[nx], hi[nx])"""
        a = np.ascontiguousarray(par, dtype=np.float64)
        if self.L.orc_set_user_initial(self.h, int(kind), dptr(a), int(a.size)) != 0:
            raise ValueError("orc_set_user_initial")

    def set_user_loglik(self, kind, par):
        """a measurement likelihood other than the Gaussian descriptor (llpf_oracle.c: 1 Laplace [b], 2 Student-t [nu, sigma, c1])"""
        a = np.ascontiguousarray(par, dtype=np.float64)
        if self.L.orc_set_user_loglik(self.h, int(kind), dptr(a), int(a.size)) != 0:
            raise ValueError("orc_set_user_loglik")

    def reset(self, xi=None):
        if xi is None:
            self.L.orc_reset(self.h)
        else:
            self.L.orc_reset_explicit(self.h, dptr(_f64(xi)))

    def correct(self, u, y, t):
        u = _f64(u)
        yy = None if y is None else _f64(y)
        return self.L.orc_correct(self.h, dptr(u), dptr(yy), float(t))

    def predict(self, u, t, xi=None, U=None):
        u = _f64(u)
        if xi is None:
            self.L.orc_predict(self.h, dptr(u), float(t))
        else:
            self.L.orc_predict_explicit(self.h, dptr(u), float(t), dptr(_f64(xi)), dptr(_f64(U)))

    def update(self, u, y, t):
        u = _f64(u)
        yy = None if y is None else _f64(y)
        return self.L.orc_update(self.h, dptr(u), dptr(yy), float(t))

    def run(self, U, Y, t_index0=0.0, ll_steps=False, xmean=False, history=False):
        U = _f64(U).reshape(-1, max(self.nu, 1)) if self.nu else np.zeros((len(Y), 0))
        Y = _f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        out = {}
        lls = np.zeros(T) if ll_steps else None
        xm = np.zeros((T, self.nx)) if xmean else None
        xh = np.zeros((T, self.N, self.nx)) if history else None
        wh = np.zeros((T, self.N)) if history else None
        weh = np.zeros((T, self.N)) if history else None
        ll = self.L.orc_run(self.h, dptr(U), dptr(Y), T, float(t_index0), dptr(lls), dptr(xm),
                            dptr(xh), dptr(wh), dptr(weh))
        out.update(ll=ll, ll_steps=lls, xmean=xm, x=xh, w=wh, we=weh)
        return out

    # AuxiliaryParticleFilter verbs (src/filtering.jl:170-217)
    def aux_correct(self):
        return self.L.orc_aux_correct(self.h)

    def aux_predict(self, u, y1, t):
        u = _f64(u)
        yy = None if y1 is None else _f64(y1)
        self.L.orc_aux_predict(self.h, dptr(u), dptr(yy), float(t))

    def aux_update(self, u, y1, t):
        u = _f64(u)
        yy = None if y1 is None else _f64(y1)
        return self.L.orc_aux_update(self.h, dptr(u), dptr(yy), float(t))

    def run_aux(self, U, Y, mode=0, ll_steps=False, xmean=False, history=False):
        """mode 0: forward_trajectory loop, mode 1: loglik loop (after reset!)."""
        U = _f64(U).reshape(-1, max(self.nu, 1)) if self.nu else np.zeros((len(Y), 0))
        Y = _f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        lls = np.zeros(T) if ll_steps else None
        xm = np.zeros((T, self.nx)) if xmean else None
        xh = np.zeros((T, self.N, self.nx)) if history else None
        wh = np.zeros((T, self.N)) if history else None
        weh = np.zeros((T, self.N)) if history else None
        ll = self.L.orc_run_aux(self.h, dptr(U), dptr(Y), T, int(mode), dptr(lls), dptr(xm), dptr(xh), dptr(wh), dptr(weh))
        return dict(ll=ll, ll_steps=lls, xmean=xm, x=xh, w=wh, we=weh)

    def smooth(self, M, U, xf, wf, wef):
        """xb [T, M, nx], idx [T, M] of smooth(pf, xf, wf, wef, ll, M, u, y) (src/smoothing.jl:116-143)."""
        xf, wf, wef = _f64(xf), _f64(wf), _f64(wef)
        T = wf.shape[0]
        U = _f64(U).reshape(T, max(self.nu, 1)) if self.nu else np.zeros((T, 1))
        xb = np.zeros((T, M, self.nx))
        idx = np.zeros((T, M), dtype=np.int64)
        rc = self.L.orc_smooth(self.h, int(M), dptr(U), T, dptr(xf), dptr(wf), dptr(wef), dptr(xb), iptr(idx))
        if rc != 0:
            raise ValueError("orc_smooth failed (%d)" % rc)
        return xb, idx

    def rb_R(self):
        nl = self.nx - self.cfg.model.nxn
        a = np.zeros((nl, nl))
        self.L.orc_rb_get_R(self.h, dptr(a))
        return a

    def rb_linear_state(self):
        """per-particle Kalman state of LLPF_MODEL_RB_BILINEAR: xl [N, nxl], R [N, nxl, nxl]"""
        nl = self.cfg.model.rb.nxl
        xl = np.zeros((self.N, nl))
        R = np.zeros((self.N, nl, nl))
        self.L.orc_rb_get_linear_state(self.h, dptr(xl), dptr(R))
        return xl, R

    def particles(self):
        a = np.empty((self.N, self.nx))
        self.L.orc_get_particles(self.h, dptr(a))
        return a

    def weights(self):
        a = np.empty(self.N)
        self.L.orc_get_weights(self.h, dptr(a))
        return a

    def expweights(self):
        a = np.empty(self.N)
        self.L.orc_get_expweights(self.h, dptr(a))
        return a

    def ancestors(self):
        a = np.empty(self.N, dtype=np.int64)
        self.L.orc_get_ancestors(self.h, iptr(a))
        return a

    def bins(self):
        a = np.empty(self.N)
        self.L.orc_get_bins(self.h, dptr(a))
        return a

    def set_particles(self, x):
        self.L.orc_set_particles(self.h, dptr(_f64(x)))

    def set_weights(self, w):
        self.L.orc_set_weights(self.h, dptr(_f64(w)))

    def set_index(self, t):
        self.L.orc_set_index(self.h, int(t))

    def index(self):
        return self.L.orc_index(self.h)

    def ess(self):
        return self.L.orc_filter_ess(self.h)

    def shouldresample(self):
        return bool(self.L.orc_shouldresample(self.h))

    def last_resampled(self):
        return bool(self.L.orc_last_resampled(self.h))

    def maxw(self):
        return self.L.orc_maxw(self.h)

    def resample_count(self):
        return self.L.orc_resample_count(self.h)

    def exact_steps(self):
        return self.L.orc_exact_steps(self.h)

    def weighted_mean(self):
        a = np.empty(self.nx)
        self.L.orc_weighted_mean(self.h, dptr(a))
        return a

    def weighted_quantile(self, q):
        q = _f64(np.atleast_1d(q))
        out = np.empty((q.size, self.nx))
        if self.L.orc_filter_weighted_quantile(self.h, dptr(q), q.size, dptr(out)):
            raise ValueError("weighted_quantile: no particle carries weight")
        return out


def weighted_quantile(v, w, q):
    """StatsBase.quantile(v, ProbabilityWeights(w), q) as restated in the oracle (orc_weighted_quantile)"""
    v, w, q = _f64(v), _f64(w), _f64(np.atleast_1d(q))
    out = np.empty(q.size)
    if lib().orc_weighted_quantile(dptr(v), dptr(w), v.size, dptr(q), q.size, dptr(out)):
        raise ValueError("weighted_quantile: empty or weightless input")
    return out


def weighted_quantile_dev(v, w, q):
    """the same quantile in device order (orc_weighted_quantile_dev: integer crossing, every particle with w > 0 present)"""
    v, w, q = _f64(v), _f64(w), _f64(np.atleast_1d(q))
    out = np.empty(q.size)
    if lib().orc_weighted_quantile_dev(dptr(v), dptr(w), v.size, dptr(q), q.size, dptr(out)):
        raise ValueError("weighted_quantile: empty or weightless input")
    return out


def draw_one_categorical(w, u, order=ORDER_REFERENCE):
    w = _f64(w).copy()
    bins = np.zeros(w.size)
    return lib().orc_draw_one_categorical(dptr(w), dptr(bins), w.size, float(u), order)


def logsumexp(w, order=ORDER_REFERENCE):
    w = _f64(w).copy()
    we = np.empty_like(w)
    mw = C.c_double(0)
    ll = lib().orc_logsumexp(dptr(w), dptr(we), w.size, order, C.byref(mw))
    return ll, w, we, mw.value


def resample(strategy, we, U, m=None, order=ORDER_REFERENCE, j0=None):
    we = _f64(we)
    n = we.size
    m = n if m is None else m
    j = np.zeros(m, dtype=np.int64) if j0 is None else np.ascontiguousarray(j0, dtype=np.int64).copy()
    bins = np.zeros(n)
    U = _f64(np.atleast_1d(U))
    rc = lib().orc_resample(strategy, dptr(we), n, m, dptr(U), iptr(j), dptr(bins), order)
    if rc != 0:
        raise ValueError("degenerate weights")
    return j, bins


def resample_uniforms(strategy, m, seed, step):
    u = np.zeros(1 if strategy == S.RESAMPLE_SYSTEMATIC else m)
    lib().orc_resample_uniforms(strategy, m, seed, step, dptr(u))
    return u


def math_vec(which, x):
    x = _f64(x)
    out = np.empty_like(x)
    lib().orc_math_vec(which, dptr(x), dptr(out), x.size)
    return out


def normals(seed, step, stream, nd, n):
    out = np.empty((n, nd))
    lib().orc_normals(seed, step, stream, nd, dptr(out), n)
    return out


def uniforms_nd(seed, step, stream, nd, n):
    """nd uniforms in [0, 1) per particle (llpf_uniforms): what a model's own noise / initial density is handed"""
    out = np.empty((n, nd))
    lib().orc_uniforms_nd(seed, step, stream, nd, dptr(out), n)
    return out


def kalman_loglik(model, U, Y):
    U = _f64(U)
    Y = _f64(Y).reshape(-1, model.ny)
    return lib().orc_kalman_loglik(C.byref(model), dptr(U), dptr(Y), Y.shape[0])
