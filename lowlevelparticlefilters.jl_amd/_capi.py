"""ctypes binding of libllpf_hip.so — one Python function per symbol of include/llpf.h.

This is the same binding a Julia maintainer writes with `ccall` (see INTEGRATION.md and
julia/LLPFAmd.jl); the tests drive the library through it.  There is no fallback: if the
shared library is missing or no GPU is visible, loading / constructing raises.
"""
import ctypes as C
import os

import numpy as np

from . import _structs as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLPF_LIB") or os.path.join(_HERE, "libllpf_hip.so")   # LLPF_LIB: A/B builds of the same engine (tools/)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_vp = C.c_void_p

# every exported symbol of include/llpf.h with its argument types (restype is int unless noted)
SYMBOLS = {
    "llpf_create": [C.POINTER(S.Config), C.POINTER(_vp)],
    "llpf_destroy": [_vp],
    "llpf_reset": [_vp],
    "llpf_seed": [_vp, C.c_uint64],
    "llpf_set_model": [_vp, _vp],
    "llpf_correct": [_vp, _dp, _dp, C.c_double, _dp],
    "llpf_predict": [_vp, _dp, C.c_double],
    "llpf_update": [_vp, _dp, _dp, C.c_double, _dp],
    "llpf_run": [_vp, _dp, _dp, C.c_int64, C.c_double, _dp, C.POINTER(S.RunOutputs)],
    "llpf_smooth": [_vp, C.c_int64, _dp, C.c_int64, _dp, _dp, _dp, _dp, _ip],
    "llpf_rb_get_covariance": [_vp, _dp],
    "llpf_rb_get_linear_state": [_vp, _dp, _dp],
    "llpf_aux_correct": [_vp, _dp],
    "llpf_aux_predict": [_vp, _dp, _dp, C.c_double],
    "llpf_aux_update": [_vp, _dp, _dp, C.c_double, _dp],
    "llpf_aux_run": [_vp, _dp, _dp, C.c_int64, C.c_int32, _dp, C.POINTER(S.RunOutputs)],
    "llpf_bank_aux_run": [_vp, _dp, _dp, C.c_int64, C.c_int32, _dp, _dp],
    "llpf_num_particles": [_vp, _ip],
    "llpf_index": [_vp, _ip],
    "llpf_get_particles": [_vp, _dp],
    "llpf_get_weights": [_vp, _dp],
    "llpf_get_expweights": [_vp, _dp],
    "llpf_get_ancestors": [_vp, _ip],
    "llpf_get_bins": [_vp, _dp],
    "llpf_set_particles": [_vp, _dp],
    "llpf_set_weights": [_vp, _dp],
    "llpf_set_index": [_vp, C.c_int64],
    "llpf_effective_particles": [_vp, _dp],
    "llpf_shouldresample": [_vp, C.POINTER(C.c_int32)],
    "llpf_weighted_mean": [_vp, _dp],
    "llpf_last_resampled": [_vp, C.POINTER(C.c_int32)],
    "llpf_maxw": [_vp, _dp],
    "llpf_logsumexp": [C.c_int32, _dp, _dp, C.c_int64, _dp],
    "llpf_resample": [C.c_int32, C.c_int32, _dp, C.c_int64, C.c_int64, _dp, _ip],
    "llpf_resample_uniforms": [C.c_int32, C.c_int64, C.c_uint64, C.c_uint32, _dp],
    "llpf_bank_create": [C.POINTER(S.Config), C.POINTER(S.Model), C.c_int32, C.POINTER(_vp)],
    "llpf_bank_destroy": [_vp],
    "llpf_bank_reset": [_vp],
    "llpf_bank_seed": [_vp, C.c_uint64],
    "llpf_bank_set_models": [_vp, _vp],
    "llpf_bank_run": [_vp, _dp, _dp, C.c_int64, C.c_double, _dp, _dp],
    "llpf_bank_run_multi": [_vp, _dp, _dp, C.c_int64, C.c_double, _dp, _dp, _dp],
    "llpf_mbank_create": [C.POINTER(S.Config), C.POINTER(S.Model), C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_vp)],
    "llpf_mbank_unique_id": [C.POINTER(C.c_uint8)],
    "llpf_mbank_partition": [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "llpf_mbank_create_rank": [C.POINTER(S.Config), C.POINTER(S.Model), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.POINTER(_vp)],
    "llpf_mbank_destroy": [_vp],
    "llpf_mbank_reset": [_vp],
    "llpf_mbank_seed": [_vp, C.c_uint64],
    "llpf_mbank_set_models": [_vp, _vp],
    "llpf_mbank_run": [_vp, _dp, _dp, C.c_int64, C.c_double, _dp, _dp],
    "llpf_mbank_aux_run": [_vp, _dp, _dp, C.c_int64, C.c_int32, _dp, _dp],
    "llpf_mbank_info": [_vp, C.POINTER(S.MBankInfo)],
    "llpf_mbank_local_devices": [_vp, C.POINTER(C.c_int32)],
    "llpf_mbank_set_profiling": [_vp, C.c_int32],
    "llpf_mbank_get_profile": [_vp, C.c_int32, _dp, _ip],
    "llpf_model_compile": [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)],
    "llpf_model_traits": [C.c_int32, C.POINTER(C.c_int32)],
    "llpf_weighted_cov": [_vp, _dp],
    "llpf_weighted_quantile": [_vp, _dp, C.c_int32, _dp],
    "llpf_set_profiling": [_vp, C.c_int32],
    "llpf_get_profile": [_vp, _dp, _ip],
    "llpf_bank_set_profiling": [_vp, C.c_int32],
    "llpf_bank_get_profile": [_vp, _dp, _ip],
    "llpf_resample_count": [_vp, _ip],
    "llpf_bank_resample_count": [_vp, _ip],
    "llpf_last_run_ms": [_vp, _dp],
    "llpf_last_run_stats": [_vp, _ip, _ip, _dp],
    "llpf_bank_last_run_ms": [_vp, _dp],
    "llpf_last_error": [],
    "llpf_version": [C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "llpf_device_count": [C.POINTER(C.c_int32)],
    "llpf_selftest_math": [C.c_int32, C.c_int32, _dp, _dp, C.c_int64],
    "llpf_selftest_normals": [C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int32, _dp, C.c_int64],
}

OK, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_DEGENERATE, ERR_ALLOC, ERR_INTERNAL = 0, 1, 2, 3, 4, 5, 6
PROF_CLASSES = 4


class LLPFError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("llpf status %d: %s" % (code, msg))
        self.code = code


class DegenerateWeights(LLPFError):
    pass


_lib = None


def lib():
    """Load libllpf_hip.so (fails loudly if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libllpf_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, args in SYMBOLS.items():
            if os.environ.get("LLPF_LIB") and not hasattr(L, name):
                continue             # A/B run against an older build of the engine (tools/ab): newer entry points are simply absent
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_char_p if name == "llpf_last_error" else C.c_int
        _lib = L
    return _lib


def check(code):
    if code != OK:
        msg = lib().llpf_last_error().decode("utf-8", "replace")
        if code == ERR_DEGENERATE:
            raise DegenerateWeights(code, msg)
        raise LLPFError(code, msg)


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int32(0)
    lib().llpf_device_count(C.byref(n))
    return n.value


class FilterHandle:
    """RAII wrapper of an `llpf_filter*` (one filter on one device)."""

    def __init__(self, cfg):
        self.L = lib()
        self.cfg = cfg
        self.h = _vp()
        check(self.L.llpf_create(C.byref(cfg), C.byref(self.h)))
        self.N = int(cfg.n_particles)
        self.nx, self.nu, self.ny = cfg.model.nx, cfg.model.nu, cfg.model.ny
        if cfg.model.model_id == S.MODEL_RB_BILINEAR:      # particles, history and means are [xn; xl] (RBParticle, reference src/rbpf.jl:24-30)
            self.nx = cfg.model.nx + cfg.model.rb.nxl

    def close(self):
        if getattr(self, "h", None):
            self.L.llpf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- step ---
    def reset(self):
        check(self.L.llpf_reset(self.h))

    def seed(self, s):
        check(self.L.llpf_seed(self.h, int(s) & 0xFFFFFFFFFFFFFFFF))

    def set_model(self, model):
        """new parameters, same model family and dimensions (the reference's filter_from_parameters(theta, pf)): nothing is reallocated"""
        check(self.L.llpf_set_model(self.h, C.byref(model)))

    def _u(self, u):
        if self.nu == 0:
            return None
        u = f64(u).reshape(-1)
        if u.size != self.nu:
            raise ValueError("u must have %d elements" % self.nu)
        return u

    def _y(self, y):
        if y is None:
            return None
        y = f64(y).reshape(-1)
        if y.size != self.ny:
            raise ValueError("y must have %d elements" % self.ny)
        return y

    def correct(self, u, y, t):
        u, y = self._u(u), self._y(y)
        ll = C.c_double(0)
        check(self.L.llpf_correct(self.h, dptr(u), dptr(y), float(t), C.byref(ll)))
        return ll.value

    def predict(self, u, t):
        u = self._u(u)
        check(self.L.llpf_predict(self.h, dptr(u), float(t)))

    def update(self, u, y, t):
        u, y = self._u(u), self._y(y)
        ll = C.c_double(0)
        check(self.L.llpf_update(self.h, dptr(u), dptr(y), float(t), C.byref(ll)))
        return ll.value

    def run(self, U, Y, t_index0=0.0, ll_steps=False, xmean=False, history=False, xcov=False, quantiles=None):
        """quantiles: probabilities q -> res["xquant"] [T, nx, len(q)], weighted_quantile of every timestep's state on the device"""
        Y = f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        U = f64(U).reshape(T, self.nu) if self.nu else None
        outs = S.RunOutputs()
        res = {}
        if quantiles is not None:
            qp = np.ascontiguousarray(np.atleast_1d(quantiles), dtype=np.float64)
            res["xquant"] = np.zeros((T, self.nx, qp.size))
            outs.xquant, outs.quant_p, outs.nq = dptr(res["xquant"]), dptr(qp), int(qp.size)
        if xcov:
            res["xcov"] = np.zeros((T, self.nx, self.nx))
            outs.xcov = dptr(res["xcov"])
        if ll_steps:
            res["ll_steps"] = np.zeros(T)
            outs.ll_steps = dptr(res["ll_steps"])
        if xmean:
            res["xmean"] = np.zeros((T, self.nx))
            outs.xmean = dptr(res["xmean"])
        if history:
            res["x"] = np.zeros((T, self.N, self.nx))
            res["w"] = np.zeros((T, self.N))
            res["we"] = np.zeros((T, self.N))
            outs.x_hist, outs.w_hist, outs.we_hist = dptr(res["x"]), dptr(res["w"]), dptr(res["we"])
        ll = C.c_double(0)
        check(self.L.llpf_run(self.h, dptr(U), dptr(Y), T, float(t_index0), C.byref(ll), C.byref(outs)))
        res["ll"] = ll.value
        return res

    def weighted_cov(self):
        """weighted_cov of the current particles under the current weights (reference src/filtering.jl:571-581), on the device"""
        a = np.zeros((self.nx, self.nx))
        check(self.L.llpf_weighted_cov(self.h, dptr(a)))
        return a

    def weighted_quantile(self, q):
        """weighted_quantile of the current particles under the current weights (reference src/filtering.jl:583-595), on the device.
        The raw handle returns the C ABI's layout, [len(q), nx]; the mirror of the reference's function (api.weighted_quantile) transposes it
        to the reference's [state][q] nesting."""
        q = np.ascontiguousarray(np.atleast_1d(q), dtype=np.float64)
        out = np.empty((q.size, self.nx))
        check(self.L.llpf_weighted_quantile(self.h, dptr(q), q.size, dptr(out)))
        return out

    def rb_covariance(self):
        """x[1].R of an RBPF: the covariance of the linear substate shared by all particles."""
        nl = self.nx - self.cfg.model.nxn
        a = np.zeros((nl, nl))
        check(self.L.llpf_rb_get_covariance(self.h, dptr(a)))
        return a

    def rb_linear_state(self):
        """per-particle Kalman state of LLPF_MODEL_RB_BILINEAR: xl [N, nxl], R [N, nxl, nxl] (fields of RBParticle)."""
        nl = self.cfg.model.rb.nxl
        xl = np.zeros((self.N, nl))
        R = np.zeros((self.N, nl, nl))
        check(self.L.llpf_rb_get_linear_state(self.h, dptr(xl), dptr(R)))
        return xl, R

    def smooth(self, M, U, xf, wf, wef):
        """xb [T, M, nx], idx [T, M]: smooth(pf, xf, wf, wef, ll, M, u, y) — reference src/smoothing.jl:116-143."""
        xf, wf, wef = f64(xf), f64(wf), f64(wef)
        T = wf.shape[0]
        U = f64(U).reshape(T, self.nu) if self.nu else None
        xb = np.zeros((T, int(M), self.nx))
        idx = np.zeros((T, int(M)), dtype=np.int64)
        check(self.L.llpf_smooth(self.h, int(M), dptr(U), T, dptr(xf), dptr(wf), dptr(wef), dptr(xb), iptr(idx)))
        return xb, idx

    # --- AuxiliaryParticleFilter verbs (reference src/filtering.jl:170-217) ---
    def aux_correct(self):
        ll = C.c_double(0)
        check(self.L.llpf_aux_correct(self.h, C.byref(ll)))
        return ll.value

    def aux_predict(self, u, y1, t):
        u, y1 = self._u(u), self._y(y1)
        check(self.L.llpf_aux_predict(self.h, dptr(u), dptr(y1), float(t)))

    def aux_update(self, u, y1, t):
        u, y1 = self._u(u), self._y(y1)
        ll = C.c_double(0)
        check(self.L.llpf_aux_update(self.h, dptr(u), dptr(y1), float(t), C.byref(ll)))
        return ll.value

    def run_aux(self, U, Y, mode=0, ll_steps=False, xmean=False, history=False):
        """mode 0: forward_trajectory loop, mode 1: loglik loop of the AuxiliaryParticleFilter (after reset)."""
        Y = f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        U = f64(U).reshape(T, self.nu) if self.nu else None
        outs = S.RunOutputs()
        res = {}
        if ll_steps:
            res["ll_steps"] = np.zeros(T)
            outs.ll_steps = dptr(res["ll_steps"])
        if xmean:
            res["xmean"] = np.zeros((T, self.nx))
            outs.xmean = dptr(res["xmean"])
        if history:
            res["x"] = np.zeros((T, self.N, self.nx))
            res["w"] = np.zeros((T, self.N))
            res["we"] = np.zeros((T, self.N))
            outs.x_hist, outs.w_hist, outs.we_hist = dptr(res["x"]), dptr(res["w"]), dptr(res["we"])
        ll = C.c_double(0)
        check(self.L.llpf_aux_run(self.h, dptr(U), dptr(Y), T, int(mode), C.byref(ll), C.byref(outs)))
        res["ll"] = ll.value
        return res

    # --- accessors ---
    def index(self):
        t = C.c_int64(0)
        check(self.L.llpf_index(self.h, C.byref(t)))
        return t.value

    def set_index(self, t):
        check(self.L.llpf_set_index(self.h, int(t)))

    def particles(self):
        a = np.empty((self.N, self.nx))
        check(self.L.llpf_get_particles(self.h, dptr(a)))
        return a

    def weights(self):
        a = np.empty(self.N)
        check(self.L.llpf_get_weights(self.h, dptr(a)))
        return a

    def expweights(self):
        a = np.empty(self.N)
        check(self.L.llpf_get_expweights(self.h, dptr(a)))
        return a

    def ancestors(self):
        a = np.empty(self.N, dtype=np.int64)
        check(self.L.llpf_get_ancestors(self.h, iptr(a)))
        return a

    def bins(self):
        a = np.empty(self.N)
        check(self.L.llpf_get_bins(self.h, dptr(a)))
        return a

    def set_particles(self, x):
        x = f64(x).reshape(self.N, self.nx)
        check(self.L.llpf_set_particles(self.h, dptr(x)))

    def set_weights(self, w):
        w = f64(w).reshape(self.N)
        check(self.L.llpf_set_weights(self.h, dptr(w)))

    def ess(self):
        v = C.c_double(0)
        check(self.L.llpf_effective_particles(self.h, C.byref(v)))
        return v.value

    def shouldresample(self):
        v = C.c_int32(0)
        check(self.L.llpf_shouldresample(self.h, C.byref(v)))
        return bool(v.value)

    def last_resampled(self):
        v = C.c_int32(0)
        check(self.L.llpf_last_resampled(self.h, C.byref(v)))
        return bool(v.value)

    def maxw(self):
        v = C.c_double(0)
        check(self.L.llpf_maxw(self.h, C.byref(v)))
        return v.value

    def weighted_mean(self):
        a = np.empty(self.nx)
        check(self.L.llpf_weighted_mean(self.h, dptr(a)))
        return a

    def resample_count(self):
        v = C.c_int64(0)
        check(self.L.llpf_resample_count(self.h, C.byref(v)))
        return v.value

    def last_run_ms(self):
        v = C.c_double(0)
        check(self.L.llpf_last_run_ms(self.h, C.byref(v)))
        return v.value

    def last_run_stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
        check(self.L.llpf_last_run_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"fused_launches": a.value, "source_side_timesteps": b.value, "survivor_fraction": c.value}

    def set_profiling(self, on):
        check(self.L.llpf_set_profiling(self.h, 1 if on else 0))

    def profile(self):
        ms = np.zeros(PROF_CLASSES)
        n = np.zeros(PROF_CLASSES, dtype=np.int64)
        check(self.L.llpf_get_profile(self.h, dptr(ms), iptr(n)))
        return ms, n


class BankHandle:
    """RAII wrapper of an `llpf_bank*` (many independent filters on one device)."""

    def __init__(self, base_cfg, models=None, n_filters=None):
        """models: one llpf_model per filter, or None with n_filters: every filter uses base_cfg.model (Monte-Carlo replicas)."""
        self.L = lib()
        self.cfg = base_cfg
        self.h = _vp()
        if models is None:
            self.F = int(n_filters)
            self._models = None
            check(self.L.llpf_bank_create(C.byref(base_cfg), None, self.F, C.byref(self.h)))
            m0 = base_cfg.model
        else:
            self.F = len(models)
            arr = (S.Model * self.F)(*models)
            self._models = arr
            check(self.L.llpf_bank_create(C.byref(base_cfg), arr, self.F, C.byref(self.h)))
            m0 = models[0]
        self.N = int(base_cfg.n_particles)
        self.nx, self.nu, self.ny = m0.nx, m0.nu, m0.ny

    def close(self):
        if getattr(self, "h", None):
            self.L.llpf_bank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self.L.llpf_bank_reset(self.h))

    def seed(self, s):
        check(self.L.llpf_bank_seed(self.h, int(s) & 0xFFFFFFFFFFFFFFFF))

    def set_models(self, models):
        """new parameters for every filter of the bank (len(models) == n_filters), same model family and dimensions"""
        if len(models) != self.F:
            raise ValueError("set_models: %d models for a bank of %d filters" % (len(models), self.F))
        arr = (S.Model * self.F)(*models)
        check(self.L.llpf_bank_set_models(self.h, arr))
        self._models = arr

    def run(self, U, Y, t_index0=0.0, ll_steps=False):
        Y = f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        U = f64(U).reshape(T, self.nu) if self.nu else None
        ll = np.zeros(self.F)
        lls = np.zeros((T, self.F)) if ll_steps else None
        check(self.L.llpf_bank_run(self.h, dptr(U), dptr(Y), T, float(t_index0), dptr(ll), dptr(lls)))
        return {"ll": ll, "ll_steps": lls}

    def run_multi(self, U, Y, t_index0=0.0, ll_steps=False, xmean=False):
        """every filter runs on inputs of its own: U [F, T, nu], Y [F, T, ny]; xmean -> [T, F, nx] weighted means after correct!"""
        Y = f64(Y).reshape(self.F, -1, self.ny)
        T = Y.shape[1]
        U = f64(U).reshape(self.F, T, self.nu) if self.nu else None
        ll = np.zeros(self.F)
        lls = np.zeros((T, self.F)) if ll_steps else None
        xm = np.zeros((T, self.F, self.nx)) if xmean else None
        check(self.L.llpf_bank_run_multi(self.h, dptr(U), dptr(Y), T, float(t_index0), dptr(ll), dptr(lls), dptr(xm)))
        return {"ll": ll, "ll_steps": lls, "xmean": xm}

    def run_aux(self, U, Y, mode=1, ll_steps=False):
        Y = f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        U = f64(U).reshape(T, self.nu) if self.nu else None
        ll = np.zeros(self.F)
        lls = np.zeros((T, self.F)) if ll_steps else None
        check(self.L.llpf_bank_aux_run(self.h, dptr(U), dptr(Y), T, int(mode), dptr(ll), dptr(lls)))
        return {"ll": ll, "ll_steps": lls}

    def last_run_ms(self):
        v = C.c_double(0)
        check(self.L.llpf_bank_last_run_ms(self.h, C.byref(v)))
        return v.value

    def resample_count(self):
        v = C.c_int64(0)
        check(self.L.llpf_bank_resample_count(self.h, C.byref(v)))
        return v.value

    def set_profiling(self, on):
        check(self.L.llpf_bank_set_profiling(self.h, 1 if on else 0))

    def profile(self):
        ms = np.zeros(PROF_CLASSES)
        n = np.zeros(PROF_CLASSES, dtype=np.int64)
        check(self.L.llpf_bank_get_profile(self.h, dptr(ms), iptr(n)))
        return ms, n

MBANK_ID_BYTES = 128
MBANK_COLL = {0: "none", 1: "rccl", 2: "host", 3: "external"}


def mbank_unique_id():
    """ncclGetUniqueId through the C ABI (rank 0 of a one-process-per-GPU job; the host distributes the bytes)."""
    buf = (C.c_uint8 * MBANK_ID_BYTES)()
    check(lib().llpf_mbank_unique_id(buf))
    return bytes(buf)


class MBankHandle:
    """RAII wrapper of an `llpf_mbank*`: a sweep of independent filters sharded over GPUs, filter k on shard k mod n_shards;
    the exchange of the log-likelihood vector (RCCL) happens inside run().

    devices=[...]             : this process drives all listed GPUs (llpf_mbank_create)
    rank=, world=, unique_id= : one process per GPU (llpf_mbank_create_rank); unique_id None with world > 1 leaves the
                                exchange to the caller (run() then returns this rank's slots, zeros elsewhere)"""

    def __init__(self, base_cfg, models=None, n_filters=None, devices=None, rank=None, world=None, unique_id=None):
        self.L = lib()
        self.cfg = base_cfg
        self.h = _vp()
        if models is None:
            self.F = int(n_filters)
            arr = None
            m0 = base_cfg.model
        else:
            self.F = len(models)
            arr = (S.Model * self.F)(*models)
            m0 = models[0]
        self._models = arr
        if rank is None:
            devs = list(devices if devices is not None else [base_cfg.device])
            darr = (C.c_int32 * len(devs))(*devs)
            check(self.L.llpf_mbank_create(C.byref(base_cfg), arr, self.F, darr, len(devs), C.byref(self.h)))
        else:
            idp = None
            if unique_id is not None:
                if len(unique_id) != MBANK_ID_BYTES:
                    raise ValueError("unique_id must have %d bytes" % MBANK_ID_BYTES)
                idp = (C.c_uint8 * MBANK_ID_BYTES)(*unique_id)
            check(self.L.llpf_mbank_create_rank(C.byref(base_cfg), arr, self.F, int(rank), int(world), idp, C.byref(self.h)))
        self.N = int(base_cfg.n_particles)
        self.nx, self.nu, self.ny = m0.nx, m0.nu, m0.ny

    def close(self):
        if getattr(self, "h", None):
            self.L.llpf_mbank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self.L.llpf_mbank_reset(self.h))

    def seed(self, s):
        check(self.L.llpf_mbank_seed(self.h, int(s) & 0xFFFFFFFFFFFFFFFF))

    def set_models(self, models):
        """new parameters for every filter of the sweep (all n_filters descriptors, on every rank)"""
        if len(models) != self.F:
            raise ValueError("set_models: %d models for a sweep of %d filters" % (len(models), self.F))
        arr = (S.Model * self.F)(*models)
        check(self.L.llpf_mbank_set_models(self.h, arr))
        self._models = arr

    def _io(self, U, Y):
        Y = f64(Y).reshape(-1, self.ny)
        T = Y.shape[0]
        U = f64(U).reshape(T, self.nu) if self.nu else None
        return U, Y, T

    def run(self, U, Y, t_index0=0.0):
        U, Y, T = self._io(U, Y)
        ll = np.zeros(self.F)
        tot = C.c_double(0)
        check(self.L.llpf_mbank_run(self.h, dptr(U), dptr(Y), T, float(t_index0), dptr(ll), C.byref(tot)))
        return {"ll": ll, "ll_sum": tot.value}

    def run_aux(self, U, Y, mode=1):
        U, Y, T = self._io(U, Y)
        ll = np.zeros(self.F)
        tot = C.c_double(0)
        check(self.L.llpf_mbank_aux_run(self.h, dptr(U), dptr(Y), T, int(mode), dptr(ll), C.byref(tot)))
        return {"ll": ll, "ll_sum": tot.value}

    def info(self):
        i = S.MBankInfo()
        check(self.L.llpf_mbank_info(self.h, C.byref(i)))
        d = {k: getattr(i, k) for k, _ in S.MBankInfo._fields_}
        d["collective"] = MBANK_COLL.get(d["collective"], d["collective"])
        devs = (C.c_int32 * max(1, i.n_local_shards))()
        check(self.L.llpf_mbank_local_devices(self.h, devs))
        d["local_devices"] = list(devs)[: i.n_local_shards]
        return d

    def last_run_ms(self):
        return self.info()["last_run_ms"]

    def resample_count(self):
        return self.info()["resample_count"]

    def set_profiling(self, on):
        check(self.L.llpf_mbank_set_profiling(self.h, 1 if on else 0))

    def profile(self, local_shard=0):
        ms = np.zeros(PROF_CLASSES)
        n = np.zeros(PROF_CLASSES, dtype=np.int64)
        check(self.L.llpf_mbank_get_profile(self.h, int(local_shard), dptr(ms), iptr(n)))
        return ms, n


def mbank_partition(n_filters, shard, n_shards):
    """llpf_mbank_partition: global indices of the filters shard `shard` of `n_shards` owns (pure host code)"""
    n = C.c_int32(0)
    check(lib().llpf_mbank_partition(int(n_filters), int(shard), int(n_shards), None, C.byref(n)))
    a = (C.c_int32 * max(1, n.value))()
    check(lib().llpf_mbank_partition(int(n_filters), int(shard), int(n_shards), a, C.byref(n)))
    return [int(a[i]) for i in range(n.value)]


def model_compile(device_src, nx, ny):
    """llpf_model_compile: JIT a user model (HIP source defining `struct UserModel`, include/llpf.h); returns the model id
    to put into llpf_model.model_id"""
    mid = C.c_int32(-1)
    check(lib().llpf_model_compile(device_src.encode("utf-8"), int(nx), int(ny), C.byref(mid)))
    return mid.value


TRAIT_LOGLIK, TRAIT_LOGLIK_BOUND, TRAIT_NOISE, TRAIT_INITIAL = 1, 2, 4, 8


def model_traits(model_id):
    """llpf_model_traits: which optional members (loglik, loglik_bound, noise, initial) a compiled model has, as TRAIT_* bits"""
    t = C.c_int32(0)
    check(lib().llpf_model_traits(int(model_id), C.byref(t)))
    return t.value


# array primitives -----------------------------------------------------------------------------------
def logsumexp(w, device=0):
    """ll, w_normalised, we = logsumexp!(w, we)  (reference src/utils.jl:18-27), computed on the GPU."""
    w = f64(w).copy()
    we = np.empty_like(w)
    ll = C.c_double(0)
    check(lib().llpf_logsumexp(device, dptr(w), dptr(we), w.size, C.byref(ll)))
    return ll.value, w, we


def resample(strategy, we, U, m=None, j0=None, device=0):
    """j = resample(strategy, we, M) (reference src/resample.jl:12-61), 0-based, computed on the GPU."""
    we = f64(we)
    n = we.size
    m = n if m is None else int(m)
    j = np.zeros(m, dtype=np.int64) if j0 is None else np.ascontiguousarray(j0, dtype=np.int64).copy()
    U = f64(np.atleast_1d(U))
    need = 1 if strategy == S.RESAMPLE_SYSTEMATIC else m
    if U.size < need or j.size < m:
        raise ValueError("resample: %d uniform(s) and %d ancestor slots are needed" % (need, m))
    check(lib().llpf_resample(device, strategy, dptr(we), n, m, dptr(U), iptr(j)))
    return j


def resample_uniforms(strategy, m, seed, step):
    u = np.zeros(1 if strategy == S.RESAMPLE_SYSTEMATIC else m)
    check(lib().llpf_resample_uniforms(strategy, m, int(seed) & 0xFFFFFFFFFFFFFFFF, step, dptr(u)))
    return u


def selftest_math(which, x, device=0):
    x = f64(x)
    out = np.empty_like(x)
    check(lib().llpf_selftest_math(device, which, dptr(x), dptr(out), x.size))
    return out


def selftest_normals(seed, step, stream, nd, n, device=0):
    out = np.empty((n, nd))
    check(lib().llpf_selftest_normals(device, int(seed), step, stream, nd, dptr(out), n))
    return out
