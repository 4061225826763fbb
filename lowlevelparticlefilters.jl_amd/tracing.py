"""Callables without device source (round 6).

The reference takes its model as Julia closures — `dynamics(x, u, p, t)`, `measurement(x, u, p, t)`, `measurement_likelihood(x, u, y, p, t)`
(src/PFtypes.jl:59-63, 189-193).  A closure cannot run on the GPU, and until round 5 a model the engine has no descriptor for had to be
written as HIP source (UserDynamics).  This module closes most of that distance for straight-line models: the callable is run ONCE on
tracer numbers that record the expression DAG (+ - * / sqrt exp log abs maximum minimum ifelse, comparisons, integer powers; constants
— everything read from `p` — are baked in with all their bits), and the DAG is emitted as the `struct UserModel` snippet that
`llpf_model_compile` builds with hiprtc.  Every node becomes one IEEE operation in the order the callable performed it (the engine is
compiled with -ffp-contract=off), so a traced model written with the built-in model's expression order reproduces it bit for bit
(tests/test_tracing.py: the quad-tank of examples/example_quadtank.jl:8-27 through rk4 of src/utils.jl:220-237).

    dyn = traced_dynamics(f, nx, nu, p=p, measurement=g, ny=ny)          # f(x, u, p, t) -> nx values, g(x, u, p, t) -> ny values
    pf  = AdvancedParticleFilter(N, dyn, dyn.measurement_model, None, df, d0)   # or ParticleFilter(N, dyn, dyn.measurement_model, df, dg, d0)

What cannot be traced: data-dependent Python control flow (`if x[0] > 0:` — use ifelse), loops whose trip count depends on the state,
calls into libraries that do not accept these numbers.  Those still go through UserDynamics.  The Julia wrapper's twin is
`trace_dynamics` in julia/LLPFAmd.jl (a `Tr <: Real` number type with the same node set)."""
import math
import struct

import numpy as np

__all__ = ["Tr", "sqrt", "exp", "log", "log1p", "abs_", "maximum", "minimum", "ifelse", "rk4", "trace", "emit_user_model", "traced_dynamics"]


class TraceError(TypeError):
    pass


class _Graph:
    def __init__(self):
        self.nodes = []          # (op, args) ; args are node ids or python floats for "const"
        self.cache = {}

    def add(self, op, *args):
        key = (op,) + args
        i = self.cache.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self.cache[key] = i
        return i


def _bits(v):
    return struct.unpack("<Q", struct.pack("<d", float(v)))[0]


class Tr:
    """A traced fp64 value: a node of the expression DAG.  Arithmetic with Python / numpy numbers bakes them in as constants."""
    __slots__ = ("g", "i")
    __array_priority__ = 1000.0        # numpy scalars defer to our reflected operators

    def __init__(self, g, i):
        self.g, self.i = g, i

    # --- construction helpers ---
    def _c(self, v):
        if isinstance(v, Tr):
            if v.g is not self.g:
                raise TraceError("values of two different traces were mixed")
            return v.i
        if isinstance(v, (bool, np.bool_)):
            raise TraceError("a Python bool met a traced value: use ifelse(cond, a, b)")
        try:
            f = float(v)
        except Exception:
            raise TraceError("cannot use %r in a traced expression" % (v,))
        return self.g.add("const", _bits(f))      # keyed by the bit pattern: -0.0 and 0.0 stay apart, every bit is kept

    def _bin(self, op, a, b):
        return Tr(self.g, self.g.add(op, a, b))

    def __add__(self, o): return self._bin("add", self.i, self._c(o))
    def __radd__(self, o): return self._bin("add", self._c(o), self.i)
    def __sub__(self, o): return self._bin("sub", self.i, self._c(o))
    def __rsub__(self, o): return self._bin("sub", self._c(o), self.i)
    def __mul__(self, o): return self._bin("mul", self.i, self._c(o))
    def __rmul__(self, o): return self._bin("mul", self._c(o), self.i)
    def __truediv__(self, o): return self._bin("div", self.i, self._c(o))
    def __rtruediv__(self, o): return self._bin("div", self._c(o), self.i)
    def __neg__(self): return Tr(self.g, self.g.add("neg", self.i))
    def __pos__(self): return self
    def __abs__(self): return Tr(self.g, self.g.add("abs", self.i))

    def __pow__(self, n):
        if isinstance(n, (int, np.integer)) and 1 <= int(n) <= 16:       # x^n as Julia's Base.power_by_squaring performs it (x^2 = x*x, x^3 = x*x*x)
            n = int(n)
            if n == 3:
                return (self * self) * self
            r, b = None, self
            while n:
                if n & 1:
                    r = b if r is None else r * b
                n >>= 1
                if n:
                    b = b * b
            return r
        if isinstance(n, float) and n == 0.5:
            return self.sqrt()
        raise TraceError("only integer powers 1..16 and 0.5 are traced (write exp(n * log(x)) for a general power)")

    # comparisons give a condition node, usable only in ifelse
    def _cmp(self, op, o): return Cond(self.g, self.g.add(op, self.i, self._c(o)))
    def __lt__(self, o): return self._cmp("lt", o)
    def __le__(self, o): return self._cmp("le", o)
    def __gt__(self, o): return self._cmp("gt", o)
    def __ge__(self, o): return self._cmp("ge", o)
    __hash__ = None

    def __eq__(self, o): return self._cmp("eq", o)
    def __ne__(self, o): return self._cmp("ne", o)

    def __bool__(self):
        raise TraceError("a traced value was used as a Python truth value (if / and / or / max / min): write ifelse(cond, a, b), maximum, minimum")

    def __float__(self):
        raise TraceError("a traced value was converted to float: the function leaves the traced operations (a library call?)")

    # numpy's ufuncs on object arrays / scalars call these methods: np.sqrt(x[0]), np.exp(...), np.abs(...)
    def sqrt(self): return Tr(self.g, self.g.add("sqrt", self.i))
    def exp(self): return Tr(self.g, self.g.add("exp", self.i))
    def log(self): return Tr(self.g, self.g.add("log", self.i))
    def log1p(self): return Tr(self.g, self.g.add("log1p", self.i))


class Cond:
    __slots__ = ("g", "i")

    def __init__(self, g, i):
        self.g, self.i = g, i

    def __bool__(self):
        raise TraceError("a comparison of traced values was used as a Python truth value: write ifelse(cond, a, b)")

    def __and__(self, o): return Cond(self.g, self.g.add("and", self.i, _cond(self, o)))
    def __or__(self, o): return Cond(self.g, self.g.add("or", self.i, _cond(self, o)))
    def __invert__(self): return Cond(self.g, self.g.add("not", self.i))


def _cond(c, o):
    if isinstance(o, Cond):
        return o.i
    raise TraceError("conditions combine with conditions only")


def _any_tr(*vals):
    for v in vals:
        if isinstance(v, (Tr, Cond)):
            return v
    return None


def _lift(ref, v):
    return v if isinstance(v, Tr) else Tr(ref.g, Tr(ref.g, 0)._c(v))


# ---- the functions a model may call; on plain numbers they are the host's (so the same callable also runs on the host, for simulate) ----
def sqrt(x): return x.sqrt() if isinstance(x, Tr) else math.sqrt(x)
def exp(x): return x.exp() if isinstance(x, Tr) else math.exp(x)
def log(x): return x.log() if isinstance(x, Tr) else math.log(x)
def log1p(x): return x.log1p() if isinstance(x, Tr) else math.log1p(x)
def abs_(x): return abs(x)


def maximum(a, b):
    """max(a, b) as Julia computes it for floats without NaN: a > b ? a : b is NOT it — Julia's max(x, y) = ifelse(x > y, x, y) up to signed
    zeros / NaN; the engine's built-in models use `v > 0 ? v : 0`, which is what this emits"""
    r = _any_tr(a, b)
    if r is None:
        return a if a > b else b
    a, b = _lift(r, a), _lift(r, b)
    return ifelse(a > b, a, b)


def minimum(a, b):
    r = _any_tr(a, b)
    if r is None:
        return a if a < b else b
    a, b = _lift(r, a), _lift(r, b)
    return ifelse(a < b, a, b)


def ifelse(c, a, b):
    """c ? a : b.  c: a comparison of traced values (or a plain bool, decided now)"""
    if isinstance(c, Cond):
        ref = Tr(c.g, 0)
        a, b = _lift(ref, a), _lift(ref, b)
        return Tr(c.g, c.g.add("sel", c.i, a.i, b.i))
    if isinstance(c, Tr):
        raise TraceError("ifelse needs a comparison, not a value")
    return a if c else b


def rk4(f, Ts, supersample=1):
    """discrete-time map of dx/dt = f(x, u, p, t) — the reference's rk4(f, Ts; supersample) (src/utils.jl:220-237), operation for operation:
    works on plain numbers and on traced ones (the loop is unrolled into the trace)"""
    ss = int(supersample)
    if ss < 1:
        raise ValueError("supersample must be >= 1")
    h = Ts / ss
    h2, h6 = h / 2.0, h / 6.0

    def step(x, u, p, t):
        x = list(x)
        n = len(x)
        for _ in range(ss):
            f1 = f(x, u, p, t)
            f2 = f([x[i] + h2 * f1[i] for i in range(n)], u, p, t + h2)
            f3 = f([x[i] + h2 * f2[i] for i in range(n)], u, p, t + h2)
            f4 = f([x[i] + h * f3[i] for i in range(n)], u, p, t + h)
            x = [x[i] + h6 * (((f1[i] + 2.0 * f2[i]) + 2.0 * f3[i]) + f4[i]) for i in range(n)]
            t = t + h
        return x
    return step


# ---- tracing and emission ---------------------------------------------------------------------------------------------------------
_INPUT_OPS = ("x", "u", "t", "y")


def trace(fn, nx, nu, p=None, ny=0, with_y=False, n_out=None, what="dynamics"):
    """run fn on tracer numbers; returns (graph, output node ids).  fn(x, u, p, t) or fn(x, u, y, p, t) (with_y)"""
    g = _Graph()
    x = [Tr(g, g.add("x", d)) for d in range(nx)]
    u = [Tr(g, g.add("u", d)) for d in range(nu)]
    t = Tr(g, g.add("t"))
    if with_y:
        y = [Tr(g, g.add("y", d)) for d in range(ny)]
        out = fn(x, u, y, p, t)
    else:
        out = fn(x, u, p, t)
    if isinstance(out, (Tr, int, float, np.floating)):
        out = [out]
    out = list(np.asarray(out, dtype=object).ravel()) if not isinstance(out, (list, tuple)) else list(out)
    if n_out is not None and len(out) != n_out:
        raise TraceError("%s returned %d values, expected %d" % (what, len(out), n_out))
    ref = Tr(g, 0)
    ids = [_lift(ref, v).i for v in out]
    return g, ids


_FMT = {"add": "({0} + {1})", "sub": "({0} - {1})", "mul": "({0} * {1})", "div": "({0} / {1})", "neg": "(-{0})", "abs": "llpf_fabs({0})",
        "sqrt": "llpf_sqrt({0})", "exp": "llpf_exp({0})", "log": "llpf_log({0})", "log1p": "llpf_log1p_nonneg({0})",
        "lt": "({0} < {1})", "le": "({0} <= {1})", "gt": "({0} > {1})", "ge": "({0} >= {1})", "eq": "({0} == {1})", "ne": "({0} != {1})",
        "and": "({0} && {1})", "or": "({0} || {1})", "not": "(!{0})", "sel": "({0} ? {1} : {2})"}
_BOOL = {"lt", "le", "gt", "ge", "eq", "ne", "and", "or", "not"}


def _const_literal(bits):
    v = struct.unpack("<d", struct.pack("<Q", bits))[0]
    if v != v:
        return "llpf_u2d(0x%016xULL)" % bits
    if math.isinf(v):
        return "(-LLPF_INF)" if v < 0 else "LLPF_INF"
    return "llpf_u2d(0x%016xULL) /* %r */" % (bits, v)       # every bit, whatever the printer of the day does with decimals


def _emit_body(g, outs, out_stmt, uses, indent="        ", tname="t_"):
    """statements computing the nodes reachable from outs (in creation order = the callable's own order), then out_stmt(k, name)"""
    need = set()
    stack = list(outs)
    while stack:
        i = stack.pop()
        if i in need:
            continue
        need.add(i)
        op = g.nodes[i][0]
        if op in _INPUT_OPS or op == "const":
            continue
        stack.extend(a for a in g.nodes[i][1:])
    lines = []
    name = {}
    for i in sorted(need):
        node = g.nodes[i]
        op = node[0]
        if op == "x":
            name[i] = "x[%d]" % node[1]
        elif op == "u":
            name[i] = "u_[%d]" % node[1]; uses.add("u")
        elif op == "t":
            name[i] = tname; uses.add("t")
        elif op == "y":
            name[i] = "y[%d]" % node[1]; uses.add("y")
        elif op == "const":
            name[i] = "c%d" % i
            lines.append("%sconst double c%d = %s;" % (indent, i, _const_literal(node[1])))
        else:
            name[i] = "v%d" % i
            lines.append("%sconst %s v%d = %s;" % (indent, "bool" if op in _BOOL else "double", i, _FMT[op].format(*[name[a] for a in node[1:]])))
    for k, i in enumerate(outs):
        lines.append(indent + out_stmt(k, name[i]))
    return "\n".join(lines)


def emit_user_model(nx, nu, ny, dynamics, measurement=None, loglik=None, loglik_bound=None, p=None):
    """the `struct UserModel` snippet (include/llpf.h: llpf_model_compile) of traced callables.
    dynamics(x, u, p, t) -> nx values; measurement(x, u, p, t) -> ny values (may use neither u nor t: the engine's Model concept hands
    neither to the measurement); loglik(x, u, y, p, t) -> one value (the AdvancedParticleFilter's measurement_likelihood; may use t, not u)
    with loglik_bound: an upper bound of it (a number), or None (every step then normalises against the true maximum)."""
    uses = set()
    gd, od = trace(dynamics, nx, nu, p, n_out=nx, what="dynamics")
    body_d = _emit_body(gd, od, lambda k, n: "out[%d] = %s;" % (k, n), uses)
    parts = ["    DEV void dynamics(const double* x, double* out) const {\n%s\n    }" % body_d]
    if measurement is not None:
        um = set()
        gm, om = trace(measurement, nx, nu, p, n_out=ny, what="measurement")
        body_m = _emit_body(gm, om, lambda k, n: "out[%d] = %s;" % (k, n), um)
        if um & {"u", "t"}:
            raise TraceError("a traced measurement may depend on the state only (the engine's Model::measurement receives neither u nor t): "
                             "put the dependence into measurement_likelihood")
        parts.append("    DEV void measurement(const double* x, double* out) const {\n%s\n    }" % body_m)
    else:
        parts.append("    DEV void measurement(const double* x, double* out) const { for (int k = 0; k < %d; ++k) out[k] = x[k]; }" % ny)
    if loglik is not None:
        ul = set()
        gl, ol = trace(loglik, nx, nu, p, ny=ny, with_y=True, n_out=1, what="measurement_likelihood")
        body_l = _emit_body(gl, ol, lambda k, n: "return %s;" % n, ul, tname="t")
        if "u" in ul:
            raise TraceError("a traced measurement_likelihood may not depend on u (the weighting of step k + 1 sees u_k on the device)")
        parts.append("    DEV double loglik(const double* x, const double* y, double t) const {\n%s\n    }" % body_l)
        if loglik_bound is not None:
            parts.append("    DEV double loglik_bound() const { return %s; }" % _const_literal(_bits(loglik_bound)))
    src = ("struct UserModel {\n    static constexpr bool RB = false;\n    double u_[%d];\n    double t_;\n"
           "    DEV void prepare(const ModelD* m, const double* u, double t) {\n"
           "        for (int j = 0; j < %d; ++j) u_[j] = (u != nullptr) ? u[j] : 0.0;\n        t_ = t;\n    }\n%s\n};\n"
           % (max(nu, 1), nu, "\n".join(parts)))
    return src


def traced_dynamics(f, nx, nu, p=None, measurement=None, ny=None, measurement_likelihood=None, loglik_bound=None, Ts=1.0):
    """UserDynamics (+ its UserMeasurement / UserLikelihood, as attributes `measurement_model` / `likelihood_model`) from ordinary Python
    callables with the reference's signatures; the callables themselves remain the host versions (simulate)."""
    from . import api
    if ny is None:
        raise ValueError("ny (the number of outputs) is needed")
    src = emit_user_model(nx, nu, ny, f, measurement, measurement_likelihood, loglik_bound, p)
    dyn = api.UserDynamics(src, nx, nu, ny, host=lambda x, u, pp, t: np.asarray(f(list(np.atleast_1d(x)), list(np.atleast_1d(u)) if u is not None else [], p, t), dtype=np.float64))
    dyn.measurement_model = api.UserMeasurement(host=None if measurement is None else
                                                (lambda x, u, pp, t: np.asarray(measurement(list(np.atleast_1d(x)), [], p, t), dtype=np.float64)))
    dyn.likelihood_model = None if measurement_likelihood is None else api.UserLikelihood(
        host=lambda x, u, y, pp, t: float(measurement_likelihood(list(np.atleast_1d(x)), [], list(np.atleast_1d(y)), p, t)))
    dyn.traced_source = src
    return dyn
