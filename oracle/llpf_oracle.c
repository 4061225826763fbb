/* llpf_oracle.c — CPU restatement of the LowLevelParticleFilters.jl particle-filter hot path.
 * TEST INFRASTRUCTURE ONLY (see llpf_oracle.h for the rules and for how parity is pinned).
 *
 * Every function cites the reference lines it follows (paths relative to the reference root,
 * v3.31.1).  Indices are 0-based here; the reference is 1-based.
 *
 * Compile with -ffp-contract=off: no multiply-add may be fused except the explicit llpf_fma()
 * calls inside llpf_detmath.h.
 */
#include "llpf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../lowlevelparticlefilters.jl_amd/csrc/shared/llpf_detmath.h"
#include "../lowlevelparticlefilters.jl_amd/csrc/shared/llpf_fixed.h"
#include "../lowlevelparticlefilters.jl_amd/csrc/shared/llpf_philox.h"

#define MAXD LLPF_MAX_DIM
#include "../lowlevelparticlefilters.jl_amd/csrc/shared/llpf_rbkf.h"
#include "../lowlevelparticlefilters.jl_amd/csrc/shared/llpf_rbfull.h"     /* llpf_rbf_*: device order only; the reference order is rbfr_* below */

/* Optional OpenMP over the per-particle loops (weighting, propagation, noise, elementwise exp): an upper bound for
 * what the reference could reach with its `threads=true` option (src/PFtypes.jl:226-259 @threads :static); the
 * reductions and the resampling scan stay serial, so results do not depend on the thread count.  Default 1 thread =
 * the reference's ParticleFilter path. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int  orc_get_threads(void) { return g_threads; }
#define ORC_PAR _Pragma("omp parallel for schedule(static) num_threads(g_threads)")

/* ------------------------------------------------------------------------------------------
 * Gaussian densities — src/utils.jl:241-270 (SimpleMvNormal), PDMats shims src/utils.jl:110-113,
 * ext/LowLevelParticleFiltersDistributionsExt.jl:16,80 (Distributions.MvNormal path)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int dim, kind;
    double mu[MAXD];
    double L[MAXD * MAXD];      /* lower Cholesky factor (FULL) */
    double scal, sqrtscal;      /* ScalMat value and its sqrt   */
    double invscal, invLd[MAXD];/* device order: reciprocals, the divisions of invquad become multiplications */
    int dev;                    /* ORC_ORDER_DEVICE */
    double diag[MAXD], invdiag[MAXD], sqrtdiag[MAXD];
    double c0;                  /* mvnormal_c0: -(k log2pi + logdet Sigma)/2, src/utils.jl:254-257 */
} gaussd;

/* cholesky(Sigma).L — Cholesky–Banachiewicz, row by row */
static int chol_lower(const double* S, int n, double* L) {
    memset(L, 0, sizeof(double) * MAXD * MAXD);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double acc = S[i * n + j];
            for (int k = 0; k < j; ++k) acc = acc - L[i * MAXD + k] * L[j * MAXD + k];
            if (i == j) {
                if (!(acc > 0.0)) return -1;
                L[i * MAXD + i] = sqrt(acc);
            } else {
                L[i * MAXD + j] = acc / L[j * MAXD + j];
            }
        }
    }
    return 0;
}

static int gauss_prepare(const llpf_gaussian* g, gaussd* d, int order) {
    memset(d, 0, sizeof(*d));
    d->dim = g->dim;
    d->kind = g->kind;
    d->dev = (order == ORC_ORDER_DEVICE);
    int n = g->dim;
    if (n < 1 || n > MAXD) return -1;
    for (int i = 0; i < n; ++i) d->mu[i] = g->mu[i];
    double logdet = 0.0;
    /* the constant uses libm log in reference order and the shared deterministic log in device
     * order (so that the engine's host code and this oracle agree to the bit on any libm) */
#define LOGF(x) (order == ORC_ORDER_DEVICE ? llpf_log(x) : log(x))
    if (g->kind == LLPF_COV_SCAL) {
        d->scal = g->cov[0];
        if (!(d->scal > 0.0)) return -1;
        d->sqrtscal = sqrt(d->scal);
        d->invscal = 1.0 / d->scal;
        logdet = (double)n * LOGF(d->scal);                 /* PDMats: logdet(ScalMat) = dim*log(value) */
        for (int i = 0; i < n; ++i) d->L[i * MAXD + i] = d->sqrtscal;
    } else if (g->kind == LLPF_COV_DIAG) {
        for (int i = 0; i < n; ++i) {
            d->diag[i] = g->cov[i];
            if (!(d->diag[i] > 0.0)) return -1;
            d->invdiag[i] = 1.0 / d->diag[i];                 /* 1 ./ a.diag, src/utils.jl:113 */
            d->sqrtdiag[i] = sqrt(d->diag[i]);
            d->L[i * MAXD + i] = d->sqrtdiag[i];
            logdet = (i == 0) ? LOGF(d->diag[i]) : logdet + LOGF(d->diag[i]);  /* sum(log, diag) */
        }
    } else if (g->kind == LLPF_COV_FULL) {
        if (chol_lower(g->cov, n, d->L) != 0) return -1;
        for (int i = 0; i < n; ++i) d->invLd[i] = 1.0 / d->L[i * MAXD + i];
        double dd = 0.0;
        for (int i = 0; i < n; ++i) dd = (i == 0) ? LOGF(d->L[i * MAXD + i]) : dd + LOGF(d->L[i * MAXD + i]);
        logdet = dd + dd;                                     /* logdet(::Cholesky) */
    } else {
        return -1;
    }
    const double log2pi = LOGF(2.0 * 3.141592653589793);     /* const log2π = log(2π), src/utils.jl:253 */
#undef LOGF
    d->c0 = -((double)n * log2pi + logdet) / 2.0;
    return 0;
}

/* extended_logpdf(d, x) = c0 - invquad(Sigma, x - mu)/2 — src/utils.jl:252; invquad per
 * src/utils.jl:110-113: ScalMat dot(x,x)/value; PDMat dot(x, chol \ x); PDiagMat wsumsq(1 ./ diag, x) */
static double gauss_logpdf(const gaussd* g, const double* x) {
    int n = g->dim;
    double d[MAXD], q;
    for (int i = 0; i < n; ++i) d[i] = x[i] - g->mu[i];
    if (g->kind == LLPF_COV_SCAL) {
        double dot = d[0] * d[0];
        for (int i = 1; i < n; ++i) dot = dot + d[i] * d[i];
        q = g->dev ? dot * g->invscal : dot / g->scal;
    } else if (g->kind == LLPF_COV_DIAG) {
        double s = (d[0] * d[0]) * g->invdiag[0];
        for (int i = 1; i < n; ++i) s = s + (d[i] * d[i]) * g->invdiag[i];
        q = s;
    } else {
        double z[MAXD], z2[MAXD];
        for (int i = 0; i < n; ++i) {                         /* L \ d */
            double acc = d[i];
            for (int j = 0; j < i; ++j) acc = acc - g->L[i * MAXD + j] * z[j];
            z[i] = g->dev ? acc * g->invLd[i] : acc / g->L[i * MAXD + i];
        }
        for (int i = n - 1; i >= 0; --i) {                    /* L' \ z */
            double acc = z[i];
            for (int j = i + 1; j < n; ++j) acc = acc - g->L[j * MAXD + i] * z2[j];
            z2[i] = g->dev ? acc * g->invLd[i] : acc / g->L[i * MAXD + i];
        }
        double dot = d[0] * z2[0];
        for (int i = 1; i < n; ++i) dot = dot + d[i] * z2[i];
        q = dot;
    }
    return g->c0 - q / 2.0;
}

/* rand!(rng, d, out): unwhiten(xi) .+ mu — src/utils.jl:264-268, Distributions _rand!(MvNormal) */
static void gauss_sample(const gaussd* g, const double* xi, double* out) {
    int n = g->dim;
    for (int i = 0; i < n; ++i) {
        double v;
        if (g->kind == LLPF_COV_SCAL) v = g->sqrtscal * xi[i];
        else if (g->kind == LLPF_COV_DIAG) v = g->sqrtdiag[i] * xi[i];
        else {
            v = g->L[i * MAXD + 0] * xi[0];
            for (int j = 1; j <= i; ++j) v = v + g->L[i * MAXD + j] * xi[j];
        }
        out[i] = v + g->mu[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * Models — SURVEY.md Appendix A.5
 * ---------------------------------------------------------------------------------------- */
/* quad-tank right-hand side — examples/example_quadtank.jl:8-27 */
static void quadtank_rhs(const double* qt, const double* h, const double* u, double t, double* xd) {
    double k1 = qt[LLPF_QT_K1], k2 = qt[LLPF_QT_K2], g = qt[LLPF_QT_G];
    double A1 = qt[LLPF_QT_A1], A2 = qt[LLPF_QT_A2], A3 = qt[LLPF_QT_A3], A4 = qt[LLPF_QT_A4];
    double a1 = qt[LLPF_QT_a1], a2 = qt[LLPF_QT_a2], a3 = qt[LLPF_QT_a3], a4 = qt[LLPF_QT_a4];
    double g1 = qt[LLPF_QT_GAMMA1], g2 = qt[LLPF_QT_GAMMA2], eps = qt[LLPF_QT_EPS];
    if (t > qt[LLPF_QT_TSWITCH]) a1 = a1 * qt[LLPF_QT_A1FACTOR];       /* :15-17 */
    double tg = 2.0 * g;
    double s[4];
    for (int i = 0; i < 4; ++i) {                                       /* ssqrt, :19 */
        double v = tg * h[i];
        s[i] = sqrt((v > 0.0 ? v : 0.0) + eps);                   /* the engine: llpf_sqrt_pos, same correctly-rounded value */
    }
    xd[0] = ((-a1) / A1) * s[0] + (a3 / A1) * s[2] + ((g1 * k1) / A1) * u[0];
    xd[1] = ((-a2) / A2) * s[1] + (a4 / A2) * s[3] + ((g2 * k2) / A2) * u[1];
    xd[2] = ((-a3) / A3) * s[2] + (((1.0 - g2) * k2) / A3) * u[1];
    xd[3] = ((-a4) / A4) * s[3] + (((1.0 - g1) * k1) / A4) * u[0];
}

/* rk4(f, Ts; supersample) — src/utils.jl:220-237 */
static void quadtank_rk4(const llpf_model* m, const double* x0, const double* u, double t, double* out) {
    int ss = m->supersample < 1 ? 1 : m->supersample;
    double Ts = m->Ts / (double)ss;
    double x[4], f1[4], f2[4], f3[4], f4[4], xt[4];
    for (int i = 0; i < 4; ++i) x[i] = x0[i];
    for (int it = 0; it < ss; ++it) {
        quadtank_rhs(m->qt, x, u, t, f1);
        for (int i = 0; i < 4; ++i) xt[i] = x[i] + (Ts / 2.0) * f1[i];
        quadtank_rhs(m->qt, xt, u, t + Ts / 2.0, f2);
        for (int i = 0; i < 4; ++i) xt[i] = x[i] + (Ts / 2.0) * f2[i];
        quadtank_rhs(m->qt, xt, u, t + Ts / 2.0, f3);
        for (int i = 0; i < 4; ++i) xt[i] = x[i] + Ts * f3[i];
        quadtank_rhs(m->qt, xt, u, t + Ts, f4);
        for (int i = 0; i < 4; ++i) x[i] = x[i] + (Ts / 6.0) * (((f1[i] + 2.0 * f2[i]) + 2.0 * f3[i]) + f4[i]);
        t = t + Ts;
    }
    for (int i = 0; i < 4; ++i) out[i] = x[i];
}

/* the rk4 known-answer case of test/runtests.jl:182-188: xdot = -1 */
void orc_rk4_scalar_decay(double x0, double Ts0, int supersample, double* out) {
    double Ts = Ts0 / (double)supersample, x = x0, t = 0.0;
    for (int it = 0; it < supersample; ++it) {
        double f1 = -1.0, f2 = -1.0, f3 = -1.0, f4 = -1.0;
        (void)t;
        x = x + (Ts / 6.0) * (((f1 + 2.0 * f2) + 2.0 * f3) + f4);
        t = t + Ts;
    }
    *out = x;
}

/* LLPF_MODEL_RB_BILINEAR: f_n / g are the linear descriptors (fn_kind 0) or the quad-tank (fn_kind 1) over xn */
static int model_is_linear(const llpf_model* m) {
    return m->model_id == LLPF_MODEL_LINEAR_GAUSSIAN || (m->model_id == LLPF_MODEL_RB_BILINEAR && m->rb.fn_kind == 0);
}

/* dynamics(x,u,p,t) without noise */
void orc_dynamics(const llpf_model* m, const double* x, const double* u, double t, double* out) {
    if (model_is_linear(m)) {          /* A*x .+ B*u, examples/example_lineargaussian.jl:28 */
        for (int r = 0; r < m->nx; ++r) {
            double ax = m->A[r * m->nx + 0] * x[0];
            for (int c = 1; c < m->nx; ++c) ax = ax + m->A[r * m->nx + c] * x[c];
            if (m->nu > 0) {
                double bu = m->B[r * m->nu + 0] * u[0];
                for (int c = 1; c < m->nu; ++c) bu = bu + m->B[r * m->nu + c] * u[c];
                ax = ax + bu;
            }
            out[r] = ax;
        }
    } else {
        quadtank_rk4(m, x, u, t, out);
    }
}

/* measurement(x,u,p,t) */
void orc_measurement(const llpf_model* m, const double* x, const double* u, double t, double* out) {
    (void)u; (void)t;
    if (model_is_linear(m)) {                                 /* C*x, examples/example_lineargaussian.jl:29 */
        for (int r = 0; r < m->ny; ++r) {
            double cx = m->C[r * m->nx + 0] * x[0];
            for (int c = 1; c < m->nx; ++c) cx = cx + m->C[r * m->nx + c] * x[c];
            out[r] = cx;
        }
    } else {                                                  /* SA[x[1], x[2]], examples/example_quadtank.jl:33 */
        out[0] = x[0];
        out[1] = x[1];
    }
}

/* ------------------------------------------------------------------------------------------
 * Julia Base semantics used by the path
 * ---------------------------------------------------------------------------------------- */
/* Base.isless for Float64: NaN is greatest, -0.0 < 0.0 */
static int jl_isless(double a, double b) {
    if (a != a) return 0;
    if (b != b) return 1;
    if (a == 0.0 && b == 0.0) return signbit(a) && !signbit(b);
    return a < b;
}
/* findmax(w): first maximal element under isless */
static double jl_findmax(const double* w, int64_t n, int64_t* idx) {
    double fm = w[0];
    int64_t im = 0;
    for (int64_t i = 1; i < n; ++i)
        if (jl_isless(fm, w[i])) { fm = w[i]; im = i; }
    *idx = im;
    return fm;
}
/* Base.mapreduce_impl pairwise summation, block size 1024.  NOTE: Julia's base case is an @simd
 * loop whose reassociation is CPU-dependent; it is restated here as a serial left-to-right loop. */
static double pairwise(const double* a, int64_t lo, int64_t hi, int sq) {
    if (lo == hi) return sq ? a[lo] * a[lo] : a[lo];
    if (hi - lo < 1024) {
        double v = sq ? (a[lo] * a[lo] + a[lo + 1] * a[lo + 1]) : (a[lo] + a[lo + 1]);
        for (int64_t i = lo + 2; i <= hi; ++i) v = sq ? v + a[i] * a[i] : v + a[i];
        return v;
    }
    int64_t mid = lo + ((hi - lo) >> 1);
    double v1 = pairwise(a, lo, mid, sq);
    double v2 = pairwise(a, mid + 1, hi, sq);
    return v1 + v2;
}
double orc_pairwise_sum(const double* a, int64_t n) { return n ? pairwise(a, 0, n - 1, 0) : 0.0; }

/* ------------------------------------------------------------------------------------------
 * logsumexp! / expnormalize! / effective_particles — src/utils.jl:3-79, src/resample.jl:1-2
 * ---------------------------------------------------------------------------------------- */
/* sum_all_but(we, i) — src/utils.jl:66-71 */
static double sum_all_but(double* we, int64_t n, int64_t i) {
    we[i] -= 1.0;
    double s = orc_pairwise_sum(we, n);
    we[i] += 1.0;
    return s;
}

/* device-order core: from raw log-weights w (unchanged) produce e_i = exp(w_i - m) and the exact
 * fixed-point sums; returns s = sum_{i != argmax} e_i rounded once */
/* weights are held as raw values plus a pending normalisation (a, b): w_norm = (w - a) - b, we = exp(w - a) * inv.
 *   exact form: a = max w, stot = fl(s + 1) with s = sum_{i != argmax} exp(w_i - a), b = log1p(s)   (utils.jl:18-27)
 *   fast  form: a = analytic upper bound of max w, stot = sum_i exp(w_i - a), b = log(stot)          (see dev_norm_bound) */
typedef struct { double m, s, l, inv, e2, stot, mtrue; uint64_t totQ; int K; int fast; } devnorm;

static void dev_expsum(const double* w, double* e, int64_t n, devnorm* o) {
    double m = w[0];
    for (int64_t i = 1; i < n; ++i) m = llpf_fmax(m, w[i]);
    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0;
    int K = llpf_qbits(n);
    for (int64_t i = 0; i < n; ++i) {
        double ei = llpf_exp_le0(w[i] - m);
        e[i] = ei;
        S = llpf_u128_add(S, llpf_fix96_unit(ei));
        E2 = llpf_u128_add(E2, llpf_fix96_unit(ei * ei));
        Q += llpf_q64_unit(ei, K);
    }
    o->m = m;
    o->K = K;
    o->totQ = Q;
    if (S.hi >= ((uint64_t)1 << 32)) o->s = llpf_fix96_to_double(llpf_fix96_minus_one(S));
    else o->s = llpf_u2d(0x7ff8000000000000ULL);              /* max is -Inf/NaN: degenerate */
    o->l = llpf_log1p_nonneg(o->s);
    o->stot = o->s + 1.0;
    o->inv = 1.0 / o->stot;
    o->e2 = llpf_fix96_to_double(E2);
    o->mtrue = m;
    o->fast = 0;
}

/* Device order, weighting path: normalisation against an ANALYTIC bound instead of the maximum, so that the GPU
 * needs no separate max pass between the weighting and the sums (one launch per timestep).
 * For the built-in Gaussian measurement densities logpdf <= c0, hence every new weight w_i = w_prev_i + logpdf_i
 * satisfies w_i <= off := max(w_prev) + c0 (also after rounding: fl is monotone), and e_i = exp(w_i - off) <= 1.
 * If the exact fixed-point sum S = sum e_i is at least 2^-10 (the particle cloud is not more than ~3.7 sigma from
 * the measurement in likelihood terms) the offset `off` replaces the maximum everywhere (logsumexp is invariant to the
 * offset; only roundings differ, by ~1e-16): ll = off + log(S), w_norm = (w - off) - log(S), we = e / S,
 * ESS = S^2 / sum(e^2), resampling quanta floor(e 2^K).  Otherwise the exact-max form above is used for this step. */
static void dev_norm_bound(const double* w, double* e, int64_t n, double off, devnorm* o) {
    double m = w[0];
    for (int64_t i = 1; i < n; ++i) m = llpf_fmax(m, w[i]);
    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0;
    int K = llpf_qbits(n);
    int bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        double ei = llpf_exp_le0(w[i] - off);
        if (ei != ei) bad = 1;
        e[i] = ei;
        S = llpf_u128_add(S, llpf_fix96_unit(ei));
        E2 = llpf_u128_add(E2, llpf_fix96_unit(ei * ei));
        Q += llpf_q64_unit(ei, K);
    }
    if (bad || S.hi < ((uint64_t)1 << 22)) {         /* S < 2^-10 (or NaN weights): exact-max form for this step */
        dev_expsum(w, e, n, o);
        return;
    }
    o->m = off;
    o->K = K;
    o->totQ = Q;
    o->stot = llpf_fix96_to_double(S);
    o->s = o->stot - 1.0;
    o->l = llpf_log(o->stot);
    o->inv = 1.0 / o->stot;
    o->e2 = llpf_fix96_to_double(E2);
    o->mtrue = m;
    o->fast = 1;
}

/* ll = logsumexp!(w, we [, maxw]) — src/utils.jl:18-27 */
double orc_logsumexp(double* w, double* we, int64_t n, int order, double* maxw) {
    if (order == ORC_ORDER_DEVICE) {
        devnorm d;
        dev_expsum(w, we, n, &d);
        for (int64_t i = 0; i < n; ++i) {
            we[i] = we[i] * d.inv;
            w[i] = (w[i] - d.m) - d.l;
        }
        if (maxw) *maxw = d.m;
        return d.l + d.m;
    }
    int64_t maxind;
    double offset = jl_findmax(w, n, &maxind);                /* :19 */
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) w[i] -= offset;           /* :20 */
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) we[i] = exp(w[i]);        /* :21 exp_map!, :3-7 */
    double s = sum_all_but(we, n, maxind);                    /* :22 */
    double sc = 1.0 / (s + 1.0);
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) we[i] *= sc;              /* :23 */
    double l = log1p(s);
    ORC_PAR
    for (int64_t i = 0; i < n; ++i) w[i] -= l;                /* :24 */
    if (maxw) *maxw = offset;                                 /* :25 */
    return l + offset;                                        /* :26 */
}

/* expnormalize!(we, w) — src/utils.jl:48-55 */
void orc_expnormalize(double* we, double* w, int64_t n) {
    int64_t maxind;
    double offset = jl_findmax(w, n, &maxind);
    for (int64_t i = 0; i < n; ++i) w[i] -= offset;
    for (int64_t i = 0; i < n; ++i) we[i] = exp(w[i]);
    for (int64_t i = 0; i < n; ++i) w[i] += offset;
    double s = sum_all_but(we, n, maxind);
    double sc = 1.0 / (s + 1.0);
    for (int64_t i = 0; i < n; ++i) we[i] *= sc;
}
/* expnormalize!(w) — src/utils.jl:57-63 */
void orc_expnormalize_inplace(double* w, int64_t n) {
    int64_t maxind;
    double offset = jl_findmax(w, n, &maxind);
    for (int64_t i = 0; i < n; ++i) w[i] -= offset;
    for (int64_t i = 0; i < n; ++i) w[i] = exp(w[i]);
    double s = sum_all_but(w, n, maxind);
    double sc = 1.0 / (s + 1.0);
    for (int64_t i = 0; i < n; ++i) w[i] *= sc;
}

/* effective_particles(we) = 1/sum(abs2, we) — src/resample.jl:1-2 */
double orc_effective_particles(const double* we, int64_t n) {
    return 1.0 / pairwise(we, 0, n - 1, 1);
}

/* ------------------------------------------------------------------------------------------
 * Resampling — src/resample.jl:17-61
 * ---------------------------------------------------------------------------------------- */
typedef struct { int strategy; int64_t m; double r, step; const double* U; } thr_ctx;

/* threshold for 0-based output i.
 * systematic: s = r:(1/M):(bins[N]+r), s[i]  (:23-24); for Float64 arguments that are not all exact
 *   short rationals Julia builds StepRangeLen{Float64,TwicePrecision,TwicePrecision}(r, 1/M) whose
 *   getindex evaluates fl(r + fl(i0 * (1/M))) (Base twiceprecision.jl unsafe_getindex; step.lo = 0).
 * stratified: u = (i - 1 + rand()) / M * bins[N]  (:49), evaluated left to right. */
static double thr_at(const thr_ctx* c, int64_t i0, double binsN) {
    if (c->strategy == LLPF_RESAMPLE_SYSTEMATIC) return c->r + (double)i0 * c->step;
    return ((double)i0 + c->U[i0]) / (double)c->m * binsN;
}

/* resample(::Type{ResampleResidual}, we, j, bins, M) — src/resample.jl:63-117.  U[m] is the uniform of output m
 * (the reference draws rand() for m = num+1..M in order; only those entries are read).
 * Device order: the weights are the integer quanta q_i (total Q), so the copy counts floor(q_i M / Q) and the
 * residuals q_i M - c_i Q are exact integers; the residuals are kept to K bits (>> ceil(log2 N)) so that their
 * cumulative sum fits 63 bits, and the multinomial part searches bins = fl(fl(cum) * fl(1/fl(total))) like the
 * other strategies.  `q` (device order): the quanta, or NULL to derive them from `we` with llpf_q64_unit. */
static int resample_residual(const double* we, const uint64_t* q, int64_t n, int64_t m, const double* U,
                             int64_t* j, double* b, int order) {
    if (order == ORC_ORDER_REFERENCE) {
        double wsum = 0.0;
        for (int64_t i = 0; i < n; ++i) wsum += we[i];                     /* :66-69 */
        double inv_wsum = 1.0 / wsum;
        int64_t num = 0;
        for (int64_t i = 0; i < n; ++i) {                                  /* :75-84 */
            double nw = we[i] * inv_wsum * (double)m;
            int64_t cnt = (int64_t)floor(nw);
            b[i] = nw - (double)cnt;
            for (int64_t k = 0; k < cnt && num < m; ++k) j[num++] = i;
        }
        if (num == m) return 0;                                            /* :86-88 */
        double rsum = 0.0;
        for (int64_t i = 0; i < n; ++i) rsum += b[i];                      /* :90-93 */
        double inv_rsum = 1.0 / rsum;
        for (int64_t i = 0; i < n; ++i) b[i] *= inv_rsum;                  /* :95-98 */
        for (int64_t i = 1; i < n; ++i) b[i] += b[i - 1];                  /* :100-103 */
        for (int64_t mm = num; mm < m; ++mm) {                             /* :105-114 */
            double u = U[mm];
            for (int64_t i = 0; i < n; ++i)
                if (u < b[i]) { j[mm] = i; break; }
        }
        return 0;
    }
    int K = llpf_qbits(n);
    int L = 62 - K;                                                        /* ceil(log2 n) */
    uint64_t Q = 0;
    for (int64_t i = 0; i < n; ++i) Q += q ? q[i] : llpf_q64_unit(we[i], K);
    if (Q == 0) return -1;
    int64_t num = 0;
    uint64_t totr = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t qi = q ? q[i] : llpf_q64_unit(we[i], K), rem;
        uint64_t cnt = llpf_muldiv_floor(qi, (uint64_t)m, Q, &rem);
        totr += rem >> L;
        for (uint64_t k = 0; k < cnt && num < m; ++k) j[num++] = i;
    }
    if (num == m || totr == 0) { for (int64_t i = 0; i < n; ++i) b[i] = 0.0; return 0; }
    double Td = (double)totr, invTd = 1.0 / Td;
    uint64_t cum = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t qi = q ? q[i] : llpf_q64_unit(we[i], K), rem;
        llpf_muldiv_floor(qi, (uint64_t)m, Q, &rem);
        cum += rem >> L;
        b[i] = (double)cum * invTd;
    }
    for (int64_t mm = num; mm < m; ++mm) {
        double u = U[mm];
        /* first i with u < bins[i] (bins non-decreasing): the reference's linear search */
        int64_t lo = 0, hi = n;
        while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (u < b[mid]) hi = mid; else lo = mid + 1; }
        if (lo < n) j[mm] = lo;
    }
    return 0;
}

int orc_resample(int strategy, const double* we, int64_t n, int64_t m, const double* U,
                 int64_t* j, double* bins, int order) {
    if (strategy == LLPF_RESAMPLE_RESIDUAL) {
        double* bb = bins ? bins : (double*)malloc(sizeof(double) * (size_t)n);
        int rc = resample_residual(we, NULL, n, m, U, j, bb, order);
        if (!bins) free(bb);
        return rc;
    }
    thr_ctx c;
    c.strategy = strategy; c.m = m; c.U = U; c.step = 1.0 / (double)m; c.r = 0.0;
    double* b = bins ? bins : (double*)malloc(sizeof(double) * (size_t)n);
    if (order == ORC_ORDER_REFERENCE) {
        b[0] = we[0];                                          /* :19-22 strictly serial cumsum */
        for (int64_t i = 1; i < n; ++i) b[i] = b[i - 1] + we[i];
        double binsN = b[n - 1];
        if (strategy == LLPF_RESAMPLE_SYSTEMATIC) c.r = U[0] * binsN / (double)n;   /* :23 */
        int64_t bo = 0;                                        /* :25-34 / :52-58 two-pointer search */
        for (int64_t i = 0; i < m; ++i) {
            double si = thr_at(&c, i, binsN);
            for (int64_t k = bo; k < n; ++k) {
                if (si < b[k]) { j[i] = k; bo = k; break; }
            }
        }
    } else {
        /* device order: bins[b] = fl( fl(sum_{k<=b} Q_k) * fl(1 / fl(sum_k Q_k)) ), Q_k = floor(we_k 2^K): an
         * integer (associative) cumulative sum, so any parallel blocking gives the same bits, and
         * bins[N-1] = fl(T * fl(1/T)) is 1 or 1 - 2^-53.  j[i] = first b with thr_i < bins[b], exactly the
         * reference's search (the GPU evaluates it through counts c(v) = #{ i : thr_i < v }). */
        int K = llpf_qbits(n);
        uint64_t cum = 0, tot = 0;
        for (int64_t i = 0; i < n; ++i) tot += llpf_q64_unit(we[i], K);
        if (tot == 0) { if (!bins) free(b); return -1; }
        double Td = (double)tot;
        double invTd = 1.0 / Td;
        for (int64_t i = 0; i < n; ++i) {
            cum += llpf_q64_unit(we[i], K);
            b[i] = (double)cum * invTd;
        }
        double binsN = Td * invTd;
        if (strategy == LLPF_RESAMPLE_SYSTEMATIC) c.r = U[0] * binsN / (double)n;
        int64_t bo = 0;
        for (int64_t i = 0; i < m; ++i) {
            double si = thr_at(&c, i, binsN);
            for (int64_t k = bo; k < n; ++k) {
                if (si < b[k]) { j[i] = k; bo = k; break; }
            }
        }
    }
    if (!bins) free(b);
    return 0;
}

void orc_resample_uniforms(int strategy, int64_t m, uint64_t seed, uint32_t step, double* u) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (strategy == LLPF_RESAMPLE_SYSTEMATIC) u[0] = llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, k0, k1);
    else for (int64_t i = 0; i < m; ++i) u[i] = llpf_uniform_idx((uint32_t)i, step, LLPF_STREAM_STRATIFY, k0, k1);
}

/* ------------------------------------------------------------------------------------------
 * The filter — PFstate src/PFtypes.jl:8-19; ParticleFilter :21-36,65-75; AdvancedParticleFilter :162-210
 * ---------------------------------------------------------------------------------------- */
struct orc_filter {
    llpf_config cfg;
    int order;
    int64_t N;
    int nx, nu, ny;
    gaussd df, dg, d0;
    /* a measurement likelihood other than the Gaussian descriptor (orc_set_user_loglik): the counterpart, in this checker, of
     * the reference's measurement_likelihood(x,u,y,p,t) callable (src/PFtypes.jl:226-239) / of a non-Gaussian measurement
     * density (ext/LowLevelParticleFiltersDistributionsExt.jl:80) */
    int user_ll_kind;
    double user_par[4], user_c, user_bound;
    /* process noise / initial density of the model's own (orc_set_user_noise / orc_set_user_initial): the counterpart of the
     * reference's AdvancedParticleFilter dynamics(x, u, p, t, noise = true) (src/PFtypes.jl:242-259) / of a dynamics_density or
     * initial_density that is not Gaussian (rand!(rng, d, noise) :135, rand(rng, initial_density) src/filtering.jl:8) */
    int user_noise_kind, user_init_kind;
    double noise_par[4], init_par[2 * MAXD];
    double* uu_buf;             /* N x nx uniforms of the particle streams LLPF_STREAM_USER / LLPF_STREAM_USER_INIT */
    double *x, *xprev;          /* N*nx, AoS like Vector{SVector} */
    double *w, *we, *bins, *e;  /* e: exp(w_raw - m) of the last normalisation (device order) */
    int64_t* j;
    double maxw;
    int64_t t;                  /* state.t[] */
    uint32_t k0, k1, n_reset, n_predict;
    int last_resampled;
    int64_t resample_count;
    int degenerate;
    int64_t n_exact_steps;      /* device order: weightings whose bound test failed (exact-max form used) */
    /* device-order scalars of the last normalisation */
    devnorm dn;
    int dn_valid;
    double wmax;                /* max of the current (normalised / uniform / installed) log-weights: the bound's input */
    /* RBPF with constant matrices (src/rbpf.jl) */
    struct {
        int on, nn, nl, zeroC, zeroAn;
        double Fn[16], Bn[16], An[16], Al[16], Bl[16], Gn[16], Cl[16], R1l[16], R1nS[16], R2S[16];
        double R[16];                 /* the covariance shared by all particles (x[1].R, "singleR") */
        double kfx[4], kfR[16];       /* fields x, R of the inner KalmanFilter object: never reset, reused as they are when C == 0 (:276) */
        gaussd dS;                    /* N(0, S) of the last measurement update */
        double K[16], L[16];
    } rb;
    /* RBPF with state-dependent coupling: every particle has its own Kalman state (src/rbpf.jl:1-5) */
    struct {
        int on, nn, nl, np;
        llpf_rbf_par par;
        double *xl, *xlprev;          /* N x nl */
        double *R, *Rprev;            /* N x np: packed lower triangles */
    } rbf;
    /* AuxiliaryParticleFilter (src/filtering.jl:170-217) */
    double* lam;                /* lambda of the last aux predict! (the reference keeps it in `we`) */
    int aux_pending;            /* w holds lambda - log N of an aux predict!, not yet normalised */
    double aux_off;             /* device order: upper bound of those weights */
    double *xi_buf, *U_buf;
};

static void set_key(orc_filter* f, uint64_t seed) {
    f->k0 = (uint32_t)seed;
    f->k1 = (uint32_t)(seed >> 32);
    f->n_reset = 0;
    f->n_predict = 0;
}

static void fill_uniform_weights(orc_filter* f, double wval) {
    double wev = 1.0 / (double)f->N;
    for (int64_t i = 0; i < f->N; ++i) { f->w[i] = wval; f->we[i] = wev; }
    f->dn_valid = 0;
    f->wmax = wval;
}

/* x_i = rand(rng, initial_density) — src/filtering.jl:8, src/PFtypes.jl:66.  User kind 1: a uniform box, x_d = lo_d + (hi_d - lo_d) uu_d */
static void init_particles(orc_filter* f, const double* xi) {
    for (int64_t i = 0; i < f->N; ++i) {
        if (f->user_init_kind == 1) {
            for (int d = 0; d < f->nx; ++d)
                f->xprev[i * f->nx + d] = f->init_par[d] + (f->init_par[f->nx + d] - f->init_par[d]) * f->uu_buf[i * f->nx + d];
        } else {
            gauss_sample(&f->d0, xi + i * f->nx, f->xprev + i * f->nx);
        }
        for (int d = 0; d < f->nx; ++d) f->x[i * f->nx + d] = f->xprev[i * f->nx + d];
    }
}

static void gen_normals(orc_filter* f, uint32_t step, uint32_t stream, double* out) {
    ORC_PAR
    for (int64_t i = 0; i < f->N; ++i)
        llpf_normals((uint32_t)i, step, stream, f->k0, f->k1, f->nx, out + i * f->nx);
}
static void gen_uniforms(orc_filter* f, uint32_t step, uint32_t stream, double* out) {
    ORC_PAR
    for (int64_t i = 0; i < f->N; ++i)
        llpf_uniforms((uint32_t)i, step, stream, f->k0, f->k1, f->nx, out + i * f->nx);
}

/* the next state from the previous state x, its noise-free prediction fx, and the particle's draws.  Gaussian descriptor:
 * fx + rand(df) (src/PFtypes.jl:135).  User kinds (test-side counterparts of the device snippets in tests/user_models.py):
 *   1  multiplicative Gaussian, standard deviation par[0] + par[1] |x_d| of the state it leaves:   fx_d + (par[0] + par[1] |x_d|) xi_d
 *   2  Laplace of scale b = par[0] through the inverse CDF of one uniform per dimension:            fx_d +/- b (-log(1 - |2 uu_d - 1|))
 * Device order: the deterministic log of llpf_detmath.h, as the device snippet; reference order: libm. */
static void apply_noise(const orc_filter* f, const double* x, const double* fx, const double* xi, const double* uu, double* out) {
    const int nx = f->nx;
    if (f->user_noise_kind == 1) {
        for (int d = 0; d < nx; ++d) {
            const double sd = f->noise_par[0] + f->noise_par[1] * fabs(x[d]);
            out[d] = fx[d] + sd * xi[d];
        }
    } else if (f->user_noise_kind == 2) {
        const int dev = f->order == ORC_ORDER_DEVICE;
        for (int d = 0; d < nx; ++d) {
            const double v = 2.0 * uu[d] - 1.0;
            double tt = 1.0 - fabs(v);
            if (!(tt > 0.0)) tt = 1.1102230246251565e-16;
            const double mg = f->noise_par[0] * (-(dev ? llpf_log(tt) : log(tt)));
            out[d] = fx[d] + (v < 0.0 ? -mg : mg);
        }
    } else {
        double nz[MAXD];
        gauss_sample(&f->df, xi, nz);
        for (int d = 0; d < nx; ++d) out[d] = fx[d] + nz[d];
    }
}

/* ------------------------------------------------------------------------------------------
 * RBPF with constant matrices — src/rbpf.jl:63-283 (IPD = IPM = AUGD = false; A, An, C, R1 plain matrices, so the
 * "singleR" branches :176,:247 apply: one covariance recursion for all particles).
 * ---------------------------------------------------------------------------------------- */
static double rb_sqrt(double x) { return sqrt(x); }
static void gauss_cov_full(const llpf_gaussian* g, double* S);

static int rb_setup(orc_filter* f, int order) {
    const llpf_model* m = &f->cfg.model;
    const int nx = m->nx, nn = m->nxn, nl = nx - nn, nu = m->nu, ny = m->ny;
    if (nn < 1 || nl < 1 || nx > LLPF_RB_MAX || ny > LLPF_RB_MAX) return -1;
    f->rb.on = 1; f->rb.nn = nn; f->rb.nl = nl;
    for (int r = 0; r < nn; ++r) {
        for (int c = 0; c < nn; ++c) f->rb.Fn[r * nn + c] = m->A[r * nx + c];
        for (int c = 0; c < nl; ++c) f->rb.An[r * nl + c] = m->A[r * nx + nn + c];
        for (int c = 0; c < nu; ++c) f->rb.Bn[r * nu + c] = m->B[r * nu + c];
    }
    for (int r = 0; r < nl; ++r) {
        for (int c = 0; c < nl; ++c) f->rb.Al[r * nl + c] = m->A[(nn + r) * nx + nn + c];
        for (int c = 0; c < nu; ++c) f->rb.Bl[r * nu + c] = m->B[(nn + r) * nu + c];
    }
    for (int r = 0; r < ny; ++r) {
        for (int c = 0; c < nn; ++c) f->rb.Gn[r * nn + c] = m->C[r * nx + c];
        for (int c = 0; c < nl; ++c) f->rb.Cl[r * nl + c] = m->C[r * nx + nn + c];
    }
    f->rb.zeroAn = 1; for (int i = 0; i < nn * nl; ++i) if (f->rb.An[i] != 0.0) f->rb.zeroAn = 0;   /* iszero(An), :175 */
    f->rb.zeroC = 1;  for (int i = 0; i < ny * nl; ++i) if (f->rb.Cl[i] != 0.0) f->rb.zeroC = 0;    /* iszero(C), :244 */
    double tmp[MAXD * MAXD];
    gauss_cov_full(&m->linear_noise, tmp);       for (int i = 0; i < nl * nl; ++i) f->rb.R1l[i] = tmp[i];
    gauss_cov_full(&m->dynamics_density, tmp);   for (int i = 0; i < nn * nn; ++i) f->rb.R1nS[i] = tmp[i];
    gauss_cov_full(&m->measurement_density, tmp);for (int i = 0; i < ny * ny; ++i) f->rb.R2S[i] = tmp[i];
    gauss_cov_full(&m->linear_initial, tmp);
    for (int i = 0; i < nl * nl; ++i) { f->rb.R[i] = tmp[i]; f->rb.kfR[i] = tmp[i]; }                /* kf.R = d0.Sigma */
    for (int i = 0; i < nl; ++i) f->rb.kfx[i] = m->linear_initial.mu[i];                             /* kf.x = d0.mu    */
    /* the density reset! draws from: xn ~ d0n, xl = d0l.mu exactly (:146-158): [mu_n; mu_l] + blockdiag(L_n, 0) xi */
    gaussd d0n = f->d0;
    memset(&f->d0, 0, sizeof(f->d0));
    f->d0.dim = nx; f->d0.kind = LLPF_COV_FULL; f->d0.dev = (order == ORC_ORDER_DEVICE);
    for (int i = 0; i < nn; ++i) {
        f->d0.mu[i] = d0n.mu[i];
        for (int j = 0; j <= i; ++j)
            f->d0.L[i * MAXD + j] = (d0n.kind == LLPF_COV_FULL) ? d0n.L[i * MAXD + j] : (i == j ? d0n.L[i * MAXD + i] : 0.0);
    }
    for (int i = 0; i < nl; ++i) f->d0.mu[nn + i] = m->linear_initial.mu[i];
    return 0;
}

/* correct!(pf::RBPF, u, y, p, t) — src/rbpf.jl:235-283 (weights only; logsumexp! by the caller) */
static void rb_correct(orc_filter* f, const double* y) {
    const int nn = f->rb.nn, nl = f->rb.nl, nx = f->nx, ny = f->ny;
    const int dev = f->order == ORC_ORDER_DEVICE;
    double Rpost[16];
    if (!f->rb.zeroC) {
        /* i == 1: kf.x = x[1].xl; kf.R = x[1].R; correct!(kf, u, y - yn, p, t) — S, K and the new R serve every particle */
        double S[16];
        llpf_rb_gain(nl, ny, f->rb.R, f->rb.Cl, f->rb.R2S, S, f->rb.K, Rpost, dev ? llpf_sqrt : rb_sqrt);
        llpf_gaussian gs;
        memset(&gs, 0, sizeof(gs));
        gs.dim = ny; gs.kind = LLPF_COV_FULL;
        for (int i = 0; i < ny * ny; ++i) gs.cov[i] = S[i];
        gauss_prepare(&gs, &f->rb.dS, f->order);               /* SimpleMvNormal(PDMat(S, S_chol)) */
    }
    ORC_PAR
    for (int64_t i = 0; i < f->N; ++i) {
        double* x = f->x + i * nx;
        double yn[4], yl[4], e[4];
        for (int r = 0; r < ny; ++r) {                         /* yn = g(xn), yl = C xl, :254-255 */
            double a = f->rb.Gn[r * nn] * x[0];
            for (int c = 1; c < nn; ++c) a = a + f->rb.Gn[r * nn + c] * x[c];
            yn[r] = a;
            double b2 = f->rb.Cl[r * nl] * x[nn];
            for (int c = 1; c < nl; ++c) b2 = b2 + f->rb.Cl[r * nl + c] * x[nn + c];
            yl[r] = b2;
        }
        if (!f->rb.zeroC) {
            /* particle 1 goes through correct!(kf, u, y - yn): e = (y - yn) - C x (filtering.jl:102); the others
             * e = y - yh with yh = yn + yl (:266) */
            for (int r = 0; r < ny; ++r) e[r] = (i == 0) ? (y[r] - yn[r]) - yl[r] : y[r] - (yn[r] + yl[r]);
            f->w[i] += gauss_logpdf(&f->rb.dS, e);             /* w[i] += ll, :272 */
            for (int r = 0; r < nl; ++r) {                     /* kf.x = xl + K e, :267 / filtering.jl:122 */
                double a = f->rb.K[r * ny] * e[0];
                for (int c = 1; c < ny; ++c) a = a + f->rb.K[r * ny + c] * e[c];
                x[nn + r] = x[nn + r] + a;
            }
        } else {
            for (int r = 0; r < ny; ++r) e[r] = y[r] - (yn[r] + yl[r]);
            f->w[i] += gauss_logpdf(&f->dg, e);                /* extended_logpdf(R2, y - yh), :275-276 */
            for (int r = 0; r < nl; ++r) x[nn + r] = f->rb.kfx[r];   /* x[i] = RBParticle(xn, kf.x, kf.R), :279: kf untouched */
        }
    }
    if (!f->rb.zeroC) {
        for (int i = 0; i < nl * nl; ++i) { f->rb.R[i] = Rpost[i]; f->rb.kfR[i] = Rpost[i]; }
        for (int r = 0; r < nl; ++r) f->rb.kfx[r] = f->x[(f->N - 1) * nx + nn + r];   /* the object keeps the last particle's mean */
    } else {
        for (int i = 0; i < nl * nl; ++i) f->rb.R[i] = f->rb.kfR[i];
    }
    memcpy(f->xprev, f->x, sizeof(double) * (size_t)f->N * nx);   /* copyto!(s.xprev, s.x), :282 */
}

/* the propagation loop of predict!(pf::RBPF, ...) — src/rbpf.jl:180-224; xi: N x nx normals of which the first nn of
 * every row are used */
static void rb_propagate(orc_filter* f, const double* u, const double* xi, int res) {
    const int nn = f->rb.nn, nl = f->rb.nl, nx = f->nx, nu = f->nu;
    double R1[16];
    llpf_rb_predcov(nl, nn, f->rb.zeroAn, f->rb.R, f->rb.Al, f->rb.An, f->rb.R1l, f->rb.R1nS, f->rb.L, R1);
    ORC_PAR
    for (int64_t i = 0; i < f->N; ++i) {
        const double* xp = f->xprev + (res ? f->j[i] : i) * nx;
        double fi[4], nz[4], xl1[4];
        for (int r = 0; r < nn; ++r) {                         /* fi = f(xn, u, p, t) = Fn xn + Bn u */
            double a = f->rb.Fn[r * nn] * xp[0];
            for (int c = 1; c < nn; ++c) a = a + f->rb.Fn[r * nn + c] * xp[c];
            if (nu > 0) {
                double b2 = f->rb.Bn[r * nu] * u[0];
                for (int c = 1; c < nu; ++c) b2 = b2 + f->rb.Bn[r * nu + c] * u[c];
                a = a + b2;
            }
            fi[r] = a;
        }
        gauss_sample(&f->df, xi + i * nx, nz);                 /* rand(pf.rng, pf.R1n) */
        for (int r = 0; r < nl; ++r) {                         /* Al*xl + Bl*u */
            double a = f->rb.Al[r * nl] * xp[nn];
            for (int c = 1; c < nl; ++c) a = a + f->rb.Al[r * nl + c] * xp[nn + c];
            if (nu > 0) {
                double b2 = f->rb.Bl[r * nu] * u[0];
                for (int c = 1; c < nu; ++c) b2 = b2 + f->rb.Bl[r * nu + c] * u[c];
                a = a + b2;
            }
            xl1[r] = a;
        }
        double* xo = f->x + i * nx;
        if (f->rb.zeroAn) {
            for (int r = 0; r < nn; ++r) xo[r] = fi[r] + nz[r];                 /* xn1 = fi + rand(R1n), :199 */
            for (int r = 0; r < nl; ++r) xo[nn + r] = xl1[r];
        } else {
            double Axl[4] = {0, 0, 0, 0}, z[4] = {0, 0, 0, 0};
            for (int r = 0; r < nn; ++r) {                     /* Axl = An xl; z = Axl + rand(R1n); xn1 = fi + z, :216-218 */
                double a = f->rb.An[r * nl] * xp[nn];
                for (int c = 1; c < nl; ++c) a = a + f->rb.An[r * nl + c] * xp[nn + c];
                Axl[r] = a;
                z[r] = a + nz[r];
                xo[r] = fi[r] + z[r];
            }
            for (int r = 0; r < nl; ++r) {                     /* xl1 = Al xl + Bl u + L (z - Axl), :220 */
                double a = f->rb.L[r * nn] * (z[0] - Axl[0]);
                for (int c = 1; c < nn; ++c) a = a + f->rb.L[r * nn + c] * (z[c] - Axl[c]);
                xo[nn + r] = xl1[r] + a;
            }
        }
    }
    for (int i = 0; i < nl * nl; ++i) f->rb.R[i] = R1[i];
}

/* shared covariance of the linear substate (x[1].R) */
void orc_rb_get_R(const orc_filter* f, double* R) { for (int i = 0; i < f->rb.nl * f->rb.nl; ++i) R[i] = f->rb.R[i]; }

/* ------------------------------------------------------------------------------------------
 * RBPF whose An is a function of the nonlinear state — src/rbpf.jl:163-283 with singleR false (:176, :247): the
 * loops below are the reference's per-particle branches.  Kalman algebra per particle:
 *   device order    csrc/shared/llpf_rbfull_body.h (shared with the HIP kernel: fused multiply-adds, packed triangles,
 *                   the time update regrouped as "condition on z, then propagate" — see that file's header);
 *   reference order rbfr_predict / rbfr_correct below: the reference's formulas as written, on full matrices, plain
 *                   multiply and add in increasing index order, libm — nothing shared with the device order, so the
 *                   two check each other to rounding (tests/test_oracle_rbfull.py, tests/test_gpu_rbfull.py).
 * The particle's covariance is STORED as its lower triangle in both orders (the reference keeps the full matrix, whose
 * two triangles differ by rounding after `Al*R*Al' + R1l - L*Nt*L'`: acknowledged deviation, ~1e-17 relative).
 * ---------------------------------------------------------------------------------------- */
#define RBM LLPF_RBF_MAXL
/* C[m x n] = A[m x k] B[k x n], row-major with explicit leading dimensions; sums in increasing index order */
static void rbfr_mm(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int m, int k, int n) {
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int q = 0; q < k; ++q) acc += A[i * lda + q] * B[q * ldb + j];
        C[i * ldc + j] = acc;
    }
}
/* C[m x n] = A[m x k] B'[k x n] with B given as n x k */
static void rbfr_mmt(const double* A, int lda, const double* B, int ldb, double* C, int ldc, int m, int k, int n) {
    for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int q = 0; q < k; ++q) acc += A[i * lda + q] * B[j * ldb + q];
        C[i * ldc + j] = acc;
    }
}
/* X = G / N for a square N (n x n), G m x n: Julia's generic right division, (N' \ G')' through an LU factorization
 * with partial pivoting of N' (LinearAlgebra: `/` -> `\` -> lu for a dense square matrix that is not triangular). */
static void rbfr_rdiv(const double* G, int m, const double* Nm, int n, double* X) {
    double LU[LLPF_RBF_MAXN * LLPF_RBF_MAXN];
    int piv[LLPF_RBF_MAXN];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) LU[i * n + j] = Nm[j * n + i];     /* N' */
    for (int k = 0; k < n; ++k) {
        int pr = k;
        double best = fabs(LU[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (fabs(LU[i * n + k]) > best) { best = fabs(LU[i * n + k]); pr = i; }
        piv[k] = pr;
        if (pr != k) for (int j = 0; j < n; ++j) { const double tmp = LU[k * n + j]; LU[k * n + j] = LU[pr * n + j]; LU[pr * n + j] = tmp; }
        const double inv = 1.0 / LU[k * n + k];
        for (int i = k + 1; i < n; ++i) LU[i * n + k] *= inv;
        for (int i = k + 1; i < n; ++i) for (int j = k + 1; j < n; ++j) LU[i * n + j] -= LU[i * n + k] * LU[k * n + j];
    }
    for (int r = 0; r < m; ++r) {                               /* column r of G': solve N' x = G[r, :]' */
        double b[LLPF_RBF_MAXN];
        for (int i = 0; i < n; ++i) b[i] = G[r * n + i];
        for (int k = 0; k < n; ++k) if (piv[k] != k) { const double tmp = b[k]; b[k] = b[piv[k]]; b[piv[k]] = tmp; }
        for (int i = 0; i < n; ++i) for (int q = 0; q < i; ++q) b[i] -= LU[i * n + q] * b[q];
        for (int i = n - 1; i >= 0; --i) { for (int q = i + 1; q < n; ++q) b[i] -= LU[i * n + q] * b[q]; b[i] /= LU[i * n + i]; }
        for (int i = 0; i < n; ++i) X[r * n + i] = b[i];
    }
}
static void rbfr_unpack(const double* Rp, int nl, double* R) {
    for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c) R[r * RBM + c] = Rp[llpf_rbf_idx(r, c)];
}
/* predict!(pf::RBPF), the !zeroAn && !singleR branch as written — src/rbpf.jl:206-221 */
static void rbfr_predict(const llpf_rbf_par* p, int nn, int nl, int nu, const double* xn, const double* xl, const double* Rp,
                         const double* u, const double* fi, const double* nz, double* xn1, double* xl1, double* R1p) {
    double R[RBM * RBM], An[LLPF_RBF_MAXN * RBM], AnR[LLPF_RBF_MAXN * RBM], Nt[LLPF_RBF_MAXN * LLPF_RBF_MAXN];
    double AlR[RBM * RBM], G[RBM * LLPF_RBF_MAXN], L[RBM * LLPF_RBF_MAXN], LN[RBM * LLPF_RBF_MAXN], T1[RBM * RBM], T2[RBM * RBM];
    rbfr_unpack(Rp, nl, R);
    for (int r = 0; r < nn; ++r) for (int c = 0; c < nl; ++c) {   /* An = get_mat(pf.An, xi.xn, u, p, t), :208 */
        double a = p->An[0][r * nl + c];
        for (int k = 0; k < nn; ++k) a += xn[k] * p->An[1 + k][r * nl + c];
        An[r * RBM + c] = a;
    }
    rbfr_mm(An, RBM, R, RBM, AnR, RBM, nn, nl, nl);             /* Nt = An*R*An' + pf.R1n.Σ, :209 */
    rbfr_mmt(AnR, RBM, An, RBM, Nt, nn, nn, nl, nn);
    for (int i = 0; i < nn * nn; ++i) Nt[i] += p->R1n[i];
    rbfr_mm(p->Al, nl, R, RBM, AlR, RBM, nl, nl, nl);           /* L = Al*R*An' / Nt, :210 */
    rbfr_mmt(AlR, RBM, An, RBM, G, nn, nl, nl, nn);
    rbfr_rdiv(G, nl, Nt, nn, L);
    rbfr_mmt(AlR, RBM, p->Al, nl, T1, RBM, nl, nl, nl);         /* R1 = Al*R*Al' + R1l - L*Nt*L', :211 */
    rbfr_mm(L, nn, Nt, nn, LN, nn, nl, nn, nn);
    rbfr_mmt(LN, nn, L, nn, T2, RBM, nl, nn, nl);
    for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c)
        R1p[llpf_rbf_idx(r, c)] = (T1[r * RBM + c] + p->R1l[llpf_rbf_idx(r, c)]) - T2[r * RBM + c];
    double dz[LLPF_RBF_MAXN];
    for (int r = 0; r < nn; ++r) {                              /* Axl = An*xi.xl ; z = Axl + rand(R1n) ; xn1 = fi + z, :213-215 */
        double a = 0.0;
        for (int c = 0; c < nl; ++c) a += An[r * RBM + c] * xl[c];
        const double z = a + nz[r];
        xn1[r] = fi[r] + z;
        dz[r] = z - a;
    }
    for (int r = 0; r < nl; ++r) {                              /* xl1 = Al*xi.xl + Bl*u + L*(z - Axl), :217 */
        double a = 0.0, b = 0.0, c2 = 0.0;
        for (int c = 0; c < nl; ++c) a += p->Al[r * nl + c] * xl[c];
        for (int c = 0; c < nu; ++c) b += p->Bl[r * nu + c] * u[c];
        for (int c = 0; c < nn; ++c) c2 += L[r * nn + c] * dz[c];
        xl1[r] = (a + b) + c2;
    }
}
/* correct!(kf, u, y - yn, p, t) for one particle — src/rbpf.jl:259-263 -> src/filtering.jl:100-128 as written */
static double rbfr_correct(const llpf_rbf_par* p, int nl, int ny, const double* y, const double* yn, double* xl, double* Rp) {
    double R[RBM * RBM], CR[LLPF_RBF_MAXY * RBM], S[LLPF_RBF_MAXY * LLPF_RBF_MAXY], U[LLPF_RBF_MAXY * LLPF_RBF_MAXY];
    double RCt[RBM * LLPF_RBF_MAXY], K[RBM * LLPF_RBF_MAXY], IKC[RBM * RBM], Rn[RBM * RBM], e[LLPF_RBF_MAXY];
    rbfr_unpack(Rp, nl, R);
    for (int i = 0; i < ny; ++i) {                              /* e = y .- Ct*x */
        double a = 0.0;
        for (int c = 0; c < nl; ++c) a += p->Cl[i * nl + c] * xl[c];
        e[i] = (y[i] - yn[i]) - a;
    }
    rbfr_mm(p->Cl, nl, R, RBM, CR, RBM, ny, nl, nl);            /* S = symmetrize(Ct*R*Ct') .+ R2 */
    rbfr_mmt(CR, RBM, p->Cl, nl, S, ny, ny, nl, ny);
    for (int i = 0; i < ny; ++i) for (int j = i + 1; j < ny; ++j) { S[i * ny + j] = 0.5 * (S[i * ny + j] + S[j * ny + i]); S[j * ny + i] = S[i * ny + j]; }
    for (int i = 0; i < ny * ny; ++i) S[i] += p->R2[i];
    double ldet = 0.0;
    for (int j = 0; j < ny; ++j) {                              /* cholesky(Symmetric(S)): S = U'U, column by column */
        for (int i = 0; i <= j; ++i) {
            double acc = S[i * ny + j];
            for (int k = 0; k < i; ++k) acc -= U[k * ny + i] * U[k * ny + j];
            U[i * ny + j] = (i == j) ? sqrt(acc) : acc / U[i * ny + i];
        }
        ldet += log(U[j * ny + j]);
    }
    ldet = 2.0 * ldet;                                          /* logdet(::Cholesky) */
    rbfr_mmt(R, RBM, p->Cl, nl, RCt, ny, nl, nl, ny);           /* K = (R*Ct')/S_chol : k U'U = row  ->  (k U') U = row */
    for (int r = 0; r < nl; ++r) {
        double t[LLPF_RBF_MAXY] = {0.0};
        for (int i = 0; i < ny; ++i) {                          /* t U = row */
            double acc = RCt[r * ny + i];
            for (int q = 0; q < i; ++q) acc -= t[q] * U[q * ny + i];
            t[i] = acc / U[i * ny + i];
        }
        for (int i = ny - 1; i >= 0; --i) {                     /* k U' = t */
            double acc = t[i];
            for (int q = i + 1; q < ny; ++q) acc -= K[r * ny + q] * U[i * ny + q];
            K[r * ny + i] = acc / U[i * ny + i];
        }
    }
    for (int r = 0; r < nl; ++r) {                              /* kf.x += K*e */
        double a = 0.0;
        for (int i = 0; i < ny; ++i) a += K[r * ny + i] * e[i];
        xl[r] += a;
    }
    for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c) { /* kf.R = symmetrize((I - K*Ct)*R) */
        double a = 0.0;
        for (int i = 0; i < ny; ++i) a += K[r * ny + i] * p->Cl[i * nl + c];
        IKC[r * RBM + c] = (r == c ? 1.0 : 0.0) - a;
    }
    rbfr_mm(IKC, RBM, R, RBM, Rn, RBM, nl, nl, nl);
    for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c)
        Rp[llpf_rbf_idx(r, c)] = (r == c) ? Rn[r * RBM + r] : 0.5 * (Rn[c * RBM + r] + Rn[r * RBM + c]);
    double quad = 0.0;                                          /* extended_logpdf: mvnormal_c0 - invquad/2, src/utils.jl:252-257 */
    {
        double z[LLPF_RBF_MAXY];
        for (int i = 0; i < ny; ++i) {                          /* U' z = e */
            double acc = e[i];
            for (int q = 0; q < i; ++q) acc -= U[q * ny + i] * z[q];
            z[i] = acc / U[i * ny + i];
            quad += z[i] * z[i];
        }
    }
    return -((double)ny * log(6.283185307179586) + ldet) / 2.0 - quad / 2.0;
}
#undef RBM

static int rbf_setup(orc_filter* f, int order) {
    const llpf_model* m = &f->cfg.model;
    const int nn = m->nx, nl = m->rb.nxl, ny = m->ny, nu = m->nu;
    if (nl < 1 || nl > LLPF_RBF_MAXL || nn > LLPF_RBF_MAXN || ny > LLPF_RBF_MAXY) return -1;
    if (m->linear_noise.dim != nl || m->linear_initial.dim != nl) return -1;
    llpf_rbf_par* q = &f->rbf.par;
    memset(q, 0, sizeof(*q));
    q->nn = nn; q->nl = nl; q->ny = ny; q->nu = nu;
    int zeroC = 1;
    for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c) q->Al[r * nl + c] = m->rb.Al[r * nl + c];
    for (int r = 0; r < nl; ++r) for (int c = 0; c < nu; ++c) q->Bl[r * nu + c] = m->rb.Bl[r * nu + c];
    for (int r = 0; r < ny; ++r) for (int c = 0; c < nl; ++c) { q->Cl[r * nl + c] = m->rb.Cl[r * nl + c]; if (q->Cl[r * nl + c] != 0.0) zeroC = 0; }
    if (zeroC) return -1;
    for (int k = 0; k <= nn; ++k) for (int i = 0; i < nn * nl; ++i) q->An[k][i] = m->rb.An[k][i];
    double S[64];
    gauss_cov_full(&m->linear_noise, S);
    for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c) q->R1l[llpf_rbf_idx(r, c)] = S[r * nl + c];
    gauss_cov_full(&m->linear_initial, S);
    for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c) q->R0[llpf_rbf_idx(r, c)] = S[r * nl + c];
    for (int r = 0; r < nl; ++r) q->xl0[r] = m->linear_initial.mu[r];
    gauss_cov_full(&m->dynamics_density, S);
    for (int i = 0; i < nn * nn; ++i) q->R1n[i] = S[i];
    gauss_cov_full(&m->measurement_density, S);
    for (int i = 0; i < ny * ny; ++i) q->R2[i] = S[i];
    q->c0y = (order == ORC_ORDER_DEVICE) ? -((double)ny * llpf_log(6.283185307179586)) / 2.0
                                         : -((double)ny * log(6.283185307179586)) / 2.0;
    f->rbf.on = 1; f->rbf.nn = nn; f->rbf.nl = nl; f->rbf.np = LLPF_RBF_NP(nl);
    size_t N = (size_t)f->N;
    f->rbf.xl = (double*)calloc(N * nl, 8); f->rbf.xlprev = (double*)calloc(N * nl, 8);
    f->rbf.R = (double*)calloc(N * f->rbf.np, 8); f->rbf.Rprev = (double*)calloc(N * f->rbf.np, 8);
    return 0;
}
/* reset!(pf::RBPF) — src/rbpf.jl:146-160: xl = copy(kf.d0.mu), R = copy(kf.d0.Sigma) */
static void rbf_reset(orc_filter* f) {
    const int nl = f->rbf.nl, np = f->rbf.np;
    for (int64_t i = 0; i < f->N; ++i) {
        for (int d = 0; d < nl; ++d) { f->rbf.xl[i * nl + d] = f->rbf.par.xl0[d]; f->rbf.xlprev[i * nl + d] = f->rbf.par.xl0[d]; }
        for (int d = 0; d < np; ++d) { f->rbf.R[i * np + d] = f->rbf.par.R0[d]; f->rbf.Rprev[i * np + d] = f->rbf.par.R0[d]; }
    }
}
/* correct!(pf::RBPF, u, y, p, t) — src/rbpf.jl:252-280, !zeroC && !singleR: every particle runs correct!(kf, u, y - yn) */
static void rbf_correct(orc_filter* f, const double* u, const double* y, double t) {
    const int nn = f->rbf.nn, nl = f->rbf.nl, np = f->rbf.np, ny = f->ny;
    const int dev = f->order == ORC_ORDER_DEVICE;
    ORC_PAR
    for (int64_t i = 0; i < f->N; ++i) {
        double yn[LLPF_RBF_MAXY];
        orc_measurement(&f->cfg.model, f->x + i * nn, u, t, yn);
        const double ll = dev ? llpf_rbf_correct(&f->rbf.par, nl, ny, y, yn, f->rbf.xl + i * nl, f->rbf.R + i * np)
                              : rbfr_correct(&f->rbf.par, nl, ny, y, yn, f->rbf.xl + i * nl, f->rbf.R + i * np);
        f->w[i] += ll;
    }
    memcpy(f->xprev, f->x, sizeof(double) * (size_t)f->N * nn);           /* copyto!(s.xprev, s.x), :282 */
    memcpy(f->rbf.xlprev, f->rbf.xl, sizeof(double) * (size_t)f->N * nl);
    memcpy(f->rbf.Rprev, f->rbf.R, sizeof(double) * (size_t)f->N * np);
}
/* the propagation loop of predict!(pf::RBPF, ...) — src/rbpf.jl:180-224, !zeroAn && !singleR */
static void rbf_propagate(orc_filter* f, const double* u, double t, const double* xi, int res) {
    const int nn = f->rbf.nn, nl = f->rbf.nl, np = f->rbf.np, nu = f->nu;
    const int dev = f->order == ORC_ORDER_DEVICE;
    ORC_PAR
    for (int64_t i = 0; i < f->N; ++i) {
        const int64_t a = res ? f->j[i] : i;                              /* xi = s.xprev[j[i]], :182 */
        double fi[LLPF_RBF_MAXN], nz[LLPF_RBF_MAXN];
        orc_dynamics(&f->cfg.model, f->xprev + a * nn, u, t, fi);
        gauss_sample(&f->df, xi + i * nn, nz);                            /* rand(pf.rng, pf.R1n), :217 */
        if (dev) llpf_rbf_predict(&f->rbf.par, nn, nl, nu, f->xprev + a * nn, f->rbf.xlprev + a * nl, f->rbf.Rprev + a * np, u, NULL, fi, nz,
                                  f->x + i * nn, f->rbf.xl + i * nl, f->rbf.R + i * np);
        else rbfr_predict(&f->rbf.par, nn, nl, nu, f->xprev + a * nn, f->rbf.xlprev + a * nl, f->rbf.Rprev + a * np, u, fi, nz,
                               f->x + i * nn, f->rbf.xl + i * nl, f->rbf.R + i * np);
    }
    memcpy(f->rbf.xlprev, f->rbf.xl, sizeof(double) * (size_t)f->N * nl);  /* copyto!(s.xprev, s.x), :227 */
    memcpy(f->rbf.Rprev, f->rbf.R, sizeof(double) * (size_t)f->N * np);
}
/* per-particle Kalman state: xl [N][nl], R [N][nl][nl] (either may be NULL) */
void orc_rb_get_linear_state(const orc_filter* f, double* xl, double* R) {
    const int nl = f->rbf.nl, np = f->rbf.np;
    for (int64_t i = 0; i < f->N; ++i) {
        if (xl) for (int d = 0; d < nl; ++d) xl[i * nl + d] = f->rbf.xl[i * nl + d];
        if (R) for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c) R[(i * nl + r) * nl + c] = f->rbf.R[i * np + llpf_rbf_idx(r, c)];
    }
}

orc_filter* orc_create(const llpf_config* cfg, int order) {
    orc_filter* f = (orc_filter*)calloc(1, sizeof(orc_filter));
    f->cfg = *cfg;
    f->order = order;
    f->N = cfg->n_particles;
    f->nx = cfg->model.nx; f->nu = cfg->model.nu; f->ny = cfg->model.ny;
    if (gauss_prepare(&cfg->model.dynamics_density, &f->df, order) ||
        gauss_prepare(&cfg->model.measurement_density, &f->dg, order) ||
        gauss_prepare(&cfg->model.initial_density, &f->d0, order)) { free(f); return NULL; }
    if (cfg->model.model_id == LLPF_MODEL_RB_LINEAR && rb_setup(f, order)) { free(f); return NULL; }
    if (cfg->model.model_id == LLPF_MODEL_RB_BILINEAR && rbf_setup(f, order)) { free(f); return NULL; }
    size_t N = (size_t)f->N;
    f->x = (double*)calloc(N * f->nx, 8); f->xprev = (double*)calloc(N * f->nx, 8);
    f->w = (double*)calloc(N, 8); f->we = (double*)calloc(N, 8);
    f->bins = (double*)calloc(N, 8); f->e = (double*)calloc(N, 8);
    f->j = (int64_t*)calloc(N, 8);
    f->xi_buf = (double*)calloc(N * f->nx, 8); f->U_buf = (double*)calloc(N, 8);
    f->uu_buf = (double*)calloc(N * f->nx, 8);
    set_key(f, cfg->seed);
    /* constructor: particles ~ d0, w = log(1/N), we = 1/N, j = 1:N, t = 0 — src/PFtypes.jl:65-75 */
    gen_normals(f, f->n_reset, LLPF_STREAM_INIT, f->xi_buf);
    f->n_reset++;
    init_particles(f, f->xi_buf);
    if (f->rbf.on) rbf_reset(f);
    fill_uniform_weights(f, order == ORC_ORDER_DEVICE ? llpf_log(1.0 / (double)f->N) : log(1.0 / (double)f->N));
    for (int64_t i = 0; i < f->N; ++i) f->j[i] = i;
    f->t = 0;
    return f;
}

void orc_destroy(orc_filter* f) {
    if (!f) return;
    free(f->x); free(f->xprev); free(f->w); free(f->we); free(f->bins); free(f->e); free(f->j);
    free(f->rbf.xl); free(f->rbf.xlprev); free(f->rbf.R); free(f->rbf.Rprev);
    free(f->xi_buf); free(f->U_buf); free(f->uu_buf); free(f->lam); free(f);
}

void orc_seed(orc_filter* f, uint64_t seed) { set_key(f, seed); }

/* reset!(pf) — src/filtering.jl:4-14 */
void orc_reset_explicit(orc_filter* f, const double* xi) {
    init_particles(f, xi);
    if (f->rbf.on) rbf_reset(f);
    f->aux_pending = 0;
    fill_uniform_weights(f, f->order == ORC_ORDER_DEVICE ? -llpf_log((double)f->N) : -log((double)f->N));
    f->t = 1;
}
void orc_reset(orc_filter* f) {
    gen_normals(f, f->n_reset, LLPF_STREAM_INIT, f->xi_buf);
    if (f->user_init_kind) gen_uniforms(f, f->n_reset, LLPF_STREAM_USER_INIT, f->uu_buf);
    f->n_reset++;
    orc_reset_explicit(f, f->xi_buf);
}

/* normalisation of the current raw log-weights in the filter's order */
static double filter_logsumexp(orc_filter* f, double off, int bound) {
    if (f->order == ORC_ORDER_DEVICE) {
        if (bound) {
            dev_norm_bound(f->w, f->e, f->N, off, &f->dn);
            if (!f->dn.fast) f->n_exact_steps++;
        } else {
            dev_expsum(f->w, f->e, f->N, &f->dn);              /* no bound known for these weights: exact-max form */
        }
        f->dn_valid = 1;
        for (int64_t i = 0; i < f->N; ++i) {
            f->we[i] = f->e[i] * f->dn.inv;
            f->w[i] = (f->w[i] - f->dn.m) - f->dn.l;
        }
        f->maxw = f->dn.mtrue;
        f->wmax = (f->dn.mtrue - f->dn.m) - f->dn.l;          /* the normalised weight of the best particle */
        double ll = f->dn.l + f->dn.m;
        if (!(ll == ll) || f->dn.mtrue == -LLPF_INF) f->degenerate = 1;
        return ll;
    }
    double ll = orc_logsumexp(f->w, f->we, f->N, ORC_ORDER_REFERENCE, &f->maxw);
    if (!(ll == ll)) f->degenerate = 1;
    return ll;
}

/* ---- user likelihoods (test-side counterparts of the device snippets in tests/user_models.py) -----------------------------------
 * kind 1  Laplace, independent components, scale b = par[0]:      ll = -(sum_k |v_k|) / b - ny log(2 b)
 * kind 2  Student-t, independent components, nu = par[0], sigma = par[1], c1 = par[2] (the per-component log-normaliser
 *         lgamma((nu+1)/2) - lgamma(nu/2) - log(nu pi)/2 - log sigma, formed by the caller):
 *                                                                 ll = sum_k ( c1 - (nu+1)/2 log1p((v_k/sigma)^2 / nu) )
 * v = y - g(x).  Device order: the deterministic log / log1p of llpf_detmath.h, as the device snippet; reference order: libm. */
int orc_set_user_loglik(orc_filter* f, int kind, const double* par, int npar) {
    if (!f || kind < 0 || kind > 2 || npar > 4) return -1;
    f->user_ll_kind = kind;
    for (int i = 0; i < 4; ++i) f->user_par[i] = (par && i < npar) ? par[i] : 0.0;
    const int dev = f->order == ORC_ORDER_DEVICE;
    if (kind == 1) {
        f->user_c = (double)f->ny * (dev ? llpf_log(2.0 * f->user_par[0]) : log(2.0 * f->user_par[0]));
        f->user_bound = -f->user_c;
    } else if (kind == 2) {
        f->user_c = (f->user_par[0] + 1.0) / 2.0;
        double b = 0.0;
        for (int k = 0; k < f->ny; ++k) b = b + f->user_par[2];
        f->user_bound = b;
    }
    return 0;
}
/* the model's own process noise (apply_noise above) / initial density (init_particles).  Installing an initial density redraws the
 * constructor's particles when nothing else has happened to the filter yet (the engine's constructor draws from the model's own) */
int orc_set_user_noise(orc_filter* f, int kind, const double* par, int npar) {
    if (!f || kind < 0 || kind > 2 || npar > 4) return -1;
    f->user_noise_kind = kind;
    for (int i = 0; i < 4; ++i) f->noise_par[i] = (par && i < npar) ? par[i] : 0.0;
    return 0;
}
int orc_set_user_initial(orc_filter* f, int kind, const double* par, int npar) {
    if (!f || kind < 0 || kind > 1 || npar > 2 * MAXD) return -1;
    f->user_init_kind = kind;
    for (int i = 0; i < 2 * MAXD; ++i) f->init_par[i] = (par && i < npar) ? par[i] : 0.0;
    if (kind && f->n_reset == 1 && f->t == 0 && f->n_predict == 0) {
        gen_uniforms(f, 0, LLPF_STREAM_USER_INIT, f->uu_buf);
        init_particles(f, f->xi_buf);
    }
    return 0;
}
static double meas_bound(const orc_filter* f) { return f->user_ll_kind ? f->user_bound : f->dg.c0; }
static double meas_loglik(const orc_filter* f, const double* v) {
    if (!f->user_ll_kind) return gauss_logpdf(&f->dg, v);
    const int dev = f->order == ORC_ORDER_DEVICE;
    if (f->user_ll_kind == 1) {
        double s = fabs(v[0]);
        for (int k = 1; k < f->ny; ++k) s = s + fabs(v[k]);
        return (-(s / f->user_par[0])) - f->user_c;
    }
    double ll = 0.0;
    for (int k = 0; k < f->ny; ++k) {
        const double z = v[k] / f->user_par[1];
        const double q = (z * z) / f->user_par[0];
        ll = ll + (f->user_par[2] - f->user_c * (dev ? llpf_log1p_nonneg(q) : log1p(q)));
    }
    return ll;
}

/* correct!(pf,u,y,p,t) — src/filtering.jl:164-168; measurement_equation! src/PFtypes.jl:107-120 (PF),
 * :226-239 (Advanced: w[i] += measurement_likelihood(x[i],u,y,p,t), which for the built-in models is
 * logpdf(dg, y - g(x))) */
double orc_correct(orc_filter* f, const double* u, const double* y, double t) {
    const int has_y = (y != NULL && y[0] == y[0]);
    const double off = has_y ? f->wmax + meas_bound(f) : f->wmax;   /* device order: upper bound of the new weights */
    if (f->rbf.on && has_y) {
        /* S_i = C R_i C' + R2 >= R2: the peak of N(0, R2) bounds every increment (+ the slack the kernel uses) */
        rbf_correct(f, u, y, t);
        f->aux_pending = 0;
        return filter_logsumexp(f, (f->wmax + f->dg.c0) + 0x1p-20, 1);
    }
    if (f->rb.on && has_y) {
        rb_correct(f, y);
        f->aux_pending = 0;
        return filter_logsumexp(f, f->wmax + (f->rb.zeroC ? f->dg.c0 : f->rb.dS.c0), 1);
    }
    if (has_y) {                                               /* any(ismissing, y) && return w */
        ORC_PAR
        for (int64_t i = 0; i < f->N; ++i) {
            double g[MAXD], v[MAXD];
            orc_measurement(&f->cfg.model, f->x + i * f->nx, u, t, g);
            for (int k = 0; k < f->ny; ++k) v[k] = y[k] - g[k];
            f->w[i] += meas_loglik(f, v);
        }
    }
    f->aux_pending = 0;
    return filter_logsumexp(f, off, 1);
}

static double filter_ess(const orc_filter* f) {
    if (f->order == ORC_ORDER_DEVICE && f->dn_valid) {
        /* 1/sum(we^2) with we = e/(s+1): (s+1)^2 / sum(e^2), sum(e^2) exact in fixed point */
        return (f->dn.stot * f->dn.stot) / f->dn.e2;
    }
    if (f->order == ORC_ORDER_DEVICE) {
        /* uniform weights: we = 1/N exactly representable product N * (1/N)^2 */
        double wev = f->we[0];
        return 1.0 / ((double)f->N * (wev * wev));
    }
    return orc_effective_particles(f->we, f->N);
}
double orc_filter_ess(const orc_filter* f) { return filter_ess(f); }

/* shouldresample(pf) — src/resample.jl:5-10 */
int orc_shouldresample(const orc_filter* f) {
    if (f->cfg.resample_threshold == 1.0) return 1;
    double th = (double)f->N * f->cfg.resample_threshold;
    if (f->order == ORC_ORDER_DEVICE && f->dn_valid) {
        /* the same test without the division: (s+1)^2 < N thr sum(e^2) */
        return f->dn.stot * f->dn.stot < th * f->dn.e2;
    }
    return filter_ess(f) < th;
}

/* device-order resample on the filter path: bins from the UNNORMALISED exp-weights e_i = exp(w_i - m)
 * (scale-invariant: bins = cumQ/totQ), so that the GPU needs no normalised-weight array */
static void filter_resample_dev(orc_filter* f, const double* U) {
    int64_t n = f->N;
    if (f->cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL) {
        uint64_t* q = (uint64_t*)malloc(8 * (size_t)n);
        for (int64_t i = 0; i < n; ++i) q[i] = llpf_q64_unit(f->e[i], f->dn.K);
        resample_residual(NULL, q, n, n, U, f->j, f->bins, ORC_ORDER_DEVICE);
        free(q);
        return;
    }
    uint64_t cum = 0;
    double Td = (double)f->dn.totQ;
    double invTd = 1.0 / Td;
    for (int64_t i = 0; i < n; ++i) {
        cum += llpf_q64_unit(f->e[i], f->dn.K);
        f->bins[i] = (double)cum * invTd;
    }
    double binsN = Td * invTd;
    thr_ctx c;
    c.strategy = f->cfg.resampling_strategy; c.m = n; c.U = U; c.step = 1.0 / (double)n;
    c.r = (c.strategy == LLPF_RESAMPLE_SYSTEMATIC) ? U[0] * binsN / (double)n : 0.0;
    int64_t bo = 0;
    for (int64_t i = 0; i < n; ++i) {
        double si = thr_at(&c, i, binsN);
        for (int64_t k = bo; k < n; ++k) {
            if (si < f->bins[k]) { f->j[i] = k; bo = k; break; }
        }
    }
}

/* predict!(pf,u,p,t) — src/filtering.jl:140-153 */
void orc_predict_explicit(orc_filter* f, const double* u, double t, const double* xi, const double* U) {
    int64_t N = f->N;
    int nx = f->nx;
    int res = orc_shouldresample(f);
    if (f->rb.on || f->rbf.on) {                              /* predict!(pf::RBPF, ...) — src/rbpf.jl:163-232 */
        if (res) {
            if (f->order == ORC_ORDER_DEVICE && f->dn_valid) filter_resample_dev(f, U);
            else orc_resample(f->cfg.resampling_strategy, f->we, N, N, U, f->j, f->bins, f->order);
        } else {
            for (int64_t i = 0; i < N; ++i) f->j[i] = i;
        }
        if (f->rbf.on) rbf_propagate(f, u, t, xi, res);
        else rb_propagate(f, u, xi, res);
        if (res) {
            fill_uniform_weights(f, f->order == ORC_ORDER_DEVICE ? llpf_log(1.0 / (double)N) : log(1.0 / (double)N));
            f->maxw = 0.0;
            f->resample_count++;
        }
        memcpy(f->xprev, f->x, sizeof(double) * (size_t)N * nx);
        f->t += 1;
        f->last_resampled = res;
        return;
    }
    if (res) {
        /* j = resample(pf) — src/resample.jl:12 */
        if (f->order == ORC_ORDER_DEVICE && f->dn_valid) filter_resample_dev(f, U);
        else orc_resample(f->cfg.resampling_strategy, f->we, N, N, U, f->j, f->bins, f->order);
        /* propagate_particles!(pf,u,j,p,t) — src/PFtypes.jl:122-139 (PF), :242-259 (Advanced) */
        ORC_PAR
        for (int64_t i = 0; i < N; ++i) {
            double fx[MAXD];
            orc_dynamics(&f->cfg.model, f->xprev + f->j[i] * nx, u, t, fx);
            apply_noise(f, f->xprev + f->j[i] * nx, fx, xi + i * nx, f->uu_buf + i * nx, f->x + i * nx);
        }
        /* reset_weights!(s) — src/utils.jl:73-79: fill!(w, log(1/N)); fill!(we, 1/N); maxw = 0 */
        fill_uniform_weights(f, f->order == ORC_ORDER_DEVICE ? llpf_log(1.0 / (double)N) : log(1.0 / (double)N));
        f->maxw = 0.0;
        f->resample_count++;
    } else {
        for (int64_t i = 0; i < N; ++i) f->j[i] = i;          /* s.j .= 1:N, :148 */
        /* propagate_particles!(pf,u,p,t) — DistributionsExt:83-93 (PF), src/PFtypes.jl:276-289 (Advanced) */
        ORC_PAR
        for (int64_t i = 0; i < N; ++i) {
            double fx[MAXD];
            orc_dynamics(&f->cfg.model, f->xprev + i * nx, u, t, fx);
            apply_noise(f, f->xprev + i * nx, fx, xi + i * nx, f->uu_buf + i * nx, f->x + i * nx);
        }
    }
    memcpy(f->xprev, f->x, sizeof(double) * (size_t)N * nx);   /* copyto!(s.xprev, s.x), :151 */
    f->t += 1;                                                /* :152 */
    f->last_resampled = res;
}

void orc_predict(orc_filter* f, const double* u, double t) {
    uint32_t step = f->n_predict++;
    gen_normals(f, step, LLPF_STREAM_DYNAMICS, f->xi_buf);
    if (f->user_noise_kind) gen_uniforms(f, step, LLPF_STREAM_USER, f->uu_buf);
    if (f->cfg.resampling_strategy == LLPF_RESAMPLE_SYSTEMATIC)
        f->U_buf[0] = llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, f->k0, f->k1);
    else
        for (int64_t i = 0; i < f->N; ++i)
            f->U_buf[i] = llpf_uniform_idx((uint32_t)i, step, LLPF_STREAM_STRATIFY, f->k0, f->k1);
    orc_predict_explicit(f, u, t, f->xi_buf, f->U_buf);
}

/* update!(pf,u,y,p,t) — src/filtering.jl:181-185 */
double orc_update(orc_filter* f, const double* u, const double* y, double t) {
    double ll = orc_correct(f, u, y, t);
    orc_predict(f, u, t);
    return ll;
}

/* ------------------------------------------------------------------------------------------
 * AuxiliaryParticleFilter{ParticleFilter} — src/PFtypes.jl:38-49, src/filtering.jl:170-217, 367-384,
 * src/smoothing.jl:232-236.  The {AdvancedParticleFilter} variant, filtering.jl:219-234, discards
 * lambda and re-propagates with noise: restated inside orc_aux_predict for filter_kind LLPF_ADVANCED_PARTICLE_FILTER.
 * ---------------------------------------------------------------------------------------- */
/* correct!(pf::AuxiliaryParticleFilter, u, y, p, t) — src/filtering.jl:170-174: the measurement update was done in
 * the predict step, only ll = logsumexp!(state) remains (so y of the very first call is never used). */
double orc_aux_correct(orc_filter* f) {
    const int pend = f->aux_pending;
    f->aux_pending = 0;
    /* device order: weights produced by an aux predict! are bounded by aux_off (lambda <= c0); anything else
     * (uniform after reset!, already normalised, installed) is normalised in the exact-max form */
    return filter_logsumexp(f, f->aux_off, pend);
}

/* predict!(pf::AuxiliaryParticleFilter, u, y1, p, t) — src/filtering.jl:195-217 */
void orc_aux_predict(orc_filter* f, const double* u, const double* y1, double t) {
    const int64_t N = f->N;
    const int nx = f->nx;
    const uint32_t step = f->n_predict++;
    gen_normals(f, step, LLPF_STREAM_DYNAMICS, f->xi_buf);
    if (f->user_noise_kind) gen_uniforms(f, step, LLPF_STREAM_USER, f->uu_buf);
    if (f->cfg.resampling_strategy == LLPF_RESAMPLE_SYSTEMATIC)
        f->U_buf[0] = llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, f->k0, f->k1);
    else
        for (int64_t i = 0; i < N; ++i)
            f->U_buf[i] = llpf_uniform_idx((uint32_t)i, step, LLPF_STREAM_STRATIFY, f->k0, f->k1);
    if (f->aux_pending) orc_aux_correct(f);                   /* (engine contract: weights are normalised first) */
    if (!f->lam) f->lam = (double*)calloc((size_t)N, 8);
    const int has_y = (y1 != NULL && y1[0] == y1[0]);
    const int dev = f->order == ORC_ORDER_DEVICE;
    /* propagate_particles!(pf.pf, u, p, t, nothing): x[i] = f(xprev[i], u, p, t), no noise — src/PFtypes.jl:261-274 */
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) orc_dynamics(&f->cfg.model, f->xprev + i * nx, u, t, f->x + i * nx);
    /* lambda = s.we; lambda .= 0; measurement_equation!(pf.pf, u, y1, p, t, lambda) — :201-203 */
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) {
        double lam = 0.0;
        if (has_y) {
            double g[MAXD], v[MAXD];
            orc_measurement(&f->cfg.model, f->x + i * nx, u, t, g);
            for (int k = 0; k < f->ny; ++k) v[k] = y1[k] - g[k];
            lam += meas_loglik(f, v);
        }
        f->lam[i] = lam;
        f->w[i] += lam;                                        /* s.w .+= lambda, :204 */
    }
    /* expnormalize!(s.w) (w used as buffer, :205) ; j = resample(strategy, s.w, s.j, s.bins), :206 */
    if (dev) {
        const double off = has_y ? f->wmax + meas_bound(f) : f->wmax;
        dev_norm_bound(f->w, f->e, N, off, &f->dn);
        if (!f->dn.fast) f->n_exact_steps++;
        f->dn_valid = 1;
        filter_resample_dev(f, f->U_buf);
    } else {
        orc_expnormalize_inplace(f->w, N);
        orc_resample(f->cfg.resampling_strategy, f->w, N, N, f->U_buf, f->j, f->bins, ORC_ORDER_REFERENCE);
    }
    if (f->cfg.filter_kind == LLPF_ADVANCED_PARTICLE_FILTER) {
        /* predict!(pf::AuxiliaryParticleFilter{<:AdvancedParticleFilter}, ...) — src/filtering.jl:219-234: lambda only steered the
         * resampling; reset_weights!(s) (:226), then propagate_particles!(pf.pf, u, j, p, t) (:228): "propagate with noise and
         * permutation" — from the PREVIOUS particles xprev[j], the noise-free prediction in s.x is overwritten */
        ORC_PAR
        for (int64_t i = 0; i < N; ++i) {
            double fx[MAXD];
            orc_dynamics(&f->cfg.model, f->xprev + f->j[i] * nx, u, t, fx);
            apply_noise(f, f->xprev + f->j[i] * nx, fx, f->xi_buf + i * nx, f->uu_buf + i * nx, f->x + i * nx);
        }
        fill_uniform_weights(f, dev ? llpf_log(1.0 / (double)N) : log(1.0 / (double)N));
        f->maxw = 0.0;
        f->t += 1;                                            /* :230 */
        memcpy(f->xprev, f->x, sizeof(double) * (size_t)N * nx);  /* :231 */
        f->dn_valid = 0;
        f->aux_pending = 0;
        f->last_resampled = 1;
        f->resample_count++;
        return;
    }
    /* permute_with_buffer!(s.x, s.xprev, j): buf[i] = x[j[i]]; copyto!(x, buf) — src/utils.jl:81-86 */
    for (int64_t i = 0; i < N; ++i)
        for (int d = 0; d < nx; ++d) f->xprev[i * nx + d] = f->x[f->j[i] * nx + d];
    /* add_noise!(pf.pf): x[i] += rand!(rng, df, noise) — src/PFtypes.jl:143-155 */
    ORC_PAR
    for (int64_t i = 0; i < N; ++i) {
        double nz[MAXD];
        gauss_sample(&f->df, f->xi_buf + i * nx, nz);
        for (int d = 0; d < nx; ++d) f->x[i * nx + d] = f->xprev[i * nx + d] + nz[d];
    }
    /* s.w[i] = lambda[i] - log(N)  ("note unresampled lambda[i] instead of lambda[j[i]]", :209-213) */
    const double lN = dev ? llpf_log((double)N) : log((double)N);
    for (int64_t i = 0; i < N; ++i) {
        f->w[i] = f->lam[i] - lN;
        f->we[i] = f->lam[i];                                  /* the reference's `we` now holds lambda */
    }
    f->t += 1;                                                /* :215 */
    memcpy(f->xprev, f->x, sizeof(double) * (size_t)N * nx);  /* :216 */
    f->dn_valid = 0;
    f->aux_pending = 1;
    f->aux_off = (has_y ? meas_bound(f) : 0.0) - lN;
    f->wmax = f->aux_off;                                      /* an upper bound of the current log-weights */
    f->last_resampled = 1;
    f->resample_count++;
}

/* update!(pf::AuxiliaryParticleFilter, u, y, y1, p, t) — src/filtering.jl:187-191 */
double orc_aux_update(orc_filter* f, const double* u, const double* y1, double t) {
    double ll = orc_aux_correct(f);
    orc_aux_predict(f, u, y1, t);
    return ll;
}

/* mode 0: forward_trajectory(pf::AuxiliaryParticleFilter, u, y, p) — src/filtering.jl:367-384 (after reset!)
 * mode 1: loglik(pf::AuxiliaryParticleFilter, u, y, p) — src/smoothing.jl:232-236: T-1 aux updates, then one update!
 *         of the wrapped ParticleFilter on (u[end], y[end]) */
double orc_run_aux(orc_filter* f, const double* U, const double* Y, int64_t T, int mode,
                   double* ll_steps, double* xmean, double* x_hist, double* w_hist, double* we_hist) {
    double ll = 0.0;
    size_t N = (size_t)f->N;
    const double Ts = f->cfg.model.Ts;
    for (int64_t k = 0; k < T; ++k) {
        const double ti = (double)k * Ts;
        const double* u = U + k * f->nu;
        double lli;
        if (mode == 1 && k == T - 1) {
            lli = orc_update(f, u, Y + k * f->ny, ti);        /* pf.pf(u[end], y[end], p, (T-1)*Ts) */
        } else {
            lli = orc_aux_correct(f);
            if (xmean) orc_weighted_mean(f, xmean + k * f->nx);
            if (x_hist) memcpy(x_hist + (size_t)k * N * f->nx, f->x, 8 * N * f->nx);
            if (w_hist) memcpy(w_hist + (size_t)k * N, f->w, 8 * N);
            if (we_hist) memcpy(we_hist + (size_t)k * N, f->we, 8 * N);
            if (k < T - 1) orc_aux_predict(f, u, Y + (k + 1) * f->ny, ti);
        }
        ll += lli;
        if (ll_steps) ll_steps[k] = lli;
    }
    return ll;
}

/* ------------------------------------------------------------------------------------------
 * Forward-filtering backward-simulation smoother — smooth(pf, xf, wf, wef, ll, M, u, y, p), src/smoothing.jl:116-143
 * ---------------------------------------------------------------------------------------- */
/* i = draw_one_categorical(pf, w) — src/resample.jl:128-152 (0-based result; w: log-weights, clobbered).
 * Reference order: logsumexp!(w, bins), serial cumsum, s = rand()*bins[end], the two half-range linear searches with
 * `<=`.  Device order: quanta of exp(w - max w), bins = fl(fl(cum) fl(1/fl(total))), first b with s <= bins[b]. */
int64_t orc_draw_one_categorical(double* w, double* bins, int64_t n, double u, int order) {
    if (order == ORC_ORDER_REFERENCE) {
        orc_logsumexp(w, bins, n, ORC_ORDER_REFERENCE, NULL);
        for (int64_t i = 1; i < n; ++i) bins[i] += bins[i - 1];
        double s = u * bins[n - 1];
        int64_t mid = n / 2;                                   /* 1-based midpoint */
        if (mid >= 1 && s < bins[mid - 1]) {
            for (int64_t b = 1; b <= mid; ++b) if (s <= bins[b - 1]) return b - 1;
        } else {
            for (int64_t b = (mid >= 1 ? mid : 1); b <= n; ++b) if (s <= bins[b - 1]) return b - 1;
        }
        return n - 1;
    }
    double m = w[0];
    for (int64_t i = 1; i < n; ++i) m = llpf_fmax(m, w[i]);
    const int K = llpf_qbits(n);
    uint64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) tot += llpf_q64_unit(llpf_exp_le0(w[i] - m), K);
    const double Td = (double)tot, invTd = 1.0 / Td;
    const double s = u * (Td * invTd);
    uint64_t cum = 0;
    for (int64_t i = 0; i < n; ++i) {
        cum += llpf_q64_unit(llpf_exp_le0(w[i] - m), K);
        bins[i] = (double)cum * invTd;
        if (s <= bins[i]) return i;
    }
    return n - 1;
}

/* xf [T][N][nx], wf / wef [T][N] (forward_trajectory history), U [T][nu]; xb [T][M][nx], idx [T][M] (optional: the
 * particle index behind every smoothed sample).  Uniforms: Philox streams SMOOTH_INIT (time-T resample) and SMOOTH
 * (one per (t, m) draw) under the filter's key — the reference uses the global rand(). */
int orc_smooth(orc_filter* f, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
               const double* wef, double* xb, int64_t* idx) {
    const int64_t N = f->N;
    const int nx = f->nx;
    if (M < 1 || M > N || T < 1) return -1;                   /* @assert M <= N, src/smoothing.jl:121 */
    const int strategy = f->cfg.resampling_strategy;
    double* Ures = (double*)malloc(8 * (size_t)(M > 1 ? M : 1));
    if (strategy == LLPF_RESAMPLE_SYSTEMATIC) Ures[0] = llpf_uniform_step((uint32_t)T, LLPF_STREAM_SMOOTH_INIT, f->k0, f->k1);
    else for (int64_t i = 0; i < M; ++i) Ures[i] = llpf_uniform_idx((uint32_t)i, (uint32_t)T, LLPF_STREAM_SMOOTH_INIT, f->k0, f->k1);
    int64_t* j = (int64_t*)calloc((size_t)M, 8);
    double* bins = (double*)malloc(8 * (size_t)N);
    /* j = resample(pf.resampling_strategy, wef[:,T], M), :123 */
    int rc = orc_resample(strategy, wef + (size_t)(T - 1) * N, N, M, Ures, j, bins, f->order);
    if (rc) { free(Ures); free(j); free(bins); return rc; }
    for (int64_t m = 0; m < M; ++m) {
        memcpy(xb + ((size_t)(T - 1) * M + m) * nx, xf + ((size_t)(T - 1) * N + j[m]) * nx, 8 * (size_t)nx);
        if (idx) idx[(size_t)(T - 1) * M + m] = j[m];
    }
    double* fx = (double*)malloc(8 * (size_t)N * nx);
    double* wb = (double*)malloc(8 * (size_t)N);
    for (int64_t t = T - 2; t >= 0; --t) {
        const double ti = (double)t * f->cfg.model.Ts;        /* ti = (t-1)*pf.Ts with 1-based t, :129 */
        const double* u = U + t * f->nu;
        ORC_PAR
        for (int64_t n = 0; n < N; ++n) orc_dynamics(&f->cfg.model, xf + ((size_t)t * N + n) * nx, u, ti, fx + n * nx);
        for (int64_t m = 0; m < M; ++m) {
            const double* xn = xb + ((size_t)(t + 1) * M + m) * nx;
            for (int64_t n = 0; n < N; ++n) {                  /* wb[n] = wf[n,t] + logpdf(df, xb[m,t+1] - f(xf[n,t],u[t],p,ti)), :133-135 */
                double v[MAXD];
                for (int d = 0; d < nx; ++d) v[d] = xn[d] - fx[n * nx + d];
                wb[n] = wf[(size_t)t * N + n] + gauss_logpdf(&f->df, v);
            }
            const double ud = llpf_uniform_idx((uint32_t)m, (uint32_t)t, LLPF_STREAM_SMOOTH, f->k0, f->k1);
            const int64_t i = orc_draw_one_categorical(wb, bins, N, ud, f->order);
            memcpy(xb + ((size_t)t * M + m) * nx, xf + ((size_t)t * N + i) * nx, 8 * (size_t)nx);
            if (idx) idx[(size_t)t * M + m] = i;
        }
    }
    free(Ures); free(j); free(bins); free(fx); free(wb);
    return 0;
}

/* weighted_mean(x, we) — src/filtering.jl:541-549 */
/* dimension of a particle as the accessors see it: an RBParticle with its own covariance indexes like [xn; xl] (src/rbpf.jl:24-30) */
static int particle_dim(const orc_filter* f) { return f->rbf.on ? f->nx + f->rbf.nl : f->nx; }
int orc_particle_dim(const orc_filter* f) { return particle_dim(f); }
static void copy_particles(const orc_filter* f, double* dst) {
    if (!f->rbf.on) { memcpy(dst, f->x, 8 * (size_t)f->N * f->nx); return; }
    const int nn = f->nx, nl = f->rbf.nl, np = nn + nl;
    for (int64_t i = 0; i < f->N; ++i) {
        for (int d = 0; d < nn; ++d) dst[i * np + d] = f->x[i * nn + d];
        for (int d = 0; d < nl; ++d) dst[i * np + nn + d] = f->rbf.xl[i * nl + d];
    }
}
void orc_weighted_mean(const orc_filter* f, double* xh) {
    const int np = particle_dim(f), nn = f->nx;
    for (int d = 0; d < np; ++d) xh[d] = 0.0;
    for (int64_t i = 0; i < f->N; ++i) {
        for (int d = 0; d < nn; ++d) xh[d] += f->x[i * nn + d] * f->we[i];
        for (int d = nn; d < np; ++d) xh[d] += f->rbf.xl[i * f->rbf.nl + (d - nn)] * f->we[i];
    }
}

/* weighted_quantile(x, we, q) (src/filtering.jl:583-595): per state dimension StatsBase.quantile(v, ProbabilityWeights(we), q).  StatsBase is
 * a dependency of the reference (Project.toml: StatsBase) that /root/reference does not vendor; this restates the published algorithm of its
 * src/weights.jl `quantile(v::RealVector, w::AbstractWeights, p::RealVector)`, the branch for weights that are not FrequencyWeights:
 *   wsum = sum(w) (Julia's pairwise sum); pairs with w == 0 dropped; pairs sorted as tuples (v, w); out prefilled with the largest v;
 *   a NaN anywhere in v -> every quantile NaN; p ascending: h = p (wsum - w_1) + w_1; `while Sk <= h` advance k (past the end: return);
 *   out = v_{k-1} + (h - S_{k-1}) / (S_k - S_{k-1}) (v_k - v_{k-1}), with v_0 = S_0 = 0.
 * Pinned by: equal weights reproduce the ordinary quantile (type 7, numpy's default), which StatsBase's own tests assert; a literal Python
 * restatement (tests/test_quantile.py).  v, w: [n]; p: [np] in [0, 1]; out: [np].  Returns 0, or 1 for an empty / weightless input. */
typedef struct { double v, w; } vw_pair;
static int vw_less(const void* a, const void* b) {
    const vw_pair *x = (const vw_pair*)a, *y = (const vw_pair*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    if (signbit(x->v) != signbit(y->v)) return signbit(x->v) ? -1 : 1;      /* isless(-0.0, 0.0); NaNs never get here (tested before the sort) */
    return x->w < y->w ? -1 : (x->w > y->w ? 1 : 0);
}
int orc_weighted_quantile(const double* v, const double* w, int64_t n, const double* p, int np, double* out) {
    if (n < 1 || np < 1) return 1;
    const double wsum = orc_pairwise_sum(w, n);
    for (int64_t i = 0; i < n; ++i) if (v[i] != v[i]) { for (int k = 0; k < np; ++k) out[k] = v[i]; return 0; }      /* (StatsBase tests this after the sort: same result) */
    vw_pair* vw = (vw_pair*)malloc(sizeof(vw_pair) * (size_t)n);
    int64_t N = 0;
    for (int64_t i = 0; i < n; ++i) if (w[i] != 0.0) { vw[N].v = v[i]; vw[N].w = w[i]; ++N; }
    if (N == 0 || !(wsum > 0.0)) { free(vw); return 1; }
    qsort(vw, (size_t)N, sizeof(vw_pair), vw_less);
    int* perm = (int*)malloc(sizeof(int) * (size_t)np);
    for (int i = 0; i < np; ++i) perm[i] = i;
    for (int i = 1; i < np; ++i) { const int k = perm[i]; int j = i - 1; while (j >= 0 && p[perm[j]] > p[k]) { perm[j + 1] = perm[j]; --j; } perm[j + 1] = k; }
    for (int i = 0; i < np; ++i) out[i] = vw[N - 1].v;
    double Sk = 0.0, Skold = 0.0, vk = 0.0, vkold = 0.0;
    int64_t k = 0;
    const double w1 = vw[0].w;
    for (int i = 0; i < np; ++i) {
        const double h = p[perm[i]] * (wsum - w1) + w1;
        while (Sk <= h) {
            k += 1;
            if (k > N) { free(vw); free(perm); return 0; }
            Skold = Sk; vkold = vk;
            vk = vw[k - 1].v;
            Sk += vw[k - 1].w;
        }
        out[perm[i]] = vkold + (h - Skold) / (Sk - Skold) * (vk - vkold);
    }
    free(vw); free(perm);
    return 0;
}
/* The same quantile in DEVICE ORDER — what the engine computes (csrc/k_quantile.hip, round 6), bit for bit.  StatsBase's running sum is a
 * serial fp64 sum over the sorted particles; the device finds the crossing by a radix selection over weight histograms, so the sums are
 * integers: m_i = floor(w_i 2^sc), capped at 2^98, with the scale chosen per quantile so that h 2^sc has its leading bit at 2^64 or above
 * (sc = 96 for h >= 2^-32) — every particle with w > 0 is PRESENT (it can be the smallest / the predecessor / the largest value) even
 * when it carries no mass at that scale, which is what makes p -> 0 return the smallest value again (round 5's accessor summed at 2^-96
 * only).  The crossing test S_k > h is exact in integers; the interpolation is StatsBase's formula in fp64 with S_{k-1} the rounded exact
 * sum and S_k = fl(S_{k-1} + w_k).  Differences from StatsBase are confined to the rounding of its serial sum (a weight below 2^-53 of
 * the running sum is absorbed there and counted here).  v, w: [n]; p: [np]; out: [np]; returns 1 when nothing carries weight. */
typedef unsigned __int128 orc_u128;
static uint64_t wq_key(double x) { uint64_t u; memcpy(&u, &x, 8); return (u >> 63) ? ~u : (u | 0x8000000000000000ULL); }
static double wq_unkey(uint64_t k) { const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k; double x; memcpy(&x, &u, 8); return x; }
static orc_u128 wq_mass(double w, int sc) {            /* min(floor(w 2^sc), 2^98); 0 for subnormal w */
    uint64_t u; memcpy(&u, &w, 8);
    const int E = (int)(u >> 52) & 0x7ff;
    if (E == 0 || E == 0x7ff || (u >> 63)) return 0;
    const uint64_t M = (u & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
    const int sh = E - 1075 + sc;
    const orc_u128 cap = (orc_u128)1 << 98;
    if (sh >= 0) return sh > 45 ? cap : ((orc_u128)M << sh);      /* M < 2^53: M << 45 < 2^98 */
    return -sh >= 53 ? 0 : (orc_u128)(M >> -sh);
}
static double wq_to_double(orc_u128 a, int sc) {       /* round-to-nearest-even of a 2^-sc */
    if (a == 0) return 0.0;
    uint64_t hi = (uint64_t)(a >> 64), lo = (uint64_t)a;
    if (hi == 0) return ldexp((double)lo, -sc);
    const int z = __builtin_clzll(hi);
    uint64_t top = z ? ((hi << z) | (lo >> (64 - z))) : hi;
    const uint64_t rest = z ? (lo << z) : lo;
    if (rest) top |= 1;
    return ldexp((double)top, 64 - z - sc);
}
typedef struct { uint64_t key; double w; } wq_pair;
static int wq_less(const void* a, const void* b) {
    const wq_pair *x = (const wq_pair*)a, *y = (const wq_pair*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->w < y->w ? -1 : (x->w > y->w ? 1 : 0);
}
int orc_weighted_quantile_dev(const double* v, const double* w, int64_t n, const double* p, int np, double* out) {
    if (n < 1 || np < 1) return 1;
    for (int64_t i = 0; i < n; ++i) if (v[i] != v[i]) { for (int k = 0; k < np; ++k) out[k] = NAN; return 0; }
    wq_pair* s = (wq_pair*)malloc(sizeof(wq_pair) * (size_t)n);
    int64_t N = 0;
    orc_u128 tot96 = 0;
    for (int64_t i = 0; i < n; ++i) if (w[i] > 0.0) { s[N].key = wq_key(v[i]); s[N].w = w[i]; ++N; tot96 += wq_mass(w[i], 96); }
    if (N == 0) { free(s); return 1; }
    qsort(s, (size_t)N, sizeof(wq_pair), wq_less);
    const double w1 = s[0].w;
    double wsum = wq_to_double(tot96, 96);
    if (wsum < w1) wsum = w1;
    for (int q = 0; q < np; ++q) {
        const double d = wsum - w1;
        const double pd = p[q] * d;
        const double h = pd + w1;
        const int eh = ilogb(h);
        const int sc = eh >= -32 ? 96 : 96 + (-32 - eh);
        orc_u128 H;                                                /* h 2^sc: an integer (its lowest set bit is at 2^(eh + sc - 52) >= 2^12) */
        { uint64_t u; memcpy(&u, &h, 8); const int E = (int)(u >> 52) & 0x7ff;
          H = (orc_u128)((u & 0x000fffffffffffffULL) | 0x0010000000000000ULL) << (E - 1075 + sc); }
        orc_u128 S = 0, Slt = 0;
        int64_t k = 0;                                             /* first index of the group in which the running sum exceeds H */
        int found = 0;
        while (k < N) {
            int64_t e = k;
            orc_u128 g = 0;
            while (e < N && s[e].key == s[k].key) { g += wq_mass(s[e].w, sc); ++e; }
            if (S + g > H) { found = 1; Slt = S; break; }
            S += g; k = e;
        }
        if (!found) { out[q] = wq_unkey(s[N - 1].key); continue; }
        const double vk = wq_unkey(s[k].key);
        if (Slt + wq_mass(s[k].w, sc) > H) {                        /* the crossing is at the group's first member (the lightest of the tie) */
            const double vkold = k ? wq_unkey(s[k - 1].key) : 0.0;
            const double Skold = k ? wq_to_double(Slt, sc) : 0.0;
            const double Sk = Skold + s[k].w;
            const double den = Sk - Skold;
            out[q] = den > 0.0 ? vkold + (h - Skold) / den * (vk - vkold) : vk;
        } else {
            out[q] = vk;                                            /* inside the tie: v_{k-1} == v_k */
        }
    }
    free(s);
    return 0;
}
/* ... of the filter's current particles and exp-weights: out [np][particle_dim] */
int orc_filter_weighted_quantile(const orc_filter* f, const double* p, int np, double* out) {
    const int pd = particle_dim(f);
    double* col = (double*)malloc(8 * (size_t)f->N);
    double* x = (double*)malloc(8 * (size_t)f->N * pd);
    double* o1 = (double*)malloc(8 * (size_t)np);
    copy_particles(f, x);
    int rc = 0;
    for (int d = 0; d < pd && !rc; ++d) {
        for (int64_t i = 0; i < f->N; ++i) col[i] = x[i * pd + d];
        rc = f->order == ORC_ORDER_DEVICE ? orc_weighted_quantile_dev(col, f->we, f->N, p, np, o1) : orc_weighted_quantile(col, f->we, f->N, p, np, o1);
        for (int k = 0; k < np; ++k) out[(size_t)k * pd + d] = o1[k];
    }
    free(col); free(x); free(o1);
    return rc;
}

/* the loop of forward_trajectory (src/filtering.jl:351-363, t_index0 = 0 after reset!) and of
 * loglik (src/smoothing.jl:227-230: t = index(pf)*Ts, index = 1 after reset!, so t_index0 = 1) */
double orc_run(orc_filter* f, const double* U, const double* Y, int64_t T, double t_index0,
               double* ll_steps, double* xmean, double* x_hist, double* w_hist, double* we_hist) {
    double ll = 0.0;
    size_t N = (size_t)f->N;
    for (int64_t k = 0; k < T; ++k) {
        double ti = (t_index0 + (double)k) * f->cfg.model.Ts;
        const double* u = U + k * f->nu;
        const double* y = Y + k * f->ny;
        double lli = orc_correct(f, u, y, ti);
        ll += lli;
        if (ll_steps) ll_steps[k] = lli;
        if (xmean) orc_weighted_mean(f, xmean + k * particle_dim(f));
        if (x_hist) copy_particles(f, x_hist + (size_t)k * N * particle_dim(f));
        if (w_hist) memcpy(w_hist + (size_t)k * N, f->w, 8 * N);
        if (we_hist) memcpy(we_hist + (size_t)k * N, f->we, 8 * N);
        orc_predict(f, u, ti);
    }
    return ll;
}

int64_t orc_num_particles(const orc_filter* f) { return f->N; }
int64_t orc_index(const orc_filter* f) { return f->t; }
void orc_get_particles(const orc_filter* f, double* dst) { copy_particles(f, dst); }
void orc_get_weights(const orc_filter* f, double* dst) { memcpy(dst, f->w, 8 * (size_t)f->N); }
void orc_get_expweights(const orc_filter* f, double* dst) { memcpy(dst, f->we, 8 * (size_t)f->N); }
void orc_get_ancestors(const orc_filter* f, int64_t* dst) { memcpy(dst, f->j, 8 * (size_t)f->N); }
void orc_get_bins(const orc_filter* f, double* dst) { memcpy(dst, f->bins, 8 * (size_t)f->N); }
void orc_set_particles(orc_filter* f, const double* src) {
    if (f->rbf.on) {
        const int nn = f->nx, nl = f->rbf.nl, np = nn + nl;
        for (int64_t i = 0; i < f->N; ++i) {
            for (int d = 0; d < nn; ++d) { f->x[i * nn + d] = src[i * np + d]; f->xprev[i * nn + d] = src[i * np + d]; }
            for (int d = 0; d < nl; ++d) { f->rbf.xl[i * nl + d] = src[i * np + nn + d]; f->rbf.xlprev[i * nl + d] = src[i * np + nn + d]; }
        }
        return;
    }
    memcpy(f->x, src, 8 * (size_t)f->N * f->nx);
    memcpy(f->xprev, src, 8 * (size_t)f->N * f->nx);
}
/* install log-weights; expweights become softmax(w) in the filter's order, w itself is kept */
void orc_set_weights(orc_filter* f, const double* w) {
    int64_t n = f->N;
    memcpy(f->w, w, 8 * (size_t)n);
    if (f->order == ORC_ORDER_DEVICE) {
        dev_expsum(f->w, f->e, n, &f->dn);
        f->dn_valid = 1;
        for (int64_t i = 0; i < n; ++i) f->we[i] = f->e[i] * f->dn.inv;
        f->maxw = f->dn.m;
        f->wmax = f->dn.m;                                     /* w is kept as installed */
    } else {
        double* tmp = (double*)malloc(8 * (size_t)n);
        memcpy(tmp, w, 8 * (size_t)n);
        orc_expnormalize(f->we, tmp, n);
        free(tmp);
    }
}
void orc_set_index(orc_filter* f, int64_t t) { f->t = t; }
int orc_last_resampled(const orc_filter* f) { return f->last_resampled; }
double orc_maxw(const orc_filter* f) { return f->maxw; }
int64_t orc_resample_count(const orc_filter* f) { return f->resample_count; }
int orc_degenerate(const orc_filter* f) { return f->degenerate; }
int64_t orc_exact_steps(const orc_filter* f) { return f->n_exact_steps; }

double orc_gauss_logpdf(const llpf_gaussian* g, const double* x) {
    gaussd d;
    if (gauss_prepare(g, &d, ORC_ORDER_REFERENCE)) return NAN;
    return gauss_logpdf(&d, x);
}
void orc_gauss_sample(const llpf_gaussian* g, const double* xi, double* out) {
    gaussd d;
    if (gauss_prepare(g, &d, ORC_ORDER_REFERENCE)) return;
    gauss_sample(&d, xi, out);
}

/* ------------------------------------------------------------------------------------------
 * Closed-form Kalman log-likelihood: independent check for linear-Gaussian inputs.
 * reset! src/kalman.jl:159-164; predict! src/filtering.jl:52-74; correct! :100-128; loop :293-314
 * ---------------------------------------------------------------------------------------- */
static void gauss_cov_full(const llpf_gaussian* g, double* S) {
    int n = g->dim;
    for (int i = 0; i < n * n; ++i) S[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        if (g->kind == LLPF_COV_SCAL) S[i * n + i] = g->cov[0];
        else if (g->kind == LLPF_COV_DIAG) S[i * n + i] = g->cov[i];
    }
    if (g->kind == LLPF_COV_FULL) for (int i = 0; i < n * n; ++i) S[i] = g->cov[i];
}

double orc_kalman_loglik(const llpf_model* m, const double* U, const double* Y, int64_t T) {
    int nx = m->nx, nu = m->nu, ny = m->ny;
    double R1[MAXD * MAXD], R2[MAXD * MAXD], P[MAXD * MAXD], x[MAXD];
    gauss_cov_full(&m->dynamics_density, R1);
    gauss_cov_full(&m->measurement_density, R2);
    gauss_cov_full(&m->initial_density, P);
    for (int i = 0; i < nx; ++i) x[i] = m->initial_density.mu[i];
    double LL = 0.0;
    for (int64_t k = 0; k < T; ++k) {
        const double* u = U + k * nu;
        const double* y = Y + k * ny;
        if (y[0] == y[0]) {
            /* e = y - C x ; S = sym(C P C') + R2 */
            double e[MAXD], CP[MAXD * MAXD], S[MAXD * MAXD], Lc[MAXD * MAXD];
            for (int r = 0; r < ny; ++r) {
                double cx = 0.0;
                for (int c = 0; c < nx; ++c) cx += m->C[r * nx + c] * x[c];
                e[r] = y[r] - cx - m->measurement_density.mu[r];
            }
            for (int r = 0; r < ny; ++r)
                for (int c = 0; c < nx; ++c) {
                    double a = 0.0;
                    for (int q = 0; q < nx; ++q) a += m->C[r * nx + q] * P[q * nx + c];
                    CP[r * nx + c] = a;
                }
            for (int r = 0; r < ny; ++r)
                for (int c = 0; c < ny; ++c) {
                    double a = 0.0;
                    for (int q = 0; q < nx; ++q) a += CP[r * nx + q] * m->C[c * nx + q];
                    S[r * ny + c] = a;
                }
            for (int r = 0; r < ny; ++r)
                for (int c = r + 1; c < ny; ++c) {
                    double a = 0.5 * (S[r * ny + c] + S[c * ny + r]);
                    S[r * ny + c] = a; S[c * ny + r] = a;
                }
            for (int i = 0; i < ny * ny; ++i) S[i] += R2[i];
            if (chol_lower(S, ny, Lc)) return NAN;
            /* Sinv e via the Cholesky factor, logdet S */
            double z[MAXD], z2[MAXD], logdet = 0.0;
            for (int i = 0; i < ny; ++i) {
                double acc = e[i];
                for (int q = 0; q < i; ++q) acc -= Lc[i * MAXD + q] * z[q];
                z[i] = acc / Lc[i * MAXD + i];
                logdet += 2.0 * log(Lc[i * MAXD + i]);
            }
            for (int i = ny - 1; i >= 0; --i) {
                double acc = z[i];
                for (int q = i + 1; q < ny; ++q) acc -= Lc[q * MAXD + i] * z2[q];
                z2[i] = acc / Lc[i * MAXD + i];
            }
            double quad = 0.0;
            for (int i = 0; i < ny; ++i) quad += e[i] * z2[i];
            LL += -((double)ny * log(2.0 * 3.141592653589793) + logdet) / 2.0 - quad / 2.0;
            /* K = P C' Sinv ; x += K e ; P = sym((I - K C) P) */
            double K[MAXD * MAXD], KC[MAXD * MAXD], Pn[MAXD * MAXD];
            for (int r = 0; r < nx; ++r) {
                /* row r of P C' is CP[:, r]; solve S k' = (P C')[r,:]' */
                double b[MAXD], t1[MAXD], t2[MAXD];
                for (int c = 0; c < ny; ++c) b[c] = CP[c * nx + r];
                for (int i = 0; i < ny; ++i) {
                    double acc = b[i];
                    for (int q = 0; q < i; ++q) acc -= Lc[i * MAXD + q] * t1[q];
                    t1[i] = acc / Lc[i * MAXD + i];
                }
                for (int i = ny - 1; i >= 0; --i) {
                    double acc = t1[i];
                    for (int q = i + 1; q < ny; ++q) acc -= Lc[q * MAXD + i] * t2[q];
                    t2[i] = acc / Lc[i * MAXD + i];
                }
                for (int c = 0; c < ny; ++c) K[r * ny + c] = t2[c];
            }
            for (int r = 0; r < nx; ++r) {
                double a = 0.0;
                for (int c = 0; c < ny; ++c) a += K[r * ny + c] * e[c];
                x[r] += a;
            }
            for (int r = 0; r < nx; ++r)
                for (int c = 0; c < nx; ++c) {
                    double a = 0.0;
                    for (int q = 0; q < ny; ++q) a += K[r * ny + q] * m->C[q * nx + c];
                    KC[r * nx + c] = (r == c ? 1.0 : 0.0) - a;
                }
            for (int r = 0; r < nx; ++r)
                for (int c = 0; c < nx; ++c) {
                    double a = 0.0;
                    for (int q = 0; q < nx; ++q) a += KC[r * nx + q] * P[q * nx + c];
                    Pn[r * nx + c] = a;
                }
            for (int r = 0; r < nx; ++r)
                for (int c = 0; c < nx; ++c) P[r * nx + c] = 0.5 * (Pn[r * nx + c] + Pn[c * nx + r]);
        }
        /* x = A x + B u (+ mean of df) ; P = sym(A P A') + R1 */
        double xn[MAXD], AP[MAXD * MAXD], Pn[MAXD * MAXD];
        for (int r = 0; r < nx; ++r) {
            double a = 0.0;
            for (int c = 0; c < nx; ++c) a += m->A[r * nx + c] * x[c];
            for (int c = 0; c < nu; ++c) a += m->B[r * nu + c] * u[c];
            xn[r] = a + m->dynamics_density.mu[r];
        }
        for (int r = 0; r < nx; ++r) x[r] = xn[r];
        for (int r = 0; r < nx; ++r)
            for (int c = 0; c < nx; ++c) {
                double a = 0.0;
                for (int q = 0; q < nx; ++q) a += m->A[r * nx + q] * P[q * nx + c];
                AP[r * nx + c] = a;
            }
        for (int r = 0; r < nx; ++r)
            for (int c = 0; c < nx; ++c) {
                double a = 0.0;
                for (int q = 0; q < nx; ++q) a += AP[r * nx + q] * m->A[c * nx + q];
                Pn[r * nx + c] = a;
            }
        for (int r = 0; r < nx; ++r)
            for (int c = 0; c < nx; ++c) P[r * nx + c] = 0.5 * (Pn[r * nx + c] + Pn[c * nx + r]) + R1[r * nx + c];
    }
    return LL;
}

/* ------------------------------------------------------------------------------------------
 * Probes of the shared primitive headers (host evaluation)
 * ---------------------------------------------------------------------------------------- */
void orc_math_vec(int which, const double* in, double* out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        double x = in[i], s, c;
        switch (which) {
            case 0: out[i] = llpf_exp(x); break;
            case 1: out[i] = llpf_log(x); break;
            case 2: out[i] = llpf_log1p_nonneg(x); break;
            case 3: llpf_sincos2pi(x, &s, &c); out[i] = s; break;
            case 4: llpf_sincos2pi(x, &s, &c); out[i] = c; break;
            case 5: out[i] = llpf_sqrt(x); break;
            case 12: out[i] = llpf_sqrt_pos(x); break;
            case 6: out[i] = 1.0 / x; break;
            case 7: out[i] = (double)llpf_d2u(x); break;
            case 8: out[i] = llpf_exp_le0(x); break;
            case 9: out[i] = llpf_log_unit(x); break;
            case 10: llpf_sincos2pi_fast(x, &s, &c); out[i] = s; break;
            case 11: llpf_sincos2pi_fast(x, &s, &c); out[i] = c; break;
            default: out[i] = NAN;
        }
    }
}
void orc_philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
    llpf_philox4 r = llpf_philox4x32_10(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
}
/* the generator the engine's draws actually go through (LLPF_PHILOX_ROUNDS rounds: 7 since round 4); returns the round count */
int orc_philox_block_engine(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
    llpf_philox4 r = llpf_philox4x32(c0, c1, c2, c3, k0, k1);
    for (int i = 0; i < 4; ++i) out4[i] = r.v[i];
    return LLPF_PHILOX_ROUNDS;
}
void orc_normals(uint64_t seed, uint32_t step, uint32_t stream, int nd, double* out, int64_t n) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t i = 0; i < n; ++i) llpf_normals((uint32_t)i, step, stream, k0, k1, nd, out + i * nd);
}
void orc_uniforms_nd(uint64_t seed, uint32_t step, uint32_t stream, int nd, double* out, int64_t n) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t i = 0; i < n; ++i) llpf_uniforms((uint32_t)i, step, stream, k0, k1, nd, out + i * nd);
}
void orc_fix96(double e, uint64_t* lo_hi) {
    llpf_u128 r = llpf_fix96(e);
    lo_hi[0] = r.lo; lo_hi[1] = r.hi;
}
uint64_t orc_q64(double e, int K) { return llpf_q64(e, K); }
void orc_fix96_unit(double e, uint64_t* lo_hi) {
    llpf_u128 r = llpf_fix96_unit(e);
    lo_hi[0] = r.lo; lo_hi[1] = r.hi;
}
uint64_t orc_q64_unit(double e, int K) { return llpf_q64_unit(e, K); }
double orc_u128_to_double(uint64_t lo, uint64_t hi) {
    llpf_u128 a; a.lo = lo; a.hi = hi;
    return llpf_u128_to_double(a);
}

