// kernels.hip — hand-written gfx950 kernels of the particle-filter step.
//
// One filter step (correct! then predict!, reference src/filtering.jl:164-168, 140-153) is
//   k_norm      : max-reduce of the block maxima, e_i = exp(w_i - m), fixed-point sums of e, e^2 and of the
//                 resampling quanta per 2048-particle tile                     (logsumexp!, utils.jl:18-27;
//                                                                               effective_particles, resample.jl:1-2)
//   k_finalize  : one block per filter sums the tile partials -> log1p(s), 1/(s+1), ll, ESS, resample decision
//   k_resample  : per tile: integer inclusive scan of the quanta (tile prefix comes from k_norm's partials, so no
//                 look-back / spinning), bins = cum/total, ancestor COUNTS c(bins) for the systematic / stratified
//                 thresholds, then expansion of the counts into ancestor indices through LDS
//                                                                              (resample, resample.jl:17-61)
//   k_step      : gather x[anc[i]] -> dynamics -> + Philox/Box–Muller process noise -> store x (SoA, 16-B vectors)
//                 -> w = w_prev + logpdf(y_next - g(x)) -> block max            (propagate_particles!, PFtypes.jl:122-139;
//                                                                               measurement_equation!, :107-120)
// All particle data is fp64 structure-of-arrays; wave64; 256-thread workgroups; grid = (tiles, filters).
// Compiled with -ffp-contract=off: the arithmetic is the same IEEE sequence as oracle/llpf_oracle.c (device order).
#include "engine.hpp"

namespace llpf {

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------
DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = llpf_fmax(v, __shfl_xor(v, o, 64));
    return v;
}
DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((unsigned long long)v, o, 64);
    return v;
}
DEV llpf_u128 wave_sum_u128(llpf_u128 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        llpf_u128 t;
        t.lo = (uint64_t)__shfl_xor((unsigned long long)v.lo, o, 64);
        t.hi = (uint64_t)__shfl_xor((unsigned long long)v.hi, o, 64);
        v = llpf_u128_add(v, t);
    }
    return v;
}
DEV double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}

DEV double block_max(double v, double* sm /* [4] */) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    double r = sm[0];
#pragma unroll
    for (int k = 1; k < BLOCK / 64; ++k) r = llpf_fmax(r, sm[k]);
    return r;
}

// ------------------------------------------------------------------------------------------------
// Gaussian pieces — operation order identical to oracle/llpf_oracle.c gauss_sample / gauss_logpdf
// (reference src/utils.jl:110-113, 252-268)
// ------------------------------------------------------------------------------------------------
template <int ND>
DEV void gauss_sample(const GaussD& g, const double* xi, double* out) {
    const int kind = g.kind;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        double v;
        if (kind == LLPF_COV_SCAL) v = g.sqrtscal * xi[i];
        else if (kind == LLPF_COV_DIAG) v = g.sqrtdiag[i] * xi[i];
        else {
            v = g.L[i * MAXD + 0] * xi[0];
#pragma unroll
            for (int j = 1; j <= i; ++j) v = v + g.L[i * MAXD + j] * xi[j];
        }
        out[i] = v + g.mu[i];
    }
}

template <int ND>
DEV double gauss_logpdf(const GaussD& g, const double* x) {
    double d[ND], q;
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = x[i] - g.mu[i];
    const int kind = g.kind;
    if (kind == LLPF_COV_SCAL) {
        double dot = d[0] * d[0];
#pragma unroll
        for (int i = 1; i < ND; ++i) dot = dot + d[i] * d[i];
        q = dot / g.scal;
    } else if (kind == LLPF_COV_DIAG) {
        double s = (d[0] * d[0]) * g.invdiag[0];
#pragma unroll
        for (int i = 1; i < ND; ++i) s = s + (d[i] * d[i]) * g.invdiag[i];
        q = s;
    } else {
        double z[ND], z2[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            double acc = d[i];
#pragma unroll
            for (int j = 0; j < i; ++j) acc = acc - g.L[i * MAXD + j] * z[j];
            z[i] = acc / g.L[i * MAXD + i];
        }
#pragma unroll
        for (int i = ND - 1; i >= 0; --i) {
            double acc = z[i];
#pragma unroll
            for (int j = i + 1; j < ND; ++j) acc = acc - g.L[j * MAXD + i] * z2[j];
            z2[i] = acc / g.L[i * MAXD + i];
        }
        double dot = d[0] * z2[0];
#pragma unroll
        for (int i = 1; i < ND; ++i) dot = dot + d[i] * z2[i];
        q = dot;
    }
    return g.c0 - q / 2.0;
}

// ------------------------------------------------------------------------------------------------
// Models.  A model is a struct with
//   prepare(md, u, t)      once per thread (particle-independent terms)
//   dynamics(x, out)       f(x,u,p,t) without noise
//   measurement(x, out)    g(x,u,p,t)
// ------------------------------------------------------------------------------------------------
template <int NX, int NY>
struct LinGauss {   // f = A x .+ B u ; g = C x   (reference examples/example_lineargaussian.jl:28-29)
    const ModelD* md;
    double bu[NX];
    bool has_u;
    DEV void prepare(const ModelD* m, const double* __restrict__ u, double /*t*/) {
        md = m;
        const int nu = m->nu;
        has_u = nu > 0 && u != nullptr;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double acc = 0.0;
            if (has_u) {
                acc = m->B[r * nu + 0] * u[0];
                for (int c = 1; c < nu; ++c) acc = acc + m->B[r * nu + c] * u[c];
            }
            bu[r] = acc;
        }
    }
    DEV void dynamics(const double* x, double* out) const {
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double ax = md->A[r * NX + 0] * x[0];
#pragma unroll
            for (int c = 1; c < NX; ++c) ax = ax + md->A[r * NX + c] * x[c];
            out[r] = has_u ? ax + bu[r] : ax;
        }
    }
    DEV void measurement(const double* x, double* out) const {
#pragma unroll
        for (int r = 0; r < NY; ++r) {
            double cx = md->C[r * NX + 0] * x[0];
#pragma unroll
            for (int c = 1; c < NX; ++c) cx = cx + md->C[r * NX + c] * x[c];
            out[r] = cx;
        }
    }
};

template <int NX, int NY>
struct QuadTank {   // reference examples/example_quadtank.jl:8-35 with rk4 of src/utils.jl:220-237
    static_assert(NX == 4 && NY == 2, "quad-tank is 4 states / 2 outputs");
    // coefficients in the reference's evaluation order: (-a/A), (a/A), (gamma k / A)
    double c1a, c1a_sw, c1b, c1u, c2a, c2b, c2u, c3a, c3u, c4a, c4u;
    double tg, eps, tsw, u0, u1, t0, Ts;
    int ss;
    DEV void prepare(const ModelD* m, const double* __restrict__ u, double t) {
        const double* q = m->qt;
        const double k1 = q[LLPF_QT_K1], k2 = q[LLPF_QT_K2], g = q[LLPF_QT_G];
        const double A1 = q[LLPF_QT_A1], A2 = q[LLPF_QT_A2], A3 = q[LLPF_QT_A3], A4 = q[LLPF_QT_A4];
        const double a1 = q[LLPF_QT_a1], a2 = q[LLPF_QT_a2], a3 = q[LLPF_QT_a3], a4 = q[LLPF_QT_a4];
        const double g1 = q[LLPF_QT_GAMMA1], g2 = q[LLPF_QT_GAMMA2];
        c1a = (-a1) / A1;
        c1a_sw = (-(a1 * q[LLPF_QT_A1FACTOR])) / A1;
        c1b = a3 / A1;
        c1u = (g1 * k1) / A1;
        c2a = (-a2) / A2;
        c2b = a4 / A2;
        c2u = (g2 * k2) / A2;
        c3a = (-a3) / A3;
        c3u = ((1.0 - g2) * k2) / A3;
        c4a = (-a4) / A4;
        c4u = ((1.0 - g1) * k1) / A4;
        tg = 2.0 * g;
        eps = q[LLPF_QT_EPS];
        tsw = q[LLPF_QT_TSWITCH];
        u0 = u[0];
        u1 = u[1];
        t0 = t;
        ss = m->supersample < 1 ? 1 : m->supersample;
        Ts = m->Ts / (double)ss;
    }
    DEV void rhs(const double* h, double t, double* xd) const {
        double s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = tg * h[i];
            s[i] = llpf_sqrt((v > 0.0 ? v : 0.0) + eps);
        }
        const double ca = (t > tsw) ? c1a_sw : c1a;
        xd[0] = ca * s[0] + c1b * s[2] + c1u * u0;
        xd[1] = c2a * s[1] + c2b * s[3] + c2u * u1;
        xd[2] = c3a * s[2] + c3u * u1;
        xd[3] = c4a * s[3] + c4u * u0;
    }
    DEV void dynamics(const double* x0, double* out) const {
        double x[4], f1[4], f2[4], f3[4], f4[4], xt[4];
        double t = t0;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = x0[i];
        for (int it = 0; it < ss; ++it) {
            rhs(x, t, f1);
#pragma unroll
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + (Ts / 2.0) * f1[i];
            rhs(xt, t + Ts / 2.0, f2);
#pragma unroll
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + (Ts / 2.0) * f2[i];
            rhs(xt, t + Ts / 2.0, f3);
#pragma unroll
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + Ts * f3[i];
            rhs(xt, t + Ts, f4);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = x[i] + (Ts / 6.0) * (((f1[i] + 2.0 * f2[i]) + 2.0 * f3[i]) + f4[i]);
            t = t + Ts;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    DEV void measurement(const double* x, double* out) const {
        out[0] = x[0];
        out[1] = x[1];
    }
};

// ------------------------------------------------------------------------------------------------
// k_init — reset!: x_i = mu0 + L0 xi_i  (reference src/filtering.jl:4-14, src/PFtypes.jl:66)
// ------------------------------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(BLOCK) void k_init(BankDev b, const ModelD* __restrict__ models,
                                                 const FilterScal* __restrict__ scal, uint32_t step, int init_anc) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    const ModelD* md = models + f;
    double xi[NX], x0[NX];
    llpf_normals((uint32_t)i, step, LLPF_STREAM_INIT, scal[f].k0, scal[f].k1, NX, xi);
    gauss_sample<NX>(md->d0, xi, x0);
    double* xc = b.xcur + (size_t)f * NX * b.Ns;
#pragma unroll
    for (int d = 0; d < NX; ++d) xc[(size_t)d * b.Ns + i] = x0[d];
    b.w[(size_t)f * b.Ns + i] = -LLPF_INF;
    if (init_anc) b.anc[(size_t)f * b.Ns + i] = (i < b.N) ? (int32_t)i : 0;
}

// ------------------------------------------------------------------------------------------------
// k_step — fused propagate + weight + block max
// ------------------------------------------------------------------------------------------------
template <class Model, int NX, int NY, int MODE>
__global__ __launch_bounds__(BLOCK) void k_step(BankDev b, const ModelD* __restrict__ models,
                                                 const FilterScal* __restrict__ scal, StepArgs a) {
    __shared__ double sm_max[BLOCK / 64];
    const int f = blockIdx.y;
    const ModelD* md = models + f;
    const FilterScal* sc = scal + f;
    const int do_res = (MODE != MODE_WEIGHT) ? sc->do_resample : 0;
    const int uniform = sc->uniform, pend = sc->norm_pending;
    const double m = sc->m, l = sc->l, wconst = sc->wconst;
    const uint32_t k0 = sc->k0, k1 = sc->k1;
    const int64_t Ns = b.Ns, N = b.N;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * Ns;
    double* __restrict__ xn = b.xnext + (size_t)f * NX * Ns;
    double* w = b.w + (size_t)f * Ns;
    const int32_t* __restrict__ anc = b.anc + (size_t)f * Ns;

    Model model;
    model.prepare(md, a.u, a.t_prop);
    double y[NY];
    if (MODE != MODE_PROP) {
#pragma unroll
        for (int k = 0; k < NY; ++k) y[k] = a.has_y ? a.y[k] : 0.0;
    }

    double bmax = -LLPF_INF;
#pragma unroll 1
    for (int it = 0; it < STEP_ITERS; ++it) {
        const int64_t i0 = ((int64_t)blockIdx.x * STEP_ITERS + it) * (BLOCK * STEP_PPT) + (int64_t)threadIdx.x * STEP_PPT;
        double xs[STEP_PPT][NX];
        if (MODE != MODE_WEIGHT) {
            double xp[STEP_PPT][NX];
            if (do_res) {
                const int2 av = *reinterpret_cast<const int2*>(anc + i0);
#pragma unroll
                for (int d = 0; d < NX; ++d) {
                    xp[0][d] = xc[(size_t)d * Ns + av.x];
                    xp[1][d] = xc[(size_t)d * Ns + av.y];
                }
            } else {
#pragma unroll
                for (int d = 0; d < NX; ++d) {
                    const double2 v = *reinterpret_cast<const double2*>(xc + (size_t)d * Ns + i0);
                    xp[0][d] = v.x;
                    xp[1][d] = v.y;
                }
            }
#pragma unroll
            for (int p = 0; p < STEP_PPT; ++p) {
                double fx[NX], xi[NX], nz[NX];
                model.dynamics(xp[p], fx);
                llpf_normals((uint32_t)(i0 + p), a.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
                gauss_sample<NX>(md->df, xi, nz);
#pragma unroll
                for (int d = 0; d < NX; ++d) xs[p][d] = fx[d] + nz[d];
            }
#pragma unroll
            for (int d = 0; d < NX; ++d) {
                double2 v;
                v.x = xs[0][d];
                v.y = xs[1][d];
                *reinterpret_cast<double2*>(xn + (size_t)d * Ns + i0) = v;
            }
        } else {
#pragma unroll
            for (int d = 0; d < NX; ++d) {
                const double2 v = *reinterpret_cast<const double2*>(xc + (size_t)d * Ns + i0);
                xs[0][d] = v.x;
                xs[1][d] = v.y;
            }
        }
        if (MODE != MODE_PROP) {
            double wp[STEP_PPT];
            if (do_res) {                          // reset_weights!: w = log(1/N)
                wp[0] = b.log1N;
                wp[1] = b.log1N;
            } else if (uniform) {
                wp[0] = wconst;
                wp[1] = wconst;
            } else {
                const double2 wv = *reinterpret_cast<const double2*>(w + i0);
                wp[0] = pend ? (wv.x - m) - l : wv.x;  // lazy w .-= offset ; w .-= log1p(s)
                wp[1] = pend ? (wv.y - m) - l : wv.y;
            }
            double wn[STEP_PPT];
#pragma unroll
            for (int p = 0; p < STEP_PPT; ++p) {
                double wv = wp[p];
                if (a.has_y) {
                    double g[NY], v[NY];
                    model.measurement(xs[p], g);
#pragma unroll
                    for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                    wv = wv + gauss_logpdf<NY>(md->dg, v);
                }
                if (i0 + p >= N) wv = -LLPF_INF;   // padding lanes carry zero weight
                wn[p] = wv;
                bmax = llpf_fmax(bmax, wv);
            }
            double2 wo;
            wo.x = wn[0];
            wo.y = wn[1];
            *reinterpret_cast<double2*>(w + i0) = wo;
        }
    }
    if (MODE != MODE_PROP) {
        const double r = block_max(bmax, sm_max);
        if (threadIdx.x == 0) b.pmax[(size_t)f * b.P1 + blockIdx.x] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// k_max — block maxima of the raw log-weights (used when no weighting kernel produced them:
// llpf_set_weights, llpf_logsumexp)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_max(BankDev b) {
    __shared__ double sm_max[BLOCK / 64];
    const int f = blockIdx.y;
    const double* w = b.w + (size_t)f * b.Ns;
    double bmax = -LLPF_INF;
#pragma unroll
    for (int it = 0; it < STEP_ITERS; ++it) {
        const int64_t i0 = ((int64_t)blockIdx.x * STEP_ITERS + it) * (BLOCK * STEP_PPT) + (int64_t)threadIdx.x * STEP_PPT;
        const double2 wv = *reinterpret_cast<const double2*>(w + i0);
        if (i0 < b.N) bmax = llpf_fmax(bmax, wv.x);
        if (i0 + 1 < b.N) bmax = llpf_fmax(bmax, wv.y);
    }
    const double r = block_max(bmax, sm_max);
    if (threadIdx.x == 0) b.pmax[(size_t)f * b.P1 + blockIdx.x] = r;
}

// max over the per-block maxima of one filter (every block of a consumer kernel recomputes it)
DEV double reduce_pmax(const double* __restrict__ pm, int P1, double* sm) {
    double v = -LLPF_INF;
    for (int k = threadIdx.x; k < P1; k += BLOCK) v = llpf_fmax(v, pm[k]);
    return block_max(v, sm);
}

// ------------------------------------------------------------------------------------------------
// k_norm — exp-weights and their exact sums per tile  (logsumexp! utils.jl:18-27, sum_all_but :66-71,
// effective_particles resample.jl:1-2; optional weighted_mean filtering.jl:541-549)
// ------------------------------------------------------------------------------------------------
template <int NX, bool XMEAN>
__global__ __launch_bounds__(BLOCK) void k_norm(BankDev b, int K) {
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_u[BLOCK / 64][6];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const double m = reduce_pmax(b.pmax + (size_t)f * b.P1, b.P1, sm_max);
    const double* __restrict__ w = b.w + (size_t)f * b.Ns;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * b.Ns;

    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0, bad = 0;
    double xm[NX > 0 ? NX : 1];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;

    double2 wv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
        wv[k] = *reinterpret_cast<const double2*>(w + i0);
    }
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
        const double e0 = llpf_exp(wv[k].x - m);
        const double e1 = llpf_exp(wv[k].y - m);
        bad += (e0 != e0) ? 1u : 0u;
        bad += (e1 != e1) ? 1u : 0u;
        S = llpf_u128_add(S, llpf_fix96(e0));
        S = llpf_u128_add(S, llpf_fix96(e1));
        E2 = llpf_u128_add(E2, llpf_fix96(e0 * e0));
        E2 = llpf_u128_add(E2, llpf_fix96(e1 * e1));
        Q += llpf_q64(e0, K);
        Q += llpf_q64(e1, K);
        if (XMEAN) {
#pragma unroll
            for (int d = 0; d < NX; ++d) {
                const double2 xv = *reinterpret_cast<const double2*>(xc + (size_t)d * b.Ns + i0);
                xm[d] = xm[d] + xv.x * e0;
                xm[d] = xm[d] + xv.y * e1;
            }
        }
    }
    S = wave_sum_u128(S);
    E2 = wave_sum_u128(E2);
    Q = wave_sum_u64(Q);
    bad = wave_sum_u64(bad);
    if (XMEAN) {
#pragma unroll
        for (int d = 0; d < NX; ++d) xm[d] = wave_sum_f64(xm[d]);
    }
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    if (lane == 0) {
        sm_u[wvid][0] = S.lo; sm_u[wvid][1] = S.hi;
        sm_u[wvid][2] = E2.lo; sm_u[wvid][3] = E2.hi;
        sm_u[wvid][4] = Q; sm_u[wvid][5] = bad;
        if (XMEAN) {
#pragma unroll
            for (int d = 0; d < NX; ++d) sm_x[wvid][d] = xm[d];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        llpf_u128 s = {sm_u[0][0], sm_u[0][1]}, e2 = {sm_u[0][2], sm_u[0][3]};
        uint64_t q = sm_u[0][4], bd = sm_u[0][5];
        for (int k = 1; k < BLOCK / 64; ++k) {
            llpf_u128 t1 = {sm_u[k][0], sm_u[k][1]}, t2 = {sm_u[k][2], sm_u[k][3]};
            s = llpf_u128_add(s, t1);
            e2 = llpf_u128_add(e2, t2);
            q += sm_u[k][4];
            bd += sm_u[k][5];
        }
        NormPartial* o = b.part + (size_t)f * b.P2 + tile;
        o->S_lo = s.lo; o->S_hi = s.hi; o->E2_lo = e2.lo; o->E2_hi = e2.hi; o->Q = q; o->bad = bd;
        if (XMEAN) {
            for (int d = 0; d < NX; ++d) {
                double acc = sm_x[0][d];
                for (int k = 1; k < BLOCK / 64; ++k) acc = acc + sm_x[k][d];
                o->xm[d] = acc;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_finalize — per filter: scalars of logsumexp!, ESS, shouldresample (resample.jl:5-10)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_finalize(BankDev b, FinalizeArgs a, int K, int nx_out, int after_predict) {
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_u[BLOCK / 64][6];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.x;
    const double m = reduce_pmax(b.pmax + (size_t)f * b.P1, b.P1, sm_max);
    const NormPartial* __restrict__ part = b.part + (size_t)f * b.P2;
    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0, bad = 0;
    double xm[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) xm[d] = 0.0;
    for (int p = threadIdx.x; p < b.P2; p += BLOCK) {
        llpf_u128 t1 = {part[p].S_lo, part[p].S_hi}, t2 = {part[p].E2_lo, part[p].E2_hi};
        S = llpf_u128_add(S, t1);
        E2 = llpf_u128_add(E2, t2);
        Q += part[p].Q;
        bad += part[p].bad;
        if (a.xmean) {
#pragma unroll
            for (int d = 0; d < MAXD; ++d)
                if (d < nx_out) xm[d] = xm[d] + part[p].xm[d];
        }
    }
    S = wave_sum_u128(S);
    E2 = wave_sum_u128(E2);
    Q = wave_sum_u64(Q);
    bad = wave_sum_u64(bad);
    if (a.xmean) {
#pragma unroll
        for (int d = 0; d < MAXD; ++d) xm[d] = wave_sum_f64(xm[d]);
    }
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    if (lane == 0) {
        sm_u[wvid][0] = S.lo; sm_u[wvid][1] = S.hi;
        sm_u[wvid][2] = E2.lo; sm_u[wvid][3] = E2.hi;
        sm_u[wvid][4] = Q; sm_u[wvid][5] = bad;
#pragma unroll
        for (int d = 0; d < MAXD; ++d) sm_x[wvid][d] = xm[d];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    llpf_u128 s128 = {sm_u[0][0], sm_u[0][1]}, e128 = {sm_u[0][2], sm_u[0][3]};
    uint64_t q = sm_u[0][4], bd = sm_u[0][5];
    for (int k = 1; k < BLOCK / 64; ++k) {
        llpf_u128 t1 = {sm_u[k][0], sm_u[k][1]}, t2 = {sm_u[k][2], sm_u[k][3]};
        s128 = llpf_u128_add(s128, t1);
        e128 = llpf_u128_add(e128, t2);
        q += sm_u[k][4];
        bd += sm_u[k][5];
    }
    FilterScal* sc = b.scal + f;
    if (after_predict) {   // bookkeeping of the predict! that ran since the last finalize
        const int r = sc->do_resample;
        sc->anc_ident = r ? 0 : 1;
        sc->last_resampled = r;
        sc->resample_count += r;
    }
    double s, l, inv, ll, ess, e2;
    int status = 0;
    if (bd != 0 || s128.hi < ((uint64_t)1 << 32)) {   // max is -Inf / NaN, or NaN weights: degenerate
        s = llpf_u2d(0x7ff8000000000000ULL);
        l = s; inv = s; ll = s; ess = s; e2 = s;
        status = LLPF_ERR_DEGENERATE;
    } else {
        s = llpf_fix96_to_double(llpf_fix96_minus_one(s128));   // sum_all_but: exact, one rounding
        l = llpf_log1p_nonneg(s);
        inv = 1.0 / (s + 1.0);
        ll = l + m;
        e2 = llpf_fix96_to_double(e128);
        ess = 1.0 / (e2 * (inv * inv));
    }
    sc->m = m; sc->s = s; sc->l = l; sc->inv = inv; sc->ll = ll; sc->ess = ess; sc->e2 = e2;
    sc->totQ = q;
    sc->K = K;
    sc->uniform = 0;
    sc->norm_pending = a.keep_norm ? 0 : 1;
    if (status) sc->status = status;
    int dr = 0;
    if (!status) dr = (b.thr == 1.0) ? 1 : (ess < (double)b.N * b.thr ? 1 : 0);
    sc->do_resample = dr;
    if (a.accumulate) sc->ll_total = sc->ll_total + ll;
    if (a.ll_steps) a.ll_steps[(size_t)a.k * b.F + f] = ll;
    if (a.xmean) {
        for (int d = 0; d < nx_out; ++d) {
            double acc = sm_x[0][d];
            for (int k = 1; k < BLOCK / 64; ++k) acc = acc + sm_x[k][d];
            a.xmean[((size_t)a.k * b.F + f) * nx_out + d] = acc * inv;
        }
    }
}

// shouldresample for a predict! that is not preceded by a correct! in the same launch sequence
__global__ void k_decide(BankDev b, int K) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= b.F) return;
    FilterScal* sc = b.scal + f;
    if (sc->status) { sc->do_resample = 0; return; }
    double ess;
    if (sc->uniform) {
        const double wev = 1.0 / (double)b.N;
        ess = 1.0 / ((double)b.N * (wev * wev));
        sc->ess = ess;
        sc->K = K;
    } else {
        ess = sc->ess;
    }
    sc->do_resample = (b.thr == 1.0) ? 1 : (ess < (double)b.N * b.thr ? 1 : 0);
}

// bookkeeping after a propagate-only predict! (reset_weights! when it resampled)
__global__ void k_post_predict(BankDev b) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= b.F) return;
    FilterScal* sc = b.scal + f;
    const int r = sc->do_resample;
    if (r) {
        sc->uniform = 1;
        sc->wconst = b.log1N;
        sc->m = 0.0;                 // maxw[] = 0, reference src/utils.jl:77
        sc->norm_pending = 0;
    }
    sc->anc_ident = r ? 0 : 1;
    sc->last_resampled = r;
    sc->resample_count += r;
    sc->do_resample = 0;
}

// ------------------------------------------------------------------------------------------------
// k_resample — scan + ancestor counts + expansion, one tile per block
// ------------------------------------------------------------------------------------------------
enum { SRC_FILTER = 0, SRC_VALUES = 1 };

struct ThrSys {   // systematic thresholds: s[i] = fl(r + fl(i0 * (1/M)))  (resample.jl:23-24)
    double r, step, Md;
    int64_t M;
    DEV double at(int64_t i0) const { return r + (double)i0 * step; }
    DEV int64_t estimate(double v) const {
        double e = (v - r) * Md;
        if (!(e > 0.0)) return 0;
        if (e >= Md) return M;
        return (int64_t)e + 1;
    }
};
struct ThrStrat { // stratified thresholds: u_i = (i0 + rand()) / M * bins[N], bins[N] = 1  (resample.jl:49)
    double Md;
    int64_t M;
    uint32_t step, k0, k1;
    const double* Uexp;
    DEV double at(int64_t i0) const {
        const double U = Uexp ? Uexp[i0] : llpf_uniform_idx((uint32_t)i0, step, LLPF_STREAM_STRATIFY, k0, k1);
        return ((double)i0 + U) / Md * 1.0;
    }
    DEV int64_t estimate(double v) const {
        double e = v * Md;
        if (!(e > 0.0)) return 0;
        if (e >= Md) return M;
        return (int64_t)e;
    }
};

// c(v) = #{ i0 in [0,M) : thr(i0) < v } for non-decreasing thr
template <class Thr>
DEV int64_t count_below(const Thr& th, double v) {
    int64_t c = th.estimate(v);
    while (c < th.M && th.at(c) < v) ++c;
    while (c > 0 && !(th.at(c - 1) < v)) --c;
    return c;
}

template <int STRATEGY, int SRC>
__global__ __launch_bounds__(BLOCK) void k_resample(BankDev b, int K, uint32_t step, const double* __restrict__ Uexp,
                                                     int64_t M, int32_t* anc_out, double* bins_out, int only_bins, int force) {
    __shared__ uint64_t sm_w[BLOCK / 64][2];
    __shared__ uint32_t cl[TILE];
    __shared__ uint64_t sm_pref[2];
    const int f = blockIdx.y;
    const FilterScal* sc = b.scal + f;
    if (!force && !sc->do_resample) return;
    if (sc->status) return;
    const int tile = blockIdx.x;
    const int64_t N = b.N;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const bool uniform = (SRC == SRC_FILTER) && sc->uniform;

    // 1. exclusive prefix of this tile and total, from the per-tile sums of k_norm (no look-back)
    uint64_t prefix, tot;
    if (uniform) {
        const uint64_t Qc = llpf_q64(1.0 / (double)N, K);
        const int64_t before = (int64_t)tile * TILE < N ? (int64_t)tile * TILE : N;
        prefix = (uint64_t)before * Qc;
        tot = (uint64_t)N * Qc;
    } else {
        const NormPartial* __restrict__ part = b.part + (size_t)f * b.P2;
        uint64_t pre = 0, all = 0;
        for (int p = threadIdx.x; p < b.P2; p += BLOCK) {
            const uint64_t q = part[p].Q;
            all += q;
            if (p < tile) pre += q;
        }
        pre = wave_sum_u64(pre);
        all = wave_sum_u64(all);
        if (lane == 0) { sm_w[wvid][0] = pre; sm_w[wvid][1] = all; }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t p0 = 0, a0 = 0;
            for (int k = 0; k < BLOCK / 64; ++k) { p0 += sm_w[k][0]; a0 += sm_w[k][1]; }
            sm_pref[0] = p0; sm_pref[1] = a0;
        }
        __syncthreads();
        prefix = sm_pref[0];
        tot = sm_pref[1];
        __syncthreads();
    }
    if (tot == 0) return;

    // 2. quanta of this thread's NORM_IPT consecutive particles, inclusive scan
    const double* __restrict__ w = b.w + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    const double m = sc->m;
    uint64_t cq[NORM_IPT];
    {
        double2 wv[NORM_IPT / 2];
#pragma unroll
        for (int k = 0; k < NORM_IPT / 2; ++k) wv[k] = *reinterpret_cast<const double2*>(w + ib + 2 * k);
        uint64_t run = 0;
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double wk = (k & 1) ? wv[k / 2].y : wv[k / 2].x;
            double v;
            if (SRC == SRC_VALUES) v = wk;
            else if (uniform) v = 1.0 / (double)N;
            else v = llpf_exp(wk - m);
            uint64_t q = llpf_q64(v, K);
            if (ib + k >= N) q = 0;
            run += q;
            cq[k] = run;
        }
    }
    uint64_t tsum = cq[NORM_IPT - 1];
    uint64_t incl = tsum;                              // wave inclusive scan of thread totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = (uint64_t)__shfl_up((unsigned long long)incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sm_w[wvid][0] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k)
        if (k < wvid) wave_off += sm_w[k][0];
    const uint64_t excl = prefix + wave_off + (incl - tsum);

    // 3. bins and ancestor counts
    const double Td = (double)tot;
    const FilterScal* scf = sc;
    uint32_t cnt[NORM_IPT];
    int64_t c_start;
    if (STRATEGY == LLPF_RESAMPLE_SYSTEMATIC) {
        ThrSys th;
        const double U = Uexp ? Uexp[0] : llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, scf->k0, scf->k1);
        th.M = M; th.Md = (double)M; th.step = 1.0 / (double)M;
        th.r = U * 1.0 / (double)N;                    // r = rand()*bins[end]/N with bins[end] == 1
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double bin = (double)(excl + cq[k]) / Td;
            if (bins_out && ib + k < N) bins_out[(size_t)f * N + ib + k] = bin;
            cnt[k] = only_bins ? 0u : (uint32_t)count_below(th, bin);
        }
        c_start = only_bins ? 0 : count_below(th, (double)prefix / Td);
    } else {
        ThrStrat th;
        th.M = M; th.Md = (double)M; th.step = step; th.k0 = scf->k0; th.k1 = scf->k1; th.Uexp = Uexp;
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double bin = (double)(excl + cq[k]) / Td;
            if (bins_out && ib + k < N) bins_out[(size_t)f * N + ib + k] = bin;
            cnt[k] = only_bins ? 0u : (uint32_t)count_below(th, bin);
        }
        c_start = only_bins ? 0 : count_below(th, (double)prefix / Td);
    }
    if (only_bins) return;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) cl[threadIdx.x * NORM_IPT + k] = cnt[k];
    __syncthreads();

    // 4. expansion: output o is produced by the first source k of this tile with cl[k] > o
    const int64_t c_end = cl[TILE - 1];
    int32_t* ao = anc_out + (size_t)f * b.Ns;
    for (int64_t o = c_start + threadIdx.x; o < c_end; o += BLOCK) {
        int lo = 0, hi = TILE - 1;
        const uint32_t ov = (uint32_t)o;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cl[mid] > ov) hi = mid; else lo = mid + 1;
        }
        ao[o] = (int32_t)((int64_t)tile * TILE + lo);
    }
    // outputs whose threshold is >= bins[N] are never written by the reference (j keeps its previous
    // value); the previous value is only materialised here if it was the identity 1:N
    if (tile == b.P2 - 1 && sc->anc_ident) {
        for (int64_t o = c_end + threadIdx.x; o < M; o += BLOCK) ao[o] = (int32_t)o;
    }
}

// per-tile sums of the quanta of plain values (standalone resample(we))
__global__ __launch_bounds__(BLOCK) void k_qpart(BankDev b, int K) {
    __shared__ uint64_t sm_w[BLOCK / 64];
    const int f = blockIdx.y, tile = blockIdx.x;
    const double* __restrict__ w = b.w + (size_t)f * b.Ns;
    uint64_t Q = 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        const int64_t i = (int64_t)tile * TILE + (int64_t)k * BLOCK + threadIdx.x;
        if (i < b.N) Q += llpf_q64(w[i], K);
    }
    Q = wave_sum_u64(Q);
    if ((threadIdx.x & 63) == 0) sm_w[threadIdx.x >> 6] = Q;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t q = 0;
        for (int k = 0; k < BLOCK / 64; ++k) q += sm_w[k];
        NormPartial* o = b.part + (size_t)f * b.P2 + tile;
        o->S_lo = 0; o->S_hi = 0; o->E2_lo = 0; o->E2_hi = 0; o->Q = q; o->bad = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// accessors
// ------------------------------------------------------------------------------------------------
// weights(pf) / expweights(pf): materialise the lazily-normalised values
__global__ __launch_bounds__(BLOCK) void k_materialize(BankDev b, double* w_out, double* we_out) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    const FilterScal* sc = b.scal + f;
    const double wr = b.w[(size_t)f * b.Ns + i];
    double wv, we;
    if (sc->uniform) {
        wv = sc->wconst;
        we = 1.0 / (double)b.N;
    } else {
        wv = sc->norm_pending ? (wr - sc->m) - sc->l : wr;
        we = llpf_exp(wr - sc->m) * sc->inv;
    }
    if (w_out) w_out[(size_t)f * b.N + i] = wv;
    if (we_out) we_out[(size_t)f * b.N + i] = we;
}

__global__ __launch_bounds__(BLOCK) void k_soa2aos(BankDev b, const double* __restrict__ xsrc, double* dst) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    for (int d = 0; d < b.nx; ++d)
        dst[((size_t)f * b.N + i) * b.nx + d] = xsrc[((size_t)f * b.nx + d) * b.Ns + i];
}
__global__ __launch_bounds__(BLOCK) void k_aos2soa(BankDev b, const double* __restrict__ src, double* xdst) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    for (int d = 0; d < b.nx; ++d)
        xdst[((size_t)f * b.nx + d) * b.Ns + i] = (i < b.N) ? src[((size_t)f * b.N + i) * b.nx + d] : 0.0;
}
__global__ __launch_bounds__(BLOCK) void k_anc64(BankDev b, int64_t* dst) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    const FilterScal* sc = b.scal + f;
    dst[(size_t)f * b.N + i] = sc->anc_ident ? i : (int64_t)b.anc[(size_t)f * b.Ns + i];
}

// weighted_mean(pf) accessor — reference src/filtering.jl:541-549,568.  One block per filter, fixed order.
__global__ __launch_bounds__(BLOCK) void k_wmean(BankDev b, double* out) {
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.x;
    const FilterScal* sc = b.scal + f;
    const double* __restrict__ xc = b.xcur + (size_t)f * b.nx * b.Ns;
    double acc[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) acc[d] = 0.0;
    for (int64_t i = threadIdx.x; i < b.N; i += BLOCK) {
        const double wr = b.w[(size_t)f * b.Ns + i];
        const double we = sc->uniform ? 1.0 / (double)b.N : llpf_exp(wr - sc->m) * sc->inv;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < b.nx) acc[d] = acc[d] + xc[(size_t)d * b.Ns + i] * we;
    }
#pragma unroll
    for (int d = 0; d < MAXD; ++d) acc[d] = wave_sum_f64(acc[d]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < MAXD; ++d) sm_x[threadIdx.x >> 6][d] = acc[d];
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int d = 0; d < b.nx; ++d) {
            double a = sm_x[0][d];
            for (int k = 1; k < BLOCK / 64; ++k) a = a + sm_x[k][d];
            out[(size_t)f * b.nx + d] = a;
        }
}

// ------------------------------------------------------------------------------------------------
// self-tests of the shared primitives on the device
// ------------------------------------------------------------------------------------------------
__global__ void k_selftest_math(int which, const double* __restrict__ in, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double s, c, r;
    switch (which) {
        case 0: r = llpf_exp(x); break;
        case 1: r = llpf_log(x); break;
        case 2: r = llpf_log1p_nonneg(x); break;
        case 3: llpf_sincos2pi(x, &s, &c); r = s; break;
        case 4: llpf_sincos2pi(x, &s, &c); r = c; break;
        case 5: r = llpf_sqrt(x); break;
        case 6: r = 1.0 / x; break;
        case 7: r = (double)llpf_d2u(x); break;
        default: r = 0.0;
    }
    out[i] = r;
}
__global__ void k_selftest_normals(uint32_t k0, uint32_t k1, uint32_t step, uint32_t stream, int nd, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xi[MAXD];
    for (int bq = 0; 2 * bq < nd; ++bq) {
        double z0, z1;
        llpf_normal_pair((uint32_t)i, step, (uint32_t)bq, stream, k0, k1, &z0, &z1);
        xi[2 * bq] = z0;
        if (2 * bq + 1 < nd) xi[2 * bq + 1] = z1;
    }
    for (int d = 0; d < nd; ++d) out[i * nd + d] = xi[d];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid1(int64_t n, int F) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK), (unsigned)F, 1); }

bool step_supported(int model_id, int nx, int ny) {
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return nx == 4 && ny == 2;
    if (model_id == LLPF_MODEL_LINEAR_GAUSSIAN) return nx >= 1 && nx <= 4 && ny >= 1 && ny <= 4;
    return false;
}

hipError_t launch_init(const BankDev& b, uint32_t step, int init_anc, hipStream_t s) {
    dim3 g = grid1(b.Ns, b.F);
    switch (b.nx) {
        case 1: hipLaunchKernelGGL(k_init<1>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 2: hipLaunchKernelGGL(k_init<2>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 3: hipLaunchKernelGGL(k_init<3>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 4: hipLaunchKernelGGL(k_init<4>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <class Model, int NX, int NY>
static hipError_t launch_step_t(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    dim3 g((unsigned)b.P1, (unsigned)b.F, 1);
    switch (mode) {
        case MODE_WEIGHT: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_WEIGHT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_PROP: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_PROP>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_PROP_WEIGHT: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_PROP_WEIGHT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int NX>
static hipError_t launch_step_lg_ny(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    switch (b.ny) {
        case 1: return launch_step_t<LinGauss<NX, 1>, NX, 1>(b, mode, a, s);
        case 2: return launch_step_t<LinGauss<NX, 2>, NX, 2>(b, mode, a, s);
        case 3: return launch_step_t<LinGauss<NX, 3>, NX, 3>(b, mode, a, s);
        case 4: return launch_step_t<LinGauss<NX, 4>, NX, 4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    const int model_id = b.model_id;
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return launch_step_t<QuadTank<4, 2>, 4, 2>(b, mode, a, s);
    switch (b.nx) {
        case 1: return launch_step_lg_ny<1>(b, mode, a, s);
        case 2: return launch_step_lg_ny<2>(b, mode, a, s);
        case 3: return launch_step_lg_ny<3>(b, mode, a, s);
        case 4: return launch_step_lg_ny<4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_max(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_max, dim3((unsigned)b.P1, (unsigned)b.F, 1), dim3(BLOCK), 0, s, b);
    return hipGetLastError();
}

static int bank_K(const BankDev& b) { return llpf_qbits(b.N); }

hipError_t launch_norm(const BankDev& b, int want_xmean, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const int K = bank_K(b);
    if (!want_xmean) { hipLaunchKernelGGL((k_norm<0, false>), g, dim3(BLOCK), 0, s, b, K); return hipGetLastError(); }
    switch (b.nx) {
        case 1: hipLaunchKernelGGL((k_norm<1, true>), g, dim3(BLOCK), 0, s, b, K); break;
        case 2: hipLaunchKernelGGL((k_norm<2, true>), g, dim3(BLOCK), 0, s, b, K); break;
        case 3: hipLaunchKernelGGL((k_norm<3, true>), g, dim3(BLOCK), 0, s, b, K); break;
        case 4: hipLaunchKernelGGL((k_norm<4, true>), g, dim3(BLOCK), 0, s, b, K); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_finalize(const BankDev& b, const FinalizeArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)b.F), dim3(BLOCK), 0, s, b, a, bank_K(b), b.nx, a.after_predict);
    return hipGetLastError();
}

hipError_t launch_decide(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3((unsigned)((b.F + 63) / 64)), dim3(64), 0, s, b, bank_K(b));
    return hipGetLastError();
}
hipError_t launch_post_predict(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_post_predict, dim3((unsigned)((b.F + 63) / 64)), dim3(64), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_resample(const BankDev& b, uint32_t step, const double* Uexp, int64_t M,
                           int32_t* anc_out, double* bins_out, int only_bins, int force, int src_values, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const int K = bank_K(b);
    const int strategy = b.strategy;
    const int src = src_values ? SRC_VALUES : SRC_FILTER;
    const int frc = force;
    if (src == SRC_VALUES) hipLaunchKernelGGL(k_qpart, g, dim3(BLOCK), 0, s, b, K);
    if (strategy == LLPF_RESAMPLE_SYSTEMATIC) {
        if (src == SRC_FILTER) hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_SYSTEMATIC, SRC_FILTER>), g, dim3(BLOCK), 0, s, b, K, step, Uexp, M, anc_out, bins_out, only_bins, frc);
        else hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_SYSTEMATIC, SRC_VALUES>), g, dim3(BLOCK), 0, s, b, K, step, Uexp, M, anc_out, bins_out, only_bins, frc);
    } else {
        if (src == SRC_FILTER) hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_STRATIFIED, SRC_FILTER>), g, dim3(BLOCK), 0, s, b, K, step, Uexp, M, anc_out, bins_out, only_bins, frc);
        else hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_STRATIFIED, SRC_VALUES>), g, dim3(BLOCK), 0, s, b, K, step, Uexp, M, anc_out, bins_out, only_bins, frc);
    }
    return hipGetLastError();
}

hipError_t launch_materialize(const BankDev& b, double* w_out, double* we_out, hipStream_t s) {
    hipLaunchKernelGGL(k_materialize, grid1(b.N, b.F), dim3(BLOCK), 0, s, b, w_out, we_out);
    return hipGetLastError();
}
hipError_t launch_soa2aos(const BankDev& b, const double* xsrc, double* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_soa2aos, grid1(b.N, b.F), dim3(BLOCK), 0, s, b, xsrc, dst);
    return hipGetLastError();
}
hipError_t launch_aos2soa(const BankDev& b, const double* src, double* xdst, hipStream_t s) {
    hipLaunchKernelGGL(k_aos2soa, grid1(b.Ns, b.F), dim3(BLOCK), 0, s, b, src, xdst);
    return hipGetLastError();
}
hipError_t launch_wmean(const BankDev& b, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_wmean, dim3((unsigned)b.F), dim3(BLOCK), 0, s, b, out);
    return hipGetLastError();
}
hipError_t launch_anc64(const BankDev& b, int64_t* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_anc64, grid1(b.N, b.F), dim3(BLOCK), 0, s, b, dst);
    return hipGetLastError();
}
hipError_t launch_selftest_math(int which, const double* in, double* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, which, in, out, n);
    return hipGetLastError();
}
hipError_t launch_selftest_normals(uint32_t k0, uint32_t k1, uint32_t step, uint32_t stream, int nd,
                                   double* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, k0, k1, step, stream, nd, out, n);
    return hipGetLastError();
}

}  // namespace llpf
