// kernels.hip — hand-written gfx950 kernels of the particle-filter step.
//
// One filter step (correct! then predict!, reference src/filtering.jl:164-168, 140-153) is three launches:
//   k_norm      : m = max w (from the sharded max accumulators), e_i = exp(w_i - m); fixed-point sums of e and e^2
//                 go to sharded integer accumulators, the per-tile sum of the resampling quanta to tileq
//                                                                              (logsumexp!, utils.jl:18-27;
//                                                                               effective_particles, resample.jl:1-2)
//   k_resample  : every block derives the scalars log1p(s), 1/(s+1), ESS and the shouldresample decision from the
//                 accumulators (block 0 publishes them); if resampling: per 1024-particle tile an integer inclusive
//                 scan of the quanta (tile prefix = masked sum of tileq, so no look-back / spinning),
//                 bins = cum * (1/total), ancestor COUNTS c(bins) for the systematic / stratified thresholds,
//                 expansion of the counts into ancestor indices by head-flag scatter + max-scan in LDS
//                                                                              (resample, resample.jl:5-61)
//   k_step      : gather x[anc[i]] -> dynamics -> + Philox/Box–Muller process noise -> store x (SoA, 16-B vectors)
//                 -> w = w_prev + logpdf(y_next - g(x)) -> block max -> atomicMax  (propagate_particles!,
//                                                                               PFtypes.jl:122-139;
//                                                                               measurement_equation!, :107-120)
// All particle data is fp64 structure-of-arrays; wave64; 256-thread workgroups; grid = (tiles, filters).
// Compiled with -ffp-contract=off: the arithmetic is the same IEEE sequence as oracle/llpf_oracle.c (device order).
#include "engine.hpp"

namespace llpf {

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// wave / block reductions and scans (wave = 64 lanes) on DPP.
// Measured on gfx950 (tools/inst_cost.hip): a ds_bpermute_b32 (what __shfl_* compiles to) costs ~10 ns of a SIMD's
// time, a DPP-modified VALU move ~1 ns; a 64-lane reduction of one 64-bit value is 12 bpermutes vs 12 DPP moves.
// Row = 16 lanes.  Inclusive scan: row_shr 1,2,4,8 (Hillis–Steele inside a row, out-of-row sources read as the
// identity), then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3.  Lane 63 ends with the total.
// ------------------------------------------------------------------------------------------------
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_QUAD_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141

template <int CTRL, int ROW_MASK, bool BOUND>
DEV uint64_t dpp_u64(uint64_t old, uint64_t v) {
    const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)old, (int)(uint32_t)v, CTRL, ROW_MASK, 0xF, BOUND);
    const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(old >> 32), (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xF, BOUND);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
DEV uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
// inclusive prefix sum over the wave
DEV uint64_t wave_scan_u64(uint64_t x) {
    x += dpp_u64<DPP_ROW_SHR(1), 0xF, true>(0, x);
    x += dpp_u64<DPP_ROW_SHR(2), 0xF, true>(0, x);
    x += dpp_u64<DPP_ROW_SHR(4), 0xF, true>(0, x);
    x += dpp_u64<DPP_ROW_SHR(8), 0xF, true>(0, x);
    x += dpp_u64<DPP_ROW_BCAST15, 0xA, false>(0, x);
    x += dpp_u64<DPP_ROW_BCAST31, 0xC, false>(0, x);
    return x;
}
DEV uint32_t wave_scan_max_u32(uint32_t x) {
#define LLPF_MAXSTEP(CTRL, RM, BC) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, RM, 0xF, BC); x = t > x ? t : x; }
    LLPF_MAXSTEP(DPP_ROW_SHR(1), 0xF, true) LLPF_MAXSTEP(DPP_ROW_SHR(2), 0xF, true) LLPF_MAXSTEP(DPP_ROW_SHR(4), 0xF, true)
    LLPF_MAXSTEP(DPP_ROW_SHR(8), 0xF, true) LLPF_MAXSTEP(DPP_ROW_BCAST15, 0xA, false) LLPF_MAXSTEP(DPP_ROW_BCAST31, 0xC, false)
#undef LLPF_MAXSTEP
    return x;
}
// total over the wave, returned uniformly to every lane
DEV uint64_t wave_sum_u64(uint64_t v) { return readlane_u64(wave_scan_u64(v), 63); }
DEV llpf_u128 wave_sum_u128(llpf_u128 v) {
    // sum the three 43-bit limbs separately (no carries between lanes), recombine: exact
    const uint64_t M43 = ((uint64_t)1 << 43) - 1;
    const uint64_t l0 = wave_sum_u64(v.lo & M43);
    const uint64_t l1 = wave_sum_u64(((v.lo >> 43) | (v.hi << 21)) & M43);
    const uint64_t l2 = wave_sum_u64(v.hi >> 22);
    llpf_u128 r = {l0, 0}, t;
    t.lo = l1 << 43; t.hi = l1 >> 21;
    r = llpf_u128_add(r, t);
    t.lo = 0; t.hi = l2 << 22;
    return llpf_u128_add(r, t);
}
DEV double wave_max(double v) {
    // running maximum with the same DPP sequence; out-of-row / masked lanes read the lane's own value
    uint64_t x = llpf_d2u(v);
#define LLPF_FMAXSTEP(CTRL, RM) { const double t = llpf_u2d(dpp_u64<CTRL, RM, false>(x, x)); const double c = llpf_u2d(x); x = llpf_d2u(llpf_fmax(c, t)); }
    LLPF_FMAXSTEP(DPP_ROW_SHR(1), 0xF) LLPF_FMAXSTEP(DPP_ROW_SHR(2), 0xF) LLPF_FMAXSTEP(DPP_ROW_SHR(4), 0xF)
    LLPF_FMAXSTEP(DPP_ROW_SHR(8), 0xF) LLPF_FMAXSTEP(DPP_ROW_BCAST15, 0xA) LLPF_FMAXSTEP(DPP_ROW_BCAST31, 0xC)
#undef LLPF_FMAXSTEP
    return llpf_u2d(readlane_u64(x, 63));
}
// fixed-order fp64 sum over the wave (used only for the weighted-mean output, never fed back)
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
        q = dot * g.invscal;
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
            z[i] = acc * g.invLd[i];
        }
#pragma unroll
        for (int i = ND - 1; i >= 0; --i) {
            double acc = z[i];
#pragma unroll
            for (int j = i + 1; j < ND; ++j) acc = acc - g.L[j * MAXD + i] * z2[j];
            z2[i] = acc * g.invLd[i];
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
    static constexpr bool RB = false;
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

// Rao-Blackwellized filter with constant matrices (reference src/rbpf.jl:163-283): the particle is [xn; xl], the
// covariance of xl is shared by all particles and advanced on the host (csrc/shared/llpf_rbkf.h).  A = [Fn An; 0 Al],
// B = [Bn; Bl], C = [Gn Cl] (row stride NX / nu / NX).  Operation order identical to oracle/llpf_oracle.c:rb_*.
template <int NX, int NY>
struct RBLin {
    static constexpr bool RB = true;
    const ModelD* md;
    const double* u;
    int nn, nl, nu;
    DEV void prepare(const ModelD* m, const double* __restrict__ uu, double /*t*/) {
        md = m; u = uu; nn = m->nxn; nl = NX - m->nxn; nu = (uu != nullptr) ? m->nu : 0;
    }
    // the propagation of predict! (:185-224): xs = [fi + z ; Al xl + Bl u + L (z - An xl)]
    DEV void rb_propagate(const double* xp, uint32_t idx, uint32_t step, uint32_t k0, uint32_t k1, const RBStep* rp, double* xs) const {
        double xi[NX], nz[NX], fi[NX], xl1[NX];
        llpf_normals(idx, step, LLPF_STREAM_DYNAMICS, k0, k1, nn, xi);
        const GaussD& g = md->df;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            if (i < nn) {                                  // rand(pf.rng, pf.R1n) = mu + L xi
                double v;
                if (g.kind == LLPF_COV_SCAL) v = g.sqrtscal * xi[i];
                else if (g.kind == LLPF_COV_DIAG) v = g.sqrtdiag[i] * xi[i];
                else {
                    v = g.L[i * MAXD + 0] * xi[0];
#pragma unroll
                    for (int j = 1; j < NX; ++j) if (j <= i) v = v + g.L[i * MAXD + j] * xi[j];
                }
                nz[i] = v + g.mu[i];
            }
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            if (r < nn) {                                  // fi = Fn xn + Bn u
                double a = md->A[r * NX] * xp[0];
#pragma unroll
                for (int c = 1; c < NX; ++c) if (c < nn) a = a + md->A[r * NX + c] * xp[c];
                if (nu > 0) {
                    double b2 = md->B[r * nu] * u[0];
                    for (int c = 1; c < nu; ++c) b2 = b2 + md->B[r * nu + c] * u[c];
                    a = a + b2;
                }
                fi[r] = a;
            }
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            if (r < nl) {                                  // Al xl + Bl u
                double a = md->A[(nn + r) * NX + nn] * xp[nn];
#pragma unroll
                for (int c = 1; c < NX; ++c) if (c < nl) a = a + md->A[(nn + r) * NX + nn + c] * xp[nn + c];
                if (nu > 0) {
                    double b2 = md->B[(nn + r) * nu] * u[0];
                    for (int c = 1; c < nu; ++c) b2 = b2 + md->B[(nn + r) * nu + c] * u[c];
                    a = a + b2;
                }
                xl1[r] = a;
            }
        }
        if (md->rb_zeroAn) {
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                if (r < nn) xs[r] = fi[r] + nz[r];
                else xs[r] = xl1[r - nn];
            }
        } else {
            double Axl[NX], z[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                if (r < nn) {
                    double a = md->A[r * NX + nn] * xp[nn];
#pragma unroll
                    for (int c = 1; c < NX; ++c) if (c < nl) a = a + md->A[r * NX + nn + c] * xp[nn + c];
                    Axl[r] = a;
                    z[r] = a + nz[r];
                    xs[r] = fi[r] + z[r];
                }
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                if (r < nl) {
                    double a = rp->L[r * nn] * (z[0] - Axl[0]);
#pragma unroll
                    for (int c = 1; c < NX; ++c) if (c < nn) a = a + rp->L[r * nn + c] * (z[c] - Axl[c]);
                    xs[nn + r] = xl1[r] + a;
                }
            }
        }
    }
    // the per-particle part of correct! (:253-280): returns ll and applies the Kalman measurement update to xl
    DEV double rb_weight(double* xs, const double* y, const RBStep* rc, bool first) const {
        double yn[NY], yl[NY], e[NY];
#pragma unroll
        for (int r = 0; r < NY; ++r) {
            double a = md->C[r * NX] * xs[0];
#pragma unroll
            for (int c = 1; c < NX; ++c) if (c < nn) a = a + md->C[r * NX + c] * xs[c];
            yn[r] = a;
            double b2 = md->C[r * NX + nn] * xs[nn];
#pragma unroll
            for (int c = 1; c < NX; ++c) if (c < nl) b2 = b2 + md->C[r * NX + nn + c] * xs[nn + c];
            yl[r] = b2;
        }
        double ll;
        if (!md->rb_zeroC) {
#pragma unroll
            for (int r = 0; r < NY; ++r) e[r] = first ? (y[r] - yn[r]) - yl[r] : y[r] - (yn[r] + yl[r]);
            ll = gauss_logpdf<NY>(rc->dS, e);
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                if (r < nl) {
                    double a = rc->K[r * NY] * e[0];
#pragma unroll
                    for (int c = 1; c < NY; ++c) a = a + rc->K[r * NY + c] * e[c];
                    xs[nn + r] = xs[nn + r] + a;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < NY; ++r) e[r] = y[r] - (yn[r] + yl[r]);
            ll = gauss_logpdf<NY>(md->dg, e);
#pragma unroll
            for (int r = 0; r < NX; ++r) if (r < nl) xs[nn + r] = rc->kfx[r];
        }
        return ll;
    }
    // unused generic hooks
    DEV void dynamics(const double* x, double* out) const { for (int d = 0; d < NX; ++d) out[d] = x[d]; }
    DEV void measurement(const double*, double*) const {}
};

template <int NX, int NY>
struct QuadTank {   // reference examples/example_quadtank.jl:8-35 with rk4 of src/utils.jl:220-237
    static constexpr bool RB = false;
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
// cross-block accumulators (see engine.hpp): order-preserving max key, limb-wise integer sums
// ------------------------------------------------------------------------------------------------
DEV uint64_t max_key(double x) {           // monotone map double -> u64; NaN (positive) maps above +inf,
    const uint64_t u = llpf_d2u(x);        // so a NaN weight wins the max like Julia's findmax; key 0 is below -inf
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
DEV double max_unkey(uint64_t k) {
    return llpf_u2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k);
}
DEV uint64_t* acc_slot(uint64_t* acc, int word, int shard) { return acc + ((size_t)word * NSHARD + shard) * ACC_STRIDE; }
DEV const uint64_t* acc_slot(const uint64_t* acc, int word, int shard) { return acc + ((size_t)word * NSHARD + shard) * ACC_STRIDE; }

DEV void acc_max(uint64_t* acc, int parity, double blockmax, bool any_nan) {
    const uint64_t key = any_nan ? max_key(llpf_u2d(0x7ff8000000000000ULL)) : max_key(blockmax);
    atomicMax(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_PM(parity), blockIdx.x & (NSHARD - 1))),
              (unsigned long long)key);
}
// every wave combines the NSHARD copies of the running max itself (lanes 0..7 load, 3 shuffles, broadcast)
DEV double acc_read_max_wave(const uint64_t* acc, int parity) {
    const int lane = threadIdx.x & 63;
    uint64_t k = (lane < NSHARD) ? *acc_slot(acc, ACC_PM(parity), lane) : 0;
#define LLPF_KSTEP(CTRL) { const uint64_t t = dpp_u64<CTRL, 0xF, false>(k, k); k = t > k ? t : k; }
    LLPF_KSTEP(DPP_QUAD_XOR1) LLPF_KSTEP(DPP_QUAD_XOR2) LLPF_KSTEP(DPP_ROW_HALF_MIRROR)
#undef LLPF_KSTEP
    k = readlane_u64(k, 0);
    return max_unkey(k);
}
constexpr uint64_t M43 = ((uint64_t)1 << 43) - 1;
DEV void acc_add_u128(uint64_t* acc, int word0, llpf_u128 v) {
    const int sh = blockIdx.x & (NSHARD - 1);
    const uint64_t limb[3] = {v.lo & M43, ((v.lo >> 43) | (v.hi << 21)) & M43, v.hi >> 22};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (limb[k]) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, word0 + k, sh)), (unsigned long long)limb[k]);
}
// limb sums (each already summed over shards) -> 128-bit value
DEV llpf_u128 acc_combine_u128(uint64_t a0, uint64_t a1, uint64_t a2) {
    llpf_u128 r = {a0, 0}, t;
    t.lo = a1 << 43; t.hi = a1 >> 21;
    r = llpf_u128_add(r, t);
    t.lo = 0; t.hi = a2 << 22;
    r = llpf_u128_add(r, t);
    return r;
}

DEV uint64_t* tileq_slot(const BankDev& b, int slot, int f) { return b.tileq + ((size_t)slot * b.F + f) * b.P2; }

// a launch of run-step k is a no-op when an EARLIER launch flagged a failed bound test (flag = 1 + its step)
DEV bool run_is_stopped(const BankDev& b, int64_t k) {
    const uint32_t fl = *b.bank_flag;
    return fl != 0 && (int64_t)(fl - 1) < k;
}

// Exp-sums of freshly computed weights against the analytic bound `off` (see oracle/llpf_oracle.c:dev_norm_bound):
// e = exp(w - off) <= 1, S += fix96(e), [E2 += fix96(e^2)], quantum q = floor(e 2^K).
struct WeightAcc {
    llpf_u128 S, E2;
    uint64_t bad;
    DEV void init() { S.lo = 0; S.hi = 0; E2.lo = 0; E2.hi = 0; bad = 0; }
    DEV uint64_t add(double w, double off, int K, bool need_e2, double* e_out = nullptr) {
        const double e = llpf_exp_le0(w - off);
        if (e_out) *e_out = e;
        bad += (e != e) ? 1u : 0u;
        S = llpf_u128_add(S, llpf_fix96_unit(e));
        if (need_e2) E2 = llpf_u128_add(E2, llpf_fix96_unit(e * e));
        return llpf_q64_unit(e, K);
    }
    // block-wide totals into the sharded accumulators of `slot`; sm: [BLOCK/64][5] u64 of LDS
    DEV void flush(uint64_t* acc, int slot, bool need_e2, uint64_t (*sm)[5]) {
        llpf_u128 s = wave_sum_u128(S), e2 = {0, 0};
        if (need_e2) e2 = wave_sum_u128(E2);
        const uint64_t bd = wave_sum_u64(bad);
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) { sm[wv][0] = s.lo; sm[wv][1] = s.hi; sm[wv][2] = e2.lo; sm[wv][3] = e2.hi; sm[wv][4] = bd; }
        __syncthreads();
        if (threadIdx.x == 0) {
            llpf_u128 ts = {sm[0][0], sm[0][1]}, te = {sm[0][2], sm[0][3]};
            uint64_t tb = sm[0][4];
            for (int k = 1; k < BLOCK / 64; ++k) {
                llpf_u128 a1 = {sm[k][0], sm[k][1]}, a2 = {sm[k][2], sm[k][3]};
                ts = llpf_u128_add(ts, a1);
                te = llpf_u128_add(te, a2);
                tb += sm[k][4];
            }
            acc_add_u128(acc, ACC_S(slot), ts);
            if (need_e2) acc_add_u128(acc, ACC_E2(slot), te);
            if (tb) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(slot), blockIdx.x & (NSHARD - 1))), (unsigned long long)tb);
        }
    }
};

// fixed-order fp64 block sum of the per-thread partial sums e_i x_i (weighted-mean output only; never fed back)
template <int NX>
DEV void block_store_xm(const double* xm, double* dst /* [MAXD] */, double (*smx)[MAXD]) {
    double v[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) v[d] = wave_sum_f64(xm[d]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) smx[wv][d] = v[d];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            double a = smx[0][d];
            for (int k = 1; k < BLOCK / 64; ++k) a = a + smx[k][d];
            dst[d] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_step — fused propagate + weight + running max
// ------------------------------------------------------------------------------------------------
template <class Model, int NX, int NY, int MODE>
__global__ __launch_bounds__(BLOCK) void k_step(BankDev b, const ModelD* __restrict__ models,
                                                 const FilterScal* scal, StepArgs a) {
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_acc[BLOCK / 64][5];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.y;
    const ModelD* md = models + f;
    const FilterScal* sc = scal + f;
    if (run_is_stopped(b, a.k)) return;
    if (a.only_fallback ? !sc->fallback : (sc->fallback != 0)) return;   // redo launches take the flagged filters, all others skip them
    const int do_res = (MODE != MODE_WEIGHT && MODE != MODE_AUX) ? sc->do_resample : 0;
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
    bool bad = false;
    // bound of the weights this kernel produces: max of the previous (normalised) weights + the density's peak
    double off = 0.0;
    WeightAcc wacc;
    uint64_t qsum = 0;
    double xm[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
    if (MODE != MODE_PROP) {
        const double wmx = do_res ? b.log1N : (uniform ? wconst : sc->wmax);
        double c0w = md->dg.c0;
        if constexpr (Model::RB) { if (!md->rb_zeroC) c0w = (a.rb_corr + f)->dS.c0; }   // peak of N(0, S) of this correct!
        off = a.has_y ? wmx + c0w : wmx;
        wacc.init();
    }
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
                if constexpr (Model::RB) {
                    model.rb_propagate(xp[p], (uint32_t)(i0 + p), a.step, k0, k1, a.rb_pred + f, xs[p]);
                    continue;
                }
                double fx[NX], xi[NX], nz[NX];
                model.dynamics(xp[p], fx);
                if (MODE == MODE_AUX) {            // propagate_particles!(pf, u, p, t, nothing): no noise
#pragma unroll
                    for (int d = 0; d < NX; ++d) xs[p][d] = fx[d];
                } else {
                    llpf_normals((uint32_t)(i0 + p), a.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
                    gauss_sample<NX>(md->df, xi, nz);
#pragma unroll
                    for (int d = 0; d < NX; ++d) xs[p][d] = fx[d] + nz[d];
                }
            }
            if (!(Model::RB && MODE == MODE_PROP_WEIGHT && a.has_y)) {
#pragma unroll
                for (int d = 0; d < NX; ++d) {
                    double2 v;
                    v.x = xs[0][d];
                    v.y = xs[1][d];
                    *reinterpret_cast<double2*>(xn + (size_t)d * Ns + i0) = v;
                }
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
            double lamv[STEP_PPT];
#pragma unroll
            for (int p = 0; p < STEP_PPT; ++p) {
                double wv = wp[p];
                if (MODE == MODE_AUX) {            // lambda .= 0; lambda += logpdf; w .+= lambda  (filtering.jl:201-204)
                    double lam = 0.0;
                    if (a.has_y) {
                        double g[NY], v[NY];
                        model.measurement(xs[p], g);
#pragma unroll
                        for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                        lam = lam + gauss_logpdf<NY>(md->dg, v);
                    }
                    lamv[p] = lam;
                    wv = wv + lam;
                } else if (a.has_y) {
                    if constexpr (Model::RB) {
                        wv = wv + model.rb_weight(xs[p], y, a.rb_corr + f, i0 + p == 0);
                    } else {
                        double g[NY], v[NY];
                        model.measurement(xs[p], g);
#pragma unroll
                        for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                        wv = wv + gauss_logpdf<NY>(md->dg, v);
                    }
                }
                if (i0 + p >= N) wv = -LLPF_INF;   // padding lanes carry zero weight
                wn[p] = wv;
                bad = bad || (wv != wv);
                bmax = llpf_fmax(bmax, wv);
            }
            double2 wo;
            wo.x = wn[0];
            wo.y = wn[1];
            *reinterpret_cast<double2*>(w + i0) = wo;
            if constexpr (Model::RB) {             // correct! has updated xl (Kalman measurement update)
                if (a.has_y) {
                    double* xdst = (MODE == MODE_WEIGHT) ? const_cast<double*>(xc) : xn;
#pragma unroll
                    for (int d = 0; d < NX; ++d) {
                        double2 v;
                        v.x = xs[0][d];
                        v.y = xs[1][d];
                        *reinterpret_cast<double2*>(xdst + (size_t)d * Ns + i0) = v;
                    }
                }
            }
            if (MODE == MODE_AUX) {
                double2 lo;
                lo.x = lamv[0];
                lo.y = lamv[1];
                *reinterpret_cast<double2*>(b.lam + (size_t)f * Ns + i0) = lo;
            }
            if (a.accumulate) {   // merged schedule: exp-sums, quanta and tile sums of the new weights formed here
                ulonglong2 qv;
                double e0, e1;
                qv.x = wacc.add(wn[0], off, a.K, a.need_e2 != 0, &e0);
                qv.y = wacc.add(wn[1], off, a.K, a.need_e2 != 0, &e1);
                *reinterpret_cast<ulonglong2*>(b.quanta_next + (size_t)f * Ns + i0) = qv;
                qsum += qv.x + qv.y;
                if (a.want_xmean) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) { xm[d] = xm[d] + xs[0][d] * e0; xm[d] = xm[d] + xs[1][d] * e1; }
                }
            }
        }
    }
    if (MODE != MODE_PROP) {
        const double r = block_max(bmax, sm_max);
        const int anybad = __syncthreads_or(bad ? 1 : 0);
        if (threadIdx.x == 0) acc_max(b.acc + (size_t)f * ACC_WORDS, a.parity, r, anybad != 0);
        if (a.accumulate) wacc.flush(b.acc + (size_t)f * ACC_WORDS, a.parity, a.need_e2 != 0, sm_acc);
        if (a.accumulate && a.want_xmean) block_store_xm<NX>(xm, b.xmpart + ((size_t)f * b.P1 + blockIdx.x) * MAXD, sm_x);
        // all particles of this block lie in one 1024-particle tile
        qsum = wave_sum_u64(qsum);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm_acc[threadIdx.x >> 6][0] = qsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t q = 0;
            for (int k = 0; k < BLOCK / 64; ++k) q += sm_acc[k][0];
            const int64_t tile = ((int64_t)blockIdx.x * STEP_TILE) / TILE;
            if (q) atomicAdd(reinterpret_cast<unsigned long long*>(tileq_slot(b, a.parity, f) + tile), (unsigned long long)q);
            if (blockIdx.x == 0) {
                FilterScal* scw = b.scal + f;
                if (a.accumulate) scw->xm_parts = b.P1;
                scw->off_slot[a.parity] = off;
                scw->e2v_slot[a.parity] = a.need_e2;
                scw->u_slot[a.parity] = llpf_uniform_step(a.next_step, LLPF_STREAM_RESAMPLE, k0, k1);
            }
        }
    }
    if (MODE != MODE_WEIGHT && MODE != MODE_AUX && blockIdx.x == 0 && threadIdx.x == 0) {
        // bookkeeping of this predict! (fields no block of this kernel reads): state.j == 1:N unless resampled
        FilterScal* scw = b.scal + f;
        scw->anc_ident_s[b.anc_slot ^ 1] = do_res ? 0 : 1;
        scw->last_resampled = do_res;
        scw->resample_count += do_res;
    }
}

// ------------------------------------------------------------------------------------------------
// k_max — maxima of the raw log-weights (when no weighting kernel produced them: llpf_set_weights,
// llpf_logsumexp); also zeroes the sum accumulators like a weighting kernel does
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_max(BankDev b, int parity) {
    __shared__ double sm_max[BLOCK / 64];
    const int f = blockIdx.y;
    const double* w = b.w + (size_t)f * b.Ns;
    double bmax = -LLPF_INF;
    bool bad = false;
#pragma unroll
    for (int it = 0; it < STEP_ITERS; ++it) {
        const int64_t i0 = ((int64_t)blockIdx.x * STEP_ITERS + it) * (BLOCK * STEP_PPT) + (int64_t)threadIdx.x * STEP_PPT;
        const double2 wv = *reinterpret_cast<const double2*>(w + i0);
        if (i0 < b.N) { bmax = llpf_fmax(bmax, wv.x); bad = bad || (wv.x != wv.x); }
        if (i0 + 1 < b.N) { bmax = llpf_fmax(bmax, wv.y); bad = bad || (wv.y != wv.y); }
    }
    const double r = block_max(bmax, sm_max);
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) acc_max(b.acc + (size_t)f * ACC_WORDS, parity, r, anybad != 0);
}

// ------------------------------------------------------------------------------------------------
// k_norm — exp-weights and their exact sums  (logsumexp! utils.jl:18-27, sum_all_but :66-71,
// effective_particles resample.jl:1-2; optional weighted_mean filtering.jl:541-549)
// ------------------------------------------------------------------------------------------------
template <int NX, bool XMEAN, bool NEED_E2>
__global__ __launch_bounds__(BLOCK) void k_norm(BankDev b, int K, int parity, uint32_t step, int only_fallback, int bound, int64_t kstep) {
    __shared__ uint64_t sm_u[BLOCK / 64][6];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    if (only_fallback && !b.scal[f].fallback) return;
    if (bound && run_is_stopped(b, kstep)) return;
    if (bound && b.scal[f].fallback) return;
    uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    const double* __restrict__ w = b.w + (size_t)f * b.Ns;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * b.Ns;

    double2 wv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
        wv[k] = *reinterpret_cast<const double2*>(w + i0);
    }
    const double m = bound ? b.scal[f].off_slot[parity] : acc_read_max_wave(acc, parity);

    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0, bad = 0;
    double xm[NX > 0 ? NX : 1];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
        const double e0 = llpf_exp_le0(wv[k].x - m);
        const double e1 = llpf_exp_le0(wv[k].y - m);
        bad += (e0 != e0) ? 1u : 0u;
        bad += (e1 != e1) ? 1u : 0u;
        S = llpf_u128_add(S, llpf_fix96_unit(e0));
        S = llpf_u128_add(S, llpf_fix96_unit(e1));
        if (NEED_E2) {
            E2 = llpf_u128_add(E2, llpf_fix96_unit(e0 * e0));
            E2 = llpf_u128_add(E2, llpf_fix96_unit(e1 * e1));
        }
        ulonglong2 qv;
        qv.x = llpf_q64_unit(e0, K);
        qv.y = llpf_q64_unit(e1, K);
        *reinterpret_cast<ulonglong2*>(b.quanta + (size_t)f * b.Ns + i0) = qv;
        Q += qv.x;
        Q += qv.y;
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
    if (NEED_E2) E2 = wave_sum_u128(E2);
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
        acc_add_u128(acc, ACC_S(parity), s);
        if (NEED_E2) acc_add_u128(acc, ACC_E2(parity), e2);
        if (tile == 0) {   // the single uniform a systematic resample of this step consumes (reference: rand(), resample.jl:23)
            FilterScal* sc = b.scal + f;
            sc->u_slot[parity] = llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1);
            sc->e2v_slot[parity] = NEED_E2 ? 1 : 0;
            sc->xm_parts = b.P2;
        }
        if (bd) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(parity), blockIdx.x & (NSHARD - 1))), (unsigned long long)bd);
        tileq_slot(b, parity, f)[tile] = q;
        if (XMEAN) {
            for (int d = 0; d < NX; ++d) {
                double a = sm_x[0][d];
                for (int k = 1; k < BLOCK / 64; ++k) a = a + sm_x[k][d];
                b.xmpart[((size_t)f * b.P1 + tile) * MAXD + d] = a;
            }
        }
    }
}

// accessor path: sum e^2 (fixed point) and ESS of the current weights when the hot loop skipped them
__global__ __launch_bounds__(BLOCK) void k_ess(BankDev b) {
    __shared__ uint64_t sm_u[BLOCK / 64][2];
    const int f = blockIdx.x;
    FilterScal* sc = b.scal + f;
    if (sc->uniform || sc->e2_valid || sc->status) return;
    const double* w = b.w + (size_t)f * b.Ns;
    const double m = sc->m;
    llpf_u128 E2 = {0, 0};
    for (int64_t i = threadIdx.x; i < b.N; i += BLOCK) {
        const double e = llpf_exp_le0(w[i] - m);
        E2 = llpf_u128_add(E2, llpf_fix96_unit(e * e));
    }
    E2 = wave_sum_u128(E2);
    if ((threadIdx.x & 63) == 0) { sm_u[threadIdx.x >> 6][0] = E2.lo; sm_u[threadIdx.x >> 6][1] = E2.hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        llpf_u128 t = {sm_u[0][0], sm_u[0][1]};
        for (int k = 1; k < BLOCK / 64; ++k) { llpf_u128 u = {sm_u[k][0], sm_u[k][1]}; t = llpf_u128_add(t, u); }
        const double e2 = llpf_fix96_to_double(t);
        sc->e2 = e2;
        sc->ess = (sc->stot * sc->stot) / e2;
        sc->e2_valid = 1;
    }
}

// after a propagate-only predict!: reset_weights! if it resampled (reference src/utils.jl:73-79)
__global__ void k_post_predict(BankDev b) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= b.F) return;
    FilterScal* sc = b.scal + f;
    if (sc->do_resample) {
        sc->uniform = 1;
        sc->wconst = b.log1N;
        sc->m = 0.0;
        sc->mtrue = 0.0;             // maxw[] = 0
        sc->wmax = b.log1N;
        sc->norm_pending = 0;
    }
    sc->do_resample = 0;
}

// ------------------------------------------------------------------------------------------------
// k_resample — [finalize] + scan + ancestor counts + expansion, one tile per block
// ------------------------------------------------------------------------------------------------
enum { SRC_FILTER = 0, SRC_VALUES = 1 };

// Thresholds are non-decreasing in i0.  count(v) = #{ i0 in [0,M) : thr(i0) < v } is evaluated with a closed
// form whenever v is not within `delta` (in index units) of a threshold, and with the exact predicate otherwise:
// thr(i0) differs from its real-arithmetic value (r + i0/M resp. (i0+U)/M) by < 4 ulp(1), i.e. by < M * 1e-15 in
// index units, far below delta, so both paths give the count defined by the reference's comparison `s[i] < bins[b]`.
struct ThrSys {   // systematic: s[i] = fl(r + fl(i0 * (1/M)))  (resample.jl:23-24, Julia StepRangeLen getindex)
    double r, step, Md, delta;
    int32_t M;
    DEV double at(int32_t i0) const { return r + (double)i0 * step; }
    DEV int32_t count(double v) const {
        const double e = (v - r) * Md;
        if (e <= -delta) return 0;
        if (e >= Md + delta) return M;
        const double fl = __builtin_floor(e);
        const double fr = e - fl;
        int32_t c = (int32_t)fl + 1;
        c = c < 0 ? 0 : (c > M ? M : c);
        if (fr > delta && fr < 1.0 - delta && e > 0.0) return c;
        while (c < M && at(c) < v) ++c;
        while (c > 0 && !(at(c - 1) < v)) --c;
        return c;
    }
};
struct ThrStrat { // stratified: u_i = (i0 + rand()) / M * bins[N]  (resample.jl:49)
    double Md, delta, binsN;
    int32_t M;
    uint32_t step, k0, k1;
    const double* Uexp;
    DEV double at(int32_t i0) const {
        const double U = Uexp ? Uexp[i0] : llpf_uniform_idx((uint32_t)i0, step, LLPF_STREAM_STRATIFY, k0, k1);
        return ((double)i0 + U) / Md * binsN;
    }
    DEV int32_t count(double v) const {
        const double e = v * Md;
        if (e <= -delta) return 0;
        if (e >= Md + delta) return M;
        const double fl = __builtin_floor(e);
        const double fr = e - fl;
        int32_t c = (int32_t)fl;
        c = c < 0 ? 0 : (c > M ? M : c);
        if (fr > delta && fr < 1.0 - delta && e > 0.0 && c < M) return at(c) < v ? c + 1 : c;
        while (c < M && at(c) < v) ++c;
        while (c > 0 && !(at(c - 1) < v)) --c;
        return c;
    }
};

// ---- shared machinery of the resample kernels -----------------------------------------------------------------
struct ResShared {                 // LDS scratch
    uint64_t red[BLOCK / 64][4];
    uint64_t accw[8];
    double dval[4];
    uint32_t cl[TILE];
};
struct ResHead {                   // block-uniform results of res_head()
    double a;                      // offset of the pending normalisation (bound or maximum)
    double mtrue;                  // true maximum of the raw weights
    double s;                      // exact form only: sum_{i != argmax} e_i
    double stot, e2;               // sum e_i (all particles), sum e_i^2 (-1: not accumulated)
    uint64_t prefix, tot;          // exclusive prefix of this tile's quanta, total of all quanta
    int dr, status, uniform, fast;
};
enum { RES_STATUS_FALLBACK = 100, RES_STATUS_SKIP = 101 };   // SKIP: this launch is a no-op for the filter   // internal: bound test failed, the host redoes this step in exact form

// shouldresample (reference src/resample.jl:5-10) without the division: ESS = stot^2 / sum(e^2) < N*thr
DEV int decide_resample(double thr, double N, double stot, double e2) {
    if (thr == 1.0) return 1;
    return (stot * stot < (N * thr) * e2) ? 1 : 0;
}
// log(sum exp(w - a)) in the form the normalisation was accumulated in
DEV double head_log(const ResHead& h) { return h.fast ? llpf_log(h.stot) : llpf_log1p_nonneg(h.s); }

// Head of a resample launch: all global loads are issued first (accumulator slots, per-tile quanta sums), one
// __syncthreads, then EVERY thread derives the block-uniform scalars (integer sums => identical everywhere).
// Tile 0 publishes the scalars of logsumexp! / effective_particles / shouldresample for later kernels.
// `defer_skip`: the caller fetched the run's stop flag and the filter's fallback flag without waiting for them; the
// launch-is-a-no-op test is made here after the barrier, so that those two loads overlap all the others.
template <int SRC>
DEV ResHead res_head(const BankDev& b, const ResArgs& a, int f, int tile, ResShared& sh,
                     bool defer_skip = false, uint32_t stop_flag = 0, int fb_flag = 0) {
    FilterScal* sc = b.scal + f;
    uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const double Nd = (double)b.N;
    ResHead h;
    const bool fin = (a.mode & RES_FINALIZE) != 0;
    const bool unif0 = (SRC == SRC_FILTER) && !fin && sc->uniform;
    // scalars of the previous launch that are needed after the barrier below: fetched now, with the other loads
    const double off_pre = sc->off_slot[a.parity];
    const int e2v_pre = sc->e2v_slot[a.parity];
    const int status_pre = sc->status;

    // loads.  wave 0: lane group g (8 lanes = 8 shards) fetches word g of this slot's accumulator set
    uint64_t accv = 0;
    const int grp = threadIdx.x / NSHARD, shard = threadIdx.x % NSHARD;
    if (fin && threadIdx.x < 64) accv = *acc_slot(acc, acc_word_of_group(grp, a.parity), shard);
    uint64_t pre = 0, all = 0;
    if ((a.mode & RES_RESAMPLE) && !unif0) {
        const uint64_t* __restrict__ tq = tileq_slot(b, a.parity, f);
        for (int p = threadIdx.x; p < b.P2; p += BLOCK) {
            const uint64_t q = tq[p];
            all += q;
            if (p < tile) pre += q;
        }
    }
    if (fin && wvid == 0) {
        // combine the 8 shards of each word inside its group of 8 lanes: xor 1, xor 2 (quad_perm), xor 4 (half mirror)
#define LLPF_ACCSTEP(CTRL) { const uint64_t t = dpp_u64<CTRL, 0xF, false>(accv, accv); accv = (grp == 0) ? (t > accv ? t : accv) : accv + t; }
        LLPF_ACCSTEP(DPP_QUAD_XOR1) LLPF_ACCSTEP(DPP_QUAD_XOR2) LLPF_ACCSTEP(DPP_ROW_HALF_MIRROR)
#undef LLPF_ACCSTEP
        if (shard == 0) sh.accw[grp] = accv;
    }
    pre = wave_sum_u64(pre);
    all = wave_sum_u64(all);
    if (lane == 0) { sh.red[wvid][0] = pre; sh.red[wvid][1] = all; }
    __syncthreads();
    h.status = 0;
    h.s = 0.0;
    if (defer_skip) {
        const bool stopped = stop_flag != 0 && (int64_t)(stop_flag - 1) < a.k;
        if (stopped || (a.only_fallback ? !fb_flag : (fb_flag != 0))) { h.status = RES_STATUS_SKIP; return h; }
    }
    h.prefix = 0; h.tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) { h.prefix += sh.red[k][0]; h.tot += sh.red[k][1]; }
    if (fin) {
        // clear the slot after next (its last reader finished two launches ago): accumulator words and tile sums
        const int clr = (a.parity + 2) % ACC_NSLOT;
        if (tile == 0 && threadIdx.x >= 64 && threadIdx.x < 128) *acc_slot(acc, acc_word_of_group(grp - 8, clr), shard) = 0;
        uint64_t* tqc = tileq_slot(b, clr, f);
        if (gridDim.x == (unsigned)b.P2) { if (threadIdx.x == 0) tqc[tile] = 0; }
        else { for (int p = threadIdx.x; p < b.P2; p += BLOCK) tqc[p] = 0; }      // finalize-only launch: one block
    }
    if (fin) {
        h.fast = a.fast_head;
        h.mtrue = max_unkey(sh.accw[0]);
        const llpf_u128 s128 = acc_combine_u128(sh.accw[1], sh.accw[2], sh.accw[3]);
        const llpf_u128 e128 = acc_combine_u128(sh.accw[4], sh.accw[5], sh.accw[6]);
        const bool bad = sh.accw[7] != 0;
        if (h.fast) {
            h.a = off_pre;                                        // published by the weighting kernel that filled this slot
            if (bad || s128.hi < ((uint64_t)1 << 22)) {           // sum exp(w - bound) < 2^-10, or NaN weights
                h.status = RES_STATUS_FALLBACK;
                h.stot = 0.0;
            } else {
                h.stot = llpf_fix96_to_double(s128);
            }
        } else {
            h.a = h.mtrue;
            if (bad || s128.hi < ((uint64_t)1 << 32)) {           // max is -Inf / NaN, or NaN weights: degenerate
                h.stot = llpf_u2d(0x7ff8000000000000ULL);
                h.s = h.stot;
                h.status = LLPF_ERR_DEGENERATE;
            } else {
                h.s = llpf_fix96_to_double(llpf_fix96_minus_one(s128));     // sum_all_but: exact, one rounding
                h.stot = h.s + 1.0;
            }
        }
        const int e2v = e2v_pre;
        h.e2 = e2v ? llpf_fix96_to_double(e128) : -1.0;           // -1: not accumulated (threshold 1: not needed)
        h.uniform = 0;
        h.dr = h.status ? 0 : decide_resample(b.thr, Nd, h.stot, h.e2);
        if (tile == 0 && threadIdx.x == 0) {
            if (h.status == RES_STATUS_FALLBACK) {
                sc->fallback = 1;
                sc->fb_step = a.k;
                *b.bank_flag = (uint32_t)(a.k + 1);
            } else {
                double l, inv, ll, ess;
                if (h.status) { l = h.stot; inv = h.stot; ll = h.stot; ess = h.stot; }
                else {
                    l = head_log(h);
                    inv = 1.0 / h.stot;
                    ll = l + h.a;
                    ess = h.e2 > 0.0 ? (h.stot * h.stot) / h.e2 : -1.0;
                }
                sc->m = h.a; sc->mtrue = h.mtrue; sc->s = h.s; sc->stot = h.stot; sc->l = l; sc->inv = inv; sc->ll = ll;
                sc->ess = ess; sc->e2 = h.e2; sc->fast = h.fast; sc->e2_valid = e2v;
                sc->wmax = (h.mtrue - h.a) - l;                   // normalised weight of the best particle
                sc->K = a.K;
                sc->uniform = 0;
                sc->norm_pending = a.keep_norm ? 0 : 1;
                if (a.keep_norm) sc->wmax = h.mtrue;              // set_weights: w stays as installed
                if (h.status) sc->status = h.status;
                sc->do_resample = h.dr;
                if (a.accumulate) sc->ll_total = sc->ll_total + ll;
                if (a.ll_steps) a.ll_steps[(size_t)a.row * b.F + f] = ll;
                sh.dval[0] = inv;
            }
        }
        if (!h.status) h.status = status_pre;          // sticky until reset! (written above only when non-zero)
    } else {
        // predict! without a preceding correct! in this launch sequence: decide from the stored state
        h.a = sc->m; h.mtrue = sc->mtrue; h.s = sc->s; h.stot = sc->stot; h.e2 = sc->e2; h.fast = sc->fast;
        h.status = sc->status; h.uniform = (SRC == SRC_FILTER) ? sc->uniform : 0;
        if (h.uniform) {
            const double wev = 1.0 / Nd;
            const double ess = 1.0 / (Nd * (wev * wev));
            h.dr = (b.thr == 1.0) ? 1 : (ess < Nd * b.thr ? 1 : 0);
            if (tile == 0 && threadIdx.x == 0 && !a.only_bins) sc->ess = ess;
            const uint64_t Qc = llpf_q64_unit(1.0 / Nd, a.K);
            const int64_t before = (int64_t)tile * TILE < b.N ? (int64_t)tile * TILE : b.N;
            h.prefix = (uint64_t)before * Qc;
            h.tot = (uint64_t)b.N * Qc;
        } else {
            h.dr = h.status ? 0 : decide_resample(b.thr, Nd, h.stot, h.e2);
        }
        if (SRC == SRC_FILTER && tile == 0 && threadIdx.x == 0 && !a.only_bins) sc->do_resample = h.dr;
    }

    // weighted_mean output (fixed-order fp64 sum of the tile partials; tile 0 only)
    if (fin && a.xmean && tile == 0 && !h.status) {
        __syncthreads();
        const double invb = sh.dval[0];
        const double* xp = b.xmpart + (size_t)f * b.P1 * MAXD;
        const int nparts = sc->xm_parts;
        for (int d = 0; d < b.nx; ++d) {
            double accx = 0.0;
            for (int p = threadIdx.x; p < nparts; p += BLOCK) accx = accx + xp[(size_t)p * MAXD + d];
            accx = wave_sum_f64(accx);
            __syncthreads();
            if (lane == 0) sh.red[wvid][2] = llpf_d2u(accx);
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = llpf_u2d(sh.red[0][2]);
                for (int k = 1; k < BLOCK / 64; ++k) t = t + llpf_u2d(sh.red[k][2]);
                a.xmean[((size_t)a.row * b.F + f) * b.nx + d] = t * invb;
            }
        }
    }
    return h;
}

// Scan of this tile's quanta + ancestor counts.  On return sh.cl[k] = c(bins[k]) for the tile's TILE sources
// (after a __syncthreads), and [c_start, c_end) is the range of outputs this tile produces.
template <int STRATEGY>
DEV void res_counts(const BankDev& b, const ResArgs& a, int f, int tile, const ResHead& h, const ulonglong2* qv,
                    ResShared& sh, int32_t& c_start, int32_t& c_end) {
    const FilterScal* sc = b.scal + f;
    const int64_t N = b.N;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    uint64_t cq[NORM_IPT];
    {
        const uint64_t Qc = h.uniform ? llpf_q64_unit(1.0 / (double)N, a.K) : 0;
        uint64_t run = 0;
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            uint64_t q = h.uniform ? Qc : ((k & 1) ? qv[k / 2].y : qv[k / 2].x);
            if (ib + k >= N) q = 0;
            run += q;
            cq[k] = run;
        }
    }
    const uint64_t tsum = cq[NORM_IPT - 1];
    const uint64_t incl = wave_scan_u64(tsum);         // wave inclusive scan of thread totals
    if (lane == 63) sh.red[wvid][3] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k)
        if (k < wvid) wave_off += sh.red[k][3];
    const uint64_t excl = h.prefix + wave_off + (incl - tsum);

    // bins = fl(fl(cum) * fl(1/fl(total))) and ancestor counts
    const double Td = (double)h.tot;
    const double invTd = 1.0 / Td;
    const double binsN = Td * invTd;                   // bins[N]: 1 or 1 - 2^-53
    const int32_t M = a.M;
    uint32_t cnt[NORM_IPT];
    if (STRATEGY == LLPF_RESAMPLE_SYSTEMATIC) {
        ThrSys th;
        const double U = a.Uexp ? a.Uexp[0] : (a.u_from_scal ? sc->u_slot[a.parity] : llpf_uniform_step(a.step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1));
        th.M = M; th.Md = (double)M; th.step = 1.0 / (double)M;
        th.delta = 1e-9 + th.Md * 1e-13;
        th.r = U * binsN / (double)N;                  // r = rand()*bins[end]/N  (resample.jl:23)
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double bin = (double)(excl + cq[k]) * invTd;
            if (a.bins_out && ib + k < N) a.bins_out[(size_t)f * N + ib + k] = bin;
            cnt[k] = a.only_bins ? 0u : (uint32_t)th.count(bin);
        }
        c_start = a.only_bins ? 0 : th.count((double)h.prefix * invTd);
    } else {
        ThrStrat th;
        th.M = M; th.Md = (double)M; th.step = a.step; th.k0 = sc->k0; th.k1 = sc->k1; th.Uexp = a.Uexp;
        th.delta = 1e-9 + th.Md * 1e-13;
        th.binsN = binsN;
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double bin = (double)(excl + cq[k]) * invTd;
            if (a.bins_out && ib + k < N) a.bins_out[(size_t)f * N + ib + k] = bin;
            cnt[k] = a.only_bins ? 0u : (uint32_t)th.count(bin);
        }
        c_start = a.only_bins ? 0 : th.count((double)h.prefix * invTd);
    }
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) sh.cl[threadIdx.x * NORM_IPT + k] = cnt[k];
    __syncthreads();
    c_end = (int32_t)sh.cl[TILE - 1];
}

// output o (c_start <= o < c_end) is produced by the first source k of the tile with cl[k] > o
DEV int res_owner(const uint32_t* cl, int32_t o) {
    // branch-free descent in power-of-two steps: p4 = 4 * #{k : cl[k] <= o} (cl is non-decreasing and o < cl[TILE-1]).
    // The probe address is one VGPR (p4) + an immediate LDS offset, so a step is ds_read + compare + select + add.
    const uint32_t ov = (uint32_t)o;
    const char* base = reinterpret_cast<const char*>(cl);
    uint32_t p4 = 0;
#pragma unroll
    for (int step = TILE / 2; step >= 1; step >>= 1) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(base + p4 + (uint32_t)(step - 1) * 4u);
        p4 += (v <= ov) ? (uint32_t)step * 4u : 0u;
    }
    return (int)(p4 >> 2);
}
static_assert(TILE == 1024, "res_owner assumes 2^10 sources per tile");

template <int STRATEGY, int SRC>
__global__ __launch_bounds__(BLOCK) void k_resample(BankDev b, ResArgs a) {
    __shared__ ResShared sh;
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !b.scal[f].fallback : (b.scal[f].fallback != 0))) return;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
    if (a.mode & RES_RESAMPLE) {
#pragma unroll
        for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
    }
    const ResHead h = res_head<SRC>(b, a, f, tile, sh);
    if (!(a.mode & RES_RESAMPLE)) return;
    if (h.status) return;
    if (!a.force && !h.dr) return;
    if (h.tot == 0) return;
    int32_t c_start, c_end;
    res_counts<STRATEGY>(b, a, f, tile, h, qv, sh, c_start, c_end);
    if (a.only_bins) return;
    int32_t* ao = a.anc_out + (size_t)f * b.Ns;
    for (int32_t o = c_start + threadIdx.x; o < c_end; o += BLOCK)
        ao[o] = (int32_t)((int64_t)tile * TILE + res_owner(sh.cl, o));
    // outputs whose threshold is >= bins[N] are never written by the reference (j keeps its previous
    // value); the previous value is only materialised here if it was the identity 1:N
    if (tile == b.P2 - 1 && SRC == SRC_FILTER && b.scal[f].anc_ident_s[b.anc_slot]) {
        for (int32_t o = c_end + threadIdx.x; o < a.M; o += BLOCK) ao[o] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Residual resampling — resample(::Type{ResampleResidual}, we, j, bins, M), reference src/resample.jl:63-117.
// Device order (oracle/llpf_oracle.c:resample_residual): with the integer quanta q_i (total Q) the copy counts
// c_i = floor(q_i M / Q) and the residuals q_i M - c_i Q are exact; residuals are kept to K bits (rho_i = rem >> L,
// L = ceil(log2 N)) so that their cumulative sum fits 63 bits.  Outputs [0, num) are the deterministic copies in
// source order; outputs m in [num, M) draw u_m and take the first i with u_m < fl(fl(cumrho_i) fl(1/fl(totrho))).
//   k_resid_prep    head (finalize / decision) + per-tile totals of c and rho + within-tile cumulative rho (scratch)
//   k_resid_scan    inclusive prefixes of the per-tile totals (one block per filter)
//   k_resid_expand  deterministic copies via the counts machinery, multinomial part by two-level binary search
// ------------------------------------------------------------------------------------------------
DEV void resid_quanta(const BankDev& b, const ResArgs& a, const ResHead& h, const ulonglong2* qv, int64_t ib, uint64_t* q) {
    const uint64_t Qc = h.uniform ? llpf_q64_unit(1.0 / (double)b.N, a.K) : 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        uint64_t v = h.uniform ? Qc : ((k & 1) ? qv[k / 2].y : qv[k / 2].x);
        if (ib + k >= b.N) v = 0;
        q[k] = v;
    }
}

template <int SRC>
__global__ __launch_bounds__(BLOCK) void k_resid_prep(BankDev b, ResArgs a) {
    __shared__ ResShared sh;
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !b.scal[f].fallback : (b.scal[f].fallback != 0))) return;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
    const ResHead h = res_head<SRC>(b, a, f, tile, sh);
    if (h.status) return;
    if (!a.force && !h.dr) return;
    if (h.tot == 0) return;
    if (tile == 0 && threadIdx.x == 0) b.scal[f].totQ = h.tot;
    const int L = 62 - a.K;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    uint64_t q[NORM_IPT], rr[NORM_IPT];
    resid_quanta(b, a, h, qv, ib, q);
    uint64_t csum = 0, run = 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        uint64_t rem;
        csum += llpf_muldiv_floor(q[k], (uint64_t)a.M, h.tot, &rem);
        run += rem >> L;
        rr[k] = run;
    }
    const uint64_t incl = wave_scan_u64(run);
    const uint64_t ctot = wave_sum_u64(csum);
    __syncthreads();
    if (lane == 63) sh.red[wvid][3] = incl;
    if (lane == 0) sh.red[wvid][2] = ctot;
    __syncthreads();
    uint64_t wave_off = 0, rtot = 0, call = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) {
        if (k < wvid) wave_off += sh.red[k][3];
        rtot += sh.red[k][3];
        call += sh.red[k][2];
    }
    const uint64_t excl = wave_off + (incl - run);
    uint64_t* cum = b.quanta_next + (size_t)f * b.Ns;      // scratch: free between the scan and the next weighting
    ulonglong2 o0, o1;
    o0.x = excl + rr[0]; o0.y = excl + rr[1]; o1.x = excl + rr[2]; o1.y = excl + rr[3];
    *reinterpret_cast<ulonglong2*>(cum + ib) = o0;
    *reinterpret_cast<ulonglong2*>(cum + ib + 2) = o1;
    if (threadIdx.x == 0) {
        uint64_t* rt = b.rtile + (size_t)f * 2 * b.P2;
        rt[tile] = call;
        rt[b.P2 + tile] = rtot;
    }
}
static_assert(NORM_IPT == 4, "k_resid_prep stores four cumulative residuals per thread");

template <int SRC>
__global__ __launch_bounds__(BLOCK) void k_resid_scan(BankDev b, ResArgs a) {
    __shared__ uint64_t sm[BLOCK / 64][2];
    const int f = blockIdx.x;
    const FilterScal* sc = b.scal + f;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !sc->fallback : (sc->fallback != 0))) return;
    if (SRC == SRC_FILTER && (sc->status || (!a.force && !sc->do_resample))) return;
    uint64_t* rt = b.rtile + (size_t)f * 2 * b.P2;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    uint64_t carry_c = 0, carry_r = 0;
    for (int base = 0; base < b.P2; base += BLOCK) {
        const int p = base + threadIdx.x;
        const uint64_t c = p < b.P2 ? rt[p] : 0, r = p < b.P2 ? rt[b.P2 + p] : 0;
        const uint64_t ic = wave_scan_u64(c), ir = wave_scan_u64(r);
        __syncthreads();
        if (lane == 63) { sm[wvid][0] = ic; sm[wvid][1] = ir; }
        __syncthreads();
        uint64_t oc = carry_c, orr = carry_r, tc = 0, tr = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) {
            if (k < wvid) { oc += sm[k][0]; orr += sm[k][1]; }
            tc += sm[k][0]; tr += sm[k][1];
        }
        if (p < b.P2) { rt[p] = oc + ic; rt[b.P2 + p] = orr + ir; }
        carry_c += tc; carry_r += tr;
    }
}

template <int SRC>
__global__ __launch_bounds__(BLOCK) void k_resid_expand(BankDev b, ResArgs a) {
    __shared__ ResShared sh;
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const FilterScal* sc = b.scal + f;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !sc->fallback : (sc->fallback != 0))) return;
    if (SRC == SRC_FILTER && (sc->status || (!a.force && !sc->do_resample))) return;
    const uint64_t Q = sc->totQ;
    if (Q == 0) return;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const uint64_t* __restrict__ rt = b.rtile + (size_t)f * 2 * b.P2;
    const uint64_t* __restrict__ cum = b.quanta_next + (size_t)f * b.Ns;
    int32_t* ao = a.anc_out + (size_t)f * b.Ns;
    const int64_t M = a.M;
    // ---- deterministic copies: cl[k] = copies of sources 0 .. k (all tiles before this one included)
    ulonglong2 qv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
    ResHead hq;
    hq.uniform = (SRC == SRC_FILTER) ? sc->uniform : 0;
    uint64_t q[NORM_IPT], cc[NORM_IPT];
    resid_quanta(b, a, hq, qv, ib, q);
    uint64_t crun = 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        uint64_t rem;
        crun += llpf_muldiv_floor(q[k], (uint64_t)M, Q, &rem);
        cc[k] = crun;
    }
    const uint64_t incl = wave_scan_u64(crun);
    if (lane == 63) sh.red[wvid][3] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k)
        if (k < wvid) wave_off += sh.red[k][3];
    const uint64_t c_prev = tile > 0 ? rt[tile - 1] : 0;
    const uint64_t excl = c_prev + wave_off + (incl - crun);
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        const uint64_t v = excl + cc[k];
        sh.cl[threadIdx.x * NORM_IPT + k] = (uint32_t)(v > (uint64_t)M ? (uint64_t)M : v);
    }
    __syncthreads();
    const int64_t c_start = (int64_t)(c_prev > (uint64_t)M ? (uint64_t)M : c_prev);
    const int64_t c_end = (int64_t)sh.cl[TILE - 1];
    for (int64_t o = c_start + threadIdx.x; o < c_end; o += BLOCK)
        ao[o] = (int32_t)((int64_t)tile * TILE + res_owner(sh.cl, (int32_t)o));
    // ---- multinomial part: outputs [num, M), an equal share per tile
    const uint64_t num64 = rt[b.P2 - 1];
    const int64_t num = (int64_t)(num64 > (uint64_t)M ? (uint64_t)M : num64);
    const int64_t R = M - num;
    if (R <= 0) return;
    const uint64_t totr = rt[2 * b.P2 - 1];
    const int anc_ident = (SRC == SRC_FILTER) ? sc->anc_ident_s[b.anc_slot] : 0;
    const int64_t chunk = (R + b.P2 - 1) / b.P2;
    const int64_t m0 = num + (int64_t)tile * chunk;
    const int64_t m1 = (m0 + chunk < M) ? m0 + chunk : M;
    const double Td = (double)totr;
    const double invTd = 1.0 / Td;
    const uint64_t* __restrict__ pr = rt + b.P2;
    for (int64_t m = m0 + threadIdx.x; m < m1; m += BLOCK) {
        const double u = a.Uexp ? a.Uexp[m] : llpf_uniform_idx((uint32_t)m, a.step, LLPF_STREAM_STRATIFY, sc->k0, sc->k1);
        int64_t src = -1;
        if (totr != 0) {
            int lo = 0, hi = b.P2;                      // first tile t with u < bins(end of t)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (u < (double)pr[mid] * invTd) hi = mid; else lo = mid + 1;
            }
            if (lo < b.P2) {
                const uint64_t before = lo > 0 ? pr[lo - 1] : 0;
                const uint64_t* ct = cum + (size_t)lo * TILE;
                int kl = 0, kh = TILE - 1;              // the tile's last bin is > u, so an index exists
                while (kl < kh) {
                    const int mid = (kl + kh) >> 1;
                    if (u < (double)(before + ct[mid]) * invTd) kh = mid; else kl = mid + 1;
                }
                src = (int64_t)lo * TILE + kl;
            }
        }
        if (src >= 0) ao[m] = (int32_t)src;
        else if (anc_ident) ao[m] = (int32_t)m;        // u >= bins[N]: j[m] keeps its previous value (identity materialised)
    }
}

// ------------------------------------------------------------------------------------------------
// k_resprop — the fused predict!: finalize + shouldresample + resample + propagate [+ weight of the next
// correct!] in ONE launch.  A block owns a tile of 1024 SOURCE particles; it derives which outputs its sources
// produce ([c_start, c_end), from the ancestor counts) and propagates exactly those outputs, reading its
// sources' states (an 8 KB window per dimension: L1/L2 hits) and writing x, w (and j, kept for the accessor and
// for the reference's "stale j" corner) coalesced.  No ancestor array round trip, no separate propagate launch.
// Load balance: a tile produces ~1024 outputs +- a few % for i.i.d.-like weights; a tile holding very heavy
// particles loops over more 256-output chunks (worst case ESS -> 1: one block does everything; still far faster
// than the serial reference).
// The per-output arithmetic is the same sequence as k_step's, so fused and unfused paths are bit-identical.
// ------------------------------------------------------------------------------------------------
template <class T> DEV T ld_off(const T* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T> DEV void st_off(T* base, uint32_t byte_off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// placeholder model of the AuxiliaryParticleFilter's second half: the dynamics were applied by k_step<MODE_AUX>
template <int NX>
struct NoModel {
    static constexpr bool RB = false;
    DEV void prepare(const ModelD*, const double*, double) {}
    DEV void dynamics(const double* x, double* out) const {
#pragma unroll
        for (int d = 0; d < NX; ++d) out[d] = x[d];
    }
    DEV void measurement(const double*, double*) const {}
};

template <class Model, int NX, int NY, bool WEIGHT>
struct PropCtx {
    const BankDev& b;
    const Model& model;
    const ModelD* md;
    const StepArgs& st;
    const double* y;
    const double* __restrict__ xc;
    double* __restrict__ xn;
    double* w;
    uint32_t k0, k1;
    int ablate;
    double off;            // bound of the new weights (offset of their exp-sums)
    uint64_t* qnext;       // quanta of the new weights
    // propagate output o from source src with previous log-weight wprev; returns the new log-weight
    // Addresses are a uniform plane base (SGPRs) + a 32-bit byte offset (one VGPR): Ns * 8 < 2^32 is checked at create.
    DEV double one(uint32_t src, uint32_t o, double wprev, bool& bad, double* xs) const {
        const int64_t Ns = b.Ns;
        const uint32_t so = src << 3, oo = o << 3;
        double xp[NX], fx[NX], xi[NX], nz[NX];
#pragma unroll
        for (int d = 0; d < NX; ++d) xp[d] = ld_off(xc + (size_t)d * Ns, so);
#ifdef LLPF_DEVTOOLS   /* ablation switches for performance experiments (results invalid); not in production builds */
        if (!(ablate & 4)) model.dynamics(xp, fx);
        else { for (int d = 0; d < NX; ++d) fx[d] = xp[d]; }
        if (!(ablate & 1)) llpf_normals(o, st.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
        else { for (int d = 0; d < NX; ++d) xi[d] = 0.25 * (double)(o & 7); }
#else
        model.dynamics(xp, fx);
        llpf_normals(o, st.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
#endif
        gauss_sample<NX>(md->df, xi, nz);
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            xs[d] = fx[d] + nz[d];
            st_off(xn + (size_t)d * Ns, oo, xs[d]);
        }
        double wv = wprev;
        if (WEIGHT) {
#ifdef LLPF_DEVTOOLS
            if (st.has_y && !(ablate & 4)) {
#else
            if (st.has_y) {
#endif
                double g[NY], v[NY];
                model.measurement(xs, g);
#pragma unroll
                for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                wv = wv + gauss_logpdf<NY>(md->dg, v);
            }
            if (o >= (uint32_t)b.N) wv = -LLPF_INF;
            bad = bad || (wv != wv);
            st_off(w, oo, wv);
        }
        return wv;
    }
};

// per-thread running sum of quanta keyed by destination tile; flushed to LDS (first 8 tiles of the block's output
// range) or straight to the global tile sums (heavier blocks) whenever the tile changes
struct TileSum {
    int32_t tcur;
    uint64_t run;
    DEV void init() { tcur = -1; run = 0; }
    DEV void flush(uint64_t* sh_tq, uint64_t* tq_global, int32_t tbase) {
        if (run) {
            const int32_t idx = tcur - tbase;
            if (idx >= 0 && idx < 8) atomicAdd(reinterpret_cast<unsigned long long*>(sh_tq + idx), (unsigned long long)run);
            else atomicAdd(reinterpret_cast<unsigned long long*>(tq_global + tcur), (unsigned long long)run);
        }
        run = 0;
    }
    DEV void add(uint32_t o, uint64_t q, uint64_t* sh_tq, uint64_t* tq_global, int32_t tbase) {
        const int32_t t = (int32_t)(o >> 10);
        if (t != tcur) { flush(sh_tq, tq_global, tbase); tcur = t; }
        run += q;
    }
};
static_assert(TILE == 1024, "TileSum assumes 1024-particle tiles");

template <class Model, int NX, int NY, bool WEIGHT, bool ACC, bool AUX = false>
__global__ __launch_bounds__(BLOCK) void k_resprop(BankDev b, const ModelD* __restrict__ models, ResArgs a, StepArgs st) {
    __shared__ ResShared sh;
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_acc[BLOCK / 64][5];
    __shared__ uint64_t sh_tq[8];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const int64_t Ns = b.Ns, N = b.N;
    const ModelD* md = models + f;
    FilterScal* sc = b.scal + f;
    const uint32_t stop_flag = *b.bank_flag;           // tested in res_head, after all other loads are in flight
    const int fb_flag = sc->fallback;
    if (threadIdx.x < 8) sh_tq[threadIdx.x] = 0;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
    const int anc_ident_prev = sc->anc_ident_s[b.anc_slot];   // this launch writes the other entry
#define LLPF_STAMP(i) if (a.dbg && threadIdx.x == 0 && f == 0) a.dbg[(size_t)tile * 8 + (i)] = wall_clock64()
    LLPF_STAMP(0);
    Model model;                                       // particle-independent terms: their loads overlap the head's
    model.prepare(md, st.u, st.t_prop);
    double y[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) y[k] = (WEIGHT && st.has_y) ? st.y[k] : 0.0;
    const uint32_t key0 = sc->k0, key1 = sc->k1;
    const ResHead h = res_head<SRC_FILTER>(b, a, f, tile, sh, true, stop_flag, fb_flag);
    if (h.status) return;
    LLPF_STAMP(1);
    PropCtx<Model, NX, NY, WEIGHT> pc{b, model, md, st, y, b.xcur + (size_t)f * NX * Ns, b.xnext + (size_t)f * NX * Ns,
                                      b.w + (size_t)f * Ns, key0, key1, a.ablate, 0.0, b.quanta_next + (size_t)f * Ns};
    int32_t* anc = b.anc + (size_t)f * Ns;
    double bmax = -LLPF_INF;
    bool bad = false;

    // One loop over the outputs this block produces (the per-output body is instantiated once):
    //   resampling : outputs [c_start, c_end) from the ancestor counts, source = tile's owner of the output;
    //                the last tile also takes [c_end, M): thresholds >= bins[N], for which the reference leaves
    //                j[i] untouched (resample.jl:25-34) -> previous ancestor (identity if the last predict! did
    //                not resample)
    //   otherwise  : s.j .= 1:N, the tile's own particles (padding lanes included so that their weight stays -Inf)
    const bool res = (h.dr || a.force) && h.tot != 0;
    int64_t first, last;
    int32_t c_end = 0;
    double l = 0.0;
    WeightAcc wacc;
    TileSum ts;
    wacc.init();
    ts.init();
    double xm[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
    uint64_t* tq_next = tileq_slot(b, st.parity, f);
    if (res) {
        int32_t c_start;
        if (b.strategy == LLPF_RESAMPLE_SYSTEMATIC) res_counts<LLPF_RESAMPLE_SYSTEMATIC>(b, a, f, tile, h, qv, sh, c_start, c_end);
        else res_counts<LLPF_RESAMPLE_STRATIFIED>(b, a, f, tile, h, qv, sh, c_start, c_end);
        first = c_start;
        last = (tile == b.P2 - 1) ? (int64_t)a.M : (int64_t)c_end;
    } else {
        l = head_log(h);
        first = (int64_t)tile * TILE;
        last = first + TILE;
    }
    {   // bound of the weights produced below: max of the previous (normalised) weights + the density's peak
        const double wmx = res ? b.log1N : (h.mtrue - h.a) - l;
        pc.off = (WEIGHT && st.has_y) ? wmx + md->dg.c0 : wmx;
    }
    const double lN = -b.mlogN;
    const double aux_off = ((st.aux == 2) ? md->dg.c0 : 0.0) - lN;     // lambda - log N <= c0 - log N (lambda = 0 if y1 is missing)
    if (AUX) pc.off = aux_off;
    const double* lamp = AUX ? b.lam + (size_t)f * Ns : nullptr;
    const int32_t tbase = (int32_t)(first >> 10);
    LLPF_STAMP(2);
    if (a.dbg && threadIdx.x == 0 && f == 0) a.dbg[(size_t)tile * 8 + 5] = (uint64_t)(last - first);
    const uint32_t tile0 = (uint32_t)tile * TILE, ulast = (uint32_t)last, ucend = (uint32_t)c_end;
#pragma unroll 1
    for (uint32_t o = (uint32_t)first + threadIdx.x; o < ulast; o += BLOCK) {
        uint32_t src = o;
        double wprev = b.log1N;                                        // reset_weights!: w = log(1/N)
        if (res) {
#ifdef LLPF_DEVTOOLS
            if (o < ucend) src = tile0 + ((a.ablate & 2) ? ((o - (uint32_t)first) & (TILE - 1)) : (uint32_t)res_owner(sh.cl, (int32_t)o));
#else
            if (o < ucend) src = tile0 + (uint32_t)res_owner(sh.cl, (int32_t)o);
#endif
            else src = anc_ident_prev ? o : (uint32_t)ld_off(anc, o << 2);
            st_off(anc, o << 2, (int32_t)src);
            if (AUX) wprev = ld_off(lamp, o << 3) - lN;               // s.w[i] = lambda[i] - log N (unresampled index, filtering.jl:209-213)
        } else if (AUX) {
            wprev = ld_off(lamp, o << 3) - lN;
        } else if (WEIGHT) {
            wprev = (ld_off(pc.w, o << 3) - h.a) - l;                  // lazy w .-= offset ; w .-= log(sum)
        }
        double xs[NX];
        const double wv = pc.one(src, o, wprev, bad, xs);
        bmax = llpf_fmax(bmax, wv);
        if (WEIGHT && ACC) {
            double e;
            const uint64_t q = wacc.add(wv, pc.off, st.K, st.need_e2 != 0, &e);
            st_off(pc.qnext, o << 3, q);
            ts.add(o, q, sh_tq, tq_next, tbase);
            if (st.want_xmean) {
#pragma unroll
                for (int d = 0; d < NX; ++d) xm[d] = xm[d] + xs[d] * e;
            }
        }
    }
    if (WEIGHT && ACC) ts.flush(sh_tq, tq_next, tbase);
    LLPF_STAMP(3);
    if (WEIGHT) {
        const double r = block_max(bmax, sm_max);
        const int anybad = __syncthreads_or(bad ? 1 : 0);
        if (threadIdx.x == 0) acc_max(b.acc + (size_t)f * ACC_WORDS, st.parity, r, anybad != 0);
        if (ACC) {
            wacc.flush(b.acc + (size_t)f * ACC_WORDS, st.parity, st.need_e2 != 0, sm_acc);
            __syncthreads();
            if (threadIdx.x < 8 && sh_tq[threadIdx.x])
                atomicAdd(reinterpret_cast<unsigned long long*>(tq_next + tbase + threadIdx.x), (unsigned long long)sh_tq[threadIdx.x]);
            if (st.want_xmean) block_store_xm<NX>(xm, b.xmpart + ((size_t)f * b.P1 + tile) * MAXD, sm_x);
        }
        if (tile == 0 && threadIdx.x == 0) {
            if (AUX) {     // the weights just written are final values (no pending normalisation); aux_off bounds them
                sc->norm_pending = 0;
                sc->uniform = 0;
                sc->wmax = aux_off;
            }
            if (ACC) sc->xm_parts = b.P2;
            sc->off_slot[st.parity] = pc.off;
            sc->e2v_slot[st.parity] = st.need_e2;
            sc->u_slot[st.parity] = llpf_uniform_step(st.next_step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1);
        }
    }
    __syncthreads();
    LLPF_STAMP(4);
#undef LLPF_STAMP
    if (tile == b.P2 - 1 && threadIdx.x == 0) {        // bookkeeping of this predict! (by the only block that reads anc_ident)
        const int r = res ? 1 : 0;
        sc->anc_ident_s[b.anc_slot ^ 1] = r ? 0 : 1;
        sc->last_resampled = r;
        sc->resample_count += r;
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
        const uint64_t q = (i < b.N) ? llpf_q64_unit(w[i], K) : 0;
        b.quanta[(size_t)f * b.Ns + i] = q;
        Q += q;
    }
    Q = wave_sum_u64(Q);
    if ((threadIdx.x & 63) == 0) sm_w[threadIdx.x >> 6] = Q;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t q = 0;
        for (int k = 0; k < BLOCK / 64; ++k) q += sm_w[k];
        tileq_slot(b, 0, f)[tile] = q;      // scratch bank of the standalone resample(we): slot 0
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
        we = llpf_exp_le0(wr - sc->m) * sc->inv;
    }
    if (w_out) w_out[(size_t)f * b.N + i] = wv;
    if (we_out) we_out[(size_t)f * b.N + i] = we;
}

// ------------------------------------------------------------------------------------------------
// FFBS particle smoother, backward step t (reference src/smoothing.jl:128-141, draw_one_categorical
// src/resample.jl:128-152).  k_smooth_fx evaluates f(xf[n,t]) once; k_smooth_draw: one block per trajectory m,
//   wb[n] = wf[n,t] + logpdf(df, xb[m,t+1] - fx[n])  (recomputed in each of the three sweeps: max, total of the
//   quanta of exp(wb - max), count of bins below s = rand()*bins[end]), index = #{b : bins[b] < s}.
// ------------------------------------------------------------------------------------------------
template <class Model, int NX, int NY>
__global__ __launch_bounds__(BLOCK) void k_smooth_fx(BankDev b, const ModelD* __restrict__ models, SmoothArgs a) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    Model model;
    model.prepare(models, a.u, a.t);
    double xp[NX], fx[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xp[d] = a.xf_t[i * NX + d];
    model.dynamics(xp, fx);
#pragma unroll
    for (int d = 0; d < NX; ++d) a.fx[(size_t)d * b.Ns + i] = fx[d];
}

template <int NX>
__global__ __launch_bounds__(BLOCK) void k_smooth_draw(BankDev b, const ModelD* __restrict__ md, SmoothArgs a) {
    __shared__ double sm_d[BLOCK / 64];
    __shared__ uint64_t sm_u[BLOCK / 64];
    const int m = blockIdx.x;
    const int64_t N = b.N, Ns = b.Ns;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    double xq[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xq[d] = a.xb_next[(size_t)m * NX + d];
    auto wb = [&](int64_t n) {
        double v[NX];
#pragma unroll
        for (int d = 0; d < NX; ++d) v[d] = xq[d] - a.fx[(size_t)d * Ns + n];
        return a.wf_t[n] + gauss_logpdf<NX>(md->df, v);
    };
    // sweep 1: maximum
    double mx = -LLPF_INF;
    for (int64_t n = threadIdx.x; n < N; n += BLOCK) mx = llpf_fmax(mx, wb(n));
    mx = block_max(mx, sm_d);
    // sweep 2: total of the quanta
    const int K = llpf_qbits(N);
    uint64_t tot = 0;
    for (int64_t n = threadIdx.x; n < N; n += BLOCK) tot += llpf_q64_unit(llpf_exp_le0(wb(n) - mx), K);
    tot = wave_sum_u64(tot);
    __syncthreads();
    if (lane == 0) sm_u[wvid] = tot;
    __syncthreads();
    tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) tot += sm_u[k];
    // sweep 3: bins = fl(fl(cum) * fl(1/fl(total))) in index order; count those below s
    const FilterScal* sc = b.scal;
    const double u = llpf_uniform_idx((uint32_t)m, a.step, LLPF_STREAM_SMOOTH, sc->k0, sc->k1);
    const double Td = (double)tot, invTd = 1.0 / Td;
    const double s = u * (Td * invTd);
    uint64_t carry = 0, count = 0;
    for (int64_t base = 0; base < N; base += (int64_t)BLOCK * 4) {
        const int64_t n0 = base + (int64_t)threadIdx.x * 4;
        uint64_t c[4], run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t n = n0 + k;
            run += (n < N) ? llpf_q64_unit(llpf_exp_le0(wb(n < N ? n : N - 1) - mx), K) : 0;
            c[k] = run;
        }
        const uint64_t incl = wave_scan_u64(run);
        __syncthreads();
        if (lane == 63) sm_u[wvid] = incl;
        __syncthreads();
        uint64_t off = carry, all = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) {
            if (k < wvid) off += sm_u[k];
            all += sm_u[k];
        }
        const uint64_t excl = off + (incl - run);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (n0 + k < N && (double)(excl + c[k]) * invTd < s) ++count;
        carry += all;
    }
    count = wave_sum_u64(count);
    __syncthreads();
    if (lane == 0) sm_u[wvid] = count;
    __syncthreads();
    count = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) count += sm_u[k];
    const int64_t idx = (int64_t)count < N ? (int64_t)count : N - 1;     // nothing found: length(bins), resample.jl:151
    if (threadIdx.x == 0 && a.idx_t) a.idx_t[m] = idx;
    if (threadIdx.x < NX) a.xb_t[(size_t)m * NX + threadIdx.x] = a.xf_t[idx * NX + threadIdx.x];
}

// w[] <- the values it stands for (uniform constant / lazily normalised / as stored); padding lanes -Inf.  The
// host clears the `uniform` / `norm_pending` flags afterwards.
__global__ __launch_bounds__(BLOCK) void k_bake_weights(BankDev b) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    const FilterScal* sc = b.scal + f;
    double* w = b.w + (size_t)f * b.Ns;
    double wv = -LLPF_INF;
    if (i < b.N) wv = sc->uniform ? sc->wconst : (sc->norm_pending ? (w[i] - sc->m) - sc->l : w[i]);
    w[i] = wv;
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
    dst[(size_t)f * b.N + i] = sc->anc_ident_s[b.anc_slot] ? i : (int64_t)b.anc[(size_t)f * b.Ns + i];
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
        const double we = sc->uniform ? 1.0 / (double)b.N : llpf_exp_le0(wr - sc->m) * sc->inv;
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
        case 8: r = llpf_exp_le0(x); break;
        case 9: r = llpf_log_unit(x); break;
        case 10: llpf_sincos2pi_fast(x, &s, &c); r = s; break;
        case 11: llpf_sincos2pi_fast(x, &s, &c); r = c; break;
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
    if (model_id == LLPF_MODEL_RB_LINEAR) return nx >= 2 && nx <= 4 && ny >= 1 && ny <= 4;
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
        case MODE_AUX: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_AUX>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
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

template <int NX>
static hipError_t launch_step_rb_ny(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    if (mode == MODE_AUX) return hipErrorInvalidValue;
    switch (b.ny) {
        case 1: return launch_step_t<RBLin<NX, 1>, NX, 1>(b, mode, a, s);
        case 2: return launch_step_t<RBLin<NX, 2>, NX, 2>(b, mode, a, s);
        case 3: return launch_step_t<RBLin<NX, 3>, NX, 3>(b, mode, a, s);
        case 4: return launch_step_t<RBLin<NX, 4>, NX, 4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    const int model_id = b.model_id;
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return launch_step_t<QuadTank<4, 2>, 4, 2>(b, mode, a, s);
    if (model_id == LLPF_MODEL_RB_LINEAR) {
        switch (b.nx) {
            case 2: return launch_step_rb_ny<2>(b, mode, a, s);
            case 3: return launch_step_rb_ny<3>(b, mode, a, s);
            case 4: return launch_step_rb_ny<4>(b, mode, a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (b.nx) {
        case 1: return launch_step_lg_ny<1>(b, mode, a, s);
        case 2: return launch_step_lg_ny<2>(b, mode, a, s);
        case 3: return launch_step_lg_ny<3>(b, mode, a, s);
        case 4: return launch_step_lg_ny<4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_max(const BankDev& b, int parity, hipStream_t s) {
    hipLaunchKernelGGL(k_max, dim3((unsigned)b.P1, (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, parity);
    return hipGetLastError();
}

template <int NX, bool XMEAN>
static void launch_norm_e2(const BankDev& b, int parity, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const int K = llpf_qbits(b.N);
    if (need_e2) hipLaunchKernelGGL((k_norm<NX, XMEAN, true>), g, dim3(BLOCK), 0, s, b, K, parity, step, only_fallback, bound, kstep);
    else hipLaunchKernelGGL((k_norm<NX, XMEAN, false>), g, dim3(BLOCK), 0, s, b, K, parity, step, only_fallback, bound, kstep);
}
hipError_t launch_norm(const BankDev& b, int parity, int want_xmean, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s) {
    if (!want_xmean) { launch_norm_e2<0, false>(b, parity, need_e2, step, only_fallback, bound, kstep, s); return hipGetLastError(); }
    switch (b.nx) {
        case 1: launch_norm_e2<1, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 2: launch_norm_e2<2, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 3: launch_norm_e2<3, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 4: launch_norm_e2<4, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_ess(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_ess, dim3((unsigned)b.F), dim3(BLOCK), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_post_predict(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_post_predict, dim3((unsigned)((b.F + 63) / 64)), dim3(64), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_resample(const BankDev& b, const ResArgs& a0, hipStream_t s) {
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    // a finalize-only launch needs just one block per filter
    dim3 g((a.mode & RES_RESAMPLE) ? (unsigned)b.P2 : 1u, (unsigned)b.F, 1);
    if (a.src_values) hipLaunchKernelGGL(k_qpart, dim3((unsigned)b.P2, (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, a.K);
    if (b.strategy == LLPF_RESAMPLE_RESIDUAL && (a.mode & RES_RESAMPLE) && !a.only_bins) {
        const dim3 gt((unsigned)b.P2, (unsigned)b.F, 1), gf((unsigned)b.F, 1, 1);
        if (!a.src_values) {
            hipLaunchKernelGGL((k_resid_prep<SRC_FILTER>), gt, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_scan<SRC_FILTER>), gf, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_expand<SRC_FILTER>), gt, dim3(BLOCK), 0, s, b, a);
        } else {
            hipLaunchKernelGGL((k_resid_prep<SRC_VALUES>), gt, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_scan<SRC_VALUES>), gf, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_expand<SRC_VALUES>), gt, dim3(BLOCK), 0, s, b, a);
        }
        return hipGetLastError();
    }
    if (b.strategy == LLPF_RESAMPLE_SYSTEMATIC) {
        if (!a.src_values) hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_SYSTEMATIC, SRC_FILTER>), g, dim3(BLOCK), 0, s, b, a);
        else hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_SYSTEMATIC, SRC_VALUES>), g, dim3(BLOCK), 0, s, b, a);
    } else {
        if (!a.src_values) hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_STRATIFIED, SRC_FILTER>), g, dim3(BLOCK), 0, s, b, a);
        else hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_STRATIFIED, SRC_VALUES>), g, dim3(BLOCK), 0, s, b, a);
    }
    return hipGetLastError();
}

template <class Model, int NX, int NY>
static hipError_t launch_resprop_t(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    if (st.aux) hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true>), g, dim3(BLOCK), 0, s, b, b.models, a, st);
    else if (weight && st.accumulate) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, true>), g, dim3(BLOCK), 0, s, b, b.models, a, st);
    else if (weight) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, false>), g, dim3(BLOCK), 0, s, b, b.models, a, st);
    else hipLaunchKernelGGL((k_resprop<Model, NX, NY, false, false>), g, dim3(BLOCK), 0, s, b, b.models, a, st);
    return hipGetLastError();
}
template <int NX>
static hipError_t launch_resprop_lg_ny(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    switch (b.ny) {
        case 1: return launch_resprop_t<LinGauss<NX, 1>, NX, 1>(b, a, st, weight, s);
        case 2: return launch_resprop_t<LinGauss<NX, 2>, NX, 2>(b, a, st, weight, s);
        case 3: return launch_resprop_t<LinGauss<NX, 3>, NX, 3>(b, a, st, weight, s);
        case 4: return launch_resprop_t<LinGauss<NX, 4>, NX, 4>(b, a, st, weight, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_resprop(const BankDev& b, const ResArgs& a0, const StepArgs& st, int weight, hipStream_t s) {
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    a.mode = RES_FINALIZE | RES_RESAMPLE;
    if (b.model_id == LLPF_MODEL_QUADTANK_RK4) return launch_resprop_t<QuadTank<4, 2>, 4, 2>(b, a, st, weight, s);
    switch (b.nx) {
        case 1: return launch_resprop_lg_ny<1>(b, a, st, weight, s);
        case 2: return launch_resprop_lg_ny<2>(b, a, st, weight, s);
        case 3: return launch_resprop_lg_ny<3>(b, a, st, weight, s);
        case 4: return launch_resprop_lg_ny<4>(b, a, st, weight, s);
        default: return hipErrorInvalidValue;
    }
}

template <class Model, int NX, int NY>
static hipError_t launch_smooth_fx_t(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((k_smooth_fx<Model, NX, NY>), grid1(b.N, 1), dim3(BLOCK), 0, s, b, b.models, a);
    return hipGetLastError();
}
template <int NX>
static hipError_t launch_smooth_fx_ny(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    switch (b.ny) {
        case 1: return launch_smooth_fx_t<LinGauss<NX, 1>, NX, 1>(b, a, s);
        case 2: return launch_smooth_fx_t<LinGauss<NX, 2>, NX, 2>(b, a, s);
        case 3: return launch_smooth_fx_t<LinGauss<NX, 3>, NX, 3>(b, a, s);
        case 4: return launch_smooth_fx_t<LinGauss<NX, 4>, NX, 4>(b, a, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_smooth_fx(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    if (b.model_id == LLPF_MODEL_QUADTANK_RK4) return launch_smooth_fx_t<QuadTank<4, 2>, 4, 2>(b, a, s);
    switch (b.nx) {
        case 1: return launch_smooth_fx_ny<1>(b, a, s);
        case 2: return launch_smooth_fx_ny<2>(b, a, s);
        case 3: return launch_smooth_fx_ny<3>(b, a, s);
        case 4: return launch_smooth_fx_ny<4>(b, a, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_smooth_draw(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    const dim3 g((unsigned)a.M, 1, 1);
    switch (b.nx) {
        case 1: hipLaunchKernelGGL((k_smooth_draw<1>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 2: hipLaunchKernelGGL((k_smooth_draw<2>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 3: hipLaunchKernelGGL((k_smooth_draw<3>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 4: hipLaunchKernelGGL((k_smooth_draw<4>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_bake_weights(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_bake_weights, grid1(b.Ns, b.F), dim3(BLOCK), 0, s, b);
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
