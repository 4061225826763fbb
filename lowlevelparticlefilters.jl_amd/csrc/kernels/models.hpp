// kernels/models.hpp — Gaussian densities and the model structs (LinGauss, RBLin, QuadTank).  Part of kernels.hip (one translation unit, namespace llpf).
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

// gauss_sample with the covariance kind tested ONCE and the (uniform) operands through scalar loads: inside the unrolled loop
// every dimension was a branch with its loads behind it — a memory round trip per dimension (k_rbfull).  Same operations.
typedef const __attribute__((address_space(4))) GaussD* gauss_cptr;
template <int ND>
DEV void gauss_sample_c(gauss_cptr g, const double* xi, double* out) {
    const int kind = g->kind;
    if (kind == LLPF_COV_SCAL) {
        const double s = g->sqrtscal;
#pragma unroll
        for (int i = 0; i < ND; ++i) out[i] = s * xi[i] + g->mu[i];
    } else if (kind == LLPF_COV_DIAG) {
#pragma unroll
        for (int i = 0; i < ND; ++i) out[i] = g->sqrtdiag[i] * xi[i] + g->mu[i];
    } else {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            double v = g->L[i * MAXD + 0] * xi[0];
#pragma unroll
            for (int j = 1; j <= i; ++j) v = v + g->L[i * MAXD + j] * xi[j];
            out[i] = v + g->mu[i];
        }
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

template <int ND>
DEV double gauss_logpdf_c(gauss_cptr g, const double* x) {   // gauss_logpdf with the (uniform) operands through scalar loads
    double d[ND], q;
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = x[i] - g->mu[i];
    const int kind = g->kind;
    if (kind == LLPF_COV_SCAL) {
        double dot = d[0] * d[0];
#pragma unroll
        for (int i = 1; i < ND; ++i) dot = dot + d[i] * d[i];
        q = dot * g->invscal;
    } else if (kind == LLPF_COV_DIAG) {
        double s = (d[0] * d[0]) * g->invdiag[0];
#pragma unroll
        for (int i = 1; i < ND; ++i) s = s + (d[i] * d[i]) * g->invdiag[i];
        q = s;
    } else {
        double z[ND], z2[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            double acc = d[i];
#pragma unroll
            for (int j = 0; j < i; ++j) acc = acc - g->L[i * MAXD + j] * z[j];
            z[i] = acc * g->invLd[i];
        }
#pragma unroll
        for (int i = ND - 1; i >= 0; --i) {
            double acc = z[i];
#pragma unroll
            for (int j = i + 1; j < ND; ++j) acc = acc - g->L[j * MAXD + i] * z2[j];
            z2[i] = acc * g->invLd[i];
        }
        double dot = d[0] * z2[0];
#pragma unroll
        for (int i = 1; i < ND; ++i) dot = dot + d[i] * z2[i];
        q = dot;
    }
    return g->c0 - q / 2.0;
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
        // the input row first, as independent loads (no wait between them), then B u in the reference's order
        // (measured: ONE block with every operand requested at once and the columns selected is slower, C2 21.14 against 20.7 us)
        double ur[MAXU];
#pragma unroll
        for (int c = 0; c < MAXU; ++c) ur[c] = (has_u && c < nu) ? u[c] : 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            double acc = 0.0;
            if (has_u) {
                acc = m->B[r * nu + 0] * ur[0];
#pragma unroll
                for (int c = 1; c < MAXU; ++c)
                    if (c < nu) acc = acc + m->B[r * nu + c] * ur[c];
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
// NN: the number of nonlinear states as a compile-time constant (round 6; the fused kernel's instantiations), or 0: read from the model at
// run time (k_step: every split of NX in one kernel).  With the run-time split the generator's loop `for (2 b < nn)` cannot be unrolled, its
// output array is indexed dynamically and lives in scratch memory (32-48 bytes per lane in every k_resprop<RBLin> of rounds 1-5), and every
// inner product runs over all NX columns under a predicate.  Same operations in the same order either way: the same bits.
template <int NX, int NY, int NN = 0>
struct RBLin {
    static constexpr bool RB = true;
    static constexpr bool LEAN = NN > 0;      // split known at compile time: the kernel can afford what the plain model's kernel has (LDS tables, owner table)
    static_assert(NN >= 0 && NN < NX, "at least one linear state");
    const ModelD* md;
    const double* u;
    int nn_rt, nu;
    DEV void prepare(const ModelD* m, const double* __restrict__ uu, double /*t*/) {
        md = m; u = uu; nn_rt = m->nxn; nu = (uu != nullptr) ? m->nu : 0;
    }
    // the propagation of predict! (:185-224): xs = [fi + z ; Al xl + Bl u + L (z - An xl)]
    // lg / sc: the block's copy of the generator's tables in LDS, or nullptr (constant memory)
    DEV void rb_propagate(const double* xp, uint32_t idx, uint32_t step, uint32_t k0, uint32_t k1, const RBStep* rp, double* xs,
                          const double* lg = nullptr, const double* sc = nullptr) const {
        const int nn = NN > 0 ? NN : nn_rt, nl = NX - nn;
        double xi[NX], nz[NX], fi[NX], xl1[NX];
        if (lg) llpf_normals_tab(idx, step, LLPF_STREAM_DYNAMICS, k0, k1, nn, xi, lg, sc);
        else llpf_normals(idx, step, LLPF_STREAM_DYNAMICS, k0, k1, nn, xi);
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
        const int nn = NN > 0 ? NN : nn_rt, nl = NX - nn;
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
    double tg, eps, tsw, u0, u1, t0, Ts, Ts2, Ts6;
    int ss;
    DEV void prepare(const ModelD* m, const double* __restrict__ u, double t) {
        // the eleven quotients (-a/A, a/A, gamma k / A) and the step sizes come from the host (ModelD::qtc, host/densities.hpp):
        // an fp64 division is a ~30-instruction expansion, and these were particle-independent
        const double* c = m->qtc;
        c1a = c[QTC_1A]; c1a_sw = c[QTC_1A_SW]; c1b = c[QTC_1B]; c1u = c[QTC_1U];
        c2a = c[QTC_2A]; c2b = c[QTC_2B]; c2u = c[QTC_2U];
        c3a = c[QTC_3A]; c3u = c[QTC_3U]; c4a = c[QTC_4A]; c4u = c[QTC_4U];
        tg = c[QTC_TG];
        eps = m->qt[LLPF_QT_EPS];
        tsw = m->qt[LLPF_QT_TSWITCH];
        u0 = u[0];
        u1 = u[1];
        t0 = t;
        ss = m->supersample < 1 ? 1 : m->supersample;
        Ts = c[QTC_H]; Ts2 = c[QTC_H2]; Ts6 = c[QTC_H6];
    }
    DEV void rhs(const double* h, double t, double* xd) const {
        double s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double v = tg * h[i];
            s[i] = llpf_sqrt_pos((v > 0.0 ? v : 0.0) + eps);        // argument >= eps = 1e-3 (ssqrt, example_quadtank.jl:19)
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
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + Ts2 * f1[i];
            rhs(xt, t + Ts2, f2);
#pragma unroll
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + Ts2 * f2[i];
            rhs(xt, t + Ts2, f3);
#pragma unroll
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + Ts * f3[i];
            rhs(xt, t + Ts, f4);
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = x[i] + Ts6 * (((f1[i] + 2.0 * f2[i]) + 2.0 * f3[i]) + f4[i]);
            t = t + Ts;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    DEV void measurement(const double* x, double* out) const {
        out[0] = x[0];
        out[1] = x[1];
    }
    // The same RK4 with the four states on the four lanes of a quad (lane & 3 = state): every lane runs the chain of ITS state — the same
    // IEEE operations in the same order as dynamics() — and the one cross term of the right-hand side (h1' takes sqrt(h3), h2' takes
    // sqrt(h4)) arrives through a quad permute.  One evaluation is a third of the dependent instructions: for the launches in which a
    // handful of lanes evaluate f while everything else waits (k_resample_fx: ~8 surviving sources per tile at BASELINE C3).
    // All four lanes of the quad must be active.  Returns the lane's own component of f(x).
    DEV double dynamics_quad(double xc, int c) const {
        const double A = c == 0 ? c1a : (c == 1 ? c2a : (c == 2 ? c3a : c4a));
        const double Asw = c == 0 ? c1a_sw : A;
        const double B = c == 0 ? c1b : c2b;
        const double pu = c == 0 ? c1u * u0 : (c == 1 ? c2u * u1 : (c == 2 ? c3u * u1 : c4u * u0));
        auto rhs1 = [&](double h, double t) {
            const double v = tg * h;
            const double sq = llpf_sqrt_pos((v > 0.0 ? v : 0.0) + eps);
            const double so = llpf_u2d(dpp_u64<0xEE /* quad_perm [2,3,2,3] */, 0xF, false>(llpf_d2u(sq), llpf_d2u(sq)));
            const double a = (t > tsw) ? Asw : A;
            const double t1 = a * sq;
            return c < 2 ? (t1 + B * so) + pu : t1 + pu;
        };
        double x = xc, t = t0;
        for (int it = 0; it < ss; ++it) {
            const double f1 = rhs1(x, t);
            const double f2 = rhs1(x + Ts2 * f1, t + Ts2);
            const double f3 = rhs1(x + Ts2 * f2, t + Ts2);
            const double f4 = rhs1(x + Ts * f3, t + Ts);
            x = x + Ts6 * (((f1 + 2.0 * f2) + 2.0 * f3) + f4);
            t = t + Ts;
        }
        return x;
    }
};

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

// Optional hooks of a model (run-time compiled user models, kernels/jit.hpp) for the random part of the step:
//   DEV void noise(const double* x, const double* fx, const double* xi, const double* uu, double* out) const
//        the NEXT state of a particle whose previous state is x and whose noise-free prediction is fx = dynamics(x), given nx standard
//        normals xi and nx uniforms uu in [0, 1) of the particle's own Philox streams — replaces out = fx + (mu + L xi) of the Gaussian
//        descriptor: state-dependent / multiplicative noise, heavy tails, anything.  This is the reference's AdvancedParticleFilter
//        contract, dynamics(x, u, p, t, noise = true) "adds its own noise" (src/PFtypes.jl:242-259), and a ParticleFilter whose
//        dynamics_density is not Gaussian (rand!(rng, d, noise), src/PFtypes.jl:122-139)
//   DEV void initial(const double* xi, const double* uu, double* out) const
//        one draw of the initial density (reset! / the constructor: x_i = rand(rng, initial_density), src/filtering.jl:4-14)
// detected without <type_traits> (hiprtc has no system headers)
// a Rao-Blackwellized model whose nonlinear / linear split is a compile-time constant (RBLin<NX, NY, NN > 0>)
template <class M, class = void> struct model_lean { static constexpr bool value = false; };
template <class M> struct model_lean<M, decltype((void)M::LEAN)> { static constexpr bool value = M::LEAN; };
template <class M, class = void> struct has_user_noise { static constexpr bool value = false; };
template <class M> struct has_user_noise<M, decltype((void)&M::noise)> { static constexpr bool value = true; };
template <class M, class = void> struct has_user_initial { static constexpr bool value = false; };
template <class M> struct has_user_initial<M, decltype((void)&M::initial)> { static constexpr bool value = true; };

// Is f(x) expensive enough to be computed once per distinct ancestor of a block and handed to the outputs that share it
// (k_step)?  Yes unless the model says otherwise: run-time compiled user models and the quad-tank's RK4 are; a matrix-vector
// product is not, and a model that forms its own noise needs x next to f(x) anyway.
template <class Model> struct share_dynamics { static constexpr bool value = !has_user_noise<Model>::value; };
template <int NX, int NY> struct share_dynamics<LinGauss<NX, NY>> { static constexpr bool value = false; };
template <int NX, int NY> struct share_dynamics<RBLin<NX, NY>> { static constexpr bool value = false; };
template <int NX> struct share_dynamics<NoModel<NX>> { static constexpr bool value = false; };
// Can the run loop of this model take the source-side form of the balanced timestep (k_resample_fx + k_step<..., MARKS>,
// kernels/resfx.hpp)?  The models whose dynamics are worth a table.  (The linear-Gaussian model is not: measured at nx = 3..8 the form
// costs it 60-130 % — f is a matrix-vector product and with healthy weights half of the particles survive — tools/bench_nx.py.)
template <class Model> struct marks_path { static constexpr bool value = share_dynamics<Model>::value && !Model::RB; };

// a model whose dynamics can also run with the states spread over the lanes of a quad (NX == 4): dynamics_quad(x_own, lane & 3)
template <class M, class = void> struct has_quad_dynamics { static constexpr bool value = false; };
template <class M> struct has_quad_dynamics<M, decltype((void)&M::dynamics_quad)> { static constexpr bool value = true; };

// Optional hooks of a model (run-time compiled user models, kernels/jit.hpp):
//   DEV double loglik(const double* x, const double* y, double t) const   log p(y | x) — the reference's measurement_likelihood(x, u, y, p, t)
//                                                                          (src/PFtypes.jl:226-239) or logpdf of ANY measurement density
//                                                                          (ext/LowLevelParticleFiltersDistributionsExt.jl:80); replaces the
//                                                                          Gaussian logpdf(dg, y - g(x)) of the descriptor
//   DEV double loglik_bound() const                                        an upper bound of loglik over x and y for these parameters (what the
//                                                                          bound-offset normalisation needs); without it every
//                                                                          step is normalised in the exact-max form
// detected here without <type_traits> (hiprtc has no system headers)
template <class M, class = void> struct has_loglik { static constexpr bool value = false; };
template <class M> struct has_loglik<M, decltype((void)&M::loglik)> { static constexpr bool value = true; };
template <class M, class = void> struct has_loglik_bound { static constexpr bool value = false; };
template <class M> struct has_loglik_bound<M, decltype((void)&M::loglik_bound)> { static constexpr bool value = true; };
constexpr double LLPF_NO_BOUND = 1e300;   // "bound" of a likelihood that declares none: every bound test fails, every step takes the exact form
