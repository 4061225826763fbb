"""Device-source snippets a USER of the C ABI would hand to llpf_model_compile (include/llpf.h): the Model concept of
csrc/kernels/models.hpp written out for two systems.  The quad-tank one restates examples/example_quadtank.jl:8-35 through rk4 of
src/utils.jl:220-237 with the constants taken from the parameter block (m->qt); it evaluates the same expressions in the same
order as the engine's built-in model, so the two must agree to the bit — which is the test."""

QUADTANK_SRC = r'''
struct UserModel {
    static constexpr bool RB = false;
    double c1a, c1a_sw, c1b, c1u, c2a, c2b, c2u, c3a, c3u, c4a, c4u, tg, eps, tsw, u0, u1, t0, h;
    int ss;
    DEV void prepare(const ModelD* m, const double* u, double t) {
        const double* q = m->qt;        // k1 k2 g A1..A4 a1..a4 gamma1 gamma2 tswitch a1factor eps
        c1a = (-q[7]) / q[3];   c1a_sw = (-(q[7] * q[14])) / q[3];   c1b = q[9] / q[3];    c1u = (q[11] * q[0]) / q[3];
        c2a = (-q[8]) / q[4];   c2b = q[10] / q[4];                  c2u = (q[12] * q[1]) / q[4];
        c3a = (-q[9]) / q[5];   c3u = ((1.0 - q[12]) * q[1]) / q[5];
        c4a = (-q[10]) / q[6];  c4u = ((1.0 - q[11]) * q[0]) / q[6];
        tg = 2.0 * q[2];  eps = q[15];  tsw = q[13];
        u0 = u[0];  u1 = u[1];  t0 = t;
        ss = m->supersample < 1 ? 1 : m->supersample;
        h = m->Ts / (double)ss;
    }
    DEV void rhs(const double* x, double t, double* xd) const {
        double s[4];
        for (int i = 0; i < 4; ++i) { const double v = tg * x[i]; s[i] = llpf_sqrt_pos((v > 0.0 ? v : 0.0) + eps); }
        const double ca = (t > tsw) ? c1a_sw : c1a;
        xd[0] = ca * s[0] + c1b * s[2] + c1u * u0;
        xd[1] = c2a * s[1] + c2b * s[3] + c2u * u1;
        xd[2] = c3a * s[2] + c3u * u1;
        xd[3] = c4a * s[3] + c4u * u0;
    }
    DEV void dynamics(const double* x0, double* out) const {
        double x[4], f1[4], f2[4], f3[4], f4[4], xt[4];
        double t = t0;
        for (int i = 0; i < 4; ++i) x[i] = x0[i];
        for (int it = 0; it < ss; ++it) {
            rhs(x, t, f1);
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + (h / 2.0) * f1[i];
            rhs(xt, t + h / 2.0, f2);
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + (h / 2.0) * f2[i];
            rhs(xt, t + h / 2.0, f3);
            for (int i = 0; i < 4; ++i) xt[i] = x[i] + h * f3[i];
            rhs(xt, t + h, f4);
            for (int i = 0; i < 4; ++i) x[i] = x[i] + (h / 6.0) * (((f1[i] + 2.0 * f2[i]) + 2.0 * f3[i]) + f4[i]);
            t = t + h;
        }
        for (int i = 0; i < 4; ++i) out[i] = x[i];
    }
    DEV void measurement(const double* x, double* out) const { out[0] = x[0]; out[1] = x[1]; }
};
'''

# a model the engine has no built-in for: a pendulum with a cubic damper, Euler-discretised, measured through sin(angle)
PENDULUM_SRC = r'''
struct UserModel {
    static constexpr bool RB = false;
    double g_over_l, damp, dt, torque;
    DEV void prepare(const ModelD* m, const double* u, double t) {
        g_over_l = m->qt[0]; damp = m->qt[1]; dt = m->Ts; torque = (m->nu > 0 && u) ? u[0] : 0.0;
    }
    DEV void dynamics(const double* x, double* out) const {
        double sn, cs;
        const double turns = x[0] * 0.15915494309189535;            // angle / (2 pi)
        llpf_sincos2pi(turns - llpf_rint(turns) < 0.0 ? turns - llpf_rint(turns) + 1.0 : turns - llpf_rint(turns), &sn, &cs);
        out[0] = x[0] + dt * x[1];
        out[1] = x[1] + dt * (torque - g_over_l * sn - damp * x[1] * x[1] * x[1]);
    }
    DEV void measurement(const double* x, double* out) const {
        double sn, cs;
        const double turns = x[0] * 0.15915494309189535;
        llpf_sincos2pi(turns - llpf_rint(turns) < 0.0 ? turns - llpf_rint(turns) + 1.0 : turns - llpf_rint(turns), &sn, &cs);
        out[0] = sn;
    }
};
'''

# A linear system (A, B, C of the parameter block: the same expressions, in the same order, as the engine's built-in linear-Gaussian
# model) with a measurement LIKELIHOOD of its own — the reference's measurement_likelihood(x, u, y, p, t) callable of an
# AdvancedParticleFilter (src/PFtypes.jl:226-239), or a non-Gaussian measurement density of a ParticleFilter
# (ext/LowLevelParticleFiltersDistributionsExt.jl:80).  `loglik_bound` is what lets the engine normalise against an analytic bound.
_LINEAR_PART = r"""
    static constexpr bool RB = false;
    const ModelD* md;
    double bu[NXU];
    bool has_u;
    DEV void prepare_linear(const ModelD* m, const double* u) {
        md = m;
        const int nu = m->nu;
        has_u = nu > 0 && u != nullptr;
        for (int r = 0; r < NXU; ++r) {
            double acc = 0.0;
            if (has_u) {
                acc = m->B[r * nu + 0] * u[0];
                for (int c = 1; c < nu; ++c) acc = acc + m->B[r * nu + c] * u[c];
            }
            bu[r] = acc;
        }
    }
    DEV void dynamics(const double* x, double* out) const {
        for (int r = 0; r < NXU; ++r) {
            double ax = md->A[r * NXU + 0] * x[0];
            for (int c = 1; c < NXU; ++c) ax = ax + md->A[r * NXU + c] * x[c];
            out[r] = has_u ? ax + bu[r] : ax;
        }
    }
    DEV void measurement(const double* x, double* out) const {
        for (int r = 0; r < NYU; ++r) {
            double cx = md->C[r * NXU + 0] * x[0];
            for (int c = 1; c < NXU; ++c) cx = cx + md->C[r * NXU + c] * x[c];
            out[r] = cx;
        }
    }
"""

# Laplace measurement noise, independent components, scale b = qt[0]:  log p = -(sum_k |y_k - g_k|) / b - ny log(2 b)
LAPLACE_SRC = r"""
struct UserModel {
    static constexpr int NXU = 2, NYU = 1;
""" + _LINEAR_PART + r"""
    double b, c;
    DEV void prepare(const ModelD* m, const double* u, double t) {
        prepare_linear(m, u);
        b = m->qt[0];
        c = (double)NYU * llpf_log(2.0 * b);
    }
    DEV double loglik(const double* x, const double* y, double t) const {
        double g[NYU];
        measurement(x, g);
        double s = llpf_fabs(y[0] - g[0]);
        for (int k = 1; k < NYU; ++k) s = s + llpf_fabs(y[k] - g[k]);
        return (-(s / b)) - c;
    }
    DEV double loglik_bound() const { return -c; }
};
"""

# Student-t measurement noise, independent components: nu = qt[0], sigma = qt[1], c1 = qt[2] (the per-component log-normaliser,
# lgamma((nu+1)/2) - lgamma(nu/2) - log(nu pi)/2 - log sigma, formed by the host):  log p = sum_k (c1 - (nu+1)/2 log1p((v_k/sigma)^2/nu))
STUDENT_T_SRC = r"""
struct UserModel {
    static constexpr int NXU = 2, NYU = 1;
""" + _LINEAR_PART + r"""
    double nu_, sigma, c1, h;
    DEV void prepare(const ModelD* m, const double* u, double t) {
        prepare_linear(m, u);
        nu_ = m->qt[0]; sigma = m->qt[1]; c1 = m->qt[2];
        h = (nu_ + 1.0) / 2.0;
    }
    DEV double loglik(const double* x, const double* y, double t) const {
        double g[NYU];
        measurement(x, g);
        double ll = 0.0;
        for (int k = 0; k < NYU; ++k) {
            const double z = (y[k] - g[k]) / sigma;
            const double q = (z * z) / nu_;
            ll = ll + (c1 - h * llpf_log1p_nonneg(q));
        }
        return ll;
    }
    DEV double loglik_bound() const {
        double bd = 0.0;
        for (int k = 0; k < NYU; ++k) bd = bd + c1;
        return bd;
    }
};
"""

# the same Laplace likelihood WITHOUT a declared bound: every step is normalised in the exact-max form (one host round trip each)
LAPLACE_NO_BOUND_SRC = LAPLACE_SRC.replace("    DEV double loglik_bound() const { return -c; }\n", "")

# ---- process noise / initial density of the model's own (include/llpf.h: optional members `noise`, `initial`) ------------------
# The reference's AdvancedParticleFilter hands the noise to the user (dynamics(x, u, p, t, noise = true), src/PFtypes.jl:242-259,
# test/runtests.jl:553-599); its ParticleFilter draws from any dynamics_density / initial_density (:135, src/filtering.jl:8).
# Multiplicative Gaussian noise — standard deviation s0 + s1 |x_d| of the state the particle leaves, (s0, s1) = qt[0..1] — and a
# uniform box as initial density, lo = qt[2..3], hi = qt[4..5]
MULT_NOISE_BOX_SRC = r"""
struct UserModel {
    static constexpr int NXU = 2, NYU = 1;
""" + _LINEAR_PART + r"""
    double s0, s1, lo[NXU], hi[NXU];
    DEV void prepare(const ModelD* m, const double* u, double t) {
        prepare_linear(m, u);
        s0 = m->qt[0]; s1 = m->qt[1];
        for (int d = 0; d < NXU; ++d) { lo[d] = m->qt[2 + d]; hi[d] = m->qt[2 + NXU + d]; }
    }
    DEV void noise(const double* x, const double* fx, const double* xi, const double* uu, double* out) const {
        for (int d = 0; d < NXU; ++d) {
            const double sd = s0 + s1 * llpf_fabs(x[d]);
            out[d] = fx[d] + sd * xi[d];
        }
    }
    DEV void initial(const double* xi, const double* uu, double* out) const {
        for (int d = 0; d < NXU; ++d) out[d] = lo[d] + (hi[d] - lo[d]) * uu[d];
    }
};
"""

# Laplace process noise of scale b = qt[0], one uniform per component through the inverse CDF (heavy tails: a non-Gaussian
# dynamics_density); the initial density stays the Gaussian descriptor
LAPLACE_NOISE_SRC = r"""
struct UserModel {
    static constexpr int NXU = 2, NYU = 1;
""" + _LINEAR_PART + r"""
    double b;
    DEV void prepare(const ModelD* m, const double* u, double t) {
        prepare_linear(m, u);
        b = m->qt[0];
    }
    DEV void noise(const double* x, const double* fx, const double* xi, const double* uu, double* out) const {
        for (int d = 0; d < NXU; ++d) {
            const double v = 2.0 * uu[d] - 1.0;
            double tt = 1.0 - llpf_fabs(v);
            if (!(tt > 0.0)) tt = 1.1102230246251565e-16;
            const double mg = b * (-llpf_log(tt));
            out[d] = fx[d] + (v < 0.0 ? -mg : mg);
        }
    }
};
"""
