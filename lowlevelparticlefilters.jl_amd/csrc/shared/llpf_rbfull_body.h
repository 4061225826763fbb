/* llpf_rbfull_body.h — macro template, included by llpf_rbfull.h (and once more by the oracle) with
 *   RBF_(name)   function-name prefix
 *   RBF_SQRT(x)  square root      RBF_LOG(x)  natural logarithm
 * All accumulations run over the summation index in increasing order with explicit fused multiply-adds. */

/* An(xn) = An[0] + sum_k xn[k] An[1+k]                       (pf.An as a function of the state, src/rbpf.jl:208) */
LLPF_HD void RBF_(coupling)(const llpf_rbf_par* p, const int nn, const int nl, const double* xn, double* An) {
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {
            double a = p->An[0][r * nl + c];
            LLPF_UNROLL
            for (int k = 0; k < nn; ++k) a = llpf_fma(xn[k], p->An[1 + k][r * nl + c], a);
            An[r * nl + c] = a;
        }
    }
}

/* Loop nests below run the inner (summation) index OUTERMOST over a set of independent accumulators: every accumulator
 * still sums its products in increasing index order (so results do not depend on the nesting), but consecutive
 * instructions are independent — a wave that runs alone on its SIMD (440 registers) otherwise waits out the latency of
 * each dependent fma. */

/* Time update of one particle — src/rbpf.jl:206-221 (An != 0 branch, !singleR):
 *   Nt = An R An' + R1n ; L = (Al R An') / Nt ; R1 = Al R Al' + R1l - L Nt L'
 *   Axl = An xl ; z = Axl + nz ; xn1 = fi + z ; xl1 = Al xl + Bl u + L (z - Axl)
 * fi = f_n(xn, u, p, t) and nz ~ R1n come from the caller.  With Nt = Lc Lc' (Cholesky, reciprocal diagonal) the gain
 * is never formed: W = (Al R An') Lc^-T gives L Nt L' = W W' and L d = W (Lc^-1 d) — the same quantities as the
 * reference's generic `/`, 200 multiply-adds fewer per particle.  (Al R An') is taken as Al (An R)'.
 * R, R1: packed lower triangles (may not alias). */
LLPF_HD void RBF_(predict)(const llpf_rbf_par* p, const int nn, const int nl, const int nu, const double* xn,
                           const double* xl, const double* R, const double* u, const double* fi, const double* nz,
                           double* xn1, double* xl1, double* R1) {
    double AnR[LLPF_RBF_MAXN * LLPF_RBF_MAXL];
    double Lc[LLPF_RBF_MAXN * LLPF_RBF_MAXN], invd[LLPF_RBF_MAXN], v[LLPF_RBF_MAXN];
    double W[LLPF_RBF_MAXL * LLPF_RBF_MAXN];
    {
        double An[LLPF_RBF_MAXN * LLPF_RBF_MAXL], Nt[LLPF_RBF_MAXN * LLPF_RBF_MAXN], dz[LLPF_RBF_MAXN], ax[LLPF_RBF_MAXN];
        RBF_(coupling)(p, nn, nl, xn, An);
        LLPF_UNROLL
        for (int r = 0; r < nn; ++r) ax[r] = An[r * nl] * xl[0];   /* z = An xl + nz ; xn1 = fi + z ; dz = z - An xl */
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) {
            LLPF_UNROLL
            for (int r = 0; r < nn; ++r) ax[r] = llpf_fma(An[r * nl + c], xl[c], ax[r]);
        }
        LLPF_UNROLL
        for (int r = 0; r < nn; ++r) {
            const double z = ax[r] + nz[r];
            xn1[r] = fi[r] + z;
            dz[r] = z - ax[r];
        }
        LLPF_UNROLL
        for (int r = 0; r < nn; ++r) {                          /* AnR = An R */
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) AnR[r * nl + c] = An[r * nl] * R[llpf_rbf_idx(0, c)];
        }
        LLPF_UNROLL
        for (int q = 1; q < nl; ++q) {
            LLPF_UNROLL
            for (int r = 0; r < nn; ++r) {
                LLPF_UNROLL
                for (int c = 0; c < nl; ++c) AnR[r * nl + c] = llpf_fma(An[r * nl + q], R[llpf_rbf_idx(q, c)], AnR[r * nl + c]);
            }
        }
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) {                          /* Nt = AnR An' + R1n (lower triangle) */
            LLPF_UNROLL
            for (int j = 0; j <= i; ++j) Nt[i * nn + j] = AnR[i * nl] * An[j * nl];
        }
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) {
            LLPF_UNROLL
            for (int i = 0; i < nn; ++i) {
                LLPF_UNROLL
                for (int j = 0; j <= i; ++j) Nt[i * nn + j] = llpf_fma(AnR[i * nl + c], An[j * nl + c], Nt[i * nn + j]);
            }
        }
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) {
            LLPF_UNROLL
            for (int j = 0; j <= i; ++j) Nt[i * nn + j] = Nt[i * nn + j] + p->R1n[i * nn + j];
        }
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) {                          /* Nt = Lc Lc' */
            LLPF_UNROLL
            for (int j = 0; j <= i; ++j) {
                double acc = Nt[i * nn + j];
                LLPF_UNROLL
                for (int k = 0; k < j; ++k) acc = llpf_fma(-Lc[i * nn + k], Lc[j * nn + k], acc);
                if (i == j) {
                    const double d = RBF_SQRT(acc);             /* not positive definite: NaN, caught as a degenerate weight */
                    Lc[i * nn + i] = d;
                    invd[i] = 1.0 / d;
                } else {
                    Lc[i * nn + j] = acc * invd[j];
                }
            }
        }
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) {                          /* Lc v = dz */
            double acc = dz[i];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * nn + q], v[q], acc);
            v[i] = acc * invd[i];
        }
    }
    {                                                           /* Al xl + Bl u for all rows */
        LLPF_UNROLL
        for (int r = 0; r < nl; ++r) xl1[r] = p->Al[r * nl] * xl[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) {
            LLPF_UNROLL
            for (int r = 0; r < nl; ++r) xl1[r] = llpf_fma(p->Al[r * nl + c], xl[c], xl1[r]);
        }
        if (nu > 0) {
            LLPF_UNROLL
            for (int r = 0; r < nl; ++r) {
                double b2 = p->Bl[r * nu] * u[0];
                for (int c = 1; c < nu; ++c) b2 = llpf_fma(p->Bl[r * nu + c], u[c], b2);
                xl1[r] = xl1[r] + b2;
            }
        }
    }
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {
        double ARr[LLPF_RBF_MAXL], g[LLPF_RBF_MAXN], acc[LLPF_RBF_MAXL];
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) g[i] = p->Al[r * nl] * AnR[i * nl];    /* row r of Al (An R)' */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) ARr[c] = p->Al[r * nl] * R[llpf_rbf_idx(0, c)];   /* row r of Al R */
        LLPF_UNROLL
        for (int q = 1; q < nl; ++q) {
            LLPF_UNROLL
            for (int i = 0; i < nn; ++i) g[i] = llpf_fma(p->Al[r * nl + q], AnR[i * nl + q], g[i]);
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) ARr[c] = llpf_fma(p->Al[r * nl + q], R[llpf_rbf_idx(q, c)], ARr[c]);
        }
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) {                          /* Lc w = g : row r of W */
            double a = g[i];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) a = llpf_fma(-Lc[i * nn + q], W[r * nn + q], a);
            W[r * nn + i] = a * invd[i];
        }
        LLPF_UNROLL
        for (int c = 0; c <= r; ++c) acc[c] = ARr[0] * p->Al[c * nl];       /* (Al R Al')[r, 0..r] */
        LLPF_UNROLL
        for (int q = 1; q < nl; ++q) {
            LLPF_UNROLL
            for (int c = 0; c <= r; ++c) acc[c] = llpf_fma(ARr[q], p->Al[c * nl + q], acc[c]);
        }
        LLPF_UNROLL
        for (int c = 0; c <= r; ++c) {                          /* R1[r,c] = (Al R Al')[r,c] + R1l[r,c] - (W W')[r,c] */
            const double a = acc[c] + p->R1l[llpf_rbf_idx(r, c)];
            double s = W[r * nn] * W[c * nn];
            LLPF_UNROLL
            for (int j = 1; j < nn; ++j) s = llpf_fma(W[r * nn + j], W[c * nn + j], s);
            R1[llpf_rbf_idx(r, c)] = a - s;
        }
        {                                                       /* xl1 = (Al xl + Bl u) + W (Lc^-1 dz) */
            double s = W[r * nn] * v[0];
            LLPF_UNROLL
            for (int j = 1; j < nn; ++j) s = llpf_fma(W[r * nn + j], v[j], s);
            xl1[r] = xl1[r] + s;
        }
    }
}

/* Measurement update of one particle — src/rbpf.jl:259-263 -> correct!(kf, u, y - yn, p, t), src/filtering.jl:100-128:
 *   e = (y - yn) - C xl ; S = symmetrize(C R C') + R2 ; K = (R C') / chol(S) ; xl += K e ;
 *   R = symmetrize((I - K C) R)   [formed as R - K (C R)] ; returns ll = logpdf(N(0, S), e).
 * xl and R (packed lower triangle) are updated in place. */
LLPF_HD double RBF_(correct)(const llpf_rbf_par* p, const int nl, const int ny, const double* y, const double* yn,
                             double* xl, double* R) {
    double e[LLPF_RBF_MAXY], CR[LLPF_RBF_MAXY * LLPF_RBF_MAXL], raw[LLPF_RBF_MAXY * LLPF_RBF_MAXY];
    double Lc[LLPF_RBF_MAXY * LLPF_RBF_MAXY], invd[LLPF_RBF_MAXY], K[LLPF_RBF_MAXL * LLPF_RBF_MAXY];
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        double a = p->Cl[i * nl] * xl[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) a = llpf_fma(p->Cl[i * nl + c], xl[c], a);
        e[i] = (y[i] - yn[i]) - a;
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {                              /* CR = C R  (its transpose is R C') */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) CR[i * nl + c] = p->Cl[i * nl] * R[llpf_rbf_idx(0, c)];
    }
    LLPF_UNROLL
    for (int q = 1; q < nl; ++q) {
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) CR[i * nl + c] = llpf_fma(p->Cl[i * nl + q], R[llpf_rbf_idx(q, c)], CR[i * nl + c]);
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        LLPF_UNROLL
        for (int j = 0; j < ny; ++j) raw[i * ny + j] = CR[i * nl] * p->Cl[j * nl];
    }
    LLPF_UNROLL
    for (int c = 1; c < nl; ++c) {
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            LLPF_UNROLL
            for (int j = 0; j < ny; ++j) raw[i * ny + j] = llpf_fma(CR[i * nl + c], p->Cl[j * nl + c], raw[i * ny + j]);
        }
    }
    double ldet = 0.0;
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {                              /* S = 0.5 (raw + raw') + R2 = Lc Lc' */
        LLPF_UNROLL
        for (int j = 0; j <= i; ++j) {
            double acc = 0.5 * (raw[i * ny + j] + raw[j * ny + i]) + p->R2[i * ny + j];
            LLPF_UNROLL
            for (int k = 0; k < j; ++k) acc = llpf_fma(-Lc[i * ny + k], Lc[j * ny + k], acc);
            if (i == j) {
                const double d = RBF_SQRT(acc);
                Lc[i * ny + i] = d;
                invd[i] = 1.0 / d;
                ldet = ldet + RBF_LOG(d);
            } else {
                Lc[i * ny + j] = acc * invd[j];
            }
        }
    }
    double quad = 0.0;
    {
        double z[LLPF_RBF_MAXY];
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {                          /* Lc z = e */
            double acc = e[i];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * ny + q], z[q], acc);
            z[i] = acc * invd[i];
            quad = llpf_fma(z[i], z[i], quad);
        }
    }
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {                              /* row r of K solves k S = (R C')[r,:] */
        double t[LLPF_RBF_MAXY];
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            double acc = CR[i * nl + r];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * ny + q], t[q], acc);
            t[i] = acc * invd[i];
        }
        LLPF_UNROLL
        for (int i = ny - 1; i >= 0; --i) {
            double acc = t[i];
            LLPF_UNROLL
            for (int q = i + 1; q < ny; ++q) acc = llpf_fma(-Lc[q * ny + i], K[r * ny + q], acc);
            K[r * ny + i] = acc * invd[i];
        }
    }
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {
        double a = K[r * ny] * e[0];
        LLPF_UNROLL
        for (int i = 1; i < ny; ++i) a = llpf_fma(K[r * ny + i], e[i], a);
        xl[r] = xl[r] + a;
    }
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {
        LLPF_UNROLL
        for (int c = 0; c <= r; ++c) {
            double a = R[llpf_rbf_idx(r, c)], b2 = a;
            LLPF_UNROLL
            for (int i = 0; i < ny; ++i) {
                a = llpf_fma(-K[r * ny + i], CR[i * nl + c], a);
                b2 = llpf_fma(-K[c * ny + i], CR[i * nl + r], b2);
            }
            R[llpf_rbf_idx(r, c)] = 0.5 * (a + b2);
        }
    }
    return (p->c0y - ldet) - 0.5 * quad;
}
