/* llpf_rbfull_body.h — macro template, included by llpf_rbfull.h (and once more by the oracle) with
 *   RBF_(name)   function-name prefix
 *   RBF_SQRT(x)  square root      RBF_LOG(x)  natural logarithm
 * All accumulations run over the inner index in increasing order with explicit fused multiply-adds. */

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

/* Time update of one particle — src/rbpf.jl:206-221 (An != 0 branch, !singleR):
 *   Nt = An R An' + R1n ; L = (Al R An') / Nt ; R1 = Al R Al' + R1l - L Nt L'
 *   Axl = An xl ; z = Axl + nz ; xn1 = fi + z ; xl1 = Al xl + Bl u + L (z - Axl)
 * fi = f_n(xn, u, p, t) and nz ~ R1n come from the caller.  The right division by Nt goes through its Cholesky
 * factor with reciprocal diagonal (the reference: a generic `/`).  R, R1: packed lower triangles (may not alias). */
LLPF_HD void RBF_(predict)(const llpf_rbf_par* p, const int nn, const int nl, const int nu, const double* xn,
                           const double* xl, const double* R, const double* u, const double* fi, const double* nz,
                           double* xn1, double* xl1, double* R1) {
    double An[LLPF_RBF_MAXN * LLPF_RBF_MAXL], AnR[LLPF_RBF_MAXN * LLPF_RBF_MAXL];
    double Nt[LLPF_RBF_MAXN * LLPF_RBF_MAXN], Lc[LLPF_RBF_MAXN * LLPF_RBF_MAXN], invd[LLPF_RBF_MAXN];
    double L[LLPF_RBF_MAXL * LLPF_RBF_MAXN];
    RBF_(coupling)(p, nn, nl, xn, An);
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {                              /* AnR = An R */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {
            double a = An[r * nl] * R[llpf_rbf_idx(0, c)];
            LLPF_UNROLL
            for (int q = 1; q < nl; ++q) a = llpf_fma(An[r * nl + q], R[llpf_rbf_idx(q, c)], a);
            AnR[r * nl + c] = a;
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {                              /* Nt = AnR An' + R1n: lower triangle, mirrored */
        LLPF_UNROLL
        for (int j = 0; j <= i; ++j) {
            double a = AnR[i * nl] * An[j * nl];
            LLPF_UNROLL
            for (int c = 1; c < nl; ++c) a = llpf_fma(AnR[i * nl + c], An[j * nl + c], a);
            a = a + p->R1n[i * nn + j];
            Nt[i * nn + j] = a;
            Nt[j * nn + i] = a;
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {                              /* Nt = Lc Lc' */
        LLPF_UNROLL
        for (int j = 0; j <= i; ++j) {
            double acc = Nt[i * nn + j];
            LLPF_UNROLL
            for (int k = 0; k < j; ++k) acc = llpf_fma(-Lc[i * nn + k], Lc[j * nn + k], acc);
            if (i == j) {
                const double d = RBF_SQRT(acc);                 /* not positive definite: NaN, caught as a degenerate weight */
                Lc[i * nn + i] = d;
                invd[i] = 1.0 / d;
            } else {
                Lc[i * nn + j] = acc * invd[j];
            }
        }
    }
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {
        double ARr[LLPF_RBF_MAXL], t[LLPF_RBF_MAXN], LN[LLPF_RBF_MAXN];
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {                          /* row r of Al R */
            double a = p->Al[r * nl] * R[llpf_rbf_idx(0, c)];
            LLPF_UNROLL
            for (int q = 1; q < nl; ++q) a = llpf_fma(p->Al[r * nl + q], R[llpf_rbf_idx(q, c)], a);
            ARr[c] = a;
        }
        LLPF_UNROLL
        for (int i = 0; i < nn; ++i) {                          /* g = row r of (Al R) An';  Lc t = g */
            double acc = ARr[0] * An[i * nl];
            LLPF_UNROLL
            for (int c = 1; c < nl; ++c) acc = llpf_fma(ARr[c], An[i * nl + c], acc);
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * nn + q], t[q], acc);
            t[i] = acc * invd[i];
        }
        LLPF_UNROLL
        for (int i = nn - 1; i >= 0; --i) {                     /* Lc' l = t : row r of L */
            double acc = t[i];
            LLPF_UNROLL
            for (int q = i + 1; q < nn; ++q) acc = llpf_fma(-Lc[q * nn + i], L[r * nn + q], acc);
            L[r * nn + i] = acc * invd[i];
        }
        LLPF_UNROLL
        for (int j = 0; j < nn; ++j) {                          /* row r of L Nt */
            double a = L[r * nn] * Nt[j];
            LLPF_UNROLL
            for (int i = 1; i < nn; ++i) a = llpf_fma(L[r * nn + i], Nt[i * nn + j], a);
            LN[j] = a;
        }
        LLPF_UNROLL
        for (int c = 0; c <= r; ++c) {                          /* R1[r,c] = (Al R Al')[r,c] + R1l[r,c] - (L Nt L')[r,c] */
            double a = ARr[0] * p->Al[c * nl];
            LLPF_UNROLL
            for (int q = 1; q < nl; ++q) a = llpf_fma(ARr[q], p->Al[c * nl + q], a);
            a = a + p->R1l[llpf_rbf_idx(r, c)];
            double s = LN[0] * L[c * nn];
            LLPF_UNROLL
            for (int j = 1; j < nn; ++j) s = llpf_fma(LN[j], L[c * nn + j], s);
            R1[llpf_rbf_idx(r, c)] = a - s;
        }
    }
    double dz[LLPF_RBF_MAXN];
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {
        double a = An[r * nl] * xl[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) a = llpf_fma(An[r * nl + c], xl[c], a);
        const double z = a + nz[r];
        xn1[r] = fi[r] + z;
        dz[r] = z - a;
    }
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {
        double a = p->Al[r * nl] * xl[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) a = llpf_fma(p->Al[r * nl + c], xl[c], a);
        if (nu > 0) {
            double b2 = p->Bl[r * nu] * u[0];
            for (int c = 1; c < nu; ++c) b2 = llpf_fma(p->Bl[r * nu + c], u[c], b2);
            a = a + b2;
        }
        double s = L[r * nn] * dz[0];
        LLPF_UNROLL
        for (int j = 1; j < nn; ++j) s = llpf_fma(L[r * nn + j], dz[j], s);
        xl1[r] = a + s;
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
        for (int c = 0; c < nl; ++c) {
            double a = p->Cl[i * nl] * R[llpf_rbf_idx(0, c)];
            LLPF_UNROLL
            for (int q = 1; q < nl; ++q) a = llpf_fma(p->Cl[i * nl + q], R[llpf_rbf_idx(q, c)], a);
            CR[i * nl + c] = a;
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        LLPF_UNROLL
        for (int j = 0; j < ny; ++j) {
            double a = CR[i * nl] * p->Cl[j * nl];
            LLPF_UNROLL
            for (int c = 1; c < nl; ++c) a = llpf_fma(CR[i * nl + c], p->Cl[j * nl + c], a);
            raw[i * ny + j] = a;
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
