/* llpf_rbfull_body.h — macro template, included by llpf_rbfull.h with
 *   RBF_(name)   function-name prefix
 *   RBF_SQRT(x)  square root      RBF_LOG(x)  natural logarithm
 *   RBF_STAGE(ptr, dep)  device only: the parameter pointer becomes opaque to the compiler at this point and the point is
 *                        tied to the value `dep` — see "Stages" below; nothing on the host.
 * All accumulations run over the summation index in increasing order with explicit fused multiply-adds.
 *
 * The recursion in the form computed here (round 3).  The reference writes the time update of the linear substate as
 *     Nt = An R An' + R1n ;  L = (Al R An') / Nt ;  R1 = Al R Al' + R1l - L Nt L' ;  xl1 = Al xl + Bl u + L (z - An xl)
 * (src/rbpf.jl:206-221).  With Nt = Lc Lc' and V = (An R)' Lc^-T this is, term by term,
 *     L Nt L' = Al (V V') Al'          L d = Al V (Lc^-1 d)
 * so that
 *     R~  = R - V V'                   x~l = xl + V (Lc^-1 (z - An xl))         "condition on the pseudo-measurement z"
 *     R1  = Al R~ Al' + R1l            xl1 = Al x~l + Bl u                      "ordinary Kalman time update"
 * — the same quantities (Schoen, Gustafsson, Nordlund 2005, eq. 22-24 group them this way), 250 multiply-adds fewer per particle
 * than forming Al R An' and W W', and — the reason for the change — every step works IN PLACE on one covariance: the live set
 * of a particle is R + one 4 x 8 panel (~90 doubles) instead of R, R1, An R, W and An at once (~200), which is what let the
 * kernel keep one wave per SIMD only (kernels/rbfull.hpp).  The reference-order oracle keeps the reference's literal formulas
 * (oracle/llpf_oracle.c: rbfr_*), so the two forms check each other to rounding.
 *
 * Stages.  The constant matrices are uniform: on the device they are scalar loads into SGPRs, of which a wave has ~100 — 50
 * doubles, against ~330 here.  Left alone the compiler hoists all of those loads to the top of the (fully unrolled) body and
 * spills the SGPRs into VGPR lanes: a third of the old kernel's instructions were v_readlane / v_writelane.  RBF_STAGE pins the
 * loads of a stage behind the value that ends the stage before it, and the stages are cut so that none needs more than ~40
 * doubles of constants: a row of one coupling matrix (8) or a pair of rows of Al (16) at a time. */

/* row r of An(xn) = An[0] + sum_k xn[k] An[1+k]              (pf.An as a function of the state, src/rbpf.jl:208) */
LLPF_HD void RBF_(coupling_row)(llpf_rbf_cptr p, const int nn, const int nl, const int r, const double* xn, double* a) {
    LLPF_UNROLL
    for (int c = 0; c < nl; ++c) a[c] = p->An[0][r * nl + c];
    LLPF_UNROLL
    for (int k = 0; k < nn; ++k) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) a[c] = llpf_fma(xn[k], p->An[1 + k][r * nl + c], a[c]);
    }
}
LLPF_HD void RBF_(coupling)(const llpf_rbf_par* p, const int nn, const int nl, const double* xn, double* An) {
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) RBF_(coupling_row)(RBF_CPTR(p), nn, nl, r, xn, An + r * nl);
}

/* Loop nests below run the inner (summation) index OUTERMOST over a set of independent accumulators: every accumulator
 * still sums its products in increasing index order (so results do not depend on the nesting), but consecutive
 * instructions are independent. */

/* rows [r0, r1) of M = Al X for a symmetric X held as a packed lower triangle; M row r is stored at row r - m0 */
#define RBF_PANEL(M, m0, r0, r1)                                                                                     \
    LLPF_UNROLL                                                                                                      \
    for (int r = (r0); r < (r1); ++r) {                                                                              \
        LLPF_UNROLL                                                                                                  \
        for (int c = 0; c < nl; ++c) M[(r - (m0)) * nl + c] = pp->Al[r * nl] * Rt[llpf_rbf_idx(0, c)];              \
    }                                                                                                                \
    LLPF_UNROLL                                                                                                      \
    for (int q = 1; q < nl; ++q) {                                                                                   \
        LLPF_UNROLL                                                                                                  \
        for (int r = (r0); r < (r1); ++r) {                                                                          \
            LLPF_UNROLL                                                                                              \
            for (int c = 0; c < nl; ++c)                                                                             \
                M[(r - (m0)) * nl + c] = llpf_fma(pp->Al[r * nl + q], Rt[llpf_rbf_idx(q, c)], M[(r - (m0)) * nl + c]); \
        }                                                                                                            \
    }
/* (Bl u)[r] — particle-independent; the device computes it once per wave (kernels/rbfull.hpp) and the body reads it back, so
 * that the body stays one straight-line block (nu is a run-time number) */
LLPF_HD double RBF_(blu_row)(llpf_rbf_cptr p, const int nu, const int r, const double* u) {
    if (nu <= 0) return -0.0;                                    /* x + (-0.0) == x for every x, signed zeros included */
    double b2 = p->Bl[r * nu] * u[0];
    for (int c = 1; c < nu; ++c) b2 = llpf_fma(p->Bl[r * nu + c], u[c], b2);
    return b2;
}
/* rows [r0, r1) of Al x~l + Bl u */
#define RBF_MEAN(r0, r1)                                                                                             \
    LLPF_UNROLL                                                                                                      \
    for (int r = (r0); r < (r1); ++r) xl1[r] = pp->Al[r * nl] * xt[0];                                               \
    LLPF_UNROLL                                                                                                      \
    for (int c = 1; c < nl; ++c) {                                                                                   \
        LLPF_UNROLL                                                                                                  \
        for (int r = (r0); r < (r1); ++r) xl1[r] = llpf_fma(pp->Al[r * nl + c], xt[c], xl1[r]);                      \
    }                                                                                                                \
    LLPF_UNROLL                                                                                                      \
    for (int r = (r0); r < (r1); ++r) xl1[r] = xl1[r] + RBF_BLU(pp, nu, r, u, blu);
/* columns [c0, c1) of a block of R1 = M Al': R1[r, c] = M[r - m0, :] . Al[c, :] for rows max(c, rlo) <= r < rhi */
#define RBF_OUT_COLS(M, m0, c0, c1, rlo, rhi)                                                                        \
    LLPF_UNROLL                                                                                                      \
    for (int c = (c0); c < (c1); ++c) {                                                                              \
        LLPF_UNROLL                                                                                                  \
        for (int r = (c > (rlo) ? c : (rlo)); r < (rhi); ++r) R1[llpf_rbf_idx(r, c)] = M[(r - (m0)) * nl] * pp->Al[c * nl]; \
    }                                                                                                                \
    LLPF_UNROLL                                                                                                      \
    for (int q = 1; q < nl; ++q) {                                                                                   \
        LLPF_UNROLL                                                                                                  \
        for (int c = (c0); c < (c1); ++c) {                                                                          \
            LLPF_UNROLL                                                                                              \
            for (int r = (c > (rlo) ? c : (rlo)); r < (rhi); ++r)                                                    \
                R1[llpf_rbf_idx(r, c)] = llpf_fma(M[(r - (m0)) * nl + q], pp->Al[c * nl + q], R1[llpf_rbf_idx(r, c)]); \
        }                                                                                                            \
    }
/* rows [r0, r1) of the lower-left block of R1 = Al M' (R~ is symmetric): R1[r, c] = Al[r, :] . M[c, :] for c < ht */
#define RBF_OUT_ROWS(M, r0, r1)                                                                                      \
    LLPF_UNROLL                                                                                                      \
    for (int r = (r0); r < (r1); ++r) {                                                                              \
        LLPF_UNROLL                                                                                                  \
        for (int c = 0; c < ht; ++c) R1[llpf_rbf_idx(r, c)] = pp->Al[r * nl] * M[c * nl];                            \
    }                                                                                                                \
    LLPF_UNROLL                                                                                                      \
    for (int q = 1; q < nl; ++q) {                                                                                   \
        LLPF_UNROLL                                                                                                  \
        for (int r = (r0); r < (r1); ++r) {                                                                          \
            LLPF_UNROLL                                                                                              \
            for (int c = 0; c < ht; ++c) R1[llpf_rbf_idx(r, c)] = llpf_fma(pp->Al[r * nl + q], M[c * nl + q], R1[llpf_rbf_idx(r, c)]); \
        }                                                                                                            \
    }
#define RBF_MIN(a, b) ((a) < (b) ? (a) : (b))
#ifndef RBF_RPS
#define RBF_RPS 2      /* rows of Al per stage */
#endif

/* Time update of one particle — src/rbpf.jl:206-221 (An != 0 branch, !singleR), in the form of the header comment.
 * fi = f_n(xn, u, p, t) and nz ~ R1n come from the caller.  R, R1: packed lower triangles (may alias: R is copied first).
 * blu: device only, Bl u as precomputed with blu_row (the host evaluates blu_row in place and ignores it). */
LLPF_HD void RBF_(predict)(const llpf_rbf_par* p, const int nn, const int nl, const int nu, const double* xn,
                           const double* xl, const double* R, const double* u, const double* blu, const double* fi,
                           const double* nz, double* xn1, double* xl1, double* R1) {
    llpf_rbf_cptr pp = RBF_CPTR(p);
    (void)blu;
    double Rt[LLPF_RBF_NP(LLPF_RBF_MAXL)];                       /* R, then R~ */
    double AnR[LLPF_RBF_MAXN * LLPF_RBF_MAXL];                   /* An R; column c becomes row c of V */
    double Nt[LLPF_RBF_MAXN * LLPF_RBF_MAXN], Lc[LLPF_RBF_MAXN * LLPF_RBF_MAXN], invd[LLPF_RBF_MAXN];
    double ax[LLPF_RBF_MAXN], v[LLPF_RBF_MAXN], xt[LLPF_RBF_MAXL];
    const int np = LLPF_RBF_NP(nl);
    LLPF_UNROLL
    for (int d = 0; d < np; ++d) Rt[d] = R[d];
    RBF_STAMP(1);
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {                              /* one row of An(xn) at a time */
        double a[LLPF_RBF_MAXL];
        RBF_STAGE(pp, r == 0 ? Rt[0] : Nt[(r - 1) * nn + (r - 1)]);
        RBF_(coupling_row)(pp, nn, nl, r, xn, a);
        ax[r] = a[0] * xl[0];                                    /* (An xl)[r] */
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) ax[r] = llpf_fma(a[c], xl[c], ax[r]);
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) AnR[r * nl + c] = a[0] * Rt[llpf_rbf_idx(0, c)];      /* (An R)[r, :] */
        LLPF_UNROLL
        for (int q = 1; q < nl; ++q) {
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) AnR[r * nl + c] = llpf_fma(a[q], Rt[llpf_rbf_idx(q, c)], AnR[r * nl + c]);
        }
        LLPF_UNROLL
        for (int j = 0; j <= r; ++j) Nt[r * nn + j] = a[0] * AnR[j * nl];                  /* Nt[r, j] = An[r, :] . (An R)[j, :] */
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) {
            LLPF_UNROLL
            for (int j = 0; j <= r; ++j) Nt[r * nn + j] = llpf_fma(a[c], AnR[j * nl + c], Nt[r * nn + j]);
        }
        LLPF_UNROLL
        for (int j = 0; j <= r; ++j) Nt[r * nn + j] = Nt[r * nn + j] + pp->R1n[r * nn + j];
        RBF_DONE(AnR, r * nl, r * nl + nl);
        RBF_DONE(Nt, r * nn, r * nn + r + 1);
        RBF_DONE(ax, r, r + 1);
    }
    RBF_STAMP(2);
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
    for (int i = 0; i < nn; ++i) {                              /* z = An xl + nz ; xn1 = fi + z ; Lc v = z - An xl */
        const double z = ax[i] + nz[i];
        xn1[i] = fi[i] + z;
        double acc = z - ax[i];
        LLPF_UNROLL
        for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * nn + q], v[q], acc);
        v[i] = acc * invd[i];
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {                              /* Lc V' = An R, in place: V[c, i] at AnR[i, c] */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {
            double acc = AnR[i * nl + c];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * nn + q], AnR[q * nl + c], acc);
            AnR[i * nl + c] = acc * invd[i];
        }
    }
    LLPF_UNROLL
    for (int c = 0; c < nl; ++c) {                              /* x~l = xl + V v */
        double s = AnR[c] * v[0];
        LLPF_UNROLL
        for (int j = 1; j < nn; ++j) s = llpf_fma(AnR[j * nl + c], v[j], s);
        xt[c] = xl[c] + s;
    }
    LLPF_UNROLL
    for (int j = 0; j < nn; ++j) {                              /* R~ = R - V V' */
        LLPF_UNROLL
        for (int r = 0; r < nl; ++r) {
            LLPF_UNROLL
            for (int c = 0; c <= r; ++c) Rt[llpf_rbf_idx(r, c)] = llpf_fma(-AnR[j * nl + r], AnR[j * nl + c], Rt[llpf_rbf_idx(r, c)]);
        }
    }
    RBF_STAMP(3);
    {   /* xl1 = Al x~l + Bl u ; R1 = Al R~ Al' + R1l.  The covariance goes by halves of Al (upper-left block from the upper
         * panel M = Al[0:ht] R~, lower-left block from the same panel against the lower rows of Al, then the lower panel and
         * the lower-right block), and every piece by pairs of rows of Al: 16 constants per stage. */
        const int ht = (nl + 1) / 2;
        double M[((LLPF_RBF_MAXL + 1) / 2) * LLPF_RBF_MAXL];
        LLPF_UNROLL
        for (int r0 = 0; r0 < nl; r0 += RBF_RPS) {
            RBF_STAGE(pp, r0 == 0 ? Rt[np - 1] : xl1[r0 - 1]);
            RBF_MEAN(r0, RBF_MIN(r0 + RBF_RPS, nl))
            RBF_DONE(xl1, r0, RBF_MIN(r0 + RBF_RPS, nl));
        }
        RBF_STAMP(4);
        LLPF_UNROLL
        for (int r0 = 0; r0 < ht; r0 += RBF_RPS) {                    /* M = Al[0:ht, :] R~ */
            RBF_STAGE(pp, r0 == 0 ? xl1[nl - 1] : M[r0 * nl - 1]);
            RBF_PANEL(M, 0, r0, RBF_MIN(r0 + RBF_RPS, ht))
            RBF_DONE(M, r0 * nl, RBF_MIN(r0 + RBF_RPS, ht) * nl);
        }
        LLPF_UNROLL
        for (int c0 = 0; c0 < ht; c0 += RBF_RPS) {                    /* upper-left block */
            RBF_STAGE(pp, c0 == 0 ? M[ht * nl - 1] : R1[llpf_rbf_idx(ht - 1, c0 - 1)]);
            RBF_OUT_COLS(M, 0, c0, RBF_MIN(c0 + RBF_RPS, ht), 0, ht)
            RBF_DONE(R1, 0, LLPF_RBF_NP(ht));
        }
        RBF_STAMP(5);
        if (ht < nl) {
            LLPF_UNROLL
            for (int r0 = ht; r0 < nl; r0 += RBF_RPS) {               /* lower-left block */
                RBF_STAGE(pp, r0 == ht ? R1[llpf_rbf_idx(ht - 1, ht - 1)] : R1[llpf_rbf_idx(r0 - 1, ht - 1)]);
                RBF_OUT_ROWS(M, r0, RBF_MIN(r0 + RBF_RPS, nl))
                LLPF_UNROLL
                for (int r = r0; r < RBF_MIN(r0 + RBF_RPS, nl); ++r) RBF_DONE(R1, llpf_rbf_idx(r, 0), llpf_rbf_idx(r, 0) + ht);
            }
            RBF_FENCE(R1[llpf_rbf_idx(nl - 1, ht - 1)]);
            RBF_STAMP(6);
            LLPF_UNROLL
            for (int r0 = ht; r0 < nl; r0 += RBF_RPS) {               /* M = Al[ht:nl, :] R~ */
                RBF_STAGE(pp, r0 == ht ? R1[llpf_rbf_idx(nl - 1, ht - 1)] : M[(r0 - ht) * nl - 1]);
                RBF_PANEL(M, ht, r0, RBF_MIN(r0 + RBF_RPS, nl))
                RBF_DONE(M, (r0 - ht) * nl, (RBF_MIN(r0 + RBF_RPS, nl) - ht) * nl);
            }
            RBF_FENCE(M[(nl - ht) * nl - 1]);
            RBF_STAMP(7);
            LLPF_UNROLL
            for (int c0 = ht; c0 < nl; c0 += RBF_RPS) {               /* lower-right block */
                RBF_STAGE(pp, c0 == ht ? M[(nl - ht) * nl - 1] : R1[llpf_rbf_idx(nl - 1, c0 - 1)]);
                RBF_OUT_COLS(M, ht, c0, RBF_MIN(c0 + RBF_RPS, nl), ht, nl)
                RBF_DONE(R1, LLPF_RBF_NP(ht), np);
            }
        }
        LLPF_UNROLL
        for (int d0 = 0; d0 < np; d0 += 12) {
            RBF_STAGE(pp, d0 == 0 ? R1[np - 1] : R1[d0 - 1]);
            LLPF_UNROLL
            for (int d = d0; d < RBF_MIN(d0 + 12, np); ++d) R1[d] = R1[d] + pp->R1l[d];
            RBF_DONE(R1, d0, RBF_MIN(d0 + 12, np));
        }
        RBF_STAMP(8);
    }
}
#undef RBF_PANEL
#undef RBF_MEAN
#undef RBF_OUT_COLS
#undef RBF_OUT_ROWS
#undef RBF_MIN

/* Measurement update of one particle — src/rbpf.jl:259-263 -> correct!(kf, u, y - yn, p, t), src/filtering.jl:100-128:
 *   e = (y - yn) - C xl ; S = symmetrize(C R C') + R2 ; K = (R C') / chol(S) ; xl += K e ;
 *   R = symmetrize((I - K C) R)   [formed as R - K (C R)] ; returns ll = logpdf(N(0, S), e).
 * xl and R (packed lower triangle) are updated in place. */
LLPF_HD double RBF_(correct)(const llpf_rbf_par* p, const int nl, const int ny, const double* y, const double* yn,
                             double* xl, double* R) {
    llpf_rbf_cptr pp = RBF_CPTR(p);
    double e[LLPF_RBF_MAXY], CR[LLPF_RBF_MAXY * LLPF_RBF_MAXL], raw[LLPF_RBF_MAXY * LLPF_RBF_MAXY];
    double Lc[LLPF_RBF_MAXY * LLPF_RBF_MAXY], invd[LLPF_RBF_MAXY], K[LLPF_RBF_MAXL * LLPF_RBF_MAXY];
    RBF_STAGE(pp, R[0]);
    RBF_STAMP(9);
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        double a = pp->Cl[i * nl] * xl[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) a = llpf_fma(pp->Cl[i * nl + c], xl[c], a);
        e[i] = (y[i] - yn[i]) - a;
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {                              /* CR = C R  (its transpose is R C') */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) CR[i * nl + c] = pp->Cl[i * nl] * R[llpf_rbf_idx(0, c)];
    }
    LLPF_UNROLL
    for (int q = 1; q < nl; ++q) {
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) CR[i * nl + c] = llpf_fma(pp->Cl[i * nl + q], R[llpf_rbf_idx(q, c)], CR[i * nl + c]);
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        LLPF_UNROLL
        for (int j = 0; j < ny; ++j) raw[i * ny + j] = CR[i * nl] * pp->Cl[j * nl];
    }
    LLPF_UNROLL
    for (int c = 1; c < nl; ++c) {
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            LLPF_UNROLL
            for (int j = 0; j < ny; ++j) raw[i * ny + j] = llpf_fma(CR[i * nl + c], pp->Cl[j * nl + c], raw[i * ny + j]);
        }
    }
    double ldet = 0.0;
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {                              /* S = 0.5 (raw + raw') + R2 = Lc Lc' */
        LLPF_UNROLL
        for (int j = 0; j <= i; ++j) {
            double acc = 0.5 * (raw[i * ny + j] + raw[j * ny + i]) + pp->R2[i * ny + j];
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
    RBF_STAMP(10);
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
    return (pp->c0y - ldet) - 0.5 * quad;
}
