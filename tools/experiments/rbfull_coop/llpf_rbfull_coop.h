/* llpf_rbfull_coop.h — the per-particle Kalman recursion of llpf_rbfull_body.h for a batch of particles that SEVERAL waves work on
 * together (round 5; kernels/rbfull.hpp "the tail").
 *
 * Why.  k_rbfull gives every particle to one lane, and a wave costs the same whatever its number of active lanes, so the work of a
 * launch comes in units of one batch (64 particles, ~4000 vector instructions).  BASELINE C5 (N = 2e5) is 3125 batches on 1024
 * SIMDs: 53 SIMDs run a fourth batch while 971 are done after three (EXPERIMENTS 4.18).  The last batches are therefore split
 * BY OUTPUT instead of by particle: a workgroup of four waves shares one batch; every lane still owns one particle, every wave
 * computes a slice of each stage's outputs, and the slices meet in LDS.  Because the split is by wave, whatever indexes a
 * constant matrix stays uniform — the constants remain scalar operands, the register indices literal.
 *
 * Roles.  Wave 0 is the NONLINEAR wave: process noise, f_n(xn), the new xn, the measurement prediction, and the weight (its
 * code is in the kernel: it calls the model).  Waves 1..3 are the KALMAN waves k = 0..2 of this file; wave k owns the columns
 * c = k (mod 3) of every matrix with a column index in the linear state: R, An R, V, R~, M = Al R~, R1, C R, and the rows of K
 * and entries of xl with that index.  Small quantities every wave needs (An(xn), Nt and its Cholesky factor, Al x~l, the
 * innovation and its 2 x 2 factor) are computed by every Kalman wave: the four waves sit on four SIMDs, so a redundant
 * instruction costs issue slots, not time, and saves an exchange.
 *
 * Bits.  Every output element is formed by the expression llpf_rbf_predict / llpf_rbf_correct use for it — the same products in
 * the same (increasing) summation order, the same constants — so the result is the sequential recursion's, bit for bit:
 * tests/test_oracle_rbfull.py runs this file on the host (four threads as the four waves, oracle/rbf_coop_emul.c) against
 * llpf_rbf_predict + llpf_rbf_correct, and the engine's N = 2e5 trajectory is held to the oracle as before.  Symmetric
 * matrices are kept as FULL columns here (entry (r, c) of an owned column c for every r): the mirrored entry is computed by
 * the formula of its lower-triangle twin, operands commuted, which is the same IEEE result.
 *
 * The includer provides (before including):
 *   RBC_CTX_DECL          trailing parameters of llpf_rbc_kalman that the three macros below use
 *   RBC_X(off)            lvalue: element `off` (< LLPF_RBC_XTOT) of this particle's exchange buffer
 *   RBC_S(off)            lvalue: element `off` (< 4) of this particle's small exchange buffer (measurement prediction, ll)
 *   RBC_SYNC()            barrier across the waves that share the batch
 * Barriers: 3 without a measurement update, 5 with one; the nonlinear wave makes the same calls (kernels/rbfull.hpp). */
#ifndef LLPF_RBFULL_COOP_H
#define LLPF_RBFULL_COOP_H

#include "llpf_rbfull.h"

#define LLPF_RBC_NK 3                /* Kalman waves */
/* exchange buffer, doubles per particle */
#define LLPF_RBC_XANR 0              /* An R, nn x nl (<= 32); dead after barrier 2 -> C R1 (ny x nl <= 16) and K (nl x ny <= 16) */
#define LLPF_RBC_XCR 0
#define LLPF_RBC_XK 16
#define LLPF_RBC_XV 32               /* V', nn x nl */
#define LLPF_RBC_XXT 64              /* x~l */
#define LLPF_RBC_XM 72               /* M = Al R~, nl x nl; until barrier 2 its first nn entries carry the process noise */
#define LLPF_RBC_XNZ 72
#define LLPF_RBC_XTOT 136
#define LLPF_RBC_SYN 0               /* small buffer: measurement prediction yn[ny] */
#define LLPF_RBC_SLL 2               /*               log-likelihood increment     */

#if !defined(RBC_CTX_DECL)
/* the engine (kernels/rbfull.hpp): both exchange buffers are LDS, one row of 64 lanes per element; the barrier is the workgroup's */
#define RBC_CTX_DECL , double* rbc_xb, double* rbc_xs, const int rbc_lane
#define RBC_X(off) (rbc_xb[(off) * 64 + rbc_lane])
#define RBC_S(off) (rbc_xs[(off) * 64 + rbc_lane])
#define RBC_SYNC() __syncthreads()
#endif
#if defined(__HIPCC__)
#define RBC_FN __device__ __forceinline__
#else
#define RBC_FN static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RBC_CPTR(p) ((llpf_rbf_cptr)(p))
#define RBC_STAGE(ptr, dep) do { const double rbc_dep_ = (dep); asm volatile("" : "+s"(ptr) : "v"(rbc_dep_)); } while (0)
#define RBC_DONE(x) asm volatile("" : : "v"(x))
#define RBC_BLU(pp, nu, r, u, blu) ((blu)[r])
#else
#define RBC_CPTR(p) (p)
#define RBC_STAGE(ptr, dep) ((void)0)
#define RBC_DONE(x) ((void)0)
#define RBC_BLU(pp, nu, r, u, blu) llpf_rbf_blu_row(pp, nu, r, u)
#endif
#define RBC_OWN(c) (((c) % LLPF_RBC_NK) == k)

/* Kalman wave k (0..2).  xn: the nonlinear state BEFORE the step; xl, Rp (packed lower triangle; only entries in an owned column or
 * row are read): the linear state before the step; has_corr: a measurement update follows the time update (its y; the prediction
 * yn comes from the nonlinear wave).  Out: xl_out[c] for owned c; Rp_out[idx(r, c)] for owned c and r >= c.  Returns the
 * log-likelihood increment (0 without a measurement update); wave 0 also leaves it in the small buffer. */
RBC_FN double llpf_rbc_kalman(const llpf_rbf_par* p, const int nn, const int nl, const int ny, const int nu, const int k,
                               const int has_corr, const double* xn, const double* xl, const double* Rp, const double* u,
                               const double* blu, const double* y, double* xl_out, double* Rp_out RBC_CTX_DECL) {
    llpf_rbf_cptr pp = RBC_CPTR(p);
    (void)blu; (void)u; (void)nu;
    const int ht = (nl + 1) / 2;
    double Rq[LLPF_RBF_MAXL * LLPF_RBF_MAXL];                    /* owned columns of R, then of R~: Rq[r * nl + c] */
    double A[LLPF_RBF_MAXN * LLPF_RBF_MAXL];                     /* An(xn) */
    double AnR[LLPF_RBF_MAXN * LLPF_RBF_MAXL];                   /* An R; row i becomes row i of V' */
    double Nt[LLPF_RBF_MAXN * LLPF_RBF_MAXN], Lc[LLPF_RBF_MAXN * LLPF_RBF_MAXN], invd[LLPF_RBF_MAXN];
    double ax[LLPF_RBF_MAXN], v[LLPF_RBF_MAXN], nz[LLPF_RBF_MAXN], xt[LLPF_RBF_MAXL], xl1[LLPF_RBF_MAXL];
    double R1q[LLPF_RBF_MAXL * LLPF_RBF_MAXL];                   /* owned columns of R1 */
    LLPF_UNROLL
    for (int c = 0; c < nl; ++c) {
        if (!RBC_OWN(c)) continue;
        LLPF_UNROLL
        for (int r = 0; r < nl; ++r) Rq[r * nl + c] = Rp[llpf_rbf_idx(r, c)];
    }
    /* ---- An(xn), An xl, owned columns of An R (llpf_rbf_predict, first loop) ---- */
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {
        RBC_STAGE(pp, r == 0 ? Rq[k] : ax[r - 1]);
        llpf_rbf_coupling_row(pp, nn, nl, r, xn, A + r * nl);
        ax[r] = A[r * nl] * xl[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) ax[r] = llpf_fma(A[r * nl + c], xl[c], ax[r]);
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) AnR[r * nl + c] = A[r * nl] * Rq[c];
        LLPF_UNROLL
        for (int q = 1; q < nl; ++q) {
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) AnR[r * nl + c] = llpf_fma(A[r * nl + q], Rq[q * nl + c], AnR[r * nl + c]);
        }
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) RBC_X(LLPF_RBC_XANR + r * nl + c) = AnR[r * nl + c];
        RBC_DONE(ax[r]);
    }
    RBC_SYNC();                                                  /* 1: An R complete, the noise is there */
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (!RBC_OWN(c)) AnR[r * nl + c] = RBC_X(LLPF_RBC_XANR + r * nl + c);
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) nz[i] = RBC_X(LLPF_RBC_XNZ + i);
    RBC_STAGE(pp, AnR[nn * nl - 1]);
    LLPF_UNROLL
    for (int r = 0; r < nn; ++r) {                              /* Nt[r, j] = An[r, :] . (An R)[j, :] + R1n */
        LLPF_UNROLL
        for (int j = 0; j <= r; ++j) Nt[r * nn + j] = A[r * nl] * AnR[j * nl];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) {
            LLPF_UNROLL
            for (int j = 0; j <= r; ++j) Nt[r * nn + j] = llpf_fma(A[r * nl + c], AnR[j * nl + c], Nt[r * nn + j]);
        }
        LLPF_UNROLL
        for (int j = 0; j <= r; ++j) Nt[r * nn + j] = Nt[r * nn + j] + pp->R1n[r * nn + j];
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {                              /* Nt = Lc Lc' */
        LLPF_UNROLL
        for (int j = 0; j <= i; ++j) {
            double acc = Nt[i * nn + j];
            LLPF_UNROLL
            for (int q = 0; q < j; ++q) acc = llpf_fma(-Lc[i * nn + q], Lc[j * nn + q], acc);
            if (i == j) {
                const double d = llpf_sqrt(acc);
                Lc[i * nn + i] = d;
                invd[i] = 1.0 / d;
            } else {
                Lc[i * nn + j] = acc * invd[j];
            }
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {                              /* z = An xl + nz ; Lc v = z - An xl */
        const double z = ax[i] + nz[i];
        double acc = z - ax[i];
        LLPF_UNROLL
        for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * nn + q], v[q], acc);
        v[i] = acc * invd[i];
    }
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {                              /* Lc V' = An R, owned columns, in place */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {
            if (!RBC_OWN(c)) continue;
            double acc = AnR[i * nl + c];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Lc[i * nn + q], AnR[q * nl + c], acc);
            AnR[i * nl + c] = acc * invd[i];
            RBC_X(LLPF_RBC_XV + i * nl + c) = AnR[i * nl + c];
        }
    }
    LLPF_UNROLL
    for (int c = 0; c < nl; ++c) {                              /* x~l = xl + V v, owned entries */
        if (!RBC_OWN(c)) continue;
        double s = AnR[c] * v[0];
        LLPF_UNROLL
        for (int j = 1; j < nn; ++j) s = llpf_fma(AnR[j * nl + c], v[j], s);
        xt[c] = xl[c] + s;
        RBC_X(LLPF_RBC_XXT + c) = xt[c];
    }
    RBC_SYNC();                                                  /* 2: V and x~l complete */
    LLPF_UNROLL
    for (int i = 0; i < nn; ++i) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (!RBC_OWN(c)) AnR[i * nl + c] = RBC_X(LLPF_RBC_XV + i * nl + c);
    }
    LLPF_UNROLL
    for (int c = 0; c < nl; ++c) if (!RBC_OWN(c)) xt[c] = RBC_X(LLPF_RBC_XXT + c);
    LLPF_UNROLL
    for (int j = 0; j < nn; ++j) {                              /* R~ = R - V V', owned columns */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {
            if (!RBC_OWN(c)) continue;
            LLPF_UNROLL
            for (int r = 0; r < nl; ++r) Rq[r * nl + c] = llpf_fma(-AnR[j * nl + r], AnR[j * nl + c], Rq[r * nl + c]);
        }
    }
    LLPF_UNROLL
    for (int r0 = 0; r0 < nl; r0 += 2) {                        /* xl1 = Al x~l + Bl u (every Kalman wave; rows by pairs) */
        const int r1 = r0 + 2 < nl ? r0 + 2 : nl;
        RBC_STAGE(pp, r0 == 0 ? Rq[(nl - 1) * nl + k] : xl1[r0 - 1]);
        LLPF_UNROLL
        for (int r = r0; r < r1; ++r) xl1[r] = pp->Al[r * nl] * xt[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) {
            LLPF_UNROLL
            for (int r = r0; r < r1; ++r) xl1[r] = llpf_fma(pp->Al[r * nl + c], xt[c], xl1[r]);
        }
        LLPF_UNROLL
        for (int r = r0; r < r1; ++r) { xl1[r] = xl1[r] + RBC_BLU(pp, nu, r, u, blu); RBC_DONE(xl1[r]); }
    }
    {
        double M[LLPF_RBF_MAXL * LLPF_RBF_MAXL];                /* owned columns of M = Al R~ (all rows: both panels of the sequential form) */
        LLPF_UNROLL
        for (int r0 = 0; r0 < nl; r0 += 2) {
            const int r1 = r0 + 2 < nl ? r0 + 2 : nl;
            RBC_STAGE(pp, r0 == 0 ? xl1[nl - 1] : M[(r0 - 1) * nl + k]);
            LLPF_UNROLL
            for (int r = r0; r < r1; ++r) {
                LLPF_UNROLL
                for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) M[r * nl + c] = pp->Al[r * nl] * Rq[c];
            }
            LLPF_UNROLL
            for (int q = 1; q < nl; ++q) {
                LLPF_UNROLL
                for (int r = r0; r < r1; ++r) {
                    LLPF_UNROLL
                    for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) M[r * nl + c] = llpf_fma(pp->Al[r * nl + q], Rq[q * nl + c], M[r * nl + c]);
                }
            }
            LLPF_UNROLL
            for (int r = r0; r < r1; ++r) {
                LLPF_UNROLL
                for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) { RBC_X(LLPF_RBC_XM + r * nl + c) = M[r * nl + c]; RBC_DONE(M[r * nl + c]); }
            }
        }
    }
    RBC_SYNC();                                                  /* 3: M complete */
    /* R1 = M Al' + R1l, owned columns.  Entry (hi, lo), hi >= lo, of the sequential form: upper-left and lower-right blocks
     * M[hi, :] . Al[lo, :], lower-left block Al[hi, :] . M[lo, :]; the owned column c holds it at row r with {r, c} = {hi, lo}.
     * One row of M at a time (from the exchange buffer), every entry that row serves. */
    LLPF_UNROLL
    for (int m = 0; m < nl; ++m) {
        double Mr[LLPF_RBF_MAXL];
        LLPF_UNROLL
        for (int q = 0; q < nl; ++q) Mr[q] = RBC_X(LLPF_RBC_XM + m * nl + q);
        LLPF_UNROLL
        for (int a0 = 0; a0 < nl; a0 += 2) {
            RBC_STAGE(pp, Mr[nl - 1]);
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) {
                if (!RBC_OWN(c)) continue;
                LLPF_UNROLL
                for (int r = 0; r < nl; ++r) {
                    const int hi = r > c ? r : c, lo = r > c ? c : r;
                    const int diag = (hi < ht) || (lo >= ht);                    /* upper-left or lower-right block */
                    const int mrow = diag ? hi : lo, arow = diag ? lo : hi;
                    if (mrow != m || arow < a0 || arow >= a0 + 2) continue;
                    double acc = Mr[0] * pp->Al[arow * nl];
                    LLPF_UNROLL
                    for (int q = 1; q < nl; ++q) acc = llpf_fma(Mr[q], pp->Al[arow * nl + q], acc);
                    R1q[r * nl + c] = acc + pp->R1l[llpf_rbf_idx(hi, lo)];
                    RBC_DONE(R1q[r * nl + c]);
                }
            }
        }
    }
    if (!has_corr) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) {
            if (!RBC_OWN(c)) continue;
            xl_out[c] = xl1[c];
            LLPF_UNROLL
            for (int r = c; r < nl; ++r) Rp_out[llpf_rbf_idx(r, c)] = R1q[r * nl + c];
        }
        return 0.0;
    }
    /* ---- measurement update (llpf_rbf_correct) on xl1, R1 ---- */
    double e[LLPF_RBF_MAXY], CR[LLPF_RBF_MAXY * LLPF_RBF_MAXL], raw[LLPF_RBF_MAXY * LLPF_RBF_MAXY];
    double Ly[LLPF_RBF_MAXY * LLPF_RBF_MAXY], invy[LLPF_RBF_MAXY], K[LLPF_RBF_MAXL * LLPF_RBF_MAXY];
    RBC_STAGE(pp, R1q[(nl - 1) * nl + k]);
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        double a = pp->Cl[i * nl] * xl1[0];
        LLPF_UNROLL
        for (int c = 1; c < nl; ++c) a = llpf_fma(pp->Cl[i * nl + c], xl1[c], a);
        e[i] = (y[i] - RBC_S(LLPF_RBC_SYN + i)) - a;
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {                              /* C R1, owned columns */
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) CR[i * nl + c] = pp->Cl[i * nl] * R1q[c];
    }
    LLPF_UNROLL
    for (int q = 1; q < nl; ++q) {
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            LLPF_UNROLL
            for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) CR[i * nl + c] = llpf_fma(pp->Cl[i * nl + q], R1q[q * nl + c], CR[i * nl + c]);
        }
    }
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (RBC_OWN(c)) RBC_X(LLPF_RBC_XCR + i * nl + c) = CR[i * nl + c];
    }
    RBC_SYNC();                                                  /* 4: C R1 complete */
    LLPF_UNROLL
    for (int i = 0; i < ny; ++i) {
        LLPF_UNROLL
        for (int c = 0; c < nl; ++c) if (!RBC_OWN(c)) CR[i * nl + c] = RBC_X(LLPF_RBC_XCR + i * nl + c);
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
    for (int i = 0; i < ny; ++i) {                              /* S = 0.5 (raw + raw') + R2 = Ly Ly' */
        LLPF_UNROLL
        for (int j = 0; j <= i; ++j) {
            double acc = 0.5 * (raw[i * ny + j] + raw[j * ny + i]) + pp->R2[i * ny + j];
            LLPF_UNROLL
            for (int q = 0; q < j; ++q) acc = llpf_fma(-Ly[i * ny + q], Ly[j * ny + q], acc);
            if (i == j) {
                const double d = llpf_sqrt(acc);
                Ly[i * ny + i] = d;
                invy[i] = 1.0 / d;
                ldet = ldet + llpf_log(d);
            } else {
                Ly[i * ny + j] = acc * invy[j];
            }
        }
    }
    double quad = 0.0;
    {
        double z[LLPF_RBF_MAXY];
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            double acc = e[i];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Ly[i * ny + q], z[q], acc);
            z[i] = acc * invy[i];
            quad = llpf_fma(z[i], z[i], quad);
        }
    }
    const double ll = (pp->c0y - ldet) - 0.5 * quad;
    if (k == 0) RBC_S(LLPF_RBC_SLL) = ll;
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {                              /* owned rows of K; owned entries of xl */
        if (!RBC_OWN(r)) continue;
        double t[LLPF_RBF_MAXY];
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) {
            double acc = CR[i * nl + r];
            LLPF_UNROLL
            for (int q = 0; q < i; ++q) acc = llpf_fma(-Ly[i * ny + q], t[q], acc);
            t[i] = acc * invy[i];
        }
        LLPF_UNROLL
        for (int i = ny - 1; i >= 0; --i) {
            double acc = t[i];
            LLPF_UNROLL
            for (int q = i + 1; q < ny; ++q) acc = llpf_fma(-Ly[q * ny + i], K[r * ny + q], acc);
            K[r * ny + i] = acc * invy[i];
        }
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) RBC_X(LLPF_RBC_XK + r * ny + i) = K[r * ny + i];
        double a = K[r * ny] * e[0];
        LLPF_UNROLL
        for (int i = 1; i < ny; ++i) a = llpf_fma(K[r * ny + i], e[i], a);
        xl_out[r] = xl1[r] + a;
    }
    RBC_SYNC();                                                  /* 5: K complete */
    LLPF_UNROLL
    for (int r = 0; r < nl; ++r) {
        if (RBC_OWN(r)) continue;
        LLPF_UNROLL
        for (int i = 0; i < ny; ++i) K[r * ny + i] = RBC_X(LLPF_RBC_XK + r * ny + i);
    }
    LLPF_UNROLL
    for (int c = 0; c < nl; ++c) {                              /* R = symmetrize(R1 - K (C R1)), owned columns, lower part */
        if (!RBC_OWN(c)) continue;
        LLPF_UNROLL
        for (int r = c; r < nl; ++r) {
            double a = R1q[r * nl + c], b2 = a;
            LLPF_UNROLL
            for (int i = 0; i < ny; ++i) {
                a = llpf_fma(-K[r * ny + i], CR[i * nl + c], a);
                b2 = llpf_fma(-K[c * ny + i], CR[i * nl + r], b2);
            }
            Rp_out[llpf_rbf_idx(r, c)] = 0.5 * (a + b2);
        }
    }
    return ll;
}
#undef RBC_OWN

#endif /* LLPF_RBFULL_COOP_H */
