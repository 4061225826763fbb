/* llpf_rbkf.h — the shared linear-substate covariance recursion of the Rao-Blackwellized particle filter with
 * constant matrices (reference src/rbpf.jl:163-283, "singleR" special case :176, :247: one Riccati recursion serves
 * all particles because A, An, C, R1 are plain matrices).  Host-only plain C, included by the engine's host code and
 * by the oracle so that both produce the same gains to the bit (compile with -ffp-contract=off).
 *
 * Dimensions: nl = linear states, nn = nonlinear states, ny = outputs, all <= 4; matrices row-major, dense. */
#ifndef LLPF_RBKF_H
#define LLPF_RBKF_H

#define LLPF_RB_MAX 4

/* C[m x n] = A[m x k] * B[k x n], accumulation over k in increasing order */
static void llpf_rb_mul(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c) {
            double a = A[r * k + 0] * B[0 * n + c];
            for (int q = 1; q < k; ++q) a = a + A[r * k + q] * B[q * n + c];
            C[r * n + c] = a;
        }
}
/* C[m x n] = A[m x k] * B'   (B is n x k) */
static void llpf_rb_mul_t(const double* A, const double* B, double* C, int m, int k, int n) {
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < n; ++c) {
            double a = A[r * k + 0] * B[c * k + 0];
            for (int q = 1; q < k; ++q) a = a + A[r * k + q] * B[c * k + q];
            C[r * n + c] = a;
        }
}
/* symmetrize(x) = 0.5 .* (x .+ x')  — reference src/filtering.jl:83-86 */
static void llpf_rb_symmetrize(double* X, int n) {
    for (int r = 0; r < n; ++r)
        for (int c = r + 1; c < n; ++c) {
            const double a = 0.5 * (X[r * n + c] + X[c * n + r]);
            X[r * n + c] = a;
            X[c * n + r] = a;
        }
    for (int r = 0; r < n; ++r) X[r * n + r] = 0.5 * (X[r * n + r] + X[r * n + r]);
}
/* lower Cholesky factor, row by row; returns nonzero if not positive definite */
static int llpf_rb_chol(const double* S, int n, double* L, double (*sq)(double)) {
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double acc = S[i * n + j];
            for (int k = 0; k < j; ++k) acc = acc - L[i * n + k] * L[j * n + k];
            if (i == j) {
                if (!(acc > 0.0)) return -1;
                L[i * n + i] = sq(acc);
            } else {
                L[i * n + j] = acc / L[j * n + j];
            }
        }
    return 0;
}

/* Measurement update of the shared covariance — correct!(kf, ...) reference src/filtering.jl:100-128 (R12 === nothing):
 *   S = symmetrize(C R C') + R2 ; S_chol = cholesky(S) ; K = (R C') / S_chol ; Rpost = symmetrize((I - K C) R)
 * R [nl x nl], C [ny x nl], R2 [ny x ny]; outputs S [ny x ny], K [nl x ny], Rpost [nl x nl]. */
static int llpf_rb_gain(int nl, int ny, const double* R, const double* C, const double* R2, double* S, double* K,
                        double* Rpost, double (*sq)(double)) {
    double CR[LLPF_RB_MAX * LLPF_RB_MAX], RCt[LLPF_RB_MAX * LLPF_RB_MAX], Lc[LLPF_RB_MAX * LLPF_RB_MAX];
    llpf_rb_mul(C, R, CR, ny, nl, nl);                 /* (C R) C' */
    llpf_rb_mul_t(CR, C, S, ny, nl, ny);
    llpf_rb_symmetrize(S, ny);
    for (int i = 0; i < ny * ny; ++i) S[i] = S[i] + R2[i];
    if (llpf_rb_chol(S, ny, Lc, sq)) return -1;
    llpf_rb_mul_t(R, C, RCt, nl, nl, ny);              /* R C' */
    for (int r = 0; r < nl; ++r) {                     /* row r of K solves k S = (R C')[r,:]  via  L L' */
        double t1[LLPF_RB_MAX] = {0, 0, 0, 0}, t2[LLPF_RB_MAX] = {0, 0, 0, 0};
        for (int i = 0; i < ny; ++i) {
            double acc = RCt[r * ny + i];
            for (int q = 0; q < i; ++q) acc = acc - Lc[i * ny + q] * t1[q];
            t1[i] = acc / Lc[i * ny + i];
        }
        for (int i = ny - 1; i >= 0; --i) {
            double acc = t1[i];
            for (int q = i + 1; q < ny; ++q) acc = acc - Lc[q * ny + i] * t2[q];
            t2[i] = acc / Lc[i * ny + i];
        }
        for (int c = 0; c < ny; ++c) K[r * ny + c] = t2[c];
    }
    double KC[LLPF_RB_MAX * LLPF_RB_MAX], IKC[LLPF_RB_MAX * LLPF_RB_MAX];
    llpf_rb_mul(K, C, KC, nl, ny, nl);
    for (int r = 0; r < nl; ++r)
        for (int c = 0; c < nl; ++c) IKC[r * nl + c] = (r == c ? 1.0 : 0.0) - KC[r * nl + c];
    llpf_rb_mul(IKC, R, Rpost, nl, nl, nl);
    llpf_rb_symmetrize(Rpost, nl);
    return 0;
}

/* Time update of the shared covariance — reference src/rbpf.jl:203-219.
 *   An == 0 : R1 = Al R Al' + R1l                                     (L unused)
 *   else    : Nt = An R An' + R1n ; L = (Al R An') / Nt ; R1 = Al R Al' + R1l - L Nt L'
 * The right division by Nt is implemented for nn == 1 (a scalar, as in the reference's own test). */
static int llpf_rb_predcov(int nl, int nn, int zeroAn, const double* R, const double* Al, const double* An,
                           const double* R1l, const double* R1n, double* L, double* R1) {
    double AR[LLPF_RB_MAX * LLPF_RB_MAX], ARA[LLPF_RB_MAX * LLPF_RB_MAX];
    llpf_rb_mul(Al, R, AR, nl, nl, nl);
    llpf_rb_mul_t(AR, Al, ARA, nl, nl, nl);
    if (zeroAn) {
        for (int i = 0; i < nl * nl; ++i) R1[i] = ARA[i] + R1l[i];
        for (int i = 0; i < nl * nn; ++i) L[i] = 0.0;
        return 0;
    }
    if (nn != 1) return -1;
    double AnR[LLPF_RB_MAX], Nt, ARAn[LLPF_RB_MAX];
    llpf_rb_mul(An, R, AnR, 1, nl, nl);                /* An R An' + R1n */
    llpf_rb_mul_t(AnR, An, &Nt, 1, nl, 1);
    Nt = Nt + R1n[0];
    llpf_rb_mul_t(AR, An, ARAn, nl, nl, 1);            /* (Al R) An' */
    for (int r = 0; r < nl; ++r) L[r] = ARAn[r] / Nt;
    for (int r = 0; r < nl; ++r)
        for (int c = 0; c < nl; ++c) {
            const double lnl = (L[r] * Nt) * L[c];     /* L*Nt*L' left to right */
            R1[r * nl + c] = (ARA[r * nl + c] + R1l[r * nl + c]) - lnl;
        }
    return 0;
}

#endif /* LLPF_RBKF_H */
