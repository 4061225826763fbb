/* parked with the experiment (EXPERIMENTS.md 5.2): this block was appended to oracle/llpf_oracle.c while the cooperative form was in the product;
 * tests/test_oracle_rbfull.py ran it (four threads as the four waves) against llpf_rbf_predict + llpf_rbf_correct: 0 differing words on all shapes. */

/* ------------------------------------------------------------------------------------------------------------------------------
 * Host emulation of the engine's COOPERATIVE form of the per-particle Kalman recursion (csrc/shared/llpf_rbfull_coop.h: one batch
 * of particles shared by four waves — wave 0 the nonlinear state, three Kalman waves owning the columns c = k mod 3).  Four
 * threads play the four waves, a pthread barrier the workgroup barrier, a plain array the LDS exchange buffer.  Checked against
 * the sequential shared form (llpf_rbf_predict + llpf_rbf_correct), which the device-order oracle and the rest of the kernel use:
 * the split must not change a bit.  Test infrastructure only (tests/test_oracle_rbfull.py).
 * ------------------------------------------------------------------------------------------------------------------------------ */
#include <pthread.h>
#define RBC_CTX_DECL , double* rbc_xb, double* rbc_xs, pthread_barrier_t* rbc_bar
#define RBC_X(off) (rbc_xb[(off)])
#define RBC_S(off) (rbc_xs[(off)])
#define RBC_SYNC() pthread_barrier_wait(rbc_bar)
#include "../lowlevelparticlefilters.jl_amd/csrc/shared/llpf_rbfull_coop.h"

typedef struct {
    const llpf_rbf_par* par;
    int nn, nl, ny, nu, role, has_corr;
    int64_t n;
    const double *xn, *xl, *Rp, *u, *y, *nz, *yn;     /* [n][...] inputs */
    double *xl_out, *Rp_out, *ll_out;                /* [n][...] outputs: every Kalman wave fills its own entries */
    double *xb, *xs;
    pthread_barrier_t* bar;
} rbc_job;

static void* rbc_thread(void* arg) {
    rbc_job* jb = (rbc_job*)arg;
    const int nn = jb->nn, nl = jb->nl, ny = jb->ny, np = LLPF_RBF_NP(jb->nl);
    for (int64_t i = 0; i < jb->n; ++i) {
        if (jb->role == 0) {
            /* the nonlinear wave: publishes the process noise before barrier 1 and the measurement prediction before barrier 2,
             * then only keeps the barrier count (its own arithmetic — dynamics, xn1 = fi + (An xl + nz) — is not part of the split) */
            for (int d = 0; d < nn; ++d) jb->xb[LLPF_RBC_XNZ + d] = jb->nz[i * nn + d];
            pthread_barrier_wait(jb->bar);
            for (int d = 0; d < ny; ++d) jb->xs[LLPF_RBC_SYN + d] = jb->yn[i * ny + d];
            pthread_barrier_wait(jb->bar);
            pthread_barrier_wait(jb->bar);
            if (jb->has_corr) { pthread_barrier_wait(jb->bar); pthread_barrier_wait(jb->bar); jb->ll_out[i] = jb->xs[LLPF_RBC_SLL]; }
        } else {
            double xlo[LLPF_RBF_MAXL], Rpo[LLPF_RBF_NP(LLPF_RBF_MAXL)];
            llpf_rbc_kalman(jb->par, nn, nl, ny, jb->nu, jb->role - 1, jb->has_corr, jb->xn + i * nn, jb->xl + i * nl, jb->Rp + i * np,
                            jb->u, NULL, jb->y + i * ny, xlo, Rpo, jb->xb, jb->xs, jb->bar);
            for (int c = 0; c < nl; ++c) {
                if (c % LLPF_RBC_NK != jb->role - 1) continue;
                jb->xl_out[i * nl + c] = xlo[c];
                for (int r = c; r < nl; ++r) jb->Rp_out[i * np + llpf_rbf_idx(r, c)] = Rpo[llpf_rbf_idx(r, c)];
            }
        }
        pthread_barrier_wait(jb->bar);       /* the exchange buffer is free for the next particle */
    }
    return NULL;
}

/* Runs n random particles of filter f's Rao-Blackwellized model (LLPF_MODEL_RB_BILINEAR) through both forms.  Returns the number of
 * output words (xl, packed R, ll) that differ in any bit; -1 on a set-up error.  xl_seq / R_seq / ll_seq (optional) receive the
 * sequential form's outputs of the LAST particle, for the caller's sanity checks. */
int64_t orc_rbf_coop_check(orc_filter* f, int has_corr, int64_t n, uint64_t seed, double* xl_seq, double* R_seq, double* ll_seq) {
    if (!f || !f->rbf.on) return -1;
    const int nn = f->rbf.nn, nl = f->rbf.nl, np = f->rbf.np, ny = f->ny, nu = f->nu;
    const llpf_rbf_par* par = &f->rbf.par;
    double *xn = calloc((size_t)n * nn, 8), *xl = calloc((size_t)n * nl, 8), *Rp = calloc((size_t)n * np, 8), *nz = calloc((size_t)n * nn, 8);
    double *fi = calloc((size_t)n * nn, 8), *y = calloc((size_t)n * ny, 8), *yn = calloc((size_t)n * ny, 8), u[8] = {0.3, -0.7, 0.2, 0.9, 0.1, -0.4, 0.6, -0.2};
    double *xl_a = calloc((size_t)n * nl, 8), *R_a = calloc((size_t)n * np, 8), *ll_a = calloc((size_t)n, 8);
    double *xl_b = calloc((size_t)n * nl, 8), *R_b = calloc((size_t)n * np, 8), *ll_b = calloc((size_t)n, 8), *xn1 = calloc((size_t)n * nn, 8);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t i = 0; i < n; ++i) {
        double g[64];
        llpf_normals((uint32_t)i, 1, 1, k0, k1, 4, g); llpf_normals((uint32_t)i, 2, 1, k0, k1, 4, g + 4);
        llpf_normals((uint32_t)i, 3, 1, k0, k1, 4, g + 8); llpf_normals((uint32_t)i, 4, 1, k0, k1, 4, g + 12);
        llpf_normals((uint32_t)i, 5, 1, k0, k1, 4, g + 16); llpf_normals((uint32_t)i, 6, 1, k0, k1, 4, g + 20);
        for (int d = 0; d < nn; ++d) { xn[i * nn + d] = 1.0 + 0.5 * g[d]; nz[i * nn + d] = 0.3 * g[4 + d]; fi[i * nn + d] = 1.0 + 0.4 * g[8 + d]; }
        for (int d = 0; d < nl; ++d) xl[i * nl + d] = g[12 + d];
        for (int d = 0; d < ny; ++d) { y[i * ny + d] = 2.0 * g[20 + d]; }
        /* a positive definite covariance: B B' + 0.5 I with B lower triangular from further normals */
        double Bm[LLPF_RBF_MAXL * LLPF_RBF_MAXL] = {0};
        for (int r = 0; r < nl; ++r) {
            double h[4];
            for (int c = 0; c <= r; c += 4) {
                llpf_normals((uint32_t)i, (uint32_t)(10 + r * 4 + c / 4), 1, k0, k1, 4, h);
                for (int q = 0; q < 4 && c + q <= r; ++q) Bm[r * nl + c + q] = 0.6 * h[q];
            }
        }
        for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c) {
            double sacc = (r == c) ? 0.5 : 0.0;
            for (int q = 0; q <= c; ++q) sacc += Bm[r * nl + q] * Bm[c * nl + q];
            Rp[i * np + llpf_rbf_idx(r, c)] = sacc;
        }
        /* sequential form */
        llpf_rbf_predict(par, nn, nl, nu, xn + i * nn, xl + i * nl, Rp + i * np, u, NULL, fi + i * nn, nz + i * nn, xn1 + i * nn, xl_a + i * nl, R_a + i * np);
        for (int d = 0; d < ny; ++d) yn[i * ny + d] = 0.5 * xn1[i * nn + (d < nn ? d : 0)] + 0.1 * d;
        ll_a[i] = has_corr ? llpf_rbf_correct(par, nl, ny, y + i * ny, yn + i * ny, xl_a + i * nl, R_a + i * np) : 0.0;
    }
    double xb[LLPF_RBC_XTOT], xs[4];
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, 1 + LLPF_RBC_NK);
    rbc_job jobs[1 + LLPF_RBC_NK];
    pthread_t th[1 + LLPF_RBC_NK];
    for (int r = 0; r < 1 + LLPF_RBC_NK; ++r) {
        rbc_job jb = {par, nn, nl, ny, nu, r, has_corr, n, xn, xl, Rp, u, y, nz, yn, xl_b, R_b, ll_b, xb, xs, &bar};
        jobs[r] = jb;
        pthread_create(&th[r], NULL, rbc_thread, &jobs[r]);
    }
    for (int r = 0; r < 1 + LLPF_RBC_NK; ++r) pthread_join(th[r], NULL);
    pthread_barrier_destroy(&bar);
    int64_t bad = 0;
    for (int64_t i = 0; i < n * nl; ++i) bad += memcmp(&xl_a[i], &xl_b[i], 8) != 0;
    for (int64_t i = 0; i < n * np; ++i) bad += memcmp(&R_a[i], &R_b[i], 8) != 0;
    if (has_corr) for (int64_t i = 0; i < n; ++i) bad += memcmp(&ll_a[i], &ll_b[i], 8) != 0;
    if (n > 0) {
        if (xl_seq) memcpy(xl_seq, xl_a + (n - 1) * nl, sizeof(double) * nl);
        if (R_seq) memcpy(R_seq, R_a + (n - 1) * np, sizeof(double) * np);
        if (ll_seq) *ll_seq = ll_a[n - 1];
    }
    free(xn); free(xl); free(Rp); free(nz); free(fi); free(y); free(yn); free(xl_a); free(R_a); free(ll_a); free(xl_b); free(R_b); free(ll_b); free(xn1);
    return bad;
}
