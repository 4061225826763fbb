/* llpf_rbfull.h — the per-particle Kalman recursion of the Rao-Blackwellized particle filter whose coupling matrix
 * An depends on the nonlinear state (reference src/rbpf.jl:163-283 with the "singleR" shortcut :176/:247 off: every
 * particle carries its own covariance R; LLPF_MODEL_RB_BILINEAR, BASELINE config C5).
 *
 * Shared by the HIP kernel (kernels/rbfull.hpp: one particle per thread, every loop below unrolls because the dimensions
 * are literal constants at the call site) and by the oracle, so that both run the same IEEE sequence.  The body
 * (llpf_rbfull_body.h) is a macro template over the square root / logarithm: this header instantiates it with the
 * deterministic device-order functions (prefix llpf_rbf_); the oracle instantiates it a second time with libm for its
 * reference order (prefix llpf_rbfr_).
 *
 * The covariance is stored as its packed lower triangle, entry (r,c), c <= r, at r(r+1)/2 + c: 36 numbers for nxl = 8. */
#ifndef LLPF_RBFULL_H
#define LLPF_RBFULL_H

#include "llpf_detmath.h"

#define LLPF_RBF_MAXN 4
#define LLPF_RBF_MAXL 8
#define LLPF_RBF_MAXY 2
#define LLPF_RBF_NP(nl) ((nl) * ((nl) + 1) / 2)

#if defined(__HIP_DEVICE_COMPILE__)
#define LLPF_UNROLL _Pragma("unroll")
#else
#define LLPF_UNROLL
#endif

/* constant parameters of one filter (device memory, read through scalar loads) */
typedef struct llpf_rbf_par {
    int32_t nn, nl, ny, nu;
    double Al[LLPF_RBF_MAXL * LLPF_RBF_MAXL];          /* nl x nl, stride nl   (kf.A)  */
    double Bl[LLPF_RBF_MAXL * 8];                      /* nl x nu, stride nu   (kf.B)  */
    double Cl[LLPF_RBF_MAXY * LLPF_RBF_MAXL];          /* ny x nl, stride nl   (kf.C)  */
    double An[1 + LLPF_RBF_MAXN][LLPF_RBF_MAXN * LLPF_RBF_MAXL];   /* An(xn) = An[0] + sum_k xn[k] An[1+k], each nn x nl */
    double R1l[LLPF_RBF_NP(LLPF_RBF_MAXL)];            /* packed lower triangle of kf.R1 */
    double R1n[LLPF_RBF_MAXN * LLPF_RBF_MAXN];         /* nn x nn dense: pf.R1n.Sigma   */
    double R2[LLPF_RBF_MAXY * LLPF_RBF_MAXY];          /* ny x ny dense                 */
    double xl0[LLPF_RBF_MAXL];                         /* kf.d0.mu                      */
    double R0[LLPF_RBF_NP(LLPF_RBF_MAXL)];             /* packed lower triangle of kf.d0.Sigma */
    double c0y;                                        /* -(ny/2) log(2 pi), in the order's own log */
} llpf_rbf_par;

LLPF_HD int llpf_rbf_idx(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }

#define RBF_(name) llpf_rbf_##name
#define RBF_SQRT(x) llpf_sqrt(x)
#define RBF_LOG(x) llpf_log(x)
#include "llpf_rbfull_body.h"
#undef RBF_
#undef RBF_SQRT
#undef RBF_LOG

#endif /* LLPF_RBFULL_H */
