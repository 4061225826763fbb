/* llpf_rbfull.h — the per-particle Kalman recursion of the Rao-Blackwellized particle filter whose coupling matrix
 * An depends on the nonlinear state (reference src/rbpf.jl:163-283 with the "singleR" shortcut :176/:247 off: every
 * particle carries its own covariance R; LLPF_MODEL_RB_BILINEAR, BASELINE config C5).
 *
 * Shared by the HIP kernel (kernels/rbfull.hpp: one particle per thread, every loop below unrolls because the dimensions
 * are literal constants at the call site) and by the oracle's DEVICE order, so that both run the same IEEE sequence.  The
 * oracle's reference order does not use this file: it restates the reference's formulas literally (oracle/llpf_oracle.c,
 * rbfr_predict / rbfr_correct), so the two orders check each other to rounding.
 *
 * The covariance is stored as its packed lower triangle, entry (r,c), c <= r, at r(r+1)/2 + c: 36 numbers for nxl = 8. */
#ifndef LLPF_RBFULL_H
#define LLPF_RBFULL_H

#include "llpf_detmath.h"

#define LLPF_RBF_MAXN 4
#define LLPF_RBF_MAXL 8
#define LLPF_RBF_MAXY 4
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
#ifndef RBF_STAMP
#define RBF_STAMP(k) ((void)0)      /* tools/rbfull_probe.hip defines it: cycle stamps at the stage boundaries */
#endif
#define RBF_SQRT(x) llpf_sqrt(x)
#define RBF_LOG(x) llpf_log(x)
#if defined(__HIP_DEVICE_COMPILE__)
/* Device: the parameters are read through the CONSTANT address space (scalar loads, whatever the compiler can or cannot prove
 * about the pointer), and at a stage boundary the pointer re-emerges from an empty asm that also reads `dep`: loads through it
 * can neither be hoisted above the point where `dep` is computed nor merged with earlier loads (llpf_rbfull_body.h, "Stages"). */
typedef const __attribute__((address_space(4))) llpf_rbf_par* llpf_rbf_cptr;
#define RBF_CPTR(p) ((llpf_rbf_cptr)(p))
#define RBF_STAGE(ptr, dep) do { const double rbf_dep_ = (dep); asm volatile("" : "+s"(ptr) : "v"(rbf_dep_)); } while (0)
#define RBF_BLU(pp, nu, r, u, blu) ((blu)[r])
/* every element of arr[i0, i1) is computed before the next stage begins: the stage's own `dep` orders only the chain that
 * value hangs on, and the compiler postponed the other rows of a stage to the end of the body — with every stage's pointer
 * and constants still alive in SGPRs */
#define RBF_DONE(arr, i0, i1) do { LLPF_UNROLL for (int rbf_i_ = (i0); rbf_i_ < (i1); ++rbf_i_) asm volatile("" : : "v"((arr)[rbf_i_])); } while (0)
/* no instruction may be scheduled across this point (keeps two panels of the time update from being live together) */
#define RBF_FENCE(dep) __builtin_amdgcn_sched_barrier(0)
#else
typedef const llpf_rbf_par* llpf_rbf_cptr;
#define RBF_CPTR(p) (p)
#define RBF_STAGE(ptr, dep) ((void)0)
#define RBF_FENCE(dep) ((void)0)
#define RBF_DONE(arr, i0, i1) ((void)0)
#define RBF_BLU(pp, nu, r, u, blu) RBF_(blu_row)(pp, nu, r, u)
#endif
#include "llpf_rbfull_body.h"
#undef RBF_
#undef RBF_SQRT
#undef RBF_LOG
#undef RBF_STAGE
#undef RBF_FENCE
#undef RBF_BLU
#undef RBF_DONE
#undef RBF_CPTR

#endif /* LLPF_RBFULL_H */
