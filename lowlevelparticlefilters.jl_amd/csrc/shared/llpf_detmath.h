/* llpf_detmath.h — deterministic fp64 elementary functions, shared by host and device.
 *
 * Why this exists: the particle-filter recursion feeds exp() of the log-weights into an
 * integer prefix sum that decides the resampling ancestors, and feeds log/sqrt/sincos
 * (Box–Muller) into the particles.  To make a GPU trajectory *bit-identical* to a CPU
 * restatement, every transcendental on the recursion's feedback path must be the same
 * sequence of IEEE-754 operations on both sides.  The functions below use only
 * + - * /, fma, sqrt, rint and integer bit manipulation, all of which are correctly
 * rounded on x86-64 and on gfx950 (verified on hardware by tests/test_gpu_math.py), and
 * the translation units that include this header are compiled with -ffp-contract=off so
 * the only fused operations are the explicit llpf_fma() calls.
 *
 * The reference calls SLEEFPirates.exp (reference src/utils.jl:5; <1 ulp, not
 * libm-identical) and Julia's randn / log1p; any <1 ulp implementation is an equally valid
 * restatement.  Accuracy of every function here is checked against 80-bit long double in
 * tests/test_detmath.py (all < 1 ulp).
 *
 * Polynomial coefficients for log / sin / cos are the classic minimax sets published with
 * FreeBSD msun (fdlibm, Sun Microsystems 1993, freely redistributable).
 */
#ifndef LLPF_DETMATH_H
#define LLPF_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define LLPF_HD __host__ __device__ __forceinline__
#else
#define LLPF_HD static inline __attribute__((always_inline))
#endif

LLPF_HD uint64_t llpf_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
LLPF_HD double   llpf_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
LLPF_HD double   llpf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* one Horner step q*r + c with a constant c.  Same fused operation as llpf_fma; on the device it is pinned to the
 * three-address v_fma_f64 (the compiler otherwise copies the constant into the destination for a two-address v_fmac) */
#ifndef LLPF_HORNER_C
#define LLPF_HORNER_C(c) "s"(c)      /* the constant as an SGPR pair: built by the scalar unit, no VGPR moves */
#endif
/* Horner step q r + c pinned to the three-address v_fma_f64.  llpf_horner_k: the engine's own polynomial sites, whose c is a
 * LITERAL — bound as an SGPR pair by default (LLPF_HORNER_C), i.e. c must be wave-uniform; llpf_horner: any operands (this header is
 * part of the prelude of run-time compiled user models, where a per-lane c would silently read lane 0's value through the "s" form). */
LLPF_HD double   llpf_horner_k(double q, double r, double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LLPF_NO_ASM_HORNER)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(q), "v"(r), LLPF_HORNER_C(c));
    return d;
#else
    return __builtin_fma(q, r, c);
#endif
}
LLPF_HD double   llpf_horner(double q, double r, double c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LLPF_NO_ASM_HORNER)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(q), "v"(r), "v"(c));
    return d;
#else
    return __builtin_fma(q, r, c);
#endif
}
LLPF_HD double   llpf_sqrt(double a) { return __builtin_sqrt(a); }
/* sqrt(a) for a NORMAL, finite, strictly positive a that is neither tiny nor huge (2^-700 < a < 2^700): the
 * correctly-rounded result, i.e. the same bits as llpf_sqrt.  On the device the compiler's IEEE expansion of sqrt
 * spends 7 of its 17 instructions on scaling of denormal inputs and on 0 / inf / NaN; this is its core alone:
 * y = rsq(a), g = a y, h = y/2, one Goldschmidt refinement of (g, h), two residual corrections of g. */
#if defined(__HIP_DEVICE_COMPILE__)
LLPF_HD double   llpf_sqrt_pos(double a) {
    const double y  = __builtin_amdgcn_rsq(a);
    const double g0 = a * y;
    const double h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, a);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, a);
    return __builtin_fma(d1, h1, g2);
}
#else
LLPF_HD double   llpf_sqrt_pos(double a) { return __builtin_sqrt(a); }
#endif
LLPF_HD double   llpf_rint(double a) { return __builtin_rint(a); }
LLPF_HD double   llpf_fmax(double a, double b) { return a > b ? a : b; }
LLPF_HD double   llpf_fabs(double a) { return __builtin_fabs(a); }

#define LLPF_INF (__builtin_inf())

/* 2^k for k in [-1022, 1023] */
LLPF_HD double llpf_pow2i(int k) { return llpf_u2d((uint64_t)(k + 1023) << 52); }

/* exp(x).  k = rint(x/ln2), r = x - k ln2 (two-term Cody–Waite with fma), degree-13 Taylor
 * polynomial on |r| <= ln2/2 (truncation 4e-18 relative), result scaled by 2^k in two exact
 * steps so that a subnormal result is rounded once.  exp(0) == 1 exactly. */
LLPF_HD double llpf_exp_core(double xc) {      /* xc in [-746, 709.78] or NaN */
    const double LOG2E  = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    double kf = llpf_rint(xc * LOG2E);
    double r = llpf_fma(-kf, LN2_HI, xc);
    r = llpf_fma(-kf, LN2_LO, r);
    /* q(r) = 1/2! + r/3! + ... + r^11/13! */
    double q = 1.6059043836821613e-10;              /* 1/13! */
    q = llpf_horner_k(q, r, 2.08767569878681e-09);       /* 1/12! */
    q = llpf_horner_k(q, r, 2.505210838544172e-08);      /* 1/11! */
    q = llpf_horner_k(q, r, 2.755731922398589e-07);      /* 1/10! */
    q = llpf_horner_k(q, r, 2.7557319223985893e-06);     /* 1/9!  */
    q = llpf_horner_k(q, r, 2.48015873015873e-05);       /* 1/8!  */
    q = llpf_horner_k(q, r, 1.984126984126984e-04);      /* 1/7!  */
    q = llpf_horner_k(q, r, 1.388888888888889e-03);      /* 1/6!  */
    q = llpf_horner_k(q, r, 8.333333333333333e-03);      /* 1/5!  */
    q = llpf_horner_k(q, r, 4.1666666666666664e-02);     /* 1/4!  */
    q = llpf_horner_k(q, r, 1.6666666666666666e-01);     /* 1/3!  */
    q = llpf_horner_k(q, r, 0.5);                        /* 1/2!  */
    double p = llpf_fma(r * r, q, r);               /* r + r^2 q(r) */
    double y = 1.0 + p;
    /* kf is an integer in [-1077, 1024] (or NaN, in which case y is NaN and the scale factors are irrelevant
     * but must not trap: the conversion below is done on a clamped copy) */
    double kc = kf == kf ? kf : 0.0;
    int k = (int)kc;
#if defined(__HIP_DEVICE_COMPILE__) && defined(LLPF_EXP_LDEXP)
    /* experiment (EXPERIMENTS 5.x): v_ldexp_f64 scales in one instruction and rounds a subnormal result once, like the two exact
     * steps below — bit-identity with the host is checked by the host-vs-device test of the shared math */
    return __builtin_amdgcn_ldexp(y, k);
#else
    int k1 = k / 2, k2 = k - k1;
    return (y * llpf_pow2i(k1)) * llpf_pow2i(k2);
#endif
}

LLPF_HD double llpf_exp(double x) {
    if (x > 709.782712893384) return LLPF_INF;
    return llpf_exp_core(x < -746.0 ? -746.0 : x);  /* NaN falls through the clamp and propagates */
}

/* exp for arguments known to be <= 0 (exp-weights exp(w - max w)); no overflow branch */
LLPF_HD double llpf_exp_le0(double x) {
    return llpf_exp_core(x < -746.0 ? -746.0 : x);
}

/* log(x) for x > 0 (fdlibm algorithm: x = 2^k (1+f), s = f/(2+f), log(1+f) = f - hfsq + s (hfsq + R(s^2))).
 * x == 0 -> -inf, x < 0 or NaN -> NaN.  < 1 ulp. */
LLPF_HD double llpf_log(double x) {
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = llpf_d2u(x);
    int k = 0;
    if (x != x) return x;
    if (x <= 0.0) return x == 0.0 ? -LLPF_INF : llpf_u2d(0x7ff8000000000000ULL);
    if ((u >> 52) == 0x7ff) return x;                 /* +inf */
    if ((u >> 52) == 0) {                             /* subnormal: scale up by 2^54 */
        x = x * 18014398509481984.0;
        u = llpf_d2u(x);
        k = -54;
    }
    uint32_t hx = (uint32_t)(u >> 32);
    hx += 0x3ff00000u - 0x3fe6a09eu;
    k += (int)(hx >> 20) - 0x3ff;
    hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
    x = llpf_u2d(((uint64_t)hx << 32) | (u & 0xffffffffULL));   /* x in [sqrt(2)/2, sqrt(2)) */
    double f = x - 1.0;
    double hfsq = 0.5 * f * f;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * llpf_fma(w, llpf_fma(w, Lg6, Lg4), Lg2);
    double t2 = z * llpf_fma(w, llpf_fma(w, llpf_fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double dk = (double)k;
    return llpf_fma(dk, LN2_HI, (llpf_fma(s, hfsq + R, dk * LN2_LO) - hfsq) + f);
}

/* log1p(s) for s >= 0 (the only use: s = sum of exp-weights minus the maximum's 1,
 * reference src/utils.jl:21-26).  u = fl(1+s); log(u) + (s - (u-1))/u   [Kahan/HP correction]. */
LLPF_HD double llpf_log1p_nonneg(double s) {
    double u = 1.0 + s;
    if (u == 1.0) return s;
    if (u == LLPF_INF) return u;
    double c = s - (u - 1.0);
    return llpf_log(u) + c / u;
}

/* kernels on |x| <= pi/4 with tail y (fdlibm __kernel_sin / __kernel_cos) */
LLPF_HD double llpf_ksin(double x, double y) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double w = z * z;
    double r = llpf_fma(z * w, llpf_fma(z, S6, S5), llpf_fma(z, llpf_fma(z, S4, S3), S2));
    double v = z * x;
    return x - ((z * llpf_fma(-v, r, 0.5 * y) - y) - v * S1);
}
LLPF_HD double llpf_kcos(double x, double y) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double w = z * z;
    double r = llpf_fma(w * w, llpf_fma(z, llpf_fma(z, C6, C5), C4), z * llpf_fma(z, llpf_fma(z, C3, C2), C1));
    double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + llpf_fma(z, r, -(x * y)));
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1).  Quadrant reduction is exact (4u and 4u - rint(4u)
 * are exact), the reduced angle f*pi/2 is carried as hi + lo. */
LLPF_HD void llpf_sincos2pi(double u, double* sn, double* cs) {
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    double t = 4.0 * u;
    double qf = llpf_rint(t);
    double f = t - qf;                       /* [-0.5, 0.5], exact */
    double a = f * PIO2_HI;
    double al = llpf_fma(f, PIO2_HI, -a) + f * PIO2_LO;
    double s = llpf_ksin(a, al);
    double c = llpf_kcos(a, al);
    int q = ((int)qf) & 3;
    double so = (q & 1) ? c : s;
    double co = (q & 1) ? s : c;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    *sn = so;
    *cs = co;
}

#endif /* LLPF_DETMATH_H */
