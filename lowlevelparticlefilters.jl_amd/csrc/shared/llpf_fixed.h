/* llpf_fixed.h — order-independent accumulation of exp-weights, shared by host and device.
 *
 * The reference normalises weights with fp64 sums whose rounding depends on summation order
 * (pairwise sum, reference src/utils.jl:66-71; strictly serial cumsum, reference
 * src/resample.jl:19-22).  A GPU cannot reproduce a serial rounding sequence, and a parallel
 * fp64 tree would make results depend on launch geometry.  This engine instead converts each
 * exp-weight e = exp(w - max w) in [0,1] to fixed point and sums integers, which is
 * associative: any blocking on any machine gives the same bits.
 *
 *   fix96(e)  = floor(e * 2^96)  as a 128-bit integer  -> sum_i e_i   (logsumexp, ESS)
 *   q64(e, K) = floor(e * 2^K)   as a  64-bit integer  -> cumulative bins for resampling,
 *               K = 62 - ceil(log2 N) so that the total fits 63 bits
 *
 * fix96 keeps every bit of any e >= 2^-44 and truncates below 2^-96; the resulting sums are
 * *more* accurate than any fp64 summation order.
 */
#ifndef LLPF_FIXED_H
#define LLPF_FIXED_H

#include "llpf_detmath.h"

typedef struct { uint64_t lo, hi; } llpf_u128;

LLPF_HD llpf_u128 llpf_u128_add(llpf_u128 a, llpf_u128 b) {
    llpf_u128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}

/* floor(e * 2^96) for 0 <= e < 2^31 (e is an exp-weight <= 1 or its square) */
LLPF_HD llpf_u128 llpf_fix96(double e) {
    llpf_u128 r; r.lo = 0; r.hi = 0;
    uint64_t u = llpf_d2u(e);
    int E = (int)(u >> 52) & 0x7ff;
    if (E == 0 || E == 0x7ff || (u >> 63)) return r; /* zero / subnormal / negative / inf / NaN -> 0 */
    uint64_t M = (u & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
    int sh = E - 979;                                /* e*2^96 = M * 2^(E-1075+96) */
    if (sh >= 0) {
        if (sh >= 64) { r.hi = M << (sh - 64); }
        else { r.lo = M << sh; r.hi = sh ? (M >> (64 - sh)) : 0; }
    } else {
        int rs = -sh;
        r.lo = rs >= 53 ? 0 : (M >> rs);
    }
    return r;
}

/* floor(e * 2^K) for 0 <= e <= 1, 0 <= K <= 62 */
LLPF_HD uint64_t llpf_q64(double e, int K) {
    uint64_t u = llpf_d2u(e);
    int E = (int)(u >> 52) & 0x7ff;
    if (E == 0 || E == 0x7ff || (u >> 63)) return 0;
    uint64_t M = (u & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
    int sh = E - 1075 + K;
    if (sh >= 0) return M << sh;                     /* e <= 1, K <= 62  =>  sh <= 10 */
    int rs = -sh;
    return rs >= 53 ? 0 : (M >> rs);
}

/* Branch-free forms for arguments in [0, 1] (exp-weights and their squares), used in the hot loops.
 * With Y = mantissa << 11 (bit 63 set):  e * 2^96 = (Y * 2^64) >> (1054 - E)  and  e * 2^K = Y >> (1086 - K - E).
 * Exact for every e in [0, 2); anything else — negative, >= 2, inf, NaN (and subnormals) — maps to 0. */
LLPF_HD llpf_u128 llpf_fix96_unit(double e) {
    const uint64_t u = llpf_d2u(e);
    const int E = (int)(u >> 52);                      /* sign bit included: negative => E >= 2048 */
    const uint64_t Y = ((u << 11) | 0x8000000000000000ULL);
    const int rs = 1054 - E;                           /* >= 31 for e <= 1 */
    const int ok = (E >= 1) & (E <= 1023);
    llpf_u128 r;
    const uint64_t hi = rs < 64 ? (Y >> (rs & 63)) : 0;
    const uint64_t lo = rs < 64 ? (Y << ((64 - rs) & 63)) : (rs < 128 ? (Y >> ((rs - 64) & 63)) : 0);
    r.hi = ok ? hi : 0;
    r.lo = ok ? lo : 0;
    return r;
}
LLPF_HD uint64_t llpf_q64_unit(double e, int K) {
    const uint64_t u = llpf_d2u(e);
    const int E = (int)(u >> 52);
    const uint64_t Y = ((u << 11) | 0x8000000000000000ULL);
    const int rs = 1086 - K - E;                       /* >= 1 for e <= 1, K <= 62 */
    const int ok = (E >= 1) & (E <= 1023) & (rs < 64);
    return ok ? (Y >> (rs & 63)) : 0;
}

/* 64 x 64 -> 128 bit product */
LLPF_HD llpf_u128 llpf_mul64(uint64_t a, uint64_t b) {
    llpf_u128 r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.lo = a * b;
    r.hi = __umul64hi(a, b);
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    r.lo = (uint64_t)p;
    r.hi = (uint64_t)(p >> 64);
#endif
    return r;
}
LLPF_HD int llpf_u128_lt(llpf_u128 a, llpf_u128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
LLPF_HD llpf_u128 llpf_u128_sub(llpf_u128 a, llpf_u128 b) {
    llpf_u128 r;
    r.lo = a.lo - b.lo;
    r.hi = a.hi - b.hi - (a.lo < b.lo ? 1u : 0u);
    return r;
}
/* c = floor(q * m / Q) and rem = q * m - c * Q, exactly, for Q >= 1 and c < 2^52 (residual resampling: q a quantum,
 * m the number of outputs, Q the total of the quanta).  A double estimate is corrected with 128-bit products. */
LLPF_HD uint64_t llpf_muldiv_floor(uint64_t q, uint64_t m, uint64_t Q, uint64_t* rem) {
    const llpf_u128 P = llpf_mul64(q, m);
    uint64_t c = (uint64_t)(((double)q * (double)m) / (double)Q);
    llpf_u128 CQ = llpf_mul64(c, Q);
    while (llpf_u128_lt(P, CQ)) { --c; CQ = llpf_mul64(c, Q); }
    llpf_u128 d = llpf_u128_sub(P, CQ);
    while (d.hi != 0 || d.lo >= Q) {
        ++c;
        const llpf_u128 Qv = {Q, 0};
        d = llpf_u128_sub(d, Qv);
    }
    *rem = d.lo;
    return c;
}

/* number of fraction bits used for the 64-bit resampling bins of an N-particle filter */
LLPF_HD int llpf_qbits(int64_t n) {
    int lg = 0;
    while (((int64_t)1 << lg) < n) ++lg;
    return 62 - lg;
}

LLPF_HD int llpf_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

/* round-to-nearest-even conversion of a 128-bit unsigned integer to double */
LLPF_HD double llpf_u128_to_double(llpf_u128 a) {
    if (a.hi == 0) return (double)a.lo;
    int z = llpf_clz64(a.hi);
    uint64_t top = z ? ((a.hi << z) | (a.lo >> (64 - z))) : a.hi;
    uint64_t rest = z ? (a.lo << z) : a.lo;
    if (rest) top |= 1;                              /* sticky bit (11 guard bits below the mantissa) */
    /* value = top * 2^(64 - z) */
    return (double)top * llpf_pow2i(64 - z);
}

/* value of a fix96 accumulator: acc * 2^-96 */
LLPF_HD double llpf_fix96_to_double(llpf_u128 a) {
    return llpf_u128_to_double(a) * llpf_pow2i(-96);
}

/* a - 2^96 (caller guarantees a >= 2^96): removes the maximum's exp(0) = 1 exactly,
 * the integer analogue of sum_all_but (reference src/utils.jl:66-71) */
LLPF_HD llpf_u128 llpf_fix96_minus_one(llpf_u128 a) {
    llpf_u128 r = a;
    r.hi -= ((uint64_t)1 << 32);
    return r;
}

#endif /* LLPF_FIXED_H */
