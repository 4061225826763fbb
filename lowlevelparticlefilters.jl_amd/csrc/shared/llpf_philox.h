/* llpf_philox.h — Philox4x32 counter-based RNG (7 rounds for the engine's draws, see LLPF_PHILOX_ROUNDS) + fp64 uniform / normal transforms,
 * shared bit-for-bit by the HIP kernels and the device-order oracle.
 *
 * The reference draws process noise sequentially from a per-filter Xoshiro via ziggurat
 * randn (reference src/PFtypes.jl:30,135) and the resampling offset from the *global* RNG
 * (reference src/resample.jl:23,49); no reference test pins either stream (SURVEY.md §8c:
 * "parity unpinned"), and a sequential stream cannot be consumed by 10^6 particles in
 * parallel.  This engine therefore defines its own stream: Philox4x32 (Salmon et al.,
 * SC'11); its 10-round instance is pinned by the published Random123 known-answer vectors in tests/test_detmath.py.
 *
 *   key     = (seed_lo, seed_hi)
 *   counter = (particle index, step counter, sub-block, stream id)
 */
#ifndef LLPF_PHILOX_H
#define LLPF_PHILOX_H

#include "llpf_detmath.h"
#include "llpf_rngmath.h"

enum {
    LLPF_STREAM_INIT     = 0,   /* reset!: x0 ~ d0                      (filtering.jl:4-14)   */
    LLPF_STREAM_DYNAMICS = 1,   /* process noise in propagate_particles! (PFtypes.jl:122-139) */
    LLPF_STREAM_RESAMPLE = 2,   /* systematic offset rand()              (resample.jl:23)     */
    LLPF_STREAM_STRATIFY = 3,   /* stratified per-stratum rand()         (resample.jl:49)     */
    LLPF_STREAM_MEASURE  = 4,   /* host-side simulate() measurement noise                     */
    LLPF_STREAM_SMOOTH   = 5,   /* backward simulation: rand() of draw_one_categorical (resample.jl:137) */
    LLPF_STREAM_SMOOTH_INIT = 6, /* backward simulation: rand() of the time-T resample (smoothing.jl:123) */
    LLPF_STREAM_USER     = 7,   /* uniforms handed to a model's own process noise (UserModel::noise; PFtypes.jl:135, :254) */
    LLPF_STREAM_USER_INIT = 8   /* uniforms handed to a model's own initial density (UserModel::initial; filtering.jl:8) */
};

typedef struct { uint32_t v[4]; } llpf_philox4;

/* Rounds of the generator every draw of the engine (and of the oracle: one header) goes through.  Round 4 made this an explicit
 * choice: Philox4x32-7.  Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11) report Philox4x32 Crush-
 * resistant (SmallCrush, Crush, BigCrush of TestU01) from 7 rounds on; 10 is their default with a safety margin.  The filters' kernels
 * are bound by instruction issue and the generator is a third of the normal pair's cost: on one MI355X, 10 -> 7 rounds took the
 * quad-tank timestep (BASELINE C3, 8 normals = 2 blocks per particle) from 32.4 to 30.9 us (-4.7 %), C5 -1 %, C2 -0.4 %
 * (profiles/r04_philox_ab.txt).  The round function and key schedule are the published ones — the 10-round instance llpf_philox4x32_10
 * below is held to the Random123 known-answer vectors (tests/test_detmath.py), and the engine's generator is the same loop stopped after
 * LLPF_PHILOX_ROUNDS rounds.  Build with -DLLPF_PHILOX_ROUNDS=10 (engine AND oracle) for the conservative variant: parity is unaffected. */
#ifndef LLPF_PHILOX_ROUNDS
#define LLPF_PHILOX_ROUNDS 7
#endif

LLPF_HD llpf_philox4 llpf_philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                       uint32_t k0, uint32_t k1, const int rounds) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    llpf_philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
/* Philox4x32-10 as published (the Random123 known-answer vectors pin this one, tests/test_detmath.py) */
LLPF_HD llpf_philox4 llpf_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return llpf_philox4x32_r(c0, c1, c2, c3, k0, k1, 10);
}
/* the generator of the engine's draws */
LLPF_HD llpf_philox4 llpf_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return llpf_philox4x32_r(c0, c1, c2, c3, k0, k1, LLPF_PHILOX_ROUNDS);
}

/* 53-bit uniforms: u_open in (0,1]  (safe for log),  u_half in [0,1) */
LLPF_HD double llpf_u01_open(uint32_t lo, uint32_t hi) {
    uint64_t a = ((uint64_t)hi << 32) | lo;
    return (double)((a >> 11) + 1) * 1.1102230246251565e-16;   /* 2^-53 */
}
LLPF_HD double llpf_u01_half(uint32_t lo, uint32_t hi) {
    uint64_t a = ((uint64_t)hi << 32) | lo;
    return (double)(a >> 11) * 1.1102230246251565e-16;
}

/* sqrt of the Box-Muller radius' argument a = -2 log u, u in (0, 1]: a is 0 (u == 1) or in [2.2e-16, 1.5e3] — the core of the
 * correctly rounded square root without the expansion's scaling of tiny arguments and its 0 / inf / NaN cases: the same bits as llpf_sqrt */
LLPF_HD double llpf_sqrt_rad(double a) { return a == 0.0 ? a : llpf_sqrt_pos(a); }     /* sqrt(-0.0) == -0.0 */
/* one Philox block -> two independent N(0,1) draws (Box–Muller on deterministic log/sqrt/sincos) */
LLPF_HD void llpf_normal_pair(uint32_t idx, uint32_t step, uint32_t sub, uint32_t stream,
                              uint32_t k0, uint32_t k1, double* z0, double* z1) {
    llpf_philox4 r = llpf_philox4x32(idx, step, sub, stream, k0, k1);
    double u1 = llpf_u01_open(r.v[0], r.v[1]);
    double u2 = llpf_u01_half(r.v[2], r.v[3]);
    double rad = llpf_sqrt_rad(-2.0 * llpf_log_unit(u1));
    double sn, cs;
    llpf_sincos2pi_fast(u2, &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}

/* the same with the four tables read from a caller's copy (lg[2 i] = 1/c_i, lg[2 i + 1] = ln c_i; sc[2 j] = sin, sc[2 j + 1] = cos):
 * identical values, so identical results */
LLPF_HD void llpf_normal_pair_tab(uint32_t idx, uint32_t step, uint32_t sub, uint32_t stream, uint32_t k0, uint32_t k1,
                                  const double* lg, const double* sc, double* z0, double* z1) {
    llpf_philox4 r = llpf_philox4x32(idx, step, sub, stream, k0, k1);
    double u1 = llpf_u01_open(r.v[0], r.v[1]);
    double u2 = llpf_u01_half(r.v[2], r.v[3]);
    double m, dk, f;
    const int i = llpf_log_unit_split(u1, &m, &dk);
    const int j = llpf_sincos2pi_split(u2, &f);
    double rad = llpf_sqrt_rad(-2.0 * llpf_log_unit_eval(m, dk, lg[2 * i], lg[2 * i + 1]));
    double sn, cs;
    llpf_sincos2pi_eval(f, sc[2 * j], sc[2 * j + 1], &sn, &cs);
    *z0 = rad * cs;
    *z1 = rad * sn;
}
LLPF_HD void llpf_normals_tab(uint32_t idx, uint32_t step, uint32_t stream, uint32_t k0, uint32_t k1,
                              int nd, double* xi, const double* lg, const double* sc) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; 2 * b < nd; ++b) {
        double z0, z1;
        llpf_normal_pair_tab(idx, step, (uint32_t)b, stream, k0, k1, lg, sc, &z0, &z1);
        xi[2 * b] = z0;
        if (2 * b + 1 < nd) xi[2 * b + 1] = z1;
    }
}

/* nd standard normals for particle idx at a step: dims (2b, 2b+1) come from sub-block b */
LLPF_HD void llpf_normals(uint32_t idx, uint32_t step, uint32_t stream, uint32_t k0, uint32_t k1,
                          int nd, double* xi) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; 2 * b < nd; ++b) {
        double z0, z1;
        llpf_normal_pair(idx, step, (uint32_t)b, stream, k0, k1, &z0, &z1);
        xi[2 * b] = z0;
        if (2 * b + 1 < nd) xi[2 * b + 1] = z1;
    }
}

/* nd uniforms in [0, 1) for particle idx at a step: dims (2b, 2b+1) come from sub-block b (a model's own noise / initial density) */
LLPF_HD void llpf_uniforms(uint32_t idx, uint32_t step, uint32_t stream, uint32_t k0, uint32_t k1, int nd, double* uu) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; 2 * b < nd; ++b) {
        llpf_philox4 r = llpf_philox4x32(idx, step, (uint32_t)b, stream, k0, k1);
        uu[2 * b] = llpf_u01_half(r.v[0], r.v[1]);
        if (2 * b + 1 < nd) uu[2 * b + 1] = llpf_u01_half(r.v[2], r.v[3]);
    }
}

/* the single uniform a systematic resample consumes at a step */
LLPF_HD double llpf_uniform_step(uint32_t step, uint32_t stream, uint32_t k0, uint32_t k1) {
    llpf_philox4 r = llpf_philox4x32(0u, step, 0u, stream, k0, k1);
    return llpf_u01_half(r.v[0], r.v[1]);
}
/* per-stratum uniform for stratified resampling (stratum index i0 is 0-based) */
LLPF_HD double llpf_uniform_idx(uint32_t i0, uint32_t step, uint32_t stream, uint32_t k0, uint32_t k1) {
    llpf_philox4 r = llpf_philox4x32(i0, step, 0u, stream, k0, k1);
    return llpf_u01_half(r.v[0], r.v[1]);
}

#endif /* LLPF_PHILOX_H */
