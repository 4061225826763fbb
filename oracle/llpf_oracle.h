/* llpf_oracle.h — CPU oracle for the particle-filter hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is a plain-C restatement of the reference algorithm
 * (baggepinnen/LowLevelParticleFilters.jl v3.31.1); it is the checker for the HIP engine and the
 * `cpu_baseline` leg of bench.py.  Nothing in the product path (lowlevelparticlefilters.jl_amd/)
 * links, imports or executes it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * may.  It shares plain-data struct declarations with include/llpf.h and the primitive headers
 * csrc/shared/llpf_{detmath,philox,fixed}.h (needed for a bit-identical "device order" mode).
 *
 * PARITY PINNING.  The reference is Julia and cannot run in the build container (no julia binary,
 * SURVEY.md §8c), and it ships no golden vectors for this path, so the oracle is pinned by
 *   (i)   the deterministic known-answer assertions of the reference's own test-suite, restated in
 *         tests/test_oracle_kat.py (test/runtests.jl:29-46, 90-105, 145-154, 182-188, 274-275);
 *   (ii)  the reference's statistical bounds (resampler proportions test/runtests.jl:108-143; PF vs
 *         Kalman log-likelihood :436-450) restated in tests/test_oracle_statistical.py;
 *   (iii) agreement with the closed-form Kalman log-likelihood (reference src/filtering.jl:52-128)
 *         within Monte-Carlo error.
 * The RNG streams (Xoshiro + ziggurat randn, global rand()) are unpinned by the reference (no
 * seeded expectation exists) and are replaced by Philox4x32 (7 rounds since round 4; the 10-round instance is held to the Random123 vectors); see llpf_philox.h.
 *
 * Two arithmetic orders:
 *   ORC_ORDER_REFERENCE — literal: findmax / SLEEF-like exp (libm) / pairwise sum / serial fp64
 *                         cumsum / two-pointer search, as cited per function in llpf_oracle.c.
 *   ORC_ORDER_DEVICE    — the same algorithm with the two order-dependent reductions (sum of
 *                         exp-weights, cumulative bins) carried out in fixed point and the
 *                         transcendentals taken from llpf_detmath.h, which is what the GPU does; it
 *                         is bit-identical to the HIP engine on every output that feeds the recursion.
 */
#ifndef LLPF_ORACLE_H
#define LLPF_ORACLE_H

#include <stdint.h>
#include "../include/llpf.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_ORDER_REFERENCE = 0, ORC_ORDER_DEVICE = 1 };

typedef struct orc_filter orc_filter;

orc_filter* orc_create(const llpf_config* cfg, int order);
void   orc_destroy(orc_filter* f);
void   orc_seed(orc_filter* f, uint64_t seed);
void   orc_reset(orc_filter* f);
void   orc_reset_explicit(orc_filter* f, const double* xi /* N*nx standard normals */);
double orc_correct(orc_filter* f, const double* u, const double* y /* NULL = missing */, double t);
void   orc_predict(orc_filter* f, const double* u, double t);
void   orc_predict_explicit(orc_filter* f, const double* u, double t,
                            const double* xi /* N*nx */, const double* U /* 1 (systematic) or N (stratified) */);
double orc_update(orc_filter* f, const double* u, const double* y, double t);
double orc_run(orc_filter* f, const double* U, const double* Y, int64_t T, double t_index0,
               double* ll_steps, double* xmean, double* x_hist, double* w_hist, double* we_hist);

/* AuxiliaryParticleFilter{ParticleFilter}: correct! (logsumexp only), predict! with the look-ahead measurement y1,
 * update!, and the loops of forward_trajectory (mode 0) / loglik (mode 1) */
double orc_aux_correct(orc_filter* f);
void   orc_aux_predict(orc_filter* f, const double* u, const double* y1 /* NULL = missing */, double t);
double orc_aux_update(orc_filter* f, const double* u, const double* y1, double t);
double orc_run_aux(orc_filter* f, const double* U, const double* Y, int64_t T, int mode,
                   double* ll_steps, double* xmean, double* x_hist, double* w_hist, double* we_hist);

/* FFBS particle smoother: smooth(pf, xf, wf, wef, ll, M, u, y, p) and draw_one_categorical (0-based) */
int64_t orc_draw_one_categorical(double* w, double* bins, int64_t n, double u, int order);
int    orc_smooth(orc_filter* f, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
                  const double* wef, double* xb, int64_t* idx);

/* RBPF (model_id LLPF_MODEL_RB_LINEAR): the covariance shared by all particles, nxl x nxl row-major */
void   orc_rb_get_R(const orc_filter* f, double* R);
/* RBPF with per-particle covariance (LLPF_MODEL_RB_BILINEAR): xl [N][nxl], R [N][nxl][nxl]; either may be NULL */
void   orc_rb_get_linear_state(const orc_filter* f, double* xl, double* R);
int    orc_particle_dim(const orc_filter* f);   /* nx, or nxn + nxl for LLPF_MODEL_RB_BILINEAR: particles / history / means are [xn; xl] */

int64_t orc_num_particles(const orc_filter* f);
int64_t orc_index(const orc_filter* f);
void   orc_get_particles(const orc_filter* f, double* dst);
void   orc_get_weights(const orc_filter* f, double* dst);
void   orc_get_expweights(const orc_filter* f, double* dst);
void   orc_get_ancestors(const orc_filter* f, int64_t* dst /* 0-based */);
void   orc_get_bins(const orc_filter* f, double* dst);
void   orc_set_particles(orc_filter* f, const double* src);
void   orc_set_weights(orc_filter* f, const double* w);
void   orc_set_index(orc_filter* f, int64_t t);
double orc_filter_ess(const orc_filter* f);
int    orc_shouldresample(const orc_filter* f);
void   orc_weighted_mean(const orc_filter* f, double* xh);
/* StatsBase.quantile(v, ProbabilityWeights(w), p) restated (reference src/filtering.jl:583-595); see llpf_oracle.c */
int    orc_weighted_quantile(const double* v, const double* w, int64_t n, const double* p, int np, double* out);
int    orc_weighted_quantile_dev(const double* v, const double* w, int64_t n, const double* p, int np, double* out);   /* device order: exact integer crossing */
int    orc_filter_weighted_quantile(const orc_filter* f, const double* p, int np, double* out /* [np][particle_dim] */);
int    orc_last_resampled(const orc_filter* f);
double orc_maxw(const orc_filter* f);
int64_t orc_resample_count(const orc_filter* f);
int    orc_degenerate(const orc_filter* f);
int64_t orc_exact_steps(const orc_filter* f);   /* device order: weightings normalised in exact-max form (bound test failed) */

/* OpenMP threads for the per-particle loops (default 1; results are independent of the count) */
void   orc_set_threads(int n);
/* a measurement likelihood other than the Gaussian descriptor: kind 0 none, 1 Laplace (par: b), 2 Student-t (par: nu, sigma, c1) */
int    orc_set_user_loglik(orc_filter* f, int kind, const double* par, int npar);
int    orc_set_user_noise(orc_filter* f, int kind, const double* par, int npar);     /* 1: multiplicative Gaussian (s0, s1); 2: Laplace (b) */
int    orc_set_user_initial(orc_filter* f, int kind, const double* par, int npar);   /* 1: uniform box (lo[nx], hi[nx]) */
int    orc_get_threads(void);

/* array primitives */
double orc_logsumexp(double* w, double* we, int64_t n, int order, double* maxw);
void   orc_expnormalize(double* we, double* w, int64_t n);
void   orc_expnormalize_inplace(double* w, int64_t n);
double orc_effective_particles(const double* we, int64_t n);
/* j is in/out (0-based; entries never reached keep their input value, as in the reference) */
int    orc_resample(int strategy, const double* we, int64_t n, int64_t m, const double* U,
                    int64_t* j, double* bins, int order);
void   orc_resample_uniforms(int strategy, int64_t m, uint64_t seed, uint32_t step, double* u);
double orc_gauss_logpdf(const llpf_gaussian* g, const double* x);
void   orc_gauss_sample(const llpf_gaussian* g, const double* xi, double* out);
void   orc_dynamics(const llpf_model* m, const double* x, const double* u, double t, double* out);
void   orc_measurement(const llpf_model* m, const double* x, const double* u, double t, double* out);
void   orc_rk4_scalar_decay(double x0, double Ts, int supersample, double* out);
double orc_kalman_loglik(const llpf_model* m, const double* U, const double* Y, int64_t T);
double orc_pairwise_sum(const double* a, int64_t n);

/* shared-primitive probes (host evaluation of csrc/shared headers) */
void   orc_math_vec(int which, const double* in, double* out, int64_t n);
void   orc_philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4);
int    orc_philox_block_engine(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4);
void   orc_normals(uint64_t seed, uint32_t step, uint32_t stream, int nd, double* out, int64_t n);
void   orc_uniforms_nd(uint64_t seed, uint32_t step, uint32_t stream, int nd, double* out, int64_t n);   /* llpf_uniforms of particles 0..n-1 */
void   orc_fix96(double e, uint64_t* lo_hi);
uint64_t orc_q64(double e, int K);
void   orc_fix96_unit(double e, uint64_t* lo_hi);
uint64_t orc_q64_unit(double e, int K);
double orc_u128_to_double(uint64_t lo, uint64_t hi);

#ifdef __cplusplus
}
#endif
#endif
