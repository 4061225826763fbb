/* llpf.h — C ABI of libllpf_hip.so, the MI355X (gfx950) engine behind the particle-filter hot path
 * of LowLevelParticleFilters.jl (predict! / correct! / update! / resample / logsumexp! /
 * forward_trajectory / loglik).
 *
 * The reference has no FFI: the "plugin interface" of this path is Julia multiple dispatch on
 * AbstractParticleFilter (reference src/PFtypes.jl:2).  Each entry point below names the
 * reference method it replaces (file:line under the reference root); INTEGRATION.md shows the
 * `ccall` a maintainer adds on the Julia side, and lowlevelparticlefilters.jl_amd/_capi.py
 * is the identical binding in ctypes (the one the tests drive, since Julia is not in this image).
 *
 * Conventions
 *   - every function returns an int status (LLPF_OK == 0); no C++ exception crosses the ABI
 *     (every export is a function-try-block: std::bad_alloc -> LLPF_ERR_ALLOC, anything else ->
 *     LLPF_ERR_INTERNAL; host threads of a multi-GPU bank catch inside the thread);
 *     llpf_last_error() returns a thread-local message for the last non-zero status.
 *   - all host pointers are borrowed for the duration of the call only.
 *   - the library owns all device memory behind the opaque handle; one handle = one device +
 *     one HIP stream; a handle is not re-entrant, distinct handles may be driven from
 *     distinct host threads.
 *   - particles cross the boundary in the reference's layout: N contiguous nx-vectors
 *     (Vector{SVector{nx,Float64}}, reference src/PFtypes.jl:9-10); on the device they live
 *     structure-of-arrays.
 *   - ancestor indices are 0-based int64 at this boundary (the Julia wrapper adds 1).
 *   - user callables (dynamics / measurement / measurement_likelihood, reference
 *     src/PFtypes.jl:59-63,189-193) cannot run on the GPU; they are replaced by a model
 *     descriptor: a built-in model id plus its parameters.
 */
#ifndef LLPF_H
#define LLPF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLPF_VERSION_MAJOR 0
#define LLPF_VERSION_MINOR 6
#define LLPF_MAX_DIM 16       /* states and outputs of a model (round 5: 8 -> 16; above 4 the linear-Gaussian model and every user model are compiled at run time) */
#define LLPF_MAX_INPUTS 8     /* inputs u */
#define LLPF_RB_MAX_LINEAR 8  /* linear states of LLPF_MODEL_RB_BILINEAR */

/* status codes */
enum {
    LLPF_OK             = 0,
    LLPF_ERR_ARG        = 1,   /* invalid argument / unsupported configuration           */
    LLPF_ERR_HIP        = 2,   /* HIP runtime error (message has the hipError string)    */
    LLPF_ERR_NO_DEVICE  = 3,   /* no gfx950 device visible: the engine has NO CPU fallback */
    LLPF_ERR_DEGENERATE = 4,   /* all weights -Inf or NaN (the reference would return NaN) */
    LLPF_ERR_ALLOC      = 5,   /* host or device memory could not be had (std::bad_alloc stops here) */
    LLPF_ERR_INTERNAL   = 6    /* any other C++ exception of the host code, caught at the boundary; message = what() */
};

/* covariance storage kinds — mirror PDMats ScalMat / PDiagMat / PDMat, whose quadratic
 * forms differ in operation order (reference src/utils.jl:110-113) */
enum { LLPF_COV_SCAL = 0, LLPF_COV_DIAG = 1, LLPF_COV_FULL = 2 };

/* Gaussian density N(mu, Sigma): replaces Distributions.MvNormal / SimpleMvNormal
 * (reference src/utils.jl:241-270, ext/LowLevelParticleFiltersDistributionsExt.jl:16,80) */
typedef struct llpf_gaussian {
    int32_t dim;
    int32_t kind;                          /* LLPF_COV_*                                        */
    double  mu[LLPF_MAX_DIM];
    double  cov[LLPF_MAX_DIM * LLPF_MAX_DIM]; /* SCAL: cov[0]=sigma^2; DIAG: cov[0..dim); FULL: row-major dim x dim */
} llpf_gaussian;

/* built-in models */
enum {
    LLPF_MODEL_LINEAR_GAUSSIAN = 0,  /* f = A x + B u, g = C x  (reference examples/example_lineargaussian.jl:28-29); nx, ny in 1..16, nu in 0..8:
                                      * precompiled up to 4 (one fused launch per timestep), compiled on demand through hiprtc above (two launches) */
    LLPF_MODEL_QUADTANK_RK4    = 1,  /* quad-tank, RK4          (reference examples/example_quadtank.jl:8-35, src/utils.jl:220-237) */
    /* Rao-Blackwellized particle filter with constant matrices (reference src/rbpf.jl:63-283, "model 2" of :92-98):
     *   xn' = Fn xn + Bn u + An xl + wn,  wn ~ dynamics_density (R1n)     xl' = Al xl + Bl u + wl,  wl ~ linear_noise (R1l)
     *   y   = Gn xn + Cl xl + e,          e  ~ measurement_density (R2)
     * state x = [xn; xl] (nx = nxn + nxl <= 4); A = [Fn An; 0 Al], B = [Bn; Bl], C = [Gn Cl];
     * initial_density = d0n (dimension nxn), linear_initial = d0l (the inner KalmanFilter's d0). */
    LLPF_MODEL_RB_LINEAR       = 2,
    /* Rao-Blackwellized particle filter whose coupling matrix depends on the nonlinear state (reference src/rbpf.jl:163-283
     * with `An` a function of x: the "singleR" shortcut :176/:247 is off, every particle carries its own covariance R and
     * runs its own Riccati recursion — BASELINE config C5):
     *   xn' = f_n(xn, u) + An(xn) xl + wn,   An(xn) = An[0] + sum_k xn[k] An[1+k]      (llpf_rb_coupling below)
     *   xl' = Al xl + Bl u + wl,             y = g(xn) + Cl xl + e
     * llpf_model.nx = nxn <= 4 (what f_n, g and the densities df, d0 see); f_n / g are the linear-Gaussian descriptors A, B, C of this
     * struct sized for nxn (rb.fn_kind 0) or the quad-tank RK4 dynamics / measurement (rb.fn_kind 1, nxn = 4, ny = 2).
     * nxl = rb.nxl <= 8, ny <= 4 (round 5; 2 before); dynamics_density = R1n (must be Gaussian), linear_noise = R1l, linear_initial = d0l.
     * Shapes (nxn, nxl, ny) = (1,2,1), (2,2,2), (4,8,2) are precompiled, every other one is compiled through hiprtc when the filter is
     * built.  Banks and multi-GPU sweeps of such filters are provided (without weighted means); the auxiliary filter and the smoother are not. */
    LLPF_MODEL_RB_BILINEAR     = 3,
    /* ids >= LLPF_MODEL_USER_BASE: models compiled at run time from device source, llpf_model_compile() below */
    LLPF_MODEL_USER_BASE       = 1000
};

/* linear substate and state-dependent coupling of LLPF_MODEL_RB_BILINEAR (ignored by the other models) */
typedef struct llpf_rb_coupling {
    int32_t nxl;                              /* number of linear states (1..8) */
    int32_t fn_kind;                          /* 0: f_n = A xn + B u, g = C xn;  1: quad-tank RK4 f_n, g (qt, supersample) */
    double  Al[LLPF_RB_MAX_LINEAR * LLPF_RB_MAX_LINEAR];  /* nxl x nxl row-major: kf.A */
    double  Bl[LLPF_RB_MAX_LINEAR * LLPF_MAX_INPUTS];     /* nxl x nu  row-major: kf.B */
    double  Cl[LLPF_RB_MAX_LINEAR * LLPF_RB_MAX_LINEAR];  /* ny  x nxl row-major: kf.C (must not be zero) */
    double  An[5][32];                        /* An[0]: constant term; An[1+k]: multiplies xn[k]; each nxn x nxl row-major */
} llpf_rb_coupling;

/* quadtank constant slots in llpf_model.qt[] */
enum { LLPF_QT_K1 = 0, LLPF_QT_K2, LLPF_QT_G, LLPF_QT_A1, LLPF_QT_A2, LLPF_QT_A3, LLPF_QT_A4,
       LLPF_QT_a1, LLPF_QT_a2, LLPF_QT_a3, LLPF_QT_a4, LLPF_QT_GAMMA1, LLPF_QT_GAMMA2,
       LLPF_QT_TSWITCH, LLPF_QT_A1FACTOR, LLPF_QT_EPS, LLPF_QT_COUNT };

/* replaces the callables + densities stored in ParticleFilter / AdvancedParticleFilter
 * (reference src/PFtypes.jl:21-36, 162-177) */
typedef struct llpf_model {
    int32_t model_id;
    int32_t nx, nu, ny;
    double  A[LLPF_MAX_DIM * LLPF_MAX_DIM];   /* nx x nx row-major (linear-Gaussian) */
    double  B[LLPF_MAX_DIM * LLPF_MAX_INPUTS]; /* nx x nu row-major                  */
    double  C[LLPF_MAX_DIM * LLPF_MAX_DIM];   /* ny x nx row-major                   */
    double  qt[LLPF_QT_COUNT];                /* quad-tank constants                 */
    int32_t supersample;                      /* rk4 supersample (reference src/utils.jl:220) */
    int32_t nxn;                              /* LLPF_MODEL_RB_LINEAR: number of nonlinear states (else 0) */
    double  Ts;                               /* sample time (reference src/PFtypes.jl:33) */
    llpf_gaussian dynamics_density;           /* df  (RB: R1n, dimension nxn) */
    llpf_gaussian measurement_density;        /* dg  (RB: R2) */
    llpf_gaussian initial_density;            /* d0  (RB: d0n, dimension nxn) */
    llpf_gaussian linear_noise;               /* RB only: N(0, R1l), dimension nx - nxn (kf.R1) */
    llpf_gaussian linear_initial;             /* RB only: d0l, dimension nx - nxn (kf.d0)       */
    llpf_rb_coupling rb;                      /* LLPF_MODEL_RB_BILINEAR only */
} llpf_model;

enum { LLPF_RESAMPLE_SYSTEMATIC = 0, LLPF_RESAMPLE_STRATIFIED = 1, LLPF_RESAMPLE_RESIDUAL = 2 };   /* reference src/LowLevelParticleFilters.jl:43-46 */
enum { LLPF_PARTICLE_FILTER = 0, LLPF_ADVANCED_PARTICLE_FILTER = 1 };

typedef struct llpf_config {
    uint32_t struct_size;            /* sizeof(llpf_config), ABI guard                     */
    int32_t  filter_kind;            /* LLPF_PARTICLE_FILTER / LLPF_ADVANCED_PARTICLE_FILTER */
    int64_t  n_particles;            /* N                                                   */
    int32_t  resampling_strategy;    /* LLPF_RESAMPLE_*                                     */
    int32_t  device;                 /* HIP device ordinal                                  */
    double   resample_threshold;     /* reference default 0.1 (PF) / 0.5 (APF)              */
    uint64_t seed;                   /* Philox key                                          */
    llpf_model model;
} llpf_config;

typedef struct llpf_filter llpf_filter;   /* opaque: PFstate + filter (reference src/PFtypes.jl:8-36) */
typedef struct llpf_bank   llpf_bank;     /* opaque: many independent filters on one device   */

/* ---- lifetime --------------------------------------------------------------------------- */
/* ParticleFilter(N, dynamics, measurement, df, dg, d0; ...) — reference src/PFtypes.jl:65-75,200-210.
 * Particles are initialised from d0 (as the reference constructor does), w = log(1/N), t = 0. */
int  llpf_create(const llpf_config* cfg, llpf_filter** out);
int  llpf_destroy(llpf_filter* f);
/* reset!(pf) — reference src/filtering.jl:4-14: x ~ d0, w = -log N, we = 1/N, t = 1 */
int  llpf_reset(llpf_filter* f);
/* re-key the RNG and zero its step counters (no reference equivalent: the reference never seeds pf.rng) */
int  llpf_seed(llpf_filter* f, uint64_t seed);
/* new parameters for this handle (same model id and dimensions): see llpf_bank_set_models */
int  llpf_set_model(llpf_filter* f, const llpf_model* model);

/* ---- the step --------------------------------------------------------------------------- */
/* correct!(pf,u,y,p,t) -> (ll, 0) — reference src/filtering.jl:164-168 (measurement_equation!
 * src/PFtypes.jl:107-120,226-239 + logsumexp! src/utils.jl:18-27).  y == NULL means "missing"
 * (weights untouched, logsumexp! still runs — reference src/PFtypes.jl:109). */
int  llpf_correct(llpf_filter* f, const double* u, const double* y, double t, double* ll);
/* predict!(pf,u,p,t) — reference src/filtering.jl:140-153 (shouldresample src/resample.jl:5-10,
 * resample :12-61, propagate_particles! src/PFtypes.jl:122-139 / DistributionsExt:83-93,
 * reset_weights! src/utils.jl:73-79) */
int  llpf_predict(llpf_filter* f, const double* u, double t);
/* update!(pf,u,y,p,t) = correct! then predict! — reference src/filtering.jl:181-185, functor :238,240 */
int  llpf_update(llpf_filter* f, const double* u, const double* y, double t, double* ll);

/* history / output selection for llpf_run */
typedef struct llpf_run_outputs {
    double* ll_steps;    /* [T] per-step log-likelihood, or NULL                               */
    double* xmean;       /* [T*nx] weighted_mean after each correct! (reference src/filtering.jl:541-549), or NULL */
    double* x_hist;      /* [T*N*nx] particles(pf) at each step, time-major (column t of the reference's N x T Matrix), or NULL */
    double* w_hist;      /* [T*N] weights(pf)    (reference src/filtering.jl:358), or NULL */
    double* we_hist;     /* [T*N] expweights(pf) (reference src/filtering.jl:359), or NULL */
    double* xcov;        /* [T*nx*nx] weighted_cov after each correct! (reference src/filtering.jl:571-581: StatsBase's corrected covariance under
                          * probability weights), computed on the device from the state the history outputs would copy out, or NULL.  llpf_run only
                          * (single filters that are not LLPF_MODEL_RB_BILINEAR); asks for the balanced two-launch timestep like the history outputs. */
    double* xquant;      /* [T*nx*nq] weighted_quantile(sol, quant_p) (reference src/filtering.jl:583-595: [t][state][q]) of the same state, computed on
                          * the device per timestep (radix selection over the exp-weights, csrc/k_quantile.hip), or NULL.  llpf_run only; same
                          * restrictions as xcov.  (ABI minor 6: the struct grew by these three fields) */
    const double* quant_p; /* [nq] probabilities in [0, 1] */
    int32_t nq;          /* 1..1024 */
    int32_t pad;
} llpf_run_outputs;

/* T iterations of {correct!(u_k,y_k,t_k); predict!(u_k,t_k)} with t_k = (t_index0 + k) * Ts, k = 0..T-1,
 * enqueued on the device without a host round trip per step.
 *   forward_trajectory(pf,u,y,p) — reference src/filtering.jl:343-365: llpf_reset, then t_index0 = 0
 *   loglik(pf,u,y,p)             — reference src/smoothing.jl:227-230: llpf_reset, then t_index0 = 1
 * U is T x nu row-major, Y is T x ny row-major; a row of Y whose first element is NaN is "missing".
 * *ll_total receives the sum of the per-step log-likelihoods. */
int  llpf_run(llpf_filter* f, const double* U, const double* Y, int64_t T, double t_index0,
              double* ll_total, const llpf_run_outputs* outs);

/* ---- AuxiliaryParticleFilter{ParticleFilter} (reference src/PFtypes.jl:38-49) ------------------------------
 * The same handle (filter_kind LLPF_PARTICLE_FILTER) driven through the auxiliary verbs; reset!, accessors and
 * resampling strategy are those of the wrapped filter (PFtypes.jl:299 @forward).  expweights(pf) between an aux
 * predict! and the next correct! returns lambda, as the reference's `we` buffer does (filtering.jl:200-203).
 *   llpf_aux_correct  correct!(pf::AuxiliaryParticleFilter,u,y,p,t)      src/filtering.jl:170-174 (logsumexp! only:
 *                     the measurement was applied by the preceding predict!, y is not used)
 *   llpf_aux_predict  predict!(pf::AuxiliaryParticleFilter,u,y1,p,t)     src/filtering.jl:195-217 (noise-free
 *                     propagate, lambda = logpdf(dg, y1 - g(x)), expnormalize!(w + lambda), resample (always),
 *                     permute, add_noise!, w = lambda - log N); y1 NULL or NaN = missing.  Weights still waiting
 *                     for their correct! are normalised first.
 *   llpf_aux_update   update!(pf::AuxiliaryParticleFilter,u,y,y1,p,t)    src/filtering.jl:187-191
 *   llpf_aux_run      mode 0: forward_trajectory(pf::AuxiliaryParticleFilter,u,y,p)  src/filtering.jl:367-384
 *                     mode 1: loglik(pf::AuxiliaryParticleFilter,u,y,p)              src/smoothing.jl:232-236
 *                     (call llpf_reset first; t_k = k * Ts; all launches are enqueued back to back unless history
 *                     outputs are requested)
 * AuxiliaryParticleFilter{AdvancedParticleFilter} (filter_kind LLPF_ADVANCED_PARTICLE_FILTER; src/filtering.jl:219-234): the same
 * verbs; its predict! uses the look-ahead weights only to choose the ancestors, then propagates AGAIN from the previous particles
 * with noise and resets the weights (lambda is discarded, so the following correct! returns ~0, as in the reference).  Driven
 * one step per call; llpf_aux_run loops the steps synchronously. */
int  llpf_aux_correct(llpf_filter* f, double* ll);
int  llpf_aux_predict(llpf_filter* f, const double* u, const double* y1, double t);
int  llpf_aux_update(llpf_filter* f, const double* u, const double* y1, double t, double* ll);
int  llpf_aux_run(llpf_filter* f, const double* U, const double* Y, int64_t T, int32_t mode,
                  double* ll_total, const llpf_run_outputs* outs);

/* ---- Rao-Blackwellized particle filter (model_id LLPF_MODEL_RB_LINEAR) ---------------------------------------
 * RBPF(N, kf, dynamics, nl_measurement_model, R1n, d0n; An, ...) — reference src/rbpf.jl:63-144 with constant
 * matrices ("singleR", :176/:247: one covariance recursion serves all particles; it runs on the host and only its
 * gains are sent to the device).  The ordinary verbs drive it: llpf_reset = reset! (:146-160), llpf_correct = correct!
 * (:235-283: w += logpdf(N(0,S), e), xl += K e — or, when C == 0, logpdf(R2, e) and the reference's reuse of the inner
 * filter's untouched x, R), llpf_predict = predict! (:163-232), llpf_run = forward_trajectory / loglik.  Particles
 * are [xn; xl].  An != 0 needs nxn == 1 (the right division by Nt, :212, is implemented for a scalar). */
int  llpf_rb_get_covariance(llpf_filter* f, double* R /* nxl*nxl row-major: x[1].R */);
/* LLPF_MODEL_RB_BILINEAR: the per-particle Kalman state (fields xl, R of every RBParticle, reference src/rbpf.jl:1-5);
 * xl [N][nxl], R [N][nxl][nxl] row-major, either may be NULL.  For this model the particle of llpf_get_particles /
 * llpf_set_particles / llpf_weighted_mean and of the x_hist / xmean outputs of llpf_run is [xn; xl] (nxn + nxl values: an
 * RBParticle indexes like that vector, :24-30). */
int  llpf_rb_get_linear_state(llpf_filter* f, double* xl, double* R);

/* ---- particle smoother ---------------------------------------------------------------------------------------
 * xb, ll = smooth(pf, xf, wf, wef, ll, M, u, y, p) — reference src/smoothing.jl:116-143: forward-filtering backward
 * simulation.  xf [T*N*nx], wf / wef [T*N] are the history outputs of llpf_run (forward_trajectory); U is T x nu.
 * The time-T indices are j = resample(strategy, wef[:,T], M) (:123); for t = T-1 .. 1 and every trajectory m:
 * wb[n] = wf[n,t] + logpdf(df, xb[m,t+1] - f(xf[n,t],u[t],p,(t-1)Ts)), i = draw_one_categorical(wb)
 * (src/resample.jl:128-152), xb[m,t] = xf[i,t] — O(M N T) density evaluations, one block per trajectory.
 * xb is [T*M*nx] (time-major), idx (optional) [T*M] the 0-based particle index behind every sample.  M <= N. */
int  llpf_smooth(llpf_filter* f, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
                 const double* wef, double* xb, int64_t* idx);

/* ---- user-supplied models ------------------------------------------------------------------------------------------
 * The reference's filters take arbitrary callables dynamics(x,u,p,t) / measurement(x,u,p,t) (src/PFtypes.jl:59-63, 189-193,
 * 226-289).  A Julia closure cannot cross this boundary, device code can: `device_src` is HIP C++ that defines
 *     struct UserModel {
 *         static constexpr bool RB = false;
 *         DEV void prepare(const ModelD* m, const double* u, double t);   // once per thread: particle-independent terms.  The
 *                                              // parameter block p is the llpf_model of the filter as the device sees it:
 *                                              // m->A[64], m->B[64], m->C[64], m->qt[16], m->Ts, m->supersample, m->nu
 *         DEV void dynamics(const double* x, double* out) const;          // x+ = f(x, u, p, t), noise-free (nx values)
 *         DEV void measurement(const double* x, double* out) const;       // y  = g(x, u, p, t)            (ny values)
 *         // optional — a measurement likelihood of the model's own: the reference's measurement_likelihood(x,u,y,p,t) callable of an
 *         // AdvancedParticleFilter (src/PFtypes.jl:226-239), or logpdf of a measurement density that is not Gaussian
 *         // (ext/LowLevelParticleFiltersDistributionsExt.jl:80); replaces logpdf(measurement_density, y - measurement(x)):
 *         DEV double loglik(const double* x, const double* y, double t) const;   // log p(y | x) at measurement time t
 *         DEV double loglik_bound() const;     // an upper bound of loglik over x and y for the parameters prepare() saw (it is
 *                                              // evaluated once, when the filter is built, with u = 0, t = 0): what the engine's
 *                                              // bound-offset normalisation needs.  Without this member every step is normalised
 *                                              // against the true maximum instead (same results to rounding; llpf_run then
 *                                              // launches the exact-form normalisation in front of every step: about 20 %
 *                                              // slower, no host round trips).  A bound that does not hold is reported as
 *                                              // LLPF_ERR_DEGENERATE.
 *     };
 * (DEV = __device__ __forceinline__; the engine's deterministic math — llpf_exp, llpf_log, llpf_sqrt_pos, ... of
 * csrc/shared/llpf_detmath.h — is in scope, and the source is compiled with -ffp-contract=off like the engine.)  The snippet is
 * compiled with hiprtc for the visible device into the engine's own step kernel; *model_id (>= LLPF_MODEL_USER_BASE) then goes
 * into llpf_model.model_id with the same nx, ny (1..4 each: the kernels around the compiled one are precompiled for those).  Process
 * noise and initial density are the Gaussian descriptors of llpf_model unless the snippet defines `noise` / `initial` (below), and so is
 * the measurement likelihood unless the snippet defines `loglik` (measurement_density is then unused, but must still be a valid Gaussian of dimension ny).  Such filters and banks
 * run the balanced two-launch timestep; the auxiliary verbs and the smoother work (their kernels are compiled with the snippet too), the
 * Rao-Blackwellized forms are not provided for them.  Compiling the same (source, nx, ny) again returns the same id.
 * On failure the compiler log is in llpf_last_error(). */
int  llpf_model_compile(const char* device_src, int32_t nx, int32_t ny, int32_t* model_id);
/* Optional members of the snippet, beyond `loglik` / `loglik_bound` (round 4) — the random part of the step in the user's hands:
 *   DEV void noise(const double* x, const double* fx, const double* xi, const double* uu, double* out) const;
 *       the next state of a particle with previous state x and noise-free prediction fx = dynamics(x), from nx standard normals xi and nx
 *       uniforms uu in [0,1) of the particle's own Philox streams; replaces out = fx + rand(dynamics_density).  The reference's
 *       AdvancedParticleFilter hands the noise to the user the same way — dynamics(x, u, p, t, noise = true), src/PFtypes.jl:242-259,
 *       test/runtests.jl:553-599 — and its ParticleFilter draws from ANY dynamics_density (rand!(rng, d, noise), src/PFtypes.jl:122-139).
 *       (dynamics_density is then unused by the propagate but must still be a valid Gaussian; the FFBS smoother, whose backward weights
 *       are logpdf(dynamics_density, .), and the auxiliary filter over a LLPF_PARTICLE_FILTER, whose add_noise! draws from it, keep using it.)
 *   DEV void initial(const double* xi, const double* uu, double* out) const;
 *       one draw of the initial density: x_i = rand(rng, initial_density) of reset! / the constructor (src/filtering.jl:4-14, src/PFtypes.jl:66).
 * llpf_model_traits reports which optional members a compiled model has (bit set of LLPF_TRAIT_*), so that a binding can refuse a
 * UserLikelihood paired with a snippet without `loglik`, or a Gaussian likelihood paired with a snippet that defines one. */
enum { LLPF_TRAIT_LOGLIK = 1, LLPF_TRAIT_LOGLIK_BOUND = 2, LLPF_TRAIT_NOISE = 4, LLPF_TRAIT_INITIAL = 8 };
int  llpf_model_traits(int32_t model_id, int32_t* traits);

/* ---- accessors (reference src/PFtypes.jl:296-334) --------------------------------------- */
int  llpf_num_particles(const llpf_filter* f, int64_t* n);                /* num_particles(pf) */
int  llpf_index(const llpf_filter* f, int64_t* t);                        /* index(pf) = state.t[] */
int  llpf_get_particles(llpf_filter* f, double* dst /* N*nx */);          /* particles(pf)     */
int  llpf_get_weights(llpf_filter* f, double* dst /* N */);               /* weights(pf): log-weights  */
int  llpf_get_expweights(llpf_filter* f, double* dst /* N */);            /* expweights(pf)    */
int  llpf_get_ancestors(llpf_filter* f, int64_t* dst /* N, 0-based */);   /* state(pf).j       */
int  llpf_get_bins(llpf_filter* f, double* dst /* N */);                  /* state(pf).bins (as used by the last resample); LLPF_ERR_ARG for ResampleResidual */
int  llpf_set_particles(llpf_filter* f, const double* src /* N*nx */);    /* state(pf).x .= , xprev .=  */
int  llpf_set_weights(llpf_filter* f, const double* w /* N log-weights */); /* state(pf).w .= w; we .= exp.(w) */
int  llpf_set_index(llpf_filter* f, int64_t t);
int  llpf_effective_particles(llpf_filter* f, double* ess);               /* reference src/resample.jl:1-2 */
int  llpf_shouldresample(llpf_filter* f, int32_t* yes);                   /* reference src/resample.jl:5-10 */
int  llpf_weighted_mean(llpf_filter* f, double* xh /* nx */);             /* reference src/filtering.jl:541-549,568 */
int  llpf_last_resampled(llpf_filter* f, int32_t* yes);                   /* did the last predict! resample? */
int  llpf_maxw(llpf_filter* f, double* maxw);                             /* state(pf).maxw[] */

/* ---- exported array primitives (operate on caller-owned host vectors, computed on the GPU) */
/* ll = logsumexp!(w, we) — reference src/utils.jl:18-27 */
int  llpf_logsumexp(int32_t device, double* w, double* we, int64_t n, double* ll);
/* j = resample(strategy, we, M) — reference src/resample.jl:12-117.  U holds the uniform draws the
 * reference takes from the global rand(): 1 value (systematic) or m values (stratified; residual: U[i] is the
 * draw of output i, read only for the outputs after the deterministic copies) — the caller's U must hold that many.  j is
 * in/out: entries whose threshold is never met keep their input value, as in the reference. */
int  llpf_resample(int32_t device, int32_t strategy, const double* we, int64_t n, int64_t m,
                   const double* U, int64_t* j /* m, 0-based */);
/* the uniforms the filter path draws for its resample at Philox step `step` (host evaluation of the
 * shared generator, so a caller / a test can reproduce what predict! used) */
int  llpf_resample_uniforms(int32_t strategy, int64_t m, uint64_t seed, uint32_t step, double* u /* 1 or m */);

/* ---- banks of independent filters (parameter sweeps) ------------------------------------ */
/* n_filters filters that share N, model_id, dimensions, strategy and threshold but have their own
 * model parameters and RNG key (seed + filter index): the `map(svec) do s ... loglik(pfs,u,y)`
 * sweep of the reference (test/runtests.jl:412-417, src/smoothing.jl:335-347), batched into
 * single launches over (tile, filter).  models == NULL: every filter uses base->model (Monte-Carlo replicas). */
int  llpf_bank_create(const llpf_config* base, const llpf_model* models, int32_t n_filters, llpf_bank** out);
int  llpf_bank_destroy(llpf_bank* b);
int  llpf_bank_reset(llpf_bank* b);
int  llpf_bank_seed(llpf_bank* b, uint64_t seed);
/* New parameters for an existing handle — what the reference's filter_from_parameters(theta, pf) of log_likelihood_fun / metropolis
 * (src/smoothing.jl:266-283, 311-330) returns: same model id and dimensions, everything else of the n_filters descriptors may differ.
 * Nothing is allocated, captured run loops stay valid (unless Ts changes); particles, weights and random streams are untouched — loglik /
 * forward_trajectory reset! first, as in the reference.  A Metropolis iteration over a bank of chains is llpf_bank_set_models + llpf_bank_run. */
int  llpf_bank_set_models(llpf_bank* b, const llpf_model* models /* [n_filters] */);
/* as llpf_run, shared U / Y, ll_total has n_filters entries; ll_steps (optional) is [T * n_filters] */
int  llpf_bank_run(llpf_bank* b, const double* U, const double* Y, int64_t T, double t_index0,
                   double* ll_total, double* ll_steps);
/* as llpf_bank_run with inputs of its own for every filter: U [n_filters][T][nu], Y [n_filters][T][ny] (e.g. the independent
 * Monte-Carlo runs of the reference's benchmark loop, examples/example_lineargaussian.jl:282-316, as one bank); xmean
 * (optional) is [T][n_filters][nx], the weighted mean after every correct!.  Missing measurements must coincide. */
int  llpf_bank_run_multi(llpf_bank* b, const double* U, const double* Y, int64_t T, double t_index0,
                         double* ll_total, double* ll_steps, double* xmean);
/* as llpf_aux_run for every filter of the bank (the ML sweep over AuxiliaryParticleFilters, test/runtests.jl:419-423) */
int  llpf_bank_aux_run(llpf_bank* b, const double* U, const double* Y, int64_t T, int32_t mode,
                       double* ll_total, double* ll_steps);

/* ---- sweeps sharded over the GPUs of one node (multi-GPU banks) ------------------------------
 * The same sweep as llpf_bank_*, with filter k on shard k mod n_shards (one shard = one GPU, one stream): the reference's
 * one-filter-per-thread layout (src/smoothing.jl:335-347, test/runtests.jl:412-417) with GPUs for threads.  Filters never
 * interact; the only exchange of the path is the all-reduce (sum) of the per-filter log-likelihood vector after a run —
 * every shard contributes its own slots and zeros — done with RCCL over xGMI on device buffers.  Filter k's RNG key is
 * seed + k wherever it lives, and a slot has exactly one non-zero contribution, so a sharded sweep returns the bits of
 * the unsharded llpf_bank_run.
 *   llpf_mbank_create       one process drives n_devices GPUs (one host thread per GPU inside the calls); communicator from
 *                           ncclCommInitAll.  n_devices == 1: no communicator, no RCCL call (llpf_bank_* behind another
 *                           handle).  A device listed twice: shards share that GPU and the vectors are summed on the host
 *                           (RCCL refuses two ranks on one device) — for testing the sharding on a one-GPU box.
 *   llpf_mbank_unique_id /  one process per GPU (torchrun, MPI, Julia Distributed): rank 0 obtains an id (ncclGetUniqueId),
 *   llpf_mbank_create_rank  the host distributes its 128 bytes, every rank r of `world` creates its shard on base->device
 *                           (ncclCommInitRank).  id == NULL with world > 1: no communicator; llpf_mbank_run then returns
 *                           this rank's slots (zeros elsewhere) and the caller owns the exchange.
 * `models` is NULL (replicas of base->model) or holds all n_filters descriptors, in every process.
 * librccl is loaded on first use (dlopen); LLPF_ERR_HIP with the loader's / RCCL's message if that fails.
 * STATUS of the RCCL exchange: exercised with one-rank communicators only (the development boxes have one GPU); sharding, the
 * host-summed exchange, the caller-owned exchange (id == NULL) and the partition are tested to return the unsharded sweep's bits.
 * Until a run on two or more GPUs has confirmed the same for ncclCommInitAll / ncclCommInitRank with more than one rank, treat
 * that mode as experimental: `id == NULL` + the caller's own all-reduce is the verified way to span processes. */
#define LLPF_MBANK_ID_BYTES 128
typedef struct llpf_mbank llpf_mbank;     /* opaque: shards + communicator */
int  llpf_mbank_create(const llpf_config* base, const llpf_model* models, int32_t n_filters,
                       const int32_t* devices, int32_t n_devices, llpf_mbank** out);
int  llpf_mbank_unique_id(uint8_t* id /* LLPF_MBANK_ID_BYTES */);
/* which filters of a sweep shard `shard` of `n_shards` owns (k with k mod n_shards == shard, ascending) — the partition both
 * constructors use, exposed for callers that own the exchange themselves (id == NULL) and for tests; pure host code, no device.
 * owned may be NULL (count only); replaces the index arithmetic of `map(svec) do s ... end` spread over workers
 * (reference test/runtests.jl:412-417, src/smoothing.jl:335-347) */
int  llpf_mbank_partition(int32_t n_filters, int32_t shard, int32_t n_shards, int32_t* owned, int32_t* n_owned);
int  llpf_mbank_create_rank(const llpf_config* base, const llpf_model* models, int32_t n_filters,
                            int32_t rank, int32_t world, const uint8_t* id /* LLPF_MBANK_ID_BYTES or NULL */, llpf_mbank** out);
int  llpf_mbank_destroy(llpf_mbank* m);
int  llpf_mbank_reset(llpf_mbank* m);
int  llpf_mbank_seed(llpf_mbank* m, uint64_t seed);
int  llpf_mbank_set_models(llpf_mbank* m, const llpf_model* models /* [n_filters], every rank passes all of them */);      /* llpf_bank_set_models of every local shard */
/* as llpf_bank_run on every shard, then the exchange: ll_total [n_filters] holds every filter's log-likelihood in every
 * process, *ll_sum their sum in index order (the global log-likelihood of the sweep); either may be NULL */
int  llpf_mbank_run(llpf_mbank* m, const double* U, const double* Y, int64_t T, double t_index0,
                    double* ll_total, double* ll_sum);
/* as llpf_bank_aux_run on every shard (mode as llpf_aux_run), then the same exchange */
int  llpf_mbank_aux_run(llpf_mbank* m, const double* U, const double* Y, int64_t T, int32_t mode,
                        double* ll_total, double* ll_sum);
enum { LLPF_MBANK_COLL_NONE = 0, LLPF_MBANK_COLL_RCCL = 1, LLPF_MBANK_COLL_HOST = 2, LLPF_MBANK_COLL_EXTERNAL = 3 };
typedef struct llpf_mbank_info_t {
    int32_t n_filters;            /* of the whole sweep                                                  */
    int32_t n_shards;             /* of the whole sweep (all processes)                                  */
    int32_t n_local_shards;       /* shards this handle drives                                           */
    int32_t first_local_shard;    /* global index of the first of them (= rank with one process per GPU) */
    int32_t n_local_filters;
    int32_t collective;           /* LLPF_MBANK_COLL_*: how the log-likelihood vector is exchanged       */
    double  last_run_ms;          /* device time of the last run: slowest local shard (hipEvents)        */
    double  last_collective_ms;   /* host time of the last exchange                                      */
    int64_t resample_count;       /* resampling predict! calls of the last run, summed over local filters */
} llpf_mbank_info_t;
int  llpf_mbank_info(llpf_mbank* m, llpf_mbank_info_t* info);
int  llpf_mbank_local_devices(llpf_mbank* m, int32_t* devices /* n_local_shards */);
int  llpf_mbank_set_profiling(llpf_mbank* m, int32_t on);
int  llpf_mbank_get_profile(llpf_mbank* m, int32_t local_shard, double* ms, int64_t* launches);

/* ---- measurement ------------------------------------------------------------------------ */
/* when enabled, every kernel launch of llpf_run / llpf_bank_run is bracketed by hipEvents on the
 * handle's stream; llpf_get_profile returns accumulated milliseconds and launch counts per kernel
 * class since the last llpf_set_profiling call.  Classes: 0 propagate+weight, 1 normalise (logsumexp
 * partials), 2 resample (scan + ancestor expansion), 3 other. */
enum { LLPF_PROF_PROPAGATE = 0, LLPF_PROF_NORMALISE = 1, LLPF_PROF_RESAMPLE = 2, LLPF_PROF_OTHER = 3, LLPF_PROF_CLASSES = 4 };
int  llpf_set_profiling(llpf_filter* f, int32_t on);
int  llpf_get_profile(llpf_filter* f, double* ms /* LLPF_PROF_CLASSES */, int64_t* launches /* LLPF_PROF_CLASSES */);
int  llpf_bank_set_profiling(llpf_bank* b, int32_t on);
int  llpf_bank_get_profile(llpf_bank* b, double* ms, int64_t* launches);
/* weighted_cov of the current particles under the current weights (reference src/filtering.jl:571-581), nx*nx row-major, on the device */
int  llpf_weighted_cov(llpf_filter* f, double* cov);
/* weighted_quantile of the current particles under the current weights, per state dimension: the reference's weighted_quantile(x, we, q)
 * (src/filtering.jl:583-595) = StatsBase.quantile(v, ProbabilityWeights(we), q) — sort, running sums, linear interpolation between the two
 * particles around h = q (sum(we) - we_first) + we_first — on the device; q [nq] in [0, 1], out [nq][nx] */
int  llpf_weighted_quantile(llpf_filter* f, const double* q, int32_t nq, double* out);
/* number of predict! calls of the last run that resampled (summed over filters for a bank) */
int  llpf_resample_count(llpf_filter* f, int64_t* n);
int  llpf_bank_resample_count(llpf_bank* b, int64_t* n);
/* elapsed device milliseconds of the last llpf_run / llpf_bank_run (hipEvents on the handle's stream) */
int  llpf_last_run_ms(llpf_filter* f, double* ms);
/* how the last llpf_run drove its timesteps: launches of the fused predict! kernel (0: the balanced form ran); timesteps that took the
 * source-side form of the balanced timestep (dynamics once per surviving source, k_resample_fx; 0: dynamics per output particle); and
 * the survivor fraction the choice for the NEXT run is made by (distinct ancestors per predict! / N, a step that did not resample
 * counting as 1; -1 when the model cannot take the source-side form).  Results do not depend on the form: the choice is a schedule. */
int  llpf_last_run_stats(llpf_filter* f, int64_t* fused_launches, int64_t* source_side_timesteps, double* survivor_fraction);
int  llpf_bank_last_run_ms(llpf_bank* b, double* ms);

/* ---- misc -------------------------------------------------------------------------------- */
const char* llpf_last_error(void);
int  llpf_version(int32_t* major, int32_t* minor);
int  llpf_device_count(int32_t* n);
/* device self-test of the shared deterministic math: fills out[] with f(in[]) computed on the GPU.
 * which: 0 exp, 1 log, 2 log1p, 3 sin2pi, 4 cos2pi, 5 sqrt, 6 reciprocal, 7 u64->f64 of the bit pattern */
int  llpf_selftest_math(int32_t device, int32_t which, const double* in, double* out, int64_t n);
/* device self-test of Philox + Box–Muller: out[i*nd + d] = xi_d for particle i at (step, stream) */
int  llpf_selftest_normals(int32_t device, uint64_t seed, uint32_t step, uint32_t stream, int32_t nd,
                           double* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* LLPF_H */
