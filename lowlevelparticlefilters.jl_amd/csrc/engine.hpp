// engine.hpp — device-side data layout and kernel launch interface of the gfx950 particle-filter
// engine.  Included by kernels.hip (device code) and capi.hip (host orchestration + C ABI).
//
// HBM layout (per bank of F independent filters of N particles each; a single filter is F = 1):
//   x      [2][F][NX][Ns]  fp64  particles, structure-of-arrays, ping-pong (cur / prev) instead of the
//                                reference's copyto!(xprev, x) (reference src/filtering.jl:151)
//   w      [F][Ns]         fp64  log-weights as last written by a weighting kernel ("raw"; the
//                                normalisation (w - m) - l of reference src/utils.jl:20,24 is applied
//                                lazily by whoever reads them, using the per-filter scalars)
//   anc    [F][Ns]         i32   ancestor indices j (reference src/PFtypes.jl:14, Int64 there)
//   acc    [F][ACC_WORDS]  u64   cross-block accumulators, sharded 8 ways, one 128-B line per (word, shard):
//                                running max of w (order-preserving key, double-buffered by weighting parity)
//                                and the fixed-point sums of the exp-weights in 43-bit limbs
//                                (integer adds commute => bit-reproducible)
//   quanta [F][Ns]         u64   q_i = floor(exp(w_i - m) 2^K), written by the normalise kernel so that the scan
//                                kernel does not recompute exp (8 B/particle of extra traffic buys ~60 VALU instr.)
//   tileq  [F][P2]         u64   per-tile sums of the resampling quanta (tile prefix for the scan kernel)
//   xmpart [3][F][P1][16]  fp64  per-block sums e_i x_i, one set per accumulator slot (weighted_mean output only; never fed back)
//   scal   [F]             FilterScal    per-filter scalars (maxw, log1p(s), 1/(s+1), ESS, flags, ...)
// Ns = N rounded up to a multiple of TILE (padding lanes carry zero weight).
// The exp-weights `we` and the cumulative `bins` of the reference (src/PFtypes.jl:12,15) are never
// stored: they are recomputed from w in registers where needed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/llpf.h"
#include "shared/llpf_detmath.h"
#include "shared/llpf_fixed.h"
#include "shared/llpf_philox.h"
#include "shared/llpf_rbfull.h"

namespace llpf {

constexpr int MAXD = LLPF_MAX_DIM;        // states / outputs
constexpr int MAXU = LLPF_MAX_INPUTS;     // inputs
constexpr int BLOCK = 256;            // 4 wave64 per workgroup
constexpr int STEP_PPT = 2;           // particles per thread per iteration in the step kernel (16-B vectors)
constexpr int STEP_ITERS = 1;         // iterations per block  -> 512 particles per block
constexpr int STEP_TILE = BLOCK * STEP_PPT * STEP_ITERS;
constexpr int NORM_IPT = 4;           // items per thread in normalise / resample kernels
constexpr int TILE = BLOCK * NORM_IPT;   // 1024 particles per tile: normalise and resample MUST share it

// Cross-block accumulators, per filter.  Every (word, shard) pair lives on its own 128-byte line: device-scope
// atomics to one line serialise at ~11 ns each (measured: 8 shards sharing a line cost +11..15 us per launch),
// so a launch of ~1000-2000 blocks puts only ~120-250 atomics on any line, spread over its whole duration.
constexpr int NSHARD = 8;
constexpr int ACC_STRIDE = 16;                  // u64 per (word, shard) slot = 128 B
// Every word exists three times ("slots").  A weighting kernel at timestep k accumulates its maximum into slot
// (k+1)%3 while — in the fused kernel — other blocks of the SAME launch are still reading slot k%3; the reader
// clears slot (k+2)%3, whose last reader finished two launches ago.  No grid-wide synchronisation is needed.
constexpr int ACC_NSLOT = 3;
constexpr int ACC_NWORDS = 24;
__host__ __device__ constexpr int ACC_PM(int p) { return p; }            // max key
__host__ __device__ constexpr int ACC_S(int p) { return 3 + 3 * p; }     // 3 limbs (43 bit) of sum fix96(e)
__host__ __device__ constexpr int ACC_E2(int p) { return 12 + 3 * p; }   // 3 limbs of sum fix96(e^2)
__host__ __device__ constexpr int ACC_BAD(int p) { return 21 + p; }      // count of NaN exp-weights
// the 8 words of one slot in the order the reader's lane groups fetch them
__host__ __device__ constexpr int acc_word_of_group(int g, int p) {
    return g == 0 ? ACC_PM(p) : (g <= 3 ? ACC_S(p) + (g - 1) : (g <= 6 ? ACC_E2(p) + (g - 4) : ACC_BAD(p)));
}
constexpr int ACC_WORDS = ACC_NWORDS * NSHARD * ACC_STRIDE;   // u64 per filter (24 KB)

// derived Gaussian (host-prepared): mirrors oracle/llpf_oracle.c:gaussd field for field
struct GaussD {
    int32_t dim, kind;
    double mu[MAXD];
    double L[MAXD * MAXD];
    double scal, sqrtscal, invscal;   // invscal = 1/scal: the device order multiplies (reference divides, utils.jl:110)
    double invLd[MAXD];               // 1 / L[i][i] for the two triangular solves of the full-covariance form
    double diag[MAXD], invdiag[MAXD], sqrtdiag[MAXD];
    double c0;
};

// Rao-Blackwellized filter with constant matrices (reference src/rbpf.jl): what one correct! / predict! needs from the
// shared covariance recursion (computed on the host, csrc/shared/llpf_rbkf.h): N(0,S) and the gain K of the
// measurement update, the gain L of the time update, and the inner KalmanFilter object's mean (C == 0 quirk, :279)
struct RBStep {
    GaussD dS;                 // SimpleMvNormal(PDMat(S, S_chol)), S = symmetrize(C R C') + R2
    double K[MAXD * MAXD];     // nxl x ny, row-major dense (stride ny)
    double L[MAXD * MAXD];     // nxl x nxn, row-major dense (stride nxn)
    double kfx[MAXD];          // kf.x as the reference leaves it in every particle when C == 0
};

enum { QTC_1A = 0, QTC_1A_SW, QTC_1B, QTC_1U, QTC_2A, QTC_2B, QTC_2U, QTC_3A, QTC_3U, QTC_4A, QTC_4U, QTC_TG, QTC_H, QTC_H2, QTC_H6, QTC_COUNT };
struct ModelD {
    int32_t model_id, nx, nu, ny;
    double A[MAXD * MAXD], B[MAXD * MAXU], C[MAXD * MAXD];
    double qt[LLPF_QT_COUNT];
    double qtc[QTC_COUNT];         // quad-tank: coefficients of the right-hand side and the RK4 step sizes, formed once on the host (host/densities.hpp)
    int32_t supersample, nxn;      // nxn: LLPF_MODEL_RB_LINEAR, number of nonlinear states (A = [Fn An; 0 Al], B = [Bn; Bl], C = [Gn Cl])
    int32_t rb_zeroC, rb_zeroAn;   // iszero(C), iszero(An)  (reference src/rbpf.jl:175,244)
    double Ts;
    GaussD df, dg, d0;
    llpf_rbf_par rbf;              // LLPF_MODEL_RB_BILINEAR: linear substate and coupling (csrc/shared/llpf_rbfull.h)
};

struct FilterScal {
    double m;            // maxw: offset of the last normalisation (reference state.maxw[])
    double s;            // sum_{i != argmax} exp(w_i - m)
    double l;            // log1p(s)
    double inv;          // 1/(s+1)
    double ll;           // l + m : log-likelihood increment of the last correct!
    double ess;          // effective_particles of the current weights
    double e2;           // sum exp(w_i - m)^2
    double wconst;       // value of the uniform log-weights (when uniform != 0)
    double ll_total;     // running sum of ll over a run
    uint64_t totQ;       // sum_i q64(e_i, K): total of the resampling bins
    int32_t K;           // fraction bits of the resampling bins
    int32_t uniform;     // weights are uniform (after reset! / after a resampling predict!)
    int32_t norm_pending;// w holds raw values; normalised value is (w - m) - l
    int32_t do_resample; // decision of shouldresample for the next propagate
    int32_t anc_ident_s[2]; // state.j == 1:N (last predict! did not resample); [n_predict & 1] is current, the predict! in flight writes the other
    int32_t status;      // 0 ok, LLPF_ERR_DEGENERATE
    int32_t last_resampled;
    int64_t resample_count;
    uint32_t k0, k1;     // Philox key of this filter
    uint32_t step_base;  // added to the Philox step arguments of every launch: a captured run (hipGraph) keeps its relative
    uint32_t pad_sb;     //   steps while successive runs still draw fresh noise
    double u_slot[ACC_NSLOT];   // the uniform of the systematic resample that consumes accumulator slot p (computed once per step)
    double stot;         // sum of exp(w - m) over ALL particles in the form the last normalisation used (exact form: s + 1)
    double mtrue;        // true maximum of the raw log-weights (state.maxw[]); m above is the OFFSET (bound or maximum)
    double wmax;         // maximum of the current normalised / uniform log-weights: input of the next bound
    double off_slot[ACC_NSLOT]; // offset (bound) the weighting kernel that filled accumulator slot p used for its exp-sums
    int32_t fast;        // last normalisation used the bound-offset form
    int32_t fallback;    // a fast head found sum exp(w - bound) < 2^-10: the host must redo this step in exact form
    int64_t fb_step;     // run-step index of that head
    int32_t xm_parts;    // number of per-block partial sums in xmpart written by the last weighting / normalise kernel
    int32_t pad2;
    int32_t e2_valid;    // sum e^2 / ESS are known for the current (finalized) weights (skipped when resample_threshold == 1)
    int32_t e2v_slot[ACC_NSLOT]; // sum e^2 was accumulated into accumulator slot p
    int32_t exact_slot[ACC_NSLOT]; // the sums of slot p are in exact-max form although a bound was available: a filter of ONE tile whose
    int32_t pad4;                //   bound test failed redoes its sums inside the producing kernel (no host round trip)
};

// arguments common to the step-path kernels
struct BankDev {
    int64_t N;           // particles per filter
    int64_t Ns;          // padded stride
    int32_t F;           // filters
    int32_t nx, nu, ny;
    int32_t strategy;
    int32_t model_id;
    int32_t P1;          // step-kernel blocks per filter
    int32_t P2;          // tiles per filter (tileq / xmpart entries)
    double thr;          // resample_threshold
    double log1N;        // log(1/N)   (reset_weights!, reference src/utils.jl:75)
    double mlogN;        // -log(N)    (reset!,         reference src/filtering.jl:11)
    const ModelD* models;
    FilterScal* scal;
    double* xcur;        // [F][NX][Ns] current particles (read by propagate, by weight-only)
    double* xnext;       // [F][NX][Ns] written by propagate
    double* w;           // [F][Ns]
    double* w_next;      // where the fused kernel stores the weights it forms: w itself, or the second buffer of a split-schedule run (host/run.hpp)
    int32_t* anc;        // [F][Ns]
    uint64_t* acc;       // [F][ACC_WORDS]
    uint64_t* quanta;    // [F][Ns]  quanta of the CURRENT weights (read by the scan)
    uint64_t* quanta_next; // [F][Ns] written by a weighting phase (ping-pong: other blocks may still read `quanta`)
    uint64_t* tileq;     // [ACC_NSLOT][F][P2] per-tile quanta sums, one set per accumulator slot
    uint64_t* tpre;      // [F][P2] filters above 1024 tiles only (else nullptr): exclusive prefix of each tile's sum inside its group of
    uint64_t* gsum;      // [F][ceil(P2/1024)] 1024 tiles, and the group totals — k_tile_prefix, in front of every kernel with a head
    uint32_t* bank_flag; // [1] 0, or 1 + the run-step index at which some filter's bound test failed: every later
                         //     launch of the run is a no-op until the host has redone that step in exact form
    double* xmpart;      // [ACC_NSLOT][F][P1][MAXD] (kernels/accum.hpp: xmpart_slot)
    double* lam;         // [F][Ns] lambda of the AuxiliaryParticleFilter predict! (nullptr until first used)
    uint64_t* rtile;     // [F][2][P2] residual resampling: per-tile copy counts / residual sums, then their inclusive prefixes
    int32_t anc_slot;    // n_predict & 1: index of the current FilterScal::anc_ident_s entry
    int32_t pad0;
    int32_t xrows;       // rows of one filter's particle plane: nx, or xn + xl + packed R for LLPF_MODEL_RB_BILINEAR (the stride between
    int32_t pad1;        //   the filters of a bank in xcur / xnext; kernels that know the model use their own constant)
    // source-side dynamics (k_resample_fx + k_step with StepArgs::marks; kernels/resfx.hpp): allocated on first use
    int32_t* mark;       // [F][Ns] run-start marks of a resampling: 1 + ancestor at the first output of every surviving source and at
                         //         every k_step block boundary inside its output range; zero everywhere else (k_step clears what it reads)
    unsigned long long* surv;   // [F][P2][4] per tile (and wave of k_resample), over the predict!s of a run: sources whose f(x) the step needed (distinct ancestors; the tile's
                         //   particles when nothing was resampled) — what the host chooses the next run's form of the balanced timestep by
                         //   (host/run.hpp).  One plain read-modify-write per block and step: same-address atomics from 977 blocks cost 8 us.
    double* fxs;         // [F][NX][Ns] f(x_j) of the surviving sources j, written by k_resample_fx, gathered by k_step
};
constexpr int32_t MARK_OWN = 0x40000000;     // flag of a mark that holds for its own output only: an output without an owner (resample.jl:27-35 writes nothing: j keeps its previous value)

// MODE_AUX: first half of the AuxiliaryParticleFilter predict! (reference src/filtering.jl:195-205): noise-free
// propagate of every particle (no ancestors), lambda = logpdf(y1 - g(x)), w <- w_norm + lambda, exp-sums of the new w
enum StepMode { MODE_WEIGHT = 0, MODE_PROP = 1, MODE_PROP_WEIGHT = 2, MODE_AUX = 3,
                MODE_AUX2 = 4 };   // second half of the auxiliary predict! in balanced form (ancestors in HBM: residual resampling), k_step<NoModel>

struct StepArgs {
    const double* u;       // device pointer to u of the propagate (nu doubles) or nullptr
    const double* y;       // device pointer to y of the weighting (ny doubles)
    double t_prop;         // time passed to dynamics
    double t_meas;         // time passed to measurement
    uint32_t step;         // Philox step counter of this predict!
    int32_t has_y;         // 0: measurement missing (weights pass through)
    int32_t parity;        // accumulator slot (0..2) this weighting writes its maximum, exp-sums and tile sums to
    int32_t need_e2;       // accumulate sum e^2 (ESS) — not needed when resample_threshold == 1
    int32_t K;             // fraction bits of the quanta
    int32_t only_fallback; // exact redo of one step: only filters whose `fallback` flag is set take part
    int64_t k;             // run-step index of this launch (for the bank_flag epoch test)
    uint32_t next_step;    // Philox step of the predict! that follows this weighting (its systematic offset is precomputed)
    int32_t want_xmean;    // also accumulate per-block sums e_i x_i for the weighted-mean output of the next finalize
    int32_t accumulate;    // 1: this weighting kernel also computes the exp-sums / quanta / tile sums of its weights (one launch
                           //    per timestep); 0: a k_norm launch in bound form does (better when the chip is saturated)
    int32_t aux;           // second half of the AuxiliaryParticleFilter predict! (k_resprop<AUX>): 1 = y1 missing, 2 = y1 present
    const RBStep* rb_corr; // LLPF_MODEL_RB_LINEAR: [F] parameters of the weighting (correct!) of this launch
    const RBStep* rb_pred; // LLPF_MODEL_RB_LINEAR: [F] parameters of the propagate (predict!) of this launch
    int32_t u_stride;      // doubles between the u / y of consecutive filters of a bank; 0: all filters share one u / y
    int32_t y_stride;
    int32_t marks;         // 1: the resampling of this predict! was done by k_resample_fx: ancestors come as run-start marks (BankDev::mark),
    int32_t pad_m;         //    f(x[ancestor]) from BankDev::fxs; k_step expands the marks and writes the ancestors itself
};

// FFBS smoother (reference src/smoothing.jl:116-143): one backward step t for all M trajectories
struct SmoothArgs {
    const double* xf_t;     // [N][nx] filter particles at time t (AoS, as forward_trajectory returns them)
    const double* wf_t;     // [N] normalised log-weights at time t
    const double* u;        // device pointer to u[t]
    double t;               // time passed to the dynamics
    double* fx;             // [nx][Ns] scratch: f(xf[n,t], u[t], p, t)
    const double* xb_next;  // [M][nx] smoothed samples at t+1
    double* xb_t;           // [M][nx] out
    int64_t* idx_t;         // [M] out: index of the particle behind every sample
    int32_t M;
    uint32_t step;          // Philox step of the draws (= t)
};
enum { RES_FINALIZE = 1, RES_RESAMPLE = 2 };

// arguments of the resample(+finalize) kernel
struct ResArgs {
    int32_t mode;          // RES_FINALIZE: derive the scalars of logsumexp!/ESS/shouldresample from the accumulators
                           // RES_RESAMPLE: scan + ancestor expansion (if the decision says so, or `force`)
    int32_t parity;        // accumulator slot (0..2) of the weighting that produced the current weights
    int32_t K;             // fraction bits of the resampling quanta
    int32_t force;         // resample regardless of the decision
    int32_t only_bins;     // write bins_out and stop
    int32_t src_values;    // w[] holds plain values in [0,1] (standalone resample(we)), not log-weights
    int32_t keep_norm;     // leave norm_pending = 0 (set_weights path: w stays as installed)
    int32_t accumulate;    // ll_total += ll
    int32_t want_xmean;
    int32_t fast_head;     // the accumulators of slot `parity` are in bound-offset form (written by a weighting phase)
    int32_t only_fallback; // exact redo of one step: only filters whose `fallback` flag is set take part
    int32_t u_from_scal;   // systematic offset U was precomputed into scal[f].u_sys by the normalise kernel
    uint32_t step;         // Philox step of this predict!
    int32_t M;             // number of outputs
    const double* Uexp;    // explicit uniforms (1 or M) or nullptr -> Philox
    int32_t* anc_out;
    double* bins_out;
    double* ll_steps;      // [T][F] or nullptr
    double* xmean;         // [T][F][nx] or nullptr
    int64_t k;             // epoch of this launch within a run (for the bank_flag stop test; recorded by a failed bound test)
    int64_t row;           // row of ll_steps / xmean this finalize writes
    int32_t count_surv;    // k_resample: add the number of distinct ancestors to BankDev::surv (models that could take the source-side form)
    int32_t ablate;        // developer aid (LLPF_ABLATE): bit0 skip RNG, bit1 skip owner search, bit2 skip model math; results invalid
    int32_t nt_id;         // k_resprop (split schedule): the steps that do not resample read and store nontemporal — set by the host for working sets well
                           // beyond the Infinity Cache (host/run.hpp: below ~7 M particles plain accesses are up to 11 % faster, above nontemporal ones)
    int32_t lazy_q;        // k_resprop (split schedule): the k_norm in front of it stored no quanta — BankDev::quanta points at the WEIGHTS and the scan
                           // forms the tile's quanta itself, floor(exp(w - offset) 2^K): the same function of the same numbers (host/run.hpp)
    uint64_t* dbg;         // optional [P2][8] phase timestamps of one launch (s_memrealtime, 100 MHz), or nullptr
};

// ---- end of the part the run-time compiled user-model kernels see (tools/gen_jit_prelude.py cuts here) ----
// launchers (kernels.hip)
hipError_t launch_init(const BankDev& b, uint32_t step, int init_anc, hipStream_t s);
hipError_t launch_init_user(const BankDev& b, const double* zero_u, uint32_t step, int init_anc, hipStream_t s);   // kernels/jit.hpp: UserModel::initial
int jit_model_traits(int model_id);   // LLPF_TRAIT_* bits of a run-time compiled model (-1: unknown id)
hipError_t launch_rbfull_jit(int fn_kind, int nn, int nl, int ny, const BankDev& b, int mode, const StepArgs& a, hipStream_t s);   // kernels/jit.hpp
int jit_prepare_rbfull(int fn_kind, int nn, int nl, int ny, std::string& err);   // compiles the shape if needed (bank construction reports the log)
int jit_builtin_lg(int nx, int ny, std::string& err);   // kernels/jit.hpp: LinGauss<nx, ny> compiled on demand (nx or ny above 4), id as a user model
hipError_t launch_smooth_fx_user(const BankDev& b, const SmoothArgs& a, hipStream_t s);   // kernels/jit.hpp: k_smooth_fx of a run-time compiled model
hipError_t launch_user_bound(int model_id, ModelD* models, int F, const double* zero_u, hipStream_t s);   // kernels/jit.hpp: bound of a user likelihood
hipError_t launch_rbfull(const BankDev& b, int mode, const StepArgs& a, hipStream_t s);   // k_rbfull.hip (called by launch_step)
hipError_t launch_rbfull_init(const BankDev& b, hipStream_t s);   // LLPF_MODEL_RB_BILINEAR: xl, R of reset!
hipError_t launch_wmean(const BankDev& b, double* out /* [F][nx] */, hipStream_t s);
// weighted_quantile of one filter's particles (k_quantile.hip): x [nx][Ns] planes, we [N] exp-weights, q / out on the device ([nq], [nq][nx]); synchronises
// k_quantile.hip: weighted_quantile by radix selection; out[q * stride_q + d * stride_d]; ws: wquantile_workspace_bytes(nx) of device memory
size_t wquantile_workspace_bytes(int nx);
hipError_t launch_wquantile(const double* x, int64_t Ns, int nx, const double* we, int64_t N, const double* q, int nq, double* out, int stride_q, int stride_d,
                            void* ws, hipStream_t s);
hipError_t launch_wcov(const BankDev& b, const double* mean /* [F][nx]: launch_wmean of the same state */, double* out /* [F][nx*nx] */, hipStream_t s);
hipError_t launch_step(const BankDev& b, int mode, const StepArgs& a, hipStream_t s);
hipError_t launch_max(const BankDev& b, int parity, hipStream_t s);           // maxima of the raw w into acc
// exp-weights, their fixed-point sums, quanta and tile sums of the current weights; offset = the maximum (exact form)
// or, with bound = 1, the analytic bound published by the weighting kernel (same results as an accumulating weighting)
hipError_t launch_norm(const BankDev& b, int parity, int want_xmean, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s);
hipError_t launch_ess(const BankDev& b, hipStream_t s);   // on-demand sum e^2 / ESS of the current weights (accessor path)
hipError_t launch_post_predict(const BankDev& b, hipStream_t s);
hipError_t launch_requant(const BankDev& b, hipStream_t s);   // quanta of the current weights from w and the stored offset (after a run whose k_norm launches stored none)
hipError_t launch_replicate_models(ModelD* models, int F, hipStream_t s);   // models[1..F) <- models[0]
// failed bound test: zero the exp-sums of `slot` (mode 0) / clear the flags (mode 1) of the filters that asked for the exact form
hipError_t launch_fb_clear(const BankDev& b, int slot, int mode, hipStream_t s);
hipError_t launch_resample(const BankDev& b, const ResArgs& a, hipStream_t s);
hipError_t launch_tile_prefix(const BankDev& b, int parity, hipStream_t s);   // no-op up to 1024 tiles; kernels/resample.hpp: k_tile_prefix
// resample with the dynamics evaluated on the SOURCE side, once per surviving particle (kernels/resfx.hpp): finalize + scan + counts,
// f(x_j) -> BankDev::fxs and run-start marks -> BankDev::mark for the k_step launch that follows with StepArgs::marks = 1
hipError_t launch_resample_fx(const BankDev& b, const ResArgs& a, const StepArgs& st, hipStream_t s);
bool resample_fx_supported(int model_id, int nx, int ny, int strategy);
// fused finalize + resample + propagate [+ weight]: one launch for predict!(u_k) and the weighting of correct!(u_{k+1}, y_{k+1})
hipError_t launch_resprop(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s);
hipError_t launch_smooth_fx(const BankDev& b, const SmoothArgs& a, hipStream_t s);
hipError_t launch_smooth_draw(const BankDev& b, const SmoothArgs& a, hipStream_t s);
hipError_t launch_bake_weights(const BankDev& b, hipStream_t s);   // w[] <- the normalised / uniform values it stands for (padding -Inf)
hipError_t launch_materialize(const BankDev& b, double* w_out, double* we_out, hipStream_t s);
hipError_t launch_soa2aos(const BankDev& b, const double* xsrc, double* dst, hipStream_t s);
hipError_t launch_aos2soa(const BankDev& b, const double* src, double* xdst, hipStream_t s);
hipError_t launch_anc64(const BankDev& b, int64_t* dst, hipStream_t s);
hipError_t launch_selftest_math(int which, const double* in, double* out, int64_t n, hipStream_t s);
hipError_t launch_selftest_normals(uint32_t k0, uint32_t k1, uint32_t step, uint32_t stream, int nd,
                                   double* out, int64_t n, hipStream_t s);
bool step_supported(int model_id, int nx, int ny);
// user models compiled at run time (host/jit in kernels.hip): returns a model id >= LLPF_MODEL_USER_BASE, or -1 with `err` set
int jit_compile_user_model(const char* device_src, int nx, int ny, std::string& err);
bool rbfull_supported(int fn_kind, int nn, int nl, int ny);
int rbfull_rows(int nn, int nl);   // rows of the particle plane: xn, xl, packed R
unsigned rbfull_grid_x(const BankDev& b, int nl, int mode);   // workgroups along x of a k_rbfull launch (persistent for the 8x8 form)

}  // namespace llpf
