// capi.hip — host orchestration and the C ABI declared in include/llpf.h.
//
// There is deliberately NO CPU implementation behind these entry points: without a gfx950 device every
// constructor fails with LLPF_ERR_NO_DEVICE.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "engine.hpp"
#include "shared/llpf_rbkf.h"

using namespace llpf;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

// (noexcept: the message is dropped, never the status, if even the copy of the message cannot be allocated)
static int fail(int code, const std::string& msg) noexcept {
    try { g_err = msg; } catch (...) { g_err.clear(); }
    return code;
}

// The barrier include/llpf.h promises ("no C++ exception crosses the ABI"; SURVEY §8(b) Errors row; the reference turns a throw
// inside the likelihood into -Inf, src/smoothing.jl:275-279, never into a dead session).  Every export is a function-try-block
//     int llpf_name(...) LLPF_TRY { ... } LLPF_GUARD (llpf_name)
// whose handler maps what the host code can throw (std::vector / std::string / std::thread / the hiprtc cache) to a status.
static int guard_catch(const char* fn) noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(LLPF_ERR_ALLOC, std::string(fn) + ": out of host memory"); }
    catch (const std::length_error& e) { return fail(LLPF_ERR_ALLOC, std::string(fn) + ": a size beyond what can be allocated (" + e.what() + ")"); }
    catch (const std::exception& e) { return fail(LLPF_ERR_INTERNAL, std::string(fn) + ": " + e.what()); }
    catch (...) { return fail(LLPF_ERR_INTERNAL, std::string(fn) + ": unknown exception"); }
}
#define LLPF_TRY try
#define LLPF_GUARD(name) catch (...) { return guard_catch(#name); }

// Fault injection for the tests of that barrier: LLPF_TEST_THROW="<kind>:<site>", kind alloc | error | other; read at a handful of
// host-side sites (one getenv per API call, none per timestep).
static void test_throw(const char* site) {
    const char* e = getenv("LLPF_TEST_THROW");
    if (!e) return;
    const char* colon = strchr(e, ':');
    if (!colon || strcmp(colon + 1, site) != 0) return;
    if (!strncmp(e, "alloc", 5)) throw std::bad_alloc();
    if (!strncmp(e, "error", 5)) throw std::runtime_error(std::string("injected at ") + site);
    throw 42;
}
// (a failed runtime call also leaves its code in the runtime's "last error", which the next launch wrapper's hipGetLastError() would
//  report as its own — a refused hipMalloc used to poison the handle's next, unrelated call: cleared here; out of memory is LLPF_ERR_ALLOC)
#define HIPC(expr)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            (void)hipGetLastError();                                                                 \
            return fail(_e == hipErrorOutOfMemory ? LLPF_ERR_ALLOC : LLPF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                            \
    } while (0)
#define CHK(expr)                                                                                    \
    do {                                                                                             \
        int _c = (expr);                                                                             \
        if (_c != LLPF_OK) return _c;                                                                \
    } while (0)

#include "host/densities.hpp"
#include "host/bank.hpp"
#include "host/fallback.hpp"
#include "host/rbkf.hpp"
#include "host/steps.hpp"
#include "host/run.hpp"
#include "host/aux.hpp"
#include "host/smooth.hpp"
#include "host/access.hpp"
#include "host/mbank.hpp"

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* llpf_last_error(void) { return g_err.c_str(); }      // (noexcept by construction: the only export without a status)

int llpf_version(int32_t* major, int32_t* minor) LLPF_TRY {
    if (major) *major = LLPF_VERSION_MAJOR;
    if (minor) *minor = LLPF_VERSION_MINOR;
    return LLPF_OK;
} LLPF_GUARD(llpf_version)

int llpf_device_count(int32_t* n) LLPF_TRY {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    if (n) *n = c;
    return LLPF_OK;
} LLPF_GUARD(llpf_device_count)

int llpf_create(const llpf_config* cfg, llpf_filter** out) LLPF_TRY {
    if (!out) return fail(LLPF_ERR_ARG, "null out pointer");
    *out = nullptr;
    llpf_filter* f = new (std::nothrow) llpf_filter();
    if (!f) return fail(LLPF_ERR_ALLOC, "out of host memory");
    int rc = bank_create(cfg, nullptr, 1, f->bank);
    if (rc != LLPF_OK) { free_bank(f->bank); delete f; return rc; }
    *out = f;
    return LLPF_OK;
} LLPF_GUARD(llpf_create)
int llpf_destroy(llpf_filter* f) LLPF_TRY {
    if (!f) return LLPF_OK;
    free_bank(f->bank);
    delete f;
    return LLPF_OK;
} LLPF_GUARD(llpf_destroy)
#define NEEDF(f) if (!(f)) return fail(LLPF_ERR_ARG, "null handle")

int llpf_reset(llpf_filter* f) LLPF_TRY { NEEDF(f); CHK(use_device(f->bank)); return bank_init_particles(f->bank, true); } LLPF_GUARD(llpf_reset)

static int bank_seed(Bank& b, uint64_t seed) {
    CHK(use_device(b));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    set_keys(b, h, seed);
    return scal_upload(b, h);
}
int llpf_seed(llpf_filter* f, uint64_t seed) LLPF_TRY { NEEDF(f); return bank_seed(f->bank, seed); } LLPF_GUARD(llpf_seed)
int llpf_set_model(llpf_filter* f, const llpf_model* model) LLPF_TRY { NEEDF(f); return bank_set_models(f->bank, model); } LLPF_GUARD(llpf_set_model)

int llpf_correct(llpf_filter* f, const double* u, const double* y, double t, double* ll) LLPF_TRY {
    NEEDF(f);
    double l = 0.0;
    int rc = bank_correct(f->bank, u, y, t, &l);
    if (ll) *ll = l;
    return rc;
} LLPF_GUARD(llpf_correct)
int llpf_predict(llpf_filter* f, const double* u, double t) LLPF_TRY { NEEDF(f); return bank_predict(f->bank, u, t); } LLPF_GUARD(llpf_predict)
int llpf_update(llpf_filter* f, const double* u, const double* y, double t, double* ll) LLPF_TRY {
    NEEDF(f);
    int rc = llpf_correct(f, u, y, t, ll);
    if (rc != LLPF_OK) return rc;
    return bank_predict(f->bank, u, t);
} LLPF_GUARD(llpf_update)

int llpf_run(llpf_filter* f, const double* U, const double* Y, int64_t T, double t_index0,
             double* ll_total, const llpf_run_outputs* o) LLPF_TRY {
    NEEDF(f);
    double lt = 0.0;
    int rc = bank_run(f->bank, U, Y, T, t_index0, &lt, o ? o->ll_steps : nullptr, o ? o->xmean : nullptr,
                      o ? o->x_hist : nullptr, o ? o->w_hist : nullptr, o ? o->we_hist : nullptr, false, o ? o->xcov : nullptr,
                      o ? o->xquant : nullptr, o ? o->quant_p : nullptr, o ? o->nq : 0);
    if (ll_total) *ll_total = lt;
    return rc;
} LLPF_GUARD(llpf_run)

int llpf_aux_correct(llpf_filter* f, double* ll) LLPF_TRY {
    NEEDF(f);
    double l = 0.0;
    int rc = bank_aux_correct(f->bank, &l, AuxOuts{}, 0);
    if (ll) *ll = l;
    return rc;
} LLPF_GUARD(llpf_aux_correct)
int llpf_aux_predict(llpf_filter* f, const double* u, const double* y1, double t) LLPF_TRY { NEEDF(f); return bank_aux_predict(f->bank, u, y1, t); } LLPF_GUARD(llpf_aux_predict)
int llpf_aux_update(llpf_filter* f, const double* u, const double* y1, double t, double* ll) LLPF_TRY {
    NEEDF(f);
    int rc = llpf_aux_correct(f, ll);
    if (rc != LLPF_OK) return rc;
    return bank_aux_predict(f->bank, u, y1, t);
} LLPF_GUARD(llpf_aux_update)
int llpf_aux_run(llpf_filter* f, const double* U, const double* Y, int64_t T, int32_t mode,
                 double* ll_total, const llpf_run_outputs* o) LLPF_TRY {
    NEEDF(f);
    if (o && (o->xcov || o->xquant)) return fail(LLPF_ERR_ARG, "the xcov / xquant outputs are provided by llpf_run only");
    double lt = 0.0;
    int rc = bank_aux_run(f->bank, U, Y, T, mode, &lt, o ? o->ll_steps : nullptr, o ? o->xmean : nullptr,
                          o ? o->x_hist : nullptr, o ? o->w_hist : nullptr, o ? o->we_hist : nullptr);
    if (ll_total) *ll_total = lt;
    return rc;
} LLPF_GUARD(llpf_aux_run)
int llpf_bank_aux_run(llpf_bank* b, const double* U, const double* Y, int64_t T, int32_t mode,
                      double* ll_total, double* ll_steps) LLPF_TRY {
    if (!b) return fail(LLPF_ERR_ARG, "null bank");
    return bank_aux_run(b->bank, U, Y, T, mode, ll_total, ll_steps, nullptr, nullptr, nullptr, nullptr);
} LLPF_GUARD(llpf_bank_aux_run)

int llpf_rb_get_covariance(llpf_filter* f, double* R) LLPF_TRY {
    NEEDF(f);
    if (!is_rb(f->bank)) return fail(LLPF_ERR_ARG, "not a Rao-Blackwellized filter");
    if (!R) return fail(LLPF_ERR_ARG, "null output");
    const int nl = f->bank.nx - f->bank.cfg.model.nxn;
    for (int i = 0; i < nl * nl; ++i) R[i] = f->bank.rb[0].R[i];
    return LLPF_OK;
} LLPF_GUARD(llpf_rb_get_covariance)

int llpf_rb_get_linear_state(llpf_filter* f, double* xl, double* R) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    if (!is_rbfull(b)) return fail(LLPF_ERR_ARG, "not a filter with per-particle covariance (LLPF_MODEL_RB_BILINEAR)");
    CHK(use_device(b));
    const int nn = b.nx, nl = b.cfg.model.rb.nxl, np = LLPF_RBF_NP(nl);
    std::vector<double> rows((size_t)(nl + np) * b.Ns);
    HIPC(hipMemcpyAsync(rows.data(), b.d_x[b.cur] + (size_t)nn * b.Ns, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    for (int64_t i = 0; i < b.N; ++i) {
        if (xl) for (int d = 0; d < nl; ++d) xl[i * nl + d] = rows[(size_t)d * b.Ns + i];
        if (R) for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c)
            R[(i * nl + r) * nl + c] = rows[(size_t)(nl + llpf_rbf_idx(r, c)) * b.Ns + i];
    }
    return LLPF_OK;
} LLPF_GUARD(llpf_rb_get_linear_state)

int llpf_smooth(llpf_filter* f, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
                const double* wef, double* xb, int64_t* idx) LLPF_TRY {
    NEEDF(f);
    return bank_smooth(f->bank, M, U, T, xf, wf, wef, xb, idx);
} LLPF_GUARD(llpf_smooth)

int llpf_num_particles(const llpf_filter* f, int64_t* n) LLPF_TRY { NEEDF(f); if (n) *n = f->bank.N; return LLPF_OK; } LLPF_GUARD(llpf_num_particles)
int llpf_index(const llpf_filter* f, int64_t* t) LLPF_TRY { NEEDF(f); if (t) *t = f->bank.t_index; return LLPF_OK; } LLPF_GUARD(llpf_index)
int llpf_set_index(llpf_filter* f, int64_t t) LLPF_TRY { NEEDF(f); f->bank.t_index = t; return LLPF_OK; } LLPF_GUARD(llpf_set_index)
int llpf_get_particles(llpf_filter* f, double* dst) LLPF_TRY { NEEDF(f); return bank_get_particles(f->bank, dst); } LLPF_GUARD(llpf_get_particles)
int llpf_get_weights(llpf_filter* f, double* dst) LLPF_TRY { NEEDF(f); return bank_get_w(f->bank, dst, false); } LLPF_GUARD(llpf_get_weights)
int llpf_get_expweights(llpf_filter* f, double* dst) LLPF_TRY { NEEDF(f); return bank_get_w(f->bank, dst, true); } LLPF_GUARD(llpf_get_expweights)

int llpf_get_ancestors(llpf_filter* f, int64_t* dst) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    HIPC(launch_anc64(b.dev(), reinterpret_cast<int64_t*>(b.d_tmp), b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(int64_t) * b.N, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
} LLPF_GUARD(llpf_get_ancestors)
int llpf_get_bins(llpf_filter* f, double* dst) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    if (!dst) return fail(LLPF_ERR_ARG, "null output");
    if (b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL)   // the reference leaves the bins of the RESIDUAL weights there (src/resample.jl:98-104)
        return fail(LLPF_ERR_ARG, "state(pf).bins is not provided for residual resampling");
    CHK(use_device(b));
    BankDev d = b.dev();
    ResArgs ra{};
    ra.mode = RES_RESAMPLE; ra.step = rel_step(b); ra.M = (int32_t)b.N; ra.anc_out = b.d_anc;
    ra.parity = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT;
    ra.bins_out = b.d_tmp; ra.only_bins = 1; ra.force = 1;
    HIPC(launch_resample(d, ra, b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(double) * b.N, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
} LLPF_GUARD(llpf_get_bins)
int llpf_set_particles(llpf_filter* f, const double* src) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    HIPC(hipMemcpyAsync(b.d_tmp, src, sizeof(double) * b.N * b.nxp, hipMemcpyHostToDevice, b.stream));
    HIPC(launch_aos2soa(b.devp(), b.d_tmp, b.d_x[b.cur], b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
} LLPF_GUARD(llpf_set_particles)
int llpf_set_weights(llpf_filter* f, const double* w) LLPF_TRY { NEEDF(f); return bank_set_weights(f->bank, w); } LLPF_GUARD(llpf_set_weights)

static int scal0(llpf_filter* f, FilterScal* out, bool decide) {
    Bank& b = f->bank;
    CHK(use_device(b));
    if (decide) HIPC(launch_ess(b.dev(), b.stream));   // sum e^2 may have been skipped by the hot loop (threshold 1)
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    *out = h[0];
    if (decide && !out->status) {   // shouldresample on the stored state (reference src/resample.jl:5-10)
        if (out->uniform) {
            const double wev = 1.0 / (double)b.N;
            out->ess = 1.0 / ((double)b.N * (wev * wev));
        }
        const double thr = b.cfg.resample_threshold;
        out->do_resample = (thr == 1.0) ? 1 : (out->ess < (double)b.N * thr ? 1 : 0);
    }
    return LLPF_OK;
}
int llpf_effective_particles(llpf_filter* f, double* ess) LLPF_TRY {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, true));
    if (ess) *ess = s.ess;
    return LLPF_OK;
} LLPF_GUARD(llpf_effective_particles)
int llpf_shouldresample(llpf_filter* f, int32_t* yes) LLPF_TRY {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, true));
    if (yes) *yes = s.do_resample;
    return LLPF_OK;
} LLPF_GUARD(llpf_shouldresample)
int llpf_last_resampled(llpf_filter* f, int32_t* yes) LLPF_TRY {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, false));
    if (yes) *yes = s.last_resampled;
    return LLPF_OK;
} LLPF_GUARD(llpf_last_resampled)
int llpf_maxw(llpf_filter* f, double* maxw) LLPF_TRY {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, false));
    if (maxw) *maxw = s.mtrue;
    return LLPF_OK;
} LLPF_GUARD(llpf_maxw)
int llpf_weighted_mean(llpf_filter* f, double* xh) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    CHK(bank_wmean(b, b.d_tmp));
    HIPC(hipMemcpyAsync(xh, b.d_tmp, sizeof(double) * b.nxp, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
} LLPF_GUARD(llpf_weighted_mean)
int llpf_weighted_cov(llpf_filter* f, double* cov) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    if (!cov) return fail(LLPF_ERR_ARG, "null output");
    if (is_rbfull(b)) return fail(LLPF_ERR_ARG, "weighted_cov is not provided for LLPF_MODEL_RB_BILINEAR (take it from the particles)");
    CHK(use_device(b));
    CHK(bank_wmean(b, b.d_tmp));
    HIPC(launch_wcov(b.dev(), b.d_tmp, b.d_tmp + MAXD, b.stream));
    HIPC(hipMemcpyAsync(cov, b.d_tmp + MAXD, sizeof(double) * b.nx * b.nx, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
} LLPF_GUARD(llpf_weighted_cov)
int llpf_weighted_quantile(llpf_filter* f, const double* q, int32_t nq, double* out) LLPF_TRY {
    NEEDF(f);
    Bank& b = f->bank;
    if (!q || !out) return fail(LLPF_ERR_ARG, "null pointer");
    if (nq < 1 || nq > 1024) return fail(LLPF_ERR_ARG, "llpf_weighted_quantile: 1 <= nq <= 1024");
    for (int i = 0; i < nq; ++i) if (!(q[i] >= 0.0 && q[i] <= 1.0)) return fail(LLPF_ERR_ARG, "llpf_weighted_quantile: a probability outside [0, 1]");
    if (is_rbfull(b)) return fail(LLPF_ERR_ARG, "weighted_quantile is not provided for LLPF_MODEL_RB_BILINEAR (take it from the particles)");
    if (b.we_is_lambda) return fail(LLPF_ERR_ARG, "weighted_quantile between the halves of an auxiliary predict!: expweights(pf) holds lambda there");
    CHK(use_device(b));
    BankDev d = b.dev();
    CHK(ensure_wq(b, q, nq));
    HIPC(hipStreamSynchronize(b.stream));                                     // q is the caller's (pageable) memory
    HIPC(launch_materialize(d, nullptr, b.d_wq_we, b.stream));               // we = expweights(pf), [N]
    CHK(ensure(&b.d_xquant, &b.cap_xq, (size_t)nq * b.nx));
    HIPC(launch_wquantile(d.xcur, b.Ns, b.nx, b.d_wq_we, b.N, b.d_wq_p, nq, b.d_xquant, b.nx, 1, b.d_wq, b.stream));      // [nq][nx]
    HIPC(hipMemcpyAsync(out, b.d_xquant, sizeof(double) * (size_t)nq * b.nx, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
} LLPF_GUARD(llpf_weighted_quantile)
int llpf_resample_count(llpf_filter* f, int64_t* n) LLPF_TRY { NEEDF(f); if (n) *n = f->bank.run_resamples; return LLPF_OK; } LLPF_GUARD(llpf_resample_count)
int llpf_model_traits(int32_t model_id, int32_t* traits) LLPF_TRY {
    if (!traits) return fail(LLPF_ERR_ARG, "null pointer");
    const int t = jit_model_traits(model_id);
    if (t < 0) return fail(LLPF_ERR_ARG, "llpf_model_traits: not the id of a run-time compiled model");
    *traits = t;
    return LLPF_OK;
} LLPF_GUARD(llpf_model_traits)
int llpf_last_run_stats(llpf_filter* f, int64_t* fused_launches, int64_t* source_side_timesteps, double* survivor_fraction) LLPF_TRY {
    NEEDF(f);
    if (fused_launches) *fused_launches = f->bank.last_run_launches;
    if (source_side_timesteps) *source_side_timesteps = f->bank.last_run_fx_steps;
    if (survivor_fraction) *survivor_fraction = f->bank.last_run_surv;
    return LLPF_OK;
} LLPF_GUARD(llpf_last_run_stats)
int llpf_last_run_ms(llpf_filter* f, double* ms) LLPF_TRY { NEEDF(f); if (ms) *ms = f->bank.last_run_ms; return LLPF_OK; } LLPF_GUARD(llpf_last_run_ms)

static int set_prof(Bank& b, int on) {
    b.profiling = on != 0;
    for (int i = 0; i < LLPF_PROF_CLASSES; ++i) { b.prof_ms[i] = 0.0; b.prof_n[i] = 0; }
    return LLPF_OK;
}
static int get_prof(Bank& b, double* ms, int64_t* n) {
    for (int i = 0; i < LLPF_PROF_CLASSES; ++i) { if (ms) ms[i] = b.prof_ms[i]; if (n) n[i] = b.prof_n[i]; }
    return LLPF_OK;
}
int llpf_set_profiling(llpf_filter* f, int32_t on) LLPF_TRY { NEEDF(f); return set_prof(f->bank, on); } LLPF_GUARD(llpf_set_profiling)
int llpf_get_profile(llpf_filter* f, double* ms, int64_t* n) LLPF_TRY { NEEDF(f); return get_prof(f->bank, ms, n); } LLPF_GUARD(llpf_get_profile)

// ---- user-supplied models (kernels/jit.hpp) -----------------------------------------------------------
int llpf_model_compile(const char* device_src, int32_t nx, int32_t ny, int32_t* model_id) LLPF_TRY {
    if (!model_id) return fail(LLPF_ERR_ARG, "null output");
    *model_id = -1;
    int ndev = 0;
    // (LLPF_JIT_COMPILE_ONLY=1: the build check on a box without a GPU — hiprtc cross-compiles for gfx950; nothing can run)
    const char* co = getenv("LLPF_JIT_COMPILE_ONLY");
    if ((hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) && !(co && atoi(co)))
        return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    std::string err;
    const int id = jit_compile_user_model(device_src, nx, ny, err);
    if (id < 0) return fail(LLPF_ERR_ARG, err);
    *model_id = id;
    return LLPF_OK;
} LLPF_GUARD(llpf_model_compile)

// ---- banks ---------------------------------------------------------------------------------------
int llpf_bank_create(const llpf_config* base, const llpf_model* models, int32_t n_filters, llpf_bank** out) LLPF_TRY {
    if (!out) return fail(LLPF_ERR_ARG, "null out pointer");
    *out = nullptr;
    // models == NULL: every filter uses base->model (Monte-Carlo replicas; seeds differ: seed + k)
    llpf_bank* b = new (std::nothrow) llpf_bank();
    if (!b) return fail(LLPF_ERR_ALLOC, "out of host memory");
    int rc = bank_create(base, models, n_filters, b->bank);
    if (rc != LLPF_OK) { free_bank(b->bank); delete b; return rc; }
    *out = b;
    return LLPF_OK;
} LLPF_GUARD(llpf_bank_create)
int llpf_bank_destroy(llpf_bank* b) LLPF_TRY {
    if (!b) return LLPF_OK;
    free_bank(b->bank);
    delete b;
    return LLPF_OK;
} LLPF_GUARD(llpf_bank_destroy)
int llpf_bank_reset(llpf_bank* b) LLPF_TRY { NEEDF(b); CHK(use_device(b->bank)); return bank_init_particles(b->bank, true); } LLPF_GUARD(llpf_bank_reset)
int llpf_bank_seed(llpf_bank* b, uint64_t seed) LLPF_TRY { NEEDF(b); return bank_seed(b->bank, seed); } LLPF_GUARD(llpf_bank_seed)
int llpf_bank_set_models(llpf_bank* b, const llpf_model* models) LLPF_TRY { NEEDF(b); return bank_set_models(b->bank, models); } LLPF_GUARD(llpf_bank_set_models)
int llpf_bank_run(llpf_bank* b, const double* U, const double* Y, int64_t T, double t_index0,
                  double* ll_total, double* ll_steps) LLPF_TRY {
    NEEDF(b);
    return bank_run(b->bank, U, Y, T, t_index0, ll_total, ll_steps, nullptr, nullptr, nullptr, nullptr);
} LLPF_GUARD(llpf_bank_run)
int llpf_bank_run_multi(llpf_bank* b, const double* U, const double* Y, int64_t T, double t_index0,
                        double* ll_total, double* ll_steps, double* xmean) LLPF_TRY {
    NEEDF(b);
    return bank_run(b->bank, U, Y, T, t_index0, ll_total, ll_steps, xmean, nullptr, nullptr, nullptr, true);
} LLPF_GUARD(llpf_bank_run_multi)
int llpf_bank_set_profiling(llpf_bank* b, int32_t on) LLPF_TRY { NEEDF(b); return set_prof(b->bank, on); } LLPF_GUARD(llpf_bank_set_profiling)
int llpf_bank_get_profile(llpf_bank* b, double* ms, int64_t* n) LLPF_TRY { NEEDF(b); return get_prof(b->bank, ms, n); } LLPF_GUARD(llpf_bank_get_profile)
int llpf_bank_resample_count(llpf_bank* b, int64_t* n) LLPF_TRY { NEEDF(b); if (n) *n = b->bank.run_resamples; return LLPF_OK; } LLPF_GUARD(llpf_bank_resample_count)
int llpf_bank_last_run_ms(llpf_bank* b, double* ms) LLPF_TRY { NEEDF(b); if (ms) *ms = b->bank.last_run_ms; return LLPF_OK; } LLPF_GUARD(llpf_bank_last_run_ms)


// ---- sweeps sharded over the GPUs of a node (host/mbank.hpp) ---------------------------------------
static int mbank_env_collective(bool distinct, int n_shards_total) {
    if (n_shards_total == 1) { const char* e = getenv("LLPF_MBANK_FORCE_RCCL"); return (e && atoi(e)) ? MBANK_COLL_RCCL : MBANK_COLL_NONE; }
    return distinct ? MBANK_COLL_RCCL : MBANK_COLL_HOST;
}
int llpf_mbank_create(const llpf_config* base, const llpf_model* models, int32_t n_filters, const int32_t* devices,
                      int32_t n_devices, llpf_mbank** out) LLPF_TRY {
    if (!out) return fail(LLPF_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (!devices || n_devices < 1) return fail(LLPF_ERR_ARG, "empty device list");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) return fail(LLPF_ERR_ARG, "device ordinal out of range");
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) distinct = false;
    }
    llpf_mbank* m = new (std::nothrow) llpf_mbank();
    if (!m) return fail(LLPF_ERR_ALLOC, "out of host memory");
    int rc = mbank_build(m, base, models, n_filters, devices, n_devices, 0, n_devices);
    if (rc == LLPF_OK) {
        m->collective = mbank_env_collective(distinct, n_devices);
        if (m->collective == MBANK_COLL_RCCL) {
            rccl_dl::Api* R = rccl_dl::api();
            if (!R->handle) rc = fail(LLPF_ERR_HIP, R->err);
            else {
                std::vector<rccl_dl::comm_t> comms((size_t)n_devices, nullptr);
                std::vector<int> devs(devices, devices + n_devices);
                (void)hipGetLastError();      // RCCL reads the runtime's sticky last-error after its own launches: start clean
                rccl_dl::result_t r = R->CommInitAll(comms.data(), n_devices, devs.data());
                if (r != rccl_dl::Success) rc = fail(LLPF_ERR_HIP, std::string("ncclCommInitAll: ") + R->GetErrorString(r));
                else for (int i = 0; i < n_devices; ++i) m->shards[i]->comm = comms[i];
            }
        }
    }
    if (rc != LLPF_OK) { const std::string keep = g_err; mbank_free(m); g_err = keep; return rc; }
    *out = m;
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_create)
int llpf_mbank_unique_id(uint8_t* id) LLPF_TRY {
    if (!id) return fail(LLPF_ERR_ARG, "null id");
    rccl_dl::Api* R = rccl_dl::api();
    if (!R->handle) return fail(LLPF_ERR_HIP, R->err);
    rccl_dl::unique_id u;
    RCCLC(R->GetUniqueId(&u));
    memcpy(id, u.internal, LLPF_MBANK_ID_BYTES);
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_unique_id)
// the partition llpf_mbank_create / llpf_mbank_create_rank use (mbank_owned), as an entry point of its own: pure host code, needs no device
int llpf_mbank_partition(int32_t n_filters, int32_t shard, int32_t n_shards, int32_t* owned, int32_t* n_owned) LLPF_TRY {
    if (n_filters < 0 || n_shards < 1 || shard < 0 || shard >= n_shards || !n_owned) return fail(LLPF_ERR_ARG, "partition: bad arguments");
    std::vector<int> o;
    mbank_owned(n_filters, shard, n_shards, o);
    *n_owned = (int32_t)o.size();
    if (owned) for (size_t i = 0; i < o.size(); ++i) owned[i] = (int32_t)o[i];
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_partition)
int llpf_mbank_create_rank(const llpf_config* base, const llpf_model* models, int32_t n_filters, int32_t rank, int32_t world,
                           const uint8_t* id, llpf_mbank** out) LLPF_TRY {
    if (!out) return fail(LLPF_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (!base) return fail(LLPF_ERR_ARG, "null config");
    if (world < 1 || rank < 0 || rank >= world) return fail(LLPF_ERR_ARG, "rank / world out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    llpf_mbank* m = new (std::nothrow) llpf_mbank();
    if (!m) return fail(LLPF_ERR_ALLOC, "out of host memory");
    const int32_t dev = base->device;
    int rc = mbank_build(m, base, models, n_filters, &dev, 1, rank, world);
    if (rc == LLPF_OK) {
        m->collective = id ? MBANK_COLL_RCCL : (world > 1 ? MBANK_COLL_EXTERNAL : mbank_env_collective(true, 1));
        if (m->collective == MBANK_COLL_RCCL) {
            rccl_dl::Api* R = rccl_dl::api();
            if (!R->handle) rc = fail(LLPF_ERR_HIP, R->err);
            else {
                rccl_dl::unique_id u;
                if (id) memcpy(u.internal, id, LLPF_MBANK_ID_BYTES);
                else { rccl_dl::result_t r0 = R->GetUniqueId(&u); if (r0 != rccl_dl::Success) rc = fail(LLPF_ERR_HIP, std::string("ncclGetUniqueId: ") + R->GetErrorString(r0)); }
                if (rc == LLPF_OK) {
                    hipSetDevice(dev);
                    (void)hipGetLastError();
                    rccl_dl::result_t r = R->CommInitRank(&m->shards[0]->comm, world, u, rank);
                    if (r != rccl_dl::Success) rc = fail(LLPF_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
                }
            }
        }
    }
    if (rc != LLPF_OK) { const std::string keep = g_err; mbank_free(m); g_err = keep; return rc; }
    *out = m;
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_create_rank)
int llpf_mbank_destroy(llpf_mbank* m) LLPF_TRY { mbank_free(m); return LLPF_OK; } LLPF_GUARD(llpf_mbank_destroy)
int llpf_mbank_reset(llpf_mbank* m) LLPF_TRY {
    NEEDF(m);
    return mbank_foreach(*m, [&](int s) -> int { Bank& b = m->shards[s]->bank; CHK(use_device(b)); return bank_init_particles(b, true); });
} LLPF_GUARD(llpf_mbank_reset)
int llpf_mbank_seed(llpf_mbank* m, uint64_t seed) LLPF_TRY {
    NEEDF(m);
    return mbank_foreach(*m, [&](int s) -> int { return bank_seed(m->shards[s]->bank, seed); });
} LLPF_GUARD(llpf_mbank_seed)
int llpf_mbank_set_models(llpf_mbank* m, const llpf_model* models) LLPF_TRY {
    NEEDF(m);
    if (!models) return fail(LLPF_ERR_ARG, "null models");
    return mbank_foreach(*m, [&](int s) -> int {
        MShard& sh = *m->shards[s];
        std::vector<llpf_model> mine;
        mine.reserve(sh.owned.size());
        for (int k : sh.owned) mine.push_back(models[k]);      // the same partition as at creation: filter k lives on shard k mod n_shards
        return bank_set_models(sh.bank, mine.data());
    });
} LLPF_GUARD(llpf_mbank_set_models)
int llpf_mbank_run(llpf_mbank* m, const double* U, const double* Y, int64_t T, double t_index0, double* ll_total, double* ll_sum) LLPF_TRY {
    NEEDF(m);
    return mbank_run(*m, U, Y, T, t_index0, ll_total, ll_sum, false, 0);
} LLPF_GUARD(llpf_mbank_run)
int llpf_mbank_aux_run(llpf_mbank* m, const double* U, const double* Y, int64_t T, int32_t mode, double* ll_total, double* ll_sum) LLPF_TRY {
    NEEDF(m);
    return mbank_run(*m, U, Y, T, 0.0, ll_total, ll_sum, true, mode);
} LLPF_GUARD(llpf_mbank_aux_run)
int llpf_mbank_info(llpf_mbank* m, llpf_mbank_info_t* info) LLPF_TRY {
    NEEDF(m);
    if (!info) return fail(LLPF_ERR_ARG, "null info");
    info->n_filters = m->n_filters;
    info->n_shards = m->n_shards_total;
    info->n_local_shards = (int32_t)m->shards.size();
    info->first_local_shard = m->first_shard;
    info->collective = m->collective;
    info->n_local_filters = 0;
    for (auto& sp : m->shards) info->n_local_filters += (int32_t)sp->owned.size();
    info->last_run_ms = m->last_run_ms;
    info->last_collective_ms = m->last_coll_ms;
    info->resample_count = 0;
    for (auto& sp : m->shards) info->resample_count += sp->bank.run_resamples;
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_info)
int llpf_mbank_local_devices(llpf_mbank* m, int32_t* devices) LLPF_TRY {
    NEEDF(m);
    if (!devices) return fail(LLPF_ERR_ARG, "devices is NULL");
    for (size_t s = 0; s < m->shards.size(); ++s) devices[s] = m->shards[s]->device;
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_local_devices)
int llpf_mbank_set_profiling(llpf_mbank* m, int32_t on) LLPF_TRY {
    NEEDF(m);
    for (auto& sp : m->shards) set_prof(sp->bank, on);
    return LLPF_OK;
} LLPF_GUARD(llpf_mbank_set_profiling)
int llpf_mbank_get_profile(llpf_mbank* m, int32_t local_shard, double* ms, int64_t* n) LLPF_TRY {
    NEEDF(m);
    if (local_shard < 0 || local_shard >= (int32_t)m->shards.size()) return fail(LLPF_ERR_ARG, "local shard out of range");
    return get_prof(m->shards[local_shard]->bank, ms, n);
} LLPF_GUARD(llpf_mbank_get_profile)

// ---- array primitives ------------------------------------------------------------------------------
// a scratch single-filter context with a dummy 1-D model, used for weights-only operations
static int scratch_bank(int32_t device, int64_t n, int strategy, std::unique_ptr<llpf_filter>& out) {
    llpf_config c;
    memset(&c, 0, sizeof(c));
    c.struct_size = sizeof(c);
    c.n_particles = n;
    c.resampling_strategy = strategy;
    c.device = device;
    c.resample_threshold = 0.1;
    c.seed = 0;
    llpf_model& m = c.model;
    m.model_id = LLPF_MODEL_LINEAR_GAUSSIAN;
    m.nx = 1; m.nu = 0; m.ny = 1;
    m.A[0] = 1.0; m.C[0] = 1.0; m.Ts = 1.0; m.supersample = 1;
    llpf_gaussian g;
    memset(&g, 0, sizeof(g));
    g.dim = 1; g.kind = LLPF_COV_SCAL; g.cov[0] = 1.0;
    m.dynamics_density = g; m.measurement_density = g; m.initial_density = g;
    out.reset(new (std::nothrow) llpf_filter());
    if (!out) return fail(LLPF_ERR_ALLOC, "out of host memory");
    int rc = bank_create(&c, nullptr, 1, out->bank);
    if (rc != LLPF_OK) { free_bank(out->bank); out.reset(); }
    return rc;
}

int llpf_logsumexp(int32_t device, double* w, double* we, int64_t n, double* ll) LLPF_TRY {
    if (!w || n < 1) return fail(LLPF_ERR_ARG, "bad arguments");
    std::unique_ptr<llpf_filter> h;
    CHK(scratch_bank(device, n, LLPF_RESAMPLE_SYSTEMATIC, h));
    Bank& b = h->bank;
    int rc = bank_set_weights(b, w);
    if (rc == LLPF_OK) {
        std::vector<FilterScal> s;
        rc = scal_download(b, s);
        if (rc == LLPF_OK) {
            if (ll) *ll = s[0].ll;
            s[0].norm_pending = 1;     // logsumexp! normalises w in place
            rc = scal_upload(b, s);
        }
        if (rc == LLPF_OK) rc = bank_get_w(b, w, false);
        if (rc == LLPF_OK && we) rc = bank_get_w(b, we, true);
    }
    free_bank(b);
    return rc;
} LLPF_GUARD(llpf_logsumexp)

int llpf_resample(int32_t device, int32_t strategy, const double* we, int64_t n, int64_t m, const double* U, int64_t* j) LLPF_TRY {
    if (!we || !U || !j || n < 1 || m < 1) return fail(LLPF_ERR_ARG, "bad arguments");
    if (m > ((int64_t)1 << 30)) return fail(LLPF_ERR_ARG, "m too large");
    std::unique_ptr<llpf_filter> h;
    CHK(scratch_bank(device, n, strategy, h));
    Bank& b = h->bank;
    int rc = LLPF_OK;
    int32_t* d_j = nullptr;
    double* d_U = nullptr;
    auto body = [&]() -> int {
        std::vector<double> stage((size_t)b.Ns, 0.0);
        memcpy(stage.data(), we, sizeof(double) * n);
        HIPC(hipMemcpyAsync(b.d_w, stage.data(), sizeof(double) * b.Ns, hipMemcpyHostToDevice, b.stream));
        const int64_t cap = (m > b.Ns ? m : b.Ns);
        std::vector<int32_t> j32((size_t)cap, 0);
        for (int64_t i = 0; i < m; ++i) j32[i] = (int32_t)j[i];
        HIPC(hipMalloc(&d_j, sizeof(int32_t) * cap));
        HIPC(hipMemcpyAsync(d_j, j32.data(), sizeof(int32_t) * cap, hipMemcpyHostToDevice, b.stream));
        const int64_t nU = (strategy == LLPF_RESAMPLE_SYSTEMATIC) ? 1 : m;
        HIPC(hipMalloc(&d_U, sizeof(double) * nU));
        HIPC(hipMemcpyAsync(d_U, U, sizeof(double) * nU, hipMemcpyHostToDevice, b.stream));
        std::vector<FilterScal> s;
        CHK(scal_download(b, s));
        s[0].uniform = 0; s[0].anc_ident_s[0] = s[0].anc_ident_s[1] = 0; s[0].status = 0; s[0].do_resample = 1;
        CHK(scal_upload(b, s));
        // ancestors are written relative to a row of stride Ns; the scratch bank has one filter, so row 0
        ResArgs ra{};
        ra.mode = RES_RESAMPLE; ra.M = (int32_t)m; ra.Uexp = d_U; ra.anc_out = d_j; ra.force = 1; ra.src_values = 1;
        HIPC(launch_resample(b.dev(), ra, b.stream));
        HIPC(hipMemcpyAsync(j32.data(), d_j, sizeof(int32_t) * m, hipMemcpyDeviceToHost, b.stream));
        HIPC(hipStreamSynchronize(b.stream));
        for (int64_t i = 0; i < m; ++i) j[i] = j32[i];
        return LLPF_OK;
    };
    rc = body();
    if (d_j) hipFree(d_j);
    if (d_U) hipFree(d_U);
    free_bank(b);
    return rc;
} LLPF_GUARD(llpf_resample)

int llpf_resample_uniforms(int32_t strategy, int64_t m, uint64_t seed, uint32_t step, double* u) LLPF_TRY {
    if (!u) return fail(LLPF_ERR_ARG, "null output");
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (strategy == LLPF_RESAMPLE_SYSTEMATIC) u[0] = llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, k0, k1);
    else for (int64_t i = 0; i < m; ++i) u[i] = llpf_uniform_idx((uint32_t)i, step, LLPF_STREAM_STRATIFY, k0, k1);
    return LLPF_OK;
} LLPF_GUARD(llpf_resample_uniforms)

// ---- device self-tests of the shared primitives ---------------------------------------------------
int llpf_selftest_math(int32_t device, int32_t which, const double* in, double* out, int64_t n) LLPF_TRY {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible");
    if (!in || !out || n < 1) return fail(LLPF_ERR_ARG, "bad arguments");
    HIPC(hipSetDevice(device));
    double *di = nullptr, *dout = nullptr;
    auto body = [&]() -> int {
        HIPC(hipMalloc(&di, sizeof(double) * n));
        HIPC(hipMalloc(&dout, sizeof(double) * n));
        HIPC(hipMemcpy(di, in, sizeof(double) * n, hipMemcpyHostToDevice));
        HIPC(launch_selftest_math(which, di, dout, n, nullptr));
        HIPC(hipDeviceSynchronize());
        HIPC(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
        return LLPF_OK;
    };
    const int rc = body();
    hipFree(di);
    hipFree(dout);
    return rc;
} LLPF_GUARD(llpf_selftest_math)
int llpf_selftest_normals(int32_t device, uint64_t seed, uint32_t step, uint32_t stream, int32_t nd, double* out, int64_t n) LLPF_TRY {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible");
    if (nd < 1 || nd > MAXD) return fail(LLPF_ERR_ARG, "nd out of range");
    if (!out || n < 1) return fail(LLPF_ERR_ARG, "bad arguments");
    HIPC(hipSetDevice(device));
    double* dout = nullptr;
    auto body = [&]() -> int {
        HIPC(hipMalloc(&dout, sizeof(double) * n * nd));
        HIPC(launch_selftest_normals((uint32_t)seed, (uint32_t)(seed >> 32), step, stream, nd, dout, n, nullptr));
        HIPC(hipDeviceSynchronize());
        HIPC(hipMemcpy(out, dout, sizeof(double) * n * nd, hipMemcpyDeviceToHost));
        return LLPF_OK;
    };
    const int rc = body();
    hipFree(dout);
    return rc;
} LLPF_GUARD(llpf_selftest_normals)

}  // extern "C"
