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
#include <string>
#include <vector>

#include "engine.hpp"
#include "shared/llpf_rbkf.h"

using namespace llpf;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPC(expr)                                                                                   \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            return fail(LLPF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)
#define CHK(expr)                                                                                    \
    do {                                                                                             \
        int _c = (expr);                                                                             \
        if (_c != LLPF_OK) return _c;                                                                \
    } while (0)

// ------------------------------------------------------------------------------------------------
// host-side preparation of the densities (same operation order as oracle/llpf_oracle.c:gauss_prepare
// in device order; transcendental = the shared deterministic log so the constant does not depend on libm)
// ------------------------------------------------------------------------------------------------
static int chol_lower(const double* S, int n, double* L) {
    memset(L, 0, sizeof(double) * MAXD * MAXD);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double acc = S[i * n + j];
            for (int k = 0; k < j; ++k) acc = acc - L[i * MAXD + k] * L[j * MAXD + k];
            if (i == j) {
                if (!(acc > 0.0)) return -1;
                L[i * MAXD + i] = llpf_sqrt(acc);
            } else {
                L[i * MAXD + j] = acc / L[j * MAXD + j];
            }
        }
    }
    return 0;
}

static int gauss_prepare(const llpf_gaussian* g, GaussD* d) {
    memset(d, 0, sizeof(*d));
    const int n = g->dim;
    if (n < 1 || n > MAXD) return -1;
    d->dim = n;
    d->kind = g->kind;
    for (int i = 0; i < n; ++i) d->mu[i] = g->mu[i];
    double logdet = 0.0;
    if (g->kind == LLPF_COV_SCAL) {
        d->scal = g->cov[0];
        if (!(d->scal > 0.0)) return -1;
        d->sqrtscal = llpf_sqrt(d->scal);
        d->invscal = 1.0 / d->scal;
        logdet = (double)n * llpf_log(d->scal);
        for (int i = 0; i < n; ++i) d->L[i * MAXD + i] = d->sqrtscal;
    } else if (g->kind == LLPF_COV_DIAG) {
        for (int i = 0; i < n; ++i) {
            d->diag[i] = g->cov[i];
            if (!(d->diag[i] > 0.0)) return -1;
            d->invdiag[i] = 1.0 / d->diag[i];
            d->sqrtdiag[i] = llpf_sqrt(d->diag[i]);
            d->L[i * MAXD + i] = d->sqrtdiag[i];
            logdet = (i == 0) ? llpf_log(d->diag[i]) : logdet + llpf_log(d->diag[i]);
        }
    } else if (g->kind == LLPF_COV_FULL) {
        if (chol_lower(g->cov, n, d->L) != 0) return -1;
        for (int i = 0; i < n; ++i) d->invLd[i] = 1.0 / d->L[i * MAXD + i];
        double dd = 0.0;
        for (int i = 0; i < n; ++i) dd = (i == 0) ? llpf_log(d->L[i * MAXD + i]) : dd + llpf_log(d->L[i * MAXD + i]);
        logdet = dd + dd;
    } else {
        return -1;
    }
    const double log2pi = llpf_log(2.0 * 3.141592653589793);
    d->c0 = -((double)n * log2pi + logdet) / 2.0;
    return 0;
}

static int model_prepare(const llpf_model* m, ModelD* d) {
    memset(d, 0, sizeof(*d));
    d->model_id = m->model_id;
    d->nx = m->nx; d->nu = m->nu; d->ny = m->ny;
    memcpy(d->A, m->A, sizeof(d->A));
    memcpy(d->B, m->B, sizeof(d->B));
    memcpy(d->C, m->C, sizeof(d->C));
    memcpy(d->qt, m->qt, sizeof(d->qt));
    d->supersample = m->supersample;
    d->Ts = m->Ts;
    if (gauss_prepare(&m->dynamics_density, &d->df)) return -1;
    if (gauss_prepare(&m->measurement_density, &d->dg)) return -2;
    if (gauss_prepare(&m->initial_density, &d->d0)) return -3;
    if (m->model_id == LLPF_MODEL_RB_LINEAR) {
        // Rao-Blackwellized model: df = R1n and d0n have dimension nxn; reset! draws xn ~ d0n and sets xl = d0l.mu exactly
        // (reference src/rbpf.jl:146-158): d0 becomes [mu_n; mu_l] + blockdiag(L_n, 0) xi
        const int nn = m->nxn, nl = m->nx - m->nxn;
        if (nn < 1 || nl < 1 || m->nx > 4) return -5;
        if (d->df.dim != nn || d->d0.dim != nn || d->dg.dim != m->ny || m->linear_noise.dim != nl || m->linear_initial.dim != nl) return -4;
        d->nxn = nn;
        d->rb_zeroAn = 1; d->rb_zeroC = 1;
        for (int r = 0; r < nn; ++r) for (int c = 0; c < nl; ++c) if (m->A[r * m->nx + nn + c] != 0.0) d->rb_zeroAn = 0;
        for (int r = 0; r < m->ny; ++r) for (int c = 0; c < nl; ++c) if (m->C[r * m->nx + nn + c] != 0.0) d->rb_zeroC = 0;
        if (!d->rb_zeroAn && nn != 1) return -6;       // L = (Al R An') / Nt is implemented for a scalar Nt
        GaussD d0n = d->d0;
        memset(&d->d0, 0, sizeof(d->d0));
        d->d0.dim = m->nx; d->d0.kind = LLPF_COV_FULL;
        for (int i = 0; i < nn; ++i) {
            d->d0.mu[i] = d0n.mu[i];
            for (int j = 0; j <= i; ++j)
                d->d0.L[i * MAXD + j] = (d0n.kind == LLPF_COV_FULL) ? d0n.L[i * MAXD + j] : (i == j ? d0n.L[i * MAXD + i] : 0.0);
        }
        for (int i = 0; i < nl; ++i) d->d0.mu[nn + i] = m->linear_initial.mu[i];
        GaussD tmp;
        if (gauss_prepare(&m->linear_noise, &tmp)) return -7;
        if (gauss_prepare(&m->linear_initial, &tmp)) return -8;
        return 0;
    }
    if (d->df.dim != m->nx || d->d0.dim != m->nx || d->dg.dim != m->ny) return -4;
    return 0;
}

// dense row-major covariance of a Gaussian descriptor
static void gauss_cov_dense(const llpf_gaussian* g, double* S) {
    const int n = g->dim;
    for (int i = 0; i < n * n; ++i) S[i] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (g->kind == LLPF_COV_SCAL) S[i * n + j] = (i == j) ? g->cov[0] : 0.0;
            else if (g->kind == LLPF_COV_DIAG) S[i * n + j] = (i == j) ? g->cov[i] : 0.0;
            else S[i * n + j] = g->cov[i * n + j];
        }
}

// ------------------------------------------------------------------------------------------------
// Bank: F independent filters of N particles on one device / one stream
// ------------------------------------------------------------------------------------------------
struct Bank {
    llpf_config cfg{};
    int F = 0;
    int64_t N = 0, Ns = 0;
    int nx = 0, nu = 0, ny = 0, P1 = 0, P2 = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    ModelD* d_models = nullptr;
    FilterScal* d_scal = nullptr;
    double* d_x[2] = {nullptr, nullptr};
    int cur = 0;
    double* d_w = nullptr;
    int32_t* d_anc = nullptr;
    uint64_t* d_acc = nullptr;
    uint64_t* d_quanta[2] = {nullptr, nullptr};
    int qcur = 0;                    // quanta buffer that holds the quanta of the current weights
    uint64_t* d_tileq = nullptr;
    uint32_t* d_flag = nullptr;
    double* d_xmpart = nullptr;
    // Rao-Blackwellized model: host side of the shared covariance recursion (csrc/shared/llpf_rbkf.h)
    struct RBHost { double R[16], kfx[4], kfR[16]; };
    std::vector<RBHost> rb;           // per filter: x[1].R and the inner KalmanFilter object's fields
    std::vector<llpf_model> hmodels;  // the F model descriptors as given at create
    RBStep* d_rb = nullptr;           // device: parameters of the single-step API ([2][F]) ...
    RBStep* d_rbseq = nullptr;        // ... and of a run ([2T+1][F]: corr_0, pred_0, corr_1, ...)
    size_t cap_rbseq = 0;
    uint64_t* d_rtile = nullptr;      // [F][2][P2] residual resampling: per-tile counts / residual sums and their prefixes
    double* d_lam = nullptr;          // [F][Ns] lambda of the AuxiliaryParticleFilter predict! (allocated on first use)
    bool aux_pending = false;         // w holds lambda - log N of an aux predict!; their exp-sums wait in slot (parity+2)%3
    bool we_is_lambda = false;        // expweights(pf) returns lambda until the next correct! (the reference keeps it in `we`)
    int parity = 0;                  // accumulator slot (0..2) the NEXT weighting kernel writes (engine.hpp ACC_NSLOT)
    double* d_uy = nullptr;          // staging for single-step u / y (2 * MAXD)
    double* d_U = nullptr;           // resident inputs of a run
    double* d_Y = nullptr;
    size_t capU = 0, capY = 0;
    double* d_ll_steps = nullptr;
    double* d_xmean = nullptr;
    size_t cap_ll = 0, cap_xm = 0;
    double* d_tmp = nullptr;         // F*N*max(nx,1) doubles (also reinterpreted as int64 / double staging)
    uint64_t seed = 0;
    uint32_t n_reset = 0, n_predict = 0;
    int64_t t_index = 0;
    // measurement
    bool profiling = false;
    double prof_ms[LLPF_PROF_CLASSES] = {0, 0, 0, 0};
    int64_t prof_n[LLPF_PROF_CLASSES] = {0, 0, 0, 0};
    struct Ev { hipEvent_t a, b; int cls; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_run0 = nullptr, ev_run1 = nullptr;
    double last_run_ms = 0.0;
    int64_t run_resamples = 0;

    BankDev dev() const {
        BankDev b;
        b.N = N; b.Ns = Ns; b.F = F; b.nx = nx; b.nu = nu; b.ny = ny;
        b.strategy = cfg.resampling_strategy;
        b.model_id = cfg.model.model_id;
        b.P1 = P1; b.P2 = P2;
        b.thr = cfg.resample_threshold;
        b.log1N = llpf_log(1.0 / (double)N);
        b.mlogN = -llpf_log((double)N);
        b.models = d_models; b.scal = d_scal;
        b.xcur = d_x[cur]; b.xnext = d_x[cur ^ 1];
        b.w = d_w; b.anc = d_anc; b.acc = d_acc; b.quanta = d_quanta[qcur]; b.quanta_next = d_quanta[qcur ^ 1]; b.tileq = d_tileq;
        b.bank_flag = d_flag; b.xmpart = d_xmpart; b.lam = d_lam; b.rtile = d_rtile;
        b.anc_slot = (int32_t)(n_predict & 1u); b.pad0 = 0;
        return b;
    }
};

struct llpf_filter { Bank bank; };
struct llpf_bank { Bank bank; };

static int use_device(const Bank& b) {
    HIPC(hipSetDevice(b.device));
    return LLPF_OK;
}

static void free_bank(Bank& b) {
    hipSetDevice(b.device);
    if (b.stream) hipStreamSynchronize(b.stream);
    hipFree(b.d_models); hipFree(b.d_scal); hipFree(b.d_x[0]); hipFree(b.d_x[1]); hipFree(b.d_w);
    hipFree(b.d_anc); hipFree(b.d_acc); hipFree(b.d_quanta[0]); hipFree(b.d_quanta[1]); hipFree(b.d_tileq); hipFree(b.d_flag); hipFree(b.d_xmpart); hipFree(b.d_lam); hipFree(b.d_rtile); hipFree(b.d_rb); hipFree(b.d_rbseq); hipFree(b.d_uy); hipFree(b.d_U); hipFree(b.d_Y);
    hipFree(b.d_ll_steps); hipFree(b.d_xmean); hipFree(b.d_tmp);
    for (auto e : b.ev_pool) hipEventDestroy(e);
    for (auto& e : b.pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    if (b.ev_run0) hipEventDestroy(b.ev_run0);
    if (b.ev_run1) hipEventDestroy(b.ev_run1);
    if (b.stream) hipStreamDestroy(b.stream);
}

static int scal_download(Bank& b, std::vector<FilterScal>& h) {
    h.resize(b.F);
    HIPC(hipMemcpyAsync(h.data(), b.d_scal, sizeof(FilterScal) * b.F, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}
static int scal_upload(Bank& b, const std::vector<FilterScal>& h) {
    HIPC(hipMemcpyAsync(b.d_scal, h.data(), sizeof(FilterScal) * b.F, hipMemcpyHostToDevice, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

static void set_keys(Bank& b, std::vector<FilterScal>& h, uint64_t seed) {
    b.seed = seed;
    b.n_reset = 0;
    for (int f = 0; f < b.F; ++f) {
        const int32_t cur = h[f].anc_ident_s[b.n_predict & 1u];     // the entry index restarts with the step counter
        h[f].anc_ident_s[0] = cur; h[f].anc_ident_s[1] = cur;
    }
    b.n_predict = 0;
    for (int f = 0; f < b.F; ++f) {
        const uint64_t s = seed + (uint64_t)f;
        h[f].k0 = (uint32_t)s;
        h[f].k1 = (uint32_t)(s >> 32);
    }
}

// profiling helpers ------------------------------------------------------------------------------
static hipEvent_t get_event(Bank& b) {
    if (!b.ev_pool.empty()) { hipEvent_t e = b.ev_pool.back(); b.ev_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
struct ProfScope {
    Bank& b; int cls; hipEvent_t e0 = nullptr;
    ProfScope(Bank& bb, int c) : b(bb), cls(c) {
        if (b.profiling) { e0 = get_event(b); hipEventRecord(e0, b.stream); }
    }
    ~ProfScope() {
        if (b.profiling) { hipEvent_t e1 = get_event(b); hipEventRecord(e1, b.stream); b.pending.push_back({e0, e1, cls}); }
    }
};
static void prof_collect(Bank& b) {
    for (auto& e : b.pending) {
        float ms = 0.f;
        hipEventSynchronize(e.b);
        hipEventElapsedTime(&ms, e.a, e.b);
        b.prof_ms[e.cls] += ms;
        b.prof_n[e.cls] += 1;
        b.ev_pool.push_back(e.a);
        b.ev_pool.push_back(e.b);
    }
    b.pending.clear();
}

static int bank_init_particles(Bank& b, bool is_reset) {
    b.aux_pending = false; b.we_is_lambda = false;
    for (size_t f = 0; f < b.rb.size(); ++f) {              // reset!(pf::RBPF): R = copy(pf.kf.d0.Sigma), src/rbpf.jl:152 (pf.kf itself is not reset)
        double S0[16];
        gauss_cov_dense(&b.hmodels[f].linear_initial, S0);
        const int nl = b.nx - b.cfg.model.nxn;
        for (int i = 0; i < nl * nl; ++i) b.rb[f].R[i] = S0[i];
    }
    // constructor (src/PFtypes.jl:65-75): x ~ d0, w = log(1/N), j = 1:N, t = 0
    // reset!      (src/filtering.jl:4-14): x ~ d0, w = -log N, we = 1/N, t = 1   (j untouched)
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    BankDev d = b.dev();
    for (int f = 0; f < b.F; ++f) {
        FilterScal& s = h[f];
        s.uniform = 1;
        s.wconst = is_reset ? d.mlogN : d.log1N;
        s.norm_pending = 0;
        s.do_resample = 0;
        s.status = 0;
        s.m = 0.0; s.s = 0.0; s.l = 0.0; s.inv = 1.0; s.ll = 0.0; s.e2 = 0.0;
        s.ess = 0.0;
        s.stot = 1.0; s.mtrue = 0.0; s.wmax = s.wconst; s.fast = 0; s.fallback = 0; s.fb_step = 0; s.e2_valid = 0;
        for (int p = 0; p < ACC_NSLOT; ++p) { s.off_slot[p] = 0.0; s.u_slot[p] = 0.0; s.e2v_slot[p] = 0; }
        s.K = llpf_qbits(b.N);
        if (!is_reset) { s.anc_ident_s[0] = s.anc_ident_s[1] = 1; s.last_resampled = 0; s.resample_count = 0; s.ll_total = 0.0; }
    }
    CHK(scal_upload(b, h));
    HIPC(hipMemsetAsync(b.d_acc, 0, sizeof(uint64_t) * (size_t)b.F * ACC_WORDS, b.stream));
    HIPC(hipMemsetAsync(b.d_tileq, 0, sizeof(uint64_t) * (size_t)ACC_NSLOT * b.F * b.P2, b.stream));
    HIPC(hipMemsetAsync(b.d_flag, 0, sizeof(uint32_t) * 4, b.stream));
    b.parity = 0;
    HIPC(launch_init(d, b.n_reset, is_reset ? 0 : 1, b.stream));
    b.n_reset++;
    b.t_index = is_reset ? 1 : 0;
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

static int bank_create(const llpf_config* cfg, const llpf_model* models, int F, Bank& b) {
    if (!cfg) return fail(LLPF_ERR_ARG, "null config");
    if (cfg->struct_size != sizeof(llpf_config)) return fail(LLPF_ERR_ARG, "llpf_config.struct_size mismatch (ABI)");
    if (F < 1) return fail(LLPF_ERR_ARG, "n_filters must be >= 1");
    const llpf_model& m0 = models ? models[0] : cfg->model;
    if (cfg->n_particles < 1 || cfg->n_particles > ((int64_t)1 << 29) - 2 * TILE)   // 32-bit byte offsets into a particle plane
        return fail(LLPF_ERR_ARG, "n_particles must be in 1..2^29-2048");
    if (m0.nx < 1 || m0.nx > MAXD || m0.ny < 1 || m0.ny > MAXD || m0.nu < 0 || m0.nu > MAXD) return fail(LLPF_ERR_ARG, "bad dimensions");
    if (!step_supported(m0.model_id, m0.nx, m0.ny))
        return fail(LLPF_ERR_ARG, "no kernel instantiated for this model/dimension (linear-Gaussian nx,ny in 1..4; quad-tank 4/2)");
    if (cfg->resampling_strategy != LLPF_RESAMPLE_SYSTEMATIC && cfg->resampling_strategy != LLPF_RESAMPLE_STRATIFIED &&
        cfg->resampling_strategy != LLPF_RESAMPLE_RESIDUAL)
        return fail(LLPF_ERR_ARG, "resampling_strategy must be systematic, stratified or residual");
    if (!(cfg->resample_threshold >= 0.0 && cfg->resample_threshold <= 1.0)) return fail(LLPF_ERR_ARG, "resample_threshold must be in [0,1]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(LLPF_ERR_ARG, "device ordinal out of range");

    b.cfg = *cfg;
    b.cfg.model = m0;
    b.F = F;
    b.N = cfg->n_particles;
    b.Ns = (b.N + TILE - 1) / TILE * TILE;
    b.nx = m0.nx; b.nu = m0.nu; b.ny = m0.ny;
    b.P1 = (int)(b.Ns / STEP_TILE);
    b.P2 = (int)(b.Ns / TILE);
    b.device = cfg->device;
    std::vector<ModelD> hm(F);
    b.hmodels.resize(F);
    for (int f = 0; f < F; ++f) {
        const llpf_model& mf = models ? models[f] : cfg->model;
        b.hmodels[f] = mf;
        if (mf.model_id != m0.model_id || mf.nx != m0.nx || mf.nu != m0.nu || mf.ny != m0.ny)
            return fail(LLPF_ERR_ARG, "all filters of a bank must share model id and dimensions");
        int rc = model_prepare(&mf, &hm[f]);
        if (rc) return fail(LLPF_ERR_ARG, "invalid density (covariance not positive definite or dimension mismatch), code " + std::to_string(rc));
    }
    HIPC(hipSetDevice(b.device));
    HIPC(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
    const size_t FN = (size_t)F * b.Ns;
    HIPC(hipMalloc(&b.d_models, sizeof(ModelD) * F));
    HIPC(hipMalloc(&b.d_scal, sizeof(FilterScal) * F));
    HIPC(hipMalloc(&b.d_x[0], sizeof(double) * FN * b.nx));
    HIPC(hipMalloc(&b.d_x[1], sizeof(double) * FN * b.nx));
    HIPC(hipMalloc(&b.d_w, sizeof(double) * FN));
    HIPC(hipMalloc(&b.d_anc, sizeof(int32_t) * FN));
    HIPC(hipMalloc(&b.d_acc, sizeof(uint64_t) * (size_t)F * ACC_WORDS));
    HIPC(hipMalloc(&b.d_quanta[0], sizeof(uint64_t) * FN));
    HIPC(hipMalloc(&b.d_quanta[1], sizeof(uint64_t) * FN));
    HIPC(hipMalloc(&b.d_tileq, sizeof(uint64_t) * (size_t)ACC_NSLOT * F * b.P2));
    HIPC(hipMalloc(&b.d_flag, sizeof(uint32_t) * 4));
    HIPC(hipMalloc(&b.d_xmpart, sizeof(double) * (size_t)F * b.P1 * MAXD));
    HIPC(hipMalloc(&b.d_rtile, sizeof(uint64_t) * (size_t)F * 2 * b.P2));
    if (m0.model_id == LLPF_MODEL_RB_LINEAR) {
        HIPC(hipMalloc(&b.d_rb, sizeof(RBStep) * 2 * (size_t)F));
        HIPC(hipMemsetAsync(b.d_rb, 0, sizeof(RBStep) * 2 * (size_t)F, b.stream));
        b.rb.resize(F);
        for (int f = 0; f < F; ++f) {                       // the inner KalmanFilter object: kf.x = d0.mu, kf.R = d0.Sigma
            double S0[16];
            gauss_cov_dense(&b.hmodels[f].linear_initial, S0);
            const int nl = m0.nx - m0.nxn;
            for (int i = 0; i < nl * nl; ++i) { b.rb[f].R[i] = S0[i]; b.rb[f].kfR[i] = S0[i]; }
            for (int i = 0; i < nl; ++i) b.rb[f].kfx[i] = b.hmodels[f].linear_initial.mu[i];
        }
    }
    HIPC(hipMemsetAsync(b.d_rtile, 0, sizeof(uint64_t) * (size_t)F * 2 * b.P2, b.stream));
    HIPC(hipMalloc(&b.d_uy, sizeof(double) * 4 * MAXD));
    HIPC(hipMalloc(&b.d_tmp, sizeof(double) * (size_t)F * b.N * (b.nx > 1 ? b.nx : 1) + 64));
    HIPC(hipMemsetAsync(b.d_x[0], 0, sizeof(double) * FN * b.nx, b.stream));
    HIPC(hipMemsetAsync(b.d_x[1], 0, sizeof(double) * FN * b.nx, b.stream));
    HIPC(hipMemsetAsync(b.d_anc, 0, sizeof(int32_t) * FN, b.stream));
    HIPC(hipMemsetAsync(b.d_scal, 0, sizeof(FilterScal) * F, b.stream));
    HIPC(hipMemsetAsync(b.d_acc, 0, sizeof(uint64_t) * (size_t)F * ACC_WORDS, b.stream));
    HIPC(hipMemsetAsync(b.d_quanta[0], 0, sizeof(uint64_t) * FN, b.stream));
    HIPC(hipMemsetAsync(b.d_quanta[1], 0, sizeof(uint64_t) * FN, b.stream));
    HIPC(hipMemsetAsync(b.d_tileq, 0, sizeof(uint64_t) * (size_t)ACC_NSLOT * F * b.P2, b.stream));
    HIPC(hipMemsetAsync(b.d_flag, 0, sizeof(uint32_t) * 4, b.stream));
    HIPC(hipMemsetAsync(b.d_xmpart, 0, sizeof(double) * (size_t)F * b.P1 * MAXD, b.stream));
    HIPC(hipMemcpyAsync(b.d_models, hm.data(), sizeof(ModelD) * F, hipMemcpyHostToDevice, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    HIPC(hipEventCreate(&b.ev_run0));
    HIPC(hipEventCreate(&b.ev_run1));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    set_keys(b, h, cfg->seed);
    CHK(scal_upload(b, h));
    return bank_init_particles(b, false);
}

static int check_status(Bank& b, std::vector<FilterScal>& h) {
    for (int f = 0; f < b.F; ++f)
        if (h[f].status) return fail(h[f].status, "degenerate weights (all -Inf or NaN) in filter " + std::to_string(f));
    return LLPF_OK;
}

// ---- exact-form redo of a normalisation whose bound test failed ------------------------------------------------
// zero the exp-sum words of accumulator slot `slot` for the filters in `fl`
static int clear_slot_sums(Bank& b, int slot, const std::vector<int>& fl) {
    for (int f : fl) {
        uint64_t* acc = b.d_acc + (size_t)f * ACC_WORDS;
        const int words[3] = {ACC_S(slot), ACC_E2(slot), ACC_BAD(slot)};
        const int nw[3] = {3, 3, 1};
        for (int q = 0; q < 3; ++q)
            HIPC(hipMemsetAsync(acc + (size_t)words[q] * NSHARD * ACC_STRIDE, 0, sizeof(uint64_t) * nw[q] * NSHARD * ACC_STRIDE, b.stream));
    }
    return LLPF_OK;
}
// which filters asked for the exact form (and at which run-step); clears nothing
static int poll_fallback(Bank& b, std::vector<int>& fl, int64_t& kf) {
    uint32_t flag = 0;
    HIPC(hipMemcpyAsync(&flag, b.d_flag, sizeof(flag), hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    fl.clear();
    kf = -1;
    if (!flag) return LLPF_OK;
    kf = (int64_t)flag - 1;
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    for (int f = 0; f < b.F; ++f) if (h[f].fallback) fl.push_back(f);
    return LLPF_OK;
}
static int clear_fallback(Bank& b, const std::vector<int>& fl) {
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    for (int f : fl) h[f].fallback = 0;
    CHK(scal_upload(b, h));
    HIPC(hipMemsetAsync(b.d_flag, 0, sizeof(uint32_t) * 4, b.stream));
    return LLPF_OK;
}
static int need_e2(const Bank& b) { return b.cfg.resample_threshold != 1.0 ? 1 : 0; }

// ---- Rao-Blackwellized model: the shared covariance recursion on the host (reference src/rbpf.jl:176-219, 247-279) ----
static bool is_rb(const Bank& b) { return b.cfg.model.model_id == LLPF_MODEL_RB_LINEAR; }
struct RBMats { int nn, nl, ny, zeroC, zeroAn; double An[16], Al[16], Cl[16], R1l[16], R1n[16], R2[16]; };
static void rb_mats(const llpf_model& m, RBMats& o) {
    const int nx = m.nx, nn = m.nxn, nl = nx - nn, ny = m.ny;
    o.nn = nn; o.nl = nl; o.ny = ny; o.zeroAn = 1; o.zeroC = 1;
    for (int r = 0; r < nn; ++r) for (int c = 0; c < nl; ++c) { o.An[r * nl + c] = m.A[r * nx + nn + c]; if (o.An[r * nl + c] != 0.0) o.zeroAn = 0; }
    for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c) o.Al[r * nl + c] = m.A[(nn + r) * nx + nn + c];
    for (int r = 0; r < ny; ++r) for (int c = 0; c < nl; ++c) { o.Cl[r * nl + c] = m.C[r * nx + nn + c]; if (o.Cl[r * nl + c] != 0.0) o.zeroC = 0; }
    gauss_cov_dense(&m.linear_noise, o.R1l);
    gauss_cov_dense(&m.dynamics_density, o.R1n);
    gauss_cov_dense(&m.measurement_density, o.R2);
}
static double rb_sqrt_host(double x) { return llpf_sqrt(x); }
// parameters of one correct! of filter f; advances the filter's shared covariance
static int rb_corr_step(Bank& b, int f, RBStep& out) {
    RBMats m;
    rb_mats(b.hmodels[f], m);
    memset(&out, 0, sizeof(out));
    for (int i = 0; i < m.nl; ++i) out.kfx[i] = b.rb[f].kfx[i];
    if (m.zeroC) {                                          // x[i] = RBParticle(xn, kf.x, kf.R) with an untouched kf, :279
        for (int i = 0; i < m.nl * m.nl; ++i) b.rb[f].R[i] = b.rb[f].kfR[i];
        return LLPF_OK;
    }
    double S[16], K[16], Rpost[16];
    if (llpf_rb_gain(m.nl, m.ny, b.rb[f].R, m.Cl, m.R2, S, K, Rpost, rb_sqrt_host)) return fail(LLPF_ERR_DEGENERATE, "RBPF: innovation covariance not positive definite");
    llpf_gaussian gs;
    memset(&gs, 0, sizeof(gs));
    gs.dim = m.ny; gs.kind = LLPF_COV_FULL;
    for (int i = 0; i < m.ny * m.ny; ++i) gs.cov[i] = S[i];
    if (gauss_prepare(&gs, &out.dS)) return fail(LLPF_ERR_DEGENERATE, "RBPF: innovation covariance not positive definite");
    for (int i = 0; i < m.nl * m.ny; ++i) out.K[i] = K[i];
    for (int i = 0; i < m.nl * m.nl; ++i) { b.rb[f].R[i] = Rpost[i]; b.rb[f].kfR[i] = Rpost[i]; }
    return LLPF_OK;
}
// parameters of one predict! of filter f; advances the filter's shared covariance
static int rb_pred_step(Bank& b, int f, RBStep& out) {
    RBMats m;
    rb_mats(b.hmodels[f], m);
    memset(&out, 0, sizeof(out));
    double L[16], R1[16];
    if (llpf_rb_predcov(m.nl, m.nn, m.zeroAn, b.rb[f].R, m.Al, m.An, m.R1l, m.R1n, L, R1)) return fail(LLPF_ERR_ARG, "RBPF: An != 0 needs one nonlinear state");
    for (int i = 0; i < m.nl * m.nn; ++i) out.L[i] = L[i];
    for (int i = 0; i < m.nl * m.nl; ++i) b.rb[f].R[i] = R1[i];
    return LLPF_OK;
}
static int rb_upload_single(Bank& b, bool corr) {
    std::vector<RBStep> hs(b.F);
    for (int f = 0; f < b.F; ++f) CHK(corr ? rb_corr_step(b, f, hs[f]) : rb_pred_step(b, f, hs[f]));
    HIPC(hipMemcpyAsync(b.d_rb + (corr ? 0 : b.F), hs.data(), sizeof(RBStep) * b.F, hipMemcpyHostToDevice, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

// ---- single steps -------------------------------------------------------------------------------
static int bank_correct(Bank& b, const double* u, const double* y, double t, double* ll_out /* [F] */) {
    CHK(use_device(b));
    b.aux_pending = false; b.we_is_lambda = false;      // new weights supersede pending aux sums
    const bool has_y = (y != nullptr) && !(y[0] != y[0]);
    double hbuf[2 * MAXD] = {0};
    if (u) for (int i = 0; i < b.nu; ++i) hbuf[i] = u[i];
    if (has_y) for (int i = 0; i < b.ny; ++i) hbuf[MAXD + i] = y[i];
    HIPC(hipMemcpyAsync(b.d_uy, hbuf, sizeof(hbuf), hipMemcpyHostToDevice, b.stream));
    const int slot = b.parity;
    {
        BankDev d = b.dev();
        StepArgs a{};
        a.u = b.d_uy; a.y = b.d_uy + MAXD; a.t_prop = t; a.t_meas = t; a.step = 0; a.has_y = has_y ? 1 : 0;
        a.parity = slot; a.need_e2 = 1; a.K = llpf_qbits(b.N); a.k = 0; a.next_step = b.n_predict; a.accumulate = 1;
        if (is_rb(b) && has_y) { CHK(rb_upload_single(b, true)); a.rb_corr = b.d_rb; }
        HIPC(launch_step(d, MODE_WEIGHT, a, b.stream));     // weights + exp-sums against the bound + quanta
    }
    b.qcur ^= 1;
    b.parity = (b.parity + 1) % ACC_NSLOT;
    BankDev d = b.dev();
    ResArgs ra{};
    ra.mode = RES_FINALIZE; ra.parity = slot; ra.M = (int32_t)b.N; ra.fast_head = 1; ra.k = 0;
    HIPC(launch_resample(d, ra, b.stream));
    std::vector<int> fl;
    int64_t kf;
    CHK(poll_fallback(b, fl, kf));
    if (!fl.empty()) {   // bound test failed: exact-max normalisation of the same weights
        CHK(clear_slot_sums(b, slot, fl));
        HIPC(launch_norm(d, slot, 0, 1, b.n_predict, 1, 0, 0, b.stream));
        ra.fast_head = 0; ra.only_fallback = 1;
        HIPC(launch_resample(d, ra, b.stream));
        CHK(clear_fallback(b, fl));
    }
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    if (ll_out) for (int f = 0; f < b.F; ++f) ll_out[f] = h[f].ll;
    return check_status(b, h);
}

static int bank_predict(Bank& b, const double* u, double t) {
    CHK(use_device(b));
    double hbuf[2 * MAXD] = {0};
    if (u) for (int i = 0; i < b.nu; ++i) hbuf[i] = u[i];
    HIPC(hipMemcpyAsync(b.d_uy, hbuf, sizeof(hbuf), hipMemcpyHostToDevice, b.stream));
    BankDev d = b.dev();
    ResArgs ra{};
    ra.mode = RES_RESAMPLE; ra.parity = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT; ra.step = b.n_predict; ra.M = (int32_t)b.N;
    ra.anc_out = b.d_anc; ra.k = 0;
    HIPC(launch_resample(d, ra, b.stream));
    StepArgs a{};
    a.u = b.d_uy; a.y = nullptr; a.t_prop = t; a.t_meas = t; a.step = b.n_predict; a.has_y = 0; a.parity = b.parity;
    a.K = llpf_qbits(b.N); a.k = 0;
    if (is_rb(b)) { CHK(rb_upload_single(b, false)); a.rb_pred = b.d_rb + b.F; }
    HIPC(launch_step(d, MODE_PROP, a, b.stream));
    HIPC(launch_post_predict(d, b.stream));
    b.cur ^= 1;
    b.n_predict++;
    b.t_index++;
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

// ---- the trajectory loop ------------------------------------------------------------------------
static int ensure(double** p, size_t* cap, size_t n) {
    if (*cap >= n && *p) return LLPF_OK;
    if (*p) hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIPC(hipMalloc(p, sizeof(double) * (n ? n : 1)));
    *cap = n;
    return LLPF_OK;
}

static int bank_run(Bank& b, const double* U, const double* Y, int64_t T, double t_index0,
                    double* ll_total /* [F] */, double* ll_steps /* [T][F] */, double* xmean /* [T][F][nx] */,
                    double* x_hist, double* w_hist, double* we_hist) {
    CHK(use_device(b));
    if (T < 1) return fail(LLPF_ERR_ARG, "T must be >= 1");
    if (!Y) return fail(LLPF_ERR_ARG, "Y is null");
    if (b.nu > 0 && !U) return fail(LLPF_ERR_ARG, "U is null");
    if ((x_hist || w_hist || we_hist) && b.F != 1) return fail(LLPF_ERR_ARG, "history outputs need a single filter");
    b.aux_pending = false; b.we_is_lambda = false;
    CHK(ensure(&b.d_U, &b.capU, (size_t)T * (b.nu > 0 ? b.nu : 1)));
    CHK(ensure(&b.d_Y, &b.capY, (size_t)T * b.ny));
    if (b.nu > 0) HIPC(hipMemcpyAsync(b.d_U, U, sizeof(double) * T * b.nu, hipMemcpyHostToDevice, b.stream));
    HIPC(hipMemcpyAsync(b.d_Y, Y, sizeof(double) * T * b.ny, hipMemcpyHostToDevice, b.stream));
    if (ll_steps) CHK(ensure(&b.d_ll_steps, &b.cap_ll, (size_t)T * b.F));
    if (xmean) CHK(ensure(&b.d_xmean, &b.cap_xm, (size_t)T * b.F * b.nx));
    {   // zero the running log-likelihood and remember the resample counter
        std::vector<FilterScal> h;
        CHK(scal_download(b, h));
        b.run_resamples = 0;
        for (int f = 0; f < b.F; ++f) { h[f].ll_total = 0.0; b.run_resamples -= h[f].resample_count; }
        CHK(scal_upload(b, h));
    }
    const double Ts = b.cfg.model.Ts;
    const int want_xm = xmean ? 1 : 0;
    const int K = llpf_qbits(b.N);
    const int ne2 = need_e2(b);
    const bool hist = x_hist || w_hist || we_hist;
    auto has_y = [&](int64_t k) { return !(Y[k * b.ny] != Y[k * b.ny]); };
    auto tk = [&](int64_t k) { return (t_index0 + (double)k) * Ts; };
    // Fused (one launch: finalize + resample + propagate + weight, a block propagates the outputs of its own source
    // tile) or balanced form (ancestors to HBM, then a uniform propagate).  The fused form saves a launch and the
    // ancestor round trip but its propagate work follows the weight distribution; models whose dynamics dominate the
    // timestep (quad-tank RK4: 32 fp64 sqrt per particle) and whose ESS is small run faster balanced (measured 69 vs
    // 121 us per timestep at N = 1e6), the linear-Gaussian model faster fused.  LLPF_UNFUSED=0/1 overrides.
    static const char* unf_env = getenv("LLPF_UNFUSED");
    const bool heavy_dynamics = b.cfg.model.model_id == LLPF_MODEL_QUADTANK_RK4;
    // residual resampling produces unsorted ancestors (copies first, multinomial draws after): always the balanced form
    const bool residual = b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL;
    const bool rbm = is_rb(b);
    const bool unfused = hist || residual || rbm || (unf_env ? atoi(unf_env) != 0 : heavy_dynamics);
    if (rbm) {
        // the whole gain schedule of the run (data independent): corr_0, pred_0, corr_1, pred_1, ..., [F] each
        const size_t need = (size_t)(2 * T + 1) * b.F;
        if (b.cap_rbseq < need) {
            if (b.d_rbseq) hipFree(b.d_rbseq);
            b.d_rbseq = nullptr; b.cap_rbseq = 0;
            HIPC(hipMalloc(&b.d_rbseq, sizeof(RBStep) * need));
            b.cap_rbseq = need;
        }
        std::vector<RBStep> seq(need);
        for (int64_t k = 0; k < T; ++k)
            for (int f = 0; f < b.F; ++f) {
                if (!(Y[k * b.ny] != Y[k * b.ny])) CHK(rb_corr_step(b, f, seq[(size_t)(2 * k) * b.F + f]));
                else memset(&seq[(size_t)(2 * k) * b.F + f], 0, sizeof(RBStep));
                CHK(rb_pred_step(b, f, seq[(size_t)(2 * k + 1) * b.F + f]));
            }
        memset(&seq[(size_t)(2 * T) * b.F], 0, sizeof(RBStep) * b.F);
        HIPC(hipMemcpyAsync(b.d_rbseq, seq.data(), sizeof(RBStep) * need, hipMemcpyHostToDevice, b.stream));
        HIPC(hipStreamSynchronize(b.stream));
    }
    // Where the exp-sums / quanta of freshly computed weights are formed (identical results either way): inside the
    // weighting phase (one launch per timestep: best when one filter of ~1e6 particles cannot fill the chip and the
    // dependent-launch latency dominates) or by a streaming k_norm launch in bound form (the fused kernel then keeps
    // its registers for the propagate and runs at higher occupancy: best when many filters saturate the SIMDs).
    // Measured on MI355X: C2 single filter 29.4 vs 30.2 us, bank 128 x 1e5: 4.3e10 vs 5.0e10 particle-steps/s.
    const char* sch_env = getenv("LLPF_SCHEDULE");       // "merged" | "split" override
    const bool merged = hist || (sch_env ? (strcmp(sch_env, "merged") == 0) : ((int64_t)b.F * b.Ns <= ((int64_t)3 << 20)));
    static const char* abl_env = getenv("LLPF_ABLATE");
    static const char* dbg_env = getenv("LLPF_DEBUG_TIMING");

    // host-side state of run-step k (the device may have to be re-driven from a step whose bound test failed)
    const int cur0 = b.cur, qcur0 = b.qcur, par0 = b.parity;
    const uint32_t np0 = b.n_predict;
    const int64_t ti0 = b.t_index;
    auto at_step = [&](int64_t k) {      // state in which step k's head runs (initial weighting done, k steps done)
        b.cur = cur0 ^ (int)(k & 1);
        b.qcur = qcur0 ^ 1 ^ (int)(k & 1);
        b.parity = (par0 + 1 + (int)(k % ACC_NSLOT)) % ACC_NSLOT;      // slot the weighting of step k writes
        b.n_predict = np0 + (uint32_t)k;
        b.t_index = ti0 + k;
    };
    auto head_slot = [&](int64_t k) { return (par0 + (int)(k % ACC_NSLOT)) % ACC_NSLOT; };

    auto res_args = [&](int64_t k, bool fast) {
        ResArgs ra{};
        ra.parity = head_slot(k); ra.step = b.n_predict; ra.M = (int32_t)b.N; ra.anc_out = b.d_anc;
        ra.accumulate = 1; ra.want_xmean = want_xm; ra.u_from_scal = 1;
        ra.ll_steps = ll_steps ? b.d_ll_steps : nullptr;
        ra.xmean = xmean ? b.d_xmean : nullptr;
        ra.k = k; ra.row = k; ra.fast_head = fast ? 1 : 0;
        ra.ablate = abl_env ? atoi(abl_env) : 0;
        return ra;
    };
    auto step_args = [&](int64_t k) {
        StepArgs st{};
        st.u = b.nu > 0 ? b.d_U + k * b.nu : nullptr;
        st.t_prop = tk(k);
        st.step = b.n_predict;
        st.parity = b.parity;
        st.need_e2 = ne2; st.K = K; st.k = k; st.next_step = b.n_predict + 1; st.want_xmean = want_xm; st.accumulate = merged ? 1 : 0;
        if (rbm) { st.rb_pred = b.d_rbseq + (size_t)(2 * k + 1) * b.F; st.rb_corr = b.d_rbseq + (size_t)(2 * k + 2) * b.F; }
        const bool weight = (k + 1 < T);
        if (weight) { st.y = b.d_Y + (k + 1) * b.ny; st.t_meas = tk(k + 1); st.has_y = has_y(k + 1) ? 1 : 0; }
        else { st.y = nullptr; st.t_meas = tk(k); st.has_y = 0; }
        return st;
    };
    // one timestep in the given form; `fast`: the head consumes the bound-offset sums of the previous weighting,
    // otherwise the exact-max sums of a k_norm launched just before (redo of a failed step, or weighted means)
    auto launch_timestep = [&](int64_t k, bool fast, int only_fb) -> int {
        at_step(k);
        BankDev d = b.dev();
        ResArgs ra = res_args(k, fast);
        ra.only_fallback = only_fb;
        StepArgs st = step_args(k);
        const bool weight = (k + 1 < T);
        if (fast && !merged) {   // split schedule: the sums of the current weights in bound form, as a streaming launch
            ProfScope ps(b, LLPF_PROF_NORMALISE);
            HIPC(launch_norm(d, ra.parity, want_xm, ne2, b.n_predict, 0, 1, k, b.stream));
        }
        if (!fast) {
            ProfScope ps(b, LLPF_PROF_NORMALISE);
            HIPC(launch_norm(d, ra.parity, want_xm, 1, b.n_predict, only_fb, 0, k, b.stream));
        }
        if (unfused) {
            {
                ra.mode = RES_FINALIZE | RES_RESAMPLE;
                ProfScope ps(b, LLPF_PROF_RESAMPLE);
                HIPC(launch_resample(d, ra, b.stream));
            }
            ProfScope ps(b, LLPF_PROF_PROPAGATE);
            st.only_fallback = only_fb;
            HIPC(launch_step(d, weight ? MODE_PROP_WEIGHT : MODE_PROP, st, b.stream));
        } else {
            uint64_t* d_dbg = nullptr;
            if (dbg_env && k == atoll(dbg_env)) {
                HIPC(hipMalloc(&d_dbg, sizeof(uint64_t) * 8 * b.P2));
                HIPC(hipMemsetAsync(d_dbg, 0, sizeof(uint64_t) * 8 * b.P2, b.stream));
                ra.dbg = d_dbg;
            }
            st.only_fallback = only_fb;
            ProfScope ps(b, LLPF_PROF_PROPAGATE);
            HIPC(launch_resprop(d, ra, st, weight ? 1 : 0, b.stream));
            if (d_dbg) {
                std::vector<uint64_t> hd((size_t)8 * b.P2);
                HIPC(hipMemcpyAsync(hd.data(), d_dbg, sizeof(uint64_t) * hd.size(), hipMemcpyDeviceToHost, b.stream));
                HIPC(hipStreamSynchronize(b.stream));
                FILE* fp = fopen("gpurun_out/llpf_timing.txt", "w");
                if (fp) {
                    for (int t = 0; t < b.P2; ++t) {
                        for (int q = 0; q < 6; ++q) fprintf(fp, "%llu ", (unsigned long long)hd[(size_t)t * 8 + q]);
                        fprintf(fp, "\n");
                    }
                    fclose(fp);
                }
                hipFree(d_dbg);
            }
        }
        return LLPF_OK;
    };

    HIPC(hipEventRecord(b.ev_run0, b.stream));
    {   // weighting of the first correct! (exp-sums against the bound, quanta, tile sums: no separate normalise pass)
        BankDev d = b.dev();
        StepArgs a{};
        a.u = b.nu > 0 ? b.d_U : nullptr; a.y = b.d_Y; a.t_prop = tk(0); a.t_meas = tk(0); a.step = 0; a.has_y = has_y(0) ? 1 : 0;
        a.parity = par0; a.need_e2 = ne2; a.K = K; a.k = 0; a.next_step = np0; a.want_xmean = want_xm; a.accumulate = merged ? 1 : 0;
        if (rbm) a.rb_corr = b.d_rbseq;
        ProfScope ps(b, LLPF_PROF_PROPAGATE);
        HIPC(launch_step(d, MODE_WEIGHT, a, b.stream));
    }
    if (hist) {
        // step-synchronous form: the normalised state between correct! and predict! is copied out
        // (forward_trajectory history, reference src/filtering.jl:357-359).  Same arithmetic as the asynchronous
        // loop below (bound-offset form, exact redo when its test fails); not a timed path.
        for (int64_t k = 0; k < T; ++k) {
            at_step(k);
            BankDev d = b.dev();
            ResArgs ra = res_args(k, true);
            ra.mode = RES_FINALIZE;
            HIPC(launch_resample(d, ra, b.stream));
            std::vector<int> fl;
            int64_t kf;
            CHK(poll_fallback(b, fl, kf));
            if (!fl.empty()) {
                CHK(clear_slot_sums(b, ra.parity, fl));
                HIPC(launch_norm(d, ra.parity, want_xm, 1, b.n_predict, 1, 0, k, b.stream));
                ra.fast_head = 0; ra.only_fallback = 1;
                HIPC(launch_resample(d, ra, b.stream));
                ra.only_fallback = 0;
                CHK(clear_fallback(b, fl));
            }
            if (x_hist) {
                HIPC(launch_soa2aos(d, b.d_x[b.cur], b.d_tmp, b.stream));
                HIPC(hipMemcpyAsync(x_hist + (size_t)k * b.N * b.nx, b.d_tmp, sizeof(double) * b.N * b.nx, hipMemcpyDeviceToHost, b.stream));
                HIPC(hipStreamSynchronize(b.stream));
            }
            if (w_hist) {
                HIPC(launch_materialize(d, b.d_tmp, nullptr, b.stream));
                HIPC(hipMemcpyAsync(w_hist + (size_t)k * b.N, b.d_tmp, sizeof(double) * b.N, hipMemcpyDeviceToHost, b.stream));
                HIPC(hipStreamSynchronize(b.stream));
            }
            if (we_hist) {
                HIPC(launch_materialize(d, nullptr, b.d_tmp, b.stream));
                HIPC(hipMemcpyAsync(we_hist + (size_t)k * b.N, b.d_tmp, sizeof(double) * b.N, hipMemcpyDeviceToHost, b.stream));
                HIPC(hipStreamSynchronize(b.stream));
            }
            ra.mode = RES_RESAMPLE;
            ra.accumulate = 0; ra.ll_steps = nullptr; ra.xmean = nullptr;
            HIPC(launch_resample(d, ra, b.stream));
            StepArgs st = step_args(k);
            HIPC(launch_step(d, (k + 1 < T) ? MODE_PROP_WEIGHT : MODE_PROP, st, b.stream));
        }
    } else {
        int64_t k0 = 0;
        while (k0 < T) {
            for (int64_t k = k0; k < T; ++k) CHK(launch_timestep(k, true, 0));
            std::vector<int> fl;
            int64_t kf;
            CHK(poll_fallback(b, fl, kf));
            if (fl.empty()) break;
            // step kf of the flagged filters: exact-max normalisation of the same weights, then the step again
            CHK(clear_slot_sums(b, head_slot(kf), fl));
            CHK(launch_timestep(kf, false, 1));
            CHK(clear_fallback(b, fl));
            k0 = kf + 1;
        }
    }
    at_step(T);
    b.qcur = qcur0 ^ (int)(T & 1);                           // the last step has no weighting phase: no quanta swap
    b.parity = (par0 + (int)(T % ACC_NSLOT)) % ACC_NSLOT;
    {
        BankDev d = b.dev();
        ProfScope ps(b, LLPF_PROF_OTHER);
        HIPC(launch_post_predict(d, b.stream));
    }
    HIPC(hipEventRecord(b.ev_run1, b.stream));
    if (ll_steps) HIPC(hipMemcpyAsync(ll_steps, b.d_ll_steps, sizeof(double) * T * b.F, hipMemcpyDeviceToHost, b.stream));
    if (xmean) HIPC(hipMemcpyAsync(xmean, b.d_xmean, sizeof(double) * T * b.F * b.nx, hipMemcpyDeviceToHost, b.stream));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, b.ev_run0, b.ev_run1));
    b.last_run_ms = ms;
    if (b.profiling) prof_collect(b);
    for (int f = 0; f < b.F; ++f) {
        if (ll_total) ll_total[f] = h[f].ll_total;
        b.run_resamples += h[f].resample_count;
    }
    return check_status(b, h);
}

// ---- AuxiliaryParticleFilter{ParticleFilter} (reference src/filtering.jl:170-217, 367-384; smoothing.jl:232-236) ----
// Launches of the auxiliary filter.  `epoch` is the position of the launch within a run (stop test of a failed
// bound); `row` the ll_steps / xmean row a finalize writes.
struct AuxOuts { double* d_ll_steps = nullptr; double* d_xmean = nullptr; int accumulate = 0; };

// correct!(pf::AuxiliaryParticleFilter): ll = logsumexp!(state) only; weights produced by an aux predict! have
// their exp-sums (against the bound c0 - log N) waiting in accumulator slot parity-1: one finalize launch
static int aux_launch_finalize(Bank& b, bool fast, int only_fb, int64_t epoch, int64_t row, const AuxOuts& o) {
    const int slot = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT;
    BankDev d = b.dev();
    if (!fast) HIPC(launch_norm(d, slot, o.d_xmean ? 1 : 0, 1, b.n_predict, only_fb, 0, epoch, b.stream));
    ResArgs ra{};
    ra.mode = RES_FINALIZE; ra.parity = slot; ra.M = (int32_t)b.N; ra.fast_head = fast ? 1 : 0; ra.K = llpf_qbits(b.N);
    ra.k = epoch; ra.row = row; ra.only_fallback = only_fb;
    ra.ll_steps = o.d_ll_steps; ra.xmean = o.d_xmean; ra.want_xmean = o.d_xmean ? 1 : 0; ra.accumulate = o.accumulate;
    ProfScope ps(b, LLPF_PROF_RESAMPLE);
    HIPC(launch_resample(d, ra, b.stream));
    return LLPF_OK;
}
// first half of predict!: k_step<MODE_AUX>  x' = f(x) (no noise), lambda = logpdf(dg, y1 - g(x')), w <- w_norm + lambda,
// exp-sums of w into slot `parity`
static int aux_launch_look(Bank& b, const double* d_u, const double* d_y1, bool has_y1, double t, int only_fb, int64_t epoch) {
    BankDev d = b.dev();
    StepArgs a{};
    a.u = d_u; a.y = d_y1; a.t_prop = t; a.t_meas = t; a.step = b.n_predict; a.has_y = has_y1 ? 1 : 0;
    a.parity = b.parity; a.need_e2 = 0; a.K = llpf_qbits(b.N); a.k = epoch; a.next_step = b.n_predict; a.accumulate = 1;
    a.only_fallback = only_fb;
    ProfScope ps(b, LLPF_PROF_NORMALISE);
    HIPC(launch_step(d, MODE_AUX, a, b.stream));
    return LLPF_OK;
}
// second half: k_resprop<AUX>  expnormalize! (head, slot parity-1) + resample (always) + x = x'[j] + noise,
// w = lambda - log N, exp-sums of the new w into slot `parity`.  Expects b.cur to point at x'.
static int aux_launch_resprop(Bank& b, bool has_y1, double t, bool fast, int only_fb, int64_t epoch, int want_xm) {
    const int slot1 = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT;
    BankDev d = b.dev();
    const int K = llpf_qbits(b.N);
    if (!fast) HIPC(launch_norm(d, slot1, 0, 0, b.n_predict, only_fb, 0, epoch, b.stream));
    ResArgs ra{};
    ra.mode = RES_FINALIZE | RES_RESAMPLE; ra.parity = slot1; ra.step = b.n_predict; ra.M = (int32_t)b.N;
    ra.anc_out = b.d_anc; ra.force = 1; ra.fast_head = fast ? 1 : 0; ra.u_from_scal = 1; ra.K = K; ra.k = epoch;
    ra.only_fallback = only_fb;
    StepArgs st{};
    st.t_prop = t; st.t_meas = t; st.step = b.n_predict; st.has_y = 0; st.parity = b.parity; st.need_e2 = 0; st.K = K;
    st.k = epoch; st.next_step = b.n_predict + 1; st.want_xmean = want_xm; st.accumulate = 1; st.aux = has_y1 ? 2 : 1;
    st.only_fallback = only_fb;
    ProfScope ps(b, LLPF_PROF_PROPAGATE);
    HIPC(launch_resprop(d, ra, st, 1, b.stream));
    return LLPF_OK;
}
static int aux_ensure_lam(Bank& b) {
    if (b.d_lam) return LLPF_OK;
    HIPC(hipMalloc(&b.d_lam, sizeof(double) * (size_t)b.F * b.Ns));
    HIPC(hipMemsetAsync(b.d_lam, 0, sizeof(double) * (size_t)b.F * b.Ns, b.stream));
    return LLPF_OK;
}

// Single-call correct!: synchronous.  Weights that do not come from an aux predict! (uniform after reset!, already
// normalised, installed) are normalised in the exact-max form.
static int bank_aux_correct(Bank& b, double* ll_out /* [F] or null */, const AuxOuts& o, int64_t row) {
    CHK(use_device(b));
    if (b.aux_pending) {
        CHK(aux_launch_finalize(b, true, 0, 0, row, o));
        std::vector<int> fl;
        int64_t kf;
        CHK(poll_fallback(b, fl, kf));
        if (!fl.empty()) {   // bound test failed: exact-max normalisation of the same weights (their max is in the slot)
            CHK(clear_slot_sums(b, (b.parity + ACC_NSLOT - 1) % ACC_NSLOT, fl));
            CHK(aux_launch_finalize(b, false, 1, 0, row, o));
            CHK(clear_fallback(b, fl));
        }
    } else {
        {
            BankDev d = b.dev();
            HIPC(launch_bake_weights(d, b.stream));
        }
        std::vector<FilterScal> h;
        CHK(scal_download(b, h));
        for (auto& s : h) { s.uniform = 0; s.norm_pending = 0; }
        CHK(scal_upload(b, h));
        HIPC(hipMemsetAsync(b.d_acc, 0, sizeof(uint64_t) * (size_t)b.F * ACC_WORDS, b.stream));
        HIPC(hipMemsetAsync(b.d_tileq, 0, sizeof(uint64_t) * (size_t)ACC_NSLOT * b.F * b.P2, b.stream));
        b.parity = 0;
        HIPC(launch_max(b.dev(), b.parity, b.stream));
        b.parity = 1;                                   // the slot just filled is parity-1
        CHK(aux_launch_finalize(b, false, 0, 0, row, o));
    }
    b.aux_pending = false;
    b.we_is_lambda = false;
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    if (ll_out) for (int f = 0; f < b.F; ++f) ll_out[f] = h[f].ll;
    return check_status(b, h);
}

// Single-call predict!(pf::AuxiliaryParticleFilter, u, y1, p, t): synchronous.  d_u / d_y1 are device pointers.
static int aux_predict_dev(Bank& b, const double* d_u, const double* d_y1, bool has_y1, double t, int want_xm) {
    if (is_rb(b)) return fail(LLPF_ERR_ARG, "the auxiliary filter is not defined for the Rao-Blackwellized model");
    if (b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL) return fail(LLPF_ERR_ARG, "the auxiliary filter supports systematic and stratified resampling");
    if (b.aux_pending) CHK(bank_aux_correct(b, nullptr, AuxOuts{}, 0));   // contract: predict! works on normalised weights
    CHK(aux_ensure_lam(b));
    CHK(aux_launch_look(b, d_u, d_y1, has_y1, t, 0, 0));
    b.parity = (b.parity + 1) % ACC_NSLOT;
    b.qcur ^= 1;
    b.cur ^= 1;                                   // the noise-free prediction is the source of the second half
    CHK(aux_launch_resprop(b, has_y1, t, true, 0, 0, want_xm));
    std::vector<int> fl;
    int64_t kf;
    CHK(poll_fallback(b, fl, kf));
    if (!fl.empty()) {   // expnormalize! of w + lambda in the exact-max form, then the second half again
        CHK(clear_slot_sums(b, (b.parity + ACC_NSLOT - 1) % ACC_NSLOT, fl));
        CHK(aux_launch_resprop(b, has_y1, t, false, 1, 0, want_xm));
        CHK(clear_fallback(b, fl));
    }
    b.parity = (b.parity + 1) % ACC_NSLOT;
    b.qcur ^= 1;
    b.cur ^= 1;
    b.n_predict++;
    b.t_index++;
    b.aux_pending = true;
    b.we_is_lambda = true;
    return LLPF_OK;
}

static int bank_aux_predict(Bank& b, const double* u, const double* y1, double t) {
    CHK(use_device(b));
    const bool has_y = (y1 != nullptr) && !(y1[0] != y1[0]);
    double hbuf[2 * MAXD] = {0};
    if (u) for (int i = 0; i < b.nu; ++i) hbuf[i] = u[i];
    if (has_y) for (int i = 0; i < b.ny; ++i) hbuf[MAXD + i] = y1[i];
    HIPC(hipMemcpyAsync(b.d_uy, hbuf, sizeof(hbuf), hipMemcpyHostToDevice, b.stream));
    CHK(aux_predict_dev(b, b.d_uy, b.d_uy + MAXD, has_y, t, 0));
    HIPC(hipStreamSynchronize(b.stream));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    return check_status(b, h);
}

// mode 0: the loop of forward_trajectory(pf::AuxiliaryParticleFilter) (src/filtering.jl:367-384, after reset!)
// mode 1: the loop of loglik(pf::AuxiliaryParticleFilter) (src/smoothing.jl:232-236): T-1 aux updates, then one update!
//         of the wrapped ParticleFilter on (u[end], y[end]).
// Without history outputs all launches are enqueued back to back (three per timestep: look-ahead, resample+propagate,
// finalize) and the bound-test flag is polled once at the end; a failed test re-drives from that launch in exact form.
static int bank_aux_run(Bank& b, const double* U, const double* Y, int64_t T, int mode, double* ll_total /* [F] */,
                        double* ll_steps, double* xmean, double* x_hist, double* w_hist, double* we_hist) {
    CHK(use_device(b));
    if (T < 1) return fail(LLPF_ERR_ARG, "T must be >= 1");
    if (!Y) return fail(LLPF_ERR_ARG, "Y is null");
    if (b.nu > 0 && !U) return fail(LLPF_ERR_ARG, "U is null");
    if (mode != 0 && mode != 1) return fail(LLPF_ERR_ARG, "mode must be 0 (forward_trajectory) or 1 (loglik)");
    if ((x_hist || w_hist || we_hist) && b.F != 1) return fail(LLPF_ERR_ARG, "history outputs need a single filter");
    CHK(ensure(&b.d_U, &b.capU, (size_t)T * (b.nu > 0 ? b.nu : 1)));
    CHK(ensure(&b.d_Y, &b.capY, (size_t)T * b.ny));
    if (b.nu > 0) HIPC(hipMemcpyAsync(b.d_U, U, sizeof(double) * T * b.nu, hipMemcpyHostToDevice, b.stream));
    HIPC(hipMemcpyAsync(b.d_Y, Y, sizeof(double) * T * b.ny, hipMemcpyHostToDevice, b.stream));
    CHK(ensure(&b.d_ll_steps, &b.cap_ll, (size_t)T * b.F));
    if (xmean) CHK(ensure(&b.d_xmean, &b.cap_xm, (size_t)T * b.F * b.nx));
    if (b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL) return fail(LLPF_ERR_ARG, "the auxiliary filter supports systematic and stratified resampling");
    CHK(aux_ensure_lam(b));
    const double Ts = b.cfg.model.Ts;
    const bool hist = x_hist || w_hist || we_hist;
    const int want_xm = xmean ? 1 : 0;
    b.run_resamples = 0;
    {
        std::vector<FilterScal> h;
        CHK(scal_download(b, h));
        for (int f = 0; f < b.F; ++f) { h[f].ll_total = 0.0; b.run_resamples -= h[f].resample_count; }
        CHK(scal_upload(b, h));
    }
    AuxOuts outs;
    outs.d_ll_steps = b.d_ll_steps; outs.d_xmean = xmean ? b.d_xmean : nullptr; outs.accumulate = 1;
    auto has_y = [&](int64_t k) { return !(Y[k * b.ny] != Y[k * b.ny]); };
    auto record = [&](int64_t k) -> int {     // x[:,t] .= particles(pf); w[:,t] .= weights(pf); we[:,t] .= expweights(pf)
        BankDev d = b.dev();
        if (x_hist) {
            HIPC(launch_soa2aos(d, b.d_x[b.cur], b.d_tmp, b.stream));
            HIPC(hipMemcpyAsync(x_hist + (size_t)k * b.N * b.nx, b.d_tmp, sizeof(double) * b.N * b.nx, hipMemcpyDeviceToHost, b.stream));
            HIPC(hipStreamSynchronize(b.stream));
        }
        if (w_hist) {
            HIPC(launch_materialize(d, b.d_tmp, nullptr, b.stream));
            HIPC(hipMemcpyAsync(w_hist + (size_t)k * b.N, b.d_tmp, sizeof(double) * b.N, hipMemcpyDeviceToHost, b.stream));
            HIPC(hipStreamSynchronize(b.stream));
        }
        if (we_hist) {
            HIPC(launch_materialize(d, nullptr, b.d_tmp, b.stream));
            HIPC(hipMemcpyAsync(we_hist + (size_t)k * b.N, b.d_tmp, sizeof(double) * b.N, hipMemcpyDeviceToHost, b.stream));
            HIPC(hipStreamSynchronize(b.stream));
        }
        return LLPF_OK;
    };
    HIPC(hipEventRecord(b.ev_run0, b.stream));
    const int64_t n_aux = T - 1;                       // aux predict! calls: k = 0 .. T-2
    // correct! of step 0 (synchronous: after reset! the weights are uniform and take the exact-max form).  loglik with
    // T = 1 consists of the wrapped filter's update! alone.
    if (mode == 0 || T > 1) CHK(bank_aux_correct(b, nullptr, outs, 0));
    if (hist) {
        // step-synchronous form (history is copied out between correct! and predict!)
        if (mode == 0 || T > 1) CHK(record(0));
        for (int64_t k = 0; k < n_aux; ++k) {
            CHK(aux_predict_dev(b, b.nu > 0 ? b.d_U + k * b.nu : nullptr, b.d_Y + (k + 1) * b.ny, has_y(k + 1), (double)k * Ts, want_xm));
            if (mode == 1 && k + 1 == T - 1) break;           // loglik: the last step is the wrapped filter's update!
            CHK(bank_aux_correct(b, nullptr, outs, k + 1));
            CHK(record(k + 1));
        }
    } else if (n_aux > 0) {
        // epochs: e = 3k+1 look-ahead(k), 3k+2 resample+propagate(k), 3k+3 finalize(k+1)
        const int P0 = b.parity, C0 = b.cur, Q0 = b.qcur;
        const uint32_t np0 = b.n_predict;
        const int64_t ti0 = b.t_index;
        const int64_t e_last = (mode == 0) ? 3 * n_aux : 3 * n_aux - 1;    // loglik: the last finalize is replaced by update!
        auto at_epoch = [&](int64_t e) {
            const int64_t k = (e - 1) / 3;
            const int r = (int)((e - 1) % 3);
            b.parity = (P0 + (int)((2 * k) % ACC_NSLOT) + (r == 0 ? 0 : (r == 1 ? 1 : 2))) % ACC_NSLOT;
            b.cur = (r == 1) ? (C0 ^ 1) : C0;
            b.qcur = (r == 1) ? (Q0 ^ 1) : Q0;
            b.n_predict = np0 + (uint32_t)k + (r == 2 ? 1u : 0u);
            b.t_index = ti0 + k + (r == 2 ? 1 : 0);
        };
        auto launch_epoch = [&](int64_t e, bool fast, int only_fb) -> int {
            at_epoch(e);
            const int64_t k = (e - 1) / 3;
            const int r = (int)((e - 1) % 3);
            const double t = (double)k * Ts;
            if (r == 0) return aux_launch_look(b, b.nu > 0 ? b.d_U + k * b.nu : nullptr, b.d_Y + (k + 1) * b.ny, has_y(k + 1), t, only_fb, e);
            if (r == 1) return aux_launch_resprop(b, has_y(k + 1), t, fast, only_fb, e, want_xm);
            return aux_launch_finalize(b, fast, only_fb, e, k + 1, outs);
        };
        int64_t e0 = 1;
        while (e0 <= e_last) {
            for (int64_t e = e0; e <= e_last; ++e) CHK(launch_epoch(e, true, 0));
            std::vector<int> fl;
            int64_t ef;
            CHK(poll_fallback(b, fl, ef));
            if (fl.empty()) break;
            // launch `ef` of the flagged filters again with an exact-max normalisation of the same weights
            at_epoch(ef);
            CHK(clear_slot_sums(b, (b.parity + ACC_NSLOT - 1) % ACC_NSLOT, fl));
            CHK(launch_epoch(ef, false, 1));
            CHK(clear_fallback(b, fl));
            e0 = ef + 1;
        }
        at_epoch(3 * n_aux);      // host state after the last resample+propagate launch (a finalize does not advance it)
        b.aux_pending = (mode == 1);
        b.we_is_lambda = (mode == 1);
    }
    std::vector<double> last(b.F, 0.0);
    if (mode == 1) {
        // pf.pf(u[end], y[end], p, (T-1)*Ts): update! of the wrapped filter
        const int64_t k = T - 1;
        CHK(bank_correct(b, b.nu > 0 ? U + k * b.nu : nullptr, Y + k * b.ny, (double)k * Ts, last.data()));
        CHK(bank_predict(b, b.nu > 0 ? U + k * b.nu : nullptr, (double)k * Ts));
    }
    HIPC(hipEventRecord(b.ev_run1, b.stream));
    std::vector<double> hl((size_t)T * b.F, 0.0);
    HIPC(hipMemcpyAsync(hl.data(), b.d_ll_steps, sizeof(double) * T * b.F, hipMemcpyDeviceToHost, b.stream));
    if (xmean) HIPC(hipMemcpyAsync(xmean, b.d_xmean, sizeof(double) * T * b.F * b.nx, hipMemcpyDeviceToHost, b.stream));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, b.ev_run0, b.ev_run1));
    b.last_run_ms = ms;
    if (b.profiling) prof_collect(b);
    for (int f = 0; f < b.F; ++f) {
        if (mode == 1) hl[(size_t)(T - 1) * b.F + f] = last[f];
        double tot = 0.0;
        for (int64_t k = 0; k < T; ++k) tot += hl[(size_t)k * b.F + f];     // same left-to-right order as the reference's sum
        if (ll_total) ll_total[f] = tot;
        b.run_resamples += h[f].resample_count;
    }
    if (ll_steps) memcpy(ll_steps, hl.data(), sizeof(double) * T * b.F);
    return check_status(b, h);
}

// ---- FFBS particle smoother (reference src/smoothing.jl:116-143) ---------------------------------------------------
extern "C" int llpf_resample(int32_t device, int32_t strategy, const double* we, int64_t n, int64_t m, const double* U, int64_t* j);

static int bank_smooth(Bank& b, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
                       const double* wef, double* xb, int64_t* idx) {
    CHK(use_device(b));
    if (b.F != 1) return fail(LLPF_ERR_ARG, "smooth needs a single filter");
    if (is_rb(b)) return fail(LLPF_ERR_ARG, "smooth is not defined for the Rao-Blackwellized model");
    if (M < 1 || M > b.N) return fail(LLPF_ERR_ARG, "M must be in 1..N (reference src/smoothing.jl:121)");
    if (T < 1 || !xf || !wf || !wef || !xb) return fail(LLPF_ERR_ARG, "bad arguments");
    if (b.nu > 0 && !U) return fail(LLPF_ERR_ARG, "U is null");
    const int64_t N = b.N;
    const int nx = b.nx;
    // j = resample(pf.resampling_strategy, wef[:,T], M) with the Philox stream SMOOTH_INIT under the filter's key
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    const uint32_t k0 = h[0].k0, k1 = h[0].k1;
    const int strategy = b.cfg.resampling_strategy;
    std::vector<double> Ures((size_t)(strategy == LLPF_RESAMPLE_SYSTEMATIC ? 1 : M));
    if (strategy == LLPF_RESAMPLE_SYSTEMATIC) Ures[0] = llpf_uniform_step((uint32_t)T, LLPF_STREAM_SMOOTH_INIT, k0, k1);
    else for (int64_t i = 0; i < M; ++i) Ures[i] = llpf_uniform_idx((uint32_t)i, (uint32_t)T, LLPF_STREAM_SMOOTH_INIT, k0, k1);
    std::vector<int64_t> j((size_t)M, 0);
    CHK(llpf_resample(b.device, strategy, wef + (size_t)(T - 1) * N, N, M, Ures.data(), j.data()));
    CHK(use_device(b));
    for (int64_t m = 0; m < M; ++m) {
        memcpy(xb + ((size_t)(T - 1) * M + m) * nx, xf + ((size_t)(T - 1) * N + j[m]) * nx, sizeof(double) * nx);
        if (idx) idx[(size_t)(T - 1) * M + m] = j[m];
    }
    if (T == 1) return LLPF_OK;
    double *d_xf = nullptr, *d_wf = nullptr, *d_fx = nullptr, *d_xb = nullptr, *d_u = nullptr;
    int64_t* d_idx = nullptr;
    auto body = [&]() -> int {
        HIPC(hipMalloc(&d_xf, sizeof(double) * (size_t)T * N * nx));
        HIPC(hipMalloc(&d_wf, sizeof(double) * (size_t)T * N));
        HIPC(hipMalloc(&d_fx, sizeof(double) * (size_t)nx * b.Ns));
        HIPC(hipMalloc(&d_xb, sizeof(double) * (size_t)T * M * nx));
        HIPC(hipMalloc(&d_idx, sizeof(int64_t) * (size_t)T * M));
        HIPC(hipMalloc(&d_u, sizeof(double) * (size_t)T * (b.nu > 0 ? b.nu : 1)));
        HIPC(hipMemcpyAsync(d_xf, xf, sizeof(double) * (size_t)T * N * nx, hipMemcpyHostToDevice, b.stream));
        HIPC(hipMemcpyAsync(d_wf, wf, sizeof(double) * (size_t)T * N, hipMemcpyHostToDevice, b.stream));
        if (b.nu > 0) HIPC(hipMemcpyAsync(d_u, U, sizeof(double) * (size_t)T * b.nu, hipMemcpyHostToDevice, b.stream));
        HIPC(hipMemcpyAsync(d_xb + (size_t)(T - 1) * M * nx, xb + (size_t)(T - 1) * M * nx, sizeof(double) * M * nx, hipMemcpyHostToDevice, b.stream));
        HIPC(hipMemsetAsync(d_idx, 0, sizeof(int64_t) * (size_t)T * M, b.stream));
        BankDev d = b.dev();
        HIPC(hipEventRecord(b.ev_run0, b.stream));
        for (int64_t t = T - 2; t >= 0; --t) {
            SmoothArgs a{};
            a.xf_t = d_xf + (size_t)t * N * nx; a.wf_t = d_wf + (size_t)t * N;
            a.u = b.nu > 0 ? d_u + t * b.nu : nullptr; a.t = (double)t * b.cfg.model.Ts;
            a.fx = d_fx; a.xb_next = d_xb + (size_t)(t + 1) * M * nx; a.xb_t = d_xb + (size_t)t * M * nx;
            a.idx_t = d_idx + (size_t)t * M; a.M = (int32_t)M; a.step = (uint32_t)t;
            HIPC(launch_smooth_fx(d, a, b.stream));
            HIPC(launch_smooth_draw(d, a, b.stream));
        }
        HIPC(hipEventRecord(b.ev_run1, b.stream));
        HIPC(hipMemcpyAsync(xb, d_xb, sizeof(double) * (size_t)(T - 1) * M * nx, hipMemcpyDeviceToHost, b.stream));
        std::vector<int64_t> hidx;
        if (idx) {
            hidx.resize((size_t)(T - 1) * M);
            HIPC(hipMemcpyAsync(hidx.data(), d_idx, sizeof(int64_t) * (size_t)(T - 1) * M, hipMemcpyDeviceToHost, b.stream));
        }
        HIPC(hipStreamSynchronize(b.stream));
        if (idx) memcpy(idx, hidx.data(), sizeof(int64_t) * (size_t)(T - 1) * M);
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, b.ev_run0, b.ev_run1));
        b.last_run_ms = ms;
        return LLPF_OK;
    };
    const int rc = body();
    hipFree(d_xf); hipFree(d_wf); hipFree(d_fx); hipFree(d_xb); hipFree(d_idx); hipFree(d_u);
    return rc;
}

// ---- accessors ----------------------------------------------------------------------------------
static int bank_get_particles(Bank& b, double* dst) {
    CHK(use_device(b));
    BankDev d = b.dev();
    HIPC(launch_soa2aos(d, b.d_x[b.cur], b.d_tmp, b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(double) * (size_t)b.F * b.N * b.nx, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}
static int bank_get_w(Bank& b, double* dst, bool expw) {
    CHK(use_device(b));
    if (expw && b.we_is_lambda) {     // after an aux predict! the reference's `we` holds lambda (src/filtering.jl:200-203)
        HIPC(hipMemcpy2DAsync(dst, sizeof(double) * b.N, b.d_lam, sizeof(double) * b.Ns, sizeof(double) * b.N, b.F,
                              hipMemcpyDeviceToHost, b.stream));
        HIPC(hipStreamSynchronize(b.stream));
        return LLPF_OK;
    }
    BankDev d = b.dev();
    HIPC(launch_materialize(d, expw ? nullptr : b.d_tmp, expw ? b.d_tmp : nullptr, b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(double) * (size_t)b.F * b.N, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

static int bank_set_weights(Bank& b, const double* w) {
    CHK(use_device(b));
    b.aux_pending = false; b.we_is_lambda = false;
    std::vector<double> stage((size_t)b.F * b.Ns, -INFINITY);
    for (int f = 0; f < b.F; ++f) memcpy(stage.data() + (size_t)f * b.Ns, w + (size_t)f * b.N, sizeof(double) * b.N);
    HIPC(hipMemcpyAsync(b.d_w, stage.data(), sizeof(double) * stage.size(), hipMemcpyHostToDevice, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    for (auto& s : h) { s.uniform = 0; s.norm_pending = 0; s.status = 0; }
    CHK(scal_upload(b, h));
    HIPC(hipMemsetAsync(b.d_acc, 0, sizeof(uint64_t) * (size_t)b.F * ACC_WORDS, b.stream));
    HIPC(hipMemsetAsync(b.d_tileq, 0, sizeof(uint64_t) * (size_t)ACC_NSLOT * b.F * b.P2, b.stream));
    b.parity = 0;
    BankDev d = b.dev();
    HIPC(launch_max(d, b.parity, b.stream));
    HIPC(launch_norm(d, b.parity, 0, 1, b.n_predict, 0, 0, 0, b.stream));
    ResArgs ra{};
    ra.mode = RES_FINALIZE; ra.parity = b.parity; ra.M = (int32_t)b.N; ra.keep_norm = 1; ra.fast_head = 0;
    HIPC(launch_resample(d, ra, b.stream));
    b.parity = (b.parity + 1) % ACC_NSLOT;
    CHK(scal_download(b, h));
    return check_status(b, h);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* llpf_last_error(void) { return g_err.c_str(); }

int llpf_version(int32_t* major, int32_t* minor) {
    if (major) *major = LLPF_VERSION_MAJOR;
    if (minor) *minor = LLPF_VERSION_MINOR;
    return LLPF_OK;
}

int llpf_device_count(int32_t* n) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    if (n) *n = c;
    return LLPF_OK;
}

int llpf_create(const llpf_config* cfg, llpf_filter** out) {
    if (!out) return fail(LLPF_ERR_ARG, "null out pointer");
    *out = nullptr;
    llpf_filter* f = new (std::nothrow) llpf_filter();
    if (!f) return fail(LLPF_ERR_ALLOC, "out of host memory");
    int rc = bank_create(cfg, nullptr, 1, f->bank);
    if (rc != LLPF_OK) { free_bank(f->bank); delete f; return rc; }
    *out = f;
    return LLPF_OK;
}
int llpf_destroy(llpf_filter* f) {
    if (!f) return LLPF_OK;
    free_bank(f->bank);
    delete f;
    return LLPF_OK;
}
#define NEEDF(f) if (!(f)) return fail(LLPF_ERR_ARG, "null handle")

int llpf_reset(llpf_filter* f) { NEEDF(f); CHK(use_device(f->bank)); return bank_init_particles(f->bank, true); }

static int bank_seed(Bank& b, uint64_t seed) {
    CHK(use_device(b));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    set_keys(b, h, seed);
    return scal_upload(b, h);
}
int llpf_seed(llpf_filter* f, uint64_t seed) { NEEDF(f); return bank_seed(f->bank, seed); }

int llpf_correct(llpf_filter* f, const double* u, const double* y, double t, double* ll) {
    NEEDF(f);
    double l = 0.0;
    int rc = bank_correct(f->bank, u, y, t, &l);
    if (ll) *ll = l;
    return rc;
}
int llpf_predict(llpf_filter* f, const double* u, double t) { NEEDF(f); return bank_predict(f->bank, u, t); }
int llpf_update(llpf_filter* f, const double* u, const double* y, double t, double* ll) {
    NEEDF(f);
    int rc = llpf_correct(f, u, y, t, ll);
    if (rc != LLPF_OK) return rc;
    return bank_predict(f->bank, u, t);
}

int llpf_run(llpf_filter* f, const double* U, const double* Y, int64_t T, double t_index0,
             double* ll_total, const llpf_run_outputs* o) {
    NEEDF(f);
    double lt = 0.0;
    int rc = bank_run(f->bank, U, Y, T, t_index0, &lt, o ? o->ll_steps : nullptr, o ? o->xmean : nullptr,
                      o ? o->x_hist : nullptr, o ? o->w_hist : nullptr, o ? o->we_hist : nullptr);
    if (ll_total) *ll_total = lt;
    return rc;
}

int llpf_aux_correct(llpf_filter* f, double* ll) {
    NEEDF(f);
    double l = 0.0;
    int rc = bank_aux_correct(f->bank, &l, AuxOuts{}, 0);
    if (ll) *ll = l;
    return rc;
}
int llpf_aux_predict(llpf_filter* f, const double* u, const double* y1, double t) { NEEDF(f); return bank_aux_predict(f->bank, u, y1, t); }
int llpf_aux_update(llpf_filter* f, const double* u, const double* y1, double t, double* ll) {
    NEEDF(f);
    int rc = llpf_aux_correct(f, ll);
    if (rc != LLPF_OK) return rc;
    return bank_aux_predict(f->bank, u, y1, t);
}
int llpf_aux_run(llpf_filter* f, const double* U, const double* Y, int64_t T, int32_t mode,
                 double* ll_total, const llpf_run_outputs* o) {
    NEEDF(f);
    double lt = 0.0;
    int rc = bank_aux_run(f->bank, U, Y, T, mode, &lt, o ? o->ll_steps : nullptr, o ? o->xmean : nullptr,
                          o ? o->x_hist : nullptr, o ? o->w_hist : nullptr, o ? o->we_hist : nullptr);
    if (ll_total) *ll_total = lt;
    return rc;
}
int llpf_bank_aux_run(llpf_bank* b, const double* U, const double* Y, int64_t T, int32_t mode,
                      double* ll_total, double* ll_steps) {
    if (!b) return fail(LLPF_ERR_ARG, "null bank");
    return bank_aux_run(b->bank, U, Y, T, mode, ll_total, ll_steps, nullptr, nullptr, nullptr, nullptr);
}

int llpf_rb_get_covariance(llpf_filter* f, double* R) {
    NEEDF(f);
    if (!is_rb(f->bank)) return fail(LLPF_ERR_ARG, "not a Rao-Blackwellized filter");
    const int nl = f->bank.nx - f->bank.cfg.model.nxn;
    for (int i = 0; i < nl * nl; ++i) R[i] = f->bank.rb[0].R[i];
    return LLPF_OK;
}

int llpf_smooth(llpf_filter* f, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
                const double* wef, double* xb, int64_t* idx) {
    NEEDF(f);
    return bank_smooth(f->bank, M, U, T, xf, wf, wef, xb, idx);
}

int llpf_num_particles(const llpf_filter* f, int64_t* n) { NEEDF(f); if (n) *n = f->bank.N; return LLPF_OK; }
int llpf_index(const llpf_filter* f, int64_t* t) { NEEDF(f); if (t) *t = f->bank.t_index; return LLPF_OK; }
int llpf_set_index(llpf_filter* f, int64_t t) { NEEDF(f); f->bank.t_index = t; return LLPF_OK; }
int llpf_get_particles(llpf_filter* f, double* dst) { NEEDF(f); return bank_get_particles(f->bank, dst); }
int llpf_get_weights(llpf_filter* f, double* dst) { NEEDF(f); return bank_get_w(f->bank, dst, false); }
int llpf_get_expweights(llpf_filter* f, double* dst) { NEEDF(f); return bank_get_w(f->bank, dst, true); }

int llpf_get_ancestors(llpf_filter* f, int64_t* dst) {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    HIPC(launch_anc64(b.dev(), reinterpret_cast<int64_t*>(b.d_tmp), b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(int64_t) * b.N, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}
int llpf_get_bins(llpf_filter* f, double* dst) {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    BankDev d = b.dev();
    ResArgs ra{};
    ra.mode = RES_RESAMPLE; ra.step = b.n_predict; ra.M = (int32_t)b.N; ra.anc_out = b.d_anc;
    ra.parity = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT;
    ra.bins_out = b.d_tmp; ra.only_bins = 1; ra.force = 1;
    HIPC(launch_resample(d, ra, b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(double) * b.N, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}
int llpf_set_particles(llpf_filter* f, const double* src) {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    HIPC(hipMemcpyAsync(b.d_tmp, src, sizeof(double) * b.N * b.nx, hipMemcpyHostToDevice, b.stream));
    HIPC(launch_aos2soa(b.dev(), b.d_tmp, b.d_x[b.cur], b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}
int llpf_set_weights(llpf_filter* f, const double* w) { NEEDF(f); return bank_set_weights(f->bank, w); }

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
int llpf_effective_particles(llpf_filter* f, double* ess) {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, true));
    if (ess) *ess = s.ess;
    return LLPF_OK;
}
int llpf_shouldresample(llpf_filter* f, int32_t* yes) {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, true));
    if (yes) *yes = s.do_resample;
    return LLPF_OK;
}
int llpf_last_resampled(llpf_filter* f, int32_t* yes) {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, false));
    if (yes) *yes = s.last_resampled;
    return LLPF_OK;
}
int llpf_maxw(llpf_filter* f, double* maxw) {
    NEEDF(f);
    FilterScal s;
    CHK(scal0(f, &s, false));
    if (maxw) *maxw = s.mtrue;
    return LLPF_OK;
}
int llpf_weighted_mean(llpf_filter* f, double* xh) {
    NEEDF(f);
    Bank& b = f->bank;
    CHK(use_device(b));
    HIPC(launch_wmean(b.dev(), b.d_tmp, b.stream));
    HIPC(hipMemcpyAsync(xh, b.d_tmp, sizeof(double) * b.nx, hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}
int llpf_resample_count(llpf_filter* f, int64_t* n) { NEEDF(f); if (n) *n = f->bank.run_resamples; return LLPF_OK; }
int llpf_last_run_ms(llpf_filter* f, double* ms) { NEEDF(f); if (ms) *ms = f->bank.last_run_ms; return LLPF_OK; }

static int set_prof(Bank& b, int on) {
    b.profiling = on != 0;
    for (int i = 0; i < LLPF_PROF_CLASSES; ++i) { b.prof_ms[i] = 0.0; b.prof_n[i] = 0; }
    return LLPF_OK;
}
static int get_prof(Bank& b, double* ms, int64_t* n) {
    for (int i = 0; i < LLPF_PROF_CLASSES; ++i) { if (ms) ms[i] = b.prof_ms[i]; if (n) n[i] = b.prof_n[i]; }
    return LLPF_OK;
}
int llpf_set_profiling(llpf_filter* f, int32_t on) { NEEDF(f); return set_prof(f->bank, on); }
int llpf_get_profile(llpf_filter* f, double* ms, int64_t* n) { NEEDF(f); return get_prof(f->bank, ms, n); }

// ---- banks ---------------------------------------------------------------------------------------
int llpf_bank_create(const llpf_config* base, const llpf_model* models, int32_t n_filters, llpf_bank** out) {
    if (!out) return fail(LLPF_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (!models) return fail(LLPF_ERR_ARG, "null models");
    llpf_bank* b = new (std::nothrow) llpf_bank();
    if (!b) return fail(LLPF_ERR_ALLOC, "out of host memory");
    int rc = bank_create(base, models, n_filters, b->bank);
    if (rc != LLPF_OK) { free_bank(b->bank); delete b; return rc; }
    *out = b;
    return LLPF_OK;
}
int llpf_bank_destroy(llpf_bank* b) {
    if (!b) return LLPF_OK;
    free_bank(b->bank);
    delete b;
    return LLPF_OK;
}
int llpf_bank_reset(llpf_bank* b) { NEEDF(b); CHK(use_device(b->bank)); return bank_init_particles(b->bank, true); }
int llpf_bank_seed(llpf_bank* b, uint64_t seed) { NEEDF(b); return bank_seed(b->bank, seed); }
int llpf_bank_run(llpf_bank* b, const double* U, const double* Y, int64_t T, double t_index0,
                  double* ll_total, double* ll_steps) {
    NEEDF(b);
    return bank_run(b->bank, U, Y, T, t_index0, ll_total, ll_steps, nullptr, nullptr, nullptr, nullptr);
}
int llpf_bank_set_profiling(llpf_bank* b, int32_t on) { NEEDF(b); return set_prof(b->bank, on); }
int llpf_bank_get_profile(llpf_bank* b, double* ms, int64_t* n) { NEEDF(b); return get_prof(b->bank, ms, n); }
int llpf_bank_resample_count(llpf_bank* b, int64_t* n) { NEEDF(b); if (n) *n = b->bank.run_resamples; return LLPF_OK; }
int llpf_bank_last_run_ms(llpf_bank* b, double* ms) { NEEDF(b); if (ms) *ms = b->bank.last_run_ms; return LLPF_OK; }

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

int llpf_logsumexp(int32_t device, double* w, double* we, int64_t n, double* ll) {
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
}

int llpf_resample(int32_t device, int32_t strategy, const double* we, int64_t n, int64_t m, const double* U, int64_t* j) {
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
}

int llpf_resample_uniforms(int32_t strategy, int64_t m, uint64_t seed, uint32_t step, double* u) {
    if (!u) return fail(LLPF_ERR_ARG, "null output");
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (strategy == LLPF_RESAMPLE_SYSTEMATIC) u[0] = llpf_uniform_step(step, LLPF_STREAM_RESAMPLE, k0, k1);
    else for (int64_t i = 0; i < m; ++i) u[i] = llpf_uniform_idx((uint32_t)i, step, LLPF_STREAM_STRATIFY, k0, k1);
    return LLPF_OK;
}

// ---- device self-tests of the shared primitives ---------------------------------------------------
int llpf_selftest_math(int32_t device, int32_t which, const double* in, double* out, int64_t n) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible");
    HIPC(hipSetDevice(device));
    double *di = nullptr, *dout = nullptr;
    HIPC(hipMalloc(&di, sizeof(double) * n));
    HIPC(hipMalloc(&dout, sizeof(double) * n));
    HIPC(hipMemcpy(di, in, sizeof(double) * n, hipMemcpyHostToDevice));
    HIPC(launch_selftest_math(which, di, dout, n, nullptr));
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    hipFree(di);
    hipFree(dout);
    return LLPF_OK;
}
int llpf_selftest_normals(int32_t device, uint64_t seed, uint32_t step, uint32_t stream, int32_t nd, double* out, int64_t n) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible");
    if (nd < 1 || nd > MAXD) return fail(LLPF_ERR_ARG, "nd out of range");
    HIPC(hipSetDevice(device));
    double* dout = nullptr;
    HIPC(hipMalloc(&dout, sizeof(double) * n * nd));
    HIPC(launch_selftest_normals((uint32_t)seed, (uint32_t)(seed >> 32), step, stream, nd, dout, n, nullptr));
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(out, dout, sizeof(double) * n * nd, hipMemcpyDeviceToHost));
    hipFree(dout);
    return LLPF_OK;
}

}  // extern "C"
