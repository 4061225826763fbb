// host/bank.hpp — the Bank (F filters on one device / stream), creation, initialisation, profiling helpers.  Part of capi.hip (one translation unit).
// ------------------------------------------------------------------------------------------------
// Bank: F independent filters of N particles on one device / one stream
// ------------------------------------------------------------------------------------------------
struct Bank {
    llpf_config cfg{};
    int F = 0;
    int64_t N = 0, Ns = 0;
    int nx = 0, nu = 0, ny = 0, P1 = 0, P2 = 0;
    int xrows = 0;                   // rows of a particle plane: nx, or xn + xl + packed R for LLPF_MODEL_RB_BILINEAR
    int nxp = 0;                     // dimension of a particle as the accessors see it: nx, or nxn + nxl for LLPF_MODEL_RB_BILINEAR (RBParticle indexes like [xn; xl])
    int device = 0;
    hipStream_t stream = nullptr;
    void* d_pool = nullptr;          // the one allocation the members below (up to d_tmp) point into
    ModelD* d_models = nullptr;
    FilterScal* d_scal = nullptr;
    double* d_x[2] = {nullptr, nullptr};
    int cur = 0;
    double* d_w = nullptr;           // the weights (the CURRENT ones: a split-schedule run of fused steps alternates between this and d_w_spare, host/run.hpp)
    double* d_w_spare = nullptr;     // second weight buffer, allocated by the first such run
    double* d_w_alloc = nullptr;     // the allocation behind whichever of the two is not part of the pool
    bool w_pingpong = false;         // inside such a run: the fused kernel writes the weights it forms to the other buffer
    int32_t* d_anc = nullptr;
    uint64_t* d_acc = nullptr;
    uint64_t* d_quanta[2] = {nullptr, nullptr};
    int qcur = 0;                    // quanta buffer that holds the quanta of the current weights
    uint64_t* d_tileq = nullptr;
    void* d_wq = nullptr;            // weighted_quantile: selection state (k_quantile.hip), the exp-weights [N], the probabilities [cap_wqp]
    double *d_wq_we = nullptr, *d_wq_p = nullptr, *d_xquant = nullptr;
    size_t cap_wqp = 0, cap_xq = 0;
    uint64_t *d_tpre = nullptr, *d_gsum = nullptr;      // filters above 1024 tiles: k_tile_prefix (kernels/resample.hpp)
    uint32_t* d_flag = nullptr;
    int64_t last_run_launches = 0, last_run_fx_steps = 0;
    double last_run_surv = -1.0;
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
    double surv_frac = -1.0;          // distinct ancestors per predict! / N over the last run of a model that can take the source-side form (-1: none yet)
    bool use_fx = true;               //   ... and the form the next run takes (hysteresis: host/run.hpp)
    unsigned long long* d_surv = nullptr;   // [F][P2][4] survivor counters of a run (BankDev::surv)
    int32_t* d_mark = nullptr;        // [F][Ns] run-start marks / [F][nx][Ns] f(x_j): resampling with source-side dynamics (kernels/resfx.hpp),
    double* d_fxs = nullptr;          //   allocated on first use (ensure_fx)
    bool aux_pending = false;         // w holds lambda - log N of an aux predict!; their exp-sums wait in slot (parity+2)%3
    bool we_is_lambda = false;        // expweights(pf) returns lambda until the next correct! (the reference keeps it in `we`)
    int parity = 0;                  // accumulator slot (0..2) the NEXT weighting kernel writes (engine.hpp ACC_NSLOT)
    double* d_uy = nullptr;          // staging for single-step u / y (2 * MAXD)
    double* d_U = nullptr;           // resident inputs of a run
    double* d_Y = nullptr;
    size_t capU = 0, capY = 0;
    double* d_ll_steps = nullptr;
    double* d_xmean = nullptr;
    double* d_xcov = nullptr;         // [T][nx*nx] + a mean: the xcov output of a run
    size_t cap_ll = 0, cap_xm = 0, cap_xc = 0;
    double* d_tmp = nullptr;         // F*N*max(nx,1) doubles (also reinterpreted as int64 / double staging)
    uint64_t seed = 0;
    uint64_t key_off = 0, key_stride = 1;   // filter f of this bank is filter key_off + f * key_stride of a sharded sweep (llpf_mbank): its
                                            // Philox key is seed + that global index, so results do not depend on the sharding
    uint32_t n_reset = 0, n_predict = 0;
    uint32_t step_base = 0;           // value of FilterScal::step_base on the device: kernels add it to the step arguments
    // Captured run loops (hipGraph): a chain of T dependent launches replays ~1 us per launch faster than it enqueues
    // (tools/launch_floor.hip: 1.6 vs 2.8 us per dependent empty launch).  Keyed by everything a launch argument depends on.
    struct RunGraph {
        int64_t T; double t_index0; int par0, cur0, qcur0, flags, np_parity;
        const void *dU, *dY, *dll, *dxm, *dxc, *drb, *dxq, *dqp, *dw, *dws;      // every device buffer a captured launch addresses that ensure() may reallocate
        int nq;
        uint64_t yhash;
        hipGraphExec_t exec;
        bool same(const RunGraph& o) const {
            return T == o.T && t_index0 == o.t_index0 && par0 == o.par0 && cur0 == o.cur0 && qcur0 == o.qcur0 && flags == o.flags &&
                   np_parity == o.np_parity && dU == o.dU && dY == o.dY && dll == o.dll && dxm == o.dxm && dxc == o.dxc && drb == o.drb && dxq == o.dxq && dqp == o.dqp && dw == o.dw && dws == o.dws && nq == o.nq && yhash == o.yhash;
        }
    };
    std::vector<RunGraph> graphs;
    double* d_hist = nullptr;         // device staging of the forward_trajectory history ([T][N][nx] x, [T][N] w, [T][N] we)
    size_t cap_hist = 0;
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
        b.w = d_w; b.w_next = (w_pingpong && d_w_spare) ? d_w_spare : d_w; b.anc = d_anc; b.acc = d_acc; b.quanta = d_quanta[qcur]; b.quanta_next = d_quanta[qcur ^ 1]; b.tileq = d_tileq; b.tpre = d_tpre; b.gsum = d_gsum;
        b.bank_flag = d_flag; b.xmpart = d_xmpart; b.lam = d_lam; b.rtile = d_rtile; b.mark = d_mark; b.fxs = d_fxs; b.surv = d_surv;
        b.anc_slot = (int32_t)(n_predict & 1u);
        b.pad0 = (cfg.model.model_id == LLPF_MODEL_RB_BILINEAR) ? (cfg.model.rb.nxl | (cfg.model.rb.fn_kind << 8))
                 : (cfg.model.model_id == LLPF_MODEL_RB_LINEAR ? cfg.model.nxn : 0);
        b.xrows = xrows; b.pad1 = 0;
        return b;
    }
    // for the layout conversions of the accessors: the first nxp rows of the plane are the particle [xn; xl]
    BankDev devp() const { BankDev b = dev(); b.nx = nxp; return b; }
};

// Philox step argument of a launch issued now (relative to the base the device adds)
static inline uint32_t rel_step(const Bank& b) { return b.n_predict - b.step_base; }

struct llpf_filter { Bank bank; };
struct llpf_bank { Bank bank; };

static int use_device(const Bank& b) {
    HIPC(hipSetDevice(b.device));
    return LLPF_OK;
}

static void free_bank(Bank& b) {
    hipSetDevice(b.device);
    if (b.stream) hipStreamSynchronize(b.stream);
    for (auto& g : b.graphs) if (g.exec) hipGraphExecDestroy(g.exec);
    b.graphs.clear();
    hipFree(b.d_pool);               // models, scal, x, w, anc, acc, quanta, tileq, flag, rtile, rb, uy, tmp
    hipFree(b.d_w_alloc);
    hipFree(b.d_wq); hipFree(b.d_wq_we); hipFree(b.d_wq_p); hipFree(b.d_xquant);
    hipFree(b.d_xmpart); hipFree(b.d_lam); hipFree(b.d_surv); hipFree(b.d_mark); hipFree(b.d_fxs); hipFree(b.d_rbseq); hipFree(b.d_hist); hipFree(b.d_U); hipFree(b.d_Y);
    hipFree(b.d_ll_steps); hipFree(b.d_xmean); hipFree(b.d_xcov);
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
    b.step_base = 0;
    for (int f = 0; f < b.F; ++f) {
        h[f].step_base = 0;
        const uint64_t s = seed + b.key_off + (uint64_t)f * b.key_stride;
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

// the scratch of the resampling with source-side dynamics: marks (zero between timesteps) and the plane of f(x_j)
// per-block parts of the weighted mean, one set per accumulator slot: [ACC_NSLOT][F][P1][MAXD] — 400 MB for a filter near 2^29 particles,
// so it is not part of every handle's pool but allocated by the first run that asks for the means (kernels touch it under want_xmean only)
static int ensure_xmpart(Bank& b) {
    if (b.d_xmpart) return LLPF_OK;
    const size_t n = (size_t)ACC_NSLOT * b.F * b.P1 * MAXD;
    HIPC(hipMalloc(&b.d_xmpart, sizeof(double) * n));
    HIPC(hipMemsetAsync(b.d_xmpart, 0, sizeof(double) * n, b.stream));
    return LLPF_OK;
}

// weighted_quantile state: allocated by the first call that asks for quantiles; p [nq] uploaded
static int ensure_wq(Bank& b, const double* p, int nq) {
    if (!b.d_wq) HIPC(hipMalloc(&b.d_wq, wquantile_workspace_bytes(b.nx)));
    if (!b.d_wq_we) HIPC(hipMalloc(&b.d_wq_we, sizeof(double) * (size_t)b.N));
    if (b.cap_wqp < (size_t)nq) {
        if (b.d_wq_p) hipFree(b.d_wq_p);
        b.d_wq_p = nullptr; b.cap_wqp = 0;
        HIPC(hipMalloc(&b.d_wq_p, sizeof(double) * (size_t)nq));
        b.cap_wqp = (size_t)nq;
    }
    HIPC(hipMemcpyAsync(b.d_wq_p, p, sizeof(double) * (size_t)nq, hipMemcpyHostToDevice, b.stream));
    return LLPF_OK;
}

static int ensure_fx(Bank& b) {
    if (b.d_mark && b.d_fxs) return LLPF_OK;
    const size_t FN = (size_t)b.F * b.Ns;
    if (!b.d_mark) { HIPC(hipMalloc(&b.d_mark, sizeof(int32_t) * FN)); HIPC(hipMemsetAsync(b.d_mark, 0, sizeof(int32_t) * FN, b.stream)); }
    if (!b.d_fxs) { HIPC(hipMalloc(&b.d_fxs, sizeof(double) * FN * b.nx)); HIPC(hipMemsetAsync(b.d_fxs, 0, sizeof(double) * FN * b.nx, b.stream)); }
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

static bool is_rbfull(const Bank& b) { return b.cfg.model.model_id == LLPF_MODEL_RB_BILINEAR; }
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
        for (int p = 0; p < ACC_NSLOT; ++p) { s.off_slot[p] = 0.0; s.u_slot[p] = 0.0; s.e2v_slot[p] = 0; s.exact_slot[p] = 0; }
        s.K = llpf_qbits(b.N);
        if (!is_reset) { s.anc_ident_s[0] = s.anc_ident_s[1] = 1; s.last_resampled = 0; s.resample_count = 0; s.ll_total = 0.0; }
    }
    CHK(scal_upload(b, h));
    HIPC(hipMemsetAsync(b.d_acc, 0, sizeof(uint64_t) * (size_t)b.F * ACC_WORDS, b.stream));
    HIPC(hipMemsetAsync(b.d_tileq, 0, sizeof(uint64_t) * (size_t)ACC_NSLOT * b.F * b.P2, b.stream));
    HIPC(hipMemsetAsync(b.d_flag, 0, sizeof(uint32_t) * 4, b.stream));
    b.parity = 0;
    if (b.cfg.model.model_id >= LLPF_MODEL_USER_BASE && (jit_model_traits(b.cfg.model.model_id) & LLPF_TRAIT_INITIAL) > 0) {
        // an initial density of the model's own: prepare() sees u = 0 — whatever the single-step verbs staged in d_uy last (a reset! after
        // predict!/correct! must equal the reset! of a fresh handle)
        HIPC(hipMemsetAsync(b.d_uy, 0, sizeof(double) * 4 * MAXD, b.stream));
        HIPC(launch_init_user(d, b.d_uy, b.n_reset, is_reset ? 0 : 1, b.stream));
    } else
        HIPC(launch_init(d, b.n_reset, is_reset ? 0 : 1, b.stream));
    if (is_rbfull(b)) HIPC(launch_rbfull_init(d, b.stream));
    b.n_reset++;
    b.t_index = is_reset ? 1 : 0;
    HIPC(hipStreamSynchronize(b.stream));
    return LLPF_OK;
}

static int bank_create(const llpf_config* cfg, const llpf_model* models, int F, Bank& b, uint64_t key_off = 0, uint64_t key_stride = 1) {
    if (!cfg) return fail(LLPF_ERR_ARG, "null config");
    if (cfg->struct_size != sizeof(llpf_config)) return fail(LLPF_ERR_ARG, "llpf_config.struct_size mismatch (ABI)");
    if (F < 1) return fail(LLPF_ERR_ARG, "n_filters must be >= 1");
    test_throw("create");
    const llpf_model& m0 = models ? models[0] : cfg->model;
    if (m0.model_id == LLPF_MODEL_LINEAR_GAUSSIAN && (m0.nx > 4 || m0.ny > 4) && m0.nx >= 1 && m0.nx <= MAXD && m0.ny >= 1 && m0.ny <= MAXD) {
        // the linear-Gaussian model above the precompiled dimensions: LinGauss<nx, ny> compiled on demand (kernels/jit.hpp), after
        // which the bank is a bank of that run-time compiled model
        std::string err;
        const int id = jit_builtin_lg(m0.nx, m0.ny, err);
        if (id < 0) return fail(LLPF_ERR_HIP, "linear-Gaussian model at nx = " + std::to_string(m0.nx) + ", ny = " + std::to_string(m0.ny) + ": " + err);
        llpf_config c2 = *cfg;
        c2.model.model_id = id;
        std::vector<llpf_model> mm;
        if (models) {
            mm.assign(models, models + F);
            for (int f = 0; f < F; ++f) {
                if (mm[f].model_id != LLPF_MODEL_LINEAR_GAUSSIAN) return fail(LLPF_ERR_ARG, "all filters of a bank must share model id and dimensions");
                mm[f].model_id = id;
            }
        }
        return bank_create(&c2, models ? mm.data() : nullptr, F, b, key_off, key_stride);
    }
    if (cfg->n_particles < 1 || cfg->n_particles > ((int64_t)1 << 29) - 2 * TILE)   // 32-bit byte offsets into a particle plane
        return fail(LLPF_ERR_ARG, "n_particles must be in 1..2^29-2048");
    if (m0.nx < 1 || m0.nx > MAXD || m0.ny < 1 || m0.ny > MAXD || m0.nu < 0 || m0.nu > MAXU) return fail(LLPF_ERR_ARG, "bad dimensions (states and outputs 1.." + std::to_string(MAXD) + ", inputs 0.." + std::to_string(MAXU) + ")");
    if (!step_supported(m0.model_id, m0.nx, m0.ny))
        return fail(LLPF_ERR_ARG, "no kernel for this model / dimension (linear-Gaussian nx, ny in 1..16; quad-tank 4 / 2; Rao-Blackwellized: see llpf.h)");
    if (m0.model_id == LLPF_MODEL_RB_BILINEAR) {
        if ((uint64_t)rbfull_rows(m0.nx, m0.rb.nxl) * (uint64_t)((cfg->n_particles + TILE - 1) / TILE * TILE) * 8u >= ((uint64_t)1 << 32))
            return fail(LLPF_ERR_ARG, "LLPF_MODEL_RB_BILINEAR: the planes of one filter (rows x particles x 8 bytes) must span less than 4 GB");
        if (!rbfull_supported(m0.rb.fn_kind, m0.nx, m0.rb.nxl, m0.ny))
            return fail(LLPF_ERR_ARG, "LLPF_MODEL_RB_BILINEAR: nxn in 1..4, nxl in 1..8, ny in 1..4 (quad-tank nonlinear part: nxn = 4, ny = 2)");
        const bool pre = (m0.rb.fn_kind == 1) ? (m0.rb.nxl == 8) : ((m0.nx == 1 && m0.rb.nxl == 2 && m0.ny == 1) || (m0.nx == 2 && m0.rb.nxl == 2 && m0.ny == 2) || (m0.nx == 4 && m0.rb.nxl == 8 && m0.ny == 2));
        if (!pre) {     // a shape the library was not precompiled for: k_rbfull compiled on demand (kernels/jit.hpp), cached
            std::string err;
            if (jit_prepare_rbfull(m0.rb.fn_kind, m0.nx, m0.rb.nxl, m0.ny, err) != 0) return fail(LLPF_ERR_HIP, "LLPF_MODEL_RB_BILINEAR shape: " + err);
        }
    }
    if (cfg->resampling_strategy != LLPF_RESAMPLE_SYSTEMATIC && cfg->resampling_strategy != LLPF_RESAMPLE_STRATIFIED &&
        cfg->resampling_strategy != LLPF_RESAMPLE_RESIDUAL)
        return fail(LLPF_ERR_ARG, "resampling_strategy must be systematic, stratified or residual");
    if (!(cfg->resample_threshold >= 0.0 && cfg->resample_threshold <= 1.0)) return fail(LLPF_ERR_ARG, "resample_threshold must be in [0,1]");
    if (cfg->filter_kind != LLPF_PARTICLE_FILTER && cfg->filter_kind != LLPF_ADVANCED_PARTICLE_FILTER)
        return fail(LLPF_ERR_ARG, "filter_kind must be LLPF_PARTICLE_FILTER or LLPF_ADVANCED_PARTICLE_FILTER");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(LLPF_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(LLPF_ERR_ARG, "device ordinal out of range");

    b.cfg = *cfg;
    b.cfg.model = m0;
    b.key_off = key_off; b.key_stride = key_stride;
    b.F = F;
    b.N = cfg->n_particles;
    b.Ns = (b.N + TILE - 1) / TILE * TILE;
    b.nx = m0.nx; b.nu = m0.nu; b.ny = m0.ny;
    b.xrows = (m0.model_id == LLPF_MODEL_RB_BILINEAR) ? rbfull_rows(m0.nx, m0.rb.nxl) : b.nx;
    b.nxp = (m0.model_id == LLPF_MODEL_RB_BILINEAR) ? m0.nx + m0.rb.nxl : b.nx;
    b.P1 = (int)(b.Ns / BLOCK);          // entries of the per-block weighted-mean partials of one filter (finest block granularity)
    b.P2 = (int)(b.Ns / TILE);
    b.device = cfg->device;
    // replicas (models == NULL): one descriptor is prepared and uploaded, the device copies it F times (a bank of
    // thousands of Monte-Carlo replicas otherwise spends its construction on host copies of identical 8 KB structs)
    const bool replicas = (models == nullptr) && m0.model_id != LLPF_MODEL_RB_LINEAR;
    const int FH = replicas ? 1 : F;
    std::vector<ModelD> hm(FH);
    b.hmodels.resize(FH);
    for (int f = 0; f < FH; ++f) {
        const llpf_model& mf = models ? models[f] : cfg->model;
        b.hmodels[f] = mf;
        if (mf.model_id != m0.model_id || mf.nx != m0.nx || mf.nu != m0.nu || mf.ny != m0.ny)
            return fail(LLPF_ERR_ARG, "all filters of a bank must share model id and dimensions");
        // the kernel instance (shape, nonlinear part), the rows of a particle plane and BankDev::pad0 are taken from models[0]
        if (m0.model_id == LLPF_MODEL_RB_BILINEAR && (mf.rb.nxl != m0.rb.nxl || mf.rb.fn_kind != m0.rb.fn_kind))
            return fail(LLPF_ERR_ARG, "all filters of a bank must share model id and dimensions (LLPF_MODEL_RB_BILINEAR: also rb.nxl and rb.fn_kind)");
        if (m0.model_id == LLPF_MODEL_RB_LINEAR && mf.nxn != m0.nxn)
            return fail(LLPF_ERR_ARG, "all filters of a bank must share model id and dimensions (LLPF_MODEL_RB_LINEAR: also nxn)");
        int rc = model_prepare(&mf, &hm[f]);
        if (rc) return fail(LLPF_ERR_ARG, "invalid density (covariance not positive definite or dimension mismatch), code " + std::to_string(rc));
    }
    HIPC(hipSetDevice(b.device));
    HIPC(hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking));
    const size_t FN = (size_t)F * b.Ns;
    {   // one device allocation for everything whose size is known here (a filter is often built per Monte-Carlo run or per
        // parameter candidate: 17 hipMalloc + as many memsets and hipFree cost more than a short run), zero-filled once
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
        const size_t o_models = take(sizeof(ModelD) * F), o_scal = take(sizeof(FilterScal) * F);
        const size_t o_x0 = take(sizeof(double) * FN * b.xrows), o_x1 = take(sizeof(double) * FN * b.xrows);
        const size_t o_w = take(sizeof(double) * FN), o_anc = take(sizeof(int32_t) * FN);
        const size_t o_acc = take(sizeof(uint64_t) * (size_t)F * ACC_WORDS);
        const size_t o_q0 = take(sizeof(uint64_t) * FN), o_q1 = take(sizeof(uint64_t) * FN);
        const size_t o_tileq = take(sizeof(uint64_t) * (size_t)ACC_NSLOT * F * b.P2), o_flag = take(sizeof(uint32_t) * 4);
        const size_t o_rtile = take(sizeof(uint64_t) * (size_t)F * 2 * b.P2);      // (xmpart: on the first run that asks for weighted means, ensure_xmpart)
        const size_t o_rb = take(m0.model_id == LLPF_MODEL_RB_LINEAR ? sizeof(RBStep) * 2 * (size_t)F : 0);
        const bool two_level = b.P2 > 4 * BLOCK;           // (kernels/resample.hpp: TQ_GROUP)
        const size_t o_tpre = take(two_level ? sizeof(uint64_t) * (size_t)F * b.P2 : 0), o_gsum = take(two_level ? sizeof(uint64_t) * (size_t)F * ((b.P2 + 4 * BLOCK - 1) / (4 * BLOCK)) : 0);
        const size_t o_uy = take(sizeof(double) * 4 * MAXD);
        const size_t o_tmp = take(sizeof(double) * (size_t)F * b.N * (b.nxp > 1 ? b.nxp : 1) + 64);
        HIPC(hipMalloc(&b.d_pool, off));
        HIPC(hipMemsetAsync(b.d_pool, 0, off, b.stream));
        char* base = static_cast<char*>(b.d_pool);
        b.d_models = reinterpret_cast<ModelD*>(base + o_models); b.d_scal = reinterpret_cast<FilterScal*>(base + o_scal);
        b.d_x[0] = reinterpret_cast<double*>(base + o_x0); b.d_x[1] = reinterpret_cast<double*>(base + o_x1);
        b.d_w = reinterpret_cast<double*>(base + o_w); b.d_anc = reinterpret_cast<int32_t*>(base + o_anc);
        b.d_acc = reinterpret_cast<uint64_t*>(base + o_acc);
        b.d_quanta[0] = reinterpret_cast<uint64_t*>(base + o_q0); b.d_quanta[1] = reinterpret_cast<uint64_t*>(base + o_q1);
        b.d_tileq = reinterpret_cast<uint64_t*>(base + o_tileq); b.d_flag = reinterpret_cast<uint32_t*>(base + o_flag);
        b.d_rtile = reinterpret_cast<uint64_t*>(base + o_rtile);
        if (m0.model_id == LLPF_MODEL_RB_LINEAR) b.d_rb = reinterpret_cast<RBStep*>(base + o_rb);
        if (two_level) { b.d_tpre = reinterpret_cast<uint64_t*>(base + o_tpre); b.d_gsum = reinterpret_cast<uint64_t*>(base + o_gsum); }
        b.d_uy = reinterpret_cast<double*>(base + o_uy); b.d_tmp = reinterpret_cast<double*>(base + o_tmp);
    }
    if (m0.model_id == LLPF_MODEL_RB_LINEAR) {
        b.rb.resize(F);
        for (int f = 0; f < F; ++f) {                       // the inner KalmanFilter object: kf.x = d0.mu, kf.R = d0.Sigma
            double S0[16];
            gauss_cov_dense(&b.hmodels[f].linear_initial, S0);
            const int nl = m0.nx - m0.nxn;
            for (int i = 0; i < nl * nl; ++i) { b.rb[f].R[i] = S0[i]; b.rb[f].kfR[i] = S0[i]; }
            for (int i = 0; i < nl; ++i) b.rb[f].kfx[i] = b.hmodels[f].linear_initial.mu[i];
        }
    }
    HIPC(hipMemcpyAsync(b.d_models, hm.data(), sizeof(ModelD) * FH, hipMemcpyHostToDevice, b.stream));
    if (FH < F) HIPC(launch_replicate_models(b.d_models, F, b.stream));
    // a run-time compiled model with a likelihood of its own declares the bound the normalisation works against (d_uy is zero-filled)
    if (m0.model_id >= LLPF_MODEL_USER_BASE) HIPC(launch_user_bound(m0.model_id, b.d_models, F, b.d_uy, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    HIPC(hipEventCreate(&b.ev_run0));
    HIPC(hipEventCreate(&b.ev_run1));
    std::vector<FilterScal> h;
    CHK(scal_download(b, h));
    set_keys(b, h, cfg->seed);
    CHK(scal_upload(b, h));
    return bank_init_particles(b, false);
}

// New parameters for an existing bank: the reference's `filter_from_parameters(theta, pf)` of log_likelihood_fun / metropolis
// (src/smoothing.jl:266-283, 311-330) hands the old filter back so that nothing is allocated per candidate.  Same model family and
// dimensions; the descriptors are prepared and uploaded in place (kernels read them from device memory: captured run loops stay valid
// unless the sampling time changes, which rides in launch arguments).  Particles, weights and the random streams are not touched —
// loglik / forward_trajectory reset! first, as in the reference.  models == NULL is not accepted for a bank of replicas of more than
// one filter (pass F descriptors).
static int bank_set_models(Bank& b, const llpf_model* models) {
    if (!models) return fail(LLPF_ERR_ARG, "null models");
    CHK(use_device(b));
    const llpf_model m0 = b.cfg.model;
    const int F = b.F;
    std::vector<ModelD> hm(F);
    std::vector<llpf_model> mm(models, models + F);
    for (int f = 0; f < F; ++f) {
        llpf_model& mf = mm[f];
        if (mf.model_id == LLPF_MODEL_LINEAR_GAUSSIAN && m0.model_id >= LLPF_MODEL_USER_BASE && (mf.nx > 4 || mf.ny > 4)) {
            std::string err;                                     // created above the precompiled dimensions: the bank holds the compiled model's id
            const int id = jit_builtin_lg(mf.nx, mf.ny, err);
            if (id == m0.model_id) mf.model_id = id;
        }
        if (mf.model_id != m0.model_id || mf.nx != m0.nx || mf.nu != m0.nu || mf.ny != m0.ny)
            return fail(LLPF_ERR_ARG, "set_model: the new model must have the model id and the dimensions the handle was created with");
        if (m0.model_id == LLPF_MODEL_RB_BILINEAR && (mf.rb.nxl != m0.rb.nxl || mf.rb.fn_kind != m0.rb.fn_kind))
            return fail(LLPF_ERR_ARG, "set_model: LLPF_MODEL_RB_BILINEAR must keep rb.nxl and rb.fn_kind");
        if (m0.model_id == LLPF_MODEL_RB_LINEAR && mf.nxn != m0.nxn) return fail(LLPF_ERR_ARG, "set_model: LLPF_MODEL_RB_LINEAR must keep nxn");
        const int rc = model_prepare(&mf, &hm[f]);
        if (rc) return fail(LLPF_ERR_ARG, "invalid density (covariance not positive definite or dimension mismatch), code " + std::to_string(rc) + ", in filter " + std::to_string(f));
    }
    if (mm[0].Ts != m0.Ts) {          // the time of a step rides in launch arguments: captured run loops are of no use any more
        for (auto& g : b.graphs) if (g.exec) hipGraphExecDestroy(g.exec);
        b.graphs.clear();
    }
    b.hmodels = mm;
    b.cfg.model = mm[0];
    HIPC(hipMemcpyAsync(b.d_models, hm.data(), sizeof(ModelD) * F, hipMemcpyHostToDevice, b.stream));
    if (m0.model_id >= LLPF_MODEL_USER_BASE) {               // the declared bound of the new parameters, evaluated as at creation: u = 0, t = 0
        HIPC(hipMemsetAsync(b.d_uy, 0, sizeof(double) * 4 * MAXD, b.stream));     // (the staging area of the single-step verbs)
        HIPC(launch_user_bound(m0.model_id, b.d_models, F, b.d_uy, b.stream));
    }
    HIPC(hipStreamSynchronize(b.stream));
    if (m0.model_id == LLPF_MODEL_RB_LINEAR) {               // the inner KalmanFilter object of the new filter: kf.x = d0.mu, kf.R = d0.Sigma
        for (int f = 0; f < F; ++f) {
            double S0[16];
            gauss_cov_dense(&b.hmodels[f].linear_initial, S0);
            const int nl = m0.nx - m0.nxn;
            for (int i = 0; i < nl * nl; ++i) { b.rb[f].R[i] = S0[i]; b.rb[f].kfR[i] = S0[i]; }
            for (int i = 0; i < nl; ++i) b.rb[f].kfx[i] = b.hmodels[f].linear_initial.mu[i];
        }
    }
    return LLPF_OK;
}

// weighted_mean of the particle as the accessors see it (nxp values per filter), MAXD rows per launch
static int bank_wmean(Bank& b, double* d_out) {
    for (int r0 = 0; r0 < b.nxp; r0 += MAXD) {
        BankDev d = b.dev();
        if (b.nxp != b.nx) { d.xcur = b.d_x[b.cur] + (size_t)r0 * b.Ns; d.nx = std::min(MAXD, b.nxp - r0); }   // single filter
        HIPC(launch_wmean(d, d_out + r0, b.stream));
    }
    return LLPF_OK;
}

static int check_status(Bank& b, std::vector<FilterScal>& h) {
    for (int f = 0; f < b.F; ++f)
        if (h[f].status) return fail(h[f].status, "degenerate weights (all -Inf or NaN) in filter " + std::to_string(f));
    return LLPF_OK;
}
