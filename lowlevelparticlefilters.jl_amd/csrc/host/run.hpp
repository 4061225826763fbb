// host/run.hpp — the trajectory loop (forward_trajectory / loglik).  Part of capi.hip (one translation unit).
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

// `multi`: every filter of the bank has its own inputs, U [F][T][nu] and Y [F][T][ny] (the Monte-Carlo loops of the
// reference's own benchmark, examples/example_lineargaussian.jl:282-316, as one bank); missing measurements must coincide.
static int bank_run(Bank& b, const double* U, const double* Y, int64_t T, double t_index0,
                    double* ll_total /* [F] */, double* ll_steps /* [T][F] */, double* xmean /* [T][F][nx] */,
                    double* x_hist, double* w_hist, double* we_hist, bool multi = false, double* xcov = nullptr,
                    double* xquant = nullptr /* [T][nx][nq] */, const double* quant_p = nullptr /* [nq] */, int nq = 0) {
    CHK(use_device(b));
    if (T < 1) return fail(LLPF_ERR_ARG, "T must be >= 1");
    if (!Y) return fail(LLPF_ERR_ARG, "Y is null");
    if (b.nu > 0 && !U) return fail(LLPF_ERR_ARG, "U is null");
    test_throw("run");
    if ((x_hist || w_hist || we_hist) && b.F != 1) return fail(LLPF_ERR_ARG, "history outputs need a single filter");
    if (xcov && (b.F != 1 || is_rbfull(b))) return fail(LLPF_ERR_ARG, "the xcov output needs a single filter that is not LLPF_MODEL_RB_BILINEAR");
    if (xcov) CHK(ensure(&b.d_xcov, &b.cap_xc, (size_t)T * b.nx * b.nx + MAXD));
    if (xquant) {      // weighted_quantile(sol, q) (src/filtering.jl:583-595) of the state the history would copy out, per timestep, on the device
        if (b.F != 1 || is_rbfull(b)) return fail(LLPF_ERR_ARG, "the xquant output needs a single filter that is not LLPF_MODEL_RB_BILINEAR");
        if (!quant_p || nq < 1 || nq > 1024) return fail(LLPF_ERR_ARG, "the xquant output needs 1 <= nq <= 1024 probabilities");
        for (int i = 0; i < nq; ++i) if (!(quant_p[i] >= 0.0 && quant_p[i] <= 1.0)) return fail(LLPF_ERR_ARG, "xquant: a probability outside [0, 1]");
        CHK(ensure_wq(b, quant_p, nq));
        HIPC(hipStreamSynchronize(b.stream));
        CHK(ensure(&b.d_xquant, &b.cap_xq, (size_t)T * b.nx * nq));
    }
    b.aux_pending = false; b.we_is_lambda = false;
    const int FM = multi ? b.F : 1;                      // input sets on the device, laid out [T][FM][nu | ny]
    CHK(ensure(&b.d_U, &b.capU, (size_t)T * FM * (b.nu > 0 ? b.nu : 1)));
    CHK(ensure(&b.d_Y, &b.capY, (size_t)T * FM * b.ny));
    std::vector<double> stageU, stageY;
    if (multi) {
        if (is_rb(b)) return fail(LLPF_ERR_ARG, "per-filter inputs are not provided for the Rao-Blackwellized model");
        stageY.resize((size_t)T * FM * b.ny);
        for (int f = 0; f < FM; ++f)
            for (int64_t k = 0; k < T; ++k) {
                const double* src = Y + ((size_t)f * T + k) * b.ny;
                if ((src[0] != src[0]) != (Y[(size_t)k * b.ny] != Y[(size_t)k * b.ny])) return fail(LLPF_ERR_ARG, "missing measurements must coincide across the filters of a bank");
                for (int i = 0; i < b.ny; ++i) stageY[((size_t)k * FM + f) * b.ny + i] = src[i];
            }
        if (b.nu > 0) {
            stageU.resize((size_t)T * FM * b.nu);
            for (int f = 0; f < FM; ++f)
                for (int64_t k = 0; k < T; ++k)
                    for (int i = 0; i < b.nu; ++i) stageU[((size_t)k * FM + f) * b.nu + i] = U[((size_t)f * T + k) * b.nu + i];
        }
    }
    if (b.nu > 0) HIPC(hipMemcpyAsync(b.d_U, multi ? stageU.data() : U, sizeof(double) * T * FM * b.nu, hipMemcpyHostToDevice, b.stream));
    HIPC(hipMemcpyAsync(b.d_Y, multi ? stageY.data() : Y, sizeof(double) * T * FM * b.ny, hipMemcpyHostToDevice, b.stream));
    if (multi) HIPC(hipStreamSynchronize(b.stream));     // the staging vectors are pageable host memory
    if (ll_steps) CHK(ensure(&b.d_ll_steps, &b.cap_ll, (size_t)T * b.F));
    if (xmean) CHK(ensure(&b.d_xmean, &b.cap_xm, (size_t)T * b.F * b.nxp));
    {   // zero the running log-likelihood and remember the resample counter
        std::vector<FilterScal> h;
        CHK(scal_download(b, h));
        b.run_resamples = 0;
        // the device adds step_base to every Philox step argument: the launches of a run carry relative steps 0, 1, ...
        b.step_base = b.n_predict;
        for (int f = 0; f < b.F; ++f) { h[f].ll_total = 0.0; h[f].step_base = b.step_base; b.run_resamples -= h[f].resample_count; }
        CHK(scal_upload(b, h));
    }
    const double Ts = b.cfg.model.Ts;
    // weighted means come out of the normalise / weighting kernels (partial sums over the nx rows they read anyway); the
    // model with per-particle covariance takes them from a k_wmean launch per step over its [xn; xl] rows instead
    const int want_xm = (xmean && !is_rbfull(b)) ? 1 : 0;
    if (want_xm) CHK(ensure_xmpart(b));
    const bool xm_launch = xmean && is_rbfull(b);
    if (xm_launch && b.F != 1) return fail(LLPF_ERR_ARG, "weighted means of a BANK of filters with per-particle covariance are not provided (run without xmean)");
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
    const char* unf_env = getenv("LLPF_UNFUSED");
    // ... and so does the linear-Gaussian model from three states on (measured at N = 1e6 on model-simulated data, tools/bench_nx.py:
    // nx 2 fused 21.1 / balanced 24.8 us per timestep, nx 3 33.8 / 28.8, nx 4 38.2 / 30.5 — the fused kernel drops to three waves per SIMD there)
    const bool heavy_dynamics = b.cfg.model.model_id == LLPF_MODEL_QUADTANK_RK4 ||
                                (b.cfg.model.model_id == LLPF_MODEL_LINEAR_GAUSSIAN && b.nx >= 3 && !is_rb(b));
    // residual resampling produces unsorted ancestors (copies first, multinomial draws after): always the balanced form
    const bool residual = b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL;
    const bool rbm = is_rb(b);
    const bool rbfull = is_rbfull(b);     // per-particle covariance: its own step kernel, balanced form, exp-sums by k_norm
    const bool user_model = b.cfg.model.model_id >= LLPF_MODEL_USER_BASE;   // run-time compiled model: only its k_step exists
    // a likelihood of the model's own that declares no bound (loglik without loglik_bound): there is nothing to normalise against ahead
    // of the weights, so every timestep takes the exact-max form — as launches of the run loop (k_norm in exact form in front of the
    // head), not as a failed bound test that the host notices and redoes (one round trip per timestep until round 4)
    const int model_traits_v = user_model ? jit_model_traits(b.cfg.model.model_id) : 0;
    const bool no_bound = user_model && model_traits_v > 0 && (model_traits_v & LLPF_TRAIT_LOGLIK) && !(model_traits_v & LLPF_TRAIT_LOGLIK_BOUND);
    // (xcov: the covariance is taken from the state between correct! and predict!, which only the balanced form leaves in memory)
    const bool unfused = user_model || rbfull || hist || residual || xcov != nullptr || xquant != nullptr || (unf_env ? atoi(unf_env) != 0 : heavy_dynamics);
    // models whose dynamics are worth a table: the resampling launch evaluates f(x_j) once per surviving source and leaves run-start marks,
    // the step kernel gathers (kernels/resfx.hpp).  LLPF_SOURCE_FX=0 takes the round-3 form (ancestors to HBM, f per distinct ancestor of a block)
    const char* sfx_env = getenv("LLPF_SOURCE_FX");
    // Which of the two pays depends on how many sources survive a resampling — every f(x) of the source-side form makes a round trip
    // through HBM.  Quad-tank, N = 1e6, us per timestep (tools/dbg/qt_regimes.py; EXPERIMENTS.md 4.13): 0.8 % distinct ancestors
    // (BASELINE C3) 31.7 source-side / 36.3 per output, 4.9 % 35.1 / 36.0, 10.5 % 39.6 / 37.2, 24.6 % 47.0 / 38.9, 71 % 54.0 / 47.2.
    // Both launches count the sources whose f the step needed (BankDev::surv, per tile); the host switches the NEXT run's form with a
    // hysteresis (below 5 % -> source-side, above 8 % -> per output).  A handle's first run takes the source-side form.
    // LLPF_SOURCE_FX=0/1 pins it.
    const bool fx_capable = unfused && resample_fx_supported(b.cfg.model.model_id, b.nx, b.ny, b.cfg.resampling_strategy);
    if (fx_capable && b.surv_frac >= 0.0) { if (b.surv_frac < 0.05) b.use_fx = true; else if (b.surv_frac > 0.08) b.use_fx = false; }
    const bool source_fx = fx_capable && (sfx_env ? atoi(sfx_env) != 0 : b.use_fx);
    const size_t n_surv = (size_t)b.F * b.P2 * 4;
    if (fx_capable) {
        if (!b.d_surv) HIPC(hipMalloc(&b.d_surv, sizeof(unsigned long long) * n_surv));
        HIPC(hipMemsetAsync(b.d_surv, 0, sizeof(unsigned long long) * n_surv, b.stream));
    }
    if (source_fx) CHK(ensure_fx(b));
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
    // (round 6: below threshold 1 the split schedule stores no quanta and moves 16 bytes per lane on the steps that do not resample — it
    //  overtakes the merged one from ~1.3 M particles on: N = 1.5e6 / 2e6 / 3e6 at threshold 0.1 27.9 / 34.4 / 43.9 against 29.6 / 36.0 / 48.3 us;
    //  at threshold 1.0 the two stay within 4 % of each other up to 3 M, profiles/r06_schedule_crossover_ab.txt)
    const int64_t merged_max = (b.cfg.resample_threshold < 1.0) ? ((int64_t)5 << 18) : ((int64_t)3 << 20);
    const bool merged = (hist || (sch_env ? (strcmp(sch_env, "merged") == 0) : ((int64_t)b.F * b.Ns <= merged_max)));
    // (a model without a bound: the weighting launches form no sums at all — a step without a measurement would otherwise leave real ones
    // in the slot, against the finite bound max(w), and the exact-form k_norm in front of the next head would add to them)
    const bool acc_in_weighting = merged && !no_bound;
    // Split schedule in front of the fused kernel, thresholds below 1: k_norm stores NO quanta (launch_norm, bound bit 1) and the fused
    // kernel's scan forms its tile's quanta from the weights (ResArgs::lazy_q) — a step that does not resample moves 16 bytes per
    // particle less (the 8 k_norm stored, the 8 the fused kernel requested before it knew), one that does the same bytes plus an exp per
    // source, which is why a filter that resamples at every step keeps the stored form.  The scan then reads weights that other blocks
    // of the same launch are replacing with the next ones: such a run alternates between two weight buffers (BankDev::w / w_next; the
    // second is allocated here on first use and starts as a copy, so that its padding holds -Inf too).  LLPF_LAZY_Q=0: stored form.
    const char* nt_env = getenv("LLPF_NT_ID");
    const char* lazy_s = getenv("LLPF_LAZY_Q");
    const bool lazy_run = !merged && !unfused && !no_bound && b.cfg.resample_threshold < 1.0 && !(lazy_s && atoi(lazy_s) == 0);
    if (lazy_run && !b.d_w_spare) {
        const size_t bytes = sizeof(double) * (size_t)b.F * b.Ns;
        if (hipMalloc(&b.d_w_alloc, bytes) != hipSuccess) { (void)hipGetLastError(); b.d_w_alloc = nullptr; return fail(LLPF_ERR_ALLOC, "second weight buffer of the split schedule"); }
        b.d_w_spare = b.d_w_alloc;
        HIPC(hipMemcpyAsync(b.d_w_spare, b.d_w, bytes, hipMemcpyDeviceToDevice, b.stream));
    }
    double* const wbuf0 = b.d_w;
    double* const wbuf1 = lazy_run ? b.d_w_spare : b.d_w;
    // The run ends in the buffer it began in (a handle's weights do not move between runs: one captured graph per shape, not two that
    // alternate): T - 1 steps have a weighting phase; when that number is odd, step 0 keeps the stored form and weights in place.
    const int64_t k_pp0 = (lazy_run && ((T - 1) & 1)) ? 1 : 0;
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
        // weights in front of step k: every step before it had a weighting phase that wrote the other buffer (the run's last step has none)
        const int64_t wsw = std::max<int64_t>(0, std::min<int64_t>(k, T - 1 > 0 ? T - 1 : 0) - k_pp0);
        b.d_w = (wsw & 1) ? wbuf1 : wbuf0;
        if (lazy_run) b.d_w_spare = (wsw & 1) ? wbuf0 : wbuf1;
        b.w_pingpong = lazy_run && k >= k_pp0;
    };
    auto head_slot = [&](int64_t k) { return (par0 + (int)(k % ACC_NSLOT)) % ACC_NSLOT; };

    auto res_args = [&](int64_t k, bool fast) {
        ResArgs ra{};
        ra.parity = head_slot(k); ra.step = rel_step(b); ra.M = (int32_t)b.N; ra.anc_out = b.d_anc;
        ra.accumulate = 1; ra.want_xmean = want_xm; ra.u_from_scal = 1; ra.count_surv = fx_capable ? 1 : 0;
        ra.ll_steps = ll_steps ? b.d_ll_steps : nullptr;
        ra.xmean = want_xm ? b.d_xmean : nullptr;
        ra.k = k; ra.row = k; ra.fast_head = fast ? 1 : 0;
        ra.ablate = abl_env ? atoi(abl_env) : 0;
        return ra;
    };
    auto step_args = [&](int64_t k) {
        StepArgs st{};
        st.u = b.nu > 0 ? b.d_U + k * FM * b.nu : nullptr;
        st.u_stride = multi ? b.nu : 0; st.y_stride = multi ? b.ny : 0;
        st.t_prop = tk(k);
        st.step = rel_step(b);
        st.parity = b.parity;
        st.need_e2 = ne2; st.K = K; st.k = k; st.next_step = rel_step(b) + 1; st.want_xmean = want_xm; st.accumulate = acc_in_weighting ? 1 : 0;
        if (rbm) { st.rb_pred = b.d_rbseq + (size_t)(2 * k + 1) * b.F; st.rb_corr = b.d_rbseq + (size_t)(2 * k + 2) * b.F; }
        const bool weight = (k + 1 < T);
        if (weight) { st.y = b.d_Y + (k + 1) * FM * b.ny; st.t_meas = tk(k + 1); st.has_y = has_y(k + 1) ? 1 : 0; }
        else { st.y = nullptr; st.t_meas = tk(k); st.has_y = 0; }
        return st;
    };
    // History outputs are staged on the device (one row per timestep, written between the head and the propagate of the
    // balanced form, no host round trip per step) and copied out in bulk at the end; beyond 16 GB of history the
    // step-synchronous loop below copies row by row instead.
    const size_t hist_rows = (size_t)T * b.N;
    const size_t hist_doubles = (x_hist ? hist_rows * b.nxp : 0) + (w_hist ? hist_rows : 0) + (we_hist ? hist_rows : 0);
    bool hist_dev = hist && hist_doubles * sizeof(double) <= ((size_t)16 << 30);
    double *dx_hist = nullptr, *dw_hist = nullptr, *dwe_hist = nullptr;
    if (hist_dev && (b.cap_hist < hist_doubles || !b.d_hist)) {
        // the staging buffer can be most of the device's memory: if it cannot be had, copy row by row instead of failing
        if (b.d_hist) hipFree(b.d_hist);
        b.d_hist = nullptr; b.cap_hist = 0;
        if (hipMalloc(&b.d_hist, sizeof(double) * hist_doubles) == hipSuccess) b.cap_hist = hist_doubles;
        else { (void)hipGetLastError(); b.d_hist = nullptr; hist_dev = false; }
    }
    if (hist_dev) {
        double* p = b.d_hist;
        if (x_hist) { dx_hist = p; p += hist_rows * b.nxp; }
        if (w_hist) { dw_hist = p; p += hist_rows; }
        if (we_hist) { dwe_hist = p; p += hist_rows; }
    }
    // one timestep in the given form; `fast`: the head consumes the bound-offset sums of the previous weighting,
    // otherwise the exact-max sums of a k_norm launched just before (redo of a failed step, or weighted means)
    auto launch_timestep = [&](int64_t k, bool fast, int only_fb) -> int {
        at_step(k);
        BankDev d = b.dev();
        ResArgs ra = res_args(k, fast);
        ra.only_fallback = only_fb;
        StepArgs st = step_args(k);
        const bool weight = (k + 1 < T);
        // (lazy_run, above; the exact redo of a failed bound test keeps the stored form — and the two weight buffers)
        const bool lazy_q = fast && lazy_run && k >= k_pp0;
        if (fast && !merged) {   // split schedule: the sums of the current weights in bound form, as a streaming launch
            ProfScope ps(b, LLPF_PROF_NORMALISE);
            HIPC(launch_norm(d, ra.parity, want_xm, ne2, rel_step(b), 0, lazy_q ? 3 : 1, k, b.stream));
        }
        ra.lazy_q = lazy_q ? 1 : 0;
        ra.nt_id = nt_env ? (atoi(nt_env) != 0 ? 1 : 0) : (((int64_t)b.F * b.Ns >= ((int64_t)7 << 20)) ? 1 : 0);      // nontemporal accesses on the steps that do not resample: working sets well beyond the Infinity Cache (LLPF_NT_ID=0|1 pins it)
        if (!fast) {
            ProfScope ps(b, LLPF_PROF_NORMALISE);
            HIPC(launch_norm(d, ra.parity, want_xm, 1, rel_step(b), only_fb, 0, k, b.stream));
        }
        if (unfused) {
            {
                ra.mode = RES_FINALIZE | RES_RESAMPLE;
                ProfScope ps(b, LLPF_PROF_RESAMPLE);
                if (source_fx) { st.only_fallback = only_fb; st.marks = 1; HIPC(launch_resample_fx(d, ra, st, b.stream)); }
                else HIPC(launch_resample(d, ra, b.stream));
            }
            if (xm_launch) { ProfScope ps(b, LLPF_PROF_OTHER); CHK(bank_wmean(b, b.d_xmean + (size_t)k * b.nxp)); }
            if (xcov) {      // weighted_cov of the state the history would copy out (src/filtering.jl:571-581): mean, then the centred moments
                ProfScope ps(b, LLPF_PROF_OTHER);
                double* mtmp = b.d_xcov + (size_t)T * b.nx * b.nx;
                HIPC(launch_wmean(d, mtmp, b.stream));
                HIPC(launch_wcov(d, mtmp, b.d_xcov + (size_t)k * b.nx * b.nx, b.stream));
            }
            if (xquant) {    // the quantiles of the same state: exp-weights materialised, then the radix selection (k_quantile.hip), [t][state][q]
                ProfScope ps(b, LLPF_PROF_OTHER);
                HIPC(launch_materialize(d, nullptr, b.d_wq_we, b.stream));
                HIPC(launch_wquantile(d.xcur, b.Ns, b.nx, b.d_wq_we, b.N, b.d_wq_p, nq, b.d_xquant + (size_t)k * b.nx * nq, 1, nq, b.d_wq, b.stream));
            }
            if (hist_dev) {   // x[:,t] .= particles(pf); w[:,t] .= weights(pf); we[:,t] .= expweights(pf)  (filtering.jl:357-359)
                ProfScope ps(b, LLPF_PROF_OTHER);
                if (dx_hist) HIPC(launch_soa2aos(b.devp(), b.d_x[b.cur], dx_hist + (size_t)k * b.N * b.nxp, b.stream));
                if (dw_hist || dwe_hist) HIPC(launch_materialize(d, dw_hist ? dw_hist + (size_t)k * b.N : nullptr, dwe_hist ? dwe_hist + (size_t)k * b.N : nullptr, b.stream));
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
            b.last_run_launches += 1;
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

    auto first_weighting = [&]() -> int {
        // weighting of the first correct! (exp-sums against the bound, quanta, tile sums: no separate normalise pass)
        b.cur = cur0; b.qcur = qcur0; b.parity = par0; b.n_predict = np0; b.t_index = ti0;      // the state at entry
        b.d_w = wbuf0; if (lazy_run) b.d_w_spare = wbuf1;
        b.w_pingpong = false;                                                                  // (k_step weights in place)
        BankDev d = b.dev();
        StepArgs a{};
        a.u = b.nu > 0 ? b.d_U : nullptr; a.y = b.d_Y; a.u_stride = multi ? b.nu : 0; a.y_stride = multi ? b.ny : 0; a.t_prop = tk(0); a.t_meas = tk(0); a.step = 0; a.has_y = has_y(0) ? 1 : 0;
        a.parity = par0; a.need_e2 = ne2; a.K = K; a.k = 0; a.next_step = 0; a.want_xmean = want_xm; a.accumulate = acc_in_weighting ? 1 : 0;
        if (rbm) a.rb_corr = b.d_rbseq;
        ProfScope ps(b, LLPF_PROF_PROPAGATE);
        HIPC(launch_step(d, MODE_WEIGHT, a, b.stream));
        return LLPF_OK;
    };
    b.last_run_launches = 0; b.last_run_fx_steps = source_fx ? T : 0; b.last_run_surv = -1.0;
    // the asynchronous loop as a captured graph, replayed when nothing a launch argument depends on has changed
    static const char* graph_env = getenv("LLPF_GRAPH");
    const bool use_graph = !hist && !b.profiling && !dbg_env && !(graph_env && atoi(graph_env) == 0);
    hipGraphExec_t gexec = nullptr;
    if (use_graph) {
        Bank::RunGraph key{};
        key.T = T; key.t_index0 = t_index0; key.par0 = par0; key.cur0 = cur0; key.qcur0 = qcur0;
        key.flags = (merged ? 1 : 0) | (unfused ? 2 : 0) | (want_xm ? 4 : 0) | (ll_steps ? 8 : 0) | (xm_launch ? 16 : 0) | (multi ? 32 : 0) | (source_fx ? 64 : 0) | (xcov ? 128 : 0) | (xquant ? 256 : 0) | ((abl_env ? atoi(abl_env) : 0) << 9) | (lazy_run ? (1 << 30) : 0) | ((nt_env && atoi(nt_env) != 0) ? (1 << 29) : 0) | ((nt_env && atoi(nt_env) == 0) ? (1 << 28) : 0);      // (no_bound is a property of the model id, which a handle keeps)
        key.np_parity = (int)(np0 & 1u);
        key.dU = b.d_U; key.dY = b.d_Y; key.dll = ll_steps ? b.d_ll_steps : nullptr; key.dxm = xmean ? b.d_xmean : nullptr; key.dxc = xcov ? b.d_xcov : nullptr; key.drb = b.d_rbseq;
        key.dw = wbuf0; key.dws = wbuf1;
        key.dxq = xquant ? b.d_xquant : nullptr; key.dqp = xquant ? b.d_wq_p : nullptr; key.nq = xquant ? nq : 0;
        key.yhash = 1469598103934665603ULL;
        for (int64_t k = 0; k < T; ++k) key.yhash = (key.yhash ^ (uint64_t)(has_y(k) ? 1 : 2)) * 1099511628211ULL;
        // a run shape is captured the second time it is seen (capture + instantiation of ~T nodes costs several ms:
        // one-off shapes are simply enqueued)
        Bank::RunGraph* slot = nullptr;
        for (auto& g : b.graphs) if (g.same(key)) { slot = &g; gexec = g.exec; break; }
        if (!slot) {
            if (b.graphs.size() >= 4) { if (b.graphs.front().exec) hipGraphExecDestroy(b.graphs.front().exec); b.graphs.erase(b.graphs.begin()); }
            key.exec = nullptr;
            b.graphs.push_back(key);
        } else if (!gexec) {
            hipGraph_t graph = nullptr;
            HIPC(hipStreamBeginCapture(b.stream, hipStreamCaptureModeThreadLocal));
            int rc = first_weighting();
            for (int64_t k = 0; rc == LLPF_OK && k < T; ++k) rc = launch_timestep(k, !no_bound, 0);
            const hipError_t ee = hipStreamEndCapture(b.stream, &graph);
            if (rc != LLPF_OK) { if (graph) hipGraphDestroy(graph); return rc; }
            if (ee != hipSuccess) return fail(LLPF_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
            const hipError_t ei = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (ei != hipSuccess) return fail(LLPF_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei));
            slot->exec = gexec;
        }
    }
    HIPC(hipEventRecord(b.ev_run0, b.stream));
    if (gexec) { HIPC(hipGraphLaunch(gexec, b.stream)); b.last_run_launches = T; }
    else CHK(first_weighting());
    if (hist && !hist_dev) {
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
                HIPC(launch_norm(d, ra.parity, want_xm, 1, rel_step(b), 1, 0, k, b.stream));
                ra.fast_head = 0; ra.only_fallback = 1;
                HIPC(launch_resample(d, ra, b.stream));
                ra.only_fallback = 0;
                CHK(clear_fallback(b, fl));
            }
            if (xm_launch) CHK(bank_wmean(b, b.d_xmean + (size_t)k * b.nxp));
            if (xcov) {
                double* mtmp = b.d_xcov + (size_t)T * b.nx * b.nx;
                HIPC(launch_wmean(d, mtmp, b.stream));
                HIPC(launch_wcov(d, mtmp, b.d_xcov + (size_t)k * b.nx * b.nx, b.stream));
            }
            if (xquant) {
                HIPC(launch_materialize(d, nullptr, b.d_wq_we, b.stream));
                HIPC(launch_wquantile(d.xcur, b.Ns, b.nx, b.d_wq_we, b.N, b.d_wq_p, nq, b.d_xquant + (size_t)k * b.nx * nq, 1, nq, b.d_wq, b.stream));
            }
            if (x_hist) {
                HIPC(launch_soa2aos(b.devp(), b.d_x[b.cur], b.d_tmp, b.stream));
                HIPC(hipMemcpyAsync(x_hist + (size_t)k * b.N * b.nxp, b.d_tmp, sizeof(double) * b.N * b.nxp, hipMemcpyDeviceToHost, b.stream));
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
        bool replayed = gexec != nullptr;       // the graph holds all T timesteps; after a failed bound test the rest is enqueued
        // Optimistic enqueue: all remaining timesteps at once, one poll at the end.  Every launch after a failed bound
        // test is a no-op, so when tests fail often (banks of many small filters: some filter fails at most steps) the
        // batch shrinks to a quarter on a failure and doubles again on a clean batch.
        int64_t batch = T;
        while (k0 < T) {
            const int64_t k1 = replayed ? T : std::min(T, k0 + batch);
            if (!replayed) {
                for (int64_t k = k0; k < k1; ++k) CHK(launch_timestep(k, !no_bound, 0));
            }
            replayed = false;
            std::vector<int> fl;
            int64_t kf;
            CHK(poll_fallback(b, fl, kf));
            if (fl.empty()) { k0 = k1; batch = std::min(T, batch * 2); continue; }
            // step kf of the flagged filters: exact-max normalisation of the same weights, then the step again
            CHK(clear_slot_sums(b, head_slot(kf), fl));
            CHK(launch_timestep(kf, false, 1));
            CHK(clear_fallback(b, fl));
            k0 = kf + 1;
            batch = std::max<int64_t>(1, std::min(batch, T) / 4);
        }
    }
    at_step(T);
    b.w_pingpong = false;                                    // the verbs outside a run weight in place, on the buffer the run ended in
    b.qcur = qcur0 ^ (int)(T & 1);                           // the last step has no weighting phase: no quanta swap
    b.parity = (par0 + (int)(T % ACC_NSLOT)) % ACC_NSLOT;
    {
        BankDev d = b.dev();
        ProfScope ps(b, LLPF_PROF_OTHER);
        // the run's k_norm launches stored no quanta: leave those of the current weights behind, as every later verb expects them
        if (lazy_run) HIPC(launch_requant(d, b.stream));
        HIPC(launch_post_predict(d, b.stream));
    }
    HIPC(hipEventRecord(b.ev_run1, b.stream));
    if (hist_dev) {
        if (x_hist) HIPC(hipMemcpyAsync(x_hist, dx_hist, sizeof(double) * hist_rows * b.nxp, hipMemcpyDeviceToHost, b.stream));
        if (w_hist) HIPC(hipMemcpyAsync(w_hist, dw_hist, sizeof(double) * hist_rows, hipMemcpyDeviceToHost, b.stream));
        if (we_hist) HIPC(hipMemcpyAsync(we_hist, dwe_hist, sizeof(double) * hist_rows, hipMemcpyDeviceToHost, b.stream));
    }
    if (hist_dev && b.cap_hist * sizeof(double) > ((size_t)256 << 20)) {
        // a large history staging buffer is not kept for the lifetime of the handle (other filters / banks need the memory)
        HIPC(hipStreamSynchronize(b.stream));
        hipFree(b.d_hist);
        b.d_hist = nullptr; b.cap_hist = 0;
    }
    if (ll_steps) HIPC(hipMemcpyAsync(ll_steps, b.d_ll_steps, sizeof(double) * T * b.F, hipMemcpyDeviceToHost, b.stream));
    if (xmean) HIPC(hipMemcpyAsync(xmean, b.d_xmean, sizeof(double) * T * b.F * b.nxp, hipMemcpyDeviceToHost, b.stream));
    if (xcov) HIPC(hipMemcpyAsync(xcov, b.d_xcov, sizeof(double) * T * b.nx * b.nx, hipMemcpyDeviceToHost, b.stream));
    if (xquant) HIPC(hipMemcpyAsync(xquant, b.d_xquant, sizeof(double) * T * b.nx * nq, hipMemcpyDeviceToHost, b.stream));
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
    double surv = 0.0;
    if (fx_capable) {
        std::vector<unsigned long long> hs(n_surv);
        HIPC(hipMemcpy(hs.data(), b.d_surv, sizeof(unsigned long long) * n_surv, hipMemcpyDeviceToHost));
        for (unsigned long long v : hs) surv += (double)v;
    }
    if (fx_capable && T >= 1) b.last_run_surv = b.surv_frac = surv / ((double)T * (double)b.N * (double)b.F);
    return check_status(b, h);
}
