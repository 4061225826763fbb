// host/aux.hpp — AuxiliaryParticleFilter verbs and run loop.  Part of capi.hip (one translation unit).
// ---- AuxiliaryParticleFilter{ParticleFilter} (reference src/filtering.jl:170-217, 367-384; smoothing.jl:232-236) ----
// Launches of the auxiliary filter.  `epoch` is the position of the launch within a run (stop test of a failed
// bound); `row` the ll_steps / xmean row a finalize writes.
struct AuxOuts { double* d_ll_steps = nullptr; double* d_xmean = nullptr; int accumulate = 0; };

// correct!(pf::AuxiliaryParticleFilter): ll = logsumexp!(state) only; weights produced by an aux predict! have
// their exp-sums (against the bound c0 - log N) waiting in accumulator slot parity-1: one finalize launch
static int aux_launch_finalize(Bank& b, bool fast, int only_fb, int64_t epoch, int64_t row, const AuxOuts& o) {
    const int slot = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT;
    BankDev d = b.dev();
    if (!fast) HIPC(launch_norm(d, slot, o.d_xmean ? 1 : 0, 1, rel_step(b), only_fb, 0, epoch, b.stream));
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
    a.u = d_u; a.y = d_y1; a.t_prop = t; a.t_meas = t; a.step = rel_step(b); a.has_y = has_y1 ? 1 : 0;
    a.parity = b.parity; a.need_e2 = 0; a.K = llpf_qbits(b.N); a.k = epoch; a.next_step = rel_step(b); a.accumulate = 1;
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
    if (!fast) HIPC(launch_norm(d, slot1, 0, 0, rel_step(b), only_fb, 0, epoch, b.stream));
    ResArgs ra{};
    ra.mode = RES_FINALIZE | RES_RESAMPLE; ra.parity = slot1; ra.step = rel_step(b); ra.M = (int32_t)b.N;
    ra.anc_out = b.d_anc; ra.force = 1; ra.fast_head = fast ? 1 : 0; ra.u_from_scal = 1; ra.K = K; ra.k = epoch;
    ra.only_fallback = only_fb;
    StepArgs st{};
    st.t_prop = t; st.t_meas = t; st.step = rel_step(b); st.has_y = 0; st.parity = b.parity; st.need_e2 = 0; st.K = K;
    st.k = epoch; st.next_step = rel_step(b) + 1; st.want_xmean = want_xm; st.accumulate = 1; st.aux = has_y1 ? 2 : 1;
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

// predict!(pf::AuxiliaryParticleFilter{<:AdvancedParticleFilter}, u, y, p, t) — reference src/filtering.jl:219-234: the look-ahead
// weights lambda (noise-free prediction) only steer the resampling; the particles are then propagated AGAIN from xprev[j], with
// noise, and the weights are reset (lambda is discarded: the following correct!, which is logsumexp! only, returns ~0).
// Device: k_step<MODE_AUX> (x' = f(x) into the scratch plane, w <- w + lambda, exp-sums) -> k_resample (expnormalize! + forced
// resample: ancestors) -> k_step<MODE_PROP> from the ORIGINAL particles -> reset_weights!.  Synchronous.
static int aux_predict_dev_advanced(Bank& b, const double* d_u, const double* d_y1, bool has_y1, double t) {
    CHK(aux_ensure_lam(b));
    CHK(aux_launch_look(b, d_u, d_y1, has_y1, t, 0, 0));
    const int slot = b.parity;
    b.parity = (b.parity + 1) % ACC_NSLOT;
    b.qcur ^= 1;                                  // the quanta of w + lambda are the current ones; b.cur stays: xnext holds x' and is overwritten
    BankDev d = b.dev();
    ResArgs ra{};
    ra.mode = RES_FINALIZE | RES_RESAMPLE; ra.parity = slot; ra.step = rel_step(b); ra.M = (int32_t)b.N; ra.anc_out = b.d_anc;
    ra.force = 1; ra.fast_head = 1; ra.u_from_scal = 1; ra.k = 0;
    HIPC(launch_resample(d, ra, b.stream));
    std::vector<int> fl;
    int64_t kf;
    CHK(poll_fallback(b, fl, kf));
    if (!fl.empty()) {   // expnormalize! of w + lambda in the exact-max form
        CHK(clear_slot_sums(b, slot, fl));
        HIPC(launch_norm(d, slot, 0, 0, rel_step(b), 1, 0, 0, b.stream));
        ra.fast_head = 0; ra.only_fallback = 1;
        HIPC(launch_resample(d, ra, b.stream));
        CHK(clear_fallback(b, fl));
    }
    StepArgs a{};
    a.u = d_u; a.y = nullptr; a.t_prop = t; a.t_meas = t; a.step = rel_step(b); a.has_y = 0; a.parity = b.parity;
    a.K = llpf_qbits(b.N); a.k = 0;
    HIPC(launch_step(d, MODE_PROP, a, b.stream));             // propagate_particles!(pf.pf, u, j, p, t): with noise, from xprev[j]
    HIPC(launch_post_predict(d, b.stream));                   // reset_weights!(s)
    b.cur ^= 1;
    b.n_predict++;
    b.t_index++;
    b.aux_pending = false;
    b.we_is_lambda = false;
    return LLPF_OK;
}

// Single-call predict!(pf::AuxiliaryParticleFilter, u, y1, p, t): synchronous.  d_u / d_y1 are device pointers.
static int aux_predict_dev(Bank& b, const double* d_u, const double* d_y1, bool has_y1, double t, int want_xm) {
    if (is_rb(b) || is_rbfull(b)) return fail(LLPF_ERR_ARG, "the auxiliary filter is not defined for the Rao-Blackwellized model");
    if (b.nx > 8) return fail(LLPF_ERR_ARG, "the auxiliary filter is compiled for up to 8 states (this filter has " + std::to_string(b.nx) + ")");
    if (b.cfg.filter_kind == LLPF_ADVANCED_PARTICLE_FILTER) {
        if (b.aux_pending) CHK(bank_aux_correct(b, nullptr, AuxOuts{}, 0));
        return aux_predict_dev_advanced(b, d_u, d_y1, has_y1, t);
    }
    if (b.aux_pending) CHK(bank_aux_correct(b, nullptr, AuxOuts{}, 0));   // contract: predict! works on normalised weights
    CHK(aux_ensure_lam(b));
    CHK(aux_launch_look(b, d_u, d_y1, has_y1, t, 0, 0));
    b.parity = (b.parity + 1) % ACC_NSLOT;
    b.qcur ^= 1;
    b.cur ^= 1;                                   // the noise-free prediction is the source of the second half
    if (b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL) {
        // resample(ResampleResidual, ...) (src/resample.jl:63-117) under the auxiliary filter: residual ancestors are not sorted, which
        // the fused second half relies on; the balanced form instead — k_resample (expnormalize! of w + lambda, forced residual
        // resample: ancestors to HBM), then k_step<NoModel, MODE_AUX2>: x = x'[j] + noise, w = lambda - log N, exp-sums
        const int slot1 = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT;
        BankDev d = b.dev();
        ResArgs ra{};
        ra.mode = RES_FINALIZE | RES_RESAMPLE; ra.parity = slot1; ra.step = rel_step(b); ra.M = (int32_t)b.N; ra.anc_out = b.d_anc;
        ra.force = 1; ra.fast_head = 1; ra.u_from_scal = 1; ra.k = 0;
        HIPC(launch_resample(d, ra, b.stream));
        std::vector<int> fl;
        int64_t kf;
        CHK(poll_fallback(b, fl, kf));
        if (!fl.empty()) {   // expnormalize! of w + lambda in the exact-max form
            CHK(clear_slot_sums(b, slot1, fl));
            HIPC(launch_norm(d, slot1, 0, 0, rel_step(b), 1, 0, 0, b.stream));
            ra.fast_head = 0; ra.only_fallback = 1;
            HIPC(launch_resample(d, ra, b.stream));
            CHK(clear_fallback(b, fl));
        }
        StepArgs st{};
        st.t_prop = t; st.t_meas = t; st.step = rel_step(b); st.has_y = 0; st.parity = b.parity; st.need_e2 = 0; st.K = llpf_qbits(b.N);
        st.k = 0; st.next_step = rel_step(b) + 1; st.want_xmean = want_xm; st.accumulate = 1; st.aux = has_y1 ? 2 : 1;
        HIPC(launch_step(d, MODE_AUX2, st, b.stream));
        b.parity = (b.parity + 1) % ACC_NSLOT;
        b.qcur ^= 1;
        b.cur ^= 1;
        b.n_predict++;
        b.t_index++;
        b.aux_pending = true;
        b.we_is_lambda = true;
        return LLPF_OK;
    }
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
    // refused HERE, before any state of the handle moves (the back-to-back epochs below never pass through aux_predict_dev's own check)
    if (is_rb(b) || is_rbfull(b)) return fail(LLPF_ERR_ARG, "the auxiliary filter is not defined for the Rao-Blackwellized model");
    if (b.nx > 8) return fail(LLPF_ERR_ARG, "the auxiliary filter is compiled for up to 8 states (this filter has " + std::to_string(b.nx) + ")");
    CHK(ensure(&b.d_U, &b.capU, (size_t)T * (b.nu > 0 ? b.nu : 1)));
    CHK(ensure(&b.d_Y, &b.capY, (size_t)T * b.ny));
    if (b.nu > 0) HIPC(hipMemcpyAsync(b.d_U, U, sizeof(double) * T * b.nu, hipMemcpyHostToDevice, b.stream));
    HIPC(hipMemcpyAsync(b.d_Y, Y, sizeof(double) * T * b.ny, hipMemcpyHostToDevice, b.stream));
    CHK(ensure(&b.d_ll_steps, &b.cap_ll, (size_t)T * b.F));
    if (xmean) CHK(ensure(&b.d_xmean, &b.cap_xm, (size_t)T * b.F * b.nx));
    const bool residual = b.cfg.resampling_strategy == LLPF_RESAMPLE_RESIDUAL;     // balanced form, driven step by step (aux_predict_dev)
    CHK(aux_ensure_lam(b));
    const double Ts = b.cfg.model.Ts;
    const bool hist = x_hist || w_hist || we_hist;
    const int want_xm = xmean ? 1 : 0;
    if (want_xm) CHK(ensure_xmpart(b));
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
    const bool advanced = b.cfg.filter_kind == LLPF_ADVANCED_PARTICLE_FILTER;     // its predict! is driven synchronously (filtering.jl:219-234)
    if (hist || advanced || residual) {
        // step-synchronous form (history is copied out between correct! and predict!)
        if (hist && (mode == 0 || T > 1)) CHK(record(0));
        for (int64_t k = 0; k < n_aux; ++k) {
            CHK(aux_predict_dev(b, b.nu > 0 ? b.d_U + k * b.nu : nullptr, b.d_Y + (k + 1) * b.ny, has_y(k + 1), (double)k * Ts, want_xm));
            if (mode == 1 && k + 1 == T - 1) break;           // loglik: the last step is the wrapped filter's update!
            CHK(bank_aux_correct(b, nullptr, outs, k + 1));
            if (hist) CHK(record(k + 1));
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
