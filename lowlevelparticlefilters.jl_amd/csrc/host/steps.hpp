// host/steps.hpp — single-step API: correct!, predict!.  Part of capi.hip (one translation unit).
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
        a.parity = slot; a.need_e2 = 1; a.K = llpf_qbits(b.N); a.k = 0; a.next_step = rel_step(b); a.accumulate = 1;
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
        HIPC(launch_norm(d, slot, 0, 1, rel_step(b), 1, 0, 0, b.stream));
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
    ra.mode = RES_RESAMPLE; ra.parity = (b.parity + ACC_NSLOT - 1) % ACC_NSLOT; ra.step = rel_step(b); ra.M = (int32_t)b.N;
    ra.anc_out = b.d_anc; ra.k = 0;
    HIPC(launch_resample(d, ra, b.stream));
    StepArgs a{};
    a.u = b.d_uy; a.y = nullptr; a.t_prop = t; a.t_meas = t; a.step = rel_step(b); a.has_y = 0; a.parity = b.parity;
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
