// host/access.hpp — accessors and set_weights.  Part of capi.hip (one translation unit).
// ---- accessors ----------------------------------------------------------------------------------
static int bank_get_particles(Bank& b, double* dst) {
    CHK(use_device(b));
    BankDev d = b.devp();
    HIPC(launch_soa2aos(d, b.d_x[b.cur], b.d_tmp, b.stream));
    HIPC(hipMemcpyAsync(dst, b.d_tmp, sizeof(double) * (size_t)b.F * b.N * b.nxp, hipMemcpyDeviceToHost, b.stream));
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
    HIPC(launch_norm(d, b.parity, 0, 1, rel_step(b), 0, 0, 0, b.stream));
    ResArgs ra{};
    ra.mode = RES_FINALIZE; ra.parity = b.parity; ra.M = (int32_t)b.N; ra.keep_norm = 1; ra.fast_head = 0;
    HIPC(launch_resample(d, ra, b.stream));
    b.parity = (b.parity + 1) % ACC_NSLOT;
    CHK(scal_download(b, h));
    return check_status(b, h);
}
