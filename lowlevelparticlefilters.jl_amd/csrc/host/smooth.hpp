// host/smooth.hpp — FFBS particle smoother driver.  Part of capi.hip (one translation unit).
// ---- FFBS particle smoother (reference src/smoothing.jl:116-143) ---------------------------------------------------
extern "C" int llpf_resample(int32_t device, int32_t strategy, const double* we, int64_t n, int64_t m, const double* U, int64_t* j);

static int bank_smooth(Bank& b, int64_t M, const double* U, int64_t T, const double* xf, const double* wf,
                       const double* wef, double* xb, int64_t* idx) {
    CHK(use_device(b));
    if (b.F != 1) return fail(LLPF_ERR_ARG, "smooth needs a single filter");
    if (is_rb(b) || is_rbfull(b)) return fail(LLPF_ERR_ARG, "smooth is not defined for the Rao-Blackwellized model");
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
