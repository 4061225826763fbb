// host/rbkf.hpp — Rao-Blackwellized model: shared covariance recursion on the host.  Part of capi.hip (one translation unit).
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
