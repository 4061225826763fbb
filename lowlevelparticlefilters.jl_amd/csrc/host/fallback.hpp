// host/fallback.hpp — host side of a failed bound test (exact-max redo).  Part of capi.hip (one translation unit).
// ---- exact-form redo of a normalisation whose bound test failed ------------------------------------------------
// zero the exp-sum words of accumulator slot `slot` for the filters whose fallback flag is set (on the device: a bank of
// thousands of small filters may flag most of them at once)
static int clear_slot_sums(Bank& b, int slot, const std::vector<int>&) {
    HIPC(launch_fb_clear(b.dev(), slot, 0, b.stream));
    return LLPF_OK;
}
// did some filter ask for the exact form (and at which run-step)?  `fl` is non-empty if so; clears nothing
static int poll_fallback(Bank& b, std::vector<int>& fl, int64_t& kf) {
    uint32_t flag = 0;
    HIPC(hipMemcpyAsync(&flag, b.d_flag, sizeof(flag), hipMemcpyDeviceToHost, b.stream));
    HIPC(hipStreamSynchronize(b.stream));
    fl.clear();
    kf = -1;
    if (!flag) return LLPF_OK;
    kf = (int64_t)flag - 1;
    fl.push_back(0);             // the flagged filters are known to the device (FilterScal::fallback); the host only needs "some"
    return LLPF_OK;
}
static int clear_fallback(Bank& b, const std::vector<int>&) {
    HIPC(launch_fb_clear(b.dev(), 0, 1, b.stream));
    return LLPF_OK;
}
static int need_e2(const Bank& b) { return b.cfg.resample_threshold != 1.0 ? 1 : 0; }
