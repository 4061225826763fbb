// host/fallback.hpp — host side of a failed bound test (exact-max redo).  Part of capi.hip (one translation unit).
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
