// kernels/accum.hpp — cross-block accumulators, WeightAcc, block-wide helpers.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// cross-block accumulators (see engine.hpp): order-preserving max key, limb-wise integer sums
// ------------------------------------------------------------------------------------------------
DEV uint64_t max_key(double x) {           // monotone map double -> u64; NaN (positive) maps above +inf,
    const uint64_t u = llpf_d2u(x);        // so a NaN weight wins the max like Julia's findmax; key 0 is below -inf
    return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
DEV double max_unkey(uint64_t k) {
    return llpf_u2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k);
}
DEV uint64_t* acc_slot(uint64_t* acc, int word, int shard) { return acc + ((size_t)word * NSHARD + shard) * ACC_STRIDE; }
DEV const uint64_t* acc_slot(const uint64_t* acc, int word, int shard) { return acc + ((size_t)word * NSHARD + shard) * ACC_STRIDE; }

DEV void acc_max(uint64_t* acc, int parity, double blockmax, bool any_nan) {
    const uint64_t key = any_nan ? max_key(llpf_u2d(0x7ff8000000000000ULL)) : max_key(blockmax);
    atomicMax(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_PM(parity), blockIdx.x & (NSHARD - 1))),
              (unsigned long long)key);
}
// every wave combines the NSHARD copies of the running max itself (lanes 0..7 load, 3 shuffles, broadcast)
DEV double acc_read_max_wave(const uint64_t* acc, int parity) {
    const int lane = threadIdx.x & 63;
    uint64_t k = (lane < NSHARD) ? *acc_slot(acc, ACC_PM(parity), lane) : 0;
#define LLPF_KSTEP(CTRL) { const uint64_t t = dpp_u64<CTRL, 0xF, false>(k, k); k = t > k ? t : k; }
    LLPF_KSTEP(DPP_QUAD_XOR1) LLPF_KSTEP(DPP_QUAD_XOR2) LLPF_KSTEP(DPP_ROW_HALF_MIRROR)
#undef LLPF_KSTEP
    k = readlane_u64(k, 0);
    return max_unkey(k);
}
constexpr uint64_t M43 = ((uint64_t)1 << 43) - 1;
DEV void acc_add_u128(uint64_t* acc, int word0, llpf_u128 v) {
    const int sh = blockIdx.x & (NSHARD - 1);
    const uint64_t limb[3] = {v.lo & M43, ((v.lo >> 43) | (v.hi << 21)) & M43, v.hi >> 22};
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (limb[k]) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, word0 + k, sh)), (unsigned long long)limb[k]);
}
// limb sums (each already summed over shards) -> 128-bit value
DEV llpf_u128 acc_combine_u128(uint64_t a0, uint64_t a1, uint64_t a2) {
    llpf_u128 r = {a0, 0}, t;
    t.lo = a1 << 43; t.hi = a1 >> 21;
    r = llpf_u128_add(r, t);
    t.lo = 0; t.hi = a2 << 22;
    r = llpf_u128_add(r, t);
    return r;
}

DEV uint64_t* tileq_slot(const BankDev& b, int slot, int f) { return b.tileq + ((size_t)slot * b.F + f) * b.P2; }
// The per-block parts of the weighted mean belong to an accumulator slot like the sums they travel with: in the fused kernel tile 0's
// head reads the parts of the PREVIOUS weighting while the other blocks of the same launch already store the next ones (round 5: one
// set of parts was a read-after-write hazard between blocks of one launch whenever block 0 ran late; seen as a rare wrong xmean row).
DEV double* xmpart_slot(const BankDev& b, int slot, int f) { return b.xmpart + ((size_t)slot * b.F + f) * b.P1 * MAXD; }

// a launch of run-step k is a no-op when an EARLIER launch flagged a failed bound test (flag = 1 + its step)
DEV bool run_is_stopped(const BankDev& b, int64_t k) {
    const uint32_t fl = *b.bank_flag;
    return fl != 0 && (int64_t)(fl - 1) < k;
}

// Exp-sums of freshly computed weights against the analytic bound `off` (see oracle/llpf_oracle.c:dev_norm_bound):
// e = exp(w - off) <= 1, S += fix96(e), [E2 += fix96(e^2)], quantum q = floor(e 2^K).
struct WeightAcc {
    llpf_u128 S, E2;
    uint64_t bad;
    DEV void init() { S.lo = 0; S.hi = 0; E2.lo = 0; E2.hi = 0; bad = 0; }
    DEV uint64_t add(double w, double off, int K, bool need_e2, double* e_out = nullptr) {
        const double e = llpf_exp_le0(w - off);
        if (e_out) *e_out = e;
        bad += (e != e) ? 1u : 0u;
        S = llpf_u128_add(S, llpf_fix96_unit(e));
        if (need_e2) E2 = llpf_u128_add(E2, llpf_fix96_unit(e * e));
        return llpf_q64_unit(e, K);
    }
    // the same for a workgroup that is one wave (k_rbfull)
    DEV void flush_wave(uint64_t* acc, int slot, bool need_e2) {
        llpf_u128 s = wave_sum_u128(S), e2 = {0, 0};
        if (need_e2) e2 = wave_sum_u128(E2);
        const uint64_t bd = (uint64_t)__builtin_popcountll(__ballot(bad != 0));
        if (threadIdx.x == 0) {
            acc_add_u128(acc, ACC_S(slot), s);
            if (need_e2) acc_add_u128(acc, ACC_E2(slot), e2);
            if (bd) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(slot), blockIdx.x & (NSHARD - 1))), (unsigned long long)bd);
        }
    }
    // block-wide totals into the sharded accumulators of `slot`; sm: [BLOCK/64][5] u64 of LDS
    DEV void flush(uint64_t* acc, int slot, bool need_e2, uint64_t (*sm)[5]) {
        llpf_u128 s = wave_sum_u128(S), e2 = {0, 0};
        if (need_e2) e2 = wave_sum_u128(E2);
        const uint64_t bd = (uint64_t)__builtin_popcountll(__ballot(bad != 0));     // only its being non-zero is ever used
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) { sm[wv][0] = s.lo; sm[wv][1] = s.hi; sm[wv][2] = e2.lo; sm[wv][3] = e2.hi; sm[wv][4] = bd; }
        __syncthreads();
        if (threadIdx.x == 0) {
            llpf_u128 ts = {sm[0][0], sm[0][1]}, te = {sm[0][2], sm[0][3]};
            uint64_t tb = sm[0][4];
            for (int k = 1; k < BLOCK / 64; ++k) {
                llpf_u128 a1 = {sm[k][0], sm[k][1]}, a2 = {sm[k][2], sm[k][3]};
                ts = llpf_u128_add(ts, a1);
                te = llpf_u128_add(te, a2);
                tb += sm[k][4];
            }
            acc_add_u128(acc, ACC_S(slot), ts);
            if (need_e2) acc_add_u128(acc, ACC_E2(slot), te);
            if (tb) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(slot), blockIdx.x & (NSHARD - 1))), (unsigned long long)tb);
        }
    }
};

// The end of a weighting block in ONE barrier: the block's maximum / NaN flag and (accumulate) the exp-sums of its weights go to the
// sharded accumulators of `slot`.  Same values as block_max + __syncthreads_or + acc_max + WeightAcc::flush (nine barriers): in the
// kernels whose blocks all reach their tails together (k_step<..., MARKS>) those were 4.5 k cycles per block with nothing to hide under.
DEV void block_flush(uint64_t* acc, int slot, double bmax, bool bad, const WeightAcc& wa, bool accumulate, bool need_e2, uint64_t (*sm)[8]) {
    const double wm = wave_max(bmax);
    const uint64_t nb = __ballot(bad) ? 1u : 0u;
    llpf_u128 s = {0, 0}, e2 = {0, 0};
    uint64_t bd = 0;
    if (accumulate) {
        s = wave_sum_u128(wa.S);
        if (need_e2) e2 = wave_sum_u128(wa.E2);
        bd = (uint64_t)__builtin_popcountll(__ballot(wa.bad != 0));
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sm[wv][0] = llpf_d2u(wm); sm[wv][1] = nb; sm[wv][2] = s.lo; sm[wv][3] = s.hi; sm[wv][4] = e2.lo; sm[wv][5] = e2.hi; sm[wv][6] = bd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = llpf_u2d(sm[0][0]);
        uint64_t anybad = sm[0][1], tb = sm[0][6];
        llpf_u128 ts = {sm[0][2], sm[0][3]}, te = {sm[0][4], sm[0][5]};
        for (int k = 1; k < BLOCK / 64; ++k) {
            r = llpf_fmax(r, llpf_u2d(sm[k][0]));
            anybad |= sm[k][1];
            llpf_u128 a1 = {sm[k][2], sm[k][3]}, a2 = {sm[k][4], sm[k][5]};
            ts = llpf_u128_add(ts, a1);
            te = llpf_u128_add(te, a2);
            tb += sm[k][6];
        }
        acc_max(acc, slot, r, anybad != 0);
        if (accumulate) {
            acc_add_u128(acc, ACC_S(slot), ts);
            if (need_e2) acc_add_u128(acc, ACC_E2(slot), te);
            if (tb) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(slot), blockIdx.x & (NSHARD - 1))), (unsigned long long)tb);
        }
    }
}

// fixed-order fp64 block sum of the per-thread partial sums e_i x_i (weighted-mean output only; never fed back)
template <int NX>
DEV void block_store_xm(const double* xm, double* dst /* [MAXD] */, double (*smx)[MAXD]) {
    double v[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) v[d] = wave_sum_f64(xm[d]);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) smx[wv][d] = v[d];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            double a = smx[0][d];
            for (int k = 1; k < BLOCK / 64; ++k) a = a + smx[k][d];
            dst[d] = a;
        }
    }
}
