// parked (EXPERIMENTS.md 5.1): one wave per tile for the split schedule's exp-sums; bit-identical, 61.5 us against k_norm's 54.9 us at 128 x 1e5 (same box)
// ------------------------------------------------------------------------------------------------
// k_norm_wt — the same sums with ONE WAVE PER TILE (round 5).  k_norm gives a tile to a block of four waves: four elements per lane,
// then every wave reduces its 128-bit sums over the lanes (~75 DPP instructions per wave for 256 elements), a barrier, LDS, one
// thread combining — a third of the kernel's instructions are that tail.  Here a wave walks a whole tile (16 elements per lane,
// eight coalesced 1 KB requests issued together), reduces once, and lane 0 hands the tile's totals to the sharded accumulators
// itself: no LDS, no barrier, the reduction amortised over four times as many elements.  All sums are integers, so the totals
// — and everything downstream — are the bits k_norm produces (tests: test_schedules_are_bit_identical, the bank tests).
// Used for the plain form only (no weighted mean, more than one tile: the one-tile filter's in-launch bound test stays in k_norm).
// ------------------------------------------------------------------------------------------------
constexpr int NORM_WT_IPT = TILE / 64;      // 16 elements per lane
template <bool NEED_E2>
__global__ __launch_bounds__(BLOCK) void k_norm_wt(BankDev b, int K, int parity, uint32_t step, int only_fallback, int bound, int64_t kstep) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int tile = (int)blockIdx.x * (BLOCK / 64) + (int)(threadIdx.x >> 6);
    if (tile >= b.P2) return;
    uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    const double* __restrict__ w = b.w + (size_t)f * b.Ns + (size_t)tile * TILE + lane * 2;
    uint64_t* __restrict__ qo = b.quanta + (size_t)f * b.Ns + (size_t)tile * TILE + lane * 2;
    // flags and weights are requested together and the flags tested afterwards
    const int fb_flag = b.scal[f].fallback;
    const uint32_t stop_flag = bound ? *b.bank_flag : 0u;
    double2 wv[NORM_WT_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_WT_IPT / 2; ++k) wv[k] = *reinterpret_cast<const double2*>(w + k * 128);
    if (only_fallback && !fb_flag) return;
    if (bound && stop_flag != 0 && (int64_t)(stop_flag - 1) < kstep) return;      // run_is_stopped
    if (bound && fb_flag) return;
    const double m = bound ? b.scal[f].off_slot[parity] : acc_read_max_wave(acc, parity);

    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < NORM_WT_IPT / 2; ++k) {
        const double e0 = llpf_exp_le0(wv[k].x - m);
        const double e1 = llpf_exp_le0(wv[k].y - m);
        bad = bad || (e0 != e0) || (e1 != e1);
        S = llpf_u128_add(S, llpf_fix96_unit(e0));
        S = llpf_u128_add(S, llpf_fix96_unit(e1));
        if (NEED_E2) {
            E2 = llpf_u128_add(E2, llpf_fix96_unit(e0 * e0));
            E2 = llpf_u128_add(E2, llpf_fix96_unit(e1 * e1));
        }
        ulonglong2 qv;
        qv.x = llpf_q64_unit(e0, K);
        qv.y = llpf_q64_unit(e1, K);
        *reinterpret_cast<ulonglong2*>(qo + k * 128) = qv;
        Q += qv.x;
        Q += qv.y;
    }
    S = wave_sum_u128(S);
    if (NEED_E2) E2 = wave_sum_u128(E2);
    Q = wave_sum_u64(Q);
    const uint64_t bd = (uint64_t)__builtin_popcountll(__ballot(bad));     // only its being non-zero is ever used
    if (lane == 0) {
        const int sh = tile & (NSHARD - 1);
        const uint64_t ls[3] = {S.lo & M43, ((S.lo >> 43) | (S.hi << 21)) & M43, S.hi >> 22};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (ls[k]) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_S(parity) + k, sh)), (unsigned long long)ls[k]);
        if (NEED_E2) {
            const uint64_t le[3] = {E2.lo & M43, ((E2.lo >> 43) | (E2.hi << 21)) & M43, E2.hi >> 22};
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (le[k]) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_E2(parity) + k, sh)), (unsigned long long)le[k]);
        }
        if (tile == 0) {   // the single uniform a systematic resample of this step consumes (reference: rand(), resample.jl:23)
            FilterScal* sc = b.scal + f;
            sc->u_slot[parity] = llpf_uniform_step(sc->step_base + step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1);
            sc->e2v_slot[parity] = NEED_E2 ? 1 : 0;
            sc->exact_slot[parity] = 0;
            sc->xm_parts = b.P2;
        }
        if (bd) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(parity), sh)), (unsigned long long)bd);
        tileq_slot(b, parity, f)[tile] = Q;
    }
}

