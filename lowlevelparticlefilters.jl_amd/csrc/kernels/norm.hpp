// kernels/norm.hpp — k_norm (exp-weights, fixed-point sums, quanta), k_ess, k_post_predict.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// k_norm — exp-weights and their exact sums  (logsumexp! utils.jl:18-27, sum_all_but :66-71,
// effective_particles resample.jl:1-2; optional weighted_mean filtering.jl:541-549)
// ------------------------------------------------------------------------------------------------
// The wave's first requests are addressed from preloaded SGPRs (kernarg preload covers the leading scalar arguments; a struct by value
// in front switches it off): the weights of the tile leave with the first instructions, the filter's flags, the bound and the rest of
// the argument block follow as ONE batch of scalar loads, and the flags are tested afterwards.  Written the other way round — and also
// when the source merely asked for the weights first, without the fence below: the compiler sank the loads behind the tests — the
// flags were four dependent scalar round trips in front of the weights' request (round 5, EXPERIMENTS 5.8).
#define LLPF_NORM_HOT_PARAMS const double* hot_w, const FilterScal* hot_scal, const uint32_t* hot_flag, int64_t hot_Ns
#define LLPF_NORM_HOT_ARGS(b) (b).w, (b).scal, (b).bank_flag, (b).Ns
// the two quanta of a thread: one 16-byte store (LLPF_NORM_ST 2: nontemporal — tools/ab/bign_store_matrix.sh)
#ifndef LLPF_NORM_ST
#define LLPF_NORM_ST 0
#endif
DEV void norm_store_q(uint64_t* p, ulonglong2 qv) {
#if LLPF_NORM_ST == 2
    typedef unsigned long long __attribute__((ext_vector_type(2))) q2_t;
    q2_t v; v.x = qv.x; v.y = qv.y;
    __builtin_nontemporal_store(v, reinterpret_cast<q2_t*>(p));
#else
    *reinterpret_cast<ulonglong2*>(p) = qv;
#endif
}
template <int NX, bool XMEAN, bool NEED_E2>
__global__ __launch_bounds__(BLOCK) void k_norm(LLPF_NORM_HOT_PARAMS, int64_t kstep, int K, int parity, int only_fallback, int bound_in, uint32_t step, BankDev b) {
    // bound_in: bit 0 = bound form, bit 1 = store no quanta (the fused kernel behind this launch forms them itself from the weights, ResArgs::lazy_q:
    // a step that does not resample never reads them, and a step that does reads 8 bytes per particle either way)
    const int bound = bound_in & 1;
    const bool store_q = (bound_in & 2) == 0;
    __shared__ uint64_t sm_u[BLOCK / 64][6];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const double* __restrict__ w = hot_w + (size_t)f * hot_Ns;
    double2 wv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
#if defined(LLPF_NORM_LD_NT) && LLPF_NORM_LD_NT      /* experiment: the weights read nontemporal */
        { typedef double __attribute__((ext_vector_type(2))) d2_t; const d2_t t_ = __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(w + i0)); wv[k].x = t_.x; wv[k].y = t_.y; }
#else
        wv[k] = *reinterpret_cast<const double2*>(w + i0);
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");                 // the requests above are not to be sunk below the tests
    // constant address space: scalar loads whatever the fence says about memory (these words are written by EARLIER launches only)
    const __attribute__((address_space(4))) FilterScal* sc0 = (const __attribute__((address_space(4))) FilterScal*)(hot_scal + f);
    const int fb_flag = sc0->fallback;
    const uint32_t stop_flag = bound ? *(const __attribute__((address_space(4))) uint32_t*)hot_flag : 0u;
    const double m_bound = sc0->off_slot[parity];
    asm volatile("" : : "s"(fb_flag), "s"(stop_flag), "s"(m_bound), "s"(b.acc), "s"(b.quanta), "s"(b.Ns), "s"(b.tileq), "s"(b.P1), "s"(b.P2), "s"(b.xcur),
                 "s"(b.xmpart));
#else
    const int fb_flag = 0; const uint32_t stop_flag = 0; const double m_bound = 0.0;
#endif
    if (only_fallback && !fb_flag) return;
    if (bound && stop_flag != 0 && (int64_t)(stop_flag - 1) < kstep) return;      // run_is_stopped
    if (bound && fb_flag) return;
    uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * b.Ns;
    const double m = bound ? m_bound : acc_read_max_wave(acc, parity);

    llpf_u128 S = {0, 0}, E2 = {0, 0};
    uint64_t Q = 0, bad = 0;
    double xm[NX > 0 ? NX : 1];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) {
        const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
        const double e0 = llpf_exp_le0(wv[k].x - m);
        const double e1 = llpf_exp_le0(wv[k].y - m);
        bad += (e0 != e0) ? 1u : 0u;
        bad += (e1 != e1) ? 1u : 0u;
        S = llpf_u128_add(S, llpf_fix96_unit(e0));
        S = llpf_u128_add(S, llpf_fix96_unit(e1));
        if (NEED_E2) {
            E2 = llpf_u128_add(E2, llpf_fix96_unit(e0 * e0));
            E2 = llpf_u128_add(E2, llpf_fix96_unit(e1 * e1));
        }
        ulonglong2 qv;
        qv.x = llpf_q64_unit(e0, K);
        qv.y = llpf_q64_unit(e1, K);
        if (store_q) norm_store_q(b.quanta + (size_t)f * b.Ns + i0, qv);
        Q += qv.x;
        Q += qv.y;
        if (XMEAN) {
#pragma unroll
            for (int d = 0; d < NX; ++d) {
                const double2 xv = *reinterpret_cast<const double2*>(xc + (size_t)d * b.Ns + i0);
                xm[d] = xm[d] + xv.x * e0;
                xm[d] = xm[d] + xv.y * e1;
            }
        }
    }
    int exact = 0;
    if (bound && gridDim.x == 1) {
        // one-tile filter in the split schedule: the coming head's bound test is made here; if it fails the sums are
        // redone at once against the true maximum (what the host would otherwise ask for in a separate launch)
        __shared__ double sm_m[BLOCK / 64];
        const llpf_u128 sw = wave_sum_u128(S);
        const uint64_t bw = wave_sum_u64(bad);
        if ((threadIdx.x & 63) == 0) { sm_u[threadIdx.x >> 6][0] = sw.lo; sm_u[threadIdx.x >> 6][1] = sw.hi; sm_u[threadIdx.x >> 6][5] = bw; }
        __syncthreads();
        llpf_u128 tot = {sm_u[0][0], sm_u[0][1]};
        uint64_t tb = sm_u[0][5];
        for (int k = 1; k < BLOCK / 64; ++k) { const llpf_u128 t1 = {sm_u[k][0], sm_u[k][1]}; tot = llpf_u128_add(tot, t1); tb += sm_u[k][5]; }
        __syncthreads();
        if (tb || tot.hi < ((uint64_t)1 << 22)) {
            exact = 1;
            double mx = -LLPF_INF;
            bool anynan = false;
#pragma unroll
            for (int k = 0; k < NORM_IPT / 2; ++k) {
                mx = llpf_fmax(mx, wv[k].x); mx = llpf_fmax(mx, wv[k].y);
                anynan = anynan || (wv[k].x != wv[k].x) || (wv[k].y != wv[k].y);
            }
            mx = block_max(mx, sm_m);
            if (__syncthreads_or(anynan ? 1 : 0)) mx = llpf_u2d(0x7ff8000000000000ULL);
            S.lo = 0; S.hi = 0; E2.lo = 0; E2.hi = 0; Q = 0; bad = 0;
#pragma unroll
            for (int d = 0; d < NX; ++d) xm[d] = 0.0;
#pragma unroll
            for (int k = 0; k < NORM_IPT / 2; ++k) {
                const int64_t i0 = (int64_t)tile * TILE + (int64_t)k * (BLOCK * 2) + threadIdx.x * 2;
                const double e0 = llpf_exp_le0(wv[k].x - mx);
                const double e1 = llpf_exp_le0(wv[k].y - mx);
                bad += (e0 != e0) ? 1u : 0u;
                bad += (e1 != e1) ? 1u : 0u;
                S = llpf_u128_add(S, llpf_fix96_unit(e0));
                S = llpf_u128_add(S, llpf_fix96_unit(e1));
                if (NEED_E2) {
                    E2 = llpf_u128_add(E2, llpf_fix96_unit(e0 * e0));
                    E2 = llpf_u128_add(E2, llpf_fix96_unit(e1 * e1));
                }
                ulonglong2 qv;
                qv.x = llpf_q64_unit(e0, K);
                qv.y = llpf_q64_unit(e1, K);
                if (store_q) norm_store_q(b.quanta + (size_t)f * b.Ns + i0, qv);
                Q += qv.x;
                Q += qv.y;
                if (XMEAN) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) {
                        const double2 xv = *reinterpret_cast<const double2*>(xc + (size_t)d * b.Ns + i0);
                        xm[d] = xm[d] + xv.x * e0;
                        xm[d] = xm[d] + xv.y * e1;
                    }
                }
            }
        }
    }
    S = wave_sum_u128(S);
    if (NEED_E2) E2 = wave_sum_u128(E2);
    Q = wave_sum_u64(Q);
    bad = wave_sum_u64(bad);
    if (XMEAN) {
#pragma unroll
        for (int d = 0; d < NX; ++d) xm[d] = wave_sum_f64(xm[d]);
    }
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    if (lane == 0) {
        sm_u[wvid][0] = S.lo; sm_u[wvid][1] = S.hi;
        sm_u[wvid][2] = E2.lo; sm_u[wvid][3] = E2.hi;
        sm_u[wvid][4] = Q; sm_u[wvid][5] = bad;
        if (XMEAN) {
#pragma unroll
            for (int d = 0; d < NX; ++d) sm_x[wvid][d] = xm[d];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        llpf_u128 s = {sm_u[0][0], sm_u[0][1]}, e2 = {sm_u[0][2], sm_u[0][3]};
        uint64_t q = sm_u[0][4], bd = sm_u[0][5];
        for (int k = 1; k < BLOCK / 64; ++k) {
            llpf_u128 t1 = {sm_u[k][0], sm_u[k][1]}, t2 = {sm_u[k][2], sm_u[k][3]};
            s = llpf_u128_add(s, t1);
            e2 = llpf_u128_add(e2, t2);
            q += sm_u[k][4];
            bd += sm_u[k][5];
        }
        acc_add_u128(acc, ACC_S(parity), s);
        if (NEED_E2) acc_add_u128(acc, ACC_E2(parity), e2);
        if (tile == 0) {   // the single uniform a systematic resample of this step consumes (reference: rand(), resample.jl:23)
            FilterScal* sc = b.scal + f;
            sc->u_slot[parity] = llpf_uniform_step(sc->step_base + step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1);
            sc->e2v_slot[parity] = NEED_E2 ? 1 : 0;
            sc->exact_slot[parity] = exact;
            sc->xm_parts = b.P2;
        }
        if (bd) atomicAdd(reinterpret_cast<unsigned long long*>(acc_slot(acc, ACC_BAD(parity), blockIdx.x & (NSHARD - 1))), (unsigned long long)bd);
        tileq_slot(b, parity, f)[tile] = q;
        if (XMEAN) {
            for (int d = 0; d < NX; ++d) {
                double a = sm_x[0][d];
                for (int k = 1; k < BLOCK / 64; ++k) a = a + sm_x[k][d];
                xmpart_slot(b, parity, f)[(size_t)tile * MAXD + d] = a;
            }
        }
    }
}

// accessor path: sum e^2 (fixed point) and ESS of the current weights when the hot loop skipped them
__global__ __launch_bounds__(BLOCK) void k_ess(BankDev b) {
    __shared__ uint64_t sm_u[BLOCK / 64][2];
    const int f = blockIdx.x;
    FilterScal* sc = b.scal + f;
    if (sc->uniform || sc->e2_valid || sc->status) return;
    const double* w = b.w + (size_t)f * b.Ns;
    const double m = sc->m;
    llpf_u128 E2 = {0, 0};
    for (int64_t i = threadIdx.x; i < b.N; i += BLOCK) {
        const double e = llpf_exp_le0(w[i] - m);
        E2 = llpf_u128_add(E2, llpf_fix96_unit(e * e));
    }
    E2 = wave_sum_u128(E2);
    if ((threadIdx.x & 63) == 0) { sm_u[threadIdx.x >> 6][0] = E2.lo; sm_u[threadIdx.x >> 6][1] = E2.hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        llpf_u128 t = {sm_u[0][0], sm_u[0][1]};
        for (int k = 1; k < BLOCK / 64; ++k) { llpf_u128 u = {sm_u[k][0], sm_u[k][1]}; t = llpf_u128_add(t, u); }
        const double e2 = llpf_fix96_to_double(t);
        sc->e2 = e2;
        sc->ess = (sc->stot * sc->stot) / e2;
        sc->e2_valid = 1;
    }
}

// after a propagate-only predict!: reset_weights! if it resampled (reference src/utils.jl:73-79)
// The quanta of the current weights, floor(exp(w - m) 2^K) with the offset and the fraction bits the last head published: what the k_norm of
// the run's last timestep would have stored had it not been told to store none (bound_in bit 1).  Launched once at the end of such a run,
// in front of k_post_predict; a filter whose last predict! resampled has uniform weights and nobody reads its quanta.
__global__ __launch_bounds__(BLOCK) void k_requant(BankDev b) {
    const int f = blockIdx.y;
    const FilterScal* sc = b.scal + f;
    if (sc->do_resample || sc->uniform) return;
    const double m = sc->m;
    const int K = sc->K;
    const int64_t i0 = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) * 2;
    if (i0 >= b.Ns) return;
    const double2 wv = *reinterpret_cast<const double2*>(b.w + (size_t)f * b.Ns + i0);
    ulonglong2 qv;
    qv.x = llpf_q64_unit(llpf_exp_le0(wv.x - m), K);
    qv.y = llpf_q64_unit(llpf_exp_le0(wv.y - m), K);
    *reinterpret_cast<ulonglong2*>(b.quanta + (size_t)f * b.Ns + i0) = qv;
}

__global__ void k_post_predict(BankDev b) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= b.F) return;
    FilterScal* sc = b.scal + f;
    if (sc->do_resample) {
        sc->uniform = 1;
        sc->wconst = b.log1N;
        sc->m = 0.0;
        sc->mtrue = 0.0;             // maxw[] = 0
        sc->wmax = b.log1N;
        sc->norm_pending = 0;
    }
    sc->do_resample = 0;
}

// host side of a failed bound test, for the filters whose `fallback` flag is set: mode 0 zeroes the exp-sum words of
// accumulator slot `slot` (the exact-form k_norm accumulates them afresh), mode 1 clears the flags and the run's stop word
__global__ void k_fb_clear(BankDev b, int slot, int mode) {
    const int f = blockIdx.x;
    FilterScal* sc = b.scal + f;
    if (!sc->fallback) return;
    if (mode == 0) {
        uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
        const int words[7] = {ACC_S(slot), ACC_S(slot) + 1, ACC_S(slot) + 2, ACC_E2(slot), ACC_E2(slot) + 1, ACC_E2(slot) + 2, ACC_BAD(slot)};
        for (int i = threadIdx.x; i < 7 * NSHARD * ACC_STRIDE; i += blockDim.x)
            acc[(size_t)words[i / (NSHARD * ACC_STRIDE)] * NSHARD * ACC_STRIDE + (i % (NSHARD * ACC_STRIDE))] = 0;
    } else if (threadIdx.x == 0) {
        sc->fallback = 0;
    }
}
__global__ void k_fb_clear_flag(BankDev b) { if (threadIdx.x < 4) b.bank_flag[threadIdx.x] = 0; }

// per-tile sums of the quanta of plain values (standalone resample(we))
__global__ __launch_bounds__(BLOCK) void k_qpart(BankDev b, int K) {
    __shared__ uint64_t sm_w[BLOCK / 64];
    const int f = blockIdx.y, tile = blockIdx.x;
    const double* __restrict__ w = b.w + (size_t)f * b.Ns;
    uint64_t Q = 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        const int64_t i = (int64_t)tile * TILE + (int64_t)k * BLOCK + threadIdx.x;
        const uint64_t q = (i < b.N) ? llpf_q64_unit(w[i], K) : 0;
        b.quanta[(size_t)f * b.Ns + i] = q;
        Q += q;
    }
    Q = wave_sum_u64(Q);
    if ((threadIdx.x & 63) == 0) sm_w[threadIdx.x >> 6] = Q;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t q = 0;
        for (int k = 0; k < BLOCK / 64; ++k) q += sm_w[k];
        tileq_slot(b, 0, f)[tile] = q;      // scratch bank of the standalone resample(we): slot 0
    }
}
