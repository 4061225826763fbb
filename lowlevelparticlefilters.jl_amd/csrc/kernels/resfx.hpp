// kernels/resfx.hpp — k_resample_fx: the resampling launch of the balanced form with the dynamics evaluated on the SOURCE side.
// Part of k_step.hip (the translation unit that knows the models), namespace llpf; needs kernels/resample.hpp.
// ------------------------------------------------------------------------------------------------
// x'_i = f(x_{j_i}, u, p, t) + noise_i  (reference src/PFtypes.jl:130-136 / :254-256): outputs that share an ancestor share f(x_j).
// Round 3 found the distinct ancestors per k_step BLOCK (1-3 RK4 evaluations per 512 outputs at BASELINE C3, but every heavy ancestor
// again in each of the hundreds of blocks it spans, four barriers and a phase in which a few lanes run the RK4 while the block waits).
// Here the block that owns a source tile — it has just formed the ancestor counts c(bins) of its 1024 sources (res_counts) — evaluates
// f(x_j) ONCE for every source j with a non-empty output range and stores it in a scratch plane (BankDev::fxs); k_step gathers
// fxs[ancestor] and adds noise and weight only.  The number of evaluations is the number of distinct ancestors (<= N, ~1e-3 N at C3).
//
// Ancestors are not written per output either (a tile that holds a very heavy particle had to store hundreds of thousands of
// identical indices through ONE CU): the launch leaves run-start MARKS — 1 + j at the first output of source j and at every k_step
// tile boundary inside its output range — in an array that is zero otherwise (BankDev::mark).  k_step<..., MARKS = true> turns the
// STEP_TILE marks of a tile into ancestors with one inclusive max-scan (ancestors of systematic / stratified resampling are
// non-decreasing), clears what it read and writes the ancestor array for the accessors.  Same thresholds, same counts, same
// comparison as k_resample: same ancestors.
// So that the step kernel needs no dynamics at all (inside its tile loop they cost it a wave per SIMD, step.hpp), this launch provides
// f for EVERY source the step will ask for: the survivors; the previous ancestors of outputs whose threshold is >= bins[N] (the
// reference writes nothing there, resample.jl:27-35: j keeps its previous value) — such an output gets a mark of its own, flagged
// MARK_OWN —; and, when shouldresample says no, all particles (j = 1:N, filtering.jl:148).
// ------------------------------------------------------------------------------------------------
constexpr int FX_INLINE_BND = 8;    // tile boundaries a lane marks by itself; longer ranges are finished by its wave

template <class Model, int NX, int STRATEGY>
__global__ __launch_bounds__(BLOCK) void k_resample_fx(BankDev b, ResArgs a, StepArgs st) {
    __shared__ ResShared sh;
    __shared__ uint32_t sh_list[TILE];       // tile-local indices of the surviving sources (any order: each is handled on its own)
    __shared__ uint32_t sh_cnt[2];
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const int t = (int)threadIdx.x, lane = t & 63;
    const bool dbg_on = st.k == 5; (void)dbg_on;
    FX_STAMP(0, t); DBG_HWID(g_fx_dbg);
    const uint32_t stop_flag = *b.bank_flag;
    const int fb_flag = b.scal[f].fallback;
    if (t < 2) sh_cnt[t] = 0;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)t * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
    Model model;
    auto overlap = [&]() {
#pragma unroll
        for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
        model.prepare(b.models + f, st.u + (size_t)f * st.u_stride, st.t_prop);   // particle-independent terms: scalar loads of their own
    };
    const ResHead h = res_head<SRC_FILTER>(b, a, f, tile, sh, true, stop_flag, fb_flag, overlap);
    FX_STAMP(1, h.tot);                 // head: loads back, scalars derived
    if (h.status == RES_STATUS_SKIP) return;
    if (h.status) return;
    const size_t Ns = (size_t)b.Ns;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * Ns;
    double* __restrict__ fxo = b.fxs + (size_t)f * NX * Ns;
    int32_t* __restrict__ mk = b.mark + (size_t)f * Ns;
    auto eval_store = [&](const int32_t j, const double* xq) {       // f(x_j) into the plane
        double fq[NX];
        model.dynamics(xq, fq);
#pragma unroll
        for (int d = 0; d < NX; ++d) wt_store(fxo + (size_t)d * Ns + j, fq[d]);
    };
    // outputs [o_begin, M) have no owner: the ancestor is the previous one (the identity if the last predict! did not resample)
    auto unowned = [&](const int64_t o_begin) {
        const int ident = b.scal[f].anc_ident_s[b.anc_slot];
        const int32_t* anc = b.anc + (size_t)f * Ns;
#pragma unroll 1
        for (int64_t o = o_begin + t; o < (int64_t)a.M; o += BLOCK) {
            const int32_t aj = ident ? (int32_t)o : anc[o];
            double xq[NX];
#pragma unroll
            for (int d = 0; d < NX; ++d) xq[d] = xc[(size_t)d * Ns + aj];
            eval_store(aj, xq);
            wt_store(mk + o, (int32_t)(MARK_OWN | (aj + 1)));
        }
    };
    if (!a.force && !h.dr) {
        if (t == 0) { const int64_t left = b.N - (int64_t)tile * TILE; if (left > 0) b.surv[((size_t)f * b.P2 + tile) * 4] += (unsigned long long)(left < TILE ? left : TILE); }
        // no resampling in this predict!: j = 1:N, every particle propagates itself — f(x_i) for the tile's particles
#pragma unroll 1
        for (int k = 0; k < NORM_IPT; ++k) {
            const int32_t j = (int32_t)((int64_t)tile * TILE + k * BLOCK + t);
            double xq[NX];
#pragma unroll
            for (int d = 0; d < NX; ++d) xq[d] = xc[(size_t)d * Ns + j];
            eval_store(j, xq);
        }
        return;
    }
    if (h.tot == 0) {                    // every bin is empty: no threshold has an owner
        if (tile == b.P2 - 1) unowned(0);
        return;
    }
    int32_t c_start, c_end;
    res_counts<STRATEGY>(b, a, f, tile, h, qv, sh, c_start, c_end);
    FX_STAMP(2, c_end);                 // scan + counts

    // ---- the surviving sources of this tile: those with a non-empty output range [prev, cur) ----
    {
        const uint4 c4 = *reinterpret_cast<const uint4*>(sh.cl + 4 * t);
        uint32_t prev = t ? sh.cl[4 * t - 1] : (uint32_t)c_start;
        const uint32_t cur[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool surv = cur[k] > prev;
            const uint64_t mask = __ballot(surv);
            if (mask) {                                   // wave-uniform
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&sh_cnt[0], (uint32_t)__popcll(mask));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (surv) sh_list[base + rank] = (uint32_t)(4 * t + k);
            }
            prev = cur[k];
        }
    }
    __syncthreads();
    const int D = (int)sh_cnt[0];
    FX_STAMP(3, D);                     // survivors listed
    if (t == 0 && D) b.surv[((size_t)f * b.P2 + tile) * 4] += (unsigned long long)D;      // what the host chooses the next run's form by (host/run.hpp)

    // A handful of survivors (peaked likelihoods: ~8 per tile at BASELINE C3) and a model whose dynamics can run with its states
    // spread over a quad: four lanes per survivor, a third of the dependent chain (QuadTank::dynamics_quad).  Survivor q sits on lanes
    // 4q .. 4q+3 counted from wave (tile & 3): the four workgroups of a CU would otherwise all put their one busy wave on SIMD 0.
    bool quad = false;                   // block-uniform
    if constexpr (has_quad_dynamics<Model>::value && NX == 4) quad = D <= BLOCK / 4;
    const int tq = (t - 64 * (tile & 3)) & (BLOCK - 1);
    const int qstep = quad ? BLOCK / 4 : BLOCK;
    const int q0 = quad ? (tq >> 2) : tq, cq = tq & 3;
    // first pass over the survivors: request x_j (the first round of them: a tile with more has near-uniform weights), leave the marks
    // a lane can leave by itself
    double xq0[NX];
    const int32_t j0 = (q0 < D) ? (int32_t)((int64_t)tile * TILE + (int)sh_list[q0]) : -1;
    if (j0 >= 0) {
        if (quad) xq0[0] = xc[(size_t)cq * Ns + j0];
        else {
#pragma unroll
            for (int d = 0; d < NX; ++d) xq0[d] = xc[(size_t)d * Ns + j0];
        }
    }
    // run-start marks: the first output of the source and every k_step tile boundary inside its range.  A lane leaves the first
    // FX_INLINE_BND boundaries itself; the rest of a long range (a heavy particle: up to N / STEP_TILE boundaries) is written by the
    // whole wave, one such source after the other — no block-wide list, no barrier
#pragma unroll 1
    for (int qb = 0; qb < D; qb += qstep) {              // block-uniform trip count
        const int q = qb + q0;
        uint32_t bnd = 0, hi = 0;
        int32_t j = 0;
        if (q < D && (!quad || cq == 0)) {
            const uint32_t idx = sh_list[q];
            j = (int32_t)((int64_t)tile * TILE + (int)idx);
            const uint32_t lo = idx ? sh.cl[idx - 1] : (uint32_t)c_start;
            hi = sh.cl[idx];
            wt_store(mk + lo, j + 1);
            bnd = (lo / STEP_TILE + 1) * STEP_TILE;
#pragma unroll 1
            for (int n = 0; n < FX_INLINE_BND && bnd < hi; ++n, bnd += STEP_TILE) wt_store(mk + bnd, j + 1);
        }
        uint64_t more = __ballot(bnd < hi);
        while (more) {                                    // wave-uniform
            const int l = __builtin_ctzll(more);
            more &= more - 1;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bnd, l), h0 = (uint32_t)__builtin_amdgcn_readlane((int)hi, l);
            const int32_t jl = __builtin_amdgcn_readlane(j, l);
            for (uint32_t bb = b0 + (uint32_t)lane * STEP_TILE; bb < h0; bb += 64u * STEP_TILE) wt_store(mk + bb, jl + 1);
        }
    }
    FX_STAMP(4, xq0[0]);                // x of the survivors back, marks left
    FX_STAMP(5, t);
    // the dynamics, once per surviving source
    if constexpr (has_quad_dynamics<Model>::value && NX == 4) {
        if (quad) {
            if (j0 >= 0) wt_store(fxo + (size_t)cq * Ns + j0, model.dynamics_quad(xq0[0], cq));      // D <= BLOCK / 4: one round
        }
    }
    if (!quad) {
#pragma unroll 1
        for (int q = q0; q < D; q += BLOCK) {
            const int32_t j = (q == q0) ? j0 : (int32_t)((int64_t)tile * TILE + (int)sh_list[q]);
            double xq[NX];
#pragma unroll
            for (int d = 0; d < NX; ++d) xq[d] = (q == q0) ? xq0[d] : xc[(size_t)d * Ns + j];
            eval_store(j, xq);
        }
    }
    // outputs whose threshold is >= bins[N] (at most a few, at the very end)
    if (tile == b.P2 - 1 && c_end < a.M) unowned((int64_t)c_end);
    FX_STAMP(6, t);                     // dynamics evaluated and stored
#if defined(LLPF_STEP_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    if (t == 0 && blockIdx.y == 0 && g_fx_dbg && dbg_on) { g_fx_dbg[(size_t)blockIdx.x * 16 + 8] = (unsigned long long)D; g_fx_dbg[(size_t)blockIdx.x * 16 + 9] = 0; g_fx_dbg[(size_t)blockIdx.x * 16 + 10] = (unsigned long long)(c_end - c_start); }
#endif
}
