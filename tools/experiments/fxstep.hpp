// kernels/fxstep.hpp — k_fxstep: one timestep of a filter whose dynamics are worth a table (quad-tank, run-time compiled models) in ONE
// launch.  Part of k_step.hip, namespace llpf; needs kernels/resample.hpp, resfx.hpp (eval / marks conventions), gridbar.hpp.
// ------------------------------------------------------------------------------------------------
// The two-launch form (k_resample_fx, then k_step<..., MARKS>) spends a third of the timestep in a resampling launch that is one long
// latency chain — cold first loads, head, scan, counts, survivors, their x, the RK4 — during which the SIMDs idle, and two thirds in a
// step launch that is issue-bound on the process noise, which depends on nothing but a counter (BASELINE C3, tools/dbg/qt_phases.py:
// resampling blocks live 22 k cycles and issue for ~2 k of them; 60 % of the step kernel's loop is Philox + Box-Muller).  Here one
// workgroup per source tile does both, all workgroups resident:
//   R  the resampling of its tile exactly as k_resample_fx does it (same head, same counts, f(x_j) once per surviving source into
//      BankDev::fxs, run-start marks into BankDev::mark), with the NOISE of the workgroup's own 1024 outputs drawn inside the chain's
//      gaps — the first 512 behind the head, the second 512 while the survivors' x travel — and kept in registers;
//   —  a barrier over the grid (kernels/gridbar.hpp; marks and f(x_j) cross workgroups through agent-scope accesses);
//   S  for its two 512-particle output tiles: marks -> ancestors (inclusive max-scan), gather f(x[ancestor]), + noise, store, weight,
//      exp-sums, quanta — the body of k_step<..., MARKS> without the generator.
// Same arithmetic, statement by statement, as the two-launch form (which remains: banks, filters larger than the resident set,
// history outputs): bit-identical results.  Reference: predict! src/filtering.jl:140-153 (resample src/resample.jl:17-61,
// propagate_particles! src/PFtypes.jl:242-259, reset_weights! src/utils.jl:73-79) + the weighting of the next correct!
// (measurement_equation! src/PFtypes.jl:226-239).
// ------------------------------------------------------------------------------------------------
template <class Model, int NX, int NY, int MODE, int STRATEGY>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_fxstep(BankDev b, ResArgs a, StepArgs st, uint32_t* bar) {
    static_assert(share_dynamics<Model>::value && !Model::RB && (MODE == MODE_PROP || MODE == MODE_PROP_WEIGHT), "k_fxstep: propagating modes of models whose dynamics are worth a table");
    constexpr int PPT = STEP_PPT;
    static_assert(TILE == 2 * BLOCK * PPT, "a source tile is two output tiles");
    __shared__ ResShared sh;
    __shared__ uint32_t sh_list[TILE];
    __shared__ uint32_t sh_heavy[TILE];
    __shared__ uint32_t sh_cnt[2];
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_acc[BLOCK / 64][5];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    __shared__ int32_t sh_mcnt[2][BLOCK / 64];
    __shared__ __attribute__((aligned(16))) double sh_rng_lg[2 * LLPF_RNG_LG_ENTRIES], sh_rng_sc[2 * LLPF_RNG_SC_ENTRIES];
    // the noise of the second output tile waits here (16 KB; in registers next to the first tile's it pushed the kernel into scratch)
    __shared__ __attribute__((aligned(16))) double sh_nzB[BLOCK][PPT * NX];
    const int f = 0;                     // single filter: the grid barrier counts the blocks along x
    const int tile = blockIdx.x;
    const int t = (int)threadIdx.x, lane = t & 63;
    const ModelD* md = b.models + f;
    FilterScal* sc = b.scal + f;
    const bool dbg_on = st.k == 5; (void)dbg_on;
    FX_STAMP(0, t); DBG_HWID(g_fx_dbg);
    // ---- everything the prologue reads is requested first ----
    const uint32_t stop_flag = *b.bank_flag;
    const int fb_flag = sc->fallback;
    const int status0 = sc->status;
    uint32_t gen = grid_barrier_generation(bar);
    double rt0 = 0.0, rt1 = 0.0;
    if (t < LLPF_RNG_SC_ENTRIES) { rt0 = LLPF_SIN64[t]; rt1 = LLPF_COS64[t]; }
    else if (t < LLPF_RNG_SC_ENTRIES + LLPF_RNG_LG_ENTRIES) { rt0 = LLPF_LOG_INVC[t - LLPF_RNG_SC_ENTRIES]; rt1 = LLPF_LOG_LNC[t - LLPF_RNG_SC_ENTRIES]; }
    const uint32_t k0 = sc->k0, k1 = sc->k1, sb = sc->step_base;
    if (t < 2) sh_cnt[t] = 0;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)t * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
    Model model;
    auto overlap = [&]() {
#pragma unroll
        for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
        model.prepare(md, st.u, st.t_prop);
        // the generator's tables into LDS: visible behind the head's barrier
        if (t < LLPF_RNG_SC_ENTRIES) { sh_rng_sc[2 * t] = rt0; sh_rng_sc[2 * t + 1] = rt1; }
        else if (t < LLPF_RNG_SC_ENTRIES + LLPF_RNG_LG_ENTRIES) { sh_rng_lg[2 * (t - LLPF_RNG_SC_ENTRIES)] = rt0; sh_rng_lg[2 * (t - LLPF_RNG_SC_ENTRIES) + 1] = rt1; }
    };
    const ResHead h = res_head<SRC_FILTER>(b, a, f, tile, sh, true, stop_flag, fb_flag, nullptr, overlap);
    FX_STAMP(1, h.tot);
    // every workgroup derives the same scalars from the same integers: the exits below are taken by all of them or by none
    if (h.status == RES_STATUS_SKIP) return;
    if (h.status) return;                                  // failed bound test (the host redoes the step) or degenerate weights
    if (__builtin_amdgcn_readfirstlane(status0) == LLPF_STATUS_BARRIER_TIMEOUT) return;   // an earlier barrier of this filter gave up: do not wait again
    const size_t Ns = (size_t)b.Ns;
    const int64_t N = b.N;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * Ns;
    double* __restrict__ xn = b.xnext + (size_t)f * NX * Ns;
    double* __restrict__ fxo = b.fxs + (size_t)f * NX * Ns;
    int32_t* __restrict__ mk = b.mark + (size_t)f * Ns;
    const bool res = (h.dr || a.force);                    // what tile 0 publishes as FilterScal::do_resample
    // ---- noise of the workgroup's own outputs (tile A = 2 * tile, tile B = 2 * tile + 1), two particles per thread ----
    auto draw = [&](const int64_t i0, double (&nz)[PPT][NX]) {
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            double xi[NX];
            llpf_normals_tab((uint32_t)(i0 + p), sb + st.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi, sh_rng_lg, sh_rng_sc);
            gauss_sample<NX>(md->df, xi, nz[p]);
        }
    };
    const int64_t iA = (int64_t)(2 * tile) * (BLOCK * PPT) + (int64_t)t * PPT, iB = iA + BLOCK * PPT;
    double nzA[PPT][NX];
    auto draw_B = [&]() {
        double nz[PPT][NX];
        draw(iB, nz);
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
#pragma unroll
            for (int d = 0; d < NX; ++d) sh_nzB[t][p * NX + d] = nz[p][d];
        }
    };
    draw(iA, nzA);
    FX_STAMP(2, nzA[0][0]);
    auto eval_store = [&](const int32_t j, const double* xq) {       // f(x_j) into the plane
        double fq[NX];
        model.dynamics(xq, fq);
#pragma unroll
        for (int d = 0; d < NX; ++d) wt_store(fxo + (size_t)d * Ns + j, fq[d]);
    };
    auto unowned = [&](const int64_t o_begin) {                      // as k_resample_fx
        const int ident = sc->anc_ident_s[b.anc_slot];
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
    if (!res) {
        // no resampling in this predict!: j = 1:N — f(x_i) of the tile's particles (the workgroup's own outputs)
        draw_B();
#pragma unroll 1
        for (int k = 0; k < NORM_IPT; ++k) {
            const int32_t j = (int32_t)((int64_t)tile * TILE + k * BLOCK + t);
            double xq[NX];
#pragma unroll
            for (int d = 0; d < NX; ++d) xq[d] = xc[(size_t)d * Ns + j];
            eval_store(j, xq);
        }
    } else if (h.tot == 0) {
        draw_B();
        if (tile == b.P2 - 1) unowned(0);
    } else {
        int32_t c_start, c_end;
        res_counts<STRATEGY>(b, a, f, tile, h, qv, sh, c_start, c_end);
        FX_STAMP(3, c_end);
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
        // survivor q is handled by thread (q + 64 * (tile & 3)) mod BLOCK: with a handful of survivors per tile the RK4 is one wave's
        // work, and the four workgroups of a CU would otherwise all put it on SIMD 0
        const int tq = (t - 64 * (tile & 3)) & (BLOCK - 1);
        double xq0[NX];
        const int32_t j0 = (tq < D) ? (int32_t)((int64_t)tile * TILE + (int)sh_list[tq]) : -1;
        if (j0 >= 0) {
#pragma unroll
            for (int d = 0; d < NX; ++d) xq0[d] = xc[(size_t)d * Ns + j0];
        }
#pragma unroll 1
        for (int q = tq; q < D; q += BLOCK) {
            const uint32_t idx = sh_list[q];
            const int32_t j = (int32_t)((int64_t)tile * TILE + (int)idx);
            const uint32_t lo = idx ? sh.cl[idx - 1] : (uint32_t)c_start, hi = sh.cl[idx];
            wt_store(mk + lo, j + 1);
            uint32_t bnd = (lo / STEP_TILE + 1) * STEP_TILE;
#pragma unroll 1
            for (int n = 0; n < FX_INLINE_BND && bnd < hi; ++n, bnd += STEP_TILE) wt_store(mk + bnd, j + 1);
            if (bnd < hi) sh_heavy[atomicAdd(&sh_cnt[1], 1u)] = idx;
        }
        draw_B();                                              // while the survivors' x travel
        FX_STAMP(4, t);
        __syncthreads();
        const int H = (int)sh_cnt[1];
        for (int hq = 0; hq < H; ++hq) {                      // block-uniform
            const uint32_t idx = sh_heavy[hq];
            const uint32_t lo = idx ? sh.cl[idx - 1] : (uint32_t)c_start, hi = sh.cl[idx];
            const int32_t j = (int32_t)((int64_t)tile * TILE + (int)idx);
            for (uint32_t bnd = (lo / STEP_TILE + 1 + FX_INLINE_BND + (uint32_t)t) * STEP_TILE; bnd < hi; bnd += BLOCK * STEP_TILE) wt_store(mk + bnd, j + 1);
        }
#pragma unroll 1
        for (int q = tq; q < D; q += BLOCK) {
            const int32_t j = (q == tq) ? j0 : (int32_t)((int64_t)tile * TILE + (int)sh_list[q]);
            double xq[NX];
#pragma unroll
            for (int d = 0; d < NX; ++d) xq[d] = (q == tq) ? xq0[d] : xc[(size_t)d * Ns + j];
            eval_store(j, xq);
        }
        if (tile == b.P2 - 1 && c_end < a.M) unowned((int64_t)c_end);
    }
    FX_STAMP(5, t);
    if (!grid_barrier(bar, gen, (int)gridDim.x)) {
        if (t == 0) sc->status = LLPF_STATUS_BARRIER_TIMEOUT;
        return;
    }
    FX_STAMP(6, t);

    // ---- S: the two output tiles of this workgroup (k_step<..., MARKS = true> without the generator) ----
    const int do_res = res ? 1 : 0;
    double y[NY];
    if (MODE != MODE_PROP) {
#pragma unroll
        for (int k = 0; k < NY; ++k) y[k] = st.has_y ? st.y[k] : 0.0;
    }
    double* w = b.w + (size_t)f * Ns;
    // what k_step reads back from FilterScal after the head has published it, derived here from the same head
    const double hl = res ? 0.0 : head_log(h);             // l of the pending normalisation (w - m) - l
    double bmax = -LLPF_INF;
    bool bad = false;
    double off = 0.0;
    WeightAcc wacc;
    double xm[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
    if (MODE != MODE_PROP) {
        const double wmx = do_res ? b.log1N : (h.mtrue - h.a) - hl;       // FilterScal::wmax of the finalized weights
        off = st.has_y ? wmx + md->dg.c0 : wmx;
        wacc.init();
    }
    auto s_tile = [&](const int64_t i0, const double (&nz)[PPT][NX], const int itn) {
        double fsh[PPT][NX];
        const double* fxp = b.fxs + (size_t)f * NX * Ns;
        if (do_res) {
            int32_t* mkp = mk + i0;
            const uint64_t mraw = Mem<1>::ld(reinterpret_cast<const uint64_t*>(mkp));
            const uint32_t m0 = (uint32_t)mraw, m1 = (uint32_t)(mraw >> 32);
            if (mraw) *reinterpret_cast<uint64_t*>(mkp) = 0;
            const uint32_t incl = wave_scan_max_u32(m1 > m0 ? m1 : m0);
            uint32_t base = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
            int32_t* wc = sh_mcnt[itn & 1];
            if (lane == 63) wc[t >> 6] = (int32_t)incl;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < BLOCK / 64 - 1; ++k) { const uint32_t c = (uint32_t)wc[k]; if (k < (t >> 6)) base = c > base ? c : base; }
            const uint32_t s0 = m0 > base ? m0 : base, s1 = m1 > s0 ? m1 : s0;
            const uint32_t am[2] = {(m0 & (uint32_t)MARK_OWN) ? m0 : s0, (m1 & (uint32_t)MARK_OWN) ? m1 : s1};
            int32_t av[PPT];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                const int32_t v = (int32_t)(am[p] & ~(uint32_t)MARK_OWN) - 1;
                av[p] = v < 0 ? 0 : v;
            }
            { int2 ao; ao.x = (i0 < N) ? av[0] : (int32_t)i0; ao.y = (i0 + 1 < N) ? av[1] : (int32_t)(i0 + 1);
              wt_store(reinterpret_cast<int2*>(b.anc + (size_t)f * Ns + i0), ao); }
#pragma unroll
            for (int d = 0; d < NX; ++d) {
#pragma unroll
                for (int p = 0; p < PPT; ++p) fsh[p][d] = Mem<1>::ld(fxp + (size_t)d * Ns + av[p]);
            }
        } else {
#pragma unroll
            for (int d = 0; d < NX; ++d) {
#pragma unroll
                for (int p = 0; p < PPT; ++p) fsh[p][d] = Mem<1>::ld(fxp + (size_t)d * Ns + i0 + p);
            }
        }
        double xs[PPT][NX];
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
#pragma unroll
            for (int d = 0; d < NX; ++d) xs[p][d] = fsh[p][d] + nz[p][d];
        }
#pragma unroll
        for (int d = 0; d < NX; ++d) { double2 v; v.x = xs[0][d]; v.y = xs[1][d]; wt_store(reinterpret_cast<double2*>(xn + (size_t)d * Ns + i0), v); }
        if (MODE != MODE_PROP) {
            double wp[PPT];
            if (do_res) {                          // reset_weights!: w = log(1/N)
#pragma unroll
                for (int p = 0; p < PPT; ++p) wp[p] = b.log1N;
            } else {
                const double2 wv = *reinterpret_cast<const double2*>(w + i0);
                wp[0] = (wv.x - h.a) - hl; wp[1] = (wv.y - h.a) - hl;      // lazy w .-= offset ; w .-= log1p(s)
            }
            double wn[PPT];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                double wv = wp[p];
                if (st.has_y) {
                    if constexpr (has_loglik<Model>::value) {
                        wv = wv + model.loglik(xs[p], y, st.t_meas);
                    } else {
                        double g[NY], v[NY];
                        model.measurement(xs[p], g);
#pragma unroll
                        for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                        wv = wv + gauss_logpdf<NY>(md->dg, v);
                    }
                }
                if (i0 + p >= N) wv = -LLPF_INF;   // padding lanes carry zero weight
                wn[p] = wv;
                bad = bad || (wv != wv);
                if constexpr (has_loglik<Model>::value) bad = bad || (st.has_y && wv > off);
                bmax = llpf_fmax(bmax, wv);
            }
            { double2 wo; wo.x = wn[0]; wo.y = wn[1]; wt_store(reinterpret_cast<double2*>(w + i0), wo); }
            if (st.accumulate) {
                uint64_t qv2[PPT], qsum = 0;
                double ev[PPT];
#pragma unroll
                for (int p = 0; p < PPT; ++p) { qv2[p] = wacc.add(wn[p], off, st.K, st.need_e2 != 0, &ev[p]); qsum += qv2[p]; }
                { ulonglong2 q2; q2.x = qv2[0]; q2.y = qv2[1]; wt_store(reinterpret_cast<ulonglong2*>(b.quanta_next + (size_t)f * Ns + i0), q2); }
                if (st.want_xmean) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) {
#pragma unroll
                        for (int p = 0; p < PPT; ++p) xm[d] = xm[d] + xs[p][d] * ev[p];
                    }
                }
                qsum = wave_sum_u64(qsum);
                if (lane == 0 && qsum)
                    atomicAdd(reinterpret_cast<unsigned long long*>(tileq_slot(b, st.parity, f) + (i0 / TILE)), (unsigned long long)qsum);
            }
        }
    };
    s_tile(iA, nzA, 0);
    {
        double nzB[PPT][NX];             // written by this thread itself: no synchronisation needed
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
#pragma unroll
            for (int d = 0; d < NX; ++d) nzB[p][d] = sh_nzB[t][p * NX + d];
        }
        s_tile(iB, nzB, 1);
    }
    FX_STAMP(7, bmax);
    if (MODE != MODE_PROP) {
        const double r = block_max(bmax, sm_max);
        const int anybad = __syncthreads_or(bad ? 1 : 0);
        if (t == 0) acc_max(b.acc + (size_t)f * ACC_WORDS, st.parity, r, anybad != 0);
        if (st.accumulate) wacc.flush(b.acc + (size_t)f * ACC_WORDS, st.parity, st.need_e2 != 0, sm_acc);
        if (st.accumulate && st.want_xmean) block_store_xm<NX>(xm, b.xmpart + ((size_t)f * b.P1 + blockIdx.x) * MAXD, sm_x);
        if (blockIdx.x == 0 && t == 0) {
            if (st.accumulate) sc->xm_parts = (int32_t)gridDim.x;
            sc->off_slot[st.parity] = off;
            sc->exact_slot[st.parity] = 0;
            sc->e2v_slot[st.parity] = st.need_e2;
            sc->u_slot[st.parity] = llpf_uniform_step(sb + st.next_step, LLPF_STREAM_RESAMPLE, k0, k1);
        }
    }
    FX_STAMP(8, t);
    if (blockIdx.x == 0 && t == 0) {
        sc->anc_ident_s[b.anc_slot ^ 1] = do_res ? 0 : 1;
        sc->last_resampled = do_res;
        sc->resample_count += do_res;
    }
}
