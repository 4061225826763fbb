// kernels/step.hpp — k_step (balanced propagate + weight) and k_max.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// k_step — fused propagate + weight + running max
// ------------------------------------------------------------------------------------------------
// PPT particles per thread (16-B vector accesses at 2; 1 halves the registers: models whose dynamics dominate, e.g. the quad-tank's
// RK4 with 32 square roots per particle, gain more from the doubled occupancy than they lose on 8-B accesses)
#if defined(LLPF_STEP_TIMING) && defined(__HIP_DEVICE_COMPILE__)   // developer build: phase stamps (tools/dbg/qt_phases.py)
#define DBG_STAMP(arr, k, dep) do { asm volatile("" : : "v"(dep)); if (threadIdx.x == 0 && blockIdx.y == 0 && arr && dbg_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); arr[(size_t)blockIdx.x * 16 + (k)] = t_; } } while (0)
#define DBG_HWID(arr) do { if (threadIdx.x == 0 && blockIdx.y == 0 && arr && dbg_on) { uint32_t hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); arr[(size_t)blockIdx.x * 16 + 13] = hw_; arr[(size_t)blockIdx.x * 16 + 14] = xcc_; } } while (0)
#else
#define DBG_STAMP(arr, k, dep) ((void)0)
#define DBG_HWID(arr) ((void)0)
#endif
#define STEP_STAMP(k, dep) DBG_STAMP(g_step_dbg, k, dep)
#define FX_STAMP(k, dep) DBG_STAMP(g_fx_dbg, k, dep)

// MARKS = true: the form that follows k_resample_fx (kernels/resfx.hpp).  The resampling launch has evaluated f(x_j) for every
// source the step needs (BankDev::fxs) and left run-start marks instead of ancestors (BankDev::mark): the kernel contains NO
// dynamics, a block walks its 512-particle tiles grid-stride (launched with four workgroups per CU: one prologue and one set of
// reductions per block, the next tile's marks in flight while this one is computed), and per tile it turns the marks into ancestors
// with one inclusive max-scan, clears them, writes the ancestor array for the accessors, gathers fxs[ancestor] and adds noise and
// weight.  The dynamics must stay out of this loop: hoisted out of it their particle-independent terms (the quad-tank's input terms
// and stage times) stay live across it and cost a wave per SIMD.  MARKS = false: one tile per block, ancestors from HBM, f inline.
template <class Model, int NX, int NY, int MODE, int PPT = STEP_PPT, bool MARKS = false>
__global__ __launch_bounds__(BLOCK) void k_step(BankDev b, const ModelD* __restrict__ models,
                                                 const FilterScal* scal, StepArgs a) {
    static_assert(!MARKS || (!Model::RB && PPT == STEP_PPT && (MODE == MODE_PROP || MODE == MODE_PROP_WEIGHT)),
                  "the marks form: propagating modes, two particles per thread (launched for models whose dynamics are worth a table)");
    __shared__ uint64_t sm_fl[BLOCK / 64][8];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    // dynamics shared between the outputs of one ancestor (below): models whose f is worth a table
    constexpr bool SHARE = !MARKS && share_dynamics<Model>::value && !Model::RB && MODE != MODE_WEIGHT && MODE != MODE_AUX;
    __shared__ int32_t sh_prev[SHARE ? BLOCK : 1], sh_run[SHARE ? BLOCK : 1], sh_wcnt[BLOCK / 64], sh_mcnt[2][BLOCK / 64];
    __shared__ double sh_fx[SHARE ? BLOCK : 1][NX];
    const int f = blockIdx.y;
    const ModelD* md = models + f;
    const FilterScal* sc = scal + f;
    const bool dbg_on = a.k == 5; (void)dbg_on;
    STEP_STAMP(0, threadIdx.x); DBG_HWID(g_step_dbg);
    // Everything the prologue reads is REQUESTED first and tested afterwards: tested one by one (stop flag, fallback flag, tables,
    // scalars) each was a memory round trip of its own at the start of every block.
    const uint32_t stop_flag = *b.bank_flag;
    const int fb_flag = sc->fallback;
    // the generator's tables in LDS (one 16-byte LDS read per lookup instead of two global loads through the GOT)
    constexpr bool LTAB = !Model::RB && MODE != MODE_WEIGHT && MODE != MODE_AUX;
    __shared__ __attribute__((aligned(16))) double sh_rng_lg[LTAB ? 2 * LLPF_RNG_LG_ENTRIES : 2], sh_rng_sc[LTAB ? 2 * LLPF_RNG_SC_ENTRIES : 2];
    double rt0 = 0.0, rt1 = 0.0;
    if (LTAB) {
        const int t = (int)threadIdx.x;
        if (t < LLPF_RNG_SC_ENTRIES) { rt0 = LLPF_SIN64[t]; rt1 = LLPF_COS64[t]; }
        else if (t < LLPF_RNG_SC_ENTRIES + LLPF_RNG_LG_ENTRIES) { rt0 = LLPF_LOG_INVC[t - LLPF_RNG_SC_ENTRIES]; rt1 = LLPF_LOG_LNC[t - LLPF_RNG_SC_ENTRIES]; }
    }
    // the first tile's marks: their address needs nothing from memory, so they travel with the scalars (zero when nothing was resampled)
    int2 mA, mB; mA.x = 0; mA.y = 0; mB = mA;      // marks of the block's next tile and of the one after it
    if constexpr (MARKS) {
        const int32_t* mk0 = b.mark + (size_t)f * b.Ns + (int64_t)blockIdx.x * (BLOCK * PPT) + (int64_t)threadIdx.x * PPT;
        mA = *reinterpret_cast<const int2*>(mk0);
        if ((int64_t)(blockIdx.x + gridDim.x) * (BLOCK * PPT) < b.Ns) mB = *reinterpret_cast<const int2*>(mk0 + (size_t)gridDim.x * (BLOCK * PPT));
    }
    const int do_res = (MODE == MODE_AUX2) ? 1 : ((MODE != MODE_WEIGHT && MODE != MODE_AUX) ? sc->do_resample : 0);   // AUX2: always resampled (filtering.jl:206)
    const int uniform = sc->uniform, pend = sc->norm_pending;
    const double m = sc->m, l = sc->l, wconst = sc->wconst;
    const uint32_t k0 = sc->k0, k1 = sc->k1, sb = sc->step_base;
#if defined(__HIP_DEVICE_COMPILE__)
    // one batch: every scalar above is wanted HERE, so their loads are issued back to back and waited for once (left to the
    // compiler they trickle in behind the branches below, a scalar-cache round trip each)
    asm volatile("" : : "s"(stop_flag), "s"(fb_flag), "s"(do_res), "s"(uniform), "s"(pend), "s"(m), "s"(l), "s"(wconst), "s"(k0), "s"(k1), "s"(sb),
                 "s"(b.Ns), "s"(b.N), "s"(a.k), "s"(a.only_fallback), "s"(a.has_y), "s"(a.step));
#endif
    if (stop_flag != 0 && (int64_t)(stop_flag - 1) < a.k) return;          // run_is_stopped
    if (a.only_fallback ? !fb_flag : (fb_flag != 0)) return;   // redo launches take the flagged filters, all others skip them
    if (LTAB) {
        const int t = (int)threadIdx.x;
        if (t < LLPF_RNG_SC_ENTRIES) { sh_rng_sc[2 * t] = rt0; sh_rng_sc[2 * t + 1] = rt1; }
        else if (t < LLPF_RNG_SC_ENTRIES + LLPF_RNG_LG_ENTRIES) { sh_rng_lg[2 * (t - LLPF_RNG_SC_ENTRIES)] = rt0; sh_rng_lg[2 * (t - LLPF_RNG_SC_ENTRIES) + 1] = rt1; }
        __syncthreads();
    }
    const int64_t Ns = b.Ns, N = b.N;
    const double* __restrict__ xc = b.xcur + (size_t)f * NX * Ns;
    double* __restrict__ xn = b.xnext + (size_t)f * NX * Ns;
    double* w = b.w + (size_t)f * Ns;
    const int32_t* __restrict__ anc = b.anc + (size_t)f * Ns;
    STEP_STAMP(1, rt0);                 // scalars back, tables in LDS
    constexpr bool WT = (MODE != MODE_AUX) && !Model::RB;     // write-through stores where they measured faster (wt_store, reduce.hpp)

    Model model;
    if constexpr (!MARKS || has_loglik<Model>::value || MODE != MODE_PROP) model.prepare(md, a.u + (size_t)f * a.u_stride, a.t_prop);
    double y[NY];
    if (MODE != MODE_PROP) {
        const double* yf = a.y + (size_t)f * a.y_stride;
#pragma unroll
        for (int k = 0; k < NY; ++k) y[k] = a.has_y ? yf[k] : 0.0;
    }

    double bmax = -LLPF_INF;
    bool bad = false;
    // bound of the weights this kernel produces: max of the previous (normalised) weights + the density's peak
    double off = 0.0;
    WeightAcc wacc;
    double xm[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
    if (MODE != MODE_PROP) {
        const double wmx = do_res ? b.log1N : (uniform ? wconst : sc->wmax);
        if constexpr (Model::RB) {     // peak of N(0, S) of this correct! (or of R2 when C == 0)
            const double c0w = md->rb_zeroC ? md->dg.c0 : (a.rb_corr + f)->dS.c0;
            off = a.has_y ? wmx + c0w : wmx;
        } else {
            off = a.has_y ? wmx + md->dg.c0 : wmx;
        }
        if (MODE == MODE_AUX2) off = ((a.aux == 2) ? md->dg.c0 : 0.0) - (-b.mlogN);    // w = lambda - log N <= c0 - log N (lambda = 0 when y1 is missing)
        wacc.init();
    }
    // ---- MARKS: f(x[ancestor]) of one tile's particles — marks -> ancestors -> gather — requested one tile ahead of the arithmetic ----
    auto fetch = [&](const int tb, const int itn, double (&fo)[PPT][NX]) {
        if constexpr (MARKS) {
            int64_t i0 = (int64_t)tb * (BLOCK * PPT) + (int64_t)threadIdx.x * PPT;
#if defined(__HIP_DEVICE_COMPILE__)
            { uint32_t il = (uint32_t)i0; asm volatile("" : "+v"(il)); i0 = (int64_t)il; }      // opaque per iteration (see tile())
#endif
            const double* __restrict__ fxp = b.fxs + (size_t)f * NX * Ns;
            if (do_res) {
                // run-start marks -> ancestors: 1 + j at the first output of every surviving source j and at every tile boundary inside
                // its range, so an inclusive max-scan over the tile gives every output its ancestor (they are non-decreasing); an
                // output without an owner carries its own flagged mark (MARK_OWN | 1 + previous j: resample.jl:27-35 writes nothing)
                const int t = (int)threadIdx.x;
                int32_t* mkp = b.mark + (size_t)f * Ns + i0;
                const int2 m2 = mA;
                mA = mB;
                mB.x = 0; mB.y = 0;
                if ((int64_t)(tb + 2 * (int)gridDim.x) * (BLOCK * PPT) < Ns) mB = *reinterpret_cast<const int2*>(mkp + (size_t)gridDim.x * (2 * BLOCK * PPT));
                if (m2.x | m2.y) { int2 z; z.x = 0; z.y = 0; *reinterpret_cast<int2*>(mkp) = z; }
                const uint32_t m0 = (uint32_t)m2.x, m1 = (uint32_t)m2.y;
                const uint32_t incl = wave_scan_max_u32(m1 > m0 ? m1 : m0);
                uint32_t base = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
                int32_t* wc = sh_mcnt[itn & 1];            // per tile parity: a wave may be one barrier ahead of the slowest reader
                if ((t & 63) == 63) wc[t >> 6] = (int32_t)incl;
                __syncthreads();
#pragma unroll
                for (int k = 0; k < BLOCK / 64 - 1; ++k) { const uint32_t c = (uint32_t)wc[k]; if (k < (t >> 6)) base = c > base ? c : base; }
                const uint32_t s0 = m0 > base ? m0 : base, s1 = m1 > s0 ? m1 : s0;
                const uint32_t am[2] = {(m0 & (uint32_t)MARK_OWN) ? m0 : s0, (m1 & (uint32_t)MARK_OWN) ? m1 : s1};
                STEP_STAMP(2, s1);           // marks back and scanned
                int32_t av[PPT];
#pragma unroll
                for (int p = 0; p < PPT; ++p) {
                    const int32_t v = (int32_t)(am[p] & ~(uint32_t)MARK_OWN) - 1;
                    av[p] = v < 0 ? 0 : v;                  // no mark at all: a launch that resampled nothing (degenerate weights), or padding
                }
                { int2 ao; ao.x = (i0 < N) ? av[0] : (int32_t)i0; ao.y = (i0 + 1 < N) ? av[1] : (int32_t)(i0 + 1);      // padding lanes keep the identity
                  uint64_t a2; __builtin_memcpy(&a2, &ao, 8);
                  wt_store(reinterpret_cast<uint64_t*>(b.anc + (size_t)f * Ns + i0), a2); }
#pragma unroll
                for (int d = 0; d < NX; ++d) {
#pragma unroll
                    for (int p = 0; p < PPT; ++p) fo[p][d] = fxp[(size_t)d * Ns + av[p]];
                }
            } else {                                        // nothing resampled: j = 1:N, f(x_i) of every particle is in the plane
#pragma unroll
                for (int d = 0; d < NX; ++d) { const double2 v = *reinterpret_cast<const double2*>(fxp + (size_t)d * Ns + i0); fo[0][d] = v.x; fo[1][d] = v.y; }
            }
        }
    };
    // ---- one tile of BLOCK * PPT particles ----
    auto tile = [&](const int tb, const int itn, const double (*fpre)[NX]) {
        int64_t i0 = (int64_t)tb * (BLOCK * PPT) + (int64_t)threadIdx.x * PPT;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MARKS) {
            // opaque per iteration: left visible, the index becomes a 64-bit induction pointer per plane, all live across the loop
            uint32_t il = (uint32_t)i0;
            asm volatile("" : "+v"(il));
            i0 = (int64_t)il;
        }
#endif
        uint64_t qsum = 0;
        double xs[PPT][NX];
        if (MODE != MODE_WEIGHT) {
            double xp[PPT][NX];
            double fsh[PPT][NX];          // f(x[ancestor]): from the scratch plane (MARKS) or from the block's table when the dynamics were shared
            bool shared = false;          // block-uniform
            if constexpr (MARKS) {
#pragma unroll
                for (int p = 0; p < PPT; ++p) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) fsh[p][d] = fpre[p][d];
                }
                shared = true;
                STEP_STAMP(3, fsh[0][0]);     // f(x[ancestor]) gathered
            } else if (do_res) {
                int32_t av[PPT];
                if constexpr (PPT == 2) { const int2 a2 = *reinterpret_cast<const int2*>(anc + i0); av[0] = a2.x; av[1] = a2.y; }
                else av[0] = anc[i0];
                if constexpr (SHARE) {
                    // x' = f(x[j]) + noise: outputs with the same ancestor share f(x[j]).  The ancestors of systematic / stratified /
                    // (the copies of) residual resampling are sorted, so the block's distinct ancestors are the starts of its runs:
                    // they are listed, the first D threads evaluate the dynamics once each, and every output reads its run's value
                    // back — the same bits, D evaluations instead of BLOCK * PPT.  (The run loop of such models takes the MARKS form;
                    // this one serves the single-step API, the auxiliary filter and residual resampling.)
                    const int t = (int)threadIdx.x;
                    sh_prev[t] = av[PPT - 1];
                    __syncthreads();
                    const int32_t prev = t ? sh_prev[t - 1] : -1;
                    int nw[PPT], c = 0;
                    nw[0] = av[0] != prev ? 1 : 0;
#pragma unroll
                    for (int p = 1; p < PPT; ++p) nw[p] = av[p] != av[p - 1] ? 1 : 0;
#pragma unroll
                    for (int p = 0; p < PPT; ++p) c += nw[p];
                    const int incl = (int)wave_scan_u64((uint64_t)c);
                    if ((t & 63) == 63) sh_wcnt[t >> 6] = incl;
                    __syncthreads();
                    int base = 0, D = 0;
#pragma unroll
                    for (int k = 0; k < BLOCK / 64; ++k) { const int v = sh_wcnt[k]; base += (k < (t >> 6)) ? v : 0; D += v; }
                    if (D <= BLOCK) {
                        int rr = base + incl - c, run[PPT];
#pragma unroll
                        for (int p = 0; p < PPT; ++p) { rr += nw[p]; run[p] = rr - 1; if (nw[p]) sh_run[rr - 1] = av[p]; }
                        __syncthreads();
                        const int kq = t;       // (rotating the evaluating wave with the block index measured no different)
                        if (kq < D) {
                            const int32_t aj = sh_run[kq];
                            double xq[NX], fq[NX];
#pragma unroll
                            for (int d = 0; d < NX; ++d) xq[d] = xc[(size_t)d * Ns + aj];
                            model.dynamics(xq, fq);
#pragma unroll
                            for (int d = 0; d < NX; ++d) sh_fx[kq][d] = fq[d];
                        }
                        __syncthreads();
#pragma unroll
                        for (int p = 0; p < PPT; ++p) {
#pragma unroll
                            for (int d = 0; d < NX; ++d) fsh[p][d] = sh_fx[run[p]][d];
                        }
                        shared = true;
                    }
                }
                if (!shared) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) {
#pragma unroll
                        for (int p = 0; p < PPT; ++p) xp[p][d] = xc[(size_t)d * Ns + av[p]];
                    }
                }
            } else {
#pragma unroll
                for (int d = 0; d < NX; ++d) {
                    if constexpr (PPT == 2) { const double2 v = *reinterpret_cast<const double2*>(xc + (size_t)d * Ns + i0); xp[0][d] = v.x; xp[1][d] = v.y; }
                    else xp[0][d] = *(xc + (size_t)d * Ns + i0);
                }
            }
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                if constexpr (Model::RB) {
                    model.rb_propagate(xp[p], (uint32_t)(i0 + p), sb + a.step, k0, k1, a.rb_pred + f, xs[p]);
                    continue;
                }
                double fx[NX], xi[NX], nz[NX];
                if constexpr (MARKS) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) fx[d] = fsh[p][d];
                } else {
                    if (shared) {
#pragma unroll
                        for (int d = 0; d < NX; ++d) fx[d] = fsh[p][d];
                    } else {
                        model.dynamics(xp[p], fx);
                    }
                }
                if (MODE == MODE_AUX) {            // propagate_particles!(pf, u, p, t, nothing): no noise
#pragma unroll
                    for (int d = 0; d < NX; ++d) xs[p][d] = fx[d];
                } else if constexpr (has_user_noise<Model>::value && !MARKS) {
                    // the model adds its own noise (AdvancedParticleFilter: dynamics(x, u, p, t, noise = true), PFtypes.jl:254; a
                    // ParticleFilter with any dynamics_density, :135): nx normals and nx uniforms of the particle's own streams
                    double uu[NX];
                    if constexpr (LTAB) llpf_normals_tab((uint32_t)(i0 + p), sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi, sh_rng_lg, sh_rng_sc);
                    else llpf_normals((uint32_t)(i0 + p), sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
                    llpf_uniforms((uint32_t)(i0 + p), sb + a.step, LLPF_STREAM_USER, k0, k1, NX, uu);
                    model.noise(xp[p], fx, xi, uu, xs[p]);
                } else {
                    if constexpr (LTAB) llpf_normals_tab((uint32_t)(i0 + p), sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi, sh_rng_lg, sh_rng_sc);
                    else llpf_normals((uint32_t)(i0 + p), sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
                    if constexpr (MARKS) gauss_sample_c<NX>((gauss_cptr)&md->df, xi, nz);     // the covariance kind tested once, operands through scalar loads
                    else gauss_sample<NX>(md->df, xi, nz);
#pragma unroll
                    for (int d = 0; d < NX; ++d) xs[p][d] = fx[d] + nz[d];
                }
            }
            STEP_STAMP(4, xs[0][0]);               // noise drawn, particles formed
            if (!(Model::RB && MODE == MODE_PROP_WEIGHT && a.has_y)) {
#pragma unroll
                for (int d = 0; d < NX; ++d) {
                    if constexpr (PPT == 2) wt_store2<WT>(xn + (size_t)d * Ns, i0, xs[0][d], xs[1][d]);
                    else wt_store<WT>(xn + (size_t)d * Ns + i0, xs[0][d]);
                }
            }
        } else {
#pragma unroll
            for (int d = 0; d < NX; ++d) {
                if constexpr (PPT == 2) { const double2 v = *reinterpret_cast<const double2*>(xc + (size_t)d * Ns + i0); xs[0][d] = v.x; xs[1][d] = v.y; }
                else xs[0][d] = *(xc + (size_t)d * Ns + i0);
            }
        }
        if (MODE != MODE_PROP) {
            double wp[PPT];
            if (MODE == MODE_AUX2) {               // w[i] = lambda[i] - log N, the unresampled lambda (filtering.jl:211-214)
                const double lN = -b.mlogN;
                const double* lamp = b.lam + (size_t)f * Ns;
                if constexpr (PPT == 2) { const double2 lv = *reinterpret_cast<const double2*>(lamp + i0); wp[0] = lv.x - lN; wp[1] = lv.y - lN; }
                else wp[0] = lamp[i0] - lN;
            } else if (do_res) {                   // reset_weights!: w = log(1/N)
#pragma unroll
                for (int p = 0; p < PPT; ++p) wp[p] = b.log1N;
            } else if (uniform) {
#pragma unroll
                for (int p = 0; p < PPT; ++p) wp[p] = wconst;
            } else {
                double wr[PPT];
                if constexpr (PPT == 2) { const double2 wv = *reinterpret_cast<const double2*>(w + i0); wr[0] = wv.x; wr[1] = wv.y; }
                else wr[0] = w[i0];
#pragma unroll
                for (int p = 0; p < PPT; ++p) wp[p] = pend ? (wr[p] - m) - l : wr[p];  // lazy w .-= offset ; w .-= log1p(s)
            }
            double wn[PPT];
            double lamv[PPT];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                double wv = wp[p];
                if (MODE == MODE_AUX) {            // lambda .= 0; lambda += logpdf; w .+= lambda  (filtering.jl:201-204)
                    double lam = 0.0;
                    if (a.has_y) {
                        if constexpr (has_loglik<Model>::value) {
                            lam = lam + model.loglik(xs[p], y, a.t_meas);
                        } else {
                            double g[NY], v[NY];
                            model.measurement(xs[p], g);
#pragma unroll
                            for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                            lam = lam + gauss_logpdf<NY>(md->dg, v);
                        }
                    }
                    lamv[p] = lam;
                    wv = wv + lam;
                } else if (MODE != MODE_AUX2 && a.has_y) {
                    if constexpr (Model::RB) {
                        wv = wv + model.rb_weight(xs[p], y, a.rb_corr + f, i0 + p == 0);
                    } else if constexpr (has_loglik<Model>::value) {
                        wv = wv + model.loglik(xs[p], y, a.t_meas);         // user measurement_likelihood; its bound sits in md->dg.c0 (k_user_bound)
                    } else {
                        double g[NY], v[NY];
                        model.measurement(xs[p], g);
#pragma unroll
                        for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                        if constexpr (MARKS) wv = wv + gauss_logpdf_c<NY>((gauss_cptr)&md->dg, v);
                        else wv = wv + gauss_logpdf<NY>(md->dg, v);
                    }
                }
                if (i0 + p >= N) wv = -LLPF_INF;   // padding lanes carry zero weight
                wn[p] = wv;
                bad = bad || (wv != wv);
                if constexpr (has_loglik<Model>::value) bad = bad || (a.has_y && wv > off);   // the user's declared bound does not hold: reported like NaN weights
                bmax = llpf_fmax(bmax, wv);
            }
            if constexpr (PPT == 2) wt_store2<WT>(w, i0, wn[0], wn[1]);
            else wt_store<WT>(w + i0, wn[0]);
            if constexpr (Model::RB) {             // correct! has updated xl (Kalman measurement update)
                if (a.has_y) {
                    double* xdst = (MODE == MODE_WEIGHT) ? const_cast<double*>(xc) : xn;
#pragma unroll
                    for (int d = 0; d < NX; ++d) {
                        if constexpr (PPT == 2) wt_store2<WT>(xdst + (size_t)d * Ns, i0, xs[0][d], xs[1][d]);
                        else wt_store<WT>(xdst + (size_t)d * Ns + i0, xs[0][d]);
                    }
                }
            }
            if (MODE == MODE_AUX) {
                if constexpr (PPT == 2) { double2 lo; lo.x = lamv[0]; lo.y = lamv[1]; *reinterpret_cast<double2*>(b.lam + (size_t)f * Ns + i0) = lo; }
                else b.lam[(size_t)f * Ns + i0] = lamv[0];
            }
            if (a.accumulate) {   // merged schedule: exp-sums, quanta and tile sums of the new weights formed here
                uint64_t qv[PPT];
                double ev[PPT];
#pragma unroll
                for (int p = 0; p < PPT; ++p) { qv[p] = wacc.add(wn[p], off, a.K, a.need_e2 != 0, &ev[p]); qsum += qv[p]; }
                if constexpr (PPT == 2) wt_store2<WT>(b.quanta_next + (size_t)f * Ns, i0, qv[0], qv[1]);
                else wt_store<WT>(b.quanta_next + (size_t)f * Ns + i0, qv[0]);
                if (a.want_xmean) {
#pragma unroll
                    for (int d = 0; d < NX; ++d) {
#pragma unroll
                        for (int p = 0; p < PPT; ++p) xm[d] = xm[d] + xs[p][d] * ev[p];
                    }
                }
                // the quanta of this wave's particles into their tile's sum (a wave never straddles a 1024-particle tile)
                qsum = wave_sum_u64(qsum);
                if ((threadIdx.x & 63) == 0 && qsum)
                    atomicAdd(reinterpret_cast<unsigned long long*>(tileq_slot(b, a.parity, f) + (i0 / TILE)), (unsigned long long)qsum);
            }
        }
    };
    if constexpr (MARKS) {
        // two buffers, the loop written out twice: a tile's gather travels under the arithmetic of the tile before it, and the FIRST tile's
        // under its own noise (copied from a "next" buffer at the top of every pass, the values were waited for before anything was drawn)
        const int ntb = (int)(Ns / (BLOCK * PPT)), G = (int)gridDim.x;
        double fA[PPT][NX], fB[PPT][NX];
        int tb = (int)blockIdx.x;
        fetch(tb, 0, fA);
#pragma unroll 1
        for (;;) {
            const int tb1 = tb + G;
            if (tb1 < ntb) fetch(tb1, 1, fB);
            tile(tb, 0, fA);
            if (tb1 >= ntb) break;
            tb = tb1 + G;
            if (tb < ntb) fetch(tb, 0, fA);
            tile(tb1, 1, fB);
            if (tb >= ntb) break;
        }
    } else {
        tile((int)blockIdx.x, 0, nullptr);
    }
    STEP_STAMP(5, bmax);                // weights, exp-sums, stores issued
    if (MODE != MODE_PROP) {
        block_flush(b.acc + (size_t)f * ACC_WORDS, a.parity, bmax, bad, wacc, a.accumulate != 0, a.need_e2 != 0, sm_fl);
        if (a.accumulate && a.want_xmean) block_store_xm<NX>(xm, xmpart_slot(b, a.parity, f) + (size_t)blockIdx.x * MAXD, sm_x);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            FilterScal* scw = b.scal + f;
            if (a.accumulate) scw->xm_parts = (int32_t)gridDim.x;
            if (MODE == MODE_AUX2) { scw->norm_pending = 0; scw->uniform = 0; scw->wmax = off; }   // final values, bounded by off (as k_resprop<AUX>)
            scw->off_slot[a.parity] = off;
            scw->exact_slot[a.parity] = 0;
            scw->e2v_slot[a.parity] = a.need_e2;
            scw->u_slot[a.parity] = llpf_uniform_step(sb + a.next_step, LLPF_STREAM_RESAMPLE, k0, k1);
        }
    }
    STEP_STAMP(6, threadIdx.x);         // block reductions and atomics done
    if (MODE != MODE_WEIGHT && MODE != MODE_AUX && blockIdx.x == 0 && threadIdx.x == 0) {
        // bookkeeping of this predict! (fields no block of this kernel reads): state.j == 1:N unless resampled
        FilterScal* scw = b.scal + f;
        scw->anc_ident_s[b.anc_slot ^ 1] = do_res ? 0 : 1;
        scw->last_resampled = do_res;
        scw->resample_count += do_res;
    }
}

// ------------------------------------------------------------------------------------------------
// k_user_bound — a model with its own likelihood: the upper bound the normalisation works against replaces the peak of the
// (unused) Gaussian descriptor, per filter, with the filter's own parameters (launched once when the bank is built)
// ------------------------------------------------------------------------------------------------
template <class Model>
__global__ void k_user_bound(ModelD* models, const double* zero_u) {
    if constexpr (has_loglik<Model>::value) {
        ModelD* md = models + blockIdx.x;
        if (threadIdx.x == 0) {
            if constexpr (has_loglik_bound<Model>::value) {
                Model model;
                model.prepare(md, zero_u, 0.0);
                md->dg.c0 = model.loglik_bound();
            } else {
                md->dg.c0 = LLPF_NO_BOUND;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_init_user — reset! / the constructor's draw for a model with an initial density of its own (UserModel::initial):
// x_i = rand(rng, initial_density), reference src/filtering.jl:4-14, src/PFtypes.jl:66.  Everything else as k_init (kernels/init.hpp).
// ------------------------------------------------------------------------------------------------
template <class Model, int NX>
__global__ __launch_bounds__(BLOCK) void k_init_user(BankDev b, const ModelD* __restrict__ models, const FilterScal* __restrict__ scal,
                                                      const double* zero_u, uint32_t step, int init_anc) {
    if constexpr (has_user_initial<Model>::value) {
        const int f = blockIdx.y;
        const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
        if (i >= b.Ns) return;
        Model model;
        model.prepare(models + f, zero_u, 0.0);
        double xi[NX], uu[NX], x0[NX];
        llpf_normals((uint32_t)i, step, LLPF_STREAM_INIT, scal[f].k0, scal[f].k1, NX, xi);
        llpf_uniforms((uint32_t)i, step, LLPF_STREAM_USER_INIT, scal[f].k0, scal[f].k1, NX, uu);
        model.initial(xi, uu, x0);
        double* xc = b.xcur + (size_t)f * b.xrows * b.Ns;
#pragma unroll
        for (int d = 0; d < NX; ++d) xc[(size_t)d * b.Ns + i] = x0[d];
        b.w[(size_t)f * b.Ns + i] = -LLPF_INF;
        if (init_anc) b.anc[(size_t)f * b.Ns + i] = (i < b.N) ? (int32_t)i : 0;
    }
}

// what a run-time compiled model provides: the instantiation's template argument is read back from its lowered name (kernels/jit.hpp)
template <int TRAITS> __global__ void k_traits_tag() {}
template <class Model> struct model_traits {
    static constexpr int value = (has_loglik<Model>::value ? LLPF_TRAIT_LOGLIK : 0) | (has_loglik_bound<Model>::value ? LLPF_TRAIT_LOGLIK_BOUND : 0) |
                                 (has_user_noise<Model>::value ? LLPF_TRAIT_NOISE : 0) | (has_user_initial<Model>::value ? LLPF_TRAIT_INITIAL : 0);
};

// ------------------------------------------------------------------------------------------------
// k_max — maxima of the raw log-weights (when no weighting kernel produced them: llpf_set_weights,
// llpf_logsumexp); also zeroes the sum accumulators like a weighting kernel does
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_max(BankDev b, int parity) {
    __shared__ double sm_max[BLOCK / 64];
    const int f = blockIdx.y;
    const double* w = b.w + (size_t)f * b.Ns;
    double bmax = -LLPF_INF;
    bool bad = false;
#pragma unroll
    for (int it = 0; it < STEP_ITERS; ++it) {
        const int64_t i0 = ((int64_t)blockIdx.x * STEP_ITERS + it) * (BLOCK * STEP_PPT) + (int64_t)threadIdx.x * STEP_PPT;
        const double2 wv = *reinterpret_cast<const double2*>(w + i0);
        if (i0 < b.N) { bmax = llpf_fmax(bmax, wv.x); bad = bad || (wv.x != wv.x); }
        if (i0 + 1 < b.N) { bmax = llpf_fmax(bmax, wv.y); bad = bad || (wv.y != wv.y); }
    }
    const double r = block_max(bmax, sm_max);
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) acc_max(b.acc + (size_t)f * ACC_WORDS, parity, r, anybad != 0);
}
