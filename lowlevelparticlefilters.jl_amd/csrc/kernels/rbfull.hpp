// kernels/rbfull.hpp — k_rbfull: propagate + weight of the Rao-Blackwellized filter with per-particle covariance
// (LLPF_MODEL_RB_BILINEAR, reference src/rbpf.jl:163-283 with singleR off).  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// Particle plane layout of this model (single filter): rows 0..NN-1 of x are xn, rows NN..NN+NL-1 the Kalman mean xl,
// the following NL(NL+1)/2 rows the packed lower triangle of the Kalman covariance R: one ancestor gather moves the
// whole RBParticle (src/rbpf.jl:1-5, :182 `xi = s.xprev[j[i]]`).  One particle per thread; all matrices in registers
// (csrc/shared/llpf_rbfull.h, shared with the oracle).  S_i = C R_i C' + R2 >= R2, so max(w_prev) + c0(R2) bounds every new
// weight: the exp-sums against that bound are formed here (merged schedule, StepArgs::accumulate) or by a k_norm launch.
// ------------------------------------------------------------------------------------------------
#define RBF_KCPTR(p) ((llpf_rbf_cptr)(p))
constexpr double RBF_BOUND_SLACK = 0x1p-20;   // keeps exp(w - bound) <= 1 when a particle's C R C' rounds to zero

// Pins every word of a prepared model object in front of `dep` (an empty asm the compiler cannot see through): whatever loads
// Model::prepare left pending are issued and waited for BEFORE the instructions that consume `dep`.
template <class M>
DEV void rbf_model_ready(const M& m, uint32_t& dep) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int W = (int)(sizeof(M) / 4);
    uint32_t w[W > 0 ? W : 1];
    __builtin_memcpy(w, &m, (size_t)W * 4);
#pragma unroll
    for (int k = 0; k < W; ++k) asm volatile("" : : "v"(w[k]));
    asm volatile("" : "+v"(dep));
#endif
}

#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
#define RBF_TSTAMP(k, dep) do { asm volatile("" : : "v"(dep)); RBF_STAMP(k); } while (0)
#else
#define RBF_TSTAMP(k, dep) ((void)0)
#endif
// element idx of an array through a 32-bit byte offset from its (uniform) base: one address register, saddr + voffset addressing
DEV double* rbf_at(double* base, uint32_t idx) { return reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (size_t)(idx * 8u)); }
DEV uint64_t* rbf_at(uint64_t* base, uint32_t idx) { return reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(base) + (size_t)(idx * 8u)); }
DEV const int32_t* rbf_at(const int32_t* base, uint32_t idx) { return reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + (size_t)(idx * 4u)); }
#define LLPF_RBF_HOT_PARAMS const FilterScal* hot_scal, const uint32_t* hot_flag, const int32_t* hot_anc, const ModelD* hot_models, const double* hot_u, \
                            int64_t hot_Ns, int32_t hot_nu, int32_t hot_ustride, const double* hot_y
#define LLPF_RBF_HOT_ARGS(b, a) (b).scal, (b).bank_flag, (b).anc, (b).models, (a).u, (b).Ns, (b).nu, (a).u_stride, (a).y
#ifndef LLPF_RBF_DMA_AUX
#define LLPF_RBF_DMA_AUX 2      /* cache policy of the global -> LDS loads: nontemporal, like the register planes (same box: +2 %) */
#endif
#ifndef LLPF_RBF_WAVES
#define LLPF_RBF_WAVES 2
#endif
// One wave per workgroup: a particle costs ~4000 vector instructions here, two waves fit a SIMD (226 VGPRs) and N = 2e5 is 3.06 batches
// of 64 particles per SIMD.  The launch is PERSISTENT for the 8x8 form: at most two waves per SIMD (launch_rbfull: gridDim.x), each
// taking batches blockIdx.x, blockIdx.x + gridDim.x, ...  While a wave runs the recursion of one batch, the covariance planes of its
// next batch travel global -> LDS without passing through registers (global_load_lds_dword: per-lane gather addresses, 35 of the 36
// planes = 17.5 KB of the 20 KB a wave may hold when eight share a CU's 160 KB); the remaining 13 planes (xn, xl, the last of R) are
// requested into registers after the recursion, before the batch's stores.  Launched as one wave per batch the third wave of every
// SIMD started its gather when the memory system was busy with the first round's stores, and the memory-bound and the issue-bound
// phases of the launch added up (tools/dbg/rbf_timing.py: 40k + 49k of 89k cycles).
constexpr int RBF_BLOCK = 64;
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void* rbf_lds_ptr;
typedef const __attribute__((address_space(1))) void* rbf_glb_ptr;
#endif
template <class Model, int NN, int NL, int NY, int MODE>
__global__ __launch_bounds__(RBF_BLOCK) __attribute__((amdgpu_waves_per_eu(NL >= 8 ? LLPF_RBF_WAVES : 1))) void k_rbfull(LLPF_RBF_HOT_PARAMS, BankDev b, StepArgs a) {
    // What the first loads are addressed with arrives in SGPRs with the wave (kernarg preload, -amdgpu-kernarg-preload-count=16; a struct
    // by value as the first argument switches it off): the filter's scalars, the ancestor index (requested whether or not this step
    // resamples), the generator's tables and the row of Bl u go out with the wave's first instructions, beside the scalar load of the rest
    // of the argument block instead of behind it — one memory round trip in front of the gather instead of three.
    const ModelD* __restrict__ models = hot_models;
    const FilterScal* scal = hot_scal;
    static_assert(MODE == MODE_WEIGHT || MODE == MODE_PROP || MODE == MODE_PROP_WEIGHT, "no auxiliary form");
    constexpr int NP = LLPF_RBF_NP(NL), ROWS = NN + NL + NP;
    constexpr bool DMA = NL >= 8 && MODE != MODE_WEIGHT;       // covariance planes through LDS (the persistent form)
    constexpr int NPD = DMA ? NP - 1 : 0, NDIR = NP - NPD;     // planes of R through LDS / straight into registers
    const int f = blockIdx.y;
    const ModelD* md = models + f;
    const FilterScal* sc = scal + f;
    uint32_t bb = blockIdx.x;                                   // the batch (64 particles) in hand
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x == 0) g_rbf_row = bb;
    __syncthreads();
    RBF_STAMP(0);
    { uint32_t hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
      g_rbf_dbg[(size_t)bb * 32 + 13] = hw_; g_rbf_dbg[(size_t)bb * 32 + 14] = xcc_; }
#endif
    // ---- with the wave's first instructions (everything here is addressed from preloaded SGPRs) ----
    const int t = (int)threadIdx.x;
    // 32-bit particle index and byte offsets from uniform bases: what stays in registers through the loop is one word per quantity
    uint32_t i = bb * (uint32_t)RBF_BLOCK + (uint32_t)t;
    const int32_t* ancf = hot_anc + (size_t)f * hot_Ns;
    const int32_t anc_i = *rbf_at(ancf, i);          // requested whether or not the step resamples: it does not wait for do_resample
    const double* uf = hot_u + (size_t)f * hot_ustride;       // banks on data of their own (llpf_bank_run_multi): filter f's row
    const llpf_rbf_par* par = &md->rbf;
    __shared__ __attribute__((aligned(16))) double sh_rng_lg[2 * LLPF_RNG_LG_ENTRIES], sh_rng_sc[2 * LLPF_RNG_SC_ENTRIES];
    __shared__ double sh_blu[LLPF_RBF_MAXL], sh_y[LLPF_RBF_MAXY + 1], sh_w[RBF_BLOCK];      // sh_y[NY]: c0 of the measurement density
    __shared__ __attribute__((aligned(16))) uint32_t sh_R[DMA ? NPD * 128 : 4];                // plane d: 64 low words, 64 high words
    static_assert(RBF_BLOCK >= LLPF_RNG_SC_ENTRIES && RBF_BLOCK >= LLPF_RNG_LG_ENTRIES, "one table entry per lane");
    double rt0 = 0.0, rt1 = 0.0, rt2 = 0.0, rt3 = 0.0, blv[8], uv[8], yv = 0.0, wN = 0.0;
    const int nu = hot_nu;
    if (MODE != MODE_WEIGHT) {
        rt0 = LLPF_SIN64[t]; rt1 = LLPF_COS64[t];
        if (t < LLPF_RNG_LG_ENTRIES) { rt2 = LLPF_LOG_INVC[t]; rt3 = LLPF_LOG_LNC[t]; }
        if (nu > 0 && t < NL) {      // lane r: row r of Bl (stride nu) and u; entries beyond nu are read (inside the arrays) and not used
#pragma unroll
            for (int c = 0; c < 8; ++c) { const int cc = c < nu ? c : 0; blv[c] = par->Bl[t * nu + cc]; uv[c] = uf[cc]; }
        }
    }
    // everything the prologue reads is REQUESTED first and tested afterwards (as in k_step): tested one by one, the stop flag, the
    // fallback flag and the scalars were eight scalar-cache round trips in a row in front of the gather
    const uint32_t stop_flag = *hot_flag;
    const int fb_flag = sc->fallback;
    const int do_res = (MODE != MODE_WEIGHT) ? sc->do_resample : 0;
    const int uniform = sc->uniform, pend = sc->norm_pending;
    const double m = sc->m, l = sc->l, wconst = sc->wconst, wmax_prev = sc->wmax;
    const uint32_t k0 = sc->k0, k1 = sc->k1, sb = sc->step_base;
    const int64_t Ns = hot_Ns, N = b.N;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : : "s"(stop_flag), "s"(fb_flag), "s"(do_res), "s"(uniform), "s"(pend), "s"(m), "s"(l), "s"(wconst), "s"(wmax_prev), "s"(k0), "s"(k1),
                 "s"(sb), "s"(Ns), "s"(N), "s"(a.k), "s"(a.only_fallback), "s"(a.has_y), "s"(a.step));
    asm volatile("" : : "s"(b.nu), "s"(b.log1N), "s"(b.anc_slot), "s"(b.F), "s"(a.next_step), "s"(a.parity), "s"(a.K), "s"(a.need_e2), "s"(a.accumulate),
                 "s"(a.t_prop), "s"(b.anc), "s"(b.xcur), "s"(b.xnext), "s"(b.w), "s"(b.quanta_next), "s"(b.acc), "s"(b.tileq), "s"(a.u), "s"(a.y));
#endif
    if (stop_flag != 0 && (int64_t)(stop_flag - 1) < a.k) return;          // run_is_stopped
    if (a.only_fallback ? !fb_flag : (fb_flag != 0)) return;
    const double* __restrict__ xc = b.xcur + (size_t)f * ROWS * Ns;
    double* __restrict__ xo = (MODE == MODE_WEIGHT) ? const_cast<double*>(xc) : b.xnext + (size_t)f * ROWS * Ns;
    double* w = b.w + (size_t)f * Ns;
    uint64_t* qnf = b.quanta_next + (size_t)f * Ns;
    const uint32_t nbatch = (uint32_t)(Ns / RBF_BLOCK), gstep = gridDim.x;
    const bool need_w = MODE != MODE_PROP && !do_res && !uniform;

    // Order of the requests.  Vector loads return in order, so whatever is requested AFTER a gather can only be waited for together
    // with all of it: the ancestor index goes first, then every particle-independent operand (generator tables, the row of Bl and u
    // of lane r < NL, y, the model's constants), then the planes; the Gaussian's operands come through scalar loads (a counter of
    // their own).  Behind the planes' issue: the generator (it needs nothing of them), then the dynamics (xn), then the recursion (R).
    if (MODE != MODE_PROP && a.has_y && t <= NY) yv = t < NY ? hot_y[(size_t)f * a.y_stride + t] : md->dg.c0;
    if (need_w) wN = *rbf_at(w, i);
    RBF_TSTAMP(16, t);                 // prologue scalars back, ancestor and operands requested
    // 32-bit byte offsets from ONE uniform base per buffer (48 planes: 64-bit addresses would hold 96 registers and cost two
    // instructions each); the launcher checks that a filter's planes span less than 4 GB
    const uint32_t stride = (uint32_t)Ns * 8u;
    uint32_t so = (do_res ? (uint32_t)anc_i : i) * 8u;
    RBF_TSTAMP(17, so);                // ancestor and operands back
    // nontemporal: every plane is read once per launch (same box: 50.0 -> 47.3 us; only on the steps that do not resample: no better)
    auto ld = [&](int row, uint32_t off) { return __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(xc) + (off + (uint32_t)row * stride))); };
    // the planes of one batch: xn, xl and the last NDIR planes of R into registers, the first NPD planes of R into LDS
    double xn[NN], xl[NL], RN[NDIR], fi[NN], nz[NN];
    auto request_regs = [&](uint32_t off) {
#pragma unroll
        for (int d = 0; d < NN; ++d) xn[d] = ld(d, off);
#pragma unroll
        for (int d = 0; d < NL; ++d) xl[d] = ld(NN + d, off);
#pragma unroll
        for (int d = 0; d < NDIR; ++d) RN[d] = ld(NN + NL + NPD + d, off);
    };
    auto request_lds = [&](uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (DMA) {
            const char* base = reinterpret_cast<const char*>(xc);
#pragma unroll
            for (int d = 0; d < NPD; ++d) {      // the instruction's offset moves the global AND the LDS address: the high word's LDS base is taken 4 back
                const uint32_t o = off + (uint32_t)(NN + NL + d) * stride;
                __builtin_amdgcn_global_load_lds((rbf_glb_ptr)(base + o), (rbf_lds_ptr)(sh_R + d * 128), 4, 0, LLPF_RBF_DMA_AUX);
                __builtin_amdgcn_global_load_lds((rbf_glb_ptr)(base + o), (rbf_lds_ptr)(sh_R + d * 128 + 63), 4, 4, LLPF_RBF_DMA_AUX);
            }
        }
#endif
    };
    request_regs(so);
    double Rf[NPD > 0 ? NPD : 1];      // the first batch's LDS planes come straight into registers (half the instructions of the LDS path)
#pragma unroll
    for (int d = 0; d < NPD; ++d) Rf[d] = ld(NN + NL + d, so);

    // Bl u is particle-independent and nu a run-time number: formed once per wave (lane r: row r), read back by the time update from
    // LDS (as a branch inside the unrolled body it cut the body into blocks that each kept their constants' SGPRs alive)
    if (MODE != MODE_WEIGHT) {
        sh_rng_sc[2 * t] = rt0; sh_rng_sc[2 * t + 1] = rt1;
        if (t < LLPF_RNG_LG_ENTRIES) { sh_rng_lg[2 * t] = rt2; sh_rng_lg[2 * t + 1] = rt3; }
        if (t < NL) {
            double b2 = -0.0;                                        // llpf_rbf_blu_row: x + (-0.0) == x for every x
            if (nu > 0) {
                b2 = blv[0] * uv[0];
#pragma unroll
                for (int c = 1; c < 8; ++c) { const double t2 = llpf_fma(blv[c], uv[c], b2); b2 = c < nu ? t2 : b2; }
            }
            sh_blu[t] = b2;
        }
    }
    if (MODE != MODE_PROP && a.has_y && t <= NY) sh_y[t] = yv;
    __syncthreads();
    RBF_TSTAMP(18, t);                 // planes requested, tables in LDS

    // The model's constants through scalar loads (their own counter: no place in the queue of the planes), where they are used: held
    // across the loop they would sit in registers through the recursion.  The pointers pass an empty asm tied to a value of the batch
    // in hand, so the loads can be neither hoisted out of the loop nor issued before that value exists.
    auto prepared = [&](auto dep) {
        Model mdl;
#if defined(__HIP_DEVICE_COMPILE__)
        const __attribute__((address_space(4))) ModelD* mk = (const __attribute__((address_space(4))) ModelD*)md;
        const __attribute__((address_space(4))) double* uk = (const __attribute__((address_space(4))) double*)uf;
        asm volatile("" : "+s"(mk), "+s"(uk) : "v"(dep));
        mdl.prepare((const ModelD*)mk, (const double*)uk, a.t_prop);
#else
        mdl.prepare(md, uf, a.t_prop);
#endif
        return mdl;
    };
    // what does not need the covariance: the noise of particle idx, then (xn back) the dynamics
    auto front = [&](uint32_t idx) {
        Model mdl;
        if (MODE != MODE_WEIGHT) {
            mdl = prepared(idx);       // the constants' scalar loads go out first and are back when the noise is done
            double xi[NN];
#if defined(LLPF_RBF_ABL_NOISE) && LLPF_RBF_ABL_NOISE == 1      /* timing experiments only (tools/ab/c5_front.sh): no generator ... */
#pragma unroll
            for (int d = 0; d < NN; ++d) xi[d] = 0x1.bp-31 * (double)(int32_t)((idx + (uint32_t)d * 977u + (sb + a.step) * 7919u) * 2654435761u);
#elif defined(LLPF_RBF_ABL_NOISE) && LLPF_RBF_ABL_NOISE == 2    /* ... or its output read from memory as if a launch before this one had written it */
#pragma unroll
            for (int d = 0; d < NN; ++d)
                xi[d] = 0x1p-40 * *rbf_at(const_cast<double*>(xc) + (size_t)(NN + d) * Ns, idx) + 0x1.bp-31 * (double)(int32_t)((idx + (uint32_t)d * 977u + (sb + a.step) * 7919u) * 2654435761u);
#else
            llpf_normals_tab(idx, sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NN, xi, sh_rng_lg, sh_rng_sc);
#endif
            gauss_sample_c<NN>((gauss_cptr)&md->df, xi, nz);
            RBF_TSTAMP(19, nz[NN - 1]);        // generator done
        }
        RBF_TSTAMP(20, xn[NN - 1]);            // xn back
        if (MODE != MODE_WEIGHT) {
#if defined(LLPF_RBF_ABL_DYN)                                   /* timing experiments only: no RK4 */
#pragma unroll
            for (int d = 0; d < NN; ++d) fi[d] = xn[d];
#else
            mdl.dynamics(xn, fi);
#endif
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : : "v"(fi[NN - 1]), "v"(nz[NN - 1]));      // RK4 and the generator are done before stamp 1
#endif
        }
    };
    front(i);
    if (DMA) {
#pragma unroll
        for (int d = 0; d < NPD; ++d) { sh_R[d * 128 + t] = (uint32_t)__double2loint(Rf[d]); sh_R[d * 128 + 64 + t] = (uint32_t)__double2hiint(Rf[d]); }
    }

    for (;;) {
        const bool has_next = DMA && bb + gstep < nbatch;          // uniform
        const uint32_t i_n = i + gstep * (uint32_t)RBF_BLOCK;
        const uint32_t io = i * 8u;
                // written through (global_store ... sc1, kernels/reduce.hpp): the planes are the next launch's input; left dirty in the L2s they are
        // written back at the kernel boundary, and the persistent wave's next batch queues behind them (same box: 51.4 -> 50.0 us)
        auto st = [&](int row, double v) { wt_store(reinterpret_cast<double*>(reinterpret_cast<char*>(xo) + (io + (uint32_t)row * stride)), v); };
        // the next batch's ancestor: requested here, wanted after the time update
        int32_t anc_n = 0;
        if (has_next && do_res) anc_n = *rbf_at(ancf, i_n);
        if (MODE != MODE_PROP) sh_w[t] = wN;
        double R[NP];
#pragma unroll
        for (int d = 0; d < NPD; ++d) R[d] = __hiloint2double((int)sh_R[d * 128 + 64 + t], (int)sh_R[d * 128 + t]);
#pragma unroll
        for (int d = 0; d < NDIR; ++d) R[NPD + d] = RN[d];
        uint32_t so_n = (do_res ? (uint32_t)anc_n : i_n) * 8u;
#if defined(__HIP_DEVICE_COMPILE__)
        if (DMA && has_next) {
            // The next batch's covariance planes start travelling while this batch's recursion runs.  The LDS planes they overwrite have
            // just been read: the reads are pinned in front of the wait below, which holds until their data is back.
#pragma unroll
            for (int d = 0; d < NPD; ++d) asm volatile("" : : "v"(R[d]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(so_n));
            request_lds(so_n);
        }
#endif
        if (MODE != MODE_WEIGHT) {
            double xn1[NN], xl1[NL], R1[NP];
            llpf_rbf_predict(par, NN, NL, nu, xn, xl, R, uf, sh_blu, fi, nz, xn1, xl1, R1);
#pragma unroll
            for (int d = 0; d < NN; ++d) xn[d] = xn1[d];
#pragma unroll
            for (int d = 0; d < NL; ++d) xl[d] = xl1[d];
#pragma unroll
            for (int d = 0; d < NP; ++d) R[d] = R1[d];
        }

        double bmax = -LLPF_INF;
        bool bad = false;
        double off = 0.0;
        WeightAcc wacc;
        uint64_t qsum = 0;
        double wv = 0.0;
        if (MODE != MODE_PROP) {
            wacc.init();
            const double wmx = do_res ? b.log1N : (uniform ? wconst : wmax_prev);
            off = a.has_y ? (wmx + sh_y[NY]) + RBF_BOUND_SLACK : wmx;
            if (do_res) wv = b.log1N;                                 // reset_weights!
            else if (uniform) wv = wconst;
            else { const double wr = sh_w[t]; wv = pend ? (wr - m) - l : wr; }
            if (a.has_y) {
                double y[NY], yn[NY];
#pragma unroll
                for (int k = 0; k < NY; ++k) y[k] = sh_y[k];
                prepared(xn[0]).measurement(xn, yn);
#if defined(LLPF_RBF_ABL_CORR)                                  /* timing experiments only: no measurement update */
                wv = wv + 0x1p-40 * (xl[0] + yn[0] + y[0]);
#else
                wv = wv + llpf_rbf_correct(par, NL, NY, y, yn, xl, R);   // w[i] += ll, src/rbpf.jl:272
#endif
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : : "v"(wv));
                RBF_STAMP(11);
#endif
            }
            if (i >= (uint32_t)N) wv = -LLPF_INF;                                // padding lanes carry zero weight
        }
        if (MODE != MODE_PROP) {
            bad = wv != wv;
            bmax = wv;
            // merged schedule: exp-sums, quantum and tile sum of the new weight formed here (otherwise by k_norm)
            if (a.accumulate) qsum = wacc.add(wv, off, a.K, a.need_e2 != 0);
        }
        if (MODE != MODE_PROP) {
            *rbf_at(w, i) = wv;
            if (a.accumulate) *rbf_at(qnf, i) = qsum;
        }
        if (MODE != MODE_WEIGHT || a.has_y) {
#pragma unroll
            for (int d = 0; d < NN; ++d) st(d, xn[d]);
#pragma unroll
            for (int d = 0; d < NL; ++d) st(NN + d, xl[d]);
#pragma unroll
            for (int d = 0; d < NP; ++d) st(NN + NL + d, R[d]);
        }
        // the next batch's register planes: behind this batch's stores in the queue (requested before them, their 26 registers do not
        // fit beside the 96 being stored and the compiler spills them with a wait); its noise is generated while they travel
        if (has_next) {
            request_regs(so_n);
            if (need_w) wN = *rbf_at(w, i_n);
        }
        if (MODE != MODE_PROP) {
            const double r = wave_max(bmax);                        // the workgroup is one wave
            const int anybad = __ballot(bad) != 0 ? 1 : 0;
            if (a.accumulate) {
                wacc.flush_wave(b.acc + (size_t)f * ACC_WORDS, a.parity, a.need_e2 != 0);
                qsum = wave_sum_u64(qsum);     // the wave's 64 particles lie in one 1024-particle tile
            }
            if (threadIdx.x == 0) {
                if (a.accumulate && qsum) atomicAdd(reinterpret_cast<unsigned long long*>(tileq_slot(b, a.parity, f) + (i / TILE)), (unsigned long long)qsum);
                acc_max(b.acc + (size_t)f * ACC_WORDS, a.parity, r, anybad != 0);
                if (bb == 0) {
                    FilterScal* scw = b.scal + f;
                    scw->off_slot[a.parity] = off;
                    scw->exact_slot[a.parity] = 0;
                    scw->e2v_slot[a.parity] = a.need_e2;
                    scw->u_slot[a.parity] = llpf_uniform_step(sb + a.next_step, LLPF_STREAM_RESAMPLE, k0, k1);
                }
            }
        }
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
        RBF_STAMP(12);
#endif
        if (MODE != MODE_WEIGHT && bb == 0 && threadIdx.x == 0) {
            FilterScal* scw = b.scal + f;
            scw->anc_ident_s[b.anc_slot ^ 1] = do_res ? 0 : 1;
            scw->last_resampled = do_res;
            scw->resample_count += do_res;
        }
        if (!DMA || !has_next) break;      // only the form with its covariance planes through LDS is launched with fewer waves than batches
        bb += gstep;
        i = i_n;
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
        if (threadIdx.x == 0) g_rbf_row = bb;
        __syncthreads();
        RBF_STAMP(0);
        { uint32_t hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
          g_rbf_dbg[(size_t)bb * 32 + 13] = hw_; g_rbf_dbg[(size_t)bb * 32 + 14] = xcc_; g_rbf_dbg[(size_t)bb * 32 + 15] = 1; }
#endif
        front(i);                          // the next batch's noise and dynamics while this batch's stores drain
    }
}

// reset!(pf::RBPF), src/rbpf.jl:146-160: xl = copy(kf.d0.mu), R = copy(kf.d0.Sigma) for every particle (xn ~ d0n by k_init); any shape
__global__ __launch_bounds__(BLOCK) void k_rbfull_init(BankDev b, const ModelD* __restrict__ models) {
    const int nn = b.nx, nl = b.pad0 & 0xff, np = LLPF_RBF_NP(nl);
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    const llpf_rbf_par* par = &models[f].rbf;
    double* xc = b.xcur + (size_t)f * b.xrows * b.Ns;
    for (int d = 0; d < nl; ++d) xc[(size_t)(nn + d) * b.Ns + i] = par->xl0[d];
    for (int d = 0; d < np; ++d) xc[(size_t)(nn + nl + d) * b.Ns + i] = par->R0[d];
}
