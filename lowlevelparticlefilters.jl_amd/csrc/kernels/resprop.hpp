// kernels/resprop.hpp — k_resprop, the fused predict!.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// k_resprop — the fused predict!: finalize + shouldresample + resample + propagate [+ weight of the next
// correct!] in ONE launch.  A block owns a tile of 1024 SOURCE particles; it derives which outputs its sources
// produce ([c_start, c_end), from the ancestor counts) and propagates exactly those outputs, reading its
// sources' states (an 8 KB window per dimension: L1/L2 hits) and writing x, w (and j, kept for the accessor and
// for the reference's "stale j" corner) coalesced.  No ancestor array round trip, no separate propagate launch.
// Load balance: a tile produces ~1024 outputs +- a few % for i.i.d.-like weights; a tile holding very heavy
// particles loops over more 256-output chunks (worst case ESS -> 1: one block does everything; still far faster
// than the serial reference).
// The per-output arithmetic is the same sequence as k_step's, so fused and unfused paths are bit-identical.
// ------------------------------------------------------------------------------------------------
template <class T> DEV T ld_off(const T* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T> DEV void st_off(T* base, uint32_t byte_off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// stores of the output loop are write-through (wt_store in reduce.hpp says why)
#ifndef LLPF_RBX_TAB
#define LLPF_RBX_TAB 1      /* RBLin with a compile-time split: generator tables in LDS + owner table, like the plain model's kernel */
#endif
#ifndef LLPF_RBX_W4
#define LLPF_RBX_W4 1       /* ... and four waves per SIMD */
#endif
#ifndef LLPF_RBX_WT
#define LLPF_RBX_WT 1       /* ... and write-through stores (same box, us per timestep at N = 1e6: run-time split 30.0; compile-time split 24.6; + tables 23.1; + four waves 23.3; + these 22.7 — profiles/r06_rbpf_lean_ab.txt) */
#endif
#ifndef LLPF_RESPROP_PF
#define LLPF_RESPROP_PF 2      /* rounds of the split-schedule output loop whose sources are requested together (1.6e7 particles, plain stores: 1: 240.5, 2: 236, 4: 250 us) */
#endif
#define LLPF_RB_PLAIN_ST (Model::RB && !(model_lean<Model>::value && LLPF_RBX_WT))
// store policy of the output loop (Mem<>::st): 1 = write-through (sc1), 0 = plain, 2 = nontemporal
#ifndef LLPF_RESPROP_ST
#define LLPF_RESPROP_ST 1
#endif
// ... and of the steps that do not resample (every output reads its own index: nothing a store leaves in the L2 is read again by this launch)
#ifndef LLPF_RESPROP_ST_ID
#define LLPF_RESPROP_ST_ID LLPF_RESPROP_ST
#endif
#ifndef LLPF_RESPROP_ID2
#define LLPF_RESPROP_ID2 1     /* steps that do not resample, split-schedule form: two adjacent outputs per thread, 16-byte accesses (LLPF_OUTPUT_LOOP_ID2) */
#endif
#ifndef LLPF_RESPROP_LD_ID
#define LLPF_RESPROP_LD_ID 0      /* 1: the sources of such a step are read nontemporal as well */
#endif
#define LLPF_STCOH ((LLPF_WT && !LLPF_RB_PLAIN_ST) ? STP : COH)
#define LLPF_STCOH0 ((LLPF_WT && !LLPF_RB_PLAIN_ST) ? LLPF_RESPROP_ST : 0)
template <class Model, int NX, int NY, bool WEIGHT, bool COH = false, bool LTAB = false>
struct PropCtx {
    const BankDev& b;
    const Model& model;
    const ModelD* md;
    const StepArgs& st;
    const double* y;
    const double* __restrict__ xc;
    double* __restrict__ xn;
    double* w;             // weights in front of this launch (read: w_prev of a step that does not resample)
    double* wn;            // weights this launch forms (BankDev::w_next)
    uint32_t k0, k1;
    uint32_t pstep;        // Philox step of this predict! (step_base + st.step)
    int ablate;
    double off;            // bound of the new weights (offset of their exp-sums)
    uint64_t* qnext;       // quanta of the new weights
    const double* rng_lg = nullptr;   // the block's copy of the generator's tables in LDS , or nullptr: the
    const double* rng_sc = nullptr;   //   tables in constant memory (a global load through the GOT per lookup)
    // propagate output o from source src with previous log-weight wprev; returns the new log-weight
    // Addresses are a uniform plane base (SGPRs) + a 32-bit byte offset (one VGPR): Ns * 8 < 2^32 is checked at create.
    // the source's state (nontemporal: C2 22.8 against 20.7 us — duplicated ancestors are re-read from the L2)
    template <bool NTL = false>
    DEV void fetch(uint32_t src, double* xp) const {
        const int64_t Ns = b.Ns;
        const uint32_t so = src << 3;
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            if constexpr (NTL) xp[d] = __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(xc + (size_t)d * Ns) + so));
            else xp[d] = Mem<COH>::ld_off(xc + (size_t)d * Ns, so);
        }
    }
    template <int STP = LLPF_RESPROP_ST, bool NTL = false>
    DEV double one(uint32_t src, uint32_t o, double wprev, bool& bad, double* xs) const {
        double xp[NX];
        fetch<NTL>(src, xp);
        return one_x<STP>(xp, o, wprev, bad, xs);
    }
    // ... from the state xp of its source (fetch); STP: the store policy of this call (LLPF_RESPROP_ST / LLPF_RESPROP_ST_ID), or -1: nothing is
    // stored here — the caller stores xs and the returned weight itself (LLPF_OUTPUT_LOOP_ID2: two outputs per 16-byte store)
    template <int STP = LLPF_RESPROP_ST>
    DEV double one_x(const double* xp, uint32_t o, double wprev, bool& bad, double* xs) const {
        const int64_t Ns = b.Ns;
        const uint32_t oo = o << 3;
        double fx[NX], xi[NX], nz[NX];
        if constexpr (Model::RB) {     // Rao-Blackwellized model: own noise structure, and correct! updates xl before the store
            if constexpr (LTAB) model.rb_propagate(xp, o, pstep, k0, k1, st.rb_pred + blockIdx.y, xs, rng_lg, rng_sc);
            else model.rb_propagate(xp, o, pstep, k0, k1, st.rb_pred + blockIdx.y, xs);
            double wr = wprev;
            if (WEIGHT) {
                if (st.has_y) wr = wr + model.rb_weight(xs, y, st.rb_corr + blockIdx.y, o == 0);
                if (o >= (uint32_t)b.N) wr = -LLPF_INF;
                bad = bad || (wr != wr);
                if constexpr (STP >= 0) Mem<LLPF_STCOH>::st_off(wn, oo, wr);
            }
            if constexpr (STP >= 0) {
#pragma unroll
                for (int d = 0; d < NX; ++d) Mem<LLPF_STCOH>::st_off(xn + (size_t)d * Ns, oo, xs[d]);
            }
            return wr;
        }
#ifdef LLPF_DEVTOOLS   /* ablation switches for performance experiments (results invalid); not in production builds */
        if (!(ablate & 4)) model.dynamics(xp, fx);
        else { for (int d = 0; d < NX; ++d) fx[d] = xp[d]; }
        if (!(ablate & 1)) { if constexpr (LTAB) llpf_normals_tab(o, pstep, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi, rng_lg, rng_sc); else llpf_normals(o, pstep, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi); }
        else { for (int d = 0; d < NX; ++d) xi[d] = 0.25 * (double)(o & 7); }
#else
        model.dynamics(xp, fx);
        if constexpr (LTAB) llpf_normals_tab(o, pstep, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi, rng_lg, rng_sc);
        else llpf_normals(o, pstep, LLPF_STREAM_DYNAMICS, k0, k1, NX, xi);
#endif
        gauss_sample_c<NX>((gauss_cptr)&md->df, xi, nz);
#pragma unroll
        for (int d = 0; d < NX; ++d) {
            xs[d] = fx[d] + nz[d];
            if constexpr (STP >= 0) Mem<LLPF_STCOH>::st_off(xn + (size_t)d * Ns, oo, xs[d]);
        }
        double wv = wprev;
        if (WEIGHT) {
#ifdef LLPF_DEVTOOLS
            if (st.has_y && !(ablate & 4)) {
#else
            if (st.has_y) {
#endif
                double g[NY], v[NY];
                model.measurement(xs, g);
#pragma unroll
                for (int k = 0; k < NY; ++k) v[k] = y[k] - g[k];
                wv = wv + gauss_logpdf_c<NY>((gauss_cptr)&md->dg, v);
            }
            if (o >= (uint32_t)b.N) wv = -LLPF_INF;
            bad = bad || (wv != wv);
            if constexpr (STP >= 0) Mem<LLPF_STCOH>::st_off(wn, oo, wv);
        }
        return wv;
    }
};

// per-thread running sum of quanta keyed by destination tile; flushed to LDS (first 8 tiles of the block's output
// range) or straight to the global tile sums (heavier blocks) whenever the tile changes
struct TileSum {
    int32_t tcur;
    uint64_t run;
    DEV void init() { tcur = -1; run = 0; }
    DEV void flush(uint64_t* sh_tq, uint64_t* tq_global, int32_t tbase) {
        if (run) {
            const int32_t idx = tcur - tbase;
            if (idx >= 0 && idx < 8) atomicAdd(reinterpret_cast<unsigned long long*>(sh_tq + idx), (unsigned long long)run);
            else atomicAdd(reinterpret_cast<unsigned long long*>(tq_global + tcur), (unsigned long long)run);
        }
        run = 0;
    }
    DEV void add(uint32_t o, uint64_t q, uint64_t* sh_tq, uint64_t* tq_global, int32_t tbase) {
        const int32_t t = (int32_t)(o >> 10);
        if (t != tcur) { flush(sh_tq, tq_global, tbase); tcur = t; }
        run += q;
    }
};
static_assert(TILE == 1024, "TileSum assumes 1024-particle tiles");

// leading scalar kernel arguments: preloaded into SGPRs by the command processor (-mllvm -amdgpu-kernarg-preload-count;
// a struct passed by value as the FIRST argument disables the preload, which is why these are not left inside BankDev)
#define LLPF_HOT_PARAMS uint64_t* hot_acc, uint64_t* hot_tileq, FilterScal* hot_scal, uint32_t* hot_flag, const uint64_t* hot_quanta, \
                        int hot_F, int hot_P2, int hot_parity
#define LLPF_HOT_ARGS(b, a) (b).acc, (b).tileq, (b).scal, (b).bank_flag, (b).quanta, (b).F, (b).P2, (a).parity

// ONE: the filter is a single tile (launched only when P2 == 1): a failed bound test is redone inside this kernel
template <class Model, int NX, int NY, bool WEIGHT, bool ACC, bool AUX = false, bool ONE = false>
// amdgpu_waves_per_eu(4): the ~3.8 blocks per CU of a 10^6-particle filter must be resident together (<= 128 VGPRs); the
// larger state dimensions would spill under that cap and keep the compiler's choice; the Rao-Blackwellized propagate uses
// 130-156 VGPRs and is pinned to three waves per SIMD (<= 168): twice in this round an unrelated change pushed it past 170 and
// cost it 28 %
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(((!Model::RB || (model_lean<Model>::value && LLPF_RBX_W4)) && NX <= 2 && NY <= 2 && !ONE) ? 4 : (Model::RB ? 3 : 1)))) void k_resprop(LLPF_HOT_PARAMS, BankDev b_in, const ModelD* __restrict__ models, ResArgs a_in, StepArgs st) {
    // what the head's first loads are addressed with arrives in SGPRs with the wave (kernarg preload) instead of through a
    // scalar load of the argument block: one memory round trip less at the start of every launch
    BankDev b = b_in;
    b.acc = hot_acc; b.tileq = hot_tileq; b.scal = hot_scal; b.bank_flag = hot_flag; b.quanta = const_cast<uint64_t*>(hot_quanta);
    b.F = hot_F; b.P2 = hot_P2;
    ResArgs a = a_in;
    a.parity = hot_parity;
    a.mode = RES_FINALIZE | RES_RESAMPLE;        // what launch_resprop always passes: known here, the head's first loads then wait for no scalar load of the argument block
    __shared__ ResShared sh;
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_acc[BLOCK / 64][5];
    __shared__ uint64_t sh_tq[8];
    __shared__ double sm_x[BLOCK / 64][MAXD];
    constexpr bool RBFAT = Model::RB && !(model_lean<Model>::value && LLPF_RBX_TAB);      // run-time split: no registers to spare for the tables (occupancy 3 -> 2)
    constexpr bool OWN_TABLE = !RBFAT;
    __shared__ __attribute__((aligned(16))) uint32_t sh_own[OWN_TABLE ? OWN_CAP : 4];
    __shared__ __attribute__((aligned(16))) double sh_rng_lg[2 * LLPF_RNG_LG_ENTRIES], sh_rng_sc[2 * LLPF_RNG_SC_ENTRIES];
    // Wave priority by phase: the head / counts / tail phases are short and latency-bound (loads, LDS, barriers, atomics), the
    // output loop is long and issue-bound.  The waves of a CU's four blocks are otherwise served oldest first, so the
    // youngest block's head is starved by the older blocks' loops and the SIMD ends the launch with that block's loop alone
    // (in-kernel stamps, tools/dbg/timing_report.py).  Same-box A/B: C2 27.0 -> 26.4 us per timestep.
    __builtin_amdgcn_s_setprio(3);
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const int64_t Ns = b.Ns;
    const ModelD* md = models + f;
    FilterScal* sc = b.scal + f;
    const uint32_t stop_flag = *b.bank_flag;     // tested in res_head, after all other loads are in flight
    const int fb_flag = sc->fallback;
    if (threadIdx.x < 8) sh_tq[threadIdx.x] = 0;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
    int anc_ident_v = sc->anc_ident_s[b.anc_slot];            // this launch writes the other entry
    if (OWN_TABLE) {   // owner table of the output loop: cleared here, filled after the counts (res_owner_table)
        const uint4 z = {0u, 0u, 0u, 0u};
        reinterpret_cast<uint4*>(sh_own)[2 * threadIdx.x] = z;
        reinterpret_cast<uint4*>(sh_own)[2 * threadIdx.x + 1] = z;
    }
    double rt0 = 0.0, rt1 = 0.0;
// phase stamps (tools/dbg/timing_c2.sh): compiled in with DEVTOOLS=1 only — the test of a.dbg at the kernel's first instruction made
// every block wait for a scalar load of the argument block before it could request anything else
#ifdef LLPF_DEVTOOLS
#define LLPF_STAMP(i) if (a.dbg && threadIdx.x == 0 && f == 0) a.dbg[(size_t)tile * 8 + (i)] = wall_clock64()
#else
#define LLPF_STAMP(i) ((void)0)
#endif
    LLPF_STAMP(0);
    Model model;
    double y[NY];
    uint32_t key0_v = sc->k0, key1_v = sc->k1, sb_v = sc->step_base;
    const double c0_pre = md->dg.c0;                   // fetched with the other loads: the bound below must not wait for it
    // particle-independent terms (B u, the measurement row): their loads run while the head's vector loads are in flight
    auto prepare = [&]() {
        model.prepare(md, st.u + (size_t)f * st.u_stride, st.t_prop);
        const double* yf = st.y + (size_t)f * st.y_stride;
#pragma unroll
        for (int k = 0; k < NY; ++k) y[k] = (WEIGHT && st.has_y) ? yf[k] : 0.0;
        // Issued LAST, after everything the head itself waits for (accumulator words, tile sums: a few KB): the tile's quanta
        // are 8 MB over the whole launch, all blocks start at once, and loads return in order — issued first they kept every
        // head waiting 2.5 us for a burst that is only needed by the counts.  Likewise the generator's tables (for the loop).
#pragma unroll
        for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
        if (!RBFAT) {
            const int t = (int)threadIdx.x;
            if (t < LLPF_RNG_SC_ENTRIES) { rt0 = LLPF_SIN64[t]; rt1 = LLPF_COS64[t]; }
            else if (t < LLPF_RNG_SC_ENTRIES + LLPF_RNG_LG_ENTRIES) { rt0 = LLPF_LOG_INVC[t - LLPF_RNG_SC_ENTRIES]; rt1 = LLPF_LOG_LNC[t - LLPF_RNG_SC_ENTRIES]; }
        }
    };
    const ResHead h = res_head<SRC_FILTER, false>(b, a, f, tile, sh, true, stop_flag, fb_flag, prepare);
    if (h.status) return;
    // Values of FilterScal fetched above with the head's loads but wanted only from here on.  FilterScal is written by this
    // kernel, so they are vector loads made uniform with v_readfirstlane — which the compiler otherwise places right behind the
    // loads, with a vmcnt wait BEFORE the head's own loads are issued (a second, serialized memory round trip at the start of
    // every block).  The asm pins the first use behind the head's barrier.
    asm volatile("" : "+v"(key0_v), "+v"(key1_v), "+v"(sb_v), "+v"(anc_ident_v));
    const uint32_t key0 = __builtin_amdgcn_readfirstlane(key0_v), key1 = __builtin_amdgcn_readfirstlane(key1_v);
    const uint32_t sb = __builtin_amdgcn_readfirstlane(sb_v);
    const int anc_ident_prev = __builtin_amdgcn_readfirstlane(anc_ident_v);
    if (!RBFAT) {      // the generator's tables -> LDS (one 16-byte LDS read per lookup instead of two global loads); the
        const int t = (int)threadIdx.x;     // barriers of the counts / the one below come before the loop reads them
        if (t < LLPF_RNG_SC_ENTRIES) { sh_rng_sc[2 * t] = rt0; sh_rng_sc[2 * t + 1] = rt1; }
        else if (t < LLPF_RNG_SC_ENTRIES + LLPF_RNG_LG_ENTRIES) { sh_rng_lg[2 * (t - LLPF_RNG_SC_ENTRIES)] = rt0; sh_rng_lg[2 * (t - LLPF_RNG_SC_ENTRIES) + 1] = rt1; }
    }
    LLPF_STAMP(1);
    PropCtx<Model, NX, NY, WEIGHT, false, !RBFAT> pc{b, model, md, st, y, b.xcur + (size_t)f * NX * Ns, b.xnext + (size_t)f * NX * Ns,
                                      b.w + (size_t)f * Ns, b.w_next + (size_t)f * Ns, key0, key1, sb + st.step, a.ablate, 0.0, b.quanta_next + (size_t)f * Ns,
                                      RBFAT ? nullptr : sh_rng_lg, RBFAT ? nullptr : sh_rng_sc};
    int32_t* anc = b.anc + (size_t)f * Ns;
    double bmax = -LLPF_INF;
    bool bad = false;

    // One loop over the outputs this block produces (the per-output body is instantiated once):
    //   resampling : outputs [c_start, c_end) from the ancestor counts, source = tile's owner of the output;
    //                the last tile also takes [c_end, M): thresholds >= bins[N], for which the reference leaves
    //                j[i] untouched (resample.jl:25-34) -> previous ancestor (identity if the last predict! did
    //                not resample)
    //   otherwise  : s.j .= 1:N, the tile's own particles (padding lanes included so that their weight stays -Inf)
    const bool res = (h.dr || a.force) && h.tot != 0;
    int64_t first, last;
    int32_t c_end = 0;
    double l = 0.0;
    WeightAcc wacc;
    TileSum ts;
    wacc.init();
    ts.init();
    double xm[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xm[d] = 0.0;
    uint64_t* tq_next = tileq_slot(b, st.parity, f);
    if (res) {
        int32_t c_start;
        if constexpr (!ACC) {
            // ResArgs::lazy_q (split schedule): the k_norm in front of this launch stored no quanta and `qsrc` is the filter's WEIGHTS — what the
            // speculative request above brought are the tile's raw log-weights, and their quanta are formed here, by k_norm's own expression
            // against the offset the head has just read back (the bound, or the maximum after a one-tile redo): identical integers.  A step
            // that does not resample (97 of 100 at the reference's threshold) thus moves neither the 8 bytes k_norm stored nor the 8 read here.
            if (a.lazy_q) {
#pragma unroll
                for (int k = 0; k < NORM_IPT / 2; ++k) {
                    qv[k].x = llpf_q64_unit(llpf_exp_le0(llpf_u2d(qv[k].x) - h.a), a.K);
                    qv[k].y = llpf_q64_unit(llpf_exp_le0(llpf_u2d(qv[k].y) - h.a), a.K);
                }
            }
        }
        if (b.strategy == LLPF_RESAMPLE_SYSTEMATIC) res_counts<LLPF_RESAMPLE_SYSTEMATIC>(b, a, f, tile, h, qv, sh, c_start, c_end);
        else res_counts<LLPF_RESAMPLE_STRATIFIED>(b, a, f, tile, h, qv, sh, c_start, c_end);
        first = c_start;
        last = (tile == b.P2 - 1) ? (int64_t)a.M : (int64_t)c_end;
        if (OWN_TABLE) res_owner_table(sh, sh, sh_own, c_start);
    } else {
        l = head_log(h);
        first = (int64_t)tile * TILE;
        last = first + TILE;
        if (!RBFAT) __syncthreads();      // generator tables in LDS
    }
    {   // bound of the weights produced below: max of the previous (normalised) weights + the density's peak
        const double wmx = res ? b.log1N : (h.mtrue - h.a) - l;
        if constexpr (Model::RB) {     // peak of N(0, S) of the coming correct! (or of R2 when C == 0)
            const double c0w = md->rb_zeroC ? md->dg.c0 : (st.rb_corr + f)->dS.c0;
            pc.off = (WEIGHT && st.has_y) ? wmx + c0w : wmx;
        } else {
            pc.off = (WEIGHT && st.has_y) ? wmx + c0_pre : wmx;
        }
    }
    const double lN = -b.mlogN;
    const double aux_off = ((st.aux == 2) ? md->dg.c0 : 0.0) - lN;     // lambda - log N <= c0 - log N (lambda = 0 if y1 is missing)
    if (AUX) pc.off = aux_off;
    const double* lamp = AUX ? b.lam + (size_t)f * Ns : nullptr;
    const int32_t tbase = (int32_t)(first >> 10);
    LLPF_STAMP(2);
#ifdef LLPF_DEVTOOLS
    if (a.dbg && threadIdx.x == 0 && f == 0) a.dbg[(size_t)tile * 8 + 5] = (uint64_t)(last - first);
#endif
    const uint32_t tile0 = (uint32_t)tile * TILE, ulast = (uint32_t)last, ucend = (uint32_t)c_end;
    __builtin_amdgcn_s_setprio(0);
    // tile-local source of output o (< c_end): the owner table, the descent beyond it
    auto owner_of = [&](uint32_t o) -> uint32_t {
#ifdef LLPF_DEVTOOLS
        if (a.ablate & 2) return (o - (uint32_t)first) & (TILE - 1);
#endif
        const uint32_t idx = o - (uint32_t)first;
        return (OWN_TABLE && idx < (uint32_t)OWN_CAP) ? sh_own[idx] - 1u : (uint32_t)res_owner(sh.cl, (int32_t)o);
    };
    // The output loop: source of output o (owner table / previous ancestor / o itself), its previous weight (log(1/N) after a
    // resampling; lambda - log N in the auxiliary filter's second half; else the lazily normalised stored weight: w .-= offset,
    // w .-= log(sum)), propagate [+ weight], exp-sum / quanta / tile-sum accumulation.  Written once, instantiated per value of
    // the block-uniform resample flag where that measured faster (split-schedule and Rao-Blackwellized kernels: each version
    // keeps only its own uniform values live; the merged single-filter kernel is 0.3 us faster with ONE loop and the flag
    // tested inside).
#define LLPF_OUTPUT_LOOP(RESX, NTLX) \
_Pragma("unroll 1") \
    for (uint32_t o = (uint32_t)first + threadIdx.x; o < ulast; o += BLOCK) { \
        uint32_t src = o; \
        double wprev = b.log1N; \
        if (RESX) { \
            if (o < ucend) src = tile0 + owner_of(o); \
            else src = anc_ident_prev ? o : (uint32_t)ld_off(anc, o << 2); \
            Mem<LLPF_STCOH0>::st_off(anc, o << 2, (int32_t)src); \
            if (AUX) wprev = ld_off(lamp, o << 3) - lN; \
        } else if (AUX) { \
            wprev = ld_off(lamp, o << 3) - lN; \
        } else if (WEIGHT) { \
            wprev = (ld_off(pc.w, o << 3) - h.a) - l; \
        } \
        double xs[NX]; \
        const double wv = pc.template one<LLPF_RESPROP_ST, NTLX>(src, o, wprev, bad, xs); \
        bmax = llpf_fmax(bmax, wv); \
        if (WEIGHT && ACC) { \
            double e; \
            const uint64_t q = wacc.add(wv, pc.off, st.K, st.need_e2 != 0, &e); \
            Mem<LLPF_STCOH0>::st_off(pc.qnext, o << 3, q); \
            ts.add(o, q, sh_tq, tq_next, tbase); \
            if (st.want_xmean) { \
_Pragma("unroll") \
                for (int d = 0; d < NX; ++d) xm[d] = xm[d] + xs[d] * e; \
            } \
        } \
    }
    // The same loop with the sources of LLPF_RESPROP_PF consecutive rounds requested together, for the split-schedule form — the kernel of
    // filters and banks beyond 3 M particles, whose states come from HBM instead of the Infinity Cache.  Worth 2 % at depth 2 and
    // nothing beyond (depth 4 is slower): the loop of such a launch does not wait for its gathers — what moved it was the store policy
    // (k_resprop_split.hip), which this loop takes per kind of step: LLPF_RESPROP_ST on a resampling step, LLPF_RESPROP_ST_ID on a step
    // whose outputs read their own index.  The per-output arithmetic and its order are untouched; the ancestors are stored behind the
    // loads (stores and loads share a counter).
#define LLPF_OUTPUT_LOOP_PF(RESX) \
_Pragma("unroll 1") \
    for (uint32_t ob = (uint32_t)first + threadIdx.x; ob < ulast; ob += LLPF_RESPROP_PF * BLOCK) { \
        uint32_t srcs[LLPF_RESPROP_PF]; \
        double wps[LLPF_RESPROP_PF], xps[LLPF_RESPROP_PF][NX]; \
_Pragma("unroll") \
        for (int k = 0; k < LLPF_RESPROP_PF; ++k) { \
            const uint32_t o = ob + (uint32_t)k * BLOCK; \
            srcs[k] = o; \
            if (RESX && o < ulast) { \
                if (o < ucend) srcs[k] = tile0 + owner_of(o); \
                else srcs[k] = anc_ident_prev ? o : (uint32_t)ld_off(anc, o << 2); \
            } \
        } \
_Pragma("unroll") \
        for (int k = 0; k < LLPF_RESPROP_PF; ++k) { \
            const uint32_t o = ob + (uint32_t)k * BLOCK; \
            wps[k] = b.log1N; \
            if (o < ulast) { \
                pc.template fetch<!(RESX) && LLPF_RESPROP_LD_ID>(srcs[k], xps[k]); \
                if (!(RESX) && WEIGHT) wps[k] = (ld_off(pc.w, o << 3) - h.a) - l; \
            } \
        } \
_Pragma("unroll") \
        for (int k = 0; k < LLPF_RESPROP_PF; ++k) { \
            const uint32_t o = ob + (uint32_t)k * BLOCK; \
            if (o < ulast) { \
                if (RESX) Mem<LLPF_STCOH0>::st_off(anc, o << 2, (int32_t)srcs[k]); \
                double xs[NX]; \
                const double wv = pc.template one_x<(RESX) ? LLPF_RESPROP_ST : LLPF_RESPROP_ST_ID>(xps[k], o, wps[k], bad, xs); \
                bmax = llpf_fmax(bmax, wv); \
            } \
        } \
    }
    // A step that does not resample, split-schedule form: output o reads index o, so a thread takes TWO ADJACENT outputs and every plane
    // is read and written 16 bytes per lane — three loads and three stores per pair instead of six and six.  With 8-byte accesses this
    // loop was bound by neither issue nor bandwidth but by the number of memory instructions in flight: cut to a third of its vector
    // instructions (the tile-loop experiment, EXPERIMENTS 6.12) it took the same time, 4.9 TB/s.  Per-output arithmetic and counters
    // untouched (PropCtx::one_x with STP = -1 returns what it would have stored).  The nontemporal forms of both (LLPF_RESPROP_LD_ID, _ST_ID) are
    // taken when the host says so (ResArgs::nt_id): from ~7 M particles on; a working set the size of the Infinity Cache prefers plain accesses
    // (N = 3.3e6 / 4e6 at threshold 0.1: 47.0 / 55.1 us plain, 52.6 / 62.1 nontemporal; 8e6: 109.0 / 108.0; 1.6e7: 209.3 / 202.4).  (The same pairing for the steps that DO resample — adjacent
    // outputs, 16-byte stores of both states, weights and ancestors — is 7.6 % slower than two rounds a block apart, and so is that loop with
    // only its STORES paired (neighbouring lanes exchange halves through quad_perm, the gathers keep their spread): 9 % slower — 16-byte
    // plain stores beside a gather cost more than twice as many 8-byte ones.  Both pairs of a thread requested before the first is computed:
    // no better, 169.2 / 170.0 us on the C4 share.  profiles/r06_paired_outputs_ab.txt.)
#define LLPF_OUTPUT_LOOP_ID2(LDNT, STNT) \
_Pragma("unroll 1") \
    for (uint32_t o2 = (uint32_t)first + 2u * threadIdx.x; o2 < ulast; o2 += 2u * BLOCK) { \
        typedef double __attribute__((ext_vector_type(2))) d2_t; \
        d2_t xv[NX], wv2 = {b.log1N, b.log1N}; \
_Pragma("unroll") \
        for (int d = 0; d < NX; ++d) { \
            const d2_t* src_ = reinterpret_cast<const d2_t*>(reinterpret_cast<const char*>(pc.xc + (size_t)d * Ns) + (o2 << 3)); \
            xv[d] = (LDNT) ? __builtin_nontemporal_load(src_) : *src_; \
        } \
        if (WEIGHT) wv2 = *reinterpret_cast<const d2_t*>(reinterpret_cast<const char*>(pc.w) + (o2 << 3)); \
        double xp0[NX], xp1[NX], xs0[NX], xs1[NX]; \
_Pragma("unroll") \
        for (int d = 0; d < NX; ++d) { xp0[d] = xv[d].x; xp1[d] = xv[d].y; } \
        const double wp0 = WEIGHT ? (wv2.x - h.a) - l : b.log1N, wp1 = WEIGHT ? (wv2.y - h.a) - l : b.log1N; \
        const double w0 = pc.template one_x<-1>(xp0, o2, wp0, bad, xs0); \
        const double w1 = pc.template one_x<-1>(xp1, o2 + 1u, wp1, bad, xs1); \
        bmax = llpf_fmax(bmax, w0); \
        bmax = llpf_fmax(bmax, w1); \
_Pragma("unroll") \
        for (int d = 0; d < NX; ++d) { \
            d2_t v_; v_.x = xs0[d]; v_.y = xs1[d]; \
            d2_t* dst_ = reinterpret_cast<d2_t*>(reinterpret_cast<char*>(pc.xn + (size_t)d * Ns) + (o2 << 3)); \
            if (STNT) __builtin_nontemporal_store(v_, dst_); else *dst_ = v_; \
        } \
        if (WEIGHT) { \
            d2_t v_; v_.x = w0; v_.y = w1; \
            d2_t* dst_ = reinterpret_cast<d2_t*>(reinterpret_cast<char*>(pc.wn) + (o2 << 3)); \
            if (STNT) __builtin_nontemporal_store(v_, dst_); else *dst_ = v_; \
        } \
    }
    constexpr bool PREFETCH = LLPF_RESPROP_PF > 1 && !(WEIGHT && ACC) && !Model::RB && !AUX;
    if constexpr (WEIGHT && ACC && !Model::RB) { LLPF_OUTPUT_LOOP(res, false) }
    else if constexpr (PREFETCH) { if (res) { LLPF_OUTPUT_LOOP_PF(true) } else if (LLPF_RESPROP_ID2) { if (a.nt_id) { LLPF_OUTPUT_LOOP_ID2(LLPF_RESPROP_LD_ID != 0, LLPF_RESPROP_ST_ID == 2) } else { LLPF_OUTPUT_LOOP_ID2(false, false) } } else { LLPF_OUTPUT_LOOP_PF(false) } }
    else if (res) { LLPF_OUTPUT_LOOP(true, false) }
    else { LLPF_OUTPUT_LOOP(false, (LLPF_RESPROP_LD_ID != 0)) }
#undef LLPF_OUTPUT_LOOP
#undef LLPF_OUTPUT_LOOP_PF
#undef LLPF_OUTPUT_LOOP_ID2
    if (WEIGHT && ACC) ts.flush(sh_tq, tq_next, tbase);
    __builtin_amdgcn_s_setprio(3);
    LLPF_STAMP(3);
    if (WEIGHT) {
        const double r = block_max(bmax, sm_max);
        const int anybad = __syncthreads_or(bad ? 1 : 0);
        if (threadIdx.x == 0) acc_max(b.acc + (size_t)f * ACC_WORDS, st.parity, r, anybad != 0);
        int exact = 0;
        if constexpr (ACC && ONE) {
            // This block is the whole filter (one tile): make the bound test of the coming head here and, if it fails,
            // redo the sums in exact-max form at once (same arithmetic as k_norm's exact form; the head is told through
            // exact_slot).  Banks of many small filters otherwise pay a host round trip at nearly every timestep.
            const llpf_u128 sw = wave_sum_u128(wacc.S);
            if ((threadIdx.x & 63) == 0) { sm_acc[threadIdx.x >> 6][0] = sw.lo; sm_acc[threadIdx.x >> 6][1] = sw.hi; }
            __syncthreads();
            llpf_u128 tot = {sm_acc[0][0], sm_acc[0][1]};
            for (int k = 1; k < BLOCK / 64; ++k) { const llpf_u128 t1 = {sm_acc[k][0], sm_acc[k][1]}; tot = llpf_u128_add(tot, t1); }
            __syncthreads();
            if (anybad || tot.hi < ((uint64_t)1 << 22)) {
                exact = 1;
                const double mx = anybad ? llpf_u2d(0x7ff8000000000000ULL) : r;
                wacc.init();
                ts.init();
                if (threadIdx.x < 8) sh_tq[threadIdx.x] = 0;
#pragma unroll
                for (int d = 0; d < NX; ++d) xm[d] = 0.0;
                __syncthreads();
#pragma unroll 1
                for (uint32_t o = (uint32_t)first + threadIdx.x; o < ulast; o += BLOCK) {
                    const double wv = ld_off(pc.wn, o << 3);
                    double e;
                    const uint64_t q = wacc.add(wv, mx, st.K, st.need_e2 != 0, &e);
                    st_off(pc.qnext, o << 3, q);
                    ts.add(o, q, sh_tq, tq_next, tbase);
                    if (st.want_xmean) {
#pragma unroll
                        for (int d = 0; d < NX; ++d) xm[d] = xm[d] + ld_off(pc.xn + (size_t)d * Ns, o << 3) * e;
                    }
                }
                ts.flush(sh_tq, tq_next, tbase);
            }
        }
        if (ACC) {
            wacc.flush(b.acc + (size_t)f * ACC_WORDS, st.parity, st.need_e2 != 0, sm_acc);
            __syncthreads();
            if (threadIdx.x < 8 && sh_tq[threadIdx.x])
                atomicAdd(reinterpret_cast<unsigned long long*>(tq_next + tbase + threadIdx.x), (unsigned long long)sh_tq[threadIdx.x]);
            if (st.want_xmean) block_store_xm<NX>(xm, xmpart_slot(b, st.parity, f) + (size_t)tile * MAXD, sm_x);
        }
        if (tile == 0 && threadIdx.x == 0) {
            if (AUX) {     // the weights just written are final values (no pending normalisation); aux_off bounds them
                sc->norm_pending = 0;
                sc->uniform = 0;
                sc->wmax = aux_off;
            }
            if (ACC) sc->xm_parts = b.P2;
            sc->off_slot[st.parity] = pc.off;
            sc->exact_slot[st.parity] = exact;
            sc->e2v_slot[st.parity] = st.need_e2;
            sc->u_slot[st.parity] = llpf_uniform_step(sb + st.next_step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1);
        }
    }
    __syncthreads();
    LLPF_STAMP(4);
#undef LLPF_STAMP
    if (tile == b.P2 - 1 && threadIdx.x == 0) {        // bookkeeping of this predict! (by the only block that reads anc_ident)
        const int r = res ? 1 : 0;
        sc->anc_ident_s[b.anc_slot ^ 1] = r ? 0 : 1;
        sc->last_resampled = r;
        sc->resample_count += r;
    }
}
