// kernels/persist.hpp — k_persist: MANY timesteps of the fused predict! (k_resprop) in ONE cooperative launch.
// Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// Why.  A particle-filter timestep has exactly one grid-wide dependency: the sums of the new weights (logsumexp!, the bins'
// total and the tile prefixes) must be complete before any block can resample.  With one launch per timestep that
// dependency is paid at the kernel boundary: ~3 us from the last wave of launch k to the first of launch k+1, 1.2 us of
// dispatch ramp, and a head whose first loads miss every cache (L2 invalidated, kernel arguments fetched again): ~8 of the
// 27.6 us of a C2 timestep.  Inside a persistent kernel the same dependency is a barrier over the resident blocks:
// 2.3 us for 977 blocks with 32 arrival shards (tools/grid_barrier.hip).
// How.  grid = P2 tiles, all co-resident (cooperative launch; the host checks the occupancy).  The eight XCDs' L2s are
// not coherent with each other inside a launch, so every datum that crosses blocks between steps — particles, weights,
// quanta, ancestors, accumulator words, tile sums — is accessed with agent-scope relaxed atomics (Mem<true>: sc1 loads /
// stores, coherent at the memory side; atomics were memory-side already), and each thread waits for its stores before
// its block arrives (s_waitcnt vmcnt(0): gfx9 counts stores there).  Block-uniform values that an ordinary launch reads
// back from FilterScal (the bound of the previous weights, the sticky status, "the last predict! did not resample") are
// carried in registers; tile 0 still publishes everything, so that a run can continue with ordinary launches (the last
// timestep, which has no weighting phase; the exact redo after a failed bound test).
// The arithmetic is k_resprop<Model, NX, NY, true, true>'s, statement by statement: results are bit-identical
// (tests/test_gpu_persist.py).  Every spin is bounded: a barrier that does not complete sets FilterScal::status and all
// blocks leave.
// ------------------------------------------------------------------------------------------------
constexpr int BAR_NSHARD = 32;                       // arrival shards, one 128-B line each
constexpr int BAR_STRIDE = 32;                       // u32 per line
constexpr int BAR_WORDS = (BAR_NSHARD + 2) * BAR_STRIDE;   // arrive[32], top, gen
constexpr int GQ_GROUPS = 32, GQ_STRIDE = 16;        // group sums of the tile sums: 32 tiles per group, one 128-B line per group
constexpr int GQ_WORDS64 = ACC_NSLOT * GQ_GROUPS * GQ_STRIDE;   // three slots, rotating like the accumulator slots
enum { LLPF_STATUS_BARRIER_TIMEOUT = 90 };           // internal: reported by the host as LLPF_ERR_HIP

// monotonic counters: arrive[sh] reaches (g+1) * blocks_in_shard, top reaches (g+1) * shards_in_use, then gen = g+1
DEV bool grid_barrier(uint32_t* bar, uint32_t& g, int nblocks) {
    __builtin_amdgcn_s_waitcnt(0);                   // this thread's stores and atomics are acknowledged
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const uint32_t sh = blockIdx.x & (BAR_NSHARD - 1);
        const uint32_t in_shard = ((uint32_t)nblocks - sh + BAR_NSHARD - 1) / BAR_NSHARD;
        const uint32_t shards = nblocks < BAR_NSHARD ? (uint32_t)nblocks : (uint32_t)BAR_NSHARD;
        uint32_t* top = bar + BAR_NSHARD * BAR_STRIDE;
        uint32_t* gen = top + BAR_STRIDE;
        const uint32_t prev = __hip_atomic_fetch_add(bar + sh * BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == (g + 1u) * in_shard) {
            const uint32_t p2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p2 + 1u == (g + 1u) * shards) __hip_atomic_store(gen, g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t spins = 0;
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
            if (++spins > (1u << 22)) { ok = false; break; }      // ~seconds: another tenant holds CUs, or a block died
        }
    }
    g++;
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

// arguments of a persistent run of timesteps [k_begin, k_end) — all of them have a weighting phase (k + 1 < T)
struct PersistArgs {
    int64_t k_begin, k_end;
    double t_index0, Ts;         // t_k = (t_index0 + k) * Ts
    const double* U;             // [T][nu] or nullptr
    const double* Y;             // [T][ny]; a row whose first element is NaN is a missing measurement
    double* x0; double* x1;      // particle planes; timestep k reads x[(cur0 + k) & 1]
    uint64_t* q0; uint64_t* q1;  // quanta; timestep k reads q[(qcur0 + k) & 1]   (qcur0 = buffer of the current quanta at k = 0)
    int32_t cur0, qcur0;
    int32_t par0;                // accumulator slot the head of timestep 0 reads
    uint32_t step0;              // Philox step (relative) of timestep 0's predict!
    int32_t np0;                 // n_predict at timestep 0 (anc_ident entry parity)
    int32_t need_e2, K;
    double* ll_steps;            // [T] or nullptr
    uint32_t* bar;               // [BAR_WORDS]
    uint64_t* gq;                // [GQ_WORDS64], zero at launch
    int32_t ablate, dbg_step;    // dbg_step: relative timestep whose phases are stamped into dbg (developer aid), or -1
    uint64_t* dbg;               // [P2][8] wall_clock64 stamps: step start, after head, after counts, after loop, after tail, #outputs, after barrier
};

template <class Model, int NX, int NY>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_persist(BankDev b0, const ModelD* __restrict__ models, PersistArgs pa) {
    constexpr bool COH = true;
    __shared__ ResShared sh;
    __shared__ double sm_max[BLOCK / 64];
    __shared__ uint64_t sm_acc[BLOCK / 64][5];
    __shared__ uint64_t sh_tq[8];
    const int f = 0;
    const int tile = blockIdx.x;
    const int64_t Ns = b0.Ns;
    const ModelD* md = models;
    FilterScal* sc = b0.scal;
    const uint32_t key0 = sc->k0, key1 = sc->k1, sb = sc->step_base;
    const double c0_pre = md->dg.c0;
    const double lN1 = b0.log1N;
    uint64_t* acc = b0.acc;
    int32_t* anc = b0.anc;
    double* wbuf = b0.w;

    // state carried between timesteps (block-uniform; written by earlier LAUNCHES, hence plain loads)
    StepCarry carry;
    carry.off = sc->off_slot[pa.par0];
    carry.e2v = sc->e2v_slot[pa.par0];
    carry.status = sc->status;
    carry.gq = nullptr;                  // the first timestep's tile sums come from an ordinary launch: no group sums yet
    int anc_ident_prev = sc->anc_ident_s[pa.np0 & 1];
    double u_sys = sc->u_slot[pa.par0];
    uint32_t stop_flag = *b0.bank_flag;
    int fb_flag = sc->fallback;
    uint32_t gen = *(pa.bar + (BAR_NSHARD + 1) * BAR_STRIDE);     // generation the barrier counters stand at

    for (int64_t k = pa.k_begin; k < pa.k_end; ++k) {
        const int kk = (int)(k - pa.k_begin);
        // ---- this timestep's view of the bank and its arguments (what the host computes per launch, host/run.hpp) ----
        BankDev b = b0;
        const int cur = (pa.cur0 + kk) & 1, qc = (pa.qcur0 + kk) & 1;
        b.xcur = cur ? pa.x1 : pa.x0;  b.xnext = cur ? pa.x0 : pa.x1;
        b.quanta = qc ? pa.q1 : pa.q0; b.quanta_next = qc ? pa.q0 : pa.q1;
        b.anc_slot = (pa.np0 + kk) & 1;
        ResArgs a{};
        a.mode = RES_FINALIZE | RES_RESAMPLE;
        a.parity = (pa.par0 + kk) % ACC_NSLOT;
        a.K = pa.K; a.M = (int32_t)b.N; a.anc_out = b.anc; a.accumulate = 1; a.u_from_scal = 1;
        a.step = pa.step0 + (uint32_t)kk;
        a.ll_steps = pa.ll_steps; a.k = k; a.row = k; a.fast_head = 1; a.ablate = pa.ablate;
        StepArgs st{};
        st.u = b.nu > 0 ? pa.U + k * b.nu : nullptr;
        st.y = pa.Y + (k + 1) * b.ny;
        st.t_prop = (pa.t_index0 + (double)k) * pa.Ts;
        st.t_meas = (pa.t_index0 + (double)(k + 1)) * pa.Ts;
        st.step = a.step; st.next_step = a.step + 1u;
        st.parity = (pa.par0 + 1 + kk) % ACC_NSLOT;
        st.need_e2 = pa.need_e2; st.K = pa.K; st.k = k; st.accumulate = 1;
        {
            const double y0 = st.y[0];
            st.has_y = (y0 != y0) ? 0 : 1;
        }

        // ---- k_resprop<Model, NX, NY, true, true>, with coherent accesses for everything other blocks wrote ----
#define LLPF_PSTAMP(i) if (pa.dbg && kk == pa.dbg_step && threadIdx.x == 0) pa.dbg[(size_t)tile * 8 + (i)] = wall_clock64()
        LLPF_PSTAMP(0);
        if (threadIdx.x < 8) sh_tq[threadIdx.x] = 0;
        const uint64_t* qsrc = b.quanta;
        const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
        ulonglong2 qv[NORM_IPT / 2];
#pragma unroll
        for (int j = 0; j < NORM_IPT / 2; ++j) {
            qv[j].x = Mem<COH>::ld(qsrc + ib + 2 * j);
            qv[j].y = Mem<COH>::ld(qsrc + ib + 2 * j + 1);
        }
        Model model;
        double y[NY];
        auto prepare = [&]() {
            model.prepare(md, st.u, st.t_prop);
#pragma unroll
            for (int j = 0; j < NY; ++j) y[j] = st.has_y ? st.y[j] : 0.0;
        };
        ResHead h = res_head<SRC_FILTER, COH>(b, a, f, tile, sh, true, stop_flag, fb_flag, &carry, prepare);
        stop_flag = 0; fb_flag = 0;              // flags of earlier launches matter for the first timestep only
        LLPF_PSTAMP(1);
        if (h.status) break;                     // skipped, failed bound test (the host redoes the step) or sticky error: uniform
        h.has_u = 1; h.u_sys = u_sys;
        PropCtx<Model, NX, NY, true, COH> pc{b, model, md, st, y, b.xcur, b.xnext, wbuf, key0, key1, sb + st.step, a.ablate, 0.0, b.quanta_next};
        double bmax = -LLPF_INF;
        bool bad = false;
        const bool res = (h.dr || a.force) && h.tot != 0;
        int64_t first, last;
        int32_t c_end = 0;
        double l = 0.0;
        WeightAcc wacc;
        TileSum ts;
        uint64_t* gq_w = pa.gq + (size_t)(kk % ACC_NSLOT) * GQ_GROUPS * GQ_STRIDE;                   // this weighting's group sums
        uint64_t* gq_c = pa.gq + (size_t)((kk + 1) % ACC_NSLOT) * GQ_GROUPS * GQ_STRIDE;             // cleared for the next one
        if (tile == 0 && threadIdx.x < GQ_GROUPS) Mem<COH>::st(gq_c + (size_t)threadIdx.x * GQ_STRIDE, (uint64_t)0);
        wacc.init();
        ts.init(gq_w);
        uint64_t* tq_next = tileq_slot(b, st.parity, f);
        if (res) {
            int32_t c_start;
            if (b.strategy == LLPF_RESAMPLE_SYSTEMATIC) res_counts<LLPF_RESAMPLE_SYSTEMATIC>(b, a, f, tile, h, qv, sh, c_start, c_end);
            else res_counts<LLPF_RESAMPLE_STRATIFIED>(b, a, f, tile, h, qv, sh, c_start, c_end);
            first = c_start;
            last = (tile == b.P2 - 1) ? (int64_t)a.M : (int64_t)c_end;
        } else {
            l = head_log(h);
            first = (int64_t)tile * TILE;
            last = first + TILE;
        }
        {
            const double wmx = res ? lN1 : (h.mtrue - h.a) - l;
            pc.off = st.has_y ? wmx + c0_pre : wmx;
        }
        const int32_t tbase = (int32_t)(first >> 10);
        const uint32_t tile0 = (uint32_t)tile * TILE, ulast = (uint32_t)last, ucend = (uint32_t)c_end;
        LLPF_PSTAMP(2);
        if (pa.dbg && kk == pa.dbg_step && threadIdx.x == 0) pa.dbg[(size_t)tile * 8 + 5] = (uint64_t)(last - first);
        __builtin_amdgcn_s_setprio(0);      // see k_resprop: loop at low, head / counts / tail at high wave priority
#pragma unroll 1
        for (uint32_t o = (uint32_t)first + threadIdx.x; o < ulast; o += BLOCK) {
            uint32_t src = o;
            double wprev = lN1;
            if (res) {
                if (o < ucend) src = tile0 + (uint32_t)res_owner(sh.cl, (int32_t)o);
                else src = anc_ident_prev ? o : (uint32_t)Mem<COH>::ld_off(anc, o << 2);
                Mem<COH>::st_off(anc, o << 2, (int32_t)src);
            } else {
                wprev = (Mem<COH>::ld_off(pc.w, o << 3) - h.a) - l;
            }
            double xs[NX];
            const double wv = pc.one(src, o, wprev, bad, xs);
            bmax = llpf_fmax(bmax, wv);
            double e;
            const uint64_t q = wacc.add(wv, pc.off, st.K, st.need_e2 != 0, &e);
            Mem<COH>::st_off(pc.qnext, o << 3, q);
            ts.add(o, q, sh_tq, tq_next, tbase);
        }
        ts.flush(sh_tq, tq_next, tbase);
        __builtin_amdgcn_s_setprio(3);
        LLPF_PSTAMP(3);
        {
            const double r = block_max(bmax, sm_max);
            const int anybad = __syncthreads_or(bad ? 1 : 0);
            if (threadIdx.x == 0) acc_max(acc, st.parity, r, anybad != 0);
            wacc.flush(acc, st.parity, st.need_e2 != 0, sm_acc);
            __syncthreads();
            if (threadIdx.x < 8 && sh_tq[threadIdx.x])
                atomicAdd(reinterpret_cast<unsigned long long*>(tq_next + tbase + threadIdx.x), (unsigned long long)sh_tq[threadIdx.x]);
            if (threadIdx.x == 0) {      // the same eight tile sums into their (at most two) groups
                uint64_t g0 = 0, g1 = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) { if (((tbase + j) >> 5) == (tbase >> 5)) g0 += sh_tq[j]; else g1 += sh_tq[j]; }
                if (g0) atomicAdd(reinterpret_cast<unsigned long long*>(gq_w + (size_t)(tbase >> 5) * GQ_STRIDE), (unsigned long long)g0);
                if (g1) atomicAdd(reinterpret_cast<unsigned long long*>(gq_w + (size_t)((tbase >> 5) + 1) * GQ_STRIDE), (unsigned long long)g1);
            }
            const double u_next = llpf_uniform_step(sb + st.next_step, LLPF_STREAM_RESAMPLE, key0, key1);
            if (tile == 0 && threadIdx.x == 0) {
                sc->xm_parts = b.P2;
                sc->off_slot[st.parity] = pc.off;
                sc->exact_slot[st.parity] = 0;
                sc->e2v_slot[st.parity] = st.need_e2;
                sc->u_slot[st.parity] = u_next;
            }
            // what the next timestep's head needs, without a round trip through memory
            carry.off = pc.off;
            carry.e2v = st.need_e2;
            carry.gq = gq_w;
            u_sys = u_next;
        }
        {
            const int r = res ? 1 : 0;
            anc_ident_prev = r ? 0 : 1;
            if (tile == b.P2 - 1 && threadIdx.x == 0) {
                sc->anc_ident_s[b.anc_slot ^ 1] = r ? 0 : 1;
                sc->last_resampled = r;
                sc->resample_count += r;
            }
        }
        LLPF_PSTAMP(4);
        if (!grid_barrier(pa.bar, gen, (int)gridDim.x)) {
            if (threadIdx.x == 0) sc->status = LLPF_STATUS_BARRIER_TIMEOUT;
            return;
        }
        LLPF_PSTAMP(6);
#undef LLPF_PSTAMP
    }
}
