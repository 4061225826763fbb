// kernels/resample.hpp — head of the resample kernels, ancestor counts, k_resample.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// k_resample — [finalize] + scan + ancestor counts + expansion, one tile per block
// ------------------------------------------------------------------------------------------------
enum { SRC_FILTER = 0, SRC_VALUES = 1 };

// Thresholds are non-decreasing in i0.  count(v) = #{ i0 in [0,M) : thr(i0) < v } is evaluated with a closed
// form whenever v is not within `delta` (in index units) of a threshold, and with the exact predicate otherwise:
// thr(i0) differs from its real-arithmetic value (r + i0/M resp. (i0+U)/M) by < 4 ulp(1), i.e. by < M * 1e-15 in
// index units, far below delta, so both paths give the count defined by the reference's comparison `s[i] < bins[b]`.
struct ThrSys {   // systematic: s[i] = fl(r + fl(i0 * (1/M)))  (resample.jl:23-24, Julia StepRangeLen getindex)
    double r, step, Md, delta;
    int32_t M;
    DEV double at(int32_t i0) const { return r + (double)i0 * step; }
    DEV int32_t count(double v) const {
        const double e = (v - r) * Md;
        if (e <= -delta) return 0;
        if (e >= Md + delta) return M;
        const double fl = __builtin_floor(e);
        const double fr = e - fl;
        int32_t c = (int32_t)fl + 1;
        c = c < 0 ? 0 : (c > M ? M : c);
        if (fr > delta && fr < 1.0 - delta && e > 0.0) return c;
        while (c < M && at(c) < v) ++c;
        while (c > 0 && !(at(c - 1) < v)) --c;
        return c;
    }
};
struct ThrStrat { // stratified: u_i = (i0 + rand()) / M * bins[N]  (resample.jl:49)
    double Md, delta, binsN;
    int32_t M;
    uint32_t step, k0, k1;
    const double* Uexp;
    DEV double at(int32_t i0) const {
        const double U = Uexp ? Uexp[i0] : llpf_uniform_idx((uint32_t)i0, step, LLPF_STREAM_STRATIFY, k0, k1);
        return ((double)i0 + U) / Md * binsN;
    }
    DEV int32_t count(double v) const {
        const double e = v * Md;
        if (e <= -delta) return 0;
        if (e >= Md + delta) return M;
        const double fl = __builtin_floor(e);
        const double fr = e - fl;
        int32_t c = (int32_t)fl;
        c = c < 0 ? 0 : (c > M ? M : c);
        if (fr > delta && fr < 1.0 - delta && e > 0.0 && c < M) return at(c) < v ? c + 1 : c;
        while (c < M && at(c) < v) ++c;
        while (c > 0 && !(at(c - 1) < v)) --c;
        return c;
    }
};

// ---- memory access policy ---------------------------------------------------------------------------------------
// COH = false: ordinary loads / stores (visible to the next LAUNCH: the kernel boundary writes the L2s back).
// COH = true : agent-scope relaxed atomics (global_load/store ... sc1): coherent at the memory side, i.e. across the eight
//              XCDs' L2s INSIDE one launch — what the persistent multi-step kernel (kernels/persist.hpp) needs for every
//              datum one block writes and another block reads after the grid barrier.  (tools/grid_barrier.hip: no stale
//              reads, 5 TB/s; __threadfence() instead costs ~30 us per step: a full L2 write-back per block.)
template <int COH>
struct Mem {
    template <class T> static DEV T ld(const T* p) {
        if constexpr (COH != 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *p;
    }
    template <class T> static DEV void st(T* p, T v) {
        if constexpr (COH == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if constexpr (COH == 2) __builtin_nontemporal_store(v, p);
        else if constexpr (COH == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else *p = v;
    }
    template <class T> static DEV T ld_off(const T* base, uint32_t byte_off) {
        return ld(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off));
    }
    template <class T> static DEV void st_off(T* base, uint32_t byte_off, T v) {
        st(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v);
    }
};
// ---- tile prefix above 1024 tiles --------------------------------------------------------------------------------
// The head needs the sum of the tile sums in front of its tile and their total.  Up to 1024 tiles every block reads all of them
// (four independent loads per thread; measured cheaper than any two-level form, EXPERIMENTS 5.4).  Above, that is a P2^2 burst —
// 15 625 tiles at N = 1.6e7: 2 GB of L2 reads per timestep; 292 969 tiles at N = 3e8: 690 GB — where the reference's cumsum is
// O(N) (src/resample.jl:19-22).  k_tile_prefix (one block per group of 1024 tiles, launched in front of every kernel with a head when
// P2 > 1024) leaves the exclusive prefix of every tile inside its group and the group totals; a head then reads <= 512 group totals
// (two per thread) and ONE prefix.  Integer sums: the same values whichever way they are added.
constexpr int TQ_GROUP = 4 * BLOCK;                   // tiles per group = what one block scans with four tiles per thread
static_assert(((int64_t)1 << 29) / TILE / TQ_GROUP <= 2 * BLOCK, "at most two group totals per thread of a head");
DEV int tq_groups(int P2) { return (P2 + TQ_GROUP - 1) / TQ_GROUP; }
template <int ONE_TU = 0>      // (a template so that the header can sit in several translation units; instantiated in kernels.hip only)
__global__ __launch_bounds__(BLOCK) void k_tile_prefix(BankDev b, int parity) {
    __shared__ uint64_t red[BLOCK / 64];
    const int f = blockIdx.y, g = blockIdx.x, t = (int)threadIdx.x, lane = t & 63, wvid = t >> 6;
    const uint64_t* __restrict__ tq = tileq_slot(b, parity, f);
    const int p0 = g * TQ_GROUP + 4 * t;
    uint64_t v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (p0 + k < b.P2) ? tq[p0 + k] : 0;
    const uint64_t tsum = (v[0] + v[1]) + (v[2] + v[3]);
    const uint64_t incl = wave_scan_u64(tsum);
    if (lane == 63) red[wvid] = incl;
    __syncthreads();
    uint64_t run = incl - tsum, tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) { if (k < wvid) run += red[k]; tot += red[k]; }
    uint64_t* tp = b.tpre + (size_t)f * b.P2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (p0 + k < b.P2) tp[p0 + k] = run; run += v[k]; }
    if (t == 0) b.gsum[(size_t)f * tq_groups(b.P2) + g] = tot;
}

// ---- shared machinery of the resample kernels -----------------------------------------------------------------
struct __attribute__((aligned(16))) ResShared {   // LDS scratch
    uint64_t red[BLOCK / 64][4];
    uint64_t accw[8];
    double dval[4];
    uint32_t cl[TILE];
};
struct ResHead {                   // block-uniform results of res_head()
    double a;                      // offset of the pending normalisation (bound or maximum)
    double mtrue;                  // true maximum of the raw weights
    double s;                      // exact form only: sum_{i != argmax} e_i
    double stot, e2;               // sum e_i (all particles), sum e_i^2 (-1: not accumulated)
    uint64_t prefix, tot;          // exclusive prefix of this tile's quanta, total of all quanta
    int dr, status, uniform, fast;
    int has_u; double u_sys;       // systematic offset of this step supplied by the caller
};
enum { RES_STATUS_FALLBACK = 100, RES_STATUS_SKIP = 101 };   // SKIP: this launch is a no-op for the filter   // internal: bound test failed, the host redoes this step in exact form

// shouldresample (reference src/resample.jl:5-10) without the division: ESS = stot^2 / sum(e^2) < N*thr
DEV int decide_resample(double thr, double N, double stot, double e2) {
    if (thr == 1.0) return 1;
    return (stot * stot < (N * thr) * e2) ? 1 : 0;
}
// log(sum exp(w - a)) in the form the normalisation was accumulated in
DEV double head_log(const ResHead& h) { return h.fast ? llpf_log(h.stot) : llpf_log1p_nonneg(h.s); }

// Head of a resample launch: all global loads are issued first (accumulator slots, per-tile quanta sums), one
// __syncthreads, then EVERY thread derives the block-uniform scalars (integer sums => identical everywhere).
// Tile 0 publishes the scalars of logsumexp! / effective_particles / shouldresample for later kernels.
// `defer_skip`: the caller fetched the run's stop flag and the filter's fallback flag without waiting for them; the
// launch-is-a-no-op test is made here after the barrier, so that those two loads overlap all the others.
struct NoOverlap { DEV void operator()() const {} };
// `overlap`: work of the caller that needs nothing from memory the head waits for (particle-independent model terms: their
// own scalar loads); it runs after the head's vector loads are issued and before the first of them is consumed
template <int SRC, bool COH = false, class Overlap = NoOverlap>
DEV ResHead res_head(const BankDev& b, const ResArgs& a, int f, int tile, ResShared& sh,
                     bool defer_skip = false, uint32_t stop_flag = 0, int fb_flag = 0, Overlap&& overlap = NoOverlap()) {
    FilterScal* sc = b.scal + f;
    uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const double Nd = (double)b.N;
    ResHead h;
    h.has_u = 0; h.u_sys = 0.0;
    const bool fin = (a.mode & RES_FINALIZE) != 0;
    const bool unif0 = (SRC == SRC_FILTER) && !fin && sc->uniform;
    // scalars of the previous launch that are needed after the barrier below: fetched now, with the other loads
    double off_pre = sc->off_slot[a.parity];
    int e2v_pre = sc->e2v_slot[a.parity];
    int exact_pre = sc->exact_slot[a.parity];
    int status_pre = sc->status;

    // loads.  wave 0: lane group g (8 lanes = 8 shards) fetches word g of this slot's accumulator set
    uint64_t accv = 0;
    const int grp = threadIdx.x / NSHARD, shard = threadIdx.x % NSHARD;
    if (fin && threadIdx.x < 64) accv = Mem<COH>::ld(acc_slot(acc, acc_word_of_group(grp, a.parity), shard));
    uint64_t pre = 0, all = 0;
    const bool want_tq = (a.mode & RES_RESAMPLE) && !unif0;
    const bool tq_small = b.P2 <= 4 * BLOCK;   // up to 1024 tiles: four INDEPENDENT loads per thread (a loop waits for every one in turn)
    const uint64_t* __restrict__ tq = tileq_slot(b, a.parity, f);
    uint64_t tqv[4] = {0, 0, 0, 0};
    if (want_tq && tq_small) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int p = (int)threadIdx.x + j * BLOCK; tqv[j] = Mem<COH>::ld(tq + (p < b.P2 ? p : 0)); }
    } else if (want_tq) {
        // above 1024 tiles: the group totals and this tile's prefix inside its group, left by k_tile_prefix (launched in front of this kernel)
        const int t = (int)threadIdx.x, G = tq_groups(b.P2);
        const uint64_t* __restrict__ gs = b.gsum + (size_t)f * G;
        tqv[0] = Mem<COH>::ld(gs + (t < G ? t : 0));
        tqv[1] = Mem<COH>::ld(gs + (t + BLOCK < G ? t + BLOCK : 0));
        if (t == 0) tqv[2] = Mem<COH>::ld(b.tpre + (size_t)f * b.P2 + tile);
    }
    overlap();
    if (want_tq) {
        if (tq_small) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = (int)threadIdx.x + j * BLOCK;
                const uint64_t q = p < b.P2 ? tqv[j] : 0;
                all += q;
                if (p < tile) pre += q;
            }
        } else {
            const int t = (int)threadIdx.x, G = tq_groups(b.P2), g = tile / TQ_GROUP;
            const uint64_t q0 = t < G ? tqv[0] : 0, q1 = t + BLOCK < G ? tqv[1] : 0;
            all = q0 + q1;
            pre = (t < g ? q0 : 0) + (t + BLOCK < g ? q1 : 0) + tqv[2];
        }
    }
    if (fin && wvid == 0) {
        // combine the 8 shards of each word inside its group of 8 lanes: xor 1, xor 2 (quad_perm), xor 4 (half mirror)
#define LLPF_ACCSTEP(CTRL) { const uint64_t t = dpp_u64<CTRL, 0xF, false>(accv, accv); accv = (grp == 0) ? (t > accv ? t : accv) : accv + t; }
        LLPF_ACCSTEP(DPP_QUAD_XOR1) LLPF_ACCSTEP(DPP_QUAD_XOR2) LLPF_ACCSTEP(DPP_ROW_HALF_MIRROR)
#undef LLPF_ACCSTEP
        if (shard == 0) sh.accw[grp] = accv;
    }
    pre = wave_sum_u64(pre);
    all = wave_sum_u64(all);
    if (lane == 0) { sh.red[wvid][0] = pre; sh.red[wvid][1] = all; }
    __syncthreads();
    if (SRC == SRC_FILTER) {
        // FilterScal is written by these kernels, so the scalars fetched above are vector loads made uniform with
        // v_readfirstlane — placed by the compiler right behind the loads, with a wait, BEFORE the accumulator and tile-sum
        // loads are issued.  The asm pins their first use here, behind the barrier.
        uint32_t ol = (uint32_t)llpf_d2u(off_pre), oh = (uint32_t)(llpf_d2u(off_pre) >> 32);
        asm volatile("" : "+v"(ol), "+v"(oh), "+v"(e2v_pre), "+v"(exact_pre), "+v"(status_pre), "+v"(stop_flag), "+v"(fb_flag));
        ol = __builtin_amdgcn_readfirstlane(ol); oh = __builtin_amdgcn_readfirstlane(oh);
        off_pre = llpf_u2d(((uint64_t)oh << 32) | ol);
        e2v_pre = __builtin_amdgcn_readfirstlane(e2v_pre); exact_pre = __builtin_amdgcn_readfirstlane(exact_pre);
        status_pre = __builtin_amdgcn_readfirstlane(status_pre);
        stop_flag = __builtin_amdgcn_readfirstlane(stop_flag); fb_flag = __builtin_amdgcn_readfirstlane(fb_flag);
    }
    h.status = 0;
    h.s = 0.0;
    if (defer_skip) {
        const bool stopped = stop_flag != 0 && (int64_t)(stop_flag - 1) < a.k;
        if (stopped || (a.only_fallback ? !fb_flag : (fb_flag != 0))) { h.status = RES_STATUS_SKIP; return h; }
    }
    h.prefix = 0; h.tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) { h.prefix += sh.red[k][0]; h.tot += sh.red[k][1]; }
    if (fin) {
        // clear the slot after next (its last reader finished two launches ago): accumulator words and tile sums
        const int clr = (a.parity + 2) % ACC_NSLOT;
        if (tile == 0 && threadIdx.x >= 64 && threadIdx.x < 128) Mem<COH>::st(acc_slot(acc, acc_word_of_group(grp - 8, clr), shard), (uint64_t)0);
        uint64_t* tqc = tileq_slot(b, clr, f);
        if (gridDim.x == (unsigned)b.P2) { if (threadIdx.x == 0) Mem<COH>::st(tqc + tile, (uint64_t)0); }
        else { for (int p = threadIdx.x; p < b.P2; p += BLOCK) tqc[p] = 0; }      // finalize-only launch: one block
    }
    if (fin) {
        h.fast = a.fast_head && !exact_pre;            // a one-tile filter may have redone its sums in exact form already
        h.mtrue = max_unkey(sh.accw[0]);
        const llpf_u128 s128 = acc_combine_u128(sh.accw[1], sh.accw[2], sh.accw[3]);
        const llpf_u128 e128 = acc_combine_u128(sh.accw[4], sh.accw[5], sh.accw[6]);
        const bool bad = sh.accw[7] != 0;
        if (h.fast) {
            h.a = off_pre;                                        // published by the weighting kernel that filled this slot
            if (bad || s128.hi < ((uint64_t)1 << 22)) {           // sum exp(w - bound) < 2^-10, or NaN weights
                h.status = RES_STATUS_FALLBACK;
                h.stot = 0.0;
            } else {
                h.stot = llpf_fix96_to_double(s128);
            }
        } else {
            h.a = h.mtrue;
            if (bad || s128.hi < ((uint64_t)1 << 32)) {           // max is -Inf / NaN, or NaN weights: degenerate
                h.stot = llpf_u2d(0x7ff8000000000000ULL);
                h.s = h.stot;
                h.status = LLPF_ERR_DEGENERATE;
            } else {
                h.s = llpf_fix96_to_double(llpf_fix96_minus_one(s128));     // sum_all_but: exact, one rounding
                h.stot = h.s + 1.0;
            }
        }
        const int e2v = e2v_pre;
        h.e2 = e2v ? llpf_fix96_to_double(e128) : -1.0;           // -1: not accumulated (threshold 1: not needed)
        h.uniform = 0;
        h.dr = h.status ? 0 : decide_resample(b.thr, Nd, h.stot, h.e2);
        if (tile == 0 && threadIdx.x == 0) {
            if (h.status == RES_STATUS_FALLBACK) {
                sc->fallback = 1;
                sc->fb_step = a.k;
                *b.bank_flag = (uint32_t)(a.k + 1);
            } else {
                double l, inv, ll, ess;
                if (h.status) { l = h.stot; inv = h.stot; ll = h.stot; ess = h.stot; }
                else {
                    l = head_log(h);
                    inv = 1.0 / h.stot;
                    ll = l + h.a;
                    ess = h.e2 > 0.0 ? (h.stot * h.stot) / h.e2 : -1.0;
                }
                sc->m = h.a; sc->mtrue = h.mtrue; sc->s = h.s; sc->stot = h.stot; sc->l = l; sc->inv = inv; sc->ll = ll;
                sc->ess = ess; sc->e2 = h.e2; sc->fast = h.fast; sc->e2_valid = e2v;
                sc->wmax = (h.mtrue - h.a) - l;                   // normalised weight of the best particle
                sc->K = a.K;
                sc->uniform = 0;
                sc->norm_pending = a.keep_norm ? 0 : 1;
                if (a.keep_norm) sc->wmax = h.mtrue;              // set_weights: w stays as installed
                if (h.status) sc->status = h.status;
                sc->do_resample = (h.dr || a.force) ? 1 : 0;      // `force`: the auxiliary filter resamples whatever the ESS (filtering.jl:206, 225)
                if (a.accumulate) sc->ll_total = sc->ll_total + ll;
                if (a.ll_steps) a.ll_steps[(size_t)a.row * b.F + f] = ll;
                sh.dval[0] = inv;
            }
        }
        if (!h.status) h.status = status_pre;          // sticky until reset! (written above only when non-zero)
    } else {
        // predict! without a preceding correct! in this launch sequence: decide from the stored state
        h.a = sc->m; h.mtrue = sc->mtrue; h.s = sc->s; h.stot = sc->stot; h.e2 = sc->e2; h.fast = sc->fast;
        h.status = sc->status; h.uniform = (SRC == SRC_FILTER) ? sc->uniform : 0;
        if (h.uniform) {
            const double wev = 1.0 / Nd;
            const double ess = 1.0 / (Nd * (wev * wev));
            h.dr = (b.thr == 1.0) ? 1 : (ess < Nd * b.thr ? 1 : 0);
            if (tile == 0 && threadIdx.x == 0 && !a.only_bins) sc->ess = ess;
            const uint64_t Qc = llpf_q64_unit(1.0 / Nd, a.K);
            const int64_t before = (int64_t)tile * TILE < b.N ? (int64_t)tile * TILE : b.N;
            h.prefix = (uint64_t)before * Qc;
            h.tot = (uint64_t)b.N * Qc;
        } else {
            h.dr = h.status ? 0 : decide_resample(b.thr, Nd, h.stot, h.e2);
        }
        if (SRC == SRC_FILTER && tile == 0 && threadIdx.x == 0 && !a.only_bins) sc->do_resample = h.dr;
    }

    // weighted_mean output (fixed-order fp64 sum of the tile partials; tile 0 only)
    if (fin && a.xmean && tile == 0 && !h.status) {
        __syncthreads();
        const double invb = sh.dval[0];
        const double* xp = xmpart_slot(b, a.parity, f);
        const int nparts = sc->xm_parts;
        for (int d = 0; d < b.nx; ++d) {
            double accx = 0.0;
            for (int p = threadIdx.x; p < nparts; p += BLOCK) accx = accx + xp[(size_t)p * MAXD + d];
            accx = wave_sum_f64(accx);
            __syncthreads();
            if (lane == 0) sh.red[wvid][2] = llpf_d2u(accx);
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = llpf_u2d(sh.red[0][2]);
                for (int k = 1; k < BLOCK / 64; ++k) t = t + llpf_u2d(sh.red[k][2]);
                a.xmean[((size_t)a.row * b.F + f) * b.nx + d] = t * invb;
            }
        }
    }
    return h;
}

// Scan of this tile's quanta + ancestor counts.  On return sh.cl[k] = c(bins[k]) for the tile's TILE sources
// (after a __syncthreads), and [c_start, c_end) is the range of outputs this tile produces.
template <int STRATEGY>
DEV void res_counts(const BankDev& b, const ResArgs& a, int f, int tile, const ResHead& h, const ulonglong2* qv,
                    ResShared& sh, int32_t& c_start, int32_t& c_end) {
    const FilterScal* sc = b.scal + f;
    const int64_t N = b.N;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    uint64_t cq[NORM_IPT];
    {
        const uint64_t Qc = h.uniform ? llpf_q64_unit(1.0 / (double)N, a.K) : 0;
        uint64_t run = 0;
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            uint64_t q = h.uniform ? Qc : ((k & 1) ? qv[k / 2].y : qv[k / 2].x);
            if (ib + k >= N) q = 0;
            run += q;
            cq[k] = run;
        }
    }
    const uint64_t tsum = cq[NORM_IPT - 1];
    const uint64_t incl = wave_scan_u64(tsum);         // wave inclusive scan of thread totals
    if (lane == 63) sh.red[wvid][3] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k)
        if (k < wvid) wave_off += sh.red[k][3];
    const uint64_t excl = h.prefix + wave_off + (incl - tsum);

    // bins = fl(fl(cum) * fl(1/fl(total))) and ancestor counts
    const double Td = (double)h.tot;
    const double invTd = 1.0 / Td;
    const double binsN = Td * invTd;                   // bins[N]: 1 or 1 - 2^-53
    const int32_t M = a.M;
    uint32_t cnt[NORM_IPT];
    if (STRATEGY == LLPF_RESAMPLE_SYSTEMATIC) {
        ThrSys th;
        const double U = h.has_u ? h.u_sys : (a.Uexp ? a.Uexp[0] : (a.u_from_scal ? sc->u_slot[a.parity] : llpf_uniform_step(sc->step_base + a.step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1)));
        th.M = M; th.Md = (double)M; th.step = 1.0 / (double)M;
        th.delta = 1e-9 + th.Md * 1e-13;
        th.r = U * binsN / (double)N;                  // r = rand()*bins[end]/N  (resample.jl:23)
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double bin = (double)(excl + cq[k]) * invTd;
            if (a.bins_out && ib + k < N) a.bins_out[(size_t)f * N + ib + k] = bin;
            cnt[k] = a.only_bins ? 0u : (uint32_t)th.count(bin);
        }
        c_start = a.only_bins ? 0 : th.count((double)h.prefix * invTd);
    } else {
        ThrStrat th;
        th.M = M; th.Md = (double)M; th.step = sc->step_base + a.step; th.k0 = sc->k0; th.k1 = sc->k1; th.Uexp = a.Uexp;
        th.delta = 1e-9 + th.Md * 1e-13;
        th.binsN = binsN;
#pragma unroll
        for (int k = 0; k < NORM_IPT; ++k) {
            const double bin = (double)(excl + cq[k]) * invTd;
            if (a.bins_out && ib + k < N) a.bins_out[(size_t)f * N + ib + k] = bin;
            cnt[k] = a.only_bins ? 0u : (uint32_t)th.count(bin);
        }
        c_start = a.only_bins ? 0 : th.count((double)h.prefix * invTd);
    }
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) sh.cl[threadIdx.x * NORM_IPT + k] = cnt[k];
    __syncthreads();
    c_end = (int32_t)sh.cl[TILE - 1];
}

// output o (c_start <= o < c_end) is produced by the first source k of the tile with cl[k] > o
DEV int res_owner(const uint32_t* cl, int32_t o) {
    // branch-free descent in power-of-two steps: p4 = 4 * #{k : cl[k] <= o} (cl is non-decreasing and o < cl[TILE-1]).
    // The probe address is one VGPR (p4) + an immediate LDS offset, so a step is ds_read + compare + select + add.
    const uint32_t ov = (uint32_t)o;
    const char* base = reinterpret_cast<const char*>(cl);
    uint32_t p4 = 0;
#pragma unroll
    for (int step = TILE / 2; step >= 1; step >>= 1) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(base + p4 + (uint32_t)(step - 1) * 4u);
        p4 += (v <= ov) ? (uint32_t)step * 4u : 0u;
    }
    return (int)(p4 >> 2);
}
static_assert(TILE == 1024, "res_owner assumes 2^10 sources per tile");

// The same map as a table: own[o - c_start] = 1 + (tile-local index of the source that produces output o) for the first
// OWN_CAP outputs of the tile (a tile produces ~TILE outputs; later ones, of tiles that hold very heavy particles, go through
// res_owner).  Each source with a non-empty range writes its index at the range's first slot, an inclusive max-scan fills
// the rest (indices grow with the slot).  ~55 instructions per thread and three barriers instead of a ten-probe descent
// (~45 instructions, ten dependent LDS round trips) per OUTPUT.  own[] must be zero on entry; sh.cl complete (res_counts).
constexpr int OWN_CAP = 2 * TILE;
// `wb` (default: the tile's first output) is the first output of the WINDOW of OWN_CAP outputs the table covers: a tile that holds very
// heavy particles walks its outputs window by window (k_resample), each window's table seeded by the sources whose ranges reach into it.
DEV void res_owner_table(const ResShared& shc, ResShared& sh, uint32_t* own, int32_t c_start, int32_t wb = -1) {
    const int t = (int)threadIdx.x, lane = t & 63, wvid = t >> 6;
    static_assert(NORM_IPT == 4 && OWN_CAP == 8 * BLOCK, "one 16-byte read of cl and two of own per thread");
    const uint32_t w0 = (uint32_t)(wb < 0 ? c_start : wb);
    const uint4 c4 = *reinterpret_cast<const uint4*>(shc.cl + 4 * t);
    uint32_t prev = t ? shc.cl[4 * t - 1] : (uint32_t)c_start;
    const uint32_t cur[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t lo = prev > w0 ? prev : w0;          // first output of this source inside the window
        const uint32_t idx = lo - w0;
        if (cur[k] > lo && idx < (uint32_t)OWN_CAP) own[idx] = (uint32_t)(4 * t + k + 1);
        prev = cur[k];
    }
    __syncthreads();
    uint4* own4 = reinterpret_cast<uint4*>(own);
    uint4 a = own4[2 * t], b = own4[2 * t + 1];
    a.y = a.y > a.x ? a.y : a.x; a.z = a.z > a.y ? a.z : a.y; a.w = a.w > a.z ? a.w : a.z;
    b.x = b.x > a.w ? b.x : a.w; b.y = b.y > b.x ? b.y : b.x; b.z = b.z > b.y ? b.z : b.y; b.w = b.w > b.z ? b.w : b.z;
    const uint32_t incl = wave_scan_max_u32(b.w);
    uint32_t base = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
    if (lane == 63) sh.red[wvid][3] = incl;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BLOCK / 64 - 1; ++k) {
        const uint32_t c = (uint32_t)sh.red[k][3];
        if (k < wvid) base = c > base ? c : base;
    }
    a.x = a.x > base ? a.x : base; a.y = a.y > base ? a.y : base; a.z = a.z > base ? a.z : base; a.w = a.w > base ? a.w : base;
    b.x = b.x > base ? b.x : base; b.y = b.y > base ? b.y : base; b.z = b.z > base ? b.z : base; b.w = b.w > base ? b.w : base;
    own4[2 * t] = a; own4[2 * t + 1] = b;
    __syncthreads();
}

template <int STRATEGY, int SRC>
__global__ __launch_bounds__(BLOCK) void k_resample(BankDev b, ResArgs a) {
    __shared__ ResShared sh;
    __shared__ __attribute__((aligned(16))) uint32_t sh_own[OWN_CAP];
    {
        const uint4 z = {0u, 0u, 0u, 0u};
        reinterpret_cast<uint4*>(sh_own)[2 * threadIdx.x] = z;
        reinterpret_cast<uint4*>(sh_own)[2 * threadIdx.x + 1] = z;
    }
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    // the stop / fallback flags are requested with the head's loads and tested behind its barrier (res_head, defer_skip):
    // tested first they were two memory round trips before anything else was asked for; the tile's quanta are requested last
    const uint32_t stop_flag = (SRC == SRC_FILTER) ? *b.bank_flag : 0u;
    const int fb_flag = (SRC == SRC_FILTER) ? b.scal[f].fallback : 0;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
    auto load_quanta = [&]() {
        if (a.mode & RES_RESAMPLE) {
#pragma unroll
            for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
        }
    };
    const ResHead h = res_head<SRC>(b, a, f, tile, sh, SRC == SRC_FILTER, stop_flag, fb_flag, load_quanta);
    if (h.status == RES_STATUS_SKIP) return;
    if (!(a.mode & RES_RESAMPLE)) return;
    if (h.status) return;
    if (!a.force && !h.dr) {
        if (SRC == SRC_FILTER && a.count_surv && threadIdx.x == 0) {      // j = 1:N: every particle is its own ancestor
            const int64_t left = b.N - (int64_t)tile * TILE;
            if (left > 0) b.surv[((size_t)f * b.P2 + tile) * 4] += (unsigned long long)(left < TILE ? left : TILE);
        }
        return;
    }
    if (h.tot == 0) return;
    int32_t c_start, c_end;
    res_counts<STRATEGY>(b, a, f, tile, h, qv, sh, c_start, c_end);
    if (a.only_bins) return;
    if (SRC == SRC_FILTER && a.count_surv) {      // distinct ancestors of this tile: sources with a non-empty output range
        const uint4 c4 = *reinterpret_cast<const uint4*>(sh.cl + 4 * threadIdx.x);
        const uint32_t p0 = threadIdx.x ? sh.cl[4 * threadIdx.x - 1] : (uint32_t)c_start;
        const int n = (c4.x > p0) + (c4.y > c4.x) + (c4.z > c4.y) + (c4.w > c4.z);
        const unsigned long long ws = wave_sum_u64((uint64_t)n);
        if ((threadIdx.x & 63) == 0 && ws) b.surv[((size_t)f * b.P2 + tile) * 4 + (threadIdx.x >> 6)] += ws;      // one entry per wave: no block reduction
    }
    int32_t* ao = a.anc_out + (size_t)f * b.Ns;
    const bool table = c_end - c_start >= BLOCK;       // block-uniform; a table for a handful of outputs would not pay
    if (table) {
        // window by window of OWN_CAP outputs: a tile that holds very heavy particles (peaked likelihoods: BASELINE C3) owns many times
        // TILE outputs, and beyond the first table every output used to cost a ten-probe descent (ten dependent LDS round trips)
        for (int32_t wb = c_start; wb < c_end; wb += OWN_CAP) {
            if (wb != c_start) {
                __syncthreads();                           // the previous window's table has been read
                const uint4 z = {0u, 0u, 0u, 0u};
                reinterpret_cast<uint4*>(sh_own)[2 * threadIdx.x] = z;
                reinterpret_cast<uint4*>(sh_own)[2 * threadIdx.x + 1] = z;
                __syncthreads();
            }
            res_owner_table(sh, sh, sh_own, c_start, wb);
            const int32_t we = (c_end - wb) < OWN_CAP ? c_end : wb + OWN_CAP;
            for (int32_t o = wb + threadIdx.x; o < we; o += BLOCK)
                wt_store(ao + o, (int32_t)((int64_t)tile * TILE + (int)sh_own[o - wb] - 1));
        }
    } else {
        for (int32_t o = c_start + threadIdx.x; o < c_end; o += BLOCK)
            wt_store(ao + o, (int32_t)((int64_t)tile * TILE + res_owner(sh.cl, o)));
    }
    // outputs whose threshold is >= bins[N] are never written by the reference (j keeps its previous
    // value); the previous value is only materialised here if it was the identity 1:N
    if (tile == b.P2 - 1 && SRC == SRC_FILTER && b.scal[f].anc_ident_s[b.anc_slot]) {
        for (int32_t o = c_end + threadIdx.x; o < a.M; o += BLOCK) ao[o] = o;
    }
}
