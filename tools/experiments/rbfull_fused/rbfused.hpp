// kernels/rbfused.hpp — the head (logsumexp! / ESS / shouldresample) and the ancestors of a resampling predict! INSIDE k_rbfull (round 5).
// Part of k_rbfull.hip, namespace llpf; needs kernels/resample.hpp (ResHead's arithmetic, the thresholds) and kernels/accum.hpp.
// ------------------------------------------------------------------------------------------------
// The timestep of the filter with per-particle covariance was two launches: k_resample (finalize the weighting's integer sums, decide,
// scan the quanta, expand the ancestors: 196 blocks at N = 2e5, a latency chain of 3.4 us when nothing is resampled — four steps out of
// five at the reference's threshold 0.1 — and 8-12 us when something is) and k_rbfull.  Both halves of the first are cheap for a WAVE
// to do for itself:
//   head       every wave combines the 64 accumulator words and the tile sums (integers: the same scalars in every wave, as in the
//              fused linear-Gaussian kernel, kernels/resprop.hpp); wave 0 publishes them.
//   ancestors  the persistent waves own OUTPUTS (batches of 64 consecutive particles), so a wave looks its 64 ancestors up instead of
//              expanding counts: ancestor of output o = first source b with thr(o) < bins[b] (reference src/resample.jl:25-34, 52-58) —
//              a bisection over the inclusive prefix of the tile sums finds the source tile, the tile's 1024 quanta are scanned by the
//              wave (16 per lane) and bisected in LDS.  thr and bins are formed exactly as kernels/resample.hpp forms them
//              (bins[b] = fl(fl(cum_b) * fl(1 / fl(total))), thr = fl(r + fl(o * fl(1 / M))) resp. (o + U_o) / M * bins[N]), and the
//              thresholds are non-decreasing in o, so "#{o' : thr(o') < bins[b]} > o" (the counts k_resample expands) and
//              "thr(o) < bins[b]" are the same predicate: the same ancestors, bit for bit (tests/test_gpu_rbfull.py runs every size
//              and threshold in both forms).
// The scratch is the wave's covariance slice of the LDS (17.5 KB: the inclusive tile prefix at [0, 1024), the tile in hand at
// [1024, 2048), u64 each), free at the two moments it is needed: before the first batch's planes arrive, and at the top of a loop
// iteration between reading this batch's planes out of it and requesting the next batch's into it.
// ------------------------------------------------------------------------------------------------
constexpr int RBF_FUSED_MAX_TILES = 1024;       // prefix array in the scratch (filters up to 2^20 particles; larger ones keep the two-launch form)

struct RbfHead {
    int status, dr, fast;
    double a, l, mtrue, stot;      // offset of the pending normalisation, log of the sum, true maximum, sum of the exp-weights
    uint64_t tot;                  // total of the quanta
};

// inclusive prefix of the tile sums of accumulator slot `parity` into lds[0, P2); returns the total (uniform)
DEV uint64_t rbf_tile_prefix(const BankDev& b, int parity, int f, int lane, uint64_t* lds) {
    const uint64_t* __restrict__ tq = tileq_slot(b, parity, f);
    const int P2 = b.P2;
    for (int p = lane; p < P2; p += 64) lds[p] = tq[p];
    __syncthreads();                                   // (the workgroup is this wave)
    const int C = (P2 + 63) >> 6, lo = lane * C;
    uint64_t tot_l = 0;
    for (int c = 0; c < C; ++c) if (lo + c < P2) tot_l += lds[lo + c];
    const uint64_t incl = wave_scan_u64(tot_l);
    uint64_t run = incl - tot_l;
    for (int c = 0; c < C; ++c) if (lo + c < P2) { run += lds[lo + c]; lds[lo + c] = run; }
    __syncthreads();
    return readlane_u64(incl, 63);
}

// The head of kernels/resample.hpp (res_head with RES_FINALIZE, a filter's own weights) by ONE wave: the same integers, the same
// arithmetic, the same scalars — in two halves, so that the caller can put its own loads between them: rbf_head_request asks for
// everything the head reads (64 accumulator words, the scalars the previous weighting left, the tile sums), rbf_head_finish consumes
// it.  `wave` = this wave's index in the launch: wave 0 publishes (FilterScal, ll outputs, the flags of a failed bound test) and clears
// the accumulator words of the slot after next, wave p < P2 clears tile sum p of that slot.  Leaves the inclusive tile prefix in
// lds[0, P2).
struct RbfHeadReq {
    double off_pre;
    int e2v_pre, exact_pre, status_pre;
    uint64_t accv;
    uint64_t tqv[RBF_FUSED_MAX_TILES / 64];       // tile sums lane, lane + 64, ...
};
DEV void rbf_head_request(const BankDev& b, const ResArgs& a, int f, int lane, RbfHeadReq& q) {
    const FilterScal* sc = b.scal + f;
    const uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    // scalars of the previous launch (vector loads: FilterScal is written by this kernel) and the accumulator words, requested together
    q.off_pre = sc->off_slot[a.parity];
    q.e2v_pre = sc->e2v_slot[a.parity]; q.exact_pre = sc->exact_slot[a.parity]; q.status_pre = sc->status;
    q.accv = *acc_slot(acc, acc_word_of_group(lane / NSHARD, a.parity), lane % NSHARD);
    const uint64_t* __restrict__ tq = tileq_slot(b, a.parity, f);
    const int P2 = b.P2;
#pragma unroll
    for (int j = 0; j < RBF_FUSED_MAX_TILES / 64; ++j) {
        q.tqv[j] = 0;
        if (j * 64 < P2) { const int p = lane + 64 * j; q.tqv[j] = tq[p < P2 ? p : 0]; }      // (uniform guard: the loads beyond the filter's tiles are not issued)
    }
}
DEV RbfHead rbf_head_finish(const BankDev& b, const ResArgs& a, int f, int lane, uint32_t wave, uint64_t* lds, RbfHeadReq& q) {
    FilterScal* sc = b.scal + f;
    uint64_t* acc = b.acc + (size_t)f * ACC_WORDS;
    const double Nd = (double)b.N;
    RbfHead h;
    double off_pre = q.off_pre;
    int e2v_pre = q.e2v_pre, exact_pre = q.exact_pre, status_pre = q.status_pre;
    const int grp = lane / NSHARD, shard = lane % NSHARD;
    uint64_t accv = q.accv;
    {   // the tile sums: raw into the scratch, then the inclusive prefix in place (as rbf_tile_prefix)
        const int P2 = b.P2;
#pragma unroll
        for (int j = 0; j < RBF_FUSED_MAX_TILES / 64; ++j) if (j * 64 < P2 && lane + 64 * j < P2) lds[lane + 64 * j] = q.tqv[j];
        __syncthreads();
        const int C = (P2 + 63) >> 6, lo = lane * C;
        uint64_t tot_l = 0;
        for (int c = 0; c < C; ++c) if (lo + c < P2) tot_l += lds[lo + c];
        const uint64_t incl = wave_scan_u64(tot_l);
        uint64_t run = incl - tot_l;
        for (int c = 0; c < C; ++c) if (lo + c < P2) { run += lds[lo + c]; lds[lo + c] = run; }
        __syncthreads();
        h.tot = readlane_u64(incl, 63);
    }
    // combine the 8 shards of each word inside its group of 8 lanes: xor 1, xor 2 (quad_perm), xor 4 (half mirror)
#define LLPF_ACCSTEP(CTRL) { const uint64_t t_ = dpp_u64<CTRL, 0xF, false>(accv, accv); accv = (grp == 0) ? (t_ > accv ? t_ : accv) : accv + t_; }
    LLPF_ACCSTEP(DPP_QUAD_XOR1) LLPF_ACCSTEP(DPP_QUAD_XOR2) LLPF_ACCSTEP(DPP_ROW_HALF_MIRROR)
#undef LLPF_ACCSTEP
    uint64_t aw[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) aw[g] = readlane_u64(accv, g * NSHARD);
    {
        uint32_t ol = (uint32_t)llpf_d2u(off_pre), oh = (uint32_t)(llpf_d2u(off_pre) >> 32);
        ol = __builtin_amdgcn_readfirstlane(ol); oh = __builtin_amdgcn_readfirstlane(oh);
        off_pre = llpf_u2d(((uint64_t)oh << 32) | ol);
        e2v_pre = __builtin_amdgcn_readfirstlane(e2v_pre); exact_pre = __builtin_amdgcn_readfirstlane(exact_pre);
        status_pre = __builtin_amdgcn_readfirstlane(status_pre);
    }
    // clear the slot after next (its last reader finished two launches ago): accumulator words and tile sums
    const int clr = (a.parity + 2) % ACC_NSLOT;
    if (wave == 0) *acc_slot(acc, acc_word_of_group(grp, clr), shard) = 0;
    if ((int)wave < b.P2 && lane == 0) tileq_slot(b, clr, f)[wave] = 0;
    h.status = 0;
    double s_all_but = 0.0;
    h.fast = a.fast_head && !exact_pre;
    h.mtrue = max_unkey(aw[0]);
    const llpf_u128 s128 = acc_combine_u128(aw[1], aw[2], aw[3]);
    const llpf_u128 e128 = acc_combine_u128(aw[4], aw[5], aw[6]);
    const bool bad = aw[7] != 0;
    if (h.fast) {
        h.a = off_pre;
        if (bad || s128.hi < ((uint64_t)1 << 22)) { h.status = RES_STATUS_FALLBACK; h.stot = 0.0; }      // sum exp(w - bound) < 2^-10, or NaN weights
        else h.stot = llpf_fix96_to_double(s128);
    } else {
        h.a = h.mtrue;
        if (bad || s128.hi < ((uint64_t)1 << 32)) {               // max is -Inf / NaN, or NaN weights: degenerate
            h.stot = llpf_u2d(0x7ff8000000000000ULL);
            s_all_but = h.stot;
            h.status = LLPF_ERR_DEGENERATE;
        } else {
            s_all_but = llpf_fix96_to_double(llpf_fix96_minus_one(s128));
            h.stot = s_all_but + 1.0;
        }
    }
    const double e2 = e2v_pre ? llpf_fix96_to_double(e128) : -1.0;
    h.dr = h.status ? 0 : decide_resample(b.thr, Nd, h.stot, e2);
    h.l = h.status ? h.stot : (h.fast ? llpf_log(h.stot) : llpf_log1p_nonneg(s_all_but));
    if (wave == 0 && lane == 0) {
        if (h.status == RES_STATUS_FALLBACK) {
            sc->fallback = 1;
            sc->fb_step = a.k;
            *b.bank_flag = (uint32_t)(a.k + 1);
        } else {
            double inv, ll, ess;
            if (h.status) { inv = h.stot; ll = h.stot; ess = h.stot; }
            else {
                inv = 1.0 / h.stot;
                ll = h.l + h.a;
                ess = e2 > 0.0 ? (h.stot * h.stot) / e2 : -1.0;
            }
            sc->m = h.a; sc->mtrue = h.mtrue; sc->s = s_all_but; sc->stot = h.stot; sc->l = h.l; sc->inv = inv; sc->ll = ll;
            sc->ess = ess; sc->e2 = e2; sc->fast = h.fast; sc->e2_valid = e2v_pre;
            sc->wmax = (h.mtrue - h.a) - h.l;
            sc->K = a.K;
            sc->uniform = 0;
            sc->norm_pending = 1;
            if (h.status) sc->status = h.status;
            sc->do_resample = (h.dr || a.force) ? 1 : 0;
            if (a.accumulate) sc->ll_total = sc->ll_total + ll;
            if (a.ll_steps) a.ll_steps[(size_t)a.row * b.F + f] = ll;
        }
    }
    if (!h.status) h.status = status_pre;          // sticky until reset!
    return h;
}

// Ancestor of output o (one per lane, o < M wherever the lane is used) of a resampling predict!.  lds[0, P2): the inclusive tile
// prefix (rbf_tile_prefix); lds[1024, 2048): scratch.  `stale`: what an output without an owner keeps (thresholds >= bins[N]: the
// reference writes nothing there, resample.jl:27-35).
template <int STRATEGY>
DEV int32_t rbf_wave_ancestor(const BankDev& b, const ResArgs& a, uint64_t tot, int f, uint32_t o, int lane, uint64_t* lds, int32_t stale) {
    const FilterScal* sc = b.scal + f;
    const int P2 = b.P2;
    const double Td = (double)tot, invTd = 1.0 / Td, binsN = Td * invTd;
    double thr;
    if (STRATEGY == LLPF_RESAMPLE_SYSTEMATIC) {
        const double U = a.Uexp ? a.Uexp[0] : (a.u_from_scal ? sc->u_slot[a.parity] : llpf_uniform_step(sc->step_base + a.step, LLPF_STREAM_RESAMPLE, sc->k0, sc->k1));
        const double r = U * binsN / (double)b.N;                // ThrSys: r = rand()*bins[end]/N  (resample.jl:23)
        thr = r + (double)(int32_t)o * (1.0 / (double)a.M);
    } else {
        const double U = a.Uexp ? a.Uexp[o] : llpf_uniform_idx(o, sc->step_base + a.step, LLPF_STREAM_STRATIFY, sc->k0, sc->k1);
        thr = ((double)(int32_t)o + U) / (double)a.M * binsN;    // ThrStrat (resample.jl:49)
    }
    // the source tile: first T with thr < bins[last source of tile T]
    int lo = 0, hi = P2;
#pragma unroll 1
    for (int s = 0; s < 11; ++s) {
        if (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (thr < (double)lds[mid] * invTd) hi = mid; else lo = mid + 1;
        }
    }
    const int T = lo;                                            // P2: no owner
    int32_t j = stale;
    // every tile some lane needs, one after the other (64 consecutive thresholds: one or two tiles, more only where the weights vanish)
    int tmin = T, tmax = T < P2 ? T : -1;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { const int x = __shfl_xor(tmin, m, 64), y = __shfl_xor(tmax, m, 64); tmin = x < tmin ? x : tmin; tmax = y > tmax ? y : tmax; }
    const uint64_t* __restrict__ qf = b.quanta + (size_t)f * b.Ns;
    uint64_t* tile = lds + RBF_FUSED_MAX_TILES;
#pragma unroll 1
    for (int Tc = tmin; Tc <= tmax; ++Tc) {
        if (__ballot(T == Tc) == 0) continue;                    // wave-uniform
        ulonglong2 qv[8];
        const uint64_t* src = qf + (size_t)Tc * TILE + lane * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(src + 2 * k);
        uint64_t c[16], run = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { run += qv[k].x; c[2 * k] = run; run += qv[k].y; c[2 * k + 1] = run; }
        const uint64_t incl = wave_scan_u64(run);
        const uint64_t base = (Tc ? lds[Tc - 1] : 0) + (incl - run);
        __syncthreads();                                         // the previous tile's bisections are done
#pragma unroll
        for (int k = 0; k < 16; ++k) tile[lane * 16 + k] = base + c[k];
        __syncthreads();
        if (T == Tc) {
            int l2 = 0, h2 = TILE - 1;                           // first b with thr < bins[b]: exists (the tile's last bin exceeds thr), so 1024 candidates: ten halvings
#pragma unroll 1
            for (int s = 0; s < 10; ++s) {
                const int mid = (l2 + h2) >> 1;
                if (thr < (double)tile[mid] * invTd) h2 = mid; else l2 = mid + 1;
            }
            j = (int32_t)(Tc * TILE + l2);
        }
    }
    return j;
}
