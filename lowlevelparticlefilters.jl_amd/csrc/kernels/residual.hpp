// kernels/residual.hpp — residual resampling kernels.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// Residual resampling — resample(::Type{ResampleResidual}, we, j, bins, M), reference src/resample.jl:63-117.
// Device order (oracle/llpf_oracle.c:resample_residual): with the integer quanta q_i (total Q) the copy counts
// c_i = floor(q_i M / Q) and the residuals q_i M - c_i Q are exact; residuals are kept to K bits (rho_i = rem >> L,
// L = ceil(log2 N)) so that their cumulative sum fits 63 bits.  Outputs [0, num) are the deterministic copies in
// source order; outputs m in [num, M) draw u_m and take the first i with u_m < fl(fl(cumrho_i) fl(1/fl(totrho))).
//   k_resid_prep    head (finalize / decision) + per-tile totals of c and rho + within-tile cumulative rho (scratch)
//   k_resid_scan    inclusive prefixes of the per-tile totals (one block per filter)
//   k_resid_expand  deterministic copies via the counts machinery, multinomial part by two-level binary search
// ------------------------------------------------------------------------------------------------
DEV void resid_quanta(const BankDev& b, const ResArgs& a, const ResHead& h, const ulonglong2* qv, int64_t ib, uint64_t* q) {
    const uint64_t Qc = h.uniform ? llpf_q64_unit(1.0 / (double)b.N, a.K) : 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        uint64_t v = h.uniform ? Qc : ((k & 1) ? qv[k / 2].y : qv[k / 2].x);
        if (ib + k >= b.N) v = 0;
        q[k] = v;
    }
}

template <int SRC>
__global__ __launch_bounds__(BLOCK) void k_resid_prep(BankDev b, ResArgs a) {
    __shared__ ResShared sh;
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !b.scal[f].fallback : (b.scal[f].fallback != 0))) return;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    ulonglong2 qv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
    const ResHead h = res_head<SRC>(b, a, f, tile, sh);
    if (h.status) return;
    if (!a.force && !h.dr) return;
    if (h.tot == 0) return;
    if (tile == 0 && threadIdx.x == 0) b.scal[f].totQ = h.tot;
    const int L = 62 - a.K;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    uint64_t q[NORM_IPT], rr[NORM_IPT];
    resid_quanta(b, a, h, qv, ib, q);
    uint64_t csum = 0, run = 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        uint64_t rem;
        csum += llpf_muldiv_floor(q[k], (uint64_t)a.M, h.tot, &rem);
        run += rem >> L;
        rr[k] = run;
    }
    const uint64_t incl = wave_scan_u64(run);
    const uint64_t ctot = wave_sum_u64(csum);
    __syncthreads();
    if (lane == 63) sh.red[wvid][3] = incl;
    if (lane == 0) sh.red[wvid][2] = ctot;
    __syncthreads();
    uint64_t wave_off = 0, rtot = 0, call = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) {
        if (k < wvid) wave_off += sh.red[k][3];
        rtot += sh.red[k][3];
        call += sh.red[k][2];
    }
    const uint64_t excl = wave_off + (incl - run);
    uint64_t* cum = b.quanta_next + (size_t)f * b.Ns;      // scratch: free between the scan and the next weighting
    ulonglong2 o0, o1;
    o0.x = excl + rr[0]; o0.y = excl + rr[1]; o1.x = excl + rr[2]; o1.y = excl + rr[3];
    *reinterpret_cast<ulonglong2*>(cum + ib) = o0;
    *reinterpret_cast<ulonglong2*>(cum + ib + 2) = o1;
    if (threadIdx.x == 0) {
        uint64_t* rt = b.rtile + (size_t)f * 2 * b.P2;
        rt[tile] = call;
        rt[b.P2 + tile] = rtot;
    }
}
static_assert(NORM_IPT == 4, "k_resid_prep stores four cumulative residuals per thread");

template <int SRC>
__global__ __launch_bounds__(BLOCK) void k_resid_scan(BankDev b, ResArgs a) {
    __shared__ uint64_t sm[BLOCK / 64][2];
    const int f = blockIdx.x;
    const FilterScal* sc = b.scal + f;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !sc->fallback : (sc->fallback != 0))) return;
    if (SRC == SRC_FILTER && (sc->status || (!a.force && !sc->do_resample))) return;
    uint64_t* rt = b.rtile + (size_t)f * 2 * b.P2;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    uint64_t carry_c = 0, carry_r = 0;
    for (int base = 0; base < b.P2; base += BLOCK) {
        const int p = base + threadIdx.x;
        const uint64_t c = p < b.P2 ? rt[p] : 0, r = p < b.P2 ? rt[b.P2 + p] : 0;
        const uint64_t ic = wave_scan_u64(c), ir = wave_scan_u64(r);
        __syncthreads();
        if (lane == 63) { sm[wvid][0] = ic; sm[wvid][1] = ir; }
        __syncthreads();
        uint64_t oc = carry_c, orr = carry_r, tc = 0, tr = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) {
            if (k < wvid) { oc += sm[k][0]; orr += sm[k][1]; }
            tc += sm[k][0]; tr += sm[k][1];
        }
        if (p < b.P2) { rt[p] = oc + ic; rt[b.P2 + p] = orr + ir; }
        carry_c += tc; carry_r += tr;
    }
}

template <int SRC>
__global__ __launch_bounds__(BLOCK) void k_resid_expand(BankDev b, ResArgs a) {
    __shared__ ResShared sh;
    const int f = blockIdx.y;
    const int tile = blockIdx.x;
    const FilterScal* sc = b.scal + f;
    if (SRC == SRC_FILTER && run_is_stopped(b, a.k)) return;
    if (SRC == SRC_FILTER && (a.only_fallback ? !sc->fallback : (sc->fallback != 0))) return;
    if (SRC == SRC_FILTER && (sc->status || (!a.force && !sc->do_resample))) return;
    const uint64_t Q = sc->totQ;
    if (Q == 0) return;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    const int64_t ib = (int64_t)tile * TILE + (int64_t)threadIdx.x * NORM_IPT;
    const uint64_t* __restrict__ qsrc = b.quanta + (size_t)f * b.Ns;
    const uint64_t* __restrict__ rt = b.rtile + (size_t)f * 2 * b.P2;
    const uint64_t* __restrict__ cum = b.quanta_next + (size_t)f * b.Ns;
    int32_t* ao = a.anc_out + (size_t)f * b.Ns;
    const int64_t M = a.M;
    // ---- deterministic copies: cl[k] = copies of sources 0 .. k (all tiles before this one included)
    ulonglong2 qv[NORM_IPT / 2];
#pragma unroll
    for (int k = 0; k < NORM_IPT / 2; ++k) qv[k] = *reinterpret_cast<const ulonglong2*>(qsrc + ib + 2 * k);
    ResHead hq;
    hq.uniform = (SRC == SRC_FILTER) ? sc->uniform : 0;
    uint64_t q[NORM_IPT], cc[NORM_IPT];
    resid_quanta(b, a, hq, qv, ib, q);
    uint64_t crun = 0;
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        uint64_t rem;
        crun += llpf_muldiv_floor(q[k], (uint64_t)M, Q, &rem);
        cc[k] = crun;
    }
    const uint64_t incl = wave_scan_u64(crun);
    if (lane == 63) sh.red[wvid][3] = incl;
    __syncthreads();
    uint64_t wave_off = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k)
        if (k < wvid) wave_off += sh.red[k][3];
    const uint64_t c_prev = tile > 0 ? rt[tile - 1] : 0;
    const uint64_t excl = c_prev + wave_off + (incl - crun);
#pragma unroll
    for (int k = 0; k < NORM_IPT; ++k) {
        const uint64_t v = excl + cc[k];
        sh.cl[threadIdx.x * NORM_IPT + k] = (uint32_t)(v > (uint64_t)M ? (uint64_t)M : v);
    }
    __syncthreads();
    const int64_t c_start = (int64_t)(c_prev > (uint64_t)M ? (uint64_t)M : c_prev);
    const int64_t c_end = (int64_t)sh.cl[TILE - 1];
    for (int64_t o = c_start + threadIdx.x; o < c_end; o += BLOCK)
        ao[o] = (int32_t)((int64_t)tile * TILE + res_owner(sh.cl, (int32_t)o));
    // ---- multinomial part: outputs [num, M), an equal share per tile
    const uint64_t num64 = rt[b.P2 - 1];
    const int64_t num = (int64_t)(num64 > (uint64_t)M ? (uint64_t)M : num64);
    const int64_t R = M - num;
    if (R <= 0) return;
    const uint64_t totr = rt[2 * b.P2 - 1];
    const int anc_ident = (SRC == SRC_FILTER) ? sc->anc_ident_s[b.anc_slot] : 0;
    const int64_t chunk = (R + b.P2 - 1) / b.P2;
    const int64_t m0 = num + (int64_t)tile * chunk;
    const int64_t m1 = (m0 + chunk < M) ? m0 + chunk : M;
    const double Td = (double)totr;
    const double invTd = 1.0 / Td;
    const uint64_t* __restrict__ pr = rt + b.P2;
    for (int64_t m = m0 + threadIdx.x; m < m1; m += BLOCK) {
        const double u = a.Uexp ? a.Uexp[m] : llpf_uniform_idx((uint32_t)m, sc->step_base + a.step, LLPF_STREAM_STRATIFY, sc->k0, sc->k1);
        int64_t src = -1;
        if (totr != 0) {
            int lo = 0, hi = b.P2;                      // first tile t with u < bins(end of t)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (u < (double)pr[mid] * invTd) hi = mid; else lo = mid + 1;
            }
            if (lo < b.P2) {
                const uint64_t before = lo > 0 ? pr[lo - 1] : 0;
                const uint64_t* ct = cum + (size_t)lo * TILE;
                int kl = 0, kh = TILE - 1;              // the tile's last bin is > u, so an index exists
                while (kl < kh) {
                    const int mid = (kl + kh) >> 1;
                    if (u < (double)(before + ct[mid]) * invTd) kh = mid; else kl = mid + 1;
                }
                src = (int64_t)lo * TILE + kl;
            }
        }
        if (src >= 0) ao[m] = (int32_t)src;
        else if (anc_ident) ao[m] = (int32_t)m;        // u >= bins[N]: j[m] keeps its previous value (identity materialised)
    }
}
