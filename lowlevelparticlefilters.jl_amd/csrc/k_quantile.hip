// k_quantile.hip — weighted_quantile on the device: a radix selection over weight histograms (round 6; rounds 4-5: two rocPRIM sorts).
// Reference: weighted_quantile(x, we, q) (src/filtering.jl:583-595) = StatsBase.quantile(v, ProbabilityWeights(we), q) per state
// dimension.  StatsBase is a dependency of the reference and not vendored; its published algorithm (src/weights.jl `quantile(v, w, p)`,
// the non-frequency-weight branch) is restated in oracle/llpf_oracle.c: orc_weighted_quantile (reference order) and, in the form computed
// here, orc_weighted_quantile_dev (device order), which this file reproduces bit for bit:
//     particles with w > 0 are PRESENT; sorted as tuples (v, w); wsum = sum(w); w1 = weight of the smallest; h = q (wsum - w1) + w1;
//     k = first index whose running sum S_k exceeds h; result v_{k-1} + (h - S_{k-1}) / (S_k - S_{k-1}) (v_k - v_{k-1});
//     past the end: the largest value; a NaN among v: NaN.
// No sort is needed for that: the crossing VALUE is found by eight passes over the particles, one per byte of the order-preserving 64-bit
// key of v, each summing the weights of the particles that share the selected prefix into 256 bins (LDS histograms per block, flushed
// with integer atomics) and a one-thread walk over the bins; a last pass finds the predecessor value and the lightest weight of the
// crossing value's ties (tuples sort by weight inside a tie).  Sums are integers, m = min(floor(w 2^sc), 2^98), with the scale chosen
// PER QUANTILE so that h 2^sc has its leading bit at 2^64 or above (sc = 96 for h >= 2^-32): the crossing test is exact, a particle
// lighter than the scale still is present as smallest / predecessor / largest value (p -> 0 returns the smallest value, which the
// 2^-96-only sums of round 5 did not), and nothing depends on the order in which blocks run.
// All launches go to the caller's stream, allocate nothing and wait for nothing: the run loop calls this once per timestep for its
// `xquant` output (host/run.hpp) and captures it into its graph; llpf_weighted_quantile is the same code on the current state.
#include "engine.hpp"

namespace llpf {

#define DEV __device__ __forceinline__
constexpr int WQ_QCH = 4;                 // quantiles per round of passes (LDS: WQ_QCH x 256 bins x 3 limbs x 8 B = 24 KB)
constexpr int WQ_TILE = 4 * BLOCK;        // particles per block
constexpr uint64_t WQ_M43 = ((uint64_t)1 << 43) - 1;

struct WqDim {                            // per state dimension
    unsigned long long tot[3];            // limb sums of fix96(w) over the present particles
    unsigned long long kmin, kmax;        // smallest / largest key among them
    unsigned long long w1bits;            // lightest weight among the particles with the smallest key
    int32_t nan, pad;
};
struct WqQ {                              // per (dimension, quantile of the round)
    double h, p;
    int32_t sc, done;                     // done: the running sum never exceeds h — the largest value
    uint64_t Hlo, Hhi;                    // h 2^sc
    uint64_t prefix;                      // bytes of the crossing value's key selected so far
    uint64_t below_lo, below_hi;          // sum of the masses of all keys below the selected range
    unsigned long long wminbits, pred;    // lightest weight among key == prefix; largest present key < prefix (0: none)
    unsigned long long hist[256][3];
};

DEV uint64_t wq_key(double x) { const uint64_t u = llpf_d2u(x); return (u >> 63) ? ~u : (u | 0x8000000000000000ULL); }
DEV double wq_val(uint64_t k) { return llpf_u2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k); }
// min(floor(w 2^sc), 2^98) for w > 0 (0 for subnormals)
DEV llpf_u128 wq_mass(double w, int sc) {
    llpf_u128 r; r.lo = 0; r.hi = 0;
    const uint64_t u = llpf_d2u(w);
    const int E = (int)(u >> 52) & 0x7ff;
    if (E == 0 || E == 0x7ff || (u >> 63)) return r;
    const uint64_t M = (u & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
    const int sh = E - 1075 + sc;
    if (sh >= 0) {
        if (sh > 45) { r.hi = (uint64_t)1 << 34; return r; }
        r.lo = M << sh; r.hi = sh ? (M >> (64 - sh)) : 0;
        return r;
    }
    r.lo = (-sh >= 53) ? 0 : (M >> -sh);
    return r;
}
// a 2^-sc, rounded once (the scale reaches 2^-1138 for weights near the bottom of the double range: two exact power-of-two steps)
DEV double wq_to_double(llpf_u128 a, int sc) { return (llpf_u128_to_double(a) * llpf_pow2i(-(sc / 2))) * llpf_pow2i(-(sc - sc / 2)); }
DEV void wq_limbs(llpf_u128 v, uint64_t* l) { l[0] = v.lo & WQ_M43; l[1] = ((v.lo >> 43) | (v.hi << 21)) & WQ_M43; l[2] = v.hi >> 22; }
DEV llpf_u128 wq_unlimb(uint64_t a0, uint64_t a1, uint64_t a2) {
    llpf_u128 r = {a0, 0}, t;
    t.lo = a1 << 43; t.hi = a1 >> 21; r = llpf_u128_add(r, t);
    t.lo = 0; t.hi = a2 << 22; r = llpf_u128_add(r, t);
    return r;
}

__global__ void k_wq_init(WqDim* D, int nx) {
    const int d = (int)threadIdx.x;
    if (d >= nx) return;
    D[d].tot[0] = D[d].tot[1] = D[d].tot[2] = 0;
    D[d].kmin = ~0ULL; D[d].kmax = 0; D[d].w1bits = ~0ULL; D[d].nan = 0; D[d].pad = 0;
}

// pass over the particles: NaN flag, total of the masses at 2^-96, smallest and largest key of the present particles
__global__ __launch_bounds__(BLOCK) void k_wq_prep(const double* __restrict__ x, int64_t Ns, const double* __restrict__ we, int64_t N, WqDim* D) {
    __shared__ unsigned long long s_tot[3], s_min, s_max;
    __shared__ int s_nan;
    const int d = blockIdx.y, t = (int)threadIdx.x;
    if (t == 0) { s_tot[0] = s_tot[1] = s_tot[2] = 0; s_min = ~0ULL; s_max = 0; s_nan = 0; }
    __syncthreads();
    const double* xd = x + (size_t)d * Ns;
    llpf_u128 sum = {0, 0};
    uint64_t kmin = ~0ULL, kmax = 0;
    bool nan = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = (int64_t)blockIdx.x * WQ_TILE + j * BLOCK + t;
        if (i >= N) continue;
        const double v = xd[i], w = we[i];
        nan = nan || (v != v);
        if (w > 0.0) {
            sum = llpf_u128_add(sum, wq_mass(w, 96));
            const uint64_t k = wq_key(v);
            kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
        }
    }
    uint64_t l[3];
    wq_limbs(sum, l);      // four masses below 2^98 each: the sum is below 2^100, the limbs canonical
#pragma unroll
    for (int k = 0; k < 3; ++k) if (l[k]) atomicAdd(&s_tot[k], (unsigned long long)l[k]);
    if (kmin != ~0ULL) { atomicMin(&s_min, (unsigned long long)kmin); atomicMax(&s_max, (unsigned long long)kmax); }
    if (nan) s_nan = 1;
    __syncthreads();
    if (t == 0) {
        const llpf_u128 bt = wq_unlimb(s_tot[0], s_tot[1], s_tot[2]);      // canonical limbs again: 2^19 blocks at most add up below 2^62
        wq_limbs(bt, l);
#pragma unroll
        for (int k = 0; k < 3; ++k) if (l[k]) atomicAdd(&D[d].tot[k], (unsigned long long)l[k]);
        if (s_min != ~0ULL) { atomicMin(&D[d].kmin, s_min); atomicMax(&D[d].kmax, s_max); }
        if (s_nan) D[d].nan = 1;
    }
}
// the lightest weight among the particles with the smallest key: w1 of StatsBase's tuple order
__global__ __launch_bounds__(BLOCK) void k_wq_first(const double* __restrict__ x, int64_t Ns, const double* __restrict__ we, int64_t N, WqDim* D) {
    const int d = blockIdx.y, t = (int)threadIdx.x;
    const uint64_t kmin = D[d].kmin;
    const double* xd = x + (size_t)d * Ns;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = (int64_t)blockIdx.x * WQ_TILE + j * BLOCK + t;
        if (i >= N) continue;
        const double w = we[i];
        if (w > 0.0 && wq_key(xd[i]) == kmin) atomicMin(&D[d].w1bits, (unsigned long long)llpf_d2u(w));
    }
}
// per (dimension, quantile of this round): h, the scale, h 2^sc; selection state cleared
__global__ __launch_bounds__(BLOCK) void k_wq_setup(const WqDim* D, WqQ* Q, const double* __restrict__ p, int q0, int nq) {
    const int d = blockIdx.y, c = blockIdx.x, t = (int)threadIdx.x;
    WqQ& s = Q[d * WQ_QCH + c];
    s.hist[t][0] = s.hist[t][1] = s.hist[t][2] = 0;
    if (t != 0) return;
    const int q = q0 + c;
    s.prefix = 0; s.below_lo = s.below_hi = 0; s.wminbits = ~0ULL; s.pred = 0; s.done = 0; s.sc = 96; s.Hlo = s.Hhi = 0; s.h = 0.0; s.p = 0.0;
    if (q >= nq || D[d].kmin == ~0ULL) { s.done = 2; return; }        // no such quantile in this round / nothing carries weight
    const double w1 = llpf_u2d(D[d].w1bits);
    double wsum = wq_to_double(wq_unlimb(D[d].tot[0], D[d].tot[1], D[d].tot[2]), 96);
    if (wsum < w1) wsum = w1;
    const double dd = wsum - w1;
    const double pd = p[q] * dd;
    const double h = pd + w1;
    const uint64_t u = llpf_d2u(h);
    const int E = (int)(u >> 52) & 0x7ff;
    const int eh = E - 1023;
    const int sc = eh >= -32 ? 96 : 96 + (-32 - eh);
    const uint64_t M = (u & 0x000fffffffffffffULL) | 0x0010000000000000ULL;
    const int sh = E - 1075 + sc;                                      // >= 12
    s.h = h; s.p = p[q]; s.sc = sc;
    if (sh >= 64) { s.Hlo = 0; s.Hhi = M << (sh - 64); }
    else { s.Hlo = M << sh; s.Hhi = M >> (64 - sh); }
}
// one byte of the selection: weights of the particles whose key carries the selected prefix, by the next byte of the key
__global__ __launch_bounds__(BLOCK) void k_wq_hist(const double* __restrict__ x, int64_t Ns, const double* __restrict__ we, int64_t N, WqQ* Q, int pass) {
    __shared__ unsigned long long sh[WQ_QCH][256][3];
    const int d = blockIdx.y, t = (int)threadIdx.x;
    const int shift = 56 - 8 * pass;
    int sc[WQ_QCH]; uint64_t pre[WQ_QCH]; bool act[WQ_QCH];
    bool any = false;
#pragma unroll
    for (int c = 0; c < WQ_QCH; ++c) {
        const WqQ& s = Q[d * WQ_QCH + c];
        sc[c] = s.sc; pre[c] = s.prefix; act[c] = s.done == 0; any = any || act[c];
        sh[c][t][0] = sh[c][t][1] = sh[c][t][2] = 0;
    }
    if (!any) return;                                                  // block-uniform
    __syncthreads();
    const double* xd = x + (size_t)d * Ns;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = (int64_t)blockIdx.x * WQ_TILE + j * BLOCK + t;
        if (i >= N) continue;
        const double w = we[i];
        if (!(w > 0.0)) continue;
        const uint64_t k = wq_key(xd[i]);
        const int dig = (int)((k >> shift) & 255u);
#pragma unroll
        for (int c = 0; c < WQ_QCH; ++c) {
            if (!act[c]) continue;
            if (pass && ((k ^ pre[c]) >> (shift + 8))) continue;       // not under the selected prefix
            uint64_t l[3];
            wq_limbs(wq_mass(w, sc[c]), l);
#pragma unroll
            for (int m = 0; m < 3; ++m) if (l[m]) atomicAdd(&sh[c][dig][m], (unsigned long long)l[m]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < WQ_QCH; ++c) {
        if (!act[c]) continue;
        const unsigned long long a0 = sh[c][t][0], a1 = sh[c][t][1], a2 = sh[c][t][2];
        if (!(a0 | a1 | a2)) continue;
        uint64_t l[3];
        wq_limbs(wq_unlimb(a0, a1, a2), l);                            // 1024 masses below 2^98: below 2^108, canonical limbs
        WqQ& s = Q[d * WQ_QCH + c];
#pragma unroll
        for (int m = 0; m < 3; ++m) if (l[m]) atomicAdd(&s.hist[t][m], (unsigned long long)l[m]);
    }
}
// ... and the walk over the 256 bins: the first bin in which the running sum exceeds h 2^sc
__global__ __launch_bounds__(BLOCK) void k_wq_pick(WqQ* Q, int pass) {
    __shared__ unsigned long long b[256][3];
    const int d = blockIdx.y, c = blockIdx.x, t = (int)threadIdx.x;
    WqQ& s = Q[d * WQ_QCH + c];
    if (s.done) return;
#pragma unroll
    for (int m = 0; m < 3; ++m) { b[t][m] = s.hist[t][m]; s.hist[t][m] = 0; }
    __syncthreads();
    if (t != 0) return;
    const llpf_u128 H = {s.Hlo, s.Hhi};
    llpf_u128 cum = {s.below_lo, s.below_hi};
    const int shift = 56 - 8 * pass;
    for (int dgt = 0; dgt < 256; ++dgt) {
        const llpf_u128 nxt = llpf_u128_add(cum, wq_unlimb(b[dgt][0], b[dgt][1], b[dgt][2]));
        if (llpf_u128_lt(H, nxt)) {
            s.prefix |= (uint64_t)dgt << shift;
            s.below_lo = cum.lo; s.below_hi = cum.hi;
            return;
        }
        cum = nxt;
    }
    s.done = 1;        // (first pass only: below a crossing bin there always is a crossing bin) the total does not exceed h
}
// the crossing value's ties and its predecessor
__global__ __launch_bounds__(BLOCK) void k_wq_finish(const double* __restrict__ x, int64_t Ns, const double* __restrict__ we, int64_t N, WqQ* Q) {
    __shared__ unsigned long long s_wmin[WQ_QCH], s_pred[WQ_QCH];
    const int d = blockIdx.y, t = (int)threadIdx.x;
    uint64_t pre[WQ_QCH]; bool act[WQ_QCH];
    bool any = false;
#pragma unroll
    for (int c = 0; c < WQ_QCH; ++c) { const WqQ& s = Q[d * WQ_QCH + c]; pre[c] = s.prefix; act[c] = s.done == 0; any = any || act[c]; }
    if (!any) return;
    if (t < WQ_QCH) { s_wmin[t] = ~0ULL; s_pred[t] = 0; }
    __syncthreads();
    const double* xd = x + (size_t)d * Ns;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = (int64_t)blockIdx.x * WQ_TILE + j * BLOCK + t;
        if (i >= N) continue;
        const double w = we[i];
        if (!(w > 0.0)) continue;
        const uint64_t k = wq_key(xd[i]);
#pragma unroll
        for (int c = 0; c < WQ_QCH; ++c) {
            if (!act[c]) continue;
            if (k == pre[c]) atomicMin(&s_wmin[c], (unsigned long long)llpf_d2u(w));
            else if (k < pre[c]) atomicMax(&s_pred[c], (unsigned long long)k);
        }
    }
    __syncthreads();
    if (t < WQ_QCH && act[t]) {
        WqQ& s = Q[d * WQ_QCH + t];
        if (s_wmin[t] != ~0ULL) atomicMin(&s.wminbits, s_wmin[t]);
        if (s_pred[t]) atomicMax(&s.pred, s_pred[t]);
    }
}
__global__ void k_wq_result(const WqDim* D, const WqQ* Q, int nx, int q0, int nq, double* __restrict__ out, int stride_q, int stride_d) {
    const int d = (int)threadIdx.x / WQ_QCH, c = (int)threadIdx.x % WQ_QCH, q = q0 + c;
    if (d >= nx || q >= nq) return;
    const WqQ& s = Q[d * WQ_QCH + c];
    double r;
    if (D[d].nan || s.done == 2) r = llpf_u2d(0x7ff8000000000000ULL);
    else if (s.done == 1) r = wq_val(D[d].kmax);
    else {
        const double vk = wq_val(s.prefix), wmin = llpf_u2d(s.wminbits);
        const llpf_u128 H = {s.Hlo, s.Hhi}, Slt = {s.below_lo, s.below_hi};
        if (llpf_u128_lt(H, llpf_u128_add(Slt, wq_mass(wmin, s.sc)))) {      // the crossing is the lightest of the ties: interpolate from the predecessor
            const double vkold = s.pred ? wq_val(s.pred) : 0.0;
            const double Skold = s.pred ? wq_to_double(Slt, s.sc) : 0.0;
            const double Sk = Skold + wmin;
            const double den = Sk - Skold;
            r = den > 0.0 ? vkold + (s.h - Skold) / den * (vk - vkold) : vk;
        } else r = vk;                                                     // inside the tie: v_{k-1} == v_k
    }
    out[(size_t)q * stride_q + (size_t)d * stride_d] = r;
}

size_t wquantile_workspace_bytes(int nx) { return sizeof(WqDim) * (size_t)nx + sizeof(WqQ) * (size_t)nx * WQ_QCH; }

// x: [nx][Ns] planes of ONE filter, we: [N] exp-weights; q (device, [nq]); out (device): out[q * stride_q + d * stride_d]; ws: wquantile_workspace_bytes(nx)
hipError_t launch_wquantile(const double* x, int64_t Ns, int nx, const double* we, int64_t N, const double* q, int nq, double* out, int stride_q, int stride_d,
                            void* ws, hipStream_t s) {
    if (N < 1 || nq < 1 || nx < 1 || nx > MAXD || !ws) return hipErrorInvalidValue;
    WqDim* D = reinterpret_cast<WqDim*>(ws);
    WqQ* Q = reinterpret_cast<WqQ*>(reinterpret_cast<char*>(ws) + sizeof(WqDim) * (size_t)nx);
    const dim3 g((unsigned)((N + WQ_TILE - 1) / WQ_TILE), (unsigned)nx, 1), gq(WQ_QCH, (unsigned)nx, 1);
    hipLaunchKernelGGL(k_wq_init, dim3(1), dim3(64), 0, s, D, nx);
    hipLaunchKernelGGL(k_wq_prep, g, dim3(BLOCK), 0, s, x, Ns, we, N, D);
    hipLaunchKernelGGL(k_wq_first, g, dim3(BLOCK), 0, s, x, Ns, we, N, D);
    for (int q0 = 0; q0 < nq; q0 += WQ_QCH) {
        hipLaunchKernelGGL(k_wq_setup, gq, dim3(BLOCK), 0, s, D, Q, q, q0, nq);
        for (int pass = 0; pass < 8; ++pass) {
            hipLaunchKernelGGL(k_wq_hist, g, dim3(BLOCK), 0, s, x, Ns, we, N, Q, pass);
            hipLaunchKernelGGL(k_wq_pick, gq, dim3(BLOCK), 0, s, Q, pass);
        }
        hipLaunchKernelGGL(k_wq_finish, g, dim3(BLOCK), 0, s, x, Ns, we, N, Q);
        hipLaunchKernelGGL(k_wq_result, dim3(1), dim3(MAXD * WQ_QCH), 0, s, D, Q, nx, q0, nq, out, stride_q, stride_d);
    }
    return hipGetLastError();
}

}  // namespace llpf
