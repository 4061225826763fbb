// kernels/smooth.hpp — FFBS smoother kernels.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// FFBS particle smoother, backward step t (reference src/smoothing.jl:128-141, draw_one_categorical
// src/resample.jl:128-152).  k_smooth_fx evaluates f(xf[n,t]) once; k_smooth_draw: one block per trajectory m,
//   wb[n] = wf[n,t] + logpdf(df, xb[m,t+1] - fx[n])  (recomputed in each of the three sweeps: max, total of the
//   quanta of exp(wb - max), count of bins below s = rand()*bins[end]), index = #{b : bins[b] < s}.
// ------------------------------------------------------------------------------------------------
#include "smooth_fx.hpp"

template <int NX>
__global__ __launch_bounds__(BLOCK) void k_smooth_draw(BankDev b, const ModelD* __restrict__ md, SmoothArgs a) {
    __shared__ double sm_d[BLOCK / 64];
    __shared__ uint64_t sm_u[BLOCK / 64];
    const int m = blockIdx.x;
    const int64_t N = b.N, Ns = b.Ns;
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
    double xq[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xq[d] = a.xb_next[(size_t)m * NX + d];
    auto wb = [&](int64_t n) {
        double v[NX];
#pragma unroll
        for (int d = 0; d < NX; ++d) v[d] = xq[d] - a.fx[(size_t)d * Ns + n];
        return a.wf_t[n] + gauss_logpdf<NX>(md->df, v);
    };
    // sweep 1: maximum
    double mx = -LLPF_INF;
    for (int64_t n = threadIdx.x; n < N; n += BLOCK) mx = llpf_fmax(mx, wb(n));
    mx = block_max(mx, sm_d);
    // sweep 2: total of the quanta
    const int K = llpf_qbits(N);
    uint64_t tot = 0;
    for (int64_t n = threadIdx.x; n < N; n += BLOCK) tot += llpf_q64_unit(llpf_exp_le0(wb(n) - mx), K);
    tot = wave_sum_u64(tot);
    __syncthreads();
    if (lane == 0) sm_u[wvid] = tot;
    __syncthreads();
    tot = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) tot += sm_u[k];
    // sweep 3: bins = fl(fl(cum) * fl(1/fl(total))) in index order; count those below s
    const FilterScal* sc = b.scal;
    const double u = llpf_uniform_idx((uint32_t)m, a.step, LLPF_STREAM_SMOOTH, sc->k0, sc->k1);
    const double Td = (double)tot, invTd = 1.0 / Td;
    const double s = u * (Td * invTd);
    uint64_t carry = 0, count = 0;
    for (int64_t base = 0; base < N; base += (int64_t)BLOCK * 4) {
        const int64_t n0 = base + (int64_t)threadIdx.x * 4;
        uint64_t c[4], run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t n = n0 + k;
            run += (n < N) ? llpf_q64_unit(llpf_exp_le0(wb(n < N ? n : N - 1) - mx), K) : 0;
            c[k] = run;
        }
        const uint64_t incl = wave_scan_u64(run);
        __syncthreads();
        if (lane == 63) sm_u[wvid] = incl;
        __syncthreads();
        uint64_t off = carry, all = 0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) {
            if (k < wvid) off += sm_u[k];
            all += sm_u[k];
        }
        const uint64_t excl = off + (incl - run);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (n0 + k < N && (double)(excl + c[k]) * invTd < s) ++count;
        carry += all;
        // bins are non-decreasing (fl and the multiplication by 1/total are monotone): once the running total is not below s
        // no later bin is, and the rest of the sweep would count nothing — on average half of it
        if (!((double)carry * invTd < s)) break;
    }
    count = wave_sum_u64(count);
    __syncthreads();
    if (lane == 0) sm_u[wvid] = count;
    __syncthreads();
    count = 0;
#pragma unroll
    for (int k = 0; k < BLOCK / 64; ++k) count += sm_u[k];
    const int64_t idx = (int64_t)count < N ? (int64_t)count : N - 1;     // nothing found: length(bins), resample.jl:151
    if (threadIdx.x == 0 && a.idx_t) a.idx_t[m] = idx;
    if (threadIdx.x < NX) a.xb_t[(size_t)m * NX + threadIdx.x] = a.xf_t[idx * NX + threadIdx.x];
}
