// k_quantile.hip — weighted_quantile of the current particles on the device.
// Reference: weighted_quantile(x, we, q) (src/filtering.jl:583-595) = StatsBase.quantile(v, ProbabilityWeights(we), q) per state
// dimension; StatsBase is a dependency of the reference and not vendored: its published algorithm (src/weights.jl, `quantile(v, w, p)`,
// the non-frequency-weight branch) is restated in oracle/llpf_oracle.c: orc_weighted_quantile, which this file is held to:
//     drop zero weights; sort the pairs (v, w) lexicographically; wsum = sum(w); w1 = weight of the smallest value;
//     h = q (wsum - w1) + w1;  advance k while S_k <= h (S_k = w_1 + ... + w_k);  result v_{k-1} + (h - S_{k-1}) / (S_k - S_{k-1}) (v_k - v_{k-1});
//     past the end: the largest value; any NaN among v: NaN.
// Off the hot path (an accessor): the two sorts are rocPRIM's device radix sort (stable: by weight first, then by value = the lexicographic
// order), the running sums are an inclusive scan in 2^-96 fixed point (128-bit integers: the same S_k whatever the scan's block
// order, and exact to 2^-96 per term), the crossing is a binary search by one thread per quantile.
#include "engine.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <vector>

namespace llpf {

struct U128Plus {
    __host__ __device__ llpf_u128 operator()(const llpf_u128& a, const llpf_u128& b) const { return llpf_u128_add(a, b); }
};
constexpr uint64_t WQ_DROPPED = ~0ULL;      // key of a particle without weight: behind every value (a NaN among v is flagged separately)

// order-preserving map double -> uint64 (negative values reversed below the positive ones) and back
__device__ __forceinline__ uint64_t wq_key(double x) { const uint64_t u = llpf_d2u(x); return (u >> 63) ? ~u : (u | 0x8000000000000000ULL); }
__device__ __forceinline__ double wq_val(uint64_t k) { return llpf_u2d((k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k); }

// keys of the first sort = the weights' bit patterns (non-negative doubles order like their bits), payload = the value keys
__global__ __launch_bounds__(BLOCK) void k_wq_pairs(const double* __restrict__ x, const double* __restrict__ we, int64_t N,
                                                     uint64_t* __restrict__ kw, uint64_t* __restrict__ kx, int32_t* __restrict__ nan_flag) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= N) return;
    const double v = x[i], w = we[i];
    if (v != v) *nan_flag = 1;
    kw[i] = llpf_d2u(w);
    kx[i] = (w > 0.0) ? wq_key(v) : WQ_DROPPED;
}
__global__ __launch_bounds__(BLOCK) void k_wq_fix(const uint64_t* __restrict__ wbits, int64_t N, llpf_u128* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < N) out[i] = llpf_fix96(llpf_u2d(wbits[i]));
}
// one thread per quantile; kx sorted ascending (dropped particles last), wbits / S in that order
__global__ void k_wq_find(const uint64_t* __restrict__ kx, const uint64_t* __restrict__ wbits, const llpf_u128* __restrict__ S, int64_t N,
                          const int32_t* __restrict__ nan_flag, const double* __restrict__ q, int nq, double* __restrict__ out, int out_stride) {
    const int t = (int)threadIdx.x;
    if (t >= nq) return;
    const double qnan = llpf_u2d(0x7ff8000000000000ULL);
    int64_t lo = 0, hi = N;                                  // number of particles that carry weight
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (kx[mid] == WQ_DROPPED) hi = mid; else lo = mid + 1; }
    const int64_t n = lo;
    if (*nan_flag || n == 0) { out[(size_t)t * out_stride] = qnan; return; }
    const double wsum = llpf_fix96_to_double(S[n - 1]), w1 = llpf_u2d(wbits[0]);
    const double h = q[t] * (wsum - w1) + w1;
    lo = 0; hi = n;                                          // first k with S_k > h
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (llpf_fix96_to_double(S[mid]) > h) hi = mid; else lo = mid + 1; }
    const int64_t k = lo;
    double r;
    if (k >= n) r = wq_val(kx[n - 1]);
    else {
        const double Sk = llpf_fix96_to_double(S[k]), Skold = k ? llpf_fix96_to_double(S[k - 1]) : 0.0;
        const double vk = wq_val(kx[k]), vkold = k ? wq_val(kx[k - 1]) : 0.0;
        r = vkold + (h - Skold) / (Sk - Skold) * (vk - vkold);
    }
    out[(size_t)t * out_stride] = r;
}

// x: [nx][Ns] planes of ONE filter, we: [N] exp-weights (launch_materialize); q, out: device [nq], [nq][nx]
hipError_t launch_wquantile(const double* x, int64_t Ns, int nx, const double* we, int64_t N, const double* q, int nq, double* out, hipStream_t s) {
    if (N < 1 || nq < 1 || nq > 1024) return hipErrorInvalidValue;
    uint64_t *ka = nullptr, *kb = nullptr, *va = nullptr, *vb = nullptr;
    llpf_u128 *fa = nullptr, *fb = nullptr;
    int32_t* flag = nullptr;
    void* tmp = nullptr;
    size_t sort_bytes = 0, scan_bytes = 0;
    hipError_t e = hipSuccess;
    auto done = [&](hipError_t r) {
        hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(fa); hipFree(fb); hipFree(flag); hipFree(tmp);
        return r;
    };
#define WQ(call) do { e = (call); if (e != hipSuccess) return done(e); } while (0)
    WQ(hipMalloc(&ka, sizeof(uint64_t) * N)); WQ(hipMalloc(&kb, sizeof(uint64_t) * N));
    WQ(hipMalloc(&va, sizeof(uint64_t) * N)); WQ(hipMalloc(&vb, sizeof(uint64_t) * N));
    WQ(hipMalloc(&fa, sizeof(llpf_u128) * N)); WQ(hipMalloc(&fb, sizeof(llpf_u128) * N));
    WQ(hipMalloc(&flag, sizeof(int32_t)));
    WQ(rocprim::radix_sort_pairs(nullptr, sort_bytes, ka, kb, va, vb, (size_t)N, 0, 64, s));
    WQ(rocprim::inclusive_scan(nullptr, scan_bytes, fa, fb, (size_t)N, U128Plus(), s));
    WQ(hipMalloc(&tmp, sort_bytes > scan_bytes ? sort_bytes : scan_bytes));
    const dim3 g((unsigned)((N + BLOCK - 1) / BLOCK));
    for (int d = 0; d < nx; ++d) {
        WQ(hipMemsetAsync(flag, 0, sizeof(int32_t), s));
        hipLaunchKernelGGL(k_wq_pairs, g, dim3(BLOCK), 0, s, x + (size_t)d * Ns, we, N, ka, va, flag);
        WQ(rocprim::radix_sort_pairs(tmp, sort_bytes, ka, kb, va, vb, (size_t)N, 0, 64, s));      // by weight: kb = weights, vb = value keys
        WQ(rocprim::radix_sort_pairs(tmp, sort_bytes, vb, va, kb, ka, (size_t)N, 0, 64, s));      // stable, by value: va = value keys, ka = weights
        hipLaunchKernelGGL(k_wq_fix, g, dim3(BLOCK), 0, s, ka, N, fa);
        WQ(rocprim::inclusive_scan(tmp, scan_bytes, fa, fb, (size_t)N, U128Plus(), s));
        hipLaunchKernelGGL(k_wq_find, dim3(1), dim3(1024), 0, s, va, ka, fb, N, flag, q, nq, out + d, nx);
        WQ(hipGetLastError());
    }
    WQ(hipStreamSynchronize(s));
#undef WQ
    return done(hipSuccess);
}

}  // namespace llpf
