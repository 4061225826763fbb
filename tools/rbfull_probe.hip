// tools/rbfull_probe.hip — where does a wave of the per-particle Kalman recursion (csrc/shared/llpf_rbfull_body.h, the
// arithmetic of k_rbfull, BASELINE config C5) spend its cycles?  Developer aid, not part of the product.
//
// A stripped kernel (gather 48 planes -> time update -> measurement update -> store; no RK4, no generator) with s_memtime
// stamps at the stage boundaries of the body (RBF_STAMP), launched with one, two, ... waves per SIMD on synthetic data.
// Prints the median cycles a wave spends in each phase, next to the number of VALU instructions of that phase: a phase at
// ~4.x cycles per instruction is issue-bound, anything above is exposed latency.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Ilowlevelparticlefilters.jl_amd/csrc -Iinclude tools/rbfull_probe.hip -o tools/rbfull_probe
//   tools/rbfull_probe [waves_per_simd=1] [reps=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NSTAMP 16
#ifndef PROBE_WAVES
#define PROBE_WAVES 2      /* register budget: 2 = the product kernel's (256 VGPRs), 1 = 512 (no spills whatever the stamps cost) */
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RBF_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); g_dbg[(size_t)blockIdx.x * NSTAMP + (k)] = t_; } while (0)   /* every lane, same address: no branch in the body */
#else
#define RBF_STAMP(k) ((void)0)
#endif
__device__ unsigned long long* g_dbg;
#include "engine.hpp"

namespace llpf {
template <int NN, int NL, int NY, bool MEMONLY>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PROBE_WAVES))) void k_probe(const double* __restrict__ xc, double* __restrict__ xo,
                                                                                      const llpf_rbf_par* __restrict__ par, int64_t Ns,
                                                                                      const double* __restrict__ u, const double* __restrict__ yy) {
    constexpr int NP = LLPF_RBF_NP(NL);
    RBF_STAMP(0);
    __shared__ double sh_blu[LLPF_RBF_MAXL];
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        const double v = llpf_rbf_blu_row((llpf_rbf_cptr)par, 2, r, u);
        if (threadIdx.x == 0) sh_blu[r] = v;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    const uint32_t stride = (uint32_t)Ns * 8u, io = i * 8u;
    auto ld = [&](int row) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xc) + (io + (uint32_t)row * stride)); };
    auto st = [&](int row, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(xo) + (io + (uint32_t)row * stride)) = v; };
    double xn[NN], xl[NL], R[NP];
#pragma unroll
    for (int d = 0; d < NN; ++d) xn[d] = ld(d);
#pragma unroll
    for (int d = 0; d < NL; ++d) xl[d] = ld(NN + d);
#pragma unroll
    for (int d = 0; d < NP; ++d) R[d] = ld(NN + NL + d);
    double fi[NN], nz[NN], xn1[NN], xl1[NL], R1[NP];
    for (int d = 0; d < NN; ++d) { fi[d] = xn[d] * 0.5; nz[d] = xn[d] * 0.25; }
    double ll = 0.0;
    if (MEMONLY) {          // the gather and the stores alone: the floor this access pattern sets
        for (int d = 0; d < NN; ++d) xn1[d] = xn[d] + fi[d];
        for (int d = 0; d < NL; ++d) xl1[d] = xl[d] + 1.0;
        for (int d = 0; d < NP; ++d) R1[d] = R[d] + 1.0;
    } else {
    llpf_rbf_predict(par, NN, NL, 2, xn, xl, R, u, sh_blu, fi, nz, xn1, xl1, R1);
    double y[NY], yn[NY];
    for (int k = 0; k < NY; ++k) { y[k] = yy[k]; yn[k] = xn1[k]; }
    ll = llpf_rbf_correct(par, NL, NY, y, yn, xl1, R1);
    }
    RBF_STAMP(11);
    xn1[0] += 1e-30 * ll;
#pragma unroll
    for (int d = 0; d < NN; ++d) st(d, xn1[d]);
#pragma unroll
    for (int d = 0; d < NL; ++d) st(NN + d, xl1[d]);
#pragma unroll
    for (int d = 0; d < NP; ++d) st(NN + NL + d, R1[d]);
    RBF_STAMP(12);
}
}  // namespace llpf

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const double wps = argc > 1 ? atof(argv[1]) : 1.0;
    const int reps = argc > 2 ? atoi(argv[2]) : 20, memonly = argc > 3 ? atoi(argv[3]) : 0;
    const int lds_bytes = argc > 4 ? atoi(argv[4]) : 0;       // dynamic LDS per (one-wave) workgroup: 20000 caps the occupancy at 2 waves per SIMD whatever the registers
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int waves = (int)(prop.multiProcessorCount * 4 * wps);
    const int64_t Ns = (int64_t)waves * 64;
    constexpr int NN = 4, NL = 8, NY = 2, NP = 36, ROWS = NN + NL + NP;
    llpf_rbf_par hp;
    memset(&hp, 0, sizeof(hp));
    hp.nn = NN; hp.nl = NL; hp.ny = NY; hp.nu = 2;
    srand(1);
    auto rnd = [] { return (double)rand() / RAND_MAX - 0.5; };
    for (int r = 0; r < NL; ++r) for (int c = 0; c < NL; ++c) hp.Al[r * NL + c] = (r == c ? 0.9 : 0.0) + 0.05 * rnd();
    for (int i = 0; i < NL * 2; ++i) hp.Bl[i] = 0.1 * rnd();
    for (int i = 0; i < NY * NL; ++i) hp.Cl[i] = rnd();
    for (int k = 0; k <= NN; ++k) for (int i = 0; i < NN * NL; ++i) hp.An[k][i] = 0.2 * rnd();
    for (int r = 0; r < NL; ++r) hp.R1l[llpf_rbf_idx(r, r)] = 0.1;
    for (int r = 0; r < NN; ++r) hp.R1n[r * NN + r] = 0.1;
    for (int r = 0; r < NY; ++r) hp.R2[r * NY + r] = 0.1;
    hp.c0y = -1.8378770664093453;
    std::vector<double> hx((size_t)ROWS * Ns);
    for (int64_t i = 0; i < Ns; ++i) {
        for (int d = 0; d < NN + NL; ++d) hx[(size_t)d * Ns + i] = rnd();
        for (int r = 0; r < NL; ++r) for (int c = 0; c <= r; ++c) hx[(size_t)(NN + NL + llpf_rbf_idx(r, c)) * Ns + i] = (r == c) ? 1.0 + 0.1 * rnd() : 0.02 * rnd();
    }
    double *dx, *dxo, *du, *dy;
    llpf_rbf_par* dp;
    unsigned long long* dd;
    CK(hipMalloc(&dx, sizeof(double) * hx.size()));
    CK(hipMalloc(&dxo, sizeof(double) * hx.size()));
    CK(hipMalloc(&dp, sizeof(hp)));
    CK(hipMalloc(&du, 64));
    CK(hipMalloc(&dy, 64));
    CK(hipMalloc(&dd, sizeof(unsigned long long) * NSTAMP * waves));
    CK(hipMemcpy(dx, hx.data(), sizeof(double) * hx.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, &hp, sizeof(hp), hipMemcpyHostToDevice));
    const double hu[2] = {0.3, -0.2}, hy[2] = {0.1, 0.2};
    CK(hipMemcpy(du, hu, 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hy, 16, hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &dd, sizeof(dd)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        if (memonly) hipLaunchKernelGGL((llpf::k_probe<NN, NL, NY, true>), dim3(waves), dim3(64), lds_bytes, 0, dx, dxo, dp, Ns, du, dy);
        else hipLaunchKernelGGL((llpf::k_probe<NN, NL, NY, false>), dim3(waves), dim3(64), lds_bytes, 0, dx, dxo, dp, Ns, du, dy);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> hd((size_t)NSTAMP * waves);
    CK(hipMemcpy(hd.data(), dd, sizeof(unsigned long long) * hd.size(), hipMemcpyDeviceToHost));
    static const char* names[13] = {"", "prologue + gather issued", "coupling rows: An, An R, Nt", "Cholesky, V, x~l, R~ = R - V V'", "Al x~l + Bl u",
                                    "upper panel + upper-left block", "lower-left block", "lower panel", "lower-right block + R1l", "(predict -> correct)",
                                    "C R, S, Cholesky, log", "gain, mean, covariance update", "stores issued"};
    printf("%d waves (%.2f per SIMD)%s, kernel %.1f us (best of %d, events), %.2f TB/s of gather + store\n", waves, wps, memonly ? " MEMORY ONLY" : "", best * 1000.0, reps, 2.0 * 48 * 8 * 64 * waves / (best * 1e-3) / 1e12);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < waves; ++w) { tmin = std::min(tmin, hd[(size_t)w * NSTAMP]); tmax = std::max(tmax, hd[(size_t)w * NSTAMP + 12]); }
    printf("first wave start -> last wave end: %llu cycles\n", tmax - tmin);
    for (int k = 1; k <= 12; ++k) {
        std::vector<unsigned long long> d(waves);
        for (int w = 0; w < waves; ++w) d[w] = hd[(size_t)w * NSTAMP + k] - hd[(size_t)w * NSTAMP + k - 1];
        std::sort(d.begin(), d.end());
        printf("  %-36s median %6llu  p10 %6llu  p90 %6llu cycles\n", names[k], d[waves / 2], d[waves / 10], d[waves * 9 / 10]);
    }
    std::vector<unsigned long long> tot(waves);
    for (int w = 0; w < waves; ++w) tot[w] = hd[(size_t)w * NSTAMP + 12] - hd[(size_t)w * NSTAMP];
    std::sort(tot.begin(), tot.end());
    printf("  %-36s median %6llu  p10 %6llu  p90 %6llu cycles\n", "whole wave", tot[waves / 2], tot[waves / 10], tot[waves * 9 / 10]);
    return 0;
}
