// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the instruction kinds that dominate the
// particle-filter kernels on gfx950: fp64 fma/mul/add, 32x32->64 mad, 32-bit logic, ds_bpermute, v_rsq_f64, div.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ void k(double* out, int iters) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.000001, c = 0.5, d = a + 1, e = a + 2, f = a + 3, g = a + 4, h = a + 5;
    uint64_t x = threadIdx.x + 12345, y = 0x9E3779B97F4A7C15ull, z = x ^ 77, w = x + 5;
    uint32_t p = threadIdx.x, q = 0xD2511F53u, r = p + 1, s2 = p + 2;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c); }
            if (KIND == 1) { x = (uint64_t)(uint32_t)x * q + y; z = (uint64_t)(uint32_t)z * q + y; w = (uint64_t)(uint32_t)w * q + y; y = (uint64_t)(uint32_t)y * q + x; }
            if (KIND == 2) { p = (p ^ q) + r; r = (r ^ p) + s2; s2 = (s2 ^ r) + q; q = (q ^ s2) + p; }
            if (KIND == 3) { p = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 1) << 2), (int)p); r = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 2) << 2), (int)r); s2 = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 4) << 2), (int)s2); q = __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 8) << 2), (int)q); }
            if (KIND == 4) { a = a / b; d = d / b; e = e / b; f = f / b; }
            if (KIND == 5) { a = __builtin_sqrt(a) + 1.0; d = __builtin_sqrt(d) + 1.0; e = __builtin_sqrt(e) + 1.0; f = __builtin_sqrt(f) + 1.0; }
            if (KIND == 6) { p = p * q + 1; r = r * q + 1; s2 = s2 * q + 1; q = q * 3 + 1; }
            if (KIND == 7) { a = a * b; d = d + c; e = e * b; f = f + c; }
            // round 5 (k_norm's fixed-point conversions): 64-bit shifts by a per-lane amount, 64-bit adds, selects, fp64 <-> u32 conversions, ldexp
            if (KIND == 8) { x = (x >> ((uint32_t)z & 63)) ^ w; z = (z << ((uint32_t)w & 63)) ^ y; w = (w >> ((uint32_t)y & 63)) ^ x; y = (y << ((uint32_t)x & 63)) ^ z; }
            if (KIND == 9) { x += z; z += w; w += y; y += x; }
            if (KIND == 10) { p = (p > q) ? r : s2; r = (r > s2) ? q : p; s2 = (s2 > p) ? r : q; q = (q > r) ? p : s2; }
            if (KIND == 11) { a = (double)(uint32_t)a + c; d = (double)(uint32_t)d + c; e = (double)(uint32_t)e + c; f = (double)(uint32_t)f + c; }
            if (KIND == 13) { a = __builtin_rint(a) + c; d = __builtin_rint(d) + c; e = __builtin_rint(e) + c; f = __builtin_rint(f) + c; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h + (double)(x + z + w + y) + (double)(p + r + s2 + q);
}
template <int KIND>
void run(const char* name, int per_iter_ops) {
    double* out; hipMalloc(&out, 1024 * 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // 1024 blocks x 256 threads = 4 waves per SIMD on 256 CUs
    hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves * iters * 16 * per_iter_ops wave-instructions
    double winst = 4.0 * iters * 16 * per_iter_ops;
    printf("%-34s %7.3f ms  -> %6.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / winst, ms * 1e6 / winst * 2.4);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f64", 4);
    run<7>("v_mul_f64 / v_add_f64", 4);
    run<1>("v_mad_u64_u32", 4);
    run<6>("v_mul_lo_u32 (+add)", 4);
    run<2>("xor+add u32 (2 ops each)", 8);
    run<3>("ds_bpermute_b32", 4);
    run<4>("fp64 divide", 4);
    run<5>("fp64 sqrt (+add)", 4);
    run<8>("v_lsh{l,r}rev_b64 by VGPR + 2 v_xor_b32 (as one)", 4);
    run<9>("64-bit add (add_co + addc)", 4);
    run<10>("v_cmp_gt_u32 + v_cndmask_b32", 8);
    run<11>("v_cvt_u32_f64 + v_cvt_f64_u32 + v_add_f64", 12);
    run<13>("v_rndne_f64 + v_add_f64", 8);
    return 0;
}
