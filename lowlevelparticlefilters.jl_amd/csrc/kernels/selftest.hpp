// kernels/selftest.hpp — device self-tests of the shared primitives.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// self-tests of the shared primitives on the device
// ------------------------------------------------------------------------------------------------
__global__ void k_selftest_math(int which, const double* __restrict__ in, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    double s, c, r;
    switch (which) {
        case 0: r = llpf_exp(x); break;
        case 1: r = llpf_log(x); break;
        case 2: r = llpf_log1p_nonneg(x); break;
        case 3: llpf_sincos2pi(x, &s, &c); r = s; break;
        case 4: llpf_sincos2pi(x, &s, &c); r = c; break;
        case 5: r = llpf_sqrt(x); break;
        case 12: r = llpf_sqrt_pos(x); break;
        case 6: r = 1.0 / x; break;
        case 7: r = (double)llpf_d2u(x); break;
        case 8: r = llpf_exp_le0(x); break;
        case 9: r = llpf_log_unit(x); break;
        case 10: llpf_sincos2pi_fast(x, &s, &c); r = s; break;
        case 11: llpf_sincos2pi_fast(x, &s, &c); r = c; break;
        // the wave primitives written in inline assembly (kernels/reduce.hpp); n must be a multiple of 64: each group of 64
        // consecutive elements is one wave, the operand is the bit pattern of x
        case 13: r = (double)(uint32_t)wave_scan_u64(llpf_d2u(x)); break;
        case 14: r = (double)(uint32_t)(wave_scan_u64(llpf_d2u(x)) >> 32); break;
        case 15: case 16: case 17: case 18: {
            const uint64_t u = llpf_d2u(x);
            llpf_u128 v; v.lo = u; v.hi = (u << 29) | (u >> 35);
            const llpf_u128 t = wave_sum_u128(v);
            const uint32_t w[4] = {(uint32_t)t.lo, (uint32_t)(t.lo >> 32), (uint32_t)t.hi, (uint32_t)(t.hi >> 32)};
            r = (double)w[which - 15];
            break;
        }
        default: r = 0.0;
    }
    out[i] = r;
}
__global__ void k_selftest_normals(uint32_t k0, uint32_t k1, uint32_t step, uint32_t stream, int nd, double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xi[MAXD];
    for (int bq = 0; 2 * bq < nd; ++bq) {
        double z0, z1;
        llpf_normal_pair((uint32_t)i, step, (uint32_t)bq, stream, k0, k1, &z0, &z1);
        xi[2 * bq] = z0;
        if (2 * bq + 1 < nd) xi[2 * bq + 1] = z1;
    }
    for (int d = 0; d < nd; ++d) out[i * nd + d] = xi[d];
}
