// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layout and the rounding sequence of its k-sum, to decide whether
// the per-particle 8x8 Kalman contractions of the C5 kernel (csrc/shared/llpf_rbfull_body.h) can move to the matrix unit
// and stay bit-identical to the oracle's fma chains.  Prints which (layout, summation order) hypothesis matches bitwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

__global__ void k(const double* a, const double* b, const double* c, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

int main() {
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::vector<double> a(64), b(64), c(64), d(64);
    int ok_total = 0;
    const char* names[4] = {"fma chain k=0..3 onto c", "fma chain k=3..0 onto c", "products rounded, then added k=0..3 onto c", "pairwise (k0+k1)+(k2+k3) + c"};
    int match[8][4] = {{0}};
    const int trials = 200;
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
    for (int t = 0; t < trials; ++t) {
        for (int i = 0; i < 64; ++i) { a[i] = U(rng) * ldexp(1.0, (int)(rng() % 20) - 10); b[i] = U(rng); c[i] = U(rng) * 1e-3; }
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        // layout hypotheses: bit0: A element (i,k) at lane 16*blk + (k*4 + i) [0] or (i*4 + k) [1]; bit1: same for B (k,j): (k*4 + j) [0] / (j*4 + k) [1];
        // bit2: D element (i,j) at (j*4 + i)?? [0] or (i*4 + j) [1]
        for (int h = 0; h < 1; ++h) {
            int good[4] = {1, 1, 1, 1};
            for (int blk = 0; blk < 4; ++blk)
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double av[4], bv[4];
                        for (int kk = 0; kk < 4; ++kk) {
                            av[kk] = a[16 * kk + 4 * blk + i];      // layout found with tools/mfma_f64_layout.hip
                            bv[kk] = b[16 * kk + 4 * blk + j];
                        }
                        const int dl = 16 * i + 4 * blk + j;
                        const double cc = c[dl], got = d[dl];
                        double r0 = cc; for (int kk = 0; kk < 4; ++kk) r0 = fma(av[kk], bv[kk], r0);
                        double r1 = cc; for (int kk = 3; kk >= 0; --kk) r1 = fma(av[kk], bv[kk], r1);
                        double r2 = cc; for (int kk = 0; kk < 4; ++kk) { volatile double p = av[kk] * bv[kk]; r2 = r2 + p; }
                        volatile double p0 = fma(av[1], bv[1], av[0] * bv[0]), p1 = fma(av[3], bv[3], av[2] * bv[2]);
                        double r3 = (p0 + p1) + cc;
                        if (bits(got) != bits(r0)) good[0] = 0;
                        if (bits(got) != bits(r1)) good[1] = 0;
                        if (bits(got) != bits(r2)) good[2] = 0;
                        if (bits(got) != bits(r3)) good[3] = 0;
                    }
            for (int q = 0; q < 4; ++q) match[h][q] += good[q];
        }
    }
    for (int h = 0; h < 8; ++h)
        for (int q = 0; q < 4; ++q)
            if (match[h][q] > trials / 2) { printf("layout A[i][k] <-> lane 16k+4blk+i, B[k][j] <-> lane 16k+4blk+j, D[i][j] <-> lane 16i+4blk+j: %s — bitwise match in %d/%d trials\n", names[q], match[h][q], trials); ok_total++; }
    if (!ok_total) {
        printf("no hypothesis matched bitwise; best counts:\n");
        for (int h = 0; h < 8; ++h) printf("  layout %d: %d %d %d %d\n", h, match[h][0], match[h][1], match[h][2], match[h][3]);
    }
    return 0;
}
