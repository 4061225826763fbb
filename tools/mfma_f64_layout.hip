// Discovers the operand layout of v_mfma_f64_4x4x4_4b_f64: for a one-hot A at lane p and B[l] = l + 1, prints which
// output lanes receive which B lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* a, const double* b, const double* c, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
int main() {
    std::vector<double> a(64), b(64), c(64, 0.0), d(64);
    double *da, *db, *dc, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&dd, 512);
    for (int i = 0; i < 64; ++i) b[i] = i + 1;
    hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice);
    for (int p = 0; p < 64; ++p) {
        for (int i = 0; i < 64; ++i) a[i] = (i == p) ? 1.0 : 0.0;
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
        printf("A lane %2d ->", p);
        for (int l = 0; l < 64; ++l) if (d[l] != 0.0) printf(" D[%d]=B[%d]", l, (int)d[l] - 1);
        printf("\n");
    }
    return 0;
}
