// tools/stream_rw.hip — what a launch shaped like k_norm can reach: 128 x 98 blocks of 256 threads, every thread reads two 16-byte pairs
// of fp64 and writes two 16-byte pairs of u64 (8 B in, 8 B out per element, 12.8 M elements = 205 MB), with WORK dependent fp64
// multiply-adds per element in between (0: a pure copy; k_norm's arithmetic is worth ~50 per element).  Three buffer pairs in
// rotation (617 MB: more than the 256 MB Infinity Cache), so that every launch finds its input in HBM as k_norm does behind k_resprop.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_rw.hip -o tools/stream_rw && tools/stream_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int F = 128, TILES = 98, BS = 256, TILE = 1024;
template <int WORK, int TPB>
__global__ __launch_bounds__(BS) void k(const double* __restrict__ in, uint64_t* __restrict__ out, double m, unsigned long long* acc) {
    const size_t base = ((size_t)blockIdx.y * TILES + (size_t)blockIdx.x * TPB) * TILE;
    uint64_t s = 0;
    for (int t = 0; t < TPB; ++t) {
        double2 v[2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) v[k2] = *reinterpret_cast<const double2*>(in + base + (size_t)t * TILE + k2 * (BS * 2) + threadIdx.x * 2);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            double a = v[k2].x - m, c = v[k2].y - m;
#pragma unroll
            for (int j = 0; j < WORK; ++j) { a = __builtin_fma(a, 0.999, 1e-3); c = __builtin_fma(c, 0.999, 1e-3); }
            ulonglong2 q; q.x = (uint64_t)__double_as_longlong(a) >> 3; q.y = (uint64_t)__double_as_longlong(c) >> 3;
            *reinterpret_cast<ulonglong2*>(out + base + (size_t)t * TILE + k2 * (BS * 2) + threadIdx.x * 2) = q;
            s += q.x + q.y;
        }
    }
    // a block-level tail like k_norm's: wave reduction by shuffles, one atomic per block
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ uint64_t sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc + (blockIdx.x & 15) * 16, (unsigned long long)(sm[0] + sm[1] + sm[2] + sm[3]));
}
template <int WORK, int TPB>
int run(double** in, uint64_t** out, unsigned long long* acc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g(TILES / TPB, F);
    for (int r = 0; r < 6; ++r) hipLaunchKernelGGL((k<WORK, TPB>), g, dim3(BS), 0, 0, in[r % 3], out[r % 3], 0.5, acc);
    CK(hipDeviceSynchronize());
    const int reps = 60;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<WORK, TPB>), g, dim3(BS), 0, 0, in[r % 3], out[r % 3], 0.5, acc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)F * TILES * TILE * 16;
    printf("fma per element %3d, tiles per block %2d: %7.2f us per launch, %5.2f TB/s\n", WORK, TPB, us, bytes / us * 1e-6);
    return 0;
}
int main() {
    const size_t n = (size_t)F * TILES * TILE;
    double* in[3]; uint64_t* out[3]; unsigned long long* acc;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&in[i], n * 8)); CK(hipMalloc(&out[i], n * 8)); CK(hipMemset(in[i], 0, n * 8)); }
    CK(hipMalloc(&acc, 16 * 16 * 8)); CK(hipMemset(acc, 0, 16 * 16 * 8));
    if (run<0, 1>(in, out, acc) || run<16, 1>(in, out, acc) || run<32, 1>(in, out, acc) || run<48, 1>(in, out, acc) || run<64, 1>(in, out, acc) || run<96, 1>(in, out, acc)) return 1;
    if (run<0, 7>(in, out, acc) || run<48, 7>(in, out, acc) || run<0, 14>(in, out, acc) || run<48, 14>(in, out, acc)) return 1;
    return 0;
}
