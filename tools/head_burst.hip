// tools/head_burst.hip — does the head's burst of tile-sum loads cost time?  Every block of a resample / fused launch (977 x 256 at C2)
// reads ALL 977 per-tile quanta sums (four loads per thread: 61 cache lines that every block asks for at the same moment) to form its own
// exclusive prefix and the total.  EXPERIMENTS 4.12 declined the two-level form (32 group sums + <= 31 tile sums per block) by argument;
// this measures it: a producer launch writes the table, the consumer stamps wall_clock64 at its first instruction and when everything it
// asked for is back, in three forms —  A all 977 sums (the product), B two-level (64 loads per block), C none (lower bound) —
// each next to the 64 accumulator words and the block's own 8 KB of quanta, as in the real head.
//   hipcc --offload-arch=gfx950 -O3 tools/head_burst.hip -o tools/head_burst && tools/head_burst
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NB = 977, BS = 256, NG = (NB + 31) / 32;

__global__ void producer(uint64_t* hot, uint64_t* big, uint64_t* tq, uint64_t* gq, uint64_t v) {
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    for (int k = 0; k < 4; ++k) __hip_atomic_store(big + i * 4 + k, v + i + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
        tq[blockIdx.x] = v + blockIdx.x;
        atomicAdd((unsigned long long*)(gq + (blockIdx.x >> 5) * 16), (unsigned long long)(v + blockIdx.x));
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) atomicAdd((unsigned long long*)(hot + threadIdx.x * 16), 1ull);
}
template <int FORM> __global__ void consumer(const uint64_t* hot, const uint64_t* big, const uint64_t* tq, const uint64_t* gq, uint64_t* stamps, uint64_t* sink) {
    const uint64_t t0 = wall_clock64();
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    const int t = threadIdx.x, tile = blockIdx.x;
    uint64_t a = 0, s = 0;
    if (t < 64) a = hot[t * 16];
    if (FORM == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int p = t + j * BS; const uint64_t q = tq[p < NB ? p : 0]; s += (p < tile) ? q : 0; }
    } else if (FORM == 1) {
        const int gs = (tile >> 5) << 5;
        if (t < NG) { const uint64_t q = gq[t * 16]; s += (t < (tile >> 5)) ? q : 0; }
        else if (t >= 32 && t < 64 && gs + (t - 32) < NB) { const uint64_t q = tq[gs + (t - 32)]; s += ((t - 32) < (tile & 31)) ? q : 0; }
    }
    const ulonglong2 q0 = *reinterpret_cast<const ulonglong2*>(big + i * 4), q1 = *reinterpret_cast<const ulonglong2*>(big + i * 4 + 2);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    const uint64_t ta = wall_clock64();                 // accumulator words and tile sums back
    const uint64_t b0 = q0.x + q0.y + q1.x + q1.y;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t tb = wall_clock64();                 // the tile's quanta back
    if (t == 0) { stamps[blockIdx.x * 4] = t0; stamps[blockIdx.x * 4 + 1] = ta; stamps[blockIdx.x * 4 + 2] = tb; }
    if (a + b0 + s == 0x1234567) sink[0] = 1;
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
    uint64_t *hot, *big, *st, *sink, *tq, *gq;
    CK(hipMalloc(&hot, 64 * 16 * 8)); CK(hipMalloc(&big, (size_t)NB * BS * 4 * 8)); CK(hipMalloc(&st, NB * 4 * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMalloc(&tq, 1024 * 8)); CK(hipMalloc(&gq, 32 * 16 * 8));
    CK(hipMemset(hot, 0, 64 * 16 * 8)); CK(hipMemset(gq, 0, 32 * 16 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<uint64_t> h(NB * 4);
    const char* names[3] = {"A all 977 tile sums (product)", "B two-level: 31 group + <= 31 tile sums", "C no tile sums (lower bound)"};
    for (int round = 0; round < 2; ++round) for (int form = 0; form < 3; ++form) {
        std::vector<double> la, lb, span, p90;
        for (int rep = 0; rep < 60; ++rep) {
            hipLaunchKernelGGL(producer, dim3(NB), dim3(BS), 0, s, hot, big, tq, gq, (uint64_t)rep);
            if (form == 0) hipLaunchKernelGGL(consumer<0>, dim3(NB), dim3(BS), 0, s, hot, big, tq, gq, st, sink);
            else if (form == 1) hipLaunchKernelGGL(consumer<1>, dim3(NB), dim3(BS), 0, s, hot, big, tq, gq, st, sink);
            else hipLaunchKernelGGL(consumer<2>, dim3(NB), dim3(BS), 0, s, hot, big, tq, gq, st, sink);
            CK(hipMemcpyAsync(h.data(), st, NB * 4 * 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (rep < 10) continue;
            std::vector<double> a, b; uint64_t tmin = ~0ull, tmax = 0;
            for (int k = 0; k < NB; ++k) { a.push_back((h[k * 4 + 1] - h[k * 4]) / 100.0); b.push_back((h[k * 4 + 2] - h[k * 4]) / 100.0); tmin = std::min(tmin, h[k * 4]); tmax = std::max(tmax, h[k * 4 + 2]); }
            std::sort(b.begin(), b.end());
            la.push_back(med(a)); lb.push_back(b[b.size() / 2]); p90.push_back(b[b.size() * 9 / 10]); span.push_back((tmax - tmin) / 100.0);
        }
        printf("%-42s first instruction -> sums back %.2f us, -> quanta back median %.2f p90 %.2f us; first block start -> last load back %.2f us\n",
               names[form], med(la), med(lb), med(p90), med(span));
    }
    return 0;
}
