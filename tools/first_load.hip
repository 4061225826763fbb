// tools/first_load.hip — how long after a workgroup's first instruction do its first loads return, at the start of a launch that
// depends on the previous one?  (The head of the fused kernel spends 2.2 us there, EXPERIMENTS.md Appendix A.)
//   hipcc --offload-arch=gfx950 -O3 tools/first_load.hip -o tools/first_load && tools/first_load
// Producer kernel writes a small "hot" array (64 words, read by every block of the consumer, like the accumulator words) and a
// large array (8 KB per block, like the tile's quanta), with plain stores or write-through (sc1) stores; the consumer (977 x 256,
// like C2) stamps wall_clock64 at its first instruction and when (a) the hot load, (b) the large load has returned.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NB = 977, BS = 256;

template <bool WT> __global__ void producer(uint64_t* hot, uint64_t* big, uint64_t v) {
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    for (int k = 0; k < 4; ++k) {
        uint64_t* p = big + i * 4 + k;
        if (WT) __hip_atomic_store(p, v + i + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v + i + k;
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) atomicAdd((unsigned long long*)(hot + threadIdx.x * 16), 1ull);
}
template <int ORDER> __global__ void consumer(const uint64_t* hot, const uint64_t* big, uint64_t* stamps, uint64_t* sink) {
    const uint64_t t0 = wall_clock64();
    const size_t i = (size_t)blockIdx.x * BS + threadIdx.x;
    uint64_t a, b0;
    uint64_t ta, tb;
    if (ORDER == 0) {        // hot first, then the block's 8 KB
        a = hot[(threadIdx.x & 63) * 16];
        const ulonglong2 q0 = *reinterpret_cast<const ulonglong2*>(big + i * 4), q1 = *reinterpret_cast<const ulonglong2*>(big + i * 4 + 2);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        ta = wall_clock64();
        b0 = q0.x + q0.y + q1.x + q1.y;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tb = wall_clock64();
    } else {                 // the 8 KB first
        const ulonglong2 q0 = *reinterpret_cast<const ulonglong2*>(big + i * 4), q1 = *reinterpret_cast<const ulonglong2*>(big + i * 4 + 2);
        a = hot[(threadIdx.x & 63) * 16];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ta = tb = wall_clock64();
        b0 = q0.x + q0.y + q1.x + q1.y;
    }
    if (threadIdx.x == 0) { stamps[blockIdx.x * 4] = t0; stamps[blockIdx.x * 4 + 1] = ta; stamps[blockIdx.x * 4 + 2] = tb; }
    if (a + b0 == 0x1234567) sink[0] = 1;
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
    uint64_t *hot, *big, *st, *sink;
    CK(hipMalloc(&hot, 64 * 16 * 8)); CK(hipMalloc(&big, (size_t)NB * BS * 4 * 8)); CK(hipMalloc(&st, NB * 4 * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(hot, 0, 64 * 16 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<uint64_t> h(NB * 4);
    for (int wt = 0; wt < 2; ++wt) for (int order = 0; order < 2; ++order) {
        std::vector<double> la, lb, span;
        for (int rep = 0; rep < 40; ++rep) {
            if (wt) hipLaunchKernelGGL(producer<true>, dim3(NB), dim3(BS), 0, s, hot, big, (uint64_t)rep);
            else hipLaunchKernelGGL(producer<false>, dim3(NB), dim3(BS), 0, s, hot, big, (uint64_t)rep);
            if (order) hipLaunchKernelGGL(consumer<1>, dim3(NB), dim3(BS), 0, s, hot, big, st, sink);
            else hipLaunchKernelGGL(consumer<0>, dim3(NB), dim3(BS), 0, s, hot, big, st, sink);
            CK(hipMemcpyAsync(h.data(), st, NB * 4 * 8, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            if (rep < 5) continue;
            std::vector<double> a, b; uint64_t tmin = ~0ull, tmax = 0;
            for (int k = 0; k < NB; ++k) { a.push_back((h[k * 4 + 1] - h[k * 4]) / 100.0); b.push_back((h[k * 4 + 2] - h[k * 4]) / 100.0); tmin = std::min(tmin, h[k * 4]); tmax = std::max(tmax, h[k * 4 + 2]); }
            la.push_back(med(a)); lb.push_back(med(b)); span.push_back((tmax - tmin) / 100.0);
        }
        printf("producer stores %-13s consumer issues %-22s: first instruction -> hot words back %.2f us, -> 8 KB back %.2f us (medians over blocks and 35 launches); first block start -> last load back %.2f us\n",
               wt ? "write-through" : "plain", order ? "the 8 KB first" : "the hot words first", med(la), med(lb), med(span));
    }
    return 0;
}
