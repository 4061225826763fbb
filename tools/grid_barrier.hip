// Micro-benchmark: what does ONE grid-wide synchronisation cost on gfx950 when it is done inside a persistent kernel
// (sharded arrival counters + a generation flag every block spins on) instead of at a kernel boundary?  And does data
// written before the barrier by one block reach a block on another XCD after it (stale-read check), with which fences?
//   variant 0  barrier only (no fences, no data)
//   variant 1  barrier with __threadfence() on both sides (agent-scope release / acquire: L2 write-back + invalidate)
//   variant 2  variant 1 + every block writes 36 KB per step (ordinary stores) and reads 12 KB a block on another XCD wrote
//   variant 3  as 2, stores are write-through (__builtin_nontemporal_store) and loads bypass (nontemporal), NO __threadfence
//   variant 4  as 2 on fine-grained (coherent) device memory, no __threadfence
//   variant 5  as 2 with agent-scope relaxed atomic loads / stores for the data (sc1: coherent at the memory side), no __threadfence
//   variant 6  as 5, gather-like 8-B loads through a permutation within a 8 KB window; no s_sleep in the spin
// All spins are bounded: a barrier that does not complete sets an error flag and every block leaves (no hang).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#ifndef NSH_
#define NSH_ 8
#endif
constexpr int NSH = NSH_, STRIDE = 32;   // arrival shards, one 128-B line each

struct Bar {
    unsigned* arrive;   // [NSH * STRIDE]
    unsigned* top;      // [STRIDE]
    unsigned* gen;      // [STRIDE]
    unsigned* err;
    unsigned per_shard[NSH];
};

template <bool FENCE, bool SLEEP = true>
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned& g) {
    __builtin_amdgcn_s_waitcnt(0);      // every thread's stores acknowledged (gfx9: vmcnt counts stores too)
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (FENCE) __threadfence();
        const unsigned sh = blockIdx.x & (NSH - 1);
        const unsigned prev = __hip_atomic_fetch_add(b.arrive + sh * STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == (g + 1) * b.per_shard[sh]) {
            const unsigned p2 = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p2 + 1 == (g + 1) * NSH) __hip_atomic_store(b.gen, g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
            if (SLEEP) __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *b.err = 1; ok = false; break; }
        }
        if (FENCE) __threadfence();
    }
    g++;
    ok = __syncthreads_and(ok ? 1 : 0) != 0;
    return ok;
}

template <int VAR>
__global__ __launch_bounds__(256) void k_persist(Bar b, double* buf0, double* buf1, int steps, unsigned* stale) {
    unsigned g = 0;
    const int nb = gridDim.x;
    const int src_blk = (blockIdx.x + 1) % nb;             // consecutive blocks sit on different XCDs
    unsigned bad = 0;
    for (int s = 0; s < steps; ++s) {
        double* wr = (s & 1) ? buf1 : buf0;
        const double* rd = (s & 1) ? buf0 : buf1;
        if (VAR >= 2) {
            // read what block src_blk wrote in the previous step (value encodes step and slot), then write this step's
            if (s > 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const size_t i = ((size_t)src_blk * 18 + k) * 256 + threadIdx.x;
                    const size_t i6 = ((size_t)src_blk * 18 + k) * 256 + ((threadIdx.x * 7 + 3 * k) & 255);
                    const double v = (VAR == 3) ? __builtin_nontemporal_load(rd + i) : (VAR == 5 ? __hip_atomic_load(rd + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (VAR == 6 ? __hip_atomic_load(rd + i6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rd[i]));
                    if (v != (double)(s - 1) * 1024.0 + (double)k) bad++;
                }
            }
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                const size_t i = ((size_t)blockIdx.x * 18 + k) * 256 + threadIdx.x;
                const double v = (double)s * 1024.0 + (double)k;
                if (VAR == 3) __builtin_nontemporal_store(v, wr + i); else if (VAR >= 5) __hip_atomic_store(wr + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else wr[i] = v;
            }
        }
        if (!grid_barrier<(VAR == 1 || VAR == 2), (VAR != 6)>(b, g)) return;
    }
    if (bad) atomicAdd(stale, bad);
}

template <int VAR>
void run(const char* name, int nblocks, int steps, bool fine) {
    Bar b;
    unsigned* ctl;
    hipMalloc(&ctl, sizeof(unsigned) * (NSH * STRIDE + 4 * STRIDE));
    hipMemset(ctl, 0, sizeof(unsigned) * (NSH * STRIDE + 4 * STRIDE));
    b.arrive = ctl; b.top = ctl + NSH * STRIDE; b.gen = b.top + STRIDE; b.err = b.gen + STRIDE;
    unsigned* stale = b.err + STRIDE;
    for (int s = 0; s < NSH; ++s) b.per_shard[s] = (nblocks - s + NSH - 1) / NSH;
    const size_t nd = (size_t)nblocks * 18 * 256;
    double *b0 = nullptr, *b1 = nullptr;
    if (fine) { hipExtMallocWithFlags((void**)&b0, nd * 8, hipDeviceMallocFinegrained); hipExtMallocWithFlags((void**)&b1, nd * 8, hipDeviceMallocFinegrained); }
    else { hipMalloc(&b0, nd * 8); hipMalloc(&b1, nd * 8); }
    hipMemset(b0, 0, nd * 8); hipMemset(b1, 0, nd * 8);
    int maxb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_persist<VAR>, 256, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    if (maxb * p.multiProcessorCount < nblocks) { printf("%-60s not co-resident (%d x %d < %d)\n", name, maxb, p.multiProcessorCount, nblocks); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void* args[] = {&b, &b0, &b1, &steps, &stale};
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(ctl, 0, sizeof(unsigned) * (NSH * STRIDE + 4 * STRIDE));
        hipEventRecord(e0);
        hipError_t e = hipLaunchCooperativeKernel((const void*)k_persist<VAR>, dim3(nblocks), dim3(256), args, 0, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        if (e != hipSuccess) { printf("%-60s launch failed: %s\n", name, hipGetErrorString(e)); return; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned h[2 * STRIDE];
        hipMemcpy(h, b.err, sizeof(h), hipMemcpyDeviceToHost);
        if (rep == 1) printf("%-60s %8.2f us per step  (blocks %d, steps %d, barrier error %u, stale reads %u)\n", name, 1e3 * ms / steps, nblocks, steps, h[0], h[STRIDE]);
    }
    hipFree(ctl); hipFree(b0); hipFree(b1);
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 2000;
    for (int nb : {977, 256}) {
        run<0>("0 barrier only", nb, steps, false);
        run<1>("1 barrier + threadfence both sides", nb, steps, false);
        run<2>("2 + 36 KB stores / 12 KB cross-block loads per block, fences", nb, steps, false);
        run<3>("3 same traffic, nontemporal stores+loads, no fences", nb, steps, false);
        run<4>("4 same traffic on fine-grained memory, no fences", nb, steps, true);
        run<5>("5 same traffic, agent-scope atomic loads/stores, no fences", nb, steps, false);
        run<6>("6 as 5, permuted 8-B loads, spin without s_sleep", nb, steps, false);
    }
    return 0;
}
