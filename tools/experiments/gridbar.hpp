// kernels/gridbar.hpp — a barrier over the workgroups of ONE launch (all of them resident).  Namespace llpf, device code.
// ------------------------------------------------------------------------------------------------
// tools/grid_barrier.hip measured it on the MI355X: 2.3 us for 977 workgroups with 32 arrival shards (3.6 us with 8); what a kernel
// boundary costs, but without the boundary's cold first loads and without a second dispatch.  The eight XCDs' L2s are not coherent with
// one another inside a launch: whatever one workgroup writes before the barrier and another reads after it goes through agent-scope
// accesses (global_load / global_store ... sc1: Mem<1> of kernels/resample.hpp, wt_store of kernels/reduce.hpp), and every thread waits
// for its own stores before its workgroup arrives (s_waitcnt vmcnt(0): gfx9 counts stores there).  Every spin is bounded: a barrier that
// does not complete returns false and the caller leaves with an error status instead of hanging the device.
// Counters are monotonic and survive across launches: arrive[sh] reaches (g+1) * blocks_in_shard, top reaches (g+1) * shards_in_use,
// then gen = g+1.  A launch reads the generation it starts from with grid_barrier_generation().  All launches that share one counter
// block must have the same number of workgroups (the host keeps one block per launch shape: Bank::d_bar).
// ------------------------------------------------------------------------------------------------
constexpr int BAR_NSHARD = 32;                       // arrival shards, one 128-B line each
constexpr int BAR_STRIDE = 32;                       // u32 per line
constexpr int BAR_WORDS = (BAR_NSHARD + 2) * BAR_STRIDE;   // arrive[32], top, gen
enum { LLPF_STATUS_BARRIER_TIMEOUT = 90 };           // internal: reported by the host as LLPF_ERR_HIP

DEV uint32_t grid_barrier_generation(const uint32_t* bar) { return *(bar + (BAR_NSHARD + 1) * BAR_STRIDE); }

DEV bool grid_barrier(uint32_t* bar, uint32_t& g, int nblocks) {
    __builtin_amdgcn_s_waitcnt(0);                   // this thread's stores and atomics are acknowledged
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const uint32_t sh = blockIdx.x & (BAR_NSHARD - 1);
        const uint32_t in_shard = ((uint32_t)nblocks - sh + BAR_NSHARD - 1) / BAR_NSHARD;
        const uint32_t shards = nblocks < BAR_NSHARD ? (uint32_t)nblocks : (uint32_t)BAR_NSHARD;
        uint32_t* top = bar + BAR_NSHARD * BAR_STRIDE;
        uint32_t* gen = top + BAR_STRIDE;
        const uint32_t prev = __hip_atomic_fetch_add(bar + sh * BAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == (g + 1u) * in_shard) {
            const uint32_t p2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p2 + 1u == (g + 1u) * shards) __hip_atomic_store(gen, g + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t spins = 0;
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g) {
            if (++spins > (1u << 22)) { ok = false; break; }      // ~seconds: another tenant holds CUs, or a block died
        }
    }
    g++;
    return __syncthreads_and(ok ? 1 : 0) != 0;
}
