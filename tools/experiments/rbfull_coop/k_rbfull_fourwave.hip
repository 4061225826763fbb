// k_rbfull.hip — the Rao-Blackwellized filter with per-particle covariance (kernels/rbfull.hpp) and its launchers
// One of the engine's device translation units (kernels.hip has the map); split so that they build in parallel.

#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
// developer build (tools/dbg/rbf_timing.py): cycle stamps of every wave at the stage boundaries of the recursion
#define RBF_STAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); g_rbf_dbg[(size_t)g_rbf_row * 32 + (k)] = t_; } while (0)
#endif
#if defined(LLPF_RBF_TIMING)
__device__ unsigned long long* g_rbf_dbg;
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ unsigned int g_rbf_rows[4];  // the batch a persistent wave has in hand: the row its stamps go to (one entry per wave of the workgroup)
#define g_rbf_row g_rbf_rows[threadIdx.x >> 6]
// the shared tail batches (kernels/rbfull.hpp): stamps of the nonlinear wave, which passes every barrier as soon as the Kalman waves reach it
#define RBF_TAILSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); g_rbf_dbg[(size_t)(nfull + tj) * 32 + (k)] = t_; } while (0)
#endif
#endif
#include <atomic>

#include "engine.hpp"

namespace llpf {

#define DEV __device__ __forceinline__

#include "kernels/reduce.hpp"
#include "kernels/models.hpp"
#include "kernels/accum.hpp"
#include "kernels/rbfull.hpp"

static inline dim3 grid1(int64_t n, int F) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK), (unsigned)F, 1); }

// LLPF_MODEL_RB_BILINEAR: the instantiated shapes (nxn, nxl, ny); fn_kind 1 = quad-tank nonlinear part
// precompiled: (1,2,1), (2,2,2), (4,8,2) and the quad-tank's (4,8,2); every other shape within the header's limits is compiled on demand
static bool rbfull_precompiled(int fn_kind, int nn, int nl, int ny) {
    if (fn_kind == 1) return nn == 4 && nl == 8 && ny == 2;
    return (nn == 1 && nl == 2 && ny == 1) || (nn == 2 && nl == 2 && ny == 2) || (nn == 4 && nl == 8 && ny == 2);
}
bool rbfull_supported(int fn_kind, int nn, int nl, int ny) {
    if (nn < 1 || nn > LLPF_RBF_MAXN || nl < 1 || nl > LLPF_RBF_MAXL || ny < 1 || ny > LLPF_RBF_MAXY) return false;
    if (fn_kind == 1) return nn == 4 && ny == 2;
    return fn_kind == 0;
}
int rbfull_rows(int nn, int nl) { return nn + nl + LLPF_RBF_NP(nl); }

// Launch shape.  The 8x8 form is persistent (kernels/rbfull.hpp): as many waves as the device holds at once — two per SIMD, eight per
// CU — shared among the bank's filters, in workgroups of four; every wave takes batches w, w + waves, ...  `tail` (StepArgs::rbf_tail):
// when the batches do not divide evenly over the SIMDs and the remainder is small, the last batches are shared by the four waves
// of a workgroup (shared/llpf_rbfull_coop.h) instead of being a whole extra batch for a few SIMDs.  LLPF_RBF_TAIL: 0 = never, k > 0 =
// exactly k batches (tests: small filters), unset = the rule below.  Every other form: one wave per batch.
RbfullShape rbfull_launch_shape(const BankDev& b, int nl, int mode) {
    RbfullShape sh;
    const unsigned nbatch = (unsigned)(b.Ns / RBF_BLOCK);
    sh.block = (unsigned)(RBF_BLOCK * rbf_wpg(nl, mode)); sh.tail = 0; sh.grid_x = nbatch;
    if (!rbf_dma(nl, mode)) return sh;
    static std::atomic<int> cus[64];     // per ordinal; filled on first use (distinct handles may be driven from distinct host threads)
    int dev = 0, nc = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        nc = cus[dev].load(std::memory_order_relaxed);
        if (nc == 0) {
            int n = 0;
            nc = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : -1;
            cus[dev].store(nc, std::memory_order_relaxed);
        }
    }
    const unsigned nreal = (unsigned)((b.N + RBF_BLOCK - 1) / RBF_BLOCK);
    unsigned per_filter = nreal;          // waves of one filter
    if (nc > 0) {
        const unsigned resident = (unsigned)nc * 4u * (unsigned)LLPF_RBF_WAVES;
        per_filter = resident / (unsigned)(b.F > 0 ? b.F : 1);
        per_filter -= per_filter % 4u;
        if (per_filter < 4u) per_filter = 4u;
    }
    const char* tail_env = getenv("LLPF_RBF_TAIL");       // (read per launch: the tests switch it inside one process)
    unsigned tail = 0;
    const unsigned simds = per_filter / (unsigned)LLPF_RBF_WAVES;      // SIMDs this filter's waves occupy
    if (tail_env) tail = (unsigned)atoi(tail_env);
    else if (nc > 0 && nreal > simds && nreal % simds != 0 && nreal % simds <= simds / 8u) tail = nreal % simds;
    unsigned waves = nreal - tail < per_filter ? nreal - tail : per_filter;
    unsigned groups = (waves + 3u) / 4u;
    if (tail >= nreal || tail > groups) { tail = 0; waves = nreal < per_filter ? nreal : per_filter; groups = (waves + 3u) / 4u; }      // at least one ordinary batch; one shared batch per workgroup at most
    sh.grid_x = groups; sh.tail = tail;
    return sh;
}
// (kept for callers that only need the number of workgroups)
unsigned rbfull_grid_x(const BankDev& b, int nl, int mode) { return rbfull_launch_shape(b, nl, mode).grid_x; }

template <class Model, int NN, int NL, int NY>
static hipError_t launch_rbfull_t(const BankDev& b, int mode, const StepArgs& a0, hipStream_t s) {
    const RbfullShape sh = rbfull_launch_shape(b, NL, mode);
    dim3 g(sh.grid_x, (unsigned)b.F, 1);
    StepArgs a = a0;
    a.rbf_tail = (int32_t)sh.tail;
    switch (mode) {
        case MODE_WEIGHT: hipLaunchKernelGGL((k_rbfull<Model, NN, NL, NY, MODE_WEIGHT>), g, dim3(sh.block), 0, s, LLPF_RBF_HOT_ARGS(b, a), b, a); break;
        case MODE_PROP: hipLaunchKernelGGL((k_rbfull<Model, NN, NL, NY, MODE_PROP>), g, dim3(sh.block), 0, s, LLPF_RBF_HOT_ARGS(b, a), b, a); break;
        case MODE_PROP_WEIGHT: hipLaunchKernelGGL((k_rbfull<Model, NN, NL, NY, MODE_PROP_WEIGHT>), g, dim3(sh.block), 0, s, LLPF_RBF_HOT_ARGS(b, a), b, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// BankDev::pad0 carries the shape of this model: nxl | fn_kind << 8
hipError_t launch_rbfull(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    const int nl = b.pad0 & 0xff, fk = (b.pad0 >> 8) & 0xff;
    if (fk == 1 && b.nx == 4 && nl == 8 && b.ny == 2) return launch_rbfull_t<QuadTank<4, 2>, 4, 8, 2>(b, mode, a, s);
    if (fk == 0 && b.nx == 4 && nl == 8 && b.ny == 2) return launch_rbfull_t<LinGauss<4, 2>, 4, 8, 2>(b, mode, a, s);
    if (fk == 0 && b.nx == 2 && nl == 2 && b.ny == 2) return launch_rbfull_t<LinGauss<2, 2>, 2, 2, 2>(b, mode, a, s);
    if (fk == 0 && b.nx == 1 && nl == 2 && b.ny == 1) return launch_rbfull_t<LinGauss<1, 1>, 1, 2, 1>(b, mode, a, s);
    if (!rbfull_supported(fk, b.nx, nl, b.ny) || rbfull_precompiled(fk, b.nx, nl, b.ny)) return hipErrorInvalidValue;
    return launch_rbfull_jit(fk, b.nx, nl, b.ny, b, mode, a, s);        // compiled on demand (kernels/jit.hpp), cached per shape
}
hipError_t launch_rbfull_init(const BankDev& b, hipStream_t s) {
    const int nl = b.pad0 & 0xff;
    dim3 g = grid1(b.Ns, b.F);
    if (nl < 1 || nl > LLPF_RBF_MAXL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_rbfull_init, g, dim3(BLOCK), 0, s, b, b.models);
    return hipGetLastError();
}


}  // namespace llpf

#if defined(LLPF_RBF_TIMING)
static unsigned long long* g_rbf_dbg_dev = nullptr;
static int64_t g_rbf_dbg_waves = 0;
extern "C" __attribute__((visibility("default"))) int llpf_debug_rbf_timing_arm(int64_t waves) {
    if (g_rbf_dbg_dev) hipFree(g_rbf_dbg_dev);
    g_rbf_dbg_waves = waves;
    if (hipMalloc(&g_rbf_dbg_dev, sizeof(unsigned long long) * 32 * (size_t)waves) != hipSuccess) return -1;
    hipMemset(g_rbf_dbg_dev, 0, sizeof(unsigned long long) * 32 * (size_t)waves);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rbf_dbg), &g_rbf_dbg_dev, sizeof(g_rbf_dbg_dev)) == hipSuccess ? 0 : -2;
}
extern "C" __attribute__((visibility("default"))) int llpf_debug_rbf_timing_read(unsigned long long* dst) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpy(dst, g_rbf_dbg_dev, sizeof(unsigned long long) * 32 * (size_t)g_rbf_dbg_waves, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
