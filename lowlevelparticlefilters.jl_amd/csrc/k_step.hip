// k_step.hip — k_step (balanced propagate + weight: every model, every mode), k_max, and the run-time compiled user models
// One of the engine's device translation units (kernels.hip has the map); split so that they build in parallel.
#include <hip/hiprtc.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "engine.hpp"
#include "jit_prelude.inc"
#if defined(LLPF_STEP_TIMING)
__device__ unsigned long long* g_step_dbg;      // developer build: phase stamps of k_step / k_resample_fx (tools/dbg/qt_phases.py)
__device__ unsigned long long* g_fx_dbg;
#endif

namespace llpf {

#define DEV __device__ __forceinline__

#include "kernels/reduce.hpp"
#include "kernels/models.hpp"
#include "kernels/accum.hpp"
#include "kernels/step.hpp"
#include "kernels/resample.hpp"
#include "kernels/resfx.hpp"
#include "kernels/jit.hpp"

bool step_supported(int model_id, int nx, int ny) {
    if (model_id >= LLPF_MODEL_USER_BASE) return jit_supported(model_id, nx, ny);
    if (model_id == LLPF_MODEL_RB_BILINEAR) return nx >= 1 && nx <= LLPF_RBF_MAXN && ny >= 1 && ny <= LLPF_RBF_MAXY;   // shape checked by rbfull_supported
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return nx == 4 && ny == 2;
    if (model_id == LLPF_MODEL_RB_LINEAR) return nx >= 2 && nx <= 4 && ny >= 1 && ny <= 4;
    if (model_id == LLPF_MODEL_LINEAR_GAUSSIAN) return nx >= 1 && nx <= MAXD && ny >= 1 && ny <= MAXD;   // above 4: compiled on demand (jit_builtin_lg)
    return false;
}

#ifndef LLPF_QT_PPT
#define LLPF_QT_PPT 2
#endif
// compute units of the current device (cached per ordinal)
static int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}
// Blocks along x of a k_step<..., MARKS = true> launch: the resident set, four workgroups per CU, shared among the filters of a bank —
// the timestep of these models is bound by the blocks' latency chains (prologue, marks, gather, reductions: 60 % of a block's life at
// BASELINE C3), so every block walks its tiles and pays the prologue and the reductions once (kernels/step.hpp)
static unsigned step_grid_x_marks(const BankDev& b) {
    const int64_t tiles = b.Ns / STEP_TILE;
    static const char* env = getenv("LLPF_STEP_PERSIST");      // 0: one block per tile
    if (!(env && atoi(env) == 0)) {
        const int64_t per_filter = ((int64_t)device_cus() * 4) / (b.F > 0 ? b.F : 1);
        if (per_filter >= 1 && per_filter < tiles) {
            const int64_t rounds = (tiles + per_filter - 1) / per_filter;      // every block the same number of tiles (C3: 1954 tiles, 977 blocks of two)
            return (unsigned)((tiles + rounds - 1) / rounds);
        }
    }
    return (unsigned)tiles;
}
template <class Model, int NX, int NY, int PPT = STEP_PPT>
static hipError_t launch_step_t(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    if constexpr (marks_path<Model>::value && PPT == STEP_PPT) {
        if (a.marks && (mode == MODE_PROP || mode == MODE_PROP_WEIGHT)) {      // the form that follows k_resample_fx
            if (!b.mark || !b.fxs) return hipErrorInvalidValue;
            dim3 g(step_grid_x_marks(b), (unsigned)b.F, 1);
            if (mode == MODE_PROP) hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_PROP, PPT, true>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a);
            else hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_PROP_WEIGHT, PPT, true>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a);
            return hipGetLastError();
        }
    }
    dim3 g((unsigned)(b.Ns / (BLOCK * PPT)), (unsigned)b.F, 1);
    switch (mode) {
        case MODE_WEIGHT: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_WEIGHT, PPT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_PROP: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_PROP, PPT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_PROP_WEIGHT: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_PROP_WEIGHT, PPT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_AUX: hipLaunchKernelGGL((k_step<Model, NX, NY, MODE_AUX, PPT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a); break;
        default: return hipErrorInvalidValue;      // MODE_AUX2 is model independent: launch_step_aux2
    }
    return hipGetLastError();
}

template <int NX>
static hipError_t launch_step_lg_ny(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    switch (b.ny) {
        case 1: return launch_step_t<LinGauss<NX, 1>, NX, 1>(b, mode, a, s);
        case 2: return launch_step_t<LinGauss<NX, 2>, NX, 2>(b, mode, a, s);
        case 3: return launch_step_t<LinGauss<NX, 3>, NX, 3>(b, mode, a, s);
        case 4: return launch_step_t<LinGauss<NX, 4>, NX, 4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

template <int NX>
static hipError_t launch_step_rb_ny(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    if (mode == MODE_AUX) return hipErrorInvalidValue;
    switch (b.ny) {
        case 1: return launch_step_t<RBLin<NX, 1>, NX, 1>(b, mode, a, s);
        case 2: return launch_step_t<RBLin<NX, 2>, NX, 2>(b, mode, a, s);
        case 3: return launch_step_t<RBLin<NX, 3>, NX, 3>(b, mode, a, s);
        case 4: return launch_step_t<RBLin<NX, 4>, NX, 4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

// second half of the auxiliary predict! in balanced form: whatever the model, it propagates nothing (NoModel); lambda does not see NY
template <int NX>
static hipError_t launch_step_aux2_t(const BankDev& b, const StepArgs& a, hipStream_t s) {
    dim3 g((unsigned)(b.Ns / (BLOCK * STEP_PPT)), (unsigned)b.F, 1);
    hipLaunchKernelGGL((k_step<NoModel<NX>, NX, 1, MODE_AUX2, STEP_PPT>), g, dim3(BLOCK), 0, s, b, b.models, b.scal, a);
    return hipGetLastError();
}
static hipError_t launch_step_aux2(const BankDev& b, const StepArgs& a, hipStream_t s) {
    switch (b.nx) {
        case 1: return launch_step_aux2_t<1>(b, a, s);
        case 2: return launch_step_aux2_t<2>(b, a, s);
        case 3: return launch_step_aux2_t<3>(b, a, s);
        case 4: return launch_step_aux2_t<4>(b, a, s);
        case 5: return launch_step_aux2_t<5>(b, a, s);
        case 6: return launch_step_aux2_t<6>(b, a, s);
        case 7: return launch_step_aux2_t<7>(b, a, s);
        case 8: return launch_step_aux2_t<8>(b, a, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_step(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    const int model_id = b.model_id;
    if (mode == MODE_AUX2) return (model_id == LLPF_MODEL_RB_LINEAR || model_id == LLPF_MODEL_RB_BILINEAR) ? hipErrorInvalidValue : launch_step_aux2(b, a, s);
    if (model_id >= LLPF_MODEL_USER_BASE) return launch_step_user(b, mode, a, s);
    if (model_id == LLPF_MODEL_RB_BILINEAR) return launch_rbfull(b, mode, a, s);
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return launch_step_t<QuadTank<4, 2>, 4, 2, LLPF_QT_PPT>(b, mode, a, s);
    if (model_id == LLPF_MODEL_RB_LINEAR) {
        switch (b.nx) {
            case 2: return launch_step_rb_ny<2>(b, mode, a, s);
            case 3: return launch_step_rb_ny<3>(b, mode, a, s);
            case 4: return launch_step_rb_ny<4>(b, mode, a, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (b.nx) {
        case 1: return launch_step_lg_ny<1>(b, mode, a, s);
        case 2: return launch_step_lg_ny<2>(b, mode, a, s);
        case 3: return launch_step_lg_ny<3>(b, mode, a, s);
        case 4: return launch_step_lg_ny<4>(b, mode, a, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- resampling with source-side dynamics (kernels/resfx.hpp) ----
bool resample_fx_supported(int model_id, int nx, int ny, int strategy) {
    if (strategy != LLPF_RESAMPLE_SYSTEMATIC && strategy != LLPF_RESAMPLE_STRATIFIED) return false;   // residual ancestors are not sorted
    if (model_id >= LLPF_MODEL_USER_BASE) return jit_marks(model_id);
    return model_id == LLPF_MODEL_QUADTANK_RK4 && nx == 4 && ny == 2 && LLPF_QT_PPT == STEP_PPT;
}
template <class Model, int NX>
static hipError_t launch_resample_fx_t(const BankDev& b, const ResArgs& a, const StepArgs& st, hipStream_t s) {
    const dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    if (b.strategy == LLPF_RESAMPLE_SYSTEMATIC) hipLaunchKernelGGL((k_resample_fx<Model, NX, LLPF_RESAMPLE_SYSTEMATIC>), g, dim3(BLOCK), 0, s, b, a, st);
    else hipLaunchKernelGGL((k_resample_fx<Model, NX, LLPF_RESAMPLE_STRATIFIED>), g, dim3(BLOCK), 0, s, b, a, st);
    return hipGetLastError();
}
hipError_t launch_resample_fx(const BankDev& b, const ResArgs& a0, const StepArgs& st, hipStream_t s) {
    if (!resample_fx_supported(b.model_id, b.nx, b.ny, b.strategy) || !b.mark || !b.fxs) return hipErrorInvalidValue;
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    { const hipError_t e = launch_tile_prefix(b, a.parity, s); if (e != hipSuccess) return e; }      // (above 1024 tiles)
    if (b.model_id >= LLPF_MODEL_USER_BASE) return launch_resample_fx_user(b, a, st, s);
    return launch_resample_fx_t<QuadTank<4, 2>, 4>(b, a, st, s);
}

hipError_t launch_max(const BankDev& b, int parity, hipStream_t s) {
    hipLaunchKernelGGL(k_max, dim3((unsigned)(b.Ns / STEP_TILE), (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, parity);
    return hipGetLastError();
}


}  // namespace llpf

#if defined(LLPF_STEP_TIMING)
static unsigned long long* g_dbg_dev[2] = {nullptr, nullptr};
static int64_t g_dbg_blocks[2] = {0, 0};
extern "C" int llpf_debug_timing_arm(int which, int64_t blocks) {
    if (g_dbg_dev[which]) hipFree(g_dbg_dev[which]);
    g_dbg_blocks[which] = blocks;
    if (hipMalloc(&g_dbg_dev[which], sizeof(unsigned long long) * 16 * (size_t)blocks) != hipSuccess) return -1;
    hipMemset(g_dbg_dev[which], 0, sizeof(unsigned long long) * 16 * (size_t)blocks);
    if (which == 0) return hipMemcpyToSymbol(HIP_SYMBOL(g_step_dbg), &g_dbg_dev[0], sizeof(void*)) == hipSuccess ? 0 : -2;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fx_dbg), &g_dbg_dev[1], sizeof(void*)) == hipSuccess ? 0 : -2;
}
extern "C" int llpf_debug_timing_read(int which, unsigned long long* dst) {
    if (!g_dbg_dev[which]) return -1;
    return hipMemcpy(dst, g_dbg_dev[which], sizeof(unsigned long long) * 16 * (size_t)g_dbg_blocks[which], hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
