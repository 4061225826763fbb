// k_step.hip — k_step (balanced propagate + weight: every model, every mode), k_max, and the run-time compiled user models
// One of the engine's device translation units (kernels.hip has the map); split so that they build in parallel.
#include <hip/hiprtc.h>

#include <mutex>
#include <string>
#include <vector>

#include "engine.hpp"
#include "jit_prelude.inc"

namespace llpf {

#define DEV __device__ __forceinline__

#include "kernels/reduce.hpp"
#include "kernels/models.hpp"
#include "kernels/accum.hpp"
#include "kernels/step.hpp"
#include "kernels/jit.hpp"

bool step_supported(int model_id, int nx, int ny) {
    if (model_id >= LLPF_MODEL_USER_BASE) return jit_supported(model_id, nx, ny);
    if (model_id == LLPF_MODEL_RB_BILINEAR) return nx >= 1 && nx <= 4 && ny >= 1 && ny <= 2;   // shape checked by rbfull_supported
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return nx == 4 && ny == 2;
    if (model_id == LLPF_MODEL_RB_LINEAR) return nx >= 2 && nx <= 4 && ny >= 1 && ny <= 4;
    if (model_id == LLPF_MODEL_LINEAR_GAUSSIAN) return nx >= 1 && nx <= MAXD && ny >= 1 && ny <= MAXD;   // above 4: compiled on demand (jit_builtin_lg)
    return false;
}

#ifndef LLPF_QT_PPT
#define LLPF_QT_PPT 2
#endif
template <class Model, int NX, int NY, int PPT = STEP_PPT>
static hipError_t launch_step_t(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    dim3 g((unsigned)(b.Ns / (BLOCK * PPT * STEP_ITERS)), (unsigned)b.F, 1);
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
    dim3 g((unsigned)(b.Ns / (BLOCK * STEP_PPT * STEP_ITERS)), (unsigned)b.F, 1);
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

hipError_t launch_max(const BankDev& b, int parity, hipStream_t s) {
    hipLaunchKernelGGL(k_max, dim3((unsigned)(b.Ns / STEP_TILE), (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, parity);
    return hipGetLastError();
}


}  // namespace llpf
