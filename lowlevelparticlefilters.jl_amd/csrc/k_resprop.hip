// k_resprop.hip — k_resprop (the fused timestep)
// One of the engine's device translation units (kernels.hip has the map); split so that they build in parallel.

// the Horner constants of the shared math stay in VGPRs here: as SGPR pairs (the other translation units) the fused single-filter kernel
// spills 112 SGPRs to VGPR lanes and loses 6 % (C2 22.0 against 20.8 us)
#ifndef LLPF_HORNER_C
#define LLPF_HORNER_C(c) "v"(c)
#endif
#include "engine.hpp"

namespace llpf {

#define DEV __device__ __forceinline__

#include "kernels/reduce.hpp"
#include "kernels/models.hpp"
#include "kernels/accum.hpp"
#include "kernels/resample.hpp"
#include "kernels/resprop.hpp"

#ifdef LLPF_RESPROP_SPLIT_TU
// This translation unit compiled a second time (k_resprop_split.hip): the split-schedule forms of the fused kernel — weights written, exp-sums
// left to k_norm (banks beyond 3 M particles), or no weighting at all — with the Horner constants as SGPR pairs (bank -4 %; the merged
// single-filter form loses 6 % with them and stays in the first unit)
template <class Model, int NX, int NY>
static hipError_t launch_resprop_t(const BankDev& b0, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    dim3 g((unsigned)b0.P2, (unsigned)b0.F, 1);
    if (st.aux) return hipErrorInvalidValue;
    BankDev b = b0;
    if (a.lazy_q) b.quanta = reinterpret_cast<uint64_t*>(b0.w);      // ResArgs::lazy_q: the scan forms the quanta from the weights (same layout, [F][Ns] of 8 bytes)
    if (weight) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, false>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else hipLaunchKernelGGL((k_resprop<Model, NX, NY, false, false>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    return hipGetLastError();
}
#else
hipError_t launch_resprop_split(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s);
template <class Model, int NX, int NY>
static hipError_t launch_resprop_t(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const bool one = b.P2 == 1;      // one-tile filters: the variant that redoes a failed bound test in place (kernels/resprop.hpp)
    if (st.aux && one) hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (st.aux) hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (weight && st.accumulate && one) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, true, false, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (weight && st.accumulate) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else return launch_resprop_split(b, a, st, weight, s);      // the split-schedule forms live in k_resprop_split.hip
    return hipGetLastError();
}
#endif
template <int NX>
static hipError_t launch_resprop_lg_ny(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    if (st.aux) return launch_resprop_t<LinGauss<NX, 1>, NX, 1>(b, a, st, weight, s);     // the auxiliary second half does not see the measurement
    switch (b.ny) {
        case 1: return launch_resprop_t<LinGauss<NX, 1>, NX, 1>(b, a, st, weight, s);
        case 2: return launch_resprop_t<LinGauss<NX, 2>, NX, 2>(b, a, st, weight, s);
        case 3: return launch_resprop_t<LinGauss<NX, 3>, NX, 3>(b, a, st, weight, s);
        case 4: return launch_resprop_t<LinGauss<NX, 4>, NX, 4>(b, a, st, weight, s);
        default: return hipErrorInvalidValue;
    }
}
template <int NX>
static hipError_t launch_resprop_rb_ny(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    // the nonlinear / linear split as a compile-time constant (kernels/models.hpp: RBLin<NX, NY, NN>; measured on the reference's own RBPF
    // benchmark system, test/test_rbpf.jl:5-31: 1 + 1 states, one output: 30.0 -> 22.7 us per timestep) for every split with one or two
    // outputs; BankDev::pad0 carries nxn for this model.  Three and four outputs: the run-time split, same bits.
    if (b.ny <= 2 && b.pad0 >= 1 && b.pad0 < NX) {
#define LLPF_RB_LEAN(NN) if constexpr (NN < NX) { if (b.pad0 == NN) return b.ny == 1 ? launch_resprop_t<RBLin<NX, 1, NN>, NX, 1>(b, a, st, weight, s) \
                                                                                     : launch_resprop_t<RBLin<NX, 2, NN>, NX, 2>(b, a, st, weight, s); }
        LLPF_RB_LEAN(1) LLPF_RB_LEAN(2) LLPF_RB_LEAN(3)
#undef LLPF_RB_LEAN
    }
    switch (b.ny) {
        case 1: return launch_resprop_t<RBLin<NX, 1>, NX, 1>(b, a, st, weight, s);
        case 2: return launch_resprop_t<RBLin<NX, 2>, NX, 2>(b, a, st, weight, s);
        case 3: return launch_resprop_t<RBLin<NX, 3>, NX, 3>(b, a, st, weight, s);
        case 4: return launch_resprop_t<RBLin<NX, 4>, NX, 4>(b, a, st, weight, s);
        default: return hipErrorInvalidValue;
    }
}
// state dimensions 5..8 (models compiled on demand): only the auxiliary filter's second half comes here, which propagates nothing
template <int NX>
static hipError_t launch_resprop_aux_only(const BankDev& b, const ResArgs& a, const StepArgs& st, hipStream_t s) {
#ifdef LLPF_RESPROP_SPLIT_TU
    return hipErrorInvalidValue;
#else
    if (!st.aux) return hipErrorInvalidValue;
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    if (b.P2 == 1) hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    return hipGetLastError();
#endif
}
#ifdef LLPF_RESPROP_SPLIT_TU
hipError_t launch_resprop_split(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
#else
hipError_t launch_resprop(const BankDev& b, const ResArgs& a0, const StepArgs& st, int weight, hipStream_t s) {
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    a.mode = RES_FINALIZE | RES_RESAMPLE;
    { const hipError_t e = launch_tile_prefix(b, a.parity, s); if (e != hipSuccess) return e; }      // (above 1024 tiles)
#endif
    // a run-time compiled model has no fused kernel: only the auxiliary second half (which propagates nothing: NoModel) may come here
    if (b.model_id >= LLPF_MODEL_USER_BASE && !st.aux) return hipErrorInvalidValue;
    if (b.model_id == LLPF_MODEL_QUADTANK_RK4) return launch_resprop_t<QuadTank<4, 2>, 4, 2>(b, a, st, weight, s);
    if (b.model_id == LLPF_MODEL_RB_LINEAR) {
        switch (b.nx) {
            case 2: return launch_resprop_rb_ny<2>(b, a, st, weight, s);
            case 3: return launch_resprop_rb_ny<3>(b, a, st, weight, s);
            case 4: return launch_resprop_rb_ny<4>(b, a, st, weight, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (b.nx) {
        case 1: return launch_resprop_lg_ny<1>(b, a, st, weight, s);
        case 2: return launch_resprop_lg_ny<2>(b, a, st, weight, s);
        case 3: return launch_resprop_lg_ny<3>(b, a, st, weight, s);
        case 4: return launch_resprop_lg_ny<4>(b, a, st, weight, s);
        case 5: return launch_resprop_aux_only<5>(b, a, st, s);
        case 6: return launch_resprop_aux_only<6>(b, a, st, s);
        case 7: return launch_resprop_aux_only<7>(b, a, st, s);
        case 8: return launch_resprop_aux_only<8>(b, a, st, s);
        default: return hipErrorInvalidValue;
    }
}


}  // namespace llpf
