// k_resprop.hip — k_resprop (the fused timestep) and, in DEVTOOLS builds, the persistent multi-step form
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
#if defined(LLPF_DEVTOOLS) && !defined(LLPF_RESPROP_SPLIT_TU)
#include "kernels/persist.hpp"      // experiment kept for reference: measured slower than the graph of per-timestep launches
#endif

#ifdef LLPF_RESPROP_SPLIT_TU
// This translation unit compiled a second time (k_resprop_split.hip): the split-schedule forms of the fused kernel — weights written, exp-sums
// left to k_norm (banks beyond 3 M particles), or no weighting at all — with the Horner constants as SGPR pairs (bank -4 %; the merged
// single-filter form loses 6 % with them and stays in the first unit)
template <class Model, int NX, int NY>
static hipError_t launch_resprop_t(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    if (st.aux) return hipErrorInvalidValue;
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

#ifndef LLPF_RESPROP_SPLIT_TU      // the persistent form and its stubs belong to the first unit only
#ifdef LLPF_DEVTOOLS
// ---- persistent multi-step launch (kernels/persist.hpp): linear-Gaussian single filters whose tiles are all co-resident ----
template <class Model, int NX, int NY>
static hipError_t launch_persist_t(const BankDev& b, const PersistArgsHost& h, hipStream_t s, int* capacity) {
    auto fn = k_persist<Model, NX, NY>;
    if (capacity) {
        int per_cu = 0, dev = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BLOCK, 0);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
        *capacity = std::min(per_cu * prop.multiProcessorCount, GQ_GROUPS * 32);   // the two-level tile prefix holds GQ_GROUPS groups of 32 tiles
        return hipSuccess;
    }
    PersistArgs pa;
    pa.k_begin = h.k_begin; pa.k_end = h.k_end; pa.t_index0 = h.t_index0; pa.Ts = h.Ts; pa.U = h.U; pa.Y = h.Y;
    pa.x0 = h.x0; pa.x1 = h.x1; pa.q0 = h.q0; pa.q1 = h.q1; pa.cur0 = h.cur0; pa.qcur0 = h.qcur0; pa.par0 = h.par0;
    pa.step0 = h.step0; pa.np0 = h.np0; pa.need_e2 = h.need_e2; pa.K = llpf_qbits(b.N); pa.ll_steps = h.ll_steps;
    pa.bar = h.bar; pa.gq = reinterpret_cast<uint64_t*>(h.bar + BAR_WORDS); pa.ablate = h.ablate; pa.dbg_step = h.dbg_step; pa.dbg = h.dbg;
    BankDev bd = b;
    const ModelD* models = b.models;
    void* args[] = {&bd, &models, &pa};
    hipError_t e0 = hipMemsetAsync(pa.gq, 0, sizeof(uint64_t) * GQ_WORDS64, s);      // group sums start from zero; the barrier counters persist
    if (e0 != hipSuccess) return e0;
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(fn), dim3((unsigned)b.P2, 1, 1), dim3(BLOCK), args, 0, s);
}
template <int NX>
static hipError_t launch_persist_ny(const BankDev& b, const PersistArgsHost& h, hipStream_t s, int* capacity) {
    switch (b.ny) {
        case 1: return launch_persist_t<LinGauss<NX, 1>, NX, 1>(b, h, s, capacity);
        case 2: return launch_persist_t<LinGauss<NX, 2>, NX, 2>(b, h, s, capacity);
        case 3: return launch_persist_t<LinGauss<NX, 3>, NX, 3>(b, h, s, capacity);
        case 4: return launch_persist_t<LinGauss<NX, 4>, NX, 4>(b, h, s, capacity);
        default: return hipErrorInvalidValue;
    }
}
static hipError_t launch_persist_any(const BankDev& b, const PersistArgsHost& h, hipStream_t s, int* capacity) {
    if (b.model_id != LLPF_MODEL_LINEAR_GAUSSIAN || b.F != 1) return hipErrorInvalidValue;
    switch (b.nx) {
        case 1: return launch_persist_ny<1>(b, h, s, capacity);
        case 2: return launch_persist_ny<2>(b, h, s, capacity);
        case 3: return launch_persist_ny<3>(b, h, s, capacity);
        case 4: return launch_persist_ny<4>(b, h, s, capacity);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_persist(const BankDev& b, const PersistArgsHost& h, hipStream_t s) { return launch_persist_any(b, h, s, nullptr); }
hipError_t persist_capacity(const BankDev& b, int* blocks) { PersistArgsHost h{}; return launch_persist_any(b, h, nullptr, blocks); }
int persist_bar_words() { return BAR_WORDS + 2 * GQ_WORDS64; }
#else   // product build: the persistent form is not compiled in (it measured slower; DEVTOOLS=1 builds it for experiments)
hipError_t launch_persist(const BankDev&, const PersistArgsHost&, hipStream_t) { return hipErrorNotSupported; }
hipError_t persist_capacity(const BankDev&, int* blocks) { *blocks = 0; return hipSuccess; }
int persist_bar_words() { return 64; }
#endif
#endif


}  // namespace llpf
