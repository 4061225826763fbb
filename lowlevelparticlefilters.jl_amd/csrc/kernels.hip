// kernels.hip — hand-written gfx950 kernels of the particle-filter step (one translation unit; the pieces live in
// kernels/*.hpp in dependency order).
//
// One timestep (correct! then predict!, reference src/filtering.jl:164-168, 140-153) of the run loop is
//   one launch   k_resprop: head (finalize logsumexp!/ESS from the sharded integer accumulators, shouldresample),
//                integer scan of the tile's quanta, ancestor COUNTS for the systematic / stratified thresholds, then
//                for every output of the block: gather x[owner] -> dynamics -> + Philox / Box-Muller noise -> store ->
//                w = w_prev + logpdf(y_next - g(x)) -> exp-sums against the analytic bound -> sharded atomics
//                                                              (single linear-Gaussian filter)
//   two launches k_norm + k_resprop (banks: split schedule) or k_resample + k_step (balanced form: quad-tank,
//                Rao-Blackwellized model, residual resampling, history outputs, single-step API)
//   three        auxiliary filter: k_step<MODE_AUX> + k_resprop<AUX> + finalize
// All particle data is fp64 structure-of-arrays; wave64; 256-thread workgroups; grid = (tiles, filters).
// Compiled with -ffp-contract=off: the arithmetic is the same IEEE sequence as oracle/llpf_oracle.c (device order).
#include <hip/hiprtc.h>

#include <mutex>
#include <vector>

#include "engine.hpp"
#include "jit_prelude.inc"

namespace llpf {

#define DEV __device__ __forceinline__

#include "kernels/reduce.hpp"
#include "kernels/models.hpp"
#include "kernels/init.hpp"
#include "kernels/accum.hpp"
#include "kernels/step.hpp"
#include "kernels/rbfull.hpp"
#include "kernels/norm.hpp"
#include "kernels/resample.hpp"
#include "kernels/residual.hpp"
#include "kernels/resprop.hpp"
#ifdef LLPF_DEVTOOLS
#include "kernels/persist.hpp"      // experiment kept for reference: measured slower than the graph of per-timestep launches
#endif
#include "kernels/access.hpp"
#include "kernels/smooth.hpp"
#include "kernels/selftest.hpp"
#include "kernels/jit.hpp"

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid1(int64_t n, int F) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK), (unsigned)F, 1); }

// LLPF_MODEL_RB_BILINEAR: the instantiated shapes (nxn, nxl, ny); fn_kind 1 = quad-tank nonlinear part
bool rbfull_supported(int fn_kind, int nn, int nl, int ny) {
    if (fn_kind == 1) return nn == 4 && nl == 8 && ny == 2;
    if (fn_kind != 0) return false;
    return (nn == 1 && nl == 2 && ny == 1) || (nn == 2 && nl == 2 && ny == 2) || (nn == 4 && nl == 8 && ny == 2);
}
int rbfull_rows(int nn, int nl) { return nn + nl + LLPF_RBF_NP(nl); }

template <class Model, int NN, int NL, int NY>
static hipError_t launch_rbfull_t(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    dim3 g((unsigned)(b.Ns / RBF_BLOCK), (unsigned)b.F, 1);
    switch (mode) {
        case MODE_WEIGHT: hipLaunchKernelGGL((k_rbfull<Model, NN, NL, NY, MODE_WEIGHT>), g, dim3(RBF_BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_PROP: hipLaunchKernelGGL((k_rbfull<Model, NN, NL, NY, MODE_PROP>), g, dim3(RBF_BLOCK), 0, s, b, b.models, b.scal, a); break;
        case MODE_PROP_WEIGHT: hipLaunchKernelGGL((k_rbfull<Model, NN, NL, NY, MODE_PROP_WEIGHT>), g, dim3(RBF_BLOCK), 0, s, b, b.models, b.scal, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// BankDev::pad0 carries the shape of this model: nxl | fn_kind << 8
static hipError_t launch_rbfull(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    const int nl = b.pad0 & 0xff, fk = (b.pad0 >> 8) & 0xff;
    if (fk == 1 && b.nx == 4 && nl == 8 && b.ny == 2) return launch_rbfull_t<QuadTank<4, 2>, 4, 8, 2>(b, mode, a, s);
    if (fk == 0 && b.nx == 4 && nl == 8 && b.ny == 2) return launch_rbfull_t<LinGauss<4, 2>, 4, 8, 2>(b, mode, a, s);
    if (fk == 0 && b.nx == 2 && nl == 2 && b.ny == 2) return launch_rbfull_t<LinGauss<2, 2>, 2, 2, 2>(b, mode, a, s);
    if (fk == 0 && b.nx == 1 && nl == 2 && b.ny == 1) return launch_rbfull_t<LinGauss<1, 1>, 1, 2, 1>(b, mode, a, s);
    return hipErrorInvalidValue;
}
hipError_t launch_rbfull_init(const BankDev& b, hipStream_t s) {
    const int nl = b.pad0 & 0xff;
    dim3 g = grid1(b.Ns, b.F);
    if (b.nx == 4 && nl == 8) hipLaunchKernelGGL((k_rbfull_init<4, 8>), g, dim3(BLOCK), 0, s, b, b.models);
    else if (b.nx == 2 && nl == 2) hipLaunchKernelGGL((k_rbfull_init<2, 2>), g, dim3(BLOCK), 0, s, b, b.models);
    else if (b.nx == 1 && nl == 2) hipLaunchKernelGGL((k_rbfull_init<1, 2>), g, dim3(BLOCK), 0, s, b, b.models);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

bool step_supported(int model_id, int nx, int ny) {
    if (model_id >= LLPF_MODEL_USER_BASE) return jit_supported(model_id, nx, ny);
    if (model_id == LLPF_MODEL_RB_BILINEAR) return nx >= 1 && nx <= 4 && ny >= 1 && ny <= 2;   // shape checked by rbfull_supported
    if (model_id == LLPF_MODEL_QUADTANK_RK4) return nx == 4 && ny == 2;
    if (model_id == LLPF_MODEL_RB_LINEAR) return nx >= 2 && nx <= 4 && ny >= 1 && ny <= 4;
    if (model_id == LLPF_MODEL_LINEAR_GAUSSIAN) return nx >= 1 && nx <= 4 && ny >= 1 && ny <= 4;
    return false;
}

hipError_t launch_init(const BankDev& b, uint32_t step, int init_anc, hipStream_t s) {
    dim3 g = grid1(b.Ns, b.F);
    switch (b.nx) {
        case 1: hipLaunchKernelGGL(k_init<1>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 2: hipLaunchKernelGGL(k_init<2>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 3: hipLaunchKernelGGL(k_init<3>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 4: hipLaunchKernelGGL(k_init<4>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
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
        default: return hipErrorInvalidValue;
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

hipError_t launch_step(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    const int model_id = b.model_id;
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

template <int NX, bool XMEAN>
static void launch_norm_e2(const BankDev& b, int parity, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const int K = llpf_qbits(b.N);
    if (need_e2) hipLaunchKernelGGL((k_norm<NX, XMEAN, true>), g, dim3(BLOCK), 0, s, b, K, parity, step, only_fallback, bound, kstep);
    else hipLaunchKernelGGL((k_norm<NX, XMEAN, false>), g, dim3(BLOCK), 0, s, b, K, parity, step, only_fallback, bound, kstep);
}
hipError_t launch_norm(const BankDev& b, int parity, int want_xmean, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s) {
    if (!want_xmean) { launch_norm_e2<0, false>(b, parity, need_e2, step, only_fallback, bound, kstep, s); return hipGetLastError(); }
    switch (b.nx) {
        case 1: launch_norm_e2<1, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 2: launch_norm_e2<2, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 3: launch_norm_e2<3, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 4: launch_norm_e2<4, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_ess(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_ess, dim3((unsigned)b.F), dim3(BLOCK), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_replicate_models(ModelD* models, int F, hipStream_t s) {
    if (F > 1) hipLaunchKernelGGL(k_replicate_models, dim3((unsigned)(F - 1)), dim3(256), 0, s, models);
    return hipGetLastError();
}
hipError_t launch_fb_clear(const BankDev& b, int slot, int mode, hipStream_t s) {
    hipLaunchKernelGGL(k_fb_clear, dim3((unsigned)b.F), dim3(64), 0, s, b, slot, mode);
    if (mode == 1) hipLaunchKernelGGL(k_fb_clear_flag, dim3(1), dim3(64), 0, s, b);     // after every filter's flag: same stream
    return hipGetLastError();
}

hipError_t launch_post_predict(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_post_predict, dim3((unsigned)((b.F + 63) / 64)), dim3(64), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_resample(const BankDev& b, const ResArgs& a0, hipStream_t s) {
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    // a finalize-only launch needs just one block per filter
    dim3 g((a.mode & RES_RESAMPLE) ? (unsigned)b.P2 : 1u, (unsigned)b.F, 1);
    if (a.src_values) hipLaunchKernelGGL(k_qpart, dim3((unsigned)b.P2, (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, a.K);
    if (b.strategy == LLPF_RESAMPLE_RESIDUAL && (a.mode & RES_RESAMPLE) && !a.only_bins) {
        const dim3 gt((unsigned)b.P2, (unsigned)b.F, 1), gf((unsigned)b.F, 1, 1);
        if (!a.src_values) {
            hipLaunchKernelGGL((k_resid_prep<SRC_FILTER>), gt, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_scan<SRC_FILTER>), gf, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_expand<SRC_FILTER>), gt, dim3(BLOCK), 0, s, b, a);
        } else {
            hipLaunchKernelGGL((k_resid_prep<SRC_VALUES>), gt, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_scan<SRC_VALUES>), gf, dim3(BLOCK), 0, s, b, a);
            hipLaunchKernelGGL((k_resid_expand<SRC_VALUES>), gt, dim3(BLOCK), 0, s, b, a);
        }
        return hipGetLastError();
    }
    if (b.strategy == LLPF_RESAMPLE_SYSTEMATIC) {
        if (!a.src_values) hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_SYSTEMATIC, SRC_FILTER>), g, dim3(BLOCK), 0, s, b, a);
        else hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_SYSTEMATIC, SRC_VALUES>), g, dim3(BLOCK), 0, s, b, a);
    } else {
        if (!a.src_values) hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_STRATIFIED, SRC_FILTER>), g, dim3(BLOCK), 0, s, b, a);
        else hipLaunchKernelGGL((k_resample<LLPF_RESAMPLE_STRATIFIED, SRC_VALUES>), g, dim3(BLOCK), 0, s, b, a);
    }
    return hipGetLastError();
}

template <class Model, int NX, int NY>
static hipError_t launch_resprop_t(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const bool one = b.P2 == 1;      // one-tile filters: the variant that redoes a failed bound test in place (kernels/resprop.hpp)
    if (st.aux && one) hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (st.aux) hipLaunchKernelGGL((k_resprop<NoModel<NX>, NX, 1, true, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (weight && st.accumulate && one) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, true, false, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (weight && st.accumulate) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, true>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else if (weight) hipLaunchKernelGGL((k_resprop<Model, NX, NY, true, false>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    else hipLaunchKernelGGL((k_resprop<Model, NX, NY, false, false>), g, dim3(BLOCK), 0, s, LLPF_HOT_ARGS(b, a), b, b.models, a, st);
    return hipGetLastError();
}
template <int NX>
static hipError_t launch_resprop_lg_ny(const BankDev& b, const ResArgs& a, const StepArgs& st, int weight, hipStream_t s) {
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
hipError_t launch_resprop(const BankDev& b, const ResArgs& a0, const StepArgs& st, int weight, hipStream_t s) {
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    a.mode = RES_FINALIZE | RES_RESAMPLE;
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
        default: return hipErrorInvalidValue;
    }
}

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

template <class Model, int NX, int NY>
static hipError_t launch_smooth_fx_t(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((k_smooth_fx<Model, NX, NY>), grid1(b.N, 1), dim3(BLOCK), 0, s, b, b.models, a);
    return hipGetLastError();
}
template <int NX>
static hipError_t launch_smooth_fx_ny(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    switch (b.ny) {
        case 1: return launch_smooth_fx_t<LinGauss<NX, 1>, NX, 1>(b, a, s);
        case 2: return launch_smooth_fx_t<LinGauss<NX, 2>, NX, 2>(b, a, s);
        case 3: return launch_smooth_fx_t<LinGauss<NX, 3>, NX, 3>(b, a, s);
        case 4: return launch_smooth_fx_t<LinGauss<NX, 4>, NX, 4>(b, a, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_smooth_fx(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    if (b.model_id >= LLPF_MODEL_USER_BASE) return hipErrorInvalidValue;     // no smoother kernel is compiled for user models
    if (b.model_id == LLPF_MODEL_QUADTANK_RK4) return launch_smooth_fx_t<QuadTank<4, 2>, 4, 2>(b, a, s);
    switch (b.nx) {
        case 1: return launch_smooth_fx_ny<1>(b, a, s);
        case 2: return launch_smooth_fx_ny<2>(b, a, s);
        case 3: return launch_smooth_fx_ny<3>(b, a, s);
        case 4: return launch_smooth_fx_ny<4>(b, a, s);
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_smooth_draw(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    const dim3 g((unsigned)a.M, 1, 1);
    switch (b.nx) {
        case 1: hipLaunchKernelGGL((k_smooth_draw<1>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 2: hipLaunchKernelGGL((k_smooth_draw<2>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 3: hipLaunchKernelGGL((k_smooth_draw<3>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 4: hipLaunchKernelGGL((k_smooth_draw<4>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_bake_weights(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_bake_weights, grid1(b.Ns, b.F), dim3(BLOCK), 0, s, b);
    return hipGetLastError();
}
hipError_t launch_materialize(const BankDev& b, double* w_out, double* we_out, hipStream_t s) {
    hipLaunchKernelGGL(k_materialize, grid1(b.N, b.F), dim3(BLOCK), 0, s, b, w_out, we_out);
    return hipGetLastError();
}
hipError_t launch_soa2aos(const BankDev& b, const double* xsrc, double* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_soa2aos, grid1(b.N, b.F), dim3(BLOCK), 0, s, b, xsrc, dst);
    return hipGetLastError();
}
hipError_t launch_aos2soa(const BankDev& b, const double* src, double* xdst, hipStream_t s) {
    hipLaunchKernelGGL(k_aos2soa, grid1(b.Ns, b.F), dim3(BLOCK), 0, s, b, src, xdst);
    return hipGetLastError();
}
hipError_t launch_wmean(const BankDev& b, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_wmean, dim3((unsigned)b.F), dim3(BLOCK), 0, s, b, out);
    return hipGetLastError();
}
hipError_t launch_anc64(const BankDev& b, int64_t* dst, hipStream_t s) {
    hipLaunchKernelGGL(k_anc64, grid1(b.N, b.F), dim3(BLOCK), 0, s, b, dst);
    return hipGetLastError();
}
hipError_t launch_selftest_math(int which, const double* in, double* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, which, in, out, n);
    return hipGetLastError();
}
hipError_t launch_selftest_normals(uint32_t k0, uint32_t k1, uint32_t step, uint32_t stream, int nd,
                                   double* out, int64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, k0, k1, step, stream, nd, out, n);
    return hipGetLastError();
}

}  // namespace llpf
