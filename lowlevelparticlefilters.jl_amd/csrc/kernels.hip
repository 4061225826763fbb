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
#include "engine.hpp"

namespace llpf {

#define DEV __device__ __forceinline__

#include "kernels/reduce.hpp"
#include "kernels/models.hpp"
#include "kernels/init.hpp"
#include "kernels/accum.hpp"
#include "kernels/norm.hpp"
#include "kernels/resample.hpp"
#include "kernels/residual.hpp"
#include "kernels/access.hpp"
#include "kernels/smooth.hpp"
#include "kernels/selftest.hpp"

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline dim3 grid1(int64_t n, int F) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK), (unsigned)F, 1); }

hipError_t launch_init(const BankDev& b, uint32_t step, int init_anc, hipStream_t s) {
    dim3 g = grid1(b.Ns, b.F);
    switch (b.nx) {
        case 1: hipLaunchKernelGGL(k_init<1>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 2: hipLaunchKernelGGL(k_init<2>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 3: hipLaunchKernelGGL(k_init<3>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 4: hipLaunchKernelGGL(k_init<4>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 5: hipLaunchKernelGGL(k_init<5>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;   // 5..8: models compiled on demand
        case 6: hipLaunchKernelGGL(k_init<6>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 7: hipLaunchKernelGGL(k_init<7>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 8: hipLaunchKernelGGL(k_init<8>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 9: hipLaunchKernelGGL(k_init<9>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 10: hipLaunchKernelGGL(k_init<10>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 11: hipLaunchKernelGGL(k_init<11>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 12: hipLaunchKernelGGL(k_init<12>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 13: hipLaunchKernelGGL(k_init<13>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 14: hipLaunchKernelGGL(k_init<14>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 15: hipLaunchKernelGGL(k_init<15>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        case 16: hipLaunchKernelGGL(k_init<16>, g, dim3(BLOCK), 0, s, b, b.models, b.scal, step, init_anc); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int NX, bool XMEAN>
static void launch_norm_e2(const BankDev& b, int parity, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s) {
    dim3 g((unsigned)b.P2, (unsigned)b.F, 1);
    const int K = llpf_qbits(b.N);
    if (need_e2) hipLaunchKernelGGL((k_norm<NX, XMEAN, true>), g, dim3(BLOCK), 0, s, LLPF_NORM_HOT_ARGS(b), kstep, K, parity, only_fallback, bound, step, b);
    else hipLaunchKernelGGL((k_norm<NX, XMEAN, false>), g, dim3(BLOCK), 0, s, LLPF_NORM_HOT_ARGS(b), kstep, K, parity, only_fallback, bound, step, b);
}
hipError_t launch_norm(const BankDev& b, int parity, int want_xmean, int need_e2, uint32_t step, int only_fallback, int bound, int64_t kstep, hipStream_t s) {
    if (!want_xmean) { launch_norm_e2<0, false>(b, parity, need_e2, step, only_fallback, bound, kstep, s); return hipGetLastError(); }
    switch (b.nx) {
        case 1: launch_norm_e2<1, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 2: launch_norm_e2<2, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 3: launch_norm_e2<3, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 4: launch_norm_e2<4, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 5: launch_norm_e2<5, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 6: launch_norm_e2<6, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 7: launch_norm_e2<7, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 8: launch_norm_e2<8, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 9: launch_norm_e2<9, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 10: launch_norm_e2<10, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 11: launch_norm_e2<11, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 12: launch_norm_e2<12, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 13: launch_norm_e2<13, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 14: launch_norm_e2<14, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 15: launch_norm_e2<15, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
        case 16: launch_norm_e2<16, true>(b, parity, need_e2, step, only_fallback, bound, kstep, s); break;
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

hipError_t launch_requant(const BankDev& b, hipStream_t s) {
    hipLaunchKernelGGL(k_requant, dim3((unsigned)((b.Ns / 2 + BLOCK - 1) / BLOCK), (unsigned)b.F, 1), dim3(BLOCK), 0, s, b);
    return hipGetLastError();
}

hipError_t launch_tile_prefix(const BankDev& b, int parity, hipStream_t s) {
    if (b.P2 <= TQ_GROUP) return hipSuccess;          // a head reads the tile sums themselves
    if (!b.tpre || !b.gsum) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_tile_prefix<0>, dim3((unsigned)((b.P2 + TQ_GROUP - 1) / TQ_GROUP), (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, parity);
    return hipGetLastError();
}

hipError_t launch_resample(const BankDev& b, const ResArgs& a0, hipStream_t s) {
    ResArgs a = a0;
    a.K = llpf_qbits(b.N);
    // a finalize-only launch needs just one block per filter
    dim3 g((a.mode & RES_RESAMPLE) ? (unsigned)b.P2 : 1u, (unsigned)b.F, 1);
    if (a.src_values) hipLaunchKernelGGL(k_qpart, dim3((unsigned)b.P2, (unsigned)b.F, 1), dim3(BLOCK), 0, s, b, a.K);
    if (a.mode & RES_RESAMPLE) { const hipError_t e = launch_tile_prefix(b, a.parity, s); if (e != hipSuccess) return e; }      // (above 1024 tiles)
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
    if (b.model_id >= LLPF_MODEL_USER_BASE) return launch_smooth_fx_user(b, a, s);     // compiled with the user's dynamics (kernels/jit.hpp)
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
        case 5: hipLaunchKernelGGL((k_smooth_draw<5>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 6: hipLaunchKernelGGL((k_smooth_draw<6>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 7: hipLaunchKernelGGL((k_smooth_draw<7>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 8: hipLaunchKernelGGL((k_smooth_draw<8>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 9: hipLaunchKernelGGL((k_smooth_draw<9>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 10: hipLaunchKernelGGL((k_smooth_draw<10>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 11: hipLaunchKernelGGL((k_smooth_draw<11>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 12: hipLaunchKernelGGL((k_smooth_draw<12>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 13: hipLaunchKernelGGL((k_smooth_draw<13>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 14: hipLaunchKernelGGL((k_smooth_draw<14>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 15: hipLaunchKernelGGL((k_smooth_draw<15>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
        case 16: hipLaunchKernelGGL((k_smooth_draw<16>), g, dim3(BLOCK), 0, s, b, b.models, a); break;
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
hipError_t launch_wcov(const BankDev& b, const double* mean, double* out, hipStream_t s) {
    if (b.nx <= 8) hipLaunchKernelGGL((k_wcov<8>), dim3((unsigned)b.F), dim3(BLOCK), 0, s, b, mean, out);
    else hipLaunchKernelGGL((k_wcov<MAXD>), dim3((unsigned)b.F), dim3(BLOCK), 0, s, b, mean, out);
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
