// kernels/rbfull.hpp — k_rbfull: propagate + weight of the Rao-Blackwellized filter with per-particle covariance
// (LLPF_MODEL_RB_BILINEAR, reference src/rbpf.jl:163-283 with singleR off).  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// Particle plane layout of this model (single filter): rows 0..NN-1 of x are xn, rows NN..NN+NL-1 the Kalman mean xl,
// the following NL(NL+1)/2 rows the packed lower triangle of the Kalman covariance R: one ancestor gather moves the
// whole RBParticle (src/rbpf.jl:1-5, :182 `xi = s.xprev[j[i]]`).  One particle per thread; all matrices in registers
// (csrc/shared/llpf_rbfull.h, shared with the oracle).  S_i = C R_i C' + R2 >= R2, so max(w_prev) + c0(R2) bounds every new
// weight: the exp-sums against that bound are formed here (merged schedule, StepArgs::accumulate) or by a k_norm launch.
// ------------------------------------------------------------------------------------------------
#define RBF_KCPTR(p) ((llpf_rbf_cptr)(p))
constexpr double RBF_BOUND_SLACK = 0x1p-20;   // keeps exp(w - bound) <= 1 when a particle's C R C' rounds to zero

#ifndef LLPF_RBF_WAVES
#define LLPF_RBF_WAVES 2
#endif
// One wave per workgroup: a particle costs ~6000 instructions here and N = 2e5 is 3.06 waves per SIMD, so the launch ends with the
// SIMDs that got a fourth wave; single-wave groups let the dispatcher hand a wave to whichever SIMD frees a slot.
constexpr int RBF_BLOCK = 64;
template <class Model, int NN, int NL, int NY, int MODE>
__global__ __launch_bounds__(RBF_BLOCK) __attribute__((amdgpu_waves_per_eu(NL >= 8 ? LLPF_RBF_WAVES : 1))) void k_rbfull(BankDev b, const ModelD* __restrict__ models,
                                                   const FilterScal* scal, StepArgs a) {
    static_assert(MODE == MODE_WEIGHT || MODE == MODE_PROP || MODE == MODE_PROP_WEIGHT, "no auxiliary form");
    constexpr int NP = LLPF_RBF_NP(NL), ROWS = NN + NL + NP;
    const int f = blockIdx.y;
    const ModelD* md = models + f;
    const FilterScal* sc = scal + f;
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    RBF_STAMP(0);
    { uint32_t hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); g_rbf_dbg[(size_t)blockIdx.x * 16 + 13] = hw_; }
#endif
    // everything the prologue reads is REQUESTED first and tested afterwards (as in k_step): tested one by one, the stop flag, the
    // fallback flag and the scalars were eight scalar-cache round trips in a row in front of the gather
    const uint32_t stop_flag = *b.bank_flag;
    const int fb_flag = sc->fallback;
    const int do_res = (MODE != MODE_WEIGHT) ? sc->do_resample : 0;
    const int uniform = sc->uniform, pend = sc->norm_pending;
    const double m = sc->m, l = sc->l, wconst = sc->wconst, wmax_prev = sc->wmax;
    const uint32_t k0 = sc->k0, k1 = sc->k1, sb = sc->step_base;
    const int64_t Ns = b.Ns, N = b.N;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : : "s"(stop_flag), "s"(fb_flag), "s"(do_res), "s"(uniform), "s"(pend), "s"(m), "s"(l), "s"(wconst), "s"(wmax_prev), "s"(k0), "s"(k1),
                 "s"(sb), "s"(Ns), "s"(N), "s"(a.k), "s"(a.only_fallback), "s"(a.has_y), "s"(a.step));
    asm volatile("" : : "s"(b.nu), "s"(b.log1N), "s"(b.anc_slot), "s"(b.F), "s"(a.next_step), "s"(a.parity), "s"(a.K), "s"(a.need_e2), "s"(a.accumulate),
                 "s"(a.t_prop), "s"(b.anc), "s"(b.xcur), "s"(b.xnext), "s"(b.w), "s"(b.quanta_next), "s"(b.acc), "s"(b.tileq), "s"(a.u), "s"(a.y));
#endif
    if (stop_flag != 0 && (int64_t)(stop_flag - 1) < a.k) return;          // run_is_stopped
    if (a.only_fallback ? !fb_flag : (fb_flag != 0)) return;
    const double* __restrict__ xc = b.xcur + (size_t)f * ROWS * Ns;
    double* __restrict__ xo = (MODE == MODE_WEIGHT) ? const_cast<double*>(xc) : b.xnext + (size_t)f * ROWS * Ns;
    double* w = b.w + (size_t)f * Ns;
    const llpf_rbf_par* par = &md->rbf;

    // The gather first: its HBM latency (~2 us) is what everything below waits for, so the particle-independent work of the
    // prologue is placed behind the loads' issue, not in front of it.
    const int64_t i = (int64_t)blockIdx.x * RBF_BLOCK + threadIdx.x;
    const int64_t src = do_res ? (int64_t)b.anc[(size_t)f * Ns + i] : i;
    // 32-bit byte offsets from ONE uniform base per buffer (48 planes: 64-bit addresses would hold 96 registers and cost two
    // instructions each); the launcher checks that a filter's planes span less than 4 GB
    const uint32_t stride = (uint32_t)Ns * 8u, so = (uint32_t)src * 8u, io = (uint32_t)i * 8u;
    auto ld = [&](int row, uint32_t off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xc) + (off + (uint32_t)row * stride)); };
    auto st = [&](int row, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(xo) + (io + (uint32_t)row * stride)) = v; };
    double xn[NN], xl[NL], R[NP];
#pragma unroll
    for (int d = 0; d < NN; ++d) xn[d] = ld(d, so);
#pragma unroll
    for (int d = 0; d < NL; ++d) xl[d] = ld(NN + d, so);
#pragma unroll
    for (int d = 0; d < NP; ++d) R[d] = ld(NN + NL + d, so);

    const double* uf = a.u + (size_t)f * a.u_stride;       // banks on data of their own (llpf_bank_run_multi): filter f's row
    Model model;
    model.prepare(md, uf, a.t_prop);
    // Bl u is particle-independent and nu a run-time number: formed once per wave, read back by the time update from LDS
    // (as a branch inside the unrolled body it cut the body into blocks that each kept their constants' SGPRs alive)
    __shared__ double sh_blu[LLPF_RBF_MAXL];
    if (MODE != MODE_WEIGHT) {
#pragma unroll
        for (int r = 0; r < NL; ++r) {      // uniform addresses only: the parameter pointer must stay scalar (RBF_STAGE takes it in SGPRs)
            const double v = llpf_rbf_blu_row(RBF_KCPTR(par), b.nu, r, uf);
            if (threadIdx.x == 0) sh_blu[r] = v;
        }
        __syncthreads();
    }

    if (MODE != MODE_WEIGHT) {
        double fi[NN], xi[NN], nz[NN], xn1[NN], xl1[NL], R1[NP];
        model.dynamics(xn, fi);
        llpf_normals((uint32_t)i, sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NN, xi);
        gauss_sample<NN>(md->df, xi, nz);
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : : "v"(fi[NN - 1]), "v"(nz[NN - 1]));      // RK4 and the generator are done before stamp 1
#endif
        llpf_rbf_predict(par, NN, NL, b.nu, xn, xl, R, uf, sh_blu, fi, nz, xn1, xl1, R1);
#pragma unroll
        for (int d = 0; d < NN; ++d) xn[d] = xn1[d];
#pragma unroll
        for (int d = 0; d < NL; ++d) xl[d] = xl1[d];
#pragma unroll
        for (int d = 0; d < NP; ++d) R[d] = R1[d];
    }

    double bmax = -LLPF_INF;
    bool bad = false;
    double off = 0.0;
    WeightAcc wacc;
    uint64_t qsum = 0;
    if (MODE != MODE_PROP) {
        wacc.init();
        const double wmx = do_res ? b.log1N : (uniform ? wconst : wmax_prev);
        off = a.has_y ? (wmx + md->dg.c0) + RBF_BOUND_SLACK : wmx;
        double wv;
        if (do_res) wv = b.log1N;                                 // reset_weights!
        else if (uniform) wv = wconst;
        else { const double wr = w[i]; wv = pend ? (wr - m) - l : wr; }
        if (a.has_y) {
            double y[NY], yn[NY];
#pragma unroll
            for (int k = 0; k < NY; ++k) y[k] = a.y[(size_t)f * a.y_stride + k];
            model.measurement(xn, yn);
            wv = wv + llpf_rbf_correct(par, NL, NY, y, yn, xl, R);   // w[i] += ll, src/rbpf.jl:272
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : : "v"(wv));
            RBF_STAMP(11);
#endif
        }
        if (i >= N) wv = -LLPF_INF;                                // padding lanes carry zero weight
        w[i] = wv;
        bad = wv != wv;
        bmax = wv;
        if (a.accumulate) {     // merged schedule: exp-sums, quantum and tile sum of the new weight formed here (otherwise by k_norm)
            qsum = wacc.add(wv, off, a.K, a.need_e2 != 0);
            b.quanta_next[(size_t)f * Ns + i] = qsum;
        }
    }
    if (MODE != MODE_WEIGHT || a.has_y) {
#pragma unroll
        for (int d = 0; d < NN; ++d) st(d, xn[d]);
#pragma unroll
        for (int d = 0; d < NL; ++d) st(NN + d, xl[d]);
#pragma unroll
        for (int d = 0; d < NP; ++d) st(NN + NL + d, R[d]);
    }
    if (MODE != MODE_PROP) {
        const double r = wave_max(bmax);                        // the workgroup is one wave
        const int anybad = __ballot(bad) != 0 ? 1 : 0;
        if (a.accumulate) {
            wacc.flush_wave(b.acc + (size_t)f * ACC_WORDS, a.parity, a.need_e2 != 0);
            qsum = wave_sum_u64(qsum);     // the wave's 64 particles lie in one 1024-particle tile
        }
        if (threadIdx.x == 0) {
            if (a.accumulate && qsum) atomicAdd(reinterpret_cast<unsigned long long*>(tileq_slot(b, a.parity, f) + (i / TILE)), (unsigned long long)qsum);
            acc_max(b.acc + (size_t)f * ACC_WORDS, a.parity, r, anybad != 0);
            if (blockIdx.x == 0) {
                FilterScal* scw = b.scal + f;
                scw->off_slot[a.parity] = off;
                scw->exact_slot[a.parity] = 0;
                scw->e2v_slot[a.parity] = a.need_e2;
                scw->u_slot[a.parity] = llpf_uniform_step(sb + a.next_step, LLPF_STREAM_RESAMPLE, k0, k1);
            }
        }
    }
#if defined(LLPF_RBF_TIMING) && defined(__HIP_DEVICE_COMPILE__)
    RBF_STAMP(12);
#endif
    if (MODE != MODE_WEIGHT && blockIdx.x == 0 && threadIdx.x == 0) {
        FilterScal* scw = b.scal + f;
        scw->anc_ident_s[b.anc_slot ^ 1] = do_res ? 0 : 1;
        scw->last_resampled = do_res;
        scw->resample_count += do_res;
    }
}

// reset!(pf::RBPF), src/rbpf.jl:146-160: xl = copy(kf.d0.mu), R = copy(kf.d0.Sigma) for every particle (xn ~ d0n by k_init); any shape
__global__ __launch_bounds__(BLOCK) void k_rbfull_init(BankDev b, const ModelD* __restrict__ models) {
    const int nn = b.nx, nl = b.pad0 & 0xff, np = LLPF_RBF_NP(nl);
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    const llpf_rbf_par* par = &models[f].rbf;
    double* xc = b.xcur + (size_t)f * b.xrows * b.Ns;
    for (int d = 0; d < nl; ++d) xc[(size_t)(nn + d) * b.Ns + i] = par->xl0[d];
    for (int d = 0; d < np; ++d) xc[(size_t)(nn + nl + d) * b.Ns + i] = par->R0[d];
}
