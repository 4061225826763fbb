// kernels/smooth_fx.hpp — k_smooth_fx: f(xf[n, t]) once per backward step of the FFBS smoother (reference src/smoothing.jl:128-141).
// Its own file because it is also part of the prelude of the run-time compiled models (tools/gen_jit_prelude.py): the smoother of a
// user model needs this one kernel compiled with the user's dynamics; the draw kernel (kernels/smooth.hpp) is model independent.
template <class Model, int NX, int NY>
__global__ __launch_bounds__(BLOCK) void k_smooth_fx(BankDev b, const ModelD* __restrict__ models, SmoothArgs a) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    Model model;
    model.prepare(models, a.u, a.t);
    double xp[NX], fx[NX];
#pragma unroll
    for (int d = 0; d < NX; ++d) xp[d] = a.xf_t[i * NX + d];
    model.dynamics(xp, fx);
#pragma unroll
    for (int d = 0; d < NX; ++d) a.fx[(size_t)d * b.Ns + i] = fx[d];
}

