// kernels/init.hpp — k_init (reset!).  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// k_init — reset!: x_i = mu0 + L0 xi_i  (reference src/filtering.jl:4-14, src/PFtypes.jl:66)
// ------------------------------------------------------------------------------------------------
template <int NX>
__global__ __launch_bounds__(BLOCK) void k_init(BankDev b, const ModelD* __restrict__ models,
                                                 const FilterScal* __restrict__ scal, uint32_t step, int init_anc) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    const ModelD* md = models + f;
    double xi[NX], x0[NX];
    llpf_normals((uint32_t)i, step, LLPF_STREAM_INIT, scal[f].k0, scal[f].k1, NX, xi);
    gauss_sample<NX>(md->d0, xi, x0);
    double* xc = b.xcur + (size_t)f * b.xrows * b.Ns;      // NX rows are drawn; a Rao-Blackwellized particle has more (k_rbfull_init)
#pragma unroll
    for (int d = 0; d < NX; ++d) xc[(size_t)d * b.Ns + i] = x0[d];
    b.w[(size_t)f * b.Ns + i] = -LLPF_INF;
    if (init_anc) b.anc[(size_t)f * b.Ns + i] = (i < b.N) ? (int32_t)i : 0;
}
