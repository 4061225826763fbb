// kernels/access.hpp — accessor kernels.  Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// accessors
// ------------------------------------------------------------------------------------------------
// weights(pf) / expweights(pf): materialise the lazily-normalised values
__global__ __launch_bounds__(BLOCK) void k_materialize(BankDev b, double* w_out, double* we_out) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    const FilterScal* sc = b.scal + f;
    const double wr = b.w[(size_t)f * b.Ns + i];
    double wv, we;
    if (sc->uniform) {
        wv = sc->wconst;
        we = 1.0 / (double)b.N;
    } else {
        wv = sc->norm_pending ? (wr - sc->m) - sc->l : wr;
        we = llpf_exp_le0(wr - sc->m) * sc->inv;
    }
    if (w_out) w_out[(size_t)f * b.N + i] = wv;
    if (we_out) we_out[(size_t)f * b.N + i] = we;
}

// w[] <- the values it stands for (uniform constant / lazily normalised / as stored); padding lanes -Inf.  The
// host clears the `uniform` / `norm_pending` flags afterwards.
__global__ __launch_bounds__(BLOCK) void k_bake_weights(BankDev b) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    const FilterScal* sc = b.scal + f;
    double* w = b.w + (size_t)f * b.Ns;
    double wv = -LLPF_INF;
    if (i < b.N) wv = sc->uniform ? sc->wconst : (sc->norm_pending ? (w[i] - sc->m) - sc->l : w[i]);
    w[i] = wv;
}

__global__ __launch_bounds__(BLOCK) void k_soa2aos(BankDev b, const double* __restrict__ xsrc, double* dst) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    for (int d = 0; d < b.nx; ++d)
        dst[((size_t)f * b.N + i) * b.nx + d] = xsrc[((size_t)f * b.xrows + d) * b.Ns + i];
}
__global__ __launch_bounds__(BLOCK) void k_aos2soa(BankDev b, const double* __restrict__ src, double* xdst) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.Ns) return;
    for (int d = 0; d < b.nx; ++d)
        xdst[((size_t)f * b.xrows + d) * b.Ns + i] = (i < b.N) ? src[((size_t)f * b.N + i) * b.nx + d] : 0.0;
}
__global__ __launch_bounds__(BLOCK) void k_anc64(BankDev b, int64_t* dst) {
    const int f = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= b.N) return;
    const FilterScal* sc = b.scal + f;
    dst[(size_t)f * b.N + i] = sc->anc_ident_s[b.anc_slot] ? i : (int64_t)b.anc[(size_t)f * b.Ns + i];
}

// weighted_mean(pf) accessor — reference src/filtering.jl:541-549,568.  One block per filter, fixed order.
__global__ __launch_bounds__(BLOCK) void k_wmean(BankDev b, double* out) {
    __shared__ double sm_x[BLOCK / 64][MAXD];
    const int f = blockIdx.x;
    const FilterScal* sc = b.scal + f;
    const double* __restrict__ xc = b.xcur + (size_t)f * b.xrows * b.Ns;
    double acc[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) acc[d] = 0.0;
    for (int64_t i = threadIdx.x; i < b.N; i += BLOCK) {
        const double wr = b.w[(size_t)f * b.Ns + i];
        const double we = sc->uniform ? 1.0 / (double)b.N : llpf_exp_le0(wr - sc->m) * sc->inv;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < b.nx) acc[d] = acc[d] + xc[(size_t)d * b.Ns + i] * we;
    }
#pragma unroll
    for (int d = 0; d < MAXD; ++d) acc[d] = wave_sum_f64(acc[d]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < MAXD; ++d) sm_x[threadIdx.x >> 6][d] = acc[d];
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int d = 0; d < b.nx; ++d) {
            double a = sm_x[0][d];
            for (int k = 1; k < BLOCK / 64; ++k) a = a + sm_x[k][d];
            out[(size_t)f * b.nx + d] = a;
        }
}

// bank of replicas: model descriptor 0 copied to the F - 1 others
__global__ void k_replicate_models(ModelD* models) {
    const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(models);
    uint32_t* dst = reinterpret_cast<uint32_t*>(models + 1 + blockIdx.x);
    for (int i = threadIdx.x; i < (int)(sizeof(ModelD) / 4); i += blockDim.x) dst[i] = src[i];
}
