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

// weighted_cov accessor — reference src/filtering.jl:571-581: cov(X, ProbabilityWeights(we), 2, corrected = true) of the current
// particles: scatter matrix sum_i we_i (x_i - mu)(x_i - mu)' around mu = sum we x / sum we, times StatsBase's correction
// n / ((n - 1) sum we), n = count(we != 0).  One block per filter, fixed order (thread-strided partial sums, a shuffle tree, four
// wave sums added in order): the same bits whatever the launch geometry of the kernels that produced the weights.
// `mean`: [F][nx] sum_i we_i x_i of the same state (k_wmean); out: [F][nx * nx] row-major.
// MD: bound of the unrolled loops (8 for nx <= 8: 36 accumulators in registers; 16 above: 136, slower — an accessor, off the hot path)
template <int MD>
__global__ __launch_bounds__(BLOCK) void k_wcov(BankDev b, const double* mean, double* out) {
    __shared__ double sm_c[BLOCK / 64][MAXD * MAXD + 2];
    const int f = blockIdx.x;
    const int nx = b.nx;
    const FilterScal* sc = b.scal + f;
    const double* __restrict__ xc = b.xcur + (size_t)f * b.xrows * b.Ns;
    // sum of the exp-weights in the same fixed order (1 to rounding), then mu = mean / sum
    double s = 0.0, cnt = 0.0;
    for (int64_t i = threadIdx.x; i < b.N; i += BLOCK) {
        const double wr = b.w[(size_t)f * b.Ns + i];
        const double we = sc->uniform ? 1.0 / (double)b.N : llpf_exp_le0(wr - sc->m) * sc->inv;
        s = s + we;
        cnt = cnt + (we != 0.0 ? 1.0 : 0.0);
    }
    s = wave_sum_f64(s); cnt = wave_sum_f64(cnt);
    if ((threadIdx.x & 63) == 0) { sm_c[threadIdx.x >> 6][0] = s; sm_c[threadIdx.x >> 6][1] = cnt; }
    __syncthreads();
    double stot = sm_c[0][0], ntot = sm_c[0][1];
    for (int k = 1; k < BLOCK / 64; ++k) { stot = stot + sm_c[k][0]; ntot = ntot + sm_c[k][1]; }
    __syncthreads();
    double mu[MD];
#pragma unroll
    for (int d = 0; d < MD; ++d) mu[d] = d < nx ? mean[(size_t)f * nx + d] / stot : 0.0;
    double acc[MD * (MD + 1) / 2];
#pragma unroll
    for (int k = 0; k < MD * (MD + 1) / 2; ++k) acc[k] = 0.0;
    for (int64_t i = threadIdx.x; i < b.N; i += BLOCK) {
        const double wr = b.w[(size_t)f * b.Ns + i];
        const double we = sc->uniform ? 1.0 / (double)b.N : llpf_exp_le0(wr - sc->m) * sc->inv;
        double dv[MD];
#pragma unroll
        for (int d = 0; d < MD; ++d) dv[d] = d < nx ? xc[(size_t)d * b.Ns + i] - mu[d] : 0.0;
        int k = 0;
#pragma unroll
        for (int r = 0; r < MD; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c, ++k)
                if (r < nx) acc[k] = acc[k] + (we * dv[r]) * dv[c];
        }
    }
    {
        int k = 0;
#pragma unroll
        for (int r = 0; r < MD; ++r) {
#pragma unroll
            for (int c = 0; c <= r; ++c, ++k) {
                const double v = wave_sum_f64(acc[k]);
                if ((threadIdx.x & 63) == 0 && r < nx) sm_c[threadIdx.x >> 6][2 + r * MAXD + c] = v;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double corr = ntot / ((ntot - 1.0) * stot);
        for (int r = 0; r < nx; ++r)
            for (int c = 0; c <= r; ++c) {
                double a = sm_c[0][2 + r * MAXD + c];
                for (int k = 1; k < BLOCK / 64; ++k) a = a + sm_c[k][2 + r * MAXD + c];
                a = a * corr;
                out[(size_t)f * nx * nx + r * nx + c] = a;
                out[(size_t)f * nx * nx + c * nx + r] = a;
            }
    }
}

// bank of replicas: model descriptor 0 copied to the F - 1 others
__global__ void k_replicate_models(ModelD* models) {
    const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(models);
    uint32_t* dst = reinterpret_cast<uint32_t*>(models + 1 + blockIdx.x);
    for (int i = threadIdx.x; i < (int)(sizeof(ModelD) / 4); i += blockDim.x) dst[i] = src[i];
}
