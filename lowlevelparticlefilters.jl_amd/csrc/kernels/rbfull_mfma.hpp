// kernels/rbfull_mfma.hpp — k_rbfull_mfma: the per-particle Kalman recursion of BASELINE config C5 on the matrix unit.
// Part of kernels.hip (one translation unit, namespace llpf).
// ------------------------------------------------------------------------------------------------
// Shape (nxn, nxl, ny) = (4, 8, 2), MODE_PROP_WEIGHT with a measurement: predict! (reference src/rbpf.jl:206-221: Nt = An R An' +
// R1n, L = Al R An' / Nt, R1 = Al R Al' + R1l - L Nt L', the means) followed by correct! (:259-263 -> src/filtering.jl:100-128:
// S = sym(C R C') + R2, K = R C' / chol(S), R = sym((I - K C) R), ll) for every particle, each with its own 8x8 covariance.
//
// k_rbfull (one particle per thread, every matrix in registers) needs 440 registers: one wave per SIMD.  Here the 8x8 / 4x8 / 2x8
// contractions run on v_mfma_f64_4x4x4_4b_f64, which multiplies FOUR independent 4x4 blocks per instruction with ONE matrix
// element per lane (lane = 16 * row + 4 * blk + col): an 8x8 covariance is 4 registers per lane instead of 36 per thread.  The
// rate is the vector unit's (78.6 TFLOP/s either way on gfx950); what changes is the register footprint.
//   * One wave = 64 particles.  Thread-per-particle ("scalar") phases do what has no matrix shape — gather, RK4, Philox /
//     Box-Muller, the 4x4 and 2x2 Cholesky factors and triangular solves, exp / log — and matrix phases loop over the 16 groups
//     of four particles (blk = particle within the group).  The two layouts meet in LDS (88 doubles per particle).
//   * Bit-identity with csrc/shared/llpf_rbfull_body.h (which the oracle runs) is kept: the instruction's k-sum is the chain
//     fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,c)))) (tools/mfma_f64_probe.hip), i.e. the header's own order: first product
//     onto 0, summation index ascending, 4-blocks of the index chained through the accumulator.  A block X held in "N layout"
//     (X[r][c] at lane 16r + 4blk + c) is X as the B operand and X' as the A operand, so every transpose below is free.
//     (fma(a, b, 0) and a * b differ only in the sign of an exact zero.)
//   * Symmetric matrices live as packed lower triangles; entries of a diagonal block above the diagonal are read back from
//     their mirror, exactly as the header reads R[idx(r, c)].
// ------------------------------------------------------------------------------------------------
constexpr int RBM_STRIDE = 89;      // doubles per particle of LDS scratch (odd: spreads the banks)
constexpr int RBM_R = 0;            // [36] packed covariance: R, then R1, then the posterior
constexpr int RBM_B = 36;           // [32] xn (4) + xl (8)  ->  G = Al (An R)' (8x4)  ->  W (8x4)  ->  K (8x2)
constexpr int RBM_A = 68;           // [16] Nt (4x4)  ->  C R (2x8)
constexpr int RBM_C = 84;           // [4]  An xl     ->  raw = (C R) C' (2x2)

#ifndef RBM_GROUP_UNROLL
#define RBM_GROUP_UNROLL 4          /* groups of four particles in flight per wave: independent MFMA / LDS chains */
#endif
#define RBM_STR2(x) #x
#define RBM_STR(x) RBM_STR2(x)
#define RBM_UNROLL_GROUPS _Pragma(RBM_STR(unroll RBM_GROUP_UNROLL))
DEV double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <class Model>
__global__ __launch_bounds__(64) void k_rbfull_mfma(BankDev b, const ModelD* __restrict__ models, const FilterScal* scal, StepArgs a) {
    constexpr int NN = 4, NL = 8, NY = 2, NP = 36, ROWS = NN + NL + NP;
    __shared__ double lds[64 * RBM_STRIDE];
    const int f = blockIdx.y;
    const ModelD* md = models + f;
    const FilterScal* sc = scal + f;
    if (run_is_stopped(b, a.k)) return;
    if (a.only_fallback ? !sc->fallback : (sc->fallback != 0)) return;
    const int do_res = sc->do_resample;
    const int uniform = sc->uniform, pend = sc->norm_pending;
    const double m = sc->m, l = sc->l, wconst = sc->wconst;
    const uint32_t k0 = sc->k0, k1 = sc->k1, sb = sc->step_base;
    const int64_t Ns = b.Ns, N = b.N;
    const double* __restrict__ xc = b.xcur + (size_t)f * ROWS * Ns;
    double* __restrict__ xo = b.xnext + (size_t)f * ROWS * Ns;
    double* w = b.w + (size_t)f * Ns;
    const llpf_rbf_par* par = &md->rbf;
    const int lane = threadIdx.x;
    const int hi = lane >> 4, blk = (lane >> 2) & 3, lo = lane & 3;
    double* mine = lds + lane * RBM_STRIDE;                 // this lane's particle in the scalar phases

    // ---- constants in matrix layout (one element per lane) ----
    double AlA[2][2], AnA0[2], AnAk[4][2], CpA[2], R1lN[2][2], R1nN;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            AlA[r][c] = par->Al[(4 * r + lo) * NL + 4 * c + hi];                       // A layout of block (r, c) of Al
            R1lN[r][c] = par->R1l[llpf_rbf_idx(4 * r + hi, 4 * c + lo)];               // N layout of block (r, c) of R1l
        }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        AnA0[c] = par->An[0][lo * NL + 4 * c + hi];                                    // A layout of block c of An[k]
#pragma unroll
        for (int k = 0; k < 4; ++k) AnAk[k][c] = par->An[1 + k][lo * NL + 4 * c + hi];
        CpA[c] = lo < NY ? par->Cl[lo * NL + 4 * c + hi] : 0.0;                        // A layout of [C; 0] (rows 2, 3 zero)
    }
    R1nN = par->R1n[hi * NN + lo];

    // ================= scalar phase 0: gather, f_n, noise =================
    Model model;
    model.prepare(md, a.u, a.t_prop);
    const int64_t i = (int64_t)blockIdx.x * 64 + lane;
    const int64_t src = do_res ? (int64_t)b.anc[(size_t)f * Ns + i] : i;
    double xn[NN], xl[NL], fi[NN], nz[NN];
#pragma unroll
    for (int d = 0; d < NN; ++d) { xn[d] = xc[(size_t)d * Ns + src]; mine[RBM_B + d] = xn[d]; }
#pragma unroll
    for (int d = 0; d < NL; ++d) { xl[d] = xc[(size_t)(NN + d) * Ns + src]; mine[RBM_B + NN + d] = xl[d]; }
#pragma unroll
    for (int d = 0; d < NP; ++d) mine[RBM_R + d] = xc[(size_t)(NN + NL + d) * Ns + src];
    {
        double xi[NN];
        model.dynamics(xn, fi);
        llpf_normals((uint32_t)i, sb + a.step, LLPF_STREAM_DYNAMICS, k0, k1, NN, xi);
        gauss_sample<NN>(md->df, xi, nz);
    }
    __syncthreads();

    // packed index of element (hi, lo) of block (r, c) of a symmetric 8x8 matrix
    int pidx[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) pidx[r][c] = llpf_rbf_idx(4 * r + hi, 4 * c + lo);

    // ================= matrix phase 1: An(xn), (An R)', Nt, G = Al (An R)', An xl =================
RBM_UNROLL_GROUPS
    for (int g = 0; g < 16; ++g) {
        double* P = lds + (4 * g + blk) * RBM_STRIDE;       // the particle this lane works for in group g
        double Rn[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) Rn[r][c] = P[RBM_R + pidx[r][c]];
        double An[2];
        {
            const double x0 = P[RBM_B + 0], x1 = P[RBM_B + 1], x2 = P[RBM_B + 2], x3 = P[RBM_B + 3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {                   // An = An[0] + sum_k xn[k] An[1+k]   (A layout)
                double v = AnA0[c];
                v = llpf_fma(x0, AnAk[0][c], v);
                v = llpf_fma(x1, AnAk[1][c], v);
                v = llpf_fma(x2, AnAk[2][c], v);
                v = llpf_fma(x3, AnAk[3][c], v);
                An[c] = v;
            }
        }
        const double xlB0 = lo == 0 ? P[RBM_B + NN + hi] : 0.0, xlB1 = lo == 0 ? P[RBM_B + NN + 4 + hi] : 0.0;
        // M1[r] = block r of R An' = (An R)':  sum_c R(r, c) An_c'   [A operand R(r, c) = N register of R(c, r); B = An_c in A layout]
        double M1[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) { M1[r] = mfma4(Rn[0][r], An[0], 0.0); M1[r] = mfma4(Rn[1][r], An[1], M1[r]); }
        // Nt = (An R) An' + R1n:  A operand M1[c] (N register = its transpose, a block of An R); B = An_c
        double Nt = mfma4(M1[0], An[0], 0.0);
        Nt = mfma4(M1[1], An[1], Nt);
        Nt = Nt + R1nN;
        // G[r] = block r of Al (An R)':  sum_c Al(r, c) M1[c]
        double G[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) { G[r] = mfma4(AlA[r][0], M1[0], 0.0); G[r] = mfma4(AlA[r][1], M1[1], G[r]); }
        // An xl (column 0 of the product with [xl 0 0 0])
        double Ax = mfma4(An[0], xlB0, 0.0);
        Ax = mfma4(An[1], xlB1, Ax);
        // hand over to the scalar phase (these lanes read xn / xl of the same particles above: program order inside the wave)
        P[RBM_A + hi * 4 + lo] = Nt;
        P[RBM_B + (hi) * 4 + lo] = G[0];                    // G[row][i], row = hi (block 0), 4 + hi (block 1)
        P[RBM_B + (4 + hi) * 4 + lo] = G[1];
        if (lo == 0) P[RBM_C + hi] = Ax;
    }
    __syncthreads();

    // ================= scalar phase 1: z, xn', Cholesky of Nt, v, W, xl' (header lines of RBF_(predict)) =================
    double xn1[NN], xl1[NL];
    {
        double Lc[NN * NN], invd[NN], v[NN], dz[NN], W[NL * NN];
#pragma unroll
        for (int r = 0; r < NN; ++r) {
            const double ax = mine[RBM_C + r];
            const double z = ax + nz[r];
            xn1[r] = fi[r] + z;
            dz[r] = z - ax;
        }
#pragma unroll
        for (int ii = 0; ii < NN; ++ii) {                    // Nt = Lc Lc'
#pragma unroll
            for (int j = 0; j <= ii; ++j) {
                double acc = mine[RBM_A + ii * 4 + j];
#pragma unroll
                for (int k = 0; k < j; ++k) acc = llpf_fma(-Lc[ii * NN + k], Lc[j * NN + k], acc);
                if (ii == j) {
                    const double d = llpf_sqrt(acc);
                    Lc[ii * NN + ii] = d;
                    invd[ii] = 1.0 / d;
                } else {
                    Lc[ii * NN + j] = acc * invd[j];
                }
            }
        }
#pragma unroll
        for (int ii = 0; ii < NN; ++ii) {                    // Lc v = dz
            double acc = dz[ii];
#pragma unroll
            for (int q = 0; q < ii; ++q) acc = llpf_fma(-Lc[ii * NN + q], v[q], acc);
            v[ii] = acc * invd[ii];
        }
#pragma unroll
        for (int r = 0; r < NL; ++r) xl1[r] = par->Al[r * NL] * xl[0];                 // Al xl + Bl u for all rows
#pragma unroll
        for (int c = 1; c < NL; ++c) {
#pragma unroll
            for (int r = 0; r < NL; ++r) xl1[r] = llpf_fma(par->Al[r * NL + c], xl[c], xl1[r]);
        }
        const int nu = b.nu;
        if (nu > 0) {
#pragma unroll
            for (int r = 0; r < NL; ++r) {
                double b2 = par->Bl[r * nu] * a.u[0];
                for (int c = 1; c < nu; ++c) b2 = llpf_fma(par->Bl[r * nu + c], a.u[c], b2);
                xl1[r] = xl1[r] + b2;
            }
        }
#pragma unroll
        for (int r = 0; r < NL; ++r) {
#pragma unroll
            for (int ii = 0; ii < NN; ++ii) {                // Lc w = g : row r of W
                double acc = mine[RBM_B + r * 4 + ii];
#pragma unroll
                for (int q = 0; q < ii; ++q) acc = llpf_fma(-Lc[ii * NN + q], W[r * NN + q], acc);
                W[r * NN + ii] = acc * invd[ii];
            }
            double s = W[r * NN] * v[0];                    // xl1 = (Al xl + Bl u) + W (Lc^-1 dz)
#pragma unroll
            for (int j = 1; j < NN; ++j) s = llpf_fma(W[r * NN + j], v[j], s);
            xl1[r] = xl1[r] + s;
        }
#pragma unroll
        for (int q = 0; q < NL * NN; ++q) mine[RBM_B + q] = W[q];
    }
    __syncthreads();

    // ================= matrix phase 2: R1 = Al R Al' + R1l - W W', then C R1, (C R1) C' =================
RBM_UNROLL_GROUPS
    for (int g = 0; g < 16; ++g) {
        double* P = lds + (4 * g + blk) * RBM_STRIDE;
        double Rn[2][2], WA[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) Rn[r][c] = P[RBM_R + pidx[r][c]];
        WA[0] = P[RBM_B + lo * 4 + hi];                     // A layout of block r of W: W[4r + lo][hi]
        WA[1] = P[RBM_B + (4 + lo) * 4 + hi];
        // U(c, r) = block (c, r) of R Al' = (Al R)':  sum_q R(c, q) Al(r, q)'   [A operand R(c, q) = N register Rn[q][c]; B = Al(r, q) in A layout]
        double U[2][2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r) { U[c][r] = mfma4(Rn[0][c], AlA[r][0], 0.0); U[c][r] = mfma4(Rn[1][c], AlA[r][1], U[c][r]); }
        // lower blocks of Al R Al', W W', R1
        double R1b[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                double p = mfma4(U[0][r], AlA[c][0], 0.0);  // sum_q (Al R)(r, q) Al(c, q)':  A operand = N register of U(q, r)
                p = mfma4(U[1][r], AlA[c][1], p);
                const double s = mfma4(WA[r], WA[c], 0.0);
                R1b[r][c] = (p + R1lN[r][c]) - s;
            }
        // lower triangle back to the packed store (all reads of this particle's R are done)
        if (hi >= lo) { P[RBM_R + pidx[0][0]] = R1b[0][0]; P[RBM_R + pidx[1][1]] = R1b[1][1]; }
        P[RBM_R + pidx[1][0]] = R1b[1][0];
        // reload in N layout with the mirrored upper entries
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) Rn[r][c] = P[RBM_R + pidx[r][c]];
        // CR[c] = block c of [C; 0] R1 (N layout: CR[i][4c + lo] at row hi = i);  V[c] = block c of R1 [C; 0]' = its transpose
        double CR[2], V[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            CR[c] = mfma4(CpA[0], Rn[0][c], 0.0); CR[c] = mfma4(CpA[1], Rn[1][c], CR[c]);
            V[c] = mfma4(Rn[0][c], CpA[0], 0.0);  V[c] = mfma4(Rn[1][c], CpA[1], V[c]);
        }
        // raw = (C R1) C':  A operand = N register of V[c] (its transpose, a block of C R1); B = [C; 0] block c in A layout
        double raw = mfma4(V[0], CpA[0], 0.0);
        raw = mfma4(V[1], CpA[1], raw);
        if (hi < NY) { P[RBM_A + hi * NL + lo] = CR[0]; P[RBM_A + hi * NL + 4 + lo] = CR[1]; }
        if (hi < NY && lo < NY) P[RBM_C + hi * NY + lo] = raw;
    }
    __syncthreads();

    // ================= scalar phase 2: innovation, S, its Cholesky factor, gain, ll (header lines of RBF_(correct)) =================
    double wv;
    {
        double y[NY], yn[NY], e[NY], Lc[NY * NY], invd[NY], K[NL * NY];
        y[0] = a.y[0]; y[1] = a.y[1];
        model.measurement(xn1, yn);
#pragma unroll
        for (int ii = 0; ii < NY; ++ii) {
            double acc = par->Cl[ii * NL] * xl1[0];
#pragma unroll
            for (int c = 1; c < NL; ++c) acc = llpf_fma(par->Cl[ii * NL + c], xl1[c], acc);
            e[ii] = (y[ii] - yn[ii]) - acc;
        }
        double ldet = 0.0;
#pragma unroll
        for (int ii = 0; ii < NY; ++ii) {                    // S = 0.5 (raw + raw') + R2 = Lc Lc'
#pragma unroll
            for (int j = 0; j <= ii; ++j) {
                double acc = 0.5 * (mine[RBM_C + ii * NY + j] + mine[RBM_C + j * NY + ii]) + par->R2[ii * NY + j];
#pragma unroll
                for (int k = 0; k < j; ++k) acc = llpf_fma(-Lc[ii * NY + k], Lc[j * NY + k], acc);
                if (ii == j) {
                    const double d = llpf_sqrt(acc);
                    Lc[ii * NY + ii] = d;
                    invd[ii] = 1.0 / d;
                    ldet = ldet + llpf_log(d);
                } else {
                    Lc[ii * NY + j] = acc * invd[j];
                }
            }
        }
        double quad = 0.0;
        {
            double z[NY];
#pragma unroll
            for (int ii = 0; ii < NY; ++ii) {                // Lc z = e
                double acc = e[ii];
#pragma unroll
                for (int q = 0; q < ii; ++q) acc = llpf_fma(-Lc[ii * NY + q], z[q], acc);
                z[ii] = acc * invd[ii];
                quad = llpf_fma(z[ii], z[ii], quad);
            }
        }
#pragma unroll
        for (int r = 0; r < NL; ++r) {                       // row r of K solves k S = (R C')[r,:]
            double t[NY];
#pragma unroll
            for (int ii = 0; ii < NY; ++ii) {
                double acc = mine[RBM_A + ii * NL + r];
#pragma unroll
                for (int q = 0; q < ii; ++q) acc = llpf_fma(-Lc[ii * NY + q], t[q], acc);
                t[ii] = acc * invd[ii];
            }
#pragma unroll
            for (int ii = NY - 1; ii >= 0; --ii) {
                double acc = t[ii];
#pragma unroll
                for (int q = ii + 1; q < NY; ++q) acc = llpf_fma(-Lc[q * NY + ii], K[r * NY + q], acc);
                K[r * NY + ii] = acc * invd[ii];
            }
        }
#pragma unroll
        for (int r = 0; r < NL; ++r) {
            double acc = K[r * NY] * e[0];
#pragma unroll
            for (int ii = 1; ii < NY; ++ii) acc = llpf_fma(K[r * NY + ii], e[ii], acc);
            xl1[r] = xl1[r] + acc;
        }
#pragma unroll
        for (int q = 0; q < NL * NY; ++q) mine[RBM_B + q] = K[q];
        const double ll = (par->c0y - ldet) - 0.5 * quad;
        if (do_res) wv = b.log1N;                            // reset_weights!
        else if (uniform) wv = wconst;
        else { const double wr = w[i]; wv = pend ? (wr - m) - l : wr; }
        wv = wv + ll;                                        // w[i] += ll, src/rbpf.jl:272
        if (i >= N) wv = -LLPF_INF;
    }
    __syncthreads();

    // ================= matrix phase 3: posterior covariance 0.5 ((R - K CR) + (R - K CR)') =================
RBM_UNROLL_GROUPS
    for (int g = 0; g < 16; ++g) {
        double* P = lds + (4 * g + blk) * RBM_STRIDE;
        double Rn[2][2], KnA[2], CRn[2];
        Rn[0][0] = P[RBM_R + pidx[0][0]]; Rn[1][0] = P[RBM_R + pidx[1][0]]; Rn[1][1] = P[RBM_R + pidx[1][1]];
        KnA[0] = hi < NY ? -P[RBM_B + lo * NY + hi] : 0.0;               // A layout of block r of -K: -K[4r + lo][hi], columns 2, 3 zero
        KnA[1] = hi < NY ? -P[RBM_B + (4 + lo) * NY + hi] : 0.0;
        CRn[0] = hi < NY ? P[RBM_A + hi * NL + lo] : 0.0;                // N layout of block c of [CR; 0]
        CRn[1] = hi < NY ? P[RBM_A + hi * NL + 4 + lo] : 0.0;
        double out[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c <= r; ++c) {
                const double d1 = mfma4(KnA[r], CRn[c], Rn[r][c]);       // R[r,c] - sum_i K[r,i] CR[i,c]
                const double d2 = mfma4(CRn[r], KnA[c], Rn[r][c]);       // R[r,c] - sum_i K[c,i] CR[i,r]   (A operand: N register = CR block transposed)
                out[r][c] = 0.5 * (d1 + d2);
            }
        if (hi >= lo) { P[RBM_R + pidx[0][0]] = out[0][0]; P[RBM_R + pidx[1][1]] = out[1][1]; }
        P[RBM_R + pidx[1][0]] = out[1][0];
    }
    __syncthreads();

    // ================= scalar phase 3: stores, block maximum =================
    w[i] = wv;
#pragma unroll
    for (int d = 0; d < NN; ++d) xo[(size_t)d * Ns + i] = xn1[d];
#pragma unroll
    for (int d = 0; d < NL; ++d) xo[(size_t)(NN + d) * Ns + i] = xl1[d];
#pragma unroll
    for (int d = 0; d < NP; ++d) xo[(size_t)(NN + NL + d) * Ns + i] = mine[RBM_R + d];
    const bool bad = wv != wv;
    const double r = wave_max(wv);
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    if (lane == 0) {
        acc_max(b.acc + (size_t)f * ACC_WORDS, a.parity, r, anybad != 0);
        if (blockIdx.x == 0) {
            const double wmx = do_res ? b.log1N : (uniform ? wconst : sc->wmax);
            FilterScal* scw = b.scal + f;
            scw->off_slot[a.parity] = (wmx + md->dg.c0) + RBF_BOUND_SLACK;
            scw->exact_slot[a.parity] = 0;
            scw->e2v_slot[a.parity] = a.need_e2;
            scw->u_slot[a.parity] = llpf_uniform_step(sb + a.next_step, LLPF_STREAM_RESAMPLE, k0, k1);
            scw->anc_ident_s[b.anc_slot ^ 1] = do_res ? 0 : 1;
            scw->last_resampled = do_res;
            scw->resample_count += do_res;
        }
    }
}
