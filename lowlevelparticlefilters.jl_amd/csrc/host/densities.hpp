// host/densities.hpp — host-side preparation of the Gaussian densities and model descriptors.  Part of capi.hip (one translation unit).
// ------------------------------------------------------------------------------------------------
// host-side preparation of the densities (same operation order as oracle/llpf_oracle.c:gauss_prepare
// in device order; transcendental = the shared deterministic log so the constant does not depend on libm)
// ------------------------------------------------------------------------------------------------
static int chol_lower(const double* S, int n, double* L) {
    memset(L, 0, sizeof(double) * MAXD * MAXD);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j <= i; ++j) {
            double acc = S[i * n + j];
            for (int k = 0; k < j; ++k) acc = acc - L[i * MAXD + k] * L[j * MAXD + k];
            if (i == j) {
                if (!(acc > 0.0)) return -1;
                L[i * MAXD + i] = llpf_sqrt(acc);
            } else {
                L[i * MAXD + j] = acc / L[j * MAXD + j];
            }
        }
    }
    return 0;
}

static int gauss_prepare(const llpf_gaussian* g, GaussD* d) {
    memset(d, 0, sizeof(*d));
    const int n = g->dim;
    if (n < 1 || n > MAXD) return -1;
    d->dim = n;
    d->kind = g->kind;
    for (int i = 0; i < n; ++i) d->mu[i] = g->mu[i];
    double logdet = 0.0;
    if (g->kind == LLPF_COV_SCAL) {
        d->scal = g->cov[0];
        if (!(d->scal > 0.0)) return -1;
        d->sqrtscal = llpf_sqrt(d->scal);
        d->invscal = 1.0 / d->scal;
        logdet = (double)n * llpf_log(d->scal);
        for (int i = 0; i < n; ++i) d->L[i * MAXD + i] = d->sqrtscal;
    } else if (g->kind == LLPF_COV_DIAG) {
        for (int i = 0; i < n; ++i) {
            d->diag[i] = g->cov[i];
            if (!(d->diag[i] > 0.0)) return -1;
            d->invdiag[i] = 1.0 / d->diag[i];
            d->sqrtdiag[i] = llpf_sqrt(d->diag[i]);
            d->L[i * MAXD + i] = d->sqrtdiag[i];
            logdet = (i == 0) ? llpf_log(d->diag[i]) : logdet + llpf_log(d->diag[i]);
        }
    } else if (g->kind == LLPF_COV_FULL) {
        if (chol_lower(g->cov, n, d->L) != 0) return -1;
        for (int i = 0; i < n; ++i) d->invLd[i] = 1.0 / d->L[i * MAXD + i];
        double dd = 0.0;
        for (int i = 0; i < n; ++i) dd = (i == 0) ? llpf_log(d->L[i * MAXD + i]) : dd + llpf_log(d->L[i * MAXD + i]);
        logdet = dd + dd;
    } else {
        return -1;
    }
    const double log2pi = llpf_log(2.0 * 3.141592653589793);
    d->c0 = -((double)n * log2pi + logdet) / 2.0;
    return 0;
}

static void gauss_cov_dense(const llpf_gaussian* g, double* S);
static int model_prepare(const llpf_model* m, ModelD* d) {
    memset(d, 0, sizeof(*d));
    d->model_id = m->model_id;
    d->nx = m->nx; d->nu = m->nu; d->ny = m->ny;
    memcpy(d->A, m->A, sizeof(d->A));
    memcpy(d->B, m->B, sizeof(d->B));
    memcpy(d->C, m->C, sizeof(d->C));
    memcpy(d->qt, m->qt, sizeof(d->qt));
    d->supersample = m->supersample;
    d->Ts = m->Ts;
    {   // quad-tank: the particle-independent coefficients, divided once here instead of by every thread of every launch — the
        // same IEEE operations in the order QuadTank::prepare used to evaluate them (examples/example_quadtank.jl:19-27)
        const double* q = m->qt;
        const double k1 = q[LLPF_QT_K1], k2 = q[LLPF_QT_K2], g = q[LLPF_QT_G];
        const double A1 = q[LLPF_QT_A1], A2 = q[LLPF_QT_A2], A3 = q[LLPF_QT_A3], A4 = q[LLPF_QT_A4];
        const double a1 = q[LLPF_QT_a1], a2 = q[LLPF_QT_a2], a3 = q[LLPF_QT_a3], a4 = q[LLPF_QT_a4];
        const double g1 = q[LLPF_QT_GAMMA1], g2 = q[LLPF_QT_GAMMA2];
        double* c = d->qtc;
        c[QTC_1A] = (-a1) / A1;
        c[QTC_1A_SW] = (-(a1 * q[LLPF_QT_A1FACTOR])) / A1;
        c[QTC_1B] = a3 / A1;
        c[QTC_1U] = (g1 * k1) / A1;
        c[QTC_2A] = (-a2) / A2;
        c[QTC_2B] = a4 / A2;
        c[QTC_2U] = (g2 * k2) / A2;
        c[QTC_3A] = (-a3) / A3;
        c[QTC_3U] = ((1.0 - g2) * k2) / A3;
        c[QTC_4A] = (-a4) / A4;
        c[QTC_4U] = ((1.0 - g1) * k1) / A4;
        c[QTC_TG] = 2.0 * g;
        const int ss = m->supersample < 1 ? 1 : m->supersample;
        const double h = m->Ts / (double)ss;
        c[QTC_H] = h;
        c[QTC_H2] = h / 2.0;
        c[QTC_H6] = h / 6.0;
    }
    // the quad-tank right-hand side takes sqrt(max(x,0) + eps) with the special-case-free llpf_sqrt_pos
    if (m->model_id == LLPF_MODEL_QUADTANK_RK4 && !(m->qt[LLPF_QT_EPS] > 1e-200)) return -9;
    if (gauss_prepare(&m->dynamics_density, &d->df)) return -1;
    if (gauss_prepare(&m->measurement_density, &d->dg)) return -2;
    if (gauss_prepare(&m->initial_density, &d->d0)) return -3;
    if (m->model_id == LLPF_MODEL_RB_LINEAR) {
        // Rao-Blackwellized model: df = R1n and d0n have dimension nxn; reset! draws xn ~ d0n and sets xl = d0l.mu exactly
        // (reference src/rbpf.jl:146-158): d0 becomes [mu_n; mu_l] + blockdiag(L_n, 0) xi
        const int nn = m->nxn, nl = m->nx - m->nxn;
        if (nn < 1 || nl < 1 || m->nx > 4) return -5;
        if (d->df.dim != nn || d->d0.dim != nn || d->dg.dim != m->ny || m->linear_noise.dim != nl || m->linear_initial.dim != nl) return -4;
        d->nxn = nn;
        d->rb_zeroAn = 1; d->rb_zeroC = 1;
        for (int r = 0; r < nn; ++r) for (int c = 0; c < nl; ++c) if (m->A[r * m->nx + nn + c] != 0.0) d->rb_zeroAn = 0;
        for (int r = 0; r < m->ny; ++r) for (int c = 0; c < nl; ++c) if (m->C[r * m->nx + nn + c] != 0.0) d->rb_zeroC = 0;
        if (!d->rb_zeroAn && nn != 1) return -6;       // L = (Al R An') / Nt is implemented for a scalar Nt
        GaussD d0n = d->d0;
        memset(&d->d0, 0, sizeof(d->d0));
        d->d0.dim = m->nx; d->d0.kind = LLPF_COV_FULL;
        for (int i = 0; i < nn; ++i) {
            d->d0.mu[i] = d0n.mu[i];
            for (int j = 0; j <= i; ++j)
                d->d0.L[i * MAXD + j] = (d0n.kind == LLPF_COV_FULL) ? d0n.L[i * MAXD + j] : (i == j ? d0n.L[i * MAXD + i] : 0.0);
        }
        for (int i = 0; i < nl; ++i) d->d0.mu[nn + i] = m->linear_initial.mu[i];
        GaussD tmp;
        if (gauss_prepare(&m->linear_noise, &tmp)) return -7;
        if (gauss_prepare(&m->linear_initial, &tmp)) return -8;
        return 0;
    }
    if (d->df.dim != m->nx || d->d0.dim != m->nx || d->dg.dim != m->ny) return -4;
    if (m->model_id == LLPF_MODEL_RB_BILINEAR) {
        // per-particle Kalman substate: constants of csrc/shared/llpf_rbfull.h (x arrays hold xn: nx = nxn)
        const int nn = m->nx, nl = m->rb.nxl, ny = m->ny, nu = m->nu;
        if (nl < 1 || nl > LLPF_RBF_MAXL || nn > LLPF_RBF_MAXN || ny > LLPF_RBF_MAXY) return -5;
        if (m->linear_noise.dim != nl || m->linear_initial.dim != nl) return -4;
        if (m->rb.fn_kind == 1 && (nn != 4 || ny != 2 || nu != 2 || !(m->qt[LLPF_QT_EPS] > 1e-200))) return -9;
        if (m->rb.fn_kind != 0 && m->rb.fn_kind != 1) return -9;
        GaussD tmp;
        if (gauss_prepare(&m->linear_noise, &tmp)) return -7;
        if (gauss_prepare(&m->linear_initial, &tmp)) return -8;
        llpf_rbf_par& q = d->rbf;
        q.nn = nn; q.nl = nl; q.ny = ny; q.nu = nu;
        bool zeroC = true;
        for (int r = 0; r < nl; ++r) for (int c = 0; c < nl; ++c) q.Al[r * nl + c] = m->rb.Al[r * nl + c];
        for (int r = 0; r < nl; ++r) for (int c = 0; c < nu; ++c) q.Bl[r * nu + c] = m->rb.Bl[r * nu + c];
        for (int r = 0; r < ny; ++r) for (int c = 0; c < nl; ++c) { q.Cl[r * nl + c] = m->rb.Cl[r * nl + c]; if (q.Cl[r * nl + c] != 0.0) zeroC = false; }
        if (zeroC) return -6;                              // C == 0 takes the reference's other branch (:274-276): use LLPF_MODEL_RB_LINEAR
        for (int k = 0; k <= nn; ++k) for (int i = 0; i < nn * nl; ++i) q.An[k][i] = m->rb.An[k][i];
        double S[64];
        gauss_cov_dense(&m->linear_noise, S);
        for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c) q.R1l[llpf_rbf_idx(r, c)] = S[r * nl + c];
        gauss_cov_dense(&m->linear_initial, S);
        for (int r = 0; r < nl; ++r) for (int c = 0; c <= r; ++c) q.R0[llpf_rbf_idx(r, c)] = S[r * nl + c];
        for (int r = 0; r < nl; ++r) q.xl0[r] = m->linear_initial.mu[r];
        gauss_cov_dense(&m->dynamics_density, S);
        for (int i = 0; i < nn * nn; ++i) q.R1n[i] = S[i];
        gauss_cov_dense(&m->measurement_density, S);
        for (int i = 0; i < ny * ny; ++i) q.R2[i] = S[i];
        q.c0y = -((double)ny * llpf_log(6.283185307179586)) / 2.0;
    }
    return 0;
}

// dense row-major covariance of a Gaussian descriptor
static void gauss_cov_dense(const llpf_gaussian* g, double* S) {
    const int n = g->dim;
    for (int i = 0; i < n * n; ++i) S[i] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (g->kind == LLPF_COV_SCAL) S[i * n + j] = (i == j) ? g->cov[0] : 0.0;
            else if (g->kind == LLPF_COV_DIAG) S[i * n + j] = (i == j) ? g->cov[i] : 0.0;
            else S[i * n + j] = g->cov[i * n + j];
        }
}
