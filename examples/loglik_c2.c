/* loglik_c2.c — the drop-in boundary from plain C: loglik(pf, u, y) of the reference's end-to-end test system
 * (test/runtests.jl:255-266) through include/llpf.h only.  Links against libllpf_hip.so; no Python, no torch.
 *
 *   cc -O2 -I include examples/loglik_c2.c -L lowlevelparticlefilters.jl_amd -lllpf_hip -lm -o examples/loglik_c2
 *   LD_LIBRARY_PATH=lowlevelparticlefilters.jl_amd examples/loglik_c2 [N] [T] [seed]
 *
 * Prints the log-likelihood, the device time of the run and particle-steps/s as one JSON line. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "llpf.h"

static uint64_t lcg(uint64_t* s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return *s >> 11; }
static double unif(uint64_t* s) { return ((double)lcg(s) + 0.5) / 9007199254740992.0; }
static double randn(uint64_t* s) { return sqrt(-2.0 * log(unif(s))) * cos(6.283185307179586 * unif(s)); }

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 100000;
    const int64_t T = argc > 2 ? atoll(argv[2]) : 200;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 1;

    llpf_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.filter_kind = LLPF_PARTICLE_FILTER;
    cfg.n_particles = N;
    cfg.resampling_strategy = LLPF_RESAMPLE_SYSTEMATIC;
    cfg.device = 0;
    cfg.resample_threshold = 0.1;                 /* the reference's default */
    cfg.seed = seed;
    llpf_model* m = &cfg.model;
    m->model_id = LLPF_MODEL_LINEAR_GAUSSIAN;
    m->nx = 2; m->nu = 1; m->ny = 1; m->supersample = 1; m->Ts = 1.0;
    const double A[4] = {0.97043, -0.097368, 0.09736, 0.970437}, B[2] = {0.1, 0.0}, C[2] = {0.0, 1.0};
    memcpy(m->A, A, sizeof(A)); memcpy(m->B, B, sizeof(B)); memcpy(m->C, C, sizeof(C));
    m->dynamics_density.dim = 2; m->dynamics_density.kind = LLPF_COV_SCAL; m->dynamics_density.cov[0] = 0.01;   /* N(0, 0.1^2 I) */
    m->measurement_density.dim = 1; m->measurement_density.kind = LLPF_COV_DIAG; m->measurement_density.cov[0] = 1.0;
    m->initial_density.dim = 2; m->initial_density.kind = LLPF_COV_SCAL; m->initial_density.cov[0] = 4.0;
    m->initial_density.mu[0] = 0.3; m->initial_density.mu[1] = -0.5;

    /* simulate(pf, T, du): x1 = mean(d0), y = C x + e, x' = A x + B u + w */
    double* U = (double*)malloc(sizeof(double) * T);
    double* Y = (double*)malloc(sizeof(double) * T);
    uint64_t s = 12345;
    double x0 = 0.3, x1 = -0.5;
    for (int64_t t = 0; t < T; ++t) {
        U[t] = randn(&s);
        Y[t] = x1 + randn(&s);
        const double n0 = A[0] * x0 + A[1] * x1 + B[0] * U[t] + 0.1 * randn(&s);
        const double n1 = A[2] * x0 + A[3] * x1 + 0.1 * randn(&s);
        x0 = n0; x1 = n1;
    }

    llpf_filter* pf = NULL;
    if (llpf_create(&cfg, &pf) != LLPF_OK) { fprintf(stderr, "llpf_create: %s\n", llpf_last_error()); return 2; }
    double ll = 0.0, ms = 0.0;
    int rc = llpf_reset(pf);                                        /* loglik = reset!, then sum of update! at t = index*Ts */
    if (rc == LLPF_OK) rc = llpf_run(pf, U, Y, T, 1.0, &ll, NULL);
    if (rc != LLPF_OK) { fprintf(stderr, "llpf_run: %s\n", llpf_last_error()); return 3; }
    llpf_last_run_ms(pf, &ms);
    printf("{\"N\": %lld, \"T\": %lld, \"loglik\": %.17g, \"device_ms\": %.3f, \"particle_steps_per_s\": %.4g}\n",
           (long long)N, (long long)T, ll, ms, (double)N * (double)T / (ms * 1e-3));
    llpf_destroy(pf);
    free(U); free(Y);
    return 0;
}
