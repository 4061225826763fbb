"""Writes tests/golden/ref_inputs_<case>.npz — everything julia/make_reference_fixtures.jl needs to run the REAL reference
(LowLevelParticleFilters.jl) on exactly the inputs and random draws this engine consumes, so that the reference's own outputs can be
frozen as tests/golden/ref_<case>.npz and held against the engine and both oracle orders (tests/test_reference_fixtures.py).

Per case: the system (matrices / quad-tank constants as the reference example writes them), N, T, resample_threshold, Ts, t_index0,
U [T, nu], Y [T, ny] (NaN row = missing), and the Philox draws of seed SEED in the order the reference consumes them:
    xi_reset [N, nx]        standard normals of reset! (the handle's reset number 1; the constructor took number 0)
    xi_dyn   [T, N, nx]     standard normals of predict! number k: particle-major, the order `rand!(pf.rng, d, noise)` is called in
                            propagate_particles! (src/PFtypes.jl:130-136, ext/LowLevelParticleFiltersDistributionsExt.jl:83-93)
    u_res    [T]            the uniform of the systematic resample of predict! number k: `rand()` of src/resample.jl:23
This file is data only (inputs); it runs here, without Julia:   python tests/golden/make_reference_inputs.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import models as M
import oracle_binding as ob
from llpf_amd import _structs as S

SEED = 11
STREAM_INIT, STREAM_DYNAMICS = 0, 1


def cases():
    """name -> case.  `family`: pf (ParticleFilter) | apf (AdvancedParticleFilter) | aux (AuxiliaryParticleFilter over the ParticleFilter,
    src/filtering.jl:195-217) | rbpf (RBPF with a state-dependent coupling An(xn): one covariance per particle, src/rbpf.jl:163-283)."""
    import rbfull_models as RM
    lg = M.lg_test_model()
    _, U, Y = M.simulate_lg(lg, 60, seed=5)
    Y = Y.copy()
    Y[23] = np.nan                                       # one missing measurement (any(ismissing, y) && return, src/PFtypes.jl:109)
    qt = M.quadtank_model()
    Uq, Yq = M.quadtank_data(40, seed=3)
    rb = RM.linear_case(2, 3, 2, seed=4)[0]              # 2 nonlinear + 3 linear states, An(xn) = An0 + sum_k xn[k] An_k, nu = 1, ny = 2
    Ur, Yr = RM.simulate_io(rb, 40, seed=6)
    base = dict(model=lg, kind=S.PARTICLE_FILTER, family="pf", strategy=S.RESAMPLE_SYSTEMATIC, N=200, T=60, thr=0.5, t_index0=0.0, U=U, Y=Y)
    return {
        "lg": dict(base),
        "quadtank": dict(model=qt, kind=S.ADVANCED_PARTICLE_FILTER, family="apf", strategy=S.RESAMPLE_SYSTEMATIC, N=200, T=40, thr=0.5, t_index0=485.0, U=Uq, Y=Yq),   # crosses t > 500
        # round 5: the rest of the path, so that ONE Julia session pins everything
        "lg_stratified": dict(base, strategy=S.RESAMPLE_STRATIFIED),            # src/resample.jl:38-61: one rand() per output
        "lg_residual": dict(base, strategy=S.RESAMPLE_RESIDUAL),                # :63-117: deterministic copies, then rand() per remaining output
        "lg_thr1": dict(base, thr=1.0),                                         # shouldresample at every step (:5-10)
        "aux_lg": dict(base, family="aux", thr=0.1),                            # the auxiliary filter resamples at every predict! whatever the threshold
        "rbpf": dict(model=rb, kind=S.PARTICLE_FILTER, family="rbpf", strategy=S.RESAMPLE_SYSTEMATIC, N=200, T=40, thr=0.5, t_index0=0.0, U=Ur, Y=Yr),
    }


def draws(case):
    """xi_reset [N, nd], xi_dyn [T, N, nd] (nd = the dimension the noise is drawn in: nx, or nxn for the RBPF, whose Kalman states draw
    nothing), u_res [T] (systematic) or [T, N] (stratified / residual: the uniform of OUTPUT m; the residual resampler reads only
    those of the outputs left after the deterministic copies, src/resample.jl:105-106)"""
    nd, N, T = case["model"].nx, case["N"], case["T"]
    strat = case.get("strategy", S.RESAMPLE_SYSTEMATIC)
    xi_reset = ob.normals(SEED, 1, STREAM_INIT, nd, N)
    xi_dyn = np.stack([ob.normals(SEED, k, STREAM_DYNAMICS, nd, N) for k in range(T)])
    if strat == S.RESAMPLE_SYSTEMATIC:
        u_res = np.array([ob.resample_uniforms(strat, N, SEED, k)[0] for k in range(T)])
    else:
        u_res = np.stack([ob.resample_uniforms(strat, N, SEED, k) for k in range(T)])
    return xi_reset, xi_dyn, u_res


def main():
    for name, c in cases().items():
        m = c["model"]
        xi_reset, xi_dyn, u_res = draws(c)
        out = dict(N=np.int64(c["N"]), T=np.int64(c["T"]), nx=np.int64(m.nx), nu=np.int64(m.nu), ny=np.int64(m.ny), thr=np.float64(c["thr"]), Ts=np.float64(m.Ts),
                   t_index0=np.float64(c["t_index0"]), seed=np.int64(SEED), U=c["U"], Y=c["Y"], xi_reset=xi_reset, xi_dyn=xi_dyn, u_res=u_res,
                   strategy=np.int64(c.get("strategy", S.RESAMPLE_SYSTEMATIC)), family=np.frombuffer(c["family"].encode(), dtype=np.uint8),
                   df_mu=S.gaussian_mean(m.dynamics_density), df_cov=S.gaussian_cov_matrix(m.dynamics_density),
                   dg_mu=S.gaussian_mean(m.measurement_density), dg_cov=S.gaussian_cov_matrix(m.measurement_density),
                   d0_mu=S.gaussian_mean(m.initial_density), d0_cov=S.gaussian_cov_matrix(m.initial_density))
        if c["family"] in ("pf", "aux"):
            out.update(A=np.array(m.A[:m.nx * m.nx]).reshape(m.nx, m.nx), B=np.array(m.B[:m.nx * m.nu]).reshape(m.nx, m.nu), C=np.array(m.C[:m.ny * m.nx]).reshape(m.ny, m.nx))
        elif c["family"] == "apf":
            out.update(supersample=np.int64(m.supersample))
        else:
            # the Rao-Blackwellized model as the reference's constructor takes it (src/rbpf.jl:84-110): the inner KalmanFilter's A, B, C, R1, d0;
            # f_n(xn, u) = Fn xn + Bn u; g(xn) = Gn xn; An(xn) = An[0] + sum_k xn[k] An[1 + k]; R1n, R2, d0n are the three densities above
            nn, nl, ny, nu = m.nx, m.rb.nxl, m.ny, m.nu
            out.update(nxl=np.int64(nl), Fn=np.array(m.A[:nn * nn]).reshape(nn, nn), Bn=np.array(m.B[:nn * nu]).reshape(nn, nu), Gn=np.array(m.C[:ny * nn]).reshape(ny, nn),
                       Al=np.array(m.rb.Al[:nl * nl]).reshape(nl, nl), Bl=np.array(m.rb.Bl[:nl * nu]).reshape(nl, nu), Cl=np.array(m.rb.Cl[:ny * nl]).reshape(ny, nl),
                       An=np.array([list(m.rb.An[k][:nn * nl]) for k in range(nn + 1)]).reshape(nn + 1, nn, nl),
                       R1l=S.gaussian_cov_matrix(m.linear_noise),
                       d0l_mu=S.gaussian_mean(m.linear_initial), d0l_cov=S.gaussian_cov_matrix(m.linear_initial))
        np.savez_compressed(os.path.join(HERE, "ref_inputs_%s.npz" % name), **out)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items() if k in ("U", "Y", "xi_dyn", "u_res")})


if __name__ == "__main__":
    main()
