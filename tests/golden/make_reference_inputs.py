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
    lg = M.lg_test_model()
    _, U, Y = M.simulate_lg(lg, 60, seed=5)
    Y = Y.copy()
    Y[23] = np.nan                                       # one missing measurement (any(ismissing, y) && return, src/PFtypes.jl:109)
    qt = M.quadtank_model()
    Uq, Yq = M.quadtank_data(40, seed=3)
    return {
        "lg": dict(model=lg, kind=S.PARTICLE_FILTER, N=200, T=60, thr=0.5, t_index0=0.0, U=U, Y=Y),
        "quadtank": dict(model=qt, kind=S.ADVANCED_PARTICLE_FILTER, N=200, T=40, thr=0.5, t_index0=485.0, U=Uq, Y=Yq),   # crosses t > 500
    }


def draws(case):
    nx, N, T = case["model"].nx, case["N"], case["T"]
    xi_reset = ob.normals(SEED, 1, STREAM_INIT, nx, N)
    xi_dyn = np.stack([ob.normals(SEED, k, STREAM_DYNAMICS, nx, N) for k in range(T)])
    u_res = np.array([ob.resample_uniforms(S.RESAMPLE_SYSTEMATIC, N, SEED, k)[0] for k in range(T)])
    return xi_reset, xi_dyn, u_res


def main():
    for name, c in cases().items():
        m = c["model"]
        xi_reset, xi_dyn, u_res = draws(c)
        out = dict(N=np.int64(c["N"]), T=np.int64(c["T"]), nx=np.int64(m.nx), nu=np.int64(m.nu), ny=np.int64(m.ny), thr=np.float64(c["thr"]), Ts=np.float64(m.Ts),
                   t_index0=np.float64(c["t_index0"]), seed=np.int64(SEED), U=c["U"], Y=c["Y"], xi_reset=xi_reset, xi_dyn=xi_dyn, u_res=u_res,
                   df_mu=S.gaussian_mean(m.dynamics_density), df_cov=S.gaussian_cov_matrix(m.dynamics_density),
                   dg_mu=S.gaussian_mean(m.measurement_density), dg_cov=S.gaussian_cov_matrix(m.measurement_density),
                   d0_mu=S.gaussian_mean(m.initial_density), d0_cov=S.gaussian_cov_matrix(m.initial_density))
        if name == "lg":
            out.update(A=np.array(m.A[:m.nx * m.nx]).reshape(m.nx, m.nx), B=np.array(m.B[:m.nx * m.nu]).reshape(m.nx, m.nu), C=np.array(m.C[:m.ny * m.nx]).reshape(m.ny, m.nx))
        else:
            out.update(supersample=np.int64(m.supersample))
        np.savez_compressed(os.path.join(HERE, "ref_inputs_%s.npz" % name), **out)
        print(name, {k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items() if k in ("U", "Y", "xi_dyn", "u_res")})


if __name__ == "__main__":
    main()
