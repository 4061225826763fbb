"""Generates tests/golden/*.npz.

IMPORTANT: the reference (Julia) cannot run in the build container and ships no golden vectors for this path
(SURVEY.md §8c), so these fixtures are NOT reference outputs.  They freeze (i) the inputs of BASELINE config C1
and of a small quad-tank case, (ii) the outputs of this repo's oracle in both arithmetic orders, and (iii) the
closed-form Kalman log-likelihood for the linear-Gaussian case.  They pin the oracle against regressions and give
the GPU tests a file-based target; the oracle itself is pinned by the reference's known-answer assertions
(tests/test_oracle_kat.py).   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import models as M            # noqa: E402
import oracle_binding as ob   # noqa: E402
from llpf_amd import _structs as S   # noqa: E402


def case(model, U, Y, N, kind, thr, strategy, seed, t0):
    out = {}
    for name, order in (("ref", ob.ORDER_REFERENCE), ("dev", ob.ORDER_DEVICE)):
        cfg = S.make_config(model, N, kind, strategy, thr, seed, 0)
        o = ob.OracleFilter(cfg, order)
        o.reset()
        r = o.run(U, Y, t0, ll_steps=True, xmean=True)
        out["ll_steps_" + name] = r["ll_steps"]
        out["xmean_" + name] = r["xmean"]
        out["x_final_" + name] = o.particles()
        out["anc_final_" + name] = o.ancestors()
        out["resamples_" + name] = np.array(o.resample_count())
    return out


def main():
    m = M.lg_c1_model()
    _, U, Y = M.simulate_lg(m, 200)
    d = case(m, U, Y, 500, S.PARTICLE_FILTER, 0.1, S.RESAMPLE_SYSTEMATIC, 7, 0.0)
    d.update(U=U, Y=Y, kalman_ll=np.array(ob.kalman_loglik(m, U, Y)),
             A=np.array(m.A[:4]).reshape(2, 2), B=np.array(m.B[:4]).reshape(2, 2), C=np.array(m.C[:4]).reshape(2, 2),
             mu0=np.array(m.initial_density.mu[:2]))
    np.savez_compressed(os.path.join(HERE, "c1_lineargaussian_N500_T200.npz"), **d)

    q = M.quadtank_model()
    Uq, Yq = M.quadtank_data(40)
    d = case(q, Uq, Yq, 2000, S.ADVANCED_PARTICLE_FILTER, 0.5, S.RESAMPLE_STRATIFIED, 7, 485.0)
    d.update(U=Uq, Y=Yq)
    np.savez_compressed(os.path.join(HERE, "quadtank_N2000_T40.npz"), **d)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
