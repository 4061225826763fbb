"""BASELINE config C4 on one GPU: a bank of independent linear-Gaussian filters (noise-level sweep,
reference test/runtests.jl:412-417) batched into single launches.  Prints particle-steps/s."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import models as M
from llpf_amd import _capi, _structs as S

ap = argparse.ArgumentParser()
ap.add_argument("--filters", type=int, default=128)
ap.add_argument("--particles", type=int, default=100000)
ap.add_argument("--T", type=int, default=1000)
ap.add_argument("--thr", type=float, default=0.1)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
svec = 10.0 ** np.linspace(-2, 0, a.filters)
models = [M.lg_test_model(s) for s in svec]
_, U, Y = M.simulate_lg(M.lg_test_model(0.1), a.T, seed=1)
cfg = S.make_config(models[0], a.particles, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, a.thr, 5, 0)
bank = _capi.BankHandle(cfg, models)
bank.reset(); bank.run(U[:20], Y[:20], 1.0)
t0 = time.perf_counter()
for _ in range(a.reps):
    bank.reset()
    r = bank.run(U, Y, 1.0)
dt = (time.perf_counter() - t0) / a.reps
ps = a.filters * a.particles * a.T / dt
print(json.dumps({"filters": a.filters, "particles": a.particles, "T": a.T, "thr": a.thr, "s_per_pass": dt,
                  "particle_steps_per_s": ps, "us_per_timestep": 1e6 * dt / a.T, "resamples": bank.resample_count(),
                  "argmax_ll": int(np.argmax(r["ll"])), "sigma_at_argmax": float(svec[int(np.argmax(r["ll"]))]),
                  "whole_timestep_roofline_frac": a.filters * a.particles * 72 / (dt / a.T) / 8e12}))
