"""FFBS smoother throughput (reference src/smoothing.jl:116-143): M*N*(T-1) transition-density evaluations.
GPU through llpf_smooth, CPU = the reference-order oracle on a few trajectories of the same problem."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import models as M
import oracle_binding as ob
from llpf_amd import _capi, _structs as S

ap = argparse.ArgumentParser()
ap.add_argument("--particles", type=int, default=100000)
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--M", type=int, default=1000)
ap.add_argument("--cpu-M", type=int, default=4)
a = ap.parse_args()
model = M.lg_test_model(0.1)
X, U, Y = M.simulate_lg(model, a.T, seed=2)
cfg = S.make_config(model, a.particles, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 3, 0)
g = _capi.FilterHandle(cfg)
g.reset()
r = g.run(U, Y, 0.0, history=True)
g.smooth(min(a.M, 16), U, r["x"], r["w"], r["we"])                 # warm-up
t0 = time.perf_counter()
xb, idx = g.smooth(a.M, U, r["x"], r["w"], r["we"])
wall = time.perf_counter() - t0
dev_s = g.last_run_ms() * 1e-3
evals = a.M * a.particles * (a.T - 1)
o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
t0 = time.perf_counter()
o.smooth(a.cpu_M, U, r["x"], r["w"], r["we"])
cpu = time.perf_counter() - t0
cpu_rate = a.cpu_M * a.particles * (a.T - 1) / cpu
print(json.dumps({"workload": "FFBS smoother, 2-D linear-Gaussian", "N": a.particles, "T": a.T, "M": a.M,
                  "density_evaluations": evals, "device_s": dev_s, "wall_s_incl_history_upload": wall,
                  "evaluations_per_s_device": evals / dev_s, "cpu_port_evaluations_per_s_1_thread": cpu_rate,
                  "speedup_vs_cpu_port": evals / dev_s / cpu_rate,
                  "smoothed_mse": float(np.mean((X - xb.mean(axis=1)) ** 2)),
                  "filter_mse": float(np.mean((X - np.einsum("tnd,tn->td", r["x"], r["we"])) ** 2))}))
