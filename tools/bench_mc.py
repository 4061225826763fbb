#!/usr/bin/env python3
"""The reference's own published particle-filter benchmark on this engine: `run_test()` of
examples/example_lineargaussian.jl:282-316 (= docs/src/benchmark.md:12-48): particle counts {10,...,1000} x time steps
{20,100,200}, 2*1000*200/(T N) independent Monte-Carlo runs per cell — 8.4e6 particle-steps in total, each run = a new
filter, T-1 updates `pf(u, y)` on freshly simulated data and a `weighted_mean` per step.  Published: 1.613 s (5.21e6
particle-steps/s, examples/...:316) and 1.140 s (7.37e6, docs/src/benchmark.md:48), hardware not stated.

Here the Monte-Carlo runs of a cell are the filters of one bank (`llpf_bank_run_multi`: every filter has inputs of its
own, seeds differ), so a cell costs T launches.  Timed like the reference's `@elapsed`: filter (bank) construction, the
simulation of the true systems (numpy on the host), the upload, all filter steps with the weighted means, and the RMSE.
One difference is stated rather than hidden: the engine's weighted means are those after `correct!` (posterior), the
reference's loop takes `weighted_mean(pf)` after `pf(u,y)` i.e. after `predict!`; the work is the same.

    python tools/bench_mc.py            # one JSON line
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def simulate_cell(A, B, Cm, m0, runs, T, rng):
    """runs independent trajectories of the example's system: x0 ~ dx0, u_t ~ randn(2), x' = A x + B u + N(0, I), y = C x + N(0, I)"""
    x = m0 + 2.0 * rng.standard_normal((runs, 2))
    U = rng.standard_normal((runs, T, 2))
    X = np.zeros((runs, T, 2))
    Y = np.zeros((runs, T, 2))
    for t in range(T):
        X[:, t] = x
        Y[:, t] = x @ Cm.T + rng.standard_normal((runs, 2))
        x = x @ A.T + U[:, t] @ B.T + rng.standard_normal((runs, 2))
    return U, Y, X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeat", type=int, default=3, help="timed repetitions of the whole benchmark (best and mean reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import models as M
    from llpf_amd import _capi, _structs as S
    model = M.lg_c1_model()
    nx = 2
    A = np.array(model.A[:4]).reshape(2, 2); B = np.array(model.B[:4]).reshape(2, 2); Cm = np.array(model.C[:4]).reshape(2, 2)
    m0 = S.gaussian_mean(model.initial_density)
    particle_count = [10, 20, 50, 100, 200, 500, 1000]
    time_steps = [20, 100, 200]

    def run_test(seed):
        rng = np.random.default_rng(seed)
        propagated = 0
        rmse = np.zeros((len(particle_count), len(time_steps)))
        launches = 0
        tb = {"simulate": 0.0, "create": 0.0, "run": 0.0, "device": 0.0, "post": 0.0}
        for Ti, T in enumerate(time_steps):
            for Ni, N in enumerate(particle_count):
                runs = 2 * max(particle_count) * max(time_steps) // T // N
                t1 = time.perf_counter()
                U, Y, X = simulate_cell(A, B, Cm, m0, runs, T, rng)
                t2 = time.perf_counter()
                cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, seed * 100003 + 7 * Ti + Ni, 0)
                bank = _capi.BankHandle(cfg, None, runs)               # "pf = ParticleFilter(N, ...)" of every Monte-Carlo run
                t3 = time.perf_counter()
                r = bank.run_multi(U, Y, 0.0, xmean=True)
                t4 = time.perf_counter()
                tb["device"] += bank.last_run_ms() * 1e-3
                err = X - np.transpose(r["xmean"], (1, 0, 2))         # [runs, T, nx]
                rmse[Ni, Ti] = np.mean(np.sqrt(np.sum(err ** 2, axis=(1, 2)) / T))
                propagated += runs * N * T
                launches += T
                del bank
                t5 = time.perf_counter()
                tb["simulate"] += t2 - t1; tb["create"] += t3 - t2; tb["run"] += t4 - t3; tb["post"] += t5 - t4
        return propagated, rmse, launches, tb

    def run_test_cpu(seed):
        """the same loops on the reference-order CPU oracle, one filter object per Monte-Carlo run, 1 thread"""
        import oracle_binding as ob
        rng = np.random.default_rng(seed)
        propagated = 0
        for Ti, T in enumerate(time_steps):
            for Ni, N in enumerate(particle_count):
                runs = 2 * max(particle_count) * max(time_steps) // T // N
                U, Y, X = simulate_cell(A, B, Cm, m0, runs, T, rng)
                for k in range(runs):
                    cfg = S.make_config(model, N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, seed * 100003 + 7 * Ti + Ni + k, 0)
                    o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
                    o.run(U[k], Y[k], 0.0, xmean=True)
                propagated += runs * N * T
        return propagated

    run_test(0)                                   # untimed: library load, first-touch of the kernels
    times = []
    for rep in range(args.repeat):
        t0 = time.perf_counter()
        propagated, rmse, launches, tb = run_test(rep + 1)
        times.append(time.perf_counter() - t0)
    best, mean = min(times), float(np.mean(times))
    published = {"examples/example_lineargaussian.jl:316": {"seconds": 1.612975455, "particle_steps_per_s": 8400000 / 1.612975455},
                 "docs/src/benchmark.md:48": {"seconds": 1.140468043, "particle_steps_per_s": 8400000 / 1.140468043}}
    out = {"metric": "particle-steps/s", "workload": "run_test() of examples/example_lineargaussian.jl:282-316: N in {10..1000} x T in {20,100,200}, "
                                                      "2*1000*200/(T N) Monte-Carlo runs per cell, new filter per run, weighted mean per step",
           "propagated_particles": int(propagated), "seconds_best": best, "seconds_mean": mean, "repetitions": args.repeat,
           "value": propagated / mean, "value_best": propagated / best,
           "timed": "bank construction + host simulation of the true systems + upload + all filter steps + weighted means + RMSE",
           "filter_launches": int(launches), "seconds_breakdown_last_repetition": tb,
           "published_reference": published,
           "vs_published_docs": (propagated / mean) / published["docs/src/benchmark.md:48"]["particle_steps_per_s"],
           "vs_published_example": (propagated / mean) / published["examples/example_lineargaussian.jl:316"]["particle_steps_per_s"],
           "note": "published numbers are from unstated CPUs; weighted means here are posterior (after correct!), the reference's loop reads "
                   "them after predict!",
           "rmse_T200": [float(v) for v in rmse[:, 2]], "particle_count": particle_count}
    if not args.no_cpu_baseline:
        t0 = time.perf_counter()
        pc = run_test_cpu(1)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": pc / dtc, "unit": "particle-steps/s", "cores": 1, "kind": "port", "seconds": dtc,
                               "sample": "the whole benchmark once on the reference-order oracle (one filter object per Monte-Carlo run, ctypes-driven)"}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
