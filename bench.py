"""bench.py — throughput of the particle-filter hot path on MI355X.

One "step" = one loglik-style pass of the hot path over one batch of synthetic input: reset! followed by
T filter timesteps (correct! + predict!, reference src/filtering.jl:164-168,140-153) for N particles,
enqueued on the device through the C ABI (llpf_run).  Metric: particle-steps/s = ranks * K * N * T / time.

Default workload = BASELINE.json configs[1]: 2-D linear-Gaussian ParticleFilter (the reference's test
system, test/runtests.jl:255-266), N = 1e6, T = 1000, systematic resampling at every step
(resample_threshold = 1.0, so the scan/expansion kernel runs in all T steps).

N > 1 GPUs (`--gpus N`): one process per GPU.  Under `python -m torch.distributed.run` the ranks come from the
environment (RANK / LOCAL_RANK / WORLD_SIZE, which must equal --gpus); started as plain `python bench.py --gpus N`
this file re-executes itself under torch.distributed.run with N ranks.  The default workload for N > 1 is BASELINE
config C4: a sweep of independent filters (128 x N=1e5 per GPU) sharded over the ranks through the C ABI
(llpf_mbank_create_rank), whose only exchange — the all-reduce of the per-filter log-likelihood vector — is an
RCCL call inside the library on the communicator built from an id rank 0 hands out (weak scaling).
"""
import argparse
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np


def build_workload(name, n_particles, T):
    import models as M
    from llpf_amd import _structs as S
    if name == "lg":
        model = M.lg_test_model()
        _, U, Y = M.simulate_lg(model, T, seed=1)
        kind, thr = S.PARTICLE_FILTER, 1.0
        label = "C2: 2-D linear-Gaussian ParticleFilter (test/runtests.jl:255-266 system), N=%d, T=%d, systematic, resample every step" % (n_particles, T)
    elif name == "quadtank":
        model = M.quadtank_model()
        U, Y = M.quadtank_data(T, seed=2)
        kind, thr = S.ADVANCED_PARTICLE_FILTER, 0.5
        label = "C3: quad-tank AdvancedParticleFilter RK4x2, N=%d, T=%d, systematic, threshold 0.5" % (n_particles, T)
    elif name == "rbpf":
        g = S.make_gaussian
        # the mixed linear/nonlinear system of the reference's test/test_rbpf.jl:5-31 (1 + 1 states, An = 0.5)
        model = S.make_rb_model([[1.0]], np.zeros((1, 0)), [[0.5]], [[0.95]], np.zeros((1, 0)), [[1.0]], [[1.0]],
                                g(np.zeros(1), np.array([[0.01]])), [[0.01]], g(np.zeros(1), np.array([[0.1]])),
                                g(np.array([1.0]), np.array([[0.01]])), g(np.array([1.0]), np.array([[1.0]])))
        rng = np.random.default_rng(1)
        xn = xl = 1.0
        Y = np.zeros((T, 1))
        for t in range(T):
            Y[t] = xn + xl + np.sqrt(0.1) * rng.standard_normal()
            xn, xl = xn + 0.5 * xl + 0.1 * rng.standard_normal(), 0.95 * xl + 0.1 * rng.standard_normal()
        U = np.zeros((T, 0))
        kind, thr = S.PARTICLE_FILTER, 0.1
        label = "RBPF (test/test_rbpf.jl:5-31 system: 1 nonlinear + 1 linear state, An = 0.5), N=%d, T=%d, threshold 0.1" % (n_particles, T)
    elif name == "rbpf_full":
        import rbfull_models as RM
        # BASELINE config C5: quad-tank levels (4 nonlinear states, RK4 x 2) + 8 linear states with a state-dependent
        # coupling An(xn): the reference's singleR shortcut (src/rbpf.jl:176,247) is off, one 8x8 Riccati recursion per particle
        model = RM.quadtank_case()
        U, Y = RM.simulate_io(model, T, seed=3)
        kind, thr = S.PARTICLE_FILTER, 0.1
        label = ("C5: RBPF, quad-tank (4 nonlinear states) + 8 linear states, An(xn) state dependent => per-particle 8x8 covariance, "
                 "N=%d, T=%d, threshold 0.1" % (n_particles, T))
    elif name == "aux":
        model = M.lg_test_model()
        _, U, Y = M.simulate_lg(model, T, seed=1)
        kind, thr = S.PARTICLE_FILTER, 0.1
        label = "AuxiliaryParticleFilter loglik (src/smoothing.jl:232-236) on the C2 system, N=%d, T=%d, systematic (always resamples)" % (n_particles, T)
    else:
        raise ValueError(name)
    return model, U, Y, kind, thr, label


def cpu_baseline(model, U, Y, kind, thr, n_particles, budget_steps, seed, ll_gpu=None, aux=False, quick=False):
    """The reference-order oracle (literal CPU restatement) timed on a bounded sample of the same workload:
    1 thread (the reference's ParticleFilter path is single-threaded, src/PFtypes.jl:107-139) and, as an upper bound
    for its `threads=true` option, OpenMP over the per-particle loops (scan and sums stay serial).
    With the GPU's per-step log-likelihoods of the same seed it also reports the accuracy figures of SURVEY 8(d)."""
    import oracle_binding as ob
    from llpf_amd import _structs as S
    cfg = S.make_config(model, n_particles, kind, S.RESAMPLE_SYSTEMATIC, thr, seed, 0)
    Ts = min(budget_steps, len(Y))
    res = {}
    # the multi-thread leg is capped at 16 threads: beyond that the serial scan/sums between the parallel loops dominate
    # and idle OpenMP workers only add wake-up cost (256 hardware threads on the GPU box: 20x slower than 1 thread)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    legs = (("cpu_baseline", 1),) if quick else (("cpu_baseline", 1), ("cpu_baseline_multithread", max(1, min(16, ncpu))))
    for label, threads in legs:
        ob.set_threads(threads)
        o = ob.OracleFilter(cfg, ob.ORDER_REFERENCE)
        o.reset()
        t0 = time.perf_counter()
        r = o.run_aux(U[:Ts], Y[:Ts], 0, ll_steps=True) if aux else o.run(U[:Ts], Y[:Ts], 1.0, ll_steps=True)
        dt = time.perf_counter() - t0
        res[label] = {"value": n_particles * Ts / dt, "unit": "particle-steps/s", "cores": threads, "kind": "port",
                      "sample": "first %d of the %d timesteps of the same workload at N=%d (%.1f s of CPU)" % (Ts, len(Y), n_particles, dt)}
        if threads == 1 and ll_gpu is not None:
            d = np.abs(np.asarray(ll_gpu[:Ts]) - r["ll_steps"])
            first = int(np.argmax(d > 1e-10)) if np.any(d > 1e-10) else int(Ts)
            res["accuracy"] = {"vs_reference_order_oracle": {
                "timesteps": int(Ts), "tolerance_per_step": 1e-10, "leading_steps_within_tolerance": first,
                "max_abs_dll_per_step_within": float(d[:first].max()) if first else 0.0,
                "max_abs_dll_per_step_all": float(d.max()),
                "note": "fp64 serial cumsum (reference) vs exact fixed-point bins (device) pick a different ancestor when a "
                        "threshold lies within ~1e-13 of a bin edge: about once per 10 timesteps at N = 1e6; from then on the "
                        "two particle systems are different, equally valid realisations and differ at Monte-Carlo level "
                        "(~1/sqrt(N) per step)"}}
        del o
    if ll_gpu is not None and not quick:      # the bit-exact contract: device-order oracle, same inputs (bounded: 20 timesteps)
        ob.set_threads(max(1, min(16, ncpu)))
        Td = min(20, Ts)
        o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
        o.reset()
        r = o.run_aux(U[:Td], Y[:Td], 0, ll_steps=True) if aux else o.run(U[:Td], Y[:Td], 1.0, ll_steps=True)
        d = np.abs(np.asarray(ll_gpu[:Td]) - r["ll_steps"])
        res["accuracy"]["vs_device_order_oracle"] = {
            "timesteps": int(Td), "max_abs_dll_per_step": float(d.max()),
            "bit_identical": bool(np.array_equal(np.asarray(ll_gpu[:Td]).view(np.uint64), r["ll_steps"].view(np.uint64)))}
        del o
    ob.set_threads(1)
    return res


RCCL_COLLECTIVE = "RCCL ncclAllReduce(fp64, sum) of the log-likelihood vector inside libllpf_hip.so (llpf_mbank_run)"


def rccl_requirement_failure(collective, ranks_seen, world):
    """--require-rccl: None when the N-rank line was produced by the in-library RCCL all-reduce with every rank reporting in, else the
    reason (the run then exits non-zero: a scaling line must never silently come from the gloo fallback or from fewer ranks)."""
    if world > 1 and collective != RCCL_COLLECTIVE:
        return "the log-likelihood exchange was not the in-library RCCL all-reduce: " + collective
    if sorted(ranks_seen) != list(range(world)):
        return "ranks seen %s, expected 0..%d" % (sorted(ranks_seen), world - 1)
    return None


def main_bank(args, rank, world, dev):
    """BASELINE config C4: a sweep of independent linear-Gaussian filters (dynamics-noise level s_k, reference
    test/runtests.jl:412-417), filter k on rank k mod world, 128 x N=1e5 per GPU by default, shared u / y, through the
    C ABI's llpf_mbank_* handle.  The only collective is the all-reduce of the per-filter log-likelihood vector: RCCL inside
    the library (communicator from llpf_mbank_unique_id + llpf_mbank_create_rank); with --dist-backend gloo (ranks sharing
    a GPU: RCCL refuses that) the handle leaves the exchange to this file, which uses torch.distributed."""
    import torch
    import torch.distributed as dist
    import models as M
    from llpf_amd import _capi, _structs as S
    T = args.T if args.T else 1000
    N = args.particles if args.particles != 1000000 else 100000
    thr = 0.1 if args.threshold is None else args.threshold
    F = args.filters_per_gpu * world
    svec = 10.0 ** np.linspace(-2, 0, F)
    models = [M.lg_test_model(s) for s in svec]
    _, U, Y = M.simulate_lg(M.lg_test_model(0.1), T, seed=1)
    cfg = S.make_config(models[0], N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, 5, dev)
    device = args.coll_device
    collective = "none (one shard)"
    external = False
    if world == 1:
        bank = _capi.MBankHandle(cfg, models, devices=[dev])
    elif args.dist_backend == "nccl":
        ids = [_capi.mbank_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        # The communicator and its first all-reduce are made under a watchdog: this path has never run on more than one GPU
        # (none was available to the build), and a rendezvous that hangs must cost the scaling run a line that says so, not the run.
        box = {}

        def make_and_try():
            try:
                h = _capi.MBankHandle(cfg, models, rank=rank, world=world, unique_id=ids[0])
                h.reset()
                h.run(U[:2], Y[:2], 1.0)
                box["bank"] = h
            except Exception as e:              # reported, never silent: the line says which collective ran
                box["err"] = "%s: %s" % (type(e).__name__, e)

        th = threading.Thread(target=make_and_try, daemon=True)
        th.start()
        th.join(timeout=args.rccl_timeout)
        ok, err = 1, ""
        if th.is_alive():
            ok, err = 0, "no answer from ncclCommInitRank / the first ncclAllReduce within %g s on rank %d" % (args.rccl_timeout, rank)
            args.stuck_thread = True
        elif "err" in box:
            ok, err = 0, box["err"]
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            bank = box["bank"]
            collective = RCCL_COLLECTIVE
        else:
            errs = [None] * world
            dist.all_gather_object(errs, err)
            bank = _capi.MBankHandle(cfg, models, rank=rank, world=world, unique_id=None)     # a working handle of a peer is left alone
            external = True
            collective = ("torch.distributed all_reduce (gloo) of the vector llpf_mbank_run returned; in-library communicator failed: "
                          + "; ".join(sorted({e for e in errs if e})))
    else:
        bank = _capi.MBankHandle(cfg, models, rank=rank, world=world, unique_id=None)
        external = True
        collective = "torch.distributed all_reduce (gloo, CPU tensors): ranks share a GPU"
    info = bank.info()

    def one_pass():
        bank.reset()
        r = bank.run(U, Y, 1.0)
        if external:
            full = torch.as_tensor(r["ll"], device=device)
            dist.all_reduce(full)
            ll = full.cpu().numpy()
            return ll, float(ll.sum())
        return r["ll"], r["ll_sum"]

    for _ in range(2):                 # setup, untimed: hipGraph capture of the run shape (see main())
        one_pass()
    for _ in range(args.warmup):
        one_pass()
    # the same per-GPU share on ONE GPU with the others idle: the reference point for the scaling of this line
    solo = None
    if world > 1:
        dist.barrier()
        if rank == 0:
            sb = _capi.MBankHandle(cfg, models[: args.filters_per_gpu], devices=[dev])
            for _ in range(2):
                sb.reset(); sb.run(U, Y, 1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, min(args.steps, 3))):
                sb.reset(); sb.run(U, Y, 1.0)
            torch.cuda.synchronize()
            solo = max(1, min(args.steps, 3)) * args.filters_per_gpu * N * T / (time.perf_counter() - t0)
            sb.close()
        dist.barrier()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = coll_ms = 0.0
    for _ in range(args.steps):
        ll_all, ll_sum = one_pass()
        i = bank.info()
        dev_ms += i["last_run_ms"]
        coll_ms += i["last_collective_ms"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ranks_seen = [{"rank": rank, "device": dev, "filters": info["n_local_filters"], "seconds": dt}]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_seen[0])
        ranks_seen = gathered
    if rank == 0:
        bank.set_profiling(True)
        bank.reset()
        bank.run(U, Y, 1.0)
        ms_cls, n_cls = bank.profile(0)
        bank.set_profiling(False)
        nx = 2
        Fl = info["n_local_filters"]
        value = args.steps * F * N * T / dt
        timestep_s = dt / (args.steps * T)
        # split schedule: k_resprop moves 16nx+20 B per particle (read quanta 8, gather + store x 16nx,
        # ancestor 4, w 8), k_norm 16 (read w, write quanta)
        step_s = ms_cls[0] / n_cls[0] * 1e-3
        b_step = 16 * nx + 20
        b_alg = 16 * nx + 40
        achieved = Fl * N * b_step / step_s / 1e9
        out = {"metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "C4: sweep of %d linear-Gaussian ParticleFilters x N=%d, T=%d, threshold %g, %d per GPU" % (F, N, T, thr, args.filters_per_gpu),
                          "filters": F, "particles": N, "timesteps": T, "nx": nx, "resample_threshold": thr,
                          "resamples_per_pass_rank0": int(bank.resample_count()),
                          "parallelism": "filter k on rank k mod %d (llpf_mbank_*), one all-reduce of the log-likelihood vector per pass" % world,
                          "collective": collective, "control_plane": "torch.distributed gloo (rendezvous, barriers, timing reduction)"},
               "collective": collective,
               "ranks_seen": sorted(r["rank"] for r in ranks_seen), "rank_devices": {str(r["rank"]): r["device"] for r in ranks_seen},
               "rank_seconds": {str(r["rank"]): r["seconds"] for r in ranks_seen},
               "device_ms_per_step": dev_ms / args.steps, "collective_ms_per_step": coll_ms / args.steps,
               "kernel_us": {"k_resprop": 1e3 * ms_cls[0] / n_cls[0], "k_norm": (1e3 * ms_cls[1] / n_cls[1]) if n_cls[1] else None},
               "argmax_sigma": float(svec[int(np.argmax(ll_all))]), "loglik_sum": ll_sum,
               "roofline": {"bound": "hbm", "kernel": "k_resprop", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                            "frac": achieved / 8000.0, "traffic": None, "bytes_per_launch": Fl * N * b_step,
                            "avg_launch_us": step_s * 1e6,
                            "method": "hipEvent pairs around every launch on the engine stream of rank 0, one profiled pass after the timed region",
                            "whole_timestep": {"algorithmic_bytes": Fl * N * b_alg, "us": timestep_s * 1e6,
                                               "achieved": Fl * N * b_alg / timestep_s / 1e9, "frac": Fl * N * b_alg / timestep_s / 8e12}}}
        if solo is not None:
            out["scaling_efficiency"] = value / (world * solo)      # = one_gpu_same_share.efficiency: THE weak-scaling figure of this line
            out["one_gpu_same_share"] = {"value": solo, "unit": "particle-steps/s", "n_gpus": 1,
                                         "note": "rank 0 alone on its per-GPU share (%d filters) while the other ranks wait: aggregate / (n_gpus x this) "
                                                 "is the weak-scaling efficiency of this workload" % args.filters_per_gpu,
                                         "efficiency": value / (world * solo)}
        # The N = 1 default line is C2 (one filter), the N > 1 default lines are C4 (a sweep): a ratio of their `value`s would mix two
        # workloads.  Every line therefore names the one-GPU rate of ITS OWN per-GPU share, the denominator of a weak-scaling ratio.
        out["scaling_reference"] = {"workload": "C4 share: %d filters x N=%d per GPU" % (Fl, N), "value_one_gpu": solo if solo is not None else value,
                                    "unit": "particle-steps/s",
                                    "how_to_use": "weak-scaling efficiency of this line = value / (n_gpus x value_one_gpu); do not divide by the N = 1 C2 line"}
        pm = load_pmc()
        if pm and pm.get("c4") and pm["c4"]["filters"] == Fl and pm["c4"]["n_particles"] == N:
            out["roofline"]["traffic"] = pm["c4"]["k_resprop"]["bytes"]
            out["roofline"]["traffic_source"] = pm["c4"]["source"] + "; " + pm["correction"]
        if world == 1 and not args.no_cpu_baseline:
            cs = args.cpu_steps if args.cpu_steps else 1000
            if args.cpu_quick:
                cs = max(2, min(cs, int(8 * 2.9e7 / N)))          # ~8 s at the measured one-core rate
            out.update(cpu_baseline(models[len(models) // 2], U, Y, S.PARTICLE_FILTER, thr, N, min(cs, T), 77, None, quick=args.cpu_quick))
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if args.require_rccl:
            why = rccl_requirement_failure(collective, [r["rank"] for r in ranks_seen], world)
            out["require_rccl"] = "ok" if why is None else "FAILED: " + why
        print(json.dumps(out))
    rccl_why = rccl_requirement_failure(collective, [r["rank"] for r in ranks_seen], world) if args.require_rccl else None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rccl_why is not None:       # every rank decides alike (the collective and the gathered ranks are agreed values): the launcher sees rc != 0
        sys.stderr.write("bench.py --require-rccl: %s\n" % rccl_why)
        sys.stdout.flush()
        os._exit(3)
    if getattr(args, "stuck_thread", False):      # a thread is still inside RCCL: do not let interpreter shutdown wait for it
        sys.stdout.flush()
        os._exit(0)


def main_reference_mc(args, rank, world):
    """The reference's own published benchmark (run_test(), examples/example_lineargaussian.jl:282-316) through
    tools/bench_mc.py, in this file's output contract: one step = the whole benchmark (8.4e6 particle-steps); vs_baseline =
    value / the faster of the two published figures (docs/src/benchmark.md:48, 7.37e6 particle-steps/s, unstated CPU).
    Single GPU (the published benchmark is a single-process loop)."""
    import subprocess
    if world > 1:
        raise SystemExit("the reference_mc workload is a single-GPU benchmark")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_mc.py"), "--repeat", str(max(1, args.steps))]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    r = json.loads(subprocess.check_output(cmd, stderr=subprocess.DEVNULL).decode().strip().splitlines()[-1])
    out = {"metric": "particle-steps/s", "value": r["value"], "unit": "particle-steps/s", "n_gpus": 1, "steps": r["repetitions"],
           "warmup": 1, "ms_per_step": 1e3 * r["seconds_mean"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": r["vs_published_docs"], "dtype": "f64", "data": "synthetic",
           "config": {"workload": r["workload"], "propagated_particles": r["propagated_particles"], "filter_launches": r["filter_launches"],
                      "timed": r["timed"], "published_reference": r["published_reference"], "note": r["note"]},
           "seconds_breakdown_last_repetition": r["seconds_breakdown_last_repetition"],
           "roofline": {"bound": "hbm", "kernel": "k_resprop (banks of 2..2000 filters of 10..1000 particles)", "achieved": None, "peak": 8000.0,
                        "unit": "GB/s", "frac": None, "traffic": None,
                        "note": "launch- and host-bound at these sizes (0.05 s of 0.15 s on the device): no roofline figure is meaningful; see the C2 / C4 lines"}}
    if "cpu_baseline" in r:
        out["cpu_baseline"] = r["cpu_baseline"]
        out["speedup_vs_cpu_baseline"] = r["speedup_vs_cpu_baseline"]
    print(json.dumps(out))


def other_configs():
    """The default line also carries the other single-GPU BASELINE configs, each from a short run of this same script (5 timed
    passes, a bounded one-thread CPU baseline of ~8 s) with its own roofline and cpu_baseline blocks: C3 quad-tank (N = 1e6, T = 2000), one GPU's share of C4 (128 filters x
    1e5), C5 RBPF with per-particle covariance (N = 2e5) — and (round 6) the C2 system at N = 1.6e7, whose working set does not fit the
    Infinity Cache.  Their full lines (with the multi-thread leg and the accuracy block) are `--workload ...`."""
    import subprocess
    res = {}
    for key, wl, extra in (("C3_quadtank", "quadtank", []), ("C4_share_128x1e5", "bank", []), ("C5_rbpf_full", "rbpf_full", []),
                           # round 6: the C2 system with a working set (1 GB) beyond the 256 MB Infinity Cache: the headline's N = 1e6 (64 MB) is
                           # served from the MALL, so its "HBM" fraction is not an HBM measurement; this one is (tools/bench_n.py has the sweep)
                           ("C2_beyond_infinity_cache_N1.6e7", "lg", ["--particles", "16000000", "--T", "100"])):
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", wl, "--steps", "5", "--warmup", "1", "--cpu-quick"] + extra,
                               capture_output=True, text=True, timeout=300)
            d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
            res[key] = {k: d.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "kernel_us", "roofline",
                                              "cpu_baseline", "speedup_vs_cpu_baseline") if k in d}
        except Exception as e:      # a failed side run must not cost the headline line
            res[key] = {"error": repr(e)[:300]}
    return res


def engine_source_hash():
    """sha256 (16 hex digits) over the engine's sources: a committed PMC summary is only quoted for the build it was taken from."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "lowlevelparticlefilters.jl_amd", "csrc")
    files = []
    for d, _, fs in os.walk(csrc):
        files += [os.path.join(d, f) for f in fs if f.endswith((".hip", ".hpp", ".h"))]
    for f in sorted(files):
        h.update(os.path.relpath(f, csrc).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc():
    """The newest profiles/r*_pmc_traffic.json whose `engine_source_hash` matches the sources of this build, else None
    (rocprofv3 --pmc needs passes of its own, so bench.py cannot collect HBM traffic inside its timed run; a summary taken
    from another build of the kernels is not quoted)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            pm = json.load(open(f))
        except Exception:
            continue
        if pm.get("engine_source_hash") == engine_source_hash():
            pm["file"] = os.path.relpath(f, ROOT)
            return pm
    return None


def self_spawn(args, argv):
    """`python bench.py --gpus N` outside a launcher: re-execute under torch.distributed.run with N ranks on this node."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


SCALING_REFERENCE_DOC = ("every bench line carries `scaling_reference.value_one_gpu`: the one-GPU rate of the per-GPU share the N > 1 lines run "
                         "(BASELINE config C4), so that a scaling ratio never divides a C4 sweep by the C2 single filter of the N = 1 default line")


def main_spawn_check(rank, local_rank, world, backend, require_rccl=False):
    """--spawn-check: every rank reports in and rank 0 prints what it saw; no GPU work (the CPU test of the launcher logic)."""
    import torch
    import torch.distributed as dist
    me = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), "cuda_devices": torch.cuda.device_count()}
    seen = [me]
    if world > 1:
        dist.init_process_group(backend="gloo")
        seen = [None] * world
        dist.all_gather_object(seen, me)
    if rank == 0:
        print(json.dumps({"spawn_check": True, "n_gpus": world, "ranks_seen": sorted(r["rank"] for r in seen),
                          "distinct_processes": len({r["pid"] for r in seen}), "ranks": seen, "require_rccl": bool(require_rccl),
                          "default_workload": "C4 share (bank) per GPU" if world > 1 else "C2 (lg)", "scaling_reference_doc": SCALING_REFERENCE_DOC}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if require_rccl and world > 1 and backend != "nccl":
        # what the real run would decide in this mode, decided here too so that the flag's wiring is testable without a GPU
        why = rccl_requirement_failure("torch.distributed all_reduce (gloo, CPU tensors): ranks share a GPU", [r["rank"] for r in seen], world)
        sys.stderr.write("bench.py --require-rccl: %s\n" % why)
        return 3
    if require_rccl:
        why = rccl_requirement_failure(RCCL_COLLECTIVE, [r["rank"] for r in seen], world)
        if why is not None:
            sys.stderr.write("bench.py --require-rccl: %s\n" % why)
            return 3
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=["lg", "quadtank", "aux", "bank", "rbpf", "rbpf_full", "reference_mc"],
                    help="default: lg (BASELINE config C2) on one GPU, bank (BASELINE config C4, the sharded sweep) on several")
    ap.add_argument("--spawn-check", action="store_true", help="only start the ranks and report them (no GPU work)")
    ap.add_argument("--filters-per-gpu", type=int, default=128, help="bank workload (BASELINE config C4): filters per GPU")
    ap.add_argument("--particles", type=int, default=1000000)
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--threshold", type=float, default=None, help="resample_threshold override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default C2 line only: skip the short runs of the other BASELINE configs (C3, one GPU's share of C4, C5) that it carries as `other_configs`")
    ap.add_argument("--cpu-steps", type=int, default=None, help="timesteps of the CPU baseline sample")
    ap.add_argument("--rccl-timeout", type=float, default=120.0,
                    help="N > 1: seconds the in-library RCCL communicator and its first all-reduce may take before the run falls back "
                         "to exchanging the log-likelihood vector through torch.distributed (and says so in config.collective)")
    ap.add_argument("--require-rccl", action="store_true",
                    help="N > 1: exit non-zero (after printing the line, which then carries require_rccl: FAILED ...) unless the log-likelihood exchange "
                         "was the in-library RCCL all-reduce and every one of the N ranks reported in — no silent gloo fallback in a scaling run.  "
                         "Since round 6 this is the DEFAULT for N > 1 with --dist-backend nccl; the flag remains for --dist-backend gloo / N = 1")
    ap.add_argument("--allow-gloo-exchange", action="store_true",
                    help="N > 1, --dist-backend nccl: opt out of the default above — if the in-library RCCL communicator cannot be built the run falls "
                         "back to exchanging the log-likelihood vector through torch.distributed (gloo), says so in `collective`, and exits 0")
    ap.add_argument("--cpu-quick", action="store_true",
                    help="bounded CPU baseline for the side runs of other_configs: the one-thread leg only, ~8 s of CPU, no accuracy block")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the measured path): one rank per GPU, the log-likelihood all-reduce over RCCL INSIDE the library; "
                         "gloo: that exchange through torch.distributed on CPU tensors, ranks may share devices (exercises the multi-rank "
                         "logic on a box with fewer GPUs than ranks).  Rendezvous, barriers and the timing reduction use gloo in both")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_spawn(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; they must agree" % (args.gpus, world))
    if args.workload is None:
        args.workload = "lg" if world == 1 else "bank"
    # a scaling line must not silently come from the fallback exchange: required by default when the measured path (nccl) was asked for
    if world > 1 and args.dist_backend == "nccl" and not args.allow_gloo_exchange:
        args.require_rccl = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.spawn_check:
        sys.exit(main_spawn_check(rank, local_rank, world, args.dist_backend, args.require_rccl))

    import torch
    import torch.distributed as dist

    dev = 0
    if world > 1:
        # one rank per GPU (nccl) / ranks folded onto the GPUs there are (gloo)
        fold = args.dist_backend != "nccl" or os.environ.get("LLPF_BENCH_FOLD_DEVICES")   # the env var: to watch the nccl mode's
        dev = local_rank % torch.cuda.device_count() if fold else local_rank                  # fallback on a box with too few GPUs
        torch.cuda.set_device(dev)
        # Control plane (unique-id broadcast, barriers, the max over ranks of the elapsed time) on gloo in BOTH modes.  The data
        # path's collective is RCCL inside libllpf_hip.so, bound to /opt/rocm's runtime; the torch wheel bundles a second RCCL
        # and HIP runtime, and with gloo here that second RCCL is never initialised in this process.
        dist.init_process_group(backend="gloo")
    args.coll_device = torch.device("cpu")

    if args.workload == "bank":
        return main_bank(args, rank, world, dev)
    if args.workload == "reference_mc":
        return main_reference_mc(args, rank, world)
    from llpf_amd import _capi, _structs as S
    T = args.T if args.T else (2000 if args.workload == "quadtank" else 1000)
    rbfull = args.workload == "rbpf_full"
    if rbfull and args.particles == 1000000:
        args.particles = 200000                      # BASELINE config C5 is quoted at N = 2e5
    model, U, Y, kind, thr, label = build_workload(args.workload, args.particles, T)
    if args.threshold is not None:
        thr = args.threshold
        label += " [resample_threshold overridden to %g]" % thr
    N = args.particles
    cfg = S.make_config(model, N, kind, S.RESAMPLE_SYSTEMATIC, thr, 1000 + rank, dev)
    pf = _capi.FilterHandle(cfg)
    ll_dev = torch.zeros(1, dtype=torch.float64, device=args.coll_device)

    aux = args.workload == "aux"

    def run_once(h, **kw):
        return h.run_aux(U, Y, 1, **kw) if aux else h.run(U, Y, 1.0, **kw)

    def one_pass():
        pf.reset()
        r = run_once(pf)
        if world > 1:
            ll_dev[0] = r["ll"]
            dist.all_reduce(ll_dev)          # the global log-likelihood: the only collective of the path
        return r["ll"]

    # setup, untimed: the engine captures a run shape into a hipGraph the second time it sees it; two passes
    # here keep that one-off capture (~10 ms) out of the warm-up accounting and of the timed region
    for _ in range(2):
        one_pass()
    for _ in range(args.warmup):
        one_pass()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev_ms = 0.0
    for _ in range(args.steps):
        ll = one_pass()
        dev_ms += pf.last_run_ms()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=args.coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    resamples = pf.resample_count()

    # per-kernel durations with HIP events on the engine's own stream (separate passes so that the
    # event records do not perturb the timed region above)
    prof = None
    if rank == 0:
        pf.set_profiling(True)
        for _ in range(max(1, min(args.steps, 3))):
            pf.reset()
            run_once(pf)
        ms, cnt = pf.profile()
        pf.set_profiling(False)
        prof = (ms, cnt)

    out = None
    if rank == 0:
        nx = model.nx
        value = world * args.steps * N * T / dt
        ms_cls, n_cls = prof
        fused = aux or not n_cls[2]               # no standalone resample launches => the fused k_resprop ran
        names = ["k_resprop(finalize+resample+propagate+weight)" if fused else "k_step(propagate+weight)",
                 "k_norm(exp-weights, sums, quanta)", "k_resample(finalize+scan+counts+ancestors)", "other"]
        if args.workload == "quadtank":
            names[0] = "k_step<MARKS>(marks->ancestors, gather f(x[anc]), noise, weight, exp-sums)"
            names[2] = "k_resample_fx(finalize+scan+counts, f(x_j) per surviving source, run-start marks)"
        if aux:
            names = ["k_resprop<AUX>(expnormalize+resample+permute+noise+weights)", "k_step<MODE_AUX>(noise-free propagate + look-ahead lambda)",
                     "finalize(logsumexp of lambda - log N)", "other"]
        kernel_us = {names[i]: (1e3 * ms_cls[i] / n_cls[i] if n_cls[i] else None) for i in range(4)}
        # Dominant kernel and its algorithmic bytes.
        #   one-launch timestep (fused k_resprop that also forms the exp-sums: no k_norm launches): the launch IS the
        #     particle-step, so it is priced with SURVEY §8(d)'s figure B_alg = 16 nx + 40 bytes per particle-step;
        #     the bytes this kernel itself has to move (its HBM model, compared with the PMC traffic) are fewer:
        #     read quanta 8 + gather x 8nx + write x 8nx + write ancestor 4 + write w 8 + write quanta 8 = 16nx + 28
        #   fused k_resprop + separate k_norm : k_resprop moves 16nx + 20 (no quanta write)
        #   k_step (balanced propagate+weight): read ancestor 4 + gather x 8nx + write x 8nx + write w 8 [+ quanta 8]
        b_alg = 16 * nx + 40                         # whole-timestep algorithmic bytes
        one_launch = fused and not n_cls[1] and not aux
        if aux:
            # k_step<MODE_AUX>: read x 8nx + read w 8 + write x' 8nx + write lambda 8 + write w 8 + write quanta 8
            # k_resprop<AUX>  : read quanta 8 + gather x' 8nx + write x 8nx + read lambda 8 + write w 8 + anc 4 + quanta 8
            b_step = b_model = 16 * nx + 36
            b_alg = 32 * nx + 68
        elif one_launch:
            b_step, b_model = b_alg, 16 * nx + 28
        elif fused:
            b_step = b_model = 16 * nx + 20
        else:
            b_step = b_model = 16 * nx + (20 if not n_cls[1] else 12)
            if args.workload == "quadtank":      # k_step<..., MARKS>: run-start marks read (4) AND ancestors written (4) by the step kernel
                b_step = b_model = 16 * nx + 24
        if rbfull:
            # the particle plane of this model has rows = xn + xl + packed lower triangle of R (4 + 8 + 36 = 48):
            # k_rbfull reads ancestor 4 + gathers 8 rows + writes 8 rows + writes w 8; whole timestep adds the k_norm /
            # k_resample traffic of SURVEY 8(d): B_alg = 16 rows + 40
            rows = nx + model.rb.nxl + model.rb.nxl * (model.rb.nxl + 1) // 2
            b_step = b_model = 16 * rows + 12 + (8 if not n_cls[1] else 0)      # + its quanta when it forms the exp-sums itself
            b_alg = 16 * rows + 40
            names[0] = "k_rbfull(gather RBParticle + Riccati time update + Kalman measurement update + weight)"
            kernel_us = {names[i]: (1e3 * ms_cls[i] / n_cls[i] if n_cls[i] else None) for i in range(4)}
        timestep_s = dt / (args.steps * T)
        if one_launch:
            # the timed region itself is bracketed by HIP events on the engine stream (llpf_last_run_ms): T launches of
            # k_resprop back to back (+ one weight-only launch and one bookkeeping launch per pass, < 0.3 %)
            step_s = dev_ms * 1e-3 / (args.steps * T)
            method = ("HIP events on the engine stream around the run of the timed region (%d passes x %d launches), "
                      "divided by the launches; kernel_us holds per-launch event pairs of %d extra profiled passes "
                      "(each pair adds ~2 us to the launch it brackets)" % (args.steps, T, max(1, min(args.steps, 3))))
        else:
            # several launches per timestep: per-launch event pairs are the only way to tell the kernels apart inside bench.py, and every
            # pair adds ~2-3 us to the launch it brackets.  That overhead is measured, not guessed: the profiled passes' event time summed
            # over all launches, minus the un-bracketed device time of the same passes in the timed region (HIP events around the whole
            # run), divided by the number of launches — and subtracted, so that avg_launch_us is comparable with the rocprofv3 kernel trace
            passes = max(1, min(args.steps, 3))
            launches = float(sum(n_cls))
            overhead_s = max(0.0, (sum(ms_cls) / passes - dev_ms / args.steps) * 1e-3 / (launches / passes))
            step_s = ms_cls[0] / n_cls[0] * 1e-3 - overhead_s
            method = ("hipEvent pairs around every launch on the engine stream, %d profiled passes after the timed region, minus the measured "
                      "per-launch overhead of a pair (method_overhead_us: profiled event time of all launches minus the un-bracketed device time "
                      "of a pass, per launch); whole_timestep is the driver-clocked figure" % passes)
        achieved = N * b_step / step_s / 1e9
        roof = {"bound": "hbm", "kernel": "k_rbfull<MODE_PROP_WEIGHT>" if rbfull else ("k_resprop" if fused else "k_step<MODE_PROP_WEIGHT>"), "achieved": achieved, "peak": 8000.0,
                "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                "bytes_per_launch": N * b_step, "kernel_model_bytes": N * b_model, "avg_launch_us": step_s * 1e6,
                "launches_per_timestep": sum(n_cls[:3]) / float(n_cls[0]),
                "method": method, "method_overhead_us": 0.0 if one_launch else overhead_s * 1e6,
                "whole_timestep": {"algorithmic_bytes": N * b_alg, "us": timestep_s * 1e6,
                                   "achieved": N * b_alg / timestep_s / 1e9, "frac": N * b_alg / timestep_s / 8e12}}
        if rbfull:
            # fp64 work of one particle-step counted from csrc/shared/llpf_rbfull_body.h at (nxn, nxl, ny) = (4, 8, 2):
            # time update ~1700 fma (An(xn) 128, An xl 32, An R 256, Nt 80, Cholesky + V 110, x~l 32, R~ = R - V V' 144, Al x~l + Bl u 80,
            # two panels Al R~ 512, the three blocks of M Al' 288, R1l 36), measurement update ~420 fma (C R 128, S 32, K 48, R - K C R 144, ...),
            # RK4 x 2 of the quad-tank ~730 flop
            flop = 2.0 * 2120 + 730.0
            roof["compute"] = {"fp64_flop_per_particle_step": flop, "achieved_tflops": N * flop / step_s / 1e12,
                               "peak_tflops": 78.6, "frac": N * flop / step_s / 78.6e12}
        if args.workload == "quadtank":
            roof["note"] = ("round 4: the RK4 runs once per SURVIVING source in the resampling launch (k_resample_fx, ~8 per 1024-particle tile "
                            "here), k_step<..., MARKS> gathers f(x[ancestor]) and is bound by instruction issue — ~430 VALU instructions per "
                            "particle, 340 of them the four normals (Philox + Box-Muller in deterministic fp64), 70 % of its loop's cycles; "
                            "its HBM figures are reported as the contract asks, whole_timestep is the number the 40 % bar is read against")
        # HBM traffic of the dominant kernel from the committed PMC summary of this same workload AND build (cannot be
        # collected inside bench.py: rocprofv3 --pmc needs its own passes); quoted only when shapes and source hash match
        pm = load_pmc()
        if pm is None:
            roof["traffic_source"] = "no profiles/r*_pmc_traffic.json matches this build's engine_source_hash %s: not quoted" % engine_source_hash()
        else:
            try:
                if one_launch and pm["n_particles"] == N and pm["nx"] == nx and args.workload == "lg" and thr == pm["resample_threshold"]:
                    roof["traffic"] = pm["k_resprop"]["bytes"]
                    roof["traffic_source"] = pm["file"] + ": " + pm["source"] + "; " + pm["correction"]
                big = pm.get("c2_big")
                if fused and not aux and big and big["n_particles"] == N and args.workload == "lg" and thr == pm["resample_threshold"]:      # (split schedule at this size: k_norm + k_resprop)
                    roof["traffic"] = big["k_resprop"]["bytes"]
                    roof["traffic_source"] = pm["file"] + ": " + big["source"] + "; " + pm["correction"]
                if args.workload == "quadtank" and pm["c3"]["n_particles"] == N:
                    roof["traffic"] = pm["c3"][roof["kernel"].split("<")[0]]["bytes"]
                    roof["traffic_source"] = pm["file"] + ": " + pm["c3"]["source"] + "; " + pm["correction"]
                if rbfull and pm["c5"]["n_particles"] == N:
                    roof["traffic"] = pm["c5"]["k_rbfull"]["bytes"]
                    roof["traffic_source"] = pm["file"] + ": " + pm["c5"]["source"] + "; " + pm["correction"]
            except KeyError:
                pass
        out = {"metric": "particle-steps/s", "value": value, "unit": "particle-steps/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": label, "particles": N, "timesteps": T, "nx": nx, "resample_threshold": thr,
                          "resamples_per_pass": int(resamples), "parallelism": "independent filter per GPU, all-reduce of log-likelihood" if world > 1 else "1 GPU"},
               "device_ms_per_step": dev_ms / args.steps, "kernel_us": kernel_us, "loglik": ll,
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            per = 2.5e7 if args.workload == "lg" else (5e5 if rbfull else 4.5e6)   # measured 1-core rates, to size a ~10 s sample
            cs = args.cpu_steps if args.cpu_steps else max(2, min(T, int((8 if args.cpu_quick else 10) * per / N)))
            ll_gpu = None
            if not args.cpu_quick:
                pf2 = _capi.FilterHandle(cfg)          # a fresh handle: same Philox counters as a fresh oracle (first reset!)
                pf2.reset()
                ll_gpu = run_once(pf2, ll_steps=True)["ll_steps"]
                del pf2
            out.update(cpu_baseline(model, U, Y, kind, thr, N, min(cs, T - 1), 1000 + rank, ll_gpu, aux, quick=args.cpu_quick))
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            if args.workload == "lg" and not args.cpu_quick:
                # ancestor mismatches against the reference order at full size, teacher forced (tests/gpu_common.py)
                from gpu_common import teacher_forced_ancestor_mismatches
                tf = teacher_forced_ancestor_mismatches(cfg, U, Y, min(50, T - 1))
                tf["note"] = ("before every correct! and every predict! the reference-order oracle's state is installed in the engine; both take the "
                              "step with the same measurement / Philox draws; %d x %d ancestor decisions and %d log-likelihood increments compared "
                              "(tolerance: |dll| <= 1e-10, exp-weights rel <= 1e-12)" % (tf["resampling_steps"], N, tf["correct_steps"]))
                out["accuracy"]["ancestor_mismatches_vs_reference_order"] = tf
        if world == 1 and args.workload == "lg" and not args.no_other_configs and not args.no_cpu_baseline and args.particles == 1000000 and not args.T:
            out["other_configs"] = other_configs()
            c4 = out["other_configs"].get("C4_share_128x1e5", {})
            out["scaling_reference"] = {"workload": (c4.get("config") or {}).get("workload"), "value_one_gpu": c4.get("value"), "unit": "particle-steps/s",
                                        "how_to_use": "`bench.py --gpus N` (N > 1) runs BASELINE config C4, a sweep of independent filters, 128 per GPU — not this "
                                                      "line's C2 single filter.  The one-GPU rate of that per-GPU share is this value: weak-scaling efficiency of an "
                                                      "N-GPU line = its value / (N x value_one_gpu)"}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
