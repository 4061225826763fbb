"""Randomised parity sweep: the HIP engine against the device-order oracle (bit for bit) over random models, sizes, thresholds, resamplers,
covariance kinds, missing measurements and drivers (whole run, run after run on one handle, single-step verbs) — the configurations
nobody thought of writing a test for.  Developer aid (needs the GPU):   python tools/fuzz_parity.py [--cases 300] [--seed 0]
Prints one line per failing case with everything needed to reproduce it, and a summary."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_binding as ob
import models as M
from llpf_amd import _capi, _structs as S

ob.set_threads(8)


def rand_cov(rng, n, scale):
    kind = int(rng.integers(0, 3))
    if kind == 0:
        return S.make_gaussian(np.zeros(n), float(scale * (0.2 + rng.random())))
    if kind == 1:
        return S.make_gaussian(np.zeros(n), scale * (0.2 + rng.random(n)))
    L = rng.standard_normal((n, n)) * 0.3 + np.eye(n)
    return S.make_gaussian(np.zeros(n), scale * (L @ L.T))


HOOKED = None    # --hooked P: the share of cases with user hooks / Rao-Blackwellized models (default 0.2, none with --big unless given)
BIG = False      # --big: few timesteps at 7e4 .. 1e6 particles (many tiles, several rounds of the persistent step kernel, heavy tiles)


def rand_case(rng):
    fam = rng.choice(["lg", "lg", "lg", "quadtank", "lg_big"])
    N = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 511, 512, 513, 1000, 1023, 1024, 1025, 2047, 2049, 3000, 4097, 10000, 33000]))
    thr = float(rng.choice([0.0, 0.1, 0.3, 0.5, 0.9, 1.0]))
    strat = int(rng.choice([S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED, S.RESAMPLE_RESIDUAL]))
    T = int(rng.integers(3, 14))
    if BIG:
        N, T = int(rng.choice([70001, 131077, 262145, 500000, 1000000, 1048577])), int(rng.integers(3, 6))
    if fam == "quadtank":
        sig = float(rng.choice([0.003, 0.01, 0.05, 0.3]))
        m = S.make_quadtank_model(S.make_gaussian(np.zeros(4), np.full(4, 0.1 * (0.5 + rng.random()))), S.make_gaussian(np.zeros(2), np.full(2, sig ** 2)),
                                  S.make_gaussian(np.array([2.0, 2.0, 3.0, 3.0]), np.full(4, 0.1)), 1.0, int(rng.integers(1, 4)))
        U, Y = M.quadtank_data(T, seed=int(rng.integers(0, 1000)))
        Y = Y + sig * rng.standard_normal(Y.shape)
        kind = S.ADVANCED_PARTICLE_FILTER
        t0 = float(rng.choice([1.0, 490.0, 497.0]))
    else:
        nx = int(rng.integers(1, 5)) if fam == "lg" else int(rng.integers(5, S.MAX_DIM + 1))       # above 4 states: compiled on demand
        ny = int(rng.integers(1, 5)) if fam == "lg" else int(rng.integers(1, 11))
        nu = int(rng.integers(0, 3)) if fam == "lg" else int(rng.integers(0, S.MAX_INPUTS + 1))
        Q, _ = np.linalg.qr(rng.standard_normal((nx, nx)))
        A = Q @ np.diag(np.linspace(0.4, 0.97, nx)) @ Q.T
        m = S.make_lg_model(A, rng.standard_normal((nx, nu)) if nu else np.zeros((nx, 0)), rng.standard_normal((ny, nx)),
                            rand_cov(rng, nx, 0.05), rand_cov(rng, ny, float(rng.choice([0.01, 0.3, 1.0]))), rand_cov(rng, nx, 2.0), 1.0)
        m.initial_density.mu[0] = float(rng.standard_normal())
        _, U, Y = M.simulate_lg(m, T, seed=int(rng.integers(0, 1000)))
        kind = int(rng.choice([S.PARTICLE_FILTER, S.ADVANCED_PARTICLE_FILTER]))
        t0 = float(rng.choice([0.0, 1.0]))
    if rng.random() < 0.4:
        Y[int(rng.integers(0, T))] = np.nan
    if rng.random() < 0.15:
        Y[int(rng.integers(0, T))] += 25.0                  # an outlier: the bound test fails, the step is redone in exact form
    driver = str(rng.choice(["run", "run_twice", "steps", "run_then_steps", "aux", "aux", "bank", "history"]))
    if driver == "aux" and m.nx > 8:
        driver = "run"                                       # the auxiliary filter stops at 8 states (host/aux.hpp)
    return dict(fam=str(fam), N=N, thr=thr, strat=strat, T=T, kind=kind, t0=t0, seed=int(rng.integers(0, 2 ** 31)),
                driver=driver, model=m, U=U, Y=Y,
                bank_scales=[float(v) for v in (0.5 + rng.random(3))])


def eq(x, y):
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    return x.shape == y.shape and np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y)


SIDES = {}       # digests of what each side produced in the last check(): a failing case is run again and the sides compared with themselves


def _note(side, name, arr):
    import hashlib
    SIDES.setdefault(side, {})[name] = hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest()[:8]


def check(c):
    SIDES.clear()
    cfg = S.make_config(c["model"], c["N"], c["kind"], c["strat"], c["thr"], c["seed"], 0)
    g = _capi.FilterHandle(cfg); o = ob.OracleFilter(cfg, ob.ORDER_DEVICE)
    g.reset(); o.reset()
    U, Y, T, t0 = c["U"], c["Y"], c["T"], c["t0"]
    Ts = c["model"].Ts
    why = []
    try:
        if c["driver"] == "aux":
            mode = c["seed"] & 1
            Yn = np.where(np.isnan(Y), 0.1, Y)                  # (the look-ahead uses y[t+1]: missing rows are not part of this sweep)
            ra, rb = g.run_aux(U, Yn, mode=mode, ll_steps=True), o.run_aux(U, Yn, mode=mode, ll_steps=True)
            if not eq(ra["ll_steps"], rb["ll_steps"]):
                why.append("aux ll_steps first diff at %s" % np.flatnonzero(ra["ll_steps"] != rb["ll_steps"])[:3])
        if c["driver"] == "history":
            want_cov = bool(c["seed"] & 2)
            ra, rb = g.run(U, Y, t0, ll_steps=True, history=True, xcov=want_cov), o.run(U, Y, t0, ll_steps=True, history=True)
            for k in ("ll_steps", "x", "w", "we"):
                if not eq(ra[k], rb[k]):
                    why.append("history " + k)
            if want_cov and not why:                           # weighted_cov of every step against numpy on the history
                for k in range(T):
                    x, we = rb["x"][k], rb["we"][k]
                    if not np.all(np.isfinite(we)) or not np.all(np.isfinite(x)):
                        continue
                    sw, nnz = we.sum(), np.count_nonzero(we)      # StatsBase's corrected covariance under probability weights (filtering.jl:571-581)
                    if nnz < 2:
                        continue
                    mu = (x * we[:, None]).sum(axis=0) / sw
                    ref = ((x - mu) * we[:, None]).T @ (x - mu) * (nnz / ((nnz - 1) * sw))
                    if not np.allclose(ra["xcov"][k], ref, rtol=1e-8, atol=1e-12):
                        why.append("xcov step %d" % k); break
            if c["seed"] & 4 and not why and c["fam"] != "quadtank":   # the backward sampler on that history (FFBS, src/smoothing.jl:103-143)
                Ms = min(c["N"], int(3 + (c["seed"] >> 3) % (6 if c["N"] > 50000 else 40)))        # M <= N (src/smoothing.jl:121)
                xg, ig = g.smooth(Ms, U, ra["x"], ra["w"], ra["we"]); xo, io = o.smooth(Ms, U, rb["x"], rb["w"], rb["we"])
                if not (eq(xg, xo) and eq(ig, io)):
                    why.append("smoother (M=%d)" % Ms)
        if c["driver"] == "bank" and c["fam"] != "quadtank":
            ms = []
            for sc in c["bank_scales"]:
                m = S.Model.from_buffer_copy(bytes(c["model"]))
                for i in range(m.nx * m.nx):
                    m.A[i] = m.A[i] * sc
                ms.append(m)
            bank = _capi.BankHandle(cfg, ms)
            bank.reset()
            rb = bank.run(U, Y, t0, ll_steps=True)
            for f, m in enumerate(ms):
                cf = S.make_config(m, c["N"], c["kind"], c["strat"], c["thr"], c["seed"] + f, 0)
                of = ob.OracleFilter(cf, ob.ORDER_DEVICE)
                of.reset()
                if not eq(rb["ll_steps"][:, f].copy(), of.run(U, Y, t0, ll_steps=True)["ll_steps"]):
                    why.append("bank filter %d" % f)
            if not why and c["N"] <= 300000:
                # the same sweep sharded over 1..4 shards that share the device (filter k on shard k mod S): the unsharded bits
                shards = 1 + (c["seed"] >> 5) % len(ms)                 # every shard needs a filter
                mb = _capi.MBankHandle(cfg, ms, devices=[0] * shards)
                mb.seed(c["seed"] + 3); mb.reset()                      # (seed + reset: the constructor's draw of the initial particles)
                ll_m = np.asarray(mb.run(U, Y, t0)["ll"])
                bank.seed(c["seed"] + 3); bank.reset()
                ll_b = np.asarray(bank.run(U, Y, t0)["ll"])
                if not eq(ll_m, ll_b):
                    why.append("mbank over %d shards" % shards)
                # new parameters for the existing bank = a fresh bank on them (llpf_bank_set_models)
                ms2 = []
                for m in ms[::-1]:
                    m2 = S.Model.from_buffer_copy(bytes(m))
                    for i in range(m2.nx * m2.nx):
                        m2.A[i] = m2.A[i] * 0.9
                    ms2.append(m2)
                fresh = _capi.BankHandle(S.make_config(ms2[0], c["N"], c["kind"], c["strat"], c["thr"], c["seed"], 0), ms2)
                bank.set_models(ms2)
                bank.seed(99); bank.reset(); fresh.seed(99); fresh.reset()
                if not eq(bank.run(U, Y, t0, ll_steps=True)["ll_steps"], fresh.run(U, Y, t0, ll_steps=True)["ll_steps"]):
                    why.append("set_models")
        if c["driver"] in ("run", "run_twice", "run_then_steps"):
            for rep in range(2 if c["driver"] == "run_twice" else 1):
                if rep:
                    g.reset(); o.reset()
                rg = g.run(U, Y, t0, ll_steps=True, xmean=True); ro = o.run(U, Y, t0, ll_steps=True, xmean=True)
                _note("engine", "xmean%d" % rep, rg["xmean"]); _note("oracle", "xmean%d" % rep, ro["xmean"])
                if not eq(rg["ll_steps"], ro["ll_steps"]):
                    why.append("ll_steps(run %d) first diff at %s" % (rep, np.flatnonzero(rg["ll_steps"] != ro["ll_steps"])[:3]))
                if not np.allclose(rg["xmean"], ro["xmean"], rtol=1e-9, atol=1e-11, equal_nan=True):
                    bad = ~np.isclose(rg["xmean"], ro["xmean"], rtol=1e-9, atol=1e-11, equal_nan=True)
                    why.append("xmean (run %d) at %s: engine %s oracle %s" % (rep, np.argwhere(bad)[:3].tolist(), rg["xmean"][bad][:3], ro["xmean"][bad][:3]))
        if c["driver"] in ("steps", "run_then_steps"):
            for k in range(T):
                t = (t0 + k) * Ts
                u = U[k] if U is not None and U.size else None
                lg_, lo_ = g.correct(u, Y[k], t), o.correct(u, Y[k], t)
                if not (lg_ == lo_ or (lg_ != lg_ and lo_ != lo_)):
                    why.append("ll step %d: %r vs %r" % (k, lg_, lo_)); break
                g.predict(u, t); o.predict(u, t)
        if c["driver"] in ("run", "steps") and c["N"] >= 63 and c["seed"] & 8 and c["thr"] >= 0.5:
            # weighted_quantile of the state the run left (src/filtering.jl:583-595), interior quantiles of a healthy weight vector.
            # (Not compared: p = 0 and 1 and degenerate weights.  The engine's running sums are 2^-96 fixed point: particles lighter
            # than 2^-96 of the heaviest carry no mass, so for p -> 0 it returns the LAST such value below the first particle with
            # mass where StatsBase returns the smallest; and StatsBase's interpolation (h - S_{k-1}) / (S_k - S_{k-1}) cancels
            # where w_k << S, in both implementations, to different roundings.)
            qs = np.array([0.1, 0.25, 0.5, 0.75, 0.9])
            qg, qo = g.weighted_quantile(qs), o.weighted_quantile(qs)
            xs = o.particles()
            spread = np.ptp(xs[np.isfinite(xs).all(axis=1)], axis=0) + 1.0 if np.isfinite(xs).all(axis=1).any() else 1.0
            # (the engine's running sums are exact 2^-96 fixed point, the restated StatsBase algorithm sums in fp64: the interpolation
            # weight differs by rounding where neighbouring weights are tiny)
            if not np.all((np.abs(qg - qo) <= 1e-7 * spread) | (np.isnan(qg) & np.isnan(qo))):
                why.append("weighted_quantile: max difference %g" % np.nanmax(np.abs(qg - qo)))
        for name, fg, fo in (("x", g.particles, o.particles), ("w", g.weights, o.weights), ("we", g.expweights, o.expweights), ("j", g.ancestors, o.ancestors)):
            vg, vo = np.ascontiguousarray(fg()), np.ascontiguousarray(fo())
            _note("engine", name, vg); _note("oracle", name, vo)
            if not eq(vg, vo):
                if vg.shape == vo.shape:
                    ne = (vg.view(np.uint64) != vo.view(np.uint64)) if vg.dtype == np.float64 else (vg != vo)
                    idx = np.argwhere(ne)
                    why.append("%s: %d of %d entries differ, first %s (engine %r, oracle %r), last %s" % (name, len(idx), vg.size, idx[0].tolist(), vg[tuple(idx[0])], vo[tuple(idx[0])], idx[-1].tolist()))
                else:
                    why.append(name + " shapes %s %s" % (vg.shape, vo.shape))
    except _capi.LLPFError as e:
        if not o.L.orc_degenerate(o.h):
            why.append("engine error, oracle fine: %s" % e)
    return why


def check_hooked(rng):
    """the cases with user hooks (likelihood / noise / initial density of the model's own) and the Rao-Blackwellized models of
    tests/independent_cases.py under random sizes, thresholds, resamplers, seeds and run lengths"""
    import independent_cases as IC
    if not hasattr(check_hooked, "cases"):
        check_hooked.cases = IC.cases()
    name = str(rng.choice([k for k in check_hooked.cases if k not in ("pf_lg_systematic", "pf_lg_stratified", "pf_quadtank")]))
    case = dict(check_hooked.cases[name])
    T = int(rng.integers(3, min(16, len(case["Y"])) + 1))
    sizes = [70001, 131077, 200000, 262145] if BIG else [1, 63, 64, 65, 300, 1023, 1025, 2500]
    case.update(N=int(rng.choice(sizes)), thr=float(rng.choice([0.1, 0.5, 0.9, 1.0])),
                strategy=int(rng.choice([S.RESAMPLE_SYSTEMATIC, S.RESAMPLE_STRATIFIED])), U=case["U"][:T], Y=case["Y"][:T])
    IC.SEED = int(rng.integers(1, 2 ** 31))
    try:
        g, o = IC.engine_of(case), IC.oracle_of(ob, case, ob.ORDER_DEVICE)
        g.reset(); o.reset()
        rg, ro = g.run(case["U"], case["Y"], case["t0"], ll_steps=True), o.run(case["U"], case["Y"], case["t0"], ll_steps=True)
        why = [] if eq(rg["ll_steps"], ro["ll_steps"]) else ["ll_steps"]
        for nm, fg, fo in (("x", g.particles, o.particles), ("w", g.weights, o.weights), ("j", g.ancestors, o.ancestors)):
            if not eq(fg(), fo()):
                why.append(nm)
    finally:
        seed_used, IC.SEED = IC.SEED, 7
    return "hooked:" + name, ("%s N=%d thr=%g strat=%d T=%d seed=%d" % (name, case["N"], case["thr"], case["strategy"], T, seed_used)), why


def sweep(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    bad, drivers = [], {}
    for i in range(cases):
        if rng.random() < (HOOKED if HOOKED is not None else 0.2) and (not BIG or HOOKED is not None):
            drv, desc, why = check_hooked(rng)
            drivers["hooked"] = drivers.get("hooked", 0) + 1
            if why:
                bad.append("case %d: %s : %s" % (i, desc, "; ".join(why)))
                if verbose:
                    print("FAIL " + bad[-1], flush=True)
            continue
        c = rand_case(rng)
        drivers[c["driver"]] = drivers.get(c["driver"], 0) + 1
        why = check(c)
        if why:
            first = {k: dict(v) for k, v in SIDES.items()}
            for again in range(2):                                   # who moves when the same case is run again?
                w2 = check(c)
                moved = ["%s.%s" % (side, k) for side in first for k in first[side] if SIDES.get(side, {}).get(k) != first[side][k]]
                why.append("[again %d: %s; changed since the failing run: %s]" % (again, "fails" if w2 else "passes", ", ".join(moved) or "nothing"))
            bad.append("case %d: %s N=%d thr=%g strat=%d T=%d kind=%d t0=%g seed=%d driver=%s nx=%d ny=%d nu=%d : %s" %
                       (i, c["fam"], c["N"], c["thr"], c["strat"], c["T"], c["kind"], c["t0"], c["seed"], c["driver"], c["model"].nx, c["model"].ny, c["model"].nu, "; ".join(why)))
            if verbose:
                print("FAIL " + bad[-1], flush=True)
    return bad, drivers


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--hooked", type=float, default=None)
    a = ap.parse_args()
    BIG = a.big
    HOOKED = a.hooked
    bad, drivers = sweep(a.cases, a.seed)
    print("%d cases (%s), %d failed (seed %d)" % (a.cases, ", ".join("%s %d" % kv for kv in sorted(drivers.items())), len(bad), a.seed))
    sys.exit(1 if bad else 0)
