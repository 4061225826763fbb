"""Sweeps sharded over GPUs through the C ABI (llpf_mbank_*, include/llpf.h; reference layout: one filter per thread,
src/smoothing.jl:335-347, test/runtests.jl:412-417).  The GPU box has ONE device, so: the one-shard handle must be the plain
bank; two shards sharing the device (host-summed exchange) and two processes over gloo must return the unsharded sweep's bits;
and the RCCL path is exercised with a one-rank communicator (ncclCommInitAll / ncclCommInitRank + ncclAllReduce on the
engine stream), which is everything of it that one GPU can run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import models as M
from llpf_amd import _capi, _structs as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sweep(F=5, N=3000, T=30, thr=0.1, seed=321):
    svec = 10.0 ** np.linspace(-2, 0, F)
    models = [M.lg_test_model(s) for s in svec]
    _, U, Y = M.simulate_lg(M.lg_test_model(0.1), T)
    cfg = S.make_config(models[0], N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, thr, seed, 0)
    return cfg, models, U, Y


def _bank_ll(cfg, models, U, Y, passes=2):
    b = _capi.BankHandle(cfg, models)
    out = []
    for _ in range(passes):
        b.reset()
        out.append(b.run(U, Y, 1.0)["ll"])
    return out


def _same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint64), np.asarray(b).view(np.uint64))


def test_one_shard_is_the_plain_bank():
    cfg, models, U, Y = _sweep()
    ref = _bank_ll(cfg, models, U, Y)
    mb = _capi.MBankHandle(cfg, models, devices=[0])
    info = mb.info()
    assert info["collective"] == "none" and info["n_shards"] == 1 and info["n_local_filters"] == 5 and info["local_devices"] == [0]
    for p in range(2):
        mb.reset()
        r = mb.run(U, Y, 1.0)
        assert _same_bits(r["ll"], ref[p])
        assert r["ll_sum"] == float(np.sum(np.asarray(ref[p])[np.arange(5)]))   # index order
    assert mb.info()["last_run_ms"] > 0.0


@pytest.mark.parametrize("F,shards", [(5, 2), (7, 3), (128, 4)])
def test_shards_sharing_the_device_return_the_unsharded_bits(F, shards):
    """filter k lives on shard k mod S with key seed + k: same numbers wherever it lives; one host thread per shard"""
    cfg, models, U, Y = _sweep(F=F, N=2000 if F > 10 else 3000)
    ref = _bank_ll(cfg, models, U, Y)
    mb = _capi.MBankHandle(cfg, models, devices=[0] * shards)
    info = mb.info()
    assert info["collective"] == "host" and info["n_shards"] == shards and info["n_local_shards"] == shards and info["n_local_filters"] == F
    for p in range(2):
        mb.reset()
        r = mb.run(U, Y, 1.0)
        assert _same_bits(r["ll"], ref[p])
    # replicas (models = NULL) shard the same way
    mr = _capi.MBankHandle(cfg, None, n_filters=F, devices=[0] * shards)
    br = _capi.BankHandle(cfg, None, n_filters=F)
    mr.reset(); br.reset()
    assert _same_bits(mr.run(U, Y, 1.0)["ll"], br.run(U, Y, 1.0)["ll"])


def test_c4_as_stated_1024_filters_over_8_shards():
    """BASELINE config C4 at its stated shape — 1024 independent linear-Gaussian filters x N = 1e5 over EIGHT shards (folded onto
    the one device of this box: the host-summed exchange) — returns the bits of the unsharded 1024-filter bank, and three of its
    filters (first, middle, the last: shard 7, Philox key seed + 1023) the device-order ORACLE's.  Reference layout: one filter
    per parameter value, `map(svec) do s ... loglik(pfs, u, y) end` (test/runtests.jl:412-417)."""
    import oracle_binding as ob
    F, N, T = 1024, 100000, 10
    svec = 10.0 ** np.linspace(-2, 0, F)
    models = [M.lg_test_model(s) for s in svec]
    _, U, Y = M.simulate_lg(M.lg_test_model(0.1), T)
    cfg = S.make_config(models[0], N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 4000, 0)
    mb = _capi.MBankHandle(cfg, models, devices=[0] * 8)
    info = mb.info()
    assert info["n_shards"] == 8 and info["n_local_filters"] == F and info["collective"] == "host"
    mb.reset()
    rm = mb.run(U, Y, 1.0)
    del mb
    b = _capi.BankHandle(cfg, models)
    b.reset()
    rb = b.run(U, Y, 1.0)
    del b
    assert rm["ll"].shape == (F,) and _same_bits(rm["ll"], rb["ll"])
    acc = 0.0
    for v in rb["ll"]:                 # the library sums in index order (numpy's pairwise sum rounds differently at F = 1024)
        acc += float(v)
    assert rm["ll_sum"] == acc
    ob.set_threads(16)
    try:
        for k in (0, 511, 1023):
            ck = S.make_config(models[k], N, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 4000 + k, 0)
            o = ob.OracleFilter(ck, ob.ORDER_DEVICE)
            o.reset()
            ro = o.run(U, Y, 1.0)
            assert _same_bits(np.array([ro["ll"]]), np.array([rm["ll"][k]])), k
    finally:
        ob.set_threads(1)


def test_set_models_of_a_sharded_sweep():
    """llpf_mbank_set_models: every shard takes its own slice of the new descriptors (the partition of the creation); a sweep that was
    given new parameters returns the bits of a sweep (and of a plain bank) created with them"""
    cfg, models, U, Y = _sweep(F=7)
    models2 = [M.lg_test_model(s) for s in 10.0 ** np.linspace(-1.5, 0.3, 7)]
    fresh = _capi.BankHandle(S.make_config(models2[0], cfg.n_particles, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, cfg.resample_threshold, cfg.seed, 0), models2)
    fresh.seed(cfg.seed); fresh.reset()                      # (seed restarts the reset! / predict! counters: the same draws on both sides)
    ref2 = fresh.run(U, Y, 1.0)["ll"]
    mb = _capi.MBankHandle(cfg, models, devices=[0, 0, 0])
    mb.reset(); mb.run(U, Y, 1.0)
    mb.set_models(models2)
    mb.seed(cfg.seed); mb.reset()
    assert _same_bits(mb.run(U, Y, 1.0)["ll"], ref2)
    with pytest.raises(ValueError):
        mb.set_models(models2[:3])


def test_reseeding_and_aux_runs_shard_too():
    cfg, models, U, Y = _sweep(F=6)
    b = _capi.BankHandle(cfg, models)
    mb = _capi.MBankHandle(cfg, models, devices=[0, 0])
    b.seed(99); mb.seed(99)
    b.reset(); mb.reset()
    assert _same_bits(b.run(U, Y, 1.0)["ll"], mb.run(U, Y, 1.0)["ll"])
    b.reset(); mb.reset()
    assert _same_bits(b.run_aux(U, Y, 1)["ll"], mb.run_aux(U, Y, 1)["ll"])


def test_rccl_one_rank_communicators():
    """ncclCommInitAll([0]) and ncclGetUniqueId + ncclCommInitRank(world = 1): the all-reduce of the vector runs through
    RCCL on the engine's stream and leaves the bank's numbers unchanged"""
    cfg, models, U, Y = _sweep()
    ref = _bank_ll(cfg, models, U, Y, passes=1)[0]
    os.environ["LLPF_MBANK_FORCE_RCCL"] = "1"
    try:
        mb = _capi.MBankHandle(cfg, models, devices=[0])
    finally:
        del os.environ["LLPF_MBANK_FORCE_RCCL"]
    assert mb.info()["collective"] == "rccl"
    mb.reset()
    assert _same_bits(mb.run(U, Y, 1.0)["ll"], ref)
    uid = _capi.mbank_unique_id()
    assert len(uid) == 128 and any(uid)
    mr = _capi.MBankHandle(cfg, models, rank=0, world=1, unique_id=uid)
    assert mr.info()["collective"] == "rccl" and mr.info()["first_local_shard"] == 0
    mr.reset()
    r = mr.run(U, Y, 1.0)
    assert _same_bits(r["ll"], ref)
    assert mr.info()["last_collective_ms"] > 0.0


def test_rank_handle_without_communicator_returns_its_slots():
    cfg, models, U, Y = _sweep(F=5)
    ref = _bank_ll(cfg, models, U, Y, passes=1)[0]
    parts = []
    for rank in range(2):
        h = _capi.MBankHandle(cfg, models, rank=rank, world=2, unique_id=None)
        assert h.info()["collective"] == "external" and h.info()["first_local_shard"] == rank
        h.reset()
        parts.append(h.run(U, Y, 1.0)["ll"])
    assert np.all(parts[0][1::2] == 0.0) and np.all(parts[1][0::2] == 0.0)
    assert _same_bits(parts[0] + parts[1], ref)


def test_argument_errors():
    cfg, models, U, Y = _sweep(F=2)
    with pytest.raises(_capi.LLPFError):
        _capi.MBankHandle(cfg, models, devices=[0, 0, 0])          # more shards than filters
    with pytest.raises(_capi.LLPFError):
        _capi.MBankHandle(cfg, models, devices=[7])                # no such device on this box
    with pytest.raises(_capi.LLPFError):
        _capi.MBankHandle(cfg, models, rank=2, world=2)


def _bench(args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_two_ranks_over_gloo_on_one_gpu():
    """`python bench.py --gpus 2` starts two ranks by itself, defaults to the sharded sweep (C4 shape, small here) and gets
    the single-process sweep's global log-likelihood"""
    common = ["--particles", "4000", "--T", "25", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    two = _bench(["--gpus", "2", "--dist-backend", "gloo", "--filters-per-gpu", "3"] + common)
    one = _bench(["--gpus", "1", "--workload", "bank", "--filters-per-gpu", "6"] + common)
    assert two["n_gpus"] == 2 and two["ranks_seen"] == [0, 1] and one["n_gpus"] == 1
    assert two["config"]["workload"].startswith("C4") and two["config"]["filters"] == 6
    assert two["loglik_sum"] == one["loglik_sum"]
    assert two["one_gpu_same_share"]["value"] > 0
