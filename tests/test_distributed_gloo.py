"""N > 1 host logic on CPU: two processes over gloo shard a sweep of independent filters with the partition of the C ABI
(llpf_mbank_partition — the function llpf_mbank_create / llpf_mbank_create_rank use themselves, pure host code) and own the exchange
like a caller of llpf_mbank_create_rank(..., id = NULL): every rank fills its own slots of the per-filter log-likelihood vector and
the vectors are summed (all-reduce).  The per-rank compute is an oracle-backed stand-in (no GPU here); filter k keeps key seed + k
wherever it lives, so the sharded sweep must return the unsharded one's bits."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FILTERS, N_PART, SEED = 5, 300, 50


def _sweep():
    import models as M
    models = [M.lg_test_model(s) for s in 10.0 ** np.linspace(-2, 0, N_FILTERS)]
    _, U, Y = M.simulate_lg(models[2], 40)
    return models, U, Y


def _shard_ll(models, U, Y, owned):
    """what a shard computes: log-likelihood of every owned filter, key seed + global index (llpf_bank_create's rule)"""
    import oracle_binding as ob
    from llpf_amd import _structs as S
    full = np.zeros(len(models))
    for k in owned:
        f = ob.OracleFilter(S.make_config(models[k], N_PART, resample_threshold=0.1, seed=SEED + k), ob.ORDER_DEVICE)
        f.reset()
        full[k] = f.run(U, Y, 1.0)["ll"]
    return full


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from llpf_amd import _capi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    models, U, Y = _sweep()
    owned = _capi.mbank_partition(len(models), rank, world)
    t = torch.from_numpy(_shard_ll(models, U, Y, owned))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)           # a slot has one non-zero contribution: the sum is exact
    q.put((rank, owned, t.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_matches_single_process():
    from llpf_amd import _capi
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    models, U, Y = _sweep()
    ll_ref = _shard_ll(models, U, Y, range(len(models)))
    seen = []
    for rank, owned, ll in res:
        assert np.array_equal(ll, ll_ref), "rank %d" % rank      # every rank ends with the full vector, bit for bit
        seen += owned
    assert sorted(seen) == list(range(N_FILTERS))


def test_partition_of_the_abi():
    from llpf_amd import _capi
    assert _capi.mbank_partition(5, 0, 2) == [0, 2, 4] and _capi.mbank_partition(5, 1, 2) == [1, 3]
    p = _capi.mbank_partition(1024, 3, 8)                       # BASELINE config C4: 128 filters per GPU
    assert p[:3] == [3, 11, 19] and len(p) == 128 and p[-1] == 1019
    assert _capi.mbank_partition(3, 7, 8) == [] and _capi.mbank_partition(0, 0, 1) == []
    with pytest.raises(_capi.LLPFError):
        _capi.mbank_partition(10, 2, 2)


def _worker_gpu(rank, world, port, q):
    """a rank of the one-process-per-GPU layout on the REAL library: llpf_mbank_create_rank(rank, world, id = NULL) — the caller owns the
    exchange — on the box's one GPU (ranks share it), the vector exchanged over gloo"""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from llpf_amd import _capi, _structs as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    models, U, Y = _sweep()
    cfg = S.make_config(models[0], N_PART, resample_threshold=0.1, seed=SEED)
    mb = _capi.MBankHandle(cfg, models, rank=rank, world=world, unique_id=None)
    info = mb.info()
    mb.reset()
    r = mb.run(U, Y, 1.0)                                   # this rank's slots filled, zeros elsewhere
    mine = _capi.mbank_partition(len(models), rank, world)
    others = [k for k in range(len(models)) if k not in mine]
    assert np.all(r["ll"][others] == 0.0) and np.all(r["ll"][mine] != 0.0)
    t = torch.from_numpy(r["ll"].copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    q.put((rank, info["n_local_filters"], info["collective"], t.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_the_real_library_match_the_unsharded_bank():
    """Round 6: the world-2 path driven through libllpf_hip.so itself when a GPU is there (the CPU test above stands the oracle in for
    the per-rank compute): two processes, llpf_mbank_create_rank with id = NULL, the exchange over gloo — every rank ends with the bits
    of the unsharded bank of the same sweep, and of the device-order oracle."""
    from llpf_amd import _capi, _structs as S
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 7) % 500)
    procs = [ctx.Process(target=_worker_gpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    models, U, Y = _sweep()
    bank = _capi.MBankHandle(S.make_config(models[0], N_PART, resample_threshold=0.1, seed=SEED), models, devices=[0])
    bank.reset()
    ll_one = bank.run(U, Y, 1.0)["ll"]
    ll_ref = _shard_ll(models, U, Y, range(len(models)))
    assert np.array_equal(ll_one, ll_ref)
    assert sorted(n for _, n, _, _ in res) == [2, 3]
    for rank, _, collective, ll in res:
        assert collective == "external", "MBANK_COLL_EXTERNAL expected"          # the handle left the exchange to the caller
        assert np.array_equal(ll, ll_one), "rank %d" % rank
