"""N > 1 host logic on CPU: two processes over gloo shard a sweep of independent filters and all-reduce the
per-filter log-likelihoods.  The per-rank compute is injected (oracle-backed stand-in here, the GPU bank in
bench.py), so this exercises exactly the sharding + collective code the GPU path uses."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleBank:
    """Stand-in with the BankHandle interface: filter k uses seed base + global index, like llpf_bank_create."""

    def __init__(self, models, owned, N, seed):
        import oracle_binding as ob
        from llpf_amd import _structs as S
        self.filters = [ob.OracleFilter(S.make_config(m, N, resample_threshold=0.1, seed=seed + g), ob.ORDER_DEVICE)
                        for m, g in zip(models, owned)]

    def reset(self):
        for f in self.filters:
            f.reset()

    def run(self, U, Y, t_index0):
        return {"ll": np.array([f.run(U, Y, t_index0)["ll"] for f in self.filters])}


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import models as M
    from llpf_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    models = [M.lg_test_model(s) for s in 10.0 ** np.linspace(-2, 0, 5)]
    _, U, Y = M.simulate_lg(models[2], 40)
    ll, tot = D.sharded_bank_loglik(lambda ms, owned: _OracleBank(ms, owned, 300, 50), models, U, Y, rank, world)
    q.put((rank, ll, tot))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sweep_matches_single_process():
    import models as M
    from llpf_amd import distributed as D
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
    models = [M.lg_test_model(s) for s in 10.0 ** np.linspace(-2, 0, 5)]
    _, U, Y = M.simulate_lg(models[2], 40)
    ll_ref, tot_ref = D.sharded_bank_loglik(lambda ms, owned: _OracleBank(ms, owned, 300, 50), models, U, Y, 0, 1)
    for rank, ll, tot in res:
        assert np.array_equal(ll, ll_ref), "rank %d" % rank      # every rank ends with the full vector
        assert tot == tot_ref
    assert D.shard_indices(5, 0, 2) == [0, 2, 4] and D.shard_indices(5, 1, 2) == [1, 3]
    assert sorted(D.shard_indices(1024, 3, 8))[:3] == [3, 11, 19] and len(D.shard_indices(1024, 3, 8)) == 128
