"""Sharding of independent filters over ranks and the single collective of the path.

The reference has no distributed code.  Independent filters (parameter sweeps,
`map(svec) do s ... loglik(pfs,u,y)`, reference test/runtests.jl:412-417) shard trivially: filter k goes to
rank k mod world.  The only exchange is the all-reduce of the per-filter log-likelihood vector (each rank fills
its own slots, zeros elsewhere), 8 KB at 1024 filters — RCCL over xGMI on the GPU node (backend "nccl"),
gloo in the CPU tests.
"""
import numpy as np


def shard_indices(n_filters, rank, world):
    """Global indices of the filters owned by `rank` (round-robin)."""
    return list(range(rank, n_filters, world))


def allreduce_logliks(ll_local, owned, n_filters, device=None, group=None):
    """Scatter the local log-likelihoods into a length-n_filters vector and sum it over ranks.
    Returns (ll_all as numpy, global sum)."""
    import torch
    import torch.distributed as dist
    full = torch.zeros(n_filters, dtype=torch.float64, device=device if device is not None else "cpu")
    if len(owned):
        full[torch.as_tensor(owned, dtype=torch.long, device=full.device)] = torch.as_tensor(
            np.asarray(ll_local, dtype=np.float64), device=full.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    out = full.cpu().numpy()
    return out, float(out.sum())


def sharded_bank_loglik(make_bank, models, U, Y, rank, world, device=None, group=None, t_index0=1.0):
    """loglik of every filter of a sweep, sharded over ranks.

    make_bank(models_subset, first_global_index) -> object with .reset() and .run(U, Y, t_index0)["ll"]
    (a llpf_amd._capi.BankHandle on the GPU; the CPU tests inject an oracle-backed stand-in to exercise the
    host logic without a GPU)."""
    owned = shard_indices(len(models), rank, world)
    ll_local = np.zeros(0)
    if owned:
        bank = make_bank([models[k] for k in owned], owned)
        bank.reset()
        ll_local = np.asarray(bank.run(U, Y, t_index0)["ll"], dtype=np.float64)
    return allreduce_logliks(ll_local, owned, len(models), device, group)
