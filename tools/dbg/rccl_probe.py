"""one-rank RCCL communicators through the C ABI (and through torch.distributed for comparison) with NCCL_DEBUG output"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import models as M
from llpf_amd import _capi, _structs as S
which = sys.argv[1]
models = [M.lg_test_model(s) for s in (0.05, 0.1, 0.2)]
_, U, Y = M.simulate_lg(models[1], 20)
cfg = S.make_config(models[0], 2000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 3, 0)
if which == "torch":
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda", dtype=torch.float64); dist.all_reduce(t); print("torch nccl ok", t.cpu().numpy())
elif which == "all":
    os.environ["LLPF_MBANK_FORCE_RCCL"] = "1"
    mb = _capi.MBankHandle(cfg, models, devices=[0]); mb.reset(); print("initall ok", mb.info()["collective"], mb.run(U, Y, 1.0))
elif which == "rank":
    uid = _capi.mbank_unique_id()
    mb = _capi.MBankHandle(cfg, models, rank=0, world=1, unique_id=uid); mb.reset(); print("initrank ok", mb.info()["collective"], mb.run(U, Y, 1.0))
elif which == "rank_torchfirst":
    import torch
    torch.cuda.init()
    uid = _capi.mbank_unique_id()
    mb = _capi.MBankHandle(cfg, models, rank=0, world=1, unique_id=uid); mb.reset(); print("initrank(torch loaded) ok", mb.info()["collective"], mb.run(U, Y, 1.0))
