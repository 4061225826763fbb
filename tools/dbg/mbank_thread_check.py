import sys, threading, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import models as M
from llpf_amd import _capi, _structs as S
models = [M.lg_test_model(s) for s in (0.05, 0.1, 0.2, 0.4)]
_, U, Y = M.simulate_lg(M.lg_test_model(0.1), 50, seed=1)
cfg = S.make_config(models[0], 20000, S.PARTICLE_FILTER, S.RESAMPLE_SYSTEMATIC, 0.1, 5, 0)
uid = _capi.mbank_unique_id()
box = {}
def mk():
    h = _capi.MBankHandle(cfg, models, rank=0, world=1, unique_id=uid)
    h.reset(); h.run(U[:2], Y[:2], 1.0)
    box['bank'] = h
t = threading.Thread(target=mk, daemon=True); t.start(); t.join(60)
assert not t.is_alive() and 'bank' in box
b = box['bank']
b.reset(); r1 = b.run(U, Y, 1.0)
ref = _capi.MBankHandle(cfg, models, devices=[0]); ref.reset(); ref.run(U[:2], Y[:2], 1.0); ref.reset(); r2 = ref.run(U, Y, 1.0)
print("created in a thread, used from main:", np.array_equal(r1['ll'], r2['ll']), r1['ll_sum'], b.info()['collective'] if 'collective' in b.info() else b.info())
