import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import models as M, oracle_binding as ob
from llpf_amd import _capi, _structs as S
for strat in (0,1):
  for N in (1000, 2000, 3000):
    model = M.quadtank_model(); U,Y = M.quadtank_data(40)
    cfg = S.make_config(model, N, 1, strat, 0.5, 7, 0)
    g=_capi.FilterHandle(cfg); o=ob.OracleFilter(cfg, ob.ORDER_DEVICE); g.reset(); o.reset()
    r=g.run(U,Y,485.0,ll_steps=True)
    lls=[]; res=[]
    for k in range(40):
        lls.append(o.correct(U[k],Y[k],(485.0+k))); o.predict(U[k],(485.0+k)); res.append(o.last_resampled())
    lls=np.array(lls)
    bad=np.nonzero(lls!=r['ll_steps'])[0]
    print("strat",strat,"N",N,"first mismatch",bad[:3], "resampled flags", ''.join('R' if x else '.' for x in res))
model = M.lg_test_model(); _,U,Y=M.simulate_lg(model,40)
for N in (1000,2000,5000):
  for thr in (0.5,1.0):
    cfg = S.make_config(model, N, 0, 0, thr, 7, 0)
    g=_capi.FilterHandle(cfg); o=ob.OracleFilter(cfg, ob.ORDER_DEVICE); g.reset(); o.reset()
    r=g.run(U,Y,0.0,ll_steps=True); ro=o.run(U,Y,0.0,ll_steps=True)
    bad=np.nonzero(ro['ll_steps']!=r['ll_steps'])[0]
    print("LG N",N,"thr",thr,"first mismatch",bad[:3])
