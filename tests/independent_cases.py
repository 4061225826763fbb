"""The cases of the independent numpy restatement (oracle/independent.py): the same systems expressed twice — as the numpy
objects that restatement takes and as the llpf_model the engine / the C oracle take — plus the Philox draws both are fed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import independent as ind          # oracle/independent.py
import models as M
import rbfull_models as RM
from llpf_amd import _structs as S

STREAM_INIT, STREAM_DYNAMICS, STREAM_USER, STREAM_USER_INIT = 0, 1, 7, 8
SEED = 7


def _gauss(g):
    return ind.Gaussian(S.gaussian_mean(g), S.gaussian_cov_matrix(g))


def _mat(arr, r, c):
    return np.array(arr[:r * c], dtype=np.float64).reshape(r, c)


def _lg_objects(m):
    return ind.LinearModel(_mat(m.A, m.nx, m.nx), _mat(m.B, m.nx, m.nu) if m.nu else None, _mat(m.C, m.ny, m.nx))


def _qt_object(m):
    q = list(m.qt)
    return ind.QuadTank(k1=q[0], k2=q[1], g=q[2], A=tuple(q[3:7]), a=tuple(q[7:11]), gamma=tuple(q[11:13]), tswitch=q[13],
                        a1factor=q[14], eps=q[15], Ts=m.Ts, supersample=m.supersample)


def cases():
    """name -> dict(model (llpf_model), kind, N, T, thr, strategy, t0, U, Y, make (-> independent filter))"""
    out = {}
    lg = M.lg_test_model()
    _, U, Y = M.simulate_lg(lg, 60, seed=3)
    Y = Y.copy()
    Y[17] = np.nan                                      # a missing measurement
    for name, thr, strat in (("pf_lg_systematic", 0.5, S.RESAMPLE_SYSTEMATIC), ("pf_lg_stratified", 1.0, S.RESAMPLE_STRATIFIED)):
        out[name] = dict(model=lg, kind=S.PARTICLE_FILTER, N=400, T=60, thr=thr, strategy=strat, t0=0.0, U=U, Y=Y,
                         make=lambda thr=thr, strat=strat: ind.ParticleFilter(400, _lg_objects(lg), _gauss(lg.dynamics_density), _gauss(lg.measurement_density),
                                                                              _gauss(lg.initial_density), thr, strat == S.RESAMPLE_STRATIFIED, lg.Ts))
    # measurement likelihoods of the user's own (run-time compiled on the engine side, tests/user_models.py): Laplace and Student-t
    # noise on the same linear system; `user` = (oracle kind, oracle parameters, device snippet, qt parameter block of the snippet)
    import user_models as UM
    st = ind.StudentT(4.0, 0.7)
    for name, dens, user in (("pf_lg_laplace", ind.Laplace(0.8), (1, [0.8], UM.LAPLACE_SRC, [0.8])),
                             ("pf_lg_student_t", st, (2, [4.0, 0.7, st.c1], UM.STUDENT_T_SRC, [4.0, 0.7, st.c1]))):
        out[name] = dict(model=lg, kind=S.ADVANCED_PARTICLE_FILTER, N=400, T=60, thr=0.5, strategy=S.RESAMPLE_SYSTEMATIC, t0=0.0, U=U, Y=Y, user=user,
                         make=lambda dens=dens: ind.ParticleFilter(400, _lg_objects(lg), _gauss(lg.dynamics_density), dens, _gauss(lg.initial_density), 0.5, False, lg.Ts))
    # process noise / initial density of the model's own (UserModel::noise / ::initial): `noise` = (oracle kind, oracle parameters),
    # `initial` likewise, `user` = (-, -, device snippet, qt parameter block)
    box = ([-1.0, 0.5], [3.0, 2.5])
    for name, dfo, d0o, extra in (
            ("pf_lg_mult_noise_box", ind.MultiplicativeGaussianNoise(0.1, 0.25), ind.UniformBox(*box),
             dict(noise=(1, [0.1, 0.25]), initial=(1, box[0] + box[1]), user=(0, [], UM.MULT_NOISE_BOX_SRC, [0.1, 0.25] + box[0] + box[1]))),
            ("pf_lg_laplace_noise", ind.LaplaceNoise(0.2), _gauss(lg.initial_density),
             dict(noise=(2, [0.2]), user=(0, [], UM.LAPLACE_NOISE_SRC, [0.2])))):
        out[name] = dict(model=lg, kind=S.ADVANCED_PARTICLE_FILTER, N=400, T=60, thr=0.5, strategy=S.RESAMPLE_SYSTEMATIC, t0=0.0, U=U, Y=Y,
                         make=lambda dfo=dfo, d0o=d0o: ind.ParticleFilter(400, _lg_objects(lg), dfo, _gauss(lg.measurement_density), d0o, 0.5, False, lg.Ts), **extra)
    qt = M.quadtank_model()
    Uq, Yq = M.quadtank_data(40, seed=2)
    out["pf_quadtank"] = dict(model=qt, kind=S.ADVANCED_PARTICLE_FILTER, N=300, T=40, thr=0.5, strategy=S.RESAMPLE_SYSTEMATIC, t0=485.0, U=Uq, Y=Yq,
                              make=lambda: ind.ParticleFilter(300, _qt_object(qt), _gauss(qt.dynamics_density), _gauss(qt.measurement_density),
                                                              _gauss(qt.initial_density), 0.5, False, qt.Ts))
    for shape in ((1, 2, 1), (2, 2, 2), (4, 8, 2), "quadtank"):
        if shape == "quadtank":
            m = RM.quadtank_case()
            name = "rb_quadtank_4_8_2"
        else:
            m, _ = RM.linear_case(*shape, seed=1, state_dependent=True)
            name = "rb_linear_%d_%d_%d" % shape
        Ur, Yr = RM.simulate_io(m, 20, seed=4)
        nn, nl, ny, nu = m.nx, m.rb.nxl, m.ny, m.nu

        def make(m=m, nn=nn, nl=nl, ny=ny, nu=nu):
            An = np.array([list(m.rb.An[k][:nn * nl]) for k in range(nn + 1)]).reshape(nn + 1, nn, nl)
            fn = _qt_object(m) if m.rb.fn_kind == 1 else ind.LinearModel(_mat(m.A, nn, nn), _mat(m.B, nn, nu) if nu else None, _mat(m.C, ny, nn))
            return ind.RBPF(200, fn, An, _mat(m.rb.Al, nl, nl), _mat(m.rb.Bl, nl, nu) if nu else None, _mat(m.rb.Cl, ny, nl),
                            _gauss(m.dynamics_density), S.gaussian_cov_matrix(m.linear_noise), _gauss(m.measurement_density),
                            _gauss(m.initial_density), _gauss(m.linear_initial), 0.5, False, m.Ts)
        out[name] = dict(model=m, kind=S.PARTICLE_FILTER, N=200, T=20, thr=0.5, strategy=S.RESAMPLE_SYSTEMATIC, t0=0.0, U=Ur, Y=Yr, make=make, rb=True)
    return out


def draws(ob, case):
    """the Philox draws a fresh handle consumes: reset! number 1 (the constructor took number 0), predict! numbers 0, 1, ..."""
    nd = case["model"].nx

    def normals(step, n, k):
        return ob.normals(SEED, step, STREAM_DYNAMICS, nd, n)[:, :k]

    def uniforms(step, n):
        return ob.resample_uniforms(case["strategy"], n, SEED, step)

    return ob.normals(SEED, 1, STREAM_INIT, nd, case["N"]), normals, uniforms


def run_independent(ob, case):
    f = case["make"]()
    xi0, normals, uniforms = draws(ob, case)
    nd = case["model"].nx
    if case.get("initial"):
        f.reset(xi0, ob.uniforms_nd(SEED, 1, STREAM_USER_INIT, nd, case["N"]))
    else:
        f.reset(xi0)
    if case.get("noise"):
        ll_steps, nres = f.run(case["U"], case["Y"], case["t0"], normals, uniforms,
                               user_uniforms=lambda step, n, k: ob.uniforms_nd(SEED, step, STREAM_USER, nd, n)[:, :k])
    else:
        ll_steps, nres = f.run(case["U"], case["Y"], case["t0"], normals, uniforms)
    res = dict(ll_steps=ll_steps, resamples=np.int64(nres), anc_final=np.asarray(f.j, dtype=np.int64))
    if case.get("rb"):
        res.update(x_final=np.hstack([f.xn, f.xl]), R_final=f.R)
    else:
        res.update(x_final=f.x)
    return res


def config_of(case):
    return S.make_config(case["model"], case["N"], case["kind"], case["strategy"], case["thr"], SEED, 0)


def oracle_of(ob, case, order):
    """the C oracle for a case (with the case's own measurement likelihood installed)"""
    o = ob.OracleFilter(config_of(case), order)
    if case.get("user") and case["user"][0]:
        o.set_user_loglik(case["user"][0], case["user"][1])
    if case.get("noise"):
        o.set_user_noise(*case["noise"])
    if case.get("initial"):
        o.set_user_initial(*case["initial"])
    return o


def engine_of(case):
    """the HIP engine for a case: cases with a measurement likelihood of their own run a model compiled from the case's device snippet"""
    from llpf_amd import _capi
    if not case.get("user"):
        return _capi.FilterHandle(config_of(case))
    kind, par, src, qt = case["user"]
    m = S.Model.from_buffer_copy(bytes(case["model"]))
    m.model_id = _capi.model_compile(src, m.nx, m.ny)
    for i, v in enumerate(qt):
        m.qt[i] = v
    return _capi.FilterHandle(S.make_config(m, case["N"], case["kind"], case["strategy"], case["thr"], SEED, 0))
