# LLPFAmd.jl — thin `ccall` layer over libllpf_hip.so (include/llpf.h) exposing the hot-path verbs of
# LowLevelParticleFilters.jl.  Julia is not available in the build image, so this file is NOT exercised by
# the test-suite; every call below is mirrored one-to-one by lowlevelparticlefilters.jl_amd/_capi.py, which is.
#
# Usage (inside a session that has LowLevelParticleFilters loaded):
#     include("LLPFAmd.jl"); using .LLPFAmd
#     pf = GPUParticleFilter(1_000_000, LinearGaussianModel(A, B, C), df, dg, d0; resample_threshold = 0.1)
#     ll = loglik(pf, u, y);  sol = forward_trajectory(pf, u, y)
module LLPFAmd

using LinearAlgebra
export GPUFilterBank, loglik_multi, GPUParticleFilter, GPUAuxiliaryParticleFilter, LinearGaussianModel, QuadTankModel, RBLinearModel, RBBilinearModel, linear_state, GaussianSpec, smooth,
       reset!, predict!, correct!, update!, loglik, forward_trajectory, particles, weights, expweights,
       num_particles, index, effective_particles, shouldresample, weighted_mean

const LIB = get(ENV, "LLPF_HIP_LIB", joinpath(@__DIR__, "..", "libllpf_hip.so"))
const MAXD = 8

# ---- plain-data mirrors of include/llpf.h (field order and padding identical) ------------------------------
struct CGaussian
    dim::Int32
    kind::Int32                       # 0 ScalMat, 1 PDiagMat, 2 PDMat
    mu::NTuple{8,Float64}
    cov::NTuple{64,Float64}
end
struct CRBCoupling                                              # llpf_rb_coupling (LLPF_MODEL_RB_BILINEAR only; zeroed otherwise)
    nxl::Int32; fn_kind::Int32
    Al::NTuple{64,Float64}; Bl::NTuple{64,Float64}; Cl::NTuple{64,Float64}
    An::NTuple{160,Float64}                                   # An[0] constant term, An[1+k] multiplies xn[k]; 5 x (nxn x nxl, row-major, 32 slots)
end
const NOCOUPLING = CRBCoupling(0, 0, ntuple(_ -> 0.0, 64), ntuple(_ -> 0.0, 64), ntuple(_ -> 0.0, 64), ntuple(_ -> 0.0, 160))
struct CModel
    model_id::Int32; nx::Int32; nu::Int32; ny::Int32
    A::NTuple{64,Float64}; B::NTuple{64,Float64}; C::NTuple{64,Float64}
    qt::NTuple{16,Float64}
    supersample::Int32; nxn::Int32
    Ts::Float64
    df::CGaussian; dg::CGaussian; d0::CGaussian
    linear_noise::CGaussian; linear_initial::CGaussian      # Rao-Blackwellized models only (zeroed otherwise)
    rb::CRBCoupling
end
struct CConfig
    struct_size::UInt32; filter_kind::Int32
    n_particles::Int64
    resampling_strategy::Int32; device::Int32
    resample_threshold::Float64
    seed::UInt64
    model::CModel
end
struct CRunOutputs
    ll_steps::Ptr{Float64}; xmean::Ptr{Float64}; x_hist::Ptr{Float64}; w_hist::Ptr{Float64}; we_hist::Ptr{Float64}
end

pad(v, n) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, n)
rowmajor(M) = vec(permutedims(Matrix{Float64}(M)))          # Julia is column-major, the ABI is row-major

"Gaussian density N(mu, Sigma); Sigma::Real => ScalMat, ::AbstractVector => PDiagMat, ::AbstractMatrix => PDMat"
struct GaussianSpec
    mu::Vector{Float64}
    cov
end
function cgauss(g::GaussianSpec)
    n = length(g.mu)
    if g.cov isa Real
        CGaussian(n, 0, pad(g.mu, 8), pad([g.cov], 64))
    elseif g.cov isa AbstractVector
        CGaussian(n, 1, pad(g.mu, 8), pad(g.cov, 64))
    else
        CGaussian(n, 2, pad(g.mu, 8), pad(rowmajor(g.cov), 64))
    end
end

const NOGAUSS = CGaussian(0, 0, ntuple(_ -> 0.0, 8), ntuple(_ -> 0.0, 64))    # unused density slot

struct LinearGaussianModel; A; B; C; end                     # dynamics A*x .+ B*u, measurement C*x
"""Rao-Blackwellized model with constant matrices (reference src/rbpf.jl:92-98): xn' = Fn xn + Bn u + An xl + wn,
xl' = Al xl + Bl u + wl, y = Gn xn + Cl xl + e; R1l the covariance of wl, d0l the inner KalmanFilter's initial density."""
struct RBLinearModel; Fn; Bn; An; Al; Bl; Gn; Cl; R1l; d0l::GaussianSpec; end
struct QuadTankModel; consts::NTuple{16,Float64}; supersample::Int; end
QuadTankModel(; supersample = 2) = QuadTankModel(
    (1.6, 1.6, 9.81, 4.9, 4.9, 4.9, 4.9, 0.03, 0.03, 0.03, 0.03, 0.2, 0.2, 500.0, 2.0, 1e-3), supersample)

function cmodel(m::LinearGaussianModel, df, dg, d0, Ts)
    nx = size(m.A, 1); nu = size(m.B, 2); ny = size(m.C, 1)
    CModel(0, nx, nu, ny, pad(rowmajor(m.A), 64), pad(rowmajor(m.B), 64), pad(rowmajor(m.C), 64),
           ntuple(_ -> 0.0, 16), 1, 0, Ts, cgauss(df), cgauss(dg), cgauss(d0), NOGAUSS, NOGAUSS, NOCOUPLING)
end
cmodel(m::QuadTankModel, df, dg, d0, Ts) =
    CModel(1, 4, 2, 2, ntuple(_ -> 0.0, 64), ntuple(_ -> 0.0, 64), ntuple(_ -> 0.0, 64), m.consts,
           m.supersample, 0, Ts, cgauss(df), cgauss(dg), cgauss(d0), NOGAUSS, NOGAUSS, NOCOUPLING)
# RBPF: df = R1n, dg = R2, d0 = d0n (all of the nonlinear substate's dimension); A = [Fn An; 0 Al], B = [Bn; Bl], C = [Gn Cl]
function cmodel(m::RBLinearModel, df, dg, d0, Ts)
    nn = size(m.Fn, 1); nl = size(m.Al, 1); nu = size(m.Bn, 2); ny = size(m.Gn, 1)
    An = m.An === nothing ? zeros(nn, nl) : m.An
    Cl = m.Cl === nothing ? zeros(ny, nl) : m.Cl
    A = [m.Fn An; zeros(nl, nn) m.Al]; B = [m.Bn; m.Bl]; C = [m.Gn Cl]
    CModel(2, nn + nl, nu, ny, pad(rowmajor(A), 64), pad(rowmajor(B), 64), pad(rowmajor(C), 64), ntuple(_ -> 0.0, 16),
           1, nn, Ts, cgauss(df), cgauss(dg), cgauss(d0), cgauss(GaussianSpec(zeros(nl), Matrix{Float64}(m.R1l))), cgauss(m.d0l), NOCOUPLING)
end

"""Rao-Blackwellized model whose coupling depends on the nonlinear state (reference src/rbpf.jl:108: `An` a function of
x): An(xn) = An0 + sum_k xn[k] Ank[k]; every particle carries its own Kalman covariance (the reference's !singleR branches,
:176/:247).  `fn` is a LinearGaussianModel (Fn, Bn, Gn over xn) or a QuadTankModel (xn = the four levels)."""
struct RBBilinearModel; fn; An0; Ank::Vector; Al; Bl; Cl; R1l; d0l::GaussianSpec; end
function cmodel(m::RBBilinearModel, df, dg, d0, Ts)
    nn, nl = size(m.An0); ny = size(m.Cl, 1)
    an = zeros(160)
    an[1:nn*nl] = rowmajor(m.An0)
    for k in 1:nn; an[32k+1:32k+nn*nl] = rowmajor(m.Ank[k]); end
    quad = m.fn isa QuadTankModel
    base = quad ? cmodel(m.fn, df, dg, d0, Ts) : cmodel(m.fn, df, dg, d0, Ts)       # A, B, C (or qt, supersample) over xn
    nu = base.nu
    rb = CRBCoupling(nl, quad ? 1 : 0, pad(rowmajor(m.Al), 64), pad(nu > 0 ? rowmajor(m.Bl) : Float64[], 64), pad(rowmajor(m.Cl), 64), Tuple(an))
    CModel(3, nn, nu, ny, base.A, base.B, base.C, base.qt, base.supersample, nn, Ts, cgauss(df), cgauss(dg), cgauss(d0),
           cgauss(GaussianSpec(zeros(nl), Matrix{Float64}(m.R1l))), cgauss(m.d0l), rb)
end

check(rc) = rc == 0 || error("llpf status $rc: " * unsafe_string(ccall((:llpf_last_error, LIB), Cstring, ())))

# ---- the filter ------------------------------------------------------------------------------------------
mutable struct GPUParticleFilter
    h::Ptr{Cvoid}
    N::Int; nx::Int; nu::Int; ny::Int; Ts::Float64
    resample_threshold::Float64
end

"ParticleFilter(N, dynamics, measurement, df, dg, d0; ...) — reference src/PFtypes.jl:65-75"
function GPUParticleFilter(N::Integer, model, df::GaussianSpec, dg::GaussianSpec, d0::GaussianSpec;
                           resample_threshold = 0.1, stratified = false, seed = 0, Ts = 1.0, device = 0, advanced = false)
    cm = cmodel(model, df, dg, d0, Float64(Ts))
    cfg = Ref(CConfig(UInt32(sizeof(CConfig)), advanced ? 1 : 0, N, stratified ? 1 : 0, device,
                      resample_threshold, UInt64(seed), cm))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:llpf_create, LIB), Cint, (Ref{CConfig}, Ref{Ptr{Cvoid}}), cfg, h))
    nxp = cm.model_id == 3 ? cm.nx + cm.rb.nxl : cm.nx         # RBBilinearModel: particles, history and means are [xn; xl]
    pf = GPUParticleFilter(h[], N, nxp, cm.nu, cm.ny, Ts, resample_threshold)
    finalizer(p -> ccall((:llpf_destroy, LIB), Cint, (Ptr{Cvoid},), p.h), pf)
    pf
end

num_particles(pf::GPUParticleFilter) = pf.N
function index(pf::GPUParticleFilter)                          # index(pf) = state.t[], src/PFtypes.jl:314
    t = Ref{Int64}(0); check(ccall((:llpf_index, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), pf.h, t)); Int(t[])
end

"reset!(pf) — src/filtering.jl:4-14"
reset!(pf::GPUParticleFilter) = (check(ccall((:llpf_reset, LIB), Cint, (Ptr{Cvoid},), pf.h)); nothing)

"correct!(pf,u,y,p,t) -> (ll, 0) — src/filtering.jl:164-168; y === missing skips the weighting"
function correct!(pf::GPUParticleFilter, u, y, p = nothing, t = index(pf) * pf.Ts)
    ll = Ref{Float64}(0)
    uy = Vector{Float64}(u)
    yp = (y === missing || any(ismissing, y)) ? Ptr{Float64}(C_NULL) : pointer(Vector{Float64}(y))
    yv = yp == C_NULL ? Float64[] : Vector{Float64}(y)
    GC.@preserve uy yv check(ccall((:llpf_correct, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}),
                                   pf.h, uy, isempty(yv) ? C_NULL : pointer(yv), Float64(t), ll))
    ll[], 0
end

"predict!(pf,u,p,t) — src/filtering.jl:140-153"
function predict!(pf::GPUParticleFilter, u, p = nothing, t = index(pf) * pf.Ts)
    check(ccall((:llpf_predict, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64), pf.h, Vector{Float64}(u), Float64(t)))
end

"update!(pf,u,y,p,t) -> (ll, 0) — src/filtering.jl:181-185; also pf(u, y)"
function update!(pf::GPUParticleFilter, u, y, p = nothing, t = index(pf) * pf.Ts)
    ll = Ref{Float64}(0)
    check(ccall((:llpf_update, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}),
                pf.h, Vector{Float64}(u), Vector{Float64}(y), Float64(t), ll))
    ll[], 0
end
(pf::GPUParticleFilter)(u, y, p = nothing, t = index(pf) * pf.Ts) = update!(pf, u, y, p, t)

rows(v) = Matrix{Float64}(reduce(hcat, v))                    # Vector of vectors -> (dim x T): column-major == ABI row-major

function run!(pf::GPUParticleFilter, u, y, tindex0; history = false)
    T = length(y)
    U = rows(u); Y = rows(y)
    ll = Ref{Float64}(0)
    x = history ? Array{Float64}(undef, pf.nx, pf.N, T) : Float64[]
    w = history ? Array{Float64}(undef, pf.N, T) : Float64[]
    we = history ? Array{Float64}(undef, pf.N, T) : Float64[]
    outs = Ref(CRunOutputs(C_NULL, C_NULL, history ? pointer(x) : C_NULL, history ? pointer(w) : C_NULL, history ? pointer(we) : C_NULL))
    GC.@preserve U Y x w we check(ccall((:llpf_run, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ref{Float64}, Ref{CRunOutputs}),
        pf.h, U, Y, T, Float64(tindex0), ll, outs))
    ll[], x, w, we
end

"loglik(pf,u,y,p) — src/smoothing.jl:227-230 (reset!, then t = index(pf)*Ts starting at 1)"
loglik(pf::GPUParticleFilter, u, y, p = nothing) = (reset!(pf); run!(pf, u, y, 1.0)[1])

"forward_trajectory(pf,u,y,p) — src/filtering.jl:343-365; returns (x[nx,N,T], w[N,T], we[N,T], ll)"
function forward_trajectory(pf::GPUParticleFilter, u, y, p = nothing)
    reset!(pf)
    ll, x, w, we = run!(pf, u, y, 0.0; history = true)
    # reinterpret(reshape, SVector{nx,Float64}, x) gives the reference's N x T Matrix{SVector}
    (; x, w, we, ll, t = range(0, step = pf.Ts, length = length(y)))
end

# ---- AuxiliaryParticleFilter{ParticleFilter} (reference src/PFtypes.jl:38-49) -----------------------------------
"AuxiliaryParticleFilter(pf): the same device handle driven through the auxiliary verbs"
struct GPUAuxiliaryParticleFilter
    pf::GPUParticleFilter
end
Base.getproperty(a::GPUAuxiliaryParticleFilter, s::Symbol) = s === :pf ? getfield(a, :pf) : getproperty(getfield(a, :pf), s)
reset!(a::GPUAuxiliaryParticleFilter) = reset!(a.pf)
index(a::GPUAuxiliaryParticleFilter) = index(a.pf)

"correct!(pf::AuxiliaryParticleFilter,u,y,p,t) -> (ll, 0) — src/filtering.jl:170-174 (logsumexp! only)"
function correct!(a::GPUAuxiliaryParticleFilter, u, y, p = nothing, t = index(a) * a.Ts)
    ll = Ref{Float64}(0)
    check(ccall((:llpf_aux_correct, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), a.pf.h, ll))
    ll[], 0
end
"predict!(pf::AuxiliaryParticleFilter,u,y1,p,t) — src/filtering.jl:195-217"
function predict!(a::GPUAuxiliaryParticleFilter, u, y1, p = nothing, t = index(a) * a.Ts)
    yv = (y1 === missing || any(ismissing, y1)) ? Float64[] : Vector{Float64}(y1)
    GC.@preserve yv check(ccall((:llpf_aux_predict, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64),
                                a.pf.h, Vector{Float64}(u), isempty(yv) ? C_NULL : pointer(yv), Float64(t)))
end
"update!(pf::AuxiliaryParticleFilter,u,y,y1,p,t) -> (ll, 0) — src/filtering.jl:187-191; also pfa(u, y, y1)"
function update!(a::GPUAuxiliaryParticleFilter, u, y, y1, p = nothing, t = index(a) * a.Ts)
    ll_e = correct!(a, u, y, p, t)
    predict!(a, u, y1, p, t)
    ll_e
end
(a::GPUAuxiliaryParticleFilter)(u, y, y1, p = nothing, t = index(a) * a.Ts) = update!(a, u, y, y1, p, t)

function run_aux!(a::GPUAuxiliaryParticleFilter, u, y, mode; history = false)
    pf = a.pf
    T = length(y)
    U = rows(u); Y = rows(y)
    ll = Ref{Float64}(0)
    x = history ? Array{Float64}(undef, pf.nx, pf.N, T) : Float64[]
    w = history ? Array{Float64}(undef, pf.N, T) : Float64[]
    we = history ? Array{Float64}(undef, pf.N, T) : Float64[]
    outs = Ref(CRunOutputs(C_NULL, C_NULL, history ? pointer(x) : C_NULL, history ? pointer(w) : C_NULL, history ? pointer(we) : C_NULL))
    GC.@preserve U Y x w we check(ccall((:llpf_aux_run, LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Ref{Float64}, Ref{CRunOutputs}),
        pf.h, U, Y, T, Int32(mode), ll, outs))
    ll[], x, w, we
end
"loglik(pf::AuxiliaryParticleFilter,u,y,p) — src/smoothing.jl:232-236"
loglik(a::GPUAuxiliaryParticleFilter, u, y, p = nothing) = (reset!(a); run_aux!(a, u, y, 1)[1])
"forward_trajectory(pf::AuxiliaryParticleFilter,u,y,p) — src/filtering.jl:367-384"
function forward_trajectory(a::GPUAuxiliaryParticleFilter, u, y, p = nothing)
    reset!(a)
    ll, x, w, we = run_aux!(a, u, y, 0; history = true)
    (; x, w, we, ll, t = range(0, step = a.Ts, length = length(y)))
end

"xb, ll = smooth(pf, M, u, y) — src/smoothing.jl:103-143 (forward filtering, backward simulation); xb is nx x M x T"
function smooth(pf::GPUParticleFilter, M::Integer, u, y, p = nothing)
    sol = forward_trajectory(pf, u, y)
    T = length(y)
    U = rows(u)
    xb = Array{Float64}(undef, pf.nx, M, T)
    GC.@preserve U check(ccall((:llpf_smooth, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
        pf.h, M, U, T, sol.x, sol.w, sol.we, xb, C_NULL))
    xb, sol.ll
end

function getvec(sym, pf, n)
    out = Vector{Float64}(undef, n)
    check(ccall((sym, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), pf.h, out)); out
end
particles(pf::GPUParticleFilter) = reshape(getvec(:llpf_get_particles, pf, pf.N * pf.nx), pf.nx, pf.N)
"(xl [nxl x N], R [nxl x nxl x N]): fields xl, R of every RBParticle (src/rbpf.jl:1-5) of a filter built from an RBBilinearModel"
function linear_state(pf::GPUParticleFilter, nxl::Integer)
    xl = Matrix{Float64}(undef, nxl, pf.N); R = Array{Float64}(undef, nxl, nxl, pf.N)      # symmetric: row- and column-major agree
    check(ccall((:llpf_rb_get_linear_state, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), pf.h, xl, R))
    xl, R
end
weights(pf::GPUParticleFilter) = getvec(:llpf_get_weights, pf, pf.N)
expweights(pf::GPUParticleFilter) = getvec(:llpf_get_expweights, pf, pf.N)
weighted_mean(pf::GPUParticleFilter) = getvec(:llpf_weighted_mean, pf, pf.nx)
function effective_particles(pf::GPUParticleFilter)
    e = Ref{Float64}(0); check(ccall((:llpf_effective_particles, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), pf.h, e)); e[]
end
function shouldresample(pf::GPUParticleFilter)
    r = Ref{Int32}(0); check(ccall((:llpf_shouldresample, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), pf.h, r)); r[] != 0
end

# ---- banks of independent filters (parameter sweeps, Monte-Carlo replicas) ---------------------------------------
"`map(svec) do s; pfs = ParticleFilter(...); loglik(pfs, u, y); end` (reference test/runtests.jl:412-417) as one device job"
mutable struct GPUFilterBank
    h::Ptr{Cvoid}
    F::Int; N::Int; nx::Int; nu::Int; ny::Int
end
function GPUFilterBank(N::Integer, models::Vector, dfs::Vector{GaussianSpec}, dg::GaussianSpec, d0::GaussianSpec;
                       resample_threshold = 0.1, seed = 0, Ts = 1.0, device = 0)
    cms = [cmodel(models[k], dfs[k], dg, d0, Float64(Ts)) for k in eachindex(models)]
    cfg = Ref(CConfig(UInt32(sizeof(CConfig)), 0, N, 0, device, resample_threshold, UInt64(seed), cms[1]))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:llpf_bank_create, LIB), Cint, (Ref{CConfig}, Ptr{CModel}, Int32, Ref{Ptr{Cvoid}}), cfg, cms, length(cms), h))
    b = GPUFilterBank(h[], length(cms), N, cms[1].nx, cms[1].nu, cms[1].ny)
    finalizer(x -> ccall((:llpf_bank_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), b)
    b
end
"log-likelihood of every filter of the bank on shared data u, y (vectors of vectors)"
function loglik(b::GPUFilterBank, u, y)
    T = length(y)
    U = b.nu > 0 ? collect(reduce(hcat, u)) : zeros(0, T); Y = collect(reduce(hcat, y))    # column-major nu x T = row-major T x nu
    ll = zeros(b.F)
    check(ccall((:llpf_bank_reset, LIB), Cint, (Ptr{Cvoid},), b.h))
    check(ccall((:llpf_bank_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}),
                b.h, U, Y, T, 1.0, ll, C_NULL))
    ll
end
"as loglik, every filter on data of its own: U nu x T x F, Y ny x T x F (column-major = the ABI's [F][T][n])"
function loglik_multi(b::GPUFilterBank, U::Array{Float64,3}, Y::Array{Float64,3})
    T = size(Y, 2)
    ll = zeros(b.F); xm = zeros(b.nx, b.F, T)
    check(ccall((:llpf_bank_reset, LIB), Cint, (Ptr{Cvoid},), b.h))
    check(ccall((:llpf_bank_run_multi, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                b.h, U, Y, T, 1.0, ll, C_NULL, xm))
    ll, xm
end

end # module
