# LLPFAmd.jl — the MI355X engine (libllpf_hip.so, include/llpf.h) behind LowLevelParticleFilters.jl's own verbs.
#
# `GPUParticleFilter <: LowLevelParticleFilters.AbstractParticleFilter` (reference src/PFtypes.jl:2), and every verb
# below is a METHOD OF THE REFERENCE'S FUNCTION (`import LowLevelParticleFilters: reset!, predict!, correct!, ...`), each a
# thin `ccall`.  Code written against the reference keeps working with the constructor name swapped:
#
#     using LowLevelParticleFilters, LLPFAmd
#     dyn, meas = LinearDynamics(A, B), LinearMeasurement(C)          # callable: also valid arguments of ParticleFilter(...)
#     pf  = ParticleFilter(N, dyn, meas, df, dg, d0)                  # the reference, CPU
#     gpf = GPUParticleFilter(N, dyn, meas, df, dg, d0)               # this engine; same signature and keyword names
#     sol = forward_trajectory(gpf, u, y)                             # ::ParticleFilteringSolution
#     ll  = loglik(gpf, u, y);  gpf(u[1], y[1]);  weighted_mean(gpf);  mean_trajectory(gpf, u, y);  smooth(gpf, M, u, y)
#
# What cannot cross the C ABI is an arbitrary Julia closure: `dynamics` / `measurement` / `measurement_likelihood` must be
# one of the descriptor types defined here (they are callable on the host, so `simulate` and the reference's CPU filters
# accept them too); densities are Gaussians (`SimpleMvNormal`, `Distributions.MvNormal`, or `GaussianSpec`).
# Julia is not installed in the build image: this file is checked statically (tests/test_julia_struct_mirror.py: struct
# mirrors field by field against include/llpf.h, every ccall symbol against the header, every verb an imported reference
# function) and call for call by the ctypes twin lowlevelparticlefilters.jl_amd/_capi.py, which the test-suite runs.
module LLPFAmd

using LinearAlgebra
using Random
using StaticArrays
using Statistics: mean, cov
import LowLevelParticleFilters
import LowLevelParticleFilters: AbstractParticleFilter, ParticleFilteringSolution, PFstate, NullParameters,
    ResamplingStrategy, ResampleSystematic, ResampleStratified, ResampleResidual, SimpleMvNormal,
    reset!, predict!, correct!, update!, forward_trajectory, loglik, smooth, sample_state,
    particles, weights, expweights, state, num_particles, index, particletype, parameters,
    effective_particles, shouldresample, weighted_mean, weighted_cov, weighted_quantile,
    dynamics, measurement, measurement_likelihood, dynamics_density, measurement_density, initial_density,
    resample_threshold, resampling_strategy

export GPUParticleFilter, GPUAdvancedParticleFilter, GPUAuxiliaryParticleFilter, GPURBPF, GPUFilterBank, GPUMultiBank,
       LinearDynamics, LinearMeasurement, QuadTankDynamics, QuadTankMeasurement, GaussianLikelihood,
       RBLinearModel, RBBilinearModel, GaussianSpec, UserDynamics, UserMeasurement, UserLikelihood, UserNoise, UserInitial, linear_state, shared_covariance, loglik_multi, mbank_unique_id,
       seed!, ancestors, last_resampled, set_parameters!, quantile_trajectory, trace_dynamics, emit_user_model

const LIB = get(ENV, "LLPF_HIP_LIB", joinpath(@__DIR__, "..", "libllpf_hip.so"))
const MAXD = 16          # LLPF_MAX_DIM: states / outputs
const MAXU = 8           # LLPF_MAX_INPUTS
const ND2 = MAXD * MAXD  # slots of a dim x dim matrix

# ---- plain-data mirrors of include/llpf.h (field order and padding identical; checked by the static test) ------------
struct CGaussian                      # llpf_gaussian
    dim::Int32
    kind::Int32                       # 0 ScalMat, 1 PDiagMat, 2 PDMat
    mu::NTuple{16,Float64}
    cov::NTuple{256,Float64}
end
struct CRBCoupling                    # llpf_rb_coupling (LLPF_MODEL_RB_BILINEAR only; zeroed otherwise)
    nxl::Int32
    fn_kind::Int32
    Al::NTuple{64,Float64}
    Bl::NTuple{64,Float64}
    Cl::NTuple{64,Float64}
    An::NTuple{160,Float64}           # An[0] constant term, An[1+k] multiplies xn[k]; 5 x (nxn x nxl, row-major, 32 slots)
end
struct CModel                         # llpf_model
    model_id::Int32
    nx::Int32
    nu::Int32
    ny::Int32
    A::NTuple{256,Float64}
    B::NTuple{128,Float64}
    C::NTuple{256,Float64}
    qt::NTuple{16,Float64}
    supersample::Int32
    nxn::Int32
    Ts::Float64
    dynamics_density::CGaussian
    measurement_density::CGaussian
    initial_density::CGaussian
    linear_noise::CGaussian           # Rao-Blackwellized models only (zeroed otherwise)
    linear_initial::CGaussian
    rb::CRBCoupling
end
struct CConfig                        # llpf_config
    struct_size::UInt32
    filter_kind::Int32
    n_particles::Int64
    resampling_strategy::Int32
    device::Int32
    resample_threshold::Float64
    seed::UInt64
    model::CModel
end
struct CRunOutputs                    # llpf_run_outputs
    ll_steps::Ptr{Float64}
    xmean::Ptr{Float64}
    x_hist::Ptr{Float64}
    w_hist::Ptr{Float64}
    we_hist::Ptr{Float64}
    xcov::Ptr{Float64}
    xquant::Ptr{Float64}              # ABI minor 6: weighted_quantile per timestep from the run loop, [nq, nx, T] in Julia's order
    quant_p::Ptr{Float64}
    nq::Int32
    pad::Int32
end
CRunOutputs(ll, xm, x, w, we, xc) = CRunOutputs(ll, xm, x, w, we, xc, C_NULL, C_NULL, Int32(0), Int32(0))
struct CMBankInfo                     # llpf_mbank_info_t
    n_filters::Int32
    n_shards::Int32
    n_local_shards::Int32
    first_local_shard::Int32
    n_local_filters::Int32
    collective::Int32
    last_run_ms::Float64
    last_collective_ms::Float64
    resample_count::Int64
end

const ZERO64 = ntuple(_ -> 0.0, 64)
const ZEROA = ntuple(_ -> 0.0, ND2)
const ZEROB = ntuple(_ -> 0.0, MAXD * MAXU)
const NOCOUPLING = CRBCoupling(0, 0, ZERO64, ZERO64, ZERO64, ntuple(_ -> 0.0, 160))
const NOGAUSS = CGaussian(0, 0, ntuple(_ -> 0.0, MAXD), ZEROA)         # unused density slot

pad(v, n) = ntuple(i -> i <= length(v) ? Float64(v[i]) : 0.0, n)
rowmajor(M) = vec(permutedims(Matrix{Float64}(M)))                   # Julia is column-major, the ABI is row-major

check(rc) = rc == 0 || error("llpf status $rc: " * unsafe_string(ccall((:llpf_last_error, LIB), Cstring, ())))

# ---- densities ------------------------------------------------------------------------------------------------------
"Gaussian N(mu, Sigma); Sigma::Real => ScalMat(sigma^2), ::AbstractVector => PDiagMat(diagonal), ::AbstractMatrix => PDMat"
struct GaussianSpec
    mu::Vector{Float64}
    cov
end
Base.length(g::GaussianSpec) = length(g.mu)

# The three PDMats storage kinds keep their identity (their quadratic forms differ in operation order, reference
# src/utils.jl:110-113): a covariance with a scalar `value` field is a ScalMat, one with a `diag` field a PDiagMat /
# Diagonal, anything else is taken densely.
function covspec(S)
    S isa Real && return Float64(S)
    S isa UniformScaling && return Float64(S.λ)
    hasproperty(S, :value) && hasproperty(S, :dim) && return Float64(S.value)
    (S isa Diagonal || hasproperty(S, :diag)) && return Vector{Float64}(S.diag)
    S isa AbstractVector && return Vector{Float64}(S)
    return Matrix{Float64}(S)
end
gaussian_spec(g::GaussianSpec) = g
gaussian_spec(d::SimpleMvNormal) = GaussianSpec(Vector{Float64}(d.μ), covspec(d.Σ))
# Distributions.MvNormal (fields μ, Σ::AbstractPDMat) and anything else that answers mean / cov
function gaussian_spec(d)
    if hasproperty(d, :μ) && hasproperty(d, :Σ)
        return GaussianSpec(Vector{Float64}(d.μ), covspec(d.Σ))
    end
    GaussianSpec(Vector{Float64}(mean(d)), Matrix{Float64}(cov(d)))
end
function cgauss(d)
    g = gaussian_spec(d)
    n = length(g.mu)
    n <= MAXD || error("density dimension $n exceeds $MAXD")
    if g.cov isa Real
        CGaussian(n, 0, pad(g.mu, MAXD), pad([g.cov], ND2))
    elseif g.cov isa AbstractVector
        CGaussian(n, 1, pad(g.mu, MAXD), pad(g.cov, ND2))
    else
        CGaussian(n, 2, pad(g.mu, MAXD), pad(rowmajor(g.cov), ND2))
    end
end

# ---- model descriptors: callable on the host, selectable on the device ---------------------------------------------
"`dynamics(x,u,p,t) = A*x + B*u` (reference examples/example_lineargaussian.jl:28)"
struct LinearDynamics{TA,TB}
    A::TA
    B::TB
end
(f::LinearDynamics)(x, u, p, t) = isempty(f.B) ? f.A * x : f.A * x + f.B * u
(f::LinearDynamics)(x, u, p, t, noise) = f(x, u, p, t)
"`measurement(x,u,p,t) = C*x` (reference examples/example_lineargaussian.jl:29)"
struct LinearMeasurement{TC}
    C::TC
end
(g::LinearMeasurement)(x, u, p, t) = g.C * x
(g::LinearMeasurement)(x, u, p, t, noise) = g.C * x

"""Quad-tank process discretised with `rk4(f, Ts; supersample)` (reference examples/example_quadtank.jl:8-35,
src/utils.jl:220-237).  `consts` follows `LLPF_QT_*` of include/llpf.h."""
struct QuadTankDynamics
    consts::NTuple{16,Float64}
    supersample::Int
    Ts::Float64
end
QuadTankDynamics(; supersample = 2, Ts = 1.0) = QuadTankDynamics(
    (1.6, 1.6, 9.81, 4.9, 4.9, 4.9, 4.9, 0.03, 0.03, 0.03, 0.03, 0.2, 0.2, 500.0, 2.0, 1e-3), supersample, Ts)
function quadtank_rhs(c, h, u, t)
    k1, k2, g = c[1], c[2], c[3]
    A1, A2, A3, A4 = c[4], c[5], c[6], c[7]
    a1, a2, a3, a4 = c[8], c[9], c[10], c[11]
    γ1, γ2 = c[12], c[13]
    if t > c[14]
        a1 *= c[15]
    end
    ssqrt(x) = sqrt(max(x, zero(x)) + c[16])
    SA[-a1 / A1 * ssqrt(2g * h[1]) + a3 / A1 * ssqrt(2g * h[3]) + γ1 * k1 / A1 * u[1],
       -a2 / A2 * ssqrt(2g * h[2]) + a4 / A2 * ssqrt(2g * h[4]) + γ2 * k2 / A2 * u[2],
       -a3 / A3 * ssqrt(2g * h[3]) + (1 - γ2) * k2 / A3 * u[2],
       -a4 / A4 * ssqrt(2g * h[4]) + (1 - γ1) * k1 / A4 * u[1]]
end
function (f::QuadTankDynamics)(x, u, p, t)
    Tss = f.Ts / f.supersample
    for _ in 1:f.supersample
        f1 = quadtank_rhs(f.consts, x, u, t)
        f2 = quadtank_rhs(f.consts, x + Tss / 2 * f1, u, t + Tss / 2)
        f3 = quadtank_rhs(f.consts, x + Tss / 2 * f2, u, t + Tss / 2)
        f4 = quadtank_rhs(f.consts, x + Tss * f3, u, t + Tss)
        x = x + Tss / 6 * (f1 + 2f2 + 2f3 + f4)
        t += Tss
    end
    x
end
(f::QuadTankDynamics)(x, u, p, t, noise) = f(x, u, p, t)
"`measurement(x,u,p,t) = x[1:2]` of the quad-tank (reference examples/example_quadtank.jl:33)"
struct QuadTankMeasurement end
(::QuadTankMeasurement)(x, u, p, t) = SA[x[1], x[2]]
(::QuadTankMeasurement)(x, u, p, t, noise) = SA[x[1], x[2]]

"`measurement_likelihood(x,u,y,p,t) = logpdf(dg, y - measurement(x,u,p,t))`: the likelihood the AdvancedParticleFilter gets"
struct GaussianLikelihood{M,D}
    measurement::M
    dg::D
end
(l::GaussianLikelihood)(x, u, y, p, t) = LowLevelParticleFilters.extended_logpdf(l.dg, y .- l.measurement(x, u, p, t))

"""Rao-Blackwellized model with constant matrices (reference src/rbpf.jl:92-98): xn' = Fn xn + Bn u + An xl + wn,
xl' = Al xl + Bl u + wl, y = Gn xn + Cl xl + e; R1l the covariance of wl, d0l the inner KalmanFilter's initial density."""
struct RBLinearModel
    Fn; Bn; An; Al; Bl; Gn; Cl; R1l; d0l
end
"""Rao-Blackwellized model whose coupling depends on the nonlinear state (reference src/rbpf.jl:108: `An` a function of
x): An(xn) = An0 + sum_k xn[k] Ank[k]; every particle carries its own Kalman covariance (the reference's !singleR branches,
:176/:247).  `fn` / `gn` are a LinearDynamics / LinearMeasurement pair over xn or the quad-tank pair (xn = the four levels)."""
struct RBBilinearModel
    fn; gn; An0; Ank::Vector; Al; Bl; Cl; R1l; d0l
end

"""
    UserDynamics(device_src, nx, nu, ny; host = nothing, A = zeros(0,0), B = zeros(0,0), C = zeros(0,0), qt = zeros(16), supersample = 1)

A model the engine has no built-in for: HIP device source defining `struct UserModel` (prepare / dynamics / measurement, see
include/llpf.h `llpf_model_compile`), compiled for the GPU with hiprtc when the filter is constructed.  `A, B, C, qt` fill the
parameter block the snippet reads (`m->A`, ...).  `host` (optional) is the same dynamics as a Julia callable `(x,u,p,t)`, used
only by host-side `simulate`.  Pair it with `UserMeasurement(host)`."""
struct UserDynamics
    src::String
    nx::Int; nu::Int; ny::Int
    host
    A; B; C
    qt::Vector{Float64}
    supersample::Int
end
UserDynamics(src, nx, nu, ny; host = nothing, A = zeros(0, 0), B = zeros(0, 0), C = zeros(0, 0), qt = zeros(16), supersample = 1) =
    UserDynamics(String(src), nx, nu, ny, host, A, B, C, collect(Float64, qt), supersample)
(f::UserDynamics)(x, u, p, t) = f.host === nothing ? error("no host version of this device model was given") : f.host(x, u, p, t)
(f::UserDynamics)(x, u, p, t, noise) = f(x, u, p, t)
struct UserMeasurement
    host
end
(g::UserMeasurement)(x, u, p, t) = g.host === nothing ? error("no host version of this device model was given") : g.host(x, u, p, t)
(g::UserMeasurement)(x, u, p, t, noise) = g(x, u, p, t)
"""
    UserLikelihood(host = nothing)

`measurement_likelihood(x,u,y,p,t)` of an `AdvancedParticleFilter` (reference src/PFtypes.jl:226-239) — or `logpdf` of a measurement
density that is not Gaussian (ext/LowLevelParticleFiltersDistributionsExt.jl:80) — as the `loglik(x, y, t)` member of the paired
`UserDynamics` snippet.  Its `loglik_bound()` member declares the upper bound of the log-density the normalisation works against;
a snippet without one is normalised against the true maximum at every step (one more launch per step, ≈ 20 % slower).  `host`: the same
likelihood as a Julia callable `(x,u,y,p,t)`, for host-side use."""
struct UserLikelihood
    host
end
UserLikelihood() = UserLikelihood(nothing)
(l::UserLikelihood)(x, u, y, p, t) = l.host === nothing ? error("no host version of this device likelihood was given") : l.host(x, u, y, p, t)
"""
    UserNoise(gaussian = nothing; host = nothing)

`dynamics_density` of a filter whose model adds its own process noise: the `noise(x, fx, xi, uu, out)` member of the paired `UserDynamics`
snippet — the reference's `AdvancedParticleFilter` contract, `dynamics(x, u, p, t, noise = true)` (src/PFtypes.jl:242-259), or a
`ParticleFilter` whose `dynamics_density` is not Gaussian (`rand!(rng, d, noise)`, :122-139).  `gaussian`: the density the FFBS smoother
and the auxiliary filter's `add_noise!` keep using (default: standard normal)."""
struct UserNoise
    gaussian
    host
end
UserNoise(gaussian = nothing; host = nothing) = UserNoise(gaussian, host)
"""
    UserInitial(host = nothing)

`initial_density` of a filter whose model draws its own initial particles: the `initial(xi, uu, out)` member of the paired `UserDynamics`
snippet (`x_i = rand(rng, initial_density)`, reference src/filtering.jl:4-14)."""
struct UserInitial
    host
end
UserInitial() = UserInitial(nothing)
const TRAIT_LOGLIK, TRAIT_LOGLIK_BOUND, TRAIT_NOISE, TRAIT_INITIAL = Int32(1), Int32(2), Int32(4), Int32(8)
function cmodel(f::UserDynamics, ::UserMeasurement, df, dg, d0, Ts; user_likelihood::Bool = false)
    id = Ref{Int32}(-1)
    check(ccall((:llpf_model_compile, LIB), Cint, (Cstring, Int32, Int32, Ref{Int32}), f.src, f.nx, f.ny, id))
    # what the snippet defines must be what the filter was told to use: a missing member would silently fall back to the Gaussian
    # descriptor, a present one silently override it
    traits = Ref{Int32}(0)
    check(ccall((:llpf_model_traits, LIB), Cint, (Int32, Ref{Int32}), id[], traits))
    for (what, wanted, bit, member) in (("measurement likelihood", user_likelihood, TRAIT_LOGLIK, "loglik"),
                                        ("dynamics_density", df isa UserNoise, TRAIT_NOISE, "noise"),
                                        ("initial_density", d0 isa UserInitial, TRAIT_INITIAL, "initial"))
        wanted && traits[] & bit == 0 && error("the $what is a User* descriptor but the snippet defines no `$member` member")
        !wanted && traits[] & bit != 0 && error("the snippet defines `$member`, which would override the Gaussian $what: pass the matching User* descriptor")
    end
    dfg = df isa UserNoise ? (df.gaussian === nothing ? GaussianSpec(zeros(f.nx), 1.0) : df.gaussian) : df
    d0g = d0 isa UserInitial ? GaussianSpec(zeros(f.nx), 1.0) : d0
    CModel(id[], f.nx, f.nu, f.ny, pad(isempty(f.A) ? Float64[] : rowmajor(f.A), ND2), pad(isempty(f.B) ? Float64[] : rowmajor(f.B), MAXD * MAXU),
           pad(isempty(f.C) ? Float64[] : rowmajor(f.C), ND2), pad(f.qt, 16), f.supersample, 0, Ts, cgauss(dfg), cgauss(dg), cgauss(d0g),
           NOGAUSS, NOGAUSS, NOCOUPLING)
end

model_dims(f::LinearDynamics, g::LinearMeasurement) = (size(f.A, 1), size(f.B, 2), size(g.C, 1))
model_dims(::QuadTankDynamics, ::QuadTankMeasurement) = (4, 2, 2)

function cmodel(f::LinearDynamics, g::LinearMeasurement, df, dg, d0, Ts)
    nx, nu, ny = model_dims(f, g)
    CModel(0, nx, nu, ny, pad(rowmajor(f.A), ND2), pad(nu > 0 ? rowmajor(f.B) : Float64[], MAXD * MAXU), pad(rowmajor(g.C), ND2),
           ntuple(_ -> 0.0, 16), 1, 0, Ts, cgauss(df), cgauss(dg), cgauss(d0), NOGAUSS, NOGAUSS, NOCOUPLING)
end
cmodel(f::QuadTankDynamics, ::QuadTankMeasurement, df, dg, d0, Ts) =
    CModel(1, 4, 2, 2, ZEROA, ZEROB, ZEROA, f.consts, f.supersample, 0, Ts, cgauss(df), cgauss(dg), cgauss(d0),
           NOGAUSS, NOGAUSS, NOCOUPLING)
# RBPF: df = R1n, dg = R2, d0 = d0n (all of the nonlinear substate's dimension); A = [Fn An; 0 Al], B = [Bn; Bl], C = [Gn Cl]
function cmodel(m::RBLinearModel, ::Nothing, df, dg, d0, Ts)
    nn = size(m.Fn, 1); nl = size(m.Al, 1); nu = size(m.Bn, 2); ny = size(m.Gn, 1)
    An = m.An === nothing ? zeros(nn, nl) : m.An
    Cl = m.Cl === nothing ? zeros(ny, nl) : m.Cl
    A = [m.Fn An; zeros(nl, nn) m.Al]; B = [m.Bn; m.Bl]; C = [m.Gn Cl]
    CModel(2, nn + nl, nu, ny, pad(rowmajor(A), ND2), pad(rowmajor(B), MAXD * MAXU), pad(rowmajor(C), ND2), ntuple(_ -> 0.0, 16),
           1, nn, Ts, cgauss(df), cgauss(dg), cgauss(d0), cgauss(GaussianSpec(zeros(nl), Matrix{Float64}(m.R1l))), cgauss(m.d0l), NOCOUPLING)
end
function cmodel(m::RBBilinearModel, ::Nothing, df, dg, d0, Ts)
    nn, nl = size(m.An0); ny = size(m.Cl, 1)
    an = zeros(160)
    an[1:nn*nl] = rowmajor(m.An0)
    for k in 1:nn
        an[32k+1:32k+nn*nl] = rowmajor(m.Ank[k])
    end
    base = cmodel(m.fn, m.gn, df, dg, d0, Ts)                         # A, B, C (or qt, supersample) over xn
    nu = Int(base.nu)
    rb = CRBCoupling(nl, m.fn isa QuadTankDynamics ? 1 : 0, pad(rowmajor(m.Al), 64), pad(nu > 0 ? rowmajor(m.Bl) : Float64[], 64),
                     pad(rowmajor(m.Cl), 64), Tuple(an))
    CModel(3, nn, nu, ny, base.A, base.B, base.C, base.qt, base.supersample, nn, Ts, cgauss(df), cgauss(dg), cgauss(d0),
           cgauss(GaussianSpec(zeros(nl), Matrix{Float64}(m.R1l))), cgauss(m.d0l), rb)
end

strategy_code(::Type{ResampleSystematic}) = Int32(0)          # reference src/LowLevelParticleFilters.jl:43-46
strategy_code(::Type{ResampleStratified}) = Int32(1)
strategy_code(::Type{ResampleResidual}) = Int32(2)

# ---- the filter ------------------------------------------------------------------------------------------------------
"""
    GPUParticleFilter(N, dynamics, measurement, dynamics_density, measurement_density, initial_density;
                      resample_threshold = 0.1, resampling_strategy = ResampleSystematic, p = NullParameters(), Ts = 1.0,
                      seed = 0, device = 0)

`ParticleFilter(N, dynamics, measurement, df, dg, d0; ...)` of the reference (src/PFtypes.jl:21-36, 65-75) on the GPU:
same positional arguments, same keyword names and defaults (`seed` keys the engine's generator, Philox4x32 (7 rounds); `rng` is
only used by the host-side `simulate`; `threads` has no meaning here).  `NX` is the dimension of a particle as the accessors return it."""
mutable struct GPUParticleFilter{NX,RST<:DataType,FT,GT,GLT,FDT,GDT,IDT,P,RNGT} <: AbstractParticleFilter
    h::Ptr{Cvoid}
    N::Int
    nx::Int
    nu::Int
    ny::Int
    Ts::Float64
    resample_threshold::Float64
    resampling_strategy::RST
    dynamics::FT
    measurement::GT
    measurement_likelihood::GLT
    dynamics_density::FDT
    measurement_density::GDT
    initial_density::IDT
    p::P
    rng::RNGT
    advanced::Bool
end

function create_filter(N, cm::CModel, advanced, thr, rst, seed, device, dyn, meas, lik, df, dg, d0, p, rng = Xoshiro())
    cfg = Ref(CConfig(UInt32(sizeof(CConfig)), advanced ? 1 : 0, N, strategy_code(rst), device, thr, UInt64(seed), cm))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:llpf_create, LIB), Cint, (Ref{CConfig}, Ref{Ptr{Cvoid}}), cfg, h))
    nxp = cm.model_id == 3 ? Int(cm.nx + cm.rb.nxl) : Int(cm.nx)      # RBBilinearModel: particles, history and means are [xn; xl]
    pf = GPUParticleFilter{nxp,typeof(rst),typeof(dyn),typeof(meas),typeof(lik),typeof(df),typeof(dg),typeof(d0),typeof(p),typeof(rng)}(
        h[], N, nxp, Int(cm.nu), Int(cm.ny), cm.Ts, thr, rst, dyn, meas, lik, df, dg, d0, p, rng, advanced)
    finalizer(x -> ccall((:llpf_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), pf)
    pf
end

function GPUParticleFilter(N::Integer, dynamics, measurement, dynamics_density, measurement_density, initial_density;
                           resample_threshold = 0.1, resampling_strategy::Type{<:ResamplingStrategy} = ResampleSystematic,
                           p = NullParameters(), Ts = 1.0, seed = 0, device = 0, rng = Xoshiro(), kwargs...)
    cm = cmodel(dynamics, measurement, dynamics_density, measurement_density, initial_density, Float64(Ts))
    create_filter(N, cm, false, Float64(resample_threshold), resampling_strategy, seed, device, dynamics, measurement,
                  GaussianLikelihood(measurement, measurement_density), dynamics_density, measurement_density, initial_density, p, rng)
end

"""
    GPUAdvancedParticleFilter(N, dynamics, measurement, measurement_likelihood::GaussianLikelihood, dynamics_density, initial_density; ...)

`AdvancedParticleFilter(N, dynamics, measurement, measurement_likelihood, dynamics_density, initial_density; ...)` of the
reference (src/PFtypes.jl:162-210; default `resample_threshold = 0.5`).  The dynamics' own noise (`noise = true`) is
`dynamics_density`, which therefore must be given."""
function GPUAdvancedParticleFilter(N::Integer, dynamics, measurement, measurement_likelihood::GaussianLikelihood,
                                   dynamics_density, initial_density; resample_threshold = 0.5,
                                   resampling_strategy::Type{<:ResamplingStrategy} = ResampleSystematic,
                                   p = NullParameters(), Ts = 1.0, seed = 0, device = 0, rng = Xoshiro(), kwargs...)
    dg = measurement_likelihood.dg
    cm = cmodel(dynamics, measurement, dynamics_density, dg, initial_density, Float64(Ts))
    create_filter(N, cm, true, Float64(resample_threshold), resampling_strategy, seed, device, dynamics, measurement,
                  measurement_likelihood, dynamics_density, dg, initial_density, p, rng)
end

# a likelihood of the user's own: the descriptor's Gaussian is not used by such a model (k_user_bound puts the declared bound in its place)
function GPUAdvancedParticleFilter(N::Integer, dynamics::UserDynamics, measurement::UserMeasurement, measurement_likelihood::UserLikelihood,
                                   dynamics_density, initial_density; resample_threshold = 0.5,
                                   resampling_strategy::Type{<:ResamplingStrategy} = ResampleSystematic,
                                   p = NullParameters(), Ts = 1.0, seed = 0, device = 0, rng = Xoshiro(), kwargs...)
    dg = GaussianSpec(zeros(dynamics.ny), 1.0)
    cm = cmodel(dynamics, measurement, dynamics_density, dg, initial_density, Float64(Ts); user_likelihood = true)
    create_filter(N, cm, true, Float64(resample_threshold), resampling_strategy, seed, device, dynamics, measurement,
                  measurement_likelihood, dynamics_density, dg, initial_density, p, rng)
end

"""
    GPURBPF(N, model::Union{RBLinearModel,RBBilinearModel}, R1n, R2, d0n; ...)

`RBPF(N, kf, dynamics, nl_measurement_model, R1n, d0n; An, ...)` of the reference (src/rbpf.jl:63-144) with the Kalman
filter `kf`, the nonlinear parts and the coupling collected in `model`; driven by the ordinary verbs."""
function GPURBPF(N::Integer, model, R1n, R2, d0n; resample_threshold = 0.1,
                 resampling_strategy::Type{<:ResamplingStrategy} = ResampleSystematic, p = NullParameters(), Ts = 1.0,
                 seed = 0, device = 0)
    cm = cmodel(model, nothing, R1n, R2, d0n, Float64(Ts))
    create_filter(N, cm, false, Float64(resample_threshold), resampling_strategy, seed, device, model, nothing, nothing, R1n, R2, d0n, p)
end

const GPF = GPUParticleFilter

"""
    set_parameters!(pf; dynamics, measurement, dynamics_density, measurement_density, initial_density)

New parameters for an existing GPU filter — same model family and dimensions, densities of the types the filter was built with — without
reallocating anything on the device (`llpf_set_model`): what the reference's `filter_from_parameters(θ, pf)` of `log_likelihood_fun` /
`metropolis` (src/smoothing.jl:266-283, 311-330) is handed the old filter for.  Both drivers are generic over `AbstractParticleFilter`
and run unchanged on a `GPUParticleFilter` (`loglik` is a method of this module).  Returns `pf`.

    filter_from_parameters(θ, pf = nothing) = pf === nothing ?
        GPUParticleFilter(N, dyn, meas, GaussianSpec(zeros(2), exp(2θ[1])), GaussianSpec(zeros(1), exp(2θ[2])), d0) :
        set_parameters!(pf; dynamics_density = GaussianSpec(zeros(2), exp(2θ[1])), measurement_density = GaussianSpec(zeros(1), exp(2θ[2])))
"""
function set_parameters!(pf::GPUParticleFilter; dynamics = pf.dynamics, measurement = pf.measurement, dynamics_density = pf.dynamics_density,
                         measurement_density = pf.measurement_density, initial_density = pf.initial_density)
    cm = pf.measurement_likelihood isa UserLikelihood ?
        cmodel(dynamics, measurement, dynamics_density, measurement_density, initial_density, pf.Ts; user_likelihood = true) :
        cmodel(dynamics, measurement, dynamics_density, measurement_density, initial_density, pf.Ts)
    check(ccall((:llpf_set_model, LIB), Cint, (Ptr{Cvoid}, Ref{CModel}), pf.h, Ref(cm)))
    pf.dynamics = dynamics; pf.measurement = measurement
    pf.dynamics_density = dynamics_density; pf.measurement_density = measurement_density; pf.initial_density = initial_density
    if pf.measurement_likelihood isa GaussianLikelihood
        pf.measurement_likelihood = GaussianLikelihood(measurement, measurement_density)
    end
    pf
end
# `pf.state` is materialised from the device on demand; every other property is a field (this method is more specific than
# the reference's getproperty(::AbstractParticleFilter, ...) of src/PFtypes.jl:84-99)
Base.getproperty(pf::GPF, s::Symbol) = s === :state ? state(pf) : getfield(pf, s)
Base.propertynames(pf::GPF) = (fieldnames(typeof(pf))..., :state)

parameters(pf::GPF) = getfield(pf, :p)
num_particles(pf::GPF) = getfield(pf, :N)
particletype(::GPF{NX}) where {NX} = SVector{NX,Float64}
dynamics(pf::GPF) = getfield(pf, :dynamics)
measurement(pf::GPF) = getfield(pf, :measurement)
measurement_likelihood(pf::GPF) = getfield(pf, :measurement_likelihood)
dynamics_density(pf::GPF) = getfield(pf, :dynamics_density)
measurement_density(pf::GPF) = getfield(pf, :measurement_density)
initial_density(pf::GPF) = getfield(pf, :initial_density)
resample_threshold(pf::GPF) = getfield(pf, :resample_threshold)
resampling_strategy(pf::GPF) = getfield(pf, :resampling_strategy)
# host-side simulate(pf, T, du) of the reference (src/filtering.jl:457-477) works on the descriptors, which are callable
sample_state(pf::GPF, x, u, p, t; noise = true) = dynamics(pf)(x, u, p, t) + noise * rand(getfield(pf, :rng), dynamics_density(pf))

"index(pf) = state.t[] — src/PFtypes.jl:314"
function index(pf::GPF)
    t = Ref{Int64}(0)
    check(ccall((:llpf_index, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}), pf.h, t))
    Int(t[])
end

"reset!(pf) — src/filtering.jl:4-14"
function reset!(pf::GPF)
    check(ccall((:llpf_reset, LIB), Cint, (Ptr{Cvoid},), pf.h))
    nothing
end
"re-key the engine's Philox generator (the reference never seeds `pf.rng`; use this for common random numbers)"
seed!(pf::GPF, s::Integer) = (check(ccall((:llpf_seed, LIB), Cint, (Ptr{Cvoid}, UInt64), pf.h, UInt64(s))); pf)

# a measurement / input as the ABI wants it: a Float64 vector, or nothing when it is `missing` (=> NULL, src/PFtypes.jl:109)
ismissingy(y) = y === missing || y === nothing || any(ismissing, y)
fvec(v) = v === nothing ? Float64[] : collect(Float64, v)
ptr_or_null(v::Vector{Float64}) = isempty(v) ? Ptr{Float64}(C_NULL) : pointer(v)

"correct!(pf,u,y,p,t) -> (ll, 0) — src/filtering.jl:164-168; a `missing` y skips the weighting, logsumexp! still runs"
function correct!(pf::GPF, u, y, p = parameters(pf), t = index(pf) * pf.Ts)
    ll = Ref{Float64}(0)
    uv = fvec(u)
    yv = ismissingy(y) ? Float64[] : fvec(y)
    GC.@preserve uv yv check(ccall((:llpf_correct, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}),
                                   pf.h, ptr_or_null(uv), ptr_or_null(yv), Float64(t), ll))
    ll[], 0
end

"predict!(pf,u,p,t) — src/filtering.jl:140-153"
function predict!(pf::GPF, u, p = parameters(pf), t = index(pf) * pf.Ts)
    uv = fvec(u)
    GC.@preserve uv check(ccall((:llpf_predict, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Float64), pf.h, ptr_or_null(uv), Float64(t)))
    nothing
end

"update!(pf,u,y,p,t) -> (ll, 0) — src/filtering.jl:181-185: correct! then predict!, one ccall"
function update!(pf::GPF, u, y, p = parameters(pf), t = index(pf) * pf.Ts)
    ll = Ref{Float64}(0)
    uv = fvec(u)
    yv = ismissingy(y) ? Float64[] : fvec(y)
    GC.@preserve uv yv check(ccall((:llpf_update, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}),
                                   pf.h, ptr_or_null(uv), ptr_or_null(yv), Float64(t), ll))
    ll[], 0
end
"pf(u, y, p, t) = update!(pf, u, y, p, t) — src/filtering.jl:238,240"
(pf::GPUParticleFilter)(u, y, p = parameters(pf), t = index(pf) * pf.Ts) = update!(pf, u, y, p, t)

# Vector of vectors -> (dim x T) matrix, column-major == the ABI's row-major T x dim; a missing measurement becomes a NaN row
function rows(v, dim)
    M = Matrix{Float64}(undef, dim, length(v))
    for (k, vk) in enumerate(v)
        if ismissingy(vk)
            M[:, k] .= NaN
        else
            M[:, k] .= vk
        end
    end
    M
end

function run!(pf::GPF, u, y, tindex0; history = false)
    T = length(y)
    U = rows(u, pf.nu)
    Y = rows(y, pf.ny)
    ll = Ref{Float64}(0)
    x = history ? Array{Float64}(undef, pf.nx, pf.N, T) : Array{Float64}(undef, 0, 0, 0)
    w = history ? Array{Float64}(undef, pf.N, T) : Array{Float64}(undef, 0, 0)
    we = history ? Array{Float64}(undef, pf.N, T) : Array{Float64}(undef, 0, 0)
    GC.@preserve U Y x w we begin
        outs = Ref(CRunOutputs(C_NULL, C_NULL, history ? pointer(x) : C_NULL, history ? pointer(w) : C_NULL, history ? pointer(we) : C_NULL, C_NULL))
        check(ccall((:llpf_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ref{Float64}, Ref{CRunOutputs}),
                    pf.h, pf.nu > 0 ? pointer(U) : C_NULL, pointer(Y), T, Float64(tindex0), ll, outs))
    end
    ll[], x, w, we
end

"""quantile_trajectory(pf, u, y, q) -> (ll, Q): forward_trajectory's loop (src/filtering.jl:351-363) with
`weighted_quantile(x[:,t], we[:,t], q)` (src/filtering.jl:583-595) of every timestep computed on the device inside the run loop
(llpf_run's xquant output: a radix selection over the exp-weights, no history crosses the bus).  Q[t][i] is the vector of the
length(q) quantiles of state i at step t — the reference's nesting."""
function quantile_trajectory(pf::GPF, u::AbstractVector, y::AbstractVector, q)
    reset!(pf)
    T = length(y)
    U = rows(u, pf.nu)
    Y = rows(y, pf.ny)
    qq = collect(Float64, q)
    out = Array{Float64}(undef, length(qq), pf.nx, T)
    ll = Ref{Float64}(0)
    GC.@preserve U Y qq out begin
        outs = Ref(CRunOutputs(C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, pointer(out), pointer(qq), Int32(length(qq)), Int32(0)))
        check(ccall((:llpf_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ref{Float64}, Ref{CRunOutputs}),
                    pf.h, pf.nu > 0 ? pointer(U) : C_NULL, pointer(Y), T, 0.0, ll, outs))
    end
    ll[], [[out[:, i, t] for i in 1:pf.nx] for t in 1:T]
end

# history [nx, N, T] -> the reference's N x T Matrix{SVector{nx,Float64}} (src/filtering.jl:347), without copying
svec_history(x::Array{Float64,3}, ::Val{NX}) where {NX} = reshape(reinterpret(SVector{NX,Float64}, vec(x)), size(x, 2), size(x, 3))

"loglik(pf,u,y,p) — src/smoothing.jl:227-230 (reset!, then T fused correct!/predict! steps with t = index(pf)*Ts starting at 1)"
function loglik(pf::GPF, u, y, p = parameters(pf))
    reset!(pf)
    run!(pf, u, y, 1.0)[1]
end

no_cb(args...) = nothing
"""forward_trajectory(pf,u,y,p) -> ParticleFilteringSolution — src/filtering.jl:343-365, src/solutions.jl:334-345.
Without callbacks the T steps run as one device job; with any of the reference's callbacks the reference's own loop is
used (it only needs the verbs below), one step per ccall."""
function forward_trajectory(pf::GPF{NX}, u::AbstractVector, y::AbstractVector, p = parameters(pf);
                            pre_correct_cb = no_cb, pre_predict_cb = no_cb, post_predict_cb = no_cb, post_correct_cb = no_cb) where {NX}
    if !(pre_correct_cb === no_cb && pre_predict_cb === no_cb && post_predict_cb === no_cb && post_correct_cb === no_cb)
        return invoke(forward_trajectory, Tuple{Any,AbstractVector,AbstractVector,Any}, pf, u, y, p;
                      pre_correct_cb, pre_predict_cb, post_predict_cb, post_correct_cb)
    end
    reset!(pf)
    ll, x, w, we = run!(pf, u, y, 0.0; history = true)
    ParticleFilteringSolution(pf, u, y, svec_history(x, Val(NX)), w, we, ll)
end

function getvec(sym, h, n)
    out = Vector{Float64}(undef, n)
    check(ccall((sym, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), h, out))
    out
end
"particles(pf)::Vector{SVector{nx,Float64}} — src/PFtypes.jl:296"
particles(pf::GPF{NX}) where {NX} = collect(reinterpret(SVector{NX,Float64}, getvec(:llpf_get_particles, pf.h, pf.N * NX)))
weights(pf::GPF) = getvec(:llpf_get_weights, pf.h, pf.N)
expweights(pf::GPF) = getvec(:llpf_get_expweights, pf.h, pf.N)
weighted_mean(pf::GPF) = getvec(:llpf_weighted_mean, pf.h, pf.nx)
"weighted_cov of the CURRENT particles and weights, on the device: one time step of the reference's weighted_cov(x, we) (src/filtering.jl:571-581)"
weighted_cov(pf::GPF) = reshape(getvec(:llpf_weighted_cov, pf.h, pf.nx * pf.nx), pf.nx, pf.nx)
"""
    weighted_quantile(pf, q)

Weighted quantile(s) `q` of the CURRENT particles and weights per state dimension, sorted and summed on the device: one time step of the
reference's `weighted_quantile(x, we, q)` (src/filtering.jl:583-595, StatsBase's `quantile(v, ProbabilityWeights(we), q)`).  A vector of
length nx for a scalar `q`, an nx x length(q) matrix for a vector.
"""
function weighted_quantile(pf::GPF, q::AbstractVector{<:Real})
    qq = collect(Float64, q)
    out = Matrix{Float64}(undef, pf.nx, length(qq))      # the ABI writes [nq][nx] row-major = nx x nq column-major
    check(ccall((:llpf_weighted_quantile, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}), pf.h, qq, Int32(length(qq)), out))
    out
end
weighted_quantile(pf::GPF, q::Real) = vec(weighted_quantile(pf, [q]))
"state(pf).j, 1-based — src/PFtypes.jl:14"
function ancestors(pf::GPF)
    j = Vector{Int64}(undef, pf.N)
    check(ccall((:llpf_get_ancestors, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), pf.h, j))
    j .+= 1
end
function effective_particles(pf::GPF)
    e = Ref{Float64}(0)
    check(ccall((:llpf_effective_particles, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), pf.h, e))
    e[]
end
function shouldresample(pf::GPF)
    r = Ref{Int32}(0)
    check(ccall((:llpf_shouldresample, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), pf.h, r))
    r[] != 0
end
function last_resampled(pf::GPF)
    r = Ref{Int32}(0)
    check(ccall((:llpf_last_resampled, LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), pf.h, r))
    r[] != 0
end
"state(pf)::PFstate — a snapshot copied from the device (src/PFtypes.jl:8-17); writing to it does not change the filter"
function state(pf::GPF)
    x = particles(pf)
    m = Ref{Float64}(0)
    check(ccall((:llpf_maxw, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), pf.h, m))
    bins = Vector{Float64}(undef, pf.N)
    check(ccall((:llpf_get_bins, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), pf.h, bins))
    PFstate(x, copy(x), weights(pf), expweights(pf), Ref(m[]), ancestors(pf), bins, Ref(index(pf)))
end

"xb, ll = smooth(pf, M, u, y, p) — src/smoothing.jl:103-143 (forward filtering, backward simulation); xb is M x T of particles"
function smooth(pf::GPF{NX}, M::Integer, u, y, p = parameters(pf)) where {NX}
    reset!(pf)
    ll, x, w, we = run!(pf, u, y, 0.0; history = true)
    T = length(y)
    U = rows(u, pf.nu)
    xb = Array{Float64}(undef, NX, M, T)
    GC.@preserve U x w we xb check(ccall((:llpf_smooth, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
        pf.h, M, pf.nu > 0 ? pointer(U) : C_NULL, T, x, w, we, xb, C_NULL))
    svec_history(xb, Val(NX)), ll
end

"(xl [nxl x N], R [nxl x nxl x N]): fields xl, R of every RBParticle (src/rbpf.jl:1-5) of a filter built from an RBBilinearModel"
function linear_state(pf::GPF)
    nxl = size(pf.dynamics.Al, 1)
    xl = Matrix{Float64}(undef, nxl, pf.N)
    R = Array{Float64}(undef, nxl, nxl, pf.N)                      # symmetric: row- and column-major agree
    check(ccall((:llpf_rb_get_linear_state, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), pf.h, xl, R))
    xl, R
end
"x[1].R of a filter built from an RBLinearModel: the covariance all particles share (src/rbpf.jl:176,247)"
function shared_covariance(pf::GPF)
    nxl = size(pf.dynamics.Al, 1)
    R = Matrix{Float64}(undef, nxl, nxl)
    check(ccall((:llpf_rb_get_covariance, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), pf.h, R))
    R
end

# ---- AuxiliaryParticleFilter{ParticleFilter} (reference src/PFtypes.jl:38-49) ----------------------------------------
"GPUAuxiliaryParticleFilter(pf) / GPUAuxiliaryParticleFilter(args...; kwargs...): the same device handle driven through the auxiliary verbs\n(over a GPUParticleFilter: src/filtering.jl:195-217; over a GPUAdvancedParticleFilter: :219-234, look-ahead resampling + re-propagation)"
struct GPUAuxiliaryParticleFilter{T<:GPUParticleFilter} <: AbstractParticleFilter
    pf::T
end
GPUAuxiliaryParticleFilter(args...; kwargs...) = GPUAuxiliaryParticleFilter(GPUParticleFilter(args...; kwargs...))
const GAPF = GPUAuxiliaryParticleFilter
Base.getproperty(a::GAPF, s::Symbol) = s === :pf ? getfield(a, :pf) : getproperty(getfield(a, :pf), s)
parameters(a::GAPF) = parameters(a.pf)
# the reference forwards these to the wrapped filter (src/PFtypes.jl:299)
for f in (:state, :particles, :weights, :expweights, :reset!, :weighted_mean, :index, :num_particles, :particletype, :dynamics,
          :measurement, :dynamics_density, :measurement_density, :initial_density, :resample_threshold, :resampling_strategy,
          :effective_particles, :shouldresample)
    @eval $f(a::GAPF) = $f(getfield(a, :pf))
end

"correct!(pf::AuxiliaryParticleFilter,u,y,p,t) -> (ll, 0) — src/filtering.jl:170-174 (logsumexp! only)"
function correct!(a::GAPF, u, y, p = parameters(a), t = index(a) * a.Ts)
    ll = Ref{Float64}(0)
    check(ccall((:llpf_aux_correct, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), a.pf.h, ll))
    ll[], 0
end
"predict!(pf::AuxiliaryParticleFilter,u,y1,p,t) — src/filtering.jl:195-217"
function predict!(a::GAPF, u, y1, p = parameters(a), t = index(a) * a.Ts)
    uv = fvec(u)
    yv = ismissingy(y1) ? Float64[] : fvec(y1)
    GC.@preserve uv yv check(ccall((:llpf_aux_predict, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64),
                                   a.pf.h, ptr_or_null(uv), ptr_or_null(yv), Float64(t)))
    nothing
end
"update!(pf::AuxiliaryParticleFilter,u,y,y1,p,t) -> (ll, 0) — src/filtering.jl:187-191"
function update!(a::GAPF, u, y, y1, p = parameters(a), t = index(a) * a.Ts)
    ll = Ref{Float64}(0)
    uv = fvec(u)
    yv = ismissingy(y1) ? Float64[] : fvec(y1)
    GC.@preserve uv yv check(ccall((:llpf_aux_update, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}),
                                   a.pf.h, ptr_or_null(uv), ptr_or_null(yv), Float64(t), ll))
    ll[], 0
end
"pfa(u, y, y1, p, t) — src/filtering.jl:239"
(a::GPUAuxiliaryParticleFilter)(u, y, y1, p = parameters(a), t = index(a) * a.Ts) = update!(a, u, y, y1, p, t)

function run_aux!(a::GAPF, u, y, mode; history = false)
    pf = a.pf
    T = length(y)
    U = rows(u, pf.nu)
    Y = rows(y, pf.ny)
    ll = Ref{Float64}(0)
    x = history ? Array{Float64}(undef, pf.nx, pf.N, T) : Array{Float64}(undef, 0, 0, 0)
    w = history ? Array{Float64}(undef, pf.N, T) : Array{Float64}(undef, 0, 0)
    we = history ? Array{Float64}(undef, pf.N, T) : Array{Float64}(undef, 0, 0)
    GC.@preserve U Y x w we begin
        outs = Ref(CRunOutputs(C_NULL, C_NULL, history ? pointer(x) : C_NULL, history ? pointer(w) : C_NULL, history ? pointer(we) : C_NULL, C_NULL))
        check(ccall((:llpf_aux_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Ref{Float64}, Ref{CRunOutputs}),
                    pf.h, pf.nu > 0 ? pointer(U) : C_NULL, pointer(Y), T, Int32(mode), ll, outs))
    end
    ll[], x, w, we
end
"loglik(pf::AuxiliaryParticleFilter,u,y,p) — src/smoothing.jl:232-236"
function loglik(a::GAPF, u, y, p = parameters(a))
    reset!(a)
    run_aux!(a, u, y, 1)[1]
end
"forward_trajectory(pf::AuxiliaryParticleFilter,u,y,p) -> ParticleFilteringSolution — src/filtering.jl:367-384"
function forward_trajectory(a::GAPF{<:GPUParticleFilter{NX}}, u::AbstractVector, y::AbstractVector, p = parameters(a)) where {NX}
    reset!(a)
    ll, x, w, we = run_aux!(a, u, y, 0; history = true)
    ParticleFilteringSolution(a, u, y, svec_history(x, Val(NX)), w, we, ll)
end

# ---- banks of independent filters: one GPU (llpf_bank_*) ----------------------------------------------------------------
"`map(svec) do s; pfs = ParticleFilter(...); loglik(pfs, u, y); end` (reference test/runtests.jl:412-417) as one device job"
mutable struct GPUFilterBank
    h::Ptr{Cvoid}
    F::Int
    N::Int
    nx::Int
    nu::Int
    ny::Int
end
bank_models(dynamics::Vector, measurement, dfs::Vector, dg, d0, Ts) =
    [cmodel(dynamics[k], measurement, dfs[k], dg, d0, Float64(Ts)) for k in eachindex(dynamics)]
function GPUFilterBank(N::Integer, dynamics::Vector, measurement, dfs::Vector, dg, d0; resample_threshold = 0.1,
                       resampling_strategy::Type{<:ResamplingStrategy} = ResampleSystematic, seed = 0, Ts = 1.0, device = 0)
    cms = bank_models(dynamics, measurement, dfs, dg, d0, Ts)
    cfg = Ref(CConfig(UInt32(sizeof(CConfig)), 0, N, strategy_code(resampling_strategy), device, resample_threshold, UInt64(seed), cms[1]))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:llpf_bank_create, LIB), Cint, (Ref{CConfig}, Ptr{CModel}, Int32, Ref{Ptr{Cvoid}}), cfg, cms, length(cms), h))
    b = GPUFilterBank(h[], length(cms), N, cms[1].nx, cms[1].nu, cms[1].ny)
    finalizer(x -> ccall((:llpf_bank_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), b)
    b
end
"""
    set_parameters!(bank, dynamics::Vector, measurement, dfs::Vector, dg, d0; Ts = 1.0)

New parameters for every filter of a `GPUFilterBank` / `GPUMultiBank` (the arguments of its constructor, same count and dimensions) without
reallocating anything: a Metropolis iteration over one chain per filter — the reference's `metropolis_threaded` (src/smoothing.jl:335-347) —
is `set_parameters!` with the chains' candidates followed by `loglik(bank, u, y)`.
"""
function set_parameters!(b::GPUFilterBank, dynamics::Vector, measurement, dfs::Vector, dg, d0; Ts = 1.0)
    cms = bank_models(dynamics, measurement, dfs, dg, d0, Ts)
    length(cms) == b.F || throw(ArgumentError("set_parameters!: $(length(cms)) models for a bank of $(b.F) filters"))
    check(ccall((:llpf_bank_set_models, LIB), Cint, (Ptr{Cvoid}, Ptr{CModel}), b.h, cms))
    b
end
"log-likelihood of every filter of the bank on shared data u, y (vectors of vectors)"
function loglik(b::GPUFilterBank, u, y)
    T = length(y)
    U = rows(u, b.nu)
    Y = rows(y, b.ny)
    ll = zeros(b.F)
    check(ccall((:llpf_bank_reset, LIB), Cint, (Ptr{Cvoid},), b.h))
    GC.@preserve U Y check(ccall((:llpf_bank_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}),
                                 b.h, b.nu > 0 ? pointer(U) : C_NULL, pointer(Y), T, 1.0, ll, C_NULL))
    ll
end
"as loglik, every filter on data of its own: U nu x T x F, Y ny x T x F (column-major = the ABI's [F][T][n]); also the weighted means nx x F x T"
function loglik_multi(b::GPUFilterBank, U::Array{Float64,3}, Y::Array{Float64,3})
    T = size(Y, 2)
    ll = zeros(b.F)
    xm = zeros(b.nx, b.F, T)
    check(ccall((:llpf_bank_reset, LIB), Cint, (Ptr{Cvoid},), b.h))
    check(ccall((:llpf_bank_run_multi, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                b.h, U, Y, T, 1.0, ll, C_NULL, xm))
    ll, xm
end

# ---- the same sweep sharded over the GPUs of a node (llpf_mbank_*): filter k on shard k mod n_shards, one RCCL all-reduce
# of the log-likelihood vector per run, inside the library ----------------------------------------------------------------
mutable struct GPUMultiBank
    h::Ptr{Cvoid}
    F::Int
    N::Int
    nx::Int
    nu::Int
    ny::Int
end
"""
    GPUMultiBank(N, dynamics::Vector, measurement, dfs::Vector, dg, d0; devices = [0, 1, ...], ...)            # this process drives all listed GPUs
    GPUMultiBank(N, dynamics::Vector, measurement, dfs::Vector, dg, d0; rank, world, unique_id, device, ...)   # one process per GPU

The reference's one-filter-per-thread sweeps (src/smoothing.jl:335-347, test/runtests.jl:412-417) with GPUs for threads.
With one process per GPU (Distributed.jl, MPI.jl), rank 0 calls `mbank_unique_id()` and sends the 128 bytes to the others."""
function GPUMultiBank(N::Integer, dynamics::Vector, measurement, dfs::Vector, dg, d0; devices = Int32[0], rank = nothing, world = 1,
                      unique_id = nothing, device = 0, resample_threshold = 0.1,
                      resampling_strategy::Type{<:ResamplingStrategy} = ResampleSystematic, seed = 0, Ts = 1.0)
    cms = bank_models(dynamics, measurement, dfs, dg, d0, Ts)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if rank === nothing
        devs = collect(Int32, devices)
        cfg = Ref(CConfig(UInt32(sizeof(CConfig)), 0, N, strategy_code(resampling_strategy), devs[1], resample_threshold, UInt64(seed), cms[1]))
        check(ccall((:llpf_mbank_create, LIB), Cint, (Ref{CConfig}, Ptr{CModel}, Int32, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                    cfg, cms, length(cms), devs, length(devs), h))
    else
        cfg = Ref(CConfig(UInt32(sizeof(CConfig)), 0, N, strategy_code(resampling_strategy), device, resample_threshold, UInt64(seed), cms[1]))
        id = unique_id === nothing ? UInt8[] : collect(UInt8, unique_id)
        GC.@preserve id check(ccall((:llpf_mbank_create_rank, LIB), Cint, (Ref{CConfig}, Ptr{CModel}, Int32, Int32, Int32, Ptr{UInt8}, Ref{Ptr{Cvoid}}),
                                    cfg, cms, length(cms), rank, world, isempty(id) ? Ptr{UInt8}(C_NULL) : pointer(id), h))
    end
    b = GPUMultiBank(h[], length(cms), N, cms[1].nx, cms[1].nu, cms[1].ny)
    finalizer(x -> ccall((:llpf_mbank_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), b)
    b
end
"ncclGetUniqueId through the C ABI: 128 bytes rank 0 hands to every rank's GPUMultiBank(...; unique_id)"
function mbank_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:llpf_mbank_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    id
end
"set_parameters! for a sweep sharded over GPUs (`llpf_mbank_set_models`): same contract as for a `GPUFilterBank`"
function set_parameters!(b::GPUMultiBank, dynamics::Vector, measurement, dfs::Vector, dg, d0; Ts = 1.0)
    cms = bank_models(dynamics, measurement, dfs, dg, d0, Ts)
    length(cms) == b.F || throw(ArgumentError("set_parameters!: $(length(cms)) models for a sweep of $(b.F) filters"))
    check(ccall((:llpf_mbank_set_models, LIB), Cint, (Ptr{Cvoid}, Ptr{CModel}), b.h, cms))
    b
end
"log-likelihood of every filter of the sweep (all-reduced: the same vector in every process); `sum(ll)` is the global log-likelihood"
function loglik(b::GPUMultiBank, u, y)
    T = length(y)
    U = rows(u, b.nu)
    Y = rows(y, b.ny)
    ll = zeros(b.F)
    tot = Ref{Float64}(0)
    check(ccall((:llpf_mbank_reset, LIB), Cint, (Ptr{Cvoid},), b.h))
    GC.@preserve U Y check(ccall((:llpf_mbank_run, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Ptr{Float64}, Ref{Float64}),
                                 b.h, b.nu > 0 ? pointer(U) : C_NULL, pointer(Y), T, 1.0, ll, tot))
    ll
end
function Base.show(io::IO, b::GPUMultiBank)
    info = Ref(CMBankInfo(0, 0, 0, 0, 0, 0, 0.0, 0.0, 0))
    check(ccall((:llpf_mbank_info, LIB), Cint, (Ptr{Cvoid}, Ref{CMBankInfo}), b.h, info))
    i = info[]
    print(io, "GPUMultiBank($(i.n_filters) filters x $(b.N) particles, $(i.n_shards) shards, $(i.n_local_shards) here, collective = ",
          ("none", "rccl", "host", "external")[i.collective+1], ")")
end

include(joinpath(@__DIR__, "tracing.jl"))      # closures -> device snippet (trace_dynamics, emit_user_model)

end # module
