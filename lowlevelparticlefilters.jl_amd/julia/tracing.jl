# tracing.jl — callables without device source (round 6); the twin of lowlevelparticlefilters.jl_amd/tracing.py, included by LLPFAmd.jl.
#
# The reference takes its model as closures — dynamics(x, u, p, t), measurement(x, u, p, t), measurement_likelihood(x, u, y, p, t)
# (src/PFtypes.jl:59-63, 189-193).  A closure cannot run on the GPU; for straight-line models it does not have to be rewritten as HIP
# source either: the closure is called ONCE on `Tr <: Real` numbers that record the expression DAG (+ - * / sqrt exp log log1p abs max min
# ifelse, comparisons, integer powers through Base.power_by_squaring; every constant with all its bits), and the DAG is emitted as the
# `struct UserModel` snippet that llpf_model_compile builds with hiprtc.  One IEEE operation per node, in the closure's own order (the
# engine is compiled with -ffp-contract=off): `trace_dynamics(rk4(quadtank, Ts; supersample = 2), 4, 2; ...)` of
# examples/example_quadtank.jl:8-35 reproduces the built-in QuadTankDynamics bit for bit (the Python twin's test does exactly this on
# the MI355X: tests/test_tracing.py).  Data-dependent control flow (`if x[1] > 0`) cannot be traced: write `ifelse(x[1] > 0, a, b)`.
# NOT EXECUTED in this image (no Julia binary): statically checked like the rest of the wrapper (tests/test_julia_struct_mirror.py).

mutable struct TraceGraph
    nodes::Vector{Tuple}                       # (op::Symbol, args...) ; args: node ids, or the UInt64 bits of a constant
    index::Dict{Tuple,Int}
end
TraceGraph() = TraceGraph(Tuple[], Dict{Tuple,Int}())
function addnode!(g::TraceGraph, key::Tuple)
    i = get(g.index, key, 0)
    i != 0 && return i
    push!(g.nodes, key)
    g.index[key] = length(g.nodes)
    length(g.nodes)
end

"a traced Float64: a node of the expression DAG"
struct Tr <: Real
    g::TraceGraph
    i::Int
end
"a traced comparison: usable in ifelse and with & | ! only"
struct TrCond
    g::TraceGraph
    i::Int
end
trconst(g::TraceGraph, v::Real) = Tr(g, addnode!(g, (:const, reinterpret(UInt64, Float64(v)))))
Base.promote_rule(::Type{Tr}, ::Type{<:Real}) = Tr
Base.convert(::Type{Tr}, x::Tr) = x
Base.convert(::Type{Tr}, x::Real) = error("a constant met a traced value without a trace to belong to (internal)")
Base.Float64(::Tr) = error("a traced value was converted to Float64: the function leaves the traced operations (a library call?)")
Base.Bool(::TrCond) = error("a comparison of traced values was used as a truth value (if / && / ||): write ifelse(cond, a, b)")
lift(r::Tr, v::Tr) = v
lift(r::Tr, v::Real) = trconst(r.g, v)
for (op, sym) in ((:+, :add), (:-, :sub), (:*, :mul), (:/, :div))
    @eval Base.$op(a::Tr, b::Tr) = Tr(a.g, addnode!(a.g, ($(QuoteNode(sym)), a.i, b.i)))
    @eval Base.$op(a::Tr, b::Real) = $op(a, lift(a, b))
    @eval Base.$op(a::Real, b::Tr) = $op(lift(b, a), b)
end
Base.:-(a::Tr) = Tr(a.g, addnode!(a.g, (:neg, a.i)))
Base.:+(a::Tr) = a
for (fn, sym) in ((:sqrt, :sqrt), (:exp, :exp), (:log, :log), (:log1p, :log1p), (:abs, :abs))
    @eval Base.$fn(a::Tr) = Tr(a.g, addnode!(a.g, ($(QuoteNode(sym)), a.i)))
end
Base.literal_pow(::typeof(^), a::Tr, ::Val{2}) = a * a
Base.literal_pow(::typeof(^), a::Tr, ::Val{3}) = (a * a) * a
Base.:^(a::Tr, n::Integer) = Base.power_by_squaring(a, n)
Base.one(a::Tr) = trconst(a.g, 1.0)
Base.zero(a::Tr) = trconst(a.g, 0.0)
for (op, sym) in ((:<, :lt), (:<=, :le), (:>, :gt), (:>=, :ge), (:(==), :eq))
    @eval Base.$op(a::Tr, b::Tr) = TrCond(a.g, addnode!(a.g, ($(QuoteNode(sym)), a.i, b.i)))
    @eval Base.$op(a::Tr, b::Real) = $op(a, lift(a, b))
    @eval Base.$op(a::Real, b::Tr) = $op(lift(b, a), b)
end
Base.:&(a::TrCond, b::TrCond) = TrCond(a.g, addnode!(a.g, (:and, a.i, b.i)))
Base.:|(a::TrCond, b::TrCond) = TrCond(a.g, addnode!(a.g, (:or, a.i, b.i)))
Base.:!(a::TrCond) = TrCond(a.g, addnode!(a.g, (:not, a.i)))
Base.ifelse(c::TrCond, a, b) = (r = Tr(c.g, 1); aa = lift(r, a); bb = lift(r, b); Tr(c.g, addnode!(c.g, (:sel, c.i, aa.i, bb.i))))
# max / min as the engine's built-in models write them (v > 0 ? v : 0): no NaN / signed-zero special cases
Base.max(a::Tr, b::Tr) = ifelse(a > b, a, b)
Base.max(a::Tr, b::Real) = max(a, lift(a, b))
Base.max(a::Real, b::Tr) = max(lift(b, a), b)
Base.min(a::Tr, b::Tr) = ifelse(a < b, a, b)
Base.min(a::Tr, b::Real) = min(a, lift(a, b))
Base.min(a::Real, b::Tr) = min(lift(b, a), b)

const TRACE_FMT = Dict(:add => "(\$1 + \$2)", :sub => "(\$1 - \$2)", :mul => "(\$1 * \$2)", :div => "(\$1 / \$2)", :neg => "(-\$1)", :abs => "llpf_fabs(\$1)",
                       :sqrt => "llpf_sqrt(\$1)", :exp => "llpf_exp(\$1)", :log => "llpf_log(\$1)", :log1p => "llpf_log1p_nonneg(\$1)",
                       :lt => "(\$1 < \$2)", :le => "(\$1 <= \$2)", :gt => "(\$1 > \$2)", :ge => "(\$1 >= \$2)", :eq => "(\$1 == \$2)",
                       :and => "(\$1 && \$2)", :or => "(\$1 || \$2)", :not => "(!\$1)", :sel => "(\$1 ? \$2 : \$3)")
const TRACE_BOOL = (:lt, :le, :gt, :ge, :eq, :and, :or, :not)
const TRACE_INPUT = (:x, :u, :t, :y, :const)

"statements computing the nodes reachable from `outs`, in creation order (= the closure's own order), then `stmt(k, name)` per output"
function emit_body(g::TraceGraph, outs::Vector{Int}, stmt, used::Set{Symbol}; tname = "t_")
    need = Set{Int}()
    stack = copy(outs)
    while !isempty(stack)
        i = pop!(stack)
        i in need && continue
        push!(need, i)
        node = g.nodes[i]
        node[1] in TRACE_INPUT && continue
        append!(stack, Int[a for a in node[2:end]])
    end
    lines = String[]
    name = Dict{Int,String}()
    for i in sort!(collect(need))
        node = g.nodes[i]
        op = node[1]
        if op === :x
            name[i] = "x[$(node[2])]"
        elseif op === :u
            name[i] = "u_[$(node[2])]"; push!(used, :u)
        elseif op === :t
            name[i] = tname; push!(used, :t)
        elseif op === :y
            name[i] = "y[$(node[2])]"; push!(used, :y)
        elseif op === :const
            name[i] = "c$i"
            push!(lines, "        const double c$i = llpf_u2d(0x$(string(node[2], base = 16, pad = 16))ULL);")
        else
            ex = TRACE_FMT[op]
            for (k, a) in enumerate(node[2:end])
                ex = replace(ex, "\$$k" => name[a])
            end
            name[i] = "v$i"
            push!(lines, "        const $(op in TRACE_BOOL ? "bool" : "double") v$i = $ex;")
        end
    end
    for (k, i) in enumerate(outs)
        push!(lines, "        " * stmt(k - 1, name[i]))
    end
    join(lines, "\n")
end

function trace_outputs(g::TraceGraph, out, n::Int, what::String)
    v = out isa Union{Real,Tr} ? [out] : collect(out)
    length(v) == n || error("$what returned $(length(v)) values, expected $n")
    r = Tr(g, 1)
    Int[lift(r, e).i for e in v]
end
trace_inputs(g::TraceGraph, sym::Symbol, n::Int) = Tr[Tr(g, addnode!(g, (sym, k - 1))) for k in 1:n]

"""
    emit_user_model(dynamics, nx, nu, ny; p = nothing, measurement = nothing, measurement_likelihood = nothing, loglik_bound = nothing) -> String

The `struct UserModel` device snippet of closures with the reference's signatures (traced once on `Tr` numbers).  The traced
measurement may depend on the state only and the likelihood not on `u` (the engine's Model concept hands neither to them)."""
function emit_user_model(dynamics, nx::Int, nu::Int, ny::Int; p = nothing, measurement = nothing, measurement_likelihood = nothing, loglik_bound = nothing)
    used = Set{Symbol}()
    g = TraceGraph()
    x = trace_inputs(g, :x, nx); u = trace_inputs(g, :u, nu); t = Tr(g, addnode!(g, (:t,)))
    od = trace_outputs(g, dynamics(x, u, p, t), nx, "dynamics")
    parts = String["    DEV void dynamics(const double* x, double* out) const {\n" * emit_body(g, od, (k, n) -> "out[$k] = $n;", used) * "\n    }"]
    if measurement === nothing
        push!(parts, "    DEV void measurement(const double* x, double* out) const { for (int k = 0; k < $ny; ++k) out[k] = x[k]; }")
    else
        gm = TraceGraph(); um = Set{Symbol}()
        xm = trace_inputs(gm, :x, nx); uu = trace_inputs(gm, :u, nu); tm = Tr(gm, addnode!(gm, (:t,)))
        om = trace_outputs(gm, measurement(xm, uu, p, tm), ny, "measurement")
        body = emit_body(gm, om, (k, n) -> "out[$k] = $n;", um)
        (:u in um || :t in um) && error("a traced measurement may depend on the state only: put the dependence into measurement_likelihood")
        push!(parts, "    DEV void measurement(const double* x, double* out) const {\n" * body * "\n    }")
    end
    if measurement_likelihood !== nothing
        gl = TraceGraph(); ul = Set{Symbol}()
        xl = trace_inputs(gl, :x, nx); uu = trace_inputs(gl, :u, nu); yl = trace_inputs(gl, :y, ny); tl = Tr(gl, addnode!(gl, (:t,)))
        ol = trace_outputs(gl, measurement_likelihood(xl, uu, yl, p, tl), 1, "measurement_likelihood")
        body = emit_body(gl, ol, (k, n) -> "return $n;", ul; tname = "t")
        :u in ul && error("a traced measurement_likelihood may not depend on u")
        push!(parts, "    DEV double loglik(const double* x, const double* y, double t) const {\n" * body * "\n    }")
        loglik_bound === nothing || push!(parts, "    DEV double loglik_bound() const { return llpf_u2d(0x$(string(reinterpret(UInt64, Float64(loglik_bound)), base = 16, pad = 16))ULL); }")
    end
    "struct UserModel {\n    static constexpr bool RB = false;\n    double u_[$(max(nu, 1))];\n    double t_;\n" *
    "    DEV void prepare(const ModelD* m, const double* u, double t) {\n        for (int j = 0; j < $nu; ++j) u_[j] = (u != nullptr) ? u[j] : 0.0;\n        t_ = t;\n    }\n" *
    join(parts, "\n") * "\n};\n"
end

"""
    trace_dynamics(dynamics, nx, nu; ny, p = nothing, measurement = nothing, measurement_likelihood = nothing, loglik_bound = nothing)
        -> (UserDynamics, UserMeasurement, UserLikelihood or nothing)

The descriptors that stand for the closures in `GPUParticleFilter` / `GPUAdvancedParticleFilter` (the closures stay their host
versions, for `simulate`)."""
function trace_dynamics(dynamics, nx::Int, nu::Int; ny::Int, p = nothing, measurement = nothing, measurement_likelihood = nothing, loglik_bound = nothing)
    src = emit_user_model(dynamics, nx, nu, ny; p = p, measurement = measurement, measurement_likelihood = measurement_likelihood, loglik_bound = loglik_bound)
    dyn = UserDynamics(src, nx, nu, ny; host = (x, u, pp, t) -> dynamics(x, u, p, t))
    meas = UserMeasurement(measurement === nothing ? nothing : (x, u, pp, t) -> measurement(x, u, p, t))
    lik = measurement_likelihood === nothing ? nothing : UserLikelihood((x, u, y, pp, t) -> measurement_likelihood(x, u, y, p, t))
    dyn, meas, lik
end
