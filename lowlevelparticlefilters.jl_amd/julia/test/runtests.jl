# test/runtests.jl of LLPFAmd — the reference's own end-to-end assertions, restated on the GPU filter.
#
#     julia --project=lowlevelparticlefilters.jl_amd/julia -e 'using Pkg; Pkg.test()'         (needs an MI355X and the built libllpf_hip.so)
#
# Every testset cites the lines of LowLevelParticleFilters.jl's test/runtests.jl it restates (v3.31.1); the bounds are the reference's.
# Nothing here has run in the build image (no Julia there): tests/test_julia_struct_mirror.py checks this file's block structure and
# that every name it calls is exported by LLPFAmd.jl or by the reference; the first machine with Julia and a GPU executes it.
using Test, LinearAlgebra, Random, Statistics, StaticArrays
using LowLevelParticleFilters, LLPFAmd
using LowLevelParticleFilters: SimpleMvNormal, ResampleSystematic, ResampleStratified, ResampleResidual, smoothed_mean
import Distributions: MvNormal

eye(n) = Matrix{Float64}(I, n, n)

@testset "LLPFAmd" begin

    # the system of test/runtests.jl:255-266
    n, m, p = 2, 1, 1
    dg = MvNormal(zeros(p), 1.0 * eye(p))
    df = MvNormal(zeros(n), 0.1^2 * eye(n))
    d0 = MvNormal([0.3, -0.5], 2.0^2 * eye(n))
    A_test = SA[0.97043 -0.097368; 0.09736 0.970437]
    B_test = SA[0.1; 0;;]
    C_test = SA[0 1.0]
    dyn, meas = LinearDynamics(A_test, B_test), LinearMeasurement(C_test)
    N, T, M = 1000, 200, 100

    @testset "a fresh filter does not resample (runtests.jl:274-275)" begin
        pf = GPUParticleFilter(N, dyn, meas, df, dg, d0)
        pfa = GPUAuxiliaryParticleFilter(N, dyn, meas, df, dg, d0)
        @test !shouldresample(pf)
        @test !shouldresample(pfa)
        @test num_particles(pf) == N
        @test effective_particles(pf) ≈ N
        @test all(w ≈ log(1 / N) for w in weights(pf))
        @test sum(expweights(pf)) ≈ 1
    end

    # data from the reference's own simulate (host code, generic over AbstractParticleFilter: it needs the callables and densities only)
    pf = GPUParticleFilter(N, dyn, meas, df, dg, d0; seed = 1)
    du = MvNormal(zeros(m), 1.0 * eye(m))
    x, u, y = LowLevelParticleFilters.simulate(pf, T, du)

    @testset "forward_trajectory and the weighted statistics (runtests.jl:283-308)" begin
        sol = forward_trajectory(pf, u, y)
        @test sol isa LowLevelParticleFilters.ParticleFilteringSolution
        @test size(sol.x) == (N, T) && size(sol.w) == (N, T) && size(sol.we) == (N, T)
        @test all(isfinite, sol.ll)
        WM = weighted_mean(sol.x, sol.we)
        @test WM == weighted_mean(sol)
        @test length(WM) == T
        @test WM[1] ≈ weighted_mean(sol.x[:, 1], sol.we[:, 1])
        WQ1 = weighted_quantile(sol, 0.1)
        WQ9 = weighted_quantile(sol, 0.9)
        @test all(all(WM[i] .< WQ9[i]) for i in eachindex(WM))
        @test all(all(WM[i] .> WQ1[i]) for i in eachindex(WM))
        C = weighted_cov(sol)
        C2 = zero(C[2])
        for (i, w) in enumerate(sol.we[:, 2])
            d = sol.x[i, 2] .- WM[2]
            C2 .+= w * d * d'
        end
        @test C2 * N / (N - 1) ≈ C[2]
        # the device accessors agree with the host functions on the filter's final state
        @test weighted_mean(pf) ≈ weighted_mean(reshape(particles(pf), :, 1), reshape(expweights(pf), :, 1))[1]
        @test weighted_cov(pf) ≈ weighted_cov(reshape(particles(pf), :, 1), reshape(expweights(pf), :, 1))[1]
        q = weighted_quantile(pf, [0.1, 0.5, 0.9])
        @test size(q) == (n, 3) && all(q[:, 1] .<= q[:, 2] .<= q[:, 3])
    end

    @testset "the GPU filter tracks like the reference's (same data, independent noise)" begin
        pfc = ParticleFilter(N, dyn, meas, df, dg, d0)                 # the reference on the CPU: the descriptors are ordinary callables
        llc = loglik(pfc, u, y)
        llg = loglik(pf, u, y)
        @test abs(llc - llg) < 20                                      # the bound runtests.jl:447 puts between a PF and the Kalman filter
        xm = reduce(hcat, x)
        xg = reduce(hcat, mean_trajectory(pf, u, y)[1])
        xc = reduce(hcat, mean_trajectory(pfc, u, y)[1])
        @test mean(abs2, xm - xg) < 2 * mean(abs2, xm - xc) + 0.1
    end

    @testset "step verbs: update! = correct! then predict!, functor (src/filtering.jl:181-185, 238-240)" begin
        a = GPUParticleFilter(N, dyn, meas, df, dg, d0; seed = 7)
        b = GPUParticleFilter(N, dyn, meas, df, dg, d0; seed = 7)
        reset!(a); reset!(b)
        for t in 1:20
            ll1, _ = a(u[t], y[t])
            ll2, _ = correct!(b, u[t], y[t])
            predict!(b, u[t])
            @test ll1 == ll2
        end
        @test particles(a) == particles(b)
        @test index(a) == index(b) == 21
        @test ancestors(a) == ancestors(b)
        # a missing measurement leaves the weights alone (src/PFtypes.jl:109)
        w0 = copy(weights(a))
        ll, _ = correct!(a, u[21], fill(missing, p))
        @test weights(a) ≈ w0
    end

    @testset "smoothing (runtests.jl:314-321)" begin
        xm = reduce(hcat, x)
        xb, ll = smooth(pf, M, u, y)
        @test size(xb) == (M, T)
        xbm = smoothed_mean(xb)
        @test mean(abs2, xm - xbm) < 5
    end

    @testset "maximum-likelihood sweep (runtests.jl:409-450)" begin
        kf = KalmanFilter(A_test, B_test, C_test, 0, 0.01eye(n), eye(p), d0)
        xk, uk, yk = LowLevelParticleFilters.simulate(kf, 2000, du)
        svec = exp10.(LinRange(-2, 0, 11))
        llspf = map(svec) do s
            pfs = GPUParticleFilter(N, dyn, meas, MvNormal(zeros(n), s^2 * eye(n)), dg, d0)
            loglik(pfs, uk, yk)
        end
        llspfa = map(svec) do s
            pfs = GPUAuxiliaryParticleFilter(N, dyn, meas, MvNormal(zeros(n), s^2 * eye(n)), dg, d0)
            loglik(pfs, uk, yk)
        end
        llskf = map(svec) do s
            loglik(KalmanFilter(A_test, B_test, C_test, 0, s^2 * eye(n), eye(p), d0), uk, yk)
        end
        @test 5 ≤ findmax(llspf)[2] ≤ 7
        @test 5 ≤ findmax(llspfa)[2] ≤ 7
        @test 5 ≤ findmax(llskf)[2] ≤ 7
        @test maximum(abs, llskf .- llspf) < 20
        @test maximum(abs, llskf .- llspfa) < 20
        # the same sweep as ONE bank: filter k is the single filter with seed + k (bit for bit)
        dfs = [MvNormal(zeros(n), s^2 * eye(n)) for s in svec]
        bank = GPUFilterBank(N, fill(dyn, length(svec)), meas, dfs, dg, d0; seed = 0)
        llb = loglik(bank, uk, yk)
        @test 5 ≤ findmax(llb)[2] ≤ 7
        single = GPUParticleFilter(N, dyn, meas, dfs[1], dg, d0; seed = 0)
        @test loglik(single, uk, yk) == llb[1]
        # new parameters for the same handles: what filter_from_parameters(θ, pf) hands the old filter back for (src/smoothing.jl:266-283)
        set_parameters!(single; dynamics_density = dfs[6])
        seed!(single, 5)
        fresh = GPUParticleFilter(N, dyn, meas, dfs[6], dg, d0; seed = 5)
        @test loglik(single, uk, yk) == loglik(fresh, uk, yk)
    end

    @testset "resampling strategies (runtests.jl:90-154 through the filter)" begin
        for rs in (ResampleSystematic, ResampleStratified, ResampleResidual)
            f = GPUParticleFilter(N, dyn, meas, df, dg, d0; resampling_strategy = rs, resample_threshold = 1.0, seed = 3)
            ll = loglik(f, u, y)
            @test isfinite(ll)
            j = ancestors(f)
            @test all(1 .<= j .<= N) && issorted(j)
            @test last_resampled(f)
        end
    end

    @testset "quad-tank AdvancedParticleFilter (examples/example_quadtank.jl:8-44, runtests.jl:553-599)" begin
        qdyn, qmeas = QuadTankDynamics(; Ts = 1.0, supersample = 2), QuadTankMeasurement()
        dfq = MvNormal(zeros(4), 0.1 * eye(4)); dgq = MvNormal(zeros(2), 0.01^2 * eye(2)); d0q = MvNormal([2.0, 2, 3, 3], 0.1 * eye(4))
        apf = GPUAdvancedParticleFilter(2000, qdyn, qmeas, GaussianLikelihood(qmeas, dgq), dfq, d0q; seed = 2)
        uq = [SA[0.25 * sign(sin(2pi * t / 200)) + 0.25, 0.25 * sign(sin(2pi * t / 200)) + 0.25] for t in 1:300]
        xq = [SVector{4}(2.0, 2, 3, 3)]
        for t in 1:299
            push!(xq, qdyn(xq[end], uq[t], nothing, t))
        end
        yq = [SA[xq[t][1], xq[t][2]] + 0.01 * randn(SVector{2}) for t in 1:300]
        xh, llq = mean_trajectory(apf, uq, yq)
        @test isfinite(llq)
        @test mean(abs2, reduce(hcat, xq) - reduce(hcat, xh)) < 5       # runtests.jl:587
    end
end
