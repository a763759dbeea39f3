# runtests.jl -- shaped like the reference's own AMD test (test/gpu/amd.jl:38-122 + test/gpu/gpu.jl): every k* primitive on the
# device type, then the solvers of this path, then the workspace API.  Unlike the reference's GPU test the VALUES are checked
# (against the same call on Vector{Float64}), and every solver runs twice: through the specialised method (the library's fused,
# device-resident loop on the workspace's own vectors) and through the generic method of Krylov.jl (one kernel per primitive),
# which must agree.
#
#   julia --project=julia/KrylovHIP -e 'using Pkg; Pkg.test()'        (needs an MI355X and krylov.jl_amd/libkrylov_hip.so)
using Test, Random, LinearAlgebra, SparseArrays
using Krylov, KrylovHIP
import KrylovHIP: HIPVector, HIPMatrix, HIPCsr, CTX, Ctx

include(joinpath(pkgdir(Krylov), "test", "get_div_grad.jl"))        # get_div_grad(n1, n2, n3): the benchmark operator
Random.seed!(666)

CTX[] = Ctx(0)

# the generic method of Krylov.jl on the device types (what `cg!` was before this package specialised it).  `invoke` needs argument
# types that are a SUBTYPE of the generic signature `(CgWorkspace{T,FC,S}, Any, AbstractVector{FC})` -- FC is shared by the workspace
# and b, so the workspace and the vector type must be fully parametrised -- and not of the specialised one (A::HIPCsr).
generic(f!, ws, A, b; kw...) = invoke(f!, Tuple{typeof(ws),Any,AbstractVector{Float64}}, ws, A, b; kw...)

# every entry point of the reference must reach the library's loop: `native(path) do ... end` runs the body and checks that exactly
# the expected number of solves went through khip_*_solve, none through the generic method, and which loop the last one ran
# (2 = device-resident / look-ahead, 1 = host-driven on the fused kernels)
function native(body, path::Integer; solves::Integer = 1)
  n0 = KrylovHIP.NATIVE_SOLVES[]; g0 = KrylovHIP.GENERIC_SOLVES[]
  out = body()
  @test KrylovHIP.NATIVE_SOLVES[] == n0 + solves
  @test KrylovHIP.GENERIC_SOLVES[] == g0
  @test KrylovHIP.LAST_PATH[] == path
  out
end

@testset "AMD -- KrylovHIP (libkrylov_hip)" begin

  @testset "documentation" begin
    A_cpu = get_div_grad(16, 16, 16)                                 # sparse_laplacian(16), test/test_cg.jl:22-28
    b_cpu = ones(size(A_cpu, 1))
    A_gpu = HIPCsr(A_cpu)
    b_gpu = HIPVector(b_cpu)
    x, stats = native(2) do; cg(A_gpu, b_gpu); end                   # the package's first call: cg(A, b) forwards callback = workspace -> false
    r = b_cpu - A_cpu * Vector(x)
    @test norm(r) / norm(b_cpu) ≤ 1e-6
    @test stats.solved
    @test stats.niter == 38                                          # the oracle's count for the package defaults (SURVEY 8c)
  end

  FC = Float64                                                       # the element type of this path (SURVEY 8: T = FC = Float64)
  S = HIPVector
  M = HIPMatrix
  n = 1000
  xh = rand(FC, n); yh = rand(FC, n)
  a = rand(FC); b = rand(FC); c = rand(FC); s = sqrt(1 - c^2)
  dev(v) = S(copy(v))

  @testset "kdot -- $FC" begin
    @test Krylov.kdot(n, dev(xh), dev(yh)) ≈ dot(xh, yh) rtol = 4eps()
    x = dev(xh); @test Krylov.kdot(n, x, x) ≈ dot(xh, xh) rtol = 4eps()          # aliased arguments (src/cg.jl:242)
  end
  @testset "kdotr -- $FC" begin
    @test Krylov.kdotr(n, dev(xh), dev(yh)) ≈ dot(xh, yh) rtol = 4eps()
  end
  @testset "knorm -- $FC" begin
    @test Krylov.knorm(n, dev(xh)) ≈ norm(xh) rtol = 4eps()
  end
  @testset "kaxpy! -- $FC" begin
    y = dev(yh); @test Krylov.kaxpy!(n, a, dev(xh), y) === y
    @test Vector(y) == Krylov.kaxpy!(n, a, copy(xh), copy(yh))
  end
  @testset "kaxpby! -- $FC" begin
    y = dev(yh); Krylov.kaxpby!(n, a, dev(xh), b, y)
    @test Vector(y) ≈ a .* xh .+ b .* yh rtol = 4eps()
  end
  @testset "kcopy! -- $FC" begin
    y = dev(yh); Krylov.kcopy!(n, y, dev(xh))                        # (dest, src)
    @test Vector(y) == xh
  end
  @testset "kfill! kscal! kdiv! kscalcopy! kdivcopy! -- $FC" begin
    x = dev(xh); Krylov.kfill!(x, a); @test all(==(a), Vector(x))
    x = dev(xh); Krylov.kscal!(n, a, x); @test Vector(x) == a .* xh
    x = dev(xh); Krylov.kdiv!(n, x, a); @test Vector(x) == xh .* (1 / a)       # kdiv! multiplies by the reciprocal, src/krylov_utils.jl:325-326
    y = dev(yh); Krylov.kscalcopy!(n, y, a, dev(xh)); @test Vector(y) == a .* xh
    y = dev(yh); Krylov.kdivcopy!(n, y, dev(xh), a); @test Vector(y) == xh ./ a
  end
  @testset "kswap! -- $FC" begin
    x = dev(xh); y = dev(yh)
    Krylov.@kswap!(x, y)
    @test Vector(x) == yh && Vector(y) == xh
  end
  @testset "kref! -- $FC" begin
    x = dev(xh); y = dev(yh); Krylov.kref!(n, x, y, c, s)
    xr = copy(xh); yr = copy(yh); Krylov.kref!(n, xr, yr, c, s)
    @test Vector(x) ≈ xr rtol = 4eps()
    @test Vector(y) ≈ yr rtol = 4eps()
  end
  @testset "kmul! -- $FC" begin
    A = sprand(n, n, 0.01) + 4I
    Ad = HIPCsr(A); y = S(undef, n)
    Krylov.kmul!(y, Ad, dev(xh))
    @test Vector(y) ≈ A * xh rtol = 1e-14
    Krylov.kmul!(y, Ad', dev(xh))                                    # adjoint products (docs/src/matrix_free.md:36-42)
    @test Vector(y) ≈ A' * xh rtol = 1e-14
    @test Ad' === Ad' && (Ad')' === Ad                               # A' is built once per matrix and owned by its own finalizer
  end
  @testset "conversion -- $FC" begin
    @test Krylov.matrix_to_vector(M) <: Vector{Float64}              # the block solver's tau / buffer vectors live on the host
    @test Krylov.ktypeof(dev(xh)) === S
    @test Krylov.ktypeof(M(rand(64, 2))) === M
  end

  ε = eps(FC); atol = √ε; rtol = √ε
  A_cpu = get_div_grad(12, 12, 12); nA = size(A_cpu, 1)
  U_cpu = A_cpu + spdiagm(1 => fill(-0.5, nA - 1))                   # nonsymmetric, diagonally dominant
  b_cpu = A_cpu * collect(1.0:nA); bu_cpu = U_cpu * collect(1.0:nA)
  A_gpu = HIPCsr(A_cpu); U_gpu = HIPCsr(U_cpu)

  @testset "CG -- $FC" begin
    b = S(b_cpu)
    x, stats = native(2) do; cg(A_gpu, b); end                       # out-of-place entry -> cg! on a fresh workspace -> the device-resident loop
    @test norm(b_cpu - A_cpu * Vector(x)) ≤ atol + rtol * norm(b_cpu)
    x, stats = native(2) do; krylov_solve(Val(:cg), A_gpu, b); end   # src/interface.jl:156
    ws = CgWorkspace(nA, nA, S)
    native(2) do; cg!(ws, A_gpu, b; history = true); end
    @test Krylov.solution(ws) === ws.x                               # test/test_interface.jl:260: the solution IS the workspace's vector
    ws2 = CgWorkspace(nA, nA, S)
    generic(cg!, ws2, A_gpu, b; history = true)
    @test ws.stats.niter == ws2.stats.niter && ws.stats.status == ws2.stats.status
    @test ws.stats.residuals ≈ ws2.stats.residuals rtol = 1e-10      # fused vs unfused reductions: <= 1 ulp per dot
    @test Vector(ws.x) ≈ Vector(ws2.x) rtol = 1e-10
    @test ws.stats.solved && ws.stats.timer > 0
    native(2) do; krylov_solve!(ws, A_gpu, b); end                   # src/interface.jl:331: forwards every default, callback included
    # warm start (all three x0 entry points, src/interface.jl:160-170, 333-347), Jacobi (native operator), zero right-hand side
    x0 = S(collect(1.0:nA) .+ 0.01)
    warm_start!(ws, x0); native(2) do; cg!(ws, A_gpu, b); end
    @test ws.stats.niter < ws2.stats.niter && !ws.warm_start
    native(2) do; cg!(ws, A_gpu, b, x0); end;            @test ws.stats.niter < ws2.stats.niter
    native(2) do; krylov_solve!(ws, A_gpu, b, x0); end;  @test ws.stats.niter < ws2.stats.niter
    x, stats = native(2) do; cg(A_gpu, b, x0); end;      @test stats.niter < ws2.stats.niter
    native(1) do; cg!(ws, A_gpu, b; M = KrylovHIP.jacobi(A_gpu)); end; @test ws.stats.solved && !isempty(ws.z)
    # a real callback runs inside the library's host-driven loop (trampoline): it sees the workspace and the history so far
    seen = Int[]
    native(1) do; cg!(ws, A_gpu, b; history = true, callback = w -> (push!(seen, length(w.stats.residuals)); length(seen) ≥ 5)); end
    @test ws.stats.status == "user-requested exit" && ws.stats.niter == 5 && seen == [2, 3, 4, 5, 6]
    @test_throws DomainError cg!(ws, A_gpu, b; callback = w -> throw(DomainError(0)))       # kept by the trampoline, rethrown after the solve
    native(2) do; cg!(ws, A_gpu, Krylov.kfill!(S(undef, nA), 0.0)); end
    @test ws.stats.niter == 0 && ws.stats.status == "x is a zero-residual solution"
    @test_throws ErrorException cg!(CgWorkspace(nA + 1, nA + 1, S), A_gpu, b)
    # what the library cannot take goes to the generic method, and that call works (VERDICT r05: the invoke signature)
    g0 = KrylovHIP.GENERIC_SOLVES[]
    cg!(ws, A_gpu, b; iostream = IOBuffer(), verbose = 1); @test ws.stats.solved && KrylovHIP.GENERIC_SOLVES[] == g0 + 1
  end

  @testset "IC(0)-CG -- $FC" begin                                   # the reference's only GPU known answer: niter <= 19 (test/gpu/nvidia.jl:57,69)
    A16 = get_div_grad(16, 16, 16); A16d = HIPCsr(A16); b = S(ones(size(A16, 1)))
    x, stats = cg(A16d, b; M = KrylovHIP.ilu0(A16d))
    @test stats.niter ≤ 19 && stats.solved
  end

  @testset "GMRES -- $FC" begin
    b = S(bu_cpu)
    x, stats = native(2) do; gmres(U_gpu, b); end
    @test norm(bu_cpu - U_cpu * Vector(x)) ≤ atol + rtol * norm(bu_cpu)
    x, stats = native(2) do; gmres(U_gpu, b; memory = 10, restart = true); end    # workspace keyword + solver keyword, src/interface.jl:177-189
    @test stats.solved
    for restart in (false, true)
      ws = GmresWorkspace(nA, nA, S; memory = 10); ws2 = GmresWorkspace(nA, nA, S; memory = 10)
      native(2) do; gmres!(ws, U_gpu, b; restart, history = true); end
      generic(gmres!, ws2, U_gpu, b; restart, history = true)
      @test ws.stats.niter == ws2.stats.niter && ws.stats.niter > 10
      @test ws.stats.residuals ≈ ws2.stats.residuals rtol = 1e-8
      @test restart ? length(ws.V) == 10 : length(ws.V) == length(ws2.V) > 10      # push!(V, similar(x)) through the grow callback
      @test length(ws.c) == length(ws2.c) && ws.inner_iter == ws2.inner_iter
      @test Krylov.solution(ws) === ws.x
    end
  end

  @testset "BiCGSTAB -- $FC" begin
    b = S(bu_cpu)
    ws = BicgstabWorkspace(nA, nA, S); ws2 = BicgstabWorkspace(nA, nA, S)
    native(2) do; bicgstab!(ws, U_gpu, b; history = true); end
    generic(bicgstab!, ws2, U_gpu, b; history = true)
    x, stats = native(2) do; bicgstab(U_gpu, b); end;  @test stats.niter == ws.stats.niter
    @test norm(bu_cpu - U_cpu * Vector(ws.x)) ≤ atol + rtol * norm(bu_cpu)
    @test ws.stats.niter == ws2.stats.niter
    @test ws.stats.residuals ≈ ws2.stats.residuals rtol = 1e-6
  end

  @testset "block-GMRES -- $FC" begin
    p = 4
    B_cpu = hcat((U_cpu * (collect(1.0:nA) .^ (j / 4)) for j in 1:p)...)
    B = M(B_cpu)
    X, stats = native(1) do; block_gmres(U_gpu, B); end              # the block solver has one loop (khip_block_gmres_last_path = 1)
    @test norm(B_cpu - U_cpu * Matrix(X)) ≤ atol + rtol * norm(B_cpu)
    ws = BlockGmresWorkspace(nA, nA, p, Vector{Float64}, M; memory = 5)
    ws2 = BlockGmresWorkspace(nA, nA, p, Vector{Float64}, M; memory = 5)
    native(1) do; block_gmres!(ws, U_gpu, B; history = true); end
    invoke(block_gmres!, Tuple{typeof(ws2),Any,AbstractMatrix{Float64}}, ws2, U_gpu, B; history = true)
    @test ws.stats.niter == ws2.stats.niter
    @test ws.stats.residuals ≈ ws2.stats.residuals rtol = 1e-8
    @test Matrix(ws.X) ≈ Matrix(ws2.X) rtol = 1e-8
  end

  @testset "other solvers through the k* methods -- $FC" begin       # the remaining solvers run unmodified on the device type
    b = S(b_cpu)
    for solver in (minres, cr, symmlq, cg_lanczos)
      x, stats = solver(A_gpu, b)
      @test norm(b_cpu - A_cpu * Vector(x)) ≤ 1e-5 * norm(b_cpu)
    end
    x, stats = bilq(U_gpu, S(bu_cpu))                                # needs A' (khip_csr_transpose)
    @test norm(bu_cpu - U_cpu * Vector(x)) ≤ 1e-5 * norm(bu_cpu)
    At = U_gpu'; bilq(U_gpu, S(bu_cpu)); @test U_gpu' === At          # the second solve reuses the transposed operator
  end

  @testset "solver -- $FC" begin                                     # test/gpu/gpu.jl test_solver
    memory = 5
    workspace = GmresWorkspace(nA, nA, S; memory)
    native(2) do; krylov_solve!(workspace, U_gpu, S(bu_cpu)); end    # the generic in-place entry reaches the specialised method AND its device loop
    @test workspace.stats.solved
  end

  @testset "ktypeof -- $FC" begin
    dv = S(rand(FC, 10))
    @test Krylov.ktypeof(dv) <: S
  end
end
