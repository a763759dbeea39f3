# KrylovHIP.jl -- Krylov.jl on libkrylov_hip.so (hand-written gfx950 kernels behind a C ABI, include/krylov_hip.h).
#
# Two layers, both through `ccall`:
#
#   1. The extension contract of Krylov.jl (docs/src/custom_workspaces.md:107-300): a device vector type `HIPVector`, a device
#      matrix type `HIPMatrix` (block solvers) and a CSR operator `HIPCsr` with methods for every `Krylov.k*` primitive and
#      `kmul!`.  With these the UNMODIFIED solvers of Krylov.jl run on the GPU, one kernel launch per primitive
#      (all 35+ solvers; 249 it/s for cg! at 512^3).
#
#   2. Method specialisations of `cg!`, `gmres!`, `bicgstab!` and `block_gmres!` for workspaces whose storage type is
#      `HIPVector` / `HIPMatrix` and an operator that is a `HIPCsr`.  They hand the device pointers of the WORKSPACE'S OWN
#      vectors to the library (`khip_*_workspace_adopt`, include/krylov_hip.h) and run its fused, device-resident loop on them
#      (`khip_*_solve`; 310-316 it/s for cg! at 512^3): `solution(ws) === ws.x`, `ws.stats` is filled as the generic method fills
#      it, nothing is copied.
#
#      EVERY entry point of the reference reaches them.  `cg(A, b)`, `krylov_solve(Val(:cg), A, b)`, `krylov_solve!(ws, A, b)`
#      and the `x0` forms are generated methods that forward ALL keywords to `cg!(ws, A, b; ...)` explicitly -- including their
#      own default `callback = workspace -> false` (src/cg.jl:110; src/interface.jl:146-154, 160-170, 306-320).  `user_callback`
#      below recognises those defaults (anonymous functions of module Krylov that capture nothing) as "no callback", so the
#      device-resident loop is what `x, stats = cg(A_gpu, b_gpu)` runs; `NATIVE_SOLVES[]` / `LAST_PATH[]` record it and
#      test/runtests.jl asserts on them.  A REAL callback runs inside the library's host-driven loop on the fused kernels through a
#      `@cfunction` trampoline (`khip_options.callback`): it sees the workspace's own vectors and the residual history so far.
#      What the library cannot take (`ldiv = true`, a preconditioner that is not a `HIPOperator`, a verbose log into a non-file
#      `IO`) falls back to the generic method of layer 1 with `invoke` on the fully parametrised generic signature: same
#      results, primitive by primitive.
#
# Julia is not installed in the image this library is built in: this file is checked mechanically (tests/test_abi.py parses every
# `ccall` against include/krylov_hip.h -- symbol, arity, return and argument types -- and checks that every `k*` primitive the
# reference solvers call has a method here), and the same entry points, in the same order, are executed by the Python mirror
# (krylov.jl_amd/__init__.py, whose workspaces adopt their own vectors exactly as below) and by tests/c/adopt_sequence.c.
module KrylovHIP

using Krylov, LinearAlgebra, SparseArrays
import Krylov: kdot, kdotr, knorm, kscal!, kdiv!, kcopy!, kscalcopy!, kdivcopy!, kaxpy!, kaxpby!, kfill!, kref!, kmul!
import Krylov: CgWorkspace, GmresWorkspace, BicgstabWorkspace, BlockGmresWorkspace

export Ctx, CTX, HIPVector, HIPMatrix, HIPCsr, HIPOperator, jacobi, ilu0

# krylov.jl_amd/libkrylov_hip.so of this repository, or wherever KHIP_LIBRARY points
const lib = get(ENV, "KHIP_LIBRARY", normpath(joinpath(@__DIR__, "..", "..", "..", "krylov.jl_amd", "libkrylov_hip.so")))

ck(rc) = rc == 0 || error(unsafe_string(ccall((:khip_last_error, lib), Cstring, ())))

# ------------------------------------------------------------------------------------------------ context
mutable struct Ctx; h::Ptr{Cvoid}; end
function Ctx(device::Integer = 0)
  major = Ref{Cint}(); minor = Ref{Cint}()
  ccall((:khip_version, lib), Cvoid, (Ref{Cint}, Ref{Cint}), major, minor)
  (major[] == 0 && minor[] >= 4) || error("libkrylov_hip $(major[]).$(minor[]) lacks khip_*_last_path / the adopt entry points (need >= 0.4)")
  r = Ref{Ptr{Cvoid}}(); ck(ccall((:khip_ctx_create, lib), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, r))
  Ctx(r[])
end
const CTX = Ref{Ctx}()                            # one context per process (one process per GPU)
synchronize() = ck(ccall((:khip_ctx_sync, lib), Cint, (Ptr{Cvoid},), CTX[].h))

# ------------------------------------------------------------------------------------------------ device vector: the storage type S
mutable struct HIPVector <: AbstractVector{Float64}
  ptr::Ptr{Float64}; n::Int
  function HIPVector(::UndefInitializer, n::Integer)          # S(undef, n) / S(undef, 0) -- src/krylov_workspaces.jl:269-285
    r = Ref{Ptr{Cvoid}}(C_NULL)
    n > 0 && ck(ccall((:khip_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), CTX[].h, 8n, r))
    v = new(Ptr{Float64}(r[]), n)
    finalizer(x -> ccall((:khip_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[].h, x.ptr), v)
  end
end
Base.size(v::HIPVector) = (v.n,);  Base.length(v::HIPVector) = v.n
Base.similar(v::HIPVector) = HIPVector(undef, v.n)
Base.similar(v::HIPVector, ::Type{Float64}, dims::Dims{1}) = HIPVector(undef, dims[1])
Base.getindex(::HIPVector, i...) = error("scalar indexing of a device vector")   # like allowscalar(false), test/gpu/amd.jl:8
Krylov.ktypeof(::HIPVector) = HIPVector                                          # src/krylov_utils.jl:204-224
function HIPVector(x::AbstractVector{<:Real}); xh = Vector{Float64}(x); v = HIPVector(undef, length(xh))
  ck(ccall((:khip_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), CTX[].h, v.ptr, xh, 8length(xh))); v; end
function Base.Vector(v::HIPVector); x = Vector{Float64}(undef, v.n)
  ck(ccall((:khip_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), CTX[].h, x, v.ptr, 8v.n)); x; end
Base.Array(v::HIPVector) = Vector(v)
Base.copy(v::HIPVector) = kcopy!(v.n, similar(v), v)
Base.show(io::IO, v::HIPVector) = print(io, "HIPVector(", v.n, ")")
Base.show(io::IO, ::MIME"text/plain", v::HIPVector) = show(io, v)

# ------------------------------------------------------------------------------------------------ k* primitives: one ccall each
# (src/krylov_utils.jl line in the comment)
function kdot(n::Integer, x::HIPVector, y::HIPVector)                                       # :309-311
  r = Ref{Cdouble}(); ck(ccall((:khip_dot, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cdouble}), CTX[].h, n, x.ptr, y.ptr, r)); r[]
end
kdotr(n::Integer, x::HIPVector, y::HIPVector) = kdot(n, x, y)                                 # :313-314
function knorm(n::Integer, x::HIPVector)                                                     # :316-317
  r = Ref{Cdouble}(); ck(ccall((:khip_nrm2, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ref{Cdouble}), CTX[].h, n, x.ptr, r)); r[]
end
kscal!(n::Integer, s::Float64, x::HIPVector) = (ck(ccall((:khip_scal, lib), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Cdouble}), CTX[].h, n, s, x.ptr)); x)                      # :321-323
kdiv!(n::Integer, x::HIPVector, s::Float64) = (ck(ccall((:khip_div, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Cdouble), CTX[].h, n, x.ptr, s)); x)                        # :325-326
kcopy!(n::Integer, y::HIPVector, x::HIPVector) = (ck(ccall((:khip_copy, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, n, y.ptr, x.ptr)); y)           # :328-329 (dest, src)
kscalcopy!(n::Integer, y::HIPVector, s::Float64, x::HIPVector) = (ck(ccall((:khip_scalcopy, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}), CTX[].h, n, y.ptr, s, x.ptr)); y)   # :331-332
kdivcopy!(n::Integer, y::HIPVector, x::HIPVector, s::Float64) = (ck(ccall((:khip_divcopy, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble), CTX[].h, n, y.ptr, x.ptr, s)); y)     # :334-335
kaxpy!(n::Integer, s::Float64, x::HIPVector, y::HIPVector) = (ck(ccall((:khip_axpy, lib), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, n, s, x.ptr, y.ptr)); y)            # :337-339
kaxpby!(n::Integer, s::Float64, x::HIPVector, t::Float64, y::HIPVector) = (ck(ccall((:khip_axpby, lib), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}), CTX[].h, n, s, x.ptr, t, y.ptr)); y)  # :341-345
kfill!(x::HIPVector, val::Float64) = (ck(ccall((:khip_fill, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Cdouble), CTX[].h, x.n, x.ptr, val)); x)                           # :347
kref!(n::Integer, x::HIPVector, y::HIPVector, c::Float64, s::Float64) = (ck(ccall((:khip_ref, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Cdouble), CTX[].h, n, x.ptr, y.ptr, c, s)); (x, y))  # :349

# ------------------------------------------------------------------------------------------------ operator: size, eltype, kmul!
# (docs/src/matrix_free.md:32-34)
mutable struct HIPCsr; h::Ptr{Cvoid}; m::Int; n::Int; adj::Union{Nothing,HIPCsr}; end      # adj: A' once it has been built
HIPCsr(h::Ptr{Cvoid}, m::Integer, n::Integer) = HIPCsr(h, m, n, nothing)
destroy_csr(A::HIPCsr) = ccall((:khip_csr_destroy, lib), Cint, (Ptr{Cvoid},), A.h)
function HIPCsr(A::SparseArrays.SparseMatrixCSC{Float64,<:Integer})    # CSC of a symmetric matrix == its CSR;
  At = SparseArrays.sparse(A')                                         # general case: CSR of A = CSC of A'
  r = Ref{Ptr{Cvoid}}()
  ck(ccall((:khip_csr_create, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Cdouble}, Cint, Cint, Ref{Ptr{Cvoid}}),
           CTX[].h, size(A, 1), size(A, 2), SparseArrays.nnz(At), Int64.(At.colptr), 64, Int32.(At.rowval), At.nzval, 1, 0, r))   # index_base = 1
  finalizer(destroy_csr, HIPCsr(r[], size(A)...))
end
Base.size(A::HIPCsr) = (A.m, A.n);  Base.size(A::HIPCsr, i::Integer) = i == 1 ? A.m : (i == 2 ? A.n : 1);  Base.eltype(::HIPCsr) = Float64
kmul!(y::HIPVector, A::HIPCsr, x::HIPVector) = (ck(ccall((:khip_spmv, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, A.h, x.ptr, y.ptr)); y)   # src/krylov_utils.jl:305
kmul!(y::HIPVector, ::UniformScaling, x::HIPVector) = kcopy!(length(x), y, x)   # the unguarded mul!(v, I, q) of src/bicgstab.jl:222
LinearAlgebra.mul!(y::HIPVector, A::HIPCsr, x::HIPVector) = kmul!(y, A, x)
Base.:*(A::HIPCsr, x::HIPVector) = kmul!(HIPVector(undef, A.m), A, x)
# adjoint products for the solvers that need A' (LSQR, LSMR, BiLQ, QMR, ...; they write `Aᴴ = A'` once per solve): the transposed
# operator is built ONCE per matrix (khip_csr_transpose makes an independent handle), cached in `A.adj`, owned by a finalizer of its
# own, and points back so that (A')' === A.  Repeated solves neither rebuild nor leak it.
function Base.adjoint(A::HIPCsr)
  A.adj === nothing || return A.adj
  r = Ref{Ptr{Cvoid}}()
  ck(ccall((:khip_csr_transpose, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), CTX[].h, A.h, r))
  At = finalizer(destroy_csr, HIPCsr(r[], A.n, A.m, A))
  A.adj = At
  return At
end
Base.transpose(A::HIPCsr) = adjoint(A)            # real element type

# ------------------------------------------------------------------------------------------------ C structs of the solver entries
struct Operator                     # khip_operator, include/krylov_hip.h
  csr::Ptr{Cvoid}; apply::Ptr{Cvoid}; self::Ptr{Cvoid}
end
Operator(A::HIPCsr) = Operator(A.h, C_NULL, C_NULL)

struct Options                      # khip_options: same members, same order (isbits struct = the C layout)
  atol::Cdouble; rtol::Cdouble; itmax::Cint; timemax::Cdouble; history::Cint; radius::Cdouble; linesearch::Cint
  restart::Cint; reorthogonalization::Cint; fused::Cint; callback::Ptr{Cvoid}; callback_data::Ptr{Cvoid}
  variant::Cint; verbose::Cint; log_fd::Cint
end
function Options(; atol, rtol, itmax, timemax, history, radius = 0.0, linesearch = false, restart = false, reorthogonalization = false,
                 fused = 2, variant = 0, verbose = 0, log_fd = 0, callback = C_NULL, callback_data = C_NULL)
  # timemax: NaN or <= 0 means "no limit" on the C side (include/krylov_hip.h); itmax above the Cint range is "no limit" too.
  # A time limit the entry point has already used up (`timemax -= elapsed_time`, src/interface.jl:151) must stay a limit: the
  # smallest positive one.
  tm = isfinite(timemax) ? max(timemax, floatmin(Float64)) : NaN
  Options(atol, rtol, Cint(clamp(itmax, 0, typemax(Cint))), tm, history, radius, linesearch, restart,
          reorthogonalization, fused, callback, callback_data, variant, verbose, log_fd)
end

struct Stats                        # khip_stats
  niter::Cint; solved::Cint; inconsistent::Cint; indefinite::Cint; npcCount::Cint
  timer::Cdouble; status::NTuple{96,UInt8}; residuals::Ptr{Cdouble}; nres::Cint; error::NTuple{160,UInt8}; allocation_timer::Cdouble
end
cstr(t::NTuple{N,UInt8}) where N = (b = collect(t); i = findfirst(==(0x00), b); String(b[1:(i === nothing ? N : i - 1)]))

# native operators for the M / N arguments (z <- M r on device pointers): Jacobi and ILU(0) / IC(0)
mutable struct HIPOperator
  op::Base.RefValue{Operator}; n::Int; kind::Symbol; A::HIPCsr          # A must outlive the operator
end
function jacobi(A::HIPCsr); op = Ref(Operator(C_NULL, C_NULL, C_NULL))
  ck(ccall((:khip_jacobi_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Operator}), CTX[].h, A.h, op))
  finalizer(x -> ccall((:khip_jacobi_destroy, lib), Cint, (Ref{Operator},), x.op), HIPOperator(op, A.m, :jacobi, A)); end
function ilu0(A::HIPCsr); op = Ref(Operator(C_NULL, C_NULL, C_NULL))      # IC(0) for SPD A: docs/src/gpu.md:74-163 without the vendor library
  ck(ccall((:khip_ilu0_create, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Operator}), CTX[].h, A.h, op))
  finalizer(x -> ccall((:khip_ilu0_destroy, lib), Cint, (Ref{Operator},), x.op), HIPOperator(op, A.m, :ilu0, A)); end
Base.size(M::HIPOperator) = (M.n, M.n);  Base.eltype(::HIPOperator) = Float64
# the generic solvers apply M through mulorldiv! -> mul!(y, M, x) (src/krylov_utils.jl:307): call the operator's own `apply`
function LinearAlgebra.mul!(y::HIPVector, M::HIPOperator, x::HIPVector)
  rc = ccall(M.op[].apply, Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}), M.op[].self, x.ptr, y.ptr)
  rc == 0 || error("preconditioner application failed"); y
end
kmul!(y::HIPVector, M::HIPOperator, x::HIPVector) = mul!(y, M, x)

# ------------------------------------------------------------------------------------------------ shared pieces of the forwarding methods
# Julia workspace -> its khip_*_workspace handle: adopted once, destroyed by the workspace's finalizer.  Weak keys (identity of the mutable
# workspace): the table must not keep a workspace alive.  The finalizers capture the handle and do not touch the table.
const HANDLES = WeakKeyDict{Any,Ptr{Cvoid}}()
dptr(v::HIPVector) = isempty(v) ? Ptr{Cdouble}(C_NULL) : v.ptr
native_precond(M) = M === I || M isa HIPOperator
opref(M) = M === I ? Ptr{Operator}(C_NULL) : Base.unsafe_convert(Ptr{Operator}, M.op)
# verbose log: the C loop writes with dprintf to a file descriptor (options.log_fd, 0 = stdout)
# (`fd(::IOStream)` is an `Int` in older and a `RawFD` in newer Julia versions: `cconvert` takes both)
logfd(io::IO) = (io === Krylov.kstdout || io === stdout) ? Cint(0) : (io isa IOStream ? Base.cconvert(Cint, fd(io))::Cint : Cint(-1))
native_log(verbose, io) = verbose <= 0 || logfd(io) >= 0

# ---- callbacks ------------------------------------------------------------------------------------------------------------
# The reference's generated entry points pass `callback = workspace -> false` EXPLICITLY (see the header of this file): one
# anonymous function per generated method, all defined in module Krylov, none capturing anything.  Those are "no callback".
default_callback(cb) = parentmodule(typeof(cb)) === Krylov && Base.issingletontype(typeof(cb))
user_callback(cb) = (cb === nothing || default_callback(cb)) ? nothing : cb
# A user's callback runs inside the library's host-driven loop (khip_options.callback, include/krylov_hip.h): the trampoline
# brings the residual history of the C side into `ws.stats.residuals` (what the reference's callbacks read, docs/src/callbacks.md),
# calls `callback(ws)::Bool` and keeps an exception for after the solve (nothing may unwind through the C frames).
mutable struct CallbackBox
  f::Any; ws::Any; stats::Ptr{Cvoid}; history::Bool; err::Any
end
function callback_trampoline(_::Ptr{Cvoid}, ud::Ptr{Cvoid})::Cint
  box = unsafe_pointer_to_objref(ud)::CallbackBox
  try
    if box.history
      st = unsafe_load(Ptr{Stats}(box.stats)); res = box.ws.stats.residuals
      for i in (length(res) + 1):Int(st.nres); push!(res, unsafe_load(st.residuals, i)); end
    end
    return box.f(box.ws) ? Cint(1) : Cint(0)
  catch e
    box.err = e
    return Cint(1)                                 # stop the solve; `finish_callback` rethrows
  end
end
const CALLBACK = Ref{Ptr{Cvoid}}(C_NULL)
# (callback, callback_data) of khip_options and the box to GC.@preserve; `stats_ptr` = khip_*_stats(h) (stable per handle)
function callback_args(cb, ws, stats_ptr::Ptr{Stats}, history::Bool)
  cb === nothing && return C_NULL, C_NULL, nothing
  history && empty!(ws.stats.residuals)
  box = CallbackBox(cb, ws, Ptr{Cvoid}(stats_ptr), history, nothing)
  return CALLBACK[], pointer_from_objref(box), box
end
finish_callback(box) = (box !== nothing && box.err !== nothing) ? throw(box.err) : nothing

# how many solves took the library's loop, and which loop the last one ran (khip_*_last_path: 2 = device-resident / look-ahead,
# 1 = host-driven on the fused kernels, 0 = one launch per primitive) -- test/runtests.jl asserts on both after every entry point
const NATIVE_SOLVES = Ref(0)
const LAST_PATH = Ref(-1)
const GENERIC_SOLVES = Ref(0)                      # solves handed to the generic method by `invoke`

function fill_stats!(stats::Krylov.SimpleStats{Float64}, sp::Ptr{Stats}, history::Bool)
  st = unsafe_load(sp)
  Krylov.reset!(stats)
  history && st.nres > 0 && append!(stats.residuals, unsafe_wrap(Array, st.residuals, Int(st.nres)))
  stats.niter = st.niter;  stats.solved = st.solved != 0;  stats.inconsistent = st.inconsistent != 0
  stats.indefinite = st.indefinite != 0;  stats.npcCount = st.npcCount
  stats.timer = st.timer;  stats.status = cstr(st.status)
  return st
end
# a failed solve throws what the generic method throws: error(...) with the library's message (argument errors, SPD violation)
failed(st::Stats) = error(isempty(cstr(st.error)) ? unsafe_string(ccall((:khip_last_error, lib), Cstring, ())) : cstr(st.error))

# ------------------------------------------------------------------------------------------------ cg!  (src/cg.jl:120-291)
const CgWs = CgWorkspace{Float64,Float64,HIPVector}
function cg_handle(ws::CgWs)
  get!(HANDLES, ws) do
    r = Ref{Ptr{Cvoid}}()
    ck(ccall((:khip_cg_workspace_adopt, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
             CTX[].h, ws.m, ws.n, ws.x.ptr, ws.r.ptr, ws.p.ptr, ws.Ap.ptr, r))
    h = r[]
    finalizer(_ -> ccall((:khip_cg_workspace_destroy, lib), Cint, (Ptr{Cvoid},), h), ws)
    h
  end
end
cg_adopt(h, name, v::HIPVector) = ck(ccall((:khip_cg_workspace_adopt_vector, lib), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cdouble}), h, name, dptr(v)))

function Krylov.cg!(ws::CgWs, A::HIPCsr, b::HIPVector; M = I, ldiv::Bool = false, radius::Float64 = 0.0, linesearch::Bool = false,
                    atol::Float64 = √eps(Float64), rtol::Float64 = √eps(Float64), itmax::Int = 0, timemax::Float64 = Inf,
                    verbose::Int = 0, history::Bool = false, callback = nothing, iostream::IO = Krylov.kstdout, fused::Int = 2)
  if ldiv || !native_precond(M) || !native_log(verbose, iostream)
    GENERIC_SOLVES[] += 1
    return invoke(Krylov.cg!, Tuple{CgWs,Any,AbstractVector{Float64}}, ws, A, b; M, ldiv, radius, linesearch, atol, rtol, itmax, timemax, verbose,
                  history, callback = callback === nothing ? (w -> false) : callback, iostream)
  end
  m, n = size(A)                                                                            # the reference's own argument checks, :128-139
  (m == ws.m && n == ws.n) || error("(workspace.m, workspace.n) = ($(ws.m), $(ws.n)) is inconsistent with size(A) = ($m, $n)")
  m == n || error("System must be square")
  length(b) == n || error("Inconsistent problem size")
  linesearch && (radius > 0) && error("`linesearch` set to `true` but trust-region radius > 0")
  (ws.warm_start && linesearch) && error("warm_start and linesearch cannot be used together")
  Krylov.allocate_if(M !== I, ws, :z, HIPVector, ws.x)                                       # :142
  Krylov.allocate_if(linesearch || (radius > 0), ws, :npc_dir, HIPVector, ws.x)              # :143
  h = cg_handle(ws)
  for (name, v) in (("x", ws.x), ("r", ws.r), ("p", ws.p), ("Ap", ws.Ap), ("z", ws.z), ("npc_dir", ws.npc_dir), ("dx", ws.Δx))
    cg_adopt(h, name, v)                                                                    # the fields as they are NOW (pointer hand-over only)
  end
  ws.warm_start && ck(ccall((:khip_cg_warm_start, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, ws.Δx.ptr))   # Δx already holds x0: sets the flag
  sp = ccall((:khip_cg_stats, lib), Ptr{Stats}, (Ptr{Cvoid},), h)
  cbf, cbd, box = callback_args(user_callback(callback), ws, sp, history)
  opts = Ref(Options(; atol, rtol, itmax, timemax, history, radius, linesearch, fused, verbose, log_fd = logfd(iostream), callback = cbf, callback_data = cbd))
  opA = Ref(Operator(A))
  rc = GC.@preserve ws A b M opts opA box ccall((:khip_cg_solve, lib), Cint, (Ptr{Cvoid}, Ref{Operator}, Ptr{Operator}, Ptr{Cdouble}, Ref{Options}),
                                                 h, opA, opref(M), b.ptr, opts)
  st = fill_stats!(ws.stats, sp, history)
  NATIVE_SOLVES[] += 1;  LAST_PATH[] = ccall((:khip_cg_last_path, lib), Cint, (Ptr{Cvoid},), h)
  ws.warm_start = false
  finish_callback(box)
  rc == 0 || failed(st)
  return ws
end

# ------------------------------------------------------------------------------------------------ gmres!  (src/gmres.jl:121-384)
const GmresWs = GmresWorkspace{Float64,Float64,HIPVector}
basis_ptrs(V::Vector{HIPVector}) = Ptr{Cdouble}[v.ptr for v in V]
function gmres_handle(ws::GmresWs)
  get!(HANDLES, ws) do
    r = Ref{Ptr{Cvoid}}(); Vp = basis_ptrs(ws.V)
    ck(ccall((:khip_gmres_workspace_adopt, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Ptr{Cdouble}}, Ref{Ptr{Cvoid}}),
             CTX[].h, ws.m, ws.n, length(ws.c), ws.x.ptr, ws.w.ptr, Vp, r))
    ck(ccall((:khip_gmres_workspace_set_grow, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), r[], GROW_VECTOR[], pointer_from_objref(ws)))
    h = r[]
    finalizer(_ -> ccall((:khip_gmres_workspace_destroy, lib), Cint, (Ptr{Cvoid},), h), ws)
    h
  end
end
# khip_grow_fn: restart = false lets the basis outgrow `memory` -- push!(V, similar(x)), src/gmres.jl:319-324
function grow_vector(ud::Ptr{Cvoid})::Ptr{Cdouble}
  ws = unsafe_pointer_to_objref(ud)::GmresWs
  t0 = time_ns()
  v = try similar(ws.x) catch; return Ptr{Cdouble}(C_NULL) end
  push!(ws.V, v)
  ws.stats.allocation_timer += (time_ns() - t0) / 1e9
  return v.ptr
end
const GROW_VECTOR = Ref{Ptr{Cvoid}}(C_NULL)
gmres_adopt(h, name, v::HIPVector) = ck(ccall((:khip_gmres_workspace_adopt_vector, lib), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cdouble}), h, name, dptr(v)))

function Krylov.gmres!(ws::GmresWs, A::HIPCsr, b::HIPVector; M = I, N = I, ldiv::Bool = false, restart::Bool = false,
                       reorthogonalization::Bool = false, atol::Float64 = √eps(Float64), rtol::Float64 = √eps(Float64), itmax::Int = 0,
                       timemax::Float64 = Inf, verbose::Int = 0, history::Bool = false, callback = nothing, iostream::IO = Krylov.kstdout,
                       fused::Int = 2)
  if ldiv || !native_precond(M) || !native_precond(N) || !native_log(verbose, iostream)
    GENERIC_SOLVES[] += 1
    return invoke(Krylov.gmres!, Tuple{GmresWs,Any,AbstractVector{Float64}}, ws, A, b; M, N, ldiv, restart, reorthogonalization, atol, rtol, itmax,
                  timemax, verbose, history, callback = callback === nothing ? (w -> false) : callback, iostream)
  end
  m, n = size(A)                                                                            # :128-140
  (m == ws.m && n == ws.n) || error("(workspace.m, workspace.n) = ($(ws.m), $(ws.n)) is inconsistent with size(A) = ($m, $n)")
  m == n || error("System must be square")
  length(b) == m || error("Inconsistent problem size")
  Krylov.allocate_if(M !== I, ws, :q, HIPVector, ws.x)                                       # :142-144
  Krylov.allocate_if(N !== I, ws, :p, HIPVector, ws.x)
  Krylov.allocate_if(restart, ws, :Δx, HIPVector, ws.x)
  h = gmres_handle(ws)
  for (name, v) in (("x", ws.x), ("w", ws.w), ("p", ws.p), ("q", ws.q), ("dx", ws.Δx))
    gmres_adopt(h, name, v)
  end
  Vp = basis_ptrs(ws.V)                                                                     # the basis as it is now (it may have grown)
  ck(ccall((:khip_gmres_workspace_adopt_basis, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cdouble}}), h, length(Vp), Vp))
  ws.warm_start && ck(ccall((:khip_gmres_warm_start, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, ws.Δx.ptr))
  sp = ccall((:khip_gmres_stats, lib), Ptr{Stats}, (Ptr{Cvoid},), h)
  cbf, cbd, box = callback_args(user_callback(callback), ws, sp, history)
  opts = Ref(Options(; atol, rtol, itmax, timemax, history, restart, reorthogonalization, fused, verbose, log_fd = logfd(iostream), callback = cbf, callback_data = cbd))
  opA = Ref(Operator(A))
  rc = GC.@preserve ws A b M N opts opA box ccall((:khip_gmres_solve, lib), Cint, (Ptr{Cvoid}, Ref{Operator}, Ptr{Operator}, Ptr{Operator}, Ptr{Cdouble}, Ref{Options}),
                                                   h, opA, opref(M), opref(N), b.ptr, opts)
  st = fill_stats!(ws.stats, sp, history)
  NATIVE_SOLVES[] += 1;  LAST_PATH[] = ccall((:khip_gmres_last_path, lib), Cint, (Ptr{Cvoid},), h)
  # the host fields of the workspace in the reference's own storage (c, s, z, packed R, inner_iter; src/krylov_workspaces.jl:2866-2871)
  len = Ref{Cint}(); inner = Ref{Cint}()
  ck(ccall((:khip_gmres_host_state, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}),
           h, 0, C_NULL, C_NULL, C_NULL, C_NULL, len, inner))
  k = Int(len[]);  resize!(ws.c, k);  resize!(ws.s, k);  resize!(ws.z, k);  resize!(ws.R, div(k * (k + 1), 2))
  ck(ccall((:khip_gmres_host_state, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}),
           h, k, ws.c, ws.s, ws.z, ws.R, len, inner))
  ws.inner_iter = inner[]
  ws.warm_start = false
  finish_callback(box)
  rc == 0 || failed(st)
  return ws
end

# ------------------------------------------------------------------------------------------------ bicgstab!  (src/bicgstab.jl:125-277)
const BicgstabWs = BicgstabWorkspace{Float64,Float64,HIPVector}
function bicgstab_handle(ws::BicgstabWs)
  get!(HANDLES, ws) do
    r = Ref{Ptr{Cvoid}}()
    ck(ccall((:khip_bicgstab_workspace_adopt, lib), Cint,
             (Ptr{Cvoid}, Int64, Int64, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Ptr{Cvoid}}),
             CTX[].h, ws.m, ws.n, ws.x.ptr, ws.r.ptr, ws.p.ptr, ws.v.ptr, ws.s.ptr, ws.qd.ptr, r))
    h = r[]
    finalizer(_ -> ccall((:khip_bicgstab_workspace_destroy, lib), Cint, (Ptr{Cvoid},), h), ws)
    h
  end
end
bicgstab_adopt(h, name, v::HIPVector) = ck(ccall((:khip_bicgstab_workspace_adopt_vector, lib), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cdouble}), h, name, dptr(v)))

function Krylov.bicgstab!(ws::BicgstabWs, A::HIPCsr, b::HIPVector; c::HIPVector = b, M = I, N = I, ldiv::Bool = false,
                          atol::Float64 = √eps(Float64), rtol::Float64 = √eps(Float64), itmax::Int = 0, timemax::Float64 = Inf,
                          verbose::Int = 0, history::Bool = false, callback = nothing, iostream::IO = Krylov.kstdout, fused::Int = 2)
  if ldiv || !native_precond(M) || !native_precond(N) || !native_log(verbose, iostream)
    GENERIC_SOLVES[] += 1
    return invoke(Krylov.bicgstab!, Tuple{BicgstabWs,Any,AbstractVector{Float64}}, ws, A, b; c, M, N, ldiv, atol, rtol, itmax, timemax, verbose, history,
                  callback = callback === nothing ? (w -> false) : callback, iostream)
  end
  m, n = size(A)                                                                            # :132-143
  (m == ws.m && n == ws.n) || error("(workspace.m, workspace.n) = ($(ws.m), $(ws.n)) is inconsistent with size(A) = ($m, $n)")
  m == n || error("System must be square")
  length(b) == m || error("Inconsistent problem size")
  Krylov.allocate_if(M !== I, ws, :t, HIPVector, ws.x)                                       # :148-149
  Krylov.allocate_if(N !== I, ws, :yz, HIPVector, ws.x)
  h = bicgstab_handle(ws)
  for (name, v) in (("x", ws.x), ("r", ws.r), ("p", ws.p), ("v", ws.v), ("s", ws.s), ("qd", ws.qd), ("yz", ws.yz), ("t", ws.t), ("dx", ws.Δx))
    bicgstab_adopt(h, name, v)
  end
  ws.warm_start && ck(ccall((:khip_bicgstab_warm_start, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, ws.Δx.ptr))
  sp = ccall((:khip_bicgstab_stats, lib), Ptr{Stats}, (Ptr{Cvoid},), h)
  cbf, cbd, box = callback_args(user_callback(callback), ws, sp, history)
  opts = Ref(Options(; atol, rtol, itmax, timemax, history, fused, verbose, log_fd = logfd(iostream), callback = cbf, callback_data = cbd))
  opA = Ref(Operator(A))
  rc = GC.@preserve ws A b c M N opts opA box ccall((:khip_bicgstab_solve, lib), Cint,
                                                 (Ptr{Cvoid}, Ref{Operator}, Ptr{Operator}, Ptr{Operator}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Options}),
                                                 h, opA, opref(M), opref(N), b.ptr, c.ptr, opts)
  st = fill_stats!(ws.stats, sp, history)
  NATIVE_SOLVES[] += 1;  LAST_PATH[] = ccall((:khip_bicgstab_last_path, lib), Cint, (Ptr{Cvoid},), h)
  ws.warm_start = false
  finish_callback(box)
  rc == 0 || failed(st)
  return ws
end

# ------------------------------------------------------------------------------------------------ device matrix (block solvers)
# Tall blocks are libkrylov_hip panels (ROW-major, rows padded to 16, padding rows zero), small blocks host matrices -- exactly
# where the library keeps them (csrc/block.cpp) and where the reference's own small LAPACK calls run.  INTEGRATION.md has the
# line-by-line table of what block_gmres! asks of its matrix type.
mutable struct HIPMatrix <: AbstractMatrix{Float64}
  ptr::Ptr{Float64}; host::Union{Nothing,Matrix{Float64}}; m::Int; k::Int
  function HIPMatrix(::UndefInitializer, m::Integer, k::Integer)
    m <= 4k && return new(C_NULL, Matrix{Float64}(undef, m, k), m, k)              # Z, C, D, R, H and the (0, 0) placeholders
    np = Ref{Int64}(); ck(ccall((:khip_panel_rows, lib), Cint, (Int64, Ref{Int64}), m, np))
    r = Ref{Ptr{Cvoid}}(C_NULL); ck(ccall((:khip_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), CTX[].h, 8np[] * k, r))
    A = new(Ptr{Float64}(r[]), nothing, m, k)
    ck(ccall((:khip_fill, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Cdouble), CTX[].h, np[] * k, A.ptr, 0.0))   # padding rows stay zero
    finalizer(x -> ccall((:khip_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), CTX[].h, x.ptr), A)
  end
end
tall(A::HIPMatrix) = A.host === nothing
plen(A::HIPMatrix) = (np = Ref{Int64}(); ccall((:khip_panel_rows, lib), Cint, (Int64, Ref{Int64}), A.m, np); np[] * A.k)
Base.size(A::HIPMatrix) = (A.m, A.k)
Base.isempty(A::HIPMatrix) = A.m == 0 || A.k == 0
Base.getindex(A::HIPMatrix, i...) = tall(A) ? error("scalar indexing of a device panel") : A.host[i...]
Base.setindex!(A::HIPMatrix, v, i...) = (A.host[i...] = v)                       # H[k][1:p,:] .= R, small blocks only
Base.view(A::HIPMatrix, I...) = view(A.host, I...)                               # D1 = view(D, 1:p, :)
Krylov.ktypeof(::HIPMatrix) = HIPMatrix
Krylov.matrix_to_vector(::Type{HIPMatrix}) = Vector{Float64}                    # tau, buffer: host
function HIPMatrix(B::Matrix{Float64}); A = HIPMatrix(undef, size(B)...); tall(A) || (A.host .= B; return A)
  t = HIPVector(vec(B)); ck(ccall((:khip_panel_from_colmajor, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, A.m, A.k, t.ptr, A.ptr)); A; end
function Base.Matrix(A::HIPMatrix); tall(A) || return copy(A.host)
  t = HIPVector(undef, A.m * A.k); ck(ccall((:khip_panel_to_colmajor, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, A.m, A.k, A.ptr, t.ptr))
  reshape(Vector(t), A.m, A.k); end
Base.fill!(A::HIPMatrix, v) = tall(A) ? (ck(ccall((:khip_fill, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Cdouble), CTX[].h, plen(A), A.ptr, v)); A) : (fill!(A.host, v); A)
Krylov.kfill!(A::HIPMatrix, v) = fill!(A, v)
Base.copyto!(D::HIPMatrix, S::HIPMatrix) = tall(D) ? (ck(ccall((:khip_copy, lib), Cint, (Ptr{Cvoid}, Int64, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, plen(D), D.ptr, S.ptr)); D) : (copyto!(D.host, S.host); D)
LinearAlgebra.norm(A::HIPMatrix) = tall(A) ? (r = Ref{Cdouble}(); ck(ccall((:khip_panel_norm, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cdouble}, Ref{Cdouble}), CTX[].h, A.m, A.k, A.ptr, r)); r[]) : norm(A.host)
# the two panel broadcasts of the solver: W .= B .- W (:160, :207) and X .+= ΔX (:161, :331); small blocks broadcast as Matrix
Base.BroadcastStyle(::Type{HIPMatrix}) = Broadcast.ArrayStyle{HIPMatrix}()
function Base.copyto!(D::HIPMatrix, bc::Broadcast.Broadcasted{Broadcast.ArrayStyle{HIPMatrix}})
  tall(D) || (copyto!(D.host, Broadcast.Broadcasted(bc.f, map(a -> a isa HIPMatrix ? a.host : a, bc.args))); return D)
  a, b = bc.args
  if bc.f === (-) && b === D      # D .= a .- D
    ck(ccall((:khip_axpby, lib), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}), CTX[].h, plen(D), 1.0, a.ptr, -1.0, D.ptr))
  elseif bc.f === (+) && a === D  # D .+= b
    ck(ccall((:khip_axpy, lib), Cint, (Ptr{Cvoid}, Int64, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, plen(D), 1.0, b.ptr, D.ptr))
  else error("panel broadcast not used by block_gmres!") end
  D
end
# products (:242, :245, :246, :325)
# (a system with n <= 4p keeps even X, W and V on the host: its products go column by column through the SpMV -- correctness only)
function LinearAlgebra.mul!(W::HIPMatrix, A::HIPCsr, P::HIPMatrix)
  if !tall(P)
    for j in 1:P.k; W.host[:, j] = Vector(kmul!(HIPVector(undef, A.m), A, HIPVector(P.host[:, j]))); end
    return W
  end
  ck(ccall((:khip_spmm, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Cint), CTX[].h, A.h, P.ptr, W.ptr, P.k)); W
end
function LinearAlgebra.mul!(Ψ::HIPMatrix, Vt::Adjoint{Float64,HIPMatrix}, Q::HIPMatrix)
  V = parent(Vt)
  tall(V) || return (mul!(Ψ.host, V.host', Q.host); Ψ)
  ck(ccall((:khip_panel_gemm_tn, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, V.m, V.k, V.ptr, Q.ptr, Ψ.host)); Ψ
end
function LinearAlgebra.mul!(Q::HIPMatrix, V::HIPMatrix, Ψ::HIPMatrix, α::Number, β::Number)
  tall(Q) || return (mul!(Q.host, V.host, Ψ.host, α, β); Q)                      # Y[i] -= R[pos] Y[j]   (:317)
  ck(ccall((:khip_panel_gemm_nn, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}, Cdouble, Ptr{Cdouble}), CTX[].h, V.m, V.k, α, V.ptr, Ψ.host, β, Q.ptr)); Q
end
LinearAlgebra.ldiv!(U::UpperTriangular{Float64,HIPMatrix}, Y::HIPMatrix) = (ldiv!(UpperTriangular(parent(U).host), Y.host); Y)   # :320
Base.:*(A::HIPCsr, X::HIPMatrix) = mul!(HIPMatrix(undef, A.m, X.k), A, X)
# householder!(Q, R, tau, buffer; compact) (src/block_krylov_utils.jl:201-208): tall Q on the device, small H blocks with LAPACK.
# The workspace's `buffer` has length 0 for any SM that is not a `Matrix` (src/block_krylov_workspaces.jl:155-158), and the
# `Matrix{Float64}` methods taking a buffer pass lwork = length(buffer) to dgeqrf / dormqr and ignore `info`
# (src/block_krylov_utils.jl:230-236, :282-290): with lwork = 0 they would silently do nothing.  The small host blocks therefore
# go through the BUFFER-LESS forms (src/block_krylov_utils.jl:192-199 and the generic fallbacks :294-300 = LAPACK.geqrf! /
# orgqr! / ormqr!, which size their own workspace).
function Krylov.householder!(Q::HIPMatrix, R::HIPMatrix, τ::Vector{Float64}, buffer::Vector{Float64}; compact::Bool=false)
  if tall(Q)                       # kgeqrf! + copy_triangle + korgqr! in one call; same Q, R, tau as LAPACK
    ck(ccall((:khip_panel_qr_tau, lib), Cint, (Ptr{Cvoid}, Int64, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}), CTX[].h, Q.m, Q.k, Q.ptr, R.host, τ))
  else
    Krylov.householder!(Q.host, R.host, τ; compact)          # 3-argument form: no (empty) buffer
  end
  Q, R
end
Krylov.kormqr!(side::Char, trans::Char, H::HIPMatrix, τ::Vector{Float64}, D::HIPMatrix, buffer::Vector{Float64}) = Krylov.kormqr!(side, trans, H.host, τ, D.host)   # :266, :279 -- 5-argument form

# ------------------------------------------------------------------------------------------------ block_gmres!  (src/block_gmres.jl:110-358)
const BlockGmresWs = BlockGmresWorkspace{Float64,Float64,Vector{Float64},HIPMatrix}
panel_ptrs(V::Vector{HIPMatrix}) = Ptr{Cdouble}[v.ptr for v in V]
pptr(A::HIPMatrix) = (isempty(A) || !tall(A)) ? Ptr{Cdouble}(C_NULL) : A.ptr
function block_gmres_handle(ws::BlockGmresWs)
  get!(HANDLES, ws) do
    r = Ref{Ptr{Cvoid}}(); Vp = panel_ptrs(ws.V)
    ck(ccall((:khip_block_gmres_workspace_adopt, lib), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Ptr{Cdouble}}, Ref{Ptr{Cvoid}}),
             CTX[].h, ws.m, ws.n, ws.p, length(ws.V), ws.X.ptr, ws.W.ptr, Vp, r))
    ck(ccall((:khip_block_gmres_workspace_set_grow, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), r[], GROW_PANEL[], pointer_from_objref(ws)))
    h = r[]
    finalizer(_ -> ccall((:khip_block_gmres_workspace_destroy, lib), Cint, (Ptr{Cvoid},), h), ws)
    h
  end
end
function grow_panel(ud::Ptr{Cvoid})::Ptr{Cdouble}            # push!(V, SM(undef, n, p)), src/block_gmres.jl:300-305 (a zeroed panel)
  ws = unsafe_pointer_to_objref(ud)::BlockGmresWs
  t0 = time_ns()
  P = try HIPMatrix(undef, ws.n, ws.p) catch; return Ptr{Cdouble}(C_NULL) end
  push!(ws.V, P)
  ws.stats.allocation_timer += (time_ns() - t0) / 1e9
  return P.ptr
end
const GROW_PANEL = Ref{Ptr{Cvoid}}(C_NULL)
block_adopt(h, name, A::HIPMatrix) = ck(ccall((:khip_block_gmres_workspace_adopt_panel, lib), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cdouble}), h, name, pptr(A)))

function Krylov.block_gmres!(ws::BlockGmresWs, A::HIPCsr, B::HIPMatrix; M = I, N = I, ldiv::Bool = false, restart::Bool = false,
                             reorthogonalization::Bool = false, atol::Float64 = √eps(Float64), rtol::Float64 = √eps(Float64), itmax::Int = 0,
                             timemax::Float64 = Inf, verbose::Int = 0, history::Bool = false, callback = nothing, iostream::IO = Krylov.kstdout)
  n, p = ws.n, ws.p
  # native loop: M = N = I (the block solver's preconditioners would be applied to panels), tall panels
  if ldiv || M !== I || N !== I || !native_log(verbose, iostream) || !tall(B) || !tall(ws.X)
    GENERIC_SOLVES[] += 1
    return invoke(Krylov.block_gmres!, Tuple{BlockGmresWs,Any,AbstractMatrix{Float64}}, ws, A, B; M, N, ldiv, restart, reorthogonalization, atol, rtol,
                  itmax, timemax, verbose, history, callback = callback === nothing ? (w -> false) : callback, iostream)
  end
  m, nA = size(A);  s, pB = size(B)                                                          # :117-128
  (m == ws.m && nA == ws.n) || error("(workspace.m, workspace.n) = ($(ws.m), $(ws.n)) is inconsistent with size(A) = ($m, $nA)")
  m == nA || error("System must be square")
  nA == s || error("Inconsistent problem size")
  pB == p || error("the workspace was built for $p right-hand sides, B has $pB")
  Krylov.allocate_if(restart, ws, :ΔX, HIPMatrix, n, p)                                       # :145
  h = block_gmres_handle(ws)
  for (name, P) in (("X", ws.X), ("W", ws.W), ("P", ws.P), ("Q", ws.Q), ("dX", ws.ΔX))
    block_adopt(h, name, P)
  end
  Vp = panel_ptrs(ws.V)
  ck(ccall((:khip_block_gmres_workspace_adopt_basis, lib), Cint, (Ptr{Cvoid}, Cint, Ptr{Ptr{Cdouble}}), h, length(Vp), Vp))
  ws.warm_start && ck(ccall((:khip_block_gmres_warm_start_panel, lib), Cint, (Ptr{Cvoid}, Ptr{Cdouble}), h, ws.ΔX.ptr))
  sp = ccall((:khip_block_gmres_stats, lib), Ptr{Stats}, (Ptr{Cvoid},), h)
  cbf, cbd, box = callback_args(user_callback(callback), ws, sp, history)
  opts = Ref(Options(; atol, rtol, itmax, timemax, history, restart, reorthogonalization, verbose, log_fd = logfd(iostream), callback = cbf, callback_data = cbd))
  opA = Ref(Operator(A))
  rc = GC.@preserve ws A B opts opA box ccall((:khip_block_gmres_solve_panel, lib), Cint,
                                           (Ptr{Cvoid}, Ref{Operator}, Ptr{Operator}, Ptr{Operator}, Ptr{Cdouble}, Ref{Options}),
                                           h, opA, C_NULL, C_NULL, B.ptr, opts)
  st = fill_stats!(ws.stats, sp, history)
  NATIVE_SOLVES[] += 1;  LAST_PATH[] = ccall((:khip_block_gmres_last_path, lib), Cint, (Ptr{Cvoid},), h)
  ws.warm_start = false
  finish_callback(box)
  rc == 0 || failed(st)
  return ws
end

function __init__()
  GROW_VECTOR[] = @cfunction(grow_vector, Ptr{Cdouble}, (Ptr{Cvoid},))
  GROW_PANEL[] = @cfunction(grow_panel, Ptr{Cdouble}, (Ptr{Cvoid},))
  CALLBACK[] = @cfunction(callback_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}))
  return nothing
end

end # module
