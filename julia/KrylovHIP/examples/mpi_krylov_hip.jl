# Multi-GPU from Julia: one process per GPU, the reference's MPIVector recipe (docs/src/custom_workspaces.md:477-637) with RCCL inside
# the library.  The slice of a rank IS a HIPVector of the local length: once the context has a communicator, khip_dot / khip_nrm2
# return the GLOBAL value on every rank and khip_spmv on a row-partitioned handle fetches the remote entries of x first, so the
# package is unchanged; what a distributed run adds is the set-up below.  MPI.jl only broadcasts the 128-byte ncclUniqueId.
# (INTEGRATION.md quotes this file; tests/test_abi.py checks its ccalls against include/krylov_hip.h.)
# mpi_krylov_hip.jl — run as: mpiexecjl -n 8 julia mpi_krylov_hip.jl     (cfg 4: cg! on get_div_grad(1024^3) over 8 MI355X)
using MPI, Krylov, SparseArrays
using KrylovHIP: CTX, Ctx, HIPVector, HIPCsr, lib, ck
MPI.Init();  comm = MPI.COMM_WORLD;  rank = MPI.Comm_rank(comm);  nranks = MPI.Comm_size(comm)
CTX[] = Ctx(rank % 8)                                                   # one GPU per process (LOCAL rank on a multi-node job)
id = zeros(UInt8, 128)
rank == 0 && ck(ccall((:khip_comm_unique_id, lib), Cint, (Ptr{Cvoid},), id))
MPI.Bcast!(id, 0, comm)
ck(ccall((:khip_comm_init, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}), CTX[].h, rank, nranks, id))

# this rank's rows [row0, row0 + m) of the global operator, columns GLOBAL (1-based here: index_base = 1), as CSR
function HIPCsrDist(Aloc_t::SparseMatrixCSC{Float64,<:Integer}, n_global::Integer, row0::Integer)   # Aloc_t = (rows of A)' : its CSC is their CSR
  m = size(Aloc_t, 2);  r = Ref{Ptr{Cvoid}}()
  ck(ccall((:khip_csr_create_dist, lib), Cint,
           (Ptr{Cvoid}, Int64, Int64, Int64, Int64, Ptr{Cvoid}, Cint, Ptr{Int32}, Ptr{Cdouble}, Cint, Cint, Ref{Ptr{Cvoid}}),
           CTX[].h, n_global, row0, m, nnz(Aloc_t), Int64.(Aloc_t.colptr), 64, Int32.(Aloc_t.rowval), Aloc_t.nzval, 1, 0, r))
  HIPCsr(r[], m, m)                                                     # size() = the LOCAL slice: workspaces hold local vectors
end

n1 = 1024;  n = n1^3;  chunk = cld(n, nranks);  row0 = rank * chunk;  m = min(chunk, n - row0)
A = HIPCsrDist(local_rows_of_get_div_grad(n1, row0, m), n, row0)        # (or khip_gen_stencil: generated in HBM, no host copy)
b = HIPVector(ones(m))
ws = CgWorkspace(KrylovConstructor(b))                                   # 4 local vectors of m entries, src/krylov_workspaces.jl:250-267
cg!(ws, A, b; atol = 0.0, rtol = 1e-8, itmax = n)                        # kdot / knorm are global, kmul! exchanges the halo
x_local = Vector(Krylov.solution(ws))                                    # this rank's slice of x
ck(ccall((:khip_comm_barrier, lib), Cint, (Ptr{Cvoid},), CTX[].h));  MPI.Finalize()
