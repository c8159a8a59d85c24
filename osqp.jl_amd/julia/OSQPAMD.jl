# OSQPAMD.jl -- thin Julia layer over the extension entry points of libosqp_amd.so.
#
# The 30 `osqp_*` symbols that osqp/OSQP.jl binds need no new Julia code: point `OSQP.osqp` at
# libosqp_amd.so (INTEGRATION.md, section 2) and the reference's `setup!/solve!/update!/warm_start!`
# and its MathOptInterface wrapper run on the MI355X unchanged.  This module only adds what the
# reference has no call for: problems generated directly in HBM, the batched small-QP path, device
# selection, statistics.  Style follows [REF src/interface.jl]: one `ccall` per entry point, a
# non-zero exit flag becomes `error(...)`.
#
# NOTE: Julia is not installed in the build image, so this file is not executed by the test suite;
# the same entry points are exercised through the Python mirror (osqp.jl_amd/interface.py, batch.py).
module OSQPAMD

using OSQP
using SparseArrays

const lib = get(ENV, "OSQP_AMD_LIB", joinpath(@__DIR__, "..", "csrc", "libosqp_amd.so"))
const Cc_int = OSQP.Cc_int

@enum ProblemKind RANDOM_QP = 0 LASSO = 1 MPC = 2

"""
    set_device(local_rank)

Select the HIP device for workspaces created afterwards (one process per GPU).
"""
function set_device(device::Integer)
    flag = ccall((:osqp_amd_set_device, lib), Cc_int, (Cc_int,), device)
    flag == 0 || error("Error selecting device $(device): $(last_error())")
    return nothing
end

last_error() = unsafe_string(ccall((:osqp_amd_last_error, lib), Cstring, ()))

"""
    setup_generated!(model, kind, n; per_row = 0, seed = 1, settings...)

Build one of the synthetic problem families of SURVEY.md 8d directly in HBM and run setup on it.
Mirrors `OSQP.setup!` [REF src/interface.jl:35-162] after the point where the data are handed to C.
"""
function setup_generated!(model::OSQP.Model, kind::ProblemKind, n::Integer; per_row::Integer = 0, seed::Integer = 1, settings...)
    settings_dict = Dict{Symbol,Any}(settings)
    stgs = OSQP.Settings(settings_dict)
    workspace = Ref{Ptr{OSQP.Workspace}}()
    flag = ccall(
        (:osqp_amd_setup_generated, lib),
        Cc_int,
        (Ptr{Ptr{OSQP.Workspace}}, Cc_int, Cc_int, Cc_int, Culonglong, Ptr{OSQP.Settings}),
        workspace, Int(kind), n, per_row, seed, Ref(stgs),
    )
    flag == 0 || error("Error in OSQP setup: $(last_error())")
    model.workspace = workspace[]
    (nn, m) = OSQP.dimensions(model)
    resize!(model.lcache, m)
    resize!(model.ucache, m)
    model.isempty = false
    return model
end

"""
    stats(model) -> Vector{Float64}

Back-end in use, non-zero counts, nnz(L), CG / ADMM iteration totals, device bytes, algorithmic bytes
of the dominant kernels (see include/osqp_amd.h, `osqp_amd_get_stats`).
"""
function stats(model::OSQP.Model)
    out = zeros(Float64, 26)  # OSQP_AMD_STATS_COUNT (include/osqp_amd.h)
    k = ccall((:osqp_amd_get_stats, lib), Cc_int, (Ptr{OSQP.Workspace}, Ptr{Cdouble}, Cc_int), model.workspace, out, length(out))
    return out[1:k]
end

"""
    iterate!(model, iters)

Run exactly `iters` ADMM iterations from the current iterate (benchmark hook).
"""
function iterate!(model::OSQP.Model, iters::Integer)
    flag = ccall((:osqp_amd_iterate, lib), Cc_int, (Ptr{OSQP.Workspace}, Cc_int), model.workspace, iters)
    flag == 0 || error("Error iterating: $(last_error())")
    return nothing
end

"""
    batch_solve(P, A, Px, Ax, q, l, u; device = 0, settings...) -> (x, y, infos)

`count` independent QPs that share the sparsity pattern of `P` (upper triangle) and `A`; the value arrays
are `count x nnz` / `count x n|m` matrices stored row-major on the C side, hence the transposes.
One workgroup per QP with the reduced KKT system factorised in LDS (csrc/batch.hip).
"""
function batch_solve(P::SparseMatrixCSC, A::SparseMatrixCSC, Px::Matrix{Float64}, Ax::Matrix{Float64},
                     q::Matrix{Float64}, l::Matrix{Float64}, u::Matrix{Float64}; device::Integer = 0, settings...)
    Pu = istriu(P) ? P : triu(P)
    n = size(A, 2); m = size(A, 1); count = size(q, 1)
    stgs = OSQP.Settings(Dict{Symbol,Any}(settings))
    Pp = convert(Vector{Cc_int}, Pu.colptr .- 1); Pi = convert(Vector{Cc_int}, Pu.rowval .- 1)
    Ap = convert(Vector{Cc_int}, A.colptr .- 1);  Ai = convert(Vector{Cc_int}, A.rowval .- 1)
    # row-major [count x .] on the C side = column-major [. x count] here
    Pxt = permutedims(Px); Axt = permutedims(Ax); qt = permutedims(q)
    lt = permutedims(max.(l, -OSQP.OSQP_INFTY)); ut = permutedims(min.(u, OSQP.OSQP_INFTY))
    x = Matrix{Float64}(undef, n, count); y = Matrix{Float64}(undef, m, count)
    infos = Vector{OSQP.CInfo}(undef, count)
    flag = ccall(
        (:osqp_amd_batch_solve, lib),
        Cc_int,
        (Cc_int, Cc_int, Cc_int, Ptr{Cc_int}, Ptr{Cc_int}, Ptr{Cdouble}, Ptr{Cc_int}, Ptr{Cc_int}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{OSQP.Settings}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{OSQP.CInfo}, Cc_int),
        count, n, m, Pp, Pi, Pxt, Ap, Ai, Axt, qt, lt, ut, Ref(stgs), x, y, infos, device,
    )
    flag == 0 || error("Error in batched solve: $(last_error())")
    results = map(infos) do ci
        info = OSQP.Info()
        OSQP.copyto!(info, ci)
        info
    end
    return permutedims(x), permutedims(y), results
end

# ---------------------------------------------------------------------------------------------
# Row-sharded solve of one large QP over several GPUs (include/osqp_amd.h, "Row-sharded workspaces")
# ---------------------------------------------------------------------------------------------

"""
Opaque handle of the library's communicator (one in-place all-gather of doubles).  Must outlive the models
set up with it.
"""
mutable struct Comm
    handle::Ptr{Cvoid}
    rank::Int
    world::Int
    keep::Any   # the @cfunction / closure of the host transport, kept alive
end

function close!(comm::Comm)
    comm.handle == C_NULL || ccall((:osqp_amd_comm_destroy, lib), Cc_int, (Ptr{Cvoid},), comm.handle)
    comm.handle = C_NULL
    return nothing
end

"""
    unique_id() -> Vector{UInt8}

The 128 bytes of an ncclUniqueId; call on one rank and distribute (e.g. `MPI.Bcast!`).
"""
function unique_id(; librccl::Union{Nothing,String} = nothing)
    id = Vector{UInt8}(undef, 128)
    flag = ccall((:osqp_amd_comm_unique_id, lib), Cc_int, (Ptr{UInt8}, Cstring), id, librccl === nothing ? C_NULL : librccl)
    flag == 0 || error("Error creating the RCCL id: $(last_error())")
    return id
end

"""
    rccl_comm(rank, world, id) -> Comm

ncclAllGather on the engine's stream (RCCL over xGMI); `id` from `unique_id()` of rank 0.
"""
function rccl_comm(rank::Integer, world::Integer, id::Vector{UInt8}; librccl::Union{Nothing,String} = nothing)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    flag = ccall((:osqp_amd_comm_create_rccl, lib), Cc_int, (Ptr{Ptr{Cvoid}}, Cc_int, Cc_int, Ptr{UInt8}, Cstring),
                 h, rank, world, id, librccl === nothing ? C_NULL : librccl)
    flag == 0 || error("Error creating the RCCL communicator: $(last_error())")
    return Comm(h[], rank, world, nothing)
end

"""
    host_comm(rank, world, allgather!) -> Comm

Host-staged transport: `allgather!(buf::Vector{Float64}, count)` receives world*count doubles with chunk `rank`
filled in and fills in the others (e.g. `MPI.Allgather!(MPI.IN_PLACE, UBuffer(buf, count), comm)`).
"""
function host_comm(rank::Integer, world::Integer, allgather!)
    function trampoline(ctx::Ptr{Cvoid}, buf::Ptr{Cdouble}, count::Clonglong)::Cint
        try
            allgather!(unsafe_wrap(Array, buf, world * count), Int(count))
            return 0
        catch
            return 1
        end
    end
    cb = @cfunction($trampoline, Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Clonglong))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    flag = ccall((:osqp_amd_comm_create_host, lib), Cc_int, (Ptr{Ptr{Cvoid}}, Cc_int, Cc_int, Ptr{Cvoid}, Ptr{Cvoid}),
                 h, rank, world, cb, C_NULL)
    flag == 0 || error("Error creating the host communicator: $(last_error())")
    return Comm(h[], rank, world, cb)
end

"""
    setup_sharded!(model, comm; P, q, A, l, u, settings...)

As `OSQP.setup!` [REF src/interface.jl:35-162] with the same (full) problem on every rank; the library keeps
this rank's row block.  `OSQP.solve!`, `update_q!`, `update_bounds!`, `warm_start!` then work unchanged (every
rank calls them in the same order and receives the full solution).
"""
function setup_sharded!(model::OSQP.Model, comm::Comm; P::SparseMatrixCSC, q::Vector{Float64}, A::SparseMatrixCSC,
                        l::Vector{Float64}, u::Vector{Float64}, settings...)
    n = size(P, 1); m = size(A, 1)
    Pu = istriu(P) ? P : triu(P)
    u = min.(u, OSQP.OSQP_INFTY); l = max.(l, -OSQP.OSQP_INFTY)
    managedP = OSQP.ManagedCcsc(Pu); managedA = OSQP.ManagedCcsc(A)
    Pdata = Ref(OSQP.Ccsc(managedP)); Adata = Ref(OSQP.Ccsc(managedA))
    stgs = OSQP.Settings(Dict{Symbol,Any}(settings))
    workspace = Ref{Ptr{OSQP.Workspace}}()
    flag = GC.@preserve managedP Pdata managedA Adata q l u begin   # as [REF src/interface.jl:132-155]
        data = OSQP.Data(n, m, Base.unsafe_convert(Ptr{OSQP.Ccsc}, Pdata), Base.unsafe_convert(Ptr{OSQP.Ccsc}, Adata),
                         pointer(q), pointer(l), pointer(u))
        ccall((:osqp_amd_setup_sharded, lib), Cc_int,
              (Ptr{Ptr{OSQP.Workspace}}, Ptr{OSQP.Data}, Ptr{OSQP.Settings}, Ptr{Cvoid}),
              workspace, Ref(data), Ref(stgs), comm.handle)
    end
    flag == 0 || error("Error in OSQP setup: $(last_error())")
    model.workspace = workspace[]
    resize!(model.lcache, m); resize!(model.ucache, m)
    model.isempty = false
    return model
end

# ---------------------------------------------------------------------------------------------------------
# The batched path over several GPUs (include/osqp_amd.h: osqp_amd_batch_mpc_*): `total` MPC instances cut into
# contiguous blocks over the ranks of a communicator, resident in HBM; `batch_mpc_solve!` = every rank its block
# (one workgroup per instance) + ONE in-place all-gather of the packed rows [x (100) | y (200) | iter, status, pri, dua]
# on the library's own RCCL communicator.  `packed` is a DEVICE pointer to total * 304 doubles (e.g. from AMDGPU.jl's
# allocator or hipMalloc through ccall); nothing else crosses ranks.
# ---------------------------------------------------------------------------------------------------------
mutable struct MpcBatch
    handle::Ptr{Cvoid}
    total::Int
end

function batch_mpc_create(total::Integer; seed::Integer = 1, comm::Ptr{Cvoid} = C_NULL, device::Integer = 0, settings...)
    stgs = OSQP.Settings(Dict{Symbol,Any}(settings))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    flag = ccall((:osqp_amd_batch_mpc_create, lib), Cc_int,
                 (Ptr{Ptr{Cvoid}}, Cc_int, Culonglong, Ptr{OSQP.Settings}, Ptr{Cvoid}, Cc_int),
                 h, total, seed, Ref(stgs), comm, device)
    flag == 0 || error("Error in batched setup: $(last_error())")
    b = MpcBatch(h[], total)
    finalizer(x -> ccall((:osqp_amd_batch_destroy, lib), Cc_int, (Ptr{Cvoid},), x.handle), b)
    return b
end

function batch_mpc_solve!(b::MpcBatch, packed::Ptr{Cdouble})
    flag = ccall((:osqp_amd_batch_mpc_solve, lib), Cc_int, (Ptr{Cvoid}, Ptr{Cdouble}), b.handle, packed)
    flag == 0 || error("Error in batched solve: $(last_error())")
    return nothing
end

"In-place all-gather of `count` doubles per rank on a device buffer, on the library's communicator."
function comm_all_gather!(comm::Ptr{Cvoid}, buf::Ptr{Cdouble}, count::Integer)
    flag = ccall((:osqp_amd_comm_all_gather, lib), Cc_int, (Ptr{Cvoid}, Ptr{Cdouble}, Cc_int), comm, buf, count)
    flag == 0 || error("Error in all-gather: $(last_error())")
    return nothing
end


end # module
