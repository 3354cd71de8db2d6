# MPFmtHIP.jl -- libmpfmt.so (MI355X) under MotionPlanning.jl's own dispatch.  `include` after `using MotionPlanning`.
# Julia 0.5/0.6 dialect like the reference (type / immutable / Ptr{Void}).  Julia is not installed where this repository is
# built: every ccall below is executed, with exactly these argument widths, by tests/abi_c/abi_caller.c (gcc, -m gpu).
import MotionPlanning: helper_data_structures, inball, is_free_state, is_free_motion, DistanceDataStructure,
                       SweptCollisionChecker, EmptyControlCache, BoxBounds, ImmutableNNC, MetricNN, statevec2mat
const libmpfmt = "libmpfmt"                      # motionplanning.jl_amd/libmpfmt.so on LD_LIBRARY_PATH
lasterr(ctx) = unsafe_string(ccall((:mpfmt_last_error, libmpfmt), Cstring, (Ptr{Void},), ctx))
chk(ctx::Ptr{Void}, rc::Int32) = rc == 0 || error(lasterr(ctx))

# ---- plugin point 1: index build, helper_data_structures(V, ::Euclidean) (geometric.jl:14) -------------------------
immutable HIPDistanceDS{T} <: DistanceDataStructure{T}
    ctx::Ptr{Void}
end
function HIPDistanceDS{S}(V::Vector{S}, device::Integer = 0)
    h = Ref{Ptr{Void}}(C_NULL)
    rc = ccall((:mpfmt_ctx_create, libmpfmt), Int32, (Int32, Ptr{Ptr{Void}}), device, h)
    rc == 0 || error(lasterr(C_NULL))
    X = statevec2mat(V)                          # zero-copy d x N view (primitivetypes.jl:21-24)
    chk(h[], ccall((:mpfmt_upload_samples, libmpfmt), Int32, (Ptr{Void}, Ptr{Float64}, Int64, Int32),
                   h[], X, size(X, 2), size(X, 1)))
    HIPDistanceDS{eltype(S)}(h[])
end
helper_data_structures{S}(V::Vector{S}, M::Euclidean) = (HIPDistanceDS(V), EmptyControlCache())

# ---- plugin point 2: r-disc query, inball (nearneighbors.jl:179-183) and the whole cache (nearneighbors.jl:23-27) --
function inball{S}(V::Vector{S}, dist::Euclidean, DS::HIPDistanceDS, v::Int, r, forwards::Bool = true)
    N = length(V); inds = Vector{Int}(N); ds = Vector{Float64}(N); k = Ref{Int64}(0)
    chk(DS.ctx, ccall((:mpfmt_rdisc_query, libmpfmt), Int32,
                      (Ptr{Void}, Int64, Float64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}),
                      DS.ctx, v, r, inds, ds, N, k))
    SparseVector(N, resize!(inds, k[]), resize!(ds, k[]))      # ascending, self excluded, 1-based
end
function hip_neighbor_graph(DS::HIPDistanceDS, N::Int, r::Float64)
    colptr = Vector{Int}(N + 1); nnz = Ref{Int64}(0)
    chk(DS.ctx, ccall((:mpfmt_rdisc_count, libmpfmt), Int32, (Ptr{Void}, Float64, Ptr{Int64}, Ptr{Int64}),
                      DS.ctx, r, colptr, nnz))
    rowval = Vector{Int}(nnz[]); nzval = Vector{Float64}(nnz[])
    chk(DS.ctx, ccall((:mpfmt_rdisc_fill, libmpfmt), Int32, (Ptr{Void}, Ptr{Int64}, Ptr{Float64}), DS.ctx, rowval, nzval))
    SparseMatrixCSC(N, N, colptr, rowval, nzval)
end

# ---- plugin point 3: collision checker (boxesND.jl:15-28) ------------------------------------------------------------
# After `precompute!` every is_free_motion(V[y], V[x], CC) of the UNMODIFIED fmtstar! (fmt.jl:75) is a lookup: the sample
# index of each end point from a Dict, a binary search for y in column x of the CSC, one bit of the edge BitVector.  States
# that are not samples (adaptive_shortcut, sample_free!) fall through to the reference's own scalar code on CC.boxes --
# identical arithmetic -- so no call ever launches a one-edge kernel.
type HIPBoxes{N,T} <: SweptCollisionChecker
    boxes::Vector{BoxBounds{N,T}}
    count::Int                                   # fmt.jl:12,106 need a mutable count
    ctx::Ptr{Void}
    index::Dict{Any,Int}                         # sample -> 1-based index
    D::SparseMatrixCSC{Float64,Int}              # the r-disc graph the edge bits refer to
    free::BitVector                              # bit e <-> is_free_motion(V[rowval[e]], V[column of e])
end
function HIPBoxes{N,T}(boxes::Vector{BoxBounds{N,T}}, DS::HIPDistanceDS, SS)
    lohi = reinterpret(T, boxes, (2N, length(boxes)))          # [lo(N); hi(N)] per box, zero copy
    chk(DS.ctx, ccall((:mpfmt_upload_boxes, libmpfmt), Int32,
                      (Ptr{Void}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32),
                      DS.ctx, lohi, length(boxes), N, collect(SS.lo), collect(SS.hi), length(SS.lo)))
    HIPBoxes(boxes, 0, DS.ctx, Dict{Any,Int}(), spzeros(0, 0), falses(0))
end
# graph + every edge bit in one go, and the reference's own neighbour cache installed: inball! becomes viewcol (nearneighbors.jl:128)
function precompute!(P::MPProblem, r::Float64)
    CC = P.CC; N = length(P.V)
    CC.D = hip_neighbor_graph(P.V.DS, N, r)
    CC.free = falses(nnz(CC.D))
    chk(CC.ctx, ccall((:mpfmt_graph_edges_free, libmpfmt), Int32, (Ptr{Void}, Ptr{UInt64}), CC.ctx, CC.free.chunks))
    CC.index = Dict{Any,Int}(zip(P.V.V, 1:N))
    P.V = MetricNN(P.V.V, P.V.dist, P.V.init, ImmutableNNC(CC.D, fill(r, N)), P.V.DS, P.V.US)
    P
end
function is_free_motion(v::AbstractVector, w::AbstractVector, CC::HIPBoxes)
    CC.count += 1                                # boxesND.jl:26
    y = get(CC.index, v, 0); x = get(CC.index, w, 0)
    if x > 0 && y > 0
        rng = CC.D.colptr[x]:(CC.D.colptr[x+1]-1)
        e = searchsortedfirst(CC.D.rowval, y, first(rng), last(rng), Base.Order.Forward)
        e <= last(rng) && CC.D.rowval[e] == y && return CC.free[e]
    end
    is_free_motion(v, w, CC.boxes)               # not a graph edge: the reference's scalar predicate (boxesND.jl:52-56)
end
is_free_state(v::AbstractVector, CC::HIPBoxes) = is_free_state(v, CC.boxes)     # boxesND.jl:42-43 (points: sampler, init)

# ---- whole solve on the device: fmtstar! (fmt.jl:3-119), recursion included (include/mpfmt.h, mpfmt_fmtstar_wavefront) ----
immutable FmtResult
    status::Int32; cost::Float64; z::Int64; collision_checks::Int64; path_len::Int64; nnz::Int64
    ms_graph::Float64; ms_sweep::Float64; ms_host_loop::Float64
end
immutable WfInfo
    done::Int32; nz::Int32; nx::Int32; nconn::Int32; ntrip::Int32; iters::Int64; checks::Int64; cmin::Float64
    tot_z::Int64; tot_x::Int64; tot_conn::Int64
end
# band = cost width of a wavefront; single = true reproduces the reference's pop order exactly (one node per step)
function fmtstar_hip!(P::MPProblem, r::Float64; init_idx = 1, checkpts = true, band = 0.25r, single = false)
    DS = P.V.DS; N = length(P.V)
    A = Vector{Int}(N); C = Vector{Float64}(N); path = Vector{Int}(N); res = Ref{FmtResult}(); info = Ref{WfInfo}()
    g = [P.goal.center; P.goal.radius]           # BallGoal (goals.jl:17-21) = MPFMT_GOAL_BALL
    rc = ccall((:mpfmt_fmtstar_wavefront, libmpfmt), Int32,
               (Ptr{Void}, Float64, Int64, Int32, Int32, Ptr{Float64}, Float64, Int32,
                Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{FmtResult}, Ptr{WfInfo}),
               DS.ctx, r, init_idx, checkpts, 1, g, band, single ? 1 : 0, A, C, path, res, info)
    rc == -6 && (warn("Initial state is infeasible!"); P.status = :failed; return Inf)      # fmt.jl:24-29
    chk(DS.ctx, rc)
    P.CC.count = res[].collision_checks
    P.status = res[].status == 1 ? :solved : :failed
    P.solution = MPSolution(P.status, res[].cost, (res[].ms_graph + res[].ms_sweep + res[].ms_host_loop) / 1e3,
                            Dict("collision_checks" => res[].collision_checks, "tree" => A, "cost" => res[].cost,
                                 "path" => path[1:res[].path_len], "planner" => "FMTstar", "r" => r, "num_samples" => N,
                                 "wavefronts" => info[].iters))
    P.status, P.solution.cost, P.solution.elapsed
end

# ---- multi-GPU: ONE Julia thread, G ctxs (SURVEY 8e); the RCCL exchange lives behind the ABI ------------------------
function hip_comm_create!(ctxs::Vector{Ptr{Void}})
    id = Vector{UInt8}(128)
    ccall((:mpfmt_comm_unique_id, libmpfmt), Int32, (Ptr{UInt8},), id) == 0 || error(lasterr(C_NULL))
    ccall((:mpfmt_group_begin, libmpfmt), Int32, ())
    for (g, c) in enumerate(ctxs)
        chk(c, ccall((:mpfmt_comm_create, libmpfmt), Int32, (Ptr{Void}, Int32, Int32, Ptr{UInt8}), c, g - 1, length(ctxs), id))
    end
    ccall((:mpfmt_group_end, libmpfmt), Int32, ())
end
