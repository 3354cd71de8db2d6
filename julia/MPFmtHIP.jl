# MPFmtHIP.jl -- libmpfmt.so (MI355X) under MotionPlanning.jl's own dispatch.  `include` after `using MotionPlanning`.
# Julia 0.5/0.6 dialect like the reference (type / immutable / Ptr{Void}).  Julia is not installed where this repository is
# built: every ccall below is executed, with exactly these argument widths, by tests/abi_c/abi_caller.c and abi_caller2.c (gcc,
# -m gpu; a width that differs from include/mpfmt.h fails their build), and what they return is compared with the oracle.
import MotionPlanning: helper_data_structures, inball, is_free_state, is_free_motion, closest, closeR, sample_free!,
                       DistanceDataStructure, SweptCollisionChecker, EmptyControlCache, BruteDistanceDS, BoxBounds, ImmutableNNC,
                       MetricNN, statevec2mat, LinearQuadratic, ChoppedMetric, ChoppedQuasiMetric, ReedsSheppExact, DubinsExact,
                       SE2State, PointRobot2D, Circle, Polygon, Compound2D, Shape2D
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
    pinned::Vector{Ptr{Void}}                    # page-locked arrays behind D and free (hip_precompute_step!)
end
function HIPBoxes{N,T}(boxes::Vector{BoxBounds{N,T}}, DS::HIPDistanceDS, SS)
    lohi = reinterpret(T, boxes, (2N, length(boxes)))          # [lo(N); hi(N)] per box, zero copy
    chk(DS.ctx, ccall((:mpfmt_upload_boxes, libmpfmt), Int32,
                      (Ptr{Void}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32),
                      DS.ctx, lohi, length(boxes), N, collect(SS.lo), collect(SS.hi), length(SS.lo)))
    HIPBoxes(boxes, 0, DS.ctx, Dict{Any,Int}(), spzeros(0, 0), falses(0), Ptr{Void}[])
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
# The same precompute! without a second build, a second sweep or a pageable copy: ONE mpfmt_graph_step_device (index, half build of the
# graph, edge tests fused into it, mask written by the ordering pass -- the path bench.py times), then ONE mpfmt_graph_export_pinned into
# the ctx's page-locked export arena (2.6 GB at the north star, at link speed).  The arena lives as long as the ctx: page-locking it
# (~0.2 s per GB) is paid by the first graph of a ctx only, every later precompute of the same problem costs the copy alone.  The
# arrays back the SparseMatrixCSC and the BitVector directly (pointer_to_array, own = false) and are overwritten by the next export of
# the ctx; hip_release!(CC) drops the references (the library frees the arena with the ctx).
function hip_precompute_step!(P::MPProblem, r::Float64)
    CC = P.CC; N = length(P.V); ctx = CC.ctx
    nnz = Ref{Int64}(0)
    chk(ctx, ccall((:mpfmt_graph_step_device, libmpfmt), Int32, (Ptr{Void}, Float64, Ptr{Int64}), ctx, r, nnz))
    # the arrays of the previous export live in the ctx's arena, which an export of a LARGER graph frees: drop them before the call
    # (include/mpfmt.h, mpfmt_graph_export_pinned)
    CC.D = spzeros(0, 0); CC.free = falses(0)
    pc = Ref{Ptr{Int64}}(C_NULL); pr = Ref{Ptr{Int64}}(C_NULL); pv = Ref{Ptr{Float64}}(C_NULL); pm = Ref{Ptr{UInt64}}(C_NULL)
    rate = Ref{Float64}(0.0)
    chk(ctx, ccall((:mpfmt_graph_export_pinned, libmpfmt), Int32,
                   (Ptr{Void}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{UInt64}}, Ptr{Int64}, Ptr{Float64}),
                   ctx, pc, pr, pv, pm, nnz, rate))
    words = (nnz[] + 63) >> 6
    colptr = pointer_to_array(pc[], N + 1, false)
    rowval = pointer_to_array(pr[], nnz[], false)
    nzval = pointer_to_array(pv[], nnz[], false)
    CC.D = SparseMatrixCSC(N, N, colptr, rowval, nzval)
    CC.free = falses(0)
    CC.free.chunks = pointer_to_array(pm[], words, false); CC.free.len = nnz[]; CC.free.dims = (nnz[],)
    CC.index = Dict{Any,Int}(zip(P.V.V, 1:N))
    P.V = MetricNN(P.V.V, P.V.dist, P.V.init, ImmutableNNC(CC.D, fill(r, N)), P.V.DS, P.V.US)
    rate[]                                       # GB/s of the export
end
# The same export into page-locked arrays that outlive the ctx (mpfmt_pinned_alloc per array -- page-locking paid per call -- returned
# by hip_release!): for a graph that is kept while the ctx goes on to other sample sets.
function hip_export_owned!(CC::HIPBoxes, N::Int, nnz::Int)
    words = (nnz + 63) >> 6
    ptrs = Ptr{Void}[]
    for bytes in (8(N + 1), 8max(nnz, 1), 8max(nnz, 1), 8max(words, 1))
        p = Ref{Ptr{Void}}(C_NULL)
        ccall((:mpfmt_pinned_alloc, libmpfmt), Int32, (Int64, Ptr{Ptr{Void}}), bytes, p) == 0 || error("mpfmt_pinned_alloc")
        push!(ptrs, p[])
    end
    rate = Ref{Float64}(0.0)
    chk(CC.ctx, ccall((:mpfmt_graph_export, libmpfmt), Int32, (Ptr{Void}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{UInt64}, Ptr{Float64}),
                      CC.ctx, ptrs[1], ptrs[2], ptrs[3], ptrs[4], rate))
    CC.D = SparseMatrixCSC(N, N, pointer_to_array(convert(Ptr{Int64}, ptrs[1]), N + 1, false),
                           pointer_to_array(convert(Ptr{Int64}, ptrs[2]), nnz, false), pointer_to_array(convert(Ptr{Float64}, ptrs[3]), nnz, false))
    CC.free = falses(0)
    CC.free.chunks = pointer_to_array(convert(Ptr{UInt64}, ptrs[4]), words, false); CC.free.len = nnz; CC.free.dims = (nnz,)
    CC.pinned = ptrs
    rate[]
end
function hip_release!(CC::HIPBoxes)
    CC.D = spzeros(0, 0); CC.free = falses(0)
    for p in CC.pinned
        ccall((:mpfmt_pinned_free, libmpfmt), Int32, (Ptr{Void},), p)
    end
    CC.pinned = Ptr{Void}[]
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


# One thread, G ctxs, one step of the eager hot path on every GPU and the free-edge masks of all shards on each of them
# (what `bench.py --gpus G` does with one process per GPU).  Launch everything, then finish everything: no call blocks on a
# peer this thread has yet to launch.  `first` = true on the first step (the thread has seen every shard's nnz: it agrees the
# capacity itself instead of the blocking lengths exchange).  Returns per-ctx (device pointer, stride in words, words, nnz).
function hip_graph_step!(ctxs::Vector{Ptr{Void}}, r::Float64; first::Bool = false)
    G = length(ctxs); nnz = Vector{Int64}(G)
    for c in ctxs
        chk(c, ccall((:mpfmt_graph_step_launch, libmpfmt), Int32, (Ptr{Void}, Float64), c, r))
    end
    for (g, c) in enumerate(ctxs)
        n = Ref{Int64}(0)
        chk(c, ccall((:mpfmt_graph_step_finish, libmpfmt), Int32, (Ptr{Void}, Ptr{Int64}), c, n)); nnz[g] = n[]
    end
    hint = first ? div(maximum(nnz) + 63, 64) + 64 : 0
    ccall((:mpfmt_group_begin, libmpfmt), Int32, ())
    for c in ctxs
        chk(c, ccall((:mpfmt_allgather_free_mask_launch, libmpfmt), Int32, (Ptr{Void}, Int64), c, hint))
    end
    ccall((:mpfmt_group_end, libmpfmt), Int32, ())
    hip_gather_finish!(ctxs)
end
function hip_gather_finish!(ctxs::Vector{Ptr{Void}})
    G = length(ctxs)
    out = Vector{Any}(G)
    while true
        again = false
        for (g, c) in enumerate(ctxs)
            ptr = Ref{Ptr{Void}}(C_NULL); stride = Ref{Int64}(0); words = Vector{Int64}(G); nnzs = Vector{Int64}(G)
            rc = ccall((:mpfmt_allgather_free_mask_finish, libmpfmt), Int32,
                       (Ptr{Void}, Ptr{Ptr{Void}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), c, ptr, stride, words, nnzs)
            rc == 1 ? (again = true) : chk(c, rc)             # MPFMT_RETRY: a shard outgrew the capacity (every ctx reports it)
            out[g] = (ptr[], stride[], words, nnzs)
        end
        again || return out
        ccall((:mpfmt_group_begin, libmpfmt), Int32, ())
        for c in ctxs
            chk(c, ccall((:mpfmt_allgather_free_mask_relaunch, libmpfmt), Int32, (Ptr{Void},), c))
        end
        ccall((:mpfmt_group_end, libmpfmt), Int32, ())
    end
end

# ---- LinearQuadratic (double integrator): helper_data_structures(V, M::LinearQuadratic) (linearquadratic.jl:68-77) ----------
# States are uploaded with mpfmt_upload_samples(ctx, X, N, 2m), obstacles with mpfmt_upload_boxes(ctx, lohi, M, m, lo, hi, 2m)
# (workspace = first m coordinates, OutputMatrix([I 0]), linearquadratic.jl:51-52).
function helper_data_structures{S}(V::Vector{S}, M::LinearQuadratic, DS::HIPDistanceDS)
    N = length(V); colptr = Vector{Int}(N + 1); nnz = Ref{Int64}(0)
    chk(DS.ctx, ccall((:mpfmt_di_graph_count, libmpfmt), Int32, (Ptr{Void}, Float64, Float64, Ptr{Int64}, Ptr{Int64}),
                      DS.ctx, M.bvp.R[1,1], M.cmax, colptr, nnz))
    rowval = Vector{Int}(nnz[]); nzval = Vector{Float64}(nnz[]); tval = Vector{Float64}(nnz[])
    chk(DS.ctx, ccall((:mpfmt_di_graph_fill, libmpfmt), Int32, (Ptr{Void}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
                      DS.ctx, rowval, nzval, tval))
    Dmat = SparseMatrixCSC(N, N, colptr, rowval, nzval)
    BruteDistanceDS(Dmat'), EmptyControlCache(), BruteDistanceDS(Dmat), EmptyControlCache()   # DSF, USF, DSB, USB
end
# every is_free_motion(V[y], V[x], CC, SS) of the steering graph (5 collision waypoints, linearquadratic.jl:85-88) and the
# number of workspace segment tests the reference would have made for it (its CC.count increments)
function hip_di_edges_free(DS::HIPDistanceDS, nnz::Int)
    free = falses(nnz); nseg = Vector{UInt8}(nnz)
    chk(DS.ctx, ccall((:mpfmt_di_graph_edges_free, libmpfmt), Int32, (Ptr{Void}, Ptr{UInt64}, Ptr{UInt8}), DS.ctx, free.chunks, nseg))
    free, nseg
end

# ---- simple cars (simplecars.jl:42-49): chopped Reeds-Shepp metric / Dubins quasi-metric ---------------------------------------
function helper_data_structures{S<:SE2State,R<:ReedsSheppExact}(V::Vector{S}, M::ChoppedMetric{R}, DS::HIPDistanceDS)
    N = length(V); colptr = Vector{Int}(N + 1); nnz = Ref{Int64}(0)
    chk(DS.ctx, ccall((:mpfmt_reedsshepp_graph_count, libmpfmt), Int32, (Ptr{Void}, Float64, Float64, Float64, Ptr{Int64}, Ptr{Int64}),
                      DS.ctx, M.m.r, M.m.s, M.chopval, colptr, nnz))
    rowval = Vector{Int}(nnz[]); nzval = Vector{Float64}(nnz[])
    chk(DS.ctx, ccall((:mpfmt_reedsshepp_graph_fill, libmpfmt), Int32, (Ptr{Void}, Ptr{Int64}, Ptr{Float64}), DS.ctx, rowval, nzval))
    BruteDistanceDS(SparseMatrixCSC(N, N, colptr, rowval, nzval)), EmptyControlCache()      # column v = inball(v), ds = d(v, w)
end
function helper_data_structures{S<:SE2State,R<:DubinsExact}(V::Vector{S}, M::ChoppedQuasiMetric{R}, DS::HIPDistanceDS)
    N = length(V); colptr = Vector{Int}(N + 1); nnz = Ref{Int64}(0)
    chk(DS.ctx, ccall((:mpfmt_dubins_graph_count, libmpfmt), Int32, (Ptr{Void}, Float64, Float64, Float64, Ptr{Int64}, Ptr{Int64}),
                      DS.ctx, M.m.r, M.m.s, M.chopval, colptr, nnz))
    rowval = Vector{Int}(nnz[]); nzval = Vector{Float64}(nnz[])
    chk(DS.ctx, ccall((:mpfmt_dubins_graph_fill, libmpfmt), Int32, (Ptr{Void}, Ptr{Int64}, Ptr{Float64}), DS.ctx, rowval, nzval))
    Dmat = SparseMatrixCSC(N, N, colptr, rowval, nzval)                                      # column x = backward set of x
    BruteDistanceDS(Dmat'), EmptyControlCache(), BruteDistanceDS(Dmat), EmptyControlCache()
end

# ---- closest / closeR (boxesND.jl:33-34,61-86; robots2D.jl:25-26), one call for a batch of query points (columns of P) ----------
function closest(P::Matrix{Float64}, CC::HIPBoxes, W::Matrix{Float64})
    n = size(P, 2); d2 = Vector{Float64}(n); v = similar(P); k = Vector{Int}(n); fails = Ref{Int64}(0)
    chk(CC.ctx, ccall((:mpfmt_closest, libmpfmt), Int32,
                      (Ptr{Void}, Ptr{Float64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}),
                      CC.ctx, P, n, W, d2, v, k, fails))
    fails[] == 0 || error("bvls returned nothing for $(fails[]) (point, box) pair(s)")     # what bvls.jl:67 leads to in the reference
    d2, v
end
function closeR(P::Matrix{Float64}, CC::HIPBoxes, W::Matrix{Float64}, r2::Float64)
    n = size(P, 2); dw = size(P, 1); ptr = Vector{Int}(n + 1); total = Ref{Int64}(0); fails = Ref{Int64}(0)
    cap = n * length(CC.boxes)                   # every (point, obstacle) pair at most
    obstacle = Vector{Int}(cap); d2 = Vector{Float64}(cap); v = Matrix{Float64}(dw, cap)
    chk(CC.ctx, ccall((:mpfmt_closeR, libmpfmt), Int32,
                      (Ptr{Void}, Ptr{Float64}, Int64, Ptr{Float64}, Float64, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Float64}, Ptr{Float64},
                       Ptr{Int64}, Ptr{Int64}),
                      CC.ctx, P, n, W, r2, ptr, cap, obstacle, d2, v, total, fails))
    ptr, obstacle[1:total[]], d2[1:total[]], v[:, 1:total[]]
end

# ---- 2-D SAT world: PointRobot2D(Compound2D(parts)) (robots2D.jl:5-14) -----------------------------------------------------------
# kinds 0 = Circle (cx, cy, r), 1 = Polygon (x1, y1, x2, y2, ...); afterwards every validity entry point of the ctx answers with
# the SAT predicates (SAT2D.jl:119-178) until mpfmt_upload_boxes switches back
function hip_upload_shapes!(ctx::Ptr{Void}, parts::Vector, lo::Vector{Float64}, hi::Vector{Float64})
    kinds = Int32[]; nverts = Int32[]; data = Float64[]
    for s in parts
        if isa(s, Circle)
            push!(kinds, 0); push!(nverts, 0); append!(data, [s.c[1], s.c[2], s.r])
        else
            push!(kinds, 1); push!(nverts, length(s.points)); for p in s.points; append!(data, [p[1], p[2]]); end
        end
    end
    chk(ctx, ccall((:mpfmt_upload_shapes2d, libmpfmt), Int32,
                   (Ptr{Void}, Int32, Ptr{Int32}, Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                   ctx, length(kinds), kinds, nverts, data, lo, hi))
end

# ---- sample_free!(P, N) (sampling.jl:11-45): the rejection loop in device batches (counter-based stream, sequential semantics) ----
function hip_sample_free!(ctx::Ptr{Void}, seed::UInt64, N::Int, init::Vector{Float64}, goal_kind::Int32, goal_params::Vector{Float64},
                          goal_ct::Int32)
    X = Matrix{Float64}(length(init), N); attempts = Ref{Int64}(0)
    chk(ctx, ccall((:mpfmt_sample_free, libmpfmt), Int32,
                   (Ptr{Void}, UInt64, Int64, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Int64}),
                   ctx, seed, N, init, goal_kind, goal_params, goal_ct, X, attempts))
    X, attempts[]
end
