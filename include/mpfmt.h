/*
 * mpfmt.h -- C ABI of libmpfmt.so: the MI355X (gfx950) implementation of the FMT* batch-expand
 * hot path of schmrlng/MotionPlanning.jl (r-disc neighbour graph, segment-vs-AABB sweep,
 * per-edge cost, FMT* expansion).
 *
 * The reference has no FFI: its plugin surface is Julia multiple dispatch on the abstract types
 * SampleSet / DistanceDataStructure / CollisionChecker / StateSpace held in the abstract fields of
 * MPProblem (src/problems.jl:12-30).  Each export below names the reference method it stands behind
 * (paths relative to the reference repository); INTEGRATION.md shows the Julia `ccall` glue.
 *
 * Conventions
 *   - every call returns int32 status: 0 = MPFMT_OK, negative = error; mpfmt_last_error(ctx) has text.
 *     No exception crosses the ABI.
 *   - the caller (Julia) owns every host array passed in; the library owns device memory inside ctx.
 *   - sample / edge INDICES ARE 1-BASED Int64 at this ABI (Julia `Int`), like every index the
 *     reference stores (rowval of SparseMatrixCSC, A = parents, path).  colptr is 1-based too.
 *   - samples: const double* X = d x N column-major, i.e. the memory of Vector{SVector{d,Float64}}
 *     (zero-copy view `statevec2mat`, src/primitivetypes.jl:21-24).
 *   - boxes:   const double* lohi = (2*dw) x M column-major: lo(dw) then hi(dw) per box, the memory of
 *     Vector{BoxBounds{dw,Float64}} (src/collisioncheckers/boxesND.jl:5-13).
 *   - bit masks: uint64 words, bit e of word e>>6, LSB first == BitVector.chunks.
 *   - calls are blocking and must come from one host thread per ctx (the reference is single-threaded).
 *   - arithmetic that decides a mask bit is IEEE binary64, unfused, in the reference's written order;
 *     masks and indices are bit-exact against the CPU oracle, costs within 1e-6 relative.
 */
#ifndef MPFMT_H
#define MPFMT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* every export carries default visibility; the library itself is built with -fvisibility=hidden, so nothing else (C++ helpers,
 * kernel launch stubs) is visible to a process that loads it beside other HIP libraries */
#define MPFMT_API __attribute__((visibility("default")))

#define MPFMT_OK              0
#define MPFMT_ERR_ARG        -1   /* bad argument (null pointer, dimension out of range, index out of range) */
#define MPFMT_ERR_STATE      -2   /* call out of order (e.g. fill before count, sweep before boxes) */
#define MPFMT_ERR_HIP        -3   /* HIP runtime error (text in mpfmt_last_error) */
#define MPFMT_ERR_NODEVICE   -4   /* no gfx950 device visible: the library has NO CPU fallback */
#define MPFMT_ERR_CAPACITY   -5   /* caller buffer too small */
#define MPFMT_ERR_INFEASIBLE -6   /* fmtstar: initial state infeasible (src/planners/fmt.jl:24-29) */
#define MPFMT_RETRY            1   /* not an error: mpfmt_allgather_free_mask_finish on a ctx driven inside mpfmt_group_begin / _end -- a shard
                                      outgrew the agreed capacity; call _relaunch for every ctx (in a group), then _finish again */

#define MPFMT_MAX_DIM 16

typedef struct mpfmt_ctx mpfmt_ctx;

/* goal kinds for mpfmt_fmtstar (src/goals.jl): params = [lo(d),hi(d)] | [center(d),radius] | [pt(d)] */
#define MPFMT_GOAL_RECT  0   /* RectangleGoal   src/goals.jl:8-12,96  */
#define MPFMT_GOAL_BALL  1   /* BallGoal        src/goals.jl:17-21,100 */
#define MPFMT_GOAL_POINT 2   /* PointGoal       src/goals.jl:45,111-114 */

/* ---- context ------------------------------------------------------------------------------------- */

/* device = HIP device ordinal (one ctx per GPU; one process per GPU in multi-GPU runs). */
MPFMT_API int32_t mpfmt_ctx_create(int32_t device, mpfmt_ctx** out);
MPFMT_API int32_t mpfmt_ctx_destroy(mpfmt_ctx* ctx);
MPFMT_API const char* mpfmt_last_error(const mpfmt_ctx* ctx);
/* Version string of the library build ("mpfmt <semver> gfx950"). */
MPFMT_API const char* mpfmt_version(void);
/* Launch on a caller-provided hipStream_t (e.g. torch's current stream); NULL = the ctx's own stream. */
MPFMT_API int32_t mpfmt_set_stream(mpfmt_ctx* ctx, void* hip_stream);
/* Multi-GPU: this ctx owns shard `rank` of `world` (contiguous ranges of the library's cell-sorted
 * sample order).  Graph build / sweep then only produce the columns of the shard.  Default (0,1). */
MPFMT_API int32_t mpfmt_set_shard(mpfmt_ctx* ctx, int32_t rank, int32_t world);

/* ---- index build: helper_data_structures(V, dist) (src/nearneighbors.jl:95-100,
 *      src/statespaces/geometric.jl:14) called by MetricNN(V, dist, init) (src/nearneighbors.jl:70-74)
 *      every time addpoints runs (src/sampling.jl:43). --------------------------------------------- */
/* X: d x N column-major host doubles (Julia's Vector{SVector{d,Float64}} as it lies in memory).  One PCIe copy; the set's bounding
 * box and the finiteness check run on the device beside it (one reduction over the uploaded copy -- a host loop over the coordinates
 * would cost more than the copy).  A non-finite coordinate returns MPFMT_ERR_ARG naming the first such sample (1-based) and leaves
 * the ctx as it was -- the sample set it had, its index, graph and capacity hints (the copy lands in a second buffer that changes
 * places with the ctx's only once the set has been accepted). */
MPFMT_API int32_t mpfmt_upload_samples(mpfmt_ctx* ctx, const double* X, int64_t N, int32_t d);
/* The same for a sample set that already lives in HBM of ctx's device (dX = device pointer, same d x N column-major layout): a batch
 * produced on the device -- the library's sampler (mpfmt_sample_free leaves its set in ctx already), a ROCArray -- becomes the
 * SampleSet of the next index build (addpoints, src/nearneighbors.jl:108-109) without crossing PCIe.  One device-to-device copy;
 * the set's bounding box and the finiteness check run on the device.
 * Ordering: dX is read on ctx's stream.  Whatever produced it must be complete on that stream -- hand the producer's stream to
 * mpfmt_set_stream first, or synchronise it before the call.  On return the copy is complete (dX may be reused).
 * A non-finite coordinate returns MPFMT_ERR_ARG and leaves the ctx as it was, as for mpfmt_upload_samples. */
MPFMT_API int32_t mpfmt_upload_samples_device(mpfmt_ctx* ctx, const double* dX, int64_t N, int32_t d);

/* ---- collision checker: PointRobotNDBoxes(boxes) (src/collisioncheckers/boxesND.jl:15-23) and the
 *      BoundedStateSpace bounds used by in_state_space (src/statespaces.jl:29-34,150).
 *      ss_lo/ss_hi: d_state doubles each, or both NULL (no bounds test).
 *      dw = workspace dimension (Identity s2w: dw == d). ------------------------------------------- */
MPFMT_API int32_t mpfmt_upload_boxes(mpfmt_ctx* ctx, const double* lohi, int32_t M, int32_t dw,
                           const double* ss_lo, const double* ss_hi, int32_t d_state);

/* ---- r-disc neighbour graph = ImmutableNNC(D::SparseMatrixCSC, r) (src/nearneighbors.jl:23-27):
 *      column v = inball(V, dist, DS, v, r) (src/nearneighbors.jl:179-183) for every v.
 *      Two-phase so Julia allocates exact sizes:
 *        count: colptr[N+1] (1-based, colptr[1] == 1), *nnz = colptr[N+1]-1
 *        fill : rowval[nnz] (1-based, strictly ascending inside a column, self excluded), nzval[nnz].
 *      With a shard set, only the shard's columns are non-empty. ------------------------------------ */
MPFMT_API int32_t mpfmt_rdisc_count(mpfmt_ctx* ctx, double r, int64_t* colptr, int64_t* nnz);
MPFMT_API int32_t mpfmt_rdisc_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval);

/* One query = inball(V, dist, DS, v, r, forwards) (src/nearneighbors.jl:179-183), the MutableNNC
 * cache-miss path (src/nearneighbors.jl:129-135).  v 1-based.  *k = neighbour count; at most cap
 * entries are written (MPFMT_ERR_CAPACITY if k > cap, *k still set). */
MPFMT_API int32_t mpfmt_rdisc_query(mpfmt_ctx* ctx, int64_t v, double r, int64_t* inds, double* ds, int64_t cap, int64_t* k);

/* ---- batch validity ------------------------------------------------------------------------------
 * points_free: bit e = is_free_state(V[idx[e]], CC, SS) (src/statespaces.jl:151-152,
 *              src/collisioncheckers/boxesND.jl:42-43); idx NULL = all N samples in order
 *              (the checkpts sweep, src/planners/fmt.jl:31-36).
 * edges_free : bit e = is_free_motion(V[src[e]], V[dst[e]], CC, SS) (src/statespaces.jl:153-158,
 *              src/collisioncheckers/boxesND.jl:26,44-56); src = parent first (src/planners/fmt.jl:75).
 * graph_edges_free: the same over every stored graph entry, CSC order: entry e in column x with
 *              row y  <->  is_free_motion(V[y], V[x]).  mask has ceil(nnz/64) words. */
MPFMT_API int32_t mpfmt_points_free(mpfmt_ctx* ctx, const int64_t* idx, int64_t n, uint64_t* mask);
MPFMT_API int32_t mpfmt_edges_free(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, uint64_t* mask);
MPFMT_API int32_t mpfmt_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask);
/* Same three on explicit points (not sample indices): P = d x n column-major; Q likewise (segment ends).
 * is_free_state(v, CC, SS) / is_free_motion(v, w, CC, SS) for states that are not samples
 * (sampler candidates src/sampling.jl:25, shortcut segments src/postprocessors.jl:6-39). */
MPFMT_API int32_t mpfmt_states_free(mpfmt_ctx* ctx, const double* P, int64_t n, uint64_t* mask);
MPFMT_API int32_t mpfmt_motions_free(mpfmt_ctx* ctx, const double* P, const double* Q, int64_t n, uint64_t* mask);
/* is_free_path(p, CC, SS) = @all [is_free_motion(p[i], p[i+1], CC, SS)] (src/statespaces.jl:159-160) for a path of n states
 * P = d x n column-major: *free_out = 1 / 0; seg_mask (may be NULL) gets the n-1 segment bits.  (The reference's box-list
 * method iterates 1:length(BL)-1 instead of the path length, src/collisioncheckers/boxesND.jl:57 -- a bug that is not
 * reproduced: every segment is tested.)  A path of fewer than 2 states is free. */
MPFMT_API int32_t mpfmt_path_free(mpfmt_ctx* ctx, const double* P, int64_t n, int32_t* free_out, uint64_t* seg_mask);

/* ---- Euclidean per-edge steer, src/statespaces/geometric.jl:18-19, batched over E edges src[e] -> dst[e]
 *      (1-based sample indices):
 *        euclid_steer     : steering_control(M::Euclidean, v, w) = StepControl(evaluate(M, v, w), normalize(w - v)):
 *                           t[e] = |w - v| (the graph's edge cost, bit for bit), u[e][d] = unit direction = inv(|w - v|) * (w - v)
 *                           (a zero-length edge gives t = 0 and NaN directions, as in the reference).
 *        euclid_propagate : propagate(M::Euclidean, v, u::StepControl) = v + u.t * u.u from v = V[src[e]]; with s != NULL the
 *                           partial form of src/statespaces.jl:79-81: s[e] <= 0 -> v, s[e] >= t[e] -> full step, else v + s[e] * u.
 *      collision_waypoints(::Euclidean, v, w) = (v, w) (geometric.jl:20) is what mpfmt_edges_free sweeps. */
MPFMT_API int32_t mpfmt_euclid_steer(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double* t, double* u);
MPFMT_API int32_t mpfmt_euclid_propagate(mpfmt_ctx* ctx, const int64_t* src, int64_t E, const double* t, const double* u, const double* s,
                               double* out);

/* ---- batch expand: the body of the FMT* loop (src/planners/fmt.jl:70-82) for a set of z.
 *      For every unvisited (W) and valid (F, may be NULL) sample x with a forward neighbour in zs:
 *        y_min = first argmin over open (H) backward neighbours y of C[y] + d(y,x)   (fmt.jl:72-74)
 *        free  = is_free_motion(V[y_min], V[x], CC, SS)                              (fmt.jl:75)
 *      Results are reported sorted by x ascending.  W,H,F: N-bit masks; C: N doubles; zs 1-based.
 *      Requires a built graph (mpfmt_rdisc_count).  xs/ymin/cmin/free_out need capacity cap. */
MPFMT_API int32_t mpfmt_expand(mpfmt_ctx* ctx, const uint64_t* W, const uint64_t* H, const uint64_t* F, const double* C,
                     const int64_t* zs, int64_t nz,
                     int64_t* xs, int64_t* ymin, double* cmin, uint8_t* free_out, int64_t cap, int64_t* nx);

/* ---- whole solve: fmtstar!(P, N; r, init_idx, checkpts) with connections = :R
 *      (src/planners/fmt.jl:3-119).  The GPU builds the r-disc graph, the checkpts bitmap and the
 *      per-edge free mask; the sequential dynamic-programming recursion (fmt.jl:68-90) then runs on the
 *      host over those arrays, so tree, costs and path equal the lazy reference loop's.
 *      A (parents, 1-based, 0 = none), C (cost-to-come): N entries.  path: capacity N.
 *      collision_checks counts the edge checks the loop ASKED for (P.CC.count, boxesND.jl:26). */
typedef struct {
    int32_t status;            /* 1 = :solved, 0 = :failed                      (fmt.jl:103) */
    double  cost;              /* C[z]                                          (fmt.jl:107) */
    int64_t z;                 /* last dequeued sample, 1-based */
    int64_t collision_checks;  /* metadata["collision_checks"]                  (fmt.jl:106) */
    int64_t path_len;          /* metadata["path"] length                       (fmt.jl:92-101) */
    int64_t nnz;               /* directed edges in the r-disc graph */
    double  ms_graph;          /* device time: r-disc graph build */
    double  ms_sweep;          /* device time: point + edge validity sweeps */
    double  ms_host_loop;      /* host time: sequential FMT* recursion */
} mpfmt_fmt_result;

MPFMT_API int32_t mpfmt_fmtstar(mpfmt_ctx* ctx, double r, int64_t init_idx, int32_t checkpts,
                      int32_t goal_kind, const double* goal_params,
                      int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);

/* The sequential part of fmtstar! on its own (src/planners/fmt.jl:43-101): host code, no ctx and no device.  Input is
 * the finished r-disc graph in the device-native format -- colptr[N+1] 0-based offsets, rowval[nnz] 0-based int32 rows
 * (ascending per column), nzval[nnz] distances, efree = one bit per entry (row -> column motion free), F = checkpts
 * bitmap over samples (NULL: checkpts = false), ss_lo/ss_hi = state-space bounds (both NULL: none).  init_idx is 1-based;
 * A / path are 1-based like mpfmt_fmtstar's; res gets status, cost, z, collision_checks, path_len, nnz.
 * mpfmt_fmtstar = graph_build_device + graph_sweep_device + this. */
MPFMT_API int32_t mpfmt_host_fmt_recursion(int64_t N, int32_t d, const double* X, const int64_t* colptr, const int32_t* rowval,
                                 const double* nzval, const uint64_t* efree, const uint64_t* F, const double* ss_lo,
                                 const double* ss_hi, int64_t init_idx, int32_t goal_kind, const double* goal_params,
                                 int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);

/* ---- Monte-Carlo collision probability of candidate edges (BASELINE.json configs[4]).  The reference
 *      has no implementation (README.md:9-10 cites the papers only); the workload is the one SURVEY 8d defines: per edge
 *      (src[e] -> dst[e], 1-based sample indices) `rollouts` perturbed copies of the 2-point trajectory, each put through
 *      is_free_motion(v', w', CC, SS) (statespaces.jl:153-158, boxesND.jl:44-56).  v' = v + sigma z with z a standardised
 *      Irwin-Hall(8) variate built from the halfwords of Philox4x32-10(key = seed, counter = (rollout, edge, coordinate, 2))
 *      -- integer sums and unfused fp64 only, so a scalar loop reproduces hits[e] (colliding rollouts) exactly.
 *      Needs the AABB checker; E and rollouts < 2^32. */
MPFMT_API int32_t mpfmt_mc_edges_collision(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double sigma, int64_t rollouts,
                                 uint64_t seed, int64_t* hits);
/*      Importance-sampling estimator of the same probability (the approach of the papers README.md:9-10 cites): rollouts are drawn from
 *      a mixture -- half of them from the nominal noise, the rest from the noise shifted (both end points alike, at most 3 sigma per
 *      coordinate) towards the closest points of the up to three obstacles nearest to the nominal segment (closest(p, BB, I) of
 *      boxesND.jl:61-86 with W = I -- a clamp -- at five points of the segment); a colliding rollout counts with weight
 *      f(y) / (0.5 f(y) + sum_j (0.5 / K) f(y - s_j)), f the product of Irwin-Hall(8) densities.  Weights are quantised to 2^-40 and
 *      summed as integers: estimate = wsum[e] / (rollouts * 2^40), which a scalar loop reproduces exactly (the arithmetic is spelled out
 *      at k_mc_is_edges, csrc/kernels_sweep.hip).  rollouts < 2^22; the whole obstacle set must fit one LDS stage (M <= 256 at d <= 8). */
MPFMT_API int32_t mpfmt_mc_edges_collision_is(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double sigma, int64_t rollouts,
                                    uint64_t seed, uint64_t* wsum);
/*      The ADAPTIVE estimator (BASELINE configs[4]: "adaptive-importance-sampling"): per edge a pilot of 4096 rollouts with the noise
 *      inflated by 1.625 finds colliding perturbations; their likelihood-ratio-weighted mean -- the mean of the nominal noise GIVEN a
 *      collision, one cross-entropy update of the proposal's mean in all 2 d noise coordinates -- becomes the shift of a two-component
 *      mixture (half nominal, half shifted); weights f(y) / (0.5 f(y) + 0.5 f(y - mu)) as above.  An edge whose pilot sees no collision
 *      is estimated by plain Monte Carlo (mu = 0, every weight 1).  shifts (may be NULL): E x 2 d doubles, the mu of every edge in
 *      noise units.  Spelled out: pilot rollout k, coordinate c < 2 d: Irwin-Hall integer S from Philox(key = seed, counter = (k, e,
 *      64 + c, 2)), Z = S - 262140, z = Z / 53509.92, y = 1.625 z; a colliding rollout's likelihood ratio lr = prod g(x(y_c)) /
 *      prod g(x(z_c)) (g, x as above), Wq = (uint64)(lr 2^30); SW = sum Wq, A_c = sum Wq Z_c over the hits; mu_c = clip(1.625 (A_c /
 *      53509.92) / SW, -3, 3).  Main rollouts: noise as in mpfmt_mc_edges_collision, shifted by mu when word 0 of Philox(counter = (k,
 *      e, 2 d, 3)) is odd.  Integer sums throughout: a scalar loop reproduces wsum and shifts exactly; same limits as above. */
MPFMT_API int32_t mpfmt_mc_edges_collision_ais(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double sigma, int64_t rollouts,
                                     uint64_t seed, uint64_t* wsum, double* shifts);

/* ---- Dubins car: DubinsQuasiMetricSpace(r_turn, s, lo, hi) of src/statespaces/simplecars.jl:32-38.
 *      Samples are SE2 states (x, y, theta) (upload_samples with d = 3); obstacles live in the workspace (x, y)
 *      (VectorView(1:2)): upload_boxes with dw = 2 and the 3 state bounds (lo_x, lo_y, 0; hi_x, hi_y, 2pi).
 *      dubins_graph_count/fill : the chopped backward sets (nearneighbors.jl:185-198) as a sparse cost matrix in CSC,
 *             column j = sources i (ascending, 1-based) with |xy_i - xy_j| <= r and dubins(i -> j) <= r
 *             (simplecars.jl:106-215), nzval = cost; the forward sets are its transpose.
 *      dubins_graph_edges_free : entry e (row y -> column x): is_free_motion(V[y], V[x], CC, SS) over the reference's
 *             collision waypoints (arcs every pi/12, :68-83; statespaces.jl:127-135,153-158); nseg[e] = workspace
 *             segment tests the reference would have counted (may be NULL).
 *      dubins_steer : batch steer on explicit pairs; controls[i][3][3] = (duration, speed, signed curvature) per segment.
 *      dubins_fmtstar : fmtstar! in this space (forward / backward neighbour sets like the double integrator's).
 *      sin / cos / atan2 / acos are the library's own (csrc/mp_math.h: fixed reductions and polynomials from + - * / sqrt), the
 *      same header the CPU oracle compiles: graphs, masks and costs are bit-identical to the oracle's; against a libm-based
 *      host (Julia) costs agree to ~1e-15 relative. */
MPFMT_API int32_t mpfmt_dubins_graph_count(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t* colptr, int64_t* nnz);
MPFMT_API int32_t mpfmt_dubins_graph_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval);
MPFMT_API int32_t mpfmt_dubins_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask, uint8_t* nseg);
MPFMT_API int32_t mpfmt_dubins_steer(mpfmt_ctx* ctx, const double* X0, const double* X1, int64_t n, double turn_radius, double speed,
                           double* cost, double* controls);
MPFMT_API int32_t mpfmt_dubins_fmtstar(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                             int32_t goal_kind, const double* goal_params, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);

/* ---- Reeds-Shepp car: ReedsSheppMetricSpace(r_turn, s, lo, hi) of src/statespaces/simplecars.jl:29-34 --
 *      the car that may reverse; a symmetric (chopped) metric, so forward and backward neighbour sets coincide
 *      (nearneighbors.jl:200-203).  Same world set-up as the Dubins car above.
 *      reedsshepp_graph_count/fill : column v = { w : |xy_w - xy_v| <= r and reedsshepp(v, w) <= r } (simplecars.jl:266-364
 *             with the word families :367-553), rows ascending 1-based, nzval = reedsshepp(v, w) -- evaluated in that
 *             argument order: the two directions can differ in the last bits, and the reference's inball(v) holds d(v, w).
 *      reedsshepp_graph_edges_free : entry e (row w of column v): is_free_motion(V[w], V[v], CC, SS) -- the motion
 *             fmtstar! tests when it connects v to the open parent w (fmt.jl:75) -- over the waypoints of
 *             steering_control(V[w], V[v]) (:68, :70-82); nseg[e] as above.
 *      reedsshepp_steer : batch steer; controls[i][5][3] = (duration, speed, signed curvature), segments >= nsegs[i]
 *             are zero (nsegs may be NULL).
 *      reedsshepp_fmtstar : fmtstar! in this space (the symmetric recursion of fmt.jl:36-100); workspace goals act on
 *             (x, y), MPFMT_GOAL_POINT takes a whole state (3 doubles). */
MPFMT_API int32_t mpfmt_reedsshepp_graph_count(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t* colptr, int64_t* nnz);
MPFMT_API int32_t mpfmt_reedsshepp_graph_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval);
MPFMT_API int32_t mpfmt_reedsshepp_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask, uint8_t* nseg);
MPFMT_API int32_t mpfmt_reedsshepp_steer(mpfmt_ctx* ctx, const double* X0, const double* X1, int64_t n, double turn_radius, double speed,
                               double* cost, double* controls, int32_t* nsegs);
MPFMT_API int32_t mpfmt_reedsshepp_fmtstar(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                 int32_t goal_kind, const double* goal_params, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);

/* ---- Closest obstacle points in a Mahalanobis metric: closest(p, CC, W) / closeR(p, CC, W, r2) of
 *      src/collisioncheckers/boxesND.jl:33-34,61-86 (boxes, through bvls of src/collisioncheckers/bvls.jl:19-218) and
 *      src/collisioncheckers/robots2D.jl:25-26 + SAT2D.jl:208-285 (circles / convex polygons / compounds), batched over n
 *      query points P [n][dw] against the ctx's current obstacle set (mpfmt_upload_boxes or mpfmt_upload_shapes2d).
 *      W [dw][dw] symmetric positive definite; W = NULL selects the Euclidean SAT2D methods (:208-211, :239, :260-269) and
 *      is an error for boxes (the reference has no such method).
 *      mpfmt_closest : d2min[i], vmin[i][dw] and kmin[i] (1-based obstacle, 0 = none) = the minimum over the obstacles
 *             with strict < (first minimum wins); no obstacle: (Inf, p) for boxes, (Inf, 0) for shapes, as the reference.
 *      mpfmt_closeR  : per point the obstacles with d2 < r2 in ascending d2 (ties in obstacle order: the reference's
 *             sort! is stable): ptr [n+1] 1-based offsets (always written), then obstacle / d2 / v [*total][dw];
 *             MPFMT_ERR_CAPACITY with *total set when *total > cap (nothing else written).
 *      *failures (may be NULL) counts the (point, box) pairs on which the reference's bvls exhausts its 10n iterations and
 *             returns `nothing` (closest() then throws a MethodError) -- the same pairs are detected here, reported and
 *             left out of the minimum / the lists; for circles, pairs whose multiplier iteration (SAT2D.jl:222-235,
 *             unbounded in the reference) has not ended after 200 Newton steps.
 *      Values agree with the reference's LAPACK route (QR least squares, eigfact) to rounding, not bit for bit. */
MPFMT_API int32_t mpfmt_closest(mpfmt_ctx* ctx, const double* P, int64_t n, const double* W, double* d2min, double* vmin, int64_t* kmin,
                      int64_t* failures);
MPFMT_API int32_t mpfmt_closeR(mpfmt_ctx* ctx, const double* P, int64_t n, const double* W, double r2, int64_t* ptr, int64_t cap,
                     int64_t* obstacle, double* d2, double* v, int64_t* total, int64_t* failures);

/* ---- 2-D SAT world: PointRobot2D(Compound2D(parts)) of src/collisioncheckers/robots2D.jl:12-14 and
 *      SAT2D.jl -- parts are Circle(c, r) (:14-28) and convex Polygon(points) (:32-58; Box2D = 4-point polygon, :59-62).
 *      Switches the ctx's collision checker: afterwards mpfmt_points_free / _states_free = is_free_state (point vs
 *      shapes, SAT2D.jl:121-133), mpfmt_edges_free / _motions_free / _graph_edges_free / graph sweep / expand / fmtstar =
 *      is_free_motion = !colliding(Line(v, w), obstacles) (SAT2D.jl:154-178), both wrapped in the state-space checks of
 *      statespaces.jl:150-158.  mpfmt_upload_boxes switches back.  Samples must be 2-D.
 *      kinds[i]: 0 = circle, data = (cx, cy, r); 1 = polygon with nverts[i] (3..16) vertices, data = (x1, y1, x2, y2, ...);
 *      data is the concatenation in shape order (nverts is ignored for circles).  The library runs the reference's
 *      constructors (orientation, unit normals, extrema per normal, AABBs; non-convex polygons are refused like :49).
 *      Nested Compound2D parts are not represented: flatten them (every basic test starts with its own AABB check). */
#define MPFMT_SHAPE_CIRCLE  0
#define MPFMT_SHAPE_POLYGON 1
MPFMT_API int32_t mpfmt_upload_shapes2d(mpfmt_ctx* ctx, int32_t n_shapes, const int32_t* kinds, const int32_t* nverts, const double* data,
                              const double* ss_lo, const double* ss_hi);
/* The 2-D SAT world under a steering space (the notebook's double-integrator and Dubins examples, docs/MotionPlanning.ipynb cells 7-11:
 * PointRobot2D with DoubleIntegrator(2) / DubinsQuasiMetricSpace): upload the shapes (workspace bounds), then the state-space
 * bounds of the steering space (4 for the double integrator, 3 for SE2; src/statespaces.jl:29-34, in_state_space :150).  The
 * steering sweeps (mpfmt_di_graph_edges_free, mpfmt_dubins_ / mpfmt_reedsshepp_graph_edges_free, the *_fmtstar calls) then test
 * their workspace segments with is_free_motion(v, w, CC::PointRobot2D) (robots2D.jl:13-14). */
MPFMT_API int32_t mpfmt_set_state_bounds(mpfmt_ctx* ctx, const double* ss_lo, const double* ss_hi, int32_t d_state);

/* ---- graph persistence: install a graph exported earlier by mpfmt_rdisc_count / mpfmt_rdisc_fill (same
 *      1-based CSC: colptr[N+1], rowval[nnz] strictly ascending per column, nzval[nnz]) for the samples now uploaded --
 *      the reference's ImmutableNNC(D, r) (src/nearneighbors.jl:23-28; its saveNN / loadNN! are commented out, :114-116).
 *      Afterwards mpfmt_graph_edges_free / mpfmt_graph_sweep_device / mpfmt_expand / mpfmt_fmtstar(r) use it without
 *      running the pair phase (mpfmt_fmtstar reuses any filled graph of the same radius).  The arrays are validated
 *      (monotone colptr, rows in range, ascending, no self loops); distances are taken as given. */
MPFMT_API int32_t mpfmt_graph_import(mpfmt_ctx* ctx, double r, const int64_t* colptr, const int64_t* rowval, const double* nzval);

/* ---- batch free-space sampler: sample_free!(P, N, true; ensure_goal_ct) of src/sampling.jl:11-45 with
 *      the rejection loop (sample_space, statespaces.jl:40; is_free_state, statespaces.jl:151-152) run in batches on the
 *      device.  Needs mpfmt_upload_boxes with state-space bounds (the BoundedStateSpace lo / hi) and an Identity
 *      state2workspace (d_state == dw).  The reference draws from Julia's unseeded global RNG; here candidates come from
 *      a counter-based stream (Philox4x32-10, key = seed, counter = candidate index), tested IN ORDER like the reference's
 *      loop, so the result is a pure function of (seed, N, bounds, obstacles, goal) -- independent of batching and
 *      reproducible by a scalar loop.  V[1] = init when init != NULL (sampling.jl:15-20); the last min(goal_ct, N-1)
 *      samples are free goal samples, V[N+1-i] = the i-th one (sampling.jl:37-41; sample_goal, goals.jl:97,101-108,115).
 *      The set is left uploaded in ctx (as by mpfmt_upload_samples); X_out (N*d, may be NULL) receives a copy;
 *      attempts = sample_space candidates the sequential loop would have consumed.
 *      sample_free_biased adds the goal_bias keyword (sampling.jl:11,28-30): each accepted sample is replaced by a free goal
 *      sample with probability goal_bias -- the decision for the k-th accepted sample is the uniform (seed, k) of a third
 *      stream, the replacements take free goal samples in stream order, before the ensure_goal tail does. */
MPFMT_API int32_t mpfmt_sample_free(mpfmt_ctx* ctx, uint64_t seed, int64_t N, const double* init, int32_t goal_kind,
                          const double* goal_params, int32_t goal_ct, double* X_out, int64_t* attempts);
MPFMT_API int32_t mpfmt_sample_free_biased(mpfmt_ctx* ctx, uint64_t seed, int64_t N, const double* init, int32_t goal_kind,
                                 const double* goal_params, int32_t goal_ct, double goal_bias, double* X_out, int64_t* attempts);

/* ---- double-integrator (LinearQuadratic quasi-metric) space: DoubleIntegrator(m; vmax, r=rho)
 *      (src/statespaces/linearquadratic.jl:46-53).  Samples are states (p, v) in R^{2m} (upload_samples with
 *      d = 2m); obstacles live in the workspace (first m coordinates, OutputMatrix C = [I 0], :51-52), so
 *      upload_boxes takes dw = m and the 2m state-space bounds (lo, -vmax..; hi, +vmax..).
 *      di_graph_count/fill : helper_data_structures(V, M::LinearQuadratic) = all-pairs steer_pairwise
 *             (:68-77,196-225): sparse cost matrix Dmat, entry (i -> j) kept when cost(i -> j) <= r.
 *             CSC: column j lists the sources i (ascending, 1-based) = DSB (backward sets, nearB);
 *             DSF (forward sets) is its transpose.  nzval = cost, tval = optimal time t* (the duration of the
 *             DurationAndTargetControl of :222-224; pass NULL to skip).
 *      di_graph_edges_free : entry e (row y -> column x): is_free_motion(V[y], V[x], CC, SS) with the 5
 *             collision waypoints x(v, w, t*, s), s = linspace(0, t*, 5) (:85-88; src/statespaces.jl:153-158).
 *             nseg[e] (may be NULL) = workspace segment tests the reference would have made for that edge
 *             (its CC.count increments, boxesND.jl:26), so collision_checks can be reproduced.
 *      di_steer : batch steer(L, x0, x1, r) -> (cost, t*) (:191-195) on explicit pairs, X0/X1 = 2m x n col-major.
 *      di_fmtstar : fmtstar! (src/planners/fmt.jl:3-119) over that graph; POINT goal = StateGoal (exact state,
 *             src/goals.jl:128-131), RECT/BALL act on the workspace coordinates. */
MPFMT_API int32_t mpfmt_di_graph_count(mpfmt_ctx* ctx, double rho, double r, int64_t* colptr, int64_t* nnz);
MPFMT_API int32_t mpfmt_di_graph_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval, double* tval);
MPFMT_API int32_t mpfmt_di_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask, uint8_t* nseg);
MPFMT_API int32_t mpfmt_di_steer(mpfmt_ctx* ctx, const double* X0, const double* X1, int64_t n, int32_t m, double rho, double r,
                       double* cost, double* topt);
/* The same graph and edge bits with every output left in HBM (the device-resident form of helper_data_structures,
 * linearquadratic.jl:68-77,196-225, + the per-edge is_free_motion of src/statespaces.jl:153-158): count, fill and -- when an obstacle
 * set is uploaded -- the 5-waypoint sweep, one host synchronisation at the end.  Pointers (device addresses, valid until the next
 * build on the ctx): colptr int64[N+1] 0-based, rowval int32[nnz] 0-based ascending per column, nzval / tval double[nnz],
 * free_mask uint64[ceil(nnz/64)] and nseg uint8[nnz] (NULL without a sweep). */
MPFMT_API int32_t mpfmt_di_graph_step_device(mpfmt_ctx* ctx, double rho, double r, int64_t* nnz);
MPFMT_API int32_t mpfmt_di_graph_device_ptrs(mpfmt_ctx* ctx, void** colptr, void** rowval, void** nzval, void** tval, void** free_mask, void** nseg);
MPFMT_API int32_t mpfmt_di_fmtstar(mpfmt_ctx* ctx, double rho, double r, int64_t init_idx, int32_t checkpts,
                         int32_t goal_kind, const double* goal_params,
                         int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);


/* ---- device-resident forms (bench / multi-GPU: no PCIe in the timed region) -----------------------
 * graph_build_device: count+fill on the device only; outputs stay in HBM.  Returns nnz.
 * graph_sweep_device: per-edge free mask of the resident graph into HBM.
 * Pointers to the resident arrays (device addresses, valid until the next build / ctx destroy):
 *   colptr int64[N+1] 0-based offsets, rowval int32[nnz] 0-based, nzval double[nnz], free uint64[ceil(nnz/64)]. */
MPFMT_API int32_t mpfmt_graph_build_device(mpfmt_ctx* ctx, double r, int64_t* nnz);
MPFMT_API int32_t mpfmt_graph_sweep_device(mpfmt_ctx* ctx);
/* graph_step_device: graph_build_device + graph_sweep_device as ONE call with one host synchronisation -- the planner's
 * whole step (fmt.jl:70-75's neighbour sets and edge checks for every sample).  The two-call form needs the host between
 * its kernels (nnz sizes the CSC and the mask).  When the previous step of the same (N, r, shard) took the single-pass
 * path, this call trusts its sizes, issues every kernel back to back, lets a device flag void the kernels after a capacity
 * that did not hold, validates after the synchronisation and transparently redoes the step the careful way if needed.
 * Results are identical to the two-call form and complete in HBM when the call returns (graph_sweep_device, by contrast,
 * only enqueues its kernel on the ctx's stream). */
MPFMT_API int32_t mpfmt_graph_step_device(mpfmt_ctx* ctx, double r, int64_t* nnz);
/* The same step in two calls, for ONE host thread that drives several ctxs (a Julia process holding one ctx per GPU, SURVEY
 * 8e): _launch issues the step's kernels and returns without waiting when the previous step's sizes can be trusted (otherwise
 * it runs the careful form to completion), _finish makes the one synchronisation, validates and repairs.  graph_step_device
 * = _launch + _finish. */
MPFMT_API int32_t mpfmt_graph_step_launch(mpfmt_ctx* ctx, double r);
MPFMT_API int32_t mpfmt_graph_step_finish(mpfmt_ctx* ctx, int64_t* nnz);
MPFMT_API int32_t mpfmt_graph_device_ptrs(mpfmt_ctx* ctx, void** colptr, void** rowval, void** nzval, void** free_mask);
/* The graph and mask a step left resident, to the host in the format of mpfmt_rdisc_count / _fill / mpfmt_graph_edges_free -- colptr[N+1]
 * and rowval[nnz] 1-based Int64, nzval[nnz], mask[(nnz+63)/64] BitVector chunks (mask may be NULL) -- WITHOUT rebuilding or re-sweeping:
 * the drop-in precompute! of julia/MPFmtHIP.jl is mpfmt_graph_step_device + this call, after which the unmodified fmtstar!
 * (src/planners/fmt.jl:68-90) reads ImmutableNNC columns (src/nearneighbors.jl:23-27,128) and answers is_free_motion from the mask.
 * Index conversion runs on the device ahead of two copy streams; *gb_per_s (may be NULL) reports the rate.  The copies run at link
 * speed when the destinations are page-locked: mpfmt_pinned_alloc hands out such memory (Julia wraps it with unsafe_wrap /
 * pointer_to_array, own = false, and returns it with mpfmt_pinned_free). */
MPFMT_API int32_t mpfmt_graph_export(mpfmt_ctx* ctx, int64_t* colptr, int64_t* rowval, double* nzval, uint64_t* mask, double* gb_per_s);
MPFMT_API int32_t mpfmt_pinned_alloc(int64_t bytes, void** out);
MPFMT_API int32_t mpfmt_pinned_free(void* p);
/* Page-locking is the expensive part of a page-locked export (hipHostMalloc: ~0.2 s per GB, several times the copy it speeds up), so the
 * ctx keeps ONE page-locked arena for it: grow-only, valid until it has to grow or the ctx is destroyed (mpfmt_ctx_destroy frees it).
 * mpfmt_export_arena returns at least `bytes` of it.  mpfmt_graph_export_pinned is mpfmt_graph_export into that arena: it returns the
 * four arrays (64-byte aligned; *mask may be asked for or NULL; *nnz_out their length) -- the memory behind the SparseMatrixCSC of
 * ImmutableNNC(D, r) (src/nearneighbors.jl:23-28) and the BitVector of free edges in julia/MPFmtHIP.jl hip_precompute_step!, which
 * every later call of the same ctx reuses: the second graph of a problem costs the copy alone.  The arrays are overwritten by the
 * next export of this ctx -- and FREED by one that needs a larger arena (a graph with more entries): pointers of an earlier export
 * must be dropped before the call (julia/MPFmtHIP.jl re-wraps them after every export).  A call that fails on the ctx's state (no
 * resident graph, mask asked for but not swept) leaves the arena and the earlier export untouched. */
MPFMT_API int32_t mpfmt_export_arena(mpfmt_ctx* ctx, int64_t bytes, void** out);
MPFMT_API int32_t mpfmt_graph_export_pinned(mpfmt_ctx* ctx, int64_t** colptr, int64_t** rowval, double** nzval, uint64_t** mask,
                                            int64_t* nnz_out, double* gb_per_s);

/* ---- streaming r-disc: per-column reductions WITHOUT the stored graph.  BASELINE configs[2] at the radius of src/planners/fmt.jl:39
 *      (R^12, N = 1e6: ~4 700 neighbours per sample, 57 GB of CSC) cannot keep ImmutableNNC resident; what the loop body
 *      src/planners/fmt.jl:70-82 needs of column x is a reduction over inball(x) (src/nearneighbors.jl:179-183):
 *        deg[x]      = |inball(x)|                                        (PRM*-style degree; *nnz = their sum)
 *        parent[x]   = argmin over y in inball(x), y in H, of C[y] + d(y, x), FIRST minimum in ascending y (findmin, fmt.jl:73),
 *                      1-based, 0 when no open neighbour; cost[x] = that minimum (Inf when none) -- C = NULL skips both;
 *                      H = NULL: every sample is open.  The caller then makes the ONE lazy edge test of fmt.jl:75 (mpfmt_edges_free).
 *        free_deg[x] = number of y in inball(x) with is_free_motion(V[y], V[x]) (src/statespaces.jl:153-158) when want_free != 0
 *      C: N doubles by sample index; H: BitVector chunks over the samples.  Outputs: N entries each, caller-owned.  Needs an
 *      unsharded ctx, d <= 12 and a radius the fp16 filter can take (the MFMA pair kernel does the work); membership and costs are
 *      the canonical fp64 arithmetic of every other entry point. ---------------------------------------------------------------- */
MPFMT_API int32_t mpfmt_rdisc_stream(mpfmt_ctx* ctx, double r, const double* C, const uint64_t* H, int32_t want_free,
                                     int64_t* deg, int64_t* free_deg, int64_t* parent, double* cost, int64_t* nnz);
/* Shard bookkeeping for the all-gather: column range (in the library's sorted order) and the number
 * of edges this shard produced. */
MPFMT_API int32_t mpfmt_shard_info(mpfmt_ctx* ctx, int64_t* col_begin, int64_t* col_end, int64_t* shard_nnz);

/* ---- wavefront solve: fmtstar! (src/planners/fmt.jl:3-119) with the dynamic-programming recursion ON THE DEVICE.
 *      mpfmt_fmtstar runs the loop fmt.jl:68-90 on one host core over GPU-built arrays; here W / H / C / A live in HBM and
 *      one step expands a whole cost band of open nodes Z = { z in H : C[z] <= min_H C + band } with the loop body
 *      fmt.jl:70-82 evaluated for every z of the batch against the same (W, H, C), the deferred H update fmt.jl:83-84 per
 *      batch, and the stop test fmt.jl:68 on the batch (answer = goal node of lowest (cost, index)).  Edge checks are lazy
 *      like the reference's (is_free_motion only for the y_min of an examined x, fmt.jl:75); collision_checks counts them.
 *        flags & MPFMT_WF_SINGLE : the batch is the ONE lowest (cost, index) open node: every step is one iteration of the
 *              reference loop -- tree, costs, path, collision_checks identical to mpfmt_fmtstar / the sequential recursion.
 *        band > 0 : every step equals mpfmt_expand (the batch form of the loop body) on the same sets; the tree is a valid
 *              FMT* tree of the same graph whose cost can exceed the sequential one slightly (reported by the tests / bench).
 *        flags & MPFMT_WF_EAGER  : answer edge tests from the swept graph mask (mpfmt_graph_step_device) instead of testing
 *              lazily; forced for checkers without a lane-per-obstacle form here (2-D SAT world, non-identity workspace).
 *              Without the flag the mask is still USED when a step left one for this graph and obstacle set (an edge's bit is a
 *              pure function of the edge: tree, costs and collision_checks are the lazy loop's) -- never computed for the solve.
 *        flags & MPFMT_WF_LAZY   : test every asked-for edge against the obstacle set even when a mask is resident (measurements).
 *      mpfmt_fmtstar_wavefront = begin + steps until done + finish.  A / C / path may be NULL (skips the 12 N-byte copy).
 *      Step-wise form (tests, drivers that interleave other work): wf_begin, wf_step ... until info.done, wf_finish;
 *      wf_state copies the sets out (W, H as the next step will see them; A 1-based, 0 = none), wf_batch the z of the
 *      last step (1-based).
 *      Sharded ctx (mpfmt_set_shard / mpfmt_comm_create): every rank holds the whole (W, H, C, A) and the columns
 *      of its own samples; a step examines the rank's own unvisited samples and ONE all-gather per wavefront carries every
 *      rank's (x, y_min, c_min) connections (counts ride in the slot headers).  Without a communicator the same exchange is
 *      made by the caller: wf_step on every ctx, wf_triples from each, wf_commit of all of them to each. */
#define MPFMT_WF_SINGLE 1
#define MPFMT_WF_EAGER  2
#define MPFMT_WF_LAZY   4
typedef struct {
    int32_t done;              /* 0 running, 1 goal reached (:solved), 2 open set exhausted (:failed) */
    int32_t nz;                /* batch size of the last step */
    int32_t nx;                /* samples examined (fmt.jl:70-74) */
    int32_t nconn;             /* samples connected (fmt.jl:76-80), all ranks */
    int32_t ntrip;             /* connections found by this rank (sharded) */
    int64_t iters;             /* steps so far */
    int64_t checks;            /* edge checks so far (this rank) */
    double  cmin;              /* lowest open cost at the last step */
    int64_t tot_z, tot_x, tot_conn;   /* sums over the steps */
} mpfmt_wf_info;
MPFMT_API int32_t mpfmt_fmtstar_wavefront(mpfmt_ctx* ctx, double r, int64_t init_idx, int32_t checkpts, int32_t goal_kind, const double* goal_params,
                                double band, int32_t flags, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info);
MPFMT_API int32_t mpfmt_wf_begin(mpfmt_ctx* ctx, double r, int64_t init_idx, int32_t checkpts, int32_t goal_kind, const double* goal_params,
                       double band, int32_t flags);
MPFMT_API int32_t mpfmt_wf_step(mpfmt_ctx* ctx, mpfmt_wf_info* info);
MPFMT_API int32_t mpfmt_wf_state(mpfmt_ctx* ctx, uint64_t* W, uint64_t* H, double* C, int64_t* A);
MPFMT_API int32_t mpfmt_wf_batch(mpfmt_ctx* ctx, int64_t* zs, int64_t cap, int64_t* nz);
MPFMT_API int32_t mpfmt_wf_triples(mpfmt_ctx* ctx, int64_t cap, int64_t* x, int64_t* y, double* c, int64_t* n);
MPFMT_API int32_t mpfmt_wf_commit(mpfmt_ctx* ctx, int64_t n, const int64_t* x, const int64_t* y, const double* c);
MPFMT_API int32_t mpfmt_wf_finish(mpfmt_ctx* ctx, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);
/* mpfmt_dubins_fmtstar_wavefront / mpfmt_reedsshepp_fmtstar_wavefront : the car planners (simplecars.jl spaces) the same way. */
MPFMT_API int32_t mpfmt_dubins_fmtstar_wavefront(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                       int32_t goal_kind, const double* goal_params, double band, int32_t flags, int64_t* A, double* C,
                                       int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info);
MPFMT_API int32_t mpfmt_reedsshepp_fmtstar_wavefront(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                           int32_t goal_kind, const double* goal_params, double band, int32_t flags, int64_t* A, double* C,
                                           int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info);
/* mpfmt_di_fmtstar_wavefront : mpfmt_di_fmtstar with the recursion on the device: the double-integrator graph is directed, so
 *        the forward sets nearF (rows of the cost matrix, linearquadratic.jl:73) are transposed on the device and the edge
 *        answers come from the 5-waypoint sweep's mask and segment counts.  MPFMT_WF_SINGLE reproduces mpfmt_di_fmtstar exactly;
 *        A / C / path may be NULL. */
MPFMT_API int32_t mpfmt_di_fmtstar_wavefront(mpfmt_ctx* ctx, double rho, double r, int64_t init_idx, int32_t checkpts, int32_t goal_kind,
                                   const double* goal_params, double band, int32_t flags, int64_t* A, double* C, int64_t* path,
                                   mpfmt_fmt_result* res, mpfmt_wf_info* info);

/* ---- multi-GPU exchange: RCCL over xGMI behind the ABI.  The reference has no analogue (single thread,
 *      src/planners/fmt.jl); the exchange assembles on every GPU what the single process of the reference holds in one
 *      address space: the free bit of every graph edge (is_free_motion results, statespaces.jl:153-158) and, in the
 *      wavefront solve below, the connections of every wavefront (fmt.jl:76-80).
 *      comm_unique_id : 128-byte RCCL id; rank 0 obtains it, the host hands it to every rank (file, socket, Julia array).
 *      comm_create    : ncclCommInitRank on the ctx's device + mpfmt_set_shard(rank, world).  world = 1 is allowed.
 *      group_begin/end: ncclGroupStart / ncclGroupEnd, for ONE host thread that drives several ctxs (a Julia process
 *                       holding G handles): bracket the comm_create calls, and each round of *_launch calls, with them.
 *      allgather_free_mask : after mpfmt_graph_step_device on every rank -- one all-gather of the per-shard free-edge masks.
 *                       *gathered = device address of [world][*stride_words] uint64: per rank (mask words, nnz, then the
 *                       mask, zero padded); words_each / nnz_each [world] on the host (may be NULL).  Steady state is ONE
 *                       collective: the lengths ride in front of the payload and every rank derives the next capacity
 *                       from the lengths it saw (a shard that outgrew it makes every rank repeat at the exact size).
 *                       _launch / _finish split the call: the gather runs on the ctx's communication stream and overlaps
 *                       whatever is enqueued next (the following step's index build); cap_hint > 0 = capacity in words
 *                       the caller guarantees to be identical on all ranks (needed by a single-thread driver on the
 *                       first step, when the blocking lengths exchange would wait on a peer it has yet to launch).
 *                       _launch snapshots the rank's mask, so the next step may overwrite the ctx's mask before _finish.
 *                       Calls that may sit inside mpfmt_group_begin / _end: mpfmt_comm_create, mpfmt_allgather_free_mask_launch
 *                       (with cap_hint > 0 the first time) and _relaunch; _finish must come after mpfmt_group_end.  A ctx
 *                       launched inside a group never repeats a gather on its own (its peers hang off the same thread): _finish
 *                       returns MPFMT_RETRY on every ctx instead, the driver calls _relaunch for all of them in a group and
 *                       _finish again.  (Verified with the tests' RCCL stand-in, which defers collectives to the end of
 *                       the group like RCCL does; a multi-GPU box has not been available.)
 *                       MPFMT_RCCL_LIB (tests: a stand-in library) is honoured only with MPFMT_ALLOW_RCCL_OVERRIDE=1. */
#define MPFMT_COMM_ID_BYTES 128
MPFMT_API int32_t mpfmt_comm_unique_id(uint8_t* id128);
MPFMT_API int32_t mpfmt_comm_create(mpfmt_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id128);
MPFMT_API int32_t mpfmt_comm_destroy(mpfmt_ctx* ctx);
MPFMT_API int32_t mpfmt_group_begin(void);
MPFMT_API int32_t mpfmt_group_end(void);
MPFMT_API int32_t mpfmt_allgather_free_mask(mpfmt_ctx* ctx, void** gathered, int64_t* stride_words, int64_t* words_each, int64_t* nnz_each);
MPFMT_API int32_t mpfmt_allgather_free_mask_launch(mpfmt_ctx* ctx, int64_t cap_hint);
MPFMT_API int32_t mpfmt_allgather_free_mask_finish(mpfmt_ctx* ctx, void** gathered, int64_t* stride_words, int64_t* words_each, int64_t* nnz_each);
MPFMT_API int32_t mpfmt_allgather_free_mask_relaunch(mpfmt_ctx* ctx);

/* ---- measurement: average device milliseconds per launch of a named kernel group since the last
 *      reset, measured with HIP events on the launch stream.  names: "rdisc_count", "rdisc_fill",
 *      "rdisc_sort", "grid", "sweep_graph" (mask preset + round table + kernel), "sweep_kernel" (the round-table sweep kernel alone),
 *      "pair_kernel" (the pair kernel alone), "exact_pairs" (k_exact_pairs, edge-test form 2),
 *      "sweep_points", "sweep_edges", "expand". */
MPFMT_API int32_t mpfmt_timing_reset(mpfmt_ctx* ctx);
MPFMT_API int32_t mpfmt_timing_get(mpfmt_ctx* ctx, const char* name, double* avg_ms, int64_t* launches);
/* Tuning / test knobs.  "rdisc_path": 0 = auto, 1 = exact fp64 VALU pair kernel, 2 = fp16 MFMA distance-matrix
 * filter + exact fp64 refine (both give bit-identical graphs).  "timing": 0/1 event timing off/on.
 * "sweep_rounds" (default 1): round-table sweep kernel where it applies
 * (d <= 8, at most 256 boxes), 0 = the task-header kernel everywhere.  Same mask bits either way.
 * "rdisc_half" (default 1): the single-pass build tests every pair of the ctx's own samples once and writes the hit records
 * of both columns (same CSC bit for bit; a build that overflows its logs is counted again whole).
 * "fuse_broad" (default 2): in mpfmt_graph_step* on such a build (AABB checker in the state space's own coordinates, d <= 6,
 * <= 256 boxes, every sample inside the state space) the edge tests ride in the graph kernels: 2 = broad phase in the pair
 * kernel, slab tests of the flagged pairs before the columns are ordered, the ordering pass writes the mask; 1 = flagged entries
 * tested after the ordering; 0 = the separate sweep kernel.  Same graph and mask bits in every combination; a sweep called on
 * its own (mpfmt_graph_sweep_device, mpfmt_graph_edges_free) is always the whole sweep.
 * "di_path" (default 0 = auto): the double-integrator build's candidate test on the vector ALUs (1) or as an fp16 bilinear form on the
 * matrix cores in front of the same fp64 tests (2; workspace dim <= 2 and a threshold the fp16 error bound leaves meaningful); same graph.
 * "wf_graphs" (default 0): a measured alternative kept for the record (LABNOTES.md).
 * "mf_target_items" (default 40000): work items (tile x slice of its chunk list, one wavefront each) the MFMA pair kernel aims for; the
 * slice count is made odd.  "mf_xcd_mode" (default -1 = by launch size): items reach the XCDs in interleaved groups of this many
 * (0 = one contiguous range per XCD, 1 = round robin).  "cell_fb_max" (default 8): position bits inside a cell that the cell sort's key
 * carries.
 * "overlap" (default 1): the step runs its per-sample obstacle masks and the counter fill beside the chunk lists, and the degree count,
 * its scan, the mask preset and the capacity check beside the exact pair tests, on a second (lowest-priority) stream of the ctx forked
 * and joined with events -- from 65536 samples on (smaller steps are shorter than the events' latencies); 2 = always; 0 = every kernel
 * on the ctx's stream.  "ord_draw" (default 1): the ordering kernel's workgroups draw their
 * quarter tiles from per-XCD counters when a quarter holds >= 1536 records on average; 2 = always; 0 = every nb-th quarter each.  "mf_tail_slices" / "mf_tail_permille" / "mf_tail_min_items"
 * (defaults 9 / 80 / 32768): the last permille of a single-pass launch's tiles are cut into that many (odd) slices instead of the
 * launch's own, in launches of at least that many items.  "index_halo" / "shard_blocks" (default 1 / 1, sharded ctxs): the index is
 * built for the shard's tiles and their neighbour cells only; cell ids are block-major so that a shard is a compact block.
 * "wf_pos_space" (default 1): the device solve keeps its sets a second time by cell-sorted position from the SECOND solve on a graph on
 * (2: from the first, 0: never).
 * Tuning knobs only: the graph and the masks are the same bit for bit whatever their values.
 * "debug_small_lists": test knob, shrinks the pending lists of the fused edge tests so that their overflow path runs. */
MPFMT_API int32_t mpfmt_set_option(mpfmt_ctx* ctx, const char* name, int64_t value);
/* Counters of the last graph build: "rdisc_path_used", "pairs_tested", "survivors" (pairs that passed the
 * MFMA filter), "nnz", "slices", "cells", "rdisc_half_used", "pool_used"; of the last step's edge tests: "sweep_form"
 * (0 / 1 / 2 as "fuse_broad"), "pair_items" (pairs listed for the exact tests; a synchronising read); diagnostics: "redo_count" /
 * "redo_reason" (builds redone since the ctx was made and why: 1 a chunk list was cut, 2 a log overflowed, 4 a column too long for the
 * ordering pass, 8 more entries than allocated, 16 the pending-pair list was cut), "qcap" (records a quarter log holds), "ord_per_cu"
 * (ordering-pass workgroups per CU), "filter_valu" (1: the last single-pass build filtered with the exact fp64 test on the vector ALUs),
 * "wf_pos_space_used" (1: the last device solve gathered by cell-sorted position), "list_cap" / "list_max" / "list_q<permille>" / "list_argmax" / "list_sum" (chunk-list lengths of
 * the last list build; synchronising reads). */
MPFMT_API int32_t mpfmt_get_stat(mpfmt_ctx* ctx, const char* name, int64_t* value);
/* work counters of the last graph build: candidate pairs distance-tested, tiles, slices. */
MPFMT_API int32_t mpfmt_graph_stats(mpfmt_ctx* ctx, int64_t* pairs_tested, int64_t* tiles, int64_t* slices, int64_t* cells);

#ifdef __cplusplus
}
#endif
#endif /* MPFMT_H */
