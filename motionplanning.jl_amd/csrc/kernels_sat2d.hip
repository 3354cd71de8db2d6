// 2-D SAT world: PointRobot2D over a Compound2D of Circle / convex Polygon parts
// (src/collisioncheckers/SAT2D.jl, robots2D.jl:12-14, utilities/vec2Dutils.jl) behind the same validity entry points as the
// AABB checker: mpfmt_upload_shapes2d switches the ctx's collision checker, after which points_free / edges_free /
// states_free / motions_free / graph sweep / expand / fmtstar run the SAT predicates instead of the box predicates.
// Arithmetic canon as everywhere: fp64, unfused, dot(a,b) = a1*b1 + a2*b2, cross(a,b) = a1*b2 - a2*b1.
// The shape table is small (tens of shapes, 800 B each) and read with wave-uniform addresses, i.e. through the scalar
// cache; lane = point / segment / CSC entry.
#include <cstring>
#include "mpfmt_internal.h"

// ---- host: the shape constructors -----------------------------------------------------------------------------------
// Circle(c, r): SAT2D.jl:26-28.  Polygon(points): SAT2D.jl:40-55 (orientation made clockwise-negative by the shoelace
// sum, edges, unit normals = normalize(perp(edge)), convexity check on consecutive normal angles, AABB, per-normal extrema).
static void extrema_on(const double (*pts)[2], int n, const double* ax, double* out)       // projectNextrema, vec2Dutils.jl:19-28
{
    double dmin = INFINITY, dmax = -INFINITY;
    for (int i = 0; i < n; ++i) {
        const double p = pts[i][0] * ax[0], q = pts[i][1] * ax[1];
        const double d = p + q;
        if (d < dmin) dmin = d;
        if (d > dmax) dmax = d;
    }
    out[0] = dmin; out[1] = dmax;
}

static int32_t build_shape(mpfmt_ctx* ctx, int idx, int32_t kind, int32_t n, const double* data, mpfmt_shape2d* S)
{
    memset(S, 0, sizeof *S);
    S->kind = kind;
    if (kind == MPFMT_SHAPE_CIRCLE) {
        if (!(data[2] > 0) || !std::isfinite(data[0]) || !std::isfinite(data[1]) || !std::isfinite(data[2]))
            return mpfmt_fail(ctx, MPFMT_ERR_ARG, "shape %d: circle radius must be positive and finite (SAT2D.jl:22)", idx + 1);
        S->c[0] = data[0]; S->c[1] = data[1]; S->r = data[2];
        S->xr[0] = data[0] - data[2]; S->xr[1] = data[0] + data[2];
        S->yr[0] = data[1] - data[2]; S->yr[1] = data[1] + data[2];
        return MPFMT_OK;
    }
    if (kind != MPFMT_SHAPE_POLYGON) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "shape %d: unknown kind %d", idx + 1, kind);
    if (n < 3) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "shape %d: polygons need at least 3 points (SAT2D.jl:42)", idx + 1);
    if (n > MPFMT_MAX_POLY) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "shape %d: %d vertices > %d", idx + 1, n, MPFMT_MAX_POLY);
    S->n = n;
    for (int i = 0; i < n; ++i) {
        S->pts[i][0] = data[2 * i]; S->pts[i][1] = data[2 * i + 1];
        if (!std::isfinite(S->pts[i][0]) || !std::isfinite(S->pts[i][1])) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "shape %d: non-finite vertex", idx + 1);
    }
    double area = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 < n) ? i + 1 : 0;
        const double a = S->pts[j][0] - S->pts[i][0], b = S->pts[j][1] + S->pts[i][1];
        const double t = a * b;
        area = (i == 0) ? t : area + t;
    }
    if (area > 0)
        for (int i = 0; i < n / 2; ++i) { std::swap(S->pts[i][0], S->pts[n - 1 - i][0]); std::swap(S->pts[i][1], S->pts[n - 1 - i][1]); }
    double ang[MPFMT_MAX_POLY];
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 < n) ? i + 1 : 0;
        const double ex = S->pts[j][0] - S->pts[i][0], ey = S->pts[j][1] - S->pts[i][1];
        const double px = ey, py = -ex;                                     // perp
        const double p = px * px, q = py * py;
        const double nrm = std::sqrt(p + q);
        S->normals[i][0] = px / nrm; S->normals[i][1] = py / nrm;
        ang[i] = std::atan2(S->normals[i][1], S->normals[i][0]);
    }
    for (int i = 0; i < n; ++i) {
        const double di = ang[(i + 1 < n) ? i + 1 : 0] - ang[i];
        if (-M_PI <= di && di <= 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "shape %d: polygon must be convex (SAT2D.jl:49)", idx + 1);
    }
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int i = 0; i < n; ++i) {
        xmin = std::min(xmin, S->pts[i][0]); xmax = std::max(xmax, S->pts[i][0]);
        ymin = std::min(ymin, S->pts[i][1]); ymax = std::max(ymax, S->pts[i][1]);
    }
    S->xr[0] = xmin; S->xr[1] = xmax; S->yr[0] = ymin; S->yr[1] = ymax;
    for (int i = 0; i < n; ++i) extrema_on(S->pts, n, S->normals[i], S->nex[i]);
    return MPFMT_OK;
}

int32_t mpfmt_upload_shapes2d(mpfmt_ctx* ctx, int32_t n_shapes, const int32_t* kinds, const int32_t* nverts, const double* data,
                              const double* ss_lo, const double* ss_hi)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (n_shapes < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n_shapes < 0");
    if (n_shapes > 0 && (!kinds || !nverts || !data)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "kinds / nverts / data is NULL");
    if ((ss_lo == nullptr) != (ss_hi == nullptr)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "ss_lo / ss_hi must both be given or both NULL");
    std::vector<mpfmt_shape2d> S((size_t)std::max(n_shapes, 1));
    const double* p = data;
    int32_t rc;
    mpfmt_aabb2d box;
    box.xr[0] = box.yr[0] = INFINITY; box.xr[1] = box.yr[1] = -INFINITY;
    for (int i = 0; i < n_shapes; ++i) {
        if ((rc = build_shape(ctx, i, kinds[i], nverts[i], p, &S[i]))) return rc;
        p += (kinds[i] == MPFMT_SHAPE_CIRCLE) ? 3 : 2 * nverts[i];
        box.xr[0] = std::min(box.xr[0], S[i].xr[0]); box.xr[1] = std::max(box.xr[1], S[i].xr[1]);       // Compound2D ctor, SAT2D.jl:93-97
        box.yr[0] = std::min(box.yr[0], S[i].yr[0]); box.yr[1] = std::max(box.yr[1], S[i].yr[1]);
    }
    if (n_shapes == 0) box.xr[0] = box.xr[1] = box.yr[0] = box.yr[1] = 0.0;                                // SAT2D.jl:90
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->shapes2d, sizeof(mpfmt_shape2d) * S.size()))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->shapes2d, S.data(), sizeof(mpfmt_shape2d) * S.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->M = n_shapes; ctx->dw = 2; ctx->have_boxes = true; ctx->cc_kind = 1; ctx->aabb2d = box;
    ctx->ss.has = ss_lo ? 1 : 0;
    ctx->ss.d = ss_lo ? 2 : 0;
    for (int i = 0; i < MPFMT_MAX_DIM; ++i) { ctx->ss.lo[i] = -INFINITY; ctx->ss.hi[i] = INFINITY; }
    if (ss_lo) for (int i = 0; i < 2; ++i) { ctx->ss.lo[i] = ss_lo[i]; ctx->ss.hi[i] = ss_hi[i]; }
    ctx->graph_swept = false;
    return MPFMT_OK;
}

#include "sat2d_predicates.h"

// ---- kernels ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k2d_points(const double* __restrict__ X, const int64_t* __restrict__ idx1, int64_t n,
                                                  const mpfmt_shape2d* __restrict__ S, int ns, mpfmt_aabb2d B, mpfmt_ss ss,
                                                  uint64_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool fr = false;
    if (e < n) {
        const int64_t s = idx1 ? idx1[e] - 1 : e;
        const double px = X[2 * s], py = X[2 * s + 1];
        fr = in_ss_2d(px, py, ss) && point_free_2d(px, py, S, ns, B);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < n) mask[(e - lane) >> 6] = bits;
}

__global__ __launch_bounds__(256) void k2d_edges(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                 const int64_t* __restrict__ dst1, const double* __restrict__ P,
                                                 const double* __restrict__ Q, int64_t E, const mpfmt_shape2d* __restrict__ S, int ns,
                                                 mpfmt_aabb2d B, mpfmt_ss ss, uint64_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool fr = false;
    if (e < E) {
        double vx, vy, wx, wy;
        if (src1) { const int64_t s = src1[e] - 1, t = dst1[e] - 1; vx = X[2 * s]; vy = X[2 * s + 1]; wx = X[2 * t]; wy = X[2 * t + 1]; }
        else { vx = P[2 * e]; vy = P[2 * e + 1]; wx = Q[2 * e]; wy = Q[2 * e + 1]; }
        fr = in_ss_2d(vx, vy, ss) && motion_free_2d(vx, vy, wx, wy, S, ns, B);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < E) mask[(e - lane) >> 6] = bits;
}

// lane = CSC entry e (row y -> column x): in_state_space(V[y]) && is_free_motion(V[y], V[x], CC)
__global__ __launch_bounds__(256) void k2d_graph(const double* __restrict__ X, int64_t N, const int64_t* __restrict__ colptr,
                                                 const int32_t* __restrict__ rowval, int64_t nnz, const mpfmt_shape2d* __restrict__ S,
                                                 int ns, mpfmt_aabb2d B, mpfmt_ss ss, uint64_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool fr = false;
    if (e < nnz) {
        int64_t lo = 0, hi = N;                                              // column of entry e: largest x with colptr[x] <= e
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (colptr[mid] <= e) lo = mid; else hi = mid; }
        const int64_t x = lo, y = rowval[e];
        const double vx = X[2 * y], vy = X[2 * y + 1], wx = X[2 * x], wy = X[2 * x + 1];
        fr = in_ss_2d(vx, vy, ss) && motion_free_2d(vx, vy, wx, wy, S, ns, B);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < nnz) mask[(e - lane) >> 6] = bits;
}

// ---- launchers (called from the validity entry points when the ctx's checker is the 2-D SAT world) -------------------------
int32_t mpfmt_2d_launch_points(mpfmt_ctx* ctx, const double* X, const int64_t* idx1, int64_t n, uint64_t* d_mask)
{
    if (n == 0) return MPFMT_OK;
    hipLaunchKernelGGL(k2d_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, X, idx1, n, ctx->shapes2d, ctx->M,
                       ctx->aabb2d, ctx->ss, d_mask);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

int32_t mpfmt_2d_launch_edges(mpfmt_ctx* ctx, const int64_t* s1, const int64_t* t1, const double* P, const double* Q, int64_t E,
                              uint64_t* d_mask)
{
    if (E == 0) return MPFMT_OK;
    hipLaunchKernelGGL(k2d_edges, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, s1, t1, P, Q, E, ctx->shapes2d,
                       ctx->M, ctx->aabb2d, ctx->ss, d_mask);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

int32_t mpfmt_2d_launch_graph(mpfmt_ctx* ctx)
{
    if (ctx->nnz == 0) return MPFMT_OK;
    hipLaunchKernelGGL(k2d_graph, dim3((unsigned)((ctx->nnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, ctx->N, ctx->colptr,
                       ctx->rowval, ctx->nnz, ctx->shapes2d, ctx->M, ctx->aabb2d, ctx->ss, (uint64_t*)ctx->graph_free);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
