// Dubins car: DubinsQuasiMetricSpace of src/statespaces/simplecars.jl -- SE2 states (x, y, theta),
// workspace (x, y), exact Dubins length as a quasi-metric "chopped" by the Euclidean lower bound on positions:
//   backward set of j = { i : |xy_i - xy_j| <= r  and  dubins(i -> j) <= r }      (nearneighbors.jl:185-198)
// Built here as: (1) the Euclidean r-disc graph of the positions with the library's own MFMA pair path (a helper ctx holding
// the N x 2 positions), (2) one lane per candidate edge: the six Dubins words (simplecars.jl:106-194), keep cost <= r,
// (3) ordered compaction into the CSC the planner consumes.  The collision sweep regenerates the steering controls per edge,
// walks the reference's collision waypoints (arcs sampled every pi/12, :68-83; target appended, statespaces.jl:135) and tests
// consecutive pairs on (x, y) against the AABB set with the SE2 bounds on the first point (statespaces.jl:153-158).
// Arithmetic: the reference's expressions in the written order, unfused; sin / cos / atan2 / acos / fmod are the device
// libm's, so costs agree with a CPU libm to a few ulp rather than bit for bit (the tests bound it).
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "mpfmt_internal.h"

#define CAR_TWOPI (2 * 3.141592653589793)

struct car_step { double t, s, k; };          // StepControl(t, (speed, signed curvature))

__host__ __device__ __forceinline__ double mod2pif(double x)      // mod(x, 2pi) with Julia's float mod (utils.jl:91)
{
    const double r = fmod(x, CAR_TWOPI);
    if (r == 0) return 0.0;
    return r < 0 ? r + CAR_TWOPI : r;
}
__host__ __device__ __forceinline__ car_step car_seg(int turn, double d)      // carsegment2stepcontrol, :91
{
    car_step u;
    u.t = fabs(d); u.s = (double)((d > 0) - (d < 0)); u.k = (double)turn;
    return u;
}

#define DUB_TRY(cnew_expr, T0, D0, T1, D1, T2, D2)                                                                        \
    do { const double cnew = (cnew_expr); if (!(c <= cnew)) { path[0] = car_seg(T0, D0); path[1] = car_seg(T1, D1); path[2] = car_seg(T2, D2); c = cnew; } } while (0)

// dubins(s1, s2, r, s) (simplecars.jl:198-215): the words are tried in the reference's order LSL RSR RSL LSR RLR LRL and a
// later word replaces the incumbent only when strictly shorter
__host__ __device__ inline double dubins_steer(const double* s1, const double* s2, double r, double s, car_step* path)
{
    const double vx = (s2[0] - s1[0]) / r, vy = (s2[1] - s1[1]) / r;
    const double d = sqrt(vx * vx + vy * vy);
    const double th = atan2(vy, vx);
    const double a = mod2pif(s1[2] - th), b = mod2pif(s2[2] - th);
    const double ca = cos(a), sa = sin(a), cb = cos(b), sb = sin(b);
    double c = INFINITY;
    for (int q = 0; q < 3; ++q) { path[q].t = 0; path[q].s = 0; path[q].k = 0; }
    {
        const double tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sa - sb));                 // LSL
        if (!(tmp < 0)) {
            const double t0 = atan2(cb - ca, d + sa - sb);
            const double t = mod2pif(-a + t0), p = sqrt(fmax(tmp, 0.0)), q = mod2pif(b - t0);
            DUB_TRY(t + p + q, 1, t, 0, p, 1, q);
        }
    }
    {
        const double tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sb - sa));                 // RSR
        if (!(tmp < 0)) {
            const double t0 = atan2(ca - cb, d - sa + sb);
            const double t = mod2pif(a - t0), p = sqrt(fmax(tmp, 0.0)), q = mod2pif(-b + t0);
            DUB_TRY(t + p + q, -1, t, 0, p, -1, q);
        }
    }
    {
        const double tmp = d * d - 2 + 2 * (ca * cb + sa * sb - d * (sa + sb));                 // RSL
        if (!(tmp < 0)) {
            const double p = sqrt(fmax(tmp, 0.0));
            const double t0 = atan2(ca + cb, d - sa - sb) - atan2(2.0, p);
            const double t = mod2pif(a - t0), q = mod2pif(b - t0);
            DUB_TRY(t + p + q, -1, t, 0, p, 1, q);
        }
    }
    {
        const double tmp = -2 + d * d + 2 * (ca * cb + sa * sb + d * (sa + sb));                // LSR
        if (!(tmp < 0)) {
            const double p = sqrt(fmax(tmp, 0.0));
            const double t0 = atan2(-ca - cb, d + sa + sb) - atan2(-2.0, p);
            const double t = mod2pif(-a + t0), q = mod2pif(-b + t0);
            DUB_TRY(t + p + q, 1, t, 0, p, -1, q);
        }
    }
    {
        const double tmp = (6 - d * d + 2 * (ca * cb + sa * sb + d * (sa - sb))) / 8;           // RLR
        if (!(fabs(tmp) >= 1)) {
            const double p = CAR_TWOPI - acos(tmp);
            const double t0 = atan2(ca - cb, d - sa + sb);
            const double t = mod2pif(a - t0 + p / 2), q = mod2pif(a - b - t + p);
            DUB_TRY(t + p + q, -1, t, 1, p, -1, q);
        }
    }
    {
        const double tmp = (6 - d * d + 2 * (ca * cb + sa * sb - d * (sa - sb))) / 8;           // LRL
        if (!(fabs(tmp) >= 1)) {
            const double p = CAR_TWOPI - acos(tmp);
            const double t0 = atan2(-ca + cb, d + sa - sb);
            const double t = mod2pif(-a + t0 + p / 2), q = mod2pif(b - a - t + p);
            DUB_TRY(t + p + q, 1, t, -1, p, 1, q);
        }
    }
    for (int q = 0; q < 3; ++q) {                 // scalespeed!(scaleradius!(pmin, r), s), :92-105,214
        path[q].t = path[q].t * r; path[q].k = path[q].k / r;
        path[q].t = path[q].t / s; path[q].s = path[q].s * s;
    }
    return c * r;
}

__host__ __device__ __forceinline__ void car_propagate(const double* v, const car_step& u, double* out)      // :52-65
{
    const double ang = u.t * u.s * u.k;
    if (fabs(ang) > 10 * 2.220446049250313e-16) {
        out[0] = v[0] + (sin(v[2] + ang) - sin(v[2])) / u.k;
        out[1] = v[1] + (cos(v[2]) - cos(v[2] + ang)) / u.k;
    } else {
        out[0] = v[0] + u.t * u.s * cos(v[2]);
        out[1] = v[1] + u.t * u.s * sin(v[2]);
    }
    out[2] = mod2pif(v[2] + ang);
}

// ---- 2-D box predicates on (x, y) (boxesND.jl:44-56 at d = 2, straight line) ---------------------------------------------
__device__ __forceinline__ bool seg_free_boxes2(double vx, double vy, double wx, double wy, const double* __restrict__ boxes, int M)
{
    const double lx = (wx < vx) ? wx : vx, hx = (vx < wx) ? wx : vx, ly = (wy < vy) ? wy : vy, hy = (vy < wy) ? wy : vy;
    const double dx = wx - vx, dy = wy - vy;
    bool fr = true;
    for (int k = 0; k < M; ++k) {
        const double lo0 = boxes[4 * k], lo1 = boxes[4 * k + 1], hi0 = boxes[4 * k + 2], hi1 = boxes[4 * k + 3];
        const int sep = (int)(hi0 < lx) | (int)(lo0 > hx) | (int)(hi1 < ly) | (int)(lo1 > hy);
        if (!sep) {
            const double c0 = (vx < lo0) ? lo0 : hi0, c1 = (vy < lo1) ? lo1 : hi1;
            const double l0 = (c0 - vx) / dx, l1 = (c1 - vy) / dy;
            const double y0 = vy + dy * l0;               // face 0: the other coordinate at lambda_0
            const double x1 = vx + dx * l1;               // face 1
            const int hit = ((int)(lo1 <= y0) & (int)(y0 <= hi1)) | ((int)(lo0 <= x1) & (int)(x1 <= hi0));
            if (hit) fr = false;
        }
    }
    return fr;
}
__device__ __forceinline__ bool in_ss3(const double* p, const mpfmt_ss& ss)
{
    if (!ss.has) return true;
    int ok = 1;
    for (int i = 0; i < 3; ++i) ok &= (int)(ss.lo[i] <= p[i]) & (int)(p[i] <= ss.hi[i]);
    return ok != 0;
}

// is_free_motion(v, w, CC, SS) over the reference's collision waypoints; *nseg = segment tests made (CC.count)
__device__ inline bool car_motion_free(const double* v0, const double* w, double rt, double sp, const double* __restrict__ boxes, int M,
                                       const mpfmt_ss& ss, int* nseg)
{
    car_step path[3];
    dubins_steer(v0, w, rt, sp, path);
    const double thres = 3.141592653589793 / 12;
    double v[3] = {v0[0], v0[1], v0[2]};
    double prev[3] = {0, 0, 0};
    bool have_prev = false, ok = true;
    int cnt = 0;
    auto visit = [&](const double* p) {               // p is the next waypoint: test the pair (prev, p)
        if (ok && have_prev) {
            if (!in_ss3(prev, ss)) ok = false;
            else { ++cnt; if (!seg_free_boxes2(prev[0], prev[1], p[0], p[1], boxes, M)) ok = false; }
        }
        prev[0] = p[0]; prev[1] = p[1]; prev[2] = p[2]; have_prev = true;
    };
    for (int q = 0; q < 3 && ok; ++q) {
        const car_step u = path[q];
        const double quo = u.t * u.s * u.k / thres;
        const long m = (long)floor(quo);
        visit(v);
        if (m != 0)
            for (long i = 1; i <= m && ok; ++i) {
                const double ai = (double)i * thres;
                const double p[3] = {v[0] + (sin(v[2] + ai) - sin(v[2])) / u.k, v[1] + (cos(v[2]) - cos(v[2] + ai)) / u.k, mod2pif(v[2] + ai)};
                visit(p);
            }
        double nv[3];
        car_propagate(v, u, nv);
        v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2];
    }
    visit(w);
    *nseg = cnt;
    return ok;
}

// ---- kernels -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_car_xy(const double* __restrict__ X, int64_t N, double* __restrict__ XY)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { XY[2 * i] = X[3 * i]; XY[2 * i + 1] = X[3 * i + 1]; }
}

// lane = candidate entry e of the positions graph (row i, column j): cost(i -> j); keep bit = cost <= r
__global__ __launch_bounds__(256) void k_car_cost(const double* __restrict__ X, int64_t N, const int64_t* __restrict__ ccolptr,
                                                  const int32_t* __restrict__ crowval, int64_t cnnz, double rt, double sp, double r,
                                                  double* __restrict__ cost, uint64_t* __restrict__ keep)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool k = false;
    if (e < cnnz) {
        int64_t lo = 0, hi = N;
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (ccolptr[mid] <= e) lo = mid; else hi = mid; }
        const int64_t j = lo, i = crowval[e];
        car_step path[3];
        const double c = dubins_steer(X + 3 * i, X + 3 * j, rt, sp, path);
        cost[e] = c;
        k = c <= r;
    }
    const unsigned long long bits = __ballot(k);
    if (lane == 0 && (e - lane) < cnnz) keep[(e - lane) >> 6] = bits;
}

__device__ __forceinline__ int popc_range(const uint64_t* __restrict__ keep, int64_t b, int64_t e)      // set bits in [b, e)
{
    int n = 0;
    for (int64_t w = b >> 6; w <= (e - 1) >> 6 && e > b; ++w) {
        uint64_t m = keep[w];
        if (w == (b >> 6)) m &= ~0ull << (b & 63);
        if (w == ((e - 1) >> 6) && ((e & 63) != 0)) m &= (1ull << (e & 63)) - 1ull;
        n += __popcll(m);
    }
    return n;
}

__global__ __launch_bounds__(256) void k_car_degree(const int64_t* __restrict__ ccolptr, const uint64_t* __restrict__ keep, int64_t N,
                                                    int64_t* __restrict__ deg)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N) deg[j] = popc_range(keep, ccolptr[j], ccolptr[j + 1]);
}

// one wavefront per column: kept entries keep their (ascending) order
__global__ __launch_bounds__(64) void k_car_compact(const int64_t* __restrict__ ccolptr, const int32_t* __restrict__ crowval,
                                                    const double* __restrict__ cost, const uint64_t* __restrict__ keep, int64_t N,
                                                    const int64_t* __restrict__ colptr, int32_t* __restrict__ rowval, double* __restrict__ nzval)
{
    const int lane = threadIdx.x;
    for (int64_t j = blockIdx.x; j < N; j += gridDim.x) {
        const int64_t b = ccolptr[j], en = ccolptr[j + 1];
        int64_t out = colptr[j];
        for (int64_t e0 = b; e0 < en; e0 += 64) {
            const int64_t e = e0 + lane;
            const bool k = e < en && ((keep[e >> 6] >> (e & 63)) & 1ull);
            const unsigned long long m = __ballot(k);
            if (k) {
                const int64_t p = out + __popcll(m & ((1ull << lane) - 1ull));
                rowval[p] = crowval[e];
                nzval[p] = cost[e];
            }
            out += __popcll(m);
        }
    }
}

// lane = CSC entry (row y -> column x): bit = is_free_motion(V[y], V[x]), nseg = segment tests the reference would count
__global__ __launch_bounds__(256) void k_car_sweep(const double* __restrict__ X, int64_t N, const int64_t* __restrict__ colptr,
                                                   const int32_t* __restrict__ rowval, int64_t nnz, double rt, double sp,
                                                   const double* __restrict__ boxes, int M, mpfmt_ss ss, uint64_t* __restrict__ mask,
                                                   uint8_t* __restrict__ nseg)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool fr = false;
    if (e < nnz) {
        int64_t lo = 0, hi = N;
        while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (colptr[mid] <= e) lo = mid; else hi = mid; }
        const int64_t x = lo, y = rowval[e];
        int ns = 0;
        fr = car_motion_free(X + 3 * y, X + 3 * x, rt, sp, boxes, M, ss, &ns);
        nseg[e] = (uint8_t)min(ns, 255);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < nnz) mask[(e - lane) >> 6] = bits;
}

__global__ __launch_bounds__(256) void k_car_steer(const double* __restrict__ X0, const double* __restrict__ X1, int64_t n, double rt,
                                                   double sp, double* __restrict__ cost, double* __restrict__ ctrl)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    car_step path[3];
    cost[i] = dubins_steer(X0 + 3 * i, X1 + 3 * i, rt, sp, path);
    for (int q = 0; q < 3; ++q) { ctrl[9 * i + 3 * q] = path[q].t; ctrl[9 * i + 3 * q + 1] = path[q].s; ctrl[9 * i + 3 * q + 2] = path[q].k; }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
static int32_t car_check(mpfmt_ctx* ctx, double rt, double sp, double r)
{
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (ctx->d != 3) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "Dubins states are SE2 (x, y, theta): state dimension must be 3 (got %d)", ctx->d);
    if (!(rt > 0.0) || !std::isfinite(rt) || !(sp > 0.0) || !std::isfinite(sp)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "turning radius and speed must be finite and > 0");
    if (!(r > 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "cost radius must be finite and > 0");
    if (ctx->world != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the Dubins build needs an unsharded ctx");
    return MPFMT_OK;
}

static int32_t car_scan(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n)
{
    size_t tb = 0;
    HIPCHK(ctx, rocprim::exclusive_scan(nullptr, tb, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), ctx->stream));
    void* tmp;
    int32_t rc;
    if ((rc = mpfmt_scratch(ctx, tb + 256, &tmp))) return rc;
    HIPCHK(ctx, rocprim::exclusive_scan(tmp, tb, in, out, (int64_t)0, n, rocprim::plus<int64_t>(), ctx->stream));
    return MPFMT_OK;
}

int32_t mpfmt_dubins_build(mpfmt_ctx* ctx, double rt, double sp, double r)
{
    int32_t rc;
    if ((rc = car_check(ctx, rt, sp, r))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t N = ctx->N;
    // (1) Euclidean r-disc graph of the positions in a helper ctx (same device)
    if (!ctx->aux && (rc = mpfmt_ctx_create(ctx->device, &ctx->aux))) return mpfmt_fail(ctx, rc, "helper ctx: %s", mpfmt_last_error(nullptr));
    mpfmt_ctx* ax = ctx->aux;
    {
        std::vector<double> Xh((size_t)N * 3), xy((size_t)N * 2);
        HIPCHK(ctx, hipMemcpy(Xh.data(), ctx->Xo, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < N; ++i) { xy[2 * i] = Xh[3 * i]; xy[2 * i + 1] = Xh[3 * i + 1]; }
        if ((rc = mpfmt_upload_samples(ax, xy.data(), N, 2))) return mpfmt_fail(ctx, rc, "helper ctx: %s", mpfmt_last_error(ax));
    }
    int64_t cnnz = 0;
    if ((rc = mpfmt_graph_build_device(ax, r, &cnnz))) return mpfmt_fail(ctx, rc, "helper ctx: %s", mpfmt_last_error(ax));
    HIPCHK(ctx, hipStreamSynchronize(ax->stream));
    // (2) exact Dubins cost per candidate edge
    mpfmt_time_begin(ctx);
    const int64_t cw = (cnnz + 63) / 64;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->valtmp, sizeof(double) * (size_t)std::max<int64_t>(cnnz, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->car_keep, sizeof(uint64_t) * (size_t)std::max<int64_t>(cw, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->deg, sizeof(int64_t) * (size_t)(N + 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->colptr, sizeof(int64_t) * (size_t)(N + 1)))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->deg, 0, sizeof(int64_t) * (size_t)(N + 1), ctx->stream));
    if (cnnz > 0) {
        hipLaunchKernelGGL(k_car_cost, dim3((unsigned)((cnnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, N, ax->colptr, ax->rowval, cnnz,
                           rt, sp, r, ctx->valtmp, ctx->car_keep);
        hipLaunchKernelGGL(k_car_degree, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, ax->colptr, ctx->car_keep, N, ctx->deg);
        HIPCHK(ctx, hipGetLastError());
    }
    if ((rc = car_scan(ctx, ctx->deg, ctx->colptr, (size_t)(N + 1)))) return rc;
    int64_t nnz = 0;
    HIPCHK(ctx, hipMemcpyAsync(&nnz, ctx->colptr + N, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // (3) ordered compaction
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rowval, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->nzval, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)))) return rc;
    if (cnnz > 0) {
        hipLaunchKernelGGL(k_car_compact, dim3((unsigned)std::min<int64_t>(N, 1 << 20)), dim3(64), 0, ctx->stream, ax->colptr, ax->rowval,
                           ctx->valtmp, ctx->car_keep, N, ctx->colptr, ctx->rowval, ctx->nzval);
        HIPCHK(ctx, hipGetLastError());
    }
    mpfmt_time_end(ctx, "car_graph");
    ctx->nnz = nnz;
    ctx->pairs_tested = cnnz;
    ctx->car_rt = rt; ctx->car_sp = sp; ctx->di_r = r;
    ctx->steer_kind = 2;
    ctx->di_counted = ctx->di_filled = true; ctx->di_swept = false;
    ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
    return MPFMT_OK;
}

int32_t mpfmt_dubins_sweep(mpfmt_ctx* ctx)
{
    if (!(ctx->di_filled && ctx->steer_kind == 2)) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "Dubins sweep before the Dubins graph is built");
    if (!ctx->have_boxes || ctx->cc_kind != 0 || ctx->dw != 2)
        return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the Dubins sweep needs 2-D boxes (mpfmt_upload_boxes with dw = 2 and the 3 SE2 bounds)");
    if (ctx->ss.has && ctx->ss.d != 3) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "state-space bounds must have 3 dims for SE2 states");
    int32_t rc;
    const int64_t nnz = ctx->nnz, words = (nnz + 63) / 64;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->di_nseg, (size_t)std::max<int64_t>(nnz, 1)))) return rc;
    mpfmt_time_begin(ctx);
    HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
    if (nnz > 0) {
        hipLaunchKernelGGL(k_car_sweep, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, ctx->N, ctx->colptr, ctx->rowval,
                           nnz, ctx->car_rt, ctx->car_sp, ctx->boxes, ctx->M, ctx->ss, ctx->graph_free, ctx->di_nseg);
        HIPCHK(ctx, hipGetLastError());
    }
    mpfmt_time_end(ctx, "car_sweep");
    ctx->di_swept = true;
    return MPFMT_OK;
}

int32_t mpfmt_dubins_steer_batch(mpfmt_ctx* ctx, const double* d_X0, const double* d_X1, int64_t n, double rt, double sp, double* d_cost,
                                 double* d_ctrl)
{
    hipLaunchKernelGGL(k_car_steer, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_X0, d_X1, n, rt, sp, d_cost, d_ctrl);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
