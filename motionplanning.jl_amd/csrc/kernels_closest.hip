// Closest obstacle points in a Mahalanobis metric: closest / closeR of
// src/collisioncheckers/boxesND.jl:61-86 (axis-aligned boxes, through the bounded-variable least squares of
// src/collisioncheckers/bvls.jl:19-218) and src/collisioncheckers/SAT2D.jl:208-285 (circles, convex polygons, compounds),
// batched over many query points -- the building block of the collision-probability estimators README.md:9-10 cites.
//
// Mapping: lane = (point, obstacle) pair, pair id = obstacle * n + point, so a wavefront shares its obstacle (scalar loads)
// and reads consecutive points.  Pass 1 writes d2[obstacle][point]; pass 2 (lane = point) takes the minimum / the r2-filtered
// sorted list over the obstacles and re-solves the winners for their closest points, so no n x M x d array of points exists.
//
// bvls on the device follows bvls.jl's active-set iteration step for step (initial bounds by magnitude, steepest locked
// gradient freed, `oops` list, criti, step length, clamp), bookkeeping quirks included -- a freed variable that is locked again
// through the oops branch keeps state 0 (:113, :151-164), which is why some pairs exhaust the 10n iterations: the reference
// then returns `nothing` and closest() throws; here the pair is reported in `failures` and skipped.  Only the projected
// solve differs in form: with A = chol(W) the least-squares problem A[:, free] \ (b - A[:, bound] x_bound) (:131-142) has the
// normal equations W_ff y = W_fb (p_b - x_b), z = p_f + y, solved by an in-register Cholesky of the masked matrix (bound
// rows / columns replaced by identity) with static indexing.  Results agree with the QR route to rounding (tests: 1e-9).
#include <vector>
#include <cmath>
#include <cstring>
#include "mpfmt_internal.h"

#define CL_THREADS 128

template <int D> struct cl_w { double W[D][D]; };

struct cl_2d {                      // per-call constants of the 2-D methods, derived from W on the host
    int32_t has_w;
    double W[2][2];
    double s1, s2, v1[2], v2[2];    // eigfact(W): ascending values, orthonormal vectors (SAT2D.jl:212)
    double U[2][2], Ui[2][2];       // chol(W) upper and its inverse (SAT2D.jl:256-257)
};

// ---- bvls for one box -------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int bvls_box(const cl_w<D>& Wm, const double (&p)[D], const double (&l)[D], const double (&u)[D],
                                        const double tol, double (&x)[D])
{
    uint32_t st1 = 0, st2 = 0, btw = 0, oops = 0;                       // state == 1 / == 2, between (atbound = ~between), oopslist
    int criti = -1, crits = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {                                       // bvls.jl:41-63
        const uint32_t b = 1u << i;
        if (u[i] >= INFINITY && l[i] <= -INFINITY) { x[i] = 0.0; btw |= b; }
        else if (u[i] >= INFINITY) { x[i] = l[i]; st1 |= b; }
        else if (l[i] <= -INFINITY) { x[i] = u[i]; st2 |= b; }
        else if (fabs(l[i]) <= fabs(u[i])) { x[i] = l[i]; st1 |= b; }
        else { x[i] = u[i]; st2 |= b; }
    }
    // The iteration is a deterministic map of (x, state, between, oopslist, criti, crits): a state seen before means the
    // remaining iterations repeat it and the 10n limit will be reached -- report that outcome at once (snapshots at
    // iterations 1, 2, 4, ..., Brent's cycle detection).  Without this a wavefront with one such lane runs all 10n rounds.
    double sx[D];
    uint32_t s_st1 = 0, s_st2 = 0, s_btw = 0, s_oops = 0;
    int s_criti = 0, s_crits = 0, next_snap = 1;
    for (int iter = 1; iter <= 10 * D; ++iter) {                        // :67-69
        if (iter > 1) {
            bool same = (st1 == s_st1) && (st2 == s_st2) && (btw == s_btw) && (oops == s_oops) && (criti == s_criti) && (crits == s_crits);
#pragma unroll
            for (int i = 0; i < D; ++i) same = same && (x[i] == sx[i]);
            if (same) return -1;
        }
        if (iter == next_snap) {
#pragma unroll
            for (int i = 0; i < D; ++i) sx[i] = x[i];
            s_st1 = st1; s_st2 = st2; s_btw = btw; s_oops = oops; s_criti = criti; s_crits = crits;
            next_snap *= 2;
        }
        double g[D];
        bool done = true;
        int newi = -1;
        double newg = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {                                   // grad = A'(Ax - b) = W(x - p)   (:73-77)
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) s = s + Wm.W[i][j] * (x[j] - p[j]);
            g[i] = ((oops >> i) & 1u) ? 0.0 : s;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {                                   // optimality (:81-89) and the variable to free (:94-112)
            const bool s1 = (st1 >> i) & 1u, s2 = (st2 >> i) & 1u;
            const bool bad = (fabs(g[i]) > tol && !s1 && !s2) || (g[i] < 0.0 && s1) || (g[i] > 0.0 && s2);
            done = done && !bad;
            if (!((btw >> i) & 1u) && i != criti) {
                if (g[i] > 0.0 && s2 && fabs(g[i]) > newg) { newi = i; newg = fabs(g[i]); }
                if (g[i] < 0.0 && s1 && fabs(g[i]) > newg) { newi = i; newg = fabs(g[i]); }
            }
        }
        if (done) return iter;
        if (newi >= 0) { const uint32_t b = 1u << newi; btw |= b; st1 &= ~b; st2 &= ~b; }     // :116-120

        // projected problem (:131-142): masked Cholesky
        double G[D][D], y[D], xnew[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) s = s + (((btw >> j) & 1u) ? 0.0 : Wm.W[i][j] * (p[j] - x[j]));
            y[i] = ((btw >> i) & 1u) ? s : 0.0;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const bool fj = (btw >> j) & 1u;
            double s = fj ? Wm.W[j][j] : 1.0;
#pragma unroll
            for (int k = 0; k < j; ++k) s = s - G[j][k] * G[j][k];
            const double gjj = sqrt(s);
            G[j][j] = gjj;
#pragma unroll
            for (int i = j + 1; i < D; ++i) {
                double t = (fj && ((btw >> i) & 1u)) ? Wm.W[i][j] : 0.0;
#pragma unroll
                for (int k = 0; k < j; ++k) t = t - G[i][k] * G[j][k];
                G[i][j] = t / gjj;
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double s = y[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s = s - G[i][k] * y[k];
            y[i] = s / G[i][i];
        }
#pragma unroll
        for (int i = D - 1; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int k = i + 1; k < D; ++k) s = s - G[k][i] * y[k];
            y[i] = s / G[i][i];
        }
        bool oopsed = false;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            xnew[i] = ((btw >> i) & 1u) ? p[i] + y[i] : x[i];
            if (i == newi) oopsed = (xnew[i] <= l[i] && x[i] == l[i]) || (xnew[i] >= u[i] && x[i] == u[i]);     // :146-147
        }
        if (oopsed) {                                                   // :151-165 (state[newi] stays 0: it was zeroed at :113)
            const uint32_t b = 1u << newi;
            oops |= b; btw &= ~b;
            continue;
        }
        oops = 0;                                                       // :169
        double alpha = 1.0;                                             // :174-195
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if ((btw >> i) & 1u) {
                if (xnew[i] > u[i]) { const double na = fmin(alpha, (u[i] - x[i]) / (xnew[i] - x[i])); if (na < alpha) { criti = i; crits = 2; alpha = na; } }
                if (xnew[i] < l[i]) { const double na = fmin(alpha, (l[i] - x[i]) / (xnew[i] - x[i])); if (na < alpha) { criti = i; crits = 1; alpha = na; } }
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = x[i] + alpha * (xnew[i] - x[i]);                    // :199
        if (alpha < 1.0) {                                              // :203-207
            const uint32_t b = 1u << criti;
            btw &= ~b;
            st1 = (st1 & ~b) | (crits == 1 ? b : 0u);
            st2 = (st2 & ~b) | (crits == 2 ? b : 0u);
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {                                   // :209-222
            const uint32_t b = 1u << i;
            if (x[i] >= u[i]) { x[i] = u[i]; st2 |= b; st1 &= ~b; btw &= ~b; }
            if (x[i] <= l[i]) { x[i] = l[i]; st1 |= b; st2 &= ~b; btw &= ~b; }
        }
    }
    return -1;
}

// closest(p, BB, W) (boxesND.jl:61-70): false when bvls ran out of iterations
template <int D>
__device__ __forceinline__ bool closest_box(const cl_w<D>& Wm, const double tol, const double (&p)[D], const double* __restrict__ box,
                                            double& d2, double (&v)[D])
{
    double l[D], u[D];
#pragma unroll
    for (int q = 0; q < D; ++q) { l[q] = box[q]; u[q] = box[D + q]; }
    if (bvls_box<D>(Wm, p, l, u, tol, v) < 0) { d2 = NAN; return false; }
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s = s + Wm.W[i][j] * (v[j] - p[j]);
        acc = acc + (v[i] - p[i]) * s;
    }
    d2 = acc;
    return true;
}

// ---- 2-D shapes (SAT2D.jl:208-259) -------------------------------------------------------------------------------------
__device__ __forceinline__ double cl_dot2(const double a0, const double a1, const double b0, const double b1)
{
    const double p = a0 * b0, q = a1 * b1;
    return p + q;
}

__device__ __forceinline__ void closest_polypts(const double p0, const double p1, const mpfmt_shape2d* __restrict__ S, const cl_2d& K,
                                                const bool xf, double& d2min, double& v0, double& v1)
{
    const int n = S->n;
    auto px = [&](int i) { return xf ? K.U[0][0] * S->pts[i][0] + K.U[0][1] * S->pts[i][1] : S->pts[i][0]; };
    auto py = [&](int i) { return xf ? K.U[1][1] * S->pts[i][1] : S->pts[i][1]; };
    d2min = INFINITY; v0 = px(0); v1 = py(0);
    for (int i = 0; i < n; ++i) {                                       // SAT2D.jl:243-252
        const int nx = (i + 1 < n) ? i + 1 : 0;
        const double ax = px(i), ay = py(i), bx = px(nx), by = py(nx);
        const double e0 = bx - ax, e1 = by - ay;
        const double t = cl_dot2(e0, e1, p0 - ax, p1 - ay) / cl_dot2(e0, e1, e0, e1);
        double c0, c1;
        if (t < 0.0) { c0 = ax; c1 = ay; }
        else if (t < 1.0) { c0 = ax + t * e0; c1 = ay + t * e1; }
        else { c0 = bx; c1 = by; }
        const double dd = cl_dot2(p0 - c0, p1 - c1, p0 - c0, p1 - c1);
        if (dd < d2min) { d2min = dd; v0 = c0; v1 = c1; }
    }
}

// closest(p, S [, W]) for one shape; false when the circle's multiplier iteration did not end (the reference's loops are
// unbounded; 200 Newton steps / 64 halvings are where this library stops a run that would not have ended there either)
__device__ __forceinline__ bool closest_shape(const double p0, const double p1, const mpfmt_shape2d* __restrict__ S, const cl_2d& K,
                                              double& d2, double& x0, double& x1)
{
    if (!K.has_w) {
        if (S->kind == MPFMT_SHAPE_CIRCLE) {                            // SAT2D.jl:208-211
            const double w0 = p0 - S->c[0], w1 = p1 - S->c[1];
            const double nv = sqrt(cl_dot2(w0, w1, w0, w1));
            x0 = S->c[0] + S->r * (w0 / nv); x1 = S->c[1] + S->r * (w1 / nv);
            d2 = cl_dot2(p0 - x0, p1 - x1, p0 - x0, p1 - x1);
        } else closest_polypts(p0, p1, S, K, false, d2, x0, x1);       // :239
        return true;
    }
    if (S->kind == MPFMT_SHAPE_CIRCLE) {                                // :213-238
        const double s1 = K.s1, s2 = K.s2, r2 = S->r * S->r;
        const double c0 = p0 - S->c[0], c1 = p1 - S->c[1];
        const double q1 = cl_dot2(K.v1[0], K.v1[1], c0, c1), q2 = cl_dot2(K.v2[0], K.v2[1], c0, c1);
        auto F = [&](double lam) { const double a = q1 * s1 / (lam + s1), b = q2 * s2 / (lam + s2); return (a * a + b * b) - r2; };
        double lambda = 1.0;
        double f = F(lambda);
        int it = 0;
        while (fabs(f) > 1e-8) {
            if (++it > 200 || !(f == f)) { d2 = NAN; x0 = x1 = NAN; return false; }
            const double a = q1 * s1 / (lambda + s1), b = q2 * s2 / (lambda + s2);
            const double fp = -2.0 / (lambda + s1) * (a * a) + -2.0 / (lambda + s2) * (b * b);
            double alpha = 1.0, lnew, fnew;
            int h = 0;
            for (;;) {
                lnew = lambda - alpha * f / fp;
                fnew = F(lnew);
                if (fabs(fnew) < fabs(f)) break;
                alpha = alpha / 2.0;
                if (++h > 64) { d2 = NAN; x0 = x1 = NAN; return false; }
            }
            f = fnew; lambda = lnew;
        }
        const double k1 = q1 * s1 / (lambda + s1), k2 = q2 * s2 / (lambda + s2);
        x0 = (S->c[0] + K.v1[0] * k1) + K.v2[0] * k2;
        x1 = (S->c[1] + K.v1[1] * k1) + K.v2[1] * k2;
        d2 = s1 * ((q1 - k1) * (q1 - k1)) + s2 * ((q2 - k2) * (q2 - k2));
        return true;
    }
    const double l0 = K.U[0][0] * p0 + K.U[0][1] * p1, l1 = K.U[1][1] * p1;                    // :255-259
    double dd, y0, y1;
    closest_polypts(l0, l1, S, K, true, dd, y0, y1);
    x0 = K.Ui[0][0] * y0 + K.Ui[0][1] * y1;
    x1 = K.Ui[1][1] * y1;
    const double t0 = x0 - p0, t1 = x1 - p1;
    const double w0 = K.W[0][0] * t0 + K.W[0][1] * t1, w1 = K.W[1][0] * t0 + K.W[1][1] * t1;
    d2 = t0 * w0 + t1 * w1;
    return true;
}

// ---- kernels -------------------------------------------------------------------------------------------------------------
// pass 1: d2all[k * n + i] = closest(P[i], obstacle k, W)[1], NaN where the reference would have thrown
template <int D>
__global__ __launch_bounds__(CL_THREADS) void k_cl_pairs_boxes(const double* __restrict__ P, int64_t n, const double* __restrict__ boxes, int32_t M,
                                                               cl_w<D> Wm, double tol_scale, double* __restrict__ d2all)
{
    const int64_t id = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x;
    if (id >= n * (int64_t)M) return;
    const int64_t k = id / n, i = id - k * n;
    double p[D], v[D], d2;
#pragma unroll
    for (int q = 0; q < D; ++q) p[q] = P[i * D + q];
    double pWp = 0.0;                                                   // norm(b)^2 = |chol(W) p|^2 = p'Wp
#pragma unroll
    for (int a = 0; a < D; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < D; ++b) s = s + Wm.W[a][b] * p[b];
        pWp = pWp + p[a] * s;
    }
    const double tol = (1.0 + sqrt(fmax(pWp, 0.0))) * tol_scale;        // (1 + norm(b)) * myeps   (bvls.jl:84)
    closest_box<D>(Wm, tol, p, boxes + (size_t)k * 2 * D, d2, v);
    d2all[id] = d2;
}

__global__ __launch_bounds__(CL_THREADS) void k_cl_pairs_shapes(const double* __restrict__ P, int64_t n, const mpfmt_shape2d* __restrict__ S, int32_t M,
                                                                cl_2d K, double* __restrict__ d2all)
{
    const int64_t id = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x;
    if (id >= n * (int64_t)M) return;
    const int64_t k = id / n, i = id - k * n;
    double d2, x0, x1;
    closest_shape(P[2 * i], P[2 * i + 1], S + k, K, d2, x0, x1);
    d2all[id] = d2;
}

// pass 2 (closest): minimum over the obstacles with strict <  (boxesND.jl:72-81, SAT2D.jl:260-279); winner re-solved for v
template <int D, bool SHAPES>
__global__ __launch_bounds__(CL_THREADS) void k_cl_min(const double* __restrict__ P, int64_t n, const void* __restrict__ obst, int32_t M,
                                                       cl_w<D> Wm, cl_2d K, double tol_scale, const double* __restrict__ d2all,
                                                       double* __restrict__ d2min, double* __restrict__ vmin, int64_t* __restrict__ kmin,
                                                       unsigned long long* __restrict__ failures)
{
    const int64_t i = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x;
    if (i >= n) return;
    double best = INFINITY;
    int32_t kb = -1, bad = 0;
    for (int32_t k = 0; k < M; ++k) {
        const double dd = d2all[(int64_t)k * n + i];
        bad += (dd != dd);
        if (dd < best) { best = dd; kb = k; }
    }
    double p[D], v[D];
#pragma unroll
    for (int q = 0; q < D; ++q) { p[q] = P[i * D + q]; v[q] = SHAPES ? 0.0 : p[q]; }         // (Inf, zeros) / (Inf, p) when nothing wins
    if (kb >= 0) {
        double dd;
        if constexpr (SHAPES) {
            double x0, x1;
            closest_shape(p[0], p[1], (const mpfmt_shape2d*)obst + kb, K, dd, x0, x1);
            v[0] = x0; v[1] = x1;
        } else {
            double pWp = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                double s = 0.0;
#pragma unroll
                for (int b = 0; b < D; ++b) s = s + Wm.W[a][b] * p[b];
                pWp = pWp + p[a] * s;
            }
            closest_box<D>(Wm, (1.0 + sqrt(fmax(pWp, 0.0))) * tol_scale, p, (const double*)obst + (size_t)kb * 2 * D, dd, v);
        }
    }
    d2min[i] = best;
    kmin[i] = (int64_t)kb + 1;
#pragma unroll
    for (int q = 0; q < D; ++q) vmin[i * D + q] = v[q];
    if (bad) atomicAdd(failures, (unsigned long long)bad);
}

// pass 2 (closeR): count of obstacles with d2 < r2 per point
__global__ __launch_bounds__(CL_THREADS) void k_cl_count(int64_t n, int32_t M, double r2, const double* __restrict__ d2all, int64_t* __restrict__ cnt,
                                                         unsigned long long* __restrict__ failures)
{
    const int64_t i = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x;
    if (i >= n) return;
    int32_t c = 0, bad = 0;
    for (int32_t k = 0; k < M; ++k) {
        const double dd = d2all[(int64_t)k * n + i];
        bad += (dd != dd);
        c += (dd < r2);
    }
    cnt[i] = c;
    if (bad) atomicAdd(failures, (unsigned long long)bad);
}

// pass 3 (closeR): the lists in ascending (d2, obstacle) order -- the stable sort of boxesND.jl:85 / SAT2D.jl:285
template <int D, bool SHAPES>
__global__ __launch_bounds__(CL_THREADS) void k_cl_fill(const double* __restrict__ P, int64_t n, const void* __restrict__ obst, int32_t M,
                                                        cl_w<D> Wm, cl_2d K, double tol_scale, double r2, const double* __restrict__ d2all,
                                                        const int64_t* __restrict__ ptr, int64_t* __restrict__ idx, double* __restrict__ d2out,
                                                        double* __restrict__ vout)
{
    const int64_t i = (int64_t)blockIdx.x * CL_THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t beg = ptr[i], cnt = ptr[i + 1] - beg;
    if (cnt == 0) return;
    double p[D];
#pragma unroll
    for (int q = 0; q < D; ++q) p[q] = P[i * D + q];
    double tol = 0.0;
    if constexpr (!SHAPES) {
        double pWp = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < D; ++b) s = s + Wm.W[a][b] * p[b];
            pWp = pWp + p[a] * s;
        }
        tol = (1.0 + sqrt(fmax(pWp, 0.0))) * tol_scale;
    }
    double last = -INFINITY;
    int32_t lastk = -1;
    for (int64_t j = 0; j < cnt; ++j) {
        double bd = INFINITY;
        int32_t bk = -1;
        for (int32_t k = 0; k < M; ++k) {                               // next (d2, k) after (last, lastk) in lexicographic order
            const double dd = d2all[(int64_t)k * n + i];
            const bool after = (dd > last) || (dd == last && k > lastk);
            if (dd < r2 && after && dd < bd) { bd = dd; bk = k; }
        }
        last = bd; lastk = bk;
        double dd, v[D];
        if constexpr (SHAPES) {
            double x0, x1;
            closest_shape(p[0], p[1], (const mpfmt_shape2d*)obst + bk, K, dd, x0, x1);
            v[0] = x0; v[1] = x1;
        } else {
            closest_box<D>(Wm, tol, p, (const double*)obst + (size_t)bk * 2 * D, dd, v);
        }
        idx[beg + j] = (int64_t)bk + 1;
        d2out[beg + j] = bd;
#pragma unroll
        for (int q = 0; q < D; ++q) vout[(beg + j) * D + q] = v[q];
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------
namespace {
struct Tmp {
    std::vector<void*> p;
    ~Tmp() { for (void* q : p) if (q) hipFree(q); }
    template <class T> hipError_t get(T** out, size_t bytes)
    {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
        if (e == hipSuccess) { p.push_back(q); *out = (T*)q; }
        return e;
    }
};

bool chol2(const double W[2][2], double U[2][2])
{
    if (!(W[0][0] > 0.0)) return false;
    U[0][0] = sqrt(W[0][0]); U[0][1] = W[0][1] / U[0][0]; U[1][0] = 0.0;
    const double s = W[1][1] - U[0][1] * U[0][1];
    if (!(s > 0.0)) return false;
    U[1][1] = sqrt(s);
    return true;
}

bool is_spd(const double* W, int d)
{
    std::vector<double> U((size_t)d * d, 0.0);
    for (int j = 0; j < d; ++j) {
        double s = W[j * d + j];
        for (int k = 0; k < j; ++k) s -= U[k * d + j] * U[k * d + j];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        U[j * d + j] = sqrt(s);
        for (int i = j + 1; i < d; ++i) {
            double t = W[j * d + i];
            for (int k = 0; k < j; ++k) t -= U[k * d + j] * U[k * d + i];
            U[j * d + i] = t / U[j * d + j];
        }
    }
    return true;
}

// eigfact of the symmetric 2x2 (closed form; LAPACK's vectors may differ in sign, which cancels in SAT2D.jl:213-238)
void eig2(const double W[2][2], cl_2d* K)
{
    const double a = W[0][0], b = 0.5 * (W[0][1] + W[1][0]), c = W[1][1];
    const double h = 0.5 * (a - c), m = 0.5 * (a + c), rad = sqrt(h * h + b * b);
    K->s1 = m - rad; K->s2 = m + rad;
    if (b == 0.0) {
        if (a <= c) { K->v1[0] = 1; K->v1[1] = 0; K->v2[0] = 0; K->v2[1] = 1; }
        else        { K->v1[0] = 0; K->v1[1] = 1; K->v2[0] = 1; K->v2[1] = 0; }
        return;
    }
    double e0, e1;
    if (fabs(K->s2 - a) >= fabs(K->s2 - c)) { e0 = b; e1 = K->s2 - a; } else { e0 = K->s2 - c; e1 = b; }
    const double nn = sqrt(e0 * e0 + e1 * e1);
    K->v2[0] = e0 / nn; K->v2[1] = e1 / nn;
    K->v1[0] = -K->v2[1]; K->v1[1] = K->v2[0];
}

struct cl_setup {
    int32_t d = 0, M = 0;
    bool shapes = false;
    const void* obst = nullptr;
    cl_2d K;
    std::vector<double> W;
};

int32_t cl_prepare(mpfmt_ctx* ctx, const double* P, int64_t n, const double* W, bool need_w, cl_setup* s)
{
    if (n < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n < 0");
    if (n > 0 && !P) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "points array is NULL");
    memset(&s->K, 0, sizeof s->K);
    if (ctx->cc_kind == 1) {
        if (!ctx->shapes2d) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_shapes2d)");
        s->shapes = true; s->d = 2; s->M = ctx->M; s->obst = ctx->shapes2d;
    } else {
        if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
        s->d = ctx->dw; s->M = ctx->M; s->obst = ctx->boxes;
        if (!W) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "closest(p, boxes) needs the weight matrix W (boxesND.jl:61)");
    }
    if (need_w && !W) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "closeR needs the weight matrix W (SAT2D.jl:281-285)");
    const int d = s->d;
    if ((int64_t)n * (int64_t)(s->M > 0 ? s->M : 1) > ((int64_t)1 << 31))
        return mpfmt_fail(ctx, MPFMT_ERR_ARG, "points x obstacles = %lld pairs > 2^31: split the batch", (long long)n * s->M);
    if (W) {
        s->W.assign(W, W + (size_t)d * d);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j)
                if (W[i * d + j] != W[j * d + i]) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "W must be symmetric");
        if (!is_spd(W, d)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "W must be positive definite (chol(W), boxesND.jl:65)");
        if (s->shapes) {
            cl_2d& K = s->K;
            K.has_w = 1;
            K.W[0][0] = W[0]; K.W[0][1] = W[1]; K.W[1][0] = W[2]; K.W[1][1] = W[3];
            eig2(K.W, &K);
            chol2(K.W, K.U);
            const double det = K.U[0][0] * K.U[1][1];                  // inv of the 2x2 SMatrix: adjugate / det
            K.Ui[0][0] = K.U[1][1] / det; K.Ui[0][1] = -K.U[0][1] / det; K.Ui[1][0] = 0.0; K.Ui[1][1] = K.U[0][0] / det;
        }
    }
    return MPFMT_OK;
}

template <int D> cl_w<D> make_w(const cl_setup& s)
{
    cl_w<D> w;
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) w.W[i][j] = s.W.empty() ? (i == j ? 1.0 : 0.0) : s.W[(size_t)i * D + j];
    return w;
}

#define CL_DISPATCH_D(DIM, EXPR)                                                                      \
    switch (DIM) {                                                                                    \
        case 1: { constexpr int DD = 1; EXPR; } break;   case 2: { constexpr int DD = 2; EXPR; } break;   \
        case 3: { constexpr int DD = 3; EXPR; } break;   case 4: { constexpr int DD = 4; EXPR; } break;   \
        case 5: { constexpr int DD = 5; EXPR; } break;   case 6: { constexpr int DD = 6; EXPR; } break;   \
        case 7: { constexpr int DD = 7; EXPR; } break;   case 8: { constexpr int DD = 8; EXPR; } break;   \
        case 9: { constexpr int DD = 9; EXPR; } break;   case 10: { constexpr int DD = 10; EXPR; } break; \
        case 11: { constexpr int DD = 11; EXPR; } break; case 12: { constexpr int DD = 12; EXPR; } break; \
        default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "closest: unsupported dimension %d (1..12)", (int)(DIM)); \
    }

constexpr double CL_MYEPS = 1.0e-10;                                    // bvls.jl:38

int32_t cl_pairs(mpfmt_ctx* ctx, const cl_setup& s, const double* dP, int64_t n, double* d2all)
{
    const int64_t pairs = n * (int64_t)s.M;
    if (pairs == 0) return MPFMT_OK;
    const unsigned nb = (unsigned)((pairs + CL_THREADS - 1) / CL_THREADS);
    mpfmt_timed tm1(ctx);
    if (s.shapes)
        hipLaunchKernelGGL(k_cl_pairs_shapes, dim3(nb), dim3(CL_THREADS), 0, ctx->stream, dP, n, (const mpfmt_shape2d*)s.obst, s.M, s.K, d2all);
    else
        CL_DISPATCH_D(s.d, hipLaunchKernelGGL((k_cl_pairs_boxes<DD>), dim3(nb), dim3(CL_THREADS), 0, ctx->stream, dP, n, (const double*)s.obst, s.M,
                                              make_w<DD>(s), CL_MYEPS, d2all));
    tm1.end("closest_pairs");
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
}  // namespace

extern "C" {

int32_t mpfmt_closest(mpfmt_ctx* ctx, const double* P, int64_t n, const double* W, double* d2min, double* vmin, int64_t* kmin, int64_t* failures)
{
    if (!ctx) return MPFMT_ERR_ARG;
    cl_setup s;
    int32_t rc;
    if ((rc = cl_prepare(ctx, P, n, W, false, &s))) return rc;
    if (failures) *failures = 0;
    if (n == 0) return MPFMT_OK;
    if (!d2min || !vmin || !kmin) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output array");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int d = s.d;
    Tmp tmp;
    double *dP, *d2all, *dd2, *dv; int64_t* dk; unsigned long long* dfail;
    HIPCHK(ctx, tmp.get(&dP, sizeof(double) * (size_t)n * d));
    HIPCHK(ctx, tmp.get(&d2all, sizeof(double) * (size_t)n * (s.M > 0 ? s.M : 1)));
    HIPCHK(ctx, tmp.get(&dd2, sizeof(double) * (size_t)n));
    HIPCHK(ctx, tmp.get(&dv, sizeof(double) * (size_t)n * d));
    HIPCHK(ctx, tmp.get(&dk, sizeof(int64_t) * (size_t)n));
    HIPCHK(ctx, tmp.get(&dfail, sizeof(unsigned long long)));
    HIPCHK(ctx, hipMemcpyAsync(dP, P, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(dfail, 0, sizeof(unsigned long long), ctx->stream));
    if ((rc = cl_pairs(ctx, s, dP, n, d2all))) return rc;
    const unsigned nb = (unsigned)((n + CL_THREADS - 1) / CL_THREADS);
    mpfmt_timed tm2(ctx);
    if (s.shapes)
        hipLaunchKernelGGL((k_cl_min<2, true>), dim3(nb), dim3(CL_THREADS), 0, ctx->stream, dP, n, s.obst, s.M, make_w<2>(s), s.K, CL_MYEPS, d2all,
                           dd2, dv, dk, dfail);
    else
        CL_DISPATCH_D(d, hipLaunchKernelGGL((k_cl_min<DD, false>), dim3(nb), dim3(CL_THREADS), 0, ctx->stream, dP, n, s.obst, s.M, make_w<DD>(s), s.K,
                                            CL_MYEPS, d2all, dd2, dv, dk, dfail));
    tm2.end("closest_select");
    HIPCHK(ctx, hipGetLastError());
    unsigned long long nf = 0;
    HIPCHK(ctx, hipMemcpyAsync(d2min, dd2, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(vmin, dv, sizeof(double) * (size_t)n * d, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(kmin, dk, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&nf, dfail, sizeof nf, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (failures) *failures = (int64_t)nf;
    return MPFMT_OK;
}

int32_t mpfmt_closeR(mpfmt_ctx* ctx, const double* P, int64_t n, const double* W, double r2, int64_t* ptr, int64_t cap, int64_t* obstacle,
                     double* d2, double* v, int64_t* total, int64_t* failures)
{
    if (!ctx) return MPFMT_ERR_ARG;
    cl_setup s;
    int32_t rc;
    if ((rc = cl_prepare(ctx, P, n, W, true, &s))) return rc;
    if (!ptr || !total) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "ptr / total is NULL");
    if (cap < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "cap < 0");
    if (failures) *failures = 0;
    *total = 0;
    ptr[0] = 1;
    if (n == 0) return MPFMT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int d = s.d;
    Tmp tmp;
    double *dP, *d2all; int64_t *dcnt, *dptr; unsigned long long* dfail;
    HIPCHK(ctx, tmp.get(&dP, sizeof(double) * (size_t)n * d));
    HIPCHK(ctx, tmp.get(&d2all, sizeof(double) * (size_t)n * (s.M > 0 ? s.M : 1)));
    HIPCHK(ctx, tmp.get(&dcnt, sizeof(int64_t) * (size_t)(n + 1)));
    HIPCHK(ctx, tmp.get(&dptr, sizeof(int64_t) * (size_t)(n + 1)));
    HIPCHK(ctx, tmp.get(&dfail, sizeof(unsigned long long)));
    HIPCHK(ctx, hipMemcpyAsync(dP, P, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(dfail, 0, sizeof(unsigned long long), ctx->stream));
    if ((rc = cl_pairs(ctx, s, dP, n, d2all))) return rc;
    const unsigned nb = (unsigned)((n + CL_THREADS - 1) / CL_THREADS);
    hipLaunchKernelGGL(k_cl_count, dim3(nb), dim3(CL_THREADS), 0, ctx->stream, n, s.M, r2, d2all, dcnt, dfail);
    HIPCHK(ctx, hipGetLastError());
    std::vector<int64_t> cnt((size_t)n), hp((size_t)n + 1);
    unsigned long long nf = 0;
    HIPCHK(ctx, hipMemcpyAsync(cnt.data(), dcnt, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&nf, dfail, sizeof nf, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (failures) *failures = (int64_t)nf;
    hp[0] = 0;
    for (int64_t i = 0; i < n; ++i) hp[i + 1] = hp[i] + cnt[i];
    const int64_t tot = hp[n];
    for (int64_t i = 0; i <= n; ++i) ptr[i] = hp[i] + 1;
    *total = tot;
    if (tot > cap) return mpfmt_fail(ctx, MPFMT_ERR_CAPACITY, "closeR: %lld entries > cap %lld", (long long)tot, (long long)cap);
    if (tot == 0) return MPFMT_OK;
    if (!obstacle || !d2 || !v) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output array");
    int64_t* didx; double *dd2, *dv;
    HIPCHK(ctx, tmp.get(&didx, sizeof(int64_t) * (size_t)tot));
    HIPCHK(ctx, tmp.get(&dd2, sizeof(double) * (size_t)tot));
    HIPCHK(ctx, tmp.get(&dv, sizeof(double) * (size_t)tot * d));
    HIPCHK(ctx, hipMemcpyAsync(dptr, hp.data(), sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice, ctx->stream));
    mpfmt_timed tm3(ctx);
    if (s.shapes)
        hipLaunchKernelGGL((k_cl_fill<2, true>), dim3(nb), dim3(CL_THREADS), 0, ctx->stream, dP, n, s.obst, s.M, make_w<2>(s), s.K, CL_MYEPS, r2, d2all,
                           dptr, didx, dd2, dv);
    else
        CL_DISPATCH_D(d, hipLaunchKernelGGL((k_cl_fill<DD, false>), dim3(nb), dim3(CL_THREADS), 0, ctx->stream, dP, n, s.obst, s.M, make_w<DD>(s), s.K,
                                            CL_MYEPS, r2, d2all, dptr, didx, dd2, dv));
    tm3.end("closest_select");
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(obstacle, didx, sizeof(int64_t) * (size_t)tot, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d2, dd2, sizeof(double) * (size_t)tot, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(v, dv, sizeof(double) * (size_t)tot * d, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

}  // extern "C"
