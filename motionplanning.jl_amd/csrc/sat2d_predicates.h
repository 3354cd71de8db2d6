// Device predicates of the 2-D SAT world (PointRobot2D over Circle / convex Polygon parts: src/collisioncheckers/SAT2D.jl:119-178,
// src/collisioncheckers/robots2D.jl:12-14, src/utilities/vec2Dutils.jl), shared by the point / segment / graph kernels of
// kernels_sat2d.hip and by the steering-graph sweeps (double integrator, Dubins, Reeds-Shepp), whose workspace is 2-D.
#pragma once
#include "mpfmt_internal.h"

// the 2-D workspace checker a steering sweep tests its waypoint segments against: AABBs (kind 0, [M][2][2]) or shapes (kind 1)
struct mpfmt_ws2d {
    int32_t kind;
    const double* boxes; int32_t M;
    const mpfmt_shape2d* shapes; int32_t ns;
    mpfmt_aabb2d aabb;
};

// ---- device predicates --------------------------------------------------------------------------------------------------
__device__ __forceinline__ double dot2(double ax, double ay, double bx, double by) { const double p = ax * bx; const double q = ay * by; return p + q; }
__device__ __forceinline__ double cross2(double ax, double ay, double bx, double by) { const double p = ax * by; const double q = ay * bx; return p - q; }
__device__ __forceinline__ bool overlapping(double a0, double a1, double b0, double b1) { return (a0 <= b1) & (b0 <= a1); }   // vec2Dutils.jl:33
__device__ __forceinline__ bool ininterval(double x, double i0, double i1) { return (i0 <= x) & (x <= i1); }                  // :34

// colliding(p, S) (SAT2D.jl:121-127).  The polygon form is the reference's as written: `@all [!ininterval(...)]`.
__device__ __forceinline__ bool point_hits(double px, double py, const mpfmt_shape2d* __restrict__ S)
{
    if (S->kind == MPFMT_SHAPE_CIRCLE) {
        const double tx = px - S->c[0], ty = py - S->c[1];
        return dot2(tx, ty, tx, ty) <= S->r * S->r;
    }
    if (!(ininterval(px, S->xr[0], S->xr[1]) && ininterval(py, S->yr[0], S->yr[1]))) return false;
    bool all = true;
    for (int i = 0; i < S->n; ++i) all = all & !ininterval(dot2(px, py, S->normals[i][0], S->normals[i][1]), S->nex[i][0], S->nex[i][1]);
    return all;
}

// colliding(L, B) = colliding_ends_free(L, B) || colliding(L.v, B) || colliding(L.w, B) (SAT2D.jl:163-176), L = Line(v, w) (:66-81)
__device__ __forceinline__ bool line_hits(double vx, double vy, double wx, double wy, const mpfmt_shape2d* __restrict__ S)
{
    const double ex = wx - vx, ey = wy - vy;
    const double lx0 = (vx < wx) ? vx : wx, lx1 = (vx < wx) ? wx : vx;       // minmaxV
    const double ly0 = (vy < wy) ? vy : wy, ly1 = (vy < wy) ? wy : vy;
    bool ends_free_hit = false;
    if (overlapping(lx0, lx1, S->xr[0], S->xr[1]) && overlapping(ly0, ly1, S->yr[0], S->yr[1])) {
        if (S->kind == MPFMT_SHAPE_CIRCLE) {
            const double cx = S->c[0] - vx, cy = S->c[1] - vy;
            const double d2 = dot2(ex, ey, ex, ey);
            const double cr = cross2(ex, ey, cx, cy);
            const double lhs = d2 * (S->r * S->r), rhs = cr * cr;
            const double t = dot2(cx, cy, ex, ey);
            ends_free_hit = !(lhs < rhs) & (0 <= t) & (t <= d2);
        } else {
            const double nx = ey, ny = -ex;                                  // perp(edge), not normalised
            const double ndotv = dot2(vx, vy, nx, ny);
            double dmin = INFINITY, dmax = -INFINITY;
            for (int i = 0; i < S->n; ++i) {
                const double d = dot2(S->pts[i][0], S->pts[i][1], nx, ny);
                dmin = (d < dmin) ? d : dmin;
                dmax = (d > dmax) ? d : dmax;
            }
            bool hit = ininterval(ndotv, dmin, dmax);                        // !is_separating_axis(L, P)
            for (int i = 0; i < S->n; ++i) {                                 // !any is_separating_axis(P, L, i)
                const double a = dot2(vx, vy, S->normals[i][0], S->normals[i][1]), b = dot2(wx, wy, S->normals[i][0], S->normals[i][1]);
                const double l0 = (a < b) ? a : b, l1 = (a < b) ? b : a;
                hit = hit & overlapping(S->nex[i][0], S->nex[i][1], l0, l1);
            }
            ends_free_hit = hit;
        }
    }
    return ends_free_hit || point_hits(vx, vy, S) || point_hits(wx, wy, S);
}

// is_free_state(v, CC) = !colliding(v, obstacles) (robots2D.jl:12; SAT2D.jl:129-132)
__device__ __forceinline__ bool point_free_2d(double px, double py, const mpfmt_shape2d* __restrict__ S, int ns, const mpfmt_aabb2d& B)
{
    if (!(ininterval(px, B.xr[0], B.xr[1]) && ininterval(py, B.yr[0], B.yr[1]))) return true;
    bool hit = false;
    for (int i = 0; i < ns; ++i) hit = hit | point_hits(px, py, S + i);
    return !hit;
}

// is_free_motion(v, w, CC) = !colliding(Line(v, w), obstacles) (robots2D.jl:13-14; SAT2D.jl:154-157,178)
__device__ __forceinline__ bool motion_free_2d(double vx, double vy, double wx, double wy, const mpfmt_shape2d* __restrict__ S, int ns,
                                               const mpfmt_aabb2d& B)
{
    const double lx0 = (vx < wx) ? vx : wx, lx1 = (vx < wx) ? wx : vx;
    const double ly0 = (vy < wy) ? vy : wy, ly1 = (vy < wy) ? wy : vy;
    if (!(overlapping(B.xr[0], B.xr[1], lx0, lx1) && overlapping(B.yr[0], B.yr[1], ly0, ly1))) return true;
    bool hit = false;
    for (int i = 0; i < ns; ++i) hit = hit | line_hits(vx, vy, wx, wy, S + i);
    return !hit;
}

__device__ __forceinline__ bool in_ss_2d(double x, double y, const mpfmt_ss& ss)
{
    if (!ss.has) return true;
    return (ss.lo[0] <= x) & (x <= ss.hi[0]) & (ss.lo[1] <= y) & (y <= ss.hi[1]);
}

