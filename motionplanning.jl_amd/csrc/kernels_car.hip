// Dubins car: DubinsQuasiMetricSpace of src/statespaces/simplecars.jl -- SE2 states (x, y, theta),
// workspace (x, y), exact Dubins length as a quasi-metric "chopped" by the Euclidean lower bound on positions:
//   backward set of j = { i : |xy_i - xy_j| <= r  and  dubins(i -> j) <= r }      (nearneighbors.jl:185-198)
// Built here as: (1) the Euclidean r-disc graph of the positions with the library's own MFMA pair path (a helper ctx holding
// the N x 2 positions), (2) one lane per candidate edge: the six Dubins words (simplecars.jl:106-194), keep cost <= r,
// (3) ordered compaction into the CSC the planner consumes.  The collision sweep regenerates the steering controls per edge,
// walks the reference's collision waypoints (arcs sampled every pi/12, :68-83; target appended, statespaces.jl:135) and tests
// consecutive pairs on (x, y) against the AABB set with the SE2 bounds on the first point (statespaces.jl:153-158).
// Arithmetic: the reference's expressions in the written order, unfused; sin / cos / atan2 / acos come from mp_math.h (fixed
// reductions and polynomials from + - * / sqrt, the SAME header the CPU oracle compiles: results are bit-identical); fmod / sqrt are the device
// libm's, so costs agree with a CPU libm to a few ulp rather than bit for bit (the tests bound it).
#include <cstring>
#include "mpfmt_internal.h"
#include "mp_math.h"
#include "sat2d_predicates.h"

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
    const double th = mp_atan2(vy, vx);
    const double a = mod2pif(s1[2] - th), b = mod2pif(s2[2] - th);
    const double ca = mp_cos(a), sa = mp_sin(a), cb = mp_cos(b), sb = mp_sin(b);
    double c = INFINITY;
    for (int q = 0; q < 3; ++q) { path[q].t = 0; path[q].s = 0; path[q].k = 0; }
    {
        const double tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sa - sb));                 // LSL
        if (!(tmp < 0)) {
            const double t0 = mp_atan2(cb - ca, d + sa - sb);
            const double t = mod2pif(-a + t0), p = sqrt(fmax(tmp, 0.0)), q = mod2pif(b - t0);
            DUB_TRY(t + p + q, 1, t, 0, p, 1, q);
        }
    }
    {
        const double tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sb - sa));                 // RSR
        if (!(tmp < 0)) {
            const double t0 = mp_atan2(ca - cb, d - sa + sb);
            const double t = mod2pif(a - t0), p = sqrt(fmax(tmp, 0.0)), q = mod2pif(-b + t0);
            DUB_TRY(t + p + q, -1, t, 0, p, -1, q);
        }
    }
    {
        const double tmp = d * d - 2 + 2 * (ca * cb + sa * sb - d * (sa + sb));                 // RSL
        if (!(tmp < 0)) {
            const double p = sqrt(fmax(tmp, 0.0));
            const double t0 = mp_atan2(ca + cb, d - sa - sb) - mp_atan2(2.0, p);
            const double t = mod2pif(a - t0), q = mod2pif(b - t0);
            DUB_TRY(t + p + q, -1, t, 0, p, 1, q);
        }
    }
    {
        const double tmp = -2 + d * d + 2 * (ca * cb + sa * sb + d * (sa + sb));                // LSR
        if (!(tmp < 0)) {
            const double p = sqrt(fmax(tmp, 0.0));
            const double t0 = mp_atan2(-ca - cb, d + sa + sb) - mp_atan2(-2.0, p);
            const double t = mod2pif(-a + t0), q = mod2pif(-b + t0);
            DUB_TRY(t + p + q, 1, t, 0, p, -1, q);
        }
    }
    {
        const double tmp = (6 - d * d + 2 * (ca * cb + sa * sb + d * (sa - sb))) / 8;           // RLR
        if (!(fabs(tmp) >= 1)) {
            const double p = CAR_TWOPI - mp_acos(tmp);
            const double t0 = mp_atan2(ca - cb, d - sa + sb);
            const double t = mod2pif(a - t0 + p / 2), q = mod2pif(a - b - t + p);
            DUB_TRY(t + p + q, -1, t, 1, p, -1, q);
        }
    }
    {
        const double tmp = (6 - d * d + 2 * (ca * cb + sa * sb - d * (sa - sb))) / 8;           // LRL
        if (!(fabs(tmp) >= 1)) {
            const double p = CAR_TWOPI - mp_acos(tmp);
            const double t0 = mp_atan2(-ca + cb, d + sa - sb);
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
        out[0] = v[0] + (mp_sin(v[2] + ang) - mp_sin(v[2])) / u.k;
        out[1] = v[1] + (mp_cos(v[2]) - mp_cos(v[2] + ang)) / u.k;
    } else {
        out[0] = v[0] + u.t * u.s * mp_cos(v[2]);
        out[1] = v[1] + u.t * u.s * mp_sin(v[2]);
    }
    out[2] = mod2pif(v[2] + ang);
}

// ---- Reeds-Shepp (simplecars.jl:228-524): nine word families tried on the target and its time-flipped / reflected /
// backwards images in the reference's order; negative segment lengths are reverse gear -----------------------------------------
#define CAR_PI 3.141592653589793
struct rs_best { double c; int l, post; car_step p[5]; };
__host__ __device__ __forceinline__ void rs_R(double x, double y, double& r, double& th) { r = sqrt(x * x + y * y); th = mp_atan2(y, x); }
__host__ __device__ __forceinline__ double rs_M(double t) { const double m = mod2pif(t); return m > CAR_PI ? m - CAR_TWOPI : m; }
__host__ __device__ inline double rs_Tau(double u, double v, double E, double N)
{
    const double delta = rs_M(u - v);
    const double A = mp_sin(u) - mp_sin(delta);
    const double B = mp_cos(u) - mp_cos(delta) - 1;
    double r, th;
    rs_R(E * A + N * B, N * A - E * B, r, th);
    const double t = 2 * mp_cos(delta) - 2 * mp_cos(v) - 2 * mp_cos(u) + 3;
    return t < 0 ? rs_M(th + CAR_PI) : rs_M(th);
}
__host__ __device__ inline double rs_Omega(double u, double v, double E, double N, double t) { return rs_M(rs_Tau(u, v, E, N) - u + v - t); }
__host__ __device__ __forceinline__ void rs_accept(rs_best& b, double cnew, int l, int post, car_step a0, car_step a1, car_step a2,
                                                   car_step a3, car_step a4)
{
    if (!(b.c <= cnew)) { b.p[0] = a0; b.p[1] = a1; b.p[2] = a2; b.p[3] = a3; b.p[4] = a4; b.c = cnew; b.l = l; b.post = post; }
}
// family f (0..8) on the transformed target T
__host__ __device__ inline void rs_family(int f, const double* T, rs_best& b, int post)
{
    const car_step z = car_seg(0, 0.0);
    const double st = mp_sin(T[2]), ct = mp_cos(T[2]);
    if (f == 0) {                                                         // (8.1) L+ S+ L+
        double r, th; rs_R(T[0] - st, T[1] - 1 + ct, r, th);
        const double u = r, t = mod2pif(th), v = mod2pif(T[2] - t);
        rs_accept(b, t + u + v, 3, post, car_seg(1, t), car_seg(0, u), car_seg(1, v), z, z);
    } else if (f == 1) {                                                  // (8.2) L+ S+ R+
        double r, th, r1, th1; rs_R(T[0] + st, T[1] - 1 - ct, r, th);
        if (r * r < 4) return;
        const double u = sqrt(r * r - 4);
        rs_R(u, 2.0, r1, th1);
        const double t = mod2pif(th + th1), v = mod2pif(t - T[2]);
        rs_accept(b, t + u + v, 3, post, car_seg(1, t), car_seg(0, u), car_seg(-1, v), z, z);
    } else if (f == 2 || f == 3) {                                        // (8.3) L+ R- L+   (8.4) L+ R- L-
        const double E = T[0] - st, N = T[1] + ct - 1;
        if (E * E + N * N > 16) return;
        double r, th; rs_R(E, N, r, th);
        double u = mp_acos(1 - r * r / 8);
        const double t = mod2pif(th - u / 2 + CAR_PI);
        double v = mod2pif(CAR_PI - u / 2 - th + T[2]);
        if (f == 3) v = v - CAR_TWOPI;
        u = -u;
        rs_accept(b, f == 2 ? t - u + v : t - u - v, 3, post, car_seg(1, t), car_seg(-1, u), car_seg(1, v), z, z);
    } else if (f == 4) {                                                  // (8.7) L+ R+u L-u R-
        const double E = T[0] + st, N = T[1] - ct - 1;
        const double p = (2 + sqrt(E * E + N * N)) / 4;
        if (p < 0 || p > 1) return;
        const double u = mp_acos(p);
        const double t = mod2pif(rs_Tau(u, -u, E, N)), v = mod2pif(rs_Omega(u, -u, E, N, T[2])) - CAR_TWOPI;
        rs_accept(b, t + 2 * u - v, 4, post, car_seg(1, t), car_seg(-1, u), car_seg(1, -u), car_seg(-1, v), z);
    } else if (f == 5) {                                                  // (8.8) L+ R-u L-u R+
        const double E = T[0] + st, N = T[1] - ct - 1;
        const double p = (20 - E * E - N * N) / 16;
        if (p < 0 || p > 1) return;
        const double u = -mp_acos(p);
        const double t = mod2pif(rs_Tau(u, u, E, N)), v = mod2pif(rs_Omega(u, u, E, N, T[2]));
        rs_accept(b, t - 2 * u + v, 4, post, car_seg(1, t), car_seg(-1, u), car_seg(1, u), car_seg(-1, v), z);
    } else if (f == 6) {                                                  // (8.9) L+ R- S- L-
        const double E = T[0] - st, N = T[1] + ct - 1;
        double D, be; rs_R(E, N, D, be);
        if (D < 2) return;
        const double ga = mp_acos(2 / D), F = sqrt(D * D / 4 - 1);
        const double t = mod2pif(CAR_PI + be - ga), u = 2 - 2 * F;
        if (u > 0) return;
        const double v = mod2pif(-3 * CAR_PI / 2 + ga + T[2] - be) - CAR_TWOPI;
        rs_accept(b, t + CAR_PI / 2 - u - v, 4, post, car_seg(1, t), car_seg(-1, -CAR_PI / 2), car_seg(0, u), car_seg(1, v), z);
    } else if (f == 7) {                                                  // (8.10) L+ R- S- R-
        const double E = T[0] + st, N = T[1] - ct - 1;
        double D, be; rs_R(E, N, D, be);
        if (D < 2) return;
        const double t = mod2pif(be + CAR_PI / 2), u = 2 - D;
        if (u > 0) return;
        const double v = mod2pif(-CAR_PI - T[2] + be) - CAR_TWOPI;
        rs_accept(b, t + CAR_PI / 2 - u - v, 4, post, car_seg(1, t), car_seg(-1, -CAR_PI / 2), car_seg(0, u), car_seg(-1, v), z);
    } else {                                                              // (8.11) L+ R- S- L- R+
        const double E = T[0] + st, N = T[1] - ct - 1;
        double D, be; rs_R(E, N, D, be);
        if (D < 2) return;
        const double ga = mp_acos(2 / D), F = sqrt(D * D / 4 - 1);
        const double t = mod2pif(CAR_PI + be - ga), u = 4 - 2 * F;
        if (u > 0) return;
        const double v = mod2pif(CAR_PI + be - T[2] - ga);
        rs_accept(b, t + CAR_PI - u + v, 5, post, car_seg(1, t), car_seg(-1, -CAR_PI / 2), car_seg(0, u), car_seg(1, -CAR_PI / 2), car_seg(-1, v));
    }
}

// reedsshepp(s1, s2, r, s) (:265-363): cost, controls path[0..L)
__host__ __device__ inline double rs_steer(const double* s1, const double* s2, double r, double s, car_step* path, int& L)
{
    const double dx = (s2[0] - s1[0]) / r, dy = (s2[1] - s1[1]) / r;
    const double ct = mp_cos(s1[2]), st = mp_sin(s1[2]);
    double T[8][3];                 // images in the reference's POST numbering: 0 id, 1 T, 2 R, 3 B, 4 R_T, 5 B_T, 6 B_R, 7 B_R_T
    T[0][0] = dx * ct + dy * st; T[0][1] = -dx * st + dy * ct; T[0][2] = mod2pif(s2[2] - s1[2]);
    T[1][0] = -T[0][0]; T[1][1] = T[0][1]; T[1][2] = -T[0][2];                       // timeflip
    T[2][0] = T[0][0]; T[2][1] = -T[0][1]; T[2][2] = -T[0][2];                       // reflect
    T[4][0] = T[1][0]; T[4][1] = -T[1][1]; T[4][2] = -T[1][2];                       // reflect(timeflip)
    T[3][0] = T[0][0] * mp_cos(T[0][2]) + T[0][1] * mp_sin(T[0][2]);                       // backwards (:247)
    T[3][1] = T[0][0] * mp_sin(T[0][2]) - T[0][1] * mp_cos(T[0][2]);
    T[3][2] = T[0][2];
    T[5][0] = -T[3][0]; T[5][1] = T[3][1]; T[5][2] = -T[3][2];
    T[6][0] = T[3][0]; T[6][1] = -T[3][1]; T[6][2] = -T[3][2];
    T[7][0] = T[5][0]; T[7][1] = -T[5][1]; T[7][2] = -T[5][2];
    rs_best b;
    b.c = INFINITY; b.l = 0; b.post = 0;
    for (int q = 0; q < 5; ++q) b.p[q] = car_seg(0, 0.0);
    const int order8[8] = {0, 1, 2, 4, 3, 5, 6, 7};
    // images tried per family: 4 (id, T, R, R_T), 2 for (8.3) (id, R), 8 for (8.4), (8.9), (8.10)
    const int count[9] = {4, 4, 2, 8, 4, 4, 8, 8, 4};
    for (int f = 0; f < 9; ++f)
        for (int q = 0; q < count[f]; ++q) {
            const int img = (f == 2) ? (q == 0 ? 0 : 2) : order8[q];
            rs_family(f, T[img], b, img);
        }
    for (int q = 0; q < b.l; ++q) {
        b.p[q].t = b.p[q].t * r; b.p[q].k = b.p[q].k / r;
        b.p[q].t = b.p[q].t / s; b.p[q].s = b.p[q].s * s;
    }
    const bool tf = (b.post == 1 || b.post == 4 || b.post == 5 || b.post == 7);     // timeflip!: negate speed
    const bool rf = (b.post == 2 || b.post == 4 || b.post == 6 || b.post == 7);     // reflect!: negate curvature
    const bool bw = (b.post == 3 || b.post == 5 || b.post == 6 || b.post == 7);     // backwards!: reverse the order
    for (int q = 0; q < b.l; ++q) { if (tf) b.p[q].s = -b.p[q].s; if (rf) b.p[q].k = -b.p[q].k; }
    for (int q = 0; q < 5; ++q) path[q] = (q < b.l) ? (bw ? b.p[b.l - 1 - q] : b.p[q]) : car_seg(0, 0.0);
    L = b.l;
    return b.c * r;
}

// Cost only (the graph's cost filter evaluates 1e8+ candidate pairs and keeps the cost alone): the same 46 evaluations in
// the same order with the same accept rule, but no controls are stored (rs_steer's path arrays and its runtime-indexed image
// table live in scratch: 336 bytes per lane, 2 wavefronts per SIMD), every image is a compile-time case, and the images'
// sin / cos come from ONE sincos of the heading difference -- the eight images only flip its sign (sin odd, cos even).
template <int F>
__device__ __forceinline__ void rs_fam_cost(const double x, const double y, const double th, const double st, const double ct, double& c,
                                            int& win, const int id)
{
    double cnew;
    if constexpr (F == 0) {
        double r, a; rs_R(x - st, y - 1 + ct, r, a);
        const double u = r, t = mod2pif(a), v = mod2pif(th - t);
        cnew = t + u + v;
    } else if constexpr (F == 1) {
        double r, a, r1, a1; rs_R(x + st, y - 1 - ct, r, a);
        if (r * r < 4) return;
        const double u = sqrt(r * r - 4);
        rs_R(u, 2.0, r1, a1);
        const double t = mod2pif(a + a1), v = mod2pif(t - th);
        cnew = t + u + v;
    } else if constexpr (F == 2 || F == 3) {
        const double E = x - st, N = y + ct - 1;
        if (E * E + N * N > 16) return;
        double r, a; rs_R(E, N, r, a);
        double u = mp_acos(1 - r * r / 8);
        const double t = mod2pif(a - u / 2 + CAR_PI);
        double v = mod2pif(CAR_PI - u / 2 - a + th);
        if (F == 3) v = v - CAR_TWOPI;
        u = -u;
        cnew = (F == 2) ? t - u + v : t - u - v;
    } else if constexpr (F == 4) {
        const double E = x + st, N = y - ct - 1;
        const double p = (2 + sqrt(E * E + N * N)) / 4;
        if (p < 0 || p > 1) return;
        const double u = mp_acos(p);
        const double t = mod2pif(rs_Tau(u, -u, E, N)), v = mod2pif(rs_Omega(u, -u, E, N, th)) - CAR_TWOPI;
        cnew = t + 2 * u - v;
    } else if constexpr (F == 5) {
        const double E = x + st, N = y - ct - 1;
        const double p = (20 - E * E - N * N) / 16;
        if (p < 0 || p > 1) return;
        const double u = -mp_acos(p);
        const double t = mod2pif(rs_Tau(u, u, E, N)), v = mod2pif(rs_Omega(u, u, E, N, th));
        cnew = t - 2 * u + v;
    } else if constexpr (F == 6) {
        const double E = x - st, N = y + ct - 1;
        double D, be; rs_R(E, N, D, be);
        if (D < 2) return;
        const double ga = mp_acos(2 / D), Fq = sqrt(D * D / 4 - 1);
        const double t = mod2pif(CAR_PI + be - ga), u = 2 - 2 * Fq;
        if (u > 0) return;
        const double v = mod2pif(-3 * CAR_PI / 2 + ga + th - be) - CAR_TWOPI;
        cnew = t + CAR_PI / 2 - u - v;
    } else if constexpr (F == 7) {
        const double E = x + st, N = y - ct - 1;
        double D, be; rs_R(E, N, D, be);
        if (D < 2) return;
        const double t = mod2pif(be + CAR_PI / 2), u = 2 - D;
        if (u > 0) return;
        const double v = mod2pif(-CAR_PI - th + be) - CAR_TWOPI;
        cnew = t + CAR_PI / 2 - u - v;
    } else {
        const double E = x + st, N = y - ct - 1;
        double D, be; rs_R(E, N, D, be);
        if (D < 2) return;
        const double ga = mp_acos(2 / D), Fq = sqrt(D * D / 4 - 1);
        const double t = mod2pif(CAR_PI + be - ga), u = 4 - 2 * Fq;
        if (u > 0) return;
        const double v = mod2pif(CAR_PI + be - th - ga);
        cnew = t + CAR_PI - u + v;
    }
    if (!(c <= cnew)) { c = cnew; win = id; }                             // rs_accept's rule; id = family * 8 + image
}

// returns the cost / r and the winning (family * 8 + image); the images are those of rs_steer's table
__device__ __forceinline__ double rs_cost_win(const double X, const double Y, const double th, const double st, const double ct,
                                              const double Xb, const double Yb, int& win)
{
    double c = INFINITY;
    win = -1;
    // image i of rs_steer's table: 0 id, 1 T, 2 R, 4 R_T, 3 B, 5 B_T, 6 B_R, 7 B_R_T (angle th or -th: sin flips, cos stays)
#define RS_IMG0(F) rs_fam_cost<F>(X, Y, th, st, ct, c, win, F * 8 + 0)
#define RS_IMG1(F) rs_fam_cost<F>(-X, Y, -th, -st, ct, c, win, F * 8 + 1)
#define RS_IMG2(F) rs_fam_cost<F>(X, -Y, -th, -st, ct, c, win, F * 8 + 2)
#define RS_IMG4(F) rs_fam_cost<F>(-X, -Y, th, st, ct, c, win, F * 8 + 4)
#define RS_IMG3(F) rs_fam_cost<F>(Xb, Yb, th, st, ct, c, win, F * 8 + 3)
#define RS_IMG5(F) rs_fam_cost<F>(-Xb, Yb, -th, -st, ct, c, win, F * 8 + 5)
#define RS_IMG6(F) rs_fam_cost<F>(Xb, -Yb, -th, -st, ct, c, win, F * 8 + 6)
#define RS_IMG7(F) rs_fam_cost<F>(-Xb, -Yb, th, st, ct, c, win, F * 8 + 7)
#define RS_FOUR(F) RS_IMG0(F); RS_IMG1(F); RS_IMG2(F); RS_IMG4(F)
#define RS_EIGHT(F) RS_FOUR(F); RS_IMG3(F); RS_IMG5(F); RS_IMG6(F); RS_IMG7(F)
    RS_FOUR(0); RS_FOUR(1);
    RS_IMG0(2); RS_IMG2(2);
    RS_EIGHT(3);
    RS_FOUR(4); RS_FOUR(5);
    RS_EIGHT(6); RS_EIGHT(7);
    RS_FOUR(8);
#undef RS_IMG0
#undef RS_IMG1
#undef RS_IMG2
#undef RS_IMG3
#undef RS_IMG4
#undef RS_IMG5
#undef RS_IMG6
#undef RS_IMG7
#undef RS_FOUR
#undef RS_EIGHT
    return c;
}

__device__ inline double rs_cost(const double* s1, const double* s2, double r)
{
    const double dx = (s2[0] - s1[0]) / r, dy = (s2[1] - s1[1]) / r;
    const double c1 = mp_cos(s1[2]), s1n = mp_sin(s1[2]);
    const double X = dx * c1 + dy * s1n, Y = -dx * s1n + dy * c1, th = mod2pif(s2[2] - s1[2]);
    const double st = mp_sin(th), ct = mp_cos(th);
    const double Xb = X * ct + Y * st, Yb = X * st - Y * ct;              // backwards image (:247)
    int win;
    return rs_cost_win(X, Y, th, st, ct, Xb, Yb, win) * r;
}

// reedsshepp with its controls, device form: the 46 evaluations run cost-only (above) and remember the winner; only the
// winning word is then evaluated again with its segments -- the same formulas on the same image, so the same cost and
// controls as rs_steer, without carrying five segments through every accept.
__device__ inline double rs_steer_dev(const double* s1, const double* s2, double r, double s, car_step* path, int& L)
{
    const double dx = (s2[0] - s1[0]) / r, dy = (s2[1] - s1[1]) / r;
    const double c1 = mp_cos(s1[2]), s1n = mp_sin(s1[2]);
    const double X = dx * c1 + dy * s1n, Y = -dx * s1n + dy * c1, th = mod2pif(s2[2] - s1[2]);
    const double st = mp_sin(th), ct = mp_cos(th);
    const double Xb = X * ct + Y * st, Yb = X * st - Y * ct;
    int win;
    rs_cost_win(X, Y, th, st, ct, Xb, Yb, win);
    rs_best b;
    b.c = INFINITY; b.l = 0; b.post = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) b.p[q] = car_seg(0, 0.0);
    if (win >= 0) {
        const int f = win >> 3, img = win & 7;
        const bool back = (img == 3 || img == 5 || img == 6 || img == 7);
        const bool fx = (img == 1 || img == 4 || img == 5 || img == 7);     // x negated (timeflip)
        const bool fy = (img == 2 || img == 4 || img == 6 || img == 7);     // y negated (reflect)
        const double bx = back ? Xb : X, by = back ? Yb : Y;
        double T[3];
        T[0] = fx ? -bx : bx; T[1] = fy ? -by : by; T[2] = (fx != fy) ? -th : th;
        rs_family(f, T, b, img);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        if (q < b.l) {
            b.p[q].t = b.p[q].t * r; b.p[q].k = b.p[q].k / r;
            b.p[q].t = b.p[q].t / s; b.p[q].s = b.p[q].s * s;
        }
    }
    const bool tf = (b.post == 1 || b.post == 4 || b.post == 5 || b.post == 7);
    const bool rf = (b.post == 2 || b.post == 4 || b.post == 6 || b.post == 7);
    const bool bw = (b.post == 3 || b.post == 5 || b.post == 6 || b.post == 7);
#pragma unroll
    for (int q = 0; q < 5; ++q) if (q < b.l) { if (tf) b.p[q].s = -b.p[q].s; if (rf) b.p[q].k = -b.p[q].k; }
    // backwards!: reverse the first l segments (static indexing: l is 3, 4 or 5)
    car_step o[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) o[q] = b.p[q];
    if (bw) {
        if (b.l == 3) { o[0] = b.p[2]; o[2] = b.p[0]; }
        else if (b.l == 4) { o[0] = b.p[3]; o[1] = b.p[2]; o[2] = b.p[1]; o[3] = b.p[0]; }
        else if (b.l == 5) { o[0] = b.p[4]; o[1] = b.p[3]; o[3] = b.p[1]; o[4] = b.p[0]; }
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) path[q] = o[q];
    L = b.l;
    return b.c * r;
}

// KIND 1 = Dubins (3 segments), 2 = Reeds-Shepp (up to 5)
template <int KIND>
__host__ __device__ __forceinline__ double car_steer(const double* s1, const double* s2, double r, double s, car_step* path, int& L)
{
    if (KIND == 2) {
#if defined(__HIP_DEVICE_COMPILE__)
        return rs_steer_dev(s1, s2, r, s, path, L);
#else
        return rs_steer(s1, s2, r, s, path, L);
#endif
    }
    L = 3;
    path[3] = car_seg(0, 0.0); path[4] = path[3];
    return dubins_steer(s1, s2, r, s, path);
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
template <int KIND>
__device__ inline bool car_motion_free(const double* v0, const double* w, double rt, double sp, const mpfmt_ws2d& cc,
                                       const mpfmt_ss& ss, int* nseg)
{
    car_step path[5];
    int L;
    car_steer<KIND>(v0, w, rt, sp, path, L);
    const double thres = 3.141592653589793 / 12;
    double v[3] = {v0[0], v0[1], v0[2]};
    double prev[3] = {0, 0, 0};
    bool have_prev = false, ok = true;
    int cnt = 0;
    auto visit = [&](const double* p) {               // p is the next waypoint: test the pair (prev, p)
        if (ok && have_prev) {
            if (!in_ss3(prev, ss)) ok = false;
            else {
                ++cnt;                                     // boxesND.jl:26 / robots2D.jl:13: one count per segment asked for
                const bool fr = cc.kind == 1 ? motion_free_2d(prev[0], prev[1], p[0], p[1], cc.shapes, cc.ns, cc.aabb)
                                             : seg_free_boxes2(prev[0], prev[1], p[0], p[1], cc.boxes, cc.M);
                if (!fr) ok = false;
            }
        }
        prev[0] = p[0]; prev[1] = p[1]; prev[2] = p[2]; have_prev = true;
    };
    for (int q = 0; q < L && ok; ++q) {
        const car_step u = path[q];
        const double quo = u.t * u.s * u.k / thres;
        const long m = (long)floor(quo);
        visit(v);
        if (m != 0)
            for (long i = 1; i <= m && ok; ++i) {
                const double ai = (double)i * thres;
                const double p[3] = {v[0] + (mp_sin(v[2] + ai) - mp_sin(v[2])) / u.k, v[1] + (mp_cos(v[2]) - mp_cos(v[2] + ai)) / u.k, mod2pif(v[2] + ai)};
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

// lane = candidate entry e of the positions graph (row i, column j): Dubins: cost(i -> j) (backward set of j); Reeds-Shepp:
// cost(j -> i) (inball(j), ds = colwise(dist, V[j], V[inds])); keep bit = cost <= r
template <int KIND>
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
        double c;
        if constexpr (KIND == 2) c = rs_cost(X + 3 * j, X + 3 * i, rt);
        else {
            car_step path[5];
            int L;
            c = car_steer<KIND>(X + 3 * i, X + 3 * j, rt, sp, path, L);
        }
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
template <int KIND>
__global__ __launch_bounds__(256) void k_car_sweep(const double* __restrict__ X, int64_t N, const int64_t* __restrict__ colptr,
                                                   const int32_t* __restrict__ rowval, int64_t nnz, double rt, double sp,
                                                   mpfmt_ws2d cc, mpfmt_ss ss, uint64_t* __restrict__ mask,
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
        fr = car_motion_free<KIND>(X + 3 * y, X + 3 * x, rt, sp, cc, ss, &ns);
        nseg[e] = (uint8_t)min(ns, 255);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < nnz) mask[(e - lane) >> 6] = bits;
}

// controls: [n][5][3] = (duration, speed, signed curvature), unused segments zero; nseg[i] = segments used
template <int KIND>
__global__ __launch_bounds__(256) void k_car_steer(const double* __restrict__ X0, const double* __restrict__ X1, int64_t n, double rt,
                                                   double sp, double* __restrict__ cost, double* __restrict__ ctrl, int32_t* __restrict__ nseg)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    car_step path[5];
    int L;
    cost[i] = car_steer<KIND>(X0 + 3 * i, X1 + 3 * i, rt, sp, path, L);
    for (int q = 0; q < 5; ++q) { ctrl[15 * i + 3 * q] = path[q].t; ctrl[15 * i + 3 * q + 1] = path[q].s; ctrl[15 * i + 3 * q + 2] = path[q].k; }
    if (nseg) nseg[i] = L;
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

static int32_t car_scan(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n) { return mpfmt_scan_i64(ctx, in, out, n); }

int32_t mpfmt_car_build(mpfmt_ctx* ctx, int kind, double rt, double sp, double r)
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
    mpfmt_timed tm1(ctx);
    const int64_t cw = (cnnz + 63) / 64;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->valtmp, sizeof(double) * (size_t)std::max<int64_t>(cnnz, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->car_keep, sizeof(uint64_t) * (size_t)std::max<int64_t>(cw, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->deg, sizeof(int64_t) * (size_t)(N + 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->colptr, sizeof(int64_t) * (size_t)(N + 1)))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->deg, 0, sizeof(int64_t) * (size_t)(N + 1), ctx->stream));
    ctx->deg_zero_valid = false;
    if (cnnz > 0) {
        if (kind == 2)
            hipLaunchKernelGGL(k_car_cost<2>, dim3((unsigned)((cnnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, N, ax->colptr, ax->rowval,
                               cnnz, rt, sp, r, ctx->valtmp, ctx->car_keep);
        else
            hipLaunchKernelGGL(k_car_cost<1>, dim3((unsigned)((cnnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, N, ax->colptr, ax->rowval,
                               cnnz, rt, sp, r, ctx->valtmp, ctx->car_keep);
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
    tm1.end("car_graph");
    ctx->nnz = nnz;
    ctx->pairs_tested = cnnz;
    ctx->car_rt = rt; ctx->car_sp = sp; ctx->di_r = r;
    ctx->steer_kind = (kind == 2) ? 3 : 2;
    ctx->di_counted = ctx->di_filled = true; ctx->di_swept = false;
    ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
    return MPFMT_OK;
}

int32_t mpfmt_car_sweep(mpfmt_ctx* ctx)
{
    if (!(ctx->di_filled && (ctx->steer_kind == 2 || ctx->steer_kind == 3))) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car sweep before the car graph is built");
    if (!ctx->have_boxes || ctx->dw != 2)
        return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the car sweep needs a 2-D workspace checker (mpfmt_upload_boxes with dw = 2, or mpfmt_upload_shapes2d) and the 3 SE2 bounds");
    mpfmt_ws2d cc;
    cc.kind = ctx->cc_kind; cc.boxes = ctx->boxes; cc.M = ctx->cc_kind == 0 ? ctx->M : 0;
    cc.shapes = ctx->shapes2d; cc.ns = ctx->cc_kind == 1 ? ctx->M : 0; cc.aabb = ctx->aabb2d;
    if (ctx->ss.has && ctx->ss.d != 3) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "state-space bounds must have 3 dims for SE2 states");
    int32_t rc;
    const int64_t nnz = ctx->nnz, words = (nnz + 63) / 64;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->di_nseg, (size_t)std::max<int64_t>(nnz, 1)))) return rc;
    mpfmt_timed tm2(ctx);
    HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
    if (nnz > 0) {
        if (ctx->steer_kind == 3)
            hipLaunchKernelGGL(k_car_sweep<2>, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, ctx->N, ctx->colptr,
                               ctx->rowval, nnz, ctx->car_rt, ctx->car_sp, cc, ctx->ss, ctx->graph_free, ctx->di_nseg);
        else
            hipLaunchKernelGGL(k_car_sweep<1>, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->Xo, ctx->N, ctx->colptr,
                               ctx->rowval, nnz, ctx->car_rt, ctx->car_sp, cc, ctx->ss, ctx->graph_free, ctx->di_nseg);
        HIPCHK(ctx, hipGetLastError());
    }
    tm2.end("car_sweep");
    ctx->di_swept = true;
    return MPFMT_OK;
}

int32_t mpfmt_car_steer_batch(mpfmt_ctx* ctx, int kind, const double* d_X0, const double* d_X1, int64_t n, double rt, double sp, double* d_cost,
                              double* d_ctrl, int32_t* d_nseg)
{
    if (kind == 2) hipLaunchKernelGGL(k_car_steer<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_X0, d_X1, n, rt, sp, d_cost, d_ctrl, d_nseg);
    else hipLaunchKernelGGL(k_car_steer<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_X0, d_X1, n, rt, sp, d_cost, d_ctrl, d_nseg);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
