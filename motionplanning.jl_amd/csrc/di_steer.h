// Double-integrator (LinearQuadratic) steer: the device functions and launch arguments shared by the vector-ALU pair kernel
// (kernels_di.hip) and the matrix-core prefilter (kernels_di_mfma.hip).  Reference: src/statespaces/linearquadratic.jl:126-157,175-195.
#pragma once
#include "mpfmt_internal.h"

#define DI_QCAP 256

struct di_coef { double a, b, c; };     // |p|^2, p.(v0+v1), |v0|^2 + v0.v1 + |v1|^2

template <int M>
__device__ __forceinline__ di_coef di_coefs(const double* x0, const double* x1)
{
    di_coef k = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const double p = x1[i] - x0[i];
        const double v0 = x0[M + i], v1 = x1[M + i];
        k.a = k.a + p * p;
        k.b = k.b + p * (v0 + v1);
        k.c = k.c + ((v0 * v0 + v0 * v1) + v1 * v1);
    }
    return k;
}
__device__ __forceinline__ double di_cost(di_coef k, double rho, double t)
{
    const double t2 = t * t, t3 = t2 * t;
    return t + rho * ((12.0 * k.a / t3 - 12.0 * k.b / t2) + 4.0 * k.c / t);
}
__device__ __forceinline__ double di_dcost(di_coef k, double rho, double t)
{
    const double t2 = t * t, t3 = t2 * t, t4 = t2 * t2;
    return 1.0 - rho * ((36.0 * k.a / t4 - 24.0 * k.b / t3) + 4.0 * k.c / t2);
}
__device__ __forceinline__ double di_ddcost(di_coef k, double rho, double t)
{
    const double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
    return rho * ((144.0 * k.a / t5 - 72.0 * k.b / t4) + 8.0 * k.c / t3);
}
// topt_newton, linearquadratic.jl:175-190 (tol = 1e-6)
__device__ __forceinline__ double di_topt_newton(di_coef k, double rho, double tm)
{
    const double tol = 1e-6;
    double b = tm;
    if (di_dcost(k, rho, b) < 0) return tm;
    double a = tm / 100;
    while (di_dcost(k, rho, a) > 0) a /= 2;
    double t = tm / 2;
    double cdval = di_dcost(k, rho, t);
    while (fabs(cdval) > tol && fabs(a - b) > tol) {
        t = t - cdval / di_ddcost(k, rho, t);
        if (t < a || t > b) t = (a + b) / 2;
        cdval = di_dcost(k, rho, t);
        if (cdval > 0) b = t; else a = t;
    }
    return t;
}
// steer, linearquadratic.jl:191-195
template <int M>
__device__ __forceinline__ void di_steer(const double* x0, const double* x1, double rho, double r, double& cost, double& topt)
{
    int same = 1;
#pragma unroll
    for (int i = 0; i < 2 * M; ++i) same &= (int)(x0[i] == x1[i]);
    if (same) { cost = 0.0; topt = 0.0; return; }
    const di_coef k = di_coefs<M>(x0, x1);
    const double t = di_topt_newton(k, rho, r);
    cost = di_cost(k, rho, t);
    topt = t;
}
// x(v, w, t, s): state on the optimal trajectory (closed form of the SymPy `x` closure, :137-138,156)
template <int M>
__device__ __forceinline__ void di_state(const double* x0, const double* x1, double t, double s, double* out)
{
    const double t2 = t * t, t3 = t2 * t;
    const double s2 = s * s, s3 = s2 * s;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const double v0 = x0[M + i], v1 = x1[M + i];
        const double dp = (x1[i] - x0[i]) - t * v0;
        const double dv = v1 - v0;
        const double d1 = 12.0 * dp / t3 - 6.0 * dv / t2;
        const double d2 = -6.0 * dp / t2 + 4.0 * dv / t;
        const double e = (t - s) * d1 + d2;
        out[i] = (x0[i] + s * v0) + (s3 / 3.0 * d1 + s2 / 2.0 * e);
        out[M + i] = v0 + (s2 / 2.0 * d1 + s * e);
    }
}

// ---- all-pairs sparse cost graph -----------------------------------------------------------------------------
struct di_args {
    const double* X;            // [N][2M] states, caller order
    int64_t N;
    double rho, r;
    double i2, i3, i4;          // 1/r^2, 1/r^3, 1/r^4 for the multiply-only candidate pre-test
    double r2;                  // r^2 (the second pre-test)
    int32_t S;                  // source slices per target tile
    int64_t ntiles;
    int32_t* slice_cnt;         // [S][ntiles*64]
    const int64_t* colptr;
    int32_t* rowtmp;
    double* valtmp;
    double* tvaltmp;
    unsigned long long* counters;   // [0] pairs tested, [1] candidates
    int64_t tile_step;              // 1; > 1 for the pilot launch that only visits every tile_step-th tile
    // single-pass slot lists (MODE 2): accepted hits kept per (item, target lane)
    int32_t* pool_i; double* pool_c; double* pool_t;
    int64_t pool_cap;
    int32_t* pool_flag;
};

