// Euclidean per-edge steer: src/statespaces/geometric.jl:18-19 and the partial propagate of
// src/statespaces.jl:79-81, batched -- lane = edge.  HBM-bound streaming: 2 gathered states in, d + 1 doubles out per edge.
//   steering_control(M::Euclidean, v, w) = StepControl(evaluate(M, v, w), normalize(w - v))
//   propagate(M::Euclidean, v, u::StepControl) = v + u.t * u.u
//   propagate(d, v, u::StepControl, s) = s <= 0 ? v : s >= duration(u) ? propagate(d, v, u) : propagate(d, v, StepControl(s, u.u))
// Canonical arithmetic (declared, like SURVEY 8c): norm = sqrt of the index-order sum of squares (the graph's edge cost, bit for
// bit); normalize multiplies by the reciprocal (StaticArrays / Base.normalize: inv(norm) * a), so a zero-length edge yields
// t = 0 and NaN directions, as IEEE arithmetic does in the reference; v + t*u is an unfused multiply then add.
#include "sweep_predicates.h"

template <int D>
__global__ __launch_bounds__(256) void k_euclid_steer(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                      const int64_t* __restrict__ dst1, int64_t E, double* __restrict__ t,
                                                      double* __restrict__ u)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t a = src1[e] - 1, b = dst1[e] - 1;
    double dlt[D];
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        dlt[i] = X[b * D + i] - X[a * D + i];                 // w - v
        const double q = X[a * D + i] - X[b * D + i];         // the graph's d2 is sum (q_i - c_i)^2 with q = the column's sample:
        const double qq = q * q;                              // (v - w)^2 == (w - v)^2 exactly, kept in this form for identity
        s = (i == 0) ? qq : s + qq;
    }
    const double n = sqrt(s);
    const double inv = 1.0 / n;
    t[e] = n;
#pragma unroll
    for (int i = 0; i < D; ++i) u[e * D + i] = inv * dlt[i];
}

template <int D>
__global__ __launch_bounds__(256) void k_euclid_propagate(const double* __restrict__ X, const int64_t* __restrict__ src1, int64_t E,
                                                          const double* __restrict__ t, const double* __restrict__ u,
                                                          const double* __restrict__ s, double* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int64_t a = src1[e] - 1;
    const double te = t[e];
    double step = te;
    bool stay = false;
    if (s) {
        const double se = s[e];
        if (se <= 0.0) stay = true;                            // statespaces.jl:80
        else if (!(se >= te)) step = se;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double v = X[a * D + i];
        const double p = step * u[e * D + i];
        out[e * D + i] = stay ? v : v + p;
    }
}

int32_t mpfmt_launch_euclid_steer(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double* d_t, double* d_u)
{
    const int d = ctx->d;
    const unsigned nb = (unsigned)((E + 255) / 256);
    mpfmt_timed tm(ctx);
    DISPATCH_D(d, hipLaunchKernelGGL((k_euclid_steer<DD>), dim3(nb), dim3(256), 0, ctx->stream, ctx->Xo, d_src1, d_dst1, E, d_t, d_u));
    HIPCHK(ctx, hipGetLastError());
    tm.end("euclid_steer");
    return MPFMT_OK;
}

int32_t mpfmt_launch_euclid_propagate(mpfmt_ctx* ctx, const int64_t* d_src1, int64_t E, const double* d_t, const double* d_u,
                                      const double* d_s, double* d_out)
{
    const int d = ctx->d;
    const unsigned nb = (unsigned)((E + 255) / 256);
    DISPATCH_D(d, hipLaunchKernelGGL((k_euclid_propagate<DD>), dim3(nb), dim3(256), 0, ctx->stream, ctx->Xo, d_src1, E, d_t, d_u, d_s, d_out));
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
