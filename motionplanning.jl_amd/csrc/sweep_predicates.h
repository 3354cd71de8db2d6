// Exact fp64 predicates of the AABB checker shared by the sweep kernels and the wavefront FMT* driver (gfx950).
// Every function restates the cited reference line with its operations in its order, unfused (-ffp-contract=off).
#pragma once
#include "mpfmt_internal.h"

#define SWEEP_CHUNK 256          // boxes per LDS stage of the culled kernels (= row stride of the SoA staging)

// ---- exact predicates -----------------------------------------------------------------------------

// in_state_space(v, SS) = @all [lo[i] <= v[i] <= hi[i]]          statespaces.jl:150
template <int D>
__device__ __forceinline__ bool in_state_space(const double (&v)[D], const mpfmt_ss& ss)
{
    if (!ss.has) return true;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < D; ++i) ok = ok && (ss.lo[i] <= v[i]) && (v[i] <= ss.hi[i]);
    return ok;
}

// is_free_state(v, BB) = @any [!(lo[i] <= v[i] <= hi[i])]         boxesND.jl:42
template <int D>
__device__ __forceinline__ bool point_outside_box(const double (&v)[D], const double* lo, const double* hi)
{
    bool out = false;
#pragma unroll
    for (int i = 0; i < D; ++i) out = out || !((lo[i] <= v[i]) && (v[i] <= hi[i]));
    return out;
}

// is_free_motion_broadphase(l, h, BB) = @any [hi[i] < l[i] || lo[i] > h[i]]     boxesND.jl:44-45
template <int D>
__device__ __forceinline__ bool broadphase_free(const double (&l)[D], const double (&h)[D], const double* lo, const double* hi)
{
    bool sep = false;
#pragma unroll
    for (int i = 0; i < D; ++i) sep = sep || (hi[i] < l[i]) || (lo[i] > h[i]);
    return sep;
}

// is_free_motion(v, w, BB)                                         boxesND.jl:46-51
template <int D>
__device__ __forceinline__ bool narrow_free(const double (&v)[D], const double (&w)[D], const double* lo, const double* hi)
{
    double v_to_w[D], lambdas[D];
#pragma unroll
    for (int i = 0; i < D; ++i) v_to_w[i] = w[i] - v[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double corner = (v[i] < lo[i]) ? lo[i] : hi[i];          // blend(v .< lo, lo, hi)
        lambdas[i] = (corner - v[i]) / v_to_w[i];                      // IEEE: may be +-Inf / NaN
    }
    bool hit = false;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j == i) continue;
            const double prod = v_to_w[j] * lambdas[i];
            const double x = v[j] + prod;                              // unfused
            all = all && (lo[j] <= x) && (x <= hi[j]);
        }
        hit = hit || all;
    }
    return !hit;
}

template <int D>
__device__ __forceinline__ void seg_bbox(const double (&v)[D], const double (&w)[D], double (&l)[D], double (&h)[D])
{
#pragma unroll
    for (int i = 0; i < D; ++i) {
        l[i] = (w[i] < v[i]) ? w[i] : v[i];        // map(min, v, w)
        h[i] = (v[i] < w[i]) ? w[i] : v[i];        // map(max, v, w)
    }
}

// ---- straight-line forms (used by the kernels) -----------------------------------------------------------------
// The && / || forms above short-circuit: when an operand is an LDS read the compiler must keep it behind a branch,
// which turns a 2*D-term predicate into 2*D serial LDS round trips.  These take the box in registers and combine the
// IEEE comparisons without control flow (same truth table, including NaN / Inf operands).
template <int D>
struct box_regs { double lo[D], hi[D]; };

template <int D>
__device__ __forceinline__ box_regs<D> load_box(const double* sbox, int k)
{
    box_regs<D> b;
    const double* p = sbox + (int64_t)k * 2 * D;
#pragma unroll
    for (int i = 0; i < D; ++i) { b.lo[i] = p[i]; b.hi[i] = p[D + i]; }
    return b;
}

template <int D>
__device__ __forceinline__ bool in_state_space_sl(const double (&v)[D], const mpfmt_ss& ss)
{
    if (!ss.has) return true;
    int ok = 1;
#pragma unroll
    for (int i = 0; i < D; ++i) ok &= (int)(ss.lo[i] <= v[i]) & (int)(v[i] <= ss.hi[i]);
    return ok != 0;
}

template <int D>
__device__ __forceinline__ bool broadphase_free_sl(const double (&l)[D], const double (&h)[D], const box_regs<D>& b)
{
    int sep = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) sep |= (int)(b.hi[i] < l[i]) | (int)(b.lo[i] > h[i]);
    return sep != 0;
}

// is_free_motion(v, w, BB) (boxesND.jl:46-51): face i is hit iff all 2*(D-1) in-range comparisons of the other
// coordinates hold; they are counted (v_cmp + add-with-carry on the vector ALU) instead of and-ed.
template <int D>
__device__ __forceinline__ bool narrow_free_sl(const double (&v)[D], const double (&w)[D], const box_regs<D>& b)
{
    double v_to_w[D];
#pragma unroll
    for (int i = 0; i < D; ++i) v_to_w[i] = w[i] - v[i];
    int best = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double corner = (v[i] < b.lo[i]) ? b.lo[i] : b.hi[i];        // blend(v .< lo, lo, hi)
        const double lambda = (corner - v[i]) / v_to_w[i];                 // IEEE: may be +-Inf / NaN
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (j == i) continue;
            const double prod = v_to_w[j] * lambda;
            const double x = v[j] + prod;                                  // unfused
            cnt += (int)(b.lo[j] <= x);
            cnt += (int)(x <= b.hi[j]);
        }
        best = max(best, cnt);
    }
    return best != 2 * (D - 1);
}

// box k of the transposed (SoA) staging [2*D][SWEEP_CHUNK] (fixed row stride: one address register, the bound index
// is an immediate offset): uniform k = broadcast reads, per-lane k = gather
template <int D>
__device__ __forceinline__ box_regs<D> load_box_T(const double* sboxT, int k)
{
    box_regs<D> b;
    const double* p = sboxT + k;
#pragma unroll
    for (int i = 0; i < D; ++i) { b.lo[i] = p[i * SWEEP_CHUNK]; b.hi[i] = p[(D + i) * SWEEP_CHUNK]; }
    return b;
}

// Stage boxes [b0, b0+nb) into LDS (whole workgroup), layout [box][2*D].
template <int D>
__device__ __forceinline__ void stage_boxes(double* sbox, const double* __restrict__ boxes, int b0, int nb)
{
    const int n = nb * 2 * D;
    for (int t = threadIdx.x; t < n; t += blockDim.x) sbox[t] = boxes[(int64_t)b0 * 2 * D + t];
}


#define DISPATCH_D(DIM, EXPR)                                                                         \
    switch (DIM) {                                                                                    \
        case 1: { constexpr int DD = 1; EXPR; } break;   case 2: { constexpr int DD = 2; EXPR; } break;   \
        case 3: { constexpr int DD = 3; EXPR; } break;   case 4: { constexpr int DD = 4; EXPR; } break;   \
        case 5: { constexpr int DD = 5; EXPR; } break;   case 6: { constexpr int DD = 6; EXPR; } break;   \
        case 7: { constexpr int DD = 7; EXPR; } break;   case 8: { constexpr int DD = 8; EXPR; } break;   \
        case 9: { constexpr int DD = 9; EXPR; } break;   case 10: { constexpr int DD = 10; EXPR; } break; \
        case 11: { constexpr int DD = 11; EXPR; } break; case 12: { constexpr int DD = 12; EXPR; } break; \
        case 13: { constexpr int DD = 13; EXPR; } break; case 14: { constexpr int DD = 14; EXPR; } break; \
        case 15: { constexpr int DD = 15; EXPR; } break; case 16: { constexpr int DD = 16; EXPR; } break; \
        default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unsupported dimension %d", (int)(DIM));       \
    }
