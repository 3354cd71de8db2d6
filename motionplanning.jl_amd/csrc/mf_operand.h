// fp16 MFMA operand of one cell-sorted sample (candidate role), shared by the index build (k_build_tiles writes it while it has the
// row in registers) and the stand-alone k_make_ops.  Layout and error bounds: kernels_rdisc_mfma.hip.
#pragma once
#include "mpfmt_internal.h"

#define MF_PAD_NORM 60000.0f        // |u|^2 stand-in for padding samples: never below any threshold

// x: the sample's d coordinates (ignored when !real: a pad position)
__device__ __forceinline__ void mf_write_operand(void* __restrict__ ops_, const mpfmt_grid& G, int d, double scale, int64_t p, bool real,
                                                 const double* x)
{
    uint4* __restrict__ ops = reinterpret_cast<uint4*>(ops_);
    _Float16 h[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) h[k] = (_Float16)0.0f;
    float n = 0.0f;
    if (real) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            if (i < d) {
                const float u = (float)((x[i] - G.lo[i]) * scale);
                const _Float16 q = (_Float16)u;                 // round to nearest even
                h[i] = q;
                const float qf = (float)q;
                n += qf * qf;                                   // exact products, fp32 sum
            }
        }
    } else {
        n = MF_PAD_NORM;
    }
    const _Float16 nh = (_Float16)n;
    const _Float16 nl = (_Float16)(n - (float)nh);
    if (d <= 6) {
        // K = 8 layout (16 B per sample): u_0..u_5, n_hi, n_lo -- the query's norm and the threshold ride in the MFMA's C
        // stored [chunk][kb][col][half]: lane (kb, col) of the pair kernel's B fragment finds slots 4 kb .. 4 kb + 3 of samples
        // col (half 0) and 32 + col (half 1) of a chunk side by side -- one 16-byte load per lane and chunk
        union { _Float16 hh[8]; uint2 v[2]; } u8;
#pragma unroll
        for (int k = 0; k < 6; ++k) u8.hh[k] = h[k];
        u8.hh[6] = nh; u8.hh[7] = nl;
        uint2* __restrict__ o2 = reinterpret_cast<uint2*>(ops);
        const int64_t chunk = p >> 6;
        const int cx = (int)(p & 31), half = (int)((p >> 5) & 1);
        o2[(chunk * 64 + cx) * 2 + half] = u8.v[0];
        o2[(chunk * 64 + 32 + cx) * 2 + half] = u8.v[1];
        return;
    }
    h[12] = (_Float16)1.0f; h[13] = (_Float16)1.0f; h[14] = nh; h[15] = nl;
    union { _Float16 hh[16]; uint4 v[2]; } u;
#pragma unroll
    for (int k = 0; k < 16; ++k) u.hh[k] = h[k];
    ops[p * 2] = u.v[0];
    ops[p * 2 + 1] = u.v[1];
}
