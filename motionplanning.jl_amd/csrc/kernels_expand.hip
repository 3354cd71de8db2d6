// FMT* batch-expand step on the resident r-disc graph (gfx950).
//
// Replaces, for a whole set of expansion nodes z at once, the inner body of the reference's main
// loop (src/planners/fmt.jl:70-82):
//   for x in nearF(V, z, r, W)            -> mark pass   (filter_neighborhood, nearneighbors.jl:104-107)
//       checkpts && !F[x] && continue
//       neighborhood = nearB(V, x, r, H)
//       c_min, y_idx = findmin(C[inds] + ds)   -> argmin pass, first minimum = lowest sample index
//       is_free_motion(V[y_min], V[x], CC, SS) -> bit lookup in the swept graph mask, or an edge sweep
// Integer / bit work only, plus one fp64 add per backward neighbour (C[y] + d(y,x), unfused).
#include "mpfmt_internal.h"
#include <cstring>
#include <algorithm>

__device__ __forceinline__ bool bit_at(const uint64_t* m, int64_t i) { return (m[i >> 6] >> (i & 63)) & 1ull; }

// mark every x in column z (symmetric metric: forward set == column) that is unvisited and valid
__global__ __launch_bounds__(64) void k_expand_mark(const int64_t* __restrict__ zs1, int64_t nz, int64_t N,
                                                    const int64_t* __restrict__ colptr, const int32_t* __restrict__ rowval,
                                                    const uint64_t* __restrict__ W, const uint64_t* __restrict__ F,
                                                    unsigned long long* __restrict__ cand)
{
    const int64_t iz = blockIdx.x;
    if (iz >= nz) return;
    const int64_t z = zs1[iz] - 1;
    if (z < 0 || z >= N) return;
    const int64_t beg = colptr[z], end = colptr[z + 1];
    for (int64_t e = beg + threadIdx.x; e < end; e += 64) {
        const int64_t x = rowval[e];
        if (bit_at(W, x) && (!F || bit_at(F, x))) atomicOr(&cand[x >> 6], 1ull << (x & 63));
    }
}

__global__ void k_popc_words(const unsigned long long* __restrict__ cand, int64_t words, int64_t* __restrict__ cnt)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < words) cnt[w] = __popcll(cand[w]);
    if (w == words) cnt[w] = 0;
}

__global__ void k_emit_xs(const unsigned long long* __restrict__ cand, int64_t words, const int64_t* __restrict__ off,
                          int64_t cap, int64_t* __restrict__ xs0)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    unsigned long long m = cand[w];
    int64_t o = off[w];
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (o < cap) xs0[o] = w * 64 + b;
        ++o;
    }
}

// one wavefront per x: first-minimum of C[y] + d(y,x) over open backward neighbours
__global__ __launch_bounds__(64) void k_expand_argmin(const int64_t* __restrict__ xs0, int64_t nx,
                                                      const int64_t* __restrict__ colptr, const int32_t* __restrict__ rowval,
                                                      const double* __restrict__ nzval, const uint64_t* __restrict__ H,
                                                      const double* __restrict__ C, const uint64_t* __restrict__ gfree,
                                                      int64_t* __restrict__ xs1, int64_t* __restrict__ ymin1,
                                                      double* __restrict__ cmin, uint8_t* __restrict__ freeflag)
{
    const int lane = threadIdx.x;
    for (int64_t ix = blockIdx.x; ix < nx; ix += gridDim.x) {
        const int64_t x = xs0[ix];
        const int64_t beg = colptr[x], end = colptr[x + 1];
        double best = 0.0;
        int64_t be = -1;
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t y = rowval[e];
            if (!bit_at(H, y)) continue;
            const double c = C[y] + nzval[e];
            if (be < 0 || c < best) { best = c; be = e; }       // ascending e per lane: keeps the first minimum
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double oc = __shfl_xor(best, off);
            const int64_t oe = __shfl_xor(be, off);
            const bool take = (oe >= 0) && (be < 0 || oc < best || (oc == best && oe < be));
            if (take) { best = oc; be = oe; }
        }
        if (lane == 0) {
            xs1[ix] = x + 1;
            ymin1[ix] = (be >= 0) ? (int64_t)rowval[be] + 1 : 0;
            cmin[ix] = (be >= 0) ? best : 0.0;
            if (gfree) freeflag[ix] = (be >= 0) ? (uint8_t)bit_at(gfree, be) : 0;
        }
    }
}

__global__ void k_unpack_bits(const uint64_t* __restrict__ mask, const int64_t* __restrict__ ymin1, int64_t n,
                              uint8_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (ymin1[i] > 0) ? (uint8_t)bit_at(mask, i) : 0;
}

// entries without an open neighbour point at sample 1 so the edge sweep has valid indices (flag is forced to 0 after)
__global__ void k_fix_ymin(const int64_t* __restrict__ ymin1, int64_t n, int64_t* __restrict__ src1)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) src1[i] = (ymin1[i] > 0) ? ymin1[i] : 1;
}

int32_t mpfmt_launch_expand(mpfmt_ctx* ctx, const uint64_t* d_W, const uint64_t* d_H, const uint64_t* d_F,
                            const double* d_C, const int64_t* d_zs1, int64_t nz,
                            int64_t* d_xs, int64_t* d_ymin, double* d_cmin, uint8_t* d_free, int64_t cap,
                            int64_t* nx_host)
{
    if (!ctx->graph_filled) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "expand needs a built r-disc graph");
    const int64_t N = ctx->N;
    const int64_t words = (N + 63) / 64;
    *nx_host = 0;
    if (nz == 0 || N == 0) return MPFMT_OK;
    int32_t rc;
    // scratch: cand[words] | cnt[words+1] | off[words+1] | xs0[cap] | src1[cap] | mask[capwords] | scan tmp
    const size_t tmp_bytes = mpfmt_scan_tmp_bytes((size_t)(words + 1));
    const size_t o_cand = 0, o_cnt = o_cand + 8 * words, o_off = o_cnt + 8 * (words + 1), o_xs = o_off + 8 * (words + 1),
                 o_src = o_xs + 8 * cap, o_mask = o_src + 8 * cap, o_tmp = (o_mask + 8 * ((cap + 63) / 64 + 1) + 255) & ~(size_t)255;
    void* scr;
    if ((rc = mpfmt_scratch(ctx, o_tmp + tmp_bytes, &scr))) return rc;
    char* s = (char*)scr;
    unsigned long long* cand = (unsigned long long*)(s + o_cand);
    int64_t* cnt = (int64_t*)(s + o_cnt);
    int64_t* off = (int64_t*)(s + o_off);
    int64_t* xs0 = (int64_t*)(s + o_xs);
    int64_t* src1 = (int64_t*)(s + o_src);
    uint64_t* emask = (uint64_t*)(s + o_mask);

    mpfmt_timed tm1(ctx);
    HIPCHK(ctx, hipMemsetAsync(cand, 0, 8 * words, ctx->stream));
    hipLaunchKernelGGL(k_expand_mark, dim3((unsigned)nz), dim3(64), 0, ctx->stream, d_zs1, nz, N, ctx->colptr, ctx->rowval,
                       d_W, d_F, cand);
    const int B = 256;
    hipLaunchKernelGGL(k_popc_words, dim3((unsigned)((words + 1 + B - 1) / B)), dim3(B), 0, ctx->stream, cand, words, cnt);
    if ((rc = mpfmt_scan_i64_tmp(ctx, cnt, off, (size_t)(words + 1), s + o_tmp))) return rc;
    int64_t nx = 0;
    HIPCHK(ctx, hipMemcpyAsync(&nx, off + words, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nx_host = nx;
    if (nx > cap) return mpfmt_fail(ctx, MPFMT_ERR_CAPACITY, "expand: %lld results exceed capacity %lld", (long long)nx, (long long)cap);
    if (nx > 0) {
        hipLaunchKernelGGL(k_emit_xs, dim3((unsigned)((words + B - 1) / B)), dim3(B), 0, ctx->stream, cand, words, off, cap, xs0);
        const uint64_t* gfree = ctx->graph_swept ? ctx->graph_free : nullptr;
        const unsigned nb = (unsigned)std::min<int64_t>(nx, 1 << 20);
        hipLaunchKernelGGL(k_expand_argmin, dim3(nb), dim3(64), 0, ctx->stream, xs0, nx, ctx->colptr, ctx->rowval, ctx->nzval,
                           d_H, d_C, gfree, d_xs, d_ymin, d_cmin, d_free);
        HIPCHK(ctx, hipGetLastError());
        if (!gfree) {
            hipLaunchKernelGGL(k_fix_ymin, dim3((unsigned)((nx + B - 1) / B)), dim3(B), 0, ctx->stream, d_ymin, nx, src1);
            if ((rc = mpfmt_launch_edges_free(ctx, src1, d_xs, nx, emask))) return rc;
            hipLaunchKernelGGL(k_unpack_bits, dim3((unsigned)((nx + B - 1) / B)), dim3(B), 0, ctx->stream, emask, d_ymin, nx, d_free);
            HIPCHK(ctx, hipGetLastError());
        }
    }
    tm1.end("expand");
    return MPFMT_OK;
}
