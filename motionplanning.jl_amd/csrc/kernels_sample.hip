// Batch free-space sampler: sample_free! of src/sampling.jl:11-45 with the rejection loop run in
// batches on the device.  The reference's loop is sequential -- candidates are tested IN ORDER and the first accepted
// ones are kept (sampling.jl:21-36) -- so candidates come from a counter-based generator (Philox4x32-10, counter =
// candidate index): the result does not depend on the batch size and a scalar restatement reproduces it bit for bit.
//   stream:  key = seed; counter = (candidate lo32, candidate hi32, coordinate pair j, stream id);
//            words (x0, x1) -> coordinate 2j, (x2, x3) -> coordinate 2j+1;  u = ((xa >> 5) * 2^26 + (xb >> 6)) * 2^-53;
//   sample_space(SS) = lo + rand .* (hi - lo)  (statespaces.jl:40), unfused;
//   is_free_state(v, CC, SS) through the same point kernel as checkpts (kernels_sweep.hip);
//   ordered compaction: per-64-candidate popcounts -> exclusive scan -> accepted candidate k goes to slot have + k.
#include <cstring>
#include "mpfmt_internal.h"

struct philox_out { uint32_t x[4]; };

__host__ __device__ __forceinline__ philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox_out o;
    o.x[0] = c0; o.x[1] = c1; o.x[2] = c2; o.x[3] = c3;
    return o;
}

__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// u[0..d) of candidate c in stream s
__host__ __device__ __forceinline__ void sample_uniforms(uint64_t seed, uint64_t c, uint32_t stream, int d, double* u)
{
    for (int j = 0; 2 * j < d; ++j) {
        const philox_out o = philox4x32_10((uint32_t)c, (uint32_t)(c >> 32), (uint32_t)j, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[2 * j] = u53(o.x[0], o.x[1]);
        if (2 * j + 1 < d) u[2 * j + 1] = u53(o.x[2], o.x[3]);
    }
}

// lane = candidate c0 + e: P[e][0..d) = lo + u .* (hi - lo)
__global__ __launch_bounds__(256) void k_sample_space(uint64_t seed, uint64_t c0, int64_t B, int d, mpfmt_ss ss, double* __restrict__ P)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    double u[MPFMT_MAX_DIM];
    sample_uniforms(seed, c0 + (uint64_t)e, 0u, d, u);
    for (int i = 0; i < d; ++i) {
        const double w = ss.hi[i] - ss.lo[i];
        const double p = u[i] * w;
        P[e * d + i] = ss.lo[i] + p;
    }
}

__global__ __launch_bounds__(256) void k_popc_words(const uint64_t* __restrict__ mask, int64_t words, int64_t* __restrict__ cnt)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < words) cnt[w] = __popcll(mask[w]);
}

// accepted candidate number k (in order) of this batch -> slot have + k, for k < need; the candidate that fills the
// last slot reports how many candidates the sequential loop would have consumed
__global__ __launch_bounds__(256) void k_sample_compact(const double* __restrict__ P, const uint64_t* __restrict__ mask,
                                                        const int64_t* __restrict__ woff, int64_t B, int d, int64_t have,
                                                        int64_t need, uint64_t c0, double* __restrict__ W,
                                                        long long* __restrict__ attempts)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    const uint64_t m = mask[e >> 6];
    if (!((m >> (e & 63)) & 1ull)) return;
    const int64_t k = woff[e >> 6] + __popcll(m & ((1ull << (e & 63)) - 1ull));
    if (k >= need) return;
    for (int i = 0; i < d; ++i) W[(have + k) * d + i] = P[e * d + i];
    if (k == need - 1) *attempts = (long long)(c0 + (uint64_t)e + 1ull);
}

// sample_goal (goals.jl:97,101-108,115) from stream 1; returns false when a Ball candidate falls outside the ball
static bool goal_candidate(uint64_t seed, uint64_t g, int d, int kind, const double* gp, double* v)
{
    double u[MPFMT_MAX_DIM];
    sample_uniforms(seed, g, 1u, d, u);
    if (kind == MPFMT_GOAL_RECT) {
        for (int i = 0; i < d; ++i) { const double w = gp[d + i] - gp[i]; const double p = w * u[i]; v[i] = gp[i] + p; }
        return true;
    }
    if (kind == MPFMT_GOAL_BALL) {
        double s = 0.0;
        for (int i = 0; i < d; ++i) {
            const double a = 2 * gp[d]; const double b = u[i] - .5; const double p = a * b;
            v[i] = gp[i] + p;
            const double t = v[i] - gp[i]; const double tt = t * t;
            s = (i == 0) ? tt : s + tt;
        }
        return std::sqrt(s) <= gp[d];
    }
    for (int i = 0; i < d; ++i) v[i] = gp[i];
    return true;
}

static int32_t sample_free_impl(mpfmt_ctx* ctx, uint64_t seed, int64_t N, const double* init, int32_t goal_kind,
                                const double* goal_params, int32_t goal_ct, double goal_bias, double* X_out, int64_t* attempts_out);

int32_t mpfmt_sample_free(mpfmt_ctx* ctx, uint64_t seed, int64_t N, const double* init, int32_t goal_kind,
                          const double* goal_params, int32_t goal_ct, double* X_out, int64_t* attempts_out)
{
    return sample_free_impl(ctx, seed, N, init, goal_kind, goal_params, goal_ct, 0.0, X_out, attempts_out);
}

int32_t mpfmt_sample_free_biased(mpfmt_ctx* ctx, uint64_t seed, int64_t N, const double* init, int32_t goal_kind,
                                 const double* goal_params, int32_t goal_ct, double goal_bias, double* X_out, int64_t* attempts_out)
{
    if (ctx && !(goal_bias >= 0.0 && goal_bias <= 1.0)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "goal_bias must lie in [0, 1]");
    if (ctx && goal_bias > 0.0 && (!goal_params || goal_kind < 0 || goal_kind > 2)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "goal_bias needs a goal");
    return sample_free_impl(ctx, seed, N, init, goal_kind, goal_params, goal_ct, goal_bias, X_out, attempts_out);
}

static int32_t sample_free_impl(mpfmt_ctx* ctx, uint64_t seed, int64_t N, const double* init, int32_t goal_kind,
                                const double* goal_params, int32_t goal_ct, double goal_bias, double* X_out, int64_t* attempts_out)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    if (!ctx->ss.has) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "sampling needs the state-space bounds (mpfmt_upload_boxes ss_lo / ss_hi)");
    const int d = ctx->dw;
    if (ctx->ss.d != d) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "device sampling covers Identity state2workspace (state dim %d != workspace dim %d)", ctx->ss.d, d);
    if (N < 1 || N >= ((int64_t)1 << 31) - 64) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "N = %lld out of range", (long long)N);
    if (goal_ct < 0 || (goal_ct > 0 && (!goal_params || goal_kind < 0 || goal_kind > 2)))
        return mpfmt_fail(ctx, MPFMT_ERR_ARG, "bad goal arguments");
    for (int i = 0; i < d; ++i)
        if (!std::isfinite(ctx->ss.lo[i]) || !std::isfinite(ctx->ss.hi[i])) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "state-space bounds must be finite");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    double* W = nullptr;                    // [N][d] on the device
    HIPCHK(ctx, hipMalloc((void**)&W, sizeof(double) * (size_t)N * d));
    struct guard { double* p; ~guard() { if (p) hipFree(p); } } gW{W};
    int64_t have = 0;
    if (init) { HIPCHK(ctx, hipMemcpyAsync(W, init, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream)); have = 1; }
    long long* d_att = nullptr;
    HIPCHK(ctx, hipMalloc((void**)&d_att, sizeof(long long)));
    struct guard2 { long long* p; ~guard2() { if (p) hipFree(p); } } gA{d_att};
    HIPCHK(ctx, hipMemsetAsync(d_att, 0, sizeof(long long), ctx->stream));
    uint64_t c0 = 0;
    mpfmt_timed tm1(ctx);
    while (have < N) {
        const int64_t need = N - have;
        const int64_t B = ((std::max<int64_t>(65536, need + need / 2 + 4096) + 63) / 64) * 64;
        const int64_t words = B / 64;
        void* scr;
        // scratch: candidates | mask | counts | offsets | scan temp
        const size_t temp_bytes = mpfmt_scan_tmp_bytes((size_t)(words + 1));
        const size_t offP = 0, offM = offP + sizeof(double) * (size_t)B * d, offC = offM + sizeof(uint64_t) * (size_t)words,
                     offO = offC + sizeof(int64_t) * (size_t)(words + 1), offT = offO + sizeof(int64_t) * (size_t)(words + 1);
        if ((rc = mpfmt_scratch(ctx, offT + temp_bytes + 256, &scr))) return rc;
        double* P = (double*)((char*)scr + offP);
        uint64_t* mask = (uint64_t*)((char*)scr + offM);
        int64_t* cnt = (int64_t*)((char*)scr + offC);
        int64_t* off = (int64_t*)((char*)scr + offO);
        void* temp = (char*)scr + offT;
        hipLaunchKernelGGL(k_sample_space, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, ctx->stream, seed, c0, B, d, ctx->ss, P);
        HIPCHK(ctx, hipGetLastError());
        if ((rc = mpfmt_launch_states_free(ctx, P, B, mask))) return rc;
        hipLaunchKernelGGL(k_popc_words, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, ctx->stream, mask, words, cnt);
        HIPCHK(ctx, hipMemsetAsync(cnt + words, 0, sizeof(int64_t), ctx->stream));
        if ((rc = mpfmt_scan_i64_tmp(ctx, cnt, off, (size_t)(words + 1), temp))) return rc;
        hipLaunchKernelGGL(k_sample_compact, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, ctx->stream, P, mask, off, B, d, have,
                           need, c0, W, d_att);
        HIPCHK(ctx, hipGetLastError());
        int64_t accepted = 0;
        HIPCHK(ctx, hipMemcpyAsync(&accepted, off + words, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        have += std::min(accepted, need);
        c0 += (uint64_t)B;
        if (accepted == 0 && c0 > (uint64_t)N * 1000ull + (1ull << 24))
            return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "free space appears to be empty: %llu candidates, none accepted", (unsigned long long)c0);
    }
    tm1.end("sample_free");
    long long att = 0;
    HIPCHK(ctx, hipMemcpy(&att, d_att, sizeof(long long), hipMemcpyDeviceToHost));
    if (attempts_out) *attempts_out = (int64_t)att;

    // the set comes to the host once (it is handed to mpfmt_upload_samples below); goal samples are written into that copy
    std::vector<double> host;
    double* H = X_out;
    if (!H) { host.resize((size_t)N * d); H = host.data(); }
    HIPCHK(ctx, hipMemcpy(H, W, sizeof(double) * (size_t)N * d, hipMemcpyDeviceToHost));
    // free goal samples, in the order the sequential loop draws them: candidates from stream 1 (sample_goal, goals.jl:97,
    // 101-108,115), validity on the device in small batches
    uint64_t g = 0;
    const int GB = 256;
    std::vector<double> cand((size_t)GB * d), ready;
    size_t ready_at = 0;
    std::vector<uint64_t> gm((GB + 63) / 64);
    auto next_goal = [&](double* out) -> int32_t {            // sample_free_goal(P), sampling.jl:3-9
        while (ready_at * d >= ready.size()) {
            ready.clear(); ready_at = 0;
            int nc = 0;
            while (nc < GB && g <= 1000000ull) {
                if (goal_candidate(seed, g++, d, goal_kind, goal_params, &cand[(size_t)nc * d])) ++nc;
            }
            if (nc == 0) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "no free goal sample among 1e6 candidates");
            int32_t rc2;
            if ((rc2 = mpfmt_states_free(ctx, cand.data(), nc, gm.data()))) return rc2;
            for (int q = 0; q < nc; ++q)
                if ((gm[q >> 6] >> (q & 63)) & 1ull) ready.insert(ready.end(), &cand[(size_t)q * d], &cand[(size_t)q * d] + d);
        }
        memcpy(out, &ready[ready_at * d], sizeof(double) * d);
        ++ready_at;
        return MPFMT_OK;
    };
    // goal_bias (sampling.jl:28-30): each accepted sample is replaced by a free goal sample with probability goal_bias;
    // the decision for the k-th accepted sample is uniform (seed, k) of stream 2 -- again a function of the sample, not of
    // the batching -- and replacements consume the goal stream in order
    if (goal_bias > 0.0) {
        double u[MPFMT_MAX_DIM];
        for (int64_t k = init ? 1 : 0; k < N; ++k) {
            sample_uniforms(seed, (uint64_t)(k - (init ? 1 : 0)), 2u, 1, u);
            if (u[0] < goal_bias && (rc = next_goal(H + (size_t)k * d))) return rc;
        }
    }
    // goal samples overwrite the tail (sampling.jl:37-41)
    const int64_t ng = std::min<int64_t>(goal_ct, N - 1);
    for (int64_t i = 1; i <= ng; ++i)
        if ((rc = next_goal(H + (size_t)(N - i) * d))) return rc;
    // hand the set to the ctx exactly like mpfmt_upload_samples (bounding box, index invalidation)
    return mpfmt_upload_samples(ctx, H, N, d);
}
