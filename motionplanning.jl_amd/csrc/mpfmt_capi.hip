// C ABI of libmpfmt.so (see include/mpfmt.h).  Host-side plumbing only: argument checks, device
// buffers for caller-owned host arrays, 1-based <-> 0-based conversion, and the sequential FMT*
// recursion of the reference (src/planners/fmt.jl:68-90) run over GPU-built arrays.
// There is NO CPU fallback for any compute entry point: without a gfx950 device ctx_create fails.
#include "mpfmt_internal.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <chrono>
#include <functional>
#include <algorithm>
#include <vector>

static thread_local std::string g_create_err;

int32_t mpfmt_fail(mpfmt_ctx* ctx, int32_t code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

int32_t mpfmt_scratch(mpfmt_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) HIPCHK(ctx, hipFree(ctx->scratch));
        ctx->scratch = nullptr; ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 4 + 4096;
        HIPCHK(ctx, hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return MPFMT_OK;
}

int32_t mpfmt_ensure(mpfmt_ctx* ctx, void** p, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    auto it = ctx->caps.find((void*)p);
    if (*p && it != ctx->caps.end() && it->second >= bytes) return MPFMT_OK;
    if (*p) { HIPCHK(ctx, hipFree(*p)); *p = nullptr; }
    HIPCHK(ctx, hipMalloc(p, bytes));
    ctx->caps[(void*)p] = bytes;
    return MPFMT_OK;
}

// ---- timing: HIP events on the launch stream.  Begin/end only RECORD events (no host sync in the hot path);
//      pending intervals are resolved when a timing is queried.  A small stack lets groups nest.
#define TIMER_DEPTH 4
struct timer_rec { std::string name; hipEvent_t a, b; };
struct timer_state {
    std::vector<hipEvent_t> free_events;
    std::vector<timer_rec> pending;
    hipEvent_t open_a[TIMER_DEPTH];
    int depth = 0;
};
static timer_state& timers_of(mpfmt_ctx* ctx)
{
    if (!ctx->timer_state) ctx->timer_state = new timer_state();
    return *(timer_state*)ctx->timer_state;
}

static hipEvent_t timer_event(timer_state& t)
{
    if (!t.free_events.empty()) { hipEvent_t e = t.free_events.back(); t.free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

static void timer_resolve(mpfmt_ctx* ctx)
{
    if (!ctx->timer_state) return;
    timer_state& t = timers_of(ctx);
    for (timer_rec& r : t.pending) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            mpfmt_timer& tm = ctx->timers[r.name];
            tm.total_ms += ms;
            tm.launches += 1;
        }
        t.free_events.push_back(r.a);
        t.free_events.push_back(r.b);
    }
    t.pending.clear();
}

int32_t mpfmt_side_fork(mpfmt_ctx* ctx, hipStream_t* main_out)
{
    if (!ctx->side_stream) {
        // (lowest priority: what runs there is never the longer of the two sides)
        int prio_lo = 0, prio_hi = 0;
        HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, prio_lo));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    int32_t rc;
    if ((rc = mpfmt_side_join(ctx))) return rc;               // (one fork at a time)
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    *main_out = ctx->stream;
    ctx->stream = ctx->side_stream;
    return MPFMT_OK;
}

int32_t mpfmt_side_back(mpfmt_ctx* ctx, hipStream_t main)
{
    const hipError_t e = hipEventRecord(ctx->ev_join, ctx->side_stream);
    ctx->stream = main;
    ctx->side_pending = true;
    HIPCHK(ctx, e);
    return MPFMT_OK;
}

int32_t mpfmt_side_join(mpfmt_ctx* ctx)
{
    if (!ctx->side_pending) return MPFMT_OK;
    ctx->side_pending = false;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return MPFMT_OK;
}

void mpfmt_time_begin(mpfmt_ctx* ctx)
{
    if (!ctx->timing_enabled) return;
    timer_state& t = timers_of(ctx);
    if (t.depth < TIMER_DEPTH) {
        t.open_a[t.depth] = timer_event(t);
        hipEventRecord(t.open_a[t.depth], ctx->stream);
    }
    ++t.depth;
}

void mpfmt_time_end(mpfmt_ctx* ctx, const char* name)
{
    if (!ctx->timing_enabled) return;
    timer_state& t = timers_of(ctx);
    if (t.depth <= 0) return;
    --t.depth;
    if (t.depth >= TIMER_DEPTH) return;
    timer_rec r;
    r.name = name; r.a = t.open_a[t.depth]; r.b = timer_event(t);
    hipEventRecord(r.b, ctx->stream);
    t.pending.push_back(r);
    if (t.pending.size() > 4096) timer_resolve(ctx);
}

void mpfmt_time_abandon(mpfmt_ctx* ctx)
{
    if (!ctx->timing_enabled) return;
    timer_state& t = timers_of(ctx);
    if (t.depth <= 0) return;
    --t.depth;
    if (t.depth < TIMER_DEPTH) t.free_events.push_back(t.open_a[t.depth]);
}

// ---- temporaries of one ABI call: device buffers freed on every exit path -----------------------------------
struct DevTmp {
    std::vector<void*> p;
    ~DevTmp() { for (void* q : p) if (q) hipFree(q); }
    template <class T> hipError_t get(T** out, size_t bytes)
    {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
        if (e == hipSuccess) { p.push_back(q); *out = (T*)q; }
        return e;
    }
};

// ---- small conversion kernels --------------------------------------------------------------------------
__global__ void k_add1_i64(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1;
}
__global__ void k_i32_to_i64_add1(const int32_t* __restrict__ in, int64_t n, int64_t* __restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i] + 1;
}

extern "C" {

const char* mpfmt_version(void) { return "mpfmt 0.2.0 gfx950"; }

const char* mpfmt_last_error(const mpfmt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int32_t mpfmt_ctx_create(int32_t device, mpfmt_ctx** out)
{
    if (!out) return mpfmt_fail(nullptr, MPFMT_ERR_ARG, "ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return mpfmt_fail(nullptr, MPFMT_ERR_NODEVICE, "no HIP device visible (%s); libmpfmt has no CPU fallback",
                          e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return mpfmt_fail(nullptr, MPFMT_ERR_ARG, "device %d out of range [0,%d)", device, ndev);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return mpfmt_fail(nullptr, MPFMT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return mpfmt_fail(nullptr, MPFMT_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    if ((e = hipSetDevice(device)) != hipSuccess)
        return mpfmt_fail(nullptr, MPFMT_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    mpfmt_ctx* ctx = new mpfmt_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->ss.has = 0; ctx->ss.d = 0;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        delete ctx;
        return mpfmt_fail(nullptr, MPFMT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return MPFMT_OK;
}

int32_t mpfmt_ctx_destroy(mpfmt_ctx* ctx)
{
    if (!ctx) return MPFMT_OK;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();
    // (refused while another thread's open group still has to post this ctx's gather: the ctx stays whole and usable)
    { const int32_t rc = mpfmt_comm_destroy(ctx); if (rc) return rc; }
    if (ctx->zarena) { hipFree(ctx->zarena); ctx->d_pairs = nullptr; ctx->pool_flag = nullptr; ctx->pair_cnt = nullptr; ctx->qlen = nullptr; }      // (they point into it)
    void* bufs[] = {ctx->Xo, ctx->perm, ctx->iperm, ctx->cellkey, ctx->idx_arena, ctx->Xt, ctx->tile_lo, ctx->tile_hi, ctx->tile_sub, ctx->tile_sub32,
                    ctx->slice_cnt, ctx->deg, ctx->colptr, ctx->rowtmp, ctx->valtmp, ctx->rowval, ctx->nzval,
                    ctx->graph_free, ctx->d_pairs, ctx->boxes, ctx->scratch, ctx->degs, ctx->tptr, ctx->Xs, ctx->ops, ctx->di_ops, ctx->Xo_next, ctx->cellcnt_pad,
                    ctx->tvaltmp, ctx->tval, ctx->di_nseg, ctx->rowpos, ctx->pool_flag, ctx->qkey, ctx->qd2, ctx->qlen, ctx->smask, ctx->st_best, ctx->st_besti, ctx->st_nfree, ctx->pend_items, ctx->pend_cnt, ctx->pair_items, ctx->pair_cnt, ctx->lists, ctx->list_len, ctx->lists_stage, ctx->sweep_ctr, ctx->rt_cnt, ctx->rt_off, ctx->rt_tmp, ctx->rt_table, ctx->rt_total, ctx->rt_ss, ctx->ssflag_dev, ctx->shapes2d, ctx->car_keep, ctx->di_pool_i, ctx->di_pool_c, ctx->di_pool_t, ctx->spec_fail, ctx->rb_dev, ctx->bb_dev};
    if (ctx->rb_host) hipHostFree(ctx->rb_host);
    if (ctx->export_arena) hipHostFree(ctx->export_arena);
    if (ctx->bb_host) hipHostFree(ctx->bb_host);
    if (ctx->side_stream) { hipStreamSynchronize(ctx->side_stream); hipStreamDestroy(ctx->side_stream); hipEventDestroy(ctx->ev_fork); hipEventDestroy(ctx->ev_join); }
    for (int k = 0; k < 2; ++k) { if (ctx->copy_stream[k]) hipStreamDestroy(ctx->copy_stream[k]); if (ctx->ev_conv[k]) hipEventDestroy(ctx->ev_conv[k]); if (ctx->ev_copy[k]) hipEventDestroy(ctx->ev_copy[k]); }
    mpfmt_wf_free(ctx);
    if (ctx->aux) { mpfmt_ctx_destroy(ctx->aux); ctx->aux = nullptr; }
    for (void* b : bufs) if (b) hipFree(b);
    timer_resolve(ctx);
    if (ctx->timer_state) {
        timer_state* t = (timer_state*)ctx->timer_state;
        for (hipEvent_t e : t->free_events) hipEventDestroy(e);
        delete t;
        ctx->timer_state = nullptr;
    }
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MPFMT_OK;
}

int32_t mpfmt_set_stream(mpfmt_ctx* ctx, void* hip_stream)
{
    if (!ctx) return MPFMT_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return MPFMT_OK;
}

int32_t mpfmt_set_shard(mpfmt_ctx* ctx, int32_t rank, int32_t world)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "bad shard %d of %d", rank, world);
    ctx->rank = rank; ctx->world = world;
    ctx->deg_zero_valid = false;
    ctx->grid_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0;      // (the cell order and the built part of the index belong to the shard)
    ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
    return MPFMT_OK;
}

// per-block partial bounding boxes of a device-resident sample set + a count of non-finite coordinates; with dst the same pass is the
// copy into the ctx's sample buffer (thread = one coordinate: coalesced whatever d is; the box per axis needs d | the stride, so a
// thread's coordinates are those of ONE axis when the block's stride is a multiple of d -- the launch rounds it)
#define BBOX_THREADS 1024
__global__ __launch_bounds__(BBOX_THREADS) void k_bbox_partials(const double* __restrict__ X, double* __restrict__ dst, int64_t N, int d, int64_t stride,
                                                                double* __restrict__ part, int32_t* __restrict__ bad)
{
    __shared__ double s_lo[MPFMT_MAX_DIM], s_hi[MPFMT_MAX_DIM];
    __shared__ unsigned long long s_lob[MPFMT_MAX_DIM], s_hib[MPFMT_MAX_DIM];
    (void)s_lo; (void)s_hi;
    // total order of doubles as unsigned integers (sign flip), so that the per-axis minimum / maximum are LDS atomics
    auto enc = [](double a) -> unsigned long long { const unsigned long long u = (unsigned long long)__double_as_longlong(a); return (u >> 63) ? ~u : (u | 0x8000000000000000ull); };
    auto dec = [](unsigned long long e) -> double { const unsigned long long u = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e; return __longlong_as_double((long long)u); };
    if (threadIdx.x < MPFMT_MAX_DIM) { s_lob[threadIdx.x] = ~0ull; s_hib[threadIdx.x] = 0ull; }
    __syncthreads();
    const int64_t total = N * d;
    // stride (a multiple of d) coordinates apart: this thread's coordinates all belong to axis (first index) mod d
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int axis = (int)(first % d);
    double lo = INFINITY, hi = -INFINITY;
    int nbad = 0;
    for (int64_t t = (first < stride) ? first : total; t < total; t += stride) {
        const double a = X[t];
        if (dst) dst[t] = a;
        if (!(fabs(a) <= 1.7976931348623157e308)) ++nbad;
        lo = fmin(lo, a); hi = fmax(hi, a);
    }
    if (lo <= hi) { atomicMin(&s_lob[axis], enc(lo)); atomicMax(&s_hib[axis], enc(hi)); }
    if (__ballot(nbad != 0) && (threadIdx.x & 63) == 0) atomicAdd(bad, 1);
    __syncthreads();
    if (threadIdx.x < d) {
        const int i = threadIdx.x;
        part[((int64_t)blockIdx.x * 2 + 0) * MPFMT_MAX_DIM + i] = (s_lob[i] == ~0ull) ? INFINITY : dec(s_lob[i]);
        part[((int64_t)blockIdx.x * 2 + 1) * MPFMT_MAX_DIM + i] = (s_hib[i] == 0ull) ? -INFINITY : dec(s_hib[i]);
    }
}

// Both uploads: the copy into the ctx's sample buffer (host-to-device or device-to-device) and, beside it on the same stream, the
// finiteness check and the bounding box as ONE reduction on the device (a host loop over 6e6 coordinates costs 3 ms -- more than the
// PCIe copy of them), one small read-back, one synchronisation.  The copy goes to a SECOND buffer that changes places with the ctx's
// sample buffer only when the check has passed: a refused set (non-finite coordinate) leaves the ctx exactly as it was -- previous
// samples, index, graph and hints included (round 5 overwrote first and left the ctx empty: ADVICE r5).
static int32_t adopt_samples(mpfmt_ctx* ctx, const double* src, bool src_on_host, int64_t N, int32_t d)
{
    if (N < 0 || N >= ((int64_t)1 << 31) - 64) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "N = %lld out of range", (long long)N);
    if (d < 1 || d > MPFMT_MAX_DIM) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "d = %d out of range [1,%d]", d, MPFMT_MAX_DIM);
    if (N > 0 && !src) return mpfmt_fail(ctx, MPFMT_ERR_ARG, src_on_host ? "X is NULL" : "dX is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    constexpr int NB = 256;
    struct bb_block { double part[NB][2][MPFMT_MAX_DIM]; int32_t bad; int32_t pad_; };
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->bb_dev, sizeof(bb_block)))) return rc;
    if (!ctx->bb_host) HIPCHK(ctx, hipHostMalloc(&ctx->bb_host, sizeof(bb_block), hipHostMallocDefault));
    bb_block* dev = (bb_block*)ctx->bb_dev;
    // blocks x threads rounded up to a multiple of d: a thread then stays on one axis (k_bbox_partials)
    const int nb = (int)std::min<int64_t>(NB, (N * d + BBOX_THREADS - 1) / BBOX_THREADS);
    const int64_t stride = std::max<int64_t>(((int64_t)std::max(nb, 1) * BBOX_THREADS / d) * d, d);      // (rounded DOWN: every residue below it has a thread)
    double lo[MPFMT_MAX_DIM], hi[MPFMT_MAX_DIM];
    for (int i = 0; i < d; ++i) { lo[i] = 0.0; hi[i] = 0.0; }
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->Xo_next, sizeof(double) * (size_t)N * d))) return rc;
    if (N > 0) {
        HIPCHK(ctx, hipMemsetAsync(&dev->bad, 0, sizeof(int32_t), ctx->stream));
        // host samples: the PCIe copy, then the reduction over the copy; device samples: the reduction IS the copy
        if (src_on_host) HIPCHK(ctx, hipMemcpyAsync(ctx->Xo_next, src, sizeof(double) * (size_t)N * d, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_bbox_partials, dim3(nb), dim3(BBOX_THREADS), 0, ctx->stream, src_on_host ? (const double*)ctx->Xo_next : src, src_on_host ? (double*)nullptr : ctx->Xo_next,
                           N, d, stride, &dev->part[0][0][0], &dev->bad);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->bb_host, dev, sizeof(bb_block), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (N > 0) {
        const bb_block* h = (const bb_block*)ctx->bb_host;
        if (h->bad) {
            if (src_on_host)                                         // (name the sample, as the host check used to)
                for (int64_t p = 0; p < N; ++p)
                    for (int i = 0; i < d; ++i)
                        if (!std::isfinite(src[p * d + i]))
                            return mpfmt_fail(ctx, MPFMT_ERR_ARG, "sample %lld has a non-finite coordinate (the ctx keeps the sample set it had)", (long long)(p + 1));
            return mpfmt_fail(ctx, MPFMT_ERR_ARG, "the sample set has a non-finite coordinate (the ctx keeps the sample set it had)");
        }
    }
    // accepted: the new buffer becomes the ctx's sample set (the two members change places, capacities with them)
    {
        std::swap(ctx->Xo, ctx->Xo_next);
        const size_t ca = ctx->caps[(void*)&ctx->Xo], cb = ctx->caps[(void*)&ctx->Xo_next];
        ctx->caps[(void*)&ctx->Xo] = cb; ctx->caps[(void*)&ctx->Xo_next] = ca;
    }
    ctx->samples_epoch += 1;
    ctx->grid_r = -1.0; ctx->graph_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0;
    ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
    ctx->di_counted = ctx->di_filled = ctx->di_swept = false;
    ctx->nnz = 0;
    if (N > 0) {
        const bb_block* h = (const bb_block*)ctx->bb_host;
        for (int i = 0; i < d; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; }
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < d; ++i) { lo[i] = std::min(lo[i], h->part[b][0][i]); hi[i] = std::max(hi[i], h->part[b][1][i]); }
    }
    for (int i = 0; i < d; ++i) { ctx->bb_lo[i] = lo[i]; ctx->bb_hi[i] = hi[i]; }
    ctx->N = N; ctx->d = d;
    return MPFMT_OK;
}

int32_t mpfmt_upload_samples(mpfmt_ctx* ctx, const double* X, int64_t N, int32_t d)
{
    if (!ctx) return MPFMT_ERR_ARG;
    return adopt_samples(ctx, X, true, N, d);
}

// The same for a sample set that already lives in HBM (a batch made on the device: the library's own sampler, a ROCArray, a torch
// tensor): one device-to-device copy instead of the PCIe one.
int32_t mpfmt_upload_samples_device(mpfmt_ctx* ctx, const double* dX, int64_t N, int32_t d)
{
    if (!ctx) return MPFMT_ERR_ARG;
    return adopt_samples(ctx, dX, false, N, d);
}

// state-space bounds on their own: the BoundedStateSpace lo / hi for any state dimension (src/statespaces.jl:29-34).  The 2-D SAT
// world's upload takes the two workspace bounds only; a steering space over it (double integrator R^4, SE2 cars) sets its own.
int32_t mpfmt_set_state_bounds(mpfmt_ctx* ctx, const double* ss_lo, const double* ss_hi, int32_t d_state)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if ((ss_lo == nullptr) != (ss_hi == nullptr)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "ss_lo / ss_hi must both be given or both NULL");
    if (ss_lo && (d_state < 1 || d_state > MPFMT_MAX_DIM)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "d_state out of range");
    ctx->ss.has = ss_lo ? 1 : 0;
    ctx->ss.d = ss_lo ? d_state : 0;
    for (int i = 0; i < MPFMT_MAX_DIM; ++i) { ctx->ss.lo[i] = -INFINITY; ctx->ss.hi[i] = INFINITY; }
    if (ss_lo) for (int i = 0; i < d_state; ++i) { ctx->ss.lo[i] = ss_lo[i]; ctx->ss.hi[i] = ss_hi[i]; }
    ctx->graph_swept = false; ctx->di_swept = false; ctx->pend_valid = false;     // (a pending list belongs to the obstacle set it was made against)
    return MPFMT_OK;
}

int32_t mpfmt_upload_boxes(mpfmt_ctx* ctx, const double* lohi, int32_t M, int32_t dw,
                           const double* ss_lo, const double* ss_hi, int32_t d_state)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (M < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "M = %d < 0", M);
    if (dw < 1 || dw > MPFMT_MAX_DIM) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "dw = %d out of range", dw);
    if (M > 0 && !lohi) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "lohi is NULL");
    if ((ss_lo == nullptr) != (ss_hi == nullptr)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "ss_lo / ss_hi must both be given or both NULL");
    if (ss_lo && (d_state < 1 || d_state > MPFMT_MAX_DIM)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "d_state = %d out of range", d_state);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->boxes, sizeof(double) * (size_t)M * 2 * dw))) return rc;
    if (M > 0) HIPCHK(ctx, hipMemcpyAsync(ctx->boxes, lohi, sizeof(double) * (size_t)M * 2 * dw, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->M = M; ctx->dw = dw; ctx->have_boxes = true; ctx->cc_kind = 0;
    ctx->boxes_host.assign(lohi, lohi + (size_t)M * 2 * dw);      // (a host copy: sizes the pending-pair list from the obstacles' extents)
    ctx->ss.has = ss_lo ? 1 : 0;
    ctx->ss.d = ss_lo ? d_state : 0;
    for (int i = 0; i < MPFMT_MAX_DIM; ++i) { ctx->ss.lo[i] = -INFINITY; ctx->ss.hi[i] = INFINITY; }
    if (ss_lo) for (int i = 0; i < d_state; ++i) { ctx->ss.lo[i] = ss_lo[i]; ctx->ss.hi[i] = ss_hi[i]; }
    ctx->graph_swept = false; ctx->pend_valid = false;
    ctx->di_swept = false;
    return MPFMT_OK;
}

// ---- r-disc graph ------------------------------------------------------------------------------------

int32_t mpfmt_graph_build_device(mpfmt_ctx* ctx, double r, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    ctx->di_counted = ctx->di_filled = ctx->di_swept = false;
    if (ctx->rebuild_index) { ctx->grid_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0; }    // index build (cell grid + operands) is part of the build
    if ((rc = mpfmt_launch_rdisc_count(ctx, r))) return rc;
    if ((rc = mpfmt_launch_rdisc_fill(ctx, r))) return rc;
    if (nnz) *nnz = ctx->nnz;
    return MPFMT_OK;
}

int32_t mpfmt_graph_step_device(mpfmt_ctx* ctx, double r, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    ctx->di_counted = ctx->di_filled = ctx->di_swept = false;
    if (ctx->rebuild_index) { ctx->grid_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0; }
    if ((rc = mpfmt_graph_step(ctx, r))) return rc;
    // results are complete on return (the speculative path has synchronised after its last launch already; the careful
    // path launched the sweep after its last read-back)
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (nnz) *nnz = ctx->nnz;
    return MPFMT_OK;
}

int32_t mpfmt_graph_step_launch(mpfmt_ctx* ctx, double r)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->di_counted = ctx->di_filled = ctx->di_swept = false;
    if (ctx->rebuild_index) { ctx->grid_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0; }
    return mpfmt_graph_step_launch_impl(ctx, r);
}

int32_t mpfmt_graph_step_finish(mpfmt_ctx* ctx, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = mpfmt_graph_step_finish_impl(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (nnz) *nnz = ctx->nnz;
    return MPFMT_OK;
}

int32_t mpfmt_rdisc_count(mpfmt_ctx* ctx, double r, int64_t* colptr, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!colptr || !nnz) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr / nnz is NULL");
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    ctx->di_counted = ctx->di_filled = ctx->di_swept = false;
    if ((rc = mpfmt_launch_rdisc_count(ctx, r))) return rc;
    const int64_t n1 = ctx->N + 1;
    void* scr;
    if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * n1, &scr))) return rc;
    hipLaunchKernelGGL(k_add1_i64, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->colptr, n1, (int64_t*)scr);
    HIPCHK(ctx, hipMemcpyAsync(colptr, scr, sizeof(int64_t) * n1, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nnz = ctx->nnz;
    return MPFMT_OK;
}

int32_t mpfmt_rdisc_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->graph_counted) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "rdisc_fill before rdisc_count");
    if (ctx->nnz > 0 && (!rowval || !nzval)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rowval / nzval is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if (!ctx->graph_filled && (rc = mpfmt_launch_rdisc_fill(ctx, ctx->graph_r))) return rc;
    const int64_t nnz = ctx->nnz;
    if (nnz > 0) {
        // convert in slabs so the staging buffer stays small
        const int64_t slab = std::min<int64_t>(nnz, (int64_t)1 << 26);
        void* scr;
        if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * slab, &scr))) return rc;
        for (int64_t o = 0; o < nnz; o += slab) {
            const int64_t n = std::min(slab, nnz - o);
            hipLaunchKernelGGL(k_i32_to_i64_add1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->rowval + o, n, (int64_t*)scr);
            HIPCHK(ctx, hipMemcpyAsync(rowval + o, scr, sizeof(int64_t) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        HIPCHK(ctx, hipMemcpyAsync(nzval, ctx->nzval, sizeof(double) * nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

// Install a previously exported r-disc graph (the reference's ImmutableNNC(D, r), nearneighbors.jl:23-28,
// whose saveNN / loadNN! were left commented out, :114-116) so that re-plans on the same samples -- new obstacles, new
// goals, another process -- skip the pair phase.  Same format mpfmt_rdisc_count / _fill hand out.
int32_t mpfmt_graph_import(mpfmt_ctx* ctx, double r, const int64_t* colptr, const int64_t* rowval, const double* nzval)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (ctx->world != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "graph_import needs an unsharded ctx");
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (!colptr) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr is NULL");
    const int64_t N = ctx->N;
    if (colptr[0] != 1) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr[1] must be 1 (1-based CSC)");
    for (int64_t j = 0; j < N; ++j)
        if (colptr[j + 1] < colptr[j]) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr decreases at column %lld", (long long)(j + 1));
    const int64_t nnz = colptr[N] - 1;
    if (nnz > 0 && (!rowval || !nzval)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rowval / nzval is NULL");
    char verr[160];
    if (mpfmt_validate_csc(N, colptr, rowval, verr, sizeof verr) != 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "%s", verr);
    std::vector<int64_t> cp0((size_t)N + 1);
    std::vector<int32_t> rv0((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t j = 0; j <= N; ++j) cp0[j] = colptr[j] - 1;
    for (int64_t e = 0; e < nnz; ++e) rv0[e] = (int32_t)(rowval[e] - 1);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->colptr, sizeof(int64_t) * (size_t)(N + 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rowval, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->nzval, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(ctx->colptr, cp0.data(), sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->rowval, rv0.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->nzval, nzval, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nnz = nnz;
    ctx->graph_r = r;
    ctx->graph_counted = ctx->graph_filled = true;
    ctx->graph_swept = false; ctx->pend_valid = false;
    ctx->pool_valid = false;
    ctx->rowpos_valid = false;
    ctx->di_counted = ctx->di_filled = ctx->di_swept = false;
    return MPFMT_OK;
}

int32_t mpfmt_rdisc_query(mpfmt_ctx* ctx, int64_t v, double r, int64_t* inds, double* ds, int64_t cap, int64_t* k)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!k) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "k is NULL");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (v < 1 || v > ctx->N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "query index %lld out of range [1,%lld]", (long long)v, (long long)ctx->N);
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (cap < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "cap < 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc = mpfmt_launch_rdisc_query(ctx, v - 1, r, k, inds, ds, cap);
    if (rc == MPFMT_ERR_CAPACITY) return mpfmt_fail(ctx, rc, "rdisc_query: %lld neighbours exceed capacity %lld", (long long)*k, (long long)cap);
    return rc;
}

// ---- validity sweeps ---------------------------------------------------------------------------------

static int32_t up_i64(mpfmt_ctx* ctx, DevTmp& tmp, const int64_t* h, int64_t n, int64_t** d)
{
    *d = nullptr;
    HIPCHK(ctx, tmp.get(d, sizeof(int64_t) * (size_t)std::max<int64_t>(n, 1)));
    if (n > 0) HIPCHK(ctx, hipMemcpyAsync(*d, h, sizeof(int64_t) * n, hipMemcpyHostToDevice, ctx->stream));
    return MPFMT_OK;
}

static int32_t check_idx(mpfmt_ctx* ctx, const int64_t* idx, int64_t n, const char* what)
{
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 1 || idx[i] > ctx->N)
            return mpfmt_fail(ctx, MPFMT_ERR_ARG, "%s[%lld] = %lld out of range [1,%lld]", what, (long long)i, (long long)idx[i], (long long)ctx->N);
    return MPFMT_OK;
}

int32_t mpfmt_points_free(mpfmt_ctx* ctx, const int64_t* idx, int64_t n, uint64_t* mask)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (!idx) n = ctx->N;
    if (n < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n < 0");
    if (n > 0 && !mask) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "mask is NULL");
    if (n == 0) return MPFMT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if (idx && (rc = check_idx(ctx, idx, n, "idx"))) return rc;
    const int64_t words = (n + 63) / 64;
    DevTmp tmp;
    int64_t* d_idx = nullptr; uint64_t* d_mask = nullptr;
    if (idx && (rc = up_i64(ctx, tmp, idx, n, &d_idx))) return rc;
    HIPCHK(ctx, tmp.get(&d_mask, sizeof(uint64_t) * words));
    rc = mpfmt_launch_points_free(ctx, d_idx, n, d_mask);
    if (rc == MPFMT_OK) {
        hipError_t e = hipMemcpyAsync(mask, d_mask, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mpfmt_fail(ctx, MPFMT_ERR_HIP, "points_free copy back: %s", hipGetErrorString(e));
    }
    return rc;
}

int32_t mpfmt_edges_free(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, uint64_t* mask)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (E < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "E < 0");
    if (E == 0) return MPFMT_OK;
    if (!src || !dst || !mask) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "src / dst / mask is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, src, E, "src")) || (rc = check_idx(ctx, dst, E, "dst"))) return rc;
    const int64_t words = (E + 63) / 64;
    DevTmp tmp;
    int64_t *d_s = nullptr, *d_t = nullptr; uint64_t* d_mask = nullptr;
    if ((rc = up_i64(ctx, tmp, src, E, &d_s))) return rc;
    if ((rc = up_i64(ctx, tmp, dst, E, &d_t))) return rc;
    HIPCHK(ctx, tmp.get(&d_mask, sizeof(uint64_t) * words));
    rc = mpfmt_launch_edges_free(ctx, d_s, d_t, E, d_mask);
    if (rc == MPFMT_OK) {
        hipError_t e = hipMemcpyAsync(mask, d_mask, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mpfmt_fail(ctx, MPFMT_ERR_HIP, "edges_free copy back: %s", hipGetErrorString(e));
    }
    return rc;
}

int32_t mpfmt_mc_edges_collision(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double sigma, int64_t rollouts,
                                 uint64_t seed, int64_t* hits)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (E < 0 || E >= ((int64_t)1 << 32)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "E out of range");
    if (rollouts < 0 || rollouts >= ((int64_t)1 << 32)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rollouts out of range [0, 2^32)");
    if (!(sigma >= 0.0) || !std::isfinite(sigma)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "sigma must be finite and >= 0");
    if (E == 0) return MPFMT_OK;
    if (!src || !dst || !hits) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "src / dst / hits is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, src, E, "src")) || (rc = check_idx(ctx, dst, E, "dst"))) return rc;
    DevTmp tmp;
    int64_t *d_s = nullptr, *d_t = nullptr; unsigned long long* d_h = nullptr;
    if ((rc = up_i64(ctx, tmp, src, E, &d_s))) return rc;
    if ((rc = up_i64(ctx, tmp, dst, E, &d_t))) return rc;
    HIPCHK(ctx, tmp.get(&d_h, sizeof(unsigned long long) * E));
    HIPCHK(ctx, hipMemsetAsync(d_h, 0, sizeof(unsigned long long) * E, ctx->stream));
    if ((rc = mpfmt_launch_mc_edges(ctx, d_s, d_t, E, sigma, rollouts, seed, d_h))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(hits, d_h, sizeof(int64_t) * E, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

int32_t mpfmt_mc_edges_collision_is(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double sigma, int64_t rollouts,
                                    uint64_t seed, uint64_t* wsum)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (E < 0 || E >= ((int64_t)1 << 32)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "E out of range");
    if (rollouts < 0 || rollouts >= ((int64_t)1 << 22)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rollouts out of range [0, 2^22): the weights are summed in 64 bits at 2^-40");
    if (!(sigma >= 0.0) || !std::isfinite(sigma)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "sigma must be finite and >= 0");
    if (E == 0) return MPFMT_OK;
    if (!src || !dst || !wsum) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "src / dst / wsum is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, src, E, "src")) || (rc = check_idx(ctx, dst, E, "dst"))) return rc;
    DevTmp tmp;
    int64_t *d_s = nullptr, *d_t = nullptr; unsigned long long* d_h = nullptr;
    if ((rc = up_i64(ctx, tmp, src, E, &d_s))) return rc;
    if ((rc = up_i64(ctx, tmp, dst, E, &d_t))) return rc;
    HIPCHK(ctx, tmp.get(&d_h, sizeof(unsigned long long) * E));
    HIPCHK(ctx, hipMemsetAsync(d_h, 0, sizeof(unsigned long long) * E, ctx->stream));
    if ((rc = mpfmt_launch_mc_is_edges(ctx, d_s, d_t, E, sigma, rollouts, seed, d_h))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(wsum, d_h, sizeof(uint64_t) * E, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

// the ADAPTIVE estimator: pilot (4096 rollouts per edge, inflated noise) -> cross-entropy mean shift per edge -> mixture run
int32_t mpfmt_mc_edges_collision_ais(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double sigma, int64_t rollouts,
                                     uint64_t seed, uint64_t* wsum, double* shifts)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (E < 0 || E >= ((int64_t)1 << 32)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "E out of range");
    if (rollouts < 0 || rollouts >= ((int64_t)1 << 22)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rollouts out of range [0, 2^22): the weights are summed in 64 bits at 2^-40");
    if (!(sigma >= 0.0) || !std::isfinite(sigma)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "sigma must be finite and >= 0");
    if (E == 0) return MPFMT_OK;
    if (!src || !dst || !wsum) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "src / dst / wsum is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, src, E, "src")) || (rc = check_idx(ctx, dst, E, "dst"))) return rc;
    DevTmp tmp;
    int64_t *d_s = nullptr, *d_t = nullptr; unsigned long long* d_h = nullptr; double* d_mu = nullptr;
    if ((rc = up_i64(ctx, tmp, src, E, &d_s))) return rc;
    if ((rc = up_i64(ctx, tmp, dst, E, &d_t))) return rc;
    HIPCHK(ctx, tmp.get(&d_h, sizeof(unsigned long long) * E));
    HIPCHK(ctx, tmp.get(&d_mu, sizeof(double) * (size_t)E * 2 * MPFMT_MAX_DIM));
    HIPCHK(ctx, hipMemsetAsync(d_h, 0, sizeof(unsigned long long) * E, ctx->stream));
    if ((rc = mpfmt_launch_mc_ais_edges(ctx, d_s, d_t, E, sigma, rollouts, seed, d_h, d_mu))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(wsum, d_h, sizeof(uint64_t) * E, hipMemcpyDeviceToHost, ctx->stream));
    if (shifts)       // [E][2 d] out of the [E][2 MPFMT_MAX_DIM] device rows
        HIPCHK(ctx, hipMemcpy2DAsync(shifts, sizeof(double) * 2 * ctx->d, d_mu, sizeof(double) * 2 * MPFMT_MAX_DIM, sizeof(double) * 2 * ctx->d, (size_t)E,
                                     hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

static int32_t explicit_sweep(mpfmt_ctx* ctx, const double* P, const double* Q, int64_t n, uint64_t* mask)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (n < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n < 0");
    if (n == 0) return MPFMT_OK;
    if (!P || !mask) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "P / mask is NULL");
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int d = ctx->dw;
    const int64_t words = (n + 63) / 64;
    const size_t pb = sizeof(double) * (size_t)n * d;
    DevTmp tmp;
    double *dP = nullptr, *dQ = nullptr; uint64_t* d_mask = nullptr;
    HIPCHK(ctx, tmp.get(&dP, pb));
    HIPCHK(ctx, hipMemcpyAsync(dP, P, pb, hipMemcpyHostToDevice, ctx->stream));
    if (Q) {
        HIPCHK(ctx, tmp.get(&dQ, pb));
        HIPCHK(ctx, hipMemcpyAsync(dQ, Q, pb, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, tmp.get(&d_mask, sizeof(uint64_t) * words));
    int32_t rc = Q ? mpfmt_launch_motions_free(ctx, dP, dQ, n, d_mask) : mpfmt_launch_states_free(ctx, dP, n, d_mask);
    if (rc == MPFMT_OK) {
        hipError_t e = hipMemcpyAsync(mask, d_mask, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mpfmt_fail(ctx, MPFMT_ERR_HIP, "sweep copy back: %s", hipGetErrorString(e));
    }
    return rc;
}

int32_t mpfmt_states_free(mpfmt_ctx* ctx, const double* P, int64_t n, uint64_t* mask) { return explicit_sweep(ctx, P, nullptr, n, mask); }

int32_t mpfmt_motions_free(mpfmt_ctx* ctx, const double* P, const double* Q, int64_t n, uint64_t* mask)
{
    if (ctx && n > 0 && !Q) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "Q is NULL");
    return explicit_sweep(ctx, P, Q, n, mask);
}

int32_t mpfmt_path_free(mpfmt_ctx* ctx, const double* P, int64_t n, int32_t* free_out, uint64_t* seg_mask)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!free_out) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "free_out is NULL");
    if (n < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n < 0");
    *free_out = 1;
    if (n < 2) return MPFMT_OK;
    if (!P) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "P is NULL");
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    const int d = ctx->dw;
    const int64_t ns = n - 1, words = (ns + 63) / 64;
    std::vector<uint64_t> m((size_t)words, 0);
    int32_t rc;
    if ((rc = explicit_sweep(ctx, P, P + d, ns, m.data()))) return rc;       // segment i = (p[i], p[i+1]): the same array, one state on
    for (int64_t i = 0; i < ns; ++i) if (!((m[i >> 6] >> (i & 63)) & 1ull)) { *free_out = 0; break; }
    if (seg_mask) memcpy(seg_mask, m.data(), sizeof(uint64_t) * (size_t)words);
    return MPFMT_OK;
}

int32_t mpfmt_graph_sweep_device(mpfmt_ctx* ctx)
{
    if (!ctx) return MPFMT_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->pend_valid = false;                                 // (outside a step: the whole sweep -- nobody checks a pending list's overflow flag here)
    return mpfmt_launch_graph_sweep(ctx);
}

int32_t mpfmt_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask)
{
    if (!ctx) return MPFMT_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if (!ctx->graph_counted) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "graph_edges_free before rdisc_count");
    if (!ctx->graph_filled && (rc = mpfmt_launch_rdisc_fill(ctx, ctx->graph_r))) return rc;
    ctx->pend_valid = false;
    if ((rc = mpfmt_launch_graph_sweep(ctx))) return rc;
    const int64_t words = (ctx->nnz + 63) / 64;
    if (words > 0) {
        if (!mask) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "mask is NULL");
        HIPCHK(ctx, hipMemcpyAsync(mask, ctx->graph_free, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

// per-column reductions of the r-disc graph without storing it (kernels_rdisc_mfma.hip, MODE 3)
int32_t mpfmt_rdisc_stream(mpfmt_ctx* ctx, double r, const double* C, const uint64_t* H, int32_t want_free,
                           int64_t* deg, int64_t* free_deg, int64_t* parent, double* cost, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo && ctx->N > 0) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (!(r >= 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and >= 0");
    if (H && !C) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "an open set without costs");
    if (C && (!parent || !cost)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "parent / cost is NULL");
    if (want_free && !free_deg) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "free_deg is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return mpfmt_rdisc_stream_impl(ctx, r, C, H, want_free, deg, free_deg, parent, cost, nnz);
}

// ---- the resident graph + mask to the host in the ABI's format, at link speed ----------------------------------------------------------
// mpfmt_rdisc_count / _fill / mpfmt_graph_edges_free rebuild or re-sweep what a step has already left in HBM, and copy slab by slab with
// a synchronisation each.  This call hands out what mpfmt_graph_step_device left resident -- colptr and rowval as 1-based Int64, nzval,
// the free mask (BitVector chunks) -- with the index conversion on the compute stream two slabs ahead of two copy streams, so the link
// is the only thing waited for when the destinations are page-locked (mpfmt_pinned_alloc).
int32_t mpfmt_pinned_alloc(int64_t bytes, void** out)
{
    if (!out || bytes < 0) return MPFMT_ERR_ARG;
    *out = nullptr;
    if (bytes == 0) return MPFMT_OK;
    return hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault) == hipSuccess ? MPFMT_OK : MPFMT_ERR_HIP;
}

int32_t mpfmt_pinned_free(void* p)
{
    if (!p) return MPFMT_OK;
    return hipHostFree(p) == hipSuccess ? MPFMT_OK : MPFMT_ERR_HIP;
}

int32_t mpfmt_graph_export(mpfmt_ctx* ctx, int64_t* colptr, int64_t* rowval, double* nzval, uint64_t* mask, double* gb_per_s)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->graph_filled) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no resident graph (mpfmt_graph_step_device / mpfmt_graph_build_device)");
    if (mask && !ctx->graph_swept) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the resident graph has not been swept");
    const int64_t N = ctx->N, nnz = ctx->nnz;
    if (!colptr || (nnz > 0 && (!rowval || !nzval))) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr / rowval / nzval is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream[0]) {
        for (int k = 0; k < 2; ++k) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream[k], hipStreamNonBlocking));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_conv[k], hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_copy[k], hipEventDisableTiming));
        }
    }
    int32_t rc;
    const int64_t slab = (int64_t)1 << 24;                    // entries per conversion slab: 128 MB of Int64, two in flight
    void* scr;
    if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * (size_t)(2 * slab + N + 1), &scr))) return rc;
    int64_t* stg[2] = {(int64_t*)scr, (int64_t*)scr + slab};
    int64_t* cp1 = (int64_t*)scr + 2 * slab;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));            // the step's kernels are done: the clock below sees the export alone
    const auto t0 = std::chrono::steady_clock::now();
    // stream B: nzval, mask; stream A: colptr, rowval slabs as they are converted
    hipLaunchKernelGGL(k_add1_i64, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->colptr, N + 1, cp1);
    HIPCHK(ctx, hipEventRecord(ctx->ev_conv[0], ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream[0], ctx->ev_conv[0], 0));
    HIPCHK(ctx, hipMemcpyAsync(colptr, cp1, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyDeviceToHost, ctx->copy_stream[0]));
    if (nnz > 0) {
        const int64_t half = nnz / 2;
        // (the distances in two pieces, one per copy stream, so both engines stay busy while the row indices are converted)
        HIPCHK(ctx, hipMemcpyAsync(nzval, ctx->nzval, sizeof(double) * (size_t)half, hipMemcpyDeviceToHost, ctx->copy_stream[1]));
        int k = 0;
        for (int64_t o = 0; o < nnz; o += slab, ++k) {
            const int64_t n = std::min(slab, nnz - o);
            const int b = k & 1;
            if (k >= 2) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy[b], 0));      // the slab's staging buffer has been copied out
            hipLaunchKernelGGL(k_i32_to_i64_add1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rowval + o, n, stg[b]);
            HIPCHK(ctx, hipEventRecord(ctx->ev_conv[b], ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream[0], ctx->ev_conv[b], 0));
            HIPCHK(ctx, hipMemcpyAsync(rowval + o, stg[b], sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->copy_stream[0]));
            HIPCHK(ctx, hipEventRecord(ctx->ev_copy[b], ctx->copy_stream[0]));
        }
        HIPCHK(ctx, hipMemcpyAsync(nzval + half, ctx->nzval + half, sizeof(double) * (size_t)(nnz - half), hipMemcpyDeviceToHost, ctx->copy_stream[1]));
        if (mask) HIPCHK(ctx, hipMemcpyAsync(mask, ctx->graph_free, sizeof(uint64_t) * (size_t)((nnz + 63) / 64), hipMemcpyDeviceToHost, ctx->copy_stream[1]));
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream[0]));
    HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream[1]));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double bytes = 8.0 * (double)(N + 1) + 16.0 * (double)nnz + (mask ? (double)((nnz + 63) / 64) * 8.0 : 0.0);
    if (gb_per_s) *gb_per_s = sec > 0.0 ? bytes / sec / 1e9 : 0.0;
    return MPFMT_OK;
}

// Page-locked export arena that lives as long as the ctx (grow-only, freed by mpfmt_ctx_destroy): page-locking gigabytes costs
// ~0.2 s per GB (hipHostMalloc), far more than the copy it speeds up -- paid once per ctx here, not once per graph.
int32_t mpfmt_export_arena(mpfmt_ctx* ctx, int64_t bytes, void** out)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!out || bytes < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "mpfmt_export_arena: bad arguments");
    *out = nullptr;
    if ((size_t)bytes > ctx->export_arena_bytes) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if (ctx->export_arena) { HIPCHK(ctx, hipHostFree(ctx->export_arena)); ctx->export_arena = nullptr; ctx->export_arena_bytes = 0; }
        // (an eighth of slack: the next graph of the same problem -- new samples, new obstacles -- finds its room in place)
        const size_t want = (size_t)bytes + (size_t)bytes / 8 + 4096;
        HIPCHK(ctx, hipHostMalloc(&ctx->export_arena, want, hipHostMallocDefault));
        ctx->export_arena_bytes = want;
    }
    *out = ctx->export_arena;
    return MPFMT_OK;
}

// mpfmt_graph_export into the ctx's own arena: the four arrays are carved out of it (64-byte aligned) and stay valid until the next
// export of this ctx or its destruction -- the drop-in precompute! of julia/MPFmtHIP.jl wraps them without a copy.
int32_t mpfmt_graph_export_pinned(mpfmt_ctx* ctx, int64_t** colptr, int64_t** rowval, double** nzval, uint64_t** mask, int64_t* nnz_out, double* gb_per_s)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!colptr || !rowval || !nzval) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr / rowval / nzval is NULL");
    if (!ctx->graph_filled) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no resident graph (mpfmt_graph_step_device / mpfmt_graph_build_device)");
    // (every state error of mpfmt_graph_export is raised BEFORE the arena is touched: growing it frees the block the previous export's
    // pointers -- still held by the caller on a failed call -- point into)
    if (mask && !ctx->graph_swept) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the resident graph has not been swept");
    const int64_t N = ctx->N, nnz = ctx->nnz, words = (nnz + 63) / 64;
    auto al = [](size_t b) { return (b + 63) & ~(size_t)63; };
    const size_t o_cp = 0, o_rv = al(8 * (size_t)(N + 1)), o_nz = o_rv + al(8 * (size_t)std::max<int64_t>(nnz, 1));
    const size_t o_mk = o_nz + al(8 * (size_t)std::max<int64_t>(nnz, 1)), total = o_mk + al(8 * (size_t)std::max<int64_t>(words, 1));
    void* base = nullptr;
    int32_t rc;
    if ((rc = mpfmt_export_arena(ctx, (int64_t)total, &base))) return rc;
    char* b = (char*)base;
    *colptr = (int64_t*)(b + o_cp); *rowval = (int64_t*)(b + o_rv); *nzval = (double*)(b + o_nz);
    if (mask) *mask = (uint64_t*)(b + o_mk);
    if (nnz_out) *nnz_out = nnz;
    return mpfmt_graph_export(ctx, *colptr, *rowval, *nzval, mask ? *mask : nullptr, gb_per_s);
}

int32_t mpfmt_graph_device_ptrs(mpfmt_ctx* ctx, void** colptr, void** rowval, void** nzval, void** free_mask)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->graph_filled) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no resident graph");
    if (colptr) *colptr = ctx->colptr;
    if (rowval) *rowval = ctx->rowval;
    if (nzval) *nzval = ctx->nzval;
    if (free_mask) *free_mask = ctx->graph_swept ? ctx->graph_free : nullptr;
    return MPFMT_OK;
}

int32_t mpfmt_shard_info(mpfmt_ctx* ctx, int64_t* col_begin, int64_t* col_end, int64_t* shard_nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->graph_counted) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no graph counted");
    if (col_begin) *col_begin = std::min<int64_t>(ctx->tile_begin * 64, ctx->N);
    if (col_end) *col_end = std::min<int64_t>(ctx->tile_end * 64, ctx->N);
    if (shard_nnz) *shard_nnz = ctx->nnz;
    return MPFMT_OK;
}

// ---- Euclidean per-edge steer (geometric.jl:18-19) ---------------------------------------------------------------------

int32_t mpfmt_euclid_steer(mpfmt_ctx* ctx, const int64_t* src, const int64_t* dst, int64_t E, double* t, double* u)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (E < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "E < 0");
    if (E == 0) return MPFMT_OK;
    if (!src || !dst || !t || !u) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "src / dst / t / u is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, src, E, "src")) || (rc = check_idx(ctx, dst, E, "dst"))) return rc;
    const int d = ctx->d;
    DevTmp tmp;
    int64_t *d_s = nullptr, *d_t = nullptr; double *dt = nullptr, *du = nullptr;
    if ((rc = up_i64(ctx, tmp, src, E, &d_s))) return rc;
    if ((rc = up_i64(ctx, tmp, dst, E, &d_t))) return rc;
    HIPCHK(ctx, tmp.get(&dt, sizeof(double) * (size_t)E));
    HIPCHK(ctx, tmp.get(&du, sizeof(double) * (size_t)E * d));
    if ((rc = mpfmt_launch_euclid_steer(ctx, d_s, d_t, E, dt, du))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(t, dt, sizeof(double) * (size_t)E, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(u, du, sizeof(double) * (size_t)E * d, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

int32_t mpfmt_euclid_propagate(mpfmt_ctx* ctx, const int64_t* src, int64_t E, const double* t, const double* u, const double* s, double* out)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (E < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "E < 0");
    if (E == 0) return MPFMT_OK;
    if (!src || !t || !u || !out) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "src / t / u / out is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, src, E, "src"))) return rc;
    const int d = ctx->d;
    DevTmp tmp;
    int64_t* d_s = nullptr; double *dt = nullptr, *du = nullptr, *ds = nullptr, *dout = nullptr;
    if ((rc = up_i64(ctx, tmp, src, E, &d_s))) return rc;
    HIPCHK(ctx, tmp.get(&dt, sizeof(double) * (size_t)E));
    HIPCHK(ctx, tmp.get(&du, sizeof(double) * (size_t)E * d));
    HIPCHK(ctx, tmp.get(&dout, sizeof(double) * (size_t)E * d));
    HIPCHK(ctx, hipMemcpyAsync(dt, t, sizeof(double) * (size_t)E, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(du, u, sizeof(double) * (size_t)E * d, hipMemcpyHostToDevice, ctx->stream));
    if (s) {
        HIPCHK(ctx, tmp.get(&ds, sizeof(double) * (size_t)E));
        HIPCHK(ctx, hipMemcpyAsync(ds, s, sizeof(double) * (size_t)E, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = mpfmt_launch_euclid_propagate(ctx, d_s, E, dt, du, ds, dout))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(out, dout, sizeof(double) * (size_t)E * d, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

// ---- expand --------------------------------------------------------------------------------------------

int32_t mpfmt_expand(mpfmt_ctx* ctx, const uint64_t* W, const uint64_t* H, const uint64_t* F, const double* C,
                     const int64_t* zs, int64_t nz,
                     int64_t* xs, int64_t* ymin, double* cmin, uint8_t* free_out, int64_t cap, int64_t* nx)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!W || !H || !C || !nx) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "W / H / C / nx is NULL");
    if (nz < 0 || cap < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "nz / cap < 0");
    if (!ctx->graph_filled) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "expand needs a built r-disc graph (mpfmt_rdisc_count + fill)");
    if (ctx->world != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "expand runs on an unsharded ctx (a shard holds only its own columns: the frontier would be partial); the sharded batch step is mpfmt_wf_step");
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    *nx = 0;
    if (nz == 0) return MPFMT_OK;
    if (!zs) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "zs is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = check_idx(ctx, zs, nz, "zs"))) return rc;
    const int64_t N = ctx->N, words = (N + 63) / 64;
    DevTmp tmp;
    uint64_t *dW = nullptr, *dH = nullptr, *dF = nullptr; double* dC = nullptr; int64_t* dz = nullptr;
    int64_t *dxs = nullptr, *dym = nullptr; double* dcm = nullptr; uint8_t* dfr = nullptr;
    const int64_t capd = std::max<int64_t>(cap, 1);
    HIPCHK(ctx, tmp.get(&dW, 8 * words)); HIPCHK(ctx, tmp.get(&dH, 8 * words));
    HIPCHK(ctx, tmp.get(&dC, 8 * N));
    HIPCHK(ctx, hipMemcpyAsync(dW, W, 8 * words, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dH, H, 8 * words, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dC, C, 8 * N, hipMemcpyHostToDevice, ctx->stream));
    if (F) { HIPCHK(ctx, tmp.get(&dF, 8 * words)); HIPCHK(ctx, hipMemcpyAsync(dF, F, 8 * words, hipMemcpyHostToDevice, ctx->stream)); }
    if ((rc = up_i64(ctx, tmp, zs, nz, &dz))) return rc;
    HIPCHK(ctx, tmp.get(&dxs, 8 * capd)); HIPCHK(ctx, tmp.get(&dym, 8 * capd));
    HIPCHK(ctx, tmp.get(&dcm, 8 * capd)); HIPCHK(ctx, tmp.get(&dfr, capd));
    rc = mpfmt_launch_expand(ctx, dW, dH, dF, dC, dz, nz, dxs, dym, dcm, dfr, cap, nx);
    if (rc == MPFMT_OK && *nx > 0) {
        const int64_t n = *nx;
        if (!xs || !ymin || !cmin || !free_out) rc = mpfmt_fail(ctx, MPFMT_ERR_ARG, "output array is NULL");
        else {
            hipMemcpyAsync(xs, dxs, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
            hipMemcpyAsync(ymin, dym, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
            hipMemcpyAsync(cmin, dcm, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
            hipMemcpyAsync(free_out, dfr, n, hipMemcpyDeviceToHost, ctx->stream);
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = mpfmt_fail(ctx, MPFMT_ERR_HIP, "expand copy back: %s", hipGetErrorString(e));
        }
    }
    return rc;
}

// ---- fmtstar -------------------------------------------------------------------------------------------

// The sequential recursion itself (fmt.jl:43-101 over finished arrays), the binary heap and the goal predicates live in
// mpfmt_host.cpp: plain C++ without HIP, so the same translation unit is also built with -fsanitize=address,undefined for the
// CPU test suite (tests/asan/).
static inline bool bit(const std::vector<uint64_t>& m, int64_t i) { return (m[i >> 6] >> (i & 63)) & 1ull; }
static inline bool is_goal_pt(const double* v, int d, int kind, const double* g) { return mpfmt_is_goal_pt(v, d, kind, g); }

int32_t mpfmt_fmtstar(mpfmt_ctx* ctx, double r, int64_t init_idx, int32_t checkpts,
                      int32_t goal_kind, const double* goal_params,
                      int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!A || !C || !path || !res || !goal_params) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output / goal pointer");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    if (ctx->world != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "fmtstar runs on an unsharded ctx");
    const int64_t N = ctx->N;
    const int d = ctx->d;
    if (init_idx < 1 || init_idx > N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "init_idx out of range");
    if (goal_kind < 0 || goal_kind > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown goal kind %d", goal_kind);
    if (!(r > 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and > 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    int32_t rc;

    // checkpts bitmap F (fmt.jl:31-36) -- also answers is_free_state(init) (fmt.jl:24-29)
    const int64_t words = (N + 63) / 64;
    std::vector<uint64_t> F(words, 0);
    auto t0 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_points_free(ctx, nullptr, N, F.data()))) return rc;
    if (!bit(F, init_idx - 1)) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "initial state is infeasible");
    auto t1 = std::chrono::steady_clock::now();

    // r-disc graph + per-edge free mask, all edges, on the device.  A filled graph of the same samples and radius (a
    // previous plan, or mpfmt_graph_import) is reused: only the obstacle-dependent sweep is redone.
    if (!(ctx->graph_filled && ctx->graph_r == r) && (rc = mpfmt_graph_build_device(ctx, r, nullptr))) return rc;
    auto t2 = std::chrono::steady_clock::now();
    ctx->pend_valid = false;
    if ((rc = mpfmt_launch_graph_sweep(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t3 = std::chrono::steady_clock::now();
    const int64_t nnz = ctx->nnz;
    std::vector<int64_t> colptr(N + 1);
    std::vector<int32_t> rowval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<double> nzval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<uint64_t> efree((size_t)std::max<int64_t>((nnz + 63) / 64, 1));
    std::vector<double> X((size_t)N * d);
    HIPCHK(ctx, hipMemcpy(colptr.data(), ctx->colptr, sizeof(int64_t) * (N + 1), hipMemcpyDeviceToHost));
    if (nnz > 0) {
        HIPCHK(ctx, hipMemcpy(rowval.data(), ctx->rowval, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(nzval.data(), ctx->nzval, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(efree.data(), ctx->graph_free, sizeof(uint64_t) * ((nnz + 63) / 64), hipMemcpyDeviceToHost));
    }
    HIPCHK(ctx, hipMemcpy(X.data(), ctx->Xo, sizeof(double) * (size_t)N * d, hipMemcpyDeviceToHost));
    auto t4 = std::chrono::steady_clock::now();

    if ((rc = mpfmt_host_fmt_recursion(N, d, X.data(), colptr.data(), rowval.data(), nzval.data(), efree.data(),
                                       checkpts ? F.data() : nullptr, ctx->ss.has ? ctx->ss.lo : nullptr,
                                       ctx->ss.has ? ctx->ss.hi : nullptr, init_idx, goal_kind, goal_params, A, C, path, res)))
        return mpfmt_fail(ctx, rc, "host recursion rejected its arguments");
    auto t5 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    res->nnz = nnz;
    res->ms_graph = ms(t1, t2);
    res->ms_sweep = ms(t0, t1) + ms(t2, t3);
    res->ms_host_loop = ms(t4, t5);
    return MPFMT_OK;
}

// ---- double integrator (LinearQuadratic quasi-metric space) ----------------------------------------------------

static int32_t di_check(mpfmt_ctx* ctx, double rho, double r)
{
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (ctx->d % 2 != 0 || ctx->d / 2 < 1 || ctx->d / 2 > 6)
        return mpfmt_fail(ctx, MPFMT_ERR_ARG, "double-integrator states need an even dimension 2..12 (got %d)", ctx->d);
    if (!(rho > 0.0) || !std::isfinite(rho)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rho must be finite and > 0");
    if (!(r > 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "cost radius must be finite and > 0");
    return MPFMT_OK;
}

int32_t mpfmt_di_graph_count(mpfmt_ctx* ctx, double rho, double r, int64_t* colptr, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!colptr || !nnz) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr / nnz is NULL");
    int32_t rc;
    if ((rc = di_check(ctx, rho, r))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = mpfmt_di_count(ctx, rho, r))) return rc;
    const int64_t n1 = ctx->N + 1;
    void* scr;
    if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * n1, &scr))) return rc;
    hipLaunchKernelGGL(k_add1_i64, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->colptr, n1, (int64_t*)scr);
    HIPCHK(ctx, hipMemcpyAsync(colptr, scr, sizeof(int64_t) * n1, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nnz = ctx->nnz;
    return MPFMT_OK;
}

int32_t mpfmt_di_graph_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval, double* tval)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->di_counted || ctx->steer_kind != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "di_graph_fill before di_graph_count");
    if (ctx->nnz > 0 && (!rowval || !nzval)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rowval / nzval is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if (!ctx->di_filled && (rc = mpfmt_di_fill(ctx))) return rc;
    const int64_t nnz = ctx->nnz;
    if (nnz > 0) {
        void* scr;
        if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * nnz, &scr))) return rc;
        hipLaunchKernelGGL(k_i32_to_i64_add1, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rowval, nnz, (int64_t*)scr);
        HIPCHK(ctx, hipMemcpyAsync(rowval, scr, sizeof(int64_t) * nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(nzval, ctx->nzval, sizeof(double) * nnz, hipMemcpyDeviceToHost, ctx->stream));
        if (tval) HIPCHK(ctx, hipMemcpyAsync(tval, ctx->tval, sizeof(double) * nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

int32_t mpfmt_di_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask, uint8_t* nseg)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->di_counted || ctx->steer_kind != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "di_graph_edges_free before di_graph_count");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if (!ctx->di_filled && (rc = mpfmt_di_fill(ctx))) return rc;
    if ((rc = mpfmt_di_sweep(ctx))) return rc;
    const int64_t nnz = ctx->nnz, words = (nnz + 63) / 64;
    if (nnz > 0) {
        if (!mask) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "mask is NULL");
        HIPCHK(ctx, hipMemcpyAsync(mask, ctx->graph_free, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, ctx->stream));
        if (nseg) HIPCHK(ctx, hipMemcpyAsync(nseg, ctx->di_nseg, (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

// the double-integrator graph and its edge bits with every output left in HBM (bench.py --workload cfg4: no PCIe in the timed region)
int32_t mpfmt_di_graph_step_device(mpfmt_ctx* ctx, double rho, double r, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    int32_t rc;
    if ((rc = di_check(ctx, rho, r))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = mpfmt_di_count(ctx, rho, r))) return rc;
    if ((rc = mpfmt_di_fill(ctx))) return rc;
    if (ctx->have_boxes && (rc = mpfmt_di_sweep(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (nnz) *nnz = ctx->nnz;
    return MPFMT_OK;
}

int32_t mpfmt_di_graph_device_ptrs(mpfmt_ctx* ctx, void** colptr, void** rowval, void** nzval, void** tval, void** free_mask, void** nseg)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->di_filled || ctx->steer_kind != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no double-integrator graph resident");
    if (colptr) *colptr = ctx->colptr;
    if (rowval) *rowval = ctx->rowval;
    if (nzval) *nzval = ctx->nzval;
    if (tval) *tval = ctx->tval;
    if (free_mask) *free_mask = ctx->di_swept ? (void*)ctx->graph_free : nullptr;
    if (nseg) *nseg = ctx->di_swept ? (void*)ctx->di_nseg : nullptr;
    return MPFMT_OK;
}

int32_t mpfmt_di_steer(mpfmt_ctx* ctx, const double* X0, const double* X1, int64_t n, int32_t m, double rho, double r,
                       double* cost, double* topt)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (n < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n < 0");
    if (n == 0) return MPFMT_OK;
    if (!X0 || !X1 || !cost || !topt) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL array");
    if (m < 1 || m > 6) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "workspace dim m must be 1..6");
    if (!(rho > 0.0) || !(r > 0.0)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rho and r must be > 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t pb = sizeof(double) * (size_t)n * 2 * m;
    DevTmp tmp;
    double *d0 = nullptr, *d1 = nullptr, *dc = nullptr, *dt = nullptr;
    HIPCHK(ctx, tmp.get(&d0, pb)); HIPCHK(ctx, tmp.get(&d1, pb));
    HIPCHK(ctx, tmp.get(&dc, 8 * n)); HIPCHK(ctx, tmp.get(&dt, 8 * n));
    HIPCHK(ctx, hipMemcpyAsync(d0, X0, pb, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d1, X1, pb, hipMemcpyHostToDevice, ctx->stream));
    int32_t rc = mpfmt_di_steer_launch(ctx, m, d0, d1, n, rho, r, dc, dt);
    if (rc == MPFMT_OK) {
        hipMemcpyAsync(cost, dc, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(topt, dt, 8 * n, hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = mpfmt_fail(ctx, MPFMT_ERR_HIP, "di_steer copy back: %s", hipGetErrorString(e));
    }
    return rc;
}

int32_t mpfmt_di_fmtstar(mpfmt_ctx* ctx, double rho, double r, int64_t init_idx, int32_t checkpts,
                         int32_t goal_kind, const double* goal_params,
                         int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!A || !C || !path || !res || !goal_params) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output / goal pointer");
    int32_t rc;
    if ((rc = di_check(ctx, rho, r))) return rc;
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    const int64_t N = ctx->N;
    const int n = ctx->d, m = n / 2;
    if (init_idx < 1 || init_idx > N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "init_idx out of range");
    if (goal_kind < 0 || goal_kind > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown goal kind %d", goal_kind);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    auto t0 = std::chrono::steady_clock::now();
    // checkpts bitmap: is_free_state(v, CC, SS) = in_state_space(v) && point-vs-boxes on the workspace coordinates.
    // The point kernel works on dw-dimensional points, so gather the workspace coordinates of every state.
    std::vector<double> X((size_t)N * n);
    HIPCHK(ctx, hipMemcpy(X.data(), ctx->Xo, sizeof(double) * (size_t)N * n, hipMemcpyDeviceToHost));
    std::vector<double> P((size_t)N * m);
    for (int64_t i = 0; i < N; ++i) for (int q = 0; q < m; ++q) P[(size_t)i * m + q] = X[(size_t)i * n + q];
    const int64_t words = (N + 63) / 64;
    std::vector<uint64_t> F(words, 0);
    {
        const mpfmt_ss keep = ctx->ss;            // the workspace sweep must not apply the 2m-dim bounds
        ctx->ss.has = 0;
        rc = mpfmt_states_free(ctx, P.data(), N, F.data());
        ctx->ss = keep;
        if (rc) return rc;
        if (keep.has)
            for (int64_t i = 0; i < N; ++i) {
                bool ok = true;
                for (int q = 0; q < n; ++q) ok = ok && (keep.lo[q] <= X[(size_t)i * n + q]) && (X[(size_t)i * n + q] <= keep.hi[q]);
                if (!ok) F[i >> 6] &= ~(1ull << (i & 63));
            }
    }
    if (!bit(F, init_idx - 1)) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "initial state is infeasible");
    auto t1 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_di_count(ctx, rho, r))) return rc;
    if ((rc = mpfmt_di_fill(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t2 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_di_sweep(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t3 = std::chrono::steady_clock::now();
    const int64_t nnz = ctx->nnz;
    std::vector<int64_t> colptr(N + 1);
    std::vector<int32_t> rowval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<double> nzval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<uint64_t> efree((size_t)std::max<int64_t>((nnz + 63) / 64, 1));
    std::vector<uint8_t> nseg((size_t)std::max<int64_t>(nnz, 1));
    HIPCHK(ctx, hipMemcpy(colptr.data(), ctx->colptr, sizeof(int64_t) * (N + 1), hipMemcpyDeviceToHost));
    if (nnz > 0) {
        HIPCHK(ctx, hipMemcpy(rowval.data(), ctx->rowval, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(nzval.data(), ctx->nzval, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(efree.data(), ctx->graph_free, sizeof(uint64_t) * ((nnz + 63) / 64), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(nseg.data(), ctx->di_nseg, (size_t)nnz, hipMemcpyDeviceToHost));
    }
    auto t4 = std::chrono::steady_clock::now();
    auto goal_hit = [&](int64_t z) {
        const double* v = &X[(size_t)z * n];
        if (goal_kind == MPFMT_GOAL_POINT) {                      // StateGoal: exact state equality (goals.jl:128-131)
            for (int q = 0; q < n; ++q) if (!(v[q] == goal_params[q])) return false;
            return true;
        }
        return is_goal_pt(v, m, goal_kind, goal_params);           // workspace goals act on C*v = first m coordinates
    };
    mpfmt_csr_host csr;
    mpfmt_csr_view csr_view;
    const mpfmt_csr_view* pre_ptr = nullptr;
    if (mpfmt_csc_transpose_device(ctx, &csr) == MPFMT_OK) { csr_view = {csr.rowptr.data(), csr.colidx.data(), csr.centry.data()}; pre_ptr = &csr_view; }
    mpfmt_directed_fmt_recursion(N, colptr.data(), rowval.data(), nzval.data(), efree.data(), nseg.data(), checkpts ? F.data() : nullptr,
                                 init_idx, goal_hit, A, C, path, res, pre_ptr);
    auto t5 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    res->nnz = nnz;
    res->ms_graph = ms(t1, t2); res->ms_sweep = ms(t0, t1) + ms(t2, t3); res->ms_host_loop = ms(t4, t5);
    return MPFMT_OK;
}

// fmtstar! in the double-integrator space with the recursion on the device (kernels_wavefront.hip, directed form): graph,
// 5-waypoint sweep and the transpose (forward sets) on the device, then cost-band batches; `single` reproduces
// mpfmt_di_fmtstar / the reference's pop order exactly.
int32_t mpfmt_di_fmtstar_wavefront(mpfmt_ctx* ctx, double rho, double r, int64_t init_idx, int32_t checkpts, int32_t goal_kind,
                                   const double* goal_params, double band, int32_t flags, int64_t* A, double* C, int64_t* path,
                                   mpfmt_fmt_result* res, mpfmt_wf_info* info)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!res || !goal_params) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output / goal pointer");
    int32_t rc;
    if ((rc = di_check(ctx, rho, r))) return rc;
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    const int64_t N = ctx->N;
    const int n = ctx->d, m = n / 2;
    if (init_idx < 1 || init_idx > N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "init_idx out of range");
    if (goal_kind < 0 || goal_kind > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown goal kind %d", goal_kind);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    auto t0 = std::chrono::steady_clock::now();
    // checkpts bitmap: in_state_space on the state, point test on the workspace coordinates (as in mpfmt_di_fmtstar)
    std::vector<double> X((size_t)N * n);
    HIPCHK(ctx, hipMemcpy(X.data(), ctx->Xo, sizeof(double) * (size_t)N * n, hipMemcpyDeviceToHost));
    std::vector<double> P((size_t)N * m);
    for (int64_t i = 0; i < N; ++i) for (int q = 0; q < m; ++q) P[(size_t)i * m + q] = X[(size_t)i * n + q];
    const int64_t words = (N + 63) / 64;
    std::vector<uint64_t> F(words, 0);
    {
        const mpfmt_ss keep = ctx->ss;
        ctx->ss.has = 0;
        rc = mpfmt_states_free(ctx, P.data(), N, F.data());
        ctx->ss = keep;
        if (rc) return rc;
        if (keep.has)
            for (int64_t i = 0; i < N; ++i) {
                bool ok = true;
                for (int q = 0; q < n; ++q) ok = ok && (keep.lo[q] <= X[(size_t)i * n + q]) && (X[(size_t)i * n + q] <= keep.hi[q]);
                if (!ok) F[i >> 6] &= ~(1ull << (i & 63));
            }
    }
    if (!bit(F, init_idx - 1)) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "initial state is infeasible");
    auto t1 = std::chrono::steady_clock::now();
    if (!(ctx->di_filled && ctx->steer_kind == 1 && ctx->di_rho == rho && ctx->di_r == r)) {
        if ((rc = mpfmt_di_count(ctx, rho, r))) return rc;
        if ((rc = mpfmt_di_fill(ctx))) return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t2 = std::chrono::steady_clock::now();
    if (!ctx->di_swept && (rc = mpfmt_di_sweep(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t3 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_wf_begin_directed(ctx, init_idx, checkpts, F.data(), goal_kind, goal_params, m, band, flags))) return rc;
    if ((rc = mpfmt_wf_run(ctx))) return rc;
    if ((rc = mpfmt_wf_finish(ctx, A, C, path, res))) return rc;
    auto t4 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    res->ms_graph = ms(t1, t2); res->ms_sweep = ms(t0, t1) + ms(t2, t3); res->ms_host_loop = ms(t3, t4);
    mpfmt_wf_info_now(ctx, info);
    return MPFMT_OK;
}

// ---- Dubins and Reeds-Shepp cars (kernels_car.hip) ------------------------------------------------------------------

static int32_t car_graph_count(mpfmt_ctx* ctx, int kind, double turn_radius, double speed, double r, int64_t* colptr, int64_t* nnz)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!colptr || !nnz) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "colptr / nnz is NULL");
    int32_t rc;
    if ((rc = mpfmt_car_build(ctx, kind, turn_radius, speed, r))) return rc;
    const int64_t n1 = ctx->N + 1;
    void* scr;
    if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * n1, &scr))) return rc;
    hipLaunchKernelGGL(k_add1_i64, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->colptr, n1, (int64_t*)scr);
    HIPCHK(ctx, hipMemcpyAsync(colptr, scr, sizeof(int64_t) * n1, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *nnz = ctx->nnz;
    return MPFMT_OK;
}

static int32_t car_graph_fill(mpfmt_ctx* ctx, int kind, int64_t* rowval, double* nzval)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!(ctx->di_filled && ctx->steer_kind == kind + 1)) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car graph_fill before graph_count");
    const int64_t nnz = ctx->nnz;
    if (nnz > 0 && (!rowval || !nzval)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rowval / nzval is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (nnz > 0) {
        int32_t rc;
        void* scr;
        if ((rc = mpfmt_scratch(ctx, sizeof(int64_t) * nnz, &scr))) return rc;
        hipLaunchKernelGGL(k_i32_to_i64_add1, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, ctx->stream, ctx->rowval, nnz, (int64_t*)scr);
        HIPCHK(ctx, hipMemcpyAsync(rowval, scr, sizeof(int64_t) * nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(nzval, ctx->nzval, sizeof(double) * nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

static int32_t car_graph_edges_free(mpfmt_ctx* ctx, int kind, uint64_t* mask, uint8_t* nseg)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!(ctx->di_filled && ctx->steer_kind == kind + 1)) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car graph_edges_free before graph_count");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = mpfmt_car_sweep(ctx))) return rc;
    const int64_t nnz = ctx->nnz, words = (nnz + 63) / 64;
    if (nnz > 0) {
        if (!mask) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "mask is NULL");
        HIPCHK(ctx, hipMemcpyAsync(mask, ctx->graph_free, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, ctx->stream));
        if (nseg) HIPCHK(ctx, hipMemcpyAsync(nseg, ctx->di_nseg, (size_t)nnz, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

// controls out: [n][nctl][3]; nctl = 3 (Dubins) or 5 (Reeds-Shepp); nsegs (may be NULL) = segments used per pair
static int32_t car_steer_pairs(mpfmt_ctx* ctx, int kind, const double* X0, const double* X1, int64_t n, double turn_radius, double speed,
                               double* cost, double* controls, int32_t* nsegs)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (n < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n < 0");
    if (n == 0) return MPFMT_OK;
    if (!X0 || !X1 || !cost) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL array");
    if (!(turn_radius > 0.0) || !(speed > 0.0)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "turning radius and speed must be > 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevTmp tmp;
    double *d0, *d1, *dc, *du; int32_t* dn;
    int32_t rc;
    HIPCHK(ctx, tmp.get(&d0, sizeof(double) * 3 * n));
    HIPCHK(ctx, tmp.get(&d1, sizeof(double) * 3 * n));
    HIPCHK(ctx, tmp.get(&dc, sizeof(double) * n));
    HIPCHK(ctx, tmp.get(&du, sizeof(double) * 15 * n));
    HIPCHK(ctx, tmp.get(&dn, sizeof(int32_t) * n));
    HIPCHK(ctx, hipMemcpyAsync(d0, X0, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d1, X1, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = mpfmt_car_steer_batch(ctx, kind, d0, d1, n, turn_radius, speed, dc, du, dn))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(cost, dc, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> u5;
    if (controls) { u5.resize((size_t)15 * n); HIPCHK(ctx, hipMemcpyAsync(u5.data(), du, sizeof(double) * 15 * n, hipMemcpyDeviceToHost, ctx->stream)); }
    if (nsegs) HIPCHK(ctx, hipMemcpyAsync(nsegs, dn, sizeof(int32_t) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (controls) {
        const int nctl = (kind == 2) ? 5 : 3;
        for (int64_t i = 0; i < n; ++i) memcpy(controls + (size_t)i * nctl * 3, u5.data() + (size_t)i * 15, sizeof(double) * nctl * 3);
    }
    return MPFMT_OK;
}

static int32_t car_fmtstar(mpfmt_ctx* ctx, int kind, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                           int32_t goal_kind, const double* goal_params, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!A || !C || !path || !res || !goal_params) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output / goal pointer");
    if (!ctx->Xo || ctx->d != 3) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car planning needs SE2 samples (d = 3)");
    if (!ctx->have_boxes || ctx->dw != 2) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car planning needs a 2-D workspace checker (mpfmt_upload_boxes with dw = 2, or mpfmt_upload_shapes2d)");
    const int64_t N = ctx->N;
    if (init_idx < 1 || init_idx > N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "init_idx out of range");
    if (goal_kind < 0 || goal_kind > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown goal kind %d", goal_kind);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    int32_t rc;
    auto t0 = std::chrono::steady_clock::now();
    // checkpts bitmap: in_state_space on the SE2 state, point test on (x, y)
    std::vector<double> X((size_t)N * 3), P((size_t)N * 2);
    HIPCHK(ctx, hipMemcpy(X.data(), ctx->Xo, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i) { P[2 * i] = X[3 * i]; P[2 * i + 1] = X[3 * i + 1]; }
    const int64_t words = (N + 63) / 64;
    std::vector<uint64_t> F(words, 0);
    {
        const mpfmt_ss keep = ctx->ss;
        ctx->ss.has = 0;
        rc = mpfmt_states_free(ctx, P.data(), N, F.data());
        ctx->ss = keep;
        if (rc) return rc;
        if (keep.has)
            for (int64_t i = 0; i < N; ++i) {
                bool ok = true;
                for (int q = 0; q < 3; ++q) ok = ok && (keep.lo[q] <= X[(size_t)i * 3 + q]) && (X[(size_t)i * 3 + q] <= keep.hi[q]);
                if (!ok) F[i >> 6] &= ~(1ull << (i & 63));
            }
    }
    if (!bit(F, init_idx - 1)) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "initial state is infeasible");
    auto t1 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_car_build(ctx, kind, turn_radius, speed, r))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t2 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_car_sweep(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t3 = std::chrono::steady_clock::now();
    const int64_t nnz = ctx->nnz;
    std::vector<int64_t> colptr(N + 1);
    std::vector<int32_t> rowval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<double> nzval((size_t)std::max<int64_t>(nnz, 1));
    std::vector<uint64_t> efree((size_t)std::max<int64_t>((nnz + 63) / 64, 1));
    std::vector<uint8_t> nseg((size_t)std::max<int64_t>(nnz, 1));
    HIPCHK(ctx, hipMemcpy(colptr.data(), ctx->colptr, sizeof(int64_t) * (N + 1), hipMemcpyDeviceToHost));
    if (nnz > 0) {
        HIPCHK(ctx, hipMemcpy(rowval.data(), ctx->rowval, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(nzval.data(), ctx->nzval, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(efree.data(), ctx->graph_free, sizeof(uint64_t) * ((nnz + 63) / 64), hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(nseg.data(), ctx->di_nseg, (size_t)nnz, hipMemcpyDeviceToHost));
    }
    auto t4 = std::chrono::steady_clock::now();
    if (kind == 2) {
        // Reeds-Shepp: a (chopped) metric -- forward and backward sets coincide (nearneighbors.jl:200-203): the symmetric
        // recursion on column = inball; the goal acts on (x, y), POINT = exact state
        std::vector<double> gp3;
        const double* gp = goal_params;
        if ((rc = mpfmt_host_fmt_recursion_impl(N, 3, X.data(), colptr.data(), rowval.data(), nzval.data(), efree.data(),
                                          checkpts ? F.data() : nullptr, nullptr, nullptr, init_idx, goal_kind, gp, goal_kind == MPFMT_GOAL_POINT ? 3 : 2,
                                          nseg.data(), A, C, path, res)))
            return mpfmt_fail(ctx, rc, "host recursion rejected its arguments");
    } else {
        auto goal_hit = [&](int64_t z) {
            const double* v = &X[(size_t)z * 3];
            if (goal_kind == MPFMT_GOAL_POINT) return v[0] == goal_params[0] && v[1] == goal_params[1] && v[2] == goal_params[2];
            return is_goal_pt(v, 2, goal_kind, goal_params);                       // workspace goals act on (x, y)
        };
        mpfmt_csr_host csr;
        mpfmt_csr_view csr_view;
        const mpfmt_csr_view* pre_ptr = nullptr;
        if (mpfmt_csc_transpose_device(ctx, &csr) == MPFMT_OK) { csr_view = {csr.rowptr.data(), csr.colidx.data(), csr.centry.data()}; pre_ptr = &csr_view; }
        mpfmt_directed_fmt_recursion(N, colptr.data(), rowval.data(), nzval.data(), efree.data(), nseg.data(), checkpts ? F.data() : nullptr,
                                     init_idx, goal_hit, A, C, path, res, pre_ptr);
    }
    auto t5 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    res->nnz = nnz;
    res->ms_graph = ms(t1, t2); res->ms_sweep = ms(t0, t1) + ms(t2, t3); res->ms_host_loop = ms(t4, t5);
    return MPFMT_OK;
}

// the car planners with the recursion on the device (directed wavefront form; the Reeds-Shepp graph is structurally symmetric,
// so its rows are its columns and the same form applies): graph + waypoint sweep as in car_fmtstar, then cost-band batches
static int32_t car_fmtstar_wavefront(mpfmt_ctx* ctx, int kind, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                     int32_t goal_kind, const double* goal_params, double band, int32_t flags, int64_t* A, double* C,
                                     int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!res || !goal_params) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "NULL output / goal pointer");
    if (!ctx->Xo || ctx->d != 3) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car planning needs SE2 samples (d = 3)");
    if (!ctx->have_boxes || ctx->dw != 2) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "car planning needs a 2-D workspace checker (mpfmt_upload_boxes with dw = 2, or mpfmt_upload_shapes2d)");
    const int64_t N = ctx->N;
    if (init_idx < 1 || init_idx > N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "init_idx out of range");
    if (goal_kind < 0 || goal_kind > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown goal kind %d", goal_kind);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<double> X((size_t)N * 3), P((size_t)N * 2);
    HIPCHK(ctx, hipMemcpy(X.data(), ctx->Xo, sizeof(double) * (size_t)N * 3, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < N; ++i) { P[2 * i] = X[3 * i]; P[2 * i + 1] = X[3 * i + 1]; }
    const int64_t words = (N + 63) / 64;
    std::vector<uint64_t> F(words, 0);
    {
        const mpfmt_ss keep = ctx->ss;
        ctx->ss.has = 0;
        rc = mpfmt_states_free(ctx, P.data(), N, F.data());
        ctx->ss = keep;
        if (rc) return rc;
        if (keep.has)
            for (int64_t i = 0; i < N; ++i) {
                bool ok = true;
                for (int q = 0; q < 3; ++q) ok = ok && (keep.lo[q] <= X[(size_t)i * 3 + q]) && (X[(size_t)i * 3 + q] <= keep.hi[q]);
                if (!ok) F[i >> 6] &= ~(1ull << (i & 63));
            }
    }
    if (!bit(F, init_idx - 1)) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "initial state is infeasible");
    auto t1 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_car_build(ctx, kind, turn_radius, speed, r))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t2 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_car_sweep(ctx))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t3 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_wf_begin_directed(ctx, init_idx, checkpts, F.data(), goal_kind, goal_params, 2, band, flags))) return rc;
    if ((rc = mpfmt_wf_run(ctx))) return rc;
    if ((rc = mpfmt_wf_finish(ctx, A, C, path, res))) return rc;
    auto t4 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    res->ms_graph = ms(t1, t2); res->ms_sweep = ms(t0, t1) + ms(t2, t3); res->ms_host_loop = ms(t3, t4);
    mpfmt_wf_info_now(ctx, info);
    return MPFMT_OK;
}

int32_t mpfmt_dubins_fmtstar_wavefront(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                       int32_t goal_kind, const double* goal_params, double band, int32_t flags, int64_t* A, double* C,
                                       int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info)
{ return car_fmtstar_wavefront(ctx, 1, turn_radius, speed, r, init_idx, checkpts, goal_kind, goal_params, band, flags, A, C, path, res, info); }
int32_t mpfmt_reedsshepp_fmtstar_wavefront(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                           int32_t goal_kind, const double* goal_params, double band, int32_t flags, int64_t* A, double* C,
                                           int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info)
{ return car_fmtstar_wavefront(ctx, 2, turn_radius, speed, r, init_idx, checkpts, goal_kind, goal_params, band, flags, A, C, path, res, info); }

int32_t mpfmt_dubins_graph_count(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t* colptr, int64_t* nnz)
{ return car_graph_count(ctx, 1, turn_radius, speed, r, colptr, nnz); }
int32_t mpfmt_dubins_graph_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval) { return car_graph_fill(ctx, 1, rowval, nzval); }
int32_t mpfmt_dubins_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask, uint8_t* nseg) { return car_graph_edges_free(ctx, 1, mask, nseg); }
int32_t mpfmt_dubins_steer(mpfmt_ctx* ctx, const double* X0, const double* X1, int64_t n, double turn_radius, double speed,
                           double* cost, double* controls)
{ return car_steer_pairs(ctx, 1, X0, X1, n, turn_radius, speed, cost, controls, nullptr); }
int32_t mpfmt_dubins_fmtstar(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                             int32_t goal_kind, const double* goal_params, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{ return car_fmtstar(ctx, 1, turn_radius, speed, r, init_idx, checkpts, goal_kind, goal_params, A, C, path, res); }

int32_t mpfmt_reedsshepp_graph_count(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t* colptr, int64_t* nnz)
{ return car_graph_count(ctx, 2, turn_radius, speed, r, colptr, nnz); }
int32_t mpfmt_reedsshepp_graph_fill(mpfmt_ctx* ctx, int64_t* rowval, double* nzval) { return car_graph_fill(ctx, 2, rowval, nzval); }
int32_t mpfmt_reedsshepp_graph_edges_free(mpfmt_ctx* ctx, uint64_t* mask, uint8_t* nseg) { return car_graph_edges_free(ctx, 2, mask, nseg); }
int32_t mpfmt_reedsshepp_steer(mpfmt_ctx* ctx, const double* X0, const double* X1, int64_t n, double turn_radius, double speed,
                               double* cost, double* controls, int32_t* nsegs)
{ return car_steer_pairs(ctx, 2, X0, X1, n, turn_radius, speed, cost, controls, nsegs); }
int32_t mpfmt_reedsshepp_fmtstar(mpfmt_ctx* ctx, double turn_radius, double speed, double r, int64_t init_idx, int32_t checkpts,
                                 int32_t goal_kind, const double* goal_params, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{ return car_fmtstar(ctx, 2, turn_radius, speed, r, init_idx, checkpts, goal_kind, goal_params, A, C, path, res); }

// ---- measurement ---------------------------------------------------------------------------------------

int32_t mpfmt_timing_reset(mpfmt_ctx* ctx)
{
    if (!ctx) return MPFMT_ERR_ARG;
    timer_resolve(ctx);
    ctx->timers.clear();
    return MPFMT_OK;
}

int32_t mpfmt_timing_get(mpfmt_ctx* ctx, const char* name, double* avg_ms, int64_t* launches)
{
    if (!ctx || !name) return MPFMT_ERR_ARG;
    timer_resolve(ctx);
    auto it = ctx->timers.find(name);
    double a = 0.0; int64_t n = 0;
    if (it != ctx->timers.end() && it->second.launches > 0) { n = it->second.launches; a = it->second.total_ms / (double)n; }
    if (avg_ms) *avg_ms = a;
    if (launches) *launches = n;
    return MPFMT_OK;
}

int32_t mpfmt_set_option(mpfmt_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return MPFMT_ERR_ARG;
    if (strcmp(name, "rdisc_path") == 0) {
        if (value < 0 || value > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "rdisc_path must be 0 (auto), 1 (fp64 VALU) or 2 (MFMA filter)");
        ctx->rdisc_path = (int32_t)value;
        ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
        return MPFMT_OK;
    }
    if (strcmp(name, "rebuild_index") == 0) { ctx->rebuild_index = value != 0; return MPFMT_OK; }
    if (strcmp(name, "rdisc_pool") == 0) { ctx->use_pool = value != 0; return MPFMT_OK; }
    if (strcmp(name, "wf_graphs") == 0) { ctx->wf_graphs = value != 0; return MPFMT_OK; }
    if (strcmp(name, "debug_small_lists") == 0) { ctx->debug_small_lists = value != 0; return MPFMT_OK; }
    if (strcmp(name, "fuse_broad") == 0) {
        ctx->fuse_broad = (int)value;                           // 0 off, 1 flagged entries listed by the ordering pass (k_sweep_pending), 2 flagged pairs tested before it (k_exact_pairs)
        ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false; ctx->spec_ready = false;
        return MPFMT_OK;
    }
    if (strcmp(name, "rdisc_half") == 0) {
        ctx->use_half = value != 0;
        ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false; ctx->spec_ready = false; ctx->lists_r = -1.0;
        return MPFMT_OK;
    }
    if (strcmp(name, "mf_tail_permille") == 0) { ctx->mf_tail_permille = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 1000); return MPFMT_OK; }
    if (strcmp(name, "ord_draw") == 0) { ctx->ord_draw = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 2); return MPFMT_OK; }
    if (strcmp(name, "mf_tail_min_items") == 0) { ctx->mf_tail_min_items = std::max<int64_t>(value, 0); return MPFMT_OK; }
    if (strcmp(name, "mf_tail_slices") == 0) { ctx->mf_tail_slices = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 63); return MPFMT_OK; }
    if (strcmp(name, "mf_xcd_mode") == 0) { ctx->mf_xcd_mode = (int32_t)value; return MPFMT_OK; }
    if (strcmp(name, "overlap") == 0) { ctx->overlap = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 2); return MPFMT_OK; }
    if (strcmp(name, "wf_pos_space") == 0) { ctx->wf_pos_space = value < 0 ? 0 : (value > 2 ? 2 : (int32_t)value); return MPFMT_OK; }
    if (strcmp(name, "shard_blocks") == 0) { ctx->shard_blocks = value != 0; ctx->grid_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0; ctx->cut_key.clear(); return MPFMT_OK; }
    if (strcmp(name, "index_halo") == 0) { ctx->index_halo = value != 0; ctx->grid_r = -1.0; ctx->ops_r = -1.0; ctx->lists_r = -1.0; return MPFMT_OK; }
    if (strcmp(name, "cell_fb_max") == 0) { ctx->cell_fb_max = (int32_t)std::min<int64_t>(8, std::max<int64_t>(0, value)); ctx->grid_r = -1.0; return MPFMT_OK; }
    if (strcmp(name, "mf_target_items") == 0) { ctx->mf_target_items = value; return MPFMT_OK; }
    if (strcmp(name, "timing") == 0) { ctx->timing_enabled = value != 0; return MPFMT_OK; }
    if (strcmp(name, "sweep_rounds") == 0) {
        ctx->sweep_rounds = value != 0;
        return MPFMT_OK;
    }
    if (strcmp(name, "di_path") == 0) { ctx->di_path = (int32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 2); ctx->di_counted = ctx->di_filled = ctx->di_swept = false; return MPFMT_OK; }
    if (strcmp(name, "wf_force_sharded") == 0) { ctx->wf_force_sharded = value != 0; return MPFMT_OK; }
    return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown option %s", name);
}

int32_t mpfmt_get_stat(mpfmt_ctx* ctx, const char* name, int64_t* value)
{
    if (!ctx || !name || !value) return MPFMT_ERR_ARG;
    if (strcmp(name, "rdisc_path_used") == 0) { *value = ctx->rdisc_path_used; return MPFMT_OK; }
    if (strcmp(name, "rdisc_half_used") == 0) { *value = ctx->half_used ? 1 : 0; return MPFMT_OK; }
    if (strcmp(name, "sweep_form") == 0) {                   // how the resident mask was made: 0 whole sweep, 1 pending entries, 2 pairs before the ordering
        *value = ctx->sweep_in_order ? 2 : (ctx->sweep_pending_used && ctx->pend_valid) ? 1 : 0; return MPFMT_OK;
    }
    if (strcmp(name, "pair_items") == 0) {                   // (a synchronising read) pending pairs the last half build listed for k_exact_pairs
        *value = 0;
        const int64_t items = 1024;
        if (ctx->bits_in_records && ctx->pair_cnt && items > 0) {
            std::vector<int32_t> h((size_t)items);
            HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->pair_cnt, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            for (int32_t v : h) *value += v;
        }
        return MPFMT_OK;
    }
    if (strcmp(name, "pend_overflowed") == 0) { *value = ctx->pend_overflowed ? 1 : 0; return MPFMT_OK; }
    if (strcmp(name, "redo_count") == 0) { *value = ctx->redo_count; return MPFMT_OK; }
    if (strcmp(name, "redo_reason") == 0) { *value = ctx->redo_reason; ctx->redo_reason = 0; return MPFMT_OK; }      // (bits since the last read)
    if (strcmp(name, "ord_per_cu") == 0) { *value = ctx->ord_per_cu; return MPFMT_OK; }
    if (strcmp(name, "qcap") == 0) { *value = ctx->qcap; return MPFMT_OK; }
    if (strcmp(name, "pool_used") == 0) { *value = ctx->pool_valid ? 1 : 0; return MPFMT_OK; }
    if (strcmp(name, "list_cap") == 0) { *value = ctx->list_cap; return MPFMT_OK; }
    if (strcmp(name, "list_max") == 0) {                     // (a synchronising read) longest chunk list of the last list build
        *value = 0;
        const int64_t nt = ctx->tile_end - ctx->tile_begin;
        if (ctx->list_len && nt > 0) {
            int32_t v = 0;
            HIPCHK(ctx, hipMemcpyAsync(&v, ctx->list_max, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            *value = v;
        }
        return MPFMT_OK;
    }
    if (strncmp(name, "list_q", 6) == 0 || strcmp(name, "list_argmax") == 0 || strcmp(name, "list_sum") == 0) {
        // (synchronising reads, diagnostics) list_q<permille>: that quantile of the tiles' chunk-list lengths; list_argmax: the tile (counted
        // from the shard's first) with the longest list; list_sum: all entries
        *value = 0;
        const int64_t nt = ctx->tile_end - ctx->tile_begin;
        if (ctx->list_len && nt > 0) {
            std::vector<int32_t> h((size_t)nt);
            HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->list_len, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (strcmp(name, "list_argmax") == 0) *value = (int64_t)(std::max_element(h.begin(), h.end()) - h.begin());
            else if (strcmp(name, "list_sum") == 0) { int64_t t = 0; for (int32_t v : h) t += v; *value = t; }
            else {
                const int64_t q = std::min<int64_t>(1000, std::max<int64_t>(0, atoll(name + 6)));
                std::sort(h.begin(), h.end());
                *value = h[(size_t)std::min<int64_t>(nt - 1, nt * q / 1000)];
            }
        }
        return MPFMT_OK;
    }
    if (strcmp(name, "survivors") == 0) { *value = ctx->survivors; return MPFMT_OK; }
    if (strcmp(name, "pairs_tested") == 0) { *value = ctx->pairs_tested; return MPFMT_OK; }
    if (strcmp(name, "nnz") == 0) { *value = ctx->nnz; return MPFMT_OK; }
    if (strcmp(name, "slices") == 0) { *value = ctx->S; return MPFMT_OK; }
    if (strcmp(name, "cells") == 0) { *value = ctx->grid.ncells; return MPFMT_OK; }
    if (strcmp(name, "filter_valu") == 0) { *value = ctx->filter_valu ? 1 : 0; return MPFMT_OK; }
    if (strcmp(name, "wf_pos_space_used") == 0) { *value = ctx->wf_pos_used; return MPFMT_OK; }
    if (strcmp(name, "di_path_used") == 0) { *value = ctx->di_mf ? 2 : 1; return MPFMT_OK; }
    return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown stat %s", name);
}

int32_t mpfmt_graph_stats(mpfmt_ctx* ctx, int64_t* pairs_tested, int64_t* tiles, int64_t* slices, int64_t* cells)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->graph_counted) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no graph counted");
    if (pairs_tested) *pairs_tested = ctx->pairs_tested;
    if (tiles) *tiles = ctx->tile_end - ctx->tile_begin;
    if (slices) *slices = ctx->S;
    if (cells) *cells = ctx->grid.ncells;
    return MPFMT_OK;
}

}  // extern "C"
