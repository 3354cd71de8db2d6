// Multi-GPU exchange behind the C ABI: RCCL over xGMI, called directly -- no Python in the data path.
//
// One communicator per ctx (= per GPU).  Two ways to drive it:
//   - one process per GPU (bench.py under torch.distributed.run): every rank calls mpfmt_comm_create with the same
//     128-byte id (rank 0 obtains it from mpfmt_comm_unique_id and the host distributes it);
//   - ONE host thread driving G ctxs (the single-threaded Julia reference, src/problems.jl:12-30 holding G handles):
//     the same calls bracketed by mpfmt_group_begin / mpfmt_group_end (ncclGroupStart / ncclGroupEnd), and the split
//     launch / finish forms so that no call blocks on a peer that the same thread has yet to launch.
// The collectives run on a communication stream of their own, ordered after the compute stream by an event, so the
// mask gather of step k overlaps the index build of step k+1.
//
// RCCL is resolved at run time (dlopen of librccl.so.1: the copy torch has already mapped when there is one, the ROCm
// one otherwise), so single-GPU users of libmpfmt.so do not need it.
#include "mpfmt_internal.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <vector>

namespace {

struct rccl_api {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

rccl_api g_rccl;
std::once_flag g_rccl_once;

void rccl_load()
{
    // MPFMT_RCCL_LIB: an explicit library (the tests put a shared-memory stand-in there to run several ranks on one GPU).  An
    // environment variable must not be able to swap the collective library of a production process by accident: it is
    // honoured only together with MPFMT_ALLOW_RCCL_OVERRIDE=1
    const char* over = getenv("MPFMT_RCCL_LIB");
    const char* allow = getenv("MPFMT_ALLOW_RCCL_OVERRIDE");
    if (over && *over && !(allow && allow[0] == '1' && allow[1] == 0)) {
        g_rccl.err = "MPFMT_RCCL_LIB is set without MPFMT_ALLOW_RCCL_OVERRIDE=1: refusing to substitute the collective library";
        return;
    }
    const char* names[] = {over, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.dl) break;
    }
    if (!g_rccl.dl) { g_rccl.err = std::string("cannot load RCCL: ") + dlerror(); return; }
#define SYM(field, name)                                                                   \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.dl, name);                          \
    if (!g_rccl.field) { g_rccl.err = std::string("RCCL lacks ") + name; return; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
}

int32_t rccl_ready(mpfmt_ctx* ctx)
{
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.err.empty()) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "%s", g_rccl.err.c_str());
    return MPFMT_OK;
}

#define NCCLCHK(ctx, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess)                                                                              \
            return mpfmt_fail((ctx), MPFMT_ERR_HIP, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), \
                              __FILE__, __LINE__);                                                          \
    } while (0)

}  // namespace

struct mpfmt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;            // communication stream
    hipEvent_t ev_compute = nullptr;         // compute stream -> communication stream
    hipEvent_t ev_done = nullptr;            // communication stream -> host / compute stream
    // free-mask gather: [world][2 + cap] words; slot g = (mask words of rank g, nnz of rank g, mask...)
    uint64_t* gbuf = nullptr;
    int64_t gbuf_words = 0;
    int64_t cap = -1;                        // mask words per rank agreed for the NEXT exchange (-1: not agreed yet)
    int64_t slot_last = 0;                   // words per rank slot (2 header words + capacity) of the last exchange
    int64_t* hdr_host = nullptr;             // pinned [world][2]
    int64_t* hdr_dev = nullptr;              // [world][2] staging of the first (lengths only) exchange
    bool pending = false;
    bool grouped = false;                    // the gather in flight was launched inside mpfmt_group_begin / _end (one thread, several ctxs)
    bool post_due = false;                   // its header copy + completion event are still to be enqueued (by mpfmt_group_end)
    int64_t my_words = 0, my_nnz = 0;
    uint64_t* snap = nullptr;                // this rank's mask as it was at _launch: a repeat at a larger capacity re-sends THAT step
    int64_t snap_words = 0;
    // generic all-gather staging (wavefront triples): [world][slot]
    void* abuf = nullptr;
    size_t abuf_bytes = 0;
};

static int32_t comm_of(mpfmt_ctx* ctx, mpfmt_comm** out)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!ctx->comm) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no communicator (mpfmt_comm_create)");
    *out = (mpfmt_comm*)ctx->comm;
    return MPFMT_OK;
}

__global__ void k_pack_mask(const uint64_t* __restrict__ mask, int64_t words, int64_t nnz, int64_t cap, uint64_t* __restrict__ slot)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { slot[0] = (uint64_t)words; slot[1] = (uint64_t)nnz; }
    if (i < cap) slot[2 + i] = (i < words) ? mask[i] : 0ull;      // zero padded, as the interface promises; words > cap: the
                                                                  // header carries the true length and _finish repeats
}

// all-gather of `bytes` per rank from the communicator's staging buffer; send data must already be in slot `rank`
int32_t mpfmt_comm_allgather_inplace(mpfmt_ctx* ctx, void* buf, size_t bytes_per_rank, hipStream_t stream)
{
    mpfmt_comm* c;
    int32_t rc;
    if ((rc = comm_of(ctx, &c))) return rc;
    NCCLCHK(ctx, g_rccl.AllGather((const char*)buf + (size_t)c->rank * bytes_per_rank, buf, bytes_per_rank, ncclChar, c->comm, stream));
    return MPFMT_OK;
}

int32_t mpfmt_comm_world(const mpfmt_ctx* ctx, int* rank, int* world)
{
    if (ctx->comm) { const mpfmt_comm* c = (const mpfmt_comm*)ctx->comm; *rank = c->rank; *world = c->world; return 1; }
    *rank = ctx->rank; *world = ctx->world;
    return 0;
}

// A group belongs to the host thread that opened it (ncclGroupStart / End are per thread too): both are thread_local, so one thread
// per ctx beside a grouping thread cannot race on them
static thread_local int g_group_depth = 0;                          // mpfmt_group_begin nesting of this thread
static thread_local std::vector<mpfmt_ctx*> g_group_post;           // ctxs whose gather was launched inside this thread's open group

// what follows the collective on the communication stream: slot headers (lengths) to pinned host memory, completion event
static int32_t gather_post(mpfmt_ctx* ctx, mpfmt_comm* c)
{
    if (!c || !c->post_due) return MPFMT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpy2DAsync(c->hdr_host, 2 * sizeof(int64_t), c->gbuf, (size_t)c->slot_last * sizeof(uint64_t), 2 * sizeof(int64_t),
                                 (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(ctx, hipEventRecord(c->ev_done, c->stream));
    c->post_due = false;
    return MPFMT_OK;
}

// pack this rank's snapshot into its slot (compute stream) and run the all-gather at capacity c->cap (communication stream)
static int32_t gather_issue(mpfmt_ctx* ctx, mpfmt_comm* c)
{
    const int64_t slot = c->cap + 2;
    if (c->gbuf_words < slot * c->world) {
        if (c->gbuf) { HIPCHK(ctx, hipStreamSynchronize(c->stream)); HIPCHK(ctx, hipFree(c->gbuf)); c->gbuf = nullptr; }
        HIPCHK(ctx, hipMalloc((void**)&c->gbuf, sizeof(uint64_t) * (size_t)(slot * c->world)));
        c->gbuf_words = slot * c->world;
    }
    // pack on the compute stream (after the sweep and the snapshot), exchange on the communication stream
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, c->ev_done, 0));        // the previous gather has left the buffer
    uint64_t* mine = c->gbuf + (size_t)c->rank * slot;
    hipLaunchKernelGGL(k_pack_mask, dim3((unsigned)((std::max<int64_t>(c->cap, 1) + 255) / 256)), dim3(256), 0, ctx->stream,
                       c->snap, c->my_words, c->my_nnz, c->cap, mine);
    HIPCHK(ctx, hipGetLastError());
    c->slot_last = slot;
    HIPCHK(ctx, hipEventRecord(c->ev_compute, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(c->stream, c->ev_compute, 0));
    NCCLCHK(ctx, g_rccl.AllGather(mine, c->gbuf, (size_t)slot, ncclUint64, c->comm, c->stream));
    c->post_due = true;
    c->grouped = g_group_depth > 0;
    if (c->grouped) g_group_post.push_back(ctx);                        // the collective is enqueued at mpfmt_group_end: so is what follows it
    else { int32_t rc; if ((rc = gather_post(ctx, c))) return rc; }
    c->pending = true;
    return MPFMT_OK;
}

extern "C" {

int32_t mpfmt_comm_unique_id(uint8_t* id128)
{
    if (!id128) return MPFMT_ERR_ARG;
    int32_t rc;
    if ((rc = rccl_ready(nullptr))) return rc;
    static_assert(sizeof(ncclUniqueId) == MPFMT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCLCHK(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return MPFMT_OK;
}

int32_t mpfmt_group_begin(void)
{
    int32_t rc;
    if ((rc = rccl_ready(nullptr))) return rc;
    NCCLCHK(nullptr, g_rccl.GroupStart());
    ++g_group_depth;
    return MPFMT_OK;
}

// Inside ncclGroupStart / ncclGroupEnd a collective is only ENQUEUED at ncclGroupEnd: whatever must follow it on the stream (the
// copy of the slot headers to the host, the completion event) is enqueued here, after the group has closed -- issued right
// behind the ncclAllGather call it would run BEFORE the collective.
int32_t mpfmt_group_end(void)
{
    int32_t rc;
    if ((rc = rccl_ready(nullptr))) return rc;
    if (g_group_depth <= 0) return mpfmt_fail(nullptr, MPFMT_ERR_STATE, "mpfmt_group_end without mpfmt_group_begin on this thread");
    --g_group_depth;
    const ncclResult_t gr = g_rccl.GroupEnd();
    if (g_group_depth > 0 && gr == ncclSuccess) return MPFMT_OK;
    // the outermost group has closed (or RCCL refused it): the list is emptied on EVERY path, every ctx is posted -- a failure on one
    // must not leave the others with post_due set for good -- and the first error is what the caller sees
    std::vector<mpfmt_ctx*> due;
    due.swap(g_group_post);
    if (gr != ncclSuccess) {
        g_group_depth = 0;
        for (mpfmt_ctx* ctx : due) { mpfmt_comm* c = (mpfmt_comm*)ctx->comm; if (c) { c->post_due = false; c->pending = false; } }
        return mpfmt_fail(nullptr, MPFMT_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(gr));
    }
    int32_t first = MPFMT_OK;
    for (mpfmt_ctx* ctx : due) {
        rc = gather_post(ctx, (mpfmt_comm*)ctx->comm);
        if (rc && !first) first = rc;
    }
    return first;
}

int32_t mpfmt_comm_create(mpfmt_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id128)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!id128) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "id128 is NULL");
    if (world < 1 || rank < 0 || rank >= world) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "bad rank %d of %d", rank, world);
    if (ctx->comm) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "ctx already has a communicator");
    int32_t rc;
    if ((rc = rccl_ready(ctx))) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    mpfmt_comm* c = new mpfmt_comm();
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return mpfmt_fail(ctx, MPFMT_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->hdr_host, sizeof(int64_t) * 2 * (size_t)world);
    if (e == hipSuccess) e = hipMalloc((void**)&c->hdr_dev, sizeof(int64_t) * 2 * (size_t)world);
    if (e != hipSuccess) {
        ctx->comm = c;
        mpfmt_comm_destroy(ctx);
        return mpfmt_fail(ctx, MPFMT_ERR_HIP, "communicator resources: %s", hipGetErrorString(e));
    }
    ctx->comm = c;
    return mpfmt_set_shard(ctx, rank, world);
}

int32_t mpfmt_comm_destroy(mpfmt_ctx* ctx)
{
    if (!ctx || !ctx->comm) return MPFMT_OK;
    mpfmt_comm* c = (mpfmt_comm*)ctx->comm;
    // a ctx destroyed inside an open group must not be posted at group_end: the opener's list is thread-local, so only the opening
    // thread can take the ctx out of it -- from any other thread the destroy is refused while the group's post is due
    const size_t before = g_group_post.size();
    g_group_post.erase(std::remove(g_group_post.begin(), g_group_post.end(), ctx), g_group_post.end());
    if (c->post_due && c->grouped && g_group_post.size() == before)
        return mpfmt_fail(ctx, MPFMT_ERR_STATE, "communicator destroyed from another thread while its gather waits in an open group (close the group first)");
    hipSetDevice(ctx->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->gbuf) hipFree(c->gbuf);
    if (c->snap) hipFree(c->snap);
    if (c->abuf) hipFree(c->abuf);
    if (c->hdr_dev) hipFree(c->hdr_dev);
    if (c->hdr_host) hipHostFree(c->hdr_host);
    if (c->ev_compute) hipEventDestroy(c->ev_compute);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    ctx->comm = nullptr;
    return MPFMT_OK;
}

// Capacity agreement: every rank must use the same per-rank word count.  cap_hint > 0: the caller guarantees the value is
// the same on every rank (a single-thread driver that has seen all shards).  cap_hint == 0: the first exchange gathers
// the lengths alone (blocking), afterwards the lengths ride in front of the payload and every rank re-derives the same
// capacity from what it saw.
int32_t mpfmt_allgather_free_mask_launch(mpfmt_ctx* ctx, int64_t cap_hint)
{
    mpfmt_comm* c;
    int32_t rc;
    if ((rc = comm_of(ctx, &c))) return rc;
    if (!ctx->graph_swept) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no swept graph (mpfmt_graph_step_device)");
    if (c->pending) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "a mask gather is already in flight (call _finish)");
    if (cap_hint < 0) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "cap_hint < 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t words = (ctx->nnz + 63) / 64;
    if (cap_hint > 0) {
        c->cap = cap_hint;
    } else if (c->cap < 0) {
        if (g_group_depth > 0)
            return mpfmt_fail(ctx, MPFMT_ERR_STATE, "inside mpfmt_group_begin / _end the first mask gather needs cap_hint > 0 (the blocking "
                                                    "lengths exchange would wait on a peer this thread has yet to launch)");
        int64_t mine[2] = {words, ctx->nnz};
        HIPCHK(ctx, hipMemcpyAsync(c->hdr_dev + 2 * c->rank, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
        NCCLCHK(ctx, g_rccl.AllGather(c->hdr_dev + 2 * c->rank, c->hdr_dev, 2, ncclInt64, c->comm, c->stream));
        HIPCHK(ctx, hipMemcpyAsync(c->hdr_host, c->hdr_dev, sizeof(int64_t) * 2 * c->world, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(ctx, hipStreamSynchronize(c->stream));
        int64_t mx = 0;
        for (int g = 0; g < c->world; ++g) mx = std::max(mx, c->hdr_host[2 * g]);
        c->cap = mx + mx / 20 + 8;
    }
    // snapshot of this rank's mask and lengths: the overlap protocol runs the NEXT step's kernels before _finish, so a repeat at a
    // larger capacity must not read the ctx's live mask (it would gather step k+1 under step k's name)
    c->my_words = words; c->my_nnz = ctx->nnz;
    if (c->snap_words < std::max<int64_t>(words, 1)) {
        if (c->snap) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(c->snap)); c->snap = nullptr; }
        const int64_t want = std::max<int64_t>(words + words / 8, 64);
        HIPCHK(ctx, hipMalloc((void**)&c->snap, sizeof(uint64_t) * (size_t)want));
        c->snap_words = want;
    }
    if (words > 0) HIPCHK(ctx, hipMemcpyAsync(c->snap, ctx->graph_free, sizeof(uint64_t) * (size_t)words, hipMemcpyDeviceToDevice, ctx->stream));
    return gather_issue(ctx, c);
}

// the second half of a repeat for single-thread drivers: _finish returned MPFMT_RETRY on every ctx (all saw the same lengths and
// adopted the same larger capacity); the driver calls this for every ctx inside mpfmt_group_begin / _end, then _finish again
int32_t mpfmt_allgather_free_mask_relaunch(mpfmt_ctx* ctx)
{
    mpfmt_comm* c;
    int32_t rc;
    if ((rc = comm_of(ctx, &c))) return rc;
    if (c->pending) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "a mask gather is already in flight (call _finish)");
    if (c->cap < 0 || !c->snap) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "nothing to relaunch (no mask gather has been launched)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return gather_issue(ctx, c);
}

int32_t mpfmt_allgather_free_mask_finish(mpfmt_ctx* ctx, void** gathered, int64_t* stride_words, int64_t* words_each, int64_t* nnz_each)
{
    mpfmt_comm* c;
    int32_t rc;
    if ((rc = comm_of(ctx, &c))) return rc;
    if (!c->pending) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no mask gather in flight");
    if (c->post_due) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "mask gather launched inside a group that is still open (mpfmt_group_end first)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(c->ev_done));
    c->pending = false;
    int64_t mx = 0;
    for (int g = 0; g < c->world; ++g) mx = std::max(mx, c->hdr_host[2 * g]);
    if (mx > c->cap) {
        // a shard outgrew the agreed capacity: every rank sees the same lengths, so all of them repeat at the exact size -- from
        // the snapshots taken at _launch, i.e. the same step's masks
        c->cap = mx + mx / 20 + 8;
        if (c->grouped) return MPFMT_RETRY;                             // single-thread driver: relaunch all ctxs in a group, then _finish
        if ((rc = gather_issue(ctx, c))) return rc;
        HIPCHK(ctx, hipEventSynchronize(c->ev_done));
        c->pending = false;
        mx = 0;
        for (int g = 0; g < c->world; ++g) mx = std::max(mx, c->hdr_host[2 * g]);
        if (mx > c->cap) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "mask gather: a rank reports %lld words after the repeat at capacity %lld", (long long)mx, (long long)c->cap);
    } else if (mx < c->cap / 2) {
        c->cap = mx + mx / 20 + 8;            // shards shrank a lot: tighten for the next step (same decision on every rank)
    }
    for (int g = 0; g < c->world; ++g) {
        if (words_each) words_each[g] = c->hdr_host[2 * g];
        if (nnz_each) nnz_each[g] = c->hdr_host[2 * g + 1];
    }
    if (gathered) *gathered = c->gbuf;
    if (stride_words) *stride_words = c->slot_last;
    return MPFMT_OK;
}

int32_t mpfmt_allgather_free_mask(mpfmt_ctx* ctx, void** gathered, int64_t* stride_words, int64_t* words_each, int64_t* nnz_each)
{
    int32_t rc;
    if ((rc = mpfmt_allgather_free_mask_launch(ctx, 0))) return rc;
    return mpfmt_allgather_free_mask_finish(ctx, gathered, stride_words, words_each, nnz_each);
}

}  // extern "C"
