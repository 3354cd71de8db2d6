// Test double of the RCCL entry points libmpfmt.so resolves at run time (mpfmt_comm.hip), so that the MULTI-PROCESS exchange
// code -- capacity agreement, launch / finish overlap, the growth path, the per-wavefront triple exchange -- can run with several
// ranks on a box that has ONE GPU (real RCCL refuses two ranks on a device).  Ranks meet in a POSIX shared-memory segment named
// by the unique id; an all-gather stages every rank's bytes through it (device -> host -> segment -> host -> device) between two
// process-shared barriers.  Synchronous where RCCL is asynchronous (a stricter ordering, not a weaker one) -- except inside
// ncclGroupStart / ncclGroupEnd, where collectives are deferred to the end of the group exactly like RCCL's, so that a single
// thread can drive several ranks and the library's ordering of the work that follows a grouped collective is really tested.
// Built by tests/test_gpu_multirank.py:  g++ -shared -fPIC mock_rccl.cpp -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64
// Selected with MPFMT_RCCL_LIB=/path/to/librccl_mock.so.  Test infrastructure only -- never shipped, never measured.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6,
               ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
constexpr size_t SLOT = (size_t)96 << 20;          // bytes a rank may contribute to one collective
struct shared_hdr {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};
struct comm_t {
    int rank, nranks;
    std::string name;
    size_t bytes;
    char* base;
    shared_hdr* hdr() { return (shared_hdr*)base; }
    char* slot(int r) { return base + 4096 + (size_t)r * SLOT; }
    std::vector<char> host;
};
int g_depth = 0;                               // ncclGroupStart nesting
size_t dtype_size(ncclDataType_t t) { return t <= 1 ? 1 : t <= 3 ? 4 : t <= 5 ? 8 : t == 6 ? 2 : t == 7 ? 4 : 8; }
void barrier(comm_t* c)
{
    shared_hdr* h = c->hdr();
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == c->nranks) { h->arrived.store(0); h->generation.fetch_add(1); }
    else while (h->generation.load() == gen) usleep(20);
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    memset(id, 0, sizeof *id);
    FILE* f = fopen("/dev/urandom", "rb");
    unsigned char r[12] = {0};
    if (f) { if (fread(r, 1, sizeof r, f) != sizeof r) r[0] = (unsigned char)getpid(); fclose(f); }
    char* p = id->internal;
    p += sprintf(p, "/mpfmt_mock_");
    for (unsigned char b : r) p += sprintf(p, "%02x", b);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank)
{
    comm_t* c = new comm_t();
    c->rank = rank; c->nranks = nranks; c->name = std::string(id.internal, strnlen(id.internal, 127));
    c->bytes = 4096 + (size_t)nranks * SLOT;
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return ncclSystemError; }      // sparse: pages appear when touched
    c->base = (char*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) { delete c; return ncclSystemError; }
    c->hdr()->attached.fetch_add(1);
    // every rank is in before the first collective; inside a group (one thread creating several ranks) the peers follow in the
    // same thread, and the first collective's rendezvous does the waiting
    if (g_depth == 0) while (c->hdr()->attached.load() < nranks) usleep(100);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(void* comm)
{
    comm_t* c = (comm_t*)comm;
    if (!c) return ncclSuccess;
    if (c->hdr()->attached.fetch_sub(1) == 1) shm_unlink(c->name.c_str());
    munmap(c->base, c->bytes);
    delete c;
    return ncclSuccess;
}

// A collective called inside ncclGroupStart / ncclGroupEnd is only queued; ncclGroupEnd runs the queue -- all the local ranks'
// sends first, then the rendezvous, then the receives.  That is RCCL's ordering (work enqueued on the stream behind the call
// but before ncclGroupEnd runs BEFORE the collective), and it is what lets ONE thread drive several ranks without waiting on
// itself.
namespace {
struct pending_op { const void* send; void* recv; size_t n; comm_t* c; hipStream_t stream; };
std::vector<pending_op> g_queue;
void arrive(comm_t* c) { shared_hdr* h = c->hdr(); if (h->arrived.fetch_add(1) + 1 == c->nranks) { h->arrived.store(0); h->generation.fetch_add(1); } }
ncclResult_t run_ops(std::vector<pending_op>& ops)
{
    if (ops.empty()) return ncclSuccess;
    for (pending_op& o : ops) {
        if (o.n > SLOT) return ncclInvalidArgument;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy(o.c->slot(o.c->rank), o.send, o.n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    }
    for (int phase = 0; phase < 2; ++phase) {
        // every local rank arrives, then all wait for the generation to move (ranks of other processes arrive on their own)
        shared_hdr* h = ops[0].c->hdr();
        const int gen = h->generation.load();
        for (pending_op& o : ops) arrive(o.c);
        while (h->generation.load() == gen) usleep(20);
        if (phase == 0)
            for (pending_op& o : ops)
                for (int r = 0; r < o.c->nranks; ++r)
                    if (hipMemcpy((char*)o.recv + (size_t)r * o.n, o.c->slot(r), o.n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}
}  // namespace

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, void* comm, hipStream_t stream)
{
    pending_op o{send, recv, count * dtype_size(dt), (comm_t*)comm, stream};
    if (g_depth > 0) { g_queue.push_back(o); return ncclSuccess; }
    std::vector<pending_op> one{o};
    return run_ops(one);
}

ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, void*, hipStream_t) { return ncclInvalidArgument; }
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (g_depth > 0 && --g_depth == 0) { std::vector<pending_op> ops; ops.swap(g_queue); return run_ops(ops); }
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : r == ncclSystemError ? "mock: shared memory error" : r == ncclInvalidArgument ? "mock: invalid argument" : "mock: hip error"; }

}
