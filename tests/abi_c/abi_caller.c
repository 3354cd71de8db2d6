/* A C caller of libmpfmt.so that makes every call julia/MPFmtHIP.jl makes, with exactly the argument widths of its `ccall`
 * signatures (Int32 -> int32_t, Int64 / Int -> int64_t, Float64 -> double, Ptr{T} -> T*, Ptr{Void} -> void*).  The typedefs
 * below are written from the Julia file, NOT from mpfmt.h; casting the library's symbols to them under -Wcast-function-type -Werror
 * is a compile-time check that the glue's widths are the header's (pointer parameters match any pointer, integer and float
 * widths and the argument count must agree -- tests/test_gpu_boundary.py also checks that a wrong width does fail the build).
 * tests/test_gpu_boundary.py builds this with gcc, runs it on the GPU box and compares what it wrote with the oracle.
 * (abi_caller2.c does the same for the glue of the other spaces and checkers.)
 * usage: abi_caller <input.bin> <output.bin> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "mpfmt.h"

typedef struct { int32_t status; double cost; int64_t z, collision_checks, path_len, nnz; double ms_graph, ms_sweep, ms_host_loop; } FmtResult;
typedef struct { int32_t done, nz, nx, nconn, ntrip; int64_t iters, checks; double cmin; int64_t tot_z, tot_x, tot_conn; } WfInfo;

/* (Int32, Ptr{Ptr{Void}}) */
typedef int32_t (*f_ctx_create)(int32_t, void**);
/* (Ptr{Void}, Ptr{Float64}, Int64, Int32) */
typedef int32_t (*f_upload_samples)(void*, const double*, int64_t, int32_t);
/* (Ptr{Void}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Int32) */
typedef int32_t (*f_upload_boxes)(void*, const double*, int32_t, int32_t, const double*, const double*, int32_t);
/* (Ptr{Void}, Int64, Float64, Ptr{Int64}, Ptr{Float64}, Int64, Ptr{Int64}) */
typedef int32_t (*f_rdisc_query)(void*, int64_t, double, int64_t*, double*, int64_t, int64_t*);
/* (Ptr{Void}, Float64, Ptr{Int64}, Ptr{Int64}) */
typedef int32_t (*f_rdisc_count)(void*, double, int64_t*, int64_t*);
/* (Ptr{Void}, Ptr{Int64}, Ptr{Float64}) */
typedef int32_t (*f_rdisc_fill)(void*, int64_t*, double*);
/* (Ptr{Void}, Ptr{UInt64}) */
typedef int32_t (*f_graph_edges_free)(void*, uint64_t*);
/* (Ptr{Void}, Float64, Int64, Int32, Int32, Ptr{Float64}, Float64, Int32, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}, Ptr{FmtResult}, Ptr{WfInfo}) */
typedef int32_t (*f_fmtstar_wavefront)(void*, double, int64_t, int32_t, int32_t, const double*, double, int32_t, int64_t*, double*,
                                       int64_t*, FmtResult*, WfInfo*);
/* (Ptr{UInt8},) ; () ; (Ptr{Void}, Int32, Int32, Ptr{UInt8}) */
typedef int32_t (*f_comm_unique_id)(uint8_t*);
typedef int32_t (*f_group)(void);
typedef int32_t (*f_comm_create)(void*, int32_t, int32_t, const uint8_t*);

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mpfmt_last_error((mpfmt_ctx*)ctx)); return 3; } } while (0)

static void put(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) { perror("write"); exit(4); } }
static void get(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short input\n"); exit(4); } }

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    /* the struct layouts the glue declares (immutable FmtResult / WfInfo) must be the library's */
    _Static_assert(sizeof(FmtResult) == sizeof(mpfmt_fmt_result), "FmtResult layout");
    _Static_assert(sizeof(WfInfo) == sizeof(mpfmt_wf_info), "WfInfo layout");
    f_ctx_create ctx_create = (f_ctx_create)mpfmt_ctx_create;
    f_upload_samples upload_samples = (f_upload_samples)mpfmt_upload_samples;
    f_upload_boxes upload_boxes = (f_upload_boxes)mpfmt_upload_boxes;
    f_rdisc_query rdisc_query = (f_rdisc_query)mpfmt_rdisc_query;
    f_rdisc_count rdisc_count = (f_rdisc_count)mpfmt_rdisc_count;
    f_rdisc_fill rdisc_fill = (f_rdisc_fill)mpfmt_rdisc_fill;
    f_graph_edges_free graph_edges_free = (f_graph_edges_free)mpfmt_graph_edges_free;
    f_fmtstar_wavefront fmtstar_wavefront = (f_fmtstar_wavefront)mpfmt_fmtstar_wavefront;
    f_comm_unique_id comm_unique_id = mpfmt_comm_unique_id;
    f_group group_begin = mpfmt_group_begin, group_end = mpfmt_group_end;
    f_comm_create comm_create = (f_comm_create)mpfmt_comm_create;

    FILE* in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 2; }
    int64_t N, d, M;
    double r, band;
    get(in, &N, 8); get(in, &d, 8); get(in, &M, 8); get(in, &r, 8); get(in, &band, 8);
    double* X = malloc(8 * N * d); double* lohi = malloc(8 * (M ? M : 1) * 2 * d); double* lo = malloc(8 * d); double* hi = malloc(8 * d);
    double* goal = malloc(8 * (d + 1));
    get(in, X, 8 * N * d); get(in, lohi, 8 * M * 2 * d); get(in, lo, 8 * d); get(in, hi, 8 * d); get(in, goal, 8 * (d + 1));
    fclose(in);

    void* ctx = NULL;
    int32_t rc = ctx_create(0, &ctx);
    if (rc != 0) { fprintf(stderr, "ctx_create -> %d: %s\n", rc, mpfmt_last_error(NULL)); return 3; }
    CHECK(upload_samples(ctx, X, N, (int32_t)d));
    CHECK(upload_boxes(ctx, lohi, (int32_t)M, (int32_t)d, lo, hi, (int32_t)d));

    FILE* out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }
    /* inball(V, dist, DS, v, r) for v = 1 and v = N */
    int64_t* inds = malloc(8 * N); double* ds = malloc(8 * N); int64_t k = 0;
    for (int64_t v = 1; v <= N; v += N - 1) {
        CHECK(rdisc_query(ctx, v, r, inds, ds, N, &k));
        put(out, &k, 8); put(out, inds, 8 * k); put(out, ds, 8 * k);
    }
    /* hip_neighbor_graph + precompute!'s edge bits */
    int64_t* colptr = malloc(8 * (N + 1)); int64_t nnz = 0;
    CHECK(rdisc_count(ctx, r, colptr, &nnz));
    int64_t* rowval = malloc(8 * (nnz ? nnz : 1)); double* nzval = malloc(8 * (nnz ? nnz : 1));
    CHECK(rdisc_fill(ctx, rowval, nzval));
    int64_t words = (nnz + 63) / 64;
    uint64_t* freeb = calloc(words ? words : 1, 8);
    CHECK(graph_edges_free(ctx, freeb));
    put(out, &nnz, 8); put(out, colptr, 8 * (N + 1)); put(out, rowval, 8 * nnz); put(out, nzval, 8 * nnz); put(out, freeb, 8 * words);
    /* fmtstar_hip!(P, r; single = true) and with a band */
    int64_t* A = malloc(8 * N); double* C = malloc(8 * N); int64_t* path = malloc(8 * N);
    for (int pass = 0; pass < 2; ++pass) {
        FmtResult res; WfInfo info;
        CHECK(fmtstar_wavefront(ctx, r, 1, 1, 1, goal, pass ? band : 0.0, pass ? 0 : 1, A, C, path, &res, &info));
        put(out, &res, sizeof res); put(out, &info, sizeof info);
        put(out, A, 8 * N); put(out, C, 8 * N); put(out, path, 8 * res.path_len);
    }
    /* hip_comm_create! with one ctx: ncclGroupStart / CommInitRank / GroupEnd from this one thread, then a step and the gather */
    uint8_t id[128];
    CHECK(comm_unique_id(id));
    CHECK(group_begin());
    CHECK(comm_create(ctx, 0, 1, id));
    CHECK(group_end());
    int64_t nnz2 = 0, stride = 0, wcount = 0, ncount = 0;
    void* gathered = NULL;
    CHECK(mpfmt_graph_step_launch((mpfmt_ctx*)ctx, r));
    CHECK(mpfmt_graph_step_finish((mpfmt_ctx*)ctx, &nnz2));
    CHECK(mpfmt_allgather_free_mask_launch((mpfmt_ctx*)ctx, words + 16));
    CHECK(mpfmt_allgather_free_mask_finish((mpfmt_ctx*)ctx, &gathered, &stride, &wcount, &ncount));
    put(out, &nnz2, 8); put(out, &stride, 8); put(out, &wcount, 8); put(out, &ncount, 8);
    fclose(out);
    CHECK(mpfmt_ctx_destroy((mpfmt_ctx*)ctx));
    printf("abi_caller ok: N=%lld nnz=%lld\n", (long long)N, (long long)nnz);
    return 0;
}
