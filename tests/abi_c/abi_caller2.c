/* Second C caller of libmpfmt.so: the calls julia/MPFmtHIP.jl makes for the spaces and checkers beyond Euclidean + boxes --
 * LinearQuadratic (double integrator), Dubins / Reeds-Shepp cars, closest / closeR, the 2-D SAT world, the device sampler -- and
 * the per-step loop of a single thread driving its ctxs (hip_graph_step! / hip_gather_finish!), each with exactly the argument
 * widths of its `ccall` signature (Int32 -> int32_t, Int64 / Int -> int64_t, UInt64 -> uint64_t, Float64 -> double, Ptr{T} -> T*,
 * Ptr{Void} -> void*).  The typedefs are written from the Julia file, NOT from mpfmt.h; the build uses -Wcast-function-type
 * -Werror, so a width in the glue that is not the header's fails the build (pointer parameters match any pointer, integer widths
 * must agree).  tests/test_gpu_boundary.py runs it on the GPU box and compares what it wrote with the oracle.
 * usage: abi_caller2 <input.bin> <output.bin> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "mpfmt.h"

/* hip_graph_step! / hip_gather_finish! */
typedef int32_t (*f_step_launch)(void*, double);                                   /* (Ptr{Void}, Float64) */
typedef int32_t (*f_step_finish)(void*, int64_t*);                                 /* (Ptr{Void}, Ptr{Int64}) */
typedef int32_t (*f_gather_launch)(void*, int64_t);                                /* (Ptr{Void}, Int64) */
typedef int32_t (*f_gather_finish)(void*, void**, int64_t*, int64_t*, int64_t*);   /* (Ptr{Void}, Ptr{Ptr{Void}}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}) */
typedef int32_t (*f_gather_relaunch)(void*);                                       /* (Ptr{Void},) */
typedef int32_t (*f_group)(void);
typedef int32_t (*f_step_device)(void*, double, int64_t*);                         /* (Ptr{Void}, Float64, Ptr{Int64}) */
typedef int32_t (*f_pinned_alloc)(int64_t, void**);                                /* (Int64, Ptr{Ptr{Void}}) */
typedef int32_t (*f_pinned_free)(void*);                                           /* (Ptr{Void},) */
typedef int32_t (*f_export_pinned)(void*, int64_t**, int64_t**, double**, uint64_t**, int64_t*, double*);
                                                                                 /* (Ptr{Void}, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Float64}}, Ptr{Ptr{UInt64}}, Ptr{Int64}, Ptr{Float64}) */
typedef int32_t (*f_export)(void*, int64_t*, int64_t*, double*, uint64_t*, double*);   /* (Ptr{Void}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{UInt64}, Ptr{Float64}) */
/* helper_data_structures(V, ::LinearQuadratic), hip_di_edges_free */
typedef int32_t (*f_di_count)(void*, double, double, int64_t*, int64_t*);          /* (Ptr{Void}, Float64, Float64, Ptr{Int64}, Ptr{Int64}) */
typedef int32_t (*f_di_fill)(void*, int64_t*, double*, double*);                   /* (Ptr{Void}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}) */
typedef int32_t (*f_di_free)(void*, uint64_t*, uint8_t*);                          /* (Ptr{Void}, Ptr{UInt64}, Ptr{UInt8}) */
/* cars */
typedef int32_t (*f_car_count)(void*, double, double, double, int64_t*, int64_t*); /* (Ptr{Void}, Float64, Float64, Float64, Ptr{Int64}, Ptr{Int64}) */
typedef int32_t (*f_car_fill)(void*, int64_t*, double*);                           /* (Ptr{Void}, Ptr{Int64}, Ptr{Float64}) */
/* closest / closeR */
typedef int32_t (*f_closest)(void*, const double*, int64_t, const double*, double*, double*, int64_t*, int64_t*);
typedef int32_t (*f_closeR)(void*, const double*, int64_t, const double*, double, int64_t*, int64_t, int64_t*, double*, double*, int64_t*, int64_t*);
/* hip_upload_shapes! */
typedef int32_t (*f_upload_shapes)(void*, int32_t, const int32_t*, const int32_t*, const double*, const double*, const double*);
/* hip_sample_free! */
typedef int32_t (*f_sample_free)(void*, uint64_t, int64_t, const double*, int32_t, const double*, int32_t, double*, int64_t*);

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mpfmt_last_error((mpfmt_ctx*)ctx)); return 3; } } while (0)

static void put(FILE* f, const void* p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { perror("write"); exit(4); } }
static void get(FILE* f, void* p, size_t n) { if (n && fread(p, 1, n, f) != n) { fprintf(stderr, "short input\n"); exit(4); } }
static void* take(FILE* f, size_t n) { void* p = malloc(n ? n : 8); get(f, p, n); return p; }

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    f_step_launch step_launch = (f_step_launch)mpfmt_graph_step_launch;
    f_step_finish step_finish = (f_step_finish)mpfmt_graph_step_finish;
    f_gather_launch gather_launch = (f_gather_launch)mpfmt_allgather_free_mask_launch;
    f_gather_finish gather_finish = (f_gather_finish)mpfmt_allgather_free_mask_finish;
    f_gather_relaunch gather_relaunch = (f_gather_relaunch)mpfmt_allgather_free_mask_relaunch;
    f_group group_begin = mpfmt_group_begin, group_end = mpfmt_group_end;
    f_di_count di_count = (f_di_count)mpfmt_di_graph_count;
    f_di_fill di_fill = (f_di_fill)mpfmt_di_graph_fill;
    f_di_free di_free = (f_di_free)mpfmt_di_graph_edges_free;
    f_car_count rs_count = (f_car_count)mpfmt_reedsshepp_graph_count, db_count = (f_car_count)mpfmt_dubins_graph_count;
    f_car_fill rs_fill = (f_car_fill)mpfmt_reedsshepp_graph_fill, db_fill = (f_car_fill)mpfmt_dubins_graph_fill;
    f_closest closest = (f_closest)mpfmt_closest;
    f_closeR closeR = (f_closeR)mpfmt_closeR;
    f_upload_shapes upload_shapes = (f_upload_shapes)mpfmt_upload_shapes2d;
    f_sample_free sample_free = (f_sample_free)mpfmt_sample_free;
    f_step_device step_device = (f_step_device)mpfmt_graph_step_device;
    f_pinned_alloc pinned_alloc = (f_pinned_alloc)mpfmt_pinned_alloc;
    f_pinned_free pinned_free = (f_pinned_free)mpfmt_pinned_free;
    f_export graph_export = (f_export)mpfmt_graph_export;
    f_export_pinned graph_export_pinned = (f_export_pinned)mpfmt_graph_export_pinned;

    FILE* in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 2; }
    int64_t h[8];          /* N4, N3, Mb, nq, ns, ndata, Ns, seed */
    double p[8];           /* rho, r_di, rt, sp, r_car, r2, r_euclid, - */
    get(in, h, sizeof h); get(in, p, sizeof p);
    const int64_t N4 = h[0], N3 = h[1], Mb = h[2], nq = h[3], ns = h[4], ndata = h[5], Ns = h[6];
    const uint64_t seed = (uint64_t)h[7];
    double* X4 = take(in, 8 * N4 * 4); double* X3 = take(in, 8 * N3 * 3); double* lohi2 = take(in, 8 * Mb * 4);
    double* lo4 = take(in, 32); double* hi4 = take(in, 32); double* lo3 = take(in, 24); double* hi3 = take(in, 24);
    double* lo2 = take(in, 16); double* hi2 = take(in, 16);
    double* Pq = take(in, 8 * nq * 2); double* W = take(in, 32);
    int32_t* kinds = take(in, 4 * ns); int32_t* nverts = take(in, 4 * ns); double* data = take(in, 8 * ndata);
    double* init2 = take(in, 16); double* goal = take(in, 24);
    fclose(in);
    FILE* out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }

    mpfmt_ctx* ctx = NULL;
    if (mpfmt_ctx_create(0, &ctx) != 0) { fprintf(stderr, "ctx_create: %s\n", mpfmt_last_error(NULL)); return 3; }

    /* ---- double integrator: states in R^4, boxes in the 2-D workspace ---- */
    CHECK(mpfmt_upload_samples(ctx, X4, N4, 4));
    CHECK(mpfmt_upload_boxes(ctx, lohi2, (int32_t)Mb, 2, lo4, hi4, 4));
    {
        int64_t* colptr = malloc(8 * (N4 + 1)); int64_t nnz = 0;
        CHECK(di_count(ctx, p[0], p[1], colptr, &nnz));
        int64_t* rowval = malloc(8 * (nnz ? nnz : 1)); double* nzval = malloc(8 * (nnz ? nnz : 1)); double* tval = malloc(8 * (nnz ? nnz : 1));
        CHECK(di_fill(ctx, rowval, nzval, tval));
        const int64_t words = (nnz + 63) / 64;
        uint64_t* fr = calloc(words ? words : 1, 8); uint8_t* nseg = calloc(nnz ? nnz : 1, 1);
        CHECK(di_free(ctx, fr, nseg));
        put(out, &nnz, 8); put(out, colptr, 8 * (N4 + 1)); put(out, rowval, 8 * nnz); put(out, nzval, 8 * nnz); put(out, tval, 8 * nnz);
        put(out, fr, 8 * words); put(out, nseg, nnz);
    }
    /* ---- cars: SE2 states, the same 2-D boxes ---- */
    CHECK(mpfmt_upload_samples(ctx, X3, N3, 3));
    CHECK(mpfmt_upload_boxes(ctx, lohi2, (int32_t)Mb, 2, lo3, hi3, 3));
    for (int car = 0; car < 2; ++car) {
        int64_t* colptr = malloc(8 * (N3 + 1)); int64_t nnz = 0;
        CHECK((car ? rs_count : db_count)(ctx, p[2], p[3], p[4], colptr, &nnz));
        int64_t* rowval = malloc(8 * (nnz ? nnz : 1)); double* nzval = malloc(8 * (nnz ? nnz : 1));
        CHECK((car ? rs_fill : db_fill)(ctx, rowval, nzval));
        put(out, &nnz, 8); put(out, colptr, 8 * (N3 + 1)); put(out, rowval, 8 * nnz); put(out, nzval, 8 * nnz);
    }
    /* ---- closest / closeR against the 2-D boxes ---- */
    {
        double* q2 = malloc(8 * nq * 2);
        memcpy(q2, Pq, 8 * nq * 2);
        CHECK(mpfmt_upload_samples(ctx, q2, nq, 2));
        CHECK(mpfmt_upload_boxes(ctx, lohi2, (int32_t)Mb, 2, lo2, hi2, 2));
        double* d2 = malloc(8 * nq); double* v = malloc(8 * nq * 2); int64_t* k = malloc(8 * nq); int64_t fails = 0;
        CHECK(closest(ctx, Pq, nq, W, d2, v, k, &fails));
        put(out, d2, 8 * nq); put(out, v, 8 * nq * 2); put(out, k, 8 * nq); put(out, &fails, 8);
        const int64_t cap = nq * Mb;
        int64_t* ptr = malloc(8 * (nq + 1)); int64_t* ob = malloc(8 * (cap ? cap : 1)); double* dd = malloc(8 * (cap ? cap : 1));
        double* vv = malloc(8 * (cap ? cap : 1) * 2); int64_t total = 0;
        CHECK(closeR(ctx, Pq, nq, W, p[5], ptr, cap, ob, dd, vv, &total, &fails));
        put(out, &total, 8); put(out, ptr, 8 * (nq + 1)); put(out, ob, 8 * total); put(out, dd, 8 * total); put(out, vv, 8 * total * 2);
    }
    /* ---- sampler (boxes, Identity workspace), then the single-thread step loop on what it left uploaded ---- */
    {
        double* Xs = malloc(8 * Ns * 2); int64_t attempts = 0;
        CHECK(sample_free(ctx, seed, Ns, init2, 1 /* MPFMT_GOAL_BALL */, goal, 3, Xs, &attempts));
        put(out, Xs, 8 * Ns * 2); put(out, &attempts, 8);
        uint8_t id[128];
        CHECK(mpfmt_comm_unique_id(id));
        CHECK(group_begin());
        CHECK(mpfmt_comm_create(ctx, 0, 1, id));
        CHECK(group_end());
        for (int step = 0; step < 2; ++step) {
            int64_t nnz = 0, stride = 0, words = 0, nn = 0;
            void* gathered = NULL;
            CHECK(step_launch(ctx, p[6] * (step ? 1.5 : 1.0)));            /* second step: a larger radius (the gather grows) */
            CHECK(step_finish(ctx, &nnz));
            CHECK(group_begin());
            CHECK(gather_launch(ctx, step == 0 ? (nnz + 63) / 64 + 64 : 0));
            CHECK(group_end());
            int32_t rc = gather_finish(ctx, &gathered, &stride, &words, &nn);
            int64_t retried = 0;
            if (rc == MPFMT_RETRY) {
                retried = 1;
                CHECK(group_begin());
                CHECK(gather_relaunch(ctx));
                CHECK(group_end());
                rc = gather_finish(ctx, &gathered, &stride, &words, &nn);
            }
            if (rc != 0) { fprintf(stderr, "gather_finish -> %d: %s\n", rc, mpfmt_last_error(ctx)); return 3; }
            put(out, &nnz, 8); put(out, &words, 8); put(out, &nn, 8); put(out, &retried, 8);
        }
        CHECK(mpfmt_comm_destroy(ctx));
        /* ---- hip_precompute_step!: one step, one export into page-locked arrays (the sampled set is still uploaded) ---- */
        {
            int64_t nnz = 0;
            CHECK(step_device(ctx, p[6], &nnz));
            const int64_t words = (nnz + 63) / 64;
            void* pp[4] = {NULL, NULL, NULL, NULL};
            const int64_t bytes[4] = {8 * (Ns + 1), 8 * (nnz ? nnz : 1), 8 * (nnz ? nnz : 1), 8 * (words ? words : 1)};
            for (int k = 0; k < 4; ++k) if (pinned_alloc(bytes[k], &pp[k]) != 0) { fprintf(stderr, "pinned_alloc failed\n"); return 3; }
            double rate = 0.0;
            CHECK(graph_export(ctx, (int64_t*)pp[0], (int64_t*)pp[1], (double*)pp[2], (uint64_t*)pp[3], &rate));
            put(out, &nnz, 8); put(out, pp[0], 8 * (Ns + 1)); put(out, pp[1], 8 * nnz); put(out, pp[2], 8 * nnz); put(out, pp[3], 8 * words);
            for (int k = 0; k < 4; ++k) if (pinned_free(pp[k]) != 0) { fprintf(stderr, "pinned_free failed\n"); return 3; }
            /* hip_precompute_step!: the same export into the ctx's own page-locked arena, twice -- the second call must hand out the
             * same memory (the arena lives as long as the ctx) and the same graph */
            int64_t *ac = NULL, *ar = NULL; double* av = NULL; uint64_t* am = NULL; int64_t annz = -1;
            CHECK(graph_export_pinned(ctx, &ac, &ar, &av, &am, &annz, &rate));
            int64_t *bc = NULL, *br = NULL; double* bv = NULL; uint64_t* bm = NULL; int64_t bnnz = -1;
            CHECK(graph_export_pinned(ctx, &bc, &br, &bv, &bm, &bnnz, &rate));
            const int64_t same = (ac == bc && ar == br && av == bv && am == bm && annz == nnz && bnnz == nnz) ? 1 : 0;
            void* arena = NULL;
            CHECK(mpfmt_export_arena(ctx, 64, &arena));
            const int64_t base_same = ((void*)ac == arena) ? 1 : 0;
            put(out, &same, 8); put(out, &base_same, 8);
            put(out, bc, 8 * (Ns + 1)); put(out, br, 8 * nnz); put(out, bv, 8 * nnz); put(out, bm, 8 * words);
        }
    }
    /* ---- 2-D SAT world: shapes uploaded, point and segment validity on the query points ---- */
    {
        CHECK(upload_shapes(ctx, (int32_t)ns, kinds, nverts, data, lo2, hi2));
        const int64_t wp = (nq + 63) / 64, we = (nq - 1 + 63) / 64;
        uint64_t* mp_ = calloc(wp ? wp : 1, 8); uint64_t* me = calloc(we ? we : 1, 8);
        CHECK(mpfmt_states_free(ctx, Pq, nq, mp_));
        CHECK(mpfmt_motions_free(ctx, Pq, Pq + 2, nq - 1, me));
        put(out, mp_, 8 * wp); put(out, me, 8 * we);
    }
    fclose(out);
    CHECK(mpfmt_ctx_destroy(ctx));
    printf("abi_caller2 ok\n");
    return 0;
}
