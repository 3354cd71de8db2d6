// Device-resident wavefront FMT* (gfx950): the reference's dynamic-programming recursion (src/planners/fmt.jl:43-101)
// with W / H / C / A held in HBM and a whole cost band of open nodes expanded per step.
//
// The reference pops ONE lowest-cost open node z per iteration (fmt.jl:66,85-89) and runs the body fmt.jl:70-82 for it.
// Here one step takes the batch Z = { z in H : C[z] <= min_H C + band } and runs that body for every z of the batch
// against the SAME sets (W, H, C) -- the batch step SURVEY.md 7(5ii) / 8e calls a wavefront (Ichter, Schmerling, Pavone,
// "Group Marching Tree", 2017, is FMT* expanded this way):
//     for x in nearF(V, z, r, W), z in Z        -> mark: x unvisited, valid (checkpts), adjacent to the batch   (fmt.jl:70-71)
//        y_min = first argmin_{y in nearB(x) & H} C[y] + d(y, x)                                                (fmt.jl:72-74)
//        is_free_motion(V[y_min], V[x], CC, SS) -> A[x] = y_min, C[x] = c_min, W[x] = false, H_new += x         (fmt.jl:75-80)
//     H = (H \ Z) + H_new                        (the reference's deferred H update, fmt.jl:83-84, per batch)
//     stop when a batch holds a goal node (fmt.jl:68): the answer is the goal node of lowest (cost, index).
// With the batch forced to the single lowest (cost, index) open node (option `single`) every step IS one iteration of the
// reference loop, and tree, costs, path and collision_checks equal the sequential recursion's exactly; with a band the
// result of every step equals the batch form of the loop body (mpfmt_expand / the oracle's orc_expand) on the same sets.
// Edge checks are LAZY like the reference's (one per examined x per step): collision_checks counts what the loop asked for.
//
// Kernels per step (no host round trip inside a step; counters live in HBM, a `done` flag voids the kernels after the end):
//   k_wf_apply_min : H = (H & ~Z) | Hnew, clear Z / Hnew / cand; lexicographic minimum (C, index) over H -> per-block partials
//   k_wf_select    : reduce the partials; Z = nodes within the band (or the one minimum); batch list; goal test
//   k_wf_mark      : wavefront per batch node: mark unvisited valid neighbours once, append to the candidate list
//                    (sharded ctx: wavefront per OWNED unvisited sample, looks for a batch node in its own column)
//   k_wf_connect   : wavefront per candidate x: first-minimum over open neighbours (lexicographic wave reduce), then the
//                    edge test with LANE = OBSTACLE (boxes in LDS, predicates of sweep_predicates.h) or a bit of the swept
//                    mask; connect in place (1 GPU) or emit a (x, y_min, c_min) triple for the exchange (sharded)
//   k_wf_commit    : apply exchanged triples (sharded: after the all-gather every rank applies every rank's triples)
#include "sweep_predicates.h"
#include <cstring>
#include <cmath>
#include <chrono>
#include <algorithm>
#include <vector>

#define WF_MAXPARTS 1024
#define WF_MAXBLK 4096            // per-block statistics slots (no hot atomics: a single address sustains ~90 atomics / us)

struct wf_goal { int32_t kind; int32_t gd; double g[2 * MPFMT_MAX_DIM + 1]; };

struct wf_ctr {
    int32_t done;                 // 0 running, 2 open set exhausted, 1 goal in batch (set by k_wf_final; while running the
                                  // goal condition is goal_cbits != ~0, see wf_stop)
    int32_t ntrip;                // triples emitted this step (sharded)
    int32_t nz, nx;               // batch nodes / candidates of this step (lists zlist / xlist)
    int64_t iters;
    unsigned long long goal_cbits;   // lowest cost (as bits) of a goal node in the batch, ~0 = none
    int64_t final_z;              // resolved by k_wf_final
    double cmin;                  // lowest open cost at this step
    int64_t imin;
    int64_t tot[4];               // sums of the per-block statistics: batch nodes, samples examined, connected, edge checks
    int32_t nz_prev;              // batch size of the step before (k_wf_compact): k_wf_apply_min takes those nodes out of Hs
    int32_t ended;                // latched by the first k_wf_apply_min that finds a goal batch recorded: k_wf_select, which must not
};
enum { WF_NZ = 0, WF_NX = 1, WF_NCONN = 2, WF_CHECKS = 3 };

// the solve has ended: every kernel enqueued after that point returns at once and the sets stay as they were
__device__ __forceinline__ bool wf_stop(const wf_ctr* c) { return c->done != 0 || c->goal_cbits != ~0ull; }

struct wf_trip { int32_t x, y; double c; };
#define WF_XCAP 16384             // triples per rank and exchange round (256 KB): a wavefront rarely connects more per rank

struct mpfmt_wf {
    int64_t N = 0, words = 0;
    uint64_t *W = nullptr, *H = nullptr, *Z = nullptr, *Zp = nullptr, *Hn = nullptr, *cand = nullptr, *F = nullptr, *WF = nullptr;
    // The sets the mark / connect passes GATHER from, kept a second time by CELL-SORTED POSITION (unsharded Euclidean graphs): a column's
    // rows are spatial neighbours, so by position their bits sit in ~14 cache lines instead of ~103 by caller index (north star, measured
    // on the host) and their costs in ~60 instead of ~106.  WFs: unvisited and valid; Hs: open (Hns: opened during this step); cands:
    // candidates of this step; Cs: cost-to-come.  The graph's rows by position: ctx->rowpos.
    uint64_t *WFs = nullptr, *Hs = nullptr, *Hns = nullptr, *cands = nullptr;
    double* Cs = nullptr;
    int64_t pwords = 0;
    int32_t pos_space = 0;
    double* C = nullptr;
    int32_t* A = nullptr;
    int32_t *zlist = nullptr, *xlist = nullptr;                   // batch nodes / candidates of the step, compacted from the masks
    int64_t* rowptr = nullptr; int32_t* colidx = nullptr;         // directed cost graphs: forward sets (CSR of the resident CSC)
    int64_t csr_nnz = 0; bool directed = false;
    double* part_c = nullptr; int64_t* part_i = nullptr;          // per-block lexicographic minima of the open set
    double* last_c = nullptr; int64_t* last_i = nullptr;          // per-block lexicographic maxima of the batch (last node in pop order)
    int64_t* stats = nullptr;     // [WF_MAXBLK][4] per-block cumulative statistics
    double* boxT = nullptr;       // obstacle set transposed [2*d][mpad]: lane = obstacle reads are coalesced
    int mpad = 0, boxT_cap = 0;
    wf_trip* mytrips = nullptr;   // [N] connections of this rank in the current step (sharded)
    wf_trip* xbuf = nullptr;      // [world][WF_XCAP + 1] exchange slots: header (x = the rank's total count) + one round's triples
    wf_trip* hdr_host = nullptr;  // pinned [world] headers of the last exchange round
    int world_alloc = 0;
    wf_ctr* ctr = nullptr;        // device
    wf_ctr* ctr_host = nullptr;   // pinned
    int64_t prev_tot[4] = {0, 0, 0, 0};
    int64_t* path_dev = nullptr;
    wf_goal goal;
    double band = 0.0;
    int32_t single = 0, checkpts = 1, use_mask = 0, sharded = 0;
    int32_t all_in = 0;           // every sample lies inside the state space (the first-point test of statespaces.jl:155 is true for every edge)
    int64_t init = 0;
    double r = 0.0;
    bool active = false;
    int nparts = 1;
    std::chrono::steady_clock::time_point t_begin;
    double ms_graph = 0.0, ms_sweep = 0.0;
};

__device__ __forceinline__ bool wf_bit(const uint64_t* m, int64_t i) { return (m[i >> 6] >> (i & 63)) & 1ull; }

// src/goals.jl:96 (Rectangle), :100 (Ball), :111-114 (Point), Identity state2workspace
__device__ __forceinline__ bool wf_is_goal(const double* v, const wf_goal& G)
{
    const int d = G.gd;
    if (G.kind == MPFMT_GOAL_RECT) {
        bool ok = true;
        for (int i = 0; i < d; ++i) ok = ok && (G.g[i] <= v[i]) && (v[i] <= G.g[d + i]);
        return ok;
    }
    if (G.kind == MPFMT_GOAL_BALL) {
        double s = 0.0;
        for (int i = 0; i < d; ++i) { const double t = v[i] - G.g[i]; const double tt = t * t; s = (i == 0) ? tt : s + tt; }
        return sqrt(s) <= G.g[d];
    }
    bool ok = true;
    for (int i = 0; i < d; ++i) ok = ok && (v[i] == G.g[i]);
    return ok;
}

__device__ __forceinline__ void wf_lexmin_wave(double& c, int64_t& i)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(c, off);
        const int64_t oi = __shfl_xor(i, off);
        if (oi >= 0 && (i < 0 || oc < c || (oc == c && oi < i))) { c = oc; i = oi; }
    }
}
// the same, carrying a payload with the winner (the entry's row: no reload of rowval[be] after the reduce)
__device__ __forceinline__ void wf_lexmin_wave_y(double& c, int64_t& i, int32_t& y)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(c, off);
        const int64_t oi = __shfl_xor(i, off);
        const int32_t oy = __shfl_xor(y, off);
        if (oi >= 0 && (i < 0 || oc < c || (oc == c && oi < i))) { c = oc; i = oi; y = oy; }
    }
}
__device__ __forceinline__ void wf_lexmax_wave(double& c, int64_t& i)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(c, off);
        const int64_t oi = __shfl_xor(i, off);
        if (oi >= 0 && (i < 0 || oc > c || (oc == c && oi > i))) { c = oc; i = oi; }
    }
}

// Compaction of a bit mask into an index list, one word per lane: wave prefix sum of the popcounts, ONE atomic per wavefront
// on the list counter, every lane then writes the indices of its own word.
__device__ __forceinline__ void wf_append_word(unsigned long long m, int64_t w, int32_t* __restrict__ list, int32_t* counter)
{
    const int lane = threadIdx.x & 63;
    const int n = __popcll(m);
    int inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
    const int total = __shfl(inc, 63);
    if (total == 0) return;
    int base = 0;
    if (lane == 63) base = atomicAdd(counter, total);
    base = __shfl(base, 63);
    int o = base + inc - n;
    while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; list[o++] = (int32_t)(w * 64 + b); }
}

// statistics of one block: added to the block's own slot (the only writer of that slot; launches on a stream are ordered)
__device__ __forceinline__ void wf_block_stat(int64_t* stats, int which, int v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd((unsigned long long*)&stats[(blockIdx.x % WF_MAXBLK) * 4 + which], (unsigned long long)v);
}

__global__ __launch_bounds__(64) void k_wf_init(int64_t N, int64_t words, int64_t init, uint64_t* W, uint64_t* H, uint64_t* Z, uint64_t* Zp,
                                                uint64_t* Hn, uint64_t* cand, double* C, int32_t* A, int64_t* stats, wf_ctr* ctr)
{
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    for (int64_t w = t; w < words; w += (int64_t)gridDim.x * 64) {
        uint64_t full = ~0ull;
        if (w == words - 1 && (N & 63)) full = (1ull << (N & 63)) - 1;
        if (w == (init >> 6)) { W[w] = full & ~(1ull << (init & 63)); H[w] = 1ull << (init & 63); }
        else { W[w] = full; H[w] = 0; }
        Z[w] = 0; Zp[w] = 0; Hn[w] = 0; cand[w] = 0;
    }
    for (int64_t i = t; i < N; i += (int64_t)gridDim.x * 64) { C[i] = 0.0; A[i] = -1; }
    for (int64_t i = t; i < WF_MAXBLK * 4; i += (int64_t)gridDim.x * 64) stats[i] = 0;
    if (t == 0) {
        wf_ctr z;
        memset(&z, 0, sizeof z);
        z.goal_cbits = ~0ull; z.final_z = init; z.imin = -1;
        *ctr = z;
    }
}

// The open set is a bit mask; what a step needs of it is a cost per open node -- a gather of C[i] behind every set bit.  Walking a
// word's bits in one lane makes those gathers a dependent chain per lane (k_wf_select took 19 us a step that way, a fifth of the
// solve).  Instead a wavefront takes WF_GW words (lane = word), lays their set bits out as a node list in LDS (prefix sum of the
// popcounts; positions inside the group) and then works lane = node: every round of 64 gathers is in flight at once.
#define WF_GW 16                  // words per wavefront group (lanes 0 .. 15 hold one each): ~1000 groups at N = 1e6 instead of ~250 -- four times the gathers in flight
#define WF_GRP_CAP (WF_GW * 64)   // nodes of one group (every bit set)
__device__ __forceinline__ int wf_expand_group(unsigned long long m, uint16_t* __restrict__ s_list)
{
    const int lane = threadIdx.x & 63;
    const int n = __popcll(m);
    int inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
    const int total = __shfl(inc, 63);
    int o = inc - n;
    while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; s_list[o++] = (uint16_t)(lane * 64 + b); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return total;
}

// start of a step, position side: the nodes opened during the step before join Hs, its batch leaves it, the candidate mask is cleared
// (part of k_wf_apply_min's launch)
struct wf_pos { int64_t pwords; unsigned long long* Hs; uint64_t* Hns; uint64_t* cands; const int32_t* zlist; const int32_t* iperm; };
__device__ __forceinline__ void wf_apply_pos(const wf_pos& P, const wf_ctr* __restrict__ ctr)
{
    const int64_t pwords = P.pwords;
    unsigned long long* __restrict__ Hs = P.Hs; uint64_t* __restrict__ Hns = P.Hns; uint64_t* __restrict__ cands = P.cands;
    const int32_t* __restrict__ zlist = P.zlist; const int32_t* __restrict__ iperm = P.iperm;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t w = t; w < pwords; w += nt) {
        const uint64_t hn = Hns[w];
        if (hn) { atomicOr(&Hs[w], (unsigned long long)hn); Hns[w] = 0; }
        cands[w] = 0;
    }
    const int nzp = ctr->nz_prev;
    for (int64_t k = t; k < nzp; k += nt) {
        const int64_t p = iperm[zlist[k]];
        atomicAnd(&Hs[p >> 6], ~(1ull << (p & 63)));
    }
}

// One workgroup = four wavefronts = one 64-word slab of the masks; each wavefront takes WF_GW = 16 of its words.  One partial minimum per
// workgroup (k_wf_select reduces ~250 of them, every workgroup for itself) and ONE atomic per workgroup on a list counter: a counter
// takes ~90 atomics per microsecond, so a thousand wavefronts appending one by one cost more than the gathers they were split up for.
#define WF_BLK_WORDS (4 * WF_GW)
__global__ __launch_bounds__(256) void k_wf_apply_min(int64_t words, uint64_t* __restrict__ H, uint64_t* __restrict__ Z, uint64_t* __restrict__ Zp,
                                                     uint64_t* __restrict__ Hn, uint64_t* __restrict__ cand,
                                                     const uint64_t* __restrict__ W, const uint64_t* __restrict__ F, uint64_t* __restrict__ WF,
                                                     const double* __restrict__ C, double* __restrict__ part_c,
                                                     int64_t* __restrict__ part_i, wf_ctr* __restrict__ ctr, wf_pos P)
{
    __shared__ uint16_t s_list_[4][WF_GRP_CAP];
    __shared__ double s_c[4];
    __shared__ long long s_i[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t* const s_list = s_list_[wave];
    if (wf_stop(ctr)) {
        // the batch of the previous step held a goal node: the steps enqueued behind it are void.  k_wf_select cannot test
        // goal_cbits itself (see there), so the end is latched here, one kernel ahead of it on the stream (without
        // the latch every later select of the group re-appended the same batch to zlist and counted it into nz again)
        if (blockIdx.x == 0 && threadIdx.x == 0 && ctr->goal_cbits != ~0ull) ctr->ended = 1;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {              // (no other thread of this kernel touches these fields)
        ctr->tot[WF_NZ] += ctr->nz; ctr->tot[WF_NX] += ctr->nx;
        ctr->iters += 1; ctr->ntrip = 0; ctr->nz = 0; ctr->nx = 0;
    }
    if (P.pwords) wf_apply_pos(P, ctr);                       // (reads nz_prev and the batch list of the step before: k_wf_select rewrites both later)
    double bc = 0.0; int64_t bi = -1;
    for (int64_t b0 = (int64_t)blockIdx.x * WF_BLK_WORDS; b0 < words; b0 += (int64_t)gridDim.x * WF_BLK_WORDS) {     // uniform trip count
        const int64_t w0 = b0 + wave * WF_GW;
        const int64_t w = w0 + lane;
        uint64_t h = 0;
        if (w < words && lane < WF_GW) {
            const uint64_t z = Z[w];
            h = (H[w] & ~z) | Hn[w];                          // fmt.jl:83-84 for the batch of the previous step
            H[w] = h; Zp[w] = z; Z[w] = 0; Hn[w] = 0;
            if (!P.pwords) { cand[w] = 0; WF[w] = W[w] & (F ? F[w] : ~0ull); }     // unvisited and valid: the one word k_wf_mark gathers per entry
        }
        const int total = wf_expand_group(h, s_list);
        for (int k = lane; k < total; k += 64) {
            const int64_t i = w0 * 64 + (int64_t)s_list[k];
            const double c = C[i];
            if (bi < 0 || c < bc || (c == bc && i < bi)) { bc = c; bi = i; }
        }
        __builtin_amdgcn_wave_barrier();                      // (the list is rewritten by the next group)
    }
    wf_lexmin_wave(bc, bi);
    if (lane == 0) { s_c[wave] = bc; s_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            const double oc = s_c[k]; const int64_t oi = s_i[k];
            if (oi >= 0 && (bi < 0 || oc < bc || (oc == bc && oi < bi))) { bc = oc; bi = oi; }
        }
        part_c[blockIdx.x] = bc; part_i[blockIdx.x] = bi;
    }
}

// goal_cbits is written (atomicMin) only by blocks that have passed the entry test, which therefore reads `done` alone: a
// block starting late must still select its words.  done = 2 is decided identically by every block from the partials.
__global__ __launch_bounds__(256) void k_wf_select(int64_t words, int nparts, const uint64_t* __restrict__ H, uint64_t* __restrict__ Z,
                                                  const double* __restrict__ C, const double* __restrict__ X, int d,
                                                  const double* __restrict__ part_c, const int64_t* __restrict__ part_i,
                                                  double band, int single, wf_goal G, int32_t* __restrict__ zlist,
                                                  wf_ctr* __restrict__ ctr)
{
    __shared__ uint16_t s_list_[4][WF_GRP_CAP];
    __shared__ unsigned long long s_z_[4][WF_GW];
    __shared__ int s_tot[4], s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t* const s_list = s_list_[wave];
    unsigned long long* const s_z = s_z_[wave];
    if (ctr->done || ctr->ended) return;
    double cm = 0.0; int64_t im = -1;
    for (int p = lane; p < nparts; p += 64) {                 // (every wavefront reduces the ~250 partials for itself)
        const double c = part_c[p]; const int64_t i = part_i[p];
        if (i >= 0 && (im < 0 || c < cm || (c == cm && i < im))) { cm = c; im = i; }
    }
    wf_lexmin_wave(cm, im);
    if (im < 0) {                                   // H is empty: fmt.jl:85-89 `break`
        if (blockIdx.x == 0 && threadIdx.x == 0) ctr->done = 2;
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctr->cmin = cm; ctr->imin = im; }
    const double thr = cm + band;
    for (int64_t b0 = (int64_t)blockIdx.x * WF_BLK_WORDS; b0 < words; b0 += (int64_t)gridDim.x * WF_BLK_WORDS) {     // uniform trip count
        const int64_t w0 = b0 + wave * WF_GW;
        const int64_t w = w0 + lane;
        const bool own = w < words && lane < WF_GW;
        const uint64_t h = own ? H[w] : 0;
        if (lane < WF_GW) s_z[lane] = 0ull;
        const int total = wf_expand_group(h, s_list);
        for (int k = lane; k < total; k += 64) {
            const int p = (int)s_list[k];
            const int64_t i = w0 * 64 + p;
            const double c = C[i];
            const bool sel = single ? (i == im) : (c <= thr);
            if (!sel) continue;
            atomicOr(&s_z[p >> 6], 1ull << (p & 63));
            if (wf_is_goal(X + i * d, G))           // fmt.jl:68
                atomicMin(&ctr->goal_cbits, (unsigned long long)__double_as_longlong(c));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t z = own ? s_z[lane] : 0ull;
        if (z) Z[w] = z;
        // the batch list: the four wavefronts' counts are summed, ONE atomic on the list counter, every lane writes its own word's nodes
        const int n = __popcll(z);
        int inc = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
        if (lane == 63) s_tot[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tot = s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
            s_base = tot ? atomicAdd(&ctr->nz, tot) : 0;
        }
        __syncthreads();
        int o = s_base + inc - n;
        for (int k = 0; k < wave; ++k) o += s_tot[k];
        uint64_t m = z;
        while (m) { const int bb = __ffsll((long long)m) - 1; m &= m - 1; zlist[o++] = (int32_t)(w * 64 + bb); }
        __syncthreads();                                      // (s_tot / s_base are rewritten by the next slab)
    }
}

// candidate mask -> candidate list (balanced work for k_wf_connect: one wavefront per candidate, whatever word it sits in)
__global__ __launch_bounds__(64) void k_wf_compact(int64_t words, const unsigned long long* __restrict__ cand, int32_t* __restrict__ xlist,
                                                   wf_ctr* __restrict__ ctr)
{
    if (wf_stop(ctr)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->nz_prev = ctr->nz;      // (k_wf_select is done, the next k_wf_apply_min has not begun)
    for (int64_t w0 = (int64_t)blockIdx.x * 64; w0 < words; w0 += (int64_t)gridDim.x * 64) {
        const int64_t w = w0 + threadIdx.x;
        wf_append_word((w < words) ? cand[w] : 0ull, w, xlist, &ctr->nx);
    }
}

// ---- position space -----------------------------------------------------------------------------------------------------
// the graph's rows as cell-sorted positions (once per graph: the ordering pass writes them itself only for the unfused sweeps)
__global__ void k_wf_rowpos(const int32_t* __restrict__ rowval, const int32_t* __restrict__ iperm, int64_t nnz, int32_t* __restrict__ rowpos)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) rowpos[e] = iperm[rowval[e]];
}
// start of a solve: unvisited-and-valid by position (pads and the initial state out), the initial state open at cost 0
__global__ __launch_bounds__(256) void k_wf_init_pos(int64_t pwords, const int32_t* __restrict__ perm, const int32_t* __restrict__ iperm, int64_t init,
                                                     const uint64_t* __restrict__ F, uint64_t* __restrict__ WFs, uint64_t* __restrict__ Hs,
                                                     uint64_t* __restrict__ Hns, uint64_t* __restrict__ cands, double* __restrict__ Cs)
{
    const int lane = threadIdx.x & 63;
    const int64_t pinit = iperm[init];
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < pwords; w += (int64_t)gridDim.x * 4) {
        const int64_t p = w * 64 + lane;
        const int32_t o = perm[p];
        const bool ok = o >= 0 && o != init && (!F || wf_bit(F, o));
        const unsigned long long m = __ballot(ok);
        Cs[p] = 0.0;
        if (lane == 0) { WFs[w] = m; Hs[w] = (w == (pinit >> 6)) ? 1ull << (pinit & 63) : 0ull; Hns[w] = 0; cands[w] = 0; }
    }
}
// the node the reference's loop would end on: goal node of lowest (cost, index) in the batch Z, or -- open set exhausted --
// the last node of the previous batch Zp in pop order = highest (cost, index) (fmt.jl:85-89 leaves z at the last dequeued node)
__global__ __launch_bounds__(1024) void k_wf_final(int64_t words, const uint64_t* __restrict__ Z, const uint64_t* __restrict__ Zp,
                                                 const double* __restrict__ C, const double* __restrict__ X, int d, wf_goal G,
                                                 wf_ctr* __restrict__ ctr)
{
    const bool goal = ctr->goal_cbits != ~0ull;
    if (!goal && ctr->done != 2) return;
    const uint64_t* S = goal ? Z : Zp;
    __shared__ double s_c[16];
    __shared__ long long s_i[16];
    double bc = 0.0; int64_t bi = -1;
    for (int64_t w = threadIdx.x; w < words; w += blockDim.x) {
        uint64_t m = S[w];
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int64_t i = w * 64 + b;
            const double c = C[i];
            if (goal) {
                if (!wf_is_goal(X + i * d, G)) continue;
                if (bi < 0 || c < bc || (c == bc && i < bi)) { bc = c; bi = i; }
            } else {
                if (bi < 0 || c > bc || (c == bc && i > bi)) { bc = c; bi = i; }
            }
        }
    }
    if (goal) wf_lexmin_wave(bc, bi); else wf_lexmax_wave(bc, bi);
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = bc; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
            const double oc = s_c[k]; const int64_t oi = s_i[k];
            if (oi < 0) continue;
            const bool take = bi < 0 || (goal ? (oc < bc || (oc == bc && oi < bi)) : (oc > bc || (oc == bc && oi > bi)));
            if (take) { bc = oc; bi = oi; }
        }
        if (bi >= 0) ctr->final_z = bi;
        if (goal) ctr->done = 1;
    }
}

__global__ __launch_bounds__(256) void k_wf_sum_stats(const int64_t* __restrict__ stats, wf_ctr* __restrict__ ctr)
{
    __shared__ long long acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0;
    __syncthreads();
    long long v[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < WF_MAXBLK; b += blockDim.x)
        for (int k = 0; k < 4; ++k) v[k] += stats[b * 4 + k];
    for (int k = 0; k < 4; ++k) {
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
        if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&acc[k], (unsigned long long)v[k]);
    }
    __syncthreads();
    if (threadIdx.x == WF_NCONN || threadIdx.x == WF_CHECKS) ctr->tot[threadIdx.x] = acc[threadIdx.x];
}

// mark pass: one wavefront per batch node walks the node's column (symmetric metric: forward set == column,
// nearneighbors.jl:200-203) and sets the candidate bit of every unvisited valid row
__global__ __launch_bounds__(256) void k_wf_mark(const int32_t* __restrict__ zlist, const int64_t* __restrict__ colptr,
                                                 const int32_t* __restrict__ rowval, const uint64_t* __restrict__ WF,
                                                 unsigned long long* __restrict__ cand, const wf_ctr* __restrict__ ctr)
{
    if (wf_stop(ctr)) return;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int nz = ctr->nz;
    // Per entry ONE gathered word (WF = unvisited and valid) and the candidate word only for the entries that pass it.  The kernel is
    // bound by the latency of a node's chain (rows -> WF words -> candidate words -> atomic) at full residency: 54 000 nodes take 37 us =
    // 32 nodes in flight per CU x ~5.6 us each.  Both 64-entry chunks of a column (mean degree 107) are therefore requested together
    // (37 -> 33 us at the widest wavefront).  (Written with scalars: an earlier form with small arrays ran 2.7 x slower.)
    for (int iz = blockIdx.x * wpb + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); iz < nz; iz += gridDim.x * wpb) {
        const int64_t z = zlist[iz];
        const int64_t beg = colptr[z], end = colptr[z + 1];
        for (int64_t e0 = beg + lane; e0 < end + lane; e0 += 128) {         // (wave-uniform trip count)
            const int64_t e1 = e0 + 64;
            const int64_t xa = e0 < end ? (int64_t)rowval[e0] : -1;
            const int64_t xb = e1 < end ? (int64_t)rowval[e1] : -1;
            const unsigned long long wa = xa >= 0 ? WF[xa >> 6] : 0ull;
            const unsigned long long wb = xb >= 0 ? WF[xb >> 6] : 0ull;
            const unsigned long long ba = 1ull << (xa & 63), bb = 1ull << (xb & 63);
            const bool pa = (wa & ba) != 0, pb = (wb & bb) != 0;            // fmt.jl:70-71
            const unsigned long long ca = pa ? cand[xa >> 6] : ~0ull;       // seen already? (a stale read only costs an atomic)
            const unsigned long long cb = pb ? cand[xb >> 6] : ~0ull;
            if (pa && !(ca & ba)) atomicOr(&cand[xa >> 6], ba);
            if (pb && !(cb & bb)) atomicOr(&cand[xb >> 6], bb);
        }
    }
}

// sharded ctx: the columns of the batch nodes live on their owners, so each rank walks its OWN unvisited samples and
// looks for a batch node in the sample's own column (x in nearF(z) <=> z in col(x) for a metric)
__global__ __launch_bounds__(256) void k_wf_mark_owned(const int32_t* __restrict__ perm, int64_t p_begin, int64_t p_end,
                                                       const int64_t* __restrict__ colptr, const int32_t* __restrict__ rowval,
                                                       const uint64_t* __restrict__ W, const uint64_t* __restrict__ F,
                                                       const uint64_t* __restrict__ Z, unsigned long long* __restrict__ cand,
                                                       const wf_ctr* __restrict__ ctr)
{
    if (wf_stop(ctr)) return;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int64_t p = p_begin + blockIdx.x * wpb + (threadIdx.x >> 6); p < p_end; p += (int64_t)gridDim.x * wpb) {
        const int64_t x = perm[p];
        if (x < 0 || !wf_bit(W, x) || (F && !wf_bit(F, x))) continue;
        const int64_t beg = colptr[x], end = colptr[x + 1];
        bool hit = false;
        for (int64_t e0 = beg; e0 < end && !hit; e0 += 64) {
            const int64_t e = e0 + lane;
            const bool mine = (e < end) && wf_bit(Z, rowval[e]);
            hit = __ballot(mine) != 0;
        }
        if (hit && lane == 0) atomicOr(&cand[x >> 6], 1ull << (x & 63));
    }
}

// obstacle k of the transposed table [2*D][mpad]: lane = obstacle reads consecutive words
template <int D>
__device__ __forceinline__ box_regs<D> wf_load_box_T(const double* __restrict__ bT, int mpad, int k)
{
    box_regs<D> b;
#pragma unroll
    for (int i = 0; i < D; ++i) { b.lo[i] = bT[i * mpad + k]; b.hi[i] = bT[(D + i) * mpad + k]; }
    return b;
}

__global__ void k_wf_box_transpose(const double* __restrict__ boxes, int M, int d2, int mpad, double* __restrict__ bT)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= mpad * d2) return;
    const int j = t / mpad, k = t - j * mpad;
    bT[t] = boxes[(int64_t)min(k, M - 1) * d2 + j];               // padding repeats the last box
}

// one wavefront per candidate x.  MODE 0: connect in place; MODE 1: emit triples.  GEOM: the edge tests run against the obstacle set here
// (the lazy form); without it they are a bit of the resident mask and the kernel holds no geometry at all (four candidates in flight)
// POS: the candidate list holds cell-sorted positions (in ascending order: neighbouring candidates are neighbours in space), the open set and
// the costs are gathered by position (Hs, Cs, rows = rowsrc = the graph's rows as positions) and every connection is recorded on both sides
struct wf_posc { const int32_t* perm; const int32_t* rowpos; const unsigned long long* Hs; double* Cs; unsigned long long* WFs; unsigned long long* Hns; };
template <int D, int MODE, bool GEOM, bool POS>
__global__ __launch_bounds__(256, 4) void k_wf_connect(const int32_t* __restrict__ xlist, const int64_t* __restrict__ colptr,
                                                    const int32_t* __restrict__ rowval, const double* __restrict__ nzval,
                                                    const uint64_t* __restrict__ H, double* __restrict__ C, int32_t* __restrict__ A,
                                                    unsigned long long* __restrict__ W, unsigned long long* __restrict__ Hn,
                                                    const double* __restrict__ X, const double* __restrict__ bT, int M, int mpad,
                                                    mpfmt_ss ss, const uint64_t* __restrict__ gfree, const uint8_t* __restrict__ nseg,
                                                    wf_trip* __restrict__ mytrips, int64_t* __restrict__ stats, wf_ctr* __restrict__ ctr, int all_in, wf_posc P)
{
    // nseg != NULL (directed steering graphs: double integrator, cars): the edge's validity and the number of segment tests the
    // reference would have counted for it (boxesND.jl:26 per waypoint segment) were precomputed by the space's own sweep
    if (wf_stop(ctr)) return;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    int my_checks = 0, my_conn = 0;                                        // lane 0 counts for its wavefront
    const int nx = ctr->nx;
    // A candidate is a chain of dependent round trips (list entry -> column bounds -> rows -> open-set words -> costs -> [reduce] ->
    // mask bit / parent state -> stores) and a wavefront has a handful of candidates per step: one at a time the kernel sat out every
    // trip (15 us of wavefront time per candidate).  NB candidates are taken per pass and every stage is issued for all of them before
    // the next stage's first use: the trips of the NB chains overlap.  The winner's row rides through the reduce (no reload of rowval).
    constexpr int NB = GEOM ? 1 : 2;                                       // (one at a time 4.55 ms per solve, two 4.13, four 4.25: registers)
    // candidates per wavefront: round robin (caller labels), or -- POS -- one contiguous run of the position-ordered list each, so that a
    // wavefront's consecutive candidates are neighbours in space and their gathers meet the lines the last one pulled in
    const int nwv = gridDim.x * wpb, gwv = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int per = POS ? (nx + nwv - 1) / nwv : 0;
    const int ix_lo = POS ? gwv * per : gwv, ix_hi = POS ? min(nx, (gwv + 1) * per) : nx;
    const int st = POS ? 1 : nwv;
    const int32_t* __restrict__ rowsrc = POS ? P.rowpos : rowval;
    const unsigned long long* __restrict__ Hg = POS ? P.Hs : (const unsigned long long*)H;
    const double* __restrict__ Cg = POS ? P.Cs : C;
    for (int ix0 = ix_lo; ix0 < ix_hi; ix0 += NB * st) {
        int64_t x[NB], beg[NB], end[NB];
        [[maybe_unused]] int64_t px[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int64_t v = (ix0 + k * st < ix_hi) ? (int64_t)xlist[ix0 + k * st] : -1;
            if constexpr (POS) { px[k] = v; x[k] = v >= 0 ? (int64_t)P.perm[v] : -1; } else x[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) { beg[k] = x[k] >= 0 ? colptr[x[k]] : 0; end[k] = x[k] >= 0 ? colptr[x[k] + 1] : 0; }
        double best[NB];
        int64_t be[NB];
        int32_t by[NB];
        int64_t span = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) { best[k] = 0.0; be[k] = -1; by[k] = -1; span = max(span, end[k] - beg[k]); }
        for (int64_t off = lane; off < span + lane; off += 64) {           // nearB(V, x, r, H) + findmin, fmt.jl:72-74 (wave-uniform trips)
            int32_t y0[NB];
            double d0[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int64_t e0 = beg[k] + off;
                y0[k] = e0 < end[k] ? rowsrc[e0] : -1;
                d0[k] = e0 < end[k] ? nzval[e0] : 0.0;
            }
            unsigned long long h0[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) h0[k] = y0[k] >= 0 ? Hg[y0[k] >> 6] : 0ull;
            double c0[NB];
            bool o0[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                o0[k] = y0[k] >= 0 && ((h0[k] >> (y0[k] & 63)) & 1ull);
                c0[k] = o0[k] ? Cg[y0[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const double s0 = c0[k] + d0[k];
                if (o0[k] && (be[k] < 0 || s0 < best[k])) { best[k] = s0; be[k] = beg[k] + off; by[k] = y0[k]; }     // ascending e per lane keeps the first minimum
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) wf_lexmin_wave_y(best[k], be[k], by[k]);    // rows ascend with e: first minimum = lowest e
        // the edge tests of the NB winners: their mask words requested together (resident mask), or one lazy test after the other
        unsigned long long gw[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) gw[k] = (gfree && be[k] >= 0) ? gfree[be[k] >> 6] : 0ull;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (be[k] < 0) continue;                                           // (wave-uniform: be is the reduced value)
            const int64_t y = POS ? (int64_t)rowval[be[k]] : (int64_t)by[k];   // (POS: the reduce carried the row's position)
            bool inb = true;
            double v[D];
            [[maybe_unused]] double w[D];
            if (!all_in || GEOM) {
#pragma unroll
                for (int i = 0; i < D; ++i) v[i] = X[y * D + i];
                inb = in_state_space_sl<D>(v, ss);                             // statespaces.jl:155: first point of the segment
            }
            if (lane == 0) my_checks += nseg ? (int)nseg[be[k]] : (inb ? 1 : 0);   // boxesND.jl:26 is reached only then
            bool fr;
            if (!GEOM) {
                fr = (gw[k] >> (be[k] & 63)) & 1ull;
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) w[i] = X[x[k] * D + i];
                double l[D], h[D];
                seg_bbox<D>(v, w, l, h);
                bool blocked = false;
                for (int k0 = 0; k0 < M; k0 += 64) {                           // lane = obstacle (boxesND.jl:52-56; @all in any order)
                    const int kk = k0 + lane;
                    const box_regs<D> b = wf_load_box_T<D>(bT, mpad, kk);      // kk < mpad always: the table is padded to 64 lanes
                    const bool pend = (kk < M) && !broadphase_free_sl<D>(l, h, b);
                    if (__ballot(pend)) {
                        if (pend) blocked = blocked || !narrow_free_sl<D>(v, w, b);
                    }
                }
                fr = inb && (__ballot(blocked) == 0);
            }
            if (fr && lane == 0) {                                             // fmt.jl:76-80
                if (MODE == 0) {
                    A[x[k]] = (int32_t)y; C[x[k]] = best[k];
                    atomicAnd(&W[x[k] >> 6], ~(1ull << (x[k] & 63)));
                    atomicOr(&Hn[x[k] >> 6], 1ull << (x[k] & 63));
                    if constexpr (POS) {
                        P.Cs[px[k]] = best[k];
                        atomicAnd(&P.WFs[px[k] >> 6], ~(1ull << (px[k] & 63)));
                        atomicOr(&P.Hns[px[k] >> 6], 1ull << (px[k] & 63));
                    }
                    ++my_conn;
                } else {
                    wf_trip r; r.x = (int32_t)x[k]; r.y = (int32_t)y; r.c = best[k];
                    mytrips[atomicAdd(&ctr->ntrip, 1)] = r;                    // at most one per owned candidate: capacity N holds
                }
            }
        }
    }
    wf_block_stat(stats, WF_CHECKS, my_checks);
    wf_block_stat(stats, WF_NCONN, my_conn);
}

// exchange round `round` of this rank: header (x = total count of the step; 0 once the solve has ended) + its chunk
__global__ __launch_bounds__(256) void k_wf_pack(const wf_trip* __restrict__ mytrips, int round, wf_trip* __restrict__ slot,
                                                 const int64_t* __restrict__ stats, const wf_ctr* __restrict__ ctr)
{
    const int total = wf_stop(ctr) ? 0 : ctr->ntrip;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        long long checks = 0;
        for (int b = 0; b < WF_MAXBLK; ++b) checks += stats[b * 4 + WF_CHECKS];
        wf_trip hd; hd.x = total; hd.y = round; hd.c = (double)checks; slot[0] = hd;            // c: this rank's edge checks so far
    }
    const int64_t src = (int64_t)round * WF_XCAP + t;
    if (t < WF_XCAP && src < total) slot[1 + t] = mytrips[src];
}

// apply round `round` of `nslots` exchange slots
__global__ __launch_bounds__(256) void k_wf_commit(const wf_trip* __restrict__ xbuf, int nslots, int round,
                                                   double* __restrict__ C, int32_t* __restrict__ A, unsigned long long* __restrict__ W,
                                                   unsigned long long* __restrict__ Hn, int64_t* __restrict__ stats, wf_ctr* __restrict__ ctr)
{
    if (wf_stop(ctr)) return;
    for (int s = 0; s < nslots; ++s) {
        const wf_trip* slot = xbuf + (int64_t)s * (WF_XCAP + 1);
        const int64_t left = (int64_t)slot[0].x - (int64_t)round * WF_XCAP;
        const int64_t n = left < 0 ? 0 : (left > WF_XCAP ? WF_XCAP : left);
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
            const wf_trip r = slot[1 + t];
            const int64_t x = r.x;
            A[x] = r.y; C[x] = r.c;
            atomicAnd(&W[x >> 6], ~(1ull << (x & 63)));
            atomicOr(&Hn[x >> 6], 1ull << (x & 63));
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && n) stats[WF_NCONN] += n;
    }
}

// back-trace fmt.jl:92-101 from final_z to the root, written root first; path[N] = length
__global__ void k_wf_path(const int32_t* __restrict__ A, int64_t N, const wf_ctr* __restrict__ ctr, int64_t* __restrict__ path)
{
    int64_t cur = ctr->final_z, n = 1;
    while (n < N && A[cur] >= 0) { cur = A[cur]; ++n; }
    path[N] = n;
    cur = ctr->final_z;
    for (int64_t k = n - 1; k >= 0; --k) { path[k] = cur + 1; if (k) cur = A[cur]; }
}

__global__ void k_wf_A_to_i64(const int32_t* __restrict__ A, int64_t N, int64_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = (int64_t)A[i] + 1;
}

// ---- host side -----------------------------------------------------------------------------------------------------

static mpfmt_wf* wf_of(mpfmt_ctx* ctx) { return (mpfmt_wf*)ctx->wf; }

void mpfmt_wf_free(mpfmt_ctx* ctx)
{
    mpfmt_wf* s = wf_of(ctx);
    if (!s) return;
    void* bufs[] = {s->W, s->H, s->Z, s->Zp, s->Hn, s->cand, s->F, s->WF, s->WFs, s->Hs, s->Hns, s->cands, s->Cs, s->C, s->A, s->zlist, s->xlist, s->rowptr, s->colidx, s->part_c, s->part_i, s->stats, s->boxT, s->mytrips, s->xbuf,
                    s->ctr, s->path_dev};
    for (void* b : bufs) if (b) hipFree(b);
    if (s->ctr_host) hipHostFree(s->ctr_host);
    if (s->hdr_host) hipHostFree(s->hdr_host);
    delete s;
    ctx->wf = nullptr;
}

static int32_t wf_alloc(mpfmt_ctx* ctx, mpfmt_wf* s, int64_t N, int world)
{
    const int64_t words = (N + 63) / 64;
    if (s->N != N) {
        void** bufs[] = {(void**)&s->W, (void**)&s->H, (void**)&s->Z, (void**)&s->Zp, (void**)&s->Hn, (void**)&s->cand, (void**)&s->F, (void**)&s->WF,
                         (void**)&s->WFs, (void**)&s->Hs, (void**)&s->Hns, (void**)&s->cands, (void**)&s->Cs,
                         (void**)&s->C, (void**)&s->A, (void**)&s->zlist, (void**)&s->xlist, (void**)&s->path_dev, (void**)&s->mytrips};
        for (void** b : bufs) if (*b) { HIPCHK(ctx, hipFree(*b)); *b = nullptr; }
        HIPCHK(ctx, hipMalloc((void**)&s->W, 8 * words)); HIPCHK(ctx, hipMalloc((void**)&s->H, 8 * words));
        HIPCHK(ctx, hipMalloc((void**)&s->Z, 8 * words)); HIPCHK(ctx, hipMalloc((void**)&s->Zp, 8 * words));
        HIPCHK(ctx, hipMalloc((void**)&s->Hn, 8 * words));
        HIPCHK(ctx, hipMalloc((void**)&s->cand, 8 * words)); HIPCHK(ctx, hipMalloc((void**)&s->F, 8 * words)); HIPCHK(ctx, hipMalloc((void**)&s->WF, 8 * words));
        HIPCHK(ctx, hipMalloc((void**)&s->C, 8 * N)); HIPCHK(ctx, hipMalloc((void**)&s->A, 4 * N));
        HIPCHK(ctx, hipMalloc((void**)&s->zlist, 4 * N)); HIPCHK(ctx, hipMalloc((void**)&s->xlist, 4 * N));
        HIPCHK(ctx, hipMalloc((void**)&s->path_dev, 8 * (N + 1)));
        HIPCHK(ctx, hipMalloc((void**)&s->WFs, 8 * words)); HIPCHK(ctx, hipMalloc((void**)&s->Hs, 8 * words));      // (ceil(N / 64) words = one per tile)
        HIPCHK(ctx, hipMalloc((void**)&s->Hns, 8 * words)); HIPCHK(ctx, hipMalloc((void**)&s->cands, 8 * words));
        HIPCHK(ctx, hipMalloc((void**)&s->Cs, 8 * 64 * words));
        s->N = N; s->words = words;
    }
    if (!s->part_c) { HIPCHK(ctx, hipMalloc((void**)&s->part_c, 8 * WF_MAXPARTS)); HIPCHK(ctx, hipMalloc((void**)&s->part_i, 8 * WF_MAXPARTS)); }
    if (!s->stats) HIPCHK(ctx, hipMalloc((void**)&s->stats, 8 * 4 * WF_MAXBLK));
    if (!s->ctr) { HIPCHK(ctx, hipMalloc((void**)&s->ctr, sizeof(wf_ctr))); HIPCHK(ctx, hipHostMalloc((void**)&s->ctr_host, sizeof(wf_ctr))); }
    if (world > 1) {
        if (!s->mytrips) HIPCHK(ctx, hipMalloc((void**)&s->mytrips, sizeof(wf_trip) * (size_t)N));
        if (s->world_alloc < world) {
            if (s->xbuf) { HIPCHK(ctx, hipFree(s->xbuf)); s->xbuf = nullptr; }
            if (s->hdr_host) { HIPCHK(ctx, hipHostFree(s->hdr_host)); s->hdr_host = nullptr; }
            HIPCHK(ctx, hipMalloc((void**)&s->xbuf, sizeof(wf_trip) * (size_t)(WF_XCAP + 1) * (size_t)world));
            HIPCHK(ctx, hipHostMalloc((void**)&s->hdr_host, sizeof(wf_trip) * (size_t)world));
            s->world_alloc = world;
        }
    }
    return MPFMT_OK;
}

// the per-rank part of one step: apply + select + mark + connect (sharded: connections are left in mytrips)
static int32_t wf_enqueue_local(mpfmt_ctx* ctx, mpfmt_wf* s)
{
    const int64_t words = s->words;
    const int nparts = s->nparts;
    hipStream_t st = ctx->stream;
    const int d = ctx->d;
    const bool pos = s->pos_space != 0;
    wf_pos PA{pos ? s->pwords : 0, (unsigned long long*)s->Hs, s->Hns, s->cands, s->zlist, ctx->iperm};
    hipLaunchKernelGGL(k_wf_apply_min, dim3(nparts), dim3(256), 0, st, words, s->H, s->Z, s->Zp, s->Hn, s->cand, s->W, s->checkpts ? s->F : nullptr, s->WF, s->C,
                       s->part_c, s->part_i, s->ctr, PA);
    hipLaunchKernelGGL(k_wf_select, dim3(nparts), dim3(256), 0, st, words, nparts, s->H, s->Z, s->C, ctx->Xo, d, s->part_c, s->part_i, s->band,
                       s->single, s->goal, s->zlist, s->ctr);
    const uint64_t* F = s->checkpts ? s->F : nullptr;
    const int grid = ctx->num_cus * 8;                       // persistent: 4 wavefronts per block, one list entry per wavefront at a time
    if (!s->sharded) {
        // forward sets: the column itself for a metric (nearneighbors.jl:200-203), the row of the cost matrix otherwise.  Position space:
        // the same kernel over the rows as positions, the position-indexed unvisited-and-valid and candidate masks
        if (pos) hipLaunchKernelGGL(k_wf_mark, dim3(grid), dim3(256), 0, st, s->zlist, ctx->colptr, (const int32_t*)ctx->rowpos, s->WFs,
                                    (unsigned long long*)s->cands, s->ctr);
        else hipLaunchKernelGGL(k_wf_mark, dim3(grid), dim3(256), 0, st, s->zlist, s->directed ? s->rowptr : ctx->colptr,
                                s->directed ? s->colidx : ctx->rowval, s->WF, (unsigned long long*)s->cand, s->ctr);
    } else {
        const int64_t pb = std::min<int64_t>(ctx->tile_begin * 64, ctx->N), pe = std::min<int64_t>(ctx->tile_end * 64, ctx->N);
        hipLaunchKernelGGL(k_wf_mark_owned, dim3(ctx->num_cus * 8), dim3(256), 0, st, ctx->perm, pb, pe, ctx->colptr, ctx->rowval, s->W, F, s->Z,
                           (unsigned long long*)s->cand, s->ctr);
    }
    if (pos) hipLaunchKernelGGL(k_wf_compact, dim3(nparts), dim3(64), 0, st, s->pwords, (const unsigned long long*)s->cands, s->xlist, s->ctr);
    else hipLaunchKernelGGL(k_wf_compact, dim3(nparts), dim3(64), 0, st, words, (const unsigned long long*)s->cand, s->xlist, s->ctr);
    const uint64_t* gfree = s->use_mask ? ctx->graph_free : nullptr;
    const uint8_t* nseg = s->directed ? ctx->di_nseg : nullptr;
    // (a directed steering graph's validity bits are its own sweep's: gfree is set there too -- the mask form)
    const bool geom = gfree == nullptr;
    // (persistent workgroups: exactly as many as are resident at once -- the kernel's loops stride by the grid, so a workgroup that has to
    // wait for a slot starts its share of the candidates when the others are done with theirs.  The runtime's occupancy query counts
    // registers in eights; the device clock at the entry of every wavefront says sixteens: the mask form's 71 registers fit six to a
    // SIMD, not the seven the query answers -- one workgroup per CU started 40 us late and the widest launch took 80 us instead of 60)
    wf_posc PC{ctx->perm, ctx->rowpos, (const unsigned long long*)s->Hs, s->Cs, (unsigned long long*)s->WFs, (unsigned long long*)s->Hns};
#define WF_CONNECT(MODE_, GEOM_, POS_, TRIPS_) DISPATCH_D(d, { int per_cu = 0; \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_wf_connect<DD, MODE_, GEOM_, POS_>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4; \
        hipFuncAttributes fa_; \
        if (hipFuncGetAttributes(&fa_, (const void*)k_wf_connect<DD, MODE_, GEOM_, POS_>) == hipSuccess && fa_.numRegs > 0) \
            per_cu = std::max(1, std::min(per_cu, 512 / (((fa_.numRegs + 15) / 16) * 16))); \
        const int cgrid = ctx->num_cus * std::min(per_cu, 8); \
        hipLaunchKernelGGL((k_wf_connect<DD, MODE_, GEOM_, POS_>), dim3(cgrid), dim3(256), 0, st, s->xlist, ctx->colptr, \
        ctx->rowval, ctx->nzval, s->H, s->C, s->A, (unsigned long long*)s->W, (unsigned long long*)s->Hn, ctx->Xo, s->boxT, ctx->M, s->mpad, ctx->ss, gfree, nseg, \
        TRIPS_, s->stats, s->ctr, s->all_in, PC); })
    if (!s->sharded) {
        if (pos) { if (geom) { WF_CONNECT(0, true, true, (wf_trip*)nullptr); } else { WF_CONNECT(0, false, true, (wf_trip*)nullptr); } }
        else { if (geom) { WF_CONNECT(0, true, false, (wf_trip*)nullptr); } else { WF_CONNECT(0, false, false, (wf_trip*)nullptr); } }
    } else { if (geom) { WF_CONNECT(1, true, false, s->mytrips); } else { WF_CONNECT(1, false, false, s->mytrips); } }
#undef WF_CONNECT
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// ONE all-gather per wavefront: every rank's (x, y_min, c_min) connections, the counts riding in the slot
// headers; a rank that connected more than WF_XCAP nodes makes every rank run further rounds -- all of them read the same
// headers, so the decision is the same everywhere and the collectives stay matched.  Costs one host look per wavefront.
static int32_t wf_exchange(mpfmt_ctx* ctx, mpfmt_wf* s)
{
    int rank = 0, world = 1;
    mpfmt_comm_world(ctx, &rank, &world);
    hipStream_t st = ctx->stream;
    const size_t slot_bytes = sizeof(wf_trip) * (size_t)(WF_XCAP + 1);
    int32_t rc;
    for (int round = 0;; ++round) {
        wf_trip* myslot = s->xbuf + (size_t)rank * (WF_XCAP + 1);
        hipLaunchKernelGGL(k_wf_pack, dim3((WF_XCAP + 255) / 256), dim3(256), 0, st, s->mytrips, round, myslot, s->stats, s->ctr);
        if ((rc = mpfmt_comm_allgather_inplace(ctx, s->xbuf, slot_bytes, st))) return rc;
        hipLaunchKernelGGL(k_wf_commit, dim3(64), dim3(256), 0, st, s->xbuf, world, round, s->C, s->A, (unsigned long long*)s->W,
                           (unsigned long long*)s->Hn, s->stats, s->ctr);
        HIPCHK(ctx, hipMemcpy2DAsync(s->hdr_host, sizeof(wf_trip), s->xbuf, slot_bytes, sizeof(wf_trip), (size_t)world, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        int64_t mx = 0;
        for (int g = 0; g < world; ++g) mx = std::max<int64_t>(mx, s->hdr_host[g].x);
        if (mx <= (int64_t)(round + 1) * WF_XCAP) break;
    }
    return MPFMT_OK;
}

static int32_t wf_read_ctr(mpfmt_ctx* ctx, mpfmt_wf* s, bool with_stats)
{
    if (with_stats) hipLaunchKernelGGL(k_wf_sum_stats, dim3(1), dim3(256), 0, ctx->stream, s->stats, s->ctr);
    HIPCHK(ctx, hipMemcpyAsync(s->ctr_host, s->ctr, sizeof(wf_ctr), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return MPFMT_OK;
}

static bool wf_ended(const mpfmt_wf* s) { return s->ctr_host->done != 0 || s->ctr_host->goal_cbits != ~0ull; }

// info of the step(s) since the previous call (ctr_host must hold summed statistics)
static void wf_fill_info(mpfmt_wf* s, mpfmt_wf_info* info)
{
    const wf_ctr& c = *s->ctr_host;
    const bool goal = c.goal_cbits != ~0ull;
    info->done = goal ? 1 : c.done;
    info->nz = c.done == 2 ? 0 : c.nz;
    info->nx = c.nx;
    info->nconn = (int32_t)(c.tot[WF_NCONN] - s->prev_tot[WF_NCONN]);
    info->ntrip = c.ntrip;
    info->iters = c.iters - (c.done == 2 ? 1 : 0);       // the step that found the open set empty expanded no batch
    info->checks = c.tot[WF_CHECKS]; info->cmin = c.cmin;
    info->tot_z = c.tot[WF_NZ] + info->nz; info->tot_x = c.tot[WF_NX] + c.nx; info->tot_conn = c.tot[WF_NCONN];     // (the device adds a step's nz / nx at the start of the next)
    for (int k = 0; k < 4; ++k) s->prev_tot[k] = c.tot[k];
}

extern "C" {

int32_t mpfmt_wf_begin(mpfmt_ctx* ctx, double r, int64_t init_idx, int32_t checkpts, int32_t goal_kind, const double* goal_params,
                       double band, int32_t flags)
{
    if (!ctx) return MPFMT_ERR_ARG;
    if (!goal_params) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "goal_params is NULL");
    if (!ctx->Xo) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no samples uploaded");
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    const int64_t N = ctx->N;
    const int d = ctx->d;
    if (N < 1) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "no samples");
    if (init_idx < 1 || init_idx > N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "init_idx out of range");
    if (goal_kind < 0 || goal_kind > 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unknown goal kind %d", goal_kind);
    if (!(r > 0.0) || !std::isfinite(r)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "radius must be finite and > 0");
    if (!(band >= 0.0) || !std::isfinite(band)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "band must be finite and >= 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if (!ctx->wf) ctx->wf = new mpfmt_wf();
    mpfmt_wf* s = wf_of(ctx);
    s->active = false;
    s->t_begin = std::chrono::steady_clock::now();
    int rank = 0, world = 1;
    mpfmt_comm_world(ctx, &rank, &world);
    s->sharded = world > 1 || ctx->wf_force_sharded;
    if ((rc = wf_alloc(ctx, s, N, s->sharded ? std::max(world, 2) : 1))) return rc;
    s->band = band; s->single = (flags & MPFMT_WF_SINGLE) ? 1 : 0; s->checkpts = checkpts ? 1 : 0;
    s->init = init_idx - 1; s->r = r; s->directed = false;
    s->goal.kind = goal_kind; s->goal.gd = d;
    const int ng = goal_kind == MPFMT_GOAL_RECT ? 2 * d : goal_kind == MPFMT_GOAL_BALL ? d + 1 : d;
    memset(s->goal.g, 0, sizeof s->goal.g);
    for (int i = 0; i < ng; ++i) s->goal.g[i] = goal_params[i];
    s->nparts = (int)std::min<int64_t>(WF_MAXPARTS, (s->words + WF_BLK_WORDS - 1) / WF_BLK_WORDS);
    if (s->nparts < 1) s->nparts = 1;
    for (int k = 0; k < 4; ++k) s->prev_tot[k] = 0;

    // checkpts bitmap F (fmt.jl:31-36); also answers is_free_state(init) (fmt.jl:24-29)
    auto t0 = std::chrono::steady_clock::now();
    if ((rc = mpfmt_launch_points_free(ctx, nullptr, N, s->F))) return rc;
    uint64_t fw = 0;
    HIPCHK(ctx, hipMemcpyAsync(&fw, s->F + (s->init >> 6), 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!((fw >> (s->init & 63)) & 1ull)) return mpfmt_fail(ctx, MPFMT_ERR_INFEASIBLE, "initial state is infeasible");
    auto t1 = std::chrono::steady_clock::now();
    // r-disc graph (reused when one of this radius is resident); the edge tests stay lazy unless the checker has no
    // lane-per-obstacle form here (2-D SAT world, non-identity workspace) or the caller asks for the eager mask
    const bool lazy_ok = ctx->cc_kind == 0 && ctx->dw == d;
    if (!(ctx->graph_filled && ctx->graph_r == r)) {
        // no graph of this radius is resident: the STEP builds it with every edge's bit (edge tests fused into the build: ~0.4 ms more
        // than the graph alone at the north star), and the recursion then reads bits instead of testing ~1e6 edges one wavefront at a
        // time (2.4 ms); MPFMT_WF_LAZY keeps the graph-only build and the lazy tests
        if (lazy_ok && !(flags & MPFMT_WF_LAZY)) rc = mpfmt_graph_step_device(ctx, r, nullptr);
        else rc = mpfmt_graph_build_device(ctx, r, nullptr);
        if (rc) return rc;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto t2 = std::chrono::steady_clock::now();
    // (a mask a step left for this graph and this obstacle set is used, not recomputed: upload_boxes / set_state_bounds void graph_swept)
    const bool resident = ctx->graph_swept && ctx->graph_filled && ctx->graph_r == r && ctx->graph_free && !(flags & MPFMT_WF_LAZY);
    s->use_mask = ((flags & MPFMT_WF_EAGER) || !lazy_ok || resident) ? 1 : 0;
    if ((rc = mpfmt_sweep_prepare_ss(ctx))) return rc;
    s->all_in = (!ctx->ss.has || ctx->ssflag_all_in) ? 1 : 0;
    if (s->use_mask && !ctx->graph_swept) {
        ctx->pend_valid = false; if ((rc = mpfmt_launch_graph_sweep(ctx))) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (!s->use_mask) {                                    // obstacle table transposed for lane = obstacle reads
        const int mpad = std::max(64, ((ctx->M + 63) / 64) * 64);
        if (s->boxT_cap < 2 * d * mpad) {
            if (s->boxT) { HIPCHK(ctx, hipFree(s->boxT)); s->boxT = nullptr; }
            HIPCHK(ctx, hipMalloc((void**)&s->boxT, sizeof(double) * 2 * (size_t)d * (size_t)mpad));
            s->boxT_cap = 2 * d * mpad;
        }
        s->mpad = mpad;
        if (ctx->M > 0)
            hipLaunchKernelGGL(k_wf_box_transpose, dim3((unsigned)((mpad * 2 * d + 255) / 256)), dim3(256), 0, ctx->stream, ctx->boxes, ctx->M, 2 * d, mpad, s->boxT);
        else
            HIPCHK(ctx, hipMemsetAsync(s->boxT, 0, sizeof(double) * 2 * (size_t)d * (size_t)mpad, ctx->stream));
    }
    auto t3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    s->ms_graph = ms(t1, t2); s->ms_sweep = ms(t0, t1) + ms(t2, t3);
    hipLaunchKernelGGL(k_wf_init, dim3(256), dim3(64), 0, ctx->stream, N, s->words, s->init, s->W, s->H, s->Z, s->Zp, s->Hn, s->cand, s->C, s->A,
                       s->stats, s->ctr);
    // position space (see mpfmt_wf): an unsharded ctx whose WHOLE index belongs to the resident graph (built here, not imported)
    s->pos_space = (!s->sharded && ctx->wf_pos_space && ctx->grid_r == r && ctx->perm && ctx->iperm && !ctx->tileneed && ctx->ntiles == s->words &&
                    ctx->nnz > 0 && ctx->nnz < ((int64_t)1 << 31)) ? 1 : 0;
    if (s->pos_space && !ctx->rowpos_valid) {
        // the entries' positions cost one gather over the whole graph (0.7 ms at the north star) and save a quarter of that per solve:
        // the FIRST solve on a graph runs by caller index, the positions are made when a second one asks for the same graph
        const bool again = ctx->wf_seen_epoch == ctx->samples_epoch && ctx->wf_seen_r == r && ctx->wf_seen_nnz == ctx->nnz;
        ctx->wf_seen_epoch = ctx->samples_epoch; ctx->wf_seen_r = r; ctx->wf_seen_nnz = ctx->nnz;
        if (!again && ctx->wf_pos_space < 2) s->pos_space = 0;
    }
    ctx->wf_pos_used = s->pos_space;
    if (s->pos_space) {
        s->pwords = s->words;
        if (!ctx->rowpos_valid) {
            if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rowpos, sizeof(int32_t) * (size_t)std::max<int64_t>(std::max(ctx->nnz, ctx->nnz_cap), 1)))) return rc;
            hipLaunchKernelGGL(k_wf_rowpos, dim3((unsigned)((ctx->nnz + 255) / 256)), dim3(256), 0, ctx->stream, (const int32_t*)ctx->rowval, (const int32_t*)ctx->iperm,
                               ctx->nnz, ctx->rowpos);
            ctx->rowpos_valid = true;
        }
        hipLaunchKernelGGL(k_wf_init_pos, dim3(ctx->num_cus * 4), dim3(256), 0, ctx->stream, s->pwords, (const int32_t*)ctx->perm, (const int32_t*)ctx->iperm, s->init,
                           (const uint64_t*)(s->checkpts ? s->F : nullptr), s->WFs, s->Hs, s->Hns, s->cands, s->Cs);
    }
    HIPCHK(ctx, hipGetLastError());
    s->active = true;
    return MPFMT_OK;
}

int32_t mpfmt_wf_step(mpfmt_ctx* ctx, mpfmt_wf_info* info)
{
    if (!ctx) return MPFMT_ERR_ARG;
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no wavefront solve in progress (mpfmt_wf_begin)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = wf_enqueue_local(ctx, s))) return rc;
    if (s->sharded && ctx->comm && (rc = wf_exchange(ctx, s))) return rc;
    if ((rc = wf_read_ctr(ctx, s, true))) return rc;
    if (info) wf_fill_info(s, info);
    return MPFMT_OK;
}

int32_t mpfmt_wf_state(mpfmt_ctx* ctx, uint64_t* W, uint64_t* H, double* C, int64_t* A)
{
    if (!ctx) return MPFMT_ERR_ARG;
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no wavefront solve in progress (mpfmt_wf_begin)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t N = s->N, words = s->words;
    if (W) HIPCHK(ctx, hipMemcpy(W, s->W, 8 * words, hipMemcpyDeviceToHost));
    if (H) {   // the open set as the NEXT step will see it: H = (H \ Z) + Hnew is applied at the start of a step
        std::vector<uint64_t> h(words), z(words), hn(words);
        HIPCHK(ctx, hipMemcpy(h.data(), s->H, 8 * words, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(z.data(), s->Z, 8 * words, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(hn.data(), s->Hn, 8 * words, hipMemcpyDeviceToHost));
        for (int64_t w = 0; w < words; ++w) H[w] = (h[w] & ~z[w]) | hn[w];
    }
    if (C) HIPCHK(ctx, hipMemcpy(C, s->C, 8 * N, hipMemcpyDeviceToHost));
    if (A) {
        void* scr;
        int32_t rc;
        if ((rc = mpfmt_scratch(ctx, 8 * (size_t)N, &scr))) return rc;
        hipLaunchKernelGGL(k_wf_A_to_i64, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, s->A, N, (int64_t*)scr);
        HIPCHK(ctx, hipMemcpyAsync(A, scr, 8 * N, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

int32_t mpfmt_wf_batch(mpfmt_ctx* ctx, int64_t* zs, int64_t cap, int64_t* nz)
{
    if (!ctx || !nz) return MPFMT_ERR_ARG;
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no wavefront solve in progress (mpfmt_wf_begin)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> z((size_t)s->words);
    HIPCHK(ctx, hipMemcpy(z.data(), s->Z, 8 * s->words, hipMemcpyDeviceToHost));        // the batch mask of the last step, ascending
    int64_t n = 0;
    for (int64_t w = 0; w < s->words; ++w) n += __builtin_popcountll(z[w]);
    *nz = n;
    if (n > cap) return mpfmt_fail(ctx, MPFMT_ERR_CAPACITY, "batch of %lld exceeds capacity %lld", (long long)n, (long long)cap);
    if (n > 0 && !zs) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "zs is NULL");
    int64_t o = 0;
    for (int64_t w = 0; w < s->words; ++w) {
        uint64_t m = z[w];
        while (m) { const int b = __builtin_ctzll(m); m &= m - 1; zs[o++] = w * 64 + b + 1; }
    }
    return MPFMT_OK;
}

// manual exchange for a sharded ctx without a communicator (one host thread driving G ctxs; the tests): read this rank's
// connections of the step just made, then hand every rank's connections to every ctx with mpfmt_wf_commit
int32_t mpfmt_wf_triples(mpfmt_ctx* ctx, int64_t cap, int64_t* x, int64_t* y, double* c, int64_t* n)
{
    if (!ctx || !n) return MPFMT_ERR_ARG;
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active || !s->sharded) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no sharded wavefront solve in progress");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t rc;
    if ((rc = wf_read_ctr(ctx, s, false))) return rc;
    const int64_t cnt = wf_ended(s) ? 0 : s->ctr_host->ntrip;
    *n = cnt;
    if (cnt > cap) return mpfmt_fail(ctx, MPFMT_ERR_CAPACITY, "%lld triples exceed capacity %lld", (long long)cnt, (long long)cap);
    if (cnt > 0) {
        if (!x || !y || !c) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "x / y / c is NULL");
        std::vector<wf_trip> t((size_t)cnt);
        HIPCHK(ctx, hipMemcpy(t.data(), s->mytrips, sizeof(wf_trip) * (size_t)cnt, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < cnt; ++k) { x[k] = (int64_t)t[k].x + 1; y[k] = (int64_t)t[k].y + 1; c[k] = t[k].c; }
    }
    return MPFMT_OK;
}

int32_t mpfmt_wf_commit(mpfmt_ctx* ctx, int64_t n, const int64_t* x, const int64_t* y, const double* c)
{
    if (!ctx) return MPFMT_ERR_ARG;
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active || !s->sharded) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no sharded wavefront solve in progress");
    if (n < 0 || n > s->N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "n out of range");
    if (n == 0) return MPFMT_OK;
    if (!x || !y || !c) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "x / y / c is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<wf_trip> t((size_t)WF_XCAP + 1);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t o = 0; o < n; o += WF_XCAP) {
        const int64_t m = std::min<int64_t>(WF_XCAP, n - o);
        t[0].x = (int32_t)m; t[0].y = 0; t[0].c = 0.0;
        for (int64_t k = 0; k < m; ++k) {
            if (x[o + k] < 1 || x[o + k] > s->N || y[o + k] < 1 || y[o + k] > s->N) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "triple %lld out of range", (long long)(o + k));
            t[k + 1].x = (int32_t)(x[o + k] - 1); t[k + 1].y = (int32_t)(y[o + k] - 1); t[k + 1].c = c[o + k];
        }
        HIPCHK(ctx, hipMemcpy(s->xbuf, t.data(), sizeof(wf_trip) * (size_t)(m + 1), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_wf_commit, dim3(64), dim3(256), 0, ctx->stream, s->xbuf, 1, 0, s->C, s->A, (unsigned long long*)s->W,
                           (unsigned long long*)s->Hn, s->stats, s->ctr);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MPFMT_OK;
}

int32_t mpfmt_wf_finish(mpfmt_ctx* ctx, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{
    if (!ctx || !res) return MPFMT_ERR_ARG;
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no wavefront solve in progress (mpfmt_wf_begin)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t N = s->N;
    int32_t rc;
    hipLaunchKernelGGL(k_wf_final, dim3(1), dim3(1024), 0, ctx->stream, s->words, s->Z, s->Zp, s->C, ctx->Xo, ctx->d, s->goal, s->ctr);
    hipLaunchKernelGGL(k_wf_path, dim3(1), dim3(1), 0, ctx->stream, s->A, N, s->ctr, s->path_dev);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = wf_read_ctr(ctx, s, true))) return rc;
    const wf_ctr& c = *s->ctr_host;
    int64_t plen = 0;
    HIPCHK(ctx, hipMemcpy(&plen, s->path_dev + N, 8, hipMemcpyDeviceToHost));
    if (path) HIPCHK(ctx, hipMemcpy(path, s->path_dev, 8 * plen, hipMemcpyDeviceToHost));
    double cz = 0.0;
    HIPCHK(ctx, hipMemcpy(&cz, s->C + c.final_z, 8, hipMemcpyDeviceToHost));
    if (A || C) { if ((rc = mpfmt_wf_state(ctx, nullptr, nullptr, C, A))) return rc; }
    memset(res, 0, sizeof *res);
    res->status = c.done == 1 ? 1 : 0;
    res->cost = cz;
    res->z = c.final_z + 1;
    res->collision_checks = c.tot[WF_CHECKS];
    if (s->sharded && ctx->comm) {          // every rank counted the checks of its own samples; the headers of the last exchange hold all counts
        int rank = 0, world = 1;
        mpfmt_comm_world(ctx, &rank, &world);
        int64_t tot = 0;
        for (int g = 0; g < world; ++g) tot += (int64_t)s->hdr_host[g].c;
        res->collision_checks = tot;
    }
    res->path_len = plen;
    res->nnz = ctx->nnz;
    res->ms_graph = s->ms_graph; res->ms_sweep = s->ms_sweep;
    res->ms_host_loop = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s->t_begin).count() - s->ms_graph - s->ms_sweep;
    return MPFMT_OK;
}

// steps until the end condition: enqueued in groups, the end condition voids the kernels issued after it, the host looks
// once per group (sharded: the exchange looks at the slot headers every step anyway)
extern "C++" int32_t mpfmt_wf_run(mpfmt_ctx* ctx)
{
    mpfmt_wf* s = wf_of(ctx);
    if (!s || !s->active) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no wavefront solve in progress (mpfmt_wf_begin)");
    if (s->sharded && !ctx->comm) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "a sharded wavefront solve needs a communicator (mpfmt_comm_create), or the manual mpfmt_wf_step / _triples / _commit loop");
    int32_t rc;
    const int group = s->sharded ? 1 : (s->single ? 32 : 8);
    const int64_t max_steps = 4 * s->N + 64;
    if (!s->sharded && ctx->wf_graphs) {
        // The kernels of a step take no host-side argument that changes from step to step (the sets, the lists and the counters live
        // on the device), so a group of steps is captured ONCE into a hipGraph and replayed: one launch per 8 wavefronts instead of
        // 48 (a wavefront's kernels are a few microseconds each -- the solve was launch-bound).  Anything that goes wrong with the
        // capture falls through to the plain loop below.
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        bool ok = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            for (int g = 0; g < group && ok; ++g) ok = wf_enqueue_local(ctx, s) == MPFMT_OK;
            ok = (hipStreamEndCapture(ctx->stream, &graph) == hipSuccess) && ok && graph != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (ok) {
            rc = MPFMT_OK;
            for (int64_t it = 0; it < max_steps && rc == MPFMT_OK; it += group) {
                if (hipGraphLaunch(exec, ctx->stream) != hipSuccess) { rc = mpfmt_fail(ctx, MPFMT_ERR_HIP, "hipGraphLaunch failed in the wavefront solve"); break; }
                if ((rc = wf_read_ctr(ctx, s, false))) break;
                if (wf_ended(s)) break;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) {
            if (rc) return rc;
            if (!wf_ended(s)) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "wavefront solve did not terminate");
            return MPFMT_OK;
        }
        (void)hipGetLastError();                              // capture not available: plain launches
    }
    for (int64_t it = 0; it < max_steps; it += group) {
        for (int g = 0; g < group; ++g) {
            if ((rc = wf_enqueue_local(ctx, s))) return rc;
            if (s->sharded && (rc = wf_exchange(ctx, s))) return rc;
        }
        if ((rc = wf_read_ctr(ctx, s, false))) return rc;
        if (wf_ended(s)) break;
    }
    if (!wf_ended(s)) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "wavefront solve did not terminate");
    return MPFMT_OK;
}

// Directed steering graphs (quasi-metric spaces: double integrator, Dubins): the resident CSC holds the BACKWARD sets (column x
// = sources y with cost(y -> x) <= r), its per-entry free bits and segment counts come from the space's own sweep
// (mpfmt_di_sweep / mpfmt_car_sweep); the forward sets are the rows, transposed on the device here.  F: checkpts bitmap computed
// by the caller for its space (host words); gd = coordinates the workspace goals read.
extern "C++" int32_t mpfmt_wf_begin_directed(mpfmt_ctx* ctx, int64_t init_idx, int32_t checkpts, const uint64_t* F_host, int32_t goal_kind,
                                const double* goal_params, int32_t gd, double band, int32_t flags)
{
    if (!ctx->di_filled || !ctx->di_swept) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "directed wavefront solve needs a built and swept steering graph");
    if (ctx->world != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "directed wavefront solve runs on an unsharded ctx");
    const int64_t N = ctx->N;
    const int d = ctx->d;
    if (!(band >= 0.0) || !std::isfinite(band)) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "band must be finite and >= 0");
    int32_t rc;
    if (!ctx->wf) ctx->wf = new mpfmt_wf();
    mpfmt_wf* s = wf_of(ctx);
    s->active = false;
    s->t_begin = std::chrono::steady_clock::now();
    if ((rc = wf_alloc(ctx, s, N, 1))) return rc;
    s->sharded = 0; s->directed = true; s->use_mask = 1; s->all_in = 0; s->pos_space = 0;
    s->band = band; s->single = (flags & MPFMT_WF_SINGLE) ? 1 : 0; s->checkpts = checkpts ? 1 : 0;
    s->init = init_idx - 1; s->r = ctx->di_r;
    s->goal.kind = goal_kind; s->goal.gd = goal_kind == MPFMT_GOAL_POINT ? d : gd;
    const int ng = goal_kind == MPFMT_GOAL_RECT ? 2 * gd : goal_kind == MPFMT_GOAL_BALL ? gd + 1 : d;
    memset(s->goal.g, 0, sizeof s->goal.g);
    for (int i = 0; i < ng; ++i) s->goal.g[i] = goal_params[i];
    s->nparts = (int)std::min<int64_t>(WF_MAXPARTS, (s->words + WF_BLK_WORDS - 1) / WF_BLK_WORDS);
    if (s->nparts < 1) s->nparts = 1;
    for (int k = 0; k < 4; ++k) s->prev_tot[k] = 0;
    HIPCHK(ctx, hipMemcpyAsync(s->F, F_host, 8 * (size_t)s->words, hipMemcpyHostToDevice, ctx->stream));
    // forward sets
    const int64_t nnz = ctx->nnz;
    if (s->csr_nnz < std::max<int64_t>(nnz, 1) || !s->rowptr) {
        if (s->rowptr) { HIPCHK(ctx, hipFree(s->rowptr)); s->rowptr = nullptr; }
        if (s->colidx) { HIPCHK(ctx, hipFree(s->colidx)); s->colidx = nullptr; }
        HIPCHK(ctx, hipMalloc((void**)&s->rowptr, sizeof(int64_t) * (size_t)(N + 1)));
        HIPCHK(ctx, hipMalloc((void**)&s->colidx, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
        s->csr_nnz = std::max<int64_t>(nnz, 1);
    }
    if ((rc = mpfmt_csc_transpose_resident(ctx, s->rowptr, s->colidx))) return rc;
    s->ms_graph = 0.0; s->ms_sweep = 0.0;
    hipLaunchKernelGGL(k_wf_init, dim3(256), dim3(64), 0, ctx->stream, N, s->words, s->init, s->W, s->H, s->Z, s->Zp, s->Hn, s->cand, s->C, s->A,
                       s->stats, s->ctr);
    HIPCHK(ctx, hipGetLastError());
    s->active = true;
    return MPFMT_OK;
}

int32_t mpfmt_fmtstar_wavefront(mpfmt_ctx* ctx, double r, int64_t init_idx, int32_t checkpts, int32_t goal_kind, const double* goal_params,
                                double band, int32_t flags, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res, mpfmt_wf_info* info)
{
    if (!ctx || !res) return MPFMT_ERR_ARG;
    int32_t rc;
    if ((rc = mpfmt_wf_begin(ctx, r, init_idx, checkpts, goal_kind, goal_params, band, flags))) return rc;
    mpfmt_wf* s = wf_of(ctx);
    if ((rc = mpfmt_wf_run(ctx))) return rc;
    if ((rc = mpfmt_wf_finish(ctx, A, C, path, res))) return rc;
    if (info) wf_fill_info(s, info);
    return MPFMT_OK;
}

}  // extern "C"

void mpfmt_wf_info_now(mpfmt_ctx* ctx, mpfmt_wf_info* info) { if (info && wf_of(ctx)) wf_fill_info(wf_of(ctx), info); }

