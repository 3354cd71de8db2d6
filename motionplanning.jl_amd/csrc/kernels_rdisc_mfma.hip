// r-disc pair sweep with an fp16 MFMA distance-matrix FILTER and an exact fp64 REFINE (gfx950).
//
// Same contract as k_rdisc in kernels_rdisc.hip (inball for every sample, reference
// src/nearneighbors.jl:179-183; canonical membership  i != v && sum_i (q_i-c_i)^2 <= r*r  in fp64,
// unfused, index order), but the N x d . d^T x N distance-matrix block is evaluated on the matrix cores:
//
//   filter : every sorted sample carries a 16-slot fp16 operand  (u_0..u_{d-1}, 0.., 1, 1, n_hi, n_lo)
//            with u = (x - lo) * s quantised to fp16 and n = |u^|^2.  With the query operand
//            (-2u_0.., n_hi, n_lo, 1, 1) one v_mfma_f32_32x32x16_f16 gives, for 32 queries x 32
//            candidates,   acc = |u^_q|^2 + |u^_c|^2 - 2 u^_q.u^_c - T = |u^_q - u^_c|^2 - T   (C input = -T).
//            u^ are EXACT coordinates of slightly moved points (|u^ - u| <= 2^-12 per coordinate), so
//            | |u^_q-u^_c| - s|q-c| | <= 2 sqrt(d) 2^-12 and the threshold
//                T = (s r + 2 sqrt(d) e_c)^2 + accumulation margin
//            can never reject a true neighbour (proof in LABNOTES.md section 3.1).  It passes a few per cent of extra pairs in a
//            thin shell around the r-ball.  (d <= 6: the K = 8 form, v_mfma_f32_32x32x8_f16, norm and threshold in the C input.)
//   extract: the sign bits of the 16 accumulators are funnelled into one 16-bit word per lane
//            (v_alignbit_b32), lanes with survivors append (query, candidate) to a wave-private LDS queue
//            (__ballot + mbcnt prefix), ~0.3 % of the pairs.
//   refine : when 64 survivors are queued, lane = survivor: exact canonical fp64 d2 from the fp64
//            coordinates, membership test, count or emit (row index, sqrt(d2)).  All 64 lanes busy.
//
// Work mapping: one independent wavefront per (64-query tile, slice of its candidate chunks), no workgroup barriers; the tile's two
// 32-row A fragments stay in VGPRs; B fragments are 1 KB coalesced loads, prefetched two chunks ahead; the chunks within r of the tile's
// sub-boxes are listed once per index build (k_chunk_lists).  MODE 2 (single pass) appends every exact hit to the log of its column's
// quarter tile (in a half build both columns' records, from the lane that holds both ends) and runs the edge tests' broad phase in the drain.
#include "mpfmt_internal.h"
#include "sweep_cmpx.h"
#include "sweep_predicates.h"
#include "mf_operand.h"
#include <algorithm>
#include <cmath>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define NXCD 8
#define MF_THREADS 256
#define MF_WAVES 4
#define MF_QCAP 256                 // survivor queue entries per wavefront (drained in batches of 64)

// a global pointer read through the constant address space: a wave-uniform address then always takes the scalar cache
typedef const __attribute__((address_space(4))) double* mf_cptr;
__device__ __forceinline__ mf_cptr mf_const(const double* p) { return (mf_cptr)(uintptr_t)p; }

struct mf_args {
    const uint4* ops;               // [npad][2] 16 fp16 slots per sorted sample (candidate role)
    const double* Xs;               // [npad][D] sorted AoS fp64 (NaN padded)
    const int32_t* perm;
    const int32_t* cellstart;
    const double* tile_lo;
    const double* tile_hi;
    double r2;                      // exact membership threshold r*r
    double rpad;                    // conservative radius for box pruning
    float negT;                     // -T, filter threshold in normalised squared units
    int32_t S;
    int32_t xcd_mode;               // item -> XCD placement: 0 contiguous range per XCD, 1 round robin, >=2 interleaved groups of that many items
    int64_t blk_begin;              // first 256-query block of the shard
    int64_t nitems;                 // tiles x slices (items_head + the tail tiles x S_tail)
    // The LAST tiles of the launch are cut into more, shorter items (single-pass build): blocks are dispatched in item order, so the
    // kernel ends when the last-dispatched items end -- with every other wave slot already idle for up to an item's duration (~150 us
    // of a 1.7 ms launch at three slices per tile; measured by launching the kernel in two parts: +140 us at a 90 % split).
    int64_t items_head;             // items of the tiles cut into S slices; the tiles after them are cut into S_tail
    int32_t S_tail;
    int64_t npad;
    int64_t ntiles;
    int32_t* slice_cnt;             // [S][npad]
    const int64_t* tptr;            // [npad+1] offsets of the sorted-order staging CSC
    int32_t* rowtmp;
    double* valtmp;
    const uint32_t* lists;          // [tiles of the shard][list_cap] candidate chunk ids of each tile (k_chunk_lists)
    const int32_t* list_len;        // [tiles of the shard]
    int64_t list_cap;
    unsigned long long* pairs;
    unsigned long long* survivors;
    // MODE 2 (single pass): every exact hit is appended -- while it is found -- to the LOG OF ITS COLUMN'S QUARTER TILE (16 consecutive
    // cell-sorted columns): one log per quarter whoever finds the hit (any slice of the column's own tile; in a half build also the
    // tiles before it, which see the pair from the other end).  A record is 12 bytes in two arrays: a key word -- row sample index
    // (26 bits) | column within the quarter << 26 | broad-phase flag << 30 (bit 31: set later by k_exact_pairs on a blocked edge) --
    // and the squared distance.  Places are reserved on the log's global cursor, one returning atomic per (drain, log).
    int32_t* pool_flag;             // set to 1 when a log overflows (the build is then redone in the two-pass form)
    long long qcap;                 // capacity of ONE quarter log, in records
    uint32_t* qkey;                 // [quarters of the shard][qcap]
    double* qd2;                    // [quarters of the shard][qcap]
    int32_t* qlen;                  // [quarters of the shard] cursors (zeroed per build; may run past qcap: readers clamp)
    // half build: the chunk lists hold only chunks >= the tile (inside the shard), every pair is found once, and the hit of a chunk
    // beyond the tile is also written as the record of the OTHER column into that column's quarter log
    int32_t half;
    int64_t ntiles_shard;           // tiles of this shard: the other column's record is written for candidates in [blk_begin, blk_begin + ntiles_shard)
    // broad phase of the edge tests in the drain (half build, d <= 12, <= 256 boxes in the state space's own coordinates, every
    // sample inside the state space): bit 30 of a record's row index = "the segment's box meets an obstacle's: exact test needed"
    int32_t fb;
    int32_t M;
    const double* boxes;            // [M][2][D]
    const unsigned long long* smask;   // [npad] per cell-sorted sample: bit (k & 63) set when box k lies within r (per axis) of the sample
                                    // (k_sample_masks): a segment's box can only meet boxes in the masks of BOTH its ends
    // fb == 2: the pairs whose segment box met an obstacle's are listed (MF_NREG regions of icap 16-byte items, one per (pair, box)
    // unit) for k_exact_pairs, which runs the slab tests in both directions and sets bit 31 of the blocked records' key
    // MODE 3 (streaming, mpfmt_rdisc_stream): no graph is stored; per column the degree, the best open parent argmin_y C[y] + d(y, x)
    // (first minimum in ascending y: fmt.jl:73) and, on request, the number of free edges, as per-slice partials
    const double* st_C;             // [N] cost-to-come by sample index (or nullptr: no parent search)
    const unsigned long long* st_H; // [N bits] open set by sample index (nullptr: every sample is open)
    int32_t st_free;                // count is_free_motion(V[y], V[x]) over the column's entries
    int32_t st_ss_has;
    const double* st_ss;            // state-space bounds lo[MPFMT_MAX_DIM], hi[MPFMT_MAX_DIM] (in_state_space of the first point)
    unsigned long long* st_best;    // [S][npad] bits of the best cost (~0: none)
    int32_t* st_besti;              // [S][npad] its sample index
    int32_t* st_nfree;              // [S][npad]
    uint4* pitems;
    int32_t* pcnt;                  // [MF_NREG] items in each region (zeroed per build; may exceed icap: the reader clamps)
    long long icap;
    int32_t* pend_over;
};

__device__ __forceinline__ int cell_of_m(double x, double lo, double inv_w, int g)
{
    double f = floor((x - lo) * inv_w);
    int c = (int)f;
    if (!(f >= 0.0)) c = 0;
    if (f >= (double)g) c = g - 1;
    return c;
}

// ---- operand construction --------------------------------------------------------------------------------
// one thread per sorted position; writes the candidate-role operand (mf_operand.h).  The index build writes the operands itself
// (k_build_tiles); this kernel serves a ctx whose index was built before the matrix-core path was asked for.
__global__ void k_make_ops(const double* __restrict__ Xs, int64_t N, int64_t npad, int d,
                           mpfmt_grid G, double scale, void* __restrict__ ops)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npad) return;
    double x[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = (i < d && p < N) ? Xs[p * d + i] : 0.0;
    mf_write_operand(ops, G, d, scale, p, p < N, x);
}

// ---- candidate chunk lists -------------------------------------------------------------------------------------
// One wavefront per tile, run once per (grid, radius): the tile's neighbourhood -- grid rows -> contiguous runs of the
// sorted array -> 64-sample chunks -- is flattened lane-parallel, every chunk's tight box is tested against the tile's
// tight box (64 box tests in flight together) and the surviving unique chunk ids are ballot-compacted into the tile's
// list in global memory (ascending).  The pair kernel's items then take contiguous slices of these lists.
// NW = 1: as described.  NW = 4 (small shards, where one wavefront per tile leaves most of the chip idle and the kernel is
// one long latency chain): the tile's rows are split into NW contiguous ranges, one per wavefront; each stages its kept
// ids in LDS, and the ranges are concatenated in order -- a chunk straddling two ranges is kept by both or by neither
// (same box test), so dropping a range's first id when it equals the previous range's last restores the dedupe exactly.
// (built for six wavefronts per SIMD -- 80 registers, no spills: left alone the compiler takes 106, and the sample masks that run beside
// this kernel on the step's side stream then find room for one wavefront per SIMD: 122 us for them instead of 59; eight per SIMD spill)
template <int D, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_chunk_lists(const int32_t* __restrict__ cellstart, const double* __restrict__ tile_lo,
                                                    const double* __restrict__ tile_hi, const double* __restrict__ tile_sub,
                                                    const float* __restrict__ tile_sub32, mpfmt_grid G, double rpad,
                                                    int64_t tile_begin, int64_t nt, int64_t list_cap,
                                                    uint32_t* __restrict__ lists, int32_t* __restrict__ list_len,
                                                    int32_t* __restrict__ max_len, int half, const uint32_t* __restrict__ cellkey, int fb,
                                                    uint32_t* __restrict__ gstage)
{
    __shared__ int32_t s_sega_[NW][64];                   // first chunk of each row's run
    __shared__ int32_t s_segp_[NW][64];                   // exclusive prefix of the runs' chunk counts
    __shared__ int32_t s_wcnt[NW], s_wfirst[NW], s_wlast[NW];
    extern __shared__ uint32_t s_stage[];                 // NW > 1: [NW][list_cap] kept ids of each wavefront's row range
    const int lane = threadIdx.x & 63;
    const int wave = (NW > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    int32_t* const s_sega = s_sega_[wave];
    int32_t* const s_segp = s_segp_[wave];
    const int64_t tl = blockIdx.x;
    if (tl >= nt) return;
    const int64_t tile = tile_begin + tl;
    const double rpad2 = rpad * rpad;
    bool use_sub = false;
#pragma unroll
    for (int i = 0; i < D; ++i) use_sub = use_sub || (G.g[i] >= 3);
    double wlo[D], whi[D];
    double qal[D], qah[D], qbl[D], qbh[D];                // the tile's two sub-boxes (k_tile_bbox)
    int clo[D], chi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        wlo[i] = tile_lo[tile * D + i];
        whi[i] = tile_hi[tile * D + i];
        qal[i] = tile_sub[(tile * 4 + 0) * D + i]; qah[i] = tile_sub[(tile * 4 + 1) * D + i];
        qbl[i] = tile_sub[(tile * 4 + 2) * D + i]; qbh[i] = tile_sub[(tile * 4 + 3) * D + i];
        clo[i] = cell_of_m(wlo[i] - rpad, G.lo[i], G.inv_w[i], G.g[i]);
        chi[i] = cell_of_m(whi[i] + rpad, G.lo[i], G.inv_w[i], G.g[i]);
    }
    // the tile's own sub-boxes in fp32, rounded outward like the candidates' (this tile's row of tile_sub32)
    float fqal[D], fqah[D], fqbl[D], fqbh[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        fqal[i] = tile_sub32[(tile * 4 + 0) * D + i]; fqah[i] = tile_sub32[(tile * 4 + 1) * D + i];
        fqbl[i] = tile_sub32[(tile * 4 + 2) * D + i]; fqbh[i] = tile_sub32[(tile * 4 + 3) * D + i];
    }
    const float frpad2 = (float)(rpad2 * (1.0 + 1e-4));
    constexpr int L = D - 1;
    // The rows (cells of the dimensions before the last) around the tile, in ASCENDING CELL-ID ORDER -- consecutive rows' runs are then
    // consecutive ranges of the sorted order, and a chunk that straddles two runs is seen twice in a row (the dedupe below).  Row-major
    // ids: the mixed-radix count over [clo, chi].  Block-major ids (sharded ctx): the half-block combinations first (axis 0 the most
    // significant), inside each the mixed-radix count over the part of [clo, chi] that lies in that half.
    const int ncombo = 1 << G.nsplit;
    auto part = [&](int i, int cb, int& lo, int& hi) {        // the cells of axis i < L in half-block combination cb
        lo = clo[i]; hi = chi[i];
        if (i < G.nsplit) {
            if ((cb >> (G.nsplit - 1 - i)) & 1) lo = max(lo, G.split[i]); else hi = min(hi, G.split[i] - 1);
        }
    };
    auto combo_rows = [&](int cb) -> uint32_t {
        uint32_t n = 1;
#pragma unroll
        for (int i = 0; i < L; ++i) { int lo, hi; part(i, cb, lo, hi); n *= (uint32_t)max(hi - lo + 1, 0); }
        return n;
    };
    uint32_t rows = 0;                                    // <= number of grid cells <= 2^24
    for (int cb = 0; cb < ncombo; ++cb) rows += combo_rows(cb);

    uint32_t* __restrict__ out = lists + tl * list_cap;
    // (NW > 1: the wavefront's kept ids are staged in LDS -- or, when the lists are too long for it, in a global scratch area)
    uint32_t* const stage = (NW > 1) ? (gstage ? gstage + ((int64_t)blockIdx.x * NW + wave) * list_cap : s_stage + (int64_t)wave * list_cap) : nullptr;
    int32_t gcount = 0;              // unique surviving chunks so far (uniform)
    int64_t carry = -1;              // last chunk id of the previous flattened batch (dedupe)
    int64_t firstkept = -1, lastkept = -1;
    // half build: only chunks >= the tile are wanted.  Rows come in ascending cell order, so every row before the one that holds
    // the tile's first sample lies wholly before the tile: the enumeration starts at that row
    uint32_t rows_lo = 0;
    if (half && tile_begin == 0 && G.nsplit == 0) {           // (a shard keeps the chunks of the shards before it: every row is enumerated)
        int64_t cid = (int64_t)(cellkey[tile * 64] >> fb);    // (a tile's first sample is never a pad)
        uint32_t mul = 1;
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            const int ci = (int)((cid / G.stride[i]) % G.g[i]);
            rows_lo += (uint32_t)max(0, min(ci, chi[i]) - clo[i]) * mul;
            mul *= (uint32_t)(chi[i] - clo[i] + 1);
        }
        rows_lo = min(rows_lo, rows);
    }
    const uint32_t nbatch = (rows - rows_lo + 63) / 64, bq = (nbatch + NW - 1) / NW;
    const uint32_t row_begin = rows_lo + (uint32_t)wave * bq * 64, row_end = min(rows, rows_lo + (uint32_t)(wave + 1) * bq * 64);
    for (uint32_t row0 = row_begin; row0 < row_end; row0 += 64) {
        // lane = one row: candidate run [ca, ca+n) in chunk units
        const uint32_t row = row0 + lane;
        int32_t ca = 0, n = 0;
        if (row < row_end) {
            uint32_t rem = row;
            int cbi = 0;
            for (; cbi < ncombo - 1; ++cbi) { const uint32_t n = combo_rows(cbi); if (rem < n) break; rem -= n; }
            int64_t cbase = 0;
            double partial = 0.0, pa = 0.0, pb = 0.0;      // row cell vs the hull / sub-box A / sub-box B
#pragma unroll
            for (int i = L - 1; i >= 0; --i) {
                int plo, phi;
                part(i, cbi, plo, phi);
                const uint32_t span = (uint32_t)max(phi - plo + 1, 1);
                const uint32_t qd = rem / span;
                const int c = plo + (int)(rem - qd * span);
                rem = qd;
                cbase += mpfmt_cell_term(G, i, c);
                const double eps = G.w[i] * 1e-9;
                const double lo = G.lo[i] + (double)c * G.w[i] - eps;
                const double hi = G.lo[i] + (double)(c + 1) * G.w[i] + eps;
                double gap = fmax(fmax(lo - whi[i], wlo[i] - hi), 0.0);
                double ga = fmax(fmax(lo - qah[i], qal[i] - hi), 0.0);
                double gb = fmax(fmax(lo - qbh[i], qbl[i] - hi), 0.0);
                if (G.g[i] == 1) { gap = 0.0; ga = 0.0; gb = (qbl[i] > qbh[i]) ? gb : 0.0; }
                partial += gap * gap;
                pa += ga * ga; pb += gb * gb;
            }
            if (partial <= rpad2 && fmin(pa, pb) <= rpad2) {
                int c0 = clo[L], c1 = chi[L];
                if (G.g[L] > 1) {
                    const double eps = G.w[L] * 1e-9;
                    while (c0 <= c1) {
                        const double hi = G.lo[L] + (double)(c0 + 1) * G.w[L] + eps;
                        const double gap = fmax(wlo[L] - hi, 0.0);
                        if (partial + gap * gap > rpad2) ++c0; else break;
                    }
                    while (c1 >= c0) {
                        const double lo = G.lo[L] + (double)c1 * G.w[L] - eps;
                        const double gap = fmax(lo - whi[L], 0.0);
                        if (partial + gap * gap > rpad2) --c1; else break;
                    }
                }
                if (c0 <= c1) {
                    const int64_t ra = cellstart[cbase + c0];
                    const int64_t rb = cellstart[cbase + c1 + 1];
                    if (rb > ra) { ca = (int32_t)(ra >> 6); n = (int32_t)(((rb - 1) >> 6) - (ra >> 6) + 1); }
                }
            }
        }
        // exclusive prefix of n over the lanes
        int32_t incl = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int32_t T = __shfl(incl, 63);
        __builtin_amdgcn_wave_barrier();
        s_sega[lane] = ca;
        s_segp[lane] = incl - n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int32_t t0 = 0; t0 < T; t0 += 64) {
            const int32_t t = t0 + lane;
            const bool act = t < T;
            // run owning flattened index t: the largest j with s_segp[j] <= t
            int j = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const int jj = j + step;
                if (jj < 64 && s_segp[jj] <= t) j = jj;
            }
            int64_t c = act ? (int64_t)s_sega[j] + (t - s_segp[j]) : -2;
            // dedupe: consecutive runs may share their boundary chunk
            int64_t prevc = __shfl_up(c, 1);
            if (lane == 0) prevc = carry;
            const int lastl = min(63, T - t0 - 1);
            carry = __shfl(c, lastl);
            // half build: the chunks before the tile find these pairs -- inside the shard; a chunk of another shard is nobody's but ours
            bool keep = act && (c != prevc) && (!half || c >= tile || c < tile_begin || c >= tile_begin + nt);
            if (keep && !use_sub) {                       // coarse grid (<= 2 cells per dimension): every cell neighbours every other
                double gap2 = 0.0;                        // one, a tile running over a row end loses nothing -- hull against hull
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double gp = fmax(fmax(tile_lo[c * D + i] - whi[i], wlo[i] - tile_hi[c * D + i]), 0.0);
                    gap2 += gp * gp;
                }
                keep = gap2 <= rpad2;
            } else if (keep) {
                // query sub-box x candidate sub-box, in fp32 on boxes rounded outward (both sides): every gap is a lower bound of
                // the fp64 one up to a few 1e-7 relative, which the 1e-4 on the threshold covers -- never a chunk less, and the
                // arithmetic (the bound of this kernel: ~150 lane-ops per candidate) at the fp32 rate; candidate boxes are 16 D
                // bytes each instead of 32 D.  (Requesting the next 64 candidates' boxes before these are tested was measured: no
                // gain -- the registers it takes cost the residency it was meant to cover for.)
                float gaa = 0.f, gab = 0.f, gba = 0.f, gbb = 0.f;
                const float* __restrict__ cs = tile_sub32 + c * 4 * D;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const float cal = cs[i], cah = cs[D + i], cbl = cs[2 * D + i], cbh = cs[3 * D + i];
                    const float g0 = fmaxf(fmaxf(cal - fqah[i], fqal[i] - cah), 0.f);
                    const float g1 = fmaxf(fmaxf(cbl - fqah[i], fqal[i] - cbh), 0.f);
                    const float g2 = fmaxf(fmaxf(cal - fqbh[i], fqbl[i] - cah), 0.f);
                    const float g3 = fmaxf(fmaxf(cbl - fqbh[i], fqbl[i] - cbh), 0.f);
                    gaa += g0 * g0; gab += g1 * g1; gba += g2 * g2; gbb += g3 * g3;
                }
                keep = fminf(fminf(gaa, gab), fminf(gba, gbb)) <= frpad2;
            }
            const unsigned long long m = __ballot(keep);
            const int32_t gidx = gcount + (int32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (NW == 1) { if (keep && gidx < list_cap) out[gidx] = (uint32_t)c; }
            else {
                if (keep && gidx < list_cap) stage[gidx] = (uint32_t)c;
                if (m) {
                    if (firstkept < 0) firstkept = __shfl(c, __builtin_ctzll(m));
                    lastkept = __shfl(c, 63 - __builtin_clzll(m));
                }
            }
            gcount += (int32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // the stored length never exceeds what was written (a truncated list voids the build: max_len tells the host / k_spec_check)
    if (NW == 1) {
        // (the maximum goes to the atomic only when it beats what is there: one atomic per tile on the one address took most of this
        // kernel's time -- ~88 atomics per microsecond)
        if (lane == 0) { list_len[tl] = min(gcount, (int32_t)list_cap); if (gcount > *(volatile int32_t*)max_len) atomicMax(max_len, gcount); }
        return;
    }
    if (lane == 0) { s_wcnt[wave] = gcount; s_wfirst[wave] = (int32_t)firstkept; s_wlast[wave] = (int32_t)lastkept; }
    __syncthreads();
    int32_t off = 0, total = 0, mydrop = 0, prevlast = -1;
#pragma unroll
    for (int v = 0; v < NW; ++v) {
        const int32_t cv = s_wcnt[v];
        const int32_t drop = (cv > 0 && prevlast >= 0 && s_wfirst[v] == prevlast) ? 1 : 0;
        if (v == wave) { off = total; mydrop = drop; }
        total += cv - drop;
        if (cv > 0) prevlast = s_wlast[v];
    }
    const int32_t staged = min(gcount, (int32_t)list_cap);
    for (int32_t i = mydrop + lane; i < staged; i += 64) {
        const int32_t o = off + i - mydrop;
        if (o < list_cap) out[o] = stage[i];
    }
    if (threadIdx.x == 0) { list_len[tl] = min(total, (int32_t)list_cap); if (total > *(volatile int32_t*)max_len) atomicMax(max_len, total); }
}

// ---- per-sample obstacle masks for the broad phase in the drain ------------------------------------------------------
// One wavefront per tile, lane = sample: bit (k & 63) of a sample's mask is set unless box k is farther than r from the sample along
// some axis (the negated comparisons of boxesND.jl:44-45, so a NaN bound keeps the bit).  Both ends of an edge lie within r of each
// other, hence the edge's box [min, max] lies within r of either end along every axis: a box that meets it has its bit in BOTH ends'
// masks.  The drain ORs (mask_q & mask_c) over its 64 pairs and walks only those boxes of the tile's cull -- 12 instead of 20 at
// the north star (measured on the host, tools/sim_drain_masks.py).
template <int D>
__global__ __launch_bounds__(256) void k_sample_masks(const double* __restrict__ Xs, const double* __restrict__ tile_lo, const double* __restrict__ tile_hi,
                                                      int64_t tile_begin, int64_t nt, double rpad, const double* __restrict__ boxes, int M,
                                                      unsigned long long* __restrict__ smask, unsigned long long* __restrict__ tile_bs,
                                                      const uint8_t* __restrict__ tileneed, unsigned long long* __restrict__ zero, int64_t zero_words)
{
    // (beside the chunk lists this kernel also clears the step's counter arena, a slice per workgroup: as a fill of its own on the
    // lowest-priority stream the 260 KB waited up to 70 us for a wave slot)
    if (zero) {
        const int64_t per = (zero_words + gridDim.x - 1) / gridDim.x;
        const int64_t z0 = (int64_t)blockIdx.x * per, z1 = min(zero_words, z0 + per);
        for (int64_t z = z0 + threadIdx.x; z < z1; z += blockDim.x) zero[z] = 0ull;
    }
    const int lane = threadIdx.x & 63;
    const int64_t tl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tl >= nt) return;
    const int64_t tile = tile_begin + tl;
    if (tileneed && !tileneed[tile]) return;                  // (a tile this rank never reads: shard + halo index)
    const double rm = rpad * (1.0 + 1e-6) + 1e-300;
    double xl[D], xh[D], ulo[D], uhi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double x = Xs[(tile * 64 + lane) * D + i];
        xl[i] = x - rm; xh[i] = x + rm;
        ulo[i] = tile_lo[tile * D + i] - rm; uhi[i] = tile_hi[tile * D + i] + rm;
    }
    // the boxes that survive the tile's cull (lane = box) are staged in LDS, then every lane (= sample) walks them with broadcast reads
    // (wave-uniform scalar loads were a chain of ~20 scalar-cache round trips per wavefront: 46 us for the 15 625 tiles)
    __shared__ double s_bx[4][64][2 * D];
    const int wv = threadIdx.x >> 6;
    unsigned long long mask = 0;
    for (int c = 0; c * 64 < M; ++c) {
        const int kbx = c * 64 + lane;
        int out = 0;
        double bl[D], bh[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { bl[i] = 0.0; bh[i] = 0.0; }
        if (kbx < M) {
            const double* bp = boxes + (int64_t)kbx * 2 * D;
#pragma unroll
            for (int i = 0; i < D; ++i) { bl[i] = bp[i]; bh[i] = bp[D + i]; out |= (int)(bh[i] < ulo[i]) | (int)(bl[i] > uhi[i]); }
        }
        const bool sv = kbx < M && !out;
        const unsigned long long mb = __ballot(sv);
        if (lane == 0) tile_bs[tile * 4 + c] = mb;            // the tile's surviving boxes: what the pair kernel's items of this tile walk
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
        if (sv) {
#pragma unroll
            for (int i = 0; i < D; ++i) { s_bx[wv][slot][i] = bl[i]; s_bx[wv][slot][D + i] = bh[i]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned long long rem = mb;
        for (int k = 0; rem; ++k) {
            const int b = __ffsll((long long)rem) - 1;
            rem &= rem - 1;
            int o2 = 0;
#pragma unroll
            for (int i = 0; i < D; ++i) o2 |= (int)(s_bx[wv][k][D + i] < xl[i]) | (int)(s_bx[wv][k][i] > xh[i]);
            if (!o2) mask |= 1ull << b;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < 4 && lane * 64 >= M) tile_bs[tile * 4 + lane] = 0ull;
    smask[tile * 64 + lane] = mask;                           // (pad samples: NaN coordinates, every comparison false -- every surviving bit set; never read for a hit)
}

int32_t mpfmt_launch_sample_masks(mpfmt_ctx* ctx, double r, void* zero, size_t zero_bytes, bool* zeroed)
{
    if (zeroed) *zeroed = false;
    // (every tile, not the shard's own: a shard's candidates come from all of them)
    const int64_t nt = ctx->ntiles;
    int32_t rc;
    // (per-sample masks [npad], then the four survivor words of every tile)
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->smask, sizeof(unsigned long long) * (size_t)std::max<int64_t>(ctx->ntiles * 68, 1)))) return rc;
    if (nt <= 0 || ctx->tile_end <= ctx->tile_begin) return MPFMT_OK;
    const double rpad = r * (1.0 + 1e-9) + 1e-300;
    void* const zp = (zero && zero_bytes % 8 == 0 && ctx->d >= 1 && ctx->d <= 12) ? zero : nullptr;
#define CASE(DD) case DD: hipLaunchKernelGGL((k_sample_masks<DD>), dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, ctx->stream, ctx->Xs, ctx->tile_lo, ctx->tile_hi, \
        (int64_t)0, nt, rpad, ctx->boxes, ctx->M, (unsigned long long*)ctx->smask, (unsigned long long*)ctx->smask + ctx->ntiles * 64, (const uint8_t*)ctx->tileneed, \
        (unsigned long long*)zp, (int64_t)(zero_bytes / 8)); break;
    switch (ctx->d) { CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) default: break; }
#undef CASE
    HIPCHK(ctx, hipGetLastError());
    if (zeroed) *zeroed = zp != nullptr;
    return MPFMT_OK;
}

// ---- the kernel ---------------------------------------------------------------------------------------------
// One independent wavefront per (tile, slice of the tile's chunk list): no workgroup barriers.
//   main loop: for each listed chunk, the B fragments are ONE 16 B/lane coalesced buffer load (d <= 6; two for 7 <= d <= 12) whose
//              address is a per-lane constant plus a SCALAR chunk offset (the chunk id comes out of a register-held block
//              of the list with v_readlane: no LDS round trip, no vector address arithmetic), prefetched PF chunks ahead;
//              4 MFMAs (2 query row blocks x 2 candidate column blocks), sign-bit extraction (16 v_alignbit per MFMA: the
//              floor for reading 1024 accumulators, and on gfx950 a half-rate VALU op -- tools/ubench/extract_rates*.hip),
//              then lanes whose 64 sign bits are not all clear push ONE record (chunk, lane, bits) to a wave-private LDS
//              record queue -- a ballot, a prefix count and two LDS writes per chunk instead of a loop over the bits
//   expand   : 64 records at a time (lane = record): prefix sum of the popcounts, every lane then writes its own
//              survivors (chunk, lane, bit) into the survivor queue at its own offset -- no ballot per bit
//   refine   : 64 survivors at a time (lane = survivor): canonical fp64 d2, membership test, and in the single-pass mode
//              the hits of the batch are appended -- ballot-compacted, one contiguous 16 B/lane store -- to the item's LOG
//              (full lines; k_order_logs regroups a tile's logs by column through LDS and writes the ordered CSC)
#define MF_NREG 1024                // regions of the pending-pair list (fb == 2)
#define MF_RCAP 128                 // record queue entries per wavefront (expanded 64 at a time)
#define MF_QSZ 320                  // survivor queue entries per wavefront (drained 64 at a time)
#ifndef MF_ABLATE                   // timing experiments only (results invalid): 1 skip extraction, 2 skip refine, 4 skip MFMA
#define MF_ABLATE 0
#endif

// MODE 0: count only   1: fill the staging CSC (needs offsets from a count pass)   2: count AND append hits to the item's log
// W4: built for 4 wavefronts per SIMD (128 VGPRs)
// VF: the filter is the canonical fp64 test itself on the vector ALUs instead of the fp16 matrix-core one -- for worlds whose radius lies
// below the fp16 shell of globally normalised coordinates (large low-dimensional worlds: 2-D, N >= 1e5 at small degrees).  The same
// pipeline otherwise: chunk lists, the 64-bit hit word per lane in the accumulator layout, record queue, drain, logs, fused edge tests.
template <int D, int MODE, bool W4, bool VF = false>
__device__ __forceinline__ void rdisc_mfma_body(mf_args a, mpfmt_grid G)
{
    __shared__ double s_q[64 * D];                        // fp64 query coordinates (AoS) for the refine
    __shared__ uint32_t s_rm[MF_RCAP];                    // record queue: chunk << 6 | finding lane
    __shared__ unsigned long long s_rh[MF_RCAP];          //               the lane's 64 sign bits of that chunk
    __shared__ uint32_t s_qs[MF_QSZ];                     // survivor queue: chunk << 12 | finding lane << 6 | sign-bit position
    __shared__ int32_t s_cnt[64];
    __shared__ int32_t s_operm[MODE == 2 ? 64 : 1];       // caller indices of the tile's own samples (the other column's record carries the query's)
    __shared__ int32_t s_lc[4];                           // own hits of the drain in work, per quarter of the tile (zero between drains)
    __shared__ unsigned long long s_qm[MODE == 2 ? 64 : 1];      // the queries' obstacle masks (broad phase in the drain)
    __shared__ int64_t s_base[MODE == 1 ? 64 : 1];
    __shared__ unsigned long long s_best[MODE == 3 ? 64 : 1];
    __shared__ int32_t s_besti[MODE == 3 ? 64 : 1], s_nfree[MODE == 3 ? 64 : 1];

    const int lane = threadIdx.x;
    const int64_t nblk = gridDim.x;
    const int64_t per_xcd = nblk / NXCD;
    const int64_t b = blockIdx.x;
    // block b runs on XCD b % 8 (private L2).  Items are handed to XCDs in interleaved groups: neighbouring tiles
    // (which share candidate chunks) stay on one L2, while every XCD sees the same mix of boundary / interior
    // tiles (whose work differs by >1.5x), so no XCD runs dry early.
    int64_t item;
    if (a.xcd_mode == 0) item = (b % NXCD) * per_xcd + (b / NXCD);
    else if (a.xcd_mode == 1) item = b;
    else {
        const int64_t gsz = a.xcd_mode;
        const int64_t x = b % NXCD, k = b / NXCD;              // k-th block of XCD x
        item = ((k / gsz) * NXCD + x) * gsz + (k % gsz);
    }
    if (item >= a.nitems) return;
    const bool tail_item = item >= a.items_head;
    const int Scur = tail_item ? a.S_tail : a.S;             // (odd: mpfmt_slices_for)
    const int64_t irel = tail_item ? item - a.items_head : item;
    const int64_t tile = a.blk_begin + (tail_item ? a.items_head / a.S : 0) + irel / Scur;
    const int slice = (int)(irel % Scur);
    const int64_t qpos = tile * 64 + lane;
    const int kb = lane >> 5, col = lane & 31;

    // ---- broad phase in the drain: the obstacles that can meet a segment out of this tile at all (lane = box).  Both ends of such a
    // segment lie within r of the tile's hull (the candidate is within r of a query of the tile), so a box farther than that from
    // the hull in some axis is out for the whole item
    unsigned long long bsurv[4] = {0ull, 0ull, 0ull, 0ull};
    if constexpr (MODE == 2) {
        if (a.fb) {
            // (culled once per tile by k_sample_masks -- with its slightly wider margin, a superset of the boxes within rpad of the hull;
            // the comparisons of the drain decide -- instead of once per item here: 4 x 12 loads and 24 comparisons per lane)
            const unsigned long long* tb = a.smask + a.npad + tile * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) bsurv[c] = tb[c];
        }
    }

    // half build, the tile's own chunk: the 64 x 64 block holds every pair of the tile's samples twice.  Only the sign bits of (query ql,
    // candidate cj) with cj > ql are kept -- the pair is refined, its segment's box tested and its item written once, and the record of
    // the other column goes to the tile's own logs like any other "same pair from the other end".  Bit 16 t + 15 - r of a lane's word
    // is query (t & 1) 32 + g(r) + 4 (lane >> 5), g(r) = (r & 3) + 8 (r >> 2) increasing in r, candidate (t >> 1) 32 + (lane & 31): per
    // 16-bit group the kept r are those below a count.
    [[maybe_unused]] unsigned long long own_keep = ~0ull;
    if constexpr (MODE == 2) {
        if (a.half) {
            own_keep = 0ull;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int A = (t >> 1) * 32 + (lane & 31) - (t & 1) * 32 - 4 * (lane >> 5);      // keep g(r) < A
                const int Ac = min(max(A, 0), 32);
                const int R = min(16, 4 * (Ac >> 3) + min(Ac & 7, 4));                              // number of r with g(r) < A
                const unsigned long long m16 = R ? ((0xFFFFull << (16 - R)) & 0xFFFFull) : 0ull;
                own_keep |= m16 << (16 * t);
            }
        }
    }

    // ---- setup: fp64 query coordinates to LDS, A fragments to VGPRs, counters --------------------------------
#pragma unroll
    for (int i = 0; i < D; ++i) s_q[lane * D + i] = a.Xs[qpos * D + i];
    s_cnt[lane] = 0;
    if (lane < 4) s_lc[lane] = 0;
    if constexpr (MODE == 2) s_operm[lane] = a.perm[qpos];
    if constexpr (MODE == 2) { if (a.fb) s_qm[lane] = a.smask[qpos]; }
    if constexpr (MODE == 3) { s_best[lane] = ~0ull; s_besti[lane] = 0x7fffffff; s_nfree[lane] = 0; }
    constexpr bool FILL = (MODE == 1);
    if (FILL) {
        int64_t base = a.tptr[qpos];
        for (int s = 0; s < slice; ++s) base += a.slice_cnt[(int64_t)s * a.npad + qpos];
        s_base[lane] = base;
    }
    // d <= 6: K = 8 operands (16 B per sample, v_mfma_f32_32x32x8_f16), half the operand traffic of the K = 16 form.
    // K = 8 operand layout (k_make_ops): [chunk][kb][col][half] x 4 fp16 -- lane (kb, col) of a B fragment load gets slots
    // 4 kb .. 4 kb + 3 of samples col and 32 + col of the chunk in ONE 16-byte load.
    constexpr bool K8 = (D <= 6);
    half8 aF[2];
    half4 aF4[2];
    f32x16 cinit[2];                                          // K8: C input = |u_q|^2 - T per accumulator row
    f32x16 zero16;
#pragma unroll
    for (int k = 0; k < 16; ++k) zero16[k] = 0.0f;
    if constexpr (VF) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) { aF[rb] = half8{}; aF4[rb] = half4{}; cinit[rb] = zero16; }
    } else if constexpr (K8) {
        const uint2* __restrict__ ops2 = reinterpret_cast<const uint2*>(a.ops);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const uint2 raw = ops2[(tile * 64 + kb * 32 + col) * 2 + rb];
            union { uint2 u; _Float16 h[4]; } cv; cv.u = raw;
            half4 v;
            if (kb == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (_Float16)(-2.0f * (float)cv.h[k]);
            } else {
                v[0] = (_Float16)(-2.0f * (float)cv.h[0]); v[1] = (_Float16)(-2.0f * (float)cv.h[1]);
                v[2] = (_Float16)1.0f; v[3] = (_Float16)1.0f;          // multiply the candidate's n_hi, n_lo
            }
            aF4[rb] = v;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
                union { uint2 u; _Float16 h[4]; } qn; qn.u = ops2[(tile * 64 + 32 + row) * 2 + rb];      // slots 4..7 of sample rb*32 + row
                cinit[rb][r] = ((float)qn.h[2] + (float)qn.h[3]) + a.negT;
            }
            aF[rb] = half8{};
        }
    } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const uint4 raw = a.ops[(tile * 64 + rb * 32 + col) * 2 + kb];
            union { uint4 u; _Float16 h[8]; } cv; cv.u = raw;
            half8 v;
            if (kb == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (_Float16)(-2.0f * (float)cv.h[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (_Float16)(-2.0f * (float)cv.h[k]);
                // |u_q|^2 - T, re-split into hi + lo: the threshold rides in the query operand, so the MFMA's C
                // input is the inline constant 0 and sign(acc) <=> filtered distance below the threshold
                const float nT = ((float)cv.h[6] + (float)cv.h[7]) + a.negT;
                const _Float16 nh = (_Float16)nT;
                v[4] = nh; v[5] = (_Float16)(nT - (float)nh);
                v[6] = (_Float16)1.0f; v[7] = (_Float16)1.0f;
            }
            aF[rb] = v;
            aF4[rb] = half4{};
            cinit[rb] = zero16;
        }
    }

    // ---- refine: exact fp64 test of n queued survivors (lane = survivor) ---------------------------------------
    int qcount = 0;                                           // wave-uniform survivor queue length
    int rcount = 0;                                           // wave-uniform record queue length
    int pool_over = 0;
    [[maybe_unused]] int ndrain = 0;                          // drains of this item so far (wave-uniform): spreads its pending-pair items over the regions
    auto drain = [&](int n) {
        // takes the LAST n queue entries (order is irrelevant: columns are sorted afterwards), so nothing moves
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int first = qcount - n;
        qcount = __builtin_amdgcn_readfirstlane(first);
        if (MF_ABLATE & 2) return;
        bool hit = false;
        uint32_t jg = 0, ql = 0;
        double d2 = 0.0;
        [[maybe_unused]] uint32_t e_pj = 0;
        [[maybe_unused]] unsigned long long e_sm = 0;
        [[maybe_unused]] double sl[MODE == 2 ? D : 1], sh[MODE == 2 ? D : 1];
        [[maybe_unused]] double qv[MODE == 3 ? D : 1], cv[MODE == 3 ? D : 1];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < D; ++i) { sl[i] = (double)INFINITY; sh[i] = -(double)INFINITY; }
        }
        if (lane < n) {
            // decode (the queue stores the raw coordinates of the sign bit: decoding once per drain -- 64 survivors wide -- is an
            // order of magnitude cheaper than in the extraction, which runs per chunk with a handful of active lanes)
            const uint32_t e = s_qs[first + lane];
            const int bpos = (int)(e & 63u), fl = (int)((e >> 6) & 63u);
            const uint32_t qc = e >> 12;
            const int t = bpos >> 4, r = 15 - (bpos & 15);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (fl >> 5);
            jg = qc * 64u + (uint32_t)((t >> 1) * 32 + (fl & 31));
            ql = (uint32_t)((t & 1) * 32 + row);
            // (requested with the candidate's coordinates, not after the membership test: one dependent round trip less per drain)
            if constexpr (MODE == 2) {
                e_pj = (uint32_t)a.perm[jg];
                if (a.fb) e_sm = a.smask[jg];
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double qi = s_q[ql * D + i], ci = a.Xs[(int64_t)jg * D + i];
                if constexpr (MODE == 3) { qv[i] = qi; cv[i] = ci; }
                if constexpr (MODE == 2) {
                    asm("v_min_f64 %0, %1, %2" : "=v"(sl[i]) : "v"(qi), "v"(ci));     // the segment's box (only compared: -0 / +0 do not show)
                    asm("v_max_f64 %0, %1, %2" : "=v"(sh[i]) : "v"(qi), "v"(ci));
                }
                const double t = qi - ci;
                const double tt = t * t;
                d2 = (i == 0) ? tt : d2 + tt;
            }
            hit = (d2 <= a.r2) && ((int64_t)jg != tile * 64 + (int64_t)ql);
            if (hit) {
                if (FILL) {
                    const int slot = atomicAdd(&s_cnt[ql], 1);
                    const int64_t pos = s_base[ql] + slot;
                    a.rowtmp[pos] = a.perm[jg];
                    a.valtmp[pos] = sqrt(d2);
                } else if (MODE == 0 || MODE == 3) {
                    atomicAdd(&s_cnt[ql], 1);                             // the column's degree (no return value needed; the single pass counts from the logs)
                }
            }
        }
        if constexpr (MODE == 3) {
            // ---- streaming reductions of the column ql over this drain's hits ----
            if (a.st_C) {
                // best open parent: lexicographic minimum of (C[y] + dist, y) -- findmin returns the FIRST minimum of the ascending
                // neighbourhood (fmt.jl:73).  Costs are non-negative doubles: their bit patterns order like the values.
                unsigned long long mine = ~0ull, old = ~0ull;
                int y = -1;
                if (hit) {
                    y = a.perm[jg];
                    const bool open = a.st_H ? ((a.st_H[y >> 6] >> (y & 63)) & 1ull) != 0 : true;
                    if (open) {
                        const double cost = a.st_C[y] + sqrt(d2);
                        if (cost >= 0.0) mine = (unsigned long long)__double_as_longlong(cost);      // (NaN: no candidate)
                    }
                    old = s_best[ql];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (mine != ~0ull) atomicMin(&s_best[ql], mine);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const unsigned long long now = hit ? s_best[ql] : ~0ull;
                if (hit && now < old) s_besti[ql] = 0x7fffffff;       // the minimum moved in this drain: the index of the old one is out
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                if (mine != ~0ull && mine == now) atomicMin(&s_besti[ql], y);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            }
            if (a.st_free) {
                // is_free_motion(V[y], V[x], CC, SS) (statespaces.jl:153-158): in_state_space of the first point, then the segment against
                // every box -- broad phase, exact slab test where the boxes meet (boxesND.jl:44-56); boxes through the scalar cache
                bool fr = hit;
                if (a.st_ss_has) {
                    const mf_cptr sp = mf_const(a.st_ss);
                    int ok = 1;
#pragma unroll
                    for (int i = 0; i < D; ++i) ok &= (int)(sp[i] <= cv[i]) & (int)(cv[i] <= sp[MPFMT_MAX_DIM + i]);
                    fr = fr && ok != 0;
                }
                double l[D], h[D];
                seg_bbox<D>(cv, qv, l, h);
                for (int k = 0; k < a.M; ++k) {
                    const mf_cptr bp = mf_const(a.boxes) + (int64_t)k * 2 * D;
                    if constexpr (D > 6) {
                        // the axes six at a time; the rest of the box is fetched only where a lane's segment box meets it so far
                        const unsigned long long pend = sweep_cmpx_groups<D>(__ballot(fr), bp, l, h);
                        if (pend) {
                            box_regs<D> bx;
#pragma unroll
                            for (int i = 0; i < D; ++i) { bx.lo[i] = bp[i]; bx.hi[i] = bp[D + i]; }
                            const bool f = narrow_free_sl<D>(cv, qv, bx);
                            if ((pend >> lane) & 1ull) fr = f;
                        }
                    } else {
                        box_regs<D> bx;
#pragma unroll
                        for (int i = 0; i < D; ++i) { bx.lo[i] = bp[i]; bx.hi[i] = bp[D + i]; }
                        const bool meet = fr && !broadphase_free_sl<D>(l, h, bx);
                        if (__ballot(meet)) { const bool f = narrow_free_sl<D>(cv, qv, bx); if (meet) fr = f; }
                    }
                }
                if (fr) atomicAdd(&s_nfree[ql], 1);
            }
        }
        // (MODE 2) the places of this drain's records are reserved BEFORE the broad phase: the returning atomics on the logs' cursors -- a
        // device-scope round trip -- run beside the ~170 comparisons of the box walk instead of after it; only the pending-pair region
        // (whose size the walk decides) is reserved afterwards
        [[maybe_unused]] int g = 0, pin = 0, fq = 0, leader = 0, pre = 0, cnt_l = 0, obase = 0, fbase = 0;
        [[maybe_unused]] int64_t fc = 0;
        [[maybe_unused]] bool fh = false;
        if constexpr (MODE == 2) {
            // own record: column = the query ql of this tile, row = the candidate -> log (tile, ql >> 4); the place inside the drain's
            // group is a returning LDS atomic on a per-drain counter (four addresses; the counters are left at zero again)
            g = (int)(ql >> 4);
            if (hit) pin = atomicAdd(&s_lc[g], 1);
            // the same pair seen from the other end (half build): column jg, row = this query -> the log of jg's quarter tile.  The hits
            // of a drain fall into a handful of such logs (its survivors come from two or three chunks): the lanes of each are found with
            // one ballot per log, the first of them reserves the places of all
            fc = (int64_t)(jg >> 6) - a.blk_begin;        // the candidate's tile, counted from the shard's first
            fh = a.half && hit && fc >= 0 && fc < a.ntiles_shard;      // (the tile's own chunk too: its pairs are kept once, own_keep)
            fq = (int)(fc * 4 + (int64_t)((jg & 63u) >> 4));
            leader = lane;
            {
                unsigned long long rem = __ballot(fh);
                while (rem) {
                    const int L = __builtin_ctzll(rem);
                    const int key = __builtin_amdgcn_readlane(fq, L);
                    const bool mine = fh && fq == key;
                    const unsigned long long mm = __ballot(mine);
                    if (mine) { leader = L; pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u)); }
                    if (lane == L) cnt_l = (int)__popcll(mm);
                    rem &= ~mm;
                }
            }
            // the reservations of the drain are requested back to back -- own logs (lanes 0..3), the other columns' logs (group leaders),
            // and consumed after the box walk
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 4) {
                const int c = s_lc[lane];
                if (c) { s_lc[lane] = 0; obase = atomicAdd(&a.qlen[(tile - a.blk_begin) * 4 + lane], c); }
            }
            if (fh && lane == leader) fbase = atomicAdd(&a.qlen[fq], cnt_l);
        }
        // broad phase of is_free_motion for the hits of this drain (boxesND.jl:44-45, symmetric in the two end points, so one test
        // serves both records of a pair): the boxes that survived the tile's cull, each through the scalar cache, 2 D v_cmpx in a row
        uint32_t pendflag = 0;
        [[maybe_unused]] unsigned pk_keep = 0, pc_keep = 0;
        if constexpr (MODE == 2) {
            if (a.fb) {
                unsigned pk = 0, pc = 0;
                sl[0] = hit ? sl[0] : (double)INFINITY;       // lanes without a hit fail the first comparison
                [[maybe_unused]] const unsigned long long hitm = __ballot(hit);
                // the boxes some pair of this drain can meet at all: OR over the lanes of (mask of the query & mask of the candidate)
                unsigned long long um = 0;
#if !(MF_ABLATE & 16)
                if (hit) um = s_qm[ql] & e_sm;
                {
                    uint32_t ul = (uint32_t)um, uh = (uint32_t)(um >> 32);
#define MF_OR_DPP(x, ctrl, rm_, bc) x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rm_, 0xf, bc)
                    MF_OR_DPP(ul, 0x111, 0xf, true); MF_OR_DPP(uh, 0x111, 0xf, true);        // row_shr:1
                    MF_OR_DPP(ul, 0x112, 0xf, true); MF_OR_DPP(uh, 0x112, 0xf, true);        // row_shr:2
                    MF_OR_DPP(ul, 0x114, 0xf, true); MF_OR_DPP(uh, 0x114, 0xf, true);        // row_shr:4
                    MF_OR_DPP(ul, 0x118, 0xf, true); MF_OR_DPP(uh, 0x118, 0xf, true);        // row_shr:8
                    MF_OR_DPP(ul, 0x142, 0xa, false); MF_OR_DPP(uh, 0x142, 0xa, false);      // row_bcast:15
                    MF_OR_DPP(ul, 0x143, 0xc, false); MF_OR_DPP(uh, 0x143, 0xc, false);      // row_bcast:31
#undef MF_OR_DPP
                    um = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)uh, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)ul, 63);
                }
#else
                um = ~0ull;
#endif
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned long long mb = bsurv[c] & um;
                    while (mb) {
                        const int kbx = c * 64 + (__ffsll((long long)mb) - 1);
                        mb &= mb - 1;
                        const mf_cptr bp = mf_const(a.boxes) + (int64_t)kbx * 2 * D;
                        if constexpr (D <= 6) {
                            double blo[D], bhi[D];
#pragma unroll
                            for (int i = 0; i < D; ++i) { blo[i] = bp[i]; bhi[i] = bp[D + i]; }
                            sweep_cmpx<D>::note(blo, bhi, sl, sh, pk, pc, kbx);
                        } else {
                            // 7 <= d <= 12: the axes six at a time (sweep_cmpx.h), the later bounds fetched only while a lane is left -- in
                            // R^12 few (segment, box) pairs survive six axes.  The same list of the last four boxes met and their number.
                            const unsigned long long pend = sweep_cmpx_groups<D>(hitm, bp, sl, sh);
                            if (pend) {
                                const bool pl = (pend >> lane) & 1ull;
                                pk = pl ? ((pk << 8) | (unsigned)kbx) : pk;
                                pc += pl ? 1u : 0u;
                            }
                        }
                    }
                }
                pendflag = pc ? (1u << 30) : 0u;
                pk_keep = pk; pc_keep = pc;
            }
        }
        if constexpr (MODE == 2) {
            // ---- single pass: the records of this drain's hits go to the quarter logs ------------------------------------------------
            // fb == 2: one item per (pair, box) unit of the pairs whose box met an obstacle's -- a pair that met k <= 4 boxes writes k
            // items, one that met more a single item that stands for "every box" -- in one of MF_NREG dense regions (this item's: item
            // mod MF_NREG), one reservation per drain
            [[maybe_unused]] int ibase = 0, iexcl = 0, iunits = 0;
            [[maybe_unused]] unsigned long long ipm = 0;
            [[maybe_unused]] bool iem = false;
            // (the region changes from drain to drain: in a low-dimensional world ONE item -- the slice that holds the tile's own chunk --
            // finds most of a tile's pairs, and by item alone its units overflowed a region sized for the mean)
            [[maybe_unused]] const int iregion = (int)((item + (int64_t)ndrain * 131) & (MF_NREG - 1));
            ndrain = __builtin_amdgcn_readfirstlane(ndrain + 1);
            {
                if (a.fb == 2) {
                    iem = hit && pendflag != 0;
                    ipm = __ballot(iem);
                    if (ipm) {
                        iunits = iem ? (pc_keep > 4u ? 1 : (int)pc_keep) : 0;
                        int incl = iunits;                                        // inclusive scan over the lanes (DPP)
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);       // row_shr:1
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);       // row_shr:2
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);       // row_shr:4
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);       // row_shr:8
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15
                        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31
                        iexcl = incl - iunits;
                        const int total = __builtin_amdgcn_readlane(incl, 63);
                        if (lane == 0) ibase = atomicAdd(&a.pcnt[iregion], total);
                    }
                }
            }
            uint32_t own_key = 0, for_key = 0;
            const uint32_t qs = (uint32_t)(tile * 64) + ql;
            if (hit) own_key = e_pj | ((ql & 15u) << 26) | pendflag;
            if (fh) for_key = (uint32_t)s_operm[ql] | ((jg & 15u) << 26) | pendflag;
            const int own_p = __shfl(obase, g) + pin;
            const int for_p = __shfl(fbase, leader) + pre;
            [[maybe_unused]] uint32_t own_w = 0xffffffffu, for_w = 0xffffffffu;      // places of the two records (pending-pair items)
            if (hit) {
                if (own_p < a.qcap) {
                    const long long o = ((long long)(tile - a.blk_begin) * 4 + g) * a.qcap + own_p;
                    a.qkey[o] = own_key; a.qd2[o] = d2;
                    own_w = (uint32_t)own_p;
                } else pool_over = 1;
            }
            if (fh) {
                if (for_p < a.qcap) {
                    const long long o = (long long)fq * a.qcap + for_p;
                    a.qkey[o] = for_key; a.qd2[o] = d2;
                    for_w = (uint32_t)for_p;
                } else pool_over = 1;
            }
            {
                if (a.fb == 2 && ipm) {
                    // item: both cell-sorted positions (their quarters are the records' logs), ONE box the segment's box met (9 bits split
                    // over the two position words; 256 = every box), the places of the two records in their logs
                    const int base = __builtin_amdgcn_readfirstlane(ibase);
                    if (iem && own_w != 0xffffffffu) {
                        for (int u = 0; u < iunits; ++u) {
                            const int pos = base + iexcl + u;
                            if (pos < a.icap) {
                                const uint32_t kbx = pc_keep > 4u ? 256u : ((pk_keep >> (8 * u)) & 255u);        // (bit 8: every box)
                                a.pitems[(long long)iregion * a.icap + pos] =
                                    make_uint4(qs | ((kbx & 63u) << 26), jg | ((kbx >> 6) << 26), own_w, for_w);
                            } else {
                                *a.pend_over = 1;
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- expand up to 64 records (lane = record, newest first) into the survivor queue ------------------------------
    unsigned long long surv = 0;
    auto expand = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nrec = min(rcount, 64);
        uint32_t meta = 0;
        unsigned long long H = 0;
        if (lane < nrec) { meta = s_rm[rcount - 1 - lane]; H = s_rh[rcount - 1 - lane]; }
        const int cnt = (int)__popcll(H);
        int incl = cnt;                                   // inclusive wave scan on the DPP network (row shifts, then row broadcasts)
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);       // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);       // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);       // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);       // row_shr:8
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
        // as many of the newest records as fit the survivor queue (a record holds at most 64 survivors; the queue is below 64 here)
        const int room = MF_QSZ - qcount;
        const unsigned long long fits = __ballot(lane < nrec && incl <= room);
        const int nsub = (int)__popcll(fits);                                       // incl is monotone: the fitting lanes are 0 .. nsub - 1
        const int total = __builtin_amdgcn_readlane(incl, nsub - 1);
        if (lane >= nsub) H = 0;
        int o = qcount + incl - cnt;
        const uint32_t m6 = meta << 6;
        while (__ballot(H != 0)) {
            if (H != 0) {
                const int bpos = __ffsll((long long)H) - 1;
                s_qs[o] = m6 | (uint32_t)bpos;
                ++o;
                H &= H - 1;
            }
        }
        rcount = __builtin_amdgcn_readfirstlane(rcount - nsub);
        qcount = __builtin_amdgcn_readfirstlane(qcount + total);
        surv += (unsigned long long)total;
        while (qcount >= 64) drain(64);
    };

    // ---- main loop over the item's slice of the tile's chunk list ------------------------------------------------------
    unsigned long long tested = 0;
    // B fragments come through a buffer descriptor: per-lane byte offset fixed for the whole kernel, chunk offset scalar
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.ops), 0, (int)(a.npad * (K8 ? 16 : 32)), 0x00020000);
    const int voff = K8 ? lane * 16 : col * 32 + kb * 16;
    [[maybe_unused]] auto load_b = [&](uint32_t c, u32x4 (&bq)[K8 ? 1 : 2]) {
        if constexpr (K8) {
            bq[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)(c * 1024u), 0);
        } else {
            bq[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)(c * 2048u), 0);
            bq[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024, (int)(c * 2048u), 0);
        }
    };
    // one chunk: 4 independent MFMAs back to back (2 query row blocks x 2 candidate column blocks), sign extraction, record push
    // (the ring slot is refilled -- chunk cn -- as soon as the MFMAs have read it: no register copies, the load is in flight during
    // this chunk's extraction and the following PF - 1 chunks)
    [[maybe_unused]] auto process = [&](uint32_t c, u32x4 (&bq)[K8 ? 1 : 2], bool refill, uint32_t cn) {
        tested += 64ull * 64ull;
        if (MF_ABLATE & 4) { asm volatile("" :: "v"(bq[0].x)); if (refill) load_b(cn, bq); return; }
        f32x16 acc0, acc1, acc2, acc3;
        if constexpr (K8) {
            union { u32x2 u; half4 h; } bf0, bf1;
            bf0.u = u32x2{bq[0].x, bq[0].y}; bf1.u = u32x2{bq[0].z, bq[0].w};
            acc0 = __builtin_amdgcn_mfma_f32_32x32x8f16(aF4[0], bf0.h, cinit[0], 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x8f16(aF4[1], bf0.h, cinit[1], 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x8f16(aF4[0], bf1.h, cinit[0], 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x8f16(aF4[1], bf1.h, cinit[1], 0, 0, 0);
        } else {
            union { u32x4 u; half8 h; } bf0, bf1;
            bf0.u = bq[0]; bf1.u = bq[1];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[0], bf0.h, zero16, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[1], bf0.h, zero16, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[0], bf1.h, zero16, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[1], bf1.h, zero16, 0, 0, 0);
        }
        if (refill) load_b(cn, bq);
        // H: 16 sign bits per 32x32 tile t = cbk*2 + rb at bits [16t, 16t+16); bit (15 - r) <-> accumulator register r.
        uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            h0 = __builtin_amdgcn_alignbit(h0, __float_as_uint(acc0[r]), 31);
            h1 = __builtin_amdgcn_alignbit(h1, __float_as_uint(acc1[r]), 31);
            h2 = __builtin_amdgcn_alignbit(h2, __float_as_uint(acc2[r]), 31);
            h3 = __builtin_amdgcn_alignbit(h3, __float_as_uint(acc3[r]), 31);
        }
        unsigned long long H = (unsigned long long)(h0 | (h1 << 16)) | ((unsigned long long)(h2 | (h3 << 16)) << 32);
        if constexpr (MODE == 2) { if (c == (uint32_t)tile) H &= own_keep; }
        if (MF_ABLATE & 1) { asm volatile("" :: "v"(H)); return; }
        const unsigned long long m = __ballot(H != 0);
        if (m) {
            while (rcount > MF_RCAP - 64) expand();
            if (H != 0) {
                const int pos = rcount + (int)__popcll(m & ((1ull << lane) - 1ull));
                s_rm[pos] = (c << 6) | (uint32_t)lane;
                s_rh[pos] = H;
            }
            rcount = __builtin_amdgcn_readfirstlane(rcount + (int)__popcll(m));
        }
    };

    // ---- VF: one chunk on the vector ALUs: this lane's two candidates (columns col and 32 + col of the chunk) against its 2 x 16 query
    // rows, canonical fp64 d2 <= r2, bits laid out like the accumulators' signs (bit 16 t + 15 - r: t = candidate half * 2 + row block)
    [[maybe_unused]] auto load_c = [&](uint32_t c, double (&cb)[2 * D]) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            cb[i] = a.Xs[((int64_t)c * 64 + col) * D + i];
            cb[D + i] = a.Xs[((int64_t)c * 64 + 32 + col) * D + i];
        }
    };
    [[maybe_unused]] auto process_vf = [&](uint32_t c, double (&cb)[2 * D], bool refill, uint32_t cn) {
        tested += 64ull * 64ull;
        double c0[D], c1[D];
#pragma unroll
        for (int i = 0; i < D; ++i) { c0[i] = cb[i]; c1[i] = cb[D + i]; }
        if (refill) load_c(cn, cb);
        unsigned long long H = 0;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            uint32_t h0 = 0, h1 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                double d20 = 0.0, d21 = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double qi = s_q[row * D + i];
                    const double t0 = qi - c0[i], t1 = qi - c1[i];
                    const double tt0 = t0 * t0, tt1 = t1 * t1;
                    d20 = (i == 0) ? tt0 : d20 + tt0;
                    d21 = (i == 0) ? tt1 : d21 + tt1;
                }
                h0 |= (d20 <= a.r2) ? (1u << (15 - r)) : 0u;
                h1 |= (d21 <= a.r2) ? (1u << (15 - r)) : 0u;
            }
            H |= ((unsigned long long)h0 << (16 * rb)) | ((unsigned long long)h1 << (16 * (2 + rb)));
        }
        if constexpr (MODE == 2) { if (c == (uint32_t)tile) H &= own_keep; }
        const unsigned long long m = __ballot(H != 0);
        if (m) {
            while (rcount > MF_RCAP - 64) expand();
            if (H != 0) {
                const int pos = rcount + (int)__popcll(m & ((1ull << lane) - 1ull));
                s_rm[pos] = (c << 6) | (uint32_t)lane;
                s_rh[pos] = H;
            }
            rcount = __builtin_amdgcn_readfirstlane(rcount + (int)__popcll(m));
        }
    };

    // ---- this item's slice of the tile's candidate chunk list (built once per tile by k_chunk_lists) ------------------
    {
        const int64_t tl = tile - a.blk_begin;
        const int64_t len = a.list_len[tl];
        // slices interleave the list: round j (entries [j*S, j*S + S)) gives slice s the entry at offset (s + h(j)) mod S,
        // h a multiplicative hash of j.  Every slice sees the same near/far mix of chunks, and -- unlike a plain k mod S --
        // a periodic structure in the list (rows of the cell grid are ~3 chunks each) cannot line its hits up in one
        // slice (seen: 90 of a column's 155 hits in one of 16 slices, overflowing that slot list on every build).
        const int S = Scur;
        const int64_t full = len / S;
        const int rem = (int)(len - full * S);
        auto off = [&](int64_t j) -> int { return (int)(((uint32_t)slice + (((uint32_t)j * 2654435761u) >> 24)) % (uint32_t)S); };
        const int cnt = (int)(full + ((rem > 0 && off(full) < rem) ? 1 : 0));
        const uint32_t* __restrict__ lst = a.lists + tl * a.list_cap;
        // the list is held 64 entries at a time in one VGPR (lane e = entry blk*64 + e), the following block is requested a block ahead
        auto load_blk = [&](int blk) -> uint32_t {
            const int e = blk * 64 + lane;
            return (e < cnt) ? lst[(int64_t)e * S + off(e)] : 0u;
        };
        // B fragments are prefetched PF chunks ahead (a ring of PF register sets, statically indexed): the stream is
        // latency x concurrency bound (L2 / MALL round trips), so more loads in flight per wavefront = more bandwidth.
        constexpr int PF = (K8 && !W4) ? 4 : 2;
        uint32_t lv = load_blk(0), lvn = load_blk(1);
        [[maybe_unused]] u32x4 ring[PF][K8 ? 1 : 2];
        [[maybe_unused]] double cring[VF ? PF : 1][2 * D];
        if (!(MF_ABLATE & 8)) {
#pragma unroll
            for (int u = 0; u < PF; ++u) if (u < cnt) {
                if constexpr (VF) load_c((uint32_t)__builtin_amdgcn_readlane((int)lv, u), cring[u]);
                else load_b((uint32_t)__builtin_amdgcn_readlane((int)lv, u), ring[u]);
            }
            for (int k = 0; k < cnt; k += PF) {
                if ((k & 63) == 0 && k > 0) { lv = lvn; lvn = load_blk((k >> 6) + 1); }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int kk = k + u;
                    if (kk < cnt) {
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)lv, kk & 63);
                        const bool refill = kk + PF < cnt;
                        const int kn = (kk + PF) & 63;
                        const uint32_t cn = (uint32_t)((kn < PF) ? __builtin_amdgcn_readlane((int)lvn, kn) : __builtin_amdgcn_readlane((int)lv, kn));
                        if constexpr (VF) process_vf(c, cring[u], refill, cn);
                        else process(c, ring[u], refill, cn);
                    }
                }
            }
        }
    }
    while (rcount > 0) expand();
    while (qcount > 0) drain(min(qcount, 64));

    if (MODE == 2) {
        if (pool_over) *a.pool_flag = 1;                          // a quarter log overflowed: the build is redone in the two-pass form
    }
    if constexpr (MODE == 3) {
        a.slice_cnt[(int64_t)slice * a.npad + qpos] = s_cnt[lane];
        a.st_best[(int64_t)slice * a.npad + qpos] = s_best[lane];
        a.st_besti[(int64_t)slice * a.npad + qpos] = s_besti[lane];
        a.st_nfree[(int64_t)slice * a.npad + qpos] = s_nfree[lane];
    }
    if (MODE != 1) {
        if (MODE == 0) a.slice_cnt[(int64_t)slice * a.npad + qpos] = s_cnt[lane];
        // per-XCD-sharded counters: a single hot address saturates at ~88 atomics/us (156k items would cost 1.8 ms)
        if (lane == 0 && a.pairs) { atomicAdd(a.pairs + 2 * (blockIdx.x & 255), tested); atomicAdd(a.pairs + 2 * (blockIdx.x & 255) + 1, surv); }
    }
}

template <int D, int MODE>
__global__ __launch_bounds__(64) void k_rdisc_mfma(mf_args a, mpfmt_grid G) { rdisc_mfma_body<D, MODE, false>(a, G); }
template <int D, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rdisc_mfma_w4(mf_args a, mpfmt_grid G)
{
    rdisc_mfma_body<D, MODE, true>(a, G);
}
// the single-pass pipeline with the exact fp64 filter on the vector ALUs (d <= 3: large low-dimensional worlds)
template <int D>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rdisc_vf_w4(mf_args a, mpfmt_grid G)
{
    rdisc_mfma_body<D, 2, true, true>(a, G);
}

// ---- host side ------------------------------------------------------------------------------------------------
// Slices per tile (work items = tiles x slices, one wavefront each).  The count is ODD: slice 0 of a tile holds the tile's own chunk --
// the block with by far the most hits, ~1.7x the drain work of the other slices -- and single-wavefront workgroups go to the SIMDs of a
// CU in turn, so with an even count the heavy items pile up on the same SIMDs (S = 4, 8: every heavy item on one SIMD in four -- pair
// kernel 2.8-2.9 ms at the north star against 2.02 at S = 3; S = 2, 6: on two -- 2.2-2.3 ms).  With an odd count the heavy items sit
// S apart and visit every residue modulo a power of two equally.  (tools/opt_sweep.sh mf_target_items; LABNOTES round 4.)
int mpfmt_slices_for(const mpfmt_ctx* ctx, int64_t units, bool mfma)
{
    if (units <= 0) return 1;
    // (the K = 16 form, d > 6, prefers more and shorter items: cfg3 55.0 vs 57.4 ms at 5 vs 3 slices)
    const int64_t target = mfma ? (ctx->d <= 6 ? ctx->mf_target_items : ctx->mf_target_items * 7 / 4) : 32768;
    int S = (int)std::min<int64_t>(MPFMT_MAXS, std::max<int64_t>(1, (target + units - 1) / units));
    if (mfma && S > 1 && (S & 1) == 0) S = (S + 1 <= MPFMT_MAXS) ? S + 1 : S - 1;
    return S;
}
int32_t mpfmt_mfma_prepare(mpfmt_ctx* ctx, double r, float* negT_out, bool* usable)
{
    // normalisation: one common scale so that every coordinate lies in [0,1]
    const int d = ctx->d;
    double ext = 0.0;
    for (int i = 0; i < d; ++i) ext = std::max(ext, ctx->bb_hi[i] - ctx->bb_lo[i]);
    *usable = false;
    // (chunk ids and cell-sorted positions travel in 20 / 26 bits of the queue entries and hit records)
    // (strictly below 2^26: position 2^26 - 1 with column 63 would make a record word of all ones, the ordering kernel's "no record" mark)
    if (d > 12 || !(ext > 0.0) || !(r > 0.0) || ctx->ntiles * 64 >= ((int64_t)1 << 26)) return MPFMT_OK;
    const double s = 1.0 / ext;
    const double e_c = 2.5e-4;                       // > 2^-12 (fp16 rounding on [0,1]) + fp32 conversion slack
    const double shell = 2.0 * std::sqrt((double)d) * e_c;
    const double Rh = s * r * (1.0 + 1e-9) + shell;
    // 2e-5 d: fp32 accumulation of <= 17 terms of magnitude <= 4d + the hi/lo split of the norms; 1.25e-4 > 2 * 2^-14: the
    // n_lo slots (and coordinates below 6.1e-5) are fp16 subnormals -- should the matrix core flush subnormal inputs, each of
    // the two norms of a pair loses at most 2^-14, so the bound holds whatever the flush behaviour is
    const double ftz = 1.25e-4;
    const double T = Rh * Rh * (1.0 + 1e-6) + 2e-5 * d + ftz;
    // the filter is only worth running when the shell is thin compared with the ball
    if (shell > 0.08 * s * r || ftz > 2.0 * (s * r) * (s * r)) return MPFMT_OK;
    *negT_out = -(float)(T * (1.0 + 1e-6));
    *usable = true;
    ctx->mf_scale = s;
    return MPFMT_OK;
}

int32_t mpfmt_mfma_build_operands(mpfmt_ctx* ctx)
{
    const int64_t npad = ctx->ntiles * 64;
    int32_t rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->ops, 32 * (size_t)npad))) return rc;
    if (npad == 0) return MPFMT_OK;
    const int B = 256;
    hipLaunchKernelGGL(k_make_ops, dim3((unsigned)((npad + B - 1) / B)), dim3(B), 0, ctx->stream,
                       ctx->Xs, ctx->N, npad, ctx->d, ctx->grid, ctx->mf_scale, ctx->ops);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// (re)build the per-tile candidate chunk lists of this ctx's shard for radius r; grows the list capacity until every
// list fits.  *usable = false when the lists would need more than 32 GB (caller then takes the exact VALU path).
int32_t mpfmt_mfma_build_lists(mpfmt_ctx* ctx, double r, bool* usable, bool spec, bool half)
{
    *usable = true;
    ctx->spec_lists = false;
    const int64_t nt = ctx->tile_end - ctx->tile_begin;
    if (nt <= 0) return MPFMT_OK;
    if (ctx->lists_r == r && ctx->lists_begin == ctx->tile_begin && ctx->lists_end == ctx->tile_end && ctx->lists && ctx->lists_half == half) return MPFMT_OK;
    if (ctx->lists_half != half) { ctx->lists_cap_trusted = -1; ctx->lists_half = half; }       // (a capacity learnt in the other form says nothing)
    int32_t rc;
    // (the north star's longest list is ~2 250 entries for most sample sets and 6 257 for one of five -- a tile of three sparse cells
    // across a row end: large index builds start with room for that; small shards keep the capacity their 4-wavefront list kernel stages)
    const bool small = (ctx->tile_end - ctx->tile_begin) < 16 * (int64_t)ctx->num_cus;
    int64_t cap = std::min<int64_t>(ctx->ntiles, std::max<int64_t>(ctx->list_cap, small ? 3072 : 8192));
    const double rpad = r * (1.0 + 1e-9) + 1e-300;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->list_len, sizeof(int32_t) * (size_t)(nt + 1)))) return rc;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if ((double)cap * (double)nt * 4.0 > 32e9) { *usable = false; return MPFMT_OK; }
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->lists, sizeof(uint32_t) * (size_t)cap * (size_t)nt))) return rc;
        // (the longest list's word lives in the index arena: zeroed by the index build's one fill, again here only when used since)
        if (!ctx->list_max_clean) HIPCHK(ctx, hipMemsetAsync(ctx->list_max, 0, sizeof(int32_t), ctx->stream));
        ctx->list_max_clean = false;
        const mpfmt_grid& G = ctx->grid;
        // few tiles (a small shard): four wavefronts per tile, kept ids staged in LDS (4 x cap x 4 bytes) -- in a global scratch area when
        // the lists are longer than LDS takes at a useful occupancy (a shard's lists hold every chunk of the OTHER shards: ~4 500 entries)
        const bool wide = ctx->lists_wide >= 0 ? ctx->lists_wide != 0 : nt < 16 * (int64_t)ctx->num_cus;
        const bool gst = wide && cap > 3072;
        if (gst && (rc = mpfmt_ensure(ctx, (void**)&ctx->lists_stage, sizeof(uint32_t) * (size_t)cap * 4 * (size_t)nt))) return rc;
        uint32_t* const gstage = gst ? (uint32_t*)ctx->lists_stage : nullptr;
#define CASE(DD) case DD: if (wide) hipLaunchKernelGGL((k_chunk_lists<DD, 4>), dim3((unsigned)nt), dim3(256), gst ? (size_t)0 : (size_t)cap * 16, ctx->stream, ctx->cellstart, \
            ctx->tile_lo, ctx->tile_hi, ctx->tile_sub, ctx->tile_sub32, G, rpad, ctx->tile_begin, nt, cap, (uint32_t*)ctx->lists, ctx->list_len, ctx->list_max, half ? 1 : 0, ctx->cellkey, ctx->cell_fb, gstage); \
        else hipLaunchKernelGGL((k_chunk_lists<DD, 1>), dim3((unsigned)nt), dim3(64), 0, ctx->stream, ctx->cellstart, \
            ctx->tile_lo, ctx->tile_hi, ctx->tile_sub, ctx->tile_sub32, G, rpad, ctx->tile_begin, nt, cap, (uint32_t*)ctx->lists, ctx->list_len, ctx->list_max, half ? 1 : 0, ctx->cellkey, ctx->cell_fb, (uint32_t*)nullptr); break;
        switch (ctx->d) {
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
            default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "MFMA r-disc path supports d <= 12 (got %d)", ctx->d);
        }
#undef CASE
        HIPCHK(ctx, hipGetLastError());
        if (spec && attempt == 0 && ctx->lists_cap_trusted == cap) {
            // same geometry as the build that established this capacity: the maximum is checked on the device
            // (k_spec_check) and by the host after the step's only synchronisation
            ctx->spec_lists = true;
            ctx->list_cap = cap; ctx->lists_r = r; ctx->lists_begin = ctx->tile_begin; ctx->lists_end = ctx->tile_end;
            return MPFMT_OK;
        }
        int32_t mx = 0;
        HIPCHK(ctx, hipMemcpyAsync(&mx, ctx->list_max, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (mx <= cap) {
            ctx->lists_cap_trusted = cap;
            ctx->list_cap = cap; ctx->lists_r = r; ctx->lists_begin = ctx->tile_begin; ctx->lists_end = ctx->tile_end;
            return MPFMT_OK;
        }
        cap = std::min<int64_t>(ctx->ntiles, ((int64_t)mx + mx / 8 + 64 + 255) / 256 * 256);       // (room for the longest list of the NEXT sample set too)
    }
    *usable = false;
    return MPFMT_OK;
}

// single pass: the degree of every column = a count over the keys of its quarter tile's log (one wavefront per quarter log), written
// straight to the two degree arrays the scans read (by original index and by cell-sorted position); the longest column (the ordering
// kernel stages whole columns) and the fullest log (the next build's capacity) go to two words behind the pair counters
// (a shard has too few logs to fill the chip; four wavefronts per log there measured slower -- 59 against 40 us at 8 ranks)
__global__ __launch_bounds__(256) void k_log_degrees(const uint32_t* __restrict__ qkey, const int32_t* __restrict__ qlen, long long qcap, int64_t nq,
                                                     int64_t pos0, const int32_t* __restrict__ perm, int64_t* __restrict__ deg,
                                                     int64_t* __restrict__ degs, int32_t* __restrict__ max_deg, int32_t* __restrict__ qmax,
                                                     int64_t N_tail, int64_t npad)
{
    __shared__ int s_c[4][16];
    __shared__ int s_m[4], s_q[4];
    // (unsharded: the scans' extra last elements are zeroed here instead of by two fill launches)
    if (N_tail >= 0 && blockIdx.x == 0 && threadIdx.x == 0) { deg[N_tail] = 0; degs[npad] = 0; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + wave;
    int kmax = 0, n = 0;
    if (q < nq) {
        if (lane < 16) s_c[wave][lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n = (int)min((long long)qlen[q], qcap);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(qkey + q * qcap);           // (qcap is a multiple of 4)
        // (four 16-byte loads in flight per lane: one wavefront per log with one load at a time was a chain of ~10 round trips, 84 % of
        // its cycles waiting)
        for (int b0 = 0; b0 < n; b0 += 4 * 256) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i0 = b0 + j * 256 + lane * 4;
                v[j] = (i0 < n) ? src[i0 >> 2] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i0 = b0 + j * 256 + lane * 4;
                const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) if (i0 + k < n) atomicAdd(&s_c[wave][(w[k] >> 26) & 15u], 1);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < 16) {
            const int k = s_c[wave][lane];
            const int64_t pos = pos0 + q * 16 + lane;
            const int32_t o = perm[pos];
            if (o >= 0) deg[o] = k;
            degs[pos] = k;                                     // (pad positions: 0 -- nothing else clears them on an unsharded ctx)
            kmax = k;
        }
    }
    for (int off = 8; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off));
    if (lane == 0) { s_m[wave] = kmax; s_q[wave] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (one candidate per workgroup, and it only goes to the atomic when it beats what is there: ~88 atomics / us on one address)
        const int m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3])), qm = max(max(s_q[0], s_q[1]), max(s_q[2], s_q[3]));
        if (m > *(volatile int32_t*)max_deg) atomicMax(max_deg, m);
        if (qm > *(volatile int32_t*)qmax) atomicMax(qmax, qm);
    }
}

int32_t mpfmt_launch_log_degrees(mpfmt_ctx* ctx)
{
    const int64_t nq = (ctx->tile_end - ctx->tile_begin) * 4;
    if (nq <= 0) return MPFMT_OK;
    const bool whole = !(ctx->world > 1);
    hipLaunchKernelGGL(k_log_degrees, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, ctx->stream, ctx->qkey, ctx->qlen, (long long)ctx->qcap, nq,
                       ctx->tile_begin * 64, ctx->perm, ctx->deg, ctx->degs, (int32_t*)(ctx->d_pairs + 512), (int32_t*)(ctx->d_pairs + 513),
                       whole ? ctx->N : (int64_t)-1, ctx->ntiles * 64);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

template <int DD, int MODE>
static void launch_pair_kernel(mpfmt_ctx* ctx, unsigned nblk, const mf_args& a, const mpfmt_grid& G)
{
    if constexpr (DD <= 6) hipLaunchKernelGGL((k_rdisc_mfma_w4<DD, MODE>), dim3(nblk), dim3(64), 0, ctx->stream, a, G);
    else hipLaunchKernelGGL((k_rdisc_mfma<DD, MODE>), dim3(nblk), dim3(64), 0, ctx->stream, a, G);
}

template <int MODE>
int32_t mpfmt_launch_rdisc_mfma(mpfmt_ctx* ctx, double r, float negT)
{
    mf_args a;
    a.ops = (const uint4*)ctx->ops; a.Xs = ctx->Xs; a.perm = ctx->perm; a.cellstart = ctx->cellstart;
    a.tile_lo = ctx->tile_lo; a.tile_hi = ctx->tile_hi;
    a.r2 = r * r; a.rpad = r * (1.0 + 1e-9) + 1e-300; a.negT = negT;
    a.S = ctx->S;
    a.blk_begin = ctx->tile_begin;                             // first tile of the shard
    a.nitems = (ctx->tile_end - ctx->tile_begin) * ctx->S;
    a.items_head = a.nitems; a.S_tail = ctx->S;
    if (MODE == 2 && ctx->mf_tail_slices > ctx->S && ctx->mf_tail_permille > 0 && a.nitems >= ctx->mf_tail_min_items) {
        // (the per-slice arrays of the other modes are laid out for S slices: only the single-pass build, which has none, cuts its tail finer)
        const int64_t nt_all = ctx->tile_end - ctx->tile_begin;
        const int64_t tail_tiles = std::min<int64_t>(nt_all, nt_all * ctx->mf_tail_permille / 1000);
        a.items_head = (nt_all - tail_tiles) * ctx->S;
        a.S_tail = ctx->mf_tail_slices | 1;
        a.nitems = a.items_head + tail_tiles * a.S_tail;
    }
    // (small groups spread a small launch evenly; large ones keep neighbouring tiles -- which read the same candidate chunks -- on one L2)
    a.xcd_mode = ctx->mf_xcd_mode >= 0 ? ctx->mf_xcd_mode : (a.nitems >= 32768 ? 256 : 64);
    a.npad = ctx->ntiles * 64; a.ntiles = ctx->ntiles;
    a.slice_cnt = ctx->slice_cnt; a.tptr = ctx->tptr; a.rowtmp = ctx->rowtmp; a.valtmp = ctx->valtmp;
    a.lists = (const uint32_t*)ctx->lists; a.list_len = ctx->list_len; a.list_cap = ctx->list_cap;
    a.pairs = (MODE == 1) ? nullptr : ctx->d_pairs;              // 256 x {tested, survivors} sharded counters
    a.survivors = nullptr;
    a.pool_flag = ctx->pool_flag; a.qcap = ctx->qcap;
    a.qkey = ctx->qkey; a.qd2 = ctx->qd2; a.qlen = ctx->qlen;
    a.half = (MODE == 2 && ctx->half_used) ? 1 : 0;
    a.ntiles_shard = ctx->tile_end - ctx->tile_begin;
    a.fb = (MODE == 2 && ctx->broad_in_drain) ? (ctx->bits_in_records ? 2 : 1) : 0; a.M = ctx->M; a.boxes = ctx->boxes;
    a.smask = (const unsigned long long*)ctx->smask;
    a.st_C = ctx->st_C; a.st_H = (const unsigned long long*)ctx->st_H; a.st_free = ctx->st_free;
    a.st_ss_has = (int32_t)(ctx->ss.has != 0); a.st_ss = ctx->rt_ss;
    a.st_best = (unsigned long long*)ctx->st_best; a.st_besti = ctx->st_besti; a.st_nfree = ctx->st_nfree;
    a.pitems = (uint4*)ctx->pair_items; a.pcnt = ctx->pair_cnt; a.icap = ctx->pair_icap; a.pend_over = ctx->pair_over;
    if (MODE != 2 && ctx->lists_half) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "two-pass r-disc kernels need whole chunk lists");
    if (a.nitems <= 0) return MPFMT_OK;
    const int64_t gran = NXCD * (int64_t)std::max(1, a.xcd_mode);
    const unsigned nblk = (unsigned)(((a.nitems + gran - 1) / gran) * gran);
    const mpfmt_grid& G = ctx->grid;
    if (ctx->filter_valu) {
        if constexpr (MODE == 2) {
            switch (ctx->d) {
                case 1: hipLaunchKernelGGL((k_rdisc_vf_w4<1>), dim3(nblk), dim3(64), 0, ctx->stream, a, G); break;
                case 2: hipLaunchKernelGGL((k_rdisc_vf_w4<2>), dim3(nblk), dim3(64), 0, ctx->stream, a, G); break;
                case 3: hipLaunchKernelGGL((k_rdisc_vf_w4<3>), dim3(nblk), dim3(64), 0, ctx->stream, a, G); break;
                default: return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the vector-ALU filter is built for d <= 3");
            }
            HIPCHK(ctx, hipGetLastError());
            return MPFMT_OK;
        }
        return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the vector-ALU filter exists for the single-pass build only");
    }
#define CASE(DD) case DD: launch_pair_kernel<DD, MODE>(ctx, nblk, a, G); break;
    switch (ctx->d) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
        default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "MFMA r-disc path supports d <= 12 (got %d)", ctx->d);
    }
#undef CASE
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

template int32_t mpfmt_launch_rdisc_mfma<0>(mpfmt_ctx*, double, float);
template int32_t mpfmt_launch_rdisc_mfma<1>(mpfmt_ctx*, double, float);
template int32_t mpfmt_launch_rdisc_mfma<2>(mpfmt_ctx*, double, float);
template int32_t mpfmt_launch_rdisc_mfma<3>(mpfmt_ctx*, double, float);

// ---- streaming mode: per-column reductions without a stored graph (mpfmt_rdisc_stream) -------------------------------------------------
// BASELINE configs[2] at the radius of fmt.jl:39 (R^12, N = 1e6, r = 0.625) has ~4 700 neighbours per sample: 57 GB of CSC.  What an
// FMT* expand step (fmt.jl:70-82) or a PRM*-style count needs of it per column x is a reduction: the degree, the best open parent
// argmin_y C[y] + d(y, x) (then ONE lazy edge test of that parent, fmt.jl:75), optionally the number of free edges.  The pair kernel
// in MODE 3 (filter on the matrix cores, exact fp64 refine, reductions in LDS per (tile, slice)) leaves per-slice partials; this
// kernel folds the slices and writes by sample index.
__global__ void k_stream_reduce(const int32_t* __restrict__ slice_cnt, const unsigned long long* __restrict__ st_best, const int32_t* __restrict__ st_besti,
                                const int32_t* __restrict__ st_nfree, int S, int64_t npad, int64_t pos_begin, int64_t pos_end,
                                const int32_t* __restrict__ perm, int64_t* __restrict__ deg, int64_t* __restrict__ nfree,
                                int64_t* __restrict__ parent, double* __restrict__ cost, unsigned long long* __restrict__ total)
{
    const int64_t s = pos_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long k = 0;
    if (s < pos_end) {
        const int32_t o = perm[s];
        if (o >= 0) {
            long long f = 0;
            unsigned long long b = ~0ull;
            int32_t bi = 0x7fffffff;
            for (int i = 0; i < S; ++i) {
                k += slice_cnt[(int64_t)i * npad + s];
                f += st_nfree[(int64_t)i * npad + s];
                const unsigned long long c = st_best[(int64_t)i * npad + s];
                const int32_t ci = st_besti[(int64_t)i * npad + s];
                if (c < b || (c == b && ci < bi)) { b = c; bi = ci; }
            }
            deg[o] = k; nfree[o] = f;
            parent[o] = (b != ~0ull) ? (int64_t)bi + 1 : 0;                      // 1-based sample index, 0 = no open neighbour
            cost[o] = (b != ~0ull) ? __longlong_as_double((long long)b) : (double)INFINITY;
        }
    }
    for (int off = 32; off > 0; off >>= 1) k += __shfl_xor(k, off);
    if ((threadIdx.x & 63) == 0 && k) atomicAdd(total, (unsigned long long)k);
}

int32_t mpfmt_rdisc_stream_impl(mpfmt_ctx* ctx, double r, const double* C_host, const uint64_t* H_host, int32_t want_free,
                                int64_t* deg, int64_t* nfree, int64_t* parent, double* cost, int64_t* nnz_out)
{
    int32_t rc;
    const int64_t N = ctx->N;
    if (ctx->world != 1) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "mpfmt_rdisc_stream needs an unsharded ctx");
    if (want_free && !(ctx->have_boxes && ctx->cc_kind == 0 && ctx->dw == ctx->d))
        return mpfmt_fail(ctx, MPFMT_ERR_STATE, "free-edge counts need PointRobotNDBoxes obstacles in the state space's own coordinates");
    if ((rc = mpfmt_build_grid(ctx, r))) return rc;
    ctx->tile_begin = 0; ctx->tile_end = ctx->ntiles;
    const int64_t nt = ctx->ntiles, npad = nt * 64;
    bool mf = false;
    float negT = 0.f;
    if ((rc = mpfmt_mfma_prepare(ctx, r, &negT, &mf))) return rc;
    if (!mf) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "mpfmt_rdisc_stream runs on the MFMA pair kernel: d <= 12 and a radius above the fp16 shell");
    if (ctx->ops_r != ctx->grid_r) {
        if ((rc = mpfmt_mfma_build_operands(ctx))) return rc;
        ctx->ops_r = ctx->grid_r; ctx->lists_r = -1.0;
    }
    bool ok = true;
    if ((rc = mpfmt_mfma_build_lists(ctx, r, &ok, false, false))) return rc;
    if (!ok) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "chunk lists exceed 32 GB");
    const int S = mpfmt_slices_for(ctx, nt, true);
    ctx->S = S;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->slice_cnt, sizeof(int32_t) * (size_t)S * std::max<int64_t>(npad, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->st_best, sizeof(uint64_t) * (size_t)S * std::max<int64_t>(npad, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->st_besti, sizeof(int32_t) * (size_t)S * std::max<int64_t>(npad, 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->st_nfree, sizeof(int32_t) * (size_t)S * std::max<int64_t>(npad, 1)))) return rc;
    // scratch: C[N], H words, deg[N], nfree[N], parent[N], cost[N], total
    const size_t w = (size_t)(N + 63) / 64;
    const size_t o_C = 0, o_H = o_C + 8 * (size_t)std::max<int64_t>(N, 1), o_deg = o_H + 8 * std::max<size_t>(w, 1), o_nf = o_deg + 8 * (size_t)std::max<int64_t>(N, 1),
                 o_par = o_nf + 8 * (size_t)std::max<int64_t>(N, 1), o_cost = o_par + 8 * (size_t)std::max<int64_t>(N, 1), o_tot = o_cost + 8 * (size_t)std::max<int64_t>(N, 1);
    void* scr;
    if ((rc = mpfmt_scratch(ctx, o_tot + 16, &scr))) return rc;
    char* sc = (char*)scr;
    if (C_host && N > 0) HIPCHK(ctx, hipMemcpyAsync(sc + o_C, C_host, 8 * (size_t)N, hipMemcpyHostToDevice, ctx->stream));
    if (H_host && N > 0) HIPCHK(ctx, hipMemcpyAsync(sc + o_H, H_host, 8 * w, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(sc + o_tot, 0, 16, ctx->stream));
    if (want_free && (rc = mpfmt_sweep_prepare_ss(ctx))) return rc;
    if (!ctx->d_pairs) HIPCHK(ctx, hipMalloc((void**)&ctx->d_pairs, 514 * sizeof(unsigned long long)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_pairs, 0, 512 * sizeof(unsigned long long), ctx->stream));
    ctx->st_C = C_host ? (const double*)(sc + o_C) : nullptr;
    ctx->st_H = (C_host && H_host) ? (const uint64_t*)(sc + o_H) : nullptr;
    ctx->st_free = want_free ? 1 : 0;
    ctx->half_used = false; ctx->broad_in_drain = false; ctx->bits_in_records = false;
    // the resident graph (if any) keeps its arrays, but the index now belongs to this radius
    if (ctx->graph_r != r) { ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false; ctx->graph_r = -1.0; }
    ctx->spec_ready = false;
    if (nt > 0) {
        mpfmt_timed tk(ctx);
        if ((rc = mpfmt_launch_rdisc_mfma<3>(ctx, r, negT))) return rc;
        tk.end("stream_kernel");
        const int B = 256;
        hipLaunchKernelGGL(k_stream_reduce, dim3((unsigned)((npad + B - 1) / B)), dim3(B), 0, ctx->stream, ctx->slice_cnt, (const unsigned long long*)ctx->st_best,
                           ctx->st_besti, ctx->st_nfree, S, npad, (int64_t)0, npad, ctx->perm, (int64_t*)(sc + o_deg), (int64_t*)(sc + o_nf), (int64_t*)(sc + o_par),
                           (double*)(sc + o_cost), (unsigned long long*)(sc + o_tot));
        HIPCHK(ctx, hipGetLastError());
    }
    unsigned long long tot = 0, pairs[512];
    if (N > 0) {
        if (deg) HIPCHK(ctx, hipMemcpyAsync(deg, sc + o_deg, 8 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
        if (nfree && want_free) HIPCHK(ctx, hipMemcpyAsync(nfree, sc + o_nf, 8 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
        if (parent && C_host) HIPCHK(ctx, hipMemcpyAsync(parent, sc + o_par, 8 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
        if (cost && C_host) HIPCHK(ctx, hipMemcpyAsync(cost, sc + o_cost, 8 * (size_t)N, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipMemcpyAsync(&tot, sc + o_tot, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(pairs, ctx->d_pairs, sizeof pairs, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 1; i < 256; ++i) { pairs[0] += pairs[2 * i]; pairs[1] += pairs[2 * i + 1]; }
    ctx->pairs_tested = (int64_t)pairs[0]; ctx->survivors = (int64_t)pairs[1];
    ctx->st_C = nullptr; ctx->st_H = nullptr; ctx->st_free = 0;
    if (nnz_out) *nnz_out = (int64_t)tot;
    return MPFMT_OK;
}
