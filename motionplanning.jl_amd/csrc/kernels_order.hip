// Column ordering of the single-pass r-disc build, with the collision sweep of the graph's edges fused into it (gfx950).
//
// The pair kernel (kernels_rdisc_mfma.hip) leaves every (tile, slice) item's exact hits in four append logs, one per 16 columns
// of the tile, in arrival order.  The reference's contract for a neighbourhood is ascending sample index (inball,
// src/nearneighbors.jl:179-183 -> the rows of a CSC column, src/nearneighbors.jl:23-28), so the hits have to be ordered -- and
// once a quarter tile's records sit in LDS grouped by column, both endpoints of every edge are one step away: the column
// states are 16 rows of the cell-sorted copy, the row states one L2-friendly gather.  k_order_logs<D, true> therefore also
// evaluates is_free_motion(V[row], V[col], CC, SS) for every entry (src/statespaces.jl:153-158: in_state_space of the first point, then
// the segment against every obstacle, src/collisioncheckers/boxesND.jl:26,44-56) and writes the free bit of the entry's final CSC
// position: the separate sweep kernel, its row-position array and its round table are not needed on that path.
#include "mpfmt_internal.h"
#include "sweep_predicates.h"
#include "sweep_cmpx.h"
#include <algorithm>

typedef const __attribute__((address_space(4))) double* ord_cptr;          // wave-uniform operands through the scalar cache
__device__ __forceinline__ ord_cptr ord_const(const double* p) { return (ord_cptr)(uintptr_t)p; }

// A counting sort on (column, bucket of the row id) through LDS, every phase one pass over the records with thread = record (no
// per-column serial chains), one workgroup per QUARTER TILE (16 columns) at a time:
//   1. headers: degree of each of the 16 columns (sum of the slice counts), prefix sums, output offsets (and the column states);
//   2. COUNT: histogram over (column, bucket): row ids are near-uniform over [0, N), so bucket = floor(id * 128 / N) -- monotone in
//      the id -- spreads a column's ~100 hits about one per bucket (non-returning LDS atomics);
//   3. segmented scan of the 16 x 128 counts (32 lanes per column on the DPP network) -> a cursor per (column, bucket);
//   4. PLACE: every record takes the next place of its (column, bucket) (returning LDS atomic) in the staging area: the quarter is
//      now grouped by column and ordered up to the arrival order inside a bucket;
//   5. WRITE: thread = staging position: the rank inside the bucket is a count over the bucket's other (typically 0-2) members,
//      the square root of d2 is taken, and rowval / nzval (/ rowpos) go out -- consecutive lanes write consecutive CSC entries.
//      Fused sweep: the row state is gathered from the cell-sorted copy, the obstacle set has been culled per column (wave-level,
//      lane = box), the broad phase of a surviving box is 2 d v_cmpx with the box as scalar operands (sweep_cmpx.h), pending exact
//      tests go through a per-wave LDS queue and run 64 wide; blocked bits are cleared in an LDS bitmap indexed by staging rank,
//      which is then copied -- shifted to the column's CSC bit offset -- into the global mask (atomicAnd: edge words are shared
//      with the neighbouring columns of other workgroups; the mask is preset to ones).
// The records are requested a quarter ahead into registers (ORD_PRE per thread); what a dense quarter holds beyond that is
// streamed from the logs in phases 2 and 4.  The staging area holds ORD_STG records; a quarter with more hits is done in several
// column ranges.  Columns longer than ORD_STG never come here: the host checks the maximum degree (k_degree) and takes the
// two-pass build.
#define ORD_ID(x) ((x) & 0x3fffffffu)      // a record's row index; bit 30 = the pair kernel's broad-phase flag (edge needs an exact test)
#define ORD_THREADS 512
#define ORD_WAVES 8
#define ORD_COLS 16              // columns per workgroup = columns per log
#define ORD_NB 128               // buckets per column
#define ORD_STG 2048             // staged records per workgroup (32 KB of LDS; with the counters 49.5 KB: THREE workgroups per CU -- 3072 allowed two: 1.18 -> 1.14 ms)
#define ORD_PRE 4                // records per thread requested ahead (4 x 512 = 2048 = the staging area)
#define ORD_QCAP 128             // pending exact tests per wavefront
#define ORD_MAXM 256             // obstacles the fused sweep handles (4 survivor words per column)
static_assert(ORD_STG >= MPFMT_ORD_MAXDEG, "a column the host lets through must fit the staging area");
static_assert(ORD_COLS * ORD_NB == ORD_THREADS * 4, "the segmented scan gives every thread four buckets");

template <int DX>
struct ord_hdr {
    int32_t k[ORD_COLS];         // column degrees
    int32_t cb[ORD_COLS + 4];    // exclusive prefix of the degrees (cb[ORD_COLS] = hits of the quarter)
    int32_t lp[MPFMT_MAXS + 4];  // exclusive prefix of the S log lengths: lp[sl] = records before log sl, lp[S] = records of the quarter
    long long out[ORD_COLS];     // colptr of each column
    int32_t ho[ORD_COLS];        // sample index of each column (pending-entry items)
    double xc[ORD_COLS][DX];     // column states (fused sweep)
};
template <int DX, bool SWEEP>
struct ord_shared {
    int32_t cnt[ORD_COLS][ORD_NB];   // records per (column, bucket)
    int32_t cur[ORD_COLS][ORD_NB];   // next staging position of each (column, bucket); after the placement: the bucket's end
    ord_hdr<DX> h[2];            // headers of the quarter in work and of the next one (prefetched)
    int32_t g1, pcount, pad_[2];     // pcount: pending-entry items this workgroup has appended
    unsigned long long cm[SWEEP ? ORD_COLS : 1][ORD_MAXM / 64];     // obstacles that survive each column's cull
    uint32_t bits[ORD_STG / 32];                                     // free bit of every staged entry, by rank position (fused sweep; blocked bits out of the records)
    uint32_t qa[SWEEP ? ORD_WAVES : 1][SWEEP ? ORD_QCAP : 1];        // pending exact tests: rank position | box << 12 | column << 20
    uint32_t qb[SWEEP ? ORD_WAVES : 1][SWEEP ? ORD_QCAP : 1];        //                      cell-sorted position of the row
};

struct ord_sweep {
    const double* Xs;            // [npad][D] cell-sorted states
    const double* boxes;         // [M][2][D]
    int32_t M;
    int32_t ss_has;              // test in_state_space of the row state (statespaces.jl:155)
    const double* ss_bounds;     // lo[MPFMT_MAX_DIM], hi[MPFMT_MAX_DIM]
    double rpad;                 // conservative radius of the column cull
    unsigned long long* mask;    // free bit per CSC entry, preset to ones
    const int64_t* nnz_dev;      // colptr + N: entries of the graph (padding bits of the last word are cleared)
};

__device__ __forceinline__ void ord_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// workgroup barrier that orders LDS traffic only: __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0) --
// every barrier would wait for the global loads requested ahead for the NEXT quarter and for the stores of this one
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Persistent workgroups, software pipelined: a quarter on its own is a chain of dependent round trips (perm -> colptr, counts ->
// log lengths -> records -> LDS -> stores), and with 2 workgroups per CU nothing covers them (first version: 75 % of the wave
// cycles waiting, 2.3 TB/s).  So while a workgroup writes quarter q out of LDS, the records of its next quarter are already on
// their way into registers, and the header of that quarter was requested a phase earlier still.
template <int D, bool SWEEP>
__global__ __launch_bounds__(ORD_THREADS) void k_order_logs(const mpfmt_hit* __restrict__ logs, int64_t capL, int S,
                                                            const int32_t* __restrict__ slice_cnt, int64_t npad,
                                                            const int32_t* __restrict__ log_len, int64_t tile_begin, int64_t tile_end,
                                                            const int64_t* __restrict__ colptr, const int32_t* __restrict__ perm,
                                                            int32_t* __restrict__ rowval, double* __restrict__ nzval, int32_t* __restrict__ rowpos,
                                                            uint32_t bucket_mul, const int32_t* __restrict__ spec_fail, ord_sweep sw,
                                                            const mpfmt_hit* __restrict__ flogs, int64_t fcapL, const int32_t* __restrict__ flen,
                                                            uint4* __restrict__ pend_items, int64_t pend_wcap, int32_t* __restrict__ pend_cnt,
                                                            int32_t* __restrict__ pend_over, int rec_bits)
{
    // rec_bits: k_exact_pairs has marked the records of blocked edges (bit 31 of the row index); their entries' bits are cleared in
    // the mask (sw.mask, preset to ones) the way the fused sweep's are: an LDS bitmap by rank position, copied to CSC positions
    if (spec_fail && *spec_fail) return;                     // speculative step whose capacities did not hold: redone by the host
    // half build (flogs != nullptr): every pair was found once, by the tile of its lower cell-sorted end, which also wrote the
    // record of the OTHER column into that column's tile's FOREIGN log (one per quarter tile, appended to by many tiles; its
    // per-column counts are row S of slice_cnt).  Here it is simply one more log of the quarter: source S of SS = S + 1.
    const int SS = S + (flogs ? 1 : 0);
    constexpr int DX = SWEEP ? D : 1;
    typedef ord_hdr<DX> hdr_t;
    typedef ord_shared<DX, SWEEP> shared_t;
    extern __shared__ __attribute__((aligned(16))) char ord_smem[];
    uint4* const stage = reinterpret_cast<uint4*>(ord_smem);
    shared_t& sh = *reinterpret_cast<shared_t*>(ord_smem + ORD_STG * 16);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nq = (tile_end - tile_begin) * 4;
    // bucket_mul = floor(2^32 * 128 / N) (0: N <= 128, the id is its own bucket)
    auto bucket = [&](uint32_t id) -> int { return bucket_mul ? min(ORD_NB - 1, (int)__umulhi(id, bucket_mul)) : (int)(id & (ORD_NB - 1)); };
    if ((SWEEP || rec_bits) && blockIdx.x == 0 && tid == 0) {              // padding bits of the mask's last word are zero
        const int64_t nnz = *sw.nnz_dev;
        if (nnz & 63) atomicAnd(&sw.mask[nnz >> 6], (1ull << (nnz & 63)) - 1ull);
    }

    // ---- pipeline pieces ----
    int hk = 0, ho = -1, hln = 0;                            // header values of the upcoming quarter, in the registers of the threads that fetch them
    long long hout = 0;
    double hx[DX];
    auto hdr_fetch1 = [&](int64_t qi) {                      // perm, slice counts, column state (threads < 16), log lengths (threads 64 .. 64 + S)
        hk = 0; ho = -1; hln = 0;
#pragma unroll
        for (int i = 0; i < DX; ++i) hx[i] = 0.0;
        if (qi >= nq) return;
        const int64_t tl = qi >> 2; const int quarter = (int)(qi & 3);
        if (tid < ORD_COLS) {
            const int64_t sp = (tile_begin + tl) * 64 + quarter * ORD_COLS + tid;
            ho = perm[sp];
            for (int sl = 0; sl < SS; ++sl) hk += slice_cnt[(int64_t)sl * npad + sp];
            if (SWEEP) {
#pragma unroll
                for (int i = 0; i < DX; ++i) hx[i] = sw.Xs[sp * D + i];
            }
        } else if (tid >= 64 && tid < 64 + S) {
            hln = log_len[(tl * S + (tid - 64)) * 4 + quarter];
        } else if (tid == 64 + S && flogs) {
            hln = min(flen[tl * 4 + quarter], (int32_t)fcapL);
        }
    };
    auto hdr_fetch2 = [&]() { hout = (tid < ORD_COLS && ho >= 0) ? colptr[ho] : 0; };
    auto hdr_publish = [&](int hb) {                         // degrees, their prefix sums, the log lengths (the output offsets follow)
        hdr_t& H = sh.h[hb];
        if (tid < 64) {
            const int k = (tid < ORD_COLS && ho >= 0) ? hk : 0;
            int inc = k;                                      // (lanes >= ORD_COLS carry zeros: one 16-lane row scan is enough)
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);       // row_shr:1
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);       // row_shr:2
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);       // row_shr:4
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);       // row_shr:8
            if (tid < ORD_COLS) {
                H.k[tid] = k; H.cb[tid + 1] = inc; H.ho[tid] = ho;
                if (SWEEP) {
#pragma unroll
                    for (int i = 0; i < DX; ++i) H.xc[tid][i] = hx[i];
                }
            }
            if (tid == 0) H.cb[0] = 0;
        } else if (tid < 128) {
            // (second wavefront, lanes 0 .. SS - 1 hold the log lengths, SS <= 16: one 16-lane row scan)
            int inc = (tid - 64 < SS) ? hln : 0;
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);       // row_shr:1
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);       // row_shr:2
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);       // row_shr:4
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);       // row_shr:8
            if (tid - 64 < SS) H.lp[tid - 64 + 1] = inc;
            if (tid == 64) H.lp[0] = 0;
        }
    };
    // The quarter's S logs are read as ONE sequence of lp[S] records: record i lives in the log sl with lp[sl] <= i < lp[sl + 1] (a
    // binary search over <= 17 prefix sums in LDS).  Dealing registers out per (log, run of records) instead left most lanes
    // idle on small shards, where a tile has 16 slices with ~100 records each.
    uint4 pre[ORD_PRE];
    auto rec_ptr = [&](const hdr_t& H, int64_t qi, int i) -> const uint4* {
        int lo = 0, base = 0;
        if (SS <= 5) {
            // (uniform) few logs -- the usual case: the prefix sums are wave-uniform LDS reads kept in scalar registers, the log
            // is a count of comparisons
            const int p1 = __builtin_amdgcn_readfirstlane(H.lp[1]), p2 = __builtin_amdgcn_readfirstlane(H.lp[SS > 2 ? 2 : SS]),
                      p3 = __builtin_amdgcn_readfirstlane(H.lp[SS > 3 ? 3 : SS]), p4 = __builtin_amdgcn_readfirstlane(H.lp[SS > 4 ? 4 : SS]);
            const int g1 = (SS > 1) & (i >= p1), g2 = (SS > 2) & (i >= p2), g3 = (SS > 3) & (i >= p3), g4 = (SS > 4) & (i >= p4);
            lo = g1 + g2 + g3 + g4;
            base = g4 ? p4 : g3 ? p3 : g2 ? p2 : g1 ? p1 : 0;
        } else {
            int hi = SS;                                     // lp[lo] <= i < lp[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (H.lp[mid] <= i) lo = mid; else hi = mid; }
            base = H.lp[lo];
        }
        const mpfmt_hit* const src = (lo == S) ? flogs + qi * fcapL
                                               : logs + (((qi >> 2) * S + lo) * 4 + (qi & 3)) * capL;
        return reinterpret_cast<const uint4*>(src) + (i - base);
    };
    auto rec_fetch = [&](const hdr_t& H, int64_t qi) {
        const int total = (qi < nq) ? H.lp[SS] : 0;
#pragma unroll
        for (int u = 0; u < ORD_PRE; ++u) {
            const int i = u * ORD_THREADS + tid;
            pre[u] = make_uint4(0u, 0xffffffffu, 0u, 0u);
            if (u * ORD_THREADS < total) { if (i < total) pre[u] = *rec_ptr(H, qi, i); }
        }
    };

    // Entries whose record carries the pair kernel's broad-phase flag (bit 30 of the row index: the segment's box meets an
    // obstacle's) are listed for k_sweep_pending -- (entry, column sample, row position) -- in this workgroup's own segment of the
    // item array; every other entry is free and stays set in the preset mask.
    if (tid == 0) sh.pcount = 0;
    int64_t qi = blockIdx.x;
    int hb = 0;
    hdr_fetch1(qi);
    hdr_fetch2();
    hdr_publish(0);
    if (tid < ORD_COLS) sh.h[0].out[tid] = hout;
    lds_barrier();
    rec_fetch(sh.h[0], qi);
    for (; qi < nq; qi += gridDim.x, hb ^= 1) {
        const hdr_t& H = sh.h[hb];
        const int64_t qn = qi + gridDim.x;                    // the workgroup's next quarter
        const int c0 = (int)(qi & 3) * ORD_COLS;              // first column (of the tile) of this quarter
        hdr_fetch1(qn);                                       // in flight during the counting sort
        int g0 = 0;
        bool first = true;
        while (g0 < ORD_COLS) {
            // ---- the next column range [g0, g1) that fits the staging area (normally the whole quarter); counts to zero ----
            if (tid < 64) {
                const bool ok = tid >= g0 && tid < ORD_COLS && (H.cb[min(tid, ORD_COLS - 1) + 1] - H.cb[g0] <= ORD_STG);
                const unsigned long long m = __ballot(ok);
                if (tid == 0) sh.g1 = g0 + (int)__popcll(m);
            }
            *reinterpret_cast<int4*>(&sh.cnt[0][tid * 4]) = make_int4(0, 0, 0, 0);
            lds_barrier();
            int g1 = sh.g1;
            const bool skip = g1 == g0;                      // a column beyond the staging area (excluded by the host): left out
            if (skip) g1 = g0 + 1;
            const int gb = H.cb[g0];
            // every record of the quarter, phase by phase: the prefetched ones from registers, the rest (dense quarters, or a
            // further column range) straight from the logs
            auto for_records = [&](auto&& f) {
#pragma unroll
                for (int u = 0; u < ORD_PRE; ++u) if (pre[u].y != 0xffffffffu) f(pre[u]);
                const int total = H.lp[SS];
                for (int i = (first ? ORD_PRE * ORD_THREADS : 0) + tid; i < total; i += ORD_THREADS) f(*rec_ptr(H, qi, i));
            };
            // ---- COUNT ----
            if (!skip) for_records([&](const uint4& r) {
                const int col = (int)(r.y >> 26) - c0;
                if (col >= g0 && col < g1) atomicAdd(&sh.cnt[col][bucket(ORD_ID(r.x))], 1);
            });
            lds_barrier();
            // ---- segmented scan: thread = four consecutive buckets, 32 threads per column ----
            {
                const int col = tid >> 5;
                const int4 c4 = *reinterpret_cast<const int4*>(&sh.cnt[0][tid * 4]);
                const int tot = c4.x + c4.y + c4.z + c4.w;
                int inc = tot;
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);       // row_shr:1
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);       // row_shr:2
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);       // row_shr:4
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);       // row_shr:8
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3: the scan of a 32-lane half
                const int b0 = H.cb[col] - gb + inc - tot;                               // staging position of the thread's first bucket
                *reinterpret_cast<int4*>(&sh.cur[0][tid * 4]) = make_int4(b0, b0 + c4.x, b0 + c4.x + c4.y, b0 + c4.x + c4.y + c4.z);
            }
            lds_barrier();
            // ---- PLACE ----
            if (!skip) for_records([&](const uint4& r) {
                const int col = (int)(r.y >> 26) - c0;
                if (col >= g0 && col < g1) {
                    const int pos = atomicAdd(&sh.cur[col][bucket(ORD_ID(r.x))], 1);
                    if (pos >= 0 && pos < ORD_STG) stage[pos] = r;
                }
            });
            if constexpr (SWEEP) {
                // the columns' obstacle culls (wavefront w: columns 2 w and 2 w + 1, lane = box): a box farther than r from the column
                // state in some axis cannot meet any of its edges; and every staged entry starts out free
                for (int c = g0 + 2 * wave; c < min(g1, g0 + 2 * wave + 2); ++c) {
                    double ulo[D], uhi[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) { const double wi = H.xc[c][i]; ulo[i] = wi - sw.rpad; uhi[i] = wi + sw.rpad; }
#pragma unroll
                    for (int w = 0; w < ORD_MAXM / 64; ++w) {
                        unsigned long long surv = 0;
                        if (w * 64 < sw.M) {
                            const int kb = w * 64 + lane;
                            int out = 0;
                            if (kb < sw.M) {
                                const double* bp = sw.boxes + (int64_t)kb * 2 * D;
#pragma unroll
                                for (int i = 0; i < D; ++i) out |= (int)(bp[D + i] < ulo[i]) | (int)(bp[i] > uhi[i]);
                            }
                            surv = __ballot(kb < sw.M && !out);
                        }
                        if (lane == 0) sh.cm[c][w] = surv;
                    }
                }
                if (tid < ORD_STG / 32) sh.bits[tid] = 0xffffffffu;
            } else {
                if (rec_bits && tid < ORD_STG / 32) sh.bits[tid] = 0xffffffffu;
            }
            lds_barrier();
            if (first) {
                // the next quarter: header to LDS, output offsets and records requested -- all in flight during the write-out below
                hdr_fetch2();
                hdr_publish(hb ^ 1);
#pragma unroll
                for (int u = 0; u < ORD_PRE; ++u) pre[u].y = 0xffffffffu;      // (consumed; a further column range streams the logs itself)
                first = false;
            }
            const bool last_range = g1 >= ORD_COLS;
            if (last_range) {
                lds_barrier();                                // (the next header's log lengths are read by every thread)
                rec_fetch(sh.h[hb ^ 1], qn);
            }
            // ---- WRITE: thread = staging position ----
            if (!skip) {
                const int nst = min(H.cb[g1] - gb, ORD_STG);
                [[maybe_unused]] int qcount = 0;              // pending exact tests of this wavefront (wave-uniform)
                [[maybe_unused]] uint32_t* const qa = sh.qa[SWEEP ? wave : 0];
                [[maybe_unused]] uint32_t* const qb = sh.qb[SWEEP ? wave : 0];
                // exact test of the last n (<= 64) queued items, lane = item (boxesND.jl:46-51); a blocked entry clears its bit
                [[maybe_unused]] auto drain = [&](int n) {
                    if constexpr (SWEEP) {
                        ord_wave_sync();
                        const bool on = lane < n;
                        const int qi_ = on ? qcount - n + lane : 0;
                        const uint32_t a = qa[qi_], jg = qb[qi_];
                        const int bitpos = (int)(a & 0xfffu), kbx = (int)((a >> 12) & 0xffu), cq = (int)((a >> 20) & 15u);
                        const bool all = on && ((a >> 24) & 1u);
                        double v[D], w[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) v[i] = sw.Xs[(int64_t)jg * D + i];
#pragma unroll
                        for (int i = 0; i < D; ++i) w[i] = H.xc[cq][i];
                        auto load_box_g = [&](int kb) {
                            box_regs<D> bx;
                            const double* bp = sw.boxes + (int64_t)kb * 2 * D;
#pragma unroll
                            for (int i = 0; i < D; ++i) { bx.lo[i] = bp[i]; bx.hi[i] = bp[D + i]; }
                            return bx;
                        };
                        bool free_ = all ? true : narrow_free_sl<D>(v, w, load_box_g(kbx));
                        if (__ballot(all)) {
                            // (rare) items that stand for every surviving box of their column: broad phase + exact test per box
                            if (all) {
                                double l[D], h[D];
                                seg_bbox<D>(v, w, l, h);
                                for (int q = 0; q < ORD_MAXM / 64; ++q) {
                                    unsigned long long m = sh.cm[cq][q];
                                    while (m) {
                                        const int kb = q * 64 + (__ffsll((long long)m) - 1);
                                        m &= m - 1;
                                        const box_regs<D> bx = load_box_g(kb);
                                        if (!broadphase_free_sl<D>(l, h, bx)) free_ = free_ && narrow_free_sl<D>(v, w, bx);
                                    }
                                }
                            }
                        }
                        if (on && !free_) atomicAnd(&sh.bits[bitpos >> 5], ~(1u << (bitpos & 31)));
                        qcount -= n;
                        ord_wave_sync();
                    }
                };
                for (int p0 = 0; p0 < nst; p0 += ORD_THREADS) {
                    const int p = p0 + tid;
                    const bool act = p < nst;
                    if (p0 + wave * 64 >= nst) break;                        // (no barrier inside: a wavefront without positions is done)
                    // (idle lanes of the last round carry the range's last column: the wave's column span below stays an interval)
                    const uint4 r = act ? stage[p] : make_uint4(0u, (uint32_t)(c0 + g1 - 1) << 26, 0u, 0u);
                    const int col = min(max((int)(r.y >> 26) - c0, 0), ORD_COLS - 1);
                    const int bk = bucket(ORD_ID(r.x));
                    const int e = sh.cur[col][bk], n = sh.cnt[col][bk];            // the bucket occupies [e - n, e)
                    int rk = e - n;
                    if (act) for (int m = e - n; m < e; ++m) rk += ((int32_t)ORD_ID(stage[m].x) < (int32_t)ORD_ID(r.x)) ? 1 : 0;
                    const int rel = rk - (H.cb[col] - gb);                          // rank inside the column
                    const bool valid = act && rel >= 0 && rel < H.k[col];          // (always, unless a log overflowed: that build is void, but stays in bounds)
                    const int64_t o = H.out[col] + rel;
                    if (valid) {
                        rowval[o] = (int32_t)ORD_ID(r.x);
                        nzval[o] = sqrt(__hiloint2double((int)r.w, (int)r.z));      // the log carries d2
                        if (rowpos) rowpos[o] = (int32_t)(r.y & 0x3ffffffu);
                    }
                    if constexpr (!SWEEP) { if (rec_bits && valid && (r.x >> 31)) atomicAnd(&sh.bits[rk >> 5], ~(1u << (rk & 31))); }
                    if (pend_items) {
                        const bool pd = valid && ((r.x >> 30) & 1u);
                        const unsigned long long pm = __ballot(pd);
                        if (pm) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(&sh.pcount, (int)__popcll(pm));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (pd) {
                                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                                if (pos < pend_wcap)
                                    pend_items[(int64_t)blockIdx.x * pend_wcap + pos] =
                                        make_uint4((uint32_t)(uint64_t)o, (uint32_t)((uint64_t)o >> 32), (uint32_t)H.ho[col], r.y & 0x3ffffffu);
                            }
                        }
                    }
                    if constexpr (SWEEP) {
                        const uint32_t jg = r.y & 0x3ffffffu;
                        double v[D], w[D];
#pragma unroll
                        for (int i = 0; i < D; ++i) v[i] = sw.Xs[(int64_t)jg * D + i];
#pragma unroll
                        for (int i = 0; i < D; ++i) w[i] = H.xc[col][i];
                        // survivors of the culls of the columns this wavefront's 64 positions belong to (consecutive positions: one or
                        // two columns at an FMT* degree)
                        const int cfirst = __builtin_amdgcn_readfirstlane(col);
                        const int clast = __builtin_amdgcn_readlane(col, 63);
                        unsigned long long smask[ORD_MAXM / 64];
#pragma unroll
                        for (int q = 0; q < ORD_MAXM / 64; ++q) smask[q] = 0;
                        for (int c = cfirst; c <= clast; ++c) {
#pragma unroll
                            for (int q = 0; q < ORD_MAXM / 64; ++q) {
                                const unsigned long long x = sh.cm[c][q];
                                smask[q] |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) |
                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
                            }
                        }
                        bool fr = valid;
                        if (sw.ss_has) {
                            const double* sq = sw.ss_bounds; asm volatile("" : "+s"(sq));
                            const ord_cptr sp = ord_const(sq);
                            int ok = 1;
#pragma unroll
                            for (int i = 0; i < D; ++i) ok &= (int)(sp[i] <= v[i]) & (int)(v[i] <= sp[MPFMT_MAX_DIM + i]);
                            fr = valid && ok != 0;
                        }
                        double l[D], h[D];
                        // map(min, v, w), map(max, v, w): only compared below, where a -0 / +0 difference to the reference's ternaries does not show
#pragma unroll
                        for (int i = 0; i < D; ++i) {
                            asm("v_min_f64 %0, %1, %2" : "=v"(l[i]) : "v"(w[i]), "v"(v[i]));
                            asm("v_max_f64 %0, %1, %2" : "=v"(h[i]) : "v"(w[i]), "v"(v[i]));
                        }
                        // boxes whose broad phase this lane failed: the last four as bytes of pk (box ids < 256), their number in pc
                        unsigned pk = 0, pc = 0;
                        if constexpr (D <= 6) l[0] = fr ? l[0] : (double)INFINITY;      // lanes that are out fail the first comparison
#pragma unroll
                        for (int q = 0; q < ORD_MAXM / 64; ++q) {
                            unsigned long long m = smask[q];
                            while (m) {
                                const int kb = q * 64 + (__ffsll((long long)m) - 1);
                                m &= m - 1;
                                box_regs<D> bx;                                       // wave-uniform box through the scalar cache
                                {
                                    const ord_cptr bp = ord_const(sw.boxes) + (int64_t)kb * 2 * D;
#pragma unroll
                                    for (int i = 0; i < D; ++i) { bx.lo[i] = bp[i]; bx.hi[i] = bp[D + i]; }
                                }
                                if constexpr (D <= 6) {
                                    sweep_cmpx<D>::note(bx.lo, bx.hi, l, h, pk, pc, kb);
                                } else {
                                    if (fr & !broadphase_free_sl<D>(l, h, bx)) { pk = (pk << 8) | (unsigned)kb; pc += 1; }
                                }
                            }
                        }
                        // (rare) a lane with more than four pending boxes queues ONE item that stands for "every surviving box of my
                        // column": the drain then walks the column's cull mask for it (no exact test -- and none of its registers -- here)
                        const bool o4 = pc > 4;
                        if (o4) { pk = 0; pc = 1; }
                        // queue the pending exact tests (the entry counts as free until a pass says otherwise)
#pragma unroll 1
                        for (int sl = 0; sl < 4; ++sl) {
                            const unsigned long long pm = __ballot(pc > (unsigned)sl);
                            if (!pm) break;
                            while (qcount > ORD_QCAP - 64) drain(min(qcount, 64));
                            if (pc > (unsigned)sl) {
                                const int pos = qcount + (int)__popcll(pm & ((1ull << lane) - 1ull));
                                qa[pos] = (uint32_t)rk | (((pk >> (8 * sl)) & 255u) << 12) | ((uint32_t)col << 20) | (o4 ? (1u << 24) : 0u);
                                qb[pos] = jg;
                            }
                            qcount += (int)__popcll(pm);
                        }
                        if (valid && !fr) atomicAnd(&sh.bits[rk >> 5], ~(1u << (rk & 31)));
                        while (qcount >= 64) drain(64);
                    }
                }
                if constexpr (SWEEP) { while (qcount > 0) drain(min(qcount, 64)); }
            }
            lds_barrier();
            if (SWEEP || rec_bits) {
                // ---- the free bits of the range's columns, from rank positions to CSC positions (wavefront w: columns 2 w, 2 w + 1) ----
                if (!skip) for (int c = g0 + 2 * wave; c < min(g1, g0 + 2 * wave + 2); ++c) {
                    const int k = H.k[c];
                    if (k == 0) continue;
                    const int sb = H.cb[c] - gb;
                    const int64_t o = H.out[c];
                    const int64_t w0 = o >> 6;
                    const int nw = (int)(((o + k - 1) >> 6) - w0) + 1;
                    for (int i = lane; i < nw; i += 64) {
                        const int64_t wd = w0 + i;
                        const int64_t lo_bit = max(wd * 64, o), hi_bit = min(wd * 64 + 64, o + k);
                        const int nb = (int)(hi_bit - lo_bit);                       // 1 .. 64 bits of this word belong to the column
                        const int src = sb + (int)(lo_bit - o);
                        const int wi = src >> 5, sh5 = src & 31;
                        const unsigned long long a0 = sh.bits[wi], a1 = sh.bits[min(wi + 1, ORD_STG / 32 - 1)], a2 = sh.bits[min(wi + 2, ORD_STG / 32 - 1)];
                        unsigned long long val = (a0 | (a1 << 32)) >> sh5;
                        if (sh5) val |= a2 << (64 - sh5);
                        const unsigned long long keep = (nb == 64) ? ~0ull : ((1ull << nb) - 1ull);
                        const int dsh = (int)(lo_bit & 63);
                        const unsigned long long rng = keep << dsh;
                        const unsigned long long w64 = ((val & keep) << dsh) | ~rng;
                        if (w64 != ~0ull) atomicAnd(&sw.mask[wd], w64);
                    }
                }
                lds_barrier();
            }
            g0 = g1;
        }
        if (tid < ORD_COLS) sh.h[hb ^ 1].out[tid] = hout;     // the next quarter's output offsets (requested before the write-out)
    }
    if (pend_items) {
        lds_barrier();
        if (tid == 0) {
            const int c = sh.pcount;
            pend_cnt[blockIdx.x] = (int32_t)min((int64_t)c, pend_wcap);
            if (c > pend_wcap) *pend_over = 1;                // a segment was too short: the host sweeps the whole graph instead
        }
    }
}

template <int D, bool SWEEP>
static int32_t launch_order(mpfmt_ctx* ctx, const int32_t* spec_fail, const ord_sweep& sw)
{
    const int64_t nt = ctx->tile_end - ctx->tile_begin;
    constexpr auto kk = k_order_logs<D, SWEEP>;
    const size_t lds = (size_t)ORD_STG * 16 + sizeof(ord_shared<SWEEP ? D : 1, SWEEP>);
    // (the attribute belongs to the kernel ON A DEVICE: set per launch -- a cached flag would cover the first device of a process
    // that drives several, and be written by concurrent ctx threads)
    HIPCHK(ctx, hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups: as many as fit the chip at once (3 per CU by their LDS and their 82 VGPRs), each takes every nb-th quarter tile
    if (ctx->ord_d != D) { ctx->ord_per_cu[0] = ctx->ord_per_cu[1] = 0; ctx->ord_d = D; }
    int& per_cu = ctx->ord_per_cu[SWEEP ? 1 : 0];
    if (per_cu == 0) {
        HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kk, ORD_THREADS, lds));
        per_cu = std::max(1, std::min(per_cu, 3));
    }
    const unsigned nb = (unsigned)std::min<int64_t>(nt * 4, (int64_t)ctx->num_cus * per_cu);
    // records flagged by the pair kernel's broad phase (step APIs, half build): their entries are listed for k_sweep_pending, one
    // segment of the item array per workgroup.  Workgroups take every nb-th quarter tile, so their shares are even: 1.5x the mean + 4096
    const bool recbits = !SWEEP && ctx->bits_in_records;      // form 2: the blocked edges are marked in the records, nothing to list
    const bool pend = !SWEEP && ctx->broad_in_drain && !recbits;
    if (pend) {
        int32_t rc;
        const int64_t entries = std::max<int64_t>(ctx->nnz, ctx->nnz_cap);
        ctx->pend_wcap = ctx->debug_small_lists ? 8 : entries * 3 / (2 * (int64_t)nb) + 4096;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->pend_items, sizeof(uint4) * (size_t)ctx->pend_wcap * nb))) return rc;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->pend_cnt, sizeof(int32_t) * (size_t)(nb + 1)))) return rc;
        ctx->pend_over = ctx->pend_cnt + nb;
        HIPCHK(ctx, hipMemsetAsync(ctx->pend_over, 0, sizeof(int32_t), ctx->stream));
    }
    hipLaunchKernelGGL(kk, dim3(nb), dim3(ORD_THREADS), lds, ctx->stream, ctx->pool, ctx->pool_cap, ctx->S,
                       ctx->slice_cnt, ctx->ntiles * 64, ctx->log_len, ctx->tile_begin, ctx->tile_end, ctx->colptr, ctx->perm,
                       ctx->rowval, ctx->nzval, (!SWEEP && ctx->sweep_sorted && !pend && !recbits) ? ctx->rowpos : nullptr,      // (the pending list carries its own row positions)
                       ctx->N > 128 ? (uint32_t)((128ull << 32) / (uint64_t)ctx->N) : 0u, spec_fail, sw,
                       ctx->half_used ? ctx->fpool : nullptr, ctx->fcap, ctx->flen,
                       pend ? (uint4*)ctx->pend_items : nullptr, ctx->pend_wcap, ctx->pend_cnt, ctx->pend_over, recbits ? 1 : 0);
    if (pend) ctx->pend_nseg = (int)nb;
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// can the step's sweep ride in the ordering kernel?  (PointRobotNDBoxes in the state space's own coordinates, d <= 8, <= 256 boxes)
bool mpfmt_order_can_fuse(const mpfmt_ctx* ctx)
{
    return ctx->fuse_sweep && ctx->cc_kind == 0 && ctx->have_boxes && ctx->dw == ctx->d && ctx->d <= 8 && ctx->M <= ORD_MAXM && ctx->Xs != nullptr;
}

// order the logs of the counted graph into the CSC; fuse = true: also sweep the edges (mask_entries as in mpfmt_launch_graph_sweep)
int32_t mpfmt_order_logs(mpfmt_ctx* ctx, const int32_t* spec_fail, bool fuse, int64_t mask_entries)
{
    const int64_t nt = ctx->tile_end - ctx->tile_begin;
    int32_t rc;
    ord_sweep sw{};
    if (fuse) {
        if (!mpfmt_order_can_fuse(ctx)) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "fused sweep requested where it does not apply");
        const int64_t words = (std::max<int64_t>(ctx->nnz, mask_entries) + 63) / 64;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1)))) return rc;
        if ((rc = mpfmt_sweep_prepare_ss(ctx))) return rc;
        // preset to ones (the sweep clears blocked entries); an empty graph keeps one zero word
        HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, ctx->nnz > 0 ? 0xFF : 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
        sw.Xs = ctx->Xs; sw.boxes = ctx->boxes; sw.M = ctx->M;
        sw.ss_has = (int)(ctx->ss.has && !ctx->ssflag_all_in); sw.ss_bounds = ctx->rt_ss;
        sw.rpad = ctx->graph_r * (1.0 + 1e-9) + 1e-300;
        sw.mask = (unsigned long long*)ctx->graph_free; sw.nnz_dev = ctx->colptr + ctx->N;
    }
    ctx->rowpos_valid = false;
    const bool recbits = !fuse && ctx->bits_in_records;
    if (recbits) {
        // the records carry the blocked bits (k_exact_pairs): this pass also writes the mask
        const int64_t words = (std::max<int64_t>(ctx->nnz, mask_entries) + 63) / 64;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1)))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, ctx->nnz > 0 ? 0xFF : 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
        sw.mask = (unsigned long long*)ctx->graph_free; sw.nnz_dev = ctx->colptr + ctx->N;
    }
    if (ctx->nnz == 0 || nt <= 0) { if (fuse || recbits) ctx->graph_swept = true; return MPFMT_OK; }
    if (!fuse) {
        // option sweep_sorted: also keep every row's cell-sorted position, so the sweep can gather from Xs (see kernels_sweep.hip)
        if (ctx->sweep_sorted && (rc = mpfmt_ensure(ctx, (void**)&ctx->rowpos, sizeof(int32_t) * (size_t)std::max<int64_t>(std::max(ctx->nnz, ctx->nnz_cap), 1)))) return rc;
        if ((rc = launch_order<1, false>(ctx, spec_fail, sw))) return rc;
        ctx->pend_valid = ctx->broad_in_drain && !recbits;        // ... or the flagged entries have been listed for k_sweep_pending
        ctx->rowpos_valid = ctx->sweep_sorted != 0 && !ctx->pend_valid && !recbits;   // every entry's row is also known by its cell-sorted position (the sweep gathers from Xs)
        if (recbits) { ctx->graph_swept = true; ctx->sweep_in_order = true; }
        return MPFMT_OK;
    }
    switch (ctx->d) {
        case 1: rc = launch_order<1, true>(ctx, spec_fail, sw); break;
        case 2: rc = launch_order<2, true>(ctx, spec_fail, sw); break;
        case 3: rc = launch_order<3, true>(ctx, spec_fail, sw); break;
        case 4: rc = launch_order<4, true>(ctx, spec_fail, sw); break;
        case 5: rc = launch_order<5, true>(ctx, spec_fail, sw); break;
        case 6: rc = launch_order<6, true>(ctx, spec_fail, sw); break;
        case 7: rc = launch_order<7, true>(ctx, spec_fail, sw); break;
        case 8: rc = launch_order<8, true>(ctx, spec_fail, sw); break;
        default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "fused sweep supports d <= 8 (got %d)", ctx->d);
    }
    if (rc) return rc;
    ctx->graph_swept = true;
    return MPFMT_OK;
}
