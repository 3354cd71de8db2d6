// Column ordering of the single-pass r-disc build (gfx950).
//
// The pair kernel (kernels_rdisc_mfma.hip) leaves the exact hits of every QUARTER TILE (16 consecutive cell-sorted columns) in one
// append log, in arrival order: a 4-byte key (row sample index | column within the quarter << 26 | flags) and the squared distance
// in two arrays.  The reference's contract for a neighbourhood is ascending sample index (inball, src/nearneighbors.jl:179-183 -> the
// rows of a CSC column, src/nearneighbors.jl:23-28), so every quarter is ordered here and written as 16 columns of the final CSC;
// the same pass takes the square roots (src/statespaces/geometric.jl:4), writes the free-edge mask from the marks k_exact_pairs left
// in the keys (edge-test form 2), or lists the entries the pair kernel's broad phase flagged for k_sweep_pending (form 1).
#include "mpfmt_internal.h"
#include <algorithm>

// A counting sort on (column, bucket of the row id) through LDS, every phase one pass over the records with thread = record (no
// per-column serial chains), one workgroup per QUARTER TILE at a time:
//   1. header: degree of each of the 16 columns (k_log_degrees / k_degree), prefix sums, output offsets, the log's length;
//   2. COUNT: histogram over (column, bucket): row ids are near-uniform over [0, N), so bucket = floor(id * 128 / N) -- monotone in
//      the id -- spreads a column's ~100 hits about one per bucket (non-returning LDS atomics);
//   3. segmented scan of the 16 x 128 counts (32 lanes per column on the DPP network) -> a cursor per (column, bucket);
//   4. PLACE: every record takes the next place of its (column, bucket) (returning LDS atomic) in the staging area (key and d2): the
//      quarter is now grouped by column and ordered up to the arrival order inside a bucket;
//   5. WRITE: thread = staging position: the rank inside the bucket is a count over the bucket's other (typically 0-2) members,
//      the square root of d2 is taken, and rowval / nzval (/ rowpos) go out -- consecutive lanes write consecutive CSC entries.
//      Blocked entries (bit 31 of the key) clear their bit in an LDS bitmap indexed by staging rank, which is then copied -- shifted to
//      the column's CSC bit offset -- into the global mask (atomicAnd: edge words are shared with the neighbouring columns of other
//      workgroups; the mask is preset to ones).
// The records are requested a quarter ahead into registers (ORD_PRE per thread = the staging area); what a dense quarter holds beyond
// that is streamed from the log in phases 2 and 4 and staged in several column ranges.  Columns longer than MPFMT_ORD_MAXDEG never
// come here: the host checks the maximum degree and takes the two-pass build.
#define ORD_ID(k) ((k) & 0x03ffffffu)        // a key's row index (26 bits)
#define ORD_COL(k) (((k) >> 26) & 15u)       // its column within the quarter
#define ORD_THREADS 512
#define ORD_WAVES 8
#define ORD_COLS 16              // columns per workgroup = columns per log
#define ORD_NB 128               // buckets per column
#define ORD_STG 2560             // staged records per workgroup: 12 bytes each (30 KB of LDS; with the 16-bit counters 39 KB -- FOUR workgroups per CU need <= 40 KB each)
#define ORD_PRE 5                // records per thread requested ahead (5 x 512 = the staging area)
#define ORD_WAVES_EU 8           // wavefronts per SIMD the kernel is built for (8: 64 VGPRs, 4 of them spilled -- four 512-thread workgroups per CU)
static_assert(ORD_STG >= MPFMT_ORD_MAXDEG, "a column the host lets through must fit the staging area");
static_assert(ORD_COLS * ORD_NB == ORD_THREADS * 4, "the segmented scan gives every thread four buckets");
static_assert(ORD_PRE * ORD_THREADS >= ORD_STG && ORD_STG % 32 == 0 && ORD_STG < 65536, "the prefetched records cover what the staging area holds; positions fit 16 bits");

struct ord_hdr {
    int32_t k[ORD_COLS];         // column degrees
    int32_t cb[ORD_COLS + 4];    // exclusive prefix of the degrees (cb[ORD_COLS] = hits of the quarter)
    long long out[ORD_COLS];     // colptr of each column
    int32_t ho[ORD_COLS];        // sample index of each column (pending-entry items)
    int32_t n, pad_[3];          // records in the quarter's log
};
struct ord_shared {
    // 16-bit counters, two per word (low half = the even bucket): a count or a staging position never reaches 65536, so the halves
    // never carry into each other and one 32-bit LDS atomic serves either
    uint32_t cnt[ORD_COLS][ORD_NB / 2];   // records per (column, bucket)
    uint32_t cur[ORD_COLS][ORD_NB / 2];   // next staging position of each (column, bucket); after the placement: the bucket's end
    ord_hdr h[2];                    // headers of the quarter in work and of the next one (prefetched)
    int32_t g1, pcount, nextk, pad_; // pcount: pending-entry items this workgroup has appended; nextk: the draw for the quarter after next
    uint32_t bits[ORD_STG / 32];     // free bit of every staged entry, by rank position (blocked bits out of the keys)
};

// workgroup barrier that orders LDS traffic only: __syncthreads() carries a workgroup-scope fence, i.e. s_waitcnt vmcnt(0) --
// every barrier would wait for the global loads requested ahead for the NEXT quarter and for the stores of this one
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct ord_args {
    const uint32_t* qkey;        // [quarters][qcap]
    const double* qd2;
    long long qcap;
    const int32_t* qlen;         // [quarters] cursors of the logs (may exceed qcap on a void build: clamped)
    const int64_t* degs;         // [npad] degree of every cell-sorted position
    int64_t tile_begin, tile_end;
    const int64_t* colptr;
    const int32_t* perm;
    const int32_t* iperm;        // sample index -> cell-sorted position (rowpos / pending items)
    int32_t* rowval;
    double* nzval;
    int32_t* rowpos;             // or nullptr
    uint32_t bucket_mul;         // floor(2^32 * 128 / N) (0: N <= 128, the id is its own bucket)
    const int32_t* spec_fail;
    unsigned long long* mask;    // rec_bits: free bit per CSC entry, preset to ones
    const int64_t* nnz_dev;      // colptr + N
    uint4* pend_items; int64_t pend_wcap; int32_t* pend_cnt; int32_t* pend_over;
    int64_t* deg_clear;          // sharded ctx: the degree array by sample index, whose entries of this shard are put back to zero here
    int rec_bits;                // k_exact_pairs has marked the keys of blocked edges (bit 31): their entries' bits are cleared in the mask
    int32_t* qctr;               // [8] (zeroed per launch) or nullptr: counters the workgroups draw their quarters from, one per XCD
};

// Which quarters a workgroup takes: its first two by its index (b, b + nb), every further one from a COUNTER (one per XCD -- a counter
// takes ~90 atomics per microsecond, the launch draws 62 500 quarters in 0.7 ms -- XCD x handing out the quarters = x mod 8 in order).
// With a fixed share (every nb-th quarter) the four workgroups of a CU finish 570 / 650 / 760 / 880 us after the launch although their
// shares are equal: the wave scheduler serves the oldest wavefronts first, so the CU's first workgroup runs ahead, and the last one ends
// alone on a CU whose other slots are idle (measured with the device clock at entry and exit of every workgroup).  Drawn from a counter,
// the quarters go to whoever is done.
// Persistent workgroups, software pipelined: a quarter on its own is a chain of dependent round trips (perm -> colptr, degrees ->
// log length -> records -> LDS -> stores), and with 3 workgroups per CU nothing covers them.  So while a workgroup writes quarter q out
// of LDS, the records of its next quarter are already on their way into registers, and the header of that quarter was requested a
// phase earlier still.
__global__ __launch_bounds__(ORD_THREADS) __attribute__((amdgpu_waves_per_eu(ORD_WAVES_EU, ORD_WAVES_EU))) void k_order_logs(ord_args a)
{
    if (a.spec_fail && *a.spec_fail) return;                 // speculative step whose capacities did not hold: redone by the host
    extern __shared__ __attribute__((aligned(16))) char ord_smem[];
    double* const stage_d2 = reinterpret_cast<double*>(ord_smem);
    uint32_t* const stage_key = reinterpret_cast<uint32_t*>(ord_smem + ORD_STG * 8);
    ord_shared& sh = *reinterpret_cast<ord_shared*>(ord_smem + ORD_STG * 12);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t nq = (a.tile_end - a.tile_begin) * 4;
    auto bucket = [&](uint32_t id) -> int { return a.bucket_mul ? min(ORD_NB - 1, (int)__umulhi(id, a.bucket_mul)) : (int)(id & (ORD_NB - 1)); };
    if (a.rec_bits && blockIdx.x == 0 && tid == 0) {         // padding bits of the mask's last word are zero
        const int64_t nnz = *a.nnz_dev;
        if (nnz & 63) atomicAnd(&a.mask[nnz >> 6], (1ull << (nnz & 63)) - 1ull);
    }

    // ---- pipeline pieces ----
    int hk = 0, ho = -1, hln = 0;                            // header values of the upcoming quarter, in the registers of the threads that fetch them
    long long hout = 0;
    auto hdr_fetch1 = [&](int64_t qi) {                      // perm, degree (threads < 16), log length (thread 64)
        hk = 0; ho = -1; hln = 0;
        if (qi >= nq) return;
        if (tid < ORD_COLS) {
            const int64_t sp = a.tile_begin * 64 + qi * ORD_COLS + tid;
            ho = a.perm[sp];
            hk = (int)a.degs[sp];
            if (a.deg_clear && ho >= 0) a.deg_clear[ho] = 0;      // (read by the colptr scan before this kernel; zero again for the next step)
        } else if (tid == 64) {
            hln = (int)min((long long)a.qlen[qi], a.qcap);
        }
    };
    auto hdr_fetch2 = [&]() { hout = (tid < ORD_COLS && ho >= 0) ? a.colptr[ho] : 0; };
    auto hdr_publish = [&](int hb) {                         // degrees, their prefix sums, the log's length (the output offsets follow)
        ord_hdr& H = sh.h[hb];
        if (tid < 64) {
            const int k = (tid < ORD_COLS && ho >= 0) ? hk : 0;
            int inc = k;                                      // (lanes >= ORD_COLS carry zeros: one 16-lane row scan is enough)
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);       // row_shr:1
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);       // row_shr:2
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);       // row_shr:4
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);       // row_shr:8
            if (tid < ORD_COLS) { H.k[tid] = k; H.cb[tid + 1] = inc; H.ho[tid] = ho; }
            if (tid == 0) H.cb[0] = 0;
        } else if (tid == 64) {
            H.n = hln;
        }
    };
    // keys and squared distances: requested a quarter ahead (during the write-out of the quarter before)
    uint32_t pk[ORD_PRE];
    double pd[ORD_PRE];
    // (buffer loads: one per-lane byte offset for all six, the slot's offset and the log's base scalar, reads past the log's end
    // return zero -- six flat loads keep six 64-bit addresses alive per array)
    auto rec_fetch_n = [&](int total, int64_t qi) {
        const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.qkey + (qi < nq ? qi : 0) * a.qcap), 0, total * 4, 0x00020000);
#pragma unroll
        for (int u = 0; u < ORD_PRE; ++u) pk[u] = __builtin_amdgcn_raw_buffer_load_b32(rk, tid * 4, u * ORD_THREADS * 4, 0);
    };
    auto rec_fetch = [&](const ord_hdr& H, int64_t qi) { rec_fetch_n((qi < nq) ? H.n : 0, qi); };
    typedef uint32_t ord_u32x2 __attribute__((ext_vector_type(2)));
    auto d2_fetch = [&](int total, int64_t qi) {
        const long long base = (qi < nq ? qi : 0) * a.qcap;
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.qd2 + base), 0, total * 8, 0x00020000);
#pragma unroll
        for (int u = 0; u < ORD_PRE; ++u) {
            const ord_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rd, tid * 8, u * ORD_THREADS * 8, 0);
            pd[u] = __hiloint2double((int)v.y, (int)v.x);
        }
    };

    if (tid == 0) sh.pcount = 0;
    int64_t qi = blockIdx.x;
    int64_t qn = qi + gridDim.x;                              // the quarter after qi (its header and records are requested while qi is in work)
    const int xcd = (int)(blockIdx.x & 7u);
    int hb = 0;
    hdr_fetch1(qi);
    hdr_fetch2();
    hdr_publish(0);
    if (tid < ORD_COLS) sh.h[0].out[tid] = hout;
    lds_barrier();
    rec_fetch(sh.h[0], qi);
    d2_fetch(sh.h[0].n, qi);
    for (; qi < nq; hb ^= 1) {
        const ord_hdr& H = sh.h[hb];
        const int total = H.n;
        const long long lbase = qi * a.qcap;
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
            *reinterpret_cast<uint2*>(&sh.cnt[0][tid * 2]) = make_uint2(0u, 0u);
            lds_barrier();
            int g1 = sh.g1;
            const bool skip = g1 == g0;                      // a column beyond the staging area (excluded by the host): left out
            if (skip) g1 = g0 + 1;
            const int gb = H.cb[g0];
            // every record of the quarter, phase by phase: the prefetched ones from registers (first range), the rest -- a quarter
            // denser than the prefetch, or a further column range -- straight from the log.  f(key, d2's register slot or -1, i)
            auto for_records = [&](auto&& f) {
                if (first) {
#pragma unroll
                    for (int u = 0; u < ORD_PRE; ++u) if (u * ORD_THREADS + tid < total) {
                        uint32_t key = pk[u];
                        asm volatile("" : "+v"(key));             // (column and bucket are worked out again in every phase: kept from COUNT to PLACE they cost ~5 registers per slot)
                        f(key, u, u * ORD_THREADS + tid);
                    }
                }
                for (int i = (first ? ORD_PRE * ORD_THREADS : 0) + tid; i < total; i += ORD_THREADS) f(a.qkey[lbase + i], -1, i);
            };
            // ---- COUNT ----
            if (!skip) for_records([&](uint32_t key, int, int) {
                const int col = (int)ORD_COL(key);
                if (col >= g0 && col < g1) { const int bk = bucket(ORD_ID(key)); atomicAdd(&sh.cnt[col][bk >> 1], 1u << ((bk & 1) * 16)); }
            });
            lds_barrier();
            // ---- segmented scan: thread = four consecutive buckets, 32 threads per column ----
            {
                const int col = tid >> 5;
                const uint2 cw = *reinterpret_cast<const uint2*>(&sh.cnt[0][tid * 2]);
                const int c0 = (int)(cw.x & 0xffffu), c1 = (int)(cw.x >> 16), c2 = (int)(cw.y & 0xffffu), c3 = (int)(cw.y >> 16);
                const int tot = c0 + c1 + c2 + c3;
                int inc = tot;
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);       // row_shr:1
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);       // row_shr:2
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);       // row_shr:4
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);       // row_shr:8
                inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3: the scan of a 32-lane half
                // staging position of the thread's first bucket (a void build -- an overflowed log -- may run past the area: clamped, the
                // 16-bit halves must not carry)
                const int b0 = min(max(H.cb[col] - gb + inc - tot, 0), ORD_STG);
                const int b1 = min(b0 + c0, ORD_STG), b2 = min(b1 + c1, ORD_STG), b3 = min(b2 + c2, ORD_STG);
                *reinterpret_cast<uint2*>(&sh.cur[0][tid * 2]) = make_uint2((uint32_t)b0 | ((uint32_t)b1 << 16), (uint32_t)b2 | ((uint32_t)b3 << 16));
            }
            lds_barrier();
            // ---- PLACE ----
            if (!skip) for_records([&](uint32_t key, int u, int i) {
                const int col = (int)ORD_COL(key);
                if (col >= g0 && col < g1) {
                    const int bk = bucket(ORD_ID(key));
                    const int pos = (int)((atomicAdd(&sh.cur[col][bk >> 1], 1u << ((bk & 1) * 16)) >> ((bk & 1) * 16)) & 0xffffu);
                    double d2 = 0.0;
                    if (u < 0) d2 = a.qd2[lbase + i];
#pragma unroll
                    for (int v = 0; v < ORD_PRE; ++v) if (u == v) d2 = pd[v];
                    if (pos >= 0 && pos < ORD_STG) { stage_key[pos] = key; stage_d2[pos] = d2; }
                }
            });
            if (a.rec_bits && tid < ORD_STG / 32) sh.bits[tid] = 0xffffffffu;
            lds_barrier();
            if (first) {
                // the next quarter: header to LDS, output offsets and records requested -- all in flight during the write-out below
                hdr_fetch2();
                hdr_publish(hb ^ 1);
                // ... and the draw for the one after it (read behind the barriers of the write-out)
                if (a.qctr && tid == 0) sh.nextk = atomicAdd(&a.qctr[xcd], 1);
                first = false;
            }
            const bool last_range = g1 >= ORD_COLS;
            if (last_range) {
                lds_barrier();                                // (the next header's log length is read by every thread)
                rec_fetch(sh.h[hb ^ 1], qn);
                d2_fetch(qn < nq ? sh.h[hb ^ 1].n : 0, qn);
            }
            // ---- WRITE: thread = staging position ----
            if (!skip) {
                const int nst = min(H.cb[g1] - gb, ORD_STG);
                for (int p0 = 0; p0 < nst; p0 += ORD_THREADS) {
                    const int p = p0 + tid;
                    const bool act = p < nst;
                    if (p0 + wave * 64 >= nst) break;                        // (no barrier inside: a wavefront without positions is done)
                    const uint32_t key = act ? stage_key[p] : ((uint32_t)(g1 - 1) << 26);
                    const int col = (int)ORD_COL(key);
                    const uint32_t id = ORD_ID(key);
                    const int bk = bucket(id);
                    const int e = (int)((sh.cur[col][bk >> 1] >> ((bk & 1) * 16)) & 0xffffu), n = (int)((sh.cnt[col][bk >> 1] >> ((bk & 1) * 16)) & 0xffffu);   // the bucket occupies [e - n, e)
                    int rk = e - n;
                    if (act) {
                        // (the bucket holds the record itself and typically 0-2 others: four independent reads, then the rare rest)
                        const int b0 = e - n, bl = e - 1;
                        const uint32_t m0 = ORD_ID(stage_key[b0]), m1 = ORD_ID(stage_key[min(b0 + 1, bl)]), m2 = ORD_ID(stage_key[min(b0 + 2, bl)]),
                                       m3 = ORD_ID(stage_key[min(b0 + 3, bl)]);
                        rk += (int)(m0 < id) + (int)(n > 1 && m1 < id) + (int)(n > 2 && m2 < id) + (int)(n > 3 && m3 < id);
                        for (int m = b0 + 4; m < e; ++m) rk += (ORD_ID(stage_key[m]) < id) ? 1 : 0;
                    }
                    const int rel = rk - (H.cb[col] - gb);                          // rank inside the column
                    const bool valid = act && rel >= 0 && rel < H.k[col];          // (always, unless a log overflowed: that build is void, but stays in bounds)
                    const int64_t o = H.out[col] + rel;
                    int32_t rp = 0;
                    const bool pd_ = valid && a.pend_items && ((key >> 30) & 1u);
                    if ((valid && a.rowpos) || pd_) rp = a.iperm[id];
                    if (valid) {
                        a.rowval[o] = (int32_t)id;
                        a.nzval[o] = sqrt(stage_d2[p]);                             // the log carries d2
                        if (a.rowpos) a.rowpos[o] = rp;
                    }
                    if (a.rec_bits && valid && (key >> 31)) atomicAnd(&sh.bits[rk >> 5], ~(1u << (rk & 31)));
                    if (a.pend_items) {
                        // entries whose key carries the pair kernel's broad-phase flag (the segment's box meets an obstacle's) are listed
                        // for k_sweep_pending -- (entry, column sample, row position) -- in this workgroup's own segment of the item array
                        const unsigned long long pm = __ballot(pd_);
                        if (pm) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(&sh.pcount, (int)__popcll(pm));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (pd_) {
                                const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                                if (pos < a.pend_wcap)
                                    a.pend_items[(int64_t)blockIdx.x * a.pend_wcap + pos] =
                                        make_uint4((uint32_t)(uint64_t)o, (uint32_t)((uint64_t)o >> 32), (uint32_t)H.ho[col], (uint32_t)rp);
                            }
                        }
                    }
                }
            }
            lds_barrier();
            if (a.rec_bits) {
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
                        if (w64 != ~0ull) atomicAnd(&a.mask[wd], w64);
                    }
                }
                lds_barrier();
            }
            g0 = g1;
        }
        if (tid < ORD_COLS) sh.h[hb ^ 1].out[tid] = hout;     // the next quarter's output offsets (requested before the write-out)
        qi = qn;
        qn = a.qctr ? 2 * (int64_t)gridDim.x + xcd + 8 * (int64_t)sh.nextk : qn + gridDim.x;      // (gridDim.x is a multiple of 4 with a counter: 2 nb = 0 mod 8)
    }
    if (a.pend_items) {
        lds_barrier();
        if (tid == 0) {
            const int c = sh.pcount;
            a.pend_cnt[blockIdx.x] = (int32_t)min((int64_t)c, a.pend_wcap);
            if (c > a.pend_wcap) *a.pend_over = 1;            // a segment was too short: the host sweeps the whole graph instead
        }
    }
}

// The mask of a speculative step preset ahead of the ordering pass (entries = the trusted capacity): issued on the side stream beside the
// exact pair tests instead of between them and the ordering kernel.  mpfmt_order_logs recognises it by its word count.
int32_t mpfmt_mask_preset(mpfmt_ctx* ctx, int64_t entries)
{
    int32_t rc;
    const int64_t words = (entries + 63) / 64;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)(words + 1)))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, 0xFF, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
    ctx->mask_preset_words = words;
    return MPFMT_OK;
}

// order the logs of the counted graph into the CSC (mask_entries: entries the mask is sized for, as in mpfmt_launch_graph_sweep)
int32_t mpfmt_order_logs(mpfmt_ctx* ctx, const int32_t* spec_fail, int64_t mask_entries)
{
    const int64_t nt = ctx->tile_end - ctx->tile_begin;
    int32_t rc;
    ctx->rowpos_valid = false;
    const bool recbits = ctx->bits_in_records;                // form 2: the blocked edges are marked in the keys, this pass also writes the mask
    const bool pend = ctx->broad_in_drain && !recbits;        // form 1: the flagged entries are listed for k_sweep_pending
    if (recbits) {
        // (sized with the slack a following speculative step asks for, so that it finds the mask in place)
        const int64_t words = (std::max<int64_t>(ctx->nnz, mask_entries) + 63) / 64;
        const int64_t words_alloc = (std::max<int64_t>((int64_t)((double)ctx->nnz * 1.02) + 4096, mask_entries) + 63) / 64 + 1;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)words_alloc))) return rc;
        // preset to ones (blocked entries are cleared); an empty graph keeps one zero word
        if (!(ctx->mask_preset_words == words && ctx->nnz > 0))
            HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, ctx->nnz > 0 ? 0xFF : 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
    }
    ctx->mask_preset_words = -1;
    if (ctx->nnz == 0 || nt <= 0) { if ((rc = mpfmt_side_join(ctx))) return rc; if (recbits) { ctx->graph_swept = true; ctx->sweep_in_order = true; } return MPFMT_OK; }
    // option sweep_sorted: also keep every row's cell-sorted position, so the whole sweep can gather from Xs (see kernels_sweep.hip)
    const bool want_rowpos = ctx->sweep_sorted && !pend && !recbits;
    if (want_rowpos && (rc = mpfmt_ensure(ctx, (void**)&ctx->rowpos, sizeof(int32_t) * (size_t)std::max<int64_t>(std::max(ctx->nnz, ctx->nnz_cap), 1)))) return rc;
    const size_t lds = (size_t)ORD_STG * 12 + sizeof(ord_shared);
    // (the attribute belongs to the kernel ON A DEVICE: set per launch -- a cached flag would cover the first device of a process
    // that drives several, and be written by concurrent ctx threads)
    HIPCHK(ctx, hipFuncSetAttribute((const void*)k_order_logs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // persistent workgroups: as many as fit the chip at once (4 per CU by their LDS and registers), each takes every nb-th quarter tile
    if (ctx->ord_per_cu == 0) {
        int per_cu = 0;
        HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_order_logs, ORD_THREADS, lds));
        ctx->ord_per_cu = std::max(1, std::min(per_cu, 4));
    }
    unsigned nb = (unsigned)std::min<int64_t>(nt * 4, (int64_t)ctx->num_cus * ctx->ord_per_cu);
    // quarters drawn from the per-XCD counters (zeroed with the step's counter arena) when the launch is a whole number of draws' strides
    // (... and the quarters are long: a counter hands out ~90 draws per microsecond, workgroups that finish a 400-record quarter every
    // few microseconds wait for it -- 2-D N = 1e6 at 440 records per quarter: 1.27 -> 1.47 ms per step; cfg2 at 1240: +2 %; the north
    // star at 1720: -9 % of the kernel)
    const bool draw = ctx->ord_draw && ctx->ord_ctr && nb >= 8 && (ctx->ord_draw > 1 || ctx->nnz >= (int64_t)1536 * nt * 4);
    if (draw) nb &= ~3u;
    if (pend) {
        // one segment of the item array per workgroup; workgroups take every nb-th quarter tile, so their shares are even: 1.5x the mean + 4096
        const int64_t entries = std::max<int64_t>(ctx->nnz, ctx->nnz_cap);
        ctx->pend_wcap = ctx->debug_small_lists ? 8 : entries * 3 / (2 * (int64_t)nb) + 4096;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->pend_items, sizeof(uint4) * (size_t)ctx->pend_wcap * nb))) return rc;
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->pend_cnt, sizeof(int32_t) * (size_t)(nb + 1)))) return rc;
        ctx->pend_over = ctx->pend_cnt + nb;
        HIPCHK(ctx, hipMemsetAsync(ctx->pend_over, 0, sizeof(int32_t), ctx->stream));
    }
    ord_args a{};
    a.qkey = ctx->qkey; a.qd2 = ctx->qd2; a.qcap = ctx->qcap; a.qlen = ctx->qlen; a.degs = ctx->degs;
    a.tile_begin = ctx->tile_begin; a.tile_end = ctx->tile_end; a.colptr = ctx->colptr; a.perm = ctx->perm; a.iperm = ctx->iperm;
    a.rowval = ctx->rowval; a.nzval = ctx->nzval; a.rowpos = want_rowpos ? ctx->rowpos : nullptr;
    a.bucket_mul = ctx->N > 128 ? (uint32_t)((128ull << 32) / (uint64_t)ctx->N) : 0u;
    a.spec_fail = spec_fail;
    a.mask = (unsigned long long*)ctx->graph_free; a.nnz_dev = ctx->colptr + ctx->N;
    a.pend_items = pend ? (uint4*)ctx->pend_items : nullptr; a.pend_wcap = ctx->pend_wcap; a.pend_cnt = ctx->pend_cnt; a.pend_over = ctx->pend_over;
    a.rec_bits = recbits ? 1 : 0;
    a.deg_clear = ctx->world > 1 ? ctx->deg : nullptr;
    a.qctr = draw ? ctx->ord_ctr : nullptr;
    if ((rc = mpfmt_side_join(ctx))) return rc;               // k_exact_pairs' marks (on the side stream since the pair kernel ended)
    hipLaunchKernelGGL(k_order_logs, dim3(nb), dim3(ORD_THREADS), lds, ctx->stream, a);
    HIPCHK(ctx, hipGetLastError());
    if (pend) ctx->pend_nseg = (int)nb;
    ctx->pend_valid = pend;                                   // the flagged entries have been listed for k_sweep_pending
    ctx->rowpos_valid = want_rowpos;                          // every entry's row is also known by its cell-sorted position (the sweep gathers from Xs)
    if (recbits) { ctx->graph_swept = true; ctx->sweep_in_order = true; }
    return MPFMT_OK;
}
