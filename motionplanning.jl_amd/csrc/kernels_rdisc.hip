// r-disc neighbour graph on gfx950: cell-grid binning + tile-vs-cell-run distance sweep.
//
// Replaces, for every sample v at once, inball(V, dist, DS::TreeDistanceDS, v, r)
// (reference src/nearneighbors.jl:179-183: KD-tree inrange + drop self + colwise distances) and
// the index build helper_data_structures(V, Euclidean) (src/statespaces/geometric.jl:14).
//
// Canonical arithmetic (must equal the CPU parity checker under oracle/ bit for bit): d2 = sum_i (q_i - c_i)^2
// accumulated in index order, fp64, unfused (-ffp-contract=off); neighbour <=> i != v && d2 <= r*r;
// dist = sqrt(d2).
//
// Design (MI355X): samples are binned into a uniform grid with cells >= r wide, sorted by cell id
// and stored as 64-sample tiles in SoA form ([tile][dim][64], 512 B contiguous per dimension, so a
// wavefront's query load is one coalesced 512 B read per dimension).  One wavefront = one tile of 64
// queries (lane = query, coordinates in VGPRs).  Candidate samples are the contiguous runs of the
// sorted array that cover the grid cells within r of the tile's tight bounding box; each 64-sample
// candidate chunk is staged once into LDS and broadcast to all 64 lanes (every lane reads the same
// LDS address: conflict-free broadcast), so HBM/L2 traffic per pair test is 1/64 of a gather.
// The kernel is fp64-VALU bound (3 ops per dimension per pair); the grid cuts the pair count from
// N^2 to ~N * (candidates in the Minkowski sum of the tile box and the r-ball).
//
// Two passes with identical arithmetic: COUNT (degrees -> scan -> colptr) and FILL (lane-private
// sequential slots, no atomics), then a per-column rank sort puts row indices in ascending order
// (the SparseVector contract of nearneighbors.jl:138-198).
#include "mpfmt_internal.h"
#include "mf_operand.h"
#include <cstring>
#include <cmath>
#include <algorithm>
#include <vector>

#define NXCD 8

// ------------------------------------------------------------------------------------------------
// grid build
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int cell_of(double x, double lo, double inv_w, int g)
{
    double f = floor((x - lo) * inv_w);
    int c = (int)f;
    if (!(f >= 0.0)) c = 0;
    if (f >= (double)g) c = g - 1;
    return c;
}

// ------------------------------------------------------------------------------------------------
// The cell sort (hand-written: a counting sort by cell id, then an ordering pass inside each cell).
// Sort key = cell id (mpfmt_grid: row-major, or block-major on a sharded ctx) followed by fb bits of the sample's position INSIDE
// its cell along the last dimension, then the sample index.  A tile is 64 consecutive samples of the sorted order and usually takes
// the tail of one cell and the head of the next one in the row; with the samples of a cell in index order both parts span their
// whole cells (tile extent two cells along the last dimension), with this key they are the upper end of one cell and the lower end
// of the next -- one cell width.  The order is a total one (index last), so every rank of a sharded run derives the SAME order, and
// with it the same shards, from the samples alone.
//   k_cellkey_count : key of every sample, its arrival number in its cell (the returning atomic that counts the cell)
//   k_scan_*        : exclusive scan of the cell counts in place -> cellstart
//   k_cell_need     : (sharded) the tiles this rank reads: its own and those of the cells next to its own cells
//   k_cell_scatter  : (fine bits, sample index) of every sample to cellstart[cell] + arrival number
//   k_cell_order    : one wavefront per cell puts the cell's items in ascending order
// ------------------------------------------------------------------------------------------------
// (the counting atomic returns the sample's arrival number in its cell: the scatter needs no atomic of its own.  A plain count with
// the scatter drawing its own numbers -- only a third of the samples is scattered on a rank of eight -- measured slower at every
// shard count: 1.12 vs 1.07 ms per step at 8 ranks, 3.66 vs 3.62 unsharded)
__global__ void k_cellkey_count(const double* __restrict__ Xo, int64_t N, int d, mpfmt_grid G, int fb,
                                uint32_t* __restrict__ key, uint32_t* __restrict__ slot, int32_t* __restrict__ cellcnt, int64_t cstride)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    int64_t id = 0;
    for (int i = 0; i < d; ++i)
        id += mpfmt_cell_term(G, i, cell_of(Xo[p * d + i], G.lo[i], G.inv_w[i], G.g[i]));
    uint32_t fine = 0;
    if (fb > 0) {
        const int L = d - 1;
        const double wdt = (G.g[L] > 1) ? G.w[L] : fmax(G.w[L], 1e-300);
        const double t = (Xo[p * d + L] - G.lo[L]) / wdt - (double)cell_of(Xo[p * d + L], G.lo[L], G.inv_w[L], G.g[L]);
        const double q = floor(fmin(fmax(t, 0.0), 1.0) * (double)(1u << fb));
        fine = min((uint32_t)q, (1u << fb) - 1u);
        if (!(t == t)) fine = 0;
    }
    key[p] = ((uint32_t)id << fb) | fine;
    slot[p] = (uint32_t)atomicAdd(&cellcnt[id * cstride], 1);
}

// exclusive scan of up to 4096 consecutive items per call of a 1024-thread workgroup (thread = 4 items); returns the block's total.
// (hand-written: the cell counts -> cellstart and the degrees -> colptr; in == out allowed)
template <typename T>
__device__ __forceinline__ T scan4096(const T* in, T* out, int64_t base, int64_t n, T carry, T* s_w)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i0 = base + (int64_t)tid * 4;
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i0 + k < n) ? in[i0 + k] : (T)0;
    const T t = v[0] + v[1] + v[2] + v[3];
    T inc = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const T u = __shfl_up(inc, off); if (lane >= off) inc += u; }
    __syncthreads();                                       // (s_w of the previous call has been read)
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        T x = lane < 16 ? s_w[lane] : (T)0;
        const T own = x;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const T u = __shfl_up(x, off); if (lane >= off) x += u; }
        if (lane < 16) s_w[lane] = x - own;                // exclusive over the wavefronts
        if (lane == 15) s_w[16] = x;                       // the block's total
    }
    __syncthreads();
    T o = carry + s_w[wave] + inc - t;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < n) out[i0 + k] = o; o += v[k]; }
    return s_w[16];
}
// one workgroup, chunk by chunk (the block sums of the long scans)
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_single(const T* in, T* out, int64_t n)
{
    __shared__ T s_w[17];
    T carry = 0;
    for (int64_t base = 0; base < n; base += 4096) carry += scan4096<T>(in, out, base, n, carry, s_w);
}
// one workgroup, ONE pass over up to 36 K items staged in LDS (the cell counters of an index build: 15 625 at the north star): loaded
// and stored coalesced, summed by thread = a contiguous run of an ODD number of items (no bank conflicts), one barrier chain instead
// of one per 4096 items
#define SCAN_LDS_MAX 36864
// the counters of a block-major id range, one per 64-byte line, back into the compact array the scan runs over
__global__ void k_cellcnt_compact(const int32_t* __restrict__ pad, int32_t* __restrict__ out, int n, int64_t cstride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pad[(size_t)i * cstride];
}
__global__ __launch_bounds__(1024) void k_scan_lds_i32(const int32_t* in, int32_t* out, int n, int64_t istride)
{
    extern __shared__ int32_t s_a[];
    __shared__ int32_t s_w[17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = ((n + 1023) / 1024) | 1;
    for (int i = tid; i < seg * 1024; i += 1024) s_a[i] = (i < n) ? in[(int64_t)i * istride] : 0;      // (istride > 1: one counter per 64-byte line, see mpfmt_build_grid)
    __syncthreads();
    int32_t t = 0;
    for (int k = 0; k < seg; ++k) t += s_a[tid * seg + k];
    int32_t inc = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int32_t u = __shfl_up(inc, off); if (lane >= off) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        int32_t x = lane < 16 ? s_w[lane] : 0;
        const int32_t own = x;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int32_t u = __shfl_up(x, off); if (lane >= off) x += u; }
        if (lane < 16) s_w[lane] = x - own;
    }
    __syncthreads();
    int32_t o = s_w[wave] + inc - t;
    for (int k = 0; k < seg; ++k) { const int32_t v = s_a[tid * seg + k]; s_a[tid * seg + k] = o; o += v; }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) out[i] = s_a[i];
}
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_block(const T* in, T* out, int64_t n, T* __restrict__ bsum)
{
    __shared__ T s_w[17];
    const T tot = scan4096<T>(in, out, (int64_t)blockIdx.x * 4096, n, (T)0, s_w);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_add(T* __restrict__ a, int64_t n, const T* __restrict__ bsum)
{
    const T add = bsum[blockIdx.x];
    const int64_t i0 = (int64_t)blockIdx.x * 4096 + (int64_t)threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i0 + k < n) a[i0 + k] += add;
}
// out = exclusive scan of in (n items; in == out allowed); bsum: room for ceil(n / 4096) + 1 items
template <typename T>
static void launch_scan(hipStream_t st, const T* in, T* out, int64_t n, T* bsum)
{
    const int64_t nsb = (n + 4095) / 4096;
    if (nsb <= 16) {
        hipLaunchKernelGGL((k_scan_single<T>), dim3(1), dim3(1024), 0, st, in, out, n);
    } else {
        hipLaunchKernelGGL((k_scan_block<T>), dim3((unsigned)nsb), dim3(1024), 0, st, in, out, n, bsum);
        hipLaunchKernelGGL((k_scan_single<T>), dim3(1), dim3(1024), 0, st, (const T*)bsum, bsum, nsb);
        hipLaunchKernelGGL((k_scan_add<T>), dim3((unsigned)nsb), dim3(1024), 0, st, out, n, (const T*)bsum);
    }
}

// does the cell [cs, ce) of the sorted order touch a tile this rank reads?  (tileneed == nullptr: the index is whole)
__device__ __forceinline__ bool cell_wanted(const uint8_t* __restrict__ tileneed, int64_t cs, int64_t ce)
{
    if (!tileneed) return true;
    const int64_t t0 = cs >> 6, t1 = (ce - 1) >> 6;
    if (t1 - t0 >= 4) return true;
    bool w = false;
    for (int64_t t = t0; t <= t1; ++t) w = w || tileneed[t] != 0;
    return w;
}

// Sharded ctx: which tiles does this rank read at all?  Its own (positions [pos_b, pos_e) of the sorted order) and those that hold a
// sample within r of an own sample: such a sample lies in a cell next to (Chebyshev distance <= 1) the own sample's cell, cells
// being at least r wide.  One workgroup per cell id; the workgroups of own cells mark the tiles of all their neighbours (thread =
// neighbour offset).  Tiles left unmarked are never built: they get empty boxes, which every candidate test rejects.
__global__ __launch_bounds__(256) void k_cell_need(const int32_t* __restrict__ cellstart, mpfmt_grid G, int d, int64_t pos_b, int64_t pos_e,
                                                   int64_t noff, uint8_t* __restrict__ tileneed)
{
    const int64_t id = blockIdx.x;                           // (one workgroup per cell id: one wavefront per cell measured 40 us against 20)
    const int64_t cs = cellstart[id], ce = cellstart[id + 1];
    // own cell <=> it holds a position of [pos_b, pos_e)
    if (ce <= cs || ce <= pos_b || cs >= pos_e) return;
    int c[MPFMT_MAX_DIM];
    for (int i = 0; i < d; ++i) c[i] = mpfmt_cell_coord(G, i, id);
    for (int64_t t = threadIdx.x; t < noff; t += blockDim.x) {
        int64_t rem = t, nid = 0;
        bool ok = true;
        for (int i = d - 1; i >= 0; --i) {
            int o = 0;
            if (G.g[i] > 1) { o = (int)(rem % 3) - 1; rem /= 3; }
            const int cc = c[i] + o;
            ok = ok && cc >= 0 && cc < G.g[i];
            nid += mpfmt_cell_term(G, i, min(max(cc, 0), G.g[i] - 1));
        }
        if (!ok) continue;
        const int64_t ns = cellstart[nid], ne = cellstart[nid + 1];
        for (int64_t tl = ns >> 6; tl <= ((ne - 1) >> 6) && ne > ns; ++tl) tileneed[tl] = 1;
    }
}

__global__ void k_cell_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ slot, int64_t N, int fb,
                               const int32_t* __restrict__ cellstart, const uint8_t* __restrict__ tileneed, unsigned long long* __restrict__ items)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const uint32_t k = key[p];
    const int64_t id = (int64_t)(k >> fb);
    const int64_t cs = cellstart[id];
    if (tileneed && !cell_wanted(tileneed, cs, cellstart[id + 1])) return;
    items[cs + slot[p]] = ((unsigned long long)(k & ((1u << fb) - 1u)) << 32) | (unsigned long long)(uint32_t)p;
}

// One wavefront per cell: the cell's items (fine bits << 32 | sample index -- all different) into ascending order, written as the
// sorted order's sample index and cell key of each position.  <= 64 items: in registers (rank = items below mine, v_readlane
// broadcasts).  <= 1024: the same count over an LDS copy.  More (clustered or duplicated samples): a stable LSD radix sort, eight
// bits a pass, between the item array and a second one -- O(n) per pass whatever the cell holds.
#define CO_LDS 1024
__global__ __launch_bounds__(256) void k_cell_order(const int32_t* __restrict__ cellstart, int64_t ncells, int fb, int pbits,
                                                    const uint8_t* __restrict__ tileneed, unsigned long long* items, unsigned long long* items2,
                                                    int32_t* __restrict__ val_out, uint32_t* __restrict__ cellkey)
{
    __shared__ unsigned long long s_it[4][CO_LDS];
    __shared__ int32_t s_cnt[4][256];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t id = (int64_t)blockIdx.x * 4 + wave;
    if (id >= ncells) return;
    const int64_t cs = cellstart[id], ce = cellstart[id + 1];
    const int64_t n = (int64_t)__builtin_amdgcn_readfirstlane((int)(ce - cs));      // (wave-uniform: one cell per wavefront)
    if (n <= 0 || !cell_wanted(tileneed, cs, ce)) return;
    const uint32_t keyhi = (uint32_t)id << fb;
    if (n <= 64) {
        const unsigned long long mine = lane < n ? items[cs + lane] : ~0ull;
        int rank = 0;
        for (int j = 0; j < (int)n; ++j) {
            const unsigned long long o = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mine >> 32), j) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)mine, j);
            rank += (o < mine) ? 1 : 0;
        }
        if (lane < n) { val_out[cs + rank] = (int32_t)(uint32_t)mine; cellkey[cs + rank] = keyhi | (uint32_t)(mine >> 32); }
        return;
    }
    if (n <= CO_LDS) {
        unsigned long long* const it = s_it[wave];
        for (int i = lane; i < (int)n; i += 64) it[i] = items[cs + i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < (int)n; base += 256) {
            unsigned long long m[4];
            int rk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = base + k * 64 + lane; m[k] = i < (int)n ? it[i] : ~0ull; rk[k] = 0; }
            for (int j = 0; j < (int)n; ++j) {
                const unsigned long long o = it[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) rk[k] += (o < m[k]) ? 1 : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * 64 + lane;
                if (i < (int)n) { val_out[cs + rk[k]] = (int32_t)(uint32_t)m[k]; cellkey[cs + rk[k]] = keyhi | (uint32_t)(m[k] >> 32); }
            }
        }
        return;
    }
    // ---- stable LSD radix sort of a large cell ----
    int32_t* const cnt = s_cnt[wave];
    unsigned long long* src = items + cs;
    unsigned long long* dst = items2 + cs;
    const int npass = (pbits + 7) / 8 + (fb > 0 ? 1 : 0);
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = (pass < (pbits + 7) / 8) ? pass * 8 : 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) cnt[lane + 64 * k] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int64_t i = lane; i < n; i += 64) atomicAdd(&cnt[(int)((src[i] >> shift) & 255ull)], 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // exclusive scan of the 256 counts (lane = four consecutive digits)
            const int c0 = cnt[lane * 4], c1 = cnt[lane * 4 + 1], c2 = cnt[lane * 4 + 2], c3 = cnt[lane * 4 + 3];
            const int t = c0 + c1 + c2 + c3;
            int inc = t;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(inc, off); if (lane >= off) inc += u; }
            const int b0 = inc - t;
            cnt[lane * 4] = b0; cnt[lane * 4 + 1] = b0 + c0; cnt[lane * 4 + 2] = b0 + c0 + c1; cnt[lane * 4 + 3] = b0 + c0 + c1 + c2;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int64_t base = 0; base < n; base += 64) {
            const int64_t i = base + lane;
            const bool act = i < n;
            const unsigned long long item = act ? src[i] : 0ull;
            const int dg = act ? (int)((item >> shift) & 255ull) : 256;
            unsigned long long rem = __ballot(act);
            while (rem) {                                    // the lanes of one digit at a time, in lane order: stable
                const int L = __builtin_ctzll(rem);
                const int dv = __builtin_amdgcn_readlane(dg, L);
                const bool mine = dg == dv;
                const unsigned long long mm = __ballot(mine);
                const int pre = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
                const int b = cnt[dv];
                if (mine) dst[b + pre] = item;
                __builtin_amdgcn_wave_barrier();
                if (lane == L) cnt[dv] = b + (int)__popcll(mm);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                rem &= ~mm;
            }
        }
        __threadfence();                                     // this pass's stores before the next pass's loads (other lanes' addresses)
        unsigned long long* const sw = src; src = dst; dst = sw;
    }
    for (int64_t i = lane; i < n; i += 64) {
        const unsigned long long item = src[i];
        val_out[cs + i] = (int32_t)(uint32_t)item;
        cellkey[cs + i] = keyhi | (uint32_t)(item >> 32);
    }
}

// One pass over the tiles this rank reads (one wavefront per tile, lane = sample): the row is gathered once from the caller's array and
// leaves as the sorted AoS row (through LDS: coalesced stores), the tile's SoA slice, perm / iperm, the hull, the two sub-boxes (fp64,
// and fp32 rounded outward: the candidate side of the chunk-list test) and -- when the matrix-core pair kernel will run -- the fp16
// operand.  A tile is 64 consecutive samples of the cell-sorted order, so one in g_last tiles runs over the end of a grid row (one in
// g_last * g_prev over the end of two, ...): its hull then spans the whole extent of the minor dimensions although its samples sit in
// two compact groups.  The samples are split where the cell key jumps the most and each side gets its own box (tile_sub [tile][A lo,
// A hi, B lo, B hi][d]; B is an empty box (+1e300, -1e300) when the tile lies in one cell) -- the candidate lists test sub-box against
// sub-box, which keeps them ~30 % shorter than hull against hull.  Any split is valid: every sample lies in A or in B.
// A tile the rank does not read (tileneed) gets empty boxes and nothing else.
#define BT_MAXD 16
__device__ __forceinline__ float f32_down(double x) { float f = (float)x; if ((double)f > x) f = nextafterf(f, -INFINITY); return f; }
__device__ __forceinline__ float f32_up(double x) { float f = (float)x; if ((double)f < x) f = nextafterf(f, INFINITY); return f; }
struct bt_ops { void* ops; double scale; };               // ops == nullptr: no operands
__global__ __launch_bounds__(256) void k_build_tiles(const double* __restrict__ Xo, const int32_t* __restrict__ perm_sorted,
                                                     const uint32_t* __restrict__ cellkey, int fb, int64_t N, int64_t ntiles, int d,
                                                     int32_t* __restrict__ perm, int32_t* __restrict__ iperm, double* __restrict__ Xs,
                                                     double* __restrict__ Xt, double* __restrict__ tile_lo, double* __restrict__ tile_hi,
                                                     double* __restrict__ tile_sub, float* __restrict__ tile_sub32,
                                                     const uint8_t* __restrict__ tileneed, bt_ops ops, mpfmt_grid G)
{
    extern __shared__ double s_rows_dyn[];                  // [4][64 * d]: sized by the dimension (12 KB at d = 6, where the 32 KB of d = 16 would halve the residency)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    if (tileneed && !tileneed[tile]) {
        for (int i = lane; i < d; i += 64) { tile_lo[tile * d + i] = 1e300; tile_hi[tile * d + i] = -1e300; }
        for (int i = lane; i < 4 * d; i += 64) {
            const bool hi = ((i / d) & 1) != 0;
            tile_sub[tile * 4 * d + i] = hi ? -1e300 : 1e300;
            tile_sub32[tile * 4 * d + i] = hi ? -INFINITY : INFINITY;
        }
        return;
    }
    const int64_t sp = tile * 64 + lane;
    const int32_t o = (sp < N) ? perm_sorted[sp] : -1;
    perm[sp] = o;
    if (o >= 0) iperm[o] = (int32_t)sp;
    // cut where the cell key jumps the most
    const int64_t key = (int64_t)(cellkey[min(sp, N - 1)] >> fb);
    const int64_t nxt = __shfl_down(key, 1);
    int64_t jump = (lane < 63 && sp + 1 < N) ? llabs(nxt - key) : 0;
    int where = lane;
    for (int off = 32; off > 0; off >>= 1) {               // arg max (first lane among equals)
        const int64_t oj = __shfl_xor(jump, off);
        const int ow = __shfl_xor(where, off);
        if (oj > jump || (oj == jump && ow < where)) { jump = oj; where = ow; }
    }
    const int split = (jump > 0) ? where + 1 : 64;         // A = lanes [0, split), B = lanes [split, 64)
    double* rows = s_rows_dyn + (size_t)wave * 64 * d;
    // the row's d coordinates are requested together (a runtime-d loop that loads and reduces in turn sat out d gather latencies)
    double xr[BT_MAXD];
#pragma unroll
    for (int i = 0; i < BT_MAXD; ++i) xr[i] = (i < d && o >= 0) ? Xo[(int64_t)o * d + i] : __builtin_nan("");
#pragma unroll
    for (int i = 0; i < BT_MAXD; ++i) if (i < d) rows[lane * d + i] = xr[i];
    if (ops.ops) mf_write_operand(ops.ops, G, d, ops.scale, sp, o >= 0, xr);
    for (int i = 0; i < d; ++i) {
        const double x = rows[lane * d + i];                 // (own row: no barrier needed)
        Xt[(tile * d + i) * 64 + lane] = x;
        double amn = lane < split ? x : NAN, amx = amn, bmn = lane < split ? NAN : x, bmx = bmn;   // fmin/fmax ignore NaN (pads too)
        for (int off = 32; off > 0; off >>= 1) {
            amn = fmin(amn, __shfl_xor(amn, off));
            amx = fmax(amx, __shfl_xor(amx, off));
            bmn = fmin(bmn, __shfl_xor(bmn, off));
            bmx = fmax(bmx, __shfl_xor(bmx, off));
        }
        if (lane == 0) {
            tile_lo[tile * d + i] = fmin(amn, bmn); tile_hi[tile * d + i] = fmax(amx, bmx);      // the hull
            double* t = tile_sub + tile * 4 * d;
            float* t32 = tile_sub32 + tile * 4 * d;
            const double al = (amn == amn) ? amn : 1e300, ah = (amx == amx) ? amx : -1e300;
            const double bl = (bmn == bmn) ? bmn : 1e300, bh = (bmx == bmx) ? bmx : -1e300;
            t[i] = al; t[d + i] = ah; t[2 * d + i] = bl; t[3 * d + i] = bh;
            t32[i] = f32_down(al); t32[d + i] = f32_up(ah); t32[2 * d + i] = f32_down(bl); t32[3 * d + i] = f32_up(bh);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double* __restrict__ out = Xs + tile * 64 * d;
    for (int t = lane; t < 64 * d; t += 64) out[t] = rows[t];
}

static inline int32_t ensure(mpfmt_ctx* ctx, void** p, size_t bytes) { return mpfmt_ensure(ctx, p, bytes); }


// shard boundaries as fractions of the cell-sorted order, at equal ESTIMATED WORK rather than equal sample counts: a column's work
// (candidate pairs, edges) goes with the number of grid cells around its own -- 2 instead of 3 per dimension at the faces of the
// domain -- so the face shards of an equal-count split would idle while the interior ones finish (30 % at 8 slabs of the north star).
// The cells of the leading axes are walked in id order (the order the samples are sorted in) with weight prod_i (cells within +-1
// along i); the cut positions assume a uniform density -- only the balance depends on that, never the result -- and every rank
// derives the same cuts from the grid alone.
// (a face cell sees 2 of 3 neighbouring cells -- two thirds of the candidate pairs -- but its columns keep more than two thirds of
// their hits, and since the half build the per-hit work of the drain outweighs the per-pair work of the filter: measured over the
// shards of the north star a face column costs 0.84 of an interior one, not 0.67 -- weight 2.35 / 3)
static void shard_cuts(mpfmt_ctx* ctx)
{
    const mpfmt_grid& G = ctx->grid;
    const int d = ctx->d, world = ctx->world;
    std::vector<int64_t> key = {(int64_t)world, (int64_t)d, (int64_t)G.nsplit};
    for (int i = 0; i < d; ++i) { key.push_back(G.g[i]); key.push_back(G.split[i]); }
    if (key == ctx->cut_key && (int)ctx->cut_frac.size() == world + 1) return;
    ctx->cut_frac.assign((size_t)world + 1, 0.0);
    ctx->cut_frac[(size_t)world] = 1.0;
    // the leading axes whose cells are walked one by one: at least the cut ones, then more while the walk stays short
    int m = std::max(1, G.nsplit);
    int64_t units = 1;
    for (int i = 0; i < m; ++i) units *= G.g[i];
    while (m < d - 1 && units * G.g[m] <= 4096) { units *= G.g[m]; ++m; }
    struct unit { int64_t id; double w; };
    std::vector<unit> u((size_t)units);
    std::vector<int> c((size_t)m, 0);
    auto nb = [](int cc, int n) { return n == 1 ? 1.0 : ((cc == 0 || cc == n - 1) ? 2.35 : 3.0); };
    for (int64_t k = 0; k < units; ++k) {
        int64_t id = 0; double w = 1.0;
        for (int i = 0; i < m; ++i) { id += mpfmt_cell_term(G, i, c[i]); w *= nb(c[i], G.g[i]); }
        u[(size_t)k] = {id, w};
        for (int i = m - 1; i >= 0; --i) { if (++c[i] < G.g[i]) break; c[i] = 0; }
    }
    std::sort(u.begin(), u.end(), [](const unit& a, const unit& b) { return a.id < b.id; });
    double total = 0.0;
    for (const unit& x : u) total += x.w;
    double acc = 0.0;
    int g = 1;
    for (int64_t k = 0; k < units && g < world; ++k) {
        const double w = u[(size_t)k].w;
        while (g < world && acc + w >= total * (double)g / (double)world) {
            ctx->cut_frac[(size_t)g] = ((double)k + (total * (double)g / (double)world - acc) / w) / (double)units;
            ++g;
        }
        acc += w;
    }
    for (; g < world; ++g) ctx->cut_frac[(size_t)g] = 1.0;
    ctx->cut_key = key;
}

int32_t mpfmt_build_grid(mpfmt_ctx* ctx, double r, bool whole)
{
    if (ctx->grid_r == r && ctx->Xt && ctx->index_rank == ctx->rank && ctx->index_world == ctx->world && !(whole && ctx->tileneed)) return MPFMT_OK;
    const int64_t N = ctx->N;
    const int d = ctx->d;
    mpfmt_timed tm1(ctx);

    // ---- choose cells: width >= r, total cells <= max(1, N/8) and < 2^31 ----------------------------
    mpfmt_grid& G = ctx->grid;
    G.gd = d;
    const int64_t cmax = std::max<int64_t>(1, std::min<int64_t>(N / 8, (int64_t)1 << 24));
    int gcap = 1024;
    for (;;) {
        int64_t prod = 1;
        for (int i = 0; i < d; ++i) {
            double ext = ctx->bb_hi[i] - ctx->bb_lo[i];
            int g = 1;
            if (r > 0.0 && ext > 0.0 && std::isfinite(ext / r)) {
                double q = std::floor(ext / r);
                g = (q >= (double)gcap) ? gcap : (int)q;
                if (g < 1) g = 1;
                // cell width ext/g must be >= r even after rounding
                while (g > 1 && ext / (double)g < r * (1.0 + 1e-9)) --g;
            } else if (!(r > 0.0) && ext > 0.0) {
                g = gcap;                        // r == 0: only exact duplicates match; finest grid allowed
            }
            G.g[i] = g;
            prod *= g;
            if (prod > ((int64_t)1 << 40)) break;
        }
        if (prod <= cmax || gcap == 1) break;
        gcap = (gcap > 2) ? gcap - std::max(1, gcap / 8) : 1;
    }
    // ---- cell ids: block-major on a sharded ctx (one cut axis per factor of two of the world, at most three, never the last axis) ----
    for (int i = 0; i < MPFMT_MAX_DIM; ++i) { G.split[i] = 0; G.hstride[i] = 0; G.ext[i] = 1; }
    G.nsplit = 0;
    if (ctx->world > 1 && ctx->shard_blocks) {
        int want = 0;
        while ((1 << want) < ctx->world && want < 3) ++want;
        int64_t lead = 1;
        for (int i = 0; i < d - 1 && G.nsplit < want; ++i) {
            if (i != G.nsplit) break;                          // (the cut axes are the leading ones)
            if (G.g[i] < 2 || lead * G.g[i] > ((int64_t)1 << 20)) break;
            lead *= G.g[i];
            G.split[i] = (G.g[i] + 1) / 2;
            ++G.nsplit;
        }
    }
    int64_t stride = 1;
    for (int i = d - 1; i >= 0; --i) {
        G.ext[i] = G.split[i] > 0 ? G.split[i] : G.g[i];
        G.stride[i] = stride;
        stride *= G.ext[i];
        double ext = ctx->bb_hi[i] - ctx->bb_lo[i];
        G.lo[i] = ctx->bb_lo[i];
        G.w[i] = (G.g[i] > 1) ? ext / (double)G.g[i] : (ext > 0 ? ext : 1.0);
        G.inv_w[i] = (G.g[i] > 1) ? (double)G.g[i] / ext : 0.0;
    }
    G.inner = stride;
    for (int i = 0; i < G.nsplit; ++i) G.hstride[i] = G.inner << (G.nsplit - 1 - i);
    G.ncells = G.inner << G.nsplit;
    for (int i = d; i < MPFMT_MAX_DIM; ++i) { G.g[i] = 1; G.stride[i] = 0; G.lo[i] = 0; G.w[i] = 1; G.inv_w[i] = 0; }

    ctx->ntiles = (N + 63) / 64;
    const int64_t npad = ctx->ntiles * 64;

    // ---- the shard: a contiguous range of 256-sample blocks (4 tiles) of the cell-sorted order ----
    {
        shard_cuts(ctx);
        const int64_t nblocks4 = (ctx->ntiles + 3) / 4;
        auto cut = [&](int g) -> int64_t {
            if (g <= 0) return 0;
            if (g >= ctx->world) return nblocks4;
            return std::min<int64_t>(nblocks4, std::max<int64_t>(0, (int64_t)std::llround(ctx->cut_frac[(size_t)g] * (double)nblocks4)));
        };
        ctx->tile_begin = std::min<int64_t>(ctx->ntiles, 4 * cut(ctx->rank));
        ctx->tile_end = std::min<int64_t>(ctx->ntiles, 4 * cut(ctx->rank + 1));
    }

    int32_t rc;
    if ((rc = ensure(ctx, (void**)&ctx->perm, sizeof(int32_t) * npad))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->iperm, sizeof(int32_t) * N))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->cellkey, sizeof(uint32_t) * N))) return rc;
    const size_t arena_cells = sizeof(int32_t) * (size_t)(G.ncells + 2);
    const size_t arena_bytes = arena_cells + (size_t)std::max<int64_t>(ctx->ntiles, 1);
    if ((rc = ensure(ctx, (void**)&ctx->idx_arena, arena_bytes))) return rc;
    ctx->cellstart = (int32_t*)ctx->idx_arena;
    ctx->list_max = ctx->cellstart + (G.ncells + 1);
    ctx->tileneed_buf = (uint8_t*)ctx->idx_arena + arena_cells;
    if ((rc = ensure(ctx, (void**)&ctx->Xt, sizeof(double) * npad * d))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->Xs, sizeof(double) * std::max<int64_t>(npad, 1) * d))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->tile_lo, sizeof(double) * ctx->ntiles * d))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->tile_hi, sizeof(double) * ctx->ntiles * d))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->tile_sub, sizeof(double) * ctx->ntiles * 4 * d))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->tile_sub32, sizeof(float) * ctx->ntiles * 4 * d))) return rc;

    // the matrix-core pair kernel's operands are written by the tile pass when that kernel is going to run
    bool mf = false;
    float negT = 0.f;
    if (ctx->rdisc_path != 1 && (rc = mpfmt_mfma_prepare(ctx, r, &negT, &mf))) return rc;
    if (mf && (rc = ensure(ctx, (void**)&ctx->ops, 32 * (size_t)std::max<int64_t>(npad, 1)))) return rc;

    // shard + halo index: the matrix-core path reads candidates only through its chunk lists, which the empty boxes of unbuilt tiles
    // keep them out of; the exact VALU kernel walks cell runs directly and needs every tile.  The halo search visits 3^(gridded
    // dimensions) neighbours of every own cell: past 1e8 visits the whole index is cheaper than the search.
    ctx->tileneed = nullptr;
    int64_t noff = 1;
    if (ctx->world > 1 && !whole && mf && ctx->index_halo && N > 0) {
        double v = 1.0;
        for (int i = 0; i < d; ++i) if (G.g[i] > 1) { noff *= 3; v *= 3.0; }
        if (v * (double)G.ncells / (double)ctx->world <= 1e8 && ctx->tile_end > ctx->tile_begin) ctx->tileneed = ctx->tileneed_buf;
    }

    if (N > 0) {
        int bits = 1;
        while (((int64_t)1 << bits) < G.ncells) ++bits;
        const int fb = std::max(0, std::min(ctx->cell_fb_max, 32 - bits));       // position bits inside the cell (k_cellkey_count)
        ctx->cell_fb = fb;
        int pbits = 1;
        while (((int64_t)1 << pbits) < N) ++pbits;
        // scratch: key[N] slot[N] val_out[N] (4 bytes each), items[N] items2[N] (8 bytes each), block sums of the scan
        const int64_t nsb = (G.ncells + 1 + 4095) / 4096;
        const size_t o_key = 0, o_slot = o_key + 4 * (size_t)N, o_val = o_slot + 4 * (size_t)N;
        const size_t o_it = (o_val + 4 * (size_t)N + 15) & ~(size_t)15, o_it2 = o_it + 8 * (size_t)N, o_bs = o_it2 + 8 * (size_t)N;
        void* scr;
        if ((rc = mpfmt_scratch(ctx, o_bs + 4 * (size_t)(nsb + 1), &scr))) return rc;
        uint32_t* key = (uint32_t*)((char*)scr + o_key);
        uint32_t* slot = (uint32_t*)((char*)scr + o_slot);
        int32_t* val_out = (int32_t*)((char*)scr + o_val);
        unsigned long long* items = (unsigned long long*)((char*)scr + o_it);
        unsigned long long* items2 = (unsigned long long*)((char*)scr + o_it2);
        int32_t* bsum = (int32_t*)((char*)scr + o_bs);
        HIPCHK(ctx, hipMemsetAsync(ctx->idx_arena, 0, ctx->tileneed ? arena_bytes : arena_cells, ctx->stream));
        ctx->list_max_clean = true;
        const int B = 256;
        // Block-major cell ids (sharded ctx): the counting atomics of k_cellkey_count on the compact counter array take 87 us instead of
        // the row-major ids' 54 for the same 1e6 samples -- memory-side atomics, and the block-major id range (holes at every odd cut
        // axis) lands its hot counters on few channels (tools/ubench/atomic_hist.hip: 24 atomics / ns whatever the layout, UNLESS the
        // used counters alias).  One counter per 64-byte line restores the 54 us (A/B on one box, rank 3 of 8: 87.4 -> 54.8; the
        // unsharded step: 54.8 -> 54.0, left compact); the padded array is cleared and compacted by two small launches.
        const int64_t cstride = (G.nsplit > 0) ? 16 : 1;
        int32_t* cnt = ctx->cellstart;
        if (cstride > 1) {
            const size_t need = sizeof(int32_t) * (size_t)(G.ncells + 2) * (size_t)cstride;
            if ((rc = ensure(ctx, (void**)&ctx->cellcnt_pad, need))) return rc;
            HIPCHK(ctx, hipMemsetAsync(ctx->cellcnt_pad, 0, need, ctx->stream));
            cnt = ctx->cellcnt_pad;
        }
        hipLaunchKernelGGL(k_cellkey_count, dim3((unsigned)((N + B - 1) / B)), dim3(B), 0, ctx->stream, ctx->Xo, N, d, G, fb, key, slot, cnt, cstride);
        const bool scan_lds = G.ncells + 1 <= SCAN_LDS_MAX;
        if (cstride > 1 && !scan_lds)                            // (the LDS scan reads the padded counters itself)
            hipLaunchKernelGGL(k_cellcnt_compact, dim3((unsigned)((G.ncells + 1 + 255) / 256)), dim3(256), 0, ctx->stream, (const int32_t*)cnt, ctx->cellstart,
                               (int)(G.ncells + 1), cstride);
        if (scan_lds) {
            const int seg = (int)(((G.ncells + 1 + 1023) / 1024) | 1);
            const size_t lds = sizeof(int32_t) * (size_t)seg * 1024;
            HIPCHK(ctx, hipFuncSetAttribute((const void*)k_scan_lds_i32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_scan_lds_i32, dim3(1), dim3(1024), lds, ctx->stream, (const int32_t*)cnt, ctx->cellstart, (int)(G.ncells + 1), cstride);
        } else {
            launch_scan<int32_t>(ctx->stream, (const int32_t*)ctx->cellstart, ctx->cellstart, G.ncells + 1, bsum);
        }
        if (ctx->tileneed)
            hipLaunchKernelGGL(k_cell_need, dim3((unsigned)G.ncells), dim3(256), 0, ctx->stream, (const int32_t*)ctx->cellstart, G, d,
                               ctx->tile_begin * 64, std::min<int64_t>(ctx->tile_end * 64, N), noff, ctx->tileneed);
        hipLaunchKernelGGL(k_cell_scatter, dim3((unsigned)((N + B - 1) / B)), dim3(B), 0, ctx->stream, (const uint32_t*)key, (const uint32_t*)slot, N, fb,
                           (const int32_t*)ctx->cellstart, (const uint8_t*)ctx->tileneed, items);
        hipLaunchKernelGGL(k_cell_order, dim3((unsigned)((G.ncells + 3) / 4)), dim3(256), 0, ctx->stream, (const int32_t*)ctx->cellstart, G.ncells, fb, pbits,
                           (const uint8_t*)ctx->tileneed, items, items2, val_out, ctx->cellkey);
        bt_ops bo{mf ? ctx->ops : nullptr, ctx->mf_scale};
        hipLaunchKernelGGL(k_build_tiles, dim3((unsigned)((ctx->ntiles + 3) / 4)), dim3(256), sizeof(double) * 4 * 64 * (size_t)d, ctx->stream, ctx->Xo, (const int32_t*)val_out,
                           (const uint32_t*)ctx->cellkey, fb, N, ctx->ntiles, d, ctx->perm, ctx->iperm, ctx->Xs, ctx->Xt, ctx->tile_lo, ctx->tile_hi,
                           ctx->tile_sub, ctx->tile_sub32, (const uint8_t*)ctx->tileneed, bo, G);
        HIPCHK(ctx, hipGetLastError());
    }
    tm1.end("grid");
    ctx->grid_r = r;
    ctx->index_rank = ctx->rank; ctx->index_world = ctx->world;
    ctx->ops_r = mf ? r : -1.0;
    ctx->lists_r = -1.0;
    ctx->graph_r = -1.0; ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
    return MPFMT_OK;
}

// ------------------------------------------------------------------------------------------------
// the pair sweep
// ------------------------------------------------------------------------------------------------
struct rdisc_args {
    const double* Xt;
    const int32_t* perm;
    const int32_t* cellstart;
    const double* tile_lo;
    const double* tile_hi;
    double r2;        // r*r  (the membership threshold, exact)
    double rpad;      // r*(1+1e-9): conservative radius for cell pruning only
    int32_t S;
    int64_t tile_begin;
    int64_t nitems;   // (tile_end - tile_begin) * S
    int64_t npad;
    int32_t* slice_cnt;
    const int64_t* tptr;       // offsets of the sorted-order staging CSC
    int32_t* rowtmp;
    double* valtmp;
    unsigned long long* pairs;
};

template <int D, bool FILL>
__global__ __launch_bounds__(64) void k_rdisc(rdisc_args a, mpfmt_grid G)
{
    __shared__ double sh[D * 64];
    __shared__ int32_t shp[64];
    const int lane = threadIdx.x;

    // XCD-aware item mapping: blocks b, b+8, b+16.. run on the same XCD (private 4 MiB L2); give each
    // XCD a contiguous range of tiles so neighbouring tiles (which share candidate cells) share an L2.
    const int64_t nblk = gridDim.x;
    const int64_t per_xcd = nblk / NXCD;
    const int64_t b = blockIdx.x;
    const int64_t item = (b % NXCD) * per_xcd + (b / NXCD);
    if (item >= a.nitems) return;
    const int64_t tile = a.tile_begin + item / a.S;
    const int slice = (int)(item % a.S);
    const int64_t qpos = tile * 64 + lane;

    double q[D];
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] = a.Xt[(tile * D + i) * 64 + lane];

    // tile bounding box and per-dimension cell ranges (wave-uniform)
    double tlo[D], thi[D];
    int clo[D], chi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        tlo[i] = a.tile_lo[tile * D + i];
        thi[i] = a.tile_hi[tile * D + i];
        clo[i] = cell_of(tlo[i] - a.rpad, G.lo[i], G.inv_w[i], G.g[i]);
        chi[i] = cell_of(thi[i] + a.rpad, G.lo[i], G.inv_w[i], G.g[i]);
    }
    const double rpad2 = a.rpad * a.rpad;
    constexpr int L = D - 1;                    // last dimension: contiguous in the sorted order

    int64_t base = 0;
    int32_t cnt = 0;
    if (FILL) {
        base = a.tptr[qpos];
        for (int s = 0; s < slice; ++s) base += a.slice_cnt[(int64_t)s * a.npad + qpos];
    }

    int64_t rows = 1;
#pragma unroll
    for (int i = 0; i < L; ++i) rows *= (chi[i] - clo[i] + 1);

    unsigned long long tested = 0;
    int64_t blk = 0;
    for (int64_t row = 0; row < rows; ++row) {
        // decode the row into cell coordinates of dims 0..L-1 (dim L-1 fastest), accumulate the
        // squared gap between the tile box and the cell slab in those dims
        int64_t rem = row;
        int64_t cbase = 0;
        double partial = 0.0;
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            const int span = chi[i] - clo[i] + 1;
            const int c = clo[i] + (int)(rem % span);
            rem /= span;
            cbase += mpfmt_cell_term(G, i, c);
            const double eps = G.w[i] * 1e-9;
            const double blo = G.lo[i] + (double)c * G.w[i] - eps;
            const double bhi = G.lo[i] + (double)(c + 1) * G.w[i] + eps;
            double gap = fmax(fmax(blo - thi[i], tlo[i] - bhi), 0.0);
            if (G.g[i] == 1) gap = 0.0;
            partial += gap * gap;
        }
        if (partial > rpad2) continue;
        // trim the run along the last dimension
        int c0 = clo[L], c1 = chi[L];
        if (G.g[L] > 1) {
            const double eps = G.w[L] * 1e-9;
            while (c0 <= c1) {
                const double bhi = G.lo[L] + (double)(c0 + 1) * G.w[L] + eps;
                const double gap = fmax(tlo[L] - bhi, 0.0);
                if (partial + gap * gap > rpad2) ++c0; else break;
            }
            while (c1 >= c0) {
                const double blo = G.lo[L] + (double)c1 * G.w[L] - eps;
                const double gap = fmax(blo - thi[L], 0.0);
                if (partial + gap * gap > rpad2) --c1; else break;
            }
            if (c0 > c1) continue;
        }
        const int64_t ra = a.cellstart[cbase + c0];
        const int64_t rb = a.cellstart[cbase + c1 + 1];
        if (rb <= ra) continue;
        for (int64_t cb = ra >> 6; cb <= ((rb - 1) >> 6); ++cb, ++blk) {
            if ((int)(blk % a.S) != slice) continue;
            const int j0 = (int)(std::max<int64_t>(ra, cb * 64) - cb * 64);
            const int j1 = (int)(std::min<int64_t>(rb, cb * 64 + 64) - cb * 64);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < D; ++i) sh[i * 64 + lane] = a.Xt[(cb * D + i) * 64 + lane];
            if (FILL) shp[lane] = a.perm[cb * 64 + lane];
            __syncthreads();
            const int64_t cpos0 = cb * 64;
            tested += (unsigned long long)(j1 - j0);
#pragma unroll 4
            for (int j = j0; j < j1; ++j) {
                double d2 = 0.0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double t = q[i] - sh[i * 64 + j];
                    const double tt = t * t;
                    d2 = (i == 0) ? tt : d2 + tt;
                }
                const bool hit = (d2 <= a.r2) && (cpos0 + j != qpos);
                if (FILL) {
                    if (hit) {
                        a.rowtmp[base + cnt] = shp[j];
                        a.valtmp[base + cnt] = sqrt(d2);
                    }
                }
                cnt += hit ? 1 : 0;
            }
        }
    }
    if (!FILL) {
        a.slice_cnt[(int64_t)slice * a.npad + qpos] = cnt;
        if (lane == 0 && a.pairs) atomicAdd(a.pairs + 2 * (blockIdx.x & 255), tested * 64ull);
    }
}

// degree of each ORIGINAL column = sum over slices of the sorted query's hits
__global__ void k_degree(const int32_t* __restrict__ slice_cnt, const int32_t* __restrict__ perm, int S, int64_t npad,
                         int64_t pos_begin, int64_t pos_end, int64_t* __restrict__ deg, int64_t* __restrict__ degs,
                         int32_t* __restrict__ max_deg, int64_t N_tail, int32_t* __restrict__ qmax)
{
    // (unsharded: the scans' extra last elements are zeroed here instead of by two fill launches)
    if (N_tail >= 0 && blockIdx.x == 0 && threadIdx.x == 0) { deg[N_tail] = 0; degs[npad] = 0; }
    int64_t s = pos_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t k = 0;
    if (s < pos_end) {
        const int32_t o = perm[s];
        if (o >= 0) {
            for (int i = 0; i < S; ++i) k += slice_cnt[(int64_t)i * npad + s];
            deg[o] = k;
        }
        degs[s] = k;                                           // (pad positions: 0 -- nothing else clears them on an unsharded ctx)
    }
    // longest column of the shard: the log-ordering kernel stages whole columns in LDS.  One candidate per workgroup, and it only
    // goes to the atomic when it beats the maximum it can see (a stale read at worst costs a redundant atomic): one atomic per
    // wavefront on the single address took 0.18 ms at N = 1e6 (~88 atomics / us), the pre-checked per-wavefront form 0.08
    // ... and the fullest quarter (16 consecutive positions: what one log of the single-pass build has to hold)
    __shared__ int s_m[4], s_q[4];
    int m = (int)min(k, (int64_t)0x7fffffff);
    int qs = m;
    for (int off = 8; off > 0; off >>= 1) qs += __shfl_xor(qs, off);
    for (int off = 32; off > 0; off >>= 1) { m = max(m, __shfl_xor(m, off)); qs = max(qs, __shfl_xor(qs, off)); }
    if ((threadIdx.x & 63) == 0) { s_m[(threadIdx.x >> 6) & 3] = m; s_q[(threadIdx.x >> 6) & 3] = qs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 1; q < (int)(blockDim.x >> 6) && q < 4; ++q) { m = max(m, s_m[q]); qs = max(qs, s_q[q]); }
        if (m > *(volatile int32_t*)max_deg) atomicMax(max_deg, m);
        if (qs > *(volatile int32_t*)qmax) atomicMax(qmax, qs);
    }
}

// Per-column ordering: rank each entry by counting smaller row indices (indices in a column are
// distinct), one wavefront per column.  Columns up to SORT_LDS entries are staged once in LDS.
// one wavefront per sorted position s: staging segment [tptr[s], tptr[s+1]) -> final column perm[s].
// Rank by counting: rank(e) = #{j : row_j < row_e} (row indices inside a column are distinct).  The compared
// index row_j is wave-uniform, so it is read with scalar loads straight from the staging array (s_load into
// SGPRs, no LDS traffic) and each comparison is one v_cmp + one v_addc.
__global__ __launch_bounds__(64) void k_sortcols(const int64_t* __restrict__ tptr, const int64_t* __restrict__ colptr,
                                                 const int32_t* __restrict__ perm, int64_t pos_begin, int64_t pos_end,
                                                 const int32_t* __restrict__ rowtmp, const double* __restrict__ valtmp,
                                                 int32_t* __restrict__ rowval, double* __restrict__ nzval)
{
    const int lane = threadIdx.x;
    for (int64_t sp = pos_begin + blockIdx.x; sp < pos_end; sp += gridDim.x) {
        const int64_t beg = __builtin_amdgcn_readfirstlane((int)(tptr[sp] & 0xffffffff)) |
                            ((int64_t)__builtin_amdgcn_readfirstlane((int)(tptr[sp] >> 32)) << 32);
        const int64_t k = tptr[sp + 1] - beg;
        if (k == 0) continue;
        const int64_t out = colptr[perm[sp]];
        const int32_t* __restrict__ col = rowtmp + beg;            // wave-uniform base
        for (int64_t e0 = 0; e0 < k; e0 += 128) {
            // two entries per lane per round: the uniform index stream is read once for both
            const int64_t ea = e0 + lane, eb = e0 + 64 + lane;
            const int32_t ma = (ea < k) ? col[ea] : 0x7fffffff;
            const int32_t mb = (eb < k) ? col[eb] : 0x7fffffff;
            int32_t ra = 0, rb = 0;
            int64_t j = 0;
            for (; j + 8 <= k; j += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int32_t v = col[j + u];
                    ra += (v < ma) ? 1 : 0;
                    rb += (v < mb) ? 1 : 0;
                }
            }
            for (; j < k; ++j) {
                const int32_t v = col[j];
                ra += (v < ma) ? 1 : 0;
                rb += (v < mb) ? 1 : 0;
            }
            if (ea < k) { rowval[out + ra] = ma; nzval[out + ra] = valtmp[beg + ea]; }
            if (eb < k) { rowval[out + rb] = mb; nzval[out + rb] = valtmp[beg + eb]; }
        }
    }
}

template <bool FILL>
static int32_t launch_rdisc(mpfmt_ctx* ctx, const rdisc_args& a, unsigned nblocks)
{
    const mpfmt_grid& G = ctx->grid;
#define CASE(DD) case DD: hipLaunchKernelGGL((k_rdisc<DD, FILL>), dim3(nblocks), dim3(64), 0, ctx->stream, a, G); break;
    switch (ctx->d) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
        default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unsupported dimension %d", ctx->d);
    }
#undef CASE
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

static void fill_args(mpfmt_ctx* ctx, double r, rdisc_args& a)
{
    a.Xt = ctx->Xt; a.perm = ctx->perm; a.cellstart = ctx->cellstart;
    a.tile_lo = ctx->tile_lo; a.tile_hi = ctx->tile_hi;
    a.r2 = r * r;
    a.rpad = r * (1.0 + 1e-9) + 1e-300;
    a.S = ctx->S;
    a.tile_begin = ctx->tile_begin;
    a.nitems = (ctx->tile_end - ctx->tile_begin) * ctx->S;
    a.npad = ctx->ntiles * 64;
    a.slice_cnt = ctx->slice_cnt;
    a.tptr = ctx->tptr;
    a.rowtmp = ctx->rowtmp; a.valtmp = ctx->valtmp;
    a.pairs = ctx->d_pairs;
}

// the library's exclusive int64 scan for the other translation units (block / single / add above): tmp holds mpfmt_scan_tmp_bytes(n)
size_t mpfmt_scan_tmp_bytes(size_t n) { return sizeof(int64_t) * ((n + 4095) / 4096 + 1); }
int32_t mpfmt_scan_i64_tmp(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n, void* tmp)
{
    if (n == 0) return MPFMT_OK;
    launch_scan<int64_t>(ctx->stream, in, out, (int64_t)n, (int64_t*)tmp);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
// ... with the temporary taken from the ctx's scratch buffer (nothing else of the caller's may live there)
int32_t mpfmt_scan_i64(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n)
{
    void* tmp;
    int32_t rc;
    if ((rc = mpfmt_scratch(ctx, mpfmt_scan_tmp_bytes(n), &tmp))) return rc;
    return mpfmt_scan_i64_tmp(ctx, in, out, n, tmp);
}
static int32_t scan_i64(mpfmt_ctx* ctx, const int64_t* in, int64_t* out, size_t n) { return mpfmt_scan_i64(ctx, in, out, n); }

// device-side verdict of a speculative step: any capacity that did not hold sets the flag every later kernel checks
__global__ void k_spec_check(const int32_t* __restrict__ pool_flag, const int32_t* __restrict__ list_max, int64_t list_cap,
                             const int64_t* __restrict__ nnz, int64_t nnz_cap, const int32_t* __restrict__ max_deg, int32_t* __restrict__ spec_fail)
{
    *spec_fail = (*pool_flag != 0) || (list_max && (int64_t)*list_max > list_cap) || (*nnz >= nnz_cap) || (*max_deg > MPFMT_ORD_MAXDEG);
}

// everything the host wants to know after a count, gathered into one block so that ONE copy into pinned memory brings it
// back (four separate copies into pageable host variables are four blocking round trips: ~100 us of idle GPU per step)
struct count_readback { unsigned long long pairs[512]; long long nnz; int pool_over, list_mx, max_deg, pad_, qmax, pad2_; };
__global__ __launch_bounds__(512) void k_count_readback(const unsigned long long* __restrict__ pairs, const int64_t* __restrict__ nnz,
                                                        const int32_t* __restrict__ pool_flag, const int32_t* __restrict__ list_max,
                                                        const int32_t* __restrict__ max_deg, count_readback* __restrict__ out,
                                                        const int32_t* __restrict__ pend_over)
{
    out->pairs[threadIdx.x] = pairs[threadIdx.x];
    if (threadIdx.x == 0) {
        out->pad_ = pend_over ? *pend_over : 0;               // (speculative step: a segment of the pending-entry list overflowed)
        out->nnz = *nnz;
        out->pool_over = pool_flag ? *pool_flag : 0;
        out->list_mx = list_max ? *list_max : 0;
        out->max_deg = *max_deg;
        out->qmax = max_deg[2];                               // (the word behind the longest column's: the fullest quarter)
    }
}

int32_t mpfmt_launch_rdisc_count(mpfmt_ctx* ctx, double r)
{
    int32_t rc;
    if ((rc = mpfmt_rdisc_count_launch(ctx, r, false))) return rc;
    return mpfmt_rdisc_count_finish(ctx, r, nullptr);
}

int32_t mpfmt_rdisc_count_launch(mpfmt_ctx* ctx, double r, bool spec)
{
    int32_t rc;
    if ((rc = mpfmt_side_join(ctx))) return rc;               // (whatever an abandoned build left on the side stream)
    ctx->masks_early = false;
    if ((rc = mpfmt_build_grid(ctx, r))) return rc;
    const int64_t N = ctx->N;
    // (the shard -- tile_begin, tile_end -- is cut by the index build: mpfmt_build_grid)
    const int64_t nt = ctx->tile_end - ctx->tile_begin;
    // the side stream pays from ~1e5 samples on: every fork and join is an event's latency (~15 us), the kernels they let run side by
    // side take 5-10 us each in a world of 2e4 samples (0.19 -> 0.22 ms per step there; cfg2, N = 1e5: 0.70 -> 0.68)
    const bool use_side = ctx->overlap > 1 || (ctx->overlap == 1 && N >= 65536);

    // The small per-build counters live in ONE arena zeroed by ONE fill: 512 pair counters + the longest column's
    // word (k_degree), the logs' overflow flag, the pending-pair list's region counters + its overflow flag.  (A ctx whose counters
    // were allocated one by one before -- by the steering spaces' builds -- keeps them and their separate fills.)
    // (... and the quarter logs' cursors behind them)
    // The fill is issued beside the chunk lists when the step forks there (a fill between two kernels of one stream costs ~15 us of
    // dispatch gaps on top of its 3), else in front of the pair kernel.
    constexpr size_t ZA_PAIRS = 0, ZA_FLAG = 4128, ZA_PCNT = 4160, ZA_OCTR = 4160 + 4112, ZA_QLEN = ZA_OCTR + 32, ZA_BYTES = ZA_QLEN;
    const size_t za_need = ZA_BYTES + sizeof(int32_t) * (size_t)std::max<int64_t>(nt * 4, 1);
    bool counters_zeroed = false;
    auto zero_counters = [&]() -> int32_t {
        if ((!ctx->zarena && !ctx->d_pairs && !ctx->pool_flag && !ctx->pair_cnt && !ctx->qlen) || (ctx->zarena && ctx->zarena_bytes < za_need)) {
            if (ctx->zarena) HIPCHK(ctx, hipFree(ctx->zarena));
            ctx->zarena = nullptr;
            HIPCHK(ctx, hipMalloc((void**)&ctx->zarena, za_need));
            ctx->zarena_bytes = za_need;
            ctx->d_pairs = (unsigned long long*)((char*)ctx->zarena + ZA_PAIRS);
            ctx->pool_flag = (int32_t*)((char*)ctx->zarena + ZA_FLAG);
            ctx->pair_cnt = (int32_t*)((char*)ctx->zarena + ZA_PCNT);
            ctx->qlen = (int32_t*)((char*)ctx->zarena + ZA_QLEN);
            ctx->ord_ctr = (int32_t*)((char*)ctx->zarena + ZA_OCTR);
        }
        if (ctx->zarena) {
            HIPCHK(ctx, hipMemsetAsync(ctx->zarena, 0, za_need, ctx->stream));
        } else {
            if (!ctx->d_pairs) HIPCHK(ctx, hipMalloc((void**)&ctx->d_pairs, 514 * sizeof(unsigned long long)));
            HIPCHK(ctx, hipMemsetAsync(ctx->d_pairs, 0, 514 * sizeof(unsigned long long), ctx->stream));
        }
        counters_zeroed = true;
        return MPFMT_OK;
    };

    // the last build of the same (N, r, shard) met a column longer than the ordering kernel stages: the logs would be written for
    // nothing (every such build ends in the fill pass), so the two-pass form is taken at once -- until a build reports a shorter
    // longest column again (k_degree measures it in every form)
    const bool hint_match = ctx->pool_hint_N == N && ctx->pool_hint_r == r && ctx->pool_hint_rank == ctx->rank && ctx->pool_hint_world == ctx->world;
    const bool too_long_hint = hint_match && ctx->pool_hint_maxdeg > MPFMT_ORD_MAXDEG;
    // path: MFMA fp16 filter + exact refine when it is usable, else the exact fp64 VALU kernel
    bool mf = false, half = false;
    float negT = 0.f;
    if (ctx->rdisc_path != 1) {
        if ((rc = mpfmt_mfma_prepare(ctx, r, &negT, &mf))) return rc;
        if (ctx->rdisc_path == 2 && !mf)
            return mpfmt_fail(ctx, MPFMT_ERR_ARG, "MFMA r-disc path requested but not usable (d > 12 or radius too small for the fp16 shell)");
    }
    // a radius below the fp16 shell (large low-dimensional worlds): the same single-pass pipeline -- chunk lists, half build, logs, fused
    // edge tests -- with the exact fp64 test itself as the filter (k_rdisc_vf_w4) instead of the two-pass exact kernel and a whole sweep
    bool vf = false;
    if (ctx->rdisc_path == 0 && !mf) {
        double ext = 0.0;
        for (int i = 0; i < ctx->d; ++i) ext = std::max(ext, ctx->bb_hi[i] - ctx->bb_lo[i]);
        vf = ctx->d <= 3 && r > 0.0 && ext > 0.0 && ctx->ntiles * 64 < ((int64_t)1 << 26) && ctx->use_pool && ctx->use_half && nt > 0 &&
             !too_long_hint && !ctx->pool_skip_once && ctx->world == 1;
    }
    ctx->filter_valu = vf;
    if (vf) mf = true;                                        // (the pair-kernel pipeline from here on)
    if (mf) {
        mpfmt_timed tm2(ctx);
        if (!vf && ctx->ops_r != ctx->grid_r) {
            if ((rc = mpfmt_mfma_build_operands(ctx))) return rc;
            ctx->ops_r = ctx->grid_r;
            ctx->lists_r = -1.0;
        }
        bool ok = true;
        // half build: the whole graph on this ctx, through the single-pass logs (decided before the lists, which differ)
        // (a shard does the same for the pairs inside it; its pairs with other shards' samples are found from its own side only)
        half = ctx->use_half && ctx->use_pool && nt > 0 && !too_long_hint && !ctx->pool_skip_once;
        // the per-sample obstacle masks of the fused broad phase need the tiles only: beside the chunk lists, on the side stream (the
        // condition is broad_in_drain's below as far as it is known here; masks made for a build that does not use them cost nothing
        // on this stream)
        if (use_side && ctx->want_broad && ctx->use_pool && nt > 0 && !too_long_hint && !ctx->pool_skip_once &&
            (ctx->d <= 6 || (ctx->d <= 12 && half && ctx->fuse_broad == 2))) {
            hipStream_t main_s;
            if ((rc = mpfmt_side_fork(ctx, &main_s))) return rc;
            // (an arena in place -- nothing is freed under the lists -- is cleared by the masks' kernel itself)
            const bool arena = ctx->zarena && ctx->zarena_bytes >= za_need;
            const int32_t rc2 = mpfmt_launch_sample_masks(ctx, r, arena ? ctx->zarena : nullptr, arena ? za_need : 0, &counters_zeroed);
            if ((rc = mpfmt_side_back(ctx, main_s)) || (rc = rc2)) return rc;
            ctx->masks_early = true;
        }
        if ((rc = mpfmt_mfma_build_lists(ctx, r, &ok, spec, half))) return rc;    // per-tile candidate chunk lists
        tm2.end("grid");
        if (!ok) {
            if (ctx->rdisc_path == 2) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "MFMA r-disc path requested but its chunk lists exceed 32 GB");
            mf = false;
        }
    }
    // (the exact VALU kernel walks cell runs directly: it needs every tile, not the shard + halo index of the matrix-core path)
    if (!mf && ctx->tileneed && (rc = mpfmt_build_grid(ctx, r, true))) return rc;
    ctx->rdisc_path_used = mf ? 2 : 1;
    ctx->mf_negT = negT;

    const int S = mpfmt_slices_for(ctx, nt, mf);                // work items = tiles x slices (one wavefront each)
    if (!mf) half = false;
    ctx->S = S;
    const int64_t npad = ctx->ntiles * 64;
    if ((rc = ensure(ctx, (void**)&ctx->slice_cnt, sizeof(int32_t) * (size_t)S * npad))) return rc;      // (per-slice counts: the two-pass forms)
    { const int64_t* was = ctx->deg; if ((rc = ensure(ctx, (void**)&ctx->deg, sizeof(int64_t) * (N + 1)))) return rc; if (ctx->deg != was) ctx->deg_zero_valid = false; }
    if ((rc = ensure(ctx, (void**)&ctx->colptr, sizeof(int64_t) * (N + 1)))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->degs, sizeof(int64_t) * (npad + 1)))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->tptr, sizeof(int64_t) * (npad + 1)))) return rc;
    if (!counters_zeroed && (rc = zero_counters())) return rc;
    const bool sparse_deg = ctx->world > 1 || nt <= 0;         // (unsharded: the degree kernels write every entry, the scans' extra last ones too)
    if (sparse_deg) {
        // a shard's degree kernel visits its own positions only: everything else must read zero.  The ordering pass puts the zeros
        // back where a step wrote (ord_args.deg_clear), so a sharded ctx pays this 8 N-byte fill once, not every step
        if (!ctx->deg_zero_valid) HIPCHK(ctx, hipMemsetAsync(ctx->deg, 0, sizeof(int64_t) * (N + 1), ctx->stream));
    }
    ctx->deg_zero_valid = false;

    // single-pass logs: ONE log per quarter tile (16 consecutive cell-sorted columns) that receives every record of its columns,
    // whoever finds the hit.  What a log must hold is the sum of 16 neighbouring columns' degrees -- a quantity with little spread
    // (no share of a tile's hits by slice or by finder enters it) -- so its capacity is the fullest quarter of the last build of the same
    // (N, r, shard) plus slack, or, on a cold ctx, the quarter of an INTERIOR column of a uniform sample set: 16 x (ball volume x
    // density) plus six standard deviations of that sum with its 16 terms taken as one.  Samples denser than uniform somewhere overflow
    // the cold estimate: that build is redone in the two-pass form, which leaves the true maximum for the next one.
    bool pool = mf && ctx->use_pool && nt > 0 && !too_long_hint && !ctx->pool_skip_once;
    ctx->pool_skip_once = false;
    double nnz_est = 0.0;
    if (pool) {
        double qwant;
        const int d = ctx->d;
        if (hint_match && ctx->pool_hint_qmax > 0) {
            const double qm = (double)ctx->pool_hint_qmax;
            qwant = qm * 1.12 + 96.0 * std::sqrt(std::max(qm / 16.0, 1.0)) + 64.0;
            // (logs that held the last build with a tenth to spare are kept as they are: growing them by a few per cent means freeing
            // and allocating gigabytes, which the allocator sometimes answers in 100+ ms)
            if ((double)ctx->qcap >= qm * 1.10 + 64.0 && (double)ctx->qcap <= qwant) qwant = (double)ctx->qcap;
            nnz_est = (double)ctx->pool_hint_nnz;
        } else {
            double vol = 1.0;
            for (int i = 0; i < d; ++i) vol *= std::max(ctx->bb_hi[i] - ctx->bb_lo[i], 1e-300);
            const double ball = std::pow(M_PI, d / 2.0) / std::tgamma(d / 2.0 + 1.0) * std::pow(r, (double)d);
            const double lam = std::min(1.0, ball / vol) * (double)N;          // neighbours of an interior sample
            // (the slack an overflow doubles belongs to this estimate only: a hint carries the true fullest quarter of the last build)
            qwant = (16.0 * lam * 1.1 + 96.0 * std::sqrt(std::max(lam, 1.0)) + 64.0) * (double)ctx->pool_slack;
            nnz_est = lam * (double)(nt * 64);
        }
        qwant = std::min(qwant, 16.0 * (double)std::min<int64_t>(N, MPFMT_ORD_MAXDEG + 1));   // (16 columns of the longest the ordering kernel takes)
        const int64_t qcap = std::max<int64_t>(64, ((int64_t)qwant + 15) / 16 * 16);
        if ((double)qcap * (double)nt * 4.0 * 12.0 > 96e9) pool = false;      // cap the logs at 96 GB of the 288
        else {
            const size_t nq = (size_t)nt * 4;
            if ((rc = ensure(ctx, (void**)&ctx->qkey, sizeof(uint32_t) * (size_t)qcap * nq))) return rc;
            if ((rc = ensure(ctx, (void**)&ctx->qd2, sizeof(double) * (size_t)qcap * nq))) return rc;
            if (!ctx->zarena) {                                    // (a ctx with separately allocated counters: their own fills)
                if ((rc = ensure(ctx, (void**)&ctx->qlen, sizeof(int32_t) * nq))) return rc;
                if (!ctx->pool_flag) HIPCHK(ctx, hipMalloc((void**)&ctx->pool_flag, sizeof(int32_t)));
                HIPCHK(ctx, hipMemsetAsync(ctx->pool_flag, 0, sizeof(int32_t), ctx->stream));
                HIPCHK(ctx, hipMemsetAsync(ctx->qlen, 0, sizeof(int32_t) * nq, ctx->stream));
            }
            ctx->qcap = qcap;
        }
    }
    if (vf && !pool) { vf = false; mf = false; half = false; ctx->filter_valu = false; ctx->rdisc_path_used = 1; }      // (no room for the logs: the exact two-pass kernel)
    if (half && !pool) {
        // (no room for the logs after all: the two-pass kernels need whole lists)
        half = false;
        bool ok = true;
        if ((rc = mpfmt_mfma_build_lists(ctx, r, &ok, spec, false))) return rc;
        if (!ok) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "chunk lists do not fit");
    }
    ctx->half_used = half;
    // the broad phase of the step's edge tests rides in any single-pass build (whole builds -- shards -- flag each entry's own record
    // and take form 1: the ordering pass lists the flagged entries); the pair form needs the half build
    // (7 <= d <= 12: only as form 2 below -- the pending-entry form's exact kernel, k_sweep_pending, is built for d <= 6)
    ctx->broad_in_drain = pool && mf && ctx->want_broad && (ctx->d <= 6 || (ctx->d <= 12 && half && ctx->fuse_broad == 2));
    ctx->bits_in_records = false; ctx->sweep_in_order = false;
    if (ctx->broad_in_drain && half && ctx->fuse_broad == 2) {
        // form 2: the pairs flagged by the drain's broad phase are listed for k_exact_pairs in 1024 dense regions (an item appends to
        // region item mod 1024 with one reservation per drain), one 16-byte item per (pair, box) unit -- up to four per pair.  Room for
        // two units per pair of the estimate (few large obstacles in a low-dimensional world flag most pairs, several times each);
        // beyond that the flag is raised, the host sweeps the whole graph and the following builds get twice the room
        const double pairs_est = 0.5 * nnz_est;
        // units per pair: the expected number of boxes an edge's box meets, from the obstacles' extents -- box k meets a segment of
        // extent ~r/2 per axis placed uniformly in the samples' bounding box with probability prod_i min(1, (w_ki + r/2) / L_i) -- times
        // 1.5 + 0.25, at most 4 (a pair that meets more writes one item): 1.4 at the north star (0.25 measured), 4 among the 30 large
        // boxes of a 2-D world of the form grid, where two units per pair overflowed
        double units = 0.0;
        if ((int64_t)ctx->boxes_host.size() == (int64_t)ctx->M * 2 * ctx->dw && ctx->dw == ctx->d) {
            for (int k = 0; k < ctx->M; ++k) {
                double p = 1.0;
                for (int i = 0; i < ctx->d; ++i) {
                    const double L = std::max(ctx->bb_hi[i] - ctx->bb_lo[i], 1e-300);
                    const double lo = std::max(ctx->boxes_host[(size_t)k * 2 * ctx->d + i], ctx->bb_lo[i] - 0.5 * r);
                    const double hi = std::min(ctx->boxes_host[(size_t)k * 2 * ctx->d + ctx->d + i], ctx->bb_hi[i] + 0.5 * r);
                    const double wdt = hi - lo;
                    p *= (wdt >= 0.0) ? std::min(1.0, (wdt + 0.5 * r) / L) : 0.0;      // (NaN bounds: counted as meeting)
                    if (!(wdt == wdt)) p = 1.0;
                }
                units += p;
            }
        } else units = 4.0;
        units = std::min(4.0, units * 1.5 + 0.25);
        ctx->pair_icap = ctx->debug_small_lists ? 8 : (int64_t)(std::min(6.0, units * (double)ctx->pair_slack) * pairs_est / 1024.0) + 4096;      // (option debug_small_lists: the overflow path, for the tests)
        if ((rc = ensure(ctx, (void**)&ctx->pair_items, 16 * (size_t)ctx->pair_icap * 1024))) return rc;
        if (!ctx->zarena) {
            if ((rc = ensure(ctx, (void**)&ctx->pair_cnt, sizeof(int32_t) * (1024 + 1)))) return rc;
            HIPCHK(ctx, hipMemsetAsync(ctx->pair_cnt, 0, sizeof(int32_t) * (1024 + 1), ctx->stream));
        }
        ctx->pair_over = ctx->pair_cnt + 1024;
        ctx->bits_in_records = true;
    }
    ctx->pool_valid = false;
    ctx->rowpos_valid = false;
    ctx->pend_valid = false;
    // (the degrees by cell-sorted position feed the staging offsets of the two-pass forms only: a single-pass build reads its own positions)
    if (sparse_deg && !pool) HIPCHK(ctx, hipMemsetAsync(ctx->degs, 0, sizeof(int64_t) * (npad + 1), ctx->stream));
    // (whatever went beside the chunk lists -- sample masks, the counters' fill -- is waited for by every form of the pair kernel)
    if (ctx->masks_early && (rc = mpfmt_side_join(ctx))) return rc;
    mpfmt_timed tm3(ctx);
    bool side_count = false;
    if (nt > 0) {
        if (mf) {
            mpfmt_timed tk(ctx);                                   // the pair kernel on its own, inside the "rdisc_count" interval
            if (pool) {
                if (!ctx->masks_early && ctx->broad_in_drain && (rc = mpfmt_launch_sample_masks(ctx, r))) return rc;
                if ((rc = mpfmt_launch_rdisc_mfma<2>(ctx, r, negT))) return rc;
                tk.end("pair_kernel");
            }
            else if ((rc = mpfmt_launch_rdisc_mfma<0>(ctx, r, negT))) return rc;
            tk.end("pair_kernel");
        } else {
            rdisc_args a;
            fill_args(ctx, r, a);
            const int64_t nblk = ((a.nitems + NXCD - 1) / NXCD) * NXCD;
            if ((rc = launch_rdisc<false>(ctx, a, (unsigned)nblk))) return rc;
        }
        const int B = 256;
        const int64_t pb = ctx->tile_begin * 64, pe = ctx->tile_end * 64;
        if (pool) {
            // single pass: the columns' degrees are counts over the logs' keys.  The flagged pairs' exact tests (form 2) MARK keys (bit 31,
            // read by the ordering pass) while the count reads other bits of them: the count and the scan of the degrees go to the side
            // stream, the tests -- the longer of the two -- stay here; joined before the first reader of colptr
            side_count = use_side && ctx->bits_in_records;
            hipStream_t main_s = nullptr;
            if (side_count && (rc = mpfmt_side_fork(ctx, &main_s))) return rc;
            int32_t rc2 = mpfmt_launch_log_degrees(ctx);
            if (side_count) {
                if (!rc2) rc2 = scan_i64(ctx, ctx->deg, ctx->colptr, (size_t)(N + 1));
                if (!rc2 && ctx->preset_entries > 0) rc2 = mpfmt_mask_preset(ctx, ctx->preset_entries);
                if ((rc = mpfmt_side_back(ctx, main_s))) return rc;
            }
            if (rc2) return rc2;
            if (ctx->bits_in_records && (rc = mpfmt_launch_exact_pairs(ctx, nullptr))) return rc;
        }
        else hipLaunchKernelGGL(k_degree, dim3((unsigned)((pe - pb + B - 1) / B)), dim3(B), 0, ctx->stream,
                                ctx->slice_cnt, ctx->perm, S, npad, pb, pe, ctx->deg, ctx->degs, (int32_t*)(ctx->d_pairs + 512),
                                (ctx->world > 1 || nt <= 0) ? (int64_t)-1 : N, (int32_t*)(ctx->d_pairs + 513));
    }
    if (!side_count && (rc = scan_i64(ctx, ctx->deg, ctx->colptr, (size_t)(N + 1)))) return rc;      // columns in original order
    // staging offsets in sorted order: only the two-pass forms read them (the single-pass build orders its logs straight into the
    // CSC) -- made here for those, and on demand by mpfmt_launch_rdisc_fill when a single-pass build has to fall back
    ctx->tptr_valid = false;
    if (!pool) { if ((rc = scan_i64(ctx, ctx->degs, ctx->tptr, (size_t)(npad + 1)))) return rc; ctx->tptr_valid = true; }
    tm3.end("rdisc_count");
    ctx->cnt_pool = pool; ctx->cnt_mf = mf;
    return MPFMT_OK;
}

// the count's read-back (nnz, pair counters, pool overflow, and -- after a speculative list build -- the list maximum)
// behind the one synchronisation.  *spec_failed (speculative callers) reports a truncated chunk list or pool overflow.
int32_t mpfmt_rdisc_count_finish(mpfmt_ctx* ctx, double r, bool* spec_failed)
{
    const int64_t N = ctx->N;
    const bool pool = ctx->cnt_pool;
    const int64_t nt = ctx->tile_end - ctx->tile_begin;
    if (spec_failed) *spec_failed = false;
    { int32_t rcj; if ((rcj = mpfmt_side_join(ctx))) return rcj; }
    if (!ctx->rb_dev) HIPCHK(ctx, hipMalloc(&ctx->rb_dev, sizeof(count_readback)));
    if (!ctx->rb_host) HIPCHK(ctx, hipHostMalloc(&ctx->rb_host, sizeof(count_readback), hipHostMallocDefault));
    hipLaunchKernelGGL(k_count_readback, dim3(1), dim3(512), 0, ctx->stream, ctx->d_pairs, ctx->colptr + N, pool ? ctx->pool_flag : nullptr,
                       (ctx->spec_lists && nt > 0) ? ctx->list_max : nullptr, (const int32_t*)(ctx->d_pairs + 512), (count_readback*)ctx->rb_dev,
                       ctx->sweep_in_order ? (const int32_t*)ctx->pair_over : (ctx->pend_valid && ctx->sweep_pending_used) ? (const int32_t*)ctx->pend_over : nullptr);
    HIPCHK(ctx, hipMemcpyAsync(ctx->rb_host, ctx->rb_dev, sizeof(count_readback), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const count_readback* rb = (const count_readback*)ctx->rb_host;
    const int64_t nnz = rb->nnz;
    ctx->pend_overflowed = rb->pad_ != 0;
    const int32_t pool_over = rb->pool_over, list_mx = rb->list_mx;
    unsigned long long pairs[512];
    memcpy(pairs, rb->pairs, sizeof pairs);
    if (ctx->spec_lists && list_mx > ctx->list_cap) {             // lists were truncated: everything after them is void
        ctx->redo_reason |= 1; ctx->redo_count += 1;
        ctx->lists_r = -1.0; ctx->lists_cap_trusted = -1;
        ctx->list_cap = std::min<int64_t>(ctx->ntiles, ((int64_t)list_mx + list_mx / 8 + 64 + 255) / 256 * 256);
        ctx->spec_lists = false;
        if (spec_failed) { *spec_failed = true; return MPFMT_OK; }
        return mpfmt_launch_rdisc_count(ctx, r);
    }
    ctx->spec_lists = false;
    ctx->max_deg = rb->max_deg;
    // the log-ordering kernel stages whole columns in LDS: a graph with a longer column takes the two-pass build
    const bool too_long = rb->max_deg > MPFMT_ORD_MAXDEG;
    if (pool && (pool_over || too_long)) {
        // a log overflowed (samples denser somewhere than the capacity assumed) or a column is too long for the ordering kernel: the
        // single pass has no per-slice counts to fill from (and a half build's lists cover half the pairs) -- the graph is counted
        // again in the two-pass form.  That count leaves the fullest quarter and the longest column in the size hint, so the next
        // build of the same (N, r, shard) is a single pass again with logs that hold (or goes straight to two passes while a column
        // stays too long); an overflow also doubles the slack of the following builds, up to 8x.
        if (pool_over && ctx->pool_slack < 8) ctx->pool_slack *= 2;
        ctx->redo_reason |= pool_over ? 2 : 4; ctx->redo_count += 1;
        ctx->pool_skip_once = true;
        ctx->lists_r = -1.0;
        ctx->graph_counted = false;
        if (spec_failed) { *spec_failed = true; return MPFMT_OK; }
        return mpfmt_launch_rdisc_count(ctx, r);
    }
    ctx->pool_valid = pool;
    if (pool) ctx->pool_slack = 1;                                // (a single pass whose logs held: the next cold estimate starts from its own margin again)
    ctx->pool_hint_N = N; ctx->pool_hint_r = r; ctx->pool_hint_nnz = nnz; ctx->pool_hint_maxdeg = rb->max_deg; ctx->pool_hint_qmax = rb->qmax;
    ctx->pool_hint_rank = ctx->rank; ctx->pool_hint_world = ctx->world;
    ctx->nnz = nnz;
    for (int i = 1; i < 256; ++i) { pairs[0] += pairs[2 * i]; pairs[1] += pairs[2 * i + 1]; }
    ctx->pairs_tested = (int64_t)pairs[0];
    ctx->survivors = (int64_t)pairs[1];
    ctx->graph_r = r;
    ctx->graph_counted = true;
    ctx->graph_filled = false;
    ctx->graph_swept = false;
    return MPFMT_OK;
}

int32_t mpfmt_launch_rdisc_fill(mpfmt_ctx* ctx, double r)
{
    if (!ctx->graph_counted || ctx->graph_r != r)
        return mpfmt_fail(ctx, MPFMT_ERR_STATE, "rdisc_fill without a matching rdisc_count");
    int32_t rc;
    const int64_t nnz = ctx->nnz;
    if (!(ctx->rdisc_path_used == 2 && ctx->pool_valid)) {        // the staging CSC is only needed by the two-pass forms
        if ((rc = ensure(ctx, (void**)&ctx->rowtmp, sizeof(int32_t) * (size_t)nnz))) return rc;
        if ((rc = ensure(ctx, (void**)&ctx->valtmp, sizeof(double) * (size_t)nnz))) return rc;
    }
    // (with the slack a following speculative step of the same (N, r) allocates for -- it then finds its arrays in place)
    const size_t nnz_alloc = (size_t)((double)nnz * 1.02) + 4096;
    if ((rc = ensure(ctx, (void**)&ctx->rowval, sizeof(int32_t) * nnz_alloc))) return rc;
    if ((rc = ensure(ctx, (void**)&ctx->nzval, sizeof(double) * nnz_alloc))) return rc;
    if (ctx->tile_end > ctx->tile_begin && nnz > 0) {
        int done = 0;
        if (ctx->rdisc_path_used == 2 && ctx->pool_valid) {
            // single pass: the hits are already in the slot lists; order each column straight into the final CSC
            mpfmt_timed tm4(ctx);
            if ((rc = mpfmt_order_logs(ctx, nullptr))) return rc;
            tm4.end("rdisc_sort");
            ctx->deg_zero_valid = ctx->world > 1;                  // (the ordering pass has put the zeros back)
            done = 1;
        }
        if (!done) {
            if (!ctx->tptr_valid) {
                if ((rc = scan_i64(ctx, ctx->degs, ctx->tptr, (size_t)(ctx->ntiles * 64 + 1)))) return rc;
                ctx->tptr_valid = true;
            }
            mpfmt_timed tm5(ctx);
            if (ctx->rdisc_path_used == 2) {
                if ((rc = mpfmt_launch_rdisc_mfma<1>(ctx, r, ctx->mf_negT))) return rc;
            } else {
                rdisc_args a;
                fill_args(ctx, r, a);
                a.pairs = nullptr;
                const int64_t nblk = ((a.nitems + NXCD - 1) / NXCD) * NXCD;
                if ((rc = launch_rdisc<true>(ctx, a, (unsigned)nblk))) return rc;
            }
            tm5.end("rdisc_fill");
            mpfmt_timed tm6(ctx);
            const int64_t pb = ctx->tile_begin * 64, pe = std::min<int64_t>(ctx->tile_end * 64, ctx->N);
            const unsigned nb = (unsigned)std::min<int64_t>(pe - pb, 1 << 20);
            hipLaunchKernelGGL(k_sortcols, dim3(nb), dim3(64), 0, ctx->stream,
                               ctx->tptr, ctx->colptr, ctx->perm, pb, pe, ctx->rowtmp, ctx->valtmp, ctx->rowval, ctx->nzval);
            HIPCHK(ctx, hipGetLastError());
            tm6.end("rdisc_sort");
        }
    }
    ctx->graph_filled = true;
    return MPFMT_OK;
}

// the sweep visited the pending-entry list only (mpfmt_launch_graph_sweep): had a segment of that list overflowed, the kernel has
// done nothing -- found out here, behind a synchronisation the careful path can afford, and answered with the whole sweep
static int32_t sweep_checked(mpfmt_ctx* ctx)
{
    int32_t rc;
    if (ctx->graph_swept && ctx->sweep_in_order) {            // form 2: the ordering pass wrote the mask out of the records' bits
        int32_t over = 0;
        HIPCHK(ctx, hipMemcpyAsync(&over, ctx->pair_over, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (!over) return MPFMT_OK;
        if (ctx->pair_slack < 8) ctx->pair_slack *= 2;
        ctx->redo_reason |= 16; ctx->redo_count += 1; ctx->pend_overflowed = true;
        ctx->sweep_in_order = false; ctx->graph_swept = false;
        return mpfmt_launch_graph_sweep(ctx);
    }
    if (ctx->graph_swept) return MPFMT_OK;
    if ((rc = mpfmt_launch_graph_sweep(ctx))) return rc;
    if (!ctx->sweep_pending_used) return MPFMT_OK;
    int32_t over = 0;
    HIPCHK(ctx, hipMemcpyAsync(&over, ctx->pend_over, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!over) return MPFMT_OK;
    ctx->pend_valid = false; ctx->graph_swept = false;
    return mpfmt_launch_graph_sweep(ctx);
}

// One whole step -- index, r-disc graph, column order, collision sweep -- with a single host synchronisation.
// mpfmt_launch_rdisc_count + _fill + mpfmt_launch_graph_sweep need the host between them (nnz sizes the CSC and the mask,
// the list maximum and the pool flag decide the path).  When the previous build of the same (N, r, shard) went through the
// single-pass path, its sizes are taken on trust instead: every launch is issued back to back, k_spec_check raises a device
// flag if a capacity does not hold (the kernels after it then return at once), and the host validates after the one
// synchronisation -- falling back to the step-by-step path when the trust was misplaced.
// The step is split in two so that ONE host thread can keep several ctxs (GPUs) busy: _launch issues the kernels of the
// speculative form without waiting (or, when nothing can be trusted yet, runs the careful form to completion), _finish makes
// the one synchronisation, validates, and redoes the step the careful way if a trusted capacity did not hold.
// can the broad phase of the step's edge tests ride in the pair kernel's drain?  (PointRobotNDBoxes in the state space's own
// coordinates, d <= 12 -- the matrix-core pair kernels --, <= 256 boxes, no sample outside the state space: the one-sided in_state_space test of
// statespaces.jl:155 is then true for every entry)
static bool step_wants_broad(mpfmt_ctx* ctx)
{
    if (!ctx->fuse_broad || ctx->cc_kind != 0 || !ctx->have_boxes || ctx->dw != ctx->d || ctx->d > 12 || ctx->M > 256) return false;
    if (mpfmt_sweep_prepare_ss(ctx) != MPFMT_OK) return false;
    return !(ctx->ss.has && !ctx->ssflag_all_in);
}

int32_t mpfmt_graph_step_launch_impl(mpfmt_ctx* ctx, double r)
{
    int32_t rc;
    const int64_t N = ctx->N;
    ctx->step_state = 0; ctx->step_r = r;
    ctx->want_broad = step_wants_broad(ctx);
    const bool spec = ctx->spec_ready && ctx->use_pool && ctx->pool_hint_N == N && ctx->pool_hint_r == r && ctx->pool_hint_rank == ctx->rank &&
                      ctx->pool_hint_world == ctx->world && ctx->pool_hint_nnz > 0 && ctx->cc_kind == 0 && ctx->have_boxes && ctx->dw == ctx->d;
    if (spec) {
        const int64_t cap = (int64_t)((double)ctx->pool_hint_nnz * 1.02) + 4096;
        ctx->preset_entries = cap; ctx->mask_preset_words = -1;
        rc = mpfmt_rdisc_count_launch(ctx, r, true);
        ctx->preset_entries = -1;
        if (rc) return rc;
        if (ctx->cnt_mf && ctx->cnt_pool) {
            if ((rc = ensure(ctx, (void**)&ctx->rowval, sizeof(int32_t) * (size_t)cap))) return rc;
            if ((rc = ensure(ctx, (void**)&ctx->nzval, sizeof(double) * (size_t)cap))) return rc;
            if (!ctx->spec_fail) HIPCHK(ctx, hipMalloc((void**)&ctx->spec_fail, sizeof(int32_t)));
            const int64_t nt = ctx->tile_end - ctx->tile_begin;
            // (behind the degree count and its scan -- colptr -- on whichever stream they ran; the join then covers the check too)
            hipLaunchKernelGGL(k_spec_check, dim3(1), dim3(1), 0, ctx->side_pending ? ctx->side_stream : ctx->stream, ctx->pool_flag,
                               (ctx->spec_lists && nt > 0) ? ctx->list_max : nullptr, ctx->list_cap, ctx->colptr + N, cap,
                               (const int32_t*)(ctx->d_pairs + 512), ctx->spec_fail);
            if (ctx->side_pending) HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->side_stream));
            ctx->nnz = ctx->pool_hint_nnz;                          // provisional: replaced by the count's own value in _finish
            ctx->nnz_cap = cap;
            ctx->pool_valid = true; ctx->rdisc_path_used = 2;
            ctx->graph_r = r; ctx->graph_counted = true;
            mpfmt_timed tm7(ctx);
            ctx->graph_swept = false;
            if ((rc = mpfmt_order_logs(ctx, ctx->spec_fail, cap))) return rc;
            tm7.end("rdisc_sort");
            ctx->graph_filled = true;
            // (form 2 of the fused edge tests: the ordering pass has written the mask already)
            if (!ctx->graph_swept && (rc = mpfmt_launch_graph_sweep(ctx, ctx->spec_fail, cap))) return rc;
            ctx->step_state = 1;                                    // speculative kernels in flight
            return MPFMT_OK;
        }
        if ((rc = mpfmt_rdisc_count_finish(ctx, r, nullptr))) return rc;
        if ((rc = mpfmt_launch_rdisc_fill(ctx, r))) return rc;
        if ((rc = sweep_checked(ctx))) return rc;
        ctx->spec_ready = ctx->rdisc_path_used == 2 && ctx->pool_valid;
        ctx->step_state = 2;
        return MPFMT_OK;
    }
    if ((rc = mpfmt_launch_rdisc_count(ctx, r))) return rc;
    if ((rc = mpfmt_launch_rdisc_fill(ctx, r))) return rc;
    if ((rc = sweep_checked(ctx))) return rc;
    ctx->spec_ready = ctx->rdisc_path_used == 2 && ctx->pool_valid;
    ctx->step_state = 2;
    return MPFMT_OK;
}

static int32_t step_finish_inner(mpfmt_ctx* ctx);
int32_t mpfmt_graph_step_finish_impl(mpfmt_ctx* ctx)
{
    const int32_t rc = step_finish_inner(ctx);
    ctx->want_broad = false;                                  // (builds outside a step never flag their records)
    return rc;
}

static int32_t step_finish_inner(mpfmt_ctx* ctx)
{
    int32_t rc;
    const double r = ctx->step_r;
    if (ctx->step_state == 0) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "graph_step_finish without graph_step_launch");
    if (ctx->step_state == 1) {
        ctx->step_state = 0;
        bool failed = false;
        if ((rc = mpfmt_rdisc_count_finish(ctx, r, &failed))) return rc;
        if (!failed && ctx->nnz < ctx->nnz_cap && ctx->pool_valid) {
            ctx->graph_filled = true; ctx->graph_swept = true;      // (finish resets the flags it owns)
            ctx->deg_zero_valid = ctx->world > 1;                   // (the ordering pass ran: the shard's degree entries are zero again)
            if (ctx->pend_overflowed) {                             // the pending-entry / pending-pair list was cut short: sweep the whole graph
                ctx->redo_reason |= 16; ctx->redo_count += 1;
                if (ctx->sweep_in_order && ctx->pair_slack < 8) ctx->pair_slack *= 2;
                ctx->pend_valid = false; ctx->sweep_in_order = false; ctx->graph_swept = false;
                return mpfmt_launch_graph_sweep(ctx);
            }
            return MPFMT_OK;
        }
        // the trust was misplaced: redo the step the careful way (capacities have been corrected by finish)
        if (!failed) { ctx->redo_reason |= 8; ctx->redo_count += 1; }       // (nnz beyond the trusted allocation)
        ctx->spec_ready = false;
        ctx->graph_counted = ctx->graph_filled = ctx->graph_swept = false;
        if ((rc = mpfmt_launch_rdisc_count(ctx, r))) return rc;
        if ((rc = mpfmt_launch_rdisc_fill(ctx, r))) return rc;
        if ((rc = sweep_checked(ctx))) return rc;
        ctx->spec_ready = ctx->rdisc_path_used == 2 && ctx->pool_valid;
        return MPFMT_OK;
    }
    ctx->step_state = 0;
    return MPFMT_OK;
}

int32_t mpfmt_graph_step(mpfmt_ctx* ctx, double r)
{
    int32_t rc;
    if ((rc = mpfmt_graph_step_launch_impl(ctx, r))) return rc;
    return mpfmt_graph_step_finish_impl(ctx);
}

// ------------------------------------------------------------------------------------------------
// single query (the MutableNNC cache-miss path, nearneighbors.jl:129-135): one workgroup scans the
// grid cells around the query, collects hits, rank-sorts them, emits 1-based indices + distances.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_rdisc_query(const double* __restrict__ Xt, const int32_t* __restrict__ perm,
                                                     const int32_t* __restrict__ cellstart, const int32_t* __restrict__ iperm,
                                                     mpfmt_grid G, int64_t v0, double r2, double rpad,
                                                     int32_t* __restrict__ hit_idx, double* __restrict__ hit_val,
                                                     int64_t* __restrict__ k_out, int64_t* __restrict__ out_idx,
                                                     double* __restrict__ out_val, int64_t cap)
{
    __shared__ int nhits;
    const int tid = threadIdx.x;
    if (tid == 0) nhits = 0;
    __syncthreads();
    const int64_t qpos = iperm[v0];
    double q[D];
    int clo[D], chi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        q[i] = Xt[((qpos >> 6) * D + i) * 64 + (qpos & 63)];
        clo[i] = cell_of(q[i] - rpad, G.lo[i], G.inv_w[i], G.g[i]);
        chi[i] = cell_of(q[i] + rpad, G.lo[i], G.inv_w[i], G.g[i]);
    }
    constexpr int L = D - 1;
    int64_t rows = 1;
#pragma unroll
    for (int i = 0; i < L; ++i) rows *= (chi[i] - clo[i] + 1);
    for (int64_t row = 0; row < rows; ++row) {
        int64_t rem = row, cbase = 0;
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            const int span = chi[i] - clo[i] + 1;
            cbase += mpfmt_cell_term(G, i, clo[i] + (int)(rem % span));
            rem /= span;
        }
        const int64_t ra = cellstart[cbase + clo[L]], rb = cellstart[cbase + chi[L] + 1];
        for (int64_t p = ra + tid; p < rb; p += 256) {
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double t = q[i] - Xt[((p >> 6) * D + i) * 64 + (p & 63)];
                const double tt = t * t;
                d2 = (i == 0) ? tt : d2 + tt;
            }
            if (d2 <= r2 && p != qpos) {
                const int slot = atomicAdd(&nhits, 1);
                hit_idx[slot] = perm[p];
                hit_val[slot] = sqrt(d2);
            }
        }
    }
    __syncthreads();
    const int64_t k = nhits;
    if (tid == 0) *k_out = k;
    __threadfence_block();
    for (int64_t e = tid; e < k; e += 256) {
        const int32_t mine = hit_idx[e];
        int64_t rank = 0;
        for (int64_t j = 0; j < k; ++j) rank += (hit_idx[j] < mine) ? 1 : 0;
        if (rank < cap) { out_idx[rank] = (int64_t)mine + 1; out_val[rank] = hit_val[e]; }
    }
}

int32_t mpfmt_launch_rdisc_query(mpfmt_ctx* ctx, int64_t v0, double r, int64_t* k_out,
                                 int64_t* inds_host, double* ds_host, int64_t cap)
{
    int32_t rc;
    if ((rc = mpfmt_build_grid(ctx, r, true))) return rc;      // (any sample may be asked for: the whole index, also on a sharded ctx)
    const int64_t N = ctx->N;
    // scratch: hit_idx[N] hit_val[N] out_idx[N] out_val[N] k
    size_t o_hi = 0, o_hv = o_hi + ((sizeof(int32_t) * N + 15) & ~(size_t)15), o_oi = o_hv + sizeof(double) * N,
           o_ov = o_oi + sizeof(int64_t) * N, o_k = o_ov + sizeof(double) * N;
    void* scr;
    if ((rc = mpfmt_scratch(ctx, o_k + 16, &scr))) return rc;
    char* s = (char*)scr;
    const double rpad = r * (1.0 + 1e-9) + 1e-300;
    const mpfmt_grid& G = ctx->grid;
#define CASE(DD) case DD: hipLaunchKernelGGL((k_rdisc_query<DD>), dim3(1), dim3(256), 0, ctx->stream, ctx->Xt, ctx->perm, \
        ctx->cellstart, ctx->iperm, G, v0, r * r, rpad, (int32_t*)(s + o_hi), (double*)(s + o_hv), (int64_t*)(s + o_k), \
        (int64_t*)(s + o_oi), (double*)(s + o_ov), N); break;
    switch (ctx->d) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
        CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15) CASE(16)
        default: return mpfmt_fail(ctx, MPFMT_ERR_ARG, "unsupported dimension %d", ctx->d);
    }
#undef CASE
    HIPCHK(ctx, hipGetLastError());
    int64_t k = 0;
    HIPCHK(ctx, hipMemcpyAsync(&k, s + o_k, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *k_out = k;
    const int64_t kk = std::min(k, cap);
    if (kk > 0 && inds_host) HIPCHK(ctx, hipMemcpy(inds_host, s + o_oi, sizeof(int64_t) * kk, hipMemcpyDeviceToHost));
    if (kk > 0 && ds_host) HIPCHK(ctx, hipMemcpy(ds_host, s + o_ov, sizeof(double) * kk, hipMemcpyDeviceToHost));
    return (k > cap) ? MPFMT_ERR_CAPACITY : MPFMT_OK;
}
