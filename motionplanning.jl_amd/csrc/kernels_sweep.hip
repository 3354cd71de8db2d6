// Segment-vs-AABB-set and point-vs-AABB-set sweeps on gfx950.
//
// Replaces the PointRobotNDBoxes checker of the reference for batches:
//   is_free_state(v, BL)      src/collisioncheckers/boxesND.jl:42-43
//   broadphase / narrow phase src/collisioncheckers/boxesND.jl:44-51 (blend: src/utilities/utils.jl:41-51)
//   is_free_motion(v, w, BL)  src/collisioncheckers/boxesND.jl:52-56
// wrapped like src/statespaces.jl:150-158 (in_state_space of the segment's first point, Identity s2w,
// collision_waypoints = (v, w) of src/statespaces/geometric.jl:20).
//
// Every predicate is a pure function of its fp64 inputs, evaluated with the reference's operations in
// the reference's order, unfused (-ffp-contract=off), IEEE division: masks are bit-exact against the
// oracle.  The reference's @any/@all short-circuits only skip work, never change a result, so the
// kernels are free to evaluate boxes in any order and to skip boxes that provably cannot matter.
//
// Design (MI355X): lane = edge (endpoints in VGPRs).  The obstacle set is staged once per workgroup in
// LDS ([box][lo(dw),hi(dw)], 16*dw bytes per box; 19.2 KB at dw=6, M=200) and reused by every wavefront
// of a persistent workgroup.  Before the per-edge loop each wavefront culls the set against the
// bounding box of ALL its 64 edges: lane l tests box 64*c+l, __ballot gives a 64-bit survivor mask,
// and the per-edge loop walks only the set bits (wave-uniform s_ff1 loop, LDS broadcast reads).  For
// graph sweeps a wavefront owns one CSC column, whose edges all live inside the r-ball of the column's
// sample, so ~7% of the boxes survive at the north-star workload.  The 64 results of a wavefront are
// one __ballot = one UInt64 chunk of a Julia BitVector (LSB = lowest edge index).
#include "mpfmt_internal.h"
#include <cstring>
#include <algorithm>

#define SWEEP_THREADS 256
// the graph sweep for d <= 6 runs 16 wavefronts per workgroup: at 4 waves per SIMD (127 VGPRs) the obstacle table (24.5 KB at
// d = 6) is staged once per CU and the 16 narrow-phase queues (7 KB each) still fit the 160 KB of LDS -- four 4-wave
// workgroups (4 x 53 KB) would not; d = 7, 8 run one 12-wavefront workgroup (3 per SIMD: their queues are larger -- 4-wave
// workgroups of 61 / 70 KB fit only twice), d > 8 has no queue and is register-bound at two
#define SWEEP_GT(D) ((D) <= 6 ? 1024 : (D) <= 8 ? 768 : 256)
#define SWEEP_LDS_BYTES (60 * 1024)

#include "sweep_predicates.h"
#include "sweep_cmpx.h"
// a global pointer read through the constant address space: a wave-uniform address then always takes the scalar cache
// (s_load into SGPRs, usable directly as the scalar operand of a vector compare) -- no alias analysis involved
typedef const __attribute__((address_space(4))) double* sweep_cptr;
__device__ __forceinline__ sweep_cptr as_const(const double* p) { return (sweep_cptr)(uintptr_t)p; }

// Wave-level cull of nb (<= SWEEP_CHUNK) staged boxes against the union box [ulo, uhi] of the
// wavefront's segments.  Survivor words stay in (wave-uniform) registers.
#define SWEEP_WORDS (SWEEP_CHUNK / 64)
template <int D>
__device__ __forceinline__ void cull_boxes(const double* sbox, int nb, const double (&ulo)[D], const double (&uhi)[D],
                                           unsigned long long (&smask)[SWEEP_WORDS], int lane)
{
#pragma unroll
    for (int c = 0; c < SWEEP_WORDS; ++c) {
        const int k = c * 64 + lane;
        const box_regs<D> b = load_box<D>(sbox, min(k, nb - 1));     // unconditional loads, no branch per term
        int out = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) out |= (int)(b.hi[i] < ulo[i]) | (int)(b.lo[i] > uhi[i]);
        smask[c] = __ballot(k < nb && !out);
    }
}

// Test one segment against the surviving staged boxes; returns "free so far".
template <int D>
__device__ __forceinline__ bool sweep_segment(const double* sbox, const unsigned long long (&smask)[SWEEP_WORDS],
                                              const double (&v)[D], const double (&w)[D], bool freeflag)
{
    double l[D], h[D];
    seg_bbox<D>(v, w, l, h);
#pragma unroll
    for (int c = 0; c < SWEEP_WORDS; ++c) {
        unsigned long long m = smask[c];
        while (m) {
            const int k = c * 64 + (__ffsll((long long)m) - 1);
            m &= m - 1;
            const box_regs<D> b = load_box<D>(sbox, k);                 // wave-uniform k: broadcast reads
            const bool pend = freeflag & !broadphase_free_sl<D>(l, h, b);
            if (__ballot(pend)) {
                if (pend) freeflag = narrow_free_sl<D>(v, w, b);
            }
        }
    }
    return freeflag;
}

// ---- points ---------------------------------------------------------------------------------------
// bit e = in_state_space(p) && all boxes: point outside box.  P: explicit points [n][D] (idx1 == null)
// or samples gathered through 1-based idx1 (or identity if both null and use_samples).
template <int D>
__global__ __launch_bounds__(SWEEP_THREADS) void k_points_free(const double* __restrict__ X, const int64_t* __restrict__ idx1,
                                                               int64_t n, const double* __restrict__ boxes, int M, int chunk,
                                                               mpfmt_ss ss, uint64_t* __restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * SWEEP_THREADS + threadIdx.x;
    const bool active = e < n;
    double v[D];
    int64_t src = 0;
    if (active) src = idx1 ? idx1[e] - 1 : e;
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = active ? X[src * D + i] : 0.0;
    bool fr = active && in_state_space_sl<D>(v, ss);
    // the boxes through the scalar cache (wave-uniform addresses): staged in LDS every box cost 2 D broadcast reads per wavefront, which
    // run at the LDS's full-width rate -- 0.22 ms for the checkpts sweep of 1e6 samples against 200 boxes, all of it LDS time
    for (int k = 0; k < M; ++k) {
        const sweep_cptr bp = as_const(boxes) + (int64_t)k * 2 * D;
        int outside = 0;                                     // @any [!(lo[i] <= v[i] <= hi[i])]
#pragma unroll
        for (int i = 0; i < D; ++i) outside |= (int)!(bp[i] <= v[i]) | (int)!(v[i] <= bp[D + i]);
        fr = fr & (outside != 0);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < n) mask[(e - lane) >> 6] = bits;
}

// ---- explicit edge lists -----------------------------------------------------------------------------
// bit e = in_state_space(v) && is_free_motion(v, w, boxes);  v/w gathered from samples via 1-based
// src1/dst1, or read from explicit arrays P/Q when src1 == null.
template <int D>
__global__ __launch_bounds__(SWEEP_THREADS) void k_edges_free(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                              const int64_t* __restrict__ dst1, const double* __restrict__ P,
                                                              const double* __restrict__ Q, int64_t E,
                                                              const double* __restrict__ boxes, int M, int chunk,
                                                              mpfmt_ss ss, uint64_t* __restrict__ mask)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sbox = (double*)smem;
    unsigned long long smask[SWEEP_WORDS];
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * SWEEP_THREADS + threadIdx.x;
    const bool active = e < E;
    double v[D], w[D];
    if (src1) {
        const int64_t s = active ? src1[e] - 1 : 0, t = active ? dst1[e] - 1 : 0;
#pragma unroll
        for (int i = 0; i < D; ++i) { v[i] = active ? X[s * D + i] : 0.0; w[i] = active ? X[t * D + i] : 0.0; }
    } else {
#pragma unroll
        for (int i = 0; i < D; ++i) { v[i] = active ? P[e * D + i] : 0.0; w[i] = active ? Q[e * D + i] : 0.0; }
    }
    bool fr = active && in_state_space_sl<D>(v, ss);
    // union box of the wavefront's active segments
    double ulo[D], uhi[D];
    {
        double l[D], h[D];
        seg_bbox<D>(v, w, l, h);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double mn = active ? l[i] : __builtin_inf(), mx = active ? h[i] : -__builtin_inf();
            for (int off = 32; off > 0; off >>= 1) {
                const double a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
                mn = (a < mn) ? a : mn;
                mx = (b > mx) ? b : mx;
            }
            ulo[i] = mn; uhi[i] = mx;
        }
    }
    for (int b0 = 0; b0 < M; b0 += chunk) {
        const int nb = min(chunk, M - b0);
        __syncthreads();
        stage_boxes<D>(sbox, boxes, b0, nb);
        __syncthreads();
        cull_boxes<D>(sbox, nb, ulo, uhi, smask, lane);
        fr = sweep_segment<D>(sbox, smask, v, w, fr);
    }
    const unsigned long long bits = __ballot(fr);
    if (lane == 0 && (e - lane) < E) mask[(e - lane) >> 6] = bits;
}

// wave-uniform lane reads (results live in SGPRs)
__device__ __forceinline__ int64_t lane_i64(int64_t v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, l);
    const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ double lane_f64(double v, int l)
{
    return __longlong_as_double(lane_i64(__double_as_longlong(v), l));
}

// ---- graph sweep ---------------------------------------------------------------------------------
// Bit e of the mask = in_state_space(V[y]) && is_free_motion(V[y], V[x]) for CSC entry e = (row y, column x).
// All rows of a column lie within rpad of V[x], so the cull box is V[x] +- rpad (~2.5 % of the boxes survive at the
// north-star workload).  Work is handed out in TASKS of SWEEP_TC (8 or 4) consecutive columns (a dynamic counter), one
// wavefront per task, 64 entries ("a round") at a time within a column:
//   - the task header -- per column its entry range and its state -- is loaded once and held in registers
//     (lane = column); per-column values are read out with v_readlane, so a column costs no memory round trip
//     of its own.  The header of the next task is requested while the current one is processed (two register sets);
//   - the dependent chain row ids -> row states is software-pipelined over rounds: row ids are requested two rounds
//     ahead, the 8*D-byte row-state gathers one round ahead, both before the current round's arithmetic.  The round
//     sequence (column, first entry) is wave-uniform scalar state that is advanced ahead of the arithmetic; an empty
//     column counts as one null round so the look-ahead never leaves the two resident headers;
//   - boxes are staged transposed (SoA, [bound][box]): the cull reads lane k = box k conflict free, the broad phase
//     reads one box by broadcast;
//   - the mask is preset to ones; a round clears the bits of its blocked entries in the two words they straddle (atomicAnd);
//   - the narrow phase is COMPACTED: only ~1 entry in 5 fails a broad phase, so running the exact slab test in the
//     round that found it would use a fifth of the lanes.  A lane instead pushes (row state, entry, column, box) to
//     its wave's LDS queue and reports the entry free for now; whenever 64 items are queued one full-width pass
//     tests them (lane = item, the column state comes from the task header by lane exchange) and clears the bits of
//     the blocked ones.  The queue is flushed before the task header is recycled;
//   - predicates are the straight-line forms on register-held boxes (see above).
#define SWEEP_QCAP 128

struct sweep_round {
    int valid, first, hs, c;          // hs: which of the two resident task headers; c: column within the task
    int64_t t, e0, end;               // task id, first entry of the round, end of the column
};

// Task header: per column its entry range and state, 2 + D eight-byte fields.  Field f of column c sits in lane
// (f % FPR) * TC + c of register f / FPR (FPR = 64 / TC fields per register): all 64 lanes of a register carry data, where a
// lane = column layout would fill TC of them and spend 2 x (2 + D) VGPR pairs on the two resident headers.
template <int D, int TC>
struct sweep_hdr {
    static constexpr int FPR = 64 / TC, NR = (2 + D + FPR - 1) / FPR;
    unsigned long long v[NR];
};
template <int D, int TC> __device__ __forceinline__ int64_t hdr_i64(const sweep_hdr<D, TC>& h, int f, int c)        // wave-uniform c
{
    constexpr int FPR = sweep_hdr<D, TC>::FPR;
    return lane_i64((int64_t)h.v[f / FPR], (f % FPR) * TC + c);
}
template <int D, int TC> __device__ __forceinline__ double hdr_f64(const sweep_hdr<D, TC>& h, int f, int c)
{
    return __longlong_as_double(hdr_i64<D, TC>(h, f, c));
}
template <int D, int TC> __device__ __forceinline__ int64_t hdr_i64_lane(const sweep_hdr<D, TC>& h, int f, int c)   // per-lane c
{
    constexpr int FPR = sweep_hdr<D, TC>::FPR;
    return (int64_t)__shfl(h.v[f / FPR], (f % FPR) * TC + c);
}

template <int D, int SWEEP_TC>
__global__ __launch_bounds__(SWEEP_GT(D), (D <= 8 ? 1 : 2)) void k_graph_sweep(const double* __restrict__ X, const int64_t* __restrict__ colptr,
                                                               const int32_t* __restrict__ rowval, int64_t N, double rpad,
                                                               const double* __restrict__ boxes, int M, int chunk,
                                                               mpfmt_ss ss, unsigned long long* __restrict__ mask,
                                                               int* __restrict__ task_ctr, const int32_t* __restrict__ perm,
                                                               int64_t sp_begin, int64_t sp_end, const int32_t* __restrict__ spec_fail,
                                                               const double* __restrict__ Xrow, const int32_t* __restrict__ rowsrc, int xcd_ranges)
{
    // Xrow / rowsrc: where row states are gathered from -- (X, rowval) = caller order, or (Xs, rowpos) = the cell-sorted copy
    // addressed by sorted position: the rows of a column are its spatial neighbours, which sit in a few contiguous runs of
    // Xs (24 full 128-byte lines per cell) instead of one line per row scattered over the caller's array.  With xcd_ranges
    // the columns are visited in cell-sorted order and every XCD (blockIdx % 8, private L2) works through its own contiguous
    // eighth of it, so the lines one wavefront pulled in serve the neighbouring columns: the gather runs out of L2 instead
    // of the fabric (tools/ubench/gather_variants.hip: 1.9e11 vs 5.0e10 rows/s).
    if (spec_fail && *spec_fail) return;                   // speculative step whose capacities did not hold: redone by the host
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sboxT = (double*)smem;                         // [2*D][SWEEP_CHUNK]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // per-wave narrow-phase queue (SoA): row state [D][QCAP], entry offset within the task, (column << 16 | box)
    double* qv = sboxT + (int64_t)SWEEP_CHUNK * 2 * D + (int64_t)wave * (D + 1) * SWEEP_QCAP;
    uint32_t* qe = (uint32_t*)(qv + (int64_t)D * SWEEP_QCAP);
    uint32_t* qk = qe + SWEEP_QCAP;
    // columns visited: every sample in caller order (perm == NULL, sp range = [0, N)), or -- on a sharded ctx -- the
    // shard's own cell-sorted positions through perm, so that a task holds SWEEP_TC non-empty columns instead of mostly
    // columns other ranks own
    const int64_t ntasks = (sp_end - sp_begin + SWEEP_TC - 1) / SWEEP_TC;
    if (blockIdx.x == 0 && threadIdx.x == 0) {               // padding bits of the last word are zero
        const int64_t nnz = colptr[N];
        if (nnz & 63) atomicAnd(&mask[nnz >> 6], (1ull << (nnz & 63)) - 1ull);
    }

    int ci = 0;
    for (int b0 = 0; b0 < M || b0 == 0; b0 += chunk, ++ci) {
        const int nb = max(0, min(chunk, M - b0));
        __syncthreads();
        for (int t = threadIdx.x; t < nb * 2 * D; t += blockDim.x) {
            const int k = t / (2 * D), i = t - k * 2 * D;
            sboxT[i * SWEEP_CHUNK + k] = boxes[(int64_t)b0 * 2 * D + t];
        }
        __syncthreads();

        int* ctr = task_ctr + ci * 8;
        int xlive = 0;                                           // wave-uniform: XCD ranges before this one (in steal order) are exhausted
        auto grab = [&]() -> int64_t {
            if (!xcd_ranges) {
                int t = 0;
                if (lane == 0) t = atomicAdd(ctr, 1);
                return (int64_t)__builtin_amdgcn_readfirstlane(t);
            }
            // own range first, then the next XCDs' (tail balance)
            const int me = (int)(blockIdx.x & 7);
            while (xlive < 8) {
                const int x = (me + xlive) & 7;
                const int64_t lo = ntasks * x / 8, hi = ntasks * (x + 1) / 8;
                int t = 0;
                if (lane == 0) t = atomicAdd(ctr + x, 1);
                t = __builtin_amdgcn_readfirstlane(t);
                if (lo + t < hi) return lo + t;
                ++xlive;
            }
            return ntasks;
        };
        sweep_hdr<D, SWEEP_TC> H0, H1;
        auto load_hdr = [&](int64_t t, sweep_hdr<D, SWEEP_TC>& h) {
            constexpr int FPR = sweep_hdr<D, SWEEP_TC>::FPR, NR = sweep_hdr<D, SWEEP_TC>::NR;
            const int c = lane % SWEEP_TC;
            const int64_t sp = sp_begin + t * SWEEP_TC + c;
            int64_t x = -1;                                       // pad positions of the sorted order hold -1
            if (sp < sp_end) x = perm ? (int64_t)perm[sp] : sp;
            const int64_t xr = x >= 0 ? x : 0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int f = q * FPR + lane / SWEEP_TC;          // this lane's field: 0 = cpb, 1 = cpe, 2 + i = w[i]
                const unsigned long long* src = (f < 2) ? reinterpret_cast<const unsigned long long*>(colptr + xr + f)
                                                        : reinterpret_cast<const unsigned long long*>(X + xr * D + min(f - 2, D - 1));
                unsigned long long val = *src;
                if (f < 2 && x < 0) val = 0;                      // empty range for a pad column
                h.v[q] = val;
            }
        };
        int64_t tset0 = grab(), tset1 = -1;
        if (tset0 >= ntasks) continue;                       // (uniform) nothing left for this wave
        load_hdr(tset0, H0);
        auto col_range = [&](const sweep_round& r, int64_t& beg, int64_t& end) {
            const int64_t b0_ = hdr_i64<D, SWEEP_TC>(H0, 0, r.c), e0_ = hdr_i64<D, SWEEP_TC>(H0, 1, r.c);
            const int64_t b1_ = hdr_i64<D, SWEEP_TC>(H1, 0, r.c), e1_ = hdr_i64<D, SWEEP_TC>(H1, 1, r.c);
            beg = r.hs ? b1_ : b0_;
            end = r.hs ? e1_ : e0_;
        };
        auto enter_col = [&](sweep_round& r) {
            int64_t beg, end;
            col_range(r, beg, end);
            r.e0 = beg; r.end = end; r.first = 1;
        };
        // next round in sequence; crossing into the other header's task reads the task id it was loaded for
        auto advance = [&](sweep_round& r) {
            if (r.e0 + 64 < r.end) { r.e0 += 64; r.first = 0; return; }
            r.c += 1;
            if (r.c >= SWEEP_TC || sp_begin + r.t * SWEEP_TC + r.c >= sp_end) {
                r.hs ^= 1;
                r.t = r.hs ? tset1 : tset0;
                r.c = 0;
                if (r.t < 0 || r.t >= ntasks) { r.valid = 0; r.e0 = r.end = 0; r.first = 0; return; }
            }
            enter_col(r);
        };
        auto request_rows = [&](const sweep_round& r) -> int32_t {
            const int64_t e = r.e0 + lane;
            return (r.valid && e < r.end) ? rowsrc[e] : 0;
        };
        auto request_states = [&](int32_t y, double (&pv)[D]) {
#pragma unroll
            for (int i = 0; i < D; ++i) pv[i] = Xrow[(int64_t)y * D + i];
        };

        // ---- narrow-phase queue ----
        int qcount = 0;                                       // wave-uniform
        auto push = [&](bool pred, const double (&v)[D], uint32_t eoff, uint32_t ck) {
            const unsigned long long m = __ballot(pred);
            if (m == 0) return;
            const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (pred) {
#pragma unroll
                for (int i = 0; i < D; ++i) qv[i * SWEEP_QCAP + pos] = v[i];
                qe[pos] = eoff; qk[pos] = ck;
            }
            qcount += __popcll(m);
        };
        // test the last n (<= 64) queued items, lane = item; hs = header set of the task they belong to
        auto drain = [&](int n, int hs) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int idx = qcount - n + lane;
            const bool on = lane < n;
            const int qi = on ? idx : 0;
            double v[D], w[D];
#pragma unroll
            for (int i = 0; i < D; ++i) v[i] = qv[i * SWEEP_QCAP + qi];
            const uint32_t eoff = qe[qi], ck = qk[qi];
            const int c = on ? (int)(ck >> 16) : 0, k = on ? (int)(ck & 0xffffu) : 0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int64_t a = hdr_i64_lane<D, SWEEP_TC>(H0, 2 + i, c), b = hdr_i64_lane<D, SWEEP_TC>(H1, 2 + i, c);
                w[i] = __longlong_as_double(hs ? b : a);
            }
            const int64_t ebase = hs ? hdr_i64_lane<D, SWEEP_TC>(H1, 0, c) : hdr_i64_lane<D, SWEEP_TC>(H0, 0, c);
            const bool free_ = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, k));
            if (on && !free_) {
                const int64_t e = ebase + (int64_t)eoff;
                atomicAnd(&mask[e >> 6], ~(1ull << (e & 63)));
            }
            qcount -= n;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };

        sweep_round R0;
        R0.valid = 1; R0.hs = 0; R0.c = 0; R0.t = tset0; R0.first = 1; R0.e0 = R0.end = 0;
        enter_col(R0);
        int32_t py0 = request_rows(R0);
        sweep_round R1 = R0;
        // the second header must be resident before the look-ahead can cross into it
        tset1 = grab();
        if (tset1 < ntasks) load_hdr(tset1, H1);
        int entered = 1;                                      // R0's task already has its successor requested
        advance(R1);
        int32_t py1 = request_rows(R1);
        double pv0[D], pv1[D];
        request_states(py0, pv0);

        double w[D], ulo[D], uhi[D];
        unsigned long long smask[SWEEP_WORDS];
#pragma unroll
        for (int i = 0; i < D; ++i) w[i] = ulo[i] = uhi[i] = 0.0;
#pragma unroll
        for (int c = 0; c < SWEEP_WORDS; ++c) smask[c] = 0;

        int hs_q = 0;                                         // header set of the task the queued items belong to
        while (true) {
            // the one place queued exact tests run (few live registers here): full passes, and a flush of the
            // remainder before the header of the task they belong to is recycled / at the end
            const bool flush = !R0.valid || (R0.first && R0.c == 0 && !entered);
            while (qcount >= 64 || (flush && qcount > 0)) drain(min(qcount, 64), hs_q);
            if (!R0.valid) break;
            // entering a task: hand its predecessor's header set to the task after it
            if (R0.first && R0.c == 0) {
                if (!entered) {
                    const int64_t tn = grab();
                    if (R0.hs) { tset0 = tn; if (tn < ntasks) load_hdr(tn, H0); }
                    else       { tset1 = tn; if (tn < ntasks) load_hdr(tn, H1); }
                }
                entered = 0;
            }
            sweep_round R2 = R1;
            advance(R2);
            const int32_t py2 = request_rows(R2);
            request_states(py1, pv1);

            if (R0.e0 < R0.end) {
                if (R0.first) {
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        const double a = hdr_f64<D, SWEEP_TC>(H0, 2 + i, R0.c), b = hdr_f64<D, SWEEP_TC>(H1, 2 + i, R0.c);
                        w[i] = R0.hs ? b : a;
                        ulo[i] = w[i] - rpad; uhi[i] = w[i] + rpad;
                    }
#pragma unroll
                    for (int c = 0; c < SWEEP_WORDS; ++c) {
                        const int k = c * 64 + lane;
                        const box_regs<D> bx = load_box_T<D>(sboxT, k);      // k < SWEEP_CHUNK: inside the staged array
                        int out = 0;
#pragma unroll
                        for (int i = 0; i < D; ++i) out |= (int)(bx.hi[i] < ulo[i]) | (int)(bx.lo[i] > uhi[i]);
                        smask[c] = __ballot(k < nb && !out);
                    }
                }
                const int64_t e0 = R0.e0, end = R0.end;
                const bool active = e0 + lane < end;
                double v[D];
#pragma unroll
                for (int i = 0; i < D; ++i) v[i] = pv0[i];
                // first chunk decides in_state_space; later chunks can only clear bits
                bool fr = active && (b0 > 0 || in_state_space_sl<D>(v, ss));
                double l[D], h[D];
                seg_bbox<D>(v, w, l, h);
                int p0 = -1, p1 = -1;                    // boxes whose broad phase this lane failed
                const uint32_t eoff = (uint32_t)(e0 + lane - (R0.hs ? hdr_i64<D, SWEEP_TC>(H1, 0, R0.c) : hdr_i64<D, SWEEP_TC>(H0, 0, R0.c)));
                hs_q = R0.hs;
#pragma unroll
                for (int c = 0; c < SWEEP_WORDS; ++c) {
                    unsigned long long m = smask[c];
                    while (m) {
                        const int k = c * 64 + (__ffsll((long long)m) - 1);
                        m &= m - 1;
                        bool pend;
                        if constexpr (D > 8) {
                            // the axes six at a time, the box through the scalar cache (sweep_cmpx.h): in R^12 few (segment, box) pairs
                            // are left after six axes, and then neither the other bounds are fetched nor their comparisons run
                            const unsigned long long pm = sweep_cmpx_groups<D>(__ballot(fr), as_const(boxes) + (int64_t)(b0 + k) * 2 * D, l, h);
                            pend = (pm >> lane) & 1ull;
                        } else pend = fr & !broadphase_free_sl<D>(l, h, load_box_T<D>(sboxT, k));      // wave-uniform k: broadcast reads
                        const bool third = pend & (p1 >= 0);
                        p1 = (pend & (p0 >= 0) & (p1 < 0)) ? k : p1;
                        p0 = (pend & (p0 < 0)) ? k : p0;
                        const unsigned long long m3 = __ballot(third);
                        if (m3) {
                            // a lane's third pending box (0.5 % of the entries, but one round in four has such a lane): queued
                            // like the other two while the p0 push below still has room; in place (a whole wave through the
                            // exact test for one or two lanes) only when the queue is that full
                            if constexpr (D <= 8) {
                                if (qcount + (int)__popcll(m3) + 64 <= SWEEP_QCAP) push(third, v, eoff, ((uint32_t)R0.c << 16) | (uint32_t)k);
                                else if (third) fr = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, k));
                            } else {
                                if (third) fr = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, k));
                            }
                        }
                    }
                }
                // queue the pending exact tests (the entry counts as free until a pass says otherwise)
                if constexpr (D <= 8) {
                    push(fr && p0 >= 0, v, eoff, ((uint32_t)R0.c << 16) | (uint32_t)max(p0, 0));
                    if (__ballot(fr && p1 >= 0)) {
                        if (qcount + 64 <= SWEEP_QCAP) push(fr && p1 >= 0, v, eoff, ((uint32_t)R0.c << 16) | (uint32_t)max(p1, 0));
                        else if (fr && p1 >= 0) fr = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, p1));   // no room: in place
                    }
                } else {
                    // d > 8: the per-wave queue would cost (d+1) KB of LDS per wavefront and halve the residency of a kernel
                    // that already needs the whole register file; the exact tests run in place
                    if (__ballot(fr && p0 >= 0)) {
                        if (fr && p0 >= 0) fr = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, p0));
                        if (__ballot(fr && p1 >= 0)) {
                            if (fr && p1 >= 0) fr = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, p1));
                        }
                    }
                }
                const unsigned long long bits = __ballot(fr);
                const int sh = (int)(e0 & 63);
                const int64_t wd = e0 >> 6;
                // the mask starts all-ones and every obstacle chunk only clears bits, so chunks (and the waves that
                // happen to claim a task in each of them) commute
                const unsigned long long clr = ~bits & __ballot(active);
                if (lane == 0 && clr) {
                    atomicAnd(&mask[wd], ~(clr << sh));
                    if (sh && (clr >> (64 - sh))) atomicAnd(&mask[wd + 1], ~(clr >> (64 - sh)));
                }
            }
            R0 = R1; R1 = R2; py1 = py2;
#pragma unroll
            for (int i = 0; i < D; ++i) pv0[i] = pv1[i];
        }
        if (M == 0) break;
    }
}

// ---- graph sweep over a ROUND TABLE ---------------------------------------------------------------
// Same bits as k_graph_sweep (d <= 8, M <= SWEEP_CHUNK), a leaner skeleton.  Stage ablation of k_graph_sweep on the north
// star (tools/ablate_sweep.sh): with the cull, the box loop and the exact tests all removed it still takes 2.29 of its 2.67 ms
// in caller order (the random 48-byte row gather at its request-rate ceiling) and 1.90 ms with rows gathered from the
// cell-sorted copy (L2 hits): the round sequencing itself -- two resident task headers read out by v_readlane, round
// descriptors advanced in scalar registers across column and task boundaries (106 SGPRs, 65 of them spilled to VGPR
// lanes) -- was the bound, not the collision arithmetic (836 instructions per round, 450 of them VALU).  Here
//   * the sequence is data: a table with one entry per 16-lane quarter of a round (column, entries, first entry), written
//     in visiting order by two trivial kernels around a scan; a wavefront loads 64 consecutive entries = 16 rounds with one
//     coalesced read (lane = quarter) and a lane fetches its quarter's entry by lane exchange.  Columns are packed end to
//     end (rounded up to quarters), so rounds are full whatever the degree is;
//   * what is wave-uniform goes through the scalar cache into SGPRs and is used as the scalar operand of the vector
//     comparisons: the obstacle box of a broad-phase iteration, the state-space bounds, the column state for the cull;
//   * the broad phase of one box is 2 d v_cmpx_*_f64 in a row that narrow EXEC, then two lane-masked updates of a packed
//     per-lane list of pending boxes: 14 VALU + 9 SALU per box where the compare / s_and chain with mask bookkeeping took
//     22 + 34 and an LDS broadcast read of the box;
//   * exact tests never run in place on the fast path (a lane or two used to occupy the whole wave for ~270 VALU).
// 420 instructions per round (270 VALU) in rounds of one column; with packed rounds the same total in 20 % fewer rounds.

// One table entry per QUARTER (16 lanes) of a round: column, number of its entries in this quarter (0..16), their first entry.
// Columns are laid end to end in visiting order, each rounded up to whole quarters, so a round of 64 lanes holds up to four
// columns (or pieces of them) and no lane is idle but the rounding (a column of 105 entries fills 7 quarters = 1.75 rounds
// where whole rounds per column took 2.19; a column of 14 fills a quarter instead of a round).
struct sweep_rd { int32_t x; uint32_t nf; int64_t e0; };          // nf = entries in the quarter (0..16)

__global__ void k_round_count(const int64_t* __restrict__ colptr, const int32_t* __restrict__ perm, int64_t sp_begin, int64_t sp_end,
                              int64_t* __restrict__ cnt, const int32_t* __restrict__ spec_fail)
{
    if (spec_fail && *spec_fail) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = sp_end - sp_begin;
    if (i > n) return;
    int64_t c = 0;
    if (i < n) {
        const int64_t x = perm ? (int64_t)perm[sp_begin + i] : sp_begin + i;       // pad positions of the sorted order hold -1
        if (x >= 0) c = (colptr[x + 1] - colptr[x] + 15) >> 4;
    }
    cnt[i] = c;                                                        // cnt[n] = 0: the scan's last element is the total
}

// One wavefront per 64 visited columns: lane = column holds (x, first entry, entries, first quarter); the wave's quarters are
// consecutive table entries, written 64 at a time (1 KB per store instruction) -- each lane finds the column of its quarter
// by a 6-step search over the lanes' first quarters (lane exchange).  (A thread per column writing its own quarters one
// after the other took 77 us for the 7e6 entries of the north star; this takes a fifth.)
__global__ __launch_bounds__(256) void k_round_fill(const int64_t* __restrict__ colptr, const int32_t* __restrict__ perm, int64_t sp_begin, int64_t sp_end,
                                                    const int64_t* __restrict__ off, sweep_rd* __restrict__ table, int64_t cap, int64_t* __restrict__ total,
                                                    const int32_t* __restrict__ spec_fail)
{
    if (spec_fail && *spec_fail) return;
    const int lane = threadIdx.x & 63;
    const int64_t n = sp_end - sp_begin;
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63;    // first column of this wavefront
    if (i0 >= n) {
        if (i0 == ((n + 63) & ~(int64_t)63) && lane == 0) {
            // the last round is padded with empty quarters; total = quarters, a multiple of 4
            int64_t q = off[n] < cap ? off[n] : cap;
            sweep_rd r; r.x = 0; r.nf = 0; r.e0 = 0;
            while ((q & 3) && q < cap) table[q++] = r;
            *total = q & ~(int64_t)3;
        }
        return;
    }
    const int64_t i = i0 + lane;
    int64_t x = -1, b = 0, len = 0, o = 0;
    if (i < n) {
        x = perm ? (int64_t)perm[sp_begin + i] : sp_begin + i;
        o = off[i];
        if (x >= 0) { b = colptr[x]; len = colptr[x + 1] - b; }
    } else {
        o = off[n];
    }
    const int64_t q0 = __shfl(o, 0);
    const int64_t qend = (i0 + 64 <= n) ? off[i0 + 64] : off[n];
    const int rel = (int)(o - q0);                                   // first quarter of this lane's column, relative (non-decreasing over lanes)
    const int tq = (int)(qend - q0);
    for (int j0 = 0; j0 < tq; j0 += 64) {
        const int j = j0 + lane;
        // the last lane whose first quarter is <= j (empty columns share their successor's first quarter: the last such lane
        // is the one that owns quarter j)
        int lo = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const int t = lo + step;
            const int rt = __shfl(rel, t & 63);
            if (t < 64 && rt <= j) lo = t;
        }
        const int64_t cx = __shfl(x, lo), cb = __shfl(b, lo), cl = __shfl(len, lo);
        const int cr = __shfl(rel, lo);
        if (j < tq && q0 + j < cap) {
            const int64_t e0 = cb + 16 * (int64_t)(j - cr);
            sweep_rd r;
            r.x = (int32_t)cx;
            r.nf = (uint32_t)(cl - 16 * (int64_t)(j - cr) < 16 ? cl - 16 * (int64_t)(j - cr) : 16);
            r.e0 = e0;
            table[q0 + j] = r;
        }
    }
}

#define RT_TASK 64                 // rounds per task = lanes of the descriptor register

template <int D>
__global__ __launch_bounds__(SWEEP_GT(D), 1) void k_graph_sweep_rt(const double* __restrict__ X, const int64_t* __restrict__ colptr, int64_t N,
                                                                     const double* __restrict__ Xrow, const int32_t* __restrict__ rowsrc,
                                                                     double rpad, const double* __restrict__ boxes, int M,
                                                                     int ss_has, const double* __restrict__ ss_bounds,
                                                                     unsigned long long* __restrict__ mask,
                                                                     const sweep_rd* __restrict__ table, const int64_t* __restrict__ total_p,
                                                                     int* __restrict__ task_ctr, const int32_t* __restrict__ spec_fail, int xcd_ranges)
{
    if (spec_fail && *spec_fail) return;                   // speculative step whose capacities did not hold: redone by the host
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sboxT = (double*)smem;                         // [2*D][SWEEP_CHUNK]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // per-wave queue of pending exact tests (SoA): row state [D][QCAP], entry (lo, hi), column, box
    double* qv = sboxT + (int64_t)SWEEP_CHUNK * 2 * D + (int64_t)wave * (D + 2) * SWEEP_QCAP;
    uint32_t* qel = (uint32_t*)(qv + (int64_t)D * SWEEP_QCAP);
    uint32_t* qeh = qel + SWEEP_QCAP;
    uint32_t* qx = qeh + SWEEP_QCAP;
    uint32_t* qk = qx + SWEEP_QCAP;
    if (blockIdx.x == 0 && threadIdx.x == 0) {               // padding bits of the last word are zero
        const int64_t nnz = colptr[N];
        if (nnz & 63) atomicAnd(&mask[nnz >> 6], (1ull << (nnz & 63)) - 1ull);
    }
    for (int t = threadIdx.x; t < M * 2 * D; t += blockDim.x) {
        const int k = t / (2 * D), i = t - k * 2 * D;
        sboxT[i * SWEEP_CHUNK + k] = boxes[t];
    }
    __syncthreads();
    const int64_t total = *total_p;
    const int64_t ntasks = (total + RT_TASK - 1) / RT_TASK;

    int xlive = 0;                                           // wave-uniform: XCD ranges before this one (in steal order) are exhausted
    auto grab = [&]() -> int64_t {
        if (!xcd_ranges) {
            int t = 0;
            if (lane == 0) t = atomicAdd(task_ctr, 1);
            return (int64_t)__builtin_amdgcn_readfirstlane(t);
        }
        const int me = (int)(blockIdx.x & 7);                // own range first, then the next XCDs' (tail balance)
        while (xlive < 8) {
            const int x = (me + xlive) & 7;
            const int64_t lo = ntasks * x / 8, hi = ntasks * (x + 1) / 8;
            int t = 0;
            if (lane == 0) t = atomicAdd(task_ctr + x, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (lo + t < hi) return lo + t;
            ++xlive;
        }
        return ntasks;
    };

    // ---- queue of pending exact tests ----
    int qcount = 0;                                          // wave-uniform
    auto push = [&](bool pred, const double (&v)[D], int64_t e, int x, int k) {
        const unsigned long long m = __ballot(pred);
        if (m == 0) return;
        const int pos = qcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (pred) {
#pragma unroll
            for (int i = 0; i < D; ++i) qv[i * SWEEP_QCAP + pos] = v[i];
            qel[pos] = (uint32_t)(uint64_t)e; qeh[pos] = (uint32_t)((uint64_t)e >> 32);
            qx[pos] = (uint32_t)x; qk[pos] = (uint32_t)k;
        }
        qcount += __popcll(m);
    };
    // exact test of the last n (<= 64) queued items, lane = item; the column state is fetched again (L2-warm)
    auto drain = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool on = lane < n;
        const int qi = on ? qcount - n + lane : 0;
        const int64_t x = (int64_t)qx[qi];
        double v[D], w[D];
#pragma unroll
        for (int i = 0; i < D; ++i) w[i] = X[x * D + i];
#pragma unroll
        for (int i = 0; i < D; ++i) v[i] = qv[i * SWEEP_QCAP + qi];
        const int64_t e = (int64_t)(((uint64_t)qeh[qi] << 32) | (uint64_t)qel[qi]);
        const int k = (int)qk[qi];
        const bool free_ = narrow_free_sl<D>(v, w, load_box_T<D>(sboxT, k));
        if (on && !free_) atomicAnd(&mask[e >> 6], ~(1ull << (e & 63)));
        qcount -= n;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // survivors of the column cull: of the column culled last (a column usually continues into the next round) and of the
    // round (union over its columns)
    unsigned long long smask[SWEEP_WORDS], slast[SWEEP_WORDS];
#pragma unroll
    for (int c = 0; c < SWEEP_WORDS; ++c) smask[c] = slast[c] = 0;
    int xlast = -1;

    for (int64_t t = grab(); t < ntasks; t = grab()) {
        const int64_t g0 = t * RT_TASK;
        const int cnt = (int)(total - g0 < RT_TASK ? total - g0 : RT_TASK) >> 2;      // rounds of this task (total is a multiple of 4)
        // lane = quarter: one coalesced read of the task's 64 quarter descriptors = 16 rounds
        int dx = 0, del = 0, deh = 0;
        if (lane < 4 * cnt) {
            const sweep_rd r = table[g0 + lane];
            // entries < 2^40: the count rides in bits 8..12 of the entry's high word (one lane exchange less per decode)
            dx = r.x; del = (int)(uint32_t)(uint64_t)r.e0; deh = (int)((uint32_t)((uint64_t)r.e0 >> 32) | (r.nf << 8));
        }
        // round k, this lane: the descriptor of its quarter (lane exchange), its entry, whether it has one
        auto lane_entry = [&](int k, int& n_, int64_t& e_) {
            const int src = 4 * k + (lane >> 4);
            const uint32_t hn = (uint32_t)__shfl(deh, src);
            n_ = (int)(hn >> 8);
            e_ = (int64_t)(((uint64_t)(hn & 0xffu) << 32) | (uint32_t)__shfl(del, src)) + (lane & 15);
        };
        // row ids two rounds ahead, row states one round ahead (the dependent chain ids -> states is pipelined over rounds;
        // beyond the task's last round the requests are null: the pipeline drains at the task boundary, 2 rounds in 16)
        auto request_rows = [&](int k) -> int32_t {
            if (k >= cnt) return 0;
            int n_; int64_t e_;
            lane_entry(k, n_, e_);
            return (lane & 15) < n_ ? rowsrc[e_] : 0;
        };
        auto request_states = [&](int32_t y, double (&pv)[D]) {
#pragma unroll
            for (int i = 0; i < D; ++i) pv[i] = Xrow[(int64_t)y * D + i];
        };
        int32_t py0 = request_rows(0);
        int32_t py1 = request_rows(1);
        double pv0[D], pv1[D];
        request_states(py0, pv0);

        for (int k = 0; k < cnt; ++k) {
            while (qcount >= 64) drain(64);                   // the one place queued exact tests run (few live registers here)
            const int32_t py2 = request_rows(k + 2);
            request_states(py1, pv1);

            int n; int64_t e;
            lane_entry(k, n, e);
            const int x = __shfl(dx, 4 * k + (lane >> 4));
            const bool active = (lane & 15) < n;
            // this lane's column state: the lanes of a quarter read the same 8 D bytes (one request)
            double w[D];
#pragma unroll
            for (int i = 0; i < D; ++i) w[i] = X[(int64_t)x * D + i];
            // (uniform) the round's columns: cull box and survivors of each column not seen in the quarter before
#pragma unroll
            for (int c = 0; c < SWEEP_WORDS; ++c) smask[c] = 0;
            int xprev = -1;
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const int xq = __builtin_amdgcn_readlane(dx, 4 * k + q);
                if (((uint32_t)__builtin_amdgcn_readlane(deh, 4 * k + q) >> 8) == 0 || xq == xprev) continue;
                xprev = xq;
                if (xq != xlast) {
                    xlast = xq;
                    const sweep_cptr xp = as_const(X) + (int64_t)xq * D;
                    double ulo[D], uhi[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) { const double wi = xp[i]; ulo[i] = wi - rpad; uhi[i] = wi + rpad; }
#pragma unroll
                    for (int c = 0; c < SWEEP_WORDS; ++c) {
                        if (c * 64 < M) {
                            const int kb = c * 64 + lane;
                            const box_regs<D> bx = load_box_T<D>(sboxT, kb);      // kb < SWEEP_CHUNK: inside the staged array
                            int out = 0;
#pragma unroll
                            for (int i = 0; i < D; ++i) out |= (int)(bx.hi[i] < ulo[i]) | (int)(bx.lo[i] > uhi[i]);
                            slast[c] = __ballot(kb < M && !out);
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < SWEEP_WORDS; ++c) smask[c] |= slast[c];
            }
            double v[D];
#pragma unroll
            for (int i = 0; i < D; ++i) v[i] = pv0[i];
            bool fr = active;
            if (ss_has) {
                // the bounds are fetched (scalar cache) where they are used: resident in SGPRs across the box loop they would
                // take 4 D of the ~100 there are
                const double* sq = ss_bounds; asm volatile("" : "+s"(sq));
                const sweep_cptr sp = as_const(sq);
                int ok = 1;
#pragma unroll
                for (int i = 0; i < D; ++i) ok &= (int)(sp[i] <= v[i]) & (int)(v[i] <= sp[MPFMT_MAX_DIM + i]);
                fr = active && ok != 0;
            }
            double l[D], h[D];
            // map(min, v, w), map(max, v, w): only compared below, where a -0 / +0 difference to the reference's ternaries does not show
#pragma unroll
            for (int i = 0; i < D; ++i) {
                asm("v_min_f64 %0, %1, %2" : "=v"(l[i]) : "v"(w[i]), "v"(v[i]));
                asm("v_max_f64 %0, %1, %2" : "=v"(h[i]) : "v"(w[i]), "v"(v[i]));
            }
            // boxes whose broad phase this lane failed: the last four as bytes of pk (box ids < 256), their number in cnt.  For
            // d <= 6 the whole test of one box is 2 d v_cmpx that narrow EXEC plus two lane-masked updates -- no scalar bookkeeping,
            // no branch in the loop but its own; lanes that are out (not active / not in the state space) fail the first comparison
            unsigned pk = 0, pc = 0;
            if constexpr (D <= 6) l[0] = fr ? l[0] : (double)INFINITY;
            [[maybe_unused]] const unsigned long long frm = __ballot(fr);
#pragma unroll
            for (int c = 0; c < SWEEP_WORDS; ++c) {
                unsigned long long m = smask[c];
                while (m) {
                    const int kb = c * 64 + (__ffsll((long long)m) - 1);
                    m &= m - 1;
                    // wave-uniform kb: the box comes through the scalar cache into SGPRs (the comparisons take it as their scalar
                    // operand); an LDS broadcast read would return 64 copies through the LDS data path
                    const sweep_cptr bp = as_const(boxes) + (int64_t)kb * 2 * D;
                    if constexpr (D <= 6) {
                        box_regs<D> bx;
#pragma unroll
                        for (int i = 0; i < D; ++i) { bx.lo[i] = bp[i]; bx.hi[i] = bp[D + i]; }
                        if constexpr (D <= 6) sweep_cmpx<D>::note(bx.lo, bx.hi, l, h, pk, pc, kb);
                        else if (fr & !broadphase_free_sl<D>(l, h, bx)) { pk = (pk << 8) | (unsigned)kb; pc += 1; }
                    } else {
                        // d > 6: the axes six at a time, the later bounds fetched only while a lane is left (sweep_cmpx.h)
                        const unsigned long long pend = sweep_cmpx_groups<D>(frm, bp, l, h);
                        if ((pend >> lane) & 1ull) { pk = (pk << 8) | (unsigned)kb; pc += 1; }
                    }
                }
            }
            if (__ballot(pc > 4)) {
                // (uniform, rare) lanes with more than four pending boxes: every surviving box tested exactly, in place
                const bool o = pc > 4;
#pragma unroll
                for (int c = 0; c < SWEEP_WORDS; ++c) {
                    unsigned long long m = smask[c];
                    while (m) {
                        const int kb = c * 64 + (__ffsll((long long)m) - 1);
                        m &= m - 1;
                        const box_regs<D> bx = load_box_T<D>(sboxT, kb);
                        if (o && fr && !broadphase_free_sl<D>(l, h, bx)) fr = narrow_free_sl<D>(v, w, bx);
                    }
                }
                if (o) pc = 0;
            }
            // queue the pending exact tests (the entry counts as free until a pass says otherwise); room for 64 before every push
#pragma unroll 1
            for (int sl = 0; sl < 4; ++sl) {
                if (!__ballot(pc > (unsigned)sl)) break;
                while (qcount > SWEEP_QCAP - 64) drain(min(qcount, 64));
                push(pc > (unsigned)sl, v, e, x, (int)((pk >> (8 * sl)) & 255u));
            }
            // the entries of a quarter are consecutive: its first lane clears the quarter's blocked bits in the word(s) they fall in
            const unsigned long long bits = __ballot(fr), act = __ballot(active);
            if ((lane & 15) == 0) {
                const unsigned clr = (unsigned)((~bits & act) >> (lane & 48)) & 0xffffu;
                if (clr) {
                    const int sh = (int)(e & 63);
                    const unsigned long long c64 = (unsigned long long)clr << sh;
                    atomicAnd(&mask[e >> 6], ~c64);
                    if (sh > 48 && (clr >> (64 - sh))) atomicAnd(&mask[(e >> 6) + 1], ~(unsigned long long)(clr >> (64 - sh)));
                }
            }
            py0 = py1; py1 = py2;
#pragma unroll
            for (int i = 0; i < D; ++i) pv0[i] = pv1[i];
        }
    }
    while (qcount > 0) drain(min(qcount, 64));
}

// ---- exact tests of the entries the pair kernel's broad phase flagged ---------------------------------------------------
// The half build's drain tests every hit's segment box against the obstacles that survive its tile's cull (symmetric: once per
// pair, kernels_rdisc_mfma.hip) and flags the records of a pair whose box meets one; k_order_logs lists the flagged entries --
// (entry, column sample, row position) -- in column order, one segment of the item array per ordering workgroup.  Everything not
// listed is free (the mask is preset to ones).  Here lane = listed entry: both states are gathered (the column's from the caller's
// array: consecutive items share it; the row's from the cell-sorted copy), the obstacle set is culled against the r-balls of the
// (few) distinct columns of the 64 items, the broad phase runs once more to find WHICH boxes (2 D v_cmpx each, per-lane packed list
// of up to four), and the slab tests run in place -- nearly every lane has one to run.  is_free_motion(V[row], V[col], CC, SS)
// with every sample inside the state space (statespaces.jl:153-158, boxesND.jl:26,44-56).
template <int D>
__global__ __launch_bounds__(256) void k_sweep_pending(const uint4* __restrict__ items, const int32_t* __restrict__ cnt, int64_t wcap,
                                                       const int32_t* __restrict__ pend_over, const double* __restrict__ X,
                                                       const double* __restrict__ Xs, double rpad, const double* __restrict__ boxes, int M,
                                                       unsigned long long* __restrict__ mask, const int64_t* __restrict__ nnz_dev,
                                                       const int32_t* __restrict__ spec_fail)
{
    if (spec_fail && *spec_fail) return;
    if (*pend_over) return;                                   // the list is incomplete: the host sweeps the whole graph instead
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sboxT = (double*)smem;                            // [2*D][SWEEP_CHUNK]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {      // padding bits of the last word are zero
        const int64_t nnz = *nnz_dev;
        if (nnz & 63) atomicAnd(&mask[nnz >> 6], (1ull << (nnz & 63)) - 1ull);
    }
    for (int t = threadIdx.x; t < M * 2 * D; t += blockDim.x) {
        const int k = t / (2 * D), i = t - k * 2 * D;
        sboxT[i * SWEEP_CHUNK + k] = boxes[t];
    }
    __syncthreads();
    const int n = cnt[blockIdx.x];
    const uint4* __restrict__ seg = items + (int64_t)blockIdx.x * wcap;
    const int wstride = (int)gridDim.y * 4;
    for (int b0 = ((int)blockIdx.y * 4 + wave) * 64; b0 < n; b0 += wstride * 64) {
        const int idx = b0 + lane;
        const bool active = idx < n;
        const uint4 it = seg[min(idx, n - 1)];
        const int64_t e = (int64_t)(((uint64_t)it.y << 32) | (uint64_t)it.x);
        const int x = (int)it.z;
        double v[D], w[D];
#pragma unroll
        for (int i = 0; i < D; ++i) w[i] = X[(int64_t)x * D + i];
#pragma unroll
        for (int i = 0; i < D; ++i) v[i] = Xs[(int64_t)it.w * D + i];
        // obstacles within reach of the columns of these 64 items (lane = box), column by column
        unsigned long long smask[SWEEP_WORDS];
#pragma unroll
        for (int c = 0; c < SWEEP_WORDS; ++c) smask[c] = 0;
        unsigned long long rem = __ballot(active);
        while (rem) {
            const int L = __builtin_ctzll(rem);
            const int xq = __builtin_amdgcn_readlane(x, L);
            const sweep_cptr xp = as_const(X) + (int64_t)xq * D;
            double ulo[D], uhi[D];
#pragma unroll
            for (int i = 0; i < D; ++i) { const double wi = xp[i]; ulo[i] = wi - rpad; uhi[i] = wi + rpad; }
#pragma unroll
            for (int c = 0; c < SWEEP_WORDS; ++c) {
                if (c * 64 < M) {
                    const int kb = c * 64 + lane;
                    const box_regs<D> bx = load_box_T<D>(sboxT, kb);
                    int out = 0;
#pragma unroll
                    for (int i = 0; i < D; ++i) out |= (int)(bx.hi[i] < ulo[i]) | (int)(bx.lo[i] > uhi[i]);
                    smask[c] |= __ballot(kb < M && !out);
                }
            }
            rem &= ~__ballot(active && x == xq);
        }
        double l[D], h[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            asm("v_min_f64 %0, %1, %2" : "=v"(l[i]) : "v"(w[i]), "v"(v[i]));
            asm("v_max_f64 %0, %1, %2" : "=v"(h[i]) : "v"(w[i]), "v"(v[i]));
        }
        unsigned pk = 0, pc = 0;
        bool fr = active;
        if constexpr (D <= 6) {
            l[0] = active ? l[0] : (double)INFINITY;
#pragma unroll
            for (int c = 0; c < SWEEP_WORDS; ++c) {
                unsigned long long mb = smask[c];
                while (mb) {
                    const int kb = c * 64 + (__ffsll((long long)mb) - 1);
                    mb &= mb - 1;
                    box_regs<D> bx;
                    const sweep_cptr bp = as_const(boxes) + (int64_t)kb * 2 * D;
#pragma unroll
                    for (int i = 0; i < D; ++i) { bx.lo[i] = bp[i]; bx.hi[i] = bp[D + i]; }
                    sweep_cmpx<D>::note(bx.lo, bx.hi, l, h, pk, pc, kb);
                }
            }
            // the slab tests of the listed boxes, most recent first; a lane with more than four takes every surviving box below
#pragma unroll 1
            for (int sl = 0; sl < 4; ++sl) {
                const bool go = fr && pc > (unsigned)sl && pc <= 4u;
                if (!__ballot(go)) continue;
                const box_regs<D> bx = load_box_T<D>(sboxT, (int)((pk >> (8 * sl)) & 255u));
                const bool f = narrow_free_sl<D>(v, w, bx);
                if (go) fr = f;
            }
        } else {
            pc = 5;
        }
        if (__ballot(fr && pc > 4u)) {
            const bool o = pc > 4u;
#pragma unroll
            for (int c = 0; c < SWEEP_WORDS; ++c) {
                unsigned long long mb = smask[c];
                while (mb) {
                    const int kb = c * 64 + (__ffsll((long long)mb) - 1);
                    mb &= mb - 1;
                    const box_regs<D> bx = load_box_T<D>(sboxT, kb);
                    if (o && fr && !broadphase_free_sl<D>(l, h, bx)) fr = narrow_free_sl<D>(v, w, bx);
                }
            }
        }
        if (active && !fr) atomicAnd(&mask[e >> 6], ~(1ull << (e & 63)));
    }
}

// ---- exact tests of the PAIRS the pair kernel's broad phase flagged, before the logs are ordered ------------------------------------
// Form 2 of the fused edge tests (option fuse_broad = 2).  The drain knows which obstacles a pair's segment box met; it lists one 16-byte
// item per (pair, box) unit: the cell-sorted positions of the pair's two ends (their quarter tiles are the logs of its two records),
// the box, and the records' places in those logs.  Here lane = unit: both states are gathered from the cell-sorted copy, the slab
// test (boxesND.jl:46-51) runs in BOTH directions -- it is not symmetric bit for bit -- and a blocked direction sets bit 31 of its
// record's key: (v = candidate -> w = query) is the query's column's entry = the own record, (query -> candidate) the other
// column's.  k_order_logs then clears the mask bits of the entries whose key carries the bit.  No gather of rows by entry, no obstacle
// cull, no box loop.
template <int D>
__global__ __launch_bounds__(256, 4) void k_exact_pairs(const uint4* __restrict__ pitems, const int32_t* __restrict__ pcnt, long long icap, int64_t nitems,
                                                     const int32_t* __restrict__ pend_over, const double* __restrict__ Xs,
                                                     const double* __restrict__ boxes, int M, uint32_t* __restrict__ qkey, long long qcap,
                                                     int64_t qbase, const int32_t* __restrict__ spec_fail)
{
    if (spec_fail && *spec_fail) return;
    if (*pend_over) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sboxT = (double*)smem;                            // [2*D][SWEEP_CHUNK]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int t = threadIdx.x; t < M * 2 * D; t += blockDim.x) {
        const int k = t / (2 * D), i = t - k * 2 * D;
        sboxT[i * SWEEP_CHUNK + k] = boxes[t];
    }
    __syncthreads();
    {
        // region blockIdx.x of the list, this workgroup's wavefronts taking every (4 gridDim.y)-th block of 64 pairs
        const int64_t it = blockIdx.x;
        const int n = (int)min((long long)pcnt[it], icap);
        const int bstep = (int)gridDim.y * 4 * 64;
        const int bfirst = ((int)blockIdx.y * 4 + wave) * 64;
        // (the next block's items are requested while this block's states are on their way: one dependent round trip per block, not two)
        uint4 inext = make_uint4(0u, 0u, 0u, 0u);
        if (bfirst < n) inext = pitems[it * icap + min(bfirst + lane, n - 1)];
        // (and a block's marks go out after the NEXT block's states are requested: the atomics' round trip runs beside the gather's instead
        // of in front of it -- the wait before the first use of a state is a wait for everything outstanding)
        long long mark0 = -1, mark1 = -1;
        for (int b0 = bfirst; b0 < n; b0 += bstep) {
            // lane = (pair, box) unit: the pair kernel writes one item per box a pair's segment box met (bit 8 of the box word: it met more
            // than four -- every box is tried).  (Items per PAIR with up to four boxes, their units laid end to end here by a scan and a
            // search among the lanes, cost 15 dependent lane exchanges per 64 units.)
            const bool on = b0 + lane < n;
            const uint4 i0 = inext;
            if (b0 + bstep < n) inext = pitems[it * icap + min(b0 + bstep + lane, n - 1)];
            const uint32_t qs = i0.x & 0x03ffffffu, jg = i0.y & 0x03ffffffu;
            const uint32_t kbw = (i0.x >> 26) | ((i0.y >> 26) << 6);
            const int kb = (int)(kbw & 255u);
            const bool all = on && (kbw & 256u) != 0;
            double q[D], c[D];
#pragma unroll
            for (int i = 0; i < D; ++i) q[i] = Xs[(int64_t)qs * D + i];
#pragma unroll
            for (int i = 0; i < D; ++i) c[i] = Xs[(int64_t)jg * D + i];
            // (a key has several writers -- one unit per box the pair met: the mark is an OR, with no return value)
            if (mark0 >= 0) atomicOr(&qkey[mark0], 0x80000000u);
            if (mark1 >= 0) atomicOr(&qkey[mark1], 0x80000000u);
            bool fwd = true, rev = true;                      // own entry: is_free_motion(c, q); foreign entry: is_free_motion(q, c)
            {
                const box_regs<D> bx = load_box_T<D>(sboxT, (on && !all) ? kb : 0);
#pragma unroll 1
                for (int dir = 0; dir < 2; ++dir) {           // (one copy of the slab test: the ends change places between the passes)
                    const bool f = narrow_free_sl<D>(c, q, bx);
                    if (on && !all) { if (dir == 0) fwd = f; else rev = f; }
#pragma unroll
                    for (int i = 0; i < D; ++i) { const double tt = c[i]; c[i] = q[i]; q[i] = tt; }
                }
            }
            if (__ballot(all)) {
                // (rare) more than four boxes met: every box, broad phase first, as the reference loops (boxesND.jl:44-51)
                double l[D], h[D];
#pragma unroll
                for (int i = 0; i < D; ++i) { l[i] = fmin(q[i], c[i]); h[i] = fmax(q[i], c[i]); }
                for (int k2 = 0; k2 < M; ++k2) {
                    const box_regs<D> bx = load_box_T<D>(sboxT, k2);
                    const bool meet = all && !broadphase_free_sl<D>(l, h, bx);
                    if (__ballot(meet)) {
#pragma unroll 1
                        for (int dir = 0; dir < 2; ++dir) {
                            const bool f = narrow_free_sl<D>(c, q, bx);
                            if (meet) { if (dir == 0) fwd = fwd && f; else rev = rev && f; }
#pragma unroll
                            for (int i = 0; i < D; ++i) { const double tt = c[i]; c[i] = q[i]; q[i] = tt; }
                        }
                    }
                }
            }
            mark0 = (on && !fwd) ? ((long long)(qs >> 4) - qbase) * qcap + (long long)i0.z : -1ll;
            mark1 = (on && !rev && i0.w != 0xffffffffu) ? ((long long)(jg >> 4) - qbase) * qcap + (long long)i0.w : -1ll;
        }
        if (mark0 >= 0) atomicOr(&qkey[mark0], 0x80000000u);
        if (mark1 >= 0) atomicOr(&qkey[mark1], 0x80000000u);
    }
}

int32_t mpfmt_launch_exact_pairs(mpfmt_ctx* ctx, const int32_t* spec_fail)
{
    const int d = ctx->d;
    const int64_t nitems = 1024;                             // regions of the list (MF_NREG in kernels_rdisc_mfma.hip)
    if (ctx->tile_end <= ctx->tile_begin || d > 12) return MPFMT_OK;
    const size_t lds = (size_t)SWEEP_CHUNK * 2 * d * sizeof(double);
    mpfmt_timed tk(ctx);
#define CASE(DD) case DD: hipLaunchKernelGGL((k_exact_pairs<DD>), dim3(1024, 4), dim3(256), lds, ctx->stream, (const uint4*)ctx->pair_items, (const int32_t*)ctx->pair_cnt, \
        (long long)ctx->pair_icap, nitems, (const int32_t*)ctx->pair_over, ctx->Xs, ctx->boxes, ctx->M, ctx->qkey, (long long)ctx->qcap, ctx->tile_begin * 4, spec_fail); break;
    switch (d) { CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) default: break; }
#undef CASE
    tk.end("exact_pairs");
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

template <int D>
static int32_t launch_sweep_pending_d(mpfmt_ctx* ctx, double rpad, const int32_t* spec_fail)
{
    const size_t lds = (size_t)SWEEP_CHUNK * 2 * D * sizeof(double);
    hipLaunchKernelGGL((k_sweep_pending<D>), dim3((unsigned)ctx->pend_nseg, 8), dim3(256), lds, ctx->stream, (const uint4*)ctx->pend_items,
                       (const int32_t*)ctx->pend_cnt, ctx->pend_wcap, (const int32_t*)ctx->pend_over, ctx->Xo, ctx->Xs, rpad, ctx->boxes, ctx->M,
                       (unsigned long long*)ctx->graph_free, ctx->colptr + ctx->N, spec_fail);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// ---- launchers -------------------------------------------------------------------------------------
static int box_chunk(int M, int D, bool culled)
{
    // boxes per LDS stage: 16*D bytes per box; culled kernels keep survivor words in registers
    int chunk = (int)((SWEEP_LDS_BYTES - 64) / (16 * D));
    chunk = std::max(64, (chunk / 64) * 64);
    if (culled) chunk = std::min(chunk, SWEEP_CHUNK);
    if (M < chunk) chunk = std::max(1, M);
    return chunk;
}
static size_t sweep_lds(int chunk, int D)
{
    return (size_t)chunk * 2 * D * sizeof(double) + 16;
}


static int32_t check_boxes(mpfmt_ctx* ctx, int d)
{
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    if (ctx->dw != d)
        return mpfmt_fail(ctx, MPFMT_ERR_ARG, "workspace dim %d != state dim %d (only Identity state2workspace on this path)", ctx->dw, d);
    return MPFMT_OK;
}

static int32_t launch_points(mpfmt_ctx* ctx, const double* X, const int64_t* idx1, int64_t n, int d, uint64_t* d_mask)
{
    if (n == 0) return MPFMT_OK;
    if (ctx->cc_kind == 1) return mpfmt_2d_launch_points(ctx, X, idx1, n, d_mask);
    const unsigned nb = (unsigned)((n + SWEEP_THREADS - 1) / SWEEP_THREADS);
    DISPATCH_D(d, hipLaunchKernelGGL((k_points_free<DD>), dim3(nb), dim3(SWEEP_THREADS), 0, ctx->stream,
                                     X, idx1, n, ctx->boxes, ctx->M, 0, ctx->ss, d_mask));
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

int32_t mpfmt_launch_points_free(mpfmt_ctx* ctx, const int64_t* d_idx1, int64_t n, uint64_t* d_mask)
{
    int32_t rc;
    if ((rc = check_boxes(ctx, ctx->d))) return rc;
    mpfmt_timed tm1(ctx);
    rc = launch_points(ctx, ctx->Xo, d_idx1, n, ctx->d, d_mask);
    tm1.end("sweep_points");
    return rc;
}

int32_t mpfmt_launch_states_free(mpfmt_ctx* ctx, const double* d_P, int64_t n, uint64_t* d_mask)
{
    int32_t rc;
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    mpfmt_timed tm2(ctx);
    rc = launch_points(ctx, d_P, nullptr, n, ctx->dw, d_mask);
    tm2.end("sweep_points");
    return rc;
}

static int32_t launch_edges(mpfmt_ctx* ctx, const int64_t* s1, const int64_t* t1, const double* P, const double* Q,
                            int64_t E, int d, uint64_t* d_mask)
{
    if (E == 0) return MPFMT_OK;
    if (ctx->cc_kind == 1) return mpfmt_2d_launch_edges(ctx, s1, t1, P, Q, E, d_mask);
    const int chunk = box_chunk(ctx->M, d, true);
    const size_t lds = sweep_lds(chunk, d);
    const unsigned nb = (unsigned)((E + SWEEP_THREADS - 1) / SWEEP_THREADS);
    DISPATCH_D(d, hipLaunchKernelGGL((k_edges_free<DD>), dim3(nb), dim3(SWEEP_THREADS), lds, ctx->stream,
                                     ctx->Xo, s1, t1, P, Q, E, ctx->boxes, ctx->M, chunk, ctx->ss, d_mask));
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

int32_t mpfmt_launch_edges_free(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, uint64_t* d_mask)
{
    int32_t rc;
    if ((rc = check_boxes(ctx, ctx->d))) return rc;
    mpfmt_timed tm3(ctx);
    rc = launch_edges(ctx, d_src1, d_dst1, nullptr, nullptr, E, ctx->d, d_mask);
    tm3.end("sweep_edges");
    return rc;
}

int32_t mpfmt_launch_motions_free(mpfmt_ctx* ctx, const double* d_P, const double* d_Q, int64_t n, uint64_t* d_mask)
{
    int32_t rc;
    if (!ctx->have_boxes) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "no obstacle set uploaded (mpfmt_upload_boxes)");
    mpfmt_timed tm4(ctx);
    rc = launch_edges(ctx, nullptr, nullptr, d_P, d_Q, n, ctx->dw, d_mask);
    tm4.end("sweep_edges");
    return rc;
}

// ---- Monte-Carlo collision probability of candidate edges (BASELINE configs[4]) ----------------------
// The reference has no implementation (README.md:9-10 cites papers only); SURVEY 8d defines the workload: per candidate
// edge, many perturbed copies of the 2-point trajectory, each swept with the segment test of boxesND.jl:44-56.  The noise
// is declared so that a scalar loop reproduces the counts bit for bit (integer sums, unfused fp64, no transcendentals):
//   rollout k of edge e, coordinate c (0..d-1 parent, d..2d-1 child): Philox4x32-10(key = seed, counter = (k, e, c, 2))
//   -> 8 halfwords, S = their sum (Irwin-Hall of 8 uniforms), z = (S - 262140) / 53509.92 (mean 0, variance 1, |z| < 4.9);
//   v' = v + sigma z.  hits[e] = rollouts with !is_free_motion(v', w', CC, SS).
// One workgroup per (edge, slice of the rollouts); lane = rollout; the boxes are culled once per workgroup against the
// edge's box grown by 4.9 sigma, so a rollout tests only the few boxes it can reach.
#define MC_SCALE (1.0 / 53509.91992145008)
#define MC_ZMAX 4.9
__device__ __forceinline__ double mc_normal(uint32_t k0, uint32_t k1, uint32_t k, uint32_t e, uint32_t c)
{
    uint32_t c0 = k, c1 = e, c2 = c, c3 = 2u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t S = (c0 & 0xffffu) + (c0 >> 16) + (c1 & 0xffffu) + (c1 >> 16) + (c2 & 0xffffu) + (c2 >> 16) + (c3 & 0xffffu) + (c3 >> 16);
    return ((double)S - 262140.0) * MC_SCALE;
}

template <int D>
__global__ __launch_bounds__(SWEEP_THREADS) void k_mc_edges(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                            const int64_t* __restrict__ dst1, double sigma, int64_t rollouts,
                                                            int64_t per_block, uint64_t seed, const double* __restrict__ boxes, int M,
                                                            int chunk, mpfmt_ss ss, unsigned long long* __restrict__ hits, int64_t e_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sbox = (double*)smem;
    __shared__ unsigned long long s_hits;
    const int lane = threadIdx.x & 63;
    const int64_t e = blockIdx.x;
    const int64_t k_begin = (int64_t)blockIdx.y * per_block, k_end = min(rollouts, k_begin + per_block);
    if (threadIdx.x == 0) s_hits = 0;
    double v0[D], w0[D], ulo[D], uhi[D];
    const int64_t s = src1[e] - 1, t = dst1[e] - 1;
    const double reach = MC_ZMAX * sigma;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        v0[i] = X[s * D + i]; w0[i] = X[t * D + i];
        ulo[i] = ((w0[i] < v0[i]) ? w0[i] : v0[i]) - reach;
        uhi[i] = ((v0[i] < w0[i]) ? w0[i] : v0[i]) + reach;
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const int nchunks = (M + chunk - 1) / max(chunk, 1);
    unsigned long long smask[SWEEP_WORDS];
#pragma unroll
    for (int c = 0; c < SWEEP_WORDS; ++c) smask[c] = 0;
    if (nchunks == 1) {                                   // the common case: the whole obstacle set staged and culled once
        __syncthreads();
        stage_boxes<D>(sbox, boxes, 0, M);
        __syncthreads();
        cull_boxes<D>(sbox, M, ulo, uhi, smask, lane);
    }
    unsigned long long mine = 0;
    for (int64_t kb = k_begin; kb < k_end; kb += SWEEP_THREADS) {        // uniform trip count: barriers inside are safe
        const int64_t k = kb + threadIdx.x;
        const bool act = k < k_end;
        double v[D], w[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double pv = sigma * mc_normal(k0, k1, (uint32_t)k, (uint32_t)(e_off + e), (uint32_t)i);
            const double pw = sigma * mc_normal(k0, k1, (uint32_t)k, (uint32_t)(e_off + e), (uint32_t)(D + i));
            v[i] = v0[i] + pv; w[i] = w0[i] + pw;
        }
        bool fr = act && in_state_space_sl<D>(v, ss);
        if (nchunks <= 1) {
            if (M > 0) fr = sweep_segment<D>(sbox, smask, v, w, fr);
        } else {
            for (int b0 = 0; b0 < M; b0 += chunk) {
                const int nb = min(chunk, M - b0);
                __syncthreads();
                stage_boxes<D>(sbox, boxes, b0, nb);
                __syncthreads();
                cull_boxes<D>(sbox, nb, ulo, uhi, smask, lane);
                fr = sweep_segment<D>(sbox, smask, v, w, fr);
            }
        }
        mine += (unsigned long long)__popcll(__ballot(act && !fr));
    }
    if (lane == 0 && mine) atomicAdd(&s_hits, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_hits) atomicAdd(&hits[e], s_hits);
}

int32_t mpfmt_launch_mc_edges(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double sigma, int64_t rollouts,
                              uint64_t seed, unsigned long long* d_hits)
{
    int32_t rc;
    if ((rc = check_boxes(ctx, ctx->d))) return rc;
    if (ctx->cc_kind != 0) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the Monte-Carlo evaluator runs against the AABB checker");
    if (E == 0 || rollouts == 0) return MPFMT_OK;
    const int d = ctx->d;
    const int chunk = box_chunk(ctx->M, d, true);
    const size_t lds = sweep_lds(std::max(chunk, 1), d);
    // enough workgroups to fill the GPU even for a single edge, at least 4 batches of rollouts per workgroup
    int64_t slices = std::max<int64_t>(1, std::min<int64_t>((rollouts + 4 * SWEEP_THREADS - 1) / (4 * SWEEP_THREADS),
                                                            std::max<int64_t>(1, (int64_t)ctx->num_cus * 8 / std::max<int64_t>(E, 1))));
    slices = std::min<int64_t>(slices, 65535);
    const int64_t per_block = ((rollouts + slices - 1) / slices + SWEEP_THREADS - 1) / SWEEP_THREADS * SWEEP_THREADS;
    slices = (rollouts + per_block - 1) / per_block;
    mpfmt_timed tm5(ctx);
    for (int64_t e0 = 0; e0 < E; e0 += 1 << 20) {                    // gridDim.x limit
        const int64_t ne = std::min<int64_t>(E - e0, 1 << 20);
        DISPATCH_D(d, hipLaunchKernelGGL((k_mc_edges<DD>), dim3((unsigned)ne, (unsigned)slices), dim3(SWEEP_THREADS), lds, ctx->stream,
                                         ctx->Xo, d_src1 + e0, d_dst1 + e0, sigma, rollouts, per_block, seed, ctx->boxes, ctx->M,
                                         std::max(chunk, 1), ctx->ss, d_hits + e0, e0));
    }
    HIPCHK(ctx, hipGetLastError());
    tm5.end("mc_edges");
    return MPFMT_OK;
}

// ---- importance-sampling estimator of the same probability (the approach of the papers README.md:9-10 cites) -------------------------
// Rollouts come from a MIXTURE of the nominal noise and the noise shifted towards the closest obstacle points; a colliding rollout
// counts with its likelihood ratio.  Declared so that a scalar loop reproduces the sums bit for bit (tests/):
//   closest points: for every box k and the five points p_t = v + t (w - v), t = 0, 1/4, 1/2, 3/4, 1, of the nominal segment,
//     c = clamp(p_t, lo_k, hi_k) (closest(p, BB, I) of boxesND.jl:61-86 with W = I) and d2 = sum_i (c_i - p_t,i)^2; the box keeps its
//     smallest d2 (first minimum over t); the K = min(3, M) boxes with the smallest d2 (first minima over k) give the shifts, in noise
//     units and the same for both end points: s_j,i = clip((c_i - p_t,i) / sigma, -3, 3);
//   rollout k: noise z as above;  Philox(key = seed, counter = (k, e, 2 d, 3)) -> words x0, x1: nominal when x0 & 1 == 0, else shift
//     j = x1 mod K;  y = z (+ s_j);  v' = v + sigma y_v, w' = w + sigma y_w;  hit = !is_free_motion(v', w', CC, SS);
//   weight = f(y) / (0.5 f(y) + sum_j (0.5 / K) f(y - s_j)),  f = product over the 2 d coordinates of the Irwin-Hall(8) density g at
//     x = (y * 53509.92 + 262140) / 65536,  g(x) = (1 / 5040) sum_{k<4} (-1)^k C(8, k) max(t - k, 0)^7 at t = min(x, 8 - x);
//   weights are quantised to 2^-40 and summed as integers (wsum[e]): the order of the sum does not matter.
#define MC_INV 53509.91992145008
#define MC_IS_K 3
__device__ __forceinline__ double mc_ih8_pdf(double x)
{
    const double t = (x < 8.0 - x) ? x : 8.0 - x;
    if (!(t > 0.0)) return 0.0;
    double acc = 0.0;
    const double cf[4] = {1.0, -8.0, 28.0, -56.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double u = t - (double)k;
        if (u > 0.0) {
            const double u2 = u * u, u4 = u2 * u2, u3 = u2 * u;
            const double u7 = u4 * u3;
            const double term = cf[k] * u7;
            acc = acc + term;
        }
    }
    return acc * (1.0 / 5040.0);
}

template <int D>
__global__ __launch_bounds__(SWEEP_THREADS) void k_mc_is_edges(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                               const int64_t* __restrict__ dst1, double sigma, int64_t rollouts,
                                                               int64_t per_block, uint64_t seed, const double* __restrict__ boxes, int M,
                                                               mpfmt_ss ss, unsigned long long* __restrict__ wsum, int64_t e_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sbox = (double*)smem;
    __shared__ unsigned long long s_sum;
    __shared__ double s_d2[SWEEP_CHUNK];
    __shared__ int s_bt[SWEEP_CHUNK], s_ch[MC_IS_K];
    const int lane = threadIdx.x & 63;
    const int64_t e = blockIdx.x;
    const int64_t k_begin = (int64_t)blockIdx.y * per_block, k_end = min(rollouts, k_begin + per_block);
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    stage_boxes<D>(sbox, boxes, 0, M);
    __syncthreads();
    double v0[D], w0[D];
    const int64_t s = src1[e] - 1, t = dst1[e] - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) { v0[i] = X[s * D + i]; w0[i] = X[t * D + i]; }
    // every box's closest approach to the five points of the nominal segment (thread = box), then the K closest boxes (thread 0)
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const double* lo = sbox + (int64_t)k * 2 * D; const double* hi = lo + D;
        double best = 0.0; int tb = -1;
        for (int tq = 0; tq < 5; ++tq) {
            const double tt = 0.25 * (double)tq;
            double d2 = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double df = w0[i] - v0[i], pr = tt * df;
                const double p = v0[i] + pr;
                const double c = (p < lo[i]) ? lo[i] : ((p > hi[i]) ? hi[i] : p);
                const double g = c - p, gg = g * g;
                d2 = (i == 0) ? gg : d2 + gg;
            }
            if (tb < 0 || d2 < best) { best = d2; tb = tq; }
        }
        s_d2[k] = best; s_bt[k] = tb;
    }
    __syncthreads();
    const int K = (M < MC_IS_K) ? M : MC_IS_K;
    if (threadIdx.x == 0) {
        for (int j = 0; j < K; ++j) {
            int kb = -1;
            for (int k = 0; k < M; ++k) {
                bool taken = false;
                for (int q = 0; q < j; ++q) taken = taken || (s_ch[q] == k);
                if (taken) continue;
                if (kb < 0 || s_d2[k] < s_d2[kb]) kb = k;
            }
            s_ch[j] = kb;
        }
    }
    __syncthreads();
    double sj[MC_IS_K][D];
#pragma unroll
    for (int j = 0; j < MC_IS_K; ++j) {
#pragma unroll
        for (int i = 0; i < D; ++i) sj[j][i] = 0.0;
        if (j < K) {
            const int kb = s_ch[j];
            const double* lo = sbox + (int64_t)kb * 2 * D; const double* hi = lo + D;
            const double tt = 0.25 * (double)s_bt[kb];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double df = w0[i] - v0[i], pr = tt * df;
                const double p = v0[i] + pr;
                const double c = (p < lo[i]) ? lo[i] : ((p > hi[i]) ? hi[i] : p);
                double q = 0.0;
                if (sigma > 0.0) {
                    q = (c - p) / sigma;
                    q = (q < -3.0) ? -3.0 : ((q > 3.0) ? 3.0 : q);
                }
                sj[j][i] = q;
            }
        }
    }
    const double cj = (K > 0) ? 0.5 / (double)K : 0.0;
    double ulo[D], uhi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double mx = 0.0;
#pragma unroll
        for (int j = 0; j < MC_IS_K; ++j) mx = fmax(mx, fabs(sj[j][i]));
        const double reach = (MC_ZMAX + mx) * sigma;
        ulo[i] = ((w0[i] < v0[i]) ? w0[i] : v0[i]) - reach;
        uhi[i] = ((v0[i] < w0[i]) ? w0[i] : v0[i]) + reach;
    }
    unsigned long long smask[SWEEP_WORDS];
    cull_boxes<D>(sbox, M, ulo, uhi, smask, lane);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    unsigned long long mine = 0;
    for (int64_t kb = k_begin; kb < k_end; kb += SWEEP_THREADS) {
        const int64_t k = kb + threadIdx.x;
        const bool act = k < k_end;
        // component: Philox(key = seed, counter = (k, e, 2 D, 3)) words 0, 1
        uint32_t c0 = (uint32_t)k, c1 = (uint32_t)(e_off + e), c2 = (uint32_t)(2 * D), c3 = 3u, q0 = k0, q1 = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ q0, n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ q1, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            q0 += 0x9E3779B9u; q1 += 0xBB67AE85u;
        }
        const int comp = (K > 0 && (c0 & 1u)) ? (int)(c1 % (uint32_t)max(K, 1)) : -1;
        double v[D], w[D], yv[D], yw[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double zv = mc_normal(k0, k1, (uint32_t)k, (uint32_t)(e_off + e), (uint32_t)i);
            const double zw = mc_normal(k0, k1, (uint32_t)k, (uint32_t)(e_off + e), (uint32_t)(D + i));
            const double shf = (comp == 0) ? sj[0][i] : (comp == 1) ? sj[1][i] : (comp == 2) ? sj[2][i] : 0.0;
            yv[i] = zv + shf; yw[i] = zw + shf;
            const double pv = sigma * yv[i], pw = sigma * yw[i];
            v[i] = v0[i] + pv; w[i] = w0[i] + pw;
        }
        bool fr = act && in_state_space_sl<D>(v, ss);
        if (M > 0) fr = sweep_segment<D>(sbox, smask, v, w, fr);
        if (act && !fr) {
            double a = 1.0;
#pragma unroll
            for (int c = 0; c < 2 * D; ++c) {
                const double yc = (c < D) ? yv[c < D ? c : 0] : yw[c < D ? 0 : c - D];
                const double xa = (yc * MC_INV + 262140.0) * (1.0 / 65536.0);
                a = a * mc_ih8_pdf(xa);
            }
            double den = 0.5 * a;
#pragma unroll
            for (int j = 0; j < MC_IS_K; ++j) {
                if (j < K) {
                    double b = 1.0;
#pragma unroll
                    for (int c = 0; c < 2 * D; ++c) {
                        const double yc = (c < D) ? yv[c < D ? c : 0] : yw[c < D ? 0 : c - D];
                        const double yb = yc - sj[j][c < D ? c : c - D];
                        const double xb = (yb * MC_INV + 262140.0) * (1.0 / 65536.0);
                        b = b * mc_ih8_pdf(xb);
                    }
                    const double tb = cj * b;
                    den = den + tb;
                }
            }
            if (K == 0) den = a;
            const double wgt = (den > 0.0) ? a / den : 0.0;
            mine += (unsigned long long)(wgt * 1099511627776.0);
        }
    }
    // integer sum: lanes -> wavefront -> workgroup -> edge
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(&s_sum, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(&wsum[e], s_sum);
}

// ---- ADAPTIVE importance sampling (BASELINE configs[4]): a pilot estimates, per edge, the mean of the nominal noise
// GIVEN a collision (a cross-entropy update of the proposal's mean), the main run samples from the mixture 0.5 nominal + 0.5 nominal
// shifted by that mean.  include/mpfmt.h (mpfmt_mc_edges_collision_ais) spells the arithmetic out; all sums are integers, so a scalar
// loop reproduces the shifts and the weight sums exactly.
#define MC_AIS_NP 4096
#define MC_AIS_ALPHA 1.625
__device__ __forceinline__ int mc_normal_int(uint32_t k0, uint32_t k1, uint32_t k, uint32_t e, uint32_t c)
{
    uint32_t c0 = k, c1 = e, c2 = c, c3 = 2u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t S = (c0 & 0xffffu) + (c0 >> 16) + (c1 & 0xffffu) + (c1 >> 16) + (c2 & 0xffffu) + (c2 >> 16) + (c3 & 0xffffu) + (c3 >> 16);
    return (int)S - 262140;
}

// pilot: one workgroup per edge, thread = pilot rollout (16 each); integer sums of the hits' quantised likelihood ratios and of
// their products with the noise integers; threads c < 2 D turn them into the shift mu[e][c]
template <int D>
__global__ __launch_bounds__(SWEEP_THREADS) void k_mc_ais_pilot(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                                const int64_t* __restrict__ dst1, double sigma, uint64_t seed,
                                                                const double* __restrict__ boxes, int M, mpfmt_ss ss,
                                                                double* __restrict__ mu, int64_t e_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sbox = (double*)smem;
    __shared__ unsigned long long s_sw;
    __shared__ long long s_a[2 * D];
    const int lane = threadIdx.x & 63;
    const int64_t e = blockIdx.x;
    if (threadIdx.x == 0) s_sw = 0;
    if (threadIdx.x < 2 * D) s_a[threadIdx.x] = 0;
    __syncthreads();
    stage_boxes<D>(sbox, boxes, 0, M);
    __syncthreads();
    double v0[D], w0[D];
    const int64_t s = src1[e] - 1, t = dst1[e] - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) { v0[i] = X[s * D + i]; w0[i] = X[t * D + i]; }
    double ulo[D], uhi[D];
    const double reach = (MC_ZMAX * MC_AIS_ALPHA) * sigma;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        ulo[i] = ((w0[i] < v0[i]) ? w0[i] : v0[i]) - reach;
        uhi[i] = ((v0[i] < w0[i]) ? w0[i] : v0[i]) + reach;
    }
    unsigned long long smask[SWEEP_WORDS];
    cull_boxes<D>(sbox, M, ulo, uhi, smask, lane);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    unsigned long long sw = 0;
    long long acc[2 * D];
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) acc[c] = 0;
    for (int k = threadIdx.x; k < MC_AIS_NP; k += SWEEP_THREADS) {
        int Z[2 * D];
        double y[2 * D];
        double num = 1.0, den = 1.0;
#pragma unroll
        for (int c = 0; c < 2 * D; ++c) {
            Z[c] = mc_normal_int(k0, k1, (uint32_t)k, (uint32_t)(e_off + e), (uint32_t)(64 + c));
            const double z = (double)Z[c] * MC_SCALE;
            y[c] = MC_AIS_ALPHA * z;
            const double xy = (y[c] * MC_INV + 262140.0) * (1.0 / 65536.0), xz = (z * MC_INV + 262140.0) * (1.0 / 65536.0);
            num = num * mc_ih8_pdf(xy); den = den * mc_ih8_pdf(xz);
        }
        double v[D], w[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double pv = sigma * y[i], pw = sigma * y[D + i];
            v[i] = v0[i] + pv; w[i] = w0[i] + pw;
        }
        bool fr = in_state_space_sl<D>(v, ss);
        if (M > 0) fr = sweep_segment<D>(sbox, smask, v, w, fr);
        if (!fr) {
            const double lr = (den > 0.0) ? num / den : 0.0;
            const unsigned long long Wq = (unsigned long long)(lr * 1073741824.0);
            sw += Wq;
#pragma unroll
            for (int c = 0; c < 2 * D; ++c) acc[c] += (long long)Wq * (long long)Z[c];
        }
    }
    for (int off = 32; off > 0; off >>= 1) sw += __shfl_xor(sw, off);
    if (lane == 0 && sw) atomicAdd(&s_sw, sw);
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) {
        long long a = acc[c];
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
        if (lane == 0 && a) atomicAdd((unsigned long long*)&s_a[c], (unsigned long long)a);
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        double m = 0.0;
        const unsigned long long SW = s_sw;
        if (SW > 0) {
            const double tt = (double)s_a[threadIdx.x] * MC_SCALE;
            m = MC_AIS_ALPHA * tt / (double)SW;
            m = (m < -3.0) ? -3.0 : ((m > 3.0) ? 3.0 : m);
        }
        mu[(e_off + e) * (2 * MPFMT_MAX_DIM) + threadIdx.x] = m;
    }
}

template <int D>
__global__ __launch_bounds__(SWEEP_THREADS) void k_mc_ais_edges(const double* __restrict__ X, const int64_t* __restrict__ src1,
                                                                const int64_t* __restrict__ dst1, double sigma, int64_t rollouts,
                                                                int64_t per_block, uint64_t seed, const double* __restrict__ boxes, int M,
                                                                mpfmt_ss ss, const double* __restrict__ mu_all,
                                                                unsigned long long* __restrict__ wsum, int64_t e_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sbox = (double*)smem;
    __shared__ unsigned long long s_sum;
    const int lane = threadIdx.x & 63;
    const int64_t e = blockIdx.x;
    const int64_t k_begin = (int64_t)blockIdx.y * per_block, k_end = min(rollouts, k_begin + per_block);
    if (threadIdx.x == 0) s_sum = 0;
    __syncthreads();
    stage_boxes<D>(sbox, boxes, 0, M);
    __syncthreads();
    double v0[D], w0[D], mu[2 * D];
    const int64_t s = src1[e] - 1, t = dst1[e] - 1;
#pragma unroll
    for (int i = 0; i < D; ++i) { v0[i] = X[s * D + i]; w0[i] = X[t * D + i]; }
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) mu[c] = mu_all[(e_off + e) * (2 * MPFMT_MAX_DIM) + c];
    double ulo[D], uhi[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double mx = fmax(fabs(mu[i]), fabs(mu[D + i]));
        const double reach = (MC_ZMAX + mx) * sigma;
        ulo[i] = ((w0[i] < v0[i]) ? w0[i] : v0[i]) - reach;
        uhi[i] = ((v0[i] < w0[i]) ? w0[i] : v0[i]) + reach;
    }
    unsigned long long smask[SWEEP_WORDS];
    cull_boxes<D>(sbox, M, ulo, uhi, smask, lane);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    unsigned long long mine = 0;
    for (int64_t kb = k_begin; kb < k_end; kb += SWEEP_THREADS) {
        const int64_t k = kb + threadIdx.x;
        const bool act = k < k_end;
        // selector: Philox(key = seed, counter = (k, e, 2 D, 3)) word 0
        uint32_t c0 = (uint32_t)k, c1 = (uint32_t)(e_off + e), c2 = (uint32_t)(2 * D), c3 = 3u, q0 = k0, q1 = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ q0, n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ q1, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            q0 += 0x9E3779B9u; q1 += 0xBB67AE85u;
        }
        const bool shifted = (c0 & 1u) != 0;
        double y[2 * D], v[D], w[D];
#pragma unroll
        for (int c = 0; c < 2 * D; ++c) {
            const double z = mc_normal(k0, k1, (uint32_t)k, (uint32_t)(e_off + e), (uint32_t)c);
            y[c] = shifted ? z + mu[c] : z;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double pv = sigma * y[i], pw = sigma * y[D + i];
            v[i] = v0[i] + pv; w[i] = w0[i] + pw;
        }
        bool fr = act && in_state_space_sl<D>(v, ss);
        if (M > 0) fr = sweep_segment<D>(sbox, smask, v, w, fr);
        if (act && !fr) {
            double a = 1.0, b = 1.0;
#pragma unroll
            for (int c = 0; c < 2 * D; ++c) {
                const double xa = (y[c] * MC_INV + 262140.0) * (1.0 / 65536.0);
                a = a * mc_ih8_pdf(xa);
                const double yb = y[c] - mu[c];
                const double xb = (yb * MC_INV + 262140.0) * (1.0 / 65536.0);
                b = b * mc_ih8_pdf(xb);
            }
            const double ha = 0.5 * a, hb = 0.5 * b;
            const double dn = ha + hb;
            const double wgt = (dn > 0.0) ? a / dn : 0.0;
            mine += (unsigned long long)(wgt * 1099511627776.0);
        }
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0 && mine) atomicAdd(&s_sum, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_sum) atomicAdd(&wsum[e], s_sum);
}

int32_t mpfmt_launch_mc_ais_edges(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double sigma, int64_t rollouts,
                                  uint64_t seed, unsigned long long* d_wsum, double* d_mu)
{
    int32_t rc;
    if ((rc = check_boxes(ctx, ctx->d))) return rc;
    if (ctx->cc_kind != 0) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the Monte-Carlo evaluator runs against the AABB checker");
    if (E == 0) return MPFMT_OK;
    const int d = ctx->d;
    const int chunk = box_chunk(ctx->M, d, true);
    if (ctx->M > chunk) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "the importance-sampling estimator stages the whole obstacle set: M <= %d at d = %d", chunk, d);
    const size_t lds = sweep_lds(std::max(chunk, 1), d);
    int64_t slices = std::max<int64_t>(1, std::min<int64_t>((rollouts + 4 * SWEEP_THREADS - 1) / (4 * SWEEP_THREADS),
                                                            std::max<int64_t>(1, (int64_t)ctx->num_cus * 8 / std::max<int64_t>(E, 1))));
    slices = std::min<int64_t>(slices, 65535);
    const int64_t per_block = ((std::max<int64_t>(rollouts, 1) + slices - 1) / slices + SWEEP_THREADS - 1) / SWEEP_THREADS * SWEEP_THREADS;
    slices = (std::max<int64_t>(rollouts, 1) + per_block - 1) / per_block;
    mpfmt_timed tm5(ctx);
    for (int64_t e0 = 0; e0 < E; e0 += 1 << 20) {
        const int64_t ne = std::min<int64_t>(E - e0, 1 << 20);
        DISPATCH_D(d, hipLaunchKernelGGL((k_mc_ais_pilot<DD>), dim3((unsigned)ne), dim3(SWEEP_THREADS), lds, ctx->stream,
                                         ctx->Xo, d_src1 + e0, d_dst1 + e0, sigma, seed, ctx->boxes, ctx->M, ctx->ss, d_mu, e0));
        if (rollouts > 0)
            DISPATCH_D(d, hipLaunchKernelGGL((k_mc_ais_edges<DD>), dim3((unsigned)ne, (unsigned)slices), dim3(SWEEP_THREADS), lds, ctx->stream,
                                             ctx->Xo, d_src1 + e0, d_dst1 + e0, sigma, rollouts, per_block, seed, ctx->boxes, ctx->M, ctx->ss,
                                             (const double*)d_mu, d_wsum + e0, e0));
    }
    HIPCHK(ctx, hipGetLastError());
    tm5.end("mc_ais_edges");
    return MPFMT_OK;
}

int32_t mpfmt_launch_mc_is_edges(mpfmt_ctx* ctx, const int64_t* d_src1, const int64_t* d_dst1, int64_t E, double sigma, int64_t rollouts,
                                 uint64_t seed, unsigned long long* d_wsum)
{
    int32_t rc;
    if ((rc = check_boxes(ctx, ctx->d))) return rc;
    if (ctx->cc_kind != 0) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the Monte-Carlo evaluator runs against the AABB checker");
    if (E == 0 || rollouts == 0) return MPFMT_OK;
    const int d = ctx->d;
    const int chunk = box_chunk(ctx->M, d, true);
    if (ctx->M > chunk) return mpfmt_fail(ctx, MPFMT_ERR_ARG, "the importance-sampling estimator stages the whole obstacle set: M <= %d at d = %d", chunk, d);
    const size_t lds = sweep_lds(std::max(chunk, 1), d);
    int64_t slices = std::max<int64_t>(1, std::min<int64_t>((rollouts + 4 * SWEEP_THREADS - 1) / (4 * SWEEP_THREADS),
                                                            std::max<int64_t>(1, (int64_t)ctx->num_cus * 8 / std::max<int64_t>(E, 1))));
    slices = std::min<int64_t>(slices, 65535);
    const int64_t per_block = ((rollouts + slices - 1) / slices + SWEEP_THREADS - 1) / SWEEP_THREADS * SWEEP_THREADS;
    slices = (rollouts + per_block - 1) / per_block;
    mpfmt_timed tm5(ctx);
    for (int64_t e0 = 0; e0 < E; e0 += 1 << 20) {
        const int64_t ne = std::min<int64_t>(E - e0, 1 << 20);
        DISPATCH_D(d, hipLaunchKernelGGL((k_mc_is_edges<DD>), dim3((unsigned)ne, (unsigned)slices), dim3(SWEEP_THREADS), lds, ctx->stream,
                                         ctx->Xo, d_src1 + e0, d_dst1 + e0, sigma, rollouts, per_block, seed, ctx->boxes, ctx->M, ctx->ss, d_wsum + e0, e0));
    }
    HIPCHK(ctx, hipGetLastError());
    tm5.end("mc_is_edges");
    return MPFMT_OK;
}

// tasks of 16 columns; of 8 when that would leave fewer than ~6 tasks per resident wavefront (a shard of a multi-GPU
// build, small graphs): finer tasks balance the tail.  One resident set of workgroups.
template <int D>
static int32_t launch_graph_sweep_d(mpfmt_ctx* ctx, size_t lds, double rpad, int chunk, const int32_t* sweep_perm, int64_t sp_begin,
                                    int64_t sp_end, const int32_t* spec_fail, bool sorted_rows)
{
    constexpr auto k8 = k_graph_sweep<D, 8>;
    constexpr auto k4 = k_graph_sweep<D, 4>;
    const int waves = SWEEP_GT(D) / 64;
    if (lds > 64 * 1024) {                                       // beyond the default dynamic-LDS limit (gfx950 has 160 KB)
        HIPCHK(ctx, hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(ctx, hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    int per_cu = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k8, SWEEP_GT(D), lds));
    const int64_t resident = (int64_t)std::max(per_cu, 1) * ctx->num_cus;
    // 8-column tasks everywhere: measured equal to 16 on the unsharded north star (2.89 vs 2.91 ms) and cfg3, 8 % better on
    // 2 shards (1.59 vs 1.73 ms) and 17-20 % on 4 and 8 (finer dynamic balance over the resident grid); fewer resident
    // workgroups than the occupancy allows is always worse (tools/run_shard_all.py)
    // (and 4-column tasks when a wavefront would get fewer than 4.5 8-column ones -- the interior ranks of 8 shards: 5-8 %;
    // with more tasks per wavefront 4 is worse: 15 % at a dozen, 19 % on the unsharded graph)
    const int tc = ((sp_end - sp_begin + 7) / 8 < (9 * resident * waves) / 2) ? 4 : 8;
    const int64_t ntasks = (sp_end - sp_begin + tc - 1) / tc;
    const unsigned nb = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ntasks + waves - 1) / waves, resident));
    hipLaunchKernelGGL(tc == 4 ? k4 : k8, dim3(nb), dim3(SWEEP_GT(D)), lds, ctx->stream, ctx->Xo, ctx->colptr, ctx->rowval, ctx->N,
                       rpad, ctx->boxes, ctx->M, chunk, ctx->ss, (unsigned long long*)ctx->graph_free, ctx->sweep_ctr, sweep_perm,
                       sp_begin, sp_end, spec_fail, sorted_rows ? ctx->Xs : ctx->Xo, sorted_rows ? ctx->rowpos : ctx->rowval,
                       sorted_rows ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// round table (count -> scan -> fill) + k_graph_sweep_rt; entries = nnz or, on a speculative step, the trusted capacity
template <int D>
static int32_t launch_sweep_rt_d(mpfmt_ctx* ctx, size_t lds, double rpad, const int32_t* spec_fail, bool sorted_rows, int64_t table_cap)
{
    constexpr auto kk = k_graph_sweep_rt<D>;
    if (lds > 64 * 1024) HIPCHK(ctx, hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kk, SWEEP_GT(D), lds));
    // one resident set at most; the task count lives on the device, its bound (table capacity / 64) is known here -- a small graph
    // gets a few workgroups instead of 4096 wavefronts queueing on the task counter
    const int64_t resident = (int64_t)std::max(per_cu, 1) * ctx->num_cus;
    const int64_t need = (((int64_t)table_cap + RT_TASK - 1) / RT_TASK + SWEEP_GT(D) / 64 - 1) / (SWEEP_GT(D) / 64);
    const unsigned nb = (unsigned)std::max<int64_t>(1, std::min(resident, need));
    hipLaunchKernelGGL(kk, dim3(nb), dim3(SWEEP_GT(D)), lds, ctx->stream, ctx->Xo, ctx->colptr, ctx->N,
                       sorted_rows ? ctx->Xs : ctx->Xo, sorted_rows ? ctx->rowpos : ctx->rowval, rpad, ctx->boxes, ctx->M,
                       (int)(ctx->ss.has && !ctx->ssflag_all_in), ctx->rt_ss, (unsigned long long*)ctx->graph_free, (const sweep_rd*)ctx->rt_table, ctx->rt_total, ctx->sweep_ctr, spec_fail,
                       sorted_rows ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// device copy of the state-space bounds (scalar loads in the sweep kernels) and, once per (sample set, bounds), the answer to
// "do all samples lie in the state space?" (the sweeps then skip the per-row in_state_space test)
int32_t mpfmt_sweep_prepare_ss(mpfmt_ctx* ctx)
{
    int32_t rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rt_ss, sizeof(double) * 2 * MPFMT_MAX_DIM))) return rc;
    if (!ctx->rt_ss_valid || memcmp(&ctx->rt_ss_host, &ctx->ss, sizeof(mpfmt_ss)) != 0) {
        double b[2 * MPFMT_MAX_DIM];
        for (int i = 0; i < MPFMT_MAX_DIM; ++i) { b[i] = ctx->ss.lo[i]; b[MPFMT_MAX_DIM + i] = ctx->ss.hi[i]; }
        HIPCHK(ctx, hipMemcpyAsync(ctx->rt_ss, b, sizeof(b), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                 // b is a stack buffer
        ctx->rt_ss_host = ctx->ss; ctx->rt_ss_valid = true;
    }
    // once per (sample set, bounds): do all samples lie in the state space?  (the sweep then drops 2 d comparisons per entry)
    if (ctx->ss.has && (ctx->ssflag_epoch != ctx->samples_epoch || memcmp(&ctx->ssflag_ss, &ctx->ss, sizeof(mpfmt_ss)) != 0)) {
        // (every sample is finite and the uploads keep the set's bounding box: all samples pass in_state_space, statespaces.jl:150,
        // exactly when the box does -- no kernel, no read-back; new samples every step then cost the step nothing here)
        bool all_in = true;
        if (ctx->N > 0) for (int k = 0; k < ctx->d; ++k) all_in = all_in && (ctx->ss.lo[k] <= ctx->bb_lo[k]) && (ctx->bb_hi[k] <= ctx->ss.hi[k]);
        ctx->ssflag_all_in = all_in;
        ctx->ssflag_epoch = ctx->samples_epoch; ctx->ssflag_ss = ctx->ss;
    }
    return MPFMT_OK;
}

static int32_t launch_graph_sweep_rt(mpfmt_ctx* ctx, double rpad, const int32_t* sweep_perm, int64_t sp_begin, int64_t sp_end,
                                     const int32_t* spec_fail, bool sorted_rows, int64_t entries)
{
    int32_t rc;
    const int d = ctx->d;
    const int64_t ncol = sp_end - sp_begin;
    const int64_t cap = entries / 16 + ncol + 8;                          // quarters: every column adds at most one partial quarter (+ padding of the last round)
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rt_cnt, sizeof(int64_t) * (size_t)(ncol + 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rt_off, sizeof(int64_t) * (size_t)(ncol + 1)))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rt_table, sizeof(sweep_rd) * (size_t)cap))) return rc;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rt_total, sizeof(int64_t)))) return rc;
    if ((rc = mpfmt_sweep_prepare_ss(ctx))) return rc;
    const size_t tmp_bytes = mpfmt_scan_tmp_bytes((size_t)(ncol + 1));
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->rt_tmp, tmp_bytes))) return rc;
    const unsigned nbk = (unsigned)((ncol + 1 + 255) / 256);
    hipLaunchKernelGGL(k_round_count, dim3(nbk), dim3(256), 0, ctx->stream, ctx->colptr, sweep_perm, sp_begin, sp_end, ctx->rt_cnt, spec_fail);
    if ((rc = mpfmt_scan_i64_tmp(ctx, ctx->rt_cnt, ctx->rt_off, (size_t)(ncol + 1), ctx->rt_tmp))) return rc;
    const unsigned nbf = (unsigned)((((ncol + 63) / 64 + 1) * 64 + 255) / 256);       // one wavefront per 64 columns + the one that pads and totals
    hipLaunchKernelGGL(k_round_fill, dim3(nbf), dim3(256), 0, ctx->stream, ctx->colptr, sweep_perm, sp_begin, sp_end, ctx->rt_off,
                       (sweep_rd*)ctx->rt_table, cap, ctx->rt_total, spec_fail);
    HIPCHK(ctx, hipGetLastError());
    const int waves = SWEEP_GT(d) / 64;
    const size_t lds = (size_t)SWEEP_CHUNK * 2 * d * sizeof(double) + (size_t)waves * (d + 2) * SWEEP_QCAP * sizeof(double);
    mpfmt_timed tk(ctx);                                           // the kernel on its own, inside the caller's "sweep_graph" interval
    DISPATCH_D(d, rc = launch_sweep_rt_d<(DD <= 8 ? DD : 8)>(ctx, lds, rpad, spec_fail, sorted_rows, cap));
    tk.end("sweep_kernel");
    return rc;
}

// spec_fail / mask_entries: the speculative step (mpfmt_graph_step) sweeps before the host knows nnz -- the mask is then
// sized and preset for mask_entries (the trusted capacity) and the kernel bails out on the device flag
int32_t mpfmt_launch_graph_sweep(mpfmt_ctx* ctx, const int32_t* spec_fail, int64_t mask_entries)
{
    int32_t rc;
    if ((rc = check_boxes(ctx, ctx->d))) return rc;
    if (!ctx->graph_filled) return mpfmt_fail(ctx, MPFMT_ERR_STATE, "graph sweep before the r-disc graph is filled");
    const int64_t words = (std::max<int64_t>(ctx->nnz, mask_entries) + 63) / 64;
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->graph_free, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1)))) return rc;
    if (ctx->cc_kind == 1) {                                         // 2-D SAT world: lane = entry, whole words written
        mpfmt_timed tm6(ctx);
        HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
        rc = mpfmt_2d_launch_graph(ctx);
        tm6.end("sweep_graph");
        if (rc) return rc;
        ctx->graph_swept = true;
        return MPFMT_OK;
    }
    mpfmt_timed tm7(ctx);
    // preset to ones (the sweep clears blocked entries); an empty graph keeps one zero word
    HIPCHK(ctx, hipMemsetAsync(ctx->graph_free, ctx->nnz > 0 ? 0xFF : 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(words, 1), ctx->stream));
    if (ctx->nnz > 0 && ctx->pend_valid && ctx->d <= 6 && ctx->M <= SWEEP_CHUNK) {
        // the step's pair kernel has done the broad phase and the ordering pass has listed what it flagged: only those entries
        // are visited.  (Should a segment of the list have overflowed, the kernel returns at once and the caller -- who reads the
        // flag behind its next synchronisation -- sweeps the whole graph.)
        const double rpad = ctx->graph_r * (1.0 + 1e-9) + 1e-300;
        mpfmt_timed tk(ctx);
        DISPATCH_D(ctx->d, rc = launch_sweep_pending_d<(DD <= 6 ? DD : 6)>(ctx, rpad, spec_fail));
        tk.end("sweep_kernel");
        if (rc) return rc;
        tm7.end("sweep_graph");
        ctx->graph_swept = true;
        ctx->sweep_pending_used = true;
        return MPFMT_OK;
    }
    ctx->sweep_pending_used = false;
    if (ctx->nnz > 0) {
        const int d = ctx->d;
        const int waves = SWEEP_GT(d) / 64;
        const int chunk = box_chunk(ctx->M, d, true);
        // transposed (SoA) boxes + one narrow-phase queue per wave
        const size_t lds = (size_t)SWEEP_CHUNK * 2 * d * sizeof(double) + (d <= 8 ? (size_t)waves * (d + 1) * SWEEP_QCAP * sizeof(double) : 0);
        const double rpad = ctx->graph_r * (1.0 + 1e-9) + 1e-300;
        // persistent workgroups (one resident set): boxes are staged once per workgroup, tasks of SWEEP_TC columns
        // are claimed from a counter per obstacle chunk
        const int nchunks = std::max(1, (ctx->M + chunk - 1) / chunk);
        if ((rc = mpfmt_ensure(ctx, (void**)&ctx->sweep_ctr, sizeof(int) * 8 * (size_t)nchunks))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->sweep_ctr, 0, sizeof(int) * 8 * (size_t)nchunks, ctx->stream));
        // unsharded: all columns in caller order; sharded: only this shard's cell-sorted positions (via perm)
        // the single-pass build also knows every row by its cell-sorted position: rows are then gathered from Xs and the
        // columns visited in cell-sorted order, one contiguous range per XCD
        const bool sorted_rows = ctx->sweep_sorted && ctx->rowpos_valid && ctx->pool_valid && ctx->rdisc_path_used == 2 && ctx->rowpos && ctx->Xs &&
                                 ctx->perm != nullptr && ctx->tile_end > ctx->tile_begin;
        const bool sharded = ctx->world > 1 && ctx->perm != nullptr && ctx->tile_end > ctx->tile_begin;
        const bool by_perm = sharded || sorted_rows;
        const int32_t* sweep_perm = by_perm ? ctx->perm : nullptr;
        const int64_t sp_begin = by_perm ? ctx->tile_begin * 64 : 0;
        const int64_t sp_end = by_perm ? std::min<int64_t>(ctx->tile_end * 64, ctx->ntiles * 64) : ctx->N;
        if (ctx->sweep_rounds && d <= 8 && ctx->M <= SWEEP_CHUNK)
            rc = launch_graph_sweep_rt(ctx, rpad, sweep_perm, sp_begin, sp_end, spec_fail, sorted_rows, std::max<int64_t>(ctx->nnz, mask_entries));
        else
            DISPATCH_D(d, rc = launch_graph_sweep_d<DD>(ctx, lds, rpad, chunk, sweep_perm, sp_begin, sp_end, spec_fail, sorted_rows));
        if (rc) return rc;
        HIPCHK(ctx, hipGetLastError());
    }
    tm7.end("sweep_graph");
    ctx->graph_swept = true;
    return MPFMT_OK;
}
