// Double-integrator all-pairs steer with the candidate test on the matrix cores (gfx950).
//
// Same contract as k_di_pairs (kernels_di.hip; reference steer_pairwise, src/statespaces/linearquadratic.jl:196-225): every ordered pair
// (source i -> target j) with dcost(r) > 0 is steered (safeguarded Newton for the optimal time) and kept when its cost is <= r.  There the
// candidate test `dcost(r) > 0` (:213) costs ~50 vector instructions per 64 pairs of ALL N^2 pairs -- two thirds of the kernel at
// BASELINE configs[3] once the Newton iteration only runs on the 1.8 % that can pass.  Here the test is ONE bilinear form per pair:
//
//   with P = 6 (p - p_c) / r^2 and V = 2 v / r (p_c: centre of the samples' box -- the form only sees differences of positions),
//     Q(i -> j) = 36 a / r^4 - 24 b / r^3 + 4 c / r^2                     (a = |p_j - p_i|^2, b = (p_j - p_i).(v_i + v_j), c = |v_i|^2 + v_i.v_j + |v_j|^2)
//               = |P_i + V_i|^2 + |P_j - V_j|^2 + P_j . (-2 P_i - 2 V_i) + V_j . (2 P_i + V_i)
//   and dcost(r) = 1 - rho Q.  Target features f = (P_j, V_j), source features g = (-2 (P_i + V_i), 2 P_i + V_i), two norms: with every
//   value split into fp16 hi + lo, sixteen slots (f_hi g_hi, f_hi g_lo, f_lo g_hi, the norms' hi / lo against ones) make
//   v_mfma_f32_32x32x16_f16 with C = -T deliver  Q - T  for 32 x 32 pairs; T = 1 / rho + E, E the bound of everything the fp16 / fp32
//   evaluation can lose (mpfmt_di_mf_prepare), so that NO pair with dcost(r) > 0 can come out non-negative.
//
// Pipeline, as in kernels_rdisc_mfma.hip: one wavefront per (tile of 64 targets, slice of the source chunks); 4 MFMAs per chunk, the
// 1024 accumulator signs of each funnelled into 16 bits per lane (v_alignbit), lanes with set bits push one record to a wave-private LDS
// queue, records are expanded into survivors (6.4 % of the pairs at configs[3]), and 64 survivors at a time (lane = survivor) take the
// multiply-only fp64 tests of k_di_pairs -- the conservative dcost test and the lower bound 2 sqrt(rho m) on the cost; what passes both
// (1.8 %) is queued once more and steered 64 at a time: exact dcost(r) > 0, Newton, cost <= r -- the drain of k_di_pairs, unchanged.
// The graph is the same entry for entry: the filter and both multiply-only tests only ever drop pairs the exact tests would drop.
#include "mpfmt_internal.h"
#include "di_steer.h"
#include <algorithm>
#include <cmath>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define DMF_RCAP 128                // record queue entries per wavefront (expanded 64 at a time)
#define DMF_QSZ 320                 // survivor queue entries per wavefront (tested 64 at a time)
#define DMF_PAD_NORM 60000.0f       // norm stand-in of a pad state: above every threshold

struct dimf_args {
    di_args a;
    const uint4* opsT;              // [npad][2] target-role operands (16 fp16 each)
    const uint4* opsS;              // [npad][2] source-role operands
    float negT;                     // -T
    int64_t npad;
};

// ---- operands -------------------------------------------------------------------------------------------------------
// one thread per state: features in fp64, every value as fp16 hi + fp16 lo (lo = the rounded remainder)
template <int M>
__global__ void k_di_make_ops(const double* __restrict__ X, int64_t N, int64_t npad, double sp, double sv, const double* __restrict__ pc,
                              uint4* __restrict__ opsT, uint4* __restrict__ opsS)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= npad) return;
    union { _Float16 h[16]; uint4 v[2]; } T, S;
#pragma unroll
    for (int k = 0; k < 16; ++k) { T.h[k] = (_Float16)0.0f; S.h[k] = (_Float16)0.0f; }
    auto split = [](double x, _Float16& hi, _Float16& lo) { hi = (_Float16)x; lo = (_Float16)(x - (double)hi); };
    if (s < N) {
        double n1 = 0.0, n0 = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const double P = (X[s * 2 * M + i] - pc[i]) * sp, V = X[s * 2 * M + M + i] * sv;
            const double f[2] = {P, V};                                  // target role: (P_j, V_j)
            const double g[2] = {-2.0 * (P + V), 2.0 * P + V};          // source role
            n1 += (P - V) * (P - V);
            n0 += (P + V) * (P + V);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = q * M + i;                                  // feature slot 0 .. 2M-1 (< 4)
                _Float16 fh, fl, gh, gl;
                split(f[q], fh, fl); split(g[q], gh, gl);
                T.h[k] = fh; T.h[4 + k] = fh; T.h[8 + k] = fl;
                S.h[k] = gh; S.h[4 + k] = gl; S.h[8 + k] = gh;
            }
        }
        _Float16 h, l;
        split(n1, h, l); T.h[12] = h; T.h[13] = l; T.h[14] = (_Float16)1.0f; T.h[15] = (_Float16)1.0f;
        split(n0, h, l); S.h[12] = (_Float16)1.0f; S.h[13] = (_Float16)1.0f; S.h[14] = h; S.h[15] = l;
    } else {
        T.h[12] = (_Float16)DMF_PAD_NORM; T.h[14] = (_Float16)1.0f; T.h[15] = (_Float16)1.0f;
        S.h[12] = (_Float16)1.0f; S.h[13] = (_Float16)1.0f; S.h[14] = (_Float16)DMF_PAD_NORM;
    }
    opsT[s * 2] = T.v[0]; opsT[s * 2 + 1] = T.v[1];
    opsS[s * 2] = S.v[0]; opsS[s * 2 + 1] = S.v[1];
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
// MODE 0: count   1: fill the staging CSC from the counts   2: count AND keep the accepted hits in slot lists (single pass)
template <int M, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_di_pairs_mf(dimf_args g)
{
    const di_args& a = g.a;
    constexpr bool FILL = (MODE == 1);
    constexpr int NS = 2 * M;
    __shared__ double s_tgt[64 * NS];                     // this tile's targets (AoS, fp64)
    __shared__ uint32_t s_rm[DMF_RCAP];                   // record queue: chunk << 6 | finding lane
    __shared__ unsigned long long s_rh[DMF_RCAP];         //               the lane's 64 sign bits of that chunk
    __shared__ uint32_t s_qs[DMF_QSZ];                    // survivor queue: chunk << 12 | finding lane << 6 | sign-bit position
    __shared__ uint32_t s_qi[DI_QCAP];                    // candidate queue (both multiply-only tests passed): source index
    __shared__ uint8_t s_ql[DI_QCAP];                     //                 target lane
    __shared__ int32_t s_cnt[64];
    __shared__ int64_t s_base[64];
    const int lane = threadIdx.x;
    const int64_t item = blockIdx.x;
    const int64_t tile = (item / a.S) * a.tile_step;
    const int slice = (int)(item % a.S);
    const int64_t slot_item = tile * a.S + slice;
    const int64_t j = tile * 64 + lane;
    const bool jact = j < a.N;
    const int64_t npad = a.ntiles * 64;
    const int kb = lane >> 5, col = lane & 31;

#pragma unroll
    for (int i = 0; i < NS; ++i) s_tgt[lane * NS + i] = jact ? a.X[j * NS + i] : 0.0;
    s_cnt[lane] = 0;
    if (FILL) {
        int64_t base = jact ? a.colptr[j] : 0;
        for (int s = 0; s < slice; ++s) base += a.slice_cnt[(int64_t)s * npad + j];
        s_base[lane] = base;
    }
    // the slice's sources: whole 64-state chunks [ch0, ch1) (the sub-slices of the pilot launch nest inside the slices)
    const int64_t nch = a.ntiles;
    const int64_t ch0 = nch * slice / a.S, ch1 = nch * (slice + 1) / a.S;

    // A fragments: target rows rb * 32 + col, slots 8 kb .. 8 kb + 7
    half8 aF[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        union { uint4 u; half8 h; } cv;
        cv.u = g.opsT[(tile * 64 + rb * 32 + col) * 2 + kb];
        aF[rb] = cv.h;
    }
    f32x16 cinit;
#pragma unroll
    for (int k = 0; k < 16; ++k) cinit[k] = g.negT;

    int qcount = 0, rcount = 0, q2 = 0;                   // wave-uniform queue lengths: survivors, records, candidates
    int pool_over = 0;
    unsigned long long ncand = 0;

    // ---- the drain of k_di_pairs: exact dcost(r) > 0, Newton, cost <= r (lane = candidate) -----------------------------------
    auto steer = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int first = q2 - n;
        if (lane < n) {
            const int64_t i = s_qi[first + lane];
            const int tl = s_ql[first + lane];
            double p0[NS], p1[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) { p0[q] = a.X[i * NS + q]; p1[q] = s_tgt[tl * NS + q]; }
            const di_coef k = di_coefs<M>(p0, p1);
            if (di_dcost(k, a.rho, a.r) > 0) {                 // candidate test `cd .> 0`, linearquadratic.jl:213
                double cost, t;
                di_steer<M>(p0, p1, a.rho, a.r, cost, t);
                if (cost <= a.r) {                             // linearquadratic.jl:221
                    const int slot = atomicAdd(&s_cnt[tl], 1);
                    if (FILL) {
                        const int64_t pos = s_base[tl] + slot;
                        a.rowtmp[pos] = (int32_t)i;
                        a.valtmp[pos] = cost;
                        a.tvaltmp[pos] = t;
                    }
                    if (MODE == 2) {
                        if (slot < a.pool_cap) {
                            const int64_t pos = (slot_item * 64 + tl) * a.pool_cap + slot;
                            a.pool_i[pos] = (int32_t)i;
                            a.pool_c[pos] = cost;
                            a.pool_t[pos] = t;
                        } else {
                            pool_over = 1;
                        }
                    }
                }
            }
        }
        q2 = __builtin_amdgcn_readfirstlane(first);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- the multiply-only fp64 tests of k_di_pairs on n queued survivors (lane = survivor); what passes is queued for the steer ----
    auto test = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int first = qcount - n;
        qcount = __builtin_amdgcn_readfirstlane(first);
        bool pend = false;
        int64_t i = 0;
        int tl = 0;
        if (lane < n) {
            const uint32_t e = s_qs[first + lane];
            const int bpos = (int)(e & 63u), fl = (int)((e >> 6) & 63u);
            const int64_t qc = (int64_t)(e >> 12);
            const int t = bpos >> 4, r = 15 - (bpos & 15);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (fl >> 5);
            i = qc * 64 + (int64_t)((t >> 1) * 32 + (fl & 31));
            tl = (t & 1) * 32 + row;
            const int64_t jj = tile * 64 + tl;
            if (i < a.N && jj < a.N && i != jj) {
                double ca = 0.0, cb = 0.0, cc = 0.0;
#pragma unroll
                for (int q = 0; q < M; ++q) {
                    const double x0p = a.X[i * NS + q], v0 = a.X[i * NS + M + q];
                    const double p = s_tgt[tl * NS + q] - x0p, v1 = s_tgt[tl * NS + M + q];
                    ca += p * p;
                    cb += p * (v0 + v1);
                    cc += (v0 * v0 + v0 * v1) + v1 * v1;
                }
                const double ta = 36.0 * ca * a.i4, tb = 24.0 * cb * a.i3, tc = 4.0 * cc * a.i2;
                const double cd = 1.0 - a.rho * ((ta - tb) + tc);
                const double slack = 1e-9 * (1.0 + a.rho * ((ta + fabs(tb)) + tc));
                // (the lower bound 2 sqrt(rho m) <= cost of k_di_pairs: see there)
                const double l1 = 4.0 * ca * cc, l2 = 3.0 * cb * cb;
                const bool far = 4.0 * a.rho * (l1 - l2) > (ca * a.r2) * (1.0 + 1e-9) + 4e-9 * a.rho * (l1 + l2);
                pend = (cd > -slack) && !far;
            }
        }
        const unsigned long long pm = __ballot(pend);
        if (pm) {
            if (q2 > DI_QCAP - 64) steer(64);
            if (pend) {
                const int pos = q2 + (int)__popcll(pm & ((1ull << lane) - 1ull));
                s_qi[pos] = (uint32_t)i;
                s_ql[pos] = (uint8_t)tl;
            }
            const int np = (int)__popcll(pm);
            q2 = __builtin_amdgcn_readfirstlane(q2 + np);
            ncand += (unsigned long long)np;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- expand up to 64 records (lane = record, newest first) into the survivor queue ------------------------------
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
        const int room = DMF_QSZ - qcount;
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
        while (qcount >= 64) test(64);
    };

    // ---- main loop over the slice's source chunks: B fragments through a buffer descriptor (per-lane offset fixed, chunk offset scalar) ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(g.opsS), 0, (int)(g.npad * 32), 0x00020000);
    const int voff = col * 32 + kb * 16;
    auto load_b = [&](int64_t c, u32x4 (&bq)[2]) {
        bq[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (int)(c * 2048), 0);
        bq[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024, (int)(c * 2048), 0);
    };
    auto process = [&](int64_t c, u32x4 (&bq)[2], bool refill, int64_t cn) {
        union { u32x4 u; half8 h; } bf0, bf1;
        bf0.u = bq[0]; bf1.u = bq[1];
        const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[0], bf0.h, cinit, 0, 0, 0);
        const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[1], bf0.h, cinit, 0, 0, 0);
        const f32x16 acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[0], bf1.h, cinit, 0, 0, 0);
        const f32x16 acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aF[1], bf1.h, cinit, 0, 0, 0);
        if (refill) load_b(cn, bq);
        // H: 16 sign bits per 32x32 block t = (source half) * 2 + (target half) at bits [16 t, 16 t + 16); bit (15 - r) <-> accumulator register r
        uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            h0 = __builtin_amdgcn_alignbit(h0, __float_as_uint(acc0[r]), 31);
            h1 = __builtin_amdgcn_alignbit(h1, __float_as_uint(acc1[r]), 31);
            h2 = __builtin_amdgcn_alignbit(h2, __float_as_uint(acc2[r]), 31);
            h3 = __builtin_amdgcn_alignbit(h3, __float_as_uint(acc3[r]), 31);
        }
        const unsigned long long H = (unsigned long long)(h0 | (h1 << 16)) | ((unsigned long long)(h2 | (h3 << 16)) << 32);
        const unsigned long long m = __ballot(H != 0);
        if (m) {
            while (rcount > DMF_RCAP - 64) expand();
            if (H != 0) {
                const int pos = rcount + (int)__popcll(m & ((1ull << lane) - 1ull));
                s_rm[pos] = ((uint32_t)c << 6) | (uint32_t)lane;
                s_rh[pos] = H;
            }
            rcount = __builtin_amdgcn_readfirstlane(rcount + (int)__popcll(m));
        }
    };
    {
        u32x4 ring[2][2];
        if (ch0 < ch1) load_b(ch0, ring[0]);
        if (ch0 + 1 < ch1) load_b(ch0 + 1, ring[1]);
        for (int64_t c = ch0; c < ch1; c += 2) {
            process(c, ring[0], c + 2 < ch1, c + 2);
            if (c + 1 < ch1) process(c + 1, ring[1], c + 3 < ch1, c + 3);
        }
    }
    while (rcount > 0) expand();
    while (qcount > 0) test(min(qcount, 64));
    while (q2 > 0) steer(min(q2, 64));

    if (MODE == 2 && pool_over) *a.pool_flag = 1;
    if (!FILL) {
        if (jact) a.slice_cnt[(int64_t)slice * npad + j] = s_cnt[lane];
        if (lane == 0 && a.counters) {
            atomicAdd(a.counters, (unsigned long long)(ch1 - ch0) * 64ull * 64ull);
            atomicAdd(a.counters + 1, ncand);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// The threshold.  Every slot value x reaches the matrix core as hi + lo with |x - hi - lo| <= 2^-22 |x| + 6.2e-5 (the second term: a
// remainder below the smallest normal fp16 number, should the core flush subnormal inputs); the products are exact in fp32 and every one
// of the 17 additions (in whatever order) rounds at 2^-24 of the running magnitude.  With F, G the largest |f|, |g| the samples' box allows:
//   E = 2M (F dG + G dF + dF dG + 2^-22 F G)      (the three products kept per feature, and the lo x lo product dropped)
//     + dN0 + dN1 + 17 x 2^-24 (3 x 2M F G + N0 + N1 + 1/rho)
// times 1.5, plus the slack of the exact test it stands in front of (1e-9 (1 + rho (ta + |tb| + tc))).  *usable = false when E is not
// small against 1 / rho (a radius far below the samples' extent: the vector-ALU kernel then).
int32_t mpfmt_di_mf_prepare(mpfmt_ctx* ctx, double rho, double r, float* negT, bool* usable, double* sp_out, double* sv_out, double* pc)
{
    *usable = false;
    const int m = ctx->d / 2;
    if (m < 1 || m > 2 || !(r > 0.0) || !(rho > 0.0) || ctx->N < 64 || ctx->N > ((int64_t)1 << 25)) return MPFMT_OK;      // (32-bit operand offsets, 20-bit chunk ids)
    const double sp = 6.0 / (r * r), sv = 2.0 / r;
    double Pm = 0.0, Vm = 0.0;
    for (int i = 0; i < m; ++i) {
        pc[i] = 0.5 * (ctx->bb_lo[i] + ctx->bb_hi[i]);
        Pm = std::max(Pm, 0.5 * (ctx->bb_hi[i] - ctx->bb_lo[i]) * sp);
        Vm = std::max(Vm, std::max(std::fabs(ctx->bb_lo[m + i]), std::fabs(ctx->bb_hi[m + i])) * sv);
    }
    if (!std::isfinite(Pm) || !std::isfinite(Vm)) return MPFMT_OK;
    Pm *= 1.0 + 1e-9; Vm *= 1.0 + 1e-9;
    const double F = std::max(Pm, Vm), G = std::max(2.0 * (Pm + Vm), 2.0 * Pm + Vm);
    const double N0 = m * (Pm + Vm) * (Pm + Vm), N1 = N0;
    if (G > 3.0e4 || N0 > 3.0e4) return MPFMT_OK;            // (fp16 range)
    const double q = std::ldexp(1.0, -22), tiny = 6.2e-5;
    const double dF = q * F + tiny, dG = q * G + tiny, dN = q * N0 + tiny;
    const double S = 3.0 * 2 * m * F * G + N0 + N1 + 1.0 / rho;
    double E = 2.0 * m * (F * dG + G * dF + dF * dG + q * F * G) + 2.0 * dN + 17.0 * std::ldexp(1.0, -24) * S;
    E *= 1.5;
    const double Sq = 4.0 * m * Pm * Pm + 8.0 * m * Pm * Vm + 3.0 * m * Vm * Vm;        // >= ta + |tb| + tc
    const double T = (1.0 / rho) * (1.0 + 1e-6) + 1e-9 * (1.0 / rho + Sq) + 1e-12 * Sq + E;
    if (E > 0.05 / rho) return MPFMT_OK;
    *negT = -(float)(T * (1.0 + 1e-6));
    *sp_out = sp; *sv_out = sv;
    *usable = true;
    return MPFMT_OK;
}

int32_t mpfmt_di_mf_build_operands(mpfmt_ctx* ctx, double sp, double sv, const double* pc_host)
{
    const int m = ctx->d / 2;
    const int64_t N = ctx->N, npad = ((N + 63) / 64) * 64;
    int32_t rc;
    // opsT | opsS | the centre: one grow-only buffer
    if ((rc = mpfmt_ensure(ctx, (void**)&ctx->di_ops, 64 * (size_t)npad + 64))) return rc;
    double* pcd = (double*)((char*)ctx->di_ops + 64 * (size_t)npad);
    HIPCHK(ctx, hipMemcpyAsync(pcd, pc_host, sizeof(double) * 2, hipMemcpyHostToDevice, ctx->stream));
    uint4* T = (uint4*)ctx->di_ops;
    uint4* S = T + 2 * npad;
    const int B = 256;
    if (m == 1) hipLaunchKernelGGL((k_di_make_ops<1>), dim3((unsigned)((npad + B - 1) / B)), dim3(B), 0, ctx->stream, ctx->Xo, N, npad, sp, sv, pcd, T, S);
    else hipLaunchKernelGGL((k_di_make_ops<2>), dim3((unsigned)((npad + B - 1) / B)), dim3(B), 0, ctx->stream, ctx->Xo, N, npad, sp, sv, pcd, T, S);
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}

// mode: 0 count, 1 fill, 2 count + slot lists; nblk = work items (tiles visited x slices)
int32_t mpfmt_di_mf_launch(mpfmt_ctx* ctx, const di_args& a, int mode, float negT, unsigned nblk)
{
    const int m = ctx->d / 2;
    dimf_args g;
    g.a = a;
    g.npad = a.ntiles * 64;
    g.opsT = (const uint4*)ctx->di_ops;
    g.opsS = g.opsT + 2 * g.npad;
    g.negT = negT;
#define LAUNCH(MM, MODE) hipLaunchKernelGGL((k_di_pairs_mf<MM, MODE>), dim3(nblk), dim3(64), 0, ctx->stream, g)
    if (m == 1) { if (mode == 0) LAUNCH(1, 0); else if (mode == 1) LAUNCH(1, 1); else LAUNCH(1, 2); }
    else if (m == 2) { if (mode == 0) LAUNCH(2, 0); else if (mode == 1) LAUNCH(2, 1); else LAUNCH(2, 2); }
    else return mpfmt_fail(ctx, MPFMT_ERR_STATE, "the matrix-core double-integrator prefilter is built for workspace dim 1, 2");
#undef LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return MPFMT_OK;
}
