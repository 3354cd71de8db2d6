/*
 * mpfmt_oracle.c -- CPU restatement of the FMT* batch-expand hot path of
 * schmrlng/MotionPlanning.jl (r-disc neighbour query, segment-vs-AABB sweep,
 * per-edge cost, and the FMT* loop that calls them).
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity checker for the HIP
 * library and the "port" CPU baseline of bench.py.  Nothing under
 * motionplanning.jl_amd/ may include, link or call it.
 *
 * PARITY UNPINNED: the reference is Julia-0.5 code with an empty test suite
 * (test/runtests.jl:1-5) and no Julia toolchain exists in this image, so the
 * restatement cannot be checked against reference outputs.  It is pinned only
 * by (i) hand-derived known answers on the reference's own AABB fixtures
 * (test/obstaclesets/ND.jl:1-14, see tests/golden/), and (ii) an independent
 * pure-Python transliteration of the same Julia lines (tests/jl_transliteration.py).
 *
 * Canonical arithmetic (declared by this build):
 *   - IEEE binary64 everywhere, no FMA contraction (build with
 *     -ffp-contract=off -fno-fast-math), operations in the written order;
 *   - d2 = ((t1*t1 + t2*t2) + t3*t3) + ...   with t_i = a_i - b_i;
 *   - Euclidean neighbour  <=>  i != v  &&  d2 <= r*r   (KD-tree "reduced
 *     distance" semantics, the method Euclidean dispatches to:
 *     src/nearneighbors.jl:179-183 + src/statespaces/geometric.jl:14);
 *   - dist = sqrt(d2) correctly rounded; edge cost C[y] + dist.
 *
 * Third-party arithmetic that is NOT under /root/reference (unpinned in
 * REQUIRE:1-9) and is restated from its published algorithm:
 *   NearestNeighbors.jl  KDTree / inrange   (membership test = sum of squares <= r^2)
 *   Distances.jl         colwise(Euclidean) (sqrt of the sequential sum of squares)
 *   StaticArrays.jl      SVector arithmetic (element-wise, unfused)
 *   Base.Collections.PriorityQueue          (binary min-heap; ties broken here by lowest index)
 *
 * All indices at this API are 0-based; the Python wrapper converts to the
 * reference's 1-based convention where a test needs it.
 */
#include <math.h>
/* Transcendental functions of the car models (src/statespaces/simplecars.jl calls Julia's libm): THIS oracle calls the C library's
 * sin / cos / atan2 / acos -- an implementation the HIP library shares nothing with, so a car-model comparison against it is an
 * independent check (costs to 1e-12, memberships exact away from the threshold).
 * A SECOND build of this same file (liboracle_devmath.so: `-DORC_DEVICE_MATH -include <the product's mp_math.h>`, see the
 * Makefile) swaps in the device's own polynomial versions.  It exists only for the explicitly labelled self-consistency tests
 * ("one set of functions on both sides: every tie between words breaks identically, trees equal bit for bit") and carries no
 * parity claim for the transcendental part.  This file itself includes nothing from motionplanning.jl_amd/. */
#ifndef ORC_DEVICE_MATH
#define mp_sin sin
#define mp_cos cos
#define mp_atan2 atan2
#define mp_acos acos
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXD 16
#define ORC_MAXPOLY 16     /* vertices per convex polygon of the 2-D SAT world */

/* ------------------------------------------------------------------------- */
/* a4: distance evaluation  (src/statespaces/geometric.jl:4-6)                */
/* ------------------------------------------------------------------------- */

/* Sequential, unfused sum of squares of (a - b). */
double orc_sqdist(const double *a, const double *b, int32_t d)
{
    double s = 0.0;
    for (int32_t i = 0; i < d; ++i) {
        double t = a[i] - b[i];
        double tt = t * t;
        s = (i == 0) ? tt : s + tt;
    }
    return s;
}

/* evaluate(Euclidean, v, w) = norm(w - v)   (geometric.jl:4) */
double orc_dist(const double *a, const double *b, int32_t d)
{
    return sqrt(orc_sqdist(a, b, d));
}

/* ------------------------------------------------------------------------- */
/* a3: r-disc query `inball`  (src/nearneighbors.jl:138-150, 179-183)         */
/* ------------------------------------------------------------------------- */

/*
 * X is d x N column-major (the zero-copy layout of Vector{SVector{d,Float64}},
 * src/primitivetypes.jl:21-24).  Returns the number of neighbours k; if
 * inds/ds are non-NULL writes up to cap of them (ascending index, self
 * excluded -- the SparseVector contract of nearneighbors.jl:138-198).
 *
 * mode 0: TreeDistanceDS semantics   d2 <= r*r          (nearneighbors.jl:179-183)
 * mode 1: generic-fallback semantics sqrt(d2) <= r      (nearneighbors.jl:138-150)
 */
int64_t orc_inball(const double *X, int64_t N, int32_t d, int64_t v, double r,
                   int32_t mode, int64_t *inds, double *ds, int64_t cap)
{
    const double *q = X + (size_t)v * d;
    const double r2 = r * r;
    int64_t k = 0;
    for (int64_t i = 0; i < N; ++i) {
        if (i == v) continue;
        double d2 = orc_sqdist(q, X + (size_t)i * d, d);
        int in = (mode == 0) ? (d2 <= r2) : (sqrt(d2) <= r);
        if (in) {
            if (inds && k < cap) inds[k] = i;
            if (ds && k < cap) ds[k] = sqrt(d2);
            ++k;
        }
    }
    return k;
}

/* Whole-graph form: the ImmutableNNC CSC (nearneighbors.jl:23-27).
 * Two-phase like the C ABI: count (colptr, 0-based offsets, N+1 entries), then fill. */
int64_t orc_rdisc_count(const double *X, int64_t N, int32_t d, double r, int32_t mode,
                        int64_t *colptr)
{
    int64_t nnz = 0;
    colptr[0] = 0;
    for (int64_t v = 0; v < N; ++v) {
        nnz += orc_inball(X, N, d, v, r, mode, NULL, NULL, 0);
        colptr[v + 1] = nnz;
    }
    return nnz;
}

void orc_rdisc_fill(const double *X, int64_t N, int32_t d, double r, int32_t mode,
                    const int64_t *colptr, int64_t *rowval, double *nzval)
{
    for (int64_t v = 0; v < N; ++v) {
        int64_t cap = colptr[v + 1] - colptr[v];
        orc_inball(X, N, d, v, r, mode, rowval + colptr[v], nzval + colptr[v], cap);
    }
}

/* ------------------------------------------------------------------------- */
/* KD-tree r-disc (the TreeDistanceDS analogue used for the CPU baseline).    */
/* Restates the published NearestNeighbors.jl KDTree scheme: split on the     */
/* widest dimension of the node's hyper-rectangle at the median, leaves of    */
/* <= 10 points, inrange prunes with the point-to-rectangle reduced distance  */
/* and tests leaf points with  sum of squares <= r^2.  The neighbour SET does */
/* not depend on the tree shape; tests check it equals orc_inball mode 0.     */
/* ------------------------------------------------------------------------- */

typedef struct {
    int64_t N;
    int32_t d;
    const double *X;     /* borrowed, d x N */
    int64_t *idx;        /* permutation, leaves are contiguous ranges */
    int64_t nnodes;
    int64_t *lo, *hi;    /* point range of each node */
    int32_t *sdim;       /* split dim, -1 for leaf */
    double *sval;
    int64_t *left, *right;
    double bmin[ORC_MAXD], bmax[ORC_MAXD];
} orc_kdtree;

#define ORC_LEAF 10

static void kd_select(int64_t *idx, int64_t lo, int64_t hi, int64_t k, const double *X, int32_t d, int32_t dim)
{
    /* quickselect on coordinate `dim`, ties by index for determinism */
    while (lo < hi) {
        int64_t p = idx[lo + (hi - lo) / 2];
        double pv = X[(size_t)p * d + dim];
        int64_t i = lo, j = hi;
        while (i <= j) {
            for (;;) { double a = X[(size_t)idx[i] * d + dim]; if (a < pv || (a == pv && idx[i] < p)) ++i; else break; }
            for (;;) { double a = X[(size_t)idx[j] * d + dim]; if (a > pv || (a == pv && idx[j] > p)) --j; else break; }
            if (i <= j) { int64_t t = idx[i]; idx[i] = idx[j]; idx[j] = t; ++i; --j; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return;
    }
}

static int64_t kd_build(orc_kdtree *T, int64_t lo, int64_t hi, double *bmin, double *bmax)
{
    int64_t id = T->nnodes++;
    T->lo[id] = lo; T->hi[id] = hi;
    T->left[id] = T->right[id] = -1; T->sdim[id] = -1; T->sval[id] = 0.0;
    if (hi - lo + 1 <= ORC_LEAF) return id;
    int32_t dim = 0; double best = -1.0;
    for (int32_t i = 0; i < T->d; ++i) { double w = bmax[i] - bmin[i]; if (w > best) { best = w; dim = i; } }
    int64_t mid = lo + (hi - lo + 1) / 2;
    kd_select(T->idx, lo, hi, mid, T->X, T->d, dim);
    double sv = T->X[(size_t)T->idx[mid] * T->d + dim];
    T->sdim[id] = dim; T->sval[id] = sv;
    double save = bmax[dim]; bmax[dim] = sv;
    int64_t l = kd_build(T, lo, mid - 1, bmin, bmax);
    bmax[dim] = save; save = bmin[dim]; bmin[dim] = sv;
    int64_t rr = kd_build(T, mid, hi, bmin, bmax);
    bmin[dim] = save;
    T->left[id] = l; T->right[id] = rr;
    return id;
}

orc_kdtree *orc_kdtree_build(const double *X, int64_t N, int32_t d)
{
    orc_kdtree *T = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    T->N = N; T->d = d; T->X = X;
    int64_t maxn = 2 * (N / 1 + 2);
    T->idx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    T->lo = (int64_t *)malloc(sizeof(int64_t) * (size_t)maxn);
    T->hi = (int64_t *)malloc(sizeof(int64_t) * (size_t)maxn);
    T->left = (int64_t *)malloc(sizeof(int64_t) * (size_t)maxn);
    T->right = (int64_t *)malloc(sizeof(int64_t) * (size_t)maxn);
    T->sdim = (int32_t *)malloc(sizeof(int32_t) * (size_t)maxn);
    T->sval = (double *)malloc(sizeof(double) * (size_t)maxn);
    for (int64_t i = 0; i < N; ++i) T->idx[i] = i;
    for (int32_t i = 0; i < d; ++i) { T->bmin[i] = INFINITY; T->bmax[i] = -INFINITY; }
    for (int64_t p = 0; p < N; ++p)
        for (int32_t i = 0; i < d; ++i) {
            double a = X[(size_t)p * d + i];
            if (a < T->bmin[i]) T->bmin[i] = a;
            if (a > T->bmax[i]) T->bmax[i] = a;
        }
    if (N > 0) {
        double bmin[ORC_MAXD], bmax[ORC_MAXD];
        memcpy(bmin, T->bmin, sizeof bmin); memcpy(bmax, T->bmax, sizeof bmax);
        kd_build(T, 0, N - 1, bmin, bmax);
    }
    return T;
}

void orc_kdtree_free(orc_kdtree *T)
{
    if (!T) return;
    free(T->idx); free(T->lo); free(T->hi); free(T->left); free(T->right); free(T->sdim); free(T->sval);
    free(T);
}

static void kd_range(const orc_kdtree *T, int64_t id, const double *q, double r2, double mind2,
                     double *off, int64_t *out, int64_t *k, int64_t cap)
{
    if (T->sdim[id] < 0) {
        for (int64_t p = T->lo[id]; p <= T->hi[id]; ++p) {
            int64_t i = T->idx[p];
            double d2 = orc_sqdist(q, T->X + (size_t)i * T->d, T->d);
            if (d2 <= r2) { if (*k < cap) out[*k] = i; ++*k; }
        }
        return;
    }
    int32_t dim = T->sdim[id];
    double diff = q[dim] - T->sval[id];
    int64_t near = diff < 0 ? T->left[id] : T->right[id];
    int64_t far = diff < 0 ? T->right[id] : T->left[id];
    kd_range(T, near, q, r2, mind2, off, out, k, cap);
    double old = off[dim];
    double nd2 = mind2 - old * old + diff * diff;
    /* conservative: tiny slack so pruning rounding can never drop a true neighbour */
    if (nd2 <= r2 * (1.0 + 1e-12)) {
        off[dim] = diff;
        kd_range(T, far, q, r2, nd2, off, out, k, cap);
        off[dim] = old;
    }
}

static int cmp_i64(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* inball through the tree: inrange(tree, V[v], r, true) ; deleteat!(self) ; colwise
 * (nearneighbors.jl:179-183).  Output sorted ascending, self excluded. */
int64_t orc_kdtree_inball(const orc_kdtree *T, int64_t v, double r, int64_t *inds, double *ds, int64_t cap)
{
    const double *q = T->X + (size_t)v * T->d;
    double off[ORC_MAXD];
    for (int32_t i = 0; i < T->d; ++i) off[i] = 0.0;
    int64_t k = 0;
    if (T->N > 0) kd_range(T, 0, q, r * r, 0.0, off, inds, &k, cap);
    int64_t kk = k < cap ? k : cap;
    qsort(inds, (size_t)kk, sizeof(int64_t), cmp_i64);
    /* delete self */
    int64_t w = 0; int found = 0;
    for (int64_t i = 0; i < kk; ++i) {
        if (inds[i] == v) { found = 1; continue; }
        inds[w++] = inds[i];
    }
    if (ds) for (int64_t i = 0; i < w; ++i) ds[i] = orc_dist(q, T->X + (size_t)inds[i] * T->d, T->d);
    return (k > cap) ? k - found : w;
}

/* ------------------------------------------------------------------------- */
/* a7: N-d AABB checker  (src/collisioncheckers/boxesND.jl:42-56)             */
/* boxes: lohi is (2*dw) x M column-major: per box lo[0..dw), hi[0..dw)       */
/* (BoxBounds(lohi::Matrix) = (lohi[:,1], lohi[:,2]) flattened, boxesND.jl:10)*/
/* ------------------------------------------------------------------------- */

/* is_free_state(v, BB) = @any [!(lo[i] <= v[i] <= hi[i])]   (boxesND.jl:42) */
static int box_point_free(const double *v, const double *lo, const double *hi, int32_t dw)
{
    for (int32_t i = 0; i < dw; ++i)
        if (!(lo[i] <= v[i] && v[i] <= hi[i])) return 1;
    return 0;
}

/* is_free_state(v, BL) = @all [...]   (boxesND.jl:43) */
int32_t orc_point_free_boxes(const double *v, const double *lohi, int32_t M, int32_t dw)
{
    for (int32_t k = 0; k < M; ++k) {
        const double *lo = lohi + (size_t)k * 2 * dw, *hi = lo + dw;
        if (!box_point_free(v, lo, hi, dw)) return 0;
    }
    return 1;
}

/* is_free_motion_broadphase(l, h, BB) = @any [hi[i] < l[i] || lo[i] > h[i]]   (boxesND.jl:44-45) */
static int box_broadphase_free(const double *l, const double *h, const double *lo, const double *hi, int32_t dw)
{
    for (int32_t i = 0; i < dw; ++i)
        if (hi[i] < l[i] || lo[i] > h[i]) return 1;
    return 0;
}

/* is_free_motion(v, w, BB)   (boxesND.jl:46-51 ; blend = utils.jl:41-51) */
static int box_narrow_free(const double *v, const double *w, const double *lo, const double *hi, int32_t dw)
{
    double v_to_w[ORC_MAXD], lambdas[ORC_MAXD];
    for (int32_t i = 0; i < dw; ++i) v_to_w[i] = w[i] - v[i];
    for (int32_t i = 0; i < dw; ++i) {
        double corner = (v[i] < lo[i]) ? lo[i] : hi[i];
        lambdas[i] = (corner - v[i]) / v_to_w[i];        /* may be +-Inf / NaN */
    }
    for (int32_t i = 0; i < dw; ++i) {                   /* @any over i */
        int all = 1;
        for (int32_t j = 0; j < dw; ++j) {               /* @all over j */
            if (i == j) continue;
            double prod = v_to_w[j] * lambdas[i];
            double x = v[j] + prod;                      /* unfused multiply-add */
            if (!(lo[j] <= x && x <= hi[j])) { all = 0; break; }
        }
        if (all) return 0;                               /* hit => not free */
    }
    return 1;
}

/* is_free_motion(v, w, BL)   (boxesND.jl:52-56) */
int32_t orc_motion_free_boxes(const double *v, const double *w, const double *lohi, int32_t M, int32_t dw)
{
    double bb_min[ORC_MAXD], bb_max[ORC_MAXD];
    for (int32_t i = 0; i < dw; ++i) {
        /* map(min, v, w): Julia min(x,y) = ifelse(y < x, y, x) for non-NaN floats */
        bb_min[i] = (w[i] < v[i]) ? w[i] : v[i];
        bb_max[i] = (v[i] < w[i]) ? w[i] : v[i];
    }
    for (int32_t k = 0; k < M; ++k) {
        const double *lo = lohi + (size_t)k * 2 * dw, *hi = lo + dw;
        if (!(box_broadphase_free(bb_min, bb_max, lo, hi, dw) || box_narrow_free(v, w, lo, hi, dw)))
            return 0;
    }
    return 1;
}

/* Individual phases, exported so the golden table can be regenerated. */
int32_t orc_box_broadphase_free(const double *v, const double *w, const double *lo, const double *hi, int32_t dw)
{
    double l[ORC_MAXD], h[ORC_MAXD];
    for (int32_t i = 0; i < dw; ++i) {
        l[i] = (w[i] < v[i]) ? w[i] : v[i];
        h[i] = (v[i] < w[i]) ? w[i] : v[i];
    }
    return box_broadphase_free(l, h, lo, hi, dw);
}
int32_t orc_box_narrow_free(const double *v, const double *w, const double *lo, const double *hi, int32_t dw)
{
    return box_narrow_free(v, w, lo, hi, dw);
}

/* ------------------------------------------------------------------------- */
/* a6: validity wrappers  (src/statespaces.jl:150-158) for the Euclidean      */
/* space: s2w = Identity, collision_waypoints = (v, w) (geometric.jl:20).     */
/* ss_lo/ss_hi may be NULL (no state-space bounds test).                      */
/* ------------------------------------------------------------------------- */

/* in_state_space(v, SS) = @all [lo[i] <= v[i] <= hi[i]]   (statespaces.jl:150) */
int32_t orc_in_state_space(const double *v, const double *ss_lo, const double *ss_hi, int32_t d)
{
    if (!ss_lo || !ss_hi) return 1;
    for (int32_t i = 0; i < d; ++i)
        if (!(ss_lo[i] <= v[i] && v[i] <= ss_hi[i])) return 0;
    return 1;
}

/* is_free_state(v, CC, SS)   (statespaces.jl:151-152) */
int32_t orc_is_free_state(const double *v, int32_t d, const double *lohi, int32_t M,
                          const double *ss_lo, const double *ss_hi)
{
    return orc_in_state_space(v, ss_lo, ss_hi, d) && orc_point_free_boxes(v, lohi, M, d);
}

/* is_free_motion(v, w, CC, SS) with waypoints (v, w): in_state_space(v) && segment test
 * (statespaces.jl:153-158; only the FIRST point of each segment is bounds-checked). */
int32_t orc_is_free_motion(const double *v, const double *w, int32_t d, const double *lohi, int32_t M,
                           const double *ss_lo, const double *ss_hi)
{
    return orc_in_state_space(v, ss_lo, ss_hi, d) && orc_motion_free_boxes(v, w, lohi, M, d);
}

static inline void set_bit(uint64_t *mask, int64_t e, int on)
{
    if (on) mask[e >> 6] |= (uint64_t)1 << (e & 63);
}

/* a10: batch point validity (fmt.jl:31-36).  idx may be NULL (= all points).
 * mask: ceil(n/64) words, bit e <-> point e, LSB first (BitVector.chunks layout). */
void orc_points_free(const double *X, int64_t N, int32_t d, const int64_t *idx, int64_t n,
                     const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    (void)N;
    memset(mask, 0, sizeof(uint64_t) * (size_t)((n + 63) / 64));
    for (int64_t e = 0; e < n; ++e) {
        int64_t i = idx ? idx[e] : e;
        set_bit(mask, e, orc_is_free_state(X + (size_t)i * d, d, lohi, M, ss_lo, ss_hi));
    }
}

/* Batch edge validity: bit e <-> is_free_motion(V[src[e]], V[dst[e]], CC, SS) (parent first, fmt.jl:75). */
void orc_edges_free(const double *X, int64_t N, int32_t d, const int64_t *src, const int64_t *dst, int64_t E,
                    const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    (void)N;
    memset(mask, 0, sizeof(uint64_t) * (size_t)((E + 63) / 64));
    for (int64_t e = 0; e < E; ++e)
        set_bit(mask, e, orc_is_free_motion(X + (size_t)src[e] * d, X + (size_t)dst[e] * d, d, lohi, M, ss_lo, ss_hi));
}

/* Whole-graph edge validity in CSC order: entry e in column x with row y  <->
 * is_free_motion(V[y], V[x])  (y = candidate parent, x = child; fmt.jl:72-75). */
void orc_graph_edges_free(const double *X, int64_t N, int32_t d, const int64_t *colptr, const int64_t *rowval,
                          const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    int64_t nnz = colptr[N];
    memset(mask, 0, sizeof(uint64_t) * (size_t)((nnz + 63) / 64));
    for (int64_t x = 0; x < N; ++x)
        for (int64_t e = colptr[x]; e < colptr[x + 1]; ++e)
            set_bit(mask, e, orc_is_free_motion(X + (size_t)rowval[e] * d, X + (size_t)x * d, d, lohi, M, ss_lo, ss_hi));
}

/* ------------------------------------------------------------------------- */
/* Goals  (src/goals.jl:96,100,111-116): kind 0 = RectangleGoal(lo,hi),       */
/* 1 = BallGoal(center,radius), 2 = PointGoal(pt).  g: [lo(d),hi(d)] |        */
/* [center(d),radius] | [pt(d)].  s2w = Identity.                             */
/* ------------------------------------------------------------------------- */
int32_t orc_is_goal_pt(const double *v, int32_t d, int32_t kind, const double *g)
{
    if (kind == 0) {
        for (int32_t i = 0; i < d; ++i) if (!(g[i] <= v[i] && v[i] <= g[d + i])) return 0;
        return 1;
    } else if (kind == 1) {
        /* norm(v - center) <= radius */
        double s = 0.0;
        for (int32_t i = 0; i < d; ++i) { double t = v[i] - g[i]; double tt = t * t; s = (i == 0) ? tt : s + tt; }
        return sqrt(s) <= g[d];
    } else {
        for (int32_t i = 0; i < d; ++i) if (!(v[i] == g[i])) return 0;
        return 1;
    }
}

/* ------------------------------------------------------------------------- */
/* a1: batch-expand step for a set of z (the body of fmt.jl:70-82 for every   */
/* z in zs, without the set updates).  Every unvisited (W), valid (F) sample  */
/* x that has a forward neighbour among zs is reported once, x ascending:     */
/*   y_min = first argmin over open (H) backward neighbours of C[y] + d(y, x) */
/*   free  = is_free_motion(V[y_min], V[x])                                   */
/* W, H, F: bitmasks (LSB-first). F may be NULL (= checkpts false).           */
/* Returns the number of reported x.                                          */
/* ------------------------------------------------------------------------- */
static inline int get_bit(const uint64_t *m, int64_t i) { return (int)((m[i >> 6] >> (i & 63)) & 1u); }

int64_t orc_expand(const double *X, int64_t N, int32_t d, double r,
                   const uint64_t *W, const uint64_t *H, const uint64_t *F, const double *C,
                   const int64_t *zs, int64_t nz,
                   const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                   int64_t *xs, int64_t *ymin, double *cmin, uint8_t *freeflag)
{
    const double r2 = r * r;
    int64_t nx = 0;
    for (int64_t x = 0; x < N; ++x) {
        if (!get_bit(W, x)) continue;                      /* filter_neighborhood(., W) */
        if (F && !get_bit(F, x)) continue;                 /* fmt.jl:71 */
        int reached = 0;
        for (int64_t iz = 0; iz < nz && !reached; ++iz) {
            int64_t z = zs[iz];
            if (z != x && orc_sqdist(X + (size_t)z * d, X + (size_t)x * d, d) <= r2) reached = 1;
        }
        if (!reached) continue;
        int64_t best = -1; double bc = 0.0;
        for (int64_t y = 0; y < N; ++y) {                  /* fmt.jl:72-74 */
            if (y == x || !get_bit(H, y)) continue;
            double d2 = orc_sqdist(X + (size_t)x * d, X + (size_t)y * d, d);
            if (!(d2 <= r2)) continue;
            double c = C[y] + sqrt(d2);
            if (best < 0 || c < bc) { best = y; bc = c; }
        }
        xs[nx] = x; ymin[nx] = best; cmin[nx] = (best >= 0) ? bc : 0.0;
        freeflag[nx] = (best >= 0) ? (uint8_t)orc_is_free_motion(X + (size_t)best * d, X + (size_t)x * d, d, lohi, M, ss_lo, ss_hi) : 0;
        ++nx;
    }
    return nx;
}

/* ------------------------------------------------------------------------- */
/* a1: fmtstar!  (src/planners/fmt.jl:3-119), connections = :R only (the :K   */
/* branch references undefined functions, fmt.jl:17-19).                      */
/* Neighbourhoods come from a lazily filled cache (MutableNNC,                */
/* nearneighbors.jl:129-135) -- nn_mode 0 = brute scan, 1 = KD-tree.          */
/* ------------------------------------------------------------------------- */

typedef struct { int64_t *inds; double *ds; int64_t k; } orc_nbr;

typedef struct { double *pri; int64_t *idx; int64_t n, cap; } orc_heap;

static int heap_less(const orc_heap *h, int64_t a, int64_t b)
{
    if (h->pri[a] < h->pri[b]) return 1;
    if (h->pri[a] > h->pri[b]) return 0;
    return h->idx[a] < h->idx[b];        /* declared tie-break: lowest sample index */
}
static void heap_swap(orc_heap *h, int64_t a, int64_t b)
{
    double p = h->pri[a]; h->pri[a] = h->pri[b]; h->pri[b] = p;
    int64_t i = h->idx[a]; h->idx[a] = h->idx[b]; h->idx[b] = i;
}
static void heap_push(orc_heap *h, int64_t i, double p)
{
    if (h->n == h->cap) {
        h->cap = h->cap ? 2 * h->cap : 1024;
        h->pri = (double *)realloc(h->pri, sizeof(double) * (size_t)h->cap);
        h->idx = (int64_t *)realloc(h->idx, sizeof(int64_t) * (size_t)h->cap);
    }
    int64_t c = h->n++;
    h->pri[c] = p; h->idx[c] = i;
    while (c > 0) { int64_t par = (c - 1) / 2; if (heap_less(h, c, par)) { heap_swap(h, c, par); c = par; } else break; }
}
static int64_t heap_pop(orc_heap *h)
{
    int64_t top = h->idx[0];
    --h->n;
    if (h->n > 0) {
        h->pri[0] = h->pri[h->n]; h->idx[0] = h->idx[h->n];
        int64_t c = 0;
        for (;;) {
            int64_t l = 2 * c + 1, rr = l + 1, m = c;
            if (l < h->n && heap_less(h, l, m)) m = l;
            if (rr < h->n && heap_less(h, rr, m)) m = rr;
            if (m == c) break;
            heap_swap(h, c, m); c = m;
        }
    }
    return top;
}

typedef struct {
    int32_t status;             /* 1 = :solved, 0 = :failed */
    double cost;                /* C[z] */
    int64_t z;                  /* final dequeued node */
    int64_t collision_checks;   /* P.CC.count */
    int64_t path_len;
    int64_t nn_queries;         /* number of inball evaluations (cache misses) */
} orc_fmt_result;

/*
 * A: parent (0-based, -1 = none) ; C: cost-to-come ; path: up to N entries.
 * init_idx 0-based.  F (checkpts bitmap) is computed inside like fmt.jl:31-36
 * when checkpts != 0.  Returns 0, or -1 when the initial state is infeasible
 * (fmt.jl:24-29).
 */
int32_t orc_fmtstar(const double *X, int64_t N, int32_t d, double r, int64_t init_idx, int32_t checkpts,
                    int32_t goal_kind, const double *goal,
                    const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                    int32_t nn_mode,
                    int64_t *A, double *C, int64_t *path, orc_fmt_result *res)
{
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    const double *init = X + (size_t)init_idx * d;
    if (!orc_is_free_state(init, d, lohi, M, ss_lo, ss_hi)) return -1;     /* fmt.jl:24-29 */

    uint8_t *F = NULL;
    if (checkpts) {                                                        /* fmt.jl:31-36 */
        F = (uint8_t *)malloc((size_t)N);
        for (int64_t i = 0; i < N; ++i) F[i] = (uint8_t)orc_is_free_state(X + (size_t)i * d, d, lohi, M, ss_lo, ss_hi);
    }
    uint8_t *Wm = (uint8_t *)malloc((size_t)N), *Hm = (uint8_t *)calloc((size_t)N, 1);   /* fmt.jl:43-46 */
    memset(Wm, 1, (size_t)N);
    for (int64_t i = 0; i < N; ++i) { A[i] = -1; C[i] = 0.0; }
    orc_nbr *cache = (orc_nbr *)calloc((size_t)N, sizeof(orc_nbr));
    orc_kdtree *T = (nn_mode == 1) ? orc_kdtree_build(X, N, d) : NULL;
    int64_t *tmp_i = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    double *tmp_d = (double *)malloc(sizeof(double) * (size_t)N);
    int64_t *Hnew = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);

    orc_heap heap = {0};
    Wm[init_idx] = 0; Hm[init_idx] = 1;                                    /* fmt.jl:48-51 */
    heap_push(&heap, init_idx, 0.0);
    int64_t z = heap_pop(&heap);                                           /* fmt.jl:66 */
    int64_t count = 0, queries = 0;

#define NBR(v) do { if (!cache[v].inds) { \
        int64_t k_ = (nn_mode == 1) ? orc_kdtree_inball(T, v, r, tmp_i, tmp_d, N) \
                                    : orc_inball(X, N, d, v, r, 0, tmp_i, tmp_d, N); \
        cache[v].k = k_; \
        cache[v].inds = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k_ > 0 ? k_ : 1)); \
        cache[v].ds = (double *)malloc(sizeof(double) * (size_t)(k_ > 0 ? k_ : 1)); \
        memcpy(cache[v].inds, tmp_i, sizeof(int64_t) * (size_t)k_); \
        memcpy(cache[v].ds, tmp_d, sizeof(double) * (size_t)k_); ++queries; } } while (0)

    while (!orc_is_goal_pt(X + (size_t)z * d, d, goal_kind, goal)) {       /* fmt.jl:68 */
        int64_t nnew = 0;
        NBR(z);
        const orc_nbr *nz = &cache[z];
        for (int64_t a = 0; a < nz->k; ++a) {                              /* fmt.jl:70 */
            int64_t x = nz->inds[a];
            if (!Wm[x]) continue;                                          /* filter_neighborhood(., W) */
            if (checkpts && !F[x]) continue;                               /* fmt.jl:71 */
            NBR(x);
            const orc_nbr *nx = &cache[x];                                 /* fmt.jl:72 */
            int64_t y_min = -1; double c_min = 0.0;
            for (int64_t b = 0; b < nx->k; ++b) {                          /* fmt.jl:73 findmin (first min) */
                int64_t y = nx->inds[b];
                if (!Hm[y]) continue;
                double c = C[y] + nx->ds[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; }
            }
            if (y_min < 0) continue;    /* unreachable for a symmetric metric (z itself is open) */
            /* boxesND.jl:26: the counter sits inside is_free_motion(v, w, CC), which statespaces.jl:155-157
             * only reaches when in_state_space(v) held */
            if (orc_in_state_space(X + (size_t)y_min * d, ss_lo, ss_hi, d)) ++count;
            if (orc_is_free_motion(X + (size_t)y_min * d, X + (size_t)x * d, d, lohi, M, ss_lo, ss_hi)) { /* fmt.jl:75 */
                A[x] = y_min; C[x] = c_min;                                /* fmt.jl:76-77 */
                heap_push(&heap, x, c_min);                                /* fmt.jl:78 */
                Hnew[nnew++] = x;                                          /* fmt.jl:79 */
                Wm[x] = 0;                                                 /* fmt.jl:80 */
            }
        }
        for (int64_t a = 0; a < nnew; ++a) Hm[Hnew[a]] = 1;                /* fmt.jl:83 */
        Hm[z] = 0;                                                         /* fmt.jl:84 */
        if (heap.n > 0) z = heap_pop(&heap); else break;                   /* fmt.jl:85-89 */
    }
#undef NBR

    /* path back-trace (fmt.jl:92-101); the reference walks until index 1 (= 0 here) */
    int64_t len = 0, cur = z;
    int64_t *rev = tmp_i;
    rev[len++] = cur;
    while (cur != 0) {
        cur = A[cur];
        if (cur < 0) break;
        rev[len++] = cur;
    }
    for (int64_t i = 0; i < len; ++i) path[i] = rev[len - 1 - i];

    res->status = orc_is_goal_pt(X + (size_t)z * d, d, goal_kind, goal);
    res->cost = C[z];
    res->z = z;
    res->collision_checks = count;
    res->path_len = len;
    res->nn_queries = queries;

    for (int64_t i = 0; i < N; ++i) { free(cache[i].inds); free(cache[i].ds); }
    free(cache); free(Wm); free(Hm); free(F); free(tmp_i); free(tmp_d); free(Hnew);
    free(heap.pri); free(heap.idx);
    orc_kdtree_free(T);
    return 0;
}

/* Same loop, but consuming a prebuilt graph (ImmutableNNC CSC, nearneighbors.jl:23-28,128)
 * and a precomputed per-edge free mask -- the "eager graph" driver the HIP library feeds.
 * free_mask may be NULL (edges are then checked lazily like above). */
int32_t orc_fmtstar_graph(const double *X, int64_t N, int32_t d, int64_t init_idx,
                          const int64_t *colptr, const int64_t *rowval, const double *nzval,
                          const uint64_t *free_mask, const uint64_t *Fmask,
                          int32_t goal_kind, const double *goal,
                          const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                          int64_t *A, double *C, int64_t *path, orc_fmt_result *res)
{
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    if (!orc_is_free_state(X + (size_t)init_idx * d, d, lohi, M, ss_lo, ss_hi)) return -1;
    uint8_t *Wm = (uint8_t *)malloc((size_t)N), *Hm = (uint8_t *)calloc((size_t)N, 1);
    memset(Wm, 1, (size_t)N);
    for (int64_t i = 0; i < N; ++i) { A[i] = -1; C[i] = 0.0; }
    int64_t *Hnew = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *rev = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    orc_heap heap = {0};
    Wm[init_idx] = 0; Hm[init_idx] = 1;
    heap_push(&heap, init_idx, 0.0);
    int64_t z = heap_pop(&heap), count = 0;
    while (!orc_is_goal_pt(X + (size_t)z * d, d, goal_kind, goal)) {
        int64_t nnew = 0;
        for (int64_t a = colptr[z]; a < colptr[z + 1]; ++a) {
            int64_t x = rowval[a];
            if (!Wm[x]) continue;
            if (Fmask && !get_bit(Fmask, x)) continue;
            int64_t y_min = -1, e_min = -1; double c_min = 0.0;
            for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {
                int64_t y = rowval[b];
                if (!Hm[y]) continue;
                double c = C[y] + nzval[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; e_min = b; }
            }
            if (y_min < 0) continue;
            if (orc_in_state_space(X + (size_t)y_min * d, ss_lo, ss_hi, d)) ++count;
            int fr = free_mask ? get_bit(free_mask, e_min)
                               : orc_is_free_motion(X + (size_t)y_min * d, X + (size_t)x * d, d, lohi, M, ss_lo, ss_hi);
            if (fr) { A[x] = y_min; C[x] = c_min; heap_push(&heap, x, c_min); Hnew[nnew++] = x; Wm[x] = 0; }
        }
        for (int64_t a = 0; a < nnew; ++a) Hm[Hnew[a]] = 1;
        Hm[z] = 0;
        if (heap.n > 0) z = heap_pop(&heap); else break;
    }
    int64_t len = 0, cur = z;
    rev[len++] = cur;
    while (cur != 0) { cur = A[cur]; if (cur < 0) break; rev[len++] = cur; }
    for (int64_t i = 0; i < len; ++i) path[i] = rev[len - 1 - i];
    res->status = orc_is_goal_pt(X + (size_t)z * d, d, goal_kind, goal);
    res->cost = C[z]; res->z = z; res->collision_checks = count; res->path_len = len; res->nn_queries = 0;
    free(Wm); free(Hm); free(Hnew); free(rev); free(heap.pri); free(heap.idx);
    return 0;
}

double orc_mp_sin(double x) { return mp_sin(x); }
double orc_mp_cos(double x) { return mp_cos(x); }
double orc_mp_atan2(double y, double x) { return mp_atan2(y, x); }
double orc_mp_acos(double x) { return mp_acos(x); }

/* The portable input stream of the synthetic workloads: SplitMix64 (Steele, Lea, Flood, "Fast splittable
 * pseudorandom number generators", OOPSLA 2014; Vigna's splitmix64.c), used counter-based.  Draw i (0-based) of seed s:
 *   z = s + (i + 1) * 0x9E3779B97F4A7C15;  z = (z ^ z >> 30) * 0xBF58476D1CE4E5B9;  z = (z ^ z >> 27) * 0x94D049BB133111EB;  z ^ z >> 31
 * and u = (z >> 11) * 2^-53.  Not a reference function (the reference uses Julia's unseeded global RNG). */
uint64_t orc_splitmix64(uint64_t seed, uint64_t i)
{
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_stream_uniform(uint64_t seed, uint64_t offset, int64_t n, double *u)
{
    for (int64_t i = 0; i < n; ++i) u[i] = (double)(orc_splitmix64(seed, offset + (uint64_t)i) >> 11) * 0x1.0p-53;
}

/* a8: Euclidean steer, src/statespaces/geometric.jl:18-19 and the partial propagate of src/statespaces.jl:79-81.
 *   steering_control(M::Euclidean, v, w) = StepControl(evaluate(M, v, w), normalize(w - v))
 *   propagate(M::Euclidean, v, u::StepControl) = v + u.t * u.u
 * Declared arithmetic: t = orc_dist (the canonical index-order norm); normalize = inv(norm) * (w - v) (StaticArrays /
 * Base.normalize multiply by the reciprocal); v + t*u unfused. */
void orc_euclid_steer(const double *v, const double *w, int32_t d, double *t, double *u)
{
    const double n = orc_dist(v, w, d);
    const double inv = 1.0 / n;
    *t = n;
    for (int32_t i = 0; i < d; ++i) u[i] = inv * (w[i] - v[i]);
}

void orc_euclid_propagate(const double *v, int32_t d, double t, const double *u, int32_t have_s, double s, double *out)
{
    double step = t;
    if (have_s) {
        if (s <= 0.0) { for (int32_t i = 0; i < d; ++i) out[i] = v[i]; return; }     /* statespaces.jl:80 */
        if (!(s >= t)) step = s;
    }
    for (int32_t i = 0; i < d; ++i) { double p = step * u[i]; out[i] = v[i] + p; }
}

/* Wavefront (batched) form of the same loop on a prebuilt graph: the checker of libmpfmt's device-resident driver
 * (mpfmt_fmtstar_wavefront).  Not a reference function: the reference pops one node per iteration (fmt.jl:66,85-89); this
 * runs the loop BODY fmt.jl:70-82 for the whole batch Z = { z in H : C[z] <= min_H C + band } (or, single != 0, the one
 * lowest (cost, index) node) against the same (W, H, C), then the deferred update fmt.jl:83-84 for the batch, and stops when a
 * batch holds a goal node (fmt.jl:68) with z = the goal node of lowest (cost, index).  With single != 0 it is
 * orc_fmtstar_graph step for step.  iters (may be NULL) receives the number of batches, the last (unexpanded) one included. */
/* general form: directed cost graphs too (quasi-metric spaces, fmt.jl with QuasiMetricNN): forward sets = rows of the cost
 * matrix given as CSR (rowptr / colidx; NULL = symmetric graph, forward set = column), nseg = per-entry count of the segment
 * tests the reference would make (NULL = one per examined edge, behind the in_state_space short circuit), gd = coordinates the
 * goal predicate reads (POINT goals compare all d), init_free = is_free_state(init) as the caller's space defines it. */
static int32_t wavefront_impl(const double *X, int64_t N, int32_t d, int32_t gd, int64_t init_idx, int32_t init_free,
                              const int64_t *colptr, const int64_t *rowval, const double *nzval,
                              const int64_t *rowptr, const int64_t *colidx,
                              const uint64_t *free_mask, const uint8_t *nseg, const uint64_t *Fmask,
                              int32_t goal_kind, const double *goal,
                              const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                              double band, int32_t single,
                              int64_t *A, double *C, int64_t *path, orc_fmt_result *res, int64_t *iters)
{
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    if (!init_free) return -1;
    if (!rowptr) { rowptr = colptr; colidx = rowval; }
#define WF_GOAL(p) ((goal_kind == 2) ? orc_is_goal_pt((p), d, 2, goal) : orc_is_goal_pt((p), gd, goal_kind, goal))
    uint8_t *Wm = (uint8_t *)malloc((size_t)N), *Hm = (uint8_t *)calloc((size_t)N, 1), *Zm = (uint8_t *)calloc((size_t)N, 1);
    uint8_t *cand = (uint8_t *)calloc((size_t)N, 1);
    memset(Wm, 1, (size_t)N);
    for (int64_t i = 0; i < N; ++i) { A[i] = -1; C[i] = 0.0; }
    int64_t *zs = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *zprev = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *xs = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *cx = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *cy = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    double *cc = (double *)malloc(sizeof(double) * (size_t)N);
    int64_t *rev = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    Wm[init_idx] = 0; Hm[init_idx] = 1;
    int64_t count = 0, z_final = init_idx, nprev = 0, it = 0;
    int32_t status = 0;
    for (;;) {
        /* lowest (cost, index) open node */
        int64_t im = -1; double cm = 0.0;
        for (int64_t i = 0; i < N; ++i) if (Hm[i] && (im < 0 || C[i] < cm)) { cm = C[i]; im = i; }
        if (im < 0) {                                     /* fmt.jl:85-89 break: z stays at the last dequeued node */
            int64_t bi = -1; double bc = 0.0;
            for (int64_t k = 0; k < nprev; ++k) { int64_t i = zprev[k]; if (bi < 0 || C[i] > bc || (C[i] == bc && i > bi)) { bc = C[i]; bi = i; } }
            if (bi >= 0) z_final = bi;
            break;
        }
        ++it;
        int64_t nz = 0;
        const double thr = cm + band;
        for (int64_t i = 0; i < N; ++i) if (Hm[i] && (single ? i == im : C[i] <= thr)) { zs[nz++] = i; Zm[i] = 1; }
        /* fmt.jl:68 on the batch */
        int64_t gi = -1; double gc = 0.0;
        for (int64_t k = 0; k < nz; ++k) {
            int64_t i = zs[k];
            if (WF_GOAL(X + (size_t)i * d) && (gi < 0 || C[i] < gc)) { gc = C[i]; gi = i; }
        }
        if (gi >= 0) { z_final = gi; status = 1; break; }
        /* fmt.jl:70-71: x unvisited, valid, adjacent to the batch */
        int64_t nx = 0;
        for (int64_t k = 0; k < nz; ++k)
            for (int64_t a = rowptr[zs[k]]; a < rowptr[zs[k] + 1]; ++a) {
                int64_t x = colidx[a];
                if (!Wm[x] || cand[x]) continue;
                if (Fmask && !get_bit(Fmask, x)) continue;
                cand[x] = 1; xs[nx++] = x;
            }
        /* fmt.jl:72-80 against the same (W, H, C) for every x */
        int64_t nconn = 0;
        for (int64_t q = 0; q < nx; ++q) {
            int64_t x = xs[q];
            cand[x] = 0;
            int64_t y_min = -1, e_min = -1; double c_min = 0.0;
            for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {
                int64_t y = rowval[b];
                if (!Hm[y]) continue;
                double c = C[y] + nzval[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; e_min = b; }
            }
            if (y_min < 0) continue;
            if (nseg) count += nseg[e_min];
            else if (orc_in_state_space(X + (size_t)y_min * d, ss_lo, ss_hi, d)) ++count;
            int fr = free_mask ? get_bit(free_mask, e_min)
                               : orc_is_free_motion(X + (size_t)y_min * d, X + (size_t)x * d, d, lohi, M, ss_lo, ss_hi);
            if (fr) { cx[nconn] = x; cy[nconn] = y_min; cc[nconn] = c_min; ++nconn; }
        }
        for (int64_t k = 0; k < nconn; ++k) { A[cx[k]] = cy[k]; C[cx[k]] = cc[k]; Wm[cx[k]] = 0; }
        for (int64_t k = 0; k < nz; ++k) { Hm[zs[k]] = 0; Zm[zs[k]] = 0; zprev[k] = zs[k]; }      /* fmt.jl:84 */
        nprev = nz;
        for (int64_t k = 0; k < nconn; ++k) Hm[cx[k]] = 1;                                          /* fmt.jl:83 */
    }
    int64_t len = 0, cur = z_final;
    rev[len++] = cur;
    while (cur != 0) { cur = A[cur]; if (cur < 0) break; rev[len++] = cur; }
    for (int64_t i = 0; i < len; ++i) path[i] = rev[len - 1 - i];
    res->status = status;
    res->cost = C[z_final]; res->z = z_final; res->collision_checks = count; res->path_len = len; res->nn_queries = 0;
    if (iters) *iters = it;
    free(Wm); free(Hm); free(Zm); free(cand); free(zs); free(zprev); free(xs); free(cx); free(cy); free(cc); free(rev);
#undef WF_GOAL
    return 0;
}

int32_t orc_fmt_wavefront_graph(const double *X, int64_t N, int32_t d, int64_t init_idx,
                                const int64_t *colptr, const int64_t *rowval, const double *nzval,
                                const uint64_t *free_mask, const uint64_t *Fmask,
                                int32_t goal_kind, const double *goal,
                                const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                                double band, int32_t single,
                                int64_t *A, double *C, int64_t *path, orc_fmt_result *res, int64_t *iters)
{
    return wavefront_impl(X, N, d, d, init_idx, orc_is_free_state(X + (size_t)init_idx * d, d, lohi, M, ss_lo, ss_hi), colptr, rowval, nzval,
                          NULL, NULL, free_mask, NULL, Fmask, goal_kind, goal, lohi, M, ss_lo, ss_hi, band, single, A, C, path, res, iters);
}

/* the batched loop over a directed cost graph with precomputed edge bits and per-edge segment counts (double integrator,
 * Dubins): checker of mpfmt_di_fmtstar_wavefront.  free_mask and nseg are required; Fmask NULL = checkpts false. */
int32_t orc_fmt_wavefront_directed(const double *X, int64_t N, int32_t d, int32_t gd, int64_t init_idx, int32_t init_free,
                                   const int64_t *colptr, const int64_t *rowval, const double *nzval,
                                   const int64_t *rowptr, const int64_t *colidx,
                                   const uint64_t *free_mask, const uint8_t *nseg, const uint64_t *Fmask,
                                   int32_t goal_kind, const double *goal, double band, int32_t single,
                                   int64_t *A, double *C, int64_t *path, orc_fmt_result *res, int64_t *iters)
{
    return wavefront_impl(X, N, d, gd, init_idx, init_free, colptr, rowval, nzval, rowptr, colidx, free_mask, nseg, Fmask, goal_kind, goal,
                          NULL, 0, NULL, NULL, band, single, A, C, path, res, iters);
}

/* ------------------------------------------------------------------------- */
/* a9: double-integrator LQ steer  (src/statespaces/linearquadratic.jl).      */
/* The reference generates cost/dcost/ddcost/x closures with SymPy at load    */
/* time (linearquadratic.jl:126-157); SymPy prints them in a version-dependent*/
/* order, so the build declares the closed forms below (SURVEY.md row a9,     */
/* R = rho*I, A = [0 I; 0 0], B = [0; I], c = 0) evaluated in written order.  */
/* State x = (p, v) in R^{2m}.                                                */
/* ------------------------------------------------------------------------- */

typedef struct { double a, b, c; } di_coef;   /* |p|^2, p.(v0+v1), |v0|^2+v0.v1+|v1|^2 */

static di_coef di_coefs(const double *x0, const double *x1, int32_t m)
{
    di_coef k = {0.0, 0.0, 0.0};
    for (int32_t i = 0; i < m; ++i) {
        double p = x1[i] - x0[i];
        double v0 = x0[m + i], v1 = x1[m + i];
        k.a = k.a + p * p;
        k.b = k.b + p * (v0 + v1);
        k.c = k.c + ((v0 * v0 + v0 * v1) + v1 * v1);
    }
    return k;
}

/* NOTE: p above is the raw position difference; the LQ cost uses the drift-corrected
 * difference (y - xbar(t)) = (p - t*v0, v1 - v0); expanding it gives these closed forms. */
static double di_cost(di_coef k, double rho, double t)
{
    double t2 = t * t, t3 = t2 * t;
    return t + rho * ((12.0 * k.a / t3 - 12.0 * k.b / t2) + 4.0 * k.c / t);
}
static double di_dcost(di_coef k, double rho, double t)
{
    double t2 = t * t, t3 = t2 * t, t4 = t2 * t2;
    return 1.0 - rho * ((36.0 * k.a / t4 - 24.0 * k.b / t3) + 4.0 * k.c / t2);
}
static double di_ddcost(di_coef k, double rho, double t)
{
    double t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
    return rho * ((144.0 * k.a / t5 - 72.0 * k.b / t4) + 8.0 * k.c / t3);
}

double orc_di_cost(const double *x0, const double *x1, int32_t m, double rho, double t) { return di_cost(di_coefs(x0, x1, m), rho, t); }
double orc_di_dcost(const double *x0, const double *x1, int32_t m, double rho, double t) { return di_dcost(di_coefs(x0, x1, m), rho, t); }
double orc_di_ddcost(const double *x0, const double *x1, int32_t m, double rho, double t) { return di_ddcost(di_coefs(x0, x1, m), rho, t); }

/* topt_newton  (linearquadratic.jl:175-190), tol = 1e-6 */
static double di_topt_newton(di_coef k, double rho, double tm)
{
    const double tol = 1e-6;
    double b = tm;
    if (di_dcost(k, rho, b) < 0) return tm;
    double a = tm / 100;
    while (di_dcost(k, rho, a) > 0) a /= 2;
    double t = tm / 2;
    double cdval = di_dcost(k, rho, t);
    while (fabs(cdval) > tol && fabs(a - b) > tol) {
        t = t - cdval / di_ddcost(k, rho, t);
        if (t < a || t > b) t = (a + b) / 2;
        cdval = di_dcost(k, rho, t);
        if (cdval > 0) b = t; else a = t;
    }
    return t;
}

/* steer(L, x0, x1, r) -> (cost, t)   (linearquadratic.jl:191-195) */
void orc_di_steer(const double *x0, const double *x1, int32_t m, double rho, double r, double *cost, double *topt)
{
    int same = 1;
    for (int32_t i = 0; i < 2 * m; ++i) if (x0[i] != x1[i]) { same = 0; break; }
    if (same) { *cost = 0.0; *topt = 0.0; return; }
    di_coef k = di_coefs(x0, x1, m);
    double t = di_topt_newton(k, rho, r);
    *cost = di_cost(k, rho, t);
    *topt = t;
}

/* x(v, w, t, s): state on the optimal trajectory at time s  (closed form of the
 * SymPy-generated `x` closure, linearquadratic.jl:137-138,156). out: 2m doubles. */
void orc_di_state(const double *x0, const double *x1, int32_t m, double rho, double t, double s, double *out)
{
    (void)rho;   /* rho cancels in the state trajectory */
    double t2 = t * t, t3 = t2 * t;
    double s2 = s * s, s3 = s2 * s;
    for (int32_t i = 0; i < m; ++i) {
        double v0 = x0[m + i], v1 = x1[m + i];
        double dp = (x1[i] - x0[i]) - t * v0;
        double dv = v1 - v0;
        double d1 = 12.0 * dp / t3 - 6.0 * dv / t2;
        double d2 = -6.0 * dp / t2 + 4.0 * dv / t;
        double e = (t - s) * d1 + d2;
        out[i] = (x0[i] + s * v0) + (s3 / 3.0 * d1 + s2 / 2.0 * e);
        out[m + i] = v0 + (s2 / 2.0 * d1 + s * e);
    }
}

/* collision_waypoints(d::LinearQuadratic, v, w): 5 states at s = linspace(0, t, 5)
 * (linearquadratic.jl:85-88).  wps: 5 x 2m row-major. */
void orc_di_waypoints(const double *x0, const double *x1, int32_t m, double rho, double r, double *wps)
{
    double cost, t;
    orc_di_steer(x0, x1, m, rho, r, &cost, &t);
    for (int32_t q = 0; q < 5; ++q) {
        /* linspace(0, t, 5)[q+1] = q*t/4 up to rounding; declared form: (q/4)*t with exact endpoints */
        double s = (q == 4) ? t : ((double)q / 4.0) * t;
        orc_di_state(x0, x1, m, rho, t, s, wps + (size_t)q * 2 * m);
    }
}

/* is_free_motion(v, w, CC, SS) for the LQ space (statespaces.jl:153-158 with 5 waypoints,
 * workspace = first m coordinates, C = [I 0], linearquadratic.jl:51-52). */
int32_t orc_di_is_free_motion(const double *x0, const double *x1, int32_t m, double rho, double r,
                              const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi)
{
    double wps[5 * ORC_MAXD];
    orc_di_waypoints(x0, x1, m, rho, r, wps);
    for (int32_t q = 0; q < 4; ++q) {
        const double *a = wps + (size_t)q * 2 * m, *b = wps + (size_t)(q + 1) * 2 * m;
        if (!(orc_in_state_space(a, ss_lo, ss_hi, 2 * m) && orc_motion_free_boxes(a, b, lohi, M, m))) return 0;
    }
    return 1;
}

/* steer_pairwise + helper_data_structures (linearquadratic.jl:68-77,196-225):
 * all ordered pairs (i -> j), i != j, prefilter dcost(r) > 0 is implied by the
 * final test, keep cost <= r.  Emits the sparse cost matrix Dmat (rows i = from,
 * columns j = to) in CSC order (column j lists sources i ascending) = DSB;
 * DSF is its transpose.  Two-phase: pass rowval == NULL to count. */
int64_t orc_di_pairwise(const double *X, int64_t N, int32_t m, double rho, double r,
                        int64_t *colptr, int64_t *rowval, double *nzval, double *tval)
{
    int32_t n = 2 * m;
    int64_t nnz = 0;
    colptr[0] = 0;
    for (int64_t j = 0; j < N; ++j) {
        for (int64_t i = 0; i < N; ++i) {
            if (i == j) continue;
            const double *x0 = X + (size_t)i * n, *x1 = X + (size_t)j * n;
            di_coef k = di_coefs(x0, x1, m);
            /* candidate filter `cd .> 0` (linearquadratic.jl:213): cd == dcost(r) */
            if (!(di_dcost(k, rho, r) > 0)) continue;
            double cost, t;
            orc_di_steer(x0, x1, m, rho, r, &cost, &t);
            if (cost <= r) {
                if (rowval) { rowval[nnz] = i; nzval[nnz] = cost; if (tval) tval[nnz] = t; }
                ++nnz;
            }
        }
        colptr[j + 1] = nnz;
    }
    return nnz;
}

/* fmt.jl:39 radius rule, evaluated left-to-right like the Julia expression. */
double orc_fmt_radius(double rm, int32_t d, double free_volume_ub, int64_t N)
{
    double dd = (double)d;
    double zeta = pow(M_PI, dd / 2) / tgamma(dd / 2 + 1);
    double inner = 1 / dd * free_volume_ub / zeta * log((double)N) / (double)N;
    return rm * 2 * pow(inner, 1 / dd);
}

/* ------------------------------------------------------------------------- */
/* a1 for the LinearQuadratic (double integrator) quasi-metric space:         */
/* fmtstar! (fmt.jl:3-119) with nearF = inballF! = row z of the sparse cost   */
/* matrix (DSF = Dmat', linearquadratic.jl:73) and nearB = column x (DSB).    */
/* colptr/rowval/nzval: Dmat in CSC (column j = sources i that reach j).      */
/* Goal kinds: RECT/BALL act on the workspace (first m coordinates, C=[I 0]); */
/* POINT is a StateGoal (exact state equality, goals.jl:128-131).             */
/* ------------------------------------------------------------------------- */
static int di_is_goal(const double *v, int32_t m, int32_t kind, const double *g)
{
    if (kind == 2) {
        for (int32_t i = 0; i < 2 * m; ++i) if (!(v[i] == g[i])) return 0;
        return 1;
    }
    return orc_is_goal_pt(v, m, kind, g);
}

int32_t orc_di_fmtstar(const double *X, int64_t N, int32_t m, double rho, double r, int64_t init_idx, int32_t checkpts,
                       const int64_t *colptr, const int64_t *rowval, const double *nzval,
                       int32_t goal_kind, const double *goal,
                       const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                       int64_t *A, double *C, int64_t *path, orc_fmt_result *res)
{
    const int32_t n = 2 * m;
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
    /* is_free_state(v, CC, SS) = in_state_space(v) && is_free_state(C*v, CC) (statespaces.jl:151-152) */
#define DI_FREE_STATE(p) (orc_in_state_space((p), ss_lo, ss_hi, n) && orc_point_free_boxes((p), lohi, M, m))
    if (!DI_FREE_STATE(X + (size_t)init_idx * n)) return -1;
    uint8_t *F = NULL;
    if (checkpts) {
        F = (uint8_t *)malloc((size_t)N);
        for (int64_t i = 0; i < N; ++i) F[i] = (uint8_t)DI_FREE_STATE(X + (size_t)i * n);
    }
#undef DI_FREE_STATE
    /* CSR of Dmat (forward sets): row i lists targets j ascending */
    int64_t nnz = colptr[N];
    int64_t *rowptr = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t));
    int64_t *colidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    for (int64_t e = 0; e < nnz; ++e) rowptr[rowval[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) rowptr[i + 1] += rowptr[i];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    memcpy(cur, rowptr, sizeof(int64_t) * (size_t)N);
    for (int64_t j = 0; j < N; ++j)
        for (int64_t e = colptr[j]; e < colptr[j + 1]; ++e) colidx[cur[rowval[e]]++] = j;

    uint8_t *Wm = (uint8_t *)malloc((size_t)N), *Hm = (uint8_t *)calloc((size_t)N, 1);
    memset(Wm, 1, (size_t)N);
    for (int64_t i = 0; i < N; ++i) { A[i] = -1; C[i] = 0.0; }
    int64_t *Hnew = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int64_t *rev = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    orc_heap heap = {0};
    Wm[init_idx] = 0; Hm[init_idx] = 1;
    heap_push(&heap, init_idx, 0.0);
    int64_t z = heap_pop(&heap), count = 0;
    while (!di_is_goal(X + (size_t)z * n, m, goal_kind, goal)) {
        int64_t nnew = 0;
        for (int64_t a = rowptr[z]; a < rowptr[z + 1]; ++a) {            /* nearF(V, z, r, W) */
            int64_t x = colidx[a];
            if (!Wm[x]) continue;
            if (checkpts && !F[x]) continue;
            int64_t y_min = -1; double c_min = 0.0;
            for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {        /* nearB(V, x, r, H) */
                int64_t y = rowval[b];
                if (!Hm[y]) continue;
                double c = C[y] + nzval[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; }
            }
            if (y_min < 0) continue;
            /* CC.count is incremented once per workspace segment actually tested (boxesND.jl:26 inside the
             * @all of statespaces.jl:155): count the segments up to and including the first failure */
            {
                double wps[5 * ORC_MAXD];
                orc_di_waypoints(X + (size_t)y_min * n, X + (size_t)x * n, m, rho, r, wps);
                int ok = 1;
                for (int32_t q = 0; q < 4 && ok; ++q) {
                    const double *p = wps + (size_t)q * n, *pn = wps + (size_t)(q + 1) * n;
                    if (!orc_in_state_space(p, ss_lo, ss_hi, n)) { ok = 0; break; }
                    ++count;
                    if (!orc_motion_free_boxes(p, pn, lohi, M, m)) ok = 0;
                }
                if (ok) { A[x] = y_min; C[x] = c_min; heap_push(&heap, x, c_min); Hnew[nnew++] = x; Wm[x] = 0; }
            }
        }
        for (int64_t a = 0; a < nnew; ++a) Hm[Hnew[a]] = 1;
        Hm[z] = 0;
        if (heap.n > 0) z = heap_pop(&heap); else break;
    }
    int64_t len = 0, c2 = z;
    rev[len++] = c2;
    while (c2 != 0) { c2 = A[c2]; if (c2 < 0) break; rev[len++] = c2; }
    for (int64_t i = 0; i < len; ++i) path[i] = rev[len - 1 - i];
    res->status = di_is_goal(X + (size_t)z * n, m, goal_kind, goal);
    res->cost = C[z]; res->z = z; res->collision_checks = count; res->path_len = len; res->nn_queries = 0;
    free(F); free(rowptr); free(colidx); free(cur); free(Wm); free(Hm); free(Hnew); free(rev); free(heap.pri); free(heap.idx);
    return 0;
}

/* Batch form of is_free_motion for the DI space over a CSC graph: entry e (row y -> column x). */
void orc_di_graph_edges_free(const double *X, int64_t N, int32_t m, double rho, double r,
                             const int64_t *colptr, const int64_t *rowval,
                             const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    int64_t nnz = colptr[N];
    int32_t n = 2 * m;
    memset(mask, 0, sizeof(uint64_t) * (size_t)((nnz + 63) / 64));
    for (int64_t x = 0; x < N; ++x)
        for (int64_t e = colptr[x]; e < colptr[x + 1]; ++e)
            set_bit(mask, e, orc_di_is_free_motion(X + (size_t)rowval[e] * n, X + (size_t)x * n, m, rho, r, lohi, M, ss_lo, ss_hi));
}

/* ---- batch free-space sampler: sample_free! of src/sampling.jl:11-45 ------------------------------
 * The reference draws from Julia's global MersenneTwister (unseeded), so its stream cannot be reproduced; the build
 * declares a counter-based stream instead so that the sequential semantics (candidates tested IN ORDER, the first
 * accepted ones kept, sampling.jl:21-36) do not depend on how candidates are batched:
 *   Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123), key = seed,
 *   counter = (candidate lo32, candidate hi32, coordinate pair j, stream) ; stream 0 = sample_space, 1 = sample_goal.
 *   A call yields words x0..x3; coordinate 2j uses (x0, x1), 2j+1 uses (x2, x3):
 *   u = ((xa >> 5) * 2^26 + (xb >> 6)) * 2^-53  in [0, 1)   (53 random bits, exact in fp64).
 * sample_space(SS) = lo + rand .* (hi - lo)              (statespaces.jl:40), evaluated unfused. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

/* uniform numbers u[0..d) of candidate c in stream s */
void orc_sample_uniforms(uint64_t seed, uint64_t c, uint32_t stream, int32_t d, double *u)
{
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int32_t j = 0; 2 * j < d; ++j) {
        const uint32_t ctr[4] = {(uint32_t)c, (uint32_t)(c >> 32), (uint32_t)j, stream};
        uint32_t x[4];
        orc_philox4x32_10(ctr, key, x);
        u[2 * j] = u53(x[0], x[1]);
        if (2 * j + 1 < d) u[2 * j + 1] = u53(x[2], x[3]);
    }
}

/* sample_goal: Rectangle lo + (hi-lo).*rand (goals.jl:97); Ball center + 2*radius*(rand - .5), rejected outside the
 * ball (goals.jl:101-108); Point = the point (goals.jl:115).  Returns 1 if a candidate was produced (Ball may reject). */
static int goal_candidate(uint64_t seed, uint64_t g, int32_t d, int32_t kind, const double *gp, double *v)
{
    double u[ORC_MAXD];
    orc_sample_uniforms(seed, g, 1u, d, u);
    if (kind == 0) {                 /* rectangle: gp = [lo(d), hi(d)] */
        for (int32_t i = 0; i < d; ++i) { const double w = gp[d + i] - gp[i]; const double p = w * u[i]; v[i] = gp[i] + p; }
        return 1;
    }
    if (kind == 1) {                 /* ball: gp = [center(d), radius] */
        double s = 0.0;
        for (int32_t i = 0; i < d; ++i) {
            const double a = 2 * gp[d]; const double b = u[i] - .5; const double p = a * b;
            v[i] = gp[i] + p;
            const double t = v[i] - gp[i]; const double tt = t * t;
            s = (i == 0) ? tt : s + tt;
        }
        return sqrt(s) <= gp[d];
    }
    for (int32_t i = 0; i < d; ++i) v[i] = gp[i];
    return 1;
}

/* sample_free!(P, N, true; ensure_goal_ct = goal_ct) into W[N][d]: W[0] = init when init != NULL (sampling.jl:15-20),
 * then free candidates in order (21-36), then W[N-i] = i-th free goal sample, i = 1..min(goal_ct, N-1) (37-41).
 * attempts = number of sample_space candidates consumed.  Returns 0, or -1 when the goal cannot be sampled
 * (max_goal_tries exhausted: the reference would loop forever). */
int32_t orc_sample_free_biased(uint64_t seed, int64_t N, int32_t d, const double *init, const double *lohi, int32_t M,
                               const double *ss_lo, const double *ss_hi, int32_t goal_kind, const double *goal_params,
                               int32_t goal_ct, double goal_bias, double *W, int64_t *attempts);

int32_t orc_sample_free(uint64_t seed, int64_t N, int32_t d, const double *init, const double *lohi, int32_t M,
                        const double *ss_lo, const double *ss_hi, int32_t goal_kind, const double *goal_params,
                        int32_t goal_ct, double *W, int64_t *attempts)
{
    return orc_sample_free_biased(seed, N, d, init, lohi, M, ss_lo, ss_hi, goal_kind, goal_params, goal_ct, 0.0, W, attempts);
}

/* the loop as written (sampling.jl:21-36), goal_bias included (:28-30): `rand() < goal_bias` for the k-th accepted sample is
 * the uniform (seed, k) of stream 2; a replacement is sample_free_goal(P) (:3-9), consuming the goal stream in order */
int32_t orc_sample_free_biased(uint64_t seed, int64_t N, int32_t d, const double *init, const double *lohi, int32_t M,
                               const double *ss_lo, const double *ss_hi, int32_t goal_kind, const double *goal_params,
                               int32_t goal_ct, double goal_bias, double *W, int64_t *attempts)
{
    int64_t have = 0;
    uint64_t c = 0, g = 0, acc = 0;
    if (N <= 0) { if (attempts) *attempts = 0; return 0; }
    if (init) { memcpy(W, init, sizeof(double) * (size_t)d); have = 1; }
    double u[ORC_MAXD], v[ORC_MAXD];
    while (have < N) {
        orc_sample_uniforms(seed, c++, 0u, d, u);
        for (int32_t i = 0; i < d; ++i) { const double w = ss_hi[i] - ss_lo[i]; const double p = u[i] * w; v[i] = ss_lo[i] + p; }
        if (orc_is_free_state(v, d, lohi, M, ss_lo, ss_hi)) {
            double ub[ORC_MAXD];
            if (0.0 < goal_bias) orc_sample_uniforms(seed, acc, 2u, 1, ub);
            if (0.0 < goal_bias && ub[0] < goal_bias) {                         /* sampling.jl:28-29 */
                double gv[ORC_MAXD];
                for (;;) {
                    if (g > 1000000u) return -1;
                    const int ok = goal_candidate(seed, g++, d, goal_kind, goal_params, gv);
                    if (ok && orc_is_free_state(gv, d, lohi, M, ss_lo, ss_hi)) break;
                }
                memcpy(W + (size_t)have * d, gv, sizeof(double) * (size_t)d);
            } else {
                memcpy(W + (size_t)have * d, v, sizeof(double) * (size_t)d);
            }
            ++have; ++acc;
        }
    }
    if (attempts) *attempts = (int64_t)c;
    const int64_t ng = (goal_ct < N - 1) ? goal_ct : N - 1;
    for (int64_t i = 1; i <= ng; ++i) {
        for (;;) {
            if (g > 1000000u) return -1;
            const int ok = goal_candidate(seed, g++, d, goal_kind, goal_params, v);
            if (ok && orc_is_free_state(v, d, lohi, M, ss_lo, ss_hi)) break;
        }
        memcpy(W + (size_t)(N - i) * d, v, sizeof(double) * (size_t)d);
    }
    return 0;
}

/* ---- 2-D SAT world: PointRobot2D over a Compound2D of Circle / convex Polygon parts ------------------
 * src/collisioncheckers/SAT2D.jl, robots2D.jl:12-14, utilities/vec2Dutils.jl.  Canon as elsewhere: fp64, unfused,
 * dot(a,b) = a1*b1 + a2*b2, cross(a,b) = a1*b2 - a2*b1 (vec2Dutils.jl:5-7), normalize(v) = v / norm(v).
 * NOTE (bug compatibility): colliding(p, P::Polygon) is written `@all [!ininterval(dot(p, n_i), nextrema_i)]`
 * (SAT2D.jl:124-127), i.e. a point collides with a polygon only if it projects OUTSIDE the polygon's extent on every
 * edge normal -- the restatement follows the reference as written. */
typedef struct {
    int32_t kind;                 /* 0 circle, 1 polygon */
    int32_t n;                    /* polygon vertex count */
    double c[2], r;               /* circle */
    double xr[2], yr[2];          /* AABB (xrange, yrange) */
    double pts[ORC_MAXPOLY][2], edges[ORC_MAXPOLY][2], normals[ORC_MAXPOLY][2], nex[ORC_MAXPOLY][2];
} orc_shape2d;

static inline double dot2(const double *a, const double *b) { const double p = a[0] * b[0]; const double q = a[1] * b[1]; return p + q; }
static inline double cross2(const double *a, const double *b) { const double p = a[0] * b[1]; const double q = a[1] * b[0]; return p - q; }
static inline int overlapping(const double *i1, const double *i2) { return i1[0] <= i2[1] && i2[0] <= i1[1]; }   /* vec2Dutils.jl:33 */
static inline int ininterval(double x, const double *i) { return i[0] <= x && x <= i[1]; }                         /* :34 */
static void project_extrema(const double (*pts)[2], int n, const double *ax, double *out)                           /* :19-28 */
{
    double dmin = INFINITY, dmax = -INFINITY;
    for (int i = 0; i < n; ++i) { const double d = dot2(pts[i], ax); if (d < dmin) dmin = d; if (d > dmax) dmax = d; }
    out[0] = dmin; out[1] = dmax;
}

/* Circle(c, r) (SAT2D.jl:26-28) / Polygon(points) (SAT2D.jl:40-55).  data: circle [cx, cy, r]; polygon [x1,y1,...].
 * Returns 0, -1 bad arguments (r <= 0, n < 3, n > ORC_MAXPOLY), -2 polygon not convex. */
int32_t orc_shape2d_build(int32_t kind, int32_t n, const double *data, orc_shape2d *S)
{
    memset(S, 0, sizeof *S);
    S->kind = kind;
    if (kind == 0) {
        if (!(data[2] > 0)) return -1;
        S->c[0] = data[0]; S->c[1] = data[1]; S->r = data[2];
        S->xr[0] = data[0] - data[2]; S->xr[1] = data[0] + data[2];
        S->yr[0] = data[1] - data[2]; S->yr[1] = data[1] + data[2];
        return 0;
    }
    if (n < 3 || n > ORC_MAXPOLY) return -1;
    S->n = n;
    for (int i = 0; i < n; ++i) { S->pts[i][0] = data[2 * i]; S->pts[i][1] = data[2 * i + 1]; }
    double area = 0.0;                                   /* sum([...]) accumulates left to right from zero */
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 < n) ? i + 1 : 0;
        const double a = S->pts[j][0] - S->pts[i][0], b = S->pts[j][1] + S->pts[i][1];
        const double t = a * b;
        area = (i == 0) ? t : area + t;
    }
    if (area > 0)                                        /* reverse!(points) */
        for (int i = 0; i < n / 2; ++i)
            for (int k = 0; k < 2; ++k) { const double t = S->pts[i][k]; S->pts[i][k] = S->pts[n - 1 - i][k]; S->pts[n - 1 - i][k] = t; }
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 < n) ? i + 1 : 0;
        S->edges[i][0] = S->pts[j][0] - S->pts[i][0]; S->edges[i][1] = S->pts[j][1] - S->pts[i][1];
        const double px = S->edges[i][1], py = -S->edges[i][0];            /* perp(g) = (g2, -g1) */
        const double p = px * px, q = py * py;
        const double nrm = sqrt(p + q);
        S->normals[i][0] = px / nrm; S->normals[i][1] = py / nrm;
    }
    for (int i = 0; i < n; ++i) {                        /* convexity: consecutive normal angles must not step in [-pi, 0] */
        const int j = (i + 1 < n) ? i + 1 : 0;
        const double di = atan2(S->normals[j][1], S->normals[j][0]) - atan2(S->normals[i][1], S->normals[i][0]);
        if (-M_PI <= di && di <= 0) return -2;
    }
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int i = 0; i < n; ++i) {
        if (S->pts[i][0] < xmin) xmin = S->pts[i][0];
        if (S->pts[i][0] > xmax) xmax = S->pts[i][0];
        if (S->pts[i][1] < ymin) ymin = S->pts[i][1];
        if (S->pts[i][1] > ymax) ymax = S->pts[i][1];
    }
    S->xr[0] = xmin; S->xr[1] = xmax; S->yr[0] = ymin; S->yr[1] = ymax;
    for (int i = 0; i < n; ++i) project_extrema(S->pts, n, S->normals[i], S->nex[i]);
    return 0;
}
int64_t orc_shape2d_sizeof(void) { return (int64_t)sizeof(orc_shape2d); }

/* colliding(p, S) (SAT2D.jl:121-133) */
static int point_colliding_shape(const double *p, const orc_shape2d *S)
{
    if (S->kind == 0) {
        const double t[2] = {p[0] - S->c[0], p[1] - S->c[1]};
        return dot2(t, t) <= S->r * S->r;
    }
    if (!(ininterval(p[0], S->xr) && ininterval(p[1], S->yr))) return 0;
    for (int i = 0; i < S->n; ++i)
        if (ininterval(dot2(p, S->normals[i]), S->nex[i])) return 0;       /* @all [!ininterval(...)] as written */
    return 1;
}

/* colliding_ends_free(L, S) (SAT2D.jl:163-174) with L = Line(v, w) (SAT2D.jl:66-81) */
static int line_colliding_ends_free(const double *v, const double *w, const orc_shape2d *S)
{
    const double edge[2] = {w[0] - v[0], w[1] - v[1]};
    const double lx[2] = {v[0] < w[0] ? v[0] : w[0], v[0] < w[0] ? w[0] : v[0]};     /* minmaxV */
    const double ly[2] = {v[1] < w[1] ? v[1] : w[1], v[1] < w[1] ? w[1] : v[1]};
    if (!(overlapping(lx, S->xr) && overlapping(ly, S->yr))) return 0;                 /* AABBseparated */
    if (S->kind == 0) {
        const double vc[2] = {S->c[0] - v[0], S->c[1] - v[1]};
        const double d2 = dot2(edge, edge);
        const double cr = cross2(edge, vc);
        const double lhs = d2 * (S->r * S->r), rhs = cr * cr;                          /* d2*C.r^2 < cross(...)^2 */
        if (lhs < rhs) return 0;
        const double t = dot2(vc, edge);
        return 0 <= t && t <= d2;
    }
    const double normal[2] = {edge[1], -edge[0]};                                      /* perp(edge), not normalised */
    const double ndotv = dot2(v, normal);
    double ex[2];
    project_extrema(S->pts, S->n, normal, ex);
    if (!ininterval(ndotv, ex)) return 0;                                              /* is_separating_axis(L, P) */
    for (int i = 0; i < S->n; ++i) {                                                   /* is_separating_axis(P, L, i) */
        const double a = dot2(v, S->normals[i]), b = dot2(w, S->normals[i]);
        const double li[2] = {a < b ? a : b, a < b ? b : a};
        if (!overlapping(S->nex[i], li)) return 0;
    }
    return 1;
}

/* colliding(L, B) = colliding_ends_free(L,B) || colliding(L.v,B) || colliding(L.w,B) (SAT2D.jl:176) */
static int line_colliding_shape(const double *v, const double *w, const orc_shape2d *S)
{
    return line_colliding_ends_free(v, w, S) || point_colliding_shape(v, S) || point_colliding_shape(w, S);
}

static void compound_aabb(const orc_shape2d *S, int32_t n, double *xr, double *yr)     /* Compound2D ctor, SAT2D.jl:88-97 */
{
    if (n == 0) { xr[0] = xr[1] = yr[0] = yr[1] = 0.0; return; }
    xr[0] = yr[0] = INFINITY; xr[1] = yr[1] = -INFINITY;
    for (int32_t i = 0; i < n; ++i) {
        if (S[i].xr[0] < xr[0]) xr[0] = S[i].xr[0];
        if (S[i].xr[1] > xr[1]) xr[1] = S[i].xr[1];
        if (S[i].yr[0] < yr[0]) yr[0] = S[i].yr[0];
        if (S[i].yr[1] > yr[1]) yr[1] = S[i].yr[1];
    }
}

/* is_free_state(v, CC::PointRobot2D) = !colliding(v, CC.obstacles) (robots2D.jl:12; compound SAT2D.jl:129-132) */
int32_t orc_2d_point_free(const double *p, const orc_shape2d *S, int32_t n)
{
    double xr[2], yr[2];
    compound_aabb(S, n, xr, yr);
    if (!(ininterval(p[0], xr) && ininterval(p[1], yr))) return 1;
    for (int32_t i = 0; i < n; ++i) if (point_colliding_shape(p, &S[i])) return 0;
    return 1;
}

/* is_free_motion(v, w, CC::PointRobot2D) = !colliding(Line(v,w), CC.obstacles) (robots2D.jl:13-14; SAT2D.jl:154-157,178) */
int32_t orc_2d_motion_free(const double *v, const double *w, const orc_shape2d *S, int32_t n)
{
    double xr[2], yr[2];
    compound_aabb(S, n, xr, yr);
    const double lx[2] = {v[0] < w[0] ? v[0] : w[0], v[0] < w[0] ? w[0] : v[0]};
    const double ly[2] = {v[1] < w[1] ? v[1] : w[1], v[1] < w[1] ? w[1] : v[1]};
    if (!(overlapping(xr, lx) && overlapping(yr, ly))) return 1;                       /* AABBseparated(C, L) */
    for (int32_t i = 0; i < n; ++i) if (line_colliding_shape(v, w, &S[i])) return 0;
    return 1;
}

/* batch forms with the state-space wrappers of statespaces.jl:150-158 (Identity state2workspace, d = 2) */
void orc_2d_points_free(const double *P, int64_t n, const orc_shape2d *S, int32_t ns, const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    memset(mask, 0, sizeof(uint64_t) * (size_t)((n + 63) / 64));
    for (int64_t e = 0; e < n; ++e)
        set_bit(mask, e, orc_in_state_space(P + 2 * e, ss_lo, ss_hi, 2) && orc_2d_point_free(P + 2 * e, S, ns));
}
void orc_2d_motions_free(const double *P, const double *Q, int64_t n, const orc_shape2d *S, int32_t ns, const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    memset(mask, 0, sizeof(uint64_t) * (size_t)((n + 63) / 64));
    for (int64_t e = 0; e < n; ++e)
        set_bit(mask, e, orc_in_state_space(P + 2 * e, ss_lo, ss_hi, 2) && orc_2d_motion_free(P + 2 * e, Q + 2 * e, S, ns));
}
void orc_2d_graph_edges_free(const double *X, int64_t N, const int64_t *colptr, const int64_t *rowval, const orc_shape2d *S, int32_t ns,
                             const double *ss_lo, const double *ss_hi, uint64_t *mask)
{
    memset(mask, 0, sizeof(uint64_t) * (size_t)((colptr[N] + 63) / 64));
    for (int64_t x = 0; x < N; ++x)
        for (int64_t e = colptr[x]; e < colptr[x + 1]; ++e) {
            const double *v = X + 2 * rowval[e], *w = X + 2 * x;
            set_bit(mask, e, orc_in_state_space(v, ss_lo, ss_hi, 2) && orc_2d_motion_free(v, w, S, ns));
        }
}

/* ---- Dubins car: DubinsQuasiMetricSpace, src/statespaces/simplecars.jl:15-21,32-38,51-83,91-215 ---------
 * States are SE2 (x, y, theta); the workspace is (x, y) (VectorView(1:2), :37).  The quasi-metric is the exact Dubins
 * length for turning radius rt and speed s; the near-neighbour sets are "chopped": candidates within Euclidean
 * (x, y) distance r (KD-tree on positions, :45-49, nearneighbors.jl:185-198), kept when the Dubins cost is <= r.
 * Arithmetic as written in the reference (unfused, operation order kept); sin / cos / atan2 / acos are mp_math.h's -- fixed
 * reductions and polynomials built from + - * / sqrt, the same header the device kernels compile, so both sides produce the
 * same bits (libm's differ between glibc and the GPU's by an ulp or two, which moved threshold decisions); sqrt / fmod are libm's (exact).
 * mod2piF(x) = mod(x, 2*pi) (utils.jl:91) with Julia's mod for floats: r = rem(x, y); r == 0 ? +0 : (r < 0 ? r + y : r). */
#define ORC_TWOPI (2 * 3.141592653589793)
static double mod2pif(double x)
{
    double r = fmod(x, ORC_TWOPI);
    if (r == 0) return 0.0;
    return r < 0 ? r + ORC_TWOPI : r;
}
typedef struct { double t, s, k; } orc_step;      /* StepControl(t, (u1 = speed sign*s, u2 = signed curvature)) */
static orc_step car_seg(int turn, double d) { orc_step u = {fabs(d), (d > 0) - (d < 0), (double)turn}; return u; }   /* :91 */

#define DUB_TRY(cnew_expr, T0, D0, T1, D1, T2, D2)                 \
    do { const double cnew = (cnew_expr); if (!(c <= cnew)) { path[0] = car_seg(T0, D0); path[1] = car_seg(T1, D1); path[2] = car_seg(T2, D2); c = cnew; } } while (0)

/* dubins(s1, s2, r, s) (simplecars.jl:198-215): cost and the 3 scaled step controls; cost = Inf when no word applies */
double orc_dubins(const double *s1, const double *s2, double r, double s, orc_step *path)
{
    const double vx = (s2[0] - s1[0]) / r, vy = (s2[1] - s1[1]) / r;
    const double d = sqrt(vx * vx + vy * vy);
    const double th = mp_atan2(vy, vx);
    const double a = mod2pif(s1[2] - th), b = mod2pif(s2[2] - th);
    const double ca = mp_cos(a), sa = mp_sin(a), cb = mp_cos(b), sb = mp_sin(b);
    double c = INFINITY;
    for (int q = 0; q < 3; ++q) { path[q].t = 0; path[q].s = 0; path[q].k = 0; }
    {   /* LSL :106-119 */
        const double tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sa - sb));
        if (!(tmp < 0)) {
            const double t0 = mp_atan2(cb - ca, d + sa - sb);
            const double t = mod2pif(-a + t0), p = sqrt(fmax(tmp, 0.0)), q = mod2pif(b - t0);
            DUB_TRY(t + p + q, 1, t, 0, p, 1, q);
        }
    }
    {   /* RSR :121-134 */
        const double tmp = 2 + d * d - 2 * (ca * cb + sa * sb - d * (sb - sa));
        if (!(tmp < 0)) {
            const double t0 = mp_atan2(ca - cb, d - sa + sb);
            const double t = mod2pif(a - t0), p = sqrt(fmax(tmp, 0.0)), q = mod2pif(-b + t0);
            DUB_TRY(t + p + q, -1, t, 0, p, -1, q);
        }
    }
    {   /* RSL :136-149 */
        const double tmp = d * d - 2 + 2 * (ca * cb + sa * sb - d * (sa + sb));
        if (!(tmp < 0)) {
            const double p = sqrt(fmax(tmp, 0.0));
            const double t0 = mp_atan2(ca + cb, d - sa - sb) - mp_atan2(2.0, p);
            const double t = mod2pif(a - t0), q = mod2pif(b - t0);
            DUB_TRY(t + p + q, -1, t, 0, p, 1, q);
        }
    }
    {   /* LSR :151-164 */
        const double tmp = -2 + d * d + 2 * (ca * cb + sa * sb + d * (sa + sb));
        if (!(tmp < 0)) {
            const double p = sqrt(fmax(tmp, 0.0));
            const double t0 = mp_atan2(-ca - cb, d + sa + sb) - mp_atan2(-2.0, p);
            const double t = mod2pif(-a + t0), q = mod2pif(-b + t0);
            DUB_TRY(t + p + q, 1, t, 0, p, -1, q);
        }
    }
    {   /* RLR :166-179 */
        const double tmp = (6 - d * d + 2 * (ca * cb + sa * sb + d * (sa - sb))) / 8;
        if (!(fabs(tmp) >= 1)) {
            const double p = ORC_TWOPI - mp_acos(tmp);
            const double t0 = mp_atan2(ca - cb, d - sa + sb);
            const double t = mod2pif(a - t0 + p / 2), q = mod2pif(a - b - t + p);
            DUB_TRY(t + p + q, -1, t, 1, p, -1, q);
        }
    }
    {   /* LRL :181-194 */
        const double tmp = (6 - d * d + 2 * (ca * cb + sa * sb - d * (sa - sb))) / 8;
        if (!(fabs(tmp) >= 1)) {
            const double p = ORC_TWOPI - mp_acos(tmp);
            const double t0 = mp_atan2(-ca + cb, d + sa - sb);
            const double t = mod2pif(-a + t0 + p / 2), q = mod2pif(b - a - t + p);
            DUB_TRY(t + p + q, 1, t, -1, p, 1, q);
        }
    }
    for (int q = 0; q < 3; ++q) {                 /* scalespeed!(scaleradius!(pmin, r), s) :92-105,214 */
        path[q].t = path[q].t * r; path[q].k = path[q].k / r;
        path[q].t = path[q].t / s; path[q].s = path[q].s * s;
    }
    return c * r;
}

/* propagate(M, v, u) :52-65 */
static void car_propagate(const double *v, orc_step u, double *out)
{
    const double ang = u.t * u.s * u.k;
    if (fabs(ang) > 10 * 2.220446049250313e-16) {
        out[0] = v[0] + (mp_sin(v[2] + ang) - mp_sin(v[2])) / u.k;
        out[1] = v[1] + (mp_cos(v[2]) - mp_cos(v[2] + ang)) / u.k;
        out[2] = mod2pif(v[2] + ang);
    } else {
        out[0] = v[0] + u.t * u.s * mp_cos(v[2]);
        out[1] = v[1] + u.t * u.s * mp_sin(v[2]);
        out[2] = mod2pif(v[2] + ang);
    }
}

/* ---- Reeds-Shepp car: simplecars.jl:228-524 (ReedsSheppMetricSpace :26-31).  Nine word families, each tried on the target
 * and its time-flipped / reflected / backwards images in the reference's order; a later candidate replaces the incumbent only
 * when strictly shorter.  Negative segment lengths mean reverse gear (carsegment2stepcontrol: speed = sign(d)). */
static void rs_R(double x, double y, double *r, double *th) { *r = sqrt(x * x + y * y); *th = mp_atan2(y, x); }      /* :230 */
static double rs_M(double t) { const double m = mod2pif(t); return m > 3.141592653589793 ? m - ORC_TWOPI : m; }      /* :232-235 */
static double rs_Tau(double u, double v, double E, double N)                                                          /* :236-243 */
{
    const double delta = rs_M(u - v);
    const double A = mp_sin(u) - mp_sin(delta);
    const double B = mp_cos(u) - mp_cos(delta) - 1;
    double r, th;
    rs_R(E * A + N * B, N * A - E * B, &r, &th);
    const double t = 2 * mp_cos(delta) - 2 * mp_cos(v) - 2 * mp_cos(u) + 3;
    return t < 0 ? rs_M(th + 3.141592653589793) : rs_M(th);
}
static double rs_Omega(double u, double v, double E, double N, double t) { return rs_M(rs_Tau(u, v, E, N) - u + v - t); }   /* :244 */

typedef struct { double c; int l; int post; orc_step p[5]; } rs_best;
#define RS_ACCEPT(B, CNEW, L, POSTV, ...)                                                                   \
    do { const double cnew_ = (CNEW); if (!((B)->c <= cnew_)) { const orc_step tmp_[5] = {__VA_ARGS__}; for (int q_ = 0; q_ < (L); ++q_) (B)->p[q_] = tmp_[q_]; (B)->c = cnew_; (B)->l = (L); (B)->post = (POSTV); } } while (0)

static void rs_LpSpLp(const double *T, rs_best *b, int post)                   /* (8.1) :365-376 */
{
    double r, th;
    rs_R(T[0] - mp_sin(T[2]), T[1] - 1 + mp_cos(T[2]), &r, &th);
    const double u = r, t = mod2pif(th), v = mod2pif(T[2] - t);
    RS_ACCEPT(b, t + u + v, 3, post, car_seg(1, t), car_seg(0, u), car_seg(1, v));
}
static void rs_LpSpRp(const double *T, rs_best *b, int post)                   /* (8.2) :378-391 */
{
    double r, th, r1, th1;
    rs_R(T[0] + mp_sin(T[2]), T[1] - 1 - mp_cos(T[2]), &r, &th);
    if (r * r < 4) return;
    const double u = sqrt(r * r - 4);
    rs_R(u, 2.0, &r1, &th1);
    const double t = mod2pif(th + th1), v = mod2pif(t - T[2]);
    RS_ACCEPT(b, t + u + v, 3, post, car_seg(1, t), car_seg(0, u), car_seg(-1, v));
}
static void rs_LpRmLp(const double *T, rs_best *b, int post)                   /* (8.3) :393-408 */
{
    const double E = T[0] - mp_sin(T[2]), N = T[1] + mp_cos(T[2]) - 1;
    if (E * E + N * N > 16) return;
    double r, th;
    rs_R(E, N, &r, &th);
    double u = mp_acos(1 - r * r / 8);
    const double t = mod2pif(th - u / 2 + 3.141592653589793), v = mod2pif(3.141592653589793 - u / 2 - th + T[2]);
    u = -u;
    RS_ACCEPT(b, t - u + v, 3, post, car_seg(1, t), car_seg(-1, u), car_seg(1, v));
}
static void rs_LpRmLm(const double *T, rs_best *b, int post)                   /* (8.4) :410-425 */
{
    const double E = T[0] - mp_sin(T[2]), N = T[1] + mp_cos(T[2]) - 1;
    if (E * E + N * N > 16) return;
    double r, th;
    rs_R(E, N, &r, &th);
    double u = mp_acos(1 - r * r / 8);
    const double t = mod2pif(th - u / 2 + 3.141592653589793), v = mod2pif(3.141592653589793 - u / 2 - th + T[2]) - ORC_TWOPI;
    u = -u;
    RS_ACCEPT(b, t - u - v, 3, post, car_seg(1, t), car_seg(-1, u), car_seg(1, v));
}
static void rs_LpRpuLmuRm(const double *T, rs_best *b, int post)               /* (8.7) :427-442 */
{
    const double E = T[0] + mp_sin(T[2]), N = T[1] - mp_cos(T[2]) - 1;
    const double p = (2 + sqrt(E * E + N * N)) / 4;
    if (p < 0 || p > 1) return;
    const double u = mp_acos(p);
    const double t = mod2pif(rs_Tau(u, -u, E, N)), v = mod2pif(rs_Omega(u, -u, E, N, T[2])) - ORC_TWOPI;
    RS_ACCEPT(b, t + 2 * u - v, 4, post, car_seg(1, t), car_seg(-1, u), car_seg(1, -u), car_seg(-1, v));
}
static void rs_LpRmuLmuRp(const double *T, rs_best *b, int post)               /* (8.8) :444-459 */
{
    const double E = T[0] + mp_sin(T[2]), N = T[1] - mp_cos(T[2]) - 1;
    const double p = (20 - E * E - N * N) / 16;
    if (p < 0 || p > 1) return;
    const double u = -mp_acos(p);
    const double t = mod2pif(rs_Tau(u, u, E, N)), v = mod2pif(rs_Omega(u, u, E, N, T[2]));
    RS_ACCEPT(b, t - 2 * u + v, 4, post, car_seg(1, t), car_seg(-1, u), car_seg(1, u), car_seg(-1, v));
}
static void rs_LpRmSmLm(const double *T, rs_best *b, int post)                 /* (8.9) :461-479 */
{
    const double E = T[0] - mp_sin(T[2]), N = T[1] + mp_cos(T[2]) - 1;
    double D, be;
    rs_R(E, N, &D, &be);
    if (D < 2) return;
    const double ga = mp_acos(2 / D), F = sqrt(D * D / 4 - 1);
    const double t = mod2pif(3.141592653589793 + be - ga), u = 2 - 2 * F;
    if (u > 0) return;
    const double v = mod2pif(-3 * 3.141592653589793 / 2 + ga + T[2] - be) - ORC_TWOPI;
    RS_ACCEPT(b, t + 3.141592653589793 / 2 - u - v, 4, post, car_seg(1, t), car_seg(-1, -3.141592653589793 / 2), car_seg(0, u), car_seg(1, v));
}
static void rs_LpRmSmRm(const double *T, rs_best *b, int post)                 /* (8.10) :481-497 */
{
    const double E = T[0] + mp_sin(T[2]), N = T[1] - mp_cos(T[2]) - 1;
    double D, be;
    rs_R(E, N, &D, &be);
    if (D < 2) return;
    const double t = mod2pif(be + 3.141592653589793 / 2), u = 2 - D;
    if (u > 0) return;
    const double v = mod2pif(-3.141592653589793 - T[2] + be) - ORC_TWOPI;
    RS_ACCEPT(b, t + 3.141592653589793 / 2 - u - v, 4, post, car_seg(1, t), car_seg(-1, -3.141592653589793 / 2), car_seg(0, u), car_seg(-1, v));
}
static void rs_LpRmSmLmRp(const double *T, rs_best *b, int post)               /* (8.11) :499-518 */
{
    const double E = T[0] + mp_sin(T[2]), N = T[1] - mp_cos(T[2]) - 1;
    double D, be;
    rs_R(E, N, &D, &be);
    if (D < 2) return;
    const double ga = mp_acos(2 / D), F = sqrt(D * D / 4 - 1);
    const double t = mod2pif(3.141592653589793 + be - ga), u = 4 - 2 * F;
    if (u > 0) return;
    const double v = mod2pif(3.141592653589793 + be - T[2] - ga);
    RS_ACCEPT(b, t + 3.141592653589793 - u + v, 5, post, car_seg(1, t), car_seg(-1, -3.141592653589793 / 2), car_seg(0, u),
              car_seg(1, -3.141592653589793 / 2), car_seg(-1, v));
}

/* reedsshepp(s1, s2, r, s) :265-363: cost, controls path[0..*L) */
double orc_reedsshepp(const double *s1, const double *s2, double r, double s, orc_step *path, int32_t *L)
{
    const double dx = (s2[0] - s1[0]) / r, dy = (s2[1] - s1[1]) / r;
    const double ct = mp_cos(s1[2]), st = mp_sin(s1[2]);
    double T[8][3];            /* POST, T, R, B, R_T, B_T, B_R, B_R_T in the reference's numbering 0..7 */
    T[0][0] = dx * ct + dy * st; T[0][1] = -dx * st + dy * ct; T[0][2] = mod2pif(s2[2] - s1[2]);
#define RS_TIMEFLIP(D, S) do { (D)[0] = -(S)[0]; (D)[1] = (S)[1]; (D)[2] = -(S)[2]; } while (0)
#define RS_REFLECT(D, S) do { (D)[0] = (S)[0]; (D)[1] = -(S)[1]; (D)[2] = -(S)[2]; } while (0)
    RS_TIMEFLIP(T[1], T[0]);                       /* tTarget   */
    RS_REFLECT(T[2], T[0]);                        /* rTarget   */
    RS_REFLECT(T[4], T[1]);                        /* trTarget  */
    T[3][0] = T[0][0] * mp_cos(T[0][2]) + T[0][1] * mp_sin(T[0][2]);      /* bTarget = backwards(target) :247 */
    T[3][1] = T[0][0] * mp_sin(T[0][2]) - T[0][1] * mp_cos(T[0][2]);
    T[3][2] = T[0][2];
    RS_TIMEFLIP(T[5], T[3]);                       /* btTarget  */
    RS_REFLECT(T[6], T[3]);                        /* brTarget  */
    RS_REFLECT(T[7], T[5]);                        /* btrTarget */
    rs_best b; b.c = INFINITY; b.l = 0; b.post = 0;
    static const int four[4] = {0, 1, 2, 4}, eight[8] = {0, 1, 2, 4, 3, 5, 6, 7};
    for (int q = 0; q < 4; ++q) rs_LpSpLp(T[four[q]], &b, four[q]);
    for (int q = 0; q < 4; ++q) rs_LpSpRp(T[four[q]], &b, four[q]);
    rs_LpRmLp(T[0], &b, 0); rs_LpRmLp(T[2], &b, 2);
    for (int q = 0; q < 8; ++q) rs_LpRmLm(T[eight[q]], &b, eight[q]);
    for (int q = 0; q < 4; ++q) rs_LpRpuLmuRm(T[four[q]], &b, four[q]);
    for (int q = 0; q < 4; ++q) rs_LpRmuLmuRp(T[four[q]], &b, four[q]);
    for (int q = 0; q < 8; ++q) rs_LpRmSmLm(T[eight[q]], &b, eight[q]);
    for (int q = 0; q < 8; ++q) rs_LpRmSmRm(T[eight[q]], &b, eight[q]);
    for (int q = 0; q < 4; ++q) rs_LpRmSmLmRp(T[four[q]], &b, four[q]);
    for (int q = 0; q < b.l; ++q) {               /* scalespeed!(scaleradius!(p[1:l], r), s) */
        b.p[q].t = b.p[q].t * r; b.p[q].k = b.p[q].k / r;
        b.p[q].t = b.p[q].t / s; b.p[q].s = b.p[q].s * s;
    }
    const int tf = (b.post == 1 || b.post == 4 || b.post == 5 || b.post == 7);      /* timeflip!: negate speed */
    const int rf = (b.post == 2 || b.post == 4 || b.post == 6 || b.post == 7);      /* reflect!: negate curvature */
    const int bw = (b.post == 3 || b.post == 5 || b.post == 6 || b.post == 7);      /* backwards!: reverse the order */
    for (int q = 0; q < b.l; ++q) { if (tf) b.p[q].s = -b.p[q].s; if (rf) b.p[q].k = -b.p[q].k; }
    for (int q = 0; q < b.l; ++q) path[q] = bw ? b.p[b.l - 1 - q] : b.p[q];
    for (int q = b.l; q < 5; ++q) { path[q].t = 0; path[q].s = 0; path[q].k = 0; }
    *L = b.l;
    return b.c * r;
}

/* kind 1 = Dubins, 2 = Reeds-Shepp */
static double car_steer(int32_t kind, const double *s1, const double *s2, double rt, double sp, orc_step *path, int32_t *L)
{
    if (kind == 2) return orc_reedsshepp(s1, s2, rt, sp, path, L);
    *L = 3;
    path[3].t = path[3].s = path[3].k = 0; path[4] = path[3];
    return orc_dubins(s1, s2, rt, sp, path);
}

#define ORC_DUB_MAXWP 160
/* collision_waypoints(d, v, w) = per-segment waypoints (:68-83; arcs sampled every pi/12 -- only for positive
 * u.t*s*invr: a negative quotient floors to m <= -1 and the range 1:m is empty) + the target (statespaces.jl:127-135) */
int32_t orc_car_waypoints(int32_t kind, const double *v0, const double *w, double rt, double sp, double *wps)
{
    orc_step path[5];
    int32_t L = 3;
    car_steer(kind, v0, w, rt, sp, path, &L);
    double v[3] = {v0[0], v0[1], v0[2]};
    int32_t n = 0;
    const double thres = 3.141592653589793 / 12;
    for (int q = 0; q < L; ++q) {
        const orc_step u = path[q];
        const double quo = u.t * u.s * u.k / thres;
        const double fl = floor(quo);
        const long m = (long)fl;
        wps[3 * n] = v[0]; wps[3 * n + 1] = v[1]; wps[3 * n + 2] = v[2]; ++n;
        if (m != 0)
            for (long i = 1; i <= m && n < ORC_DUB_MAXWP - 2; ++i) {
                const double ai = (double)i * thres;
                wps[3 * n] = v[0] + (mp_sin(v[2] + ai) - mp_sin(v[2])) / u.k;
                wps[3 * n + 1] = v[1] + (mp_cos(v[2]) - mp_cos(v[2] + ai)) / u.k;
                wps[3 * n + 2] = mod2pif(v[2] + ai);
                ++n;
            }
        double nv[3];
        car_propagate(v, u, nv);
        v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2];
    }
    wps[3 * n] = w[0]; wps[3 * n + 1] = w[1]; wps[3 * n + 2] = w[2]; ++n;
    return n;
}
int32_t orc_dubins_waypoints(const double *v0, const double *w, double rt, double sp, double *wps) { return orc_car_waypoints(1, v0, w, rt, sp, wps); }

/* is_free_motion(v, w, CC, SS) (statespaces.jl:153-158): every consecutive waypoint pair, in_state_space on the first
 * point (SE2 bounds), segment test on (x, y) against 2-D boxes; *nseg = segment tests made (CC.count, boxesND.jl:26) */
int32_t orc_car_is_free_motion(int32_t kind, const double *v, const double *w, double rt, double sp, const double *lohi, int32_t M,
                               const double *ss_lo, const double *ss_hi, int32_t *nseg)
{
    double wps[3 * ORC_DUB_MAXWP];
    const int32_t n = orc_car_waypoints(kind, v, w, rt, sp, wps);
    int32_t cnt = 0, ok = 1;
    for (int32_t i = 0; i + 1 < n && ok; ++i) {
        if (!orc_in_state_space(wps + 3 * i, ss_lo, ss_hi, 3)) { ok = 0; break; }
        ++cnt;
        if (!orc_motion_free_boxes(wps + 3 * i, wps + 3 * (i + 1), lohi, M, 2)) ok = 0;
    }
    if (nseg) *nseg = cnt;
    return ok;
}
int32_t orc_dubins_is_free_motion(const double *v, const double *w, double rt, double sp, const double *lohi, int32_t M,
                                  const double *ss_lo, const double *ss_hi, int32_t *nseg)
{
    return orc_car_is_free_motion(1, v, w, rt, sp, lohi, M, ss_lo, ss_hi, nseg);
}

/* chopped backward sets as a CSC (column j = sources i with |xy_i - xy_j| <= r and dubins(i -> j) <= r), nearneighbors.jl:185-198.
 * Two-phase: colptr != NULL counts (colptr[N+1], returns nnz); then rowval / nzval are filled in a second call. */
int64_t orc_dubins_graph(const double *X, int64_t N, double rt, double sp, double r, int64_t *colptr, int64_t *rowval, double *nzval)
{
    int64_t nnz = 0;
    orc_step path[3];
    for (int64_t j = 0; j < N; ++j) {
        if (colptr) colptr[j] = nnz;
        for (int64_t i = 0; i < N; ++i) {
            if (i == j) continue;
            const double dx = X[3 * i] - X[3 * j], dy = X[3 * i + 1] - X[3 * j + 1];
            const double px = dx * dx, py = dy * dy;
            if (!(px + py <= r * r)) continue;                       /* inrange on positions (reduced distance, like the tree) */
            const double c = orc_dubins(X + 3 * i, X + 3 * j, rt, sp, path);
            if (c <= r) { if (rowval) { rowval[nnz] = i; nzval[nnz] = c; } ++nnz; }
        }
    }
    if (colptr) colptr[N] = nnz;
    return nnz;
}

void orc_dubins_graph_edges_free(const double *X, int64_t N, double rt, double sp, const int64_t *colptr, const int64_t *rowval,
                                 const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *mask, uint8_t *nseg)
{
    memset(mask, 0, sizeof(uint64_t) * (size_t)((colptr[N] + 63) / 64));
    for (int64_t x = 0; x < N; ++x)
        for (int64_t e = colptr[x]; e < colptr[x + 1]; ++e) {
            int32_t ns = 0;
            set_bit(mask, e, orc_dubins_is_free_motion(X + 3 * rowval[e], X + 3 * x, rt, sp, lohi, M, ss_lo, ss_hi, &ns));
            if (nseg) nseg[e] = (uint8_t)ns;
        }
}

/* fmtstar! over the Dubins space: forward sets = rows of the cost matrix, backward sets = columns (like orc_di_fmtstar).
 * Goals act on the workspace (x, y) for RECT / BALL; POINT is exact state equality. */
int32_t orc_dubins_fmtstar(const double *X, int64_t N, double rt, double sp, int64_t init_idx, int32_t checkpts,
                           const int64_t *colptr, const int64_t *rowval, const double *nzval,
                           int32_t goal_kind, const double *goal, const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                           int64_t *A, double *C, int64_t *path, orc_fmt_result *res)
{
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
#define CAR_FREE_STATE(p) (orc_in_state_space((p), ss_lo, ss_hi, 3) && orc_point_free_boxes((p), lohi, M, 2))
#define CAR_GOAL(p) ((goal_kind == 2) ? ((p)[0] == goal[0] && (p)[1] == goal[1] && (p)[2] == goal[2]) : orc_is_goal_pt((p), 2, goal_kind, goal))
    if (!CAR_FREE_STATE(X + 3 * init_idx)) return -1;
    uint8_t *F = NULL;
    if (checkpts) { F = (uint8_t *)malloc((size_t)N); for (int64_t i = 0; i < N; ++i) F[i] = (uint8_t)CAR_FREE_STATE(X + 3 * i); }
    const int64_t nnz = colptr[N];
    int64_t *rowptr = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t));
    int64_t *colidx = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    for (int64_t e = 0; e < nnz; ++e) rowptr[rowval[e] + 1]++;
    for (int64_t i = 0; i < N; ++i) rowptr[i + 1] += rowptr[i];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(N > 0 ? N : 1));
    memcpy(cur, rowptr, sizeof(int64_t) * (size_t)N);
    for (int64_t j = 0; j < N; ++j) for (int64_t e = colptr[j]; e < colptr[j + 1]; ++e) colidx[cur[rowval[e]]++] = j;
    uint8_t *Wm = (uint8_t *)malloc((size_t)N), *Hm = (uint8_t *)calloc((size_t)N, 1);
    memset(Wm, 1, (size_t)N);
    for (int64_t i = 0; i < N; ++i) { A[i] = -1; C[i] = 0.0; }
    int64_t *Hnew = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *rev = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    orc_heap heap = {0};
    Wm[init_idx] = 0; Hm[init_idx] = 1;
    heap_push(&heap, init_idx, 0.0);
    int64_t z = heap_pop(&heap), count = 0;
    while (!CAR_GOAL(X + 3 * z)) {
        int64_t nnew = 0;
        for (int64_t a = rowptr[z]; a < rowptr[z + 1]; ++a) {
            const int64_t x = colidx[a];
            if (!Wm[x]) continue;
            if (checkpts && !F[x]) continue;
            int64_t y_min = -1; double c_min = 0.0;
            for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {
                const int64_t y = rowval[b];
                if (!Hm[y]) continue;
                const double c = C[y] + nzval[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; }
            }
            if (y_min < 0) continue;
            int32_t ns = 0;
            const int ok = orc_dubins_is_free_motion(X + 3 * y_min, X + 3 * x, rt, sp, lohi, M, ss_lo, ss_hi, &ns);
            count += ns;
            if (ok) { A[x] = y_min; C[x] = c_min; heap_push(&heap, x, c_min); Hnew[nnew++] = x; Wm[x] = 0; }
        }
        for (int64_t a = 0; a < nnew; ++a) Hm[Hnew[a]] = 1;
        Hm[z] = 0;
        if (heap.n > 0) z = heap_pop(&heap); else break;
    }
    int64_t len = 0, c2 = z;
    rev[len++] = c2;
    while (c2 != 0) { c2 = A[c2]; if (c2 < 0) break; rev[len++] = c2; }
    for (int64_t i = 0; i < len; ++i) path[i] = rev[len - 1 - i];
    res->status = CAR_GOAL(X + 3 * z);
    res->cost = C[z]; res->z = z; res->collision_checks = count; res->path_len = len; res->nn_queries = 0;
#undef CAR_FREE_STATE
#undef CAR_GOAL
    free(F); free(rowptr); free(colidx); free(cur); free(Wm); free(Hm); free(Hnew); free(rev); free(heap.pri); free(heap.idx);
    return 0;
}

/* ---- Monte-Carlo collision probability of an edge (BASELINE configs[4] / SURVEY 8d cfg5) ------------------------------
 * The reference has no implementation (README.md:9-10 cites papers only); SURVEY 8d defines the workload: per candidate
 * edge, many perturbed copies of the 2-point trajectory, each swept with the segment test of boxesND.jl:44-56.  Declared
 * here so that a scalar loop and the device kernel agree bit for bit (integer sums, unfused fp64, no transcendentals):
 *   rollout k of edge e, coordinate c (0..d-1 of the parent v, d..2d-1 of the child w):
 *     Philox4x32-10 (key = seed, counter = (k, e, c, 2)) -> 8 halfwords h_0..h_7;  S = sum h_i  (Irwin-Hall of 8 uniforms)
 *     z = ((double)S - 262140.0) * ORC_MC_SCALE     (mean 0, variance 1: 8 * (65536^2 - 1) / 12 = 53509.92...^2)
 *     v'_c = v_c + sigma * z   (resp. w');
 *   outcome = !is_free_motion(v', w', CC, SS)  (statespaces.jl:153-158: in_state_space(v') && segment test);
 *   hits[e] = number of colliding rollouts; the estimate is hits / rollouts. */
#define ORC_MC_SCALE (1.0 / 53509.91992145008)
static double mc_normal(uint64_t seed, uint32_t k, uint32_t e, uint32_t c)
{
    const uint32_t ctr[4] = {k, e, c, 2u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    orc_philox4x32_10(ctr, key, x);
    uint32_t S = 0;
    for (int i = 0; i < 4; ++i) S += (x[i] & 0xffffu) + (x[i] >> 16);
    return ((double)S - 262140.0) * ORC_MC_SCALE;
}

void orc_mc_edges(const double *X, int32_t d, const int64_t *src, const int64_t *dst, int64_t E, double sigma, int64_t rollouts,
                  uint64_t seed, const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, int64_t *hits)
{
    double v[ORC_MAXD], w[ORC_MAXD];
    for (int64_t e = 0; e < E; ++e) {
        const double *v0 = X + (size_t)src[e] * d, *w0 = X + (size_t)dst[e] * d;
        int64_t h = 0;
        for (int64_t k = 0; k < rollouts; ++k) {
            for (int32_t c = 0; c < d; ++c) {
                const double zv = mc_normal(seed, (uint32_t)k, (uint32_t)e, (uint32_t)c);
                const double zw = mc_normal(seed, (uint32_t)k, (uint32_t)e, (uint32_t)(d + c));
                const double pv = sigma * zv, pw = sigma * zw;
                v[c] = v0[c] + pv; w[c] = w0[c] + pw;
            }
            if (!orc_is_free_motion(v, w, d, lohi, M, ss_lo, ss_hi)) ++h;
        }
        hits[e] = h;
    }
}

/* ---- Importance-sampling estimator of the same probability ---------------------------------------------
 * The papers README.md:9-10 cites estimate the collision probability of a trajectory under Gaussian-like tracking error by sampling
 * from a MIXTURE: the nominal noise and the noise shifted towards the closest obstacle points, with likelihood-ratio weights.
 * Declared here so that the scalar loop and the device kernel agree bit for bit:
 *   closest points: for every box k and the five points p_t = v + t (w - v), t = 0, 1/4, 1/2, 3/4, 1, of the nominal segment,
 *     c = clamp(p_t, lo_k, hi_k) (closest(p, BB, I) of boxesND.jl:61-86 with W = I) and d2 = sum_i (c_i - p_t,i)^2; the box keeps its
 *     smallest d2 (first minimum over t); the K = min(3, M) boxes with the smallest d2 (first minima over k) give the shifts, in noise
 *     units and the same for both end points: s_j,i = clip((c_i - p_t,i) / sigma, -3, 3);
 *   rollout k: noise z as in orc_mc_edges;  Philox(key = seed, counter = (k, e, 2 d, 3)) -> words x0, x1: the rollout is nominal when
 *     x0 & 1 == 0, else it takes shift j = x1 mod K;  y = z (+ s_j);  v' = v + sigma y_v, w' = w + sigma y_w;
 *     hit = !is_free_motion(v', w', CC, SS);
 *   weight = f(y) / (0.5 f(y) + sum_j (0.5 / K) f(y - s_j)),  f = product over the 2 d coordinates of the Irwin-Hall(8) density g at
 *     x = (y * 53509.92 + 262140) / 65536,  g(x) = (1 / 5040) sum_{k<4} (-1)^k C(8, k) max(t - k, 0)^7 at t = min(x, 8 - x);
 *   the estimate is sum_k hit_k weight_k / rollouts; weights are quantised to 2^-40 and summed as integers (wsum[e]), so the order of
 *   the sum does not matter. */
#define ORC_MC_INV 53509.91992145008
#define ORC_IS_K 3
static double ih8_pdf(double x)
{
    const double t = (x < 8.0 - x) ? x : 8.0 - x;
    if (!(t > 0.0)) return 0.0;
    double acc = 0.0;
    static const double cf[4] = {1.0, -8.0, 28.0, -56.0};
    for (int k = 0; k < 4; ++k) {
        const double u = t - (double)k;
        if (!(u > 0.0)) break;
        const double u2 = u * u, u4 = u2 * u2, u3 = u2 * u;
        const double u7 = u4 * u3;
        const double term = cf[k] * u7;
        acc = acc + term;
    }
    return acc * (1.0 / 5040.0);
}

void orc_mc_is_edges(const double *X, int32_t d, const int64_t *src, const int64_t *dst, int64_t E, double sigma, int64_t rollouts,
                     uint64_t seed, const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *wsum)
{
    double v[ORC_MAXD], w[ORC_MAXD], s[ORC_IS_K][ORC_MAXD], y[2 * ORC_MAXD];
    double *bd2 = (double *)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    int32_t *bt = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    for (int64_t e = 0; e < E; ++e) {
        const double *v0 = X + (size_t)src[e] * d, *w0 = X + (size_t)dst[e] * d;
        /* every box's closest approach to the five points of the nominal segment */
        for (int32_t k = 0; k < M; ++k) {
            const double *lo = lohi + (size_t)k * 2 * d, *hi = lo + d;
            double best = 0.0; int32_t tb = -1;
            for (int32_t t = 0; t < 5; ++t) {
                const double tt = 0.25 * (double)t;
                double d2 = 0.0;
                for (int32_t i = 0; i < d; ++i) {
                    const double df = w0[i] - v0[i], pr = tt * df;
                    const double p = v0[i] + pr;
                    const double c = (p < lo[i]) ? lo[i] : ((p > hi[i]) ? hi[i] : p);
                    const double g = c - p, gg = g * g;
                    d2 = (i == 0) ? gg : d2 + gg;
                }
                if (tb < 0 || d2 < best) { best = d2; tb = t; }
            }
            bd2[k] = best; bt[k] = tb;
        }
        const int32_t K = (M < ORC_IS_K) ? M : ORC_IS_K;
        int32_t chosen[ORC_IS_K];
        for (int32_t j = 0; j < K; ++j) {
            int32_t kb = -1;
            for (int32_t k = 0; k < M; ++k) {
                int taken = 0;
                for (int32_t q = 0; q < j; ++q) taken |= (chosen[q] == k);
                if (taken) continue;
                if (kb < 0 || bd2[k] < bd2[kb]) kb = k;
            }
            chosen[j] = kb;
            const double *lo = lohi + (size_t)kb * 2 * d, *hi = lo + d;
            const double tt = 0.25 * (double)bt[kb];
            for (int32_t i = 0; i < d; ++i) {
                const double df = w0[i] - v0[i], pr = tt * df;
                const double p = v0[i] + pr;
                const double c = (p < lo[i]) ? lo[i] : ((p > hi[i]) ? hi[i] : p);
                double q = 0.0;
                if (sigma > 0.0) {
                    q = (c - p) / sigma;
                    q = (q < -3.0) ? -3.0 : ((q > 3.0) ? 3.0 : q);
                }
                s[j][i] = q;
            }
        }
        const double cj = (K > 0) ? 0.5 / (double)K : 0.0;
        uint64_t acc = 0;
        for (int64_t k = 0; k < rollouts; ++k) {
            const uint32_t ctr[4] = {(uint32_t)k, (uint32_t)e, (uint32_t)(2 * d), 3u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
            uint32_t xw[4];
            orc_philox4x32_10(ctr, key, xw);
            const int32_t comp = (K > 0 && (xw[0] & 1u)) ? (int32_t)(xw[1] % (uint32_t)K) : -1;
            for (int32_t c = 0; c < d; ++c) {
                const double zv = mc_normal(seed, (uint32_t)k, (uint32_t)e, (uint32_t)c);
                const double zw = mc_normal(seed, (uint32_t)k, (uint32_t)e, (uint32_t)(d + c));
                const double sh = (comp >= 0) ? s[comp][c] : 0.0;
                y[c] = zv + sh; y[d + c] = zw + sh;
                const double pv = sigma * y[c], pw = sigma * y[d + c];
                v[c] = v0[c] + pv; w[c] = w0[c] + pw;
            }
            if (orc_is_free_motion(v, w, d, lohi, M, ss_lo, ss_hi)) continue;
            double a = 1.0;                                  /* f(y) up to the common constant */
            for (int32_t c = 0; c < 2 * d; ++c) {
                const double xa = (y[c] * ORC_MC_INV + 262140.0) * (1.0 / 65536.0);
                a = a * ih8_pdf(xa);
            }
            double den = 0.5 * a;
            for (int32_t j = 0; j < K; ++j) {
                double b = 1.0;                              /* f(y - s_j) */
                for (int32_t c = 0; c < 2 * d; ++c) {
                    const double yb = y[c] - s[j][c < d ? c : c - d];
                    const double xb = (yb * ORC_MC_INV + 262140.0) * (1.0 / 65536.0);
                    b = b * ih8_pdf(xb);
                }
                const double tb = cj * b;
                den = den + tb;
            }
            if (K == 0) den = a;                             /* no obstacle: plain Monte Carlo (nothing can collide anyway) */
            const double wgt = (den > 0.0) ? a / den : 0.0;
            acc += (uint64_t)(wgt * 1099511627776.0);          /* 2^40 */
        }
        wsum[e] = acc;
    }
    free(bd2); free(bt);
}

/* ---- ADAPTIVE importance sampling of the same probability (BASELINE configs[4]: "adaptive-importance-sampling") ----
 * Two stages per edge, a cross-entropy update of the proposal's mean from a pilot:
 *   pilot: ORC_AIS_NP = 4096 rollouts with the noise INFLATED by ALPHA = 1.625: rollout k, coordinate c < 2 d: Irwin-Hall integer S from
 *     Philox(key = seed, counter = (k, e, 64 + c, 2)), Z = S - 262140, z = Z * ORC_MC_SCALE, y = ALPHA * z; v' = v + sigma y_v,
 *     w' = w + sigma y_w; hit = !is_free_motion(v', w', CC, SS).  A hit's likelihood ratio against the nominal noise, up to the
 *     factor ALPHA^(2 d) common to all rollouts:  lr = prod_c g(x(y_c)) / prod_c g(x(z_c))  (g, x as in orc_mc_is_edges; each factor
 *     is <= 1), quantised Wq = (uint64)(lr * 2^30).  Integer sums over the hits: SW = sum Wq, A_c = sum Wq * Z_c.
 *   shift (the mean of the nominal noise GIVEN a collision, in noise units):  mu_c = clip(ALPHA * ((double)A_c * ORC_MC_SCALE) /
 *     (double)SW, -3, 3), 0 when the pilot saw no collision (the estimator is then plain Monte Carlo).
 *   main: rollout k: z as in orc_mc_edges; shifted when Philox(key = seed, counter = (k, e, 2 d, 3)) word 0 is odd: y = z (+ mu);
 *     weight = a / (0.5 a + 0.5 b), a = prod_c g(x(y_c)), b = prod_c g(x(y_c - mu_c)); wsum[e] = sum over colliding rollouts of
 *     (uint64)(weight * 2^40); estimate = wsum / (rollouts * 2^40).  All sums are integers: any order gives the same result. */
#define ORC_AIS_NP 4096
#define ORC_AIS_ALPHA 1.625
static int32_t mc_normal_int(uint64_t seed, uint32_t k, uint32_t e, uint32_t c)
{
    const uint32_t ctr[4] = {k, e, c, 2u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t x[4];
    orc_philox4x32_10(ctr, key, x);
    uint32_t S = 0;
    for (int i = 0; i < 4; ++i) S += (x[i] & 0xffffu) + (x[i] >> 16);
    return (int32_t)S - 262140;
}

void orc_mc_ais_edges(const double *X, int32_t d, const int64_t *src, const int64_t *dst, int64_t E, double sigma, int64_t rollouts,
                      uint64_t seed, const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *wsum, double *shifts)
{
    double v[ORC_MAXD], w[ORC_MAXD], y[2 * ORC_MAXD], mu[2 * ORC_MAXD];
    int32_t Z[2 * ORC_MAXD];
    for (int64_t e = 0; e < E; ++e) {
        const double *v0 = X + (size_t)src[e] * d, *w0 = X + (size_t)dst[e] * d;
        uint64_t SW = 0;
        int64_t A[2 * ORC_MAXD];
        for (int32_t c = 0; c < 2 * d; ++c) A[c] = 0;
        for (int64_t k = 0; k < ORC_AIS_NP; ++k) {
            double num = 1.0, den = 1.0;
            for (int32_t c = 0; c < 2 * d; ++c) {
                Z[c] = mc_normal_int(seed, (uint32_t)k, (uint32_t)e, (uint32_t)(64 + c));
                const double z = (double)Z[c] * ORC_MC_SCALE;
                y[c] = ORC_AIS_ALPHA * z;
                const double xy = (y[c] * ORC_MC_INV + 262140.0) * (1.0 / 65536.0), xz = (z * ORC_MC_INV + 262140.0) * (1.0 / 65536.0);
                num = num * ih8_pdf(xy); den = den * ih8_pdf(xz);
            }
            for (int32_t c = 0; c < d; ++c) {
                const double pv = sigma * y[c], pw = sigma * y[d + c];
                v[c] = v0[c] + pv; w[c] = w0[c] + pw;
            }
            if (orc_is_free_motion(v, w, d, lohi, M, ss_lo, ss_hi)) continue;
            const double lr = (den > 0.0) ? num / den : 0.0;
            const uint64_t Wq = (uint64_t)(lr * 1073741824.0);               /* 2^30 */
            SW += Wq;
            for (int32_t c = 0; c < 2 * d; ++c) A[c] += (int64_t)Wq * (int64_t)Z[c];
        }
        for (int32_t c = 0; c < 2 * d; ++c) {
            double m = 0.0;
            if (SW > 0) {
                const double t = (double)A[c] * ORC_MC_SCALE;
                m = ORC_AIS_ALPHA * t / (double)SW;
                m = (m < -3.0) ? -3.0 : ((m > 3.0) ? 3.0 : m);
            }
            mu[c] = m;
            if (shifts) shifts[(size_t)e * 2 * d + c] = m;
        }
        uint64_t acc = 0;
        for (int64_t k = 0; k < rollouts; ++k) {
            const uint32_t ctr[4] = {(uint32_t)k, (uint32_t)e, (uint32_t)(2 * d), 3u}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
            uint32_t xw[4];
            orc_philox4x32_10(ctr, key, xw);
            const int shifted = (int)(xw[0] & 1u);
            for (int32_t c = 0; c < 2 * d; ++c) {
                const double z = mc_normal(seed, (uint32_t)k, (uint32_t)e, (uint32_t)c);
                y[c] = shifted ? z + mu[c] : z;
            }
            for (int32_t c = 0; c < d; ++c) {
                const double pv = sigma * y[c], pw = sigma * y[d + c];
                v[c] = v0[c] + pv; w[c] = w0[c] + pw;
            }
            if (orc_is_free_motion(v, w, d, lohi, M, ss_lo, ss_hi)) continue;
            double a = 1.0, b = 1.0;
            for (int32_t c = 0; c < 2 * d; ++c) {
                const double xa = (y[c] * ORC_MC_INV + 262140.0) * (1.0 / 65536.0);
                a = a * ih8_pdf(xa);
                const double yb = y[c] - mu[c];
                const double xb = (yb * ORC_MC_INV + 262140.0) * (1.0 / 65536.0);
                b = b * ih8_pdf(xb);
            }
            const double ha = 0.5 * a, hb = 0.5 * b;
            const double dn = ha + hb;
            const double wgt = (dn > 0.0) ? a / dn : 0.0;
            acc += (uint64_t)(wgt * 1099511627776.0);          /* 2^40 */
        }
        wsum[e] = acc;
    }
}

/* ---- Reeds-Shepp space: chopped METRIC (ChoppedMetric, MetricNN): inball(v) = { w : |xy_v - xy_w| <= r, rs(v -> w) <= r } with
 * ds = rs(v -> w) (colwise(dist, V[v], V[inds]), nearneighbors.jl:185-198); forward and backward sets coincide (:200-203).
 * CSC: column v = inball(v).  Two-phase like orc_dubins_graph. */
int64_t orc_rs_graph(const double *X, int64_t N, double rt, double sp, double r, int64_t *colptr, int64_t *rowval, double *nzval)
{
    int64_t nnz = 0;
    orc_step path[5];
    int32_t L;
    for (int64_t v = 0; v < N; ++v) {
        if (colptr) colptr[v] = nnz;
        for (int64_t w = 0; w < N; ++w) {
            if (w == v) continue;
            const double dx = X[3 * w] - X[3 * v], dy = X[3 * w + 1] - X[3 * v + 1];
            const double px = dx * dx, py = dy * dy;
            if (!(px + py <= r * r)) continue;
            const double c = orc_reedsshepp(X + 3 * v, X + 3 * w, rt, sp, path, &L);
            if (c <= r) { if (rowval) { rowval[nnz] = w; nzval[nnz] = c; } ++nnz; }
        }
    }
    if (colptr) colptr[N] = nnz;
    return nnz;
}

/* entry e (row y, column x): is_free_motion(V[y], V[x], CC, SS) (fmt.jl:75: parent first), kind 1 Dubins / 2 Reeds-Shepp */
void orc_car_graph_edges_free(int32_t kind, const double *X, int64_t N, double rt, double sp, const int64_t *colptr, const int64_t *rowval,
                              const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi, uint64_t *mask, uint8_t *nseg)
{
    memset(mask, 0, sizeof(uint64_t) * (size_t)((colptr[N] + 63) / 64));
    for (int64_t x = 0; x < N; ++x)
        for (int64_t e = colptr[x]; e < colptr[x + 1]; ++e) {
            int32_t ns = 0;
            set_bit(mask, e, orc_car_is_free_motion(kind, X + 3 * rowval[e], X + 3 * x, rt, sp, lohi, M, ss_lo, ss_hi, &ns));
            if (nseg) nseg[e] = (uint8_t)ns;
        }
}

/* fmtstar! over the Reeds-Shepp space (symmetric neighbour sets: near(z) = column z, near(x) & H scans column x with
 * c = C[y] + ds(x, y), fmt.jl:70-75); lazy edge checks counted per waypoint segment. */
int32_t orc_rs_fmtstar(const double *X, int64_t N, double rt, double sp, int64_t init_idx, int32_t checkpts,
                       const int64_t *colptr, const int64_t *rowval, const double *nzval,
                       int32_t goal_kind, const double *goal, const double *lohi, int32_t M, const double *ss_lo, const double *ss_hi,
                       int64_t *A, double *C, int64_t *path, orc_fmt_result *res)
{
    memset(res, 0, sizeof *res);
    res->cost = INFINITY;
#define CAR_FREE_STATE(p) (orc_in_state_space((p), ss_lo, ss_hi, 3) && orc_point_free_boxes((p), lohi, M, 2))
#define CAR_GOAL(p) ((goal_kind == 2) ? ((p)[0] == goal[0] && (p)[1] == goal[1] && (p)[2] == goal[2]) : orc_is_goal_pt((p), 2, goal_kind, goal))
    if (!CAR_FREE_STATE(X + 3 * init_idx)) return -1;
    uint8_t *F = NULL;
    if (checkpts) { F = (uint8_t *)malloc((size_t)N); for (int64_t i = 0; i < N; ++i) F[i] = (uint8_t)CAR_FREE_STATE(X + 3 * i); }
    uint8_t *Wm = (uint8_t *)malloc((size_t)N), *Hm = (uint8_t *)calloc((size_t)N, 1);
    memset(Wm, 1, (size_t)N);
    for (int64_t i = 0; i < N; ++i) { A[i] = -1; C[i] = 0.0; }
    int64_t *Hnew = (int64_t *)malloc(sizeof(int64_t) * (size_t)N), *rev = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    orc_heap heap = {0};
    Wm[init_idx] = 0; Hm[init_idx] = 1;
    heap_push(&heap, init_idx, 0.0);
    int64_t z = heap_pop(&heap), count = 0;
    while (!CAR_GOAL(X + 3 * z)) {
        int64_t nnew = 0;
        for (int64_t a = colptr[z]; a < colptr[z + 1]; ++a) {
            const int64_t x = rowval[a];
            if (!Wm[x]) continue;
            if (checkpts && !F[x]) continue;
            int64_t y_min = -1; double c_min = 0.0;
            for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {
                const int64_t y = rowval[b];
                if (!Hm[y]) continue;
                const double c = C[y] + nzval[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; }
            }
            if (y_min < 0) continue;
            int32_t ns = 0;
            const int ok = orc_car_is_free_motion(2, X + 3 * y_min, X + 3 * x, rt, sp, lohi, M, ss_lo, ss_hi, &ns);
            count += ns;
            if (ok) { A[x] = y_min; C[x] = c_min; heap_push(&heap, x, c_min); Hnew[nnew++] = x; Wm[x] = 0; }
        }
        for (int64_t a = 0; a < nnew; ++a) Hm[Hnew[a]] = 1;
        Hm[z] = 0;
        if (heap.n > 0) z = heap_pop(&heap); else break;
    }
    int64_t len = 0, c2 = z;
    rev[len++] = c2;
    while (c2 != 0) { c2 = A[c2]; if (c2 < 0) break; rev[len++] = c2; }
    for (int64_t i = 0; i < len; ++i) path[i] = rev[len - 1 - i];
    res->status = CAR_GOAL(X + 3 * z);
    res->cost = C[z]; res->z = z; res->collision_checks = count; res->path_len = len; res->nn_queries = 0;
#undef CAR_FREE_STATE
#undef CAR_GOAL
    free(F); free(Wm); free(Hm); free(Hnew); free(rev); free(heap.pri); free(heap.idx);
    return 0;
}

/* ======================================================================================================================
 * Closest obstacle points in a Mahalanobis metric: closest / closeR of
 * src/collisioncheckers/boxesND.jl:61-86 (boxes, through bvls.jl:19-222) and src/collisioncheckers/SAT2D.jl:208-285
 * (circles, convex polygons, compounds).  Test infrastructure like the rest of this file.
 * ==================================================================================================================== */

/* chol(W) (boxesND.jl:65): the upper factor U with U'U = W; returns -1 when W is not positive definite */
static int chol_upper(const double *W, int n, double U[ORC_MAXD][ORC_MAXD])
{
    for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) U[j][i] = 0.0;
        double s = W[j * n + j];
        for (int k = 0; k < j; ++k) s = s - U[k][j] * U[k][j];
        if (!(s > 0.0)) return -1;
        U[j][j] = sqrt(s);
        for (int i = j + 1; i < n; ++i) {
            double t = W[j * n + i];
            for (int k = 0; k < j; ++k) t = t - U[k][j] * U[k][i];
            U[j][i] = t / U[j][j];
        }
    }
    return 0;
}

/* z = A \ b for an m x k matrix of full column rank, k <= m (bvls.jl:136): Householder QR, then back substitution.
 * A and b are overwritten. */
static void ls_solve(int m, int k, double A[ORC_MAXD][ORC_MAXD], double *b, double *z)
{
    for (int j = 0; j < k; ++j) {
        double nrm = 0.0;
        for (int i = j; i < m; ++i) nrm = nrm + A[i][j] * A[i][j];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = (A[j][j] > 0.0) ? -nrm : nrm;
        double v[ORC_MAXD];
        for (int i = j; i < m; ++i) v[i] = A[i][j];
        v[j] = v[j] - alpha;
        double vv = 0.0;
        for (int i = j; i < m; ++i) vv = vv + v[i] * v[i];
        if (vv == 0.0) continue;
        for (int c = j; c < k; ++c) {
            double s = 0.0;
            for (int i = j; i < m; ++i) s = s + v[i] * A[i][c];
            s = 2.0 * s / vv;
            for (int i = j; i < m; ++i) A[i][c] = A[i][c] - s * v[i];
        }
        double s = 0.0;
        for (int i = j; i < m; ++i) s = s + v[i] * b[i];
        s = 2.0 * s / vv;
        for (int i = j; i < m; ++i) b[i] = b[i] - s * v[i];
    }
    for (int j = k - 1; j >= 0; --j) {
        double s = b[j];
        for (int c = j + 1; c < k; ++c) s = s - A[j][c] * z[c];
        z[j] = s / A[j][j];
    }
}

/* bvls(A, b, l, u) (bvls.jl:19-218), A n x n here.  Returns the iteration count, or -1 when the 10n iterations run out
 * (the reference then returns `nothing`).  The bookkeeping quirks are kept: a freed variable that wants to leave through
 * its own bound is locked again with state 0 (:151-164, state was zeroed at :113), and criti persists across iterations. */
int32_t orc_bvls(int32_t n, const double *Aflat, const double *b, const double *l, const double *u, double *x)
{
    double A[ORC_MAXD][ORC_MAXD];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i][j] = Aflat[i * n + j];
    int oops[ORC_MAXD] = {0}, state[ORC_MAXD], atbound[ORC_MAXD] = {0}, between[ORC_MAXD] = {0};
    int criti = -1, crits = 0;
    const double myeps = 1.0e-10;
    for (int i = 0; i < n; ++i) {                                                            /* :41-63 */
        if (u[i] >= INFINITY && l[i] <= -INFINITY) { x[i] = 0.0; state[i] = 0; between[i] = 1; }
        else if (u[i] >= INFINITY) { x[i] = l[i]; state[i] = 1; atbound[i] = 1; }
        else if (l[i] <= -INFINITY) { x[i] = u[i]; state[i] = 2; atbound[i] = 1; }
        else if (fabs(l[i]) <= fabs(u[i])) { x[i] = l[i]; state[i] = 1; atbound[i] = 1; }
        else { x[i] = u[i]; state[i] = 2; atbound[i] = 1; }
    }
    double nb = 0.0;
    for (int i = 0; i < n; ++i) nb = nb + b[i] * b[i];
    nb = sqrt(nb);
    for (int iter = 1; iter <= 10 * n; ++iter) {                                            /* :67-69 */
        double res[ORC_MAXD], grad[ORC_MAXD];
        for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s = s + A[i][j] * x[j]; res[i] = s - b[i]; }
        for (int j = 0; j < n; ++j) { double s = 0.0; for (int i = 0; i < n; ++i) s = s + A[i][j] * res[i]; grad[j] = oops[j] ? 0.0 : s; }   /* :73-77 */
        int done = 1;
        for (int i = 0; i < n; ++i)                                                          /* :81-89 */
            if ((fabs(grad[i]) > (1.0 + nb) * myeps && state[i] == 0) || (grad[i] < 0.0 && state[i] == 1) || (grad[i] > 0.0 && state[i] == 2)) { done = 0; break; }
        if (done) return iter;
        int newi = -1; double newg = 0.0;                                                    /* :94-112 */
        for (int i = 0; i < n; ++i) {
            if (!atbound[i] || i == criti) continue;
            if (grad[i] > 0.0 && state[i] == 2 && fabs(grad[i]) > newg) { newi = i; newg = fabs(grad[i]); }
            if (grad[i] < 0.0 && state[i] == 1 && fabs(grad[i]) > newg) { newi = i; newg = fabs(grad[i]); }
        }
        if (newi >= 0) { atbound[newi] = 0; state[newi] = 0; between[newi] = 1; }           /* :116-120 */
        double Ap[ORC_MAXD][ORC_MAXD], bp[ORC_MAXD], z[ORC_MAXD], xnew[ORC_MAXD];
        int cols[ORC_MAXD], k = 0;
        for (int j = 0; j < n; ++j) if (between[j]) cols[k++] = j;
        for (int i = 0; i < n; ++i) {                                                        /* :131-138 */
            double s = 0.0;
            for (int j = 0; j < n; ++j) if (atbound[j]) s = s + A[i][j] * x[j];
            bp[i] = b[i] - s;
            for (int c = 0; c < k; ++c) Ap[i][c] = A[i][cols[c]];
        }
        ls_solve(n, k, Ap, bp, z);                                                           /* :142 */
        for (int i = 0; i < n; ++i) xnew[i] = x[i];
        for (int c = 0; c < k; ++c) xnew[cols[c]] = z[c];
        if (newi >= 0 && ((xnew[newi] <= l[newi] && x[newi] == l[newi]) || (xnew[newi] >= u[newi] && x[newi] == u[newi]))) {   /* :146-165 */
            oops[newi] = 1;
            if (xnew[newi] <= l[newi] && state[newi] == 1) { state[newi] = 1; x[newi] = l[newi]; }
            if (xnew[newi] >= u[newi] && state[newi] == 2) { state[newi] = 2; x[newi] = u[newi]; }
            atbound[newi] = 1; between[newi] = 0;
            continue;
        }
        for (int i = 0; i < n; ++i) oops[i] = 0;                                             /* :169 */
        double alpha = 1.0;                                                                  /* :174-195 */
        for (int i = 0; i < n; ++i) {
            if (!between[i]) continue;
            if (xnew[i] > u[i]) { const double na = fmin(alpha, (u[i] - x[i]) / (xnew[i] - x[i])); if (na < alpha) { criti = i; crits = 2; alpha = na; } }
            if (xnew[i] < l[i]) { const double na = fmin(alpha, (l[i] - x[i]) / (xnew[i] - x[i])); if (na < alpha) { criti = i; crits = 1; alpha = na; } }
        }
        for (int i = 0; i < n; ++i) x[i] = x[i] + alpha * (xnew[i] - x[i]);                  /* :199 */
        if (alpha < 1.0) { between[criti] = 0; atbound[criti] = 1; state[criti] = crits; }   /* :203-207 */
        for (int i = 0; i < n; ++i) {                                                        /* :209-222 */
            if (x[i] >= u[i]) { x[i] = u[i]; state[i] = 2; between[i] = 0; atbound[i] = 1; }
            if (x[i] <= l[i]) { x[i] = l[i]; state[i] = 1; between[i] = 0; atbound[i] = 1; }
        }
    }
    return -1;
}

static double quad_form(const double *W, int n, const double *v, const double *p)      /* dot(v - p, W*(v - p)) */
{
    double t[ORC_MAXD], acc = 0.0;
    for (int i = 0; i < n; ++i) t[i] = v[i] - p[i];
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int j = 0; j < n; ++j) s = s + W[i * n + j] * t[j]; acc = acc + t[i] * s; }
    return acc;
}

/* closest(p, BB, W) (boxesND.jl:61-70): d2min; v = closest point of the box lo..hi.  Returns bvls' iteration count, -1 on
 * its failure, -2 when W is not positive definite. */
int32_t orc_closest_box(const double *p, const double *lo, const double *hi, const double *W, int32_t n, double *d2, double *v)
{
    double U[ORC_MAXD][ORC_MAXD], A[ORC_MAXD * ORC_MAXD], b[ORC_MAXD];
    if (chol_upper(W, n, U)) return -2;
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) { A[i * n + j] = U[i][j]; s = s + U[i][j] * p[j]; }
        b[i] = s;
    }
    const int32_t it = orc_bvls(n, A, b, lo, hi, v);
    if (it < 0) { *d2 = NAN; return it; }
    *d2 = quad_form(W, n, v, p);
    return it;
}

/* closest(p, BL, W) (boxesND.jl:72-81) for each of the n points: strict <, so the first minimum wins; no boxes: (Inf, p).
 * lohi [M][2][d]; kmin 0-based, -1 = none.  Returns the number of bvls failures. */
int64_t orc_closest_boxes(const double *P, int64_t n, const double *lohi, int32_t M, const double *W, int32_t d,
                          double *d2min, double *vmin, int64_t *kmin)
{
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = P + (size_t)i * d;
        double best = INFINITY; int64_t kb = -1;
        for (int q = 0; q < d; ++q) vmin[(size_t)i * d + q] = p[q];
        for (int32_t k = 0; k < M; ++k) {
            double d2, v[ORC_MAXD];
            if (orc_closest_box(p, lohi + (size_t)k * 2 * d, lohi + (size_t)k * 2 * d + d, W, d, &d2, v) < 0) { ++bad; continue; }
            if (d2 < best) { best = d2; kb = k; for (int q = 0; q < d; ++q) vmin[(size_t)i * d + q] = v[q]; }
        }
        d2min[i] = best; if (kmin) kmin[i] = kb;
    }
    return bad;
}

typedef struct { double d2; int32_t k; double v[ORC_MAXD]; } orc_cp;
static void cp_sort(orc_cp *a, int n)                      /* stable (sort! by=first is a merge sort): insertion sort, strict > */
{
    for (int i = 1; i < n; ++i) {
        orc_cp t = a[i]; int j = i - 1;
        while (j >= 0 && a[j].d2 > t.d2) { a[j + 1] = a[j]; --j; }
        a[j + 1] = t;
    }
}

/* closeR(p, BL, W, r2) (boxesND.jl:83-86) for each point: obstacles with d2 < r2, ascending d2 (ties in obstacle order).
 * ptr [n+1] always written; idx / d2 / v (may be NULL: counting pass) receive the lists back to back. */
int64_t orc_closeR_boxes(const double *P, int64_t n, const double *lohi, int32_t M, const double *W, int32_t d, double r2,
                         int64_t *ptr, int64_t *idx, double *d2out, double *vout)
{
    orc_cp *cps = (orc_cp *)malloc(sizeof(orc_cp) * (size_t)(M > 0 ? M : 1));
    int64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) {
        const double *p = P + (size_t)i * d;
        int c = 0;
        for (int32_t k = 0; k < M; ++k) {
            orc_cp t; t.k = k;
            if (orc_closest_box(p, lohi + (size_t)k * 2 * d, lohi + (size_t)k * 2 * d + d, W, d, &t.d2, t.v) < 0) continue;
            if (t.d2 < r2) cps[c++] = t;
        }
        cp_sort(cps, c);
        ptr[i] = tot;
        if (idx) for (int j = 0; j < c; ++j) {
            idx[tot + j] = cps[j].k; d2out[tot + j] = cps[j].d2;
            for (int q = 0; q < d; ++q) vout[(size_t)(tot + j) * d + q] = cps[j].v[q];
        }
        tot += c;
    }
    ptr[n] = tot;
    free(cps);
    return tot;
}

/* ---- 2-D shapes (SAT2D.jl:208-285) ---- */

/* eigfact of the symmetric 2x2 W: ascending values, orthonormal vectors (closed form; LAPACK differs in the last bits and
 * possibly in the vectors' signs, which cancel in every use below) */
static void eig2(const double *W, double *s1, double *s2, double *v1, double *v2)
{
    const double a = W[0], b = 0.5 * (W[1] + W[2]), c = W[3];
    const double h = 0.5 * (a - c), m = 0.5 * (a + c), rad = sqrt(h * h + b * b);
    *s1 = m - rad; *s2 = m + rad;
    if (b == 0.0) {
        if (a <= c) { v1[0] = 1.0; v1[1] = 0.0; v2[0] = 0.0; v2[1] = 1.0; }
        else        { v1[0] = 0.0; v1[1] = 1.0; v2[0] = 1.0; v2[1] = 0.0; }
        return;
    }
    /* eigenvector of s2: (b, s2 - a) or (s2 - c, b), whichever has the larger first-step magnitude */
    double e0, e1;
    if (fabs(*s2 - a) >= fabs(*s2 - c)) { e0 = b; e1 = *s2 - a; } else { e0 = *s2 - c; e1 = b; }
    const double nn = sqrt(e0 * e0 + e1 * e1);
    v2[0] = e0 / nn; v2[1] = e1 / nn;
    v1[0] = -v2[1]; v1[1] = v2[0];
}

#define CIRC_F(lam) (((p1 * s1 / ((lam) + s1)) * (p1 * s1 / ((lam) + s1)) + (p2 * s2 / ((lam) + s2)) * (p2 * s2 / ((lam) + s2))) - C->r * C->r)
/* closest(p, C::Circle, EF) (SAT2D.jl:213-238): Newton on the multiplier with the reference's halving line search.  The
 * reference's loops are unbounded; 200 Newton steps / 64 halvings stop a run that would not have ended there either
 * (returns -1, outputs NaN). */
static int closest_circle_W(const double *p, const orc_shape2d *C, const double *W, double *d2, double *x)
{
    double s1, s2, v1[2], v2[2];
    eig2(W, &s1, &s2, v1, v2);
    const double ctop[2] = {p[0] - C->c[0], p[1] - C->c[1]};
    const double p1 = dot2(v1, ctop), p2 = dot2(v2, ctop);
    double lambda = 1.0;
    double f = CIRC_F(lambda);
    int it = 0;
    while (fabs(f) > 1e-8) {
        if (++it > 200 || !(f == f)) { *d2 = NAN; x[0] = x[1] = NAN; return -1; }
        const double q1 = p1 * s1 / (lambda + s1), q2 = p2 * s2 / (lambda + s2);
        const double fp = -2.0 / (lambda + s1) * (q1 * q1) + -2.0 / (lambda + s2) * (q2 * q2);
        double alpha = 1.0, lnew = 1.0, fnew = 1.0;
        int h = 0;
        for (;;) {
            lnew = lambda - alpha * f / fp;
            fnew = CIRC_F(lnew);
            if (fabs(fnew) < fabs(f)) break;
            alpha = alpha / 2.0;
            if (++h > 64) { *d2 = NAN; x[0] = x[1] = NAN; return -1; }
        }
        f = fnew; lambda = lnew;
    }
    const double k1 = p1 * s1 / (lambda + s1), k2 = p2 * s2 / (lambda + s2);
    x[0] = (C->c[0] + v1[0] * k1) + v2[0] * k2;
    x[1] = (C->c[1] + v1[1] * k1) + v2[1] * k2;
    *d2 = s1 * ((p1 - k1) * (p1 - k1)) + s2 * ((p2 - k2) * (p2 - k2));
    return 0;
}

/* closest_polypts (SAT2D.jl:240-254) */
static void closest_polypts(const double *p, const double (*pts)[2], int n, double *d2min, double *vmin)
{
    *d2min = INFINITY; vmin[0] = pts[0][0]; vmin[1] = pts[0][1];
    for (int i = 0; i < n; ++i) {
        const int nx = (i + 1 < n) ? i + 1 : 0;
        const double e[2] = {pts[nx][0] - pts[i][0], pts[nx][1] - pts[i][1]};
        const double w[2] = {p[0] - pts[i][0], p[1] - pts[i][1]};
        const double t = dot2(e, w) / dot2(e, e);
        double v[2];
        if (t < 0.0) { v[0] = pts[i][0]; v[1] = pts[i][1]; }
        else if (t < 1.0) { v[0] = pts[i][0] + t * e[0]; v[1] = pts[i][1] + t * e[1]; }
        else { v[0] = pts[nx][0]; v[1] = pts[nx][1]; }
        const double u[2] = {p[0] - v[0], p[1] - v[1]};
        const double dd = dot2(u, u);
        if (dd < *d2min) { *d2min = dd; vmin[0] = v[0]; vmin[1] = v[1]; }
    }
}

/* closest(p, S [, W]) for one shape; W == NULL: the Euclidean methods (SAT2D.jl:208-211, 239).  Returns 0, -1 (circle
 * iteration did not end), -2 (W not positive definite). */
int32_t orc_closest_shape(const double *p, const void *shape, const double *W, double *d2, double *x)
{
    const orc_shape2d *S = (const orc_shape2d *)shape;
    if (!W) {
        if (S->kind == 0) {
            const double v[2] = {p[0] - S->c[0], p[1] - S->c[1]};
            const double nv = sqrt(dot2(v, v));
            x[0] = S->c[0] + S->r * (v[0] / nv); x[1] = S->c[1] + S->r * (v[1] / nv);
            const double u[2] = {p[0] - x[0], p[1] - x[1]};
            *d2 = dot2(u, u);
        } else closest_polypts(p, S->pts, S->n, d2, x);
        return 0;
    }
    if (S->kind == 0) return closest_circle_W(p, S, W, d2, x);
    double U[ORC_MAXD][ORC_MAXD];                                                            /* SAT2D.jl:255-259 */
    if (chol_upper(W, 2, U)) return -2;
    double tp[ORC_MAXPOLY][2];
    for (int i = 0; i < S->n; ++i) { tp[i][0] = U[0][0] * S->pts[i][0] + U[0][1] * S->pts[i][1]; tp[i][1] = U[1][1] * S->pts[i][1]; }
    const double lp[2] = {U[0][0] * p[0] + U[0][1] * p[1], U[1][1] * p[1]};
    double dd, y[2];
    closest_polypts(lp, tp, S->n, &dd, y);
    const double det = U[0][0] * U[1][1];                                                    /* inv of the 2x2 SMatrix: adjugate / det */
    x[0] = (U[1][1] / det) * y[0] + (-U[0][1] / det) * y[1];
    x[1] = (U[0][0] / det) * y[1];
    *d2 = quad_form(W, 2, x, p);
    return 0;
}

/* closest(p, C::Compound2D [, W]) (SAT2D.jl:260-279) per point: strict <, init (Inf, zeros).  Returns failures. */
int64_t orc_closest_shapes(const double *P, int64_t n, const void *shapes, int32_t M, const double *W, double *d2min, double *vmin, int64_t *kmin)
{
    const orc_shape2d *S = (const orc_shape2d *)shapes;
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        double best = INFINITY; int64_t kb = -1;
        vmin[2 * i] = 0.0; vmin[2 * i + 1] = 0.0;
        for (int32_t k = 0; k < M; ++k) {
            double d2, v[2];
            if (orc_closest_shape(P + 2 * i, S + k, W, &d2, v)) { ++bad; continue; }
            if (d2 < best) { best = d2; kb = k; vmin[2 * i] = v[0]; vmin[2 * i + 1] = v[1]; }
        }
        d2min[i] = best; if (kmin) kmin[i] = kb;
    }
    return bad;
}

/* closeR(p, C::Compound2D, W, r2) (SAT2D.jl:281-285) per point */
int64_t orc_closeR_shapes(const double *P, int64_t n, const void *shapes, int32_t M, const double *W, double r2,
                          int64_t *ptr, int64_t *idx, double *d2out, double *vout)
{
    const orc_shape2d *S = (const orc_shape2d *)shapes;
    orc_cp *cps = (orc_cp *)malloc(sizeof(orc_cp) * (size_t)(M > 0 ? M : 1));
    int64_t tot = 0;
    for (int64_t i = 0; i < n; ++i) {
        int c = 0;
        for (int32_t k = 0; k < M; ++k) {
            orc_cp t; t.k = k;
            if (orc_closest_shape(P + 2 * i, S + k, W, &t.d2, t.v)) continue;
            if (t.d2 < r2) cps[c++] = t;
        }
        cp_sort(cps, c);
        ptr[i] = tot;
        if (idx) for (int j = 0; j < c; ++j) { idx[tot + j] = cps[j].k; d2out[tot + j] = cps[j].d2; vout[2 * (tot + j)] = cps[j].v[0]; vout[2 * (tot + j) + 1] = cps[j].v[1]; }
        tot += c;
    }
    ptr[n] = tot;
    free(cps);
    return tot;
}
