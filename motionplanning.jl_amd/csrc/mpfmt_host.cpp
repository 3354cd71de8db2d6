// Host-only part of libmpfmt.so: the sequential dynamic-programming recursion of the reference (src/planners/fmt.jl:43-101)
// over GPU-built arrays, its priority queue, the goal predicates and the validation of imported graphs.  Plain C++17, no HIP:
// this file is compiled into the library by hipcc and, unchanged, into the sanitizer test binary of tests/asan/ by g++
// (-fsanitize=address,undefined).
#include "mpfmt_host.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// binary min-heap on (cost, index): Base.Collections.PriorityQueue of fmt.jl:51,66,78,86.  Ties on cost are
// broken by the lowest sample index (the reference leaves the order of equal priorities unspecified).
struct Heap {
    std::vector<double> pri; std::vector<int64_t> idx;
    bool less(size_t a, size_t b) const { return pri[a] < pri[b] || (pri[a] == pri[b] && idx[a] < idx[b]); }
    void push(int64_t i, double p)
    {
        pri.push_back(p); idx.push_back(i);
        size_t c = pri.size() - 1;
        while (c > 0) { size_t par = (c - 1) / 2; if (less(c, par)) { std::swap(pri[c], pri[par]); std::swap(idx[c], idx[par]); c = par; } else break; }
    }
    int64_t pop()
    {
        int64_t top = idx[0];
        pri[0] = pri.back(); idx[0] = idx.back(); pri.pop_back(); idx.pop_back();
        size_t n = pri.size(), c = 0;
        for (;;) {
            size_t l = 2 * c + 1, r = l + 1, m = c;
            if (l < n && less(l, m)) m = l;
            if (r < n && less(r, m)) m = r;
            if (m == c) break;
            std::swap(pri[c], pri[m]); std::swap(idx[c], idx[m]); c = m;
        }
        return top;
    }
    bool empty() const { return pri.empty(); }
};

// goal predicates, src/goals.jl:96 (Rectangle), :100 (Ball), :111-114 (Point), Identity state2workspace
}  // namespace

bool mpfmt_is_goal_pt(const double* v, int d, int kind, const double* g)
{
    if (kind == MPFMT_GOAL_RECT) {
        for (int i = 0; i < d; ++i) if (!(g[i] <= v[i] && v[i] <= g[d + i])) return false;
        return true;
    }
    if (kind == MPFMT_GOAL_BALL) {
        double s = 0.0;
        for (int i = 0; i < d; ++i) { double t = v[i] - g[i]; double tt = t * t; s = (i == 0) ? tt : s + tt; }
        return std::sqrt(s) <= g[d];
    }
    for (int i = 0; i < d; ++i) if (!(v[i] == g[i])) return false;
    return true;
}


// The sequential recursion of fmt.jl:43-101 on a finished r-disc graph: CSC (0-based colptr / int32 rows, ascending
// rows = the order the reference's neighbourhood scans run in), per-entry free bits (row -> column motions) and the
// optional checkpts bitmap F.  Pure host code, no device use -- the GPU's job ends where this starts.
//   - W and H are bit sets (125 KB each at N = 1e6, cache resident): the inner scan touches C[y] / nzval only for the
//     few open neighbours;
//   - the candidates x in near(z) & W are collected first, so the adjacency rows of the NEXT candidates can be
//     prefetched while the current one is scanned (each row is a random ~400-byte read from a GB-sized array).
// gd = coordinates the goal predicate reads (d for Euclidean spaces; 2 = workspace (x, y) / 3 = whole state for SE2 cars);
// nseg != NULL: per-entry count of the segment tests the reference would make (car spaces), else one test per edge check
int32_t mpfmt_host_fmt_recursion_impl(int64_t N, int32_t d, const double* X, const int64_t* colptr, const int32_t* rowval,
                                       const double* nzval, const uint64_t* efree, const uint64_t* F, const double* ss_lo,
                                       const double* ss_hi, int64_t init_idx, int32_t goal_kind, const double* goal_params, int32_t gd,
                                       const uint8_t* nseg, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);

int32_t mpfmt_host_fmt_recursion(int64_t N, int32_t d, const double* X, const int64_t* colptr, const int32_t* rowval,
                                 const double* nzval, const uint64_t* efree, const uint64_t* F, const double* ss_lo,
                                 const double* ss_hi, int64_t init_idx, int32_t goal_kind, const double* goal_params,
                                 int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{
    return mpfmt_host_fmt_recursion_impl(N, d, X, colptr, rowval, nzval, efree, F, ss_lo, ss_hi, init_idx, goal_kind, goal_params, d, nullptr,
                                   A, C, path, res);
}

int32_t mpfmt_host_fmt_recursion_impl(int64_t N, int32_t d, const double* X, const int64_t* colptr, const int32_t* rowval,
                                       const double* nzval, const uint64_t* efree, const uint64_t* F, const double* ss_lo,
                                       const double* ss_hi, int64_t init_idx, int32_t goal_kind, const double* goal_params, int32_t gd,
                                       const uint8_t* nseg, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res)
{
    if (!X || !colptr || !rowval || !nzval || !efree || !goal_params || !A || !C || !path || !res) return MPFMT_ERR_ARG;
    if (N < 1 || d < 1 || d > MPFMT_MAX_DIM || init_idx < 1 || init_idx > N || goal_kind < 0 || goal_kind > 2) return MPFMT_ERR_ARG;
    if ((ss_lo == nullptr) != (ss_hi == nullptr)) return MPFMT_ERR_ARG;
    const auto t_begin = std::chrono::steady_clock::now();
    const int64_t words = (N + 63) / 64;
    std::vector<uint64_t> Wb((size_t)words, ~0ull), Hb((size_t)words, 0ull);
    auto getb = [](const uint64_t* m, int64_t i) { return (m[(size_t)(i >> 6)] >> (i & 63)) & 1ull; };
    auto setb = [](std::vector<uint64_t>& m, int64_t i) { m[(size_t)(i >> 6)] |= 1ull << (i & 63); };
    auto clrb = [](std::vector<uint64_t>& m, int64_t i) { m[(size_t)(i >> 6)] &= ~(1ull << (i & 63)); };
    std::vector<int64_t> Hnew, cand;
    for (int64_t i = 0; i < N; ++i) { A[i] = 0; C[i] = 0.0; }
    Heap heap;
    const int64_t i0 = init_idx - 1;
    clrb(Wb, i0); setb(Hb, i0);
    heap.push(i0, 0.0);
    int64_t z = heap.pop();
    int64_t count = 0;
    auto prefetch_row = [&](int64_t x) {
        const char* p = (const char*)(rowval + colptr[x]);
        const char* e = (const char*)(rowval + colptr[x + 1]);
        for (int q = 0; q < 8 && p < e; ++q, p += 64) __builtin_prefetch(p, 0, 1);
    };
    while (!mpfmt_is_goal_pt(&X[(size_t)z * d], gd, goal_kind, goal_params)) {
        Hnew.clear();
        cand.clear();
        for (int64_t a = colptr[z]; a < colptr[z + 1]; ++a) {                 // fmt.jl:70-71
            const int64_t x = rowval[a];
            if (getb(Wb.data(), x) && (!F || getb(F, x))) cand.push_back(x);
        }
        const size_t nc = cand.size();
        for (size_t q = 0; q < nc && q < 3; ++q) prefetch_row(cand[q]);
        for (size_t q = 0; q < nc; ++q) {
            if (q + 3 < nc) prefetch_row(cand[q + 3]);
            const int64_t x = cand[q];
            int64_t y_min = -1, e_min = -1; double c_min = 0.0;
            for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {             // fmt.jl:72-74
                const int64_t y = rowval[b];
                if (!getb(Hb.data(), y)) continue;
                const double c = C[y] + nzval[b];
                if (y_min < 0 || c < c_min) { y_min = y; c_min = c; e_min = b; }
            }
            if (y_min < 0) continue;
            if (nseg) {
                count += nseg[e_min];
            } else {   // boxesND.jl:26 is only reached when in_state_space(V[y_min]) held (statespaces.jl:155-157)
                bool inb = true;
                if (ss_lo) for (int k = 0; k < d; ++k) inb = inb && (ss_lo[k] <= X[(size_t)y_min * d + k]) && (X[(size_t)y_min * d + k] <= ss_hi[k]);
                if (inb) ++count;
            }
            if (getb(efree, e_min)) {                                         // fmt.jl:75
                A[x] = y_min + 1; C[x] = c_min;
                heap.push(x, c_min);
                Hnew.push_back(x);
                clrb(Wb, x);
            }
        }
        for (int64_t x : Hnew) setb(Hb, x);                                   // fmt.jl:83
        clrb(Hb, z);                                                          // fmt.jl:84
        if (!heap.empty()) z = heap.pop(); else break;                        // fmt.jl:85-89
    }
    // path back-trace, fmt.jl:92-101 (walks until sample 1)
    std::vector<int64_t> rev;
    int64_t cur = z;
    rev.push_back(cur + 1);
    while (cur != 0) {
        const int64_t p = A[cur];
        if (p == 0) break;
        cur = p - 1;
        rev.push_back(cur + 1);
    }
    for (size_t i = 0; i < rev.size(); ++i) path[i] = rev[rev.size() - 1 - i];
    res->status = mpfmt_is_goal_pt(&X[(size_t)z * d], gd, goal_kind, goal_params) ? 1 : 0;
    res->cost = C[z];
    res->z = z + 1;
    res->collision_checks = count;
    res->path_len = (int64_t)rev.size();
    res->nnz = colptr[N];
    res->ms_host_loop = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return MPFMT_OK;
}


// fmt.jl:43-101 over a DIRECTED cost graph (quasi-metric spaces: double integrator, Dubins car): forward sets = rows of the
// cost matrix (DSF = Dmat', linearquadratic.jl:73), backward sets = its columns (the CSC given).  efree / nseg are per CSC
// entry (row -> column motion free; segment tests the reference would have counted), F the checkpts bitmap (may be NULL).
void mpfmt_directed_fmt_recursion(int64_t N, const int64_t* colptr_, const int32_t* rowval_, const double* nzval_, const uint64_t* efree_,
                                  const uint8_t* nseg_, const uint64_t* F_, int64_t init_idx, const std::function<bool(int64_t)>& goal_hit,
                                  int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res, const mpfmt_csr_view* pre)
{
    const int64_t nnz = colptr_[N];
    struct view64 { const int64_t* p; int64_t operator[](int64_t i) const { return p[i]; } };
    struct view32 { const int32_t* p; int32_t operator[](int64_t i) const { return p[i]; } };
    struct viewd { const double* p; double operator[](int64_t i) const { return p[i]; } };
    struct view8 { const uint8_t* p; uint8_t operator[](int64_t i) const { return p[i]; } };
    const view64 colptr{colptr_}; const view32 rowval{rowval_}; const viewd nzval{nzval_}; const view8 nseg{nseg_};
    const bool checkpts = F_ != nullptr;
    auto bitp = [](const uint64_t* m, int64_t i) { return (m[i >> 6] >> (i & 63)) & 1ull; };
    // forward sets: CSR of the cost matrix (DSF = Dmat', linearquadratic.jl:73), rows ascending in target index -- taken from
    // the device transpose when the caller has one (mpfmt_csc_transpose_device), else built here
    std::vector<int64_t> rowptr_own, centry_own;
    std::vector<int32_t> colidx_own;
    const int64_t* rowptr;
    const int32_t* colidx;
    const uint32_t* centry32 = nullptr;
    if (pre) {
        rowptr = pre->rowptr; colidx = pre->colidx; centry32 = pre->centry;
    } else {
        rowptr_own.assign((size_t)N + 1, 0); colidx_own.resize((size_t)std::max<int64_t>(nnz, 1)); centry_own.resize((size_t)std::max<int64_t>(nnz, 1));
        std::vector<int64_t> cur((size_t)N);
        for (int64_t e = 0; e < nnz; ++e) rowptr_own[rowval[e] + 1]++;
        for (int64_t i = 0; i < N; ++i) rowptr_own[i + 1] += rowptr_own[i];
        for (int64_t i = 0; i < N; ++i) cur[i] = rowptr_own[i];
        for (int64_t j = 0; j < N; ++j)
            for (int64_t e = colptr[j]; e < colptr[j + 1]; ++e) { const int64_t a = cur[rowval[e]]++; colidx_own[a] = (int32_t)j; centry_own[a] = e; }
        rowptr = rowptr_own.data(); colidx = colidx_own.data();
    }
    auto centry_at = [&](int64_t a) -> int64_t { return centry32 ? (int64_t)centry32[a] : centry_own[a]; };
    // The recursion of fmt.jl:43-90.  DI neighbourhoods are large (hundreds of entries) and arcs are often blocked, so a
    // sample can be examined by many expanding neighbours; rescanning nearB(x) & H each time is what the reference does
    // and is O(N deg^2).  Here the argmin over the OPEN backward neighbours is maintained instead: when y opens it
    // relaxes best[x] of its forward neighbours still in W; when the best itself has closed, x is rescanned once.  The
    // order is the reference's (lowest cost, then lowest index = first minimum of its scan), so A, C, the path and the
    // collision count are unchanged.
    std::vector<uint8_t> Wm(N, 1), Hm(N, 0);
    std::vector<int64_t> Hnew;
    std::vector<int64_t> by(N, -1), be(N, -1);       // best open parent of x and its CSC entry (-1 none, -2 rescan)
    std::vector<double> bc(N, 0.0);
    for (int64_t i = 0; i < N; ++i) { A[i] = 0; C[i] = 0.0; }
    auto open_node = [&](int64_t y) {                 // y has just entered H: offer it to its forward neighbours
        Hm[y] = 1;
        const double cy = C[y];
        for (int64_t a = rowptr[y]; a < rowptr[y + 1]; ++a) {
            const int64_t x = colidx[a];
            if (!Wm[x] || by[x] == -2) continue;
            if (by[x] >= 0 && !Hm[by[x]]) { by[x] = -2; continue; }           // its best has closed: rescan when examined
            const int64_t e = centry_at(a);
            const double c = cy + nzval[e];
            if (by[x] < 0 || c < bc[x] || (c == bc[x] && y < by[x])) { by[x] = y; bc[x] = c; be[x] = e; }
        }
    };
    Heap heap;
    const int64_t i0 = init_idx - 1;
    Wm[i0] = 0;
    open_node(i0);
    heap.push(i0, 0.0);
    int64_t z = heap.pop();
    int64_t count = 0;
    while (!goal_hit(z)) {
        Hnew.clear();
        for (int64_t a = rowptr[z]; a < rowptr[z + 1]; ++a) {                  // nearF(V, z, r, W), fmt.jl:70
            const int64_t x = colidx[a];
            if (!Wm[x]) continue;
            if (checkpts && !bitp(F_, x)) continue;
            if (by[x] == -2 || (by[x] >= 0 && !Hm[by[x]])) {                   // nearB(V, x, r, H), fmt.jl:72-74
                int64_t y_min = -1, e_min = -1; double c_min = 0.0;
                for (int64_t b = colptr[x]; b < colptr[x + 1]; ++b) {
                    const int64_t y = rowval[b];
                    if (!Hm[y]) continue;
                    const double c = C[y] + nzval[b];
                    if (y_min < 0 || c < c_min) { y_min = y; c_min = c; e_min = b; }
                }
                by[x] = y_min; bc[x] = c_min; be[x] = e_min;
            }
            if (by[x] < 0) continue;
            const int64_t y_min = by[x], e_min = be[x];
            count += nseg[e_min];                                              // boxesND.jl:26 per tested segment
            if (bitp(efree_, e_min)) {
                A[x] = y_min + 1; C[x] = bc[x];
                heap.push(x, bc[x]);
                Hnew.push_back(x);
                Wm[x] = 0;
            }
        }
        Hm[z] = 0;                                                             // fmt.jl:84 (before 83: same final sets)
        for (int64_t x : Hnew) open_node(x);                                   // fmt.jl:83
        if (!heap.empty()) z = heap.pop(); else break;
    }
    std::vector<int64_t> rev;
    int64_t cu = z;
    rev.push_back(cu + 1);
    while (cu != 0) { const int64_t p = A[cu]; if (p == 0) break; cu = p - 1; rev.push_back(cu + 1); }
    for (size_t i = 0; i < rev.size(); ++i) path[i] = rev[rev.size() - 1 - i];
    res->status = goal_hit(z) ? 1 : 0;
    res->cost = C[z]; res->z = z + 1; res->collision_checks = count; res->path_len = (int64_t)rev.size(); res->nnz = nnz;
}


// ImmutableNNC(D, r) handed in from outside (mpfmt_graph_import; nearneighbors.jl:23-28): 1-based CSC, monotone colptr,
// rows in range, strictly ascending inside a column, no self loops.  Returns 0, or the 1-based column at fault (negative:
// -1 colptr[1] != 1, -2 colptr decreases) with a message in err.
int64_t mpfmt_validate_csc(int64_t N, const int64_t* colptr, const int64_t* rowval, char* err, size_t errlen)
{
    if (colptr[0] != 1) { snprintf(err, errlen, "colptr[1] must be 1 (1-based CSC)"); return -1; }
    for (int64_t j = 0; j < N; ++j)
        if (colptr[j + 1] < colptr[j]) { snprintf(err, errlen, "colptr decreases at column %lld", (long long)(j + 1)); return -2; }
    for (int64_t j = 0; j < N; ++j)
        for (int64_t e = colptr[j] - 1; e < colptr[j + 1] - 1; ++e) {
            const int64_t y = rowval[e];
            if (y < 1 || y > N || y == j + 1) { snprintf(err, errlen, "column %lld: row %lld out of range or a self loop", (long long)(j + 1), (long long)y); return j + 1; }
            if (e > colptr[j] - 1 && rowval[e - 1] >= y) { snprintf(err, errlen, "column %lld: rows are not strictly ascending", (long long)(j + 1)); return j + 1; }
        }
    return 0;
}
