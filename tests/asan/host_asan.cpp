// Sanitizer run of the host side of libmpfmt.so ("host: ASan/UBSan build of oracle + shim tests").
// Built by tests/test_host_cpu.py with g++ -fsanitize=address,undefined together with motionplanning.jl_amd/csrc/mpfmt_host.cpp
// (the same translation unit the library ships).  Reads a graph written by the test (the oracle's), runs the symmetric and the
// directed recursion and the CSC validation on good and malformed inputs, writes the trees back for comparison.
#include "../../motionplanning.jl_amd/csrc/mpfmt_host.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <class T> static std::vector<T> rd(FILE* f, size_t n)
{
    std::vector<T> v(n ? n : 1);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short input\n"); exit(4); }
    v.resize(n);
    return v;
}

int main(int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int64_t hdr[4];
    if (fread(hdr, 8, 4, f) != 4) return 4;
    const int64_t N = hdr[0], d = hdr[1], nnz = hdr[2], gk = hdr[3];
    auto X = rd<double>(f, N * d);
    auto colptr = rd<int64_t>(f, N + 1);
    auto rowval = rd<int32_t>(f, nnz);
    auto nzval = rd<double>(f, nnz);
    auto efree = rd<uint64_t>(f, (nnz + 63) / 64);
    auto F = rd<uint64_t>(f, (N + 63) / 64);
    auto lo = rd<double>(f, d); auto hi = rd<double>(f, d);
    auto goal = rd<double>(f, d + 1);
    fclose(f);
    std::vector<int64_t> A(N), path(N), A2(N), path2(N);
    std::vector<double> C(N), C2(N);
    mpfmt_fmt_result res, res2;
    if (mpfmt_host_fmt_recursion(N, (int32_t)d, X.data(), colptr.data(), rowval.data(), nzval.data(), efree.data(), F.data(), lo.data(), hi.data(), 1,
                                 (int32_t)gk, goal.data(), A.data(), C.data(), path.data(), &res) != 0) return 5;
    // the directed recursion on the same (symmetric) graph: forward sets built inside, one segment test per edge
    std::vector<uint8_t> nseg(nnz ? nnz : 1, 1);
    auto goal_hit = [&](int64_t z) { return mpfmt_is_goal_pt(&X[z * d], (int)d, (int)gk, goal.data()); };
    memset(&res2, 0, sizeof res2);
    mpfmt_directed_fmt_recursion(N, colptr.data(), rowval.data(), nzval.data(), efree.data(), nseg.data(), F.data(), 1, goal_hit, A2.data(), C2.data(),
                                 path2.data(), &res2, nullptr);
    // argument checks must reject, not read, bad input
    if (mpfmt_host_fmt_recursion(N, (int32_t)d, X.data(), colptr.data(), rowval.data(), nzval.data(), efree.data(), nullptr, lo.data(), nullptr, 1, (int32_t)gk,
                                 goal.data(), A2.data(), C2.data(), path2.data(), &res2) != MPFMT_ERR_ARG) return 6;
    if (mpfmt_host_fmt_recursion(N, (int32_t)d, X.data(), colptr.data(), rowval.data(), nzval.data(), efree.data(), nullptr, nullptr, nullptr, N + 1, (int32_t)gk,
                                 goal.data(), A2.data(), C2.data(), path2.data(), &res2) != MPFMT_ERR_ARG) return 6;
    // CSC validation: the good graph (1-based copy), then one defect at a time
    std::vector<int64_t> cp1(N + 1), rv1(nnz ? nnz : 1);
    for (int64_t j = 0; j <= N; ++j) cp1[j] = colptr[j] + 1;
    for (int64_t e = 0; e < nnz; ++e) rv1[e] = (int64_t)rowval[e] + 1;
    char err[160];
    int bad = 0;
    if (mpfmt_validate_csc(N, cp1.data(), rv1.data(), err, sizeof err) != 0) bad |= 1;
    if (nnz > 2) {
        auto t = rv1; t[nnz / 2] = N + 7;                       if (mpfmt_validate_csc(N, cp1.data(), t.data(), err, sizeof err) <= 0) bad |= 2;
        t = rv1; t[nnz / 2] = 0;                                 if (mpfmt_validate_csc(N, cp1.data(), t.data(), err, sizeof err) <= 0) bad |= 4;
        auto c2 = cp1; c2[0] = 0;                                if (mpfmt_validate_csc(N, c2.data(), rv1.data(), err, sizeof err) != -1) bad |= 8;
        c2 = cp1; c2[N / 2] = c2[N / 2 + 1] + 1;                 if (mpfmt_validate_csc(N, c2.data(), rv1.data(), err, sizeof err) != -2) bad |= 16;
        int64_t col = 0; while (col < N && cp1[col + 1] - cp1[col] < 2) ++col;
        if (col < N) { t = rv1; std::swap(t[cp1[col] - 1], t[cp1[col]]); if (mpfmt_validate_csc(N, cp1.data(), t.data(), err, sizeof err) != col + 1) bad |= 32; }
    }
    if (bad) { fprintf(stderr, "validate_csc: defect mask %d\n", bad); return 7; }
    FILE* o = fopen(argv[2], "wb");
    if (!o) return 2;
    fwrite(&res, sizeof res, 1, o); fwrite(A.data(), 8, N, o); fwrite(C.data(), 8, N, o); fwrite(path.data(), 8, res.path_len, o);
    fwrite(&res2, sizeof res2, 1, o);
    fclose(o);
    printf("host_asan ok: N=%lld nnz=%lld status=%d cost=%.6f checks=%lld\n", (long long)N, (long long)nnz, res.status, res.cost, (long long)res.collision_checks);
    return 0;
}
