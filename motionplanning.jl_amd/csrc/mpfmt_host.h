// Declarations of the host-only part of libmpfmt.so (mpfmt_host.cpp): no HIP types, so the sanitizer build needs no ROCm.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <cstdio>
#include <functional>
#include <vector>
#include "../../include/mpfmt.h"

// forward sets of a directed cost graph (CSR view of the CSC held in the ctx): rowptr[N+1], colidx[nnz] (target of each
// entry, ascending inside a row), centry[nnz] (the CSC entry it came from)
struct mpfmt_csr_view { const int64_t* rowptr; const int32_t* colidx; const uint32_t* centry; };
struct mpfmt_csr_host { std::vector<int64_t> rowptr; std::vector<int32_t> colidx; std::vector<uint32_t> centry; };

bool mpfmt_is_goal_pt(const double* v, int d, int kind, const double* g);
int32_t mpfmt_host_fmt_recursion_impl(int64_t N, int32_t d, const double* X, const int64_t* colptr, const int32_t* rowval,
                                      const double* nzval, const uint64_t* efree, const uint64_t* F, const double* ss_lo,
                                      const double* ss_hi, int64_t init_idx, int32_t goal_kind, const double* goal_params, int32_t gd,
                                      const uint8_t* nseg, int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res);
void mpfmt_directed_fmt_recursion(int64_t N, const int64_t* colptr, const int32_t* rowval, const double* nzval, const uint64_t* efree,
                                  const uint8_t* nseg, const uint64_t* F, int64_t init_idx, const std::function<bool(int64_t)>& goal_hit,
                                  int64_t* A, double* C, int64_t* path, mpfmt_fmt_result* res, const mpfmt_csr_view* pre);
int64_t mpfmt_validate_csc(int64_t N, const int64_t* colptr, const int64_t* rowval, char* err, size_t errlen);
