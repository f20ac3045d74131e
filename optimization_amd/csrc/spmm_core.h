// spmm_core.h -- sliced-ELL-64 sparse x tall-skinny (n x P, row-major) product core, shared by the
// plain SpMM kernel (sparse.hip) and the fused Stiefel Hessian kernel (stiefel.hip).
#pragma once
#include "mi_internal.h"

namespace mi {

struct SellView {
  size_t n, nslices;
  const long long *__restrict__ slice_ptr;
  const int *__restrict__ col;
  const double *__restrict__ val;
  const double *__restrict__ halo;  // may be null
};

inline SellView sell_view(const mi_csr *A) {
  return SellView{A->n, A->nslices, A->slice_ptr, A->col, A->val, A->halo};
}

constexpr int kSlicesPerGroup = kWaves;  // one workgroup pass covers 16 slices = 1024 rows

// acc[0..P) = row `row` of A times V.  One lane per row; a wave owns one slice, so the loads of
// val/col at (base + k) * 64 + lane are contiguous across the wave (512 B + 256 B per k).
template <int P>
__device__ __forceinline__ void sell_row_times(const SellView &A, size_t slice, int lane,
                                               const double *__restrict__ V, double (&acc)[P]) {
  const long long b0 = A.slice_ptr[slice], b1 = A.slice_ptr[slice + 1];
#pragma unroll
  for (int c = 0; c < P; ++c) acc[c] = 0;
#pragma unroll 4
  for (long long k = b0; k < b1; ++k) {
    const size_t e = (size_t)k * 64 + lane;
    const double a = A.val[e];
    const size_t cidx = (size_t)A.col[e];
    const double *src = (cidx < A.n) ? (V + cidx * P) : (A.halo + (cidx - A.n) * P);
#pragma unroll
    for (int c = 0; c < P; ++c) acc[c] += a * src[c];
  }
}

// Workgroup -> contiguous range of slice groups, XCD-aware: the workgroups of one XCD cover one
// contiguous row range, so the gathers of V (rows i +- 1, +- nx, +- nx*ny of a stencil) stay inside
// that XCD's L2.
__device__ __forceinline__ void group_range(size_t ngroups, size_t &g0, size_t &g1) {
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  g0 = (ngroups * lb) / nb;
  g1 = (ngroups * (lb + 1)) / nb;
}

inline size_t sell_groups(const mi_csr *A) { return (A->nslices + kSlicesPerGroup - 1) / kSlicesPerGroup; }

}  // namespace mi
