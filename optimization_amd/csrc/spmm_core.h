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

#ifndef MI_SPMM_CHUNK
#define MI_SPMM_CHUNK 4  // measured on cfg2: 4 -> 33.4 us, 8 -> 34.1 us, entry-at-a-time -> 35 us
#endif
constexpr int kSlicesPerGroup = kWaves;  // one workgroup pass covers 16 slices = 1024 rows

// acc[0..P) = row `row` of A times V.  One lane per row; a wave owns one slice, so the loads of
// val/col at (base + k) * 64 + lane are contiguous across the wave (512 B + 256 B per k).
template <int P>
__device__ __forceinline__ void sell_row_times(const SellView &A, size_t slice, int lane,
                                               const double *__restrict__ V, double (&acc)[P]) {
  const long long b0 = A.slice_ptr[slice], b1 = A.slice_ptr[slice + 1];
#pragma unroll
  for (int c = 0; c < P; ++c) acc[c] = 0;
  // Chunks of MI_SPMM_CHUNK entries: all (value, column) pairs of a chunk are loaded first, then the
  // CH x P gathers are issued back to back (memory-level parallelism instead of a load -> gather
  // dependency per entry).  Entries beyond the slice width are predicated off by a select.
  // (Non-temporal loads of the matrix stream were measured SLOWER, 40 vs 33 us: the 87 MB matrix is
  // Infinity-Cache resident across iterations and `nt` forfeits that.)
  constexpr int CH = MI_SPMM_CHUNK;
  for (long long k = b0; k < b1; k += CH) {
    double a[CH];
    size_t cidx[CH];
    bool valid[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      valid[j] = k + j < b1;
      const size_t e = (size_t)(valid[j] ? k + j : k) * 64 + lane;
      a[j] = A.val[e];
      cidx[j] = (size_t)A.col[e];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const double *src = (cidx[j] < A.n) ? (V + cidx[j] * P) : (A.halo + (cidx[j] - A.n) * P);
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const double t = a[j] * src[c];
        acc[c] += valid[j] ? t : 0.0;  // select, not multiply-by-zero: never manufactures a NaN
      }
    }
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
