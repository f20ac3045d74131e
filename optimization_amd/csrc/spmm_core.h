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
  const uint32_t *__restrict__ pk;  // value-indexed packed entries (mi_csr::pk), may be null
  const double *__restrict__ vtab;
};

inline SellView sell_view(const mi_csr *A) {
  return SellView{A->n, A->nslices, A->slice_ptr, A->col, A->val, A->halo_cur(), A->pk, A->vtab};
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
#ifdef MI_DEBUG_NO_GATHER  // timing experiment only: every entry reads the lane's own row
      const double *src = V + (slice * 64 + lane < A.n ? slice * 64 + lane : 0) * P + 0 * cidx[j];
#else
      const double *src = (cidx[j] < A.n) ? (V + cidx[j] * P) : (A.halo + (cidx[j] - A.n) * P);
#endif
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const double t = a[j] * src[c];
        acc[c] += valid[j] ? t : 0.0;  // select, not multiply-by-zero: never manufactures a NaN
      }
    }
  }
}

// Lean, software-pipelined form of the same product for a wave that owns the slices first, first + kWaves,
// ... (< end); used by the one-pass Stiefel Hessian.
//   * 32-bit byte offsets from scalar bases (global_load with saddr) instead of 64-bit multiply-adds per
//     gathered entry: the caller guarantees that V, the halo, the value and the column arrays each span
//     < 4 GiB (sell_stream_ok);
//   * predication once per entry (the VALUE is zeroed), then one multiply and one add per component;
//   * HALO is a template flag, so unsharded matrices carry no column-range select;
//   * the value/column operands of the NEXT chunk -- of this slice or of the wave's next slice -- are
//     requested right after the current chunk's gathers (unconditionally: behind a branch the compiler can no
//     longer count outstanding loads and drains the queue), and the row epilogue's operands one slice ahead:
//       epi.begin(slice)      issue the loads the epilogue of `slice` will need
//       epi.end(slice, acc)   consume acc[0..P) = (A V)(row slice*64+lane, :)
// `first` must be wave-uniform (readfirstlane): slice bounds then come from scalar loads.
// What this does NOT buy on cfg2 is time: with wall_clock64 stamps inside the kernel the
// main loop runs at ~6.4 TB/s, the practical HBM rate, in every variant tried -- ~3x fewer instructions (this
// form), deeper prefetch, chunk sizes 2..8, L1-bypassing matrix loads, an LDS copy of the workgroup's rows of
// V, a column-major V -- and the rest of the 33 us is the prologue reduction and the ramp-down.
// PK: the matrix is read from its value-indexed packed copy (one dword per entry: (col - row) << 8 | value
// index; `vt` = the 256-entry value table, staged in LDS by the caller) -- 4 instead of 12 bytes per entry.
template <int P, bool HALO, bool PK, class Epi>
__device__ __forceinline__ void sell_stream(const SellView &A, size_t first, size_t end, int lane,
                                            const double *__restrict__ V, const double *vt, Epi &epi) {
  constexpr int CH = MI_SPMM_CHUNK;
  if (first >= end) return;
  size_t slice = first;
  long long k = A.slice_ptr[slice], b1 = A.slice_ptr[slice + 1];
  const unsigned lane8 = (unsigned)lane * 8u, lane4 = (unsigned)lane * 4u;
  const unsigned nloc = (unsigned)A.n;
  // operands of a chunk as loaded: PK: packed words (decoded when the chunk becomes current); else value + column
  struct Ops {
    double a[PK ? 1 : CH];
    unsigned c[CH];
  };
  // entries beyond the slice width re-read the chunk's first entry -- or, for an empty slice, entry 0 of
  // the matrix (mi_csr always stores >= 64 entries); their value is zeroed when the chunk is consumed
  auto load_chunk = [&](Ops &o, long long kk, long long bb) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const long long kj = (kk + j < bb) ? kk + j : ((kk < bb) ? kk : 0);
      if (PK) {
        const char *pb = reinterpret_cast<const char *>(A.pk + (size_t)kj * 64);  // scalar base
        o.c[j] = *reinterpret_cast<const unsigned *>(pb + lane4);
      } else {
        const char *vb = reinterpret_cast<const char *>(A.val + (size_t)kj * 64);
        const char *cb = reinterpret_cast<const char *>(A.col + (size_t)kj * 64);
        o.a[j] = *reinterpret_cast<const double *>(vb + lane8);
        o.c[j] = *reinterpret_cast<const unsigned *>(cb + lane4);
      }
    }
  };
  // bounds of the wave's NEXT slice, requested a whole slice ahead: a scalar load issued where its result is
  // needed costs a full memory latency per slice while the vector queue is saturated (measured: a young
  // wave waited 4-6 us for its first pair)
  size_t cand = slice + kWaves;
  long long q0 = k, q1 = b1;
  if (cand < end) { q0 = A.slice_ptr[cand]; q1 = A.slice_ptr[cand + 1]; }
  Ops cur;
  load_chunk(cur, k, b1);
  epi.begin(slice);
  double acc[P];
#pragma unroll
  for (int c = 0; c < P; ++c) acc[c] = 0;
  for (;;) {
    // the wave's next chunk: all selects on scalars, no branches
    const bool row_done = k + CH >= b1;
    const bool more_slices = cand < end;
    const bool have_next = !row_done || more_slices;
    const size_t nslice = (row_done && more_slices) ? cand : slice;
    const long long nk = row_done ? (more_slices ? q0 : k) : k + CH;
    const long long nb1 = (row_done && more_slices) ? q1 : b1;
    double a[CH];
    unsigned ci[CH];
    if (PK) {
      const unsigned row = (unsigned)(slice * 64) + (unsigned)lane;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        a[j] = vt[cur.c[j] & 255u];
        ci[j] = row + (unsigned)((int)cur.c[j] >> 8);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) { a[j] = cur.a[j]; ci[j] = cur.c[j]; }
    }
    double g[CH][P];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const char *base = reinterpret_cast<const char *>(V);
      unsigned boff = ci[j] * (unsigned)(P * 8);
      if (HALO && ci[j] >= nloc) {
        base = reinterpret_cast<const char *>(A.halo);
        boff = (ci[j] - nloc) * (unsigned)(P * 8);
      }
      const double *src = reinterpret_cast<const double *>(base + boff);
#pragma unroll
      for (int c = 0; c < P; ++c) g[j][c] = src[c];
    }
    Ops nxt;
    load_chunk(nxt, nk, nb1);  // (re-reads the first chunk of the last slice when nothing follows)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const double aj = (k + j < b1) ? a[j] : 0.0;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        // product and sum rounded separately, entry by entry in storage order: bit-identical to a plain CSR
        // loop on the host (the checks of the sharded products rely on that)
#pragma clang fp contract(off)
        const double t = aj * g[j][c];
        acc[c] = acc[c] + t;
      }
    }
    if (row_done) {
      epi.end(slice, acc);
#pragma unroll
      for (int c = 0; c < P; ++c) acc[c] = 0;
      if (have_next) epi.begin(nslice);
    }
    if (!have_next) break;
    if (row_done) {  // moved on to `cand`: request the bounds of the slice after it
      cand += kWaves;
      if (cand < end) { q0 = A.slice_ptr[cand]; q1 = A.slice_ptr[cand + 1]; }
    }
    slice = nslice; k = nk; b1 = nb1;
    cur = nxt;
  }
}

// sell_stream's 32-bit offsets: every array it indexes must span < 4 GiB
inline bool sell_stream_ok(const mi_csr *A, int p) {
  const size_t lim = (size_t)1 << 32;
  return (A->n + A->halo_lo + A->halo_hi + 64) * (size_t)p * 8 < lim && A->padded * 8 < lim;
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
