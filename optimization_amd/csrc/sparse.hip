// sparse.hip -- sparse SPD operator storage (CSR in, sliced-ELL-64 in HBM) and W = A V for
// tall-skinny V (n x p row-major, p <= 8).  This is the user-side HVP building block: the reference
// has no sparse code (its HVP is a user callable invoked at IterativeSolvers.h:294, TNT.h:512).
//
// Algorithmic bytes: 12*nnz + 4*(n+1) + 16*n*p.
#include "spmm_core.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include <algorithm>

using namespace mi;

namespace {

template <int P>
__global__ __launch_bounds__(kBlock) void k_spmm(SellView A, const double *__restrict__ V,
                                                 double *__restrict__ W) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t ngroups = (A.nslices + kSlicesPerGroup - 1) / kSlicesPerGroup;
  size_t g0, g1;
  group_range(ngroups, g0, g1);
  for (size_t g = g0; g < g1; ++g) {
    const size_t slice = g * kSlicesPerGroup + w;
    if (slice >= A.nslices) continue;
    const size_t row = slice * 64 + lane;
    double acc[P];
    sell_row_times<P>(A, slice, lane, V, acc);
    if (row < A.n) {
#pragma unroll
      for (int c = 0; c < P; ++c) W[row * P + c] = acc[c];
    }
  }
}

// The same product on the lean pipelined core (spmm_core.h sell_stream): 32-bit offsets, one workgroup per CU,
// the matrix from its value-indexed packed copy when it has one.  Same per-row arithmetic as k_spmm (bit-identical
// results); used whenever the fields span < 4 GiB.
template <int P, bool HALO, bool PK>
__global__ __launch_bounds__(kBlock) void k_spmm_stream(SellView A, const double *__restrict__ V,
                                                        double *__restrict__ W) {
  __shared__ double vt[PK ? 256 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PK) {
    if (threadIdx.x < 256) vt[threadIdx.x] = A.vtab[threadIdx.x];
    __syncthreads();
  }
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = (A.nslices * lb) / nb, s1 = (A.nslices * (lb + 1)) / nb;
  struct Epi {
    const SellView &A;
    double *__restrict__ W;
    int lane;
    __device__ __forceinline__ void begin(size_t) {}
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P]) {
      if (slice * 64 + lane >= A.n) return;
      double *ws = reinterpret_cast<double *>(reinterpret_cast<char *>(W + slice * 64 * P) +
                                              (unsigned)lane * (unsigned)(P * 8));
#pragma unroll
      for (int c = 0; c < P; ++c) ws[c] = acc[c];
    }
  } epi{A, W, lane};
  sell_stream<P, HALO, PK>(A, s0 + (size_t)__builtin_amdgcn_readfirstlane(w), s1, lane, V, vt, epi);
}

// W = A V with the three curvature partials of STPCG, <V,W>, <W,W>, <V,V> (IterativeSolvers.h:300,305-306), in the
// product's own pass: a plain CSR Hessian then needs no separate dot kernel (2N doubles re-read) per iteration
template <int P, bool HALO, bool PK>
__global__ __launch_bounds__(kBlock) void k_spmm_dots_stream(SellView A, const CgState *__restrict__ st,
                                                             const double *__restrict__ V, double *__restrict__ W,
                                                             double *__restrict__ partials) {
  __shared__ double lds[3 * kWaves];
  __shared__ double vt[PK ? 256 : 1];
  if (st && st->mode != CG_RUN) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PK) {
    if (threadIdx.x < 256) vt[threadIdx.x] = A.vtab[threadIdx.x];
    __syncthreads();
  }
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = (A.nslices * lb) / nb, s1 = (A.nslices * (lb + 1)) / nb;
  double a[3] = {0, 0, 0};
  struct Epi {
    const SellView &A;
    const double *__restrict__ V;
    double *__restrict__ W;
    double (&a)[3];
    int lane;
    double v[P];
    __device__ __forceinline__ unsigned lane_off(size_t slice) const {
      return (slice * 64 + lane < A.n) ? (unsigned)lane * (unsigned)(P * 8) : 0u;
    }
    __device__ __forceinline__ void begin(size_t slice) {
      const double *vs = reinterpret_cast<const double *>(reinterpret_cast<const char *>(V + slice * 64 * P) +
                                                          lane_off(slice));
#pragma unroll
      for (int c = 0; c < P; ++c) v[c] = vs[c];
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P]) {
      if (slice * 64 + lane >= A.n) return;
      double *ws = reinterpret_cast<double *>(reinterpret_cast<char *>(W + slice * 64 * P) + lane_off(slice));
#pragma unroll
      for (int c = 0; c < P; ++c) {
        ws[c] = acc[c];
        a[0] += v[c] * acc[c]; a[1] += acc[c] * acc[c]; a[2] += v[c] * v[c];
      }
    }
  } epi{A, V, W, a, lane, {}};
  sell_stream<P, HALO, PK>(A, s0 + (size_t)__builtin_amdgcn_readfirstlane(w), s1, lane, V, vt, epi);
  block_partials_store<3>(a, lds, partials);
}

// W = A V - s W (p = 1) and this workgroup's partial of |W|^2: LSQR's `u = A v - alpha u` / `v = A'u - beta v`
// with the norm for the following normalisation, in the SpMV's own pass
__global__ __launch_bounds__(kBlock) void k_spmv_sub_scaled(SellView A, const double *__restrict__ V,
                                                            const double *__restrict__ scale,
                                                            const int *__restrict__ mode,
                                                            const int *__restrict__ gate, double *__restrict__ W,
                                                            double *__restrict__ partials) {
  __shared__ double lds[kWaves + 1];
  if (*mode != 0 || (gate && !*gate)) return;
  const double sc = *scale;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t ngroups = (A.nslices + kSlicesPerGroup - 1) / kSlicesPerGroup;
  size_t g0, g1;
  group_range(ngroups, g0, g1);
  double a[1] = {0};
  for (size_t g = g0; g < g1; ++g) {
    const size_t slice = g * kSlicesPerGroup + w;
    if (slice >= A.nslices) continue;
    const size_t row = slice * 64 + lane;
    double acc[1];
    sell_row_times<1>(A, slice, lane, V, acc);
    if (row < A.n) {
      const double out = acc[0] - sc * W[row];
      W[row] = out;
      a[0] += out * out;
    }
  }
  block_partials_store<1>(a, lds, partials);
}

int upload(void **dst, const void *src, size_t bytes) {
  MI_HIP(hipMalloc(dst, bytes ? bytes : 8));
  if (bytes) MI_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return MI_OK;
}

// The same on the lean pipelined core (packed matrix when there is one)
template <bool HALO, bool PK>
__global__ __launch_bounds__(kBlock) void k_spmv_sub_scaled_stream(SellView A, const double *__restrict__ V,
                                                                   const double *__restrict__ scale,
                                                                   const int *__restrict__ mode,
                                                                   const int *__restrict__ gate,
                                                                   double *__restrict__ W,
                                                                   double *__restrict__ partials) {
  __shared__ double lds[kWaves + 1];
  __shared__ double vt[PK ? 256 : 1];
  if (*mode != 0 || (gate && !*gate)) return;
  const double sc = *scale;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PK) {
    if (threadIdx.x < 256) vt[threadIdx.x] = A.vtab[threadIdx.x];
    __syncthreads();
  }
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = (A.nslices * lb) / nb, s1 = (A.nslices * (lb + 1)) / nb;
  double a[1] = {0};
  struct Epi {
    const SellView &A;
    double *__restrict__ W;
    double sc;
    double (&a)[1];
    int lane;
    double wold;
    __device__ __forceinline__ void begin(size_t slice) {
      const size_t row = slice * 64 + lane;
      wold = W[row < A.n ? row : slice * 64];
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[1]) {
      const size_t row = slice * 64 + lane;
      if (row >= A.n) return;
      const double out = acc[0] - sc * wold;
      W[row] = out;
      a[0] += out * out;
    }
  } epi{A, W, sc, a, lane, 0.0};
  sell_stream<1, HALO, PK>(A, s0 + (size_t)__builtin_amdgcn_readfirstlane(w), s1, lane, V, vt, epi);
  block_partials_store<1>(a, lds, partials);
}

// Window form of the matrix for the LDS-ring kernels (spmm_core.h sell_window; mi_csr::wk, wfar).
// The window half-width wc (in 64-row chunks) is the smallest of {1, 2, 4} that already serves, from the ring,
// 90 % of the entries the widest would -- provided that is at least one off-diagonal entry per row on average (a
// matrix without such a band, e.g. a random graph, keeps the plain gather kernels), every slice is at most kWinHead
// entries wide, every row has at most kFarCap entries outside the window, and 0.0 has (or can get) a place in the
// value table.
int build_window(mi_csr *A, size_t n, size_t nnz, const int32_t *rowptr, const int32_t *col,
                 const std::vector<long long> &sp, const std::vector<uint32_t> &pk, std::vector<double> &table,
                 int ntable, size_t halo_lo, size_t halo_hi) {
  if (n == 0 || nnz == 0) return MI_OK;
  const int ncand = 3, cand[ncand] = {1, 2, 4};
  static_assert(kMaxWinChunks <= 4, "candidates");
  int wc = 0;
  {
    size_t near[ncand] = {0, 0, 0};
    for (size_t r = 0; r < n; ++r)
      for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        if ((size_t)col[k] >= n) continue;  // halo column: never in the ring
        const long long d = (long long)col[k] - (long long)r;
        const unsigned long long ad = (unsigned long long)(d < 0 ? -d : d);
        for (int i = 0; i < ncand; ++i)
          if (cand[i] <= kMaxWinChunks && ad <= 64ull * cand[i]) ++near[i];
      }
    int last = 0;
    for (int i = 0; i < ncand; ++i)
      if (cand[i] <= kMaxWinChunks) last = i;
    if (near[last] < n + n / 2 + 1) return MI_OK;
    wc = cand[last];
    for (int i = 0; i <= last; ++i)
      if (near[i] * 10 >= near[last] * 9) { wc = cand[i]; break; }
  }
  const size_t nslices = A->nslices;
  int head = 0;
  for (size_t s_ = 0; s_ < nslices; ++s_) head = std::max<int>(head, (int)(sp[s_ + 1] - sp[s_]));
  if (head > kWinHead) return MI_OK;
  // index of 0.0 (bit pattern +0.0) in the value table
  int zidx = -1;
  for (int i = 0; i < ntable; ++i) {
    uint64_t bits;
    memcpy(&bits, &table[i], sizeof bits);
    if (bits == 0) { zidx = i; break; }
  }
  if (zidx < 0) {
    if (ntable >= 256) return MI_OK;
    zidx = ntable;  // (the table is zero-filled up to 256 entries)
  }
  const int nc = 2 * kWinWaves + 2 * wc;
  const uint32_t zrow = (uint32_t)nc * 64u;
  const uint32_t zw = (zrow << 8) | (uint32_t)zidx;
  const size_t stored = pk.size();
  std::vector<uint32_t> wk(stored + (size_t)kWinHead * 64, zw);
  std::vector<int32_t> wfar((nslices + 1) * kFarCap * 64, 0);
  std::unordered_map<unsigned long long, size_t> far_strides;  // |column - row| of the far entries
  size_t far_total = 0;
  // "Pure" far structure: EVERY entry outside the window lies exactly D rows above or below its row (the two plane
  // neighbours of a 3-D stencil).  The far slots are then assigned by direction -- slot 0 = row + D, slot 1 = row - D
  // -- and the kernels compute the far columns instead of loading them (mi_csr::win_far_pure; 8 bytes per row less to
  // read).  A HALO column counts as the row it stands for in the neighbour's slab: column n + h is row h - halo_lo
  // (h < halo_lo: the LAST rows of rank - 1, i.e. negative local rows) or row n + (h - halo_lo) (the first rows of
  // rank + 1), so a z-slab of a stencil is pure too and its halo columns are computed as well (spmm_core.h load_far).
  auto virtual_row = [&](long long c) -> long long {
    if ((size_t)c < n) return c;
    const long long h = c - (long long)n;
    return h < (long long)halo_lo ? h - (long long)halo_lo : (long long)n + (h - (long long)halo_lo);
  };
  size_t pure_D = 0;
  bool pure = true;
  for (size_t r = 0; r < n && pure; ++r)
    for (int32_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
      const long long c = col[k], d = virtual_row(c) - (long long)r;
      const unsigned long long ad = (unsigned long long)(d < 0 ? -d : d);
      if ((size_t)c < n && ad <= 64ull * wc) continue;  // in the window
      if (pure_D == 0) pure_D = (size_t)ad;
      if ((size_t)ad != pure_D) { pure = false; break; }
      // (a computed halo column must exist: row - D >= -halo_lo, row + D < n + halo_hi)
      if (d < 0 ? (long long)r - (long long)pure_D < -(long long)halo_lo : r + pure_D >= n + halo_hi) { pure = false; break; }
    }
  if (pure_D == 0) pure = false;
  for (size_t sl = 0; sl < nslices; ++sl) {
    for (int lane = 0; lane < 64; ++lane) {
      const size_t r = sl * 64 + lane;
      const int32_t self = (int32_t)std::min(r, n - 1);
      for (int f = 0; f < kFarCap; ++f) wfar[(sl * kFarCap + f) * 64 + lane] = self;
      if (r >= n) continue;  // (its words stay zw)
      const int len = rowptr[r + 1] - rowptr[r];
      int nfar = 0;
      for (int k = 0; k < len; ++k) {
        const size_t e = (size_t)(sp[sl] + k) * 64 + lane;
        const uint32_t vi = pk[e] & 255u;
        const long long c = col[rowptr[r] + k];
        const long long d = c - (long long)r;
        const bool nearj = (size_t)c < n && (unsigned long long)(d < 0 ? -d : d) <= 64ull * wc;
        uint32_t rowidx;
        if (nearj) {
          rowidx = (uint32_t)(((size_t)c >> 6) % (size_t)nc) * 64u + (uint32_t)(c & 63);
        } else {
          if (nfar == kFarCap) return MI_OK;  // a row with a third far entry: not eligible
          const int slot = pure ? (virtual_row(c) > (long long)r ? 0 : 1) : nfar;
          rowidx = zrow + 1u + (uint32_t)(sl % kWinWaves) * (uint32_t)(kFarCap * 64) + (uint32_t)slot * 64u +
                   (uint32_t)lane;
          wfar[(sl * kFarCap + slot) * 64 + lane] = (int32_t)c;
          ++nfar;
          if ((size_t)c < n) {
            ++far_strides[(unsigned long long)(d < 0 ? -d : d)];
            ++far_total;
          }
        }
        wk[e] = (rowidx << 8) | vi;
      }
      // padding entries inside the slice (k >= len): zw already
    }
  }
  MI_TRY(upload((void **)&A->wk, wk.data(), wk.size() * sizeof(uint32_t)));
  MI_TRY(upload((void **)&A->wfar, wfar.data(), wfar.size() * sizeof(int32_t)));
  {  // the 16-bit form, when every value index (0.0 included) fits 5 bits and every LDS row 11 bits
    const uint32_t max_row = zrow + 1u + (uint32_t)(kWinWaves * kFarCap * 64);
    const int nvals = std::max(ntable, zidx + 1);
    if (nvals <= 32 && max_row < 2048u && head <= 8) {
      std::vector<uint32_t> w16((nslices + 1) * 4 * 64, 0u);
      const uint32_t z16 = (zrow << 5) | (uint32_t)zidx;
      for (size_t sl = 0; sl <= nslices; ++sl) {
        const int width = sl < nslices ? (int)(sp[sl + 1] - sp[sl]) : 0;
        for (int lane = 0; lane < 64; ++lane)
          for (int q = 0; q < 4; ++q) {
            uint32_t e[2];
            for (int h = 0; h < 2; ++h) {
              const int j = 2 * q + h;
              const uint32_t w32 = j < width ? wk[(size_t)(sp[sl] + j) * 64 + lane] : zw;
              e[h] = ((w32 >> 8) << 5) | (w32 & 255u);
              (void)z16;
            }
            w16[(sl * 4 + q) * 64 + lane] = e[0] | (e[1] << 16);
          }
      }
      MI_TRY(upload((void **)&A->wk16, w16.data(), w16.size() * sizeof(uint32_t)));
    }
  }
  A->win_chunks = wc;
  A->win_head = head;
  A->win_zero = zw;
  A->win_far_pure = pure ? pure_D : 0;
  // one stride carrying most of the far entries (the plane stride of a 3-D stencil): the workgroup ranges are then
  // cut so that this stride is a whole number of ranges (stiefel.hip window_bounds)
  A->win_far_stride = 0;
  for (const auto &kv : far_strides)
    if (kv.second * 10 >= far_total * 8) A->win_far_stride = (size_t)kv.first;
  return MI_OK;
}

// Build the sliced-ELL image on the host from CSR with LOCAL column indices.
// halo_lo: of the ncols - n halo columns behind the local ones, how many belong to rank - 1 (they come first)
int build_sell(mi_ctx *ctx, size_t n, size_t ncols, size_t nnz, const int32_t *rowptr,
               const int32_t *col, const double *val, mi_csr **out, size_t halo_lo = 0) {
  const size_t nslices = (n + 63) / 64;
  std::vector<long long> sp(nslices + 1, 0);
  for (size_t s = 0; s < nslices; ++s) {
    int w = 0;
    for (size_t r = s * 64; r < std::min(n, (s + 1) * 64); ++r) w = std::max(w, rowptr[r + 1] - rowptr[r]);
    sp[s + 1] = sp[s] + w;
  }
  const size_t padded = (size_t)sp[nslices] * 64;
  // at least one 64-entry chunk is always stored (zeros, column 0): the pipelined kernels park the loads of
  // predicated-off entries of EMPTY slices there (spmm_core.h sell_stream)
  const size_t stored = std::max<size_t>(padded, 64);
  std::vector<int> pcol(stored, 0);
  std::vector<double> pval(stored, 0.0);
  for (size_t s = 0; s < nslices; ++s) {
    const long long w = sp[s + 1] - sp[s];
    for (int lane = 0; lane < 64; ++lane) {
      const size_t r = s * 64 + lane;
      const int len = (r < n) ? rowptr[r + 1] - rowptr[r] : 0;
      const int self = (int)std::min(r, n ? n - 1 : 0);
      for (long long k = 0; k < w; ++k) {
        const size_t e = (size_t)(sp[s] + k) * 64 + lane;
        if (k < len) {
          pcol[e] = col[rowptr[r] + k];
          pval[e] = val[rowptr[r] + k];
        } else {
          pcol[e] = self;
        }
      }
    }
  }
  mi_csr *A = new mi_csr();
  A->ctx = ctx;
  A->n = n;
  A->ncols = ncols;
  A->nnz = nnz;
  A->padded = padded;
  A->nslices = nslices;
  MI_TRY(upload((void **)&A->slice_ptr, sp.data(), sp.size() * sizeof(long long)));
  MI_TRY(upload((void **)&A->col, pcol.data(), stored * sizeof(int)));
  MI_TRY(upload((void **)&A->val, pval.data(), stored * sizeof(double)));
  // value-indexed packed copy (mi_csr::pk): distinct stored values by BIT PATTERN (so -0.0, NaN payloads
  // and denormals survive), column as a signed 24-bit offset from the row
  // (the context's switch as it stands at THIS creation: mi_ctx_set_option("NO_PACKED") between two creations builds
  // the same matrix both ways in one process)
  const bool no_pack = A->ctx->cfg.no_packed;
  if (!no_pack) {
    std::unordered_map<uint64_t, int> index;
    std::vector<double> table;
    std::vector<uint32_t> pk(stored, 0u);
    bool ok = true;
    for (size_t e = 0; e < stored && ok; ++e) {
      uint64_t bits;
      memcpy(&bits, &pval[e], sizeof bits);
      auto it = index.find(bits);
      int vi;
      if (it == index.end()) {
        if (table.size() == 256) { ok = false; break; }
        vi = (int)table.size();
        index.emplace(bits, vi);
        table.push_back(pval[e]);
      } else {
        vi = it->second;
      }
      pk[e] = (uint32_t)vi;  // column part filled below, slice by slice
    }
    if (ok) {
      for (size_t sl = 0; sl < nslices && ok; ++sl)
        for (long long k = sp[sl]; k < sp[sl + 1]; ++k)
          for (int lane = 0; lane < 64; ++lane) {
            const size_t e = (size_t)k * 64 + lane;
            const long long delta = (long long)pcol[e] - (long long)(sl * 64 + lane);
            if (delta < -(1LL << 23) || delta >= (1LL << 23)) ok = false;  // 24-bit signed column offset
            pk[e] |= (uint32_t)((int32_t)delta) << 8;
          }
    }
    if (ok) {
      const int ntable = (int)table.size();
      table.resize(256, 0.0);
      MI_TRY(upload((void **)&A->pk, pk.data(), pk.size() * sizeof(uint32_t)));
      MI_TRY(upload((void **)&A->vtab, table.data(), 256 * sizeof(double)));
      A->nvtab = (int)index.size();
      // the window form of the same matrix, when it qualifies (local columns only: col, not pcol)
      MI_TRY(build_window(A, n, nnz, rowptr, col, sp, pk, table, ntable, halo_lo, ncols - n - halo_lo));
    }
  }
  *out = A;
  return MI_OK;
}

}  // namespace

namespace mi {

// Workgroup runs of the window kernels: how many workgroups, and which tiles each takes.
//
// Cost of a plan with nb workgroups whose longest run has L tiles, on a chip of C CUs: the busiest CU works through
// ceil(nb / C) runs -- a CU's tile rate is the same with 8 or 16 resident waves -- and every run costs about half a
// tile on top (ring fill, the 2 wc chunks of V re-read around the run):  ceil(nb / C) * (L + 1/2).  Measured on
// cfg2 (3907 tiles): 977 runs of 4 -> 28.7 us (model 18), 500 of 8 -> 27.6 us (17), 600 of 7 -> 30.9 us (22.5),
// 400 of 10 -> 31.5 us (21).  Plans within 2 % of the best: the one with the most workgroups (St(8e6,3), memory
// latency beyond the Infinity Cache: 1000 runs 253 us, 500 runs 261 us).
//
// When most far entries of the matrix share one stride D (rows; the plane stride of a 3-D stencil) the runs are
// D / m rows long (m integer), rounded to whole tiles: a far row is some other workgroup's own row, and with runs
// that divide D both touch it at the same moment of their runs, i.e. of the kernel -- one fetch into the XCD's L2
// instead of two (138 -> 133 MB read per launch on cfg2, exact request-size counters).
struct WinPlan {
  int nb = 0;
  double run = 0;  // tiles per run when cut by the far stride, else 0: equal runs of `per`
  int per = 0;
};
WinPlan window_plan(int ntiles, int max_wgs, int num_cu, size_t far_stride) {
  const double tile_rows = 64.0 * kWinWaves;
  const int min_wgs = std::min(max_wgs, num_cu + num_cu / 2);  // at least 6 waves on most CUs
  struct Cand { WinPlan p; double cost; };
  std::vector<Cand> cands;
  auto add = [&](WinPlan p, int longest) {
    if (p.nb < 1 || p.nb > max_wgs) return;
    if (p.nb < min_wgs && p.nb < ntiles) return;
    cands.push_back({p, std::ceil(p.nb / (double)num_cu) * (longest + .5)});
  };
  if (far_stride)
    for (int m = 1; far_stride / (double)m >= tile_rows; ++m) {
      WinPlan p;
      p.run = far_stride / (double)m / tile_rows;
      p.nb = (int)std::ceil(ntiles / p.run - 1e-9);
      add(p, (int)std::ceil(p.run - 1e-9));
    }
  if (cands.empty())
    for (int per = 1; per <= ntiles; ++per) {
      WinPlan p;
      p.per = per;
      p.nb = (ntiles + per - 1) / per;
      add(p, per);
    }
  WinPlan best;
  double best_cost = 0;
  for (const Cand &c : cands)
    if (!best.nb || c.cost < best_cost) { best = c.p; best_cost = c.cost; }
  for (const Cand &c : cands)
    if (c.cost <= 1.02 * best_cost && c.p.nb > best.nb) best = c.p;
  if (!best.nb) {  // nothing between the occupancy wish and the budget (e.g. budget < ntiles < 2 min_wgs): equal runs
    best.run = 0;
    best.per = std::max(1, (ntiles + std::max(1, max_wgs) - 1) / std::max(1, max_wgs));
    best.nb = (ntiles + best.per - 1) / best.per;
  }
  return best;
}

// first tile of every run (+ the end): strictly increasing from 0 to ntiles, at most max_wgs runs
std::vector<int> window_runs(int ntiles, int max_wgs, int num_cu, size_t far_stride) {
  std::vector<int> b;
  if (ntiles <= 0 || max_wgs <= 0 || num_cu <= 0) {
    b.push_back(0);
    b.push_back(std::max(0, ntiles));
    return b;
  }
  const WinPlan plan = window_plan(ntiles, max_wgs, num_cu, far_stride);
  if (plan.run >= 1.0) {
    for (int k = 0;; ++k) {
      const int t = std::min(ntiles, (int)std::llround(k * plan.run));
      b.push_back(t);
      if (t >= ntiles) break;
    }
  }
  if (b.empty() || (int)b.size() - 1 > max_wgs) {  // equal runs (also when rounding produced one run too many)
    const int per = plan.run >= 1.0 || plan.per < 1 ? std::max(1, (ntiles + max_wgs - 1) / max_wgs) : plan.per;
    b.clear();
    for (int t = 0; t < ntiles; t += per) b.push_back(t);
    b.push_back(ntiles);
  }
  return b;
}

// the plan of a matrix for a workgroup budget, cached on the matrix: grid and (if cut by the far stride) the
// workgroup -> first tile table on the device
int window_bounds(mi_ctx *ctx, const mi_csr *A, int wgs, int ntiles, int *grid, const int **bounds_out) {
  *bounds_out = nullptr;
  const bool off = ctx->cfg.no_win_bounds;
  const int key = off ? -wgs : wgs;
  auto it = A->win_plans.find(key);
  if (it == A->win_plans.end()) {
    const std::vector<int> b = window_runs(ntiles, wgs, ctx->num_cu, off ? 0 : A->win_far_stride);
    mi_csr::WinPlan plan;
    plan.n = (int)b.size() - 1;
    MI_HIP(hipMalloc((void **)&plan.bounds, b.size() * sizeof(int)));
    const hipError_t e = hipMemcpy(plan.bounds, b.data(), b.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(plan.bounds);
      MI_HIP(e);
    }
    it = A->win_plans.emplace(key, plan).first;  // (only a complete plan is ever recorded)
  }
  *grid = it->second.n;
  *bounds_out = it->second.bounds;
  return MI_OK;
}
int csr_spmm_launch(const mi_csr *A, int p, const double *V, double *W) {
  mi_ctx *ctx = A->ctx;
  if (A->n == 0) return MI_OK;
  const size_t ngroups = sell_groups(A);
  const int grid = (int)std::min<size_t>(ngroups, kMaxGrid);
  SellView view = sell_view(A);
  KScope ks(ctx, MI_K_SPMM);
  const bool no_stream = ctx->cfg.no_spmm_stream;
  if (!no_stream && p >= 1 && p <= kMaxP && sell_stream_ok(A, p)) {
    const int sgrid = (int)std::min<size_t>(ngroups, 256);  // one workgroup per CU, one round
#define SS(PV, HL, PKV) \
  hipLaunchKernelGGL((k_spmm_stream<PV, HL, PKV>), dim3(sgrid), dim3(kBlock), 0, ctx->stream, view, V, W)
#define SSP(PV)                                                 \
  if (A->halo) { if (A->pk) SS(PV, true, true); else SS(PV, true, false); } \
  else { if (A->pk) SS(PV, false, true); else SS(PV, false, false); }
    switch (p) {
      case 1: SSP(1); break;
      case 2: SSP(2); break;
      case 3: SSP(3); break;
      case 4: SSP(4); break;
      case 5: SSP(5); break;
      case 6: SSP(6); break;
      case 7: SSP(7); break;
      default: SSP(8); break;
    }
#undef SSP
#undef SS
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  switch (p) {
    case 1: hipLaunchKernelGGL(k_spmm<1>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 2: hipLaunchKernelGGL(k_spmm<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 3: hipLaunchKernelGGL(k_spmm<3>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 4: hipLaunchKernelGGL(k_spmm<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 5: hipLaunchKernelGGL(k_spmm<5>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 6: hipLaunchKernelGGL(k_spmm<6>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 7: hipLaunchKernelGGL(k_spmm<7>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    case 8: hipLaunchKernelGGL(k_spmm<8>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, V, W); break;
    default: set_error("p must be in [1,%d], got %d", kMaxP, p); return MI_ERR_INVALID_ARGUMENT;
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// W = A V + curvature partials into ctx->partials (mi_op::apply_dots of the CSR operator); MI_ERR_INTERNAL-free
// fallback: returns 1 in *unsupported when the pipelined core cannot address the fields (the caller then uses
// the product + k_cg_dot3)
int csr_spmm_dots(const mi_csr *A, int p, const mi_vec *V, mi_vec *W, int *nparts, bool *unsupported) {
  mi_ctx *ctx = A->ctx;
  *unsupported = !(p >= 1 && p <= kMaxP && sell_stream_ok(A, p)) || A->n == 0;
  if (*unsupported) return MI_OK;
  MI_TRY(comm_halo_exchange(ctx, A, p, V->d));
  int grid = uniform_grid(ctx, sell_groups(A));
  if (!ctx->uniform_grid && grid > 256) grid = 256;
  SellView view = sell_view(A);
  KScope ks(ctx, MI_K_SPMM);
#define SD(PV, HL, PKV)                                                                                       \
  hipLaunchKernelGGL((k_spmm_dots_stream<PV, HL, PKV>), dim3(grid), dim3(kBlock), 0, ctx->stream, view,       \
                     (const CgState *)ctx->cg_live, (const double *)V->d, W->d, ctx->partials)
#define SDP(PV)                                                                     \
  if (A->halo) { if (A->pk) SD(PV, true, true); else SD(PV, true, false); }         \
  else { if (A->pk) SD(PV, false, true); else SD(PV, false, false); }
  switch (p) {
    case 1: SDP(1); break;
    case 2: SDP(2); break;
    case 3: SDP(3); break;
    case 4: SDP(4); break;
    case 5: SDP(5); break;
    case 6: SDP(6); break;
    case 7: SDP(7); break;
    default: SDP(8); break;
  }
#undef SDP
#undef SD
  *nparts = grid;
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int csr_spmv_sub_scaled(const mi_csr *A, const mi_vec *V, const double *scale, const int *mode, const int *gate,
                        mi_vec *W, double *partials, int *nparts) {
  mi_ctx *ctx = A->ctx;
  MI_TRY(comm_halo_exchange(ctx, A, 1, V->d));
  int grid = uniform_grid(ctx, sell_groups(A));
  KScope ks(ctx, MI_K_SPMM);
  const bool no_stream = ctx->cfg.no_spmm_stream;
  if (!no_stream && sell_stream_ok(A, 1)) {
    if (!ctx->uniform_grid && grid > 256) grid = 256;  // one workgroup per CU, one round
#define SV(HL, PKV)                                                                                          \
  hipLaunchKernelGGL((k_spmv_sub_scaled_stream<HL, PKV>), dim3(grid), dim3(kBlock), 0, ctx->stream, sell_view(A), \
                     (const double *)V->d, scale, mode, gate, W->d, partials)
    if (A->halo) { if (A->pk) SV(true, true); else SV(true, false); }
    else { if (A->pk) SV(false, true); else SV(false, false); }
#undef SV
  } else {
    hipLaunchKernelGGL(k_spmv_sub_scaled, dim3(grid), dim3(kBlock), 0, ctx->stream, sell_view(A),
                       (const double *)V->d, scale, mode, gate, W->d, partials);
  }
  *nparts = grid;
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// A == A' with bitwise-equal values, on the square block of columns [0, n) (all of a single-GPU matrix; the diagonal
// block of a row shard, whose halo columns >= n are skipped).  O(nnz): counting-sort transpose, then the (column, value)
// lists of every row are compared after sorting them (rows are short).
bool csr_is_symmetric(size_t n, const int32_t *rowptr, const int32_t *col, const double *val) {
  std::vector<size_t> tp(n + 1, 0);
  size_t inside = 0;
  for (size_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if ((size_t)col[k] < n) { ++tp[(size_t)col[k] + 1]; ++inside; }
  for (size_t i = 0; i < n; ++i) tp[i + 1] += tp[i];
  std::vector<std::pair<int32_t, uint64_t>> t(inside);
  std::vector<size_t> fill(tp.begin(), tp.end() - 1);
  auto bits = [](double v) { uint64_t b; std::memcpy(&b, &v, sizeof b); return b; };
  for (size_t i = 0; i < n; ++i)
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if ((size_t)col[k] < n) t[fill[(size_t)col[k]]++] = {(int32_t)i, bits(val[k])};
  std::vector<std::pair<int32_t, uint64_t>> row;
  for (size_t i = 0; i < n; ++i) {
    row.clear();
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
      if ((size_t)col[k] < n) row.push_back({col[k], bits(val[k])});
    if (row.size() != tp[i + 1] - tp[i]) return false;
    std::sort(row.begin(), row.end());
    std::sort(t.begin() + tp[i], t.begin() + tp[i + 1]);  // (already ordered by row of origin; ties by value)
    if (!std::equal(row.begin(), row.end(), t.begin() + tp[i])) return false;
  }
  return true;
}

}  // namespace mi

extern "C" {

int mi_csr_create(mi_ctx *ctx, size_t n, size_t nnz, const int32_t *rowptr, const int32_t *col,
                  const double *val, mi_csr **out) {
  MI_REQUIRE(ctx && rowptr && out && (nnz == 0 || (col && val)), "null argument");
  MI_REQUIRE(n < (size_t)INT32_MAX, "n too large for int32 column indices");
  MI_REQUIRE(rowptr[0] == 0 && (size_t)rowptr[n] == nnz, "rowptr inconsistent with nnz");
  for (size_t i = 0; i < n; ++i) MI_REQUIRE(rowptr[i + 1] >= rowptr[i], "rowptr not monotone at row %zu", i);
  for (size_t k = 0; k < nnz; ++k)
    MI_REQUIRE(col[k] >= 0 && (size_t)col[k] < n, "column index out of range at entry %zu", k);
  MI_TRY(build_sell(ctx, n, n, nnz, rowptr, col, val, out));
  (*out)->symmetric = csr_is_symmetric(n, rowptr, col, val);
  return MI_OK;
}

int mi_csr_destroy(mi_csr *A) {
  if (!A) return MI_OK;
  (void)hipFree(A->slice_ptr);
  (void)hipFree(A->col);
  (void)hipFree(A->val);
  (void)hipFree(A->pk);
  (void)hipFree(A->vtab);
  (void)hipFree(A->wk);
  (void)hipFree(A->wfar);
  (void)hipFree(A->wk16);
  for (auto &kv : A->win_plans) (void)hipFree(kv.second.bounds);
  if (A->halo) comm_halo_free(A->ctx, A->halo, A->halo_in_arena);
  if (A->halo_r) comm_halo_free(A->ctx, A->halo_r, A->halo_r_in_arena);
  delete A;
  return MI_OK;
}

int mi_csr_spmm(const mi_csr *A, int p, const mi_vec *V, mi_vec *W) {
  MI_REQUIRE(A && V && W, "null argument");
  MI_REQUIRE(p >= 1 && p <= kMaxP, "p must be in [1,%d], got %d", kMaxP, p);
  MI_REQUIRE(V->n == A->n * (size_t)p && W->n == A->n * (size_t)p,
             "SpMM dimension mismatch: A has %zu rows, p=%d, V %zu, W %zu", A->n, p, V->n, W->n);
  MI_REQUIRE(V->d != W->d, "SpMM input and output must not alias");
  touch(W);
  MI_TRY(comm_halo_exchange(A->ctx, A, p, V->d));
  return csr_spmm_launch(A, p, V->d, W->d);
}

// Host-only planning step of a row-sharded matrix (no GPU needed): halo extents and the remap of
// global column indices to local ones ([0,n) local rows, [n,n+need_lo) halo from rank-1,
// [n+need_lo, n+need_lo+need_hi) halo from rank+1).  Columns outside the local range must belong to the
// ADJACENT ranks (slab-partitioned stencils); anything else is rejected.
int mi_csr_shard_plan(size_t n_global, int world_size, int rank, const size_t *row_starts, size_t nnz_local,
                      const int64_t *col_global, int32_t *col_local, size_t *need_lo_out, size_t *need_hi_out) {
  MI_REQUIRE(row_starts && (nnz_local == 0 || (col_global && col_local)) && need_lo_out && need_hi_out,
             "null argument");
  MI_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size/rank %d/%d", world_size, rank);
  MI_REQUIRE(row_starts[0] == 0 && row_starts[world_size] == n_global, "row_starts must span [0, n_global]");
  for (int r = 0; r < world_size; ++r)
    MI_REQUIRE(row_starts[r] <= row_starts[r + 1], "row_starts must be non-decreasing");
  const size_t row_begin = row_starts[rank], row_end = row_starts[rank + 1], n = row_end - row_begin;
  size_t need_lo = 0, need_hi = 0;
  for (size_t k = 0; k < nnz_local; ++k) {
    const int64_t c = col_global[k];
    MI_REQUIRE(c >= 0 && (size_t)c < n_global, "global column out of range at entry %zu", k);
    if ((size_t)c < row_begin) need_lo = std::max(need_lo, row_begin - (size_t)c);
    if ((size_t)c >= row_end) need_hi = std::max(need_hi, (size_t)c - row_end + 1);
  }
  if (need_lo)
    MI_REQUIRE(rank > 0 && need_lo <= row_begin - row_starts[rank - 1],
               "rank %d needs %zu rows below its range: not nearest-neighbour banded", rank, need_lo);
  if (need_hi)
    MI_REQUIRE(rank + 1 < world_size && need_hi <= row_starts[rank + 2] - row_end,
               "rank %d needs %zu rows above its range: not nearest-neighbour banded", rank, need_hi);
  MI_REQUIRE(n + need_lo + need_hi < (size_t)INT32_MAX, "local problem too large for int32 indices");
  for (size_t k = 0; k < nnz_local; ++k) {
    const size_t c = (size_t)col_global[k];
    if (c < row_begin) col_local[k] = (int32_t)(n + (need_lo - (row_begin - c)));
    else if (c >= row_end) col_local[k] = (int32_t)(n + need_lo + (c - row_end));
    else col_local[k] = (int32_t)(c - row_begin);
  }
  *need_lo_out = need_lo;
  *need_hi_out = need_hi;
  return MI_OK;
}

int mi_csr_create_sharded(mi_ctx *ctx, size_t n_global, size_t row_begin, size_t row_end,
                          size_t nnz_local, const int32_t *rowptr, const int64_t *col_global,
                          const double *val, const size_t *row_starts, mi_csr **out) {
  MI_REQUIRE(ctx && rowptr && col_global && val && row_starts && out, "null argument");
  MI_REQUIRE(row_begin <= row_end && row_end <= n_global, "bad local row range");
  const size_t n = row_end - row_begin;
  MI_REQUIRE(n < (size_t)INT32_MAX, "local row count too large for int32 indices");
  MI_REQUIRE(rowptr[0] == 0 && (size_t)rowptr[n] == nnz_local, "rowptr inconsistent with nnz_local");
  for (size_t i = 0; i < n; ++i) MI_REQUIRE(rowptr[i + 1] >= rowptr[i], "rowptr not monotone at row %zu", i);
  const int ws = ctx->world_size, rk = ctx->rank;
  MI_REQUIRE(row_starts[rk] == row_begin && row_starts[rk + 1] == row_end,
             "row_starts does not match this rank's range");
  size_t need_lo = 0, need_hi = 0;
  std::vector<int32_t> lcol(nnz_local);
  MI_TRY(mi_csr_shard_plan(n_global, ws, rk, row_starts, nnz_local, col_global, lcol.data(), &need_lo,
                           &need_hi));
  mi_csr *A = nullptr;
  MI_TRY(build_sell(ctx, n, n + need_lo + need_hi, nnz_local, rowptr, lcol.data(), val, &A, need_lo));
  A->symmetric = csr_is_symmetric(n, rowptr, lcol.data(), val);  // the diagonal block (the couplings are the caller's word)
  A->halo_lo = need_lo;
  A->halo_hi = need_hi;
  // what we must SEND equals what the neighbours need; exchanged once through the communicator
  size_t max_halo_rows = 0;
  int st = comm_exchange_halo_counts(ctx, need_lo, need_hi, &A->send_lo, &A->send_hi, &A->peer_lo_rows,
                                     &max_halo_rows);
  if (st == MI_OK) {
    // same size on every rank, so that the arena offsets of the peer-memory layer agree
    A->halo_stride = std::max<size_t>(1, max_halo_rows * kMaxP);  // doubles per buffer (p <= kMaxP); two buffers
    st = comm_halo_alloc(ctx, 2 * A->halo_stride * sizeof(double), &A->halo, &A->halo_in_arena, &A->halo_off);
  }
  if (st == MI_OK && !(A->send_lo <= n && A->send_hi <= n)) {
    set_error("neighbour halo request exceeds local rows");
    st = MI_ERR_INVALID_ARGUMENT;
  }
  if (st != MI_OK) {  // (the matrix and its device buffers must not leak on these paths)
    mi_csr_destroy(A);
    return st;
  }
  *out = A;
  return MI_OK;
}

// Verification hook: which form of the sparse kernels a matrix got (tests): out = {window half-width in chunks (0: no
// window form), widest slice, far stride D when the far structure is pure (computed far columns) else 0, halo rows}
int mi_debug_csr_window_info(const mi_csr *A, size_t out[4]) {
  MI_REQUIRE(A && out, "null argument");
  out[0] = (size_t)A->win_chunks;
  out[1] = (size_t)A->win_head;
  out[2] = A->win_far_pure;
  out[3] = A->halo_lo + A->halo_hi;
  return MI_OK;
}

// Host-only: the run plan of the LDS-window kernels (window_runs) for ntiles tiles, a workgroup budget, a CU count
// and the matrix's far stride in rows (0: none).  bounds_out (capacity cap) receives the first tile of every run and
// the end; *nb_out the number of runs.  No GPU needed (tests/test_cpu_oracle_templates.py).
int mi_debug_window_runs(int ntiles, int max_wgs, int num_cu, size_t far_stride, int *bounds_out, int cap,
                         int *nb_out) {
  MI_REQUIRE(bounds_out && nb_out && cap >= 2, "null argument");
  const std::vector<int> b = mi::window_runs(ntiles, max_wgs, num_cu, far_stride);
  MI_REQUIRE((int)b.size() <= cap, "plan of %d runs exceeds the output capacity %d", (int)b.size() - 1, cap - 1);
  for (size_t i = 0; i < b.size(); ++i) bounds_out[i] = b[i];
  *nb_out = (int)b.size() - 1;
  return MI_OK;
}

}  // extern "C"
