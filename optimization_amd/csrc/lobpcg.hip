// lobpcg.hip -- device building blocks of Optimization::LinearAlgebra::LOBPCG
// (reference: LinearAlgebra/LOBPCG.h:131-337) on tall-skinny COLUMN-MAJOR panels (m x k, ld = m,
// k <= 96), the layout of the reference's Eigen dense matrices:
//
//   mi_lobpcg_gram      G = S' T               LOBPCG.h:223,271-272   fp64 MFMA (v_mfma_f64_16x16x4_f64)
//   mi_lobpcg_update    Y = S C                LOBPCG.h:226-227,278,288
//   mi_lobpcg_residual  R = AX - BX diag(th);  column norms of R and X   LOBPCG.h:230,285,293,302
//   mi_rayleigh_ritz    small dense generalized symmetric-definite eigenproblem (host)   LOBPCG.h:53-62
//   mi_csr_spmm_colmajor  Y = A X for a column-major panel (the user operator of cfg5)
//
// Gram kernels: square panels of <= 80 columns (every Gram inside the LOBPCG loop) take k_gram_direct --
// no LDS, MFMA operands loaded straight from global memory in operand layout, one wave owning all output
// tiles (see there).  Other shapes take k_gram: a workgroup (8 waves) takes every gridDim-th 32-row tile,
// both panels staged in LDS column-major with leading dimension 34 (== 2 mod 32, so the 16 columns x 2
// rows that a half-wave reads for one MFMA operand hit 32 distinct 8-byte bank pairs); each wave
// accumulates its share (<= 5) of the (ka/16) x (kb/16) output tiles in registers, the next tile's
// global loads in flight during the MFMA phase.  Partial Grams (one per wave / workgroup) are summed in
// fixed order by k_gram_reduce (deterministic).
// Algorithmic bytes: 8 m (ka + kb)  (8 m ka when S == T); flops 2 m ka kb.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "mi_internal.h"
#include "spmm_core.h"

namespace mi {
int rr_host(int n, const double *A, const double *B, double *Theta, double *C);  // rr_host.cpp (g++, AVX2, no FMA)
int rr_host_lowest(int n, int k, const double *A, const double *B, double *Theta, double *C);
}

using namespace mi;

namespace {

typedef double double4v __attribute__((ext_vector_type(4)));

// A panel held as up to three column blocks that need not be adjacent (r04): LOBPCG's search basis
// S = [X, W(:, nc:), P(:, nc:)] (LOBPCG.h:254-264) as the three blocks lie, instead of a copy that moves the unlocked
// columns together every iteration.  Logical column c: [0, n0) -> block 0, [n0, n01) -> block 1, the rest -> block 2.
// One contiguous panel: base[0] with n0 = n01 = its width.
struct ColBlocks {
  const double *base[3];
  int n0, n01;
  __host__ __device__ __forceinline__ const double *col(int c, size_t m) const {
    return c < n0 ? base[0] + (size_t)c * m
                  : (c < n01 ? base[1] + (size_t)(c - n0) * m : base[2] + (size_t)(c - n01) * m);
  }
};

constexpr int kGramRows = 32;       // rows per LDS tile
constexpr int kGramLd = 34;         // LDS leading dimension
constexpr int kGramMaxK = 96;       // max panel width
constexpr int kGramThreads = 512;
constexpr int kGramWaves = kGramThreads / 64;
constexpr int kGramSegs = kGramThreads / 16;  // 16 threads x 16 B cover one 32-row column segment

constexpr int kColIters = kGramMaxK / kGramSegs;  // column segments a thread may own per panel

// Global -> registers: the 32-row x k-column tile at rows r0.. (zero-filled past m and past k).
// 16 threads (16 B each) cover one 32-row column segment; thread owns columns seg, seg+16, ...
template <bool ALIGNED>
__device__ __forceinline__ void load_tile(const double *__restrict__ P, size_t m, int k, size_t r0,
                                          double2 (&reg)[kColIters]) {
  const int seg = threadIdx.x >> 4, part = threadIdx.x & 15;
  const size_t r = r0 + 2 * (size_t)part;
#pragma unroll
  for (int i = 0; i < kColIters; ++i) {
    // columns past k are clamped to k-1: their products land in output rows/columns that are never
    // stored, and the loads stay branch-free (a load behind a divergent branch gets its own
    // s_waitcnt at the merge, which serialises the whole tile fetch)
    const int c = std::min(seg + kGramSegs * i, k - 1);
    double2 v = make_double2(0.0, 0.0);
    {
      const double *src = P + (size_t)c * m + r;
      if (ALIGNED && r0 + kGramRows <= m) {  // whole tile inside the panel (uniform test)
        v = *reinterpret_cast<const double2 *>(src);
      } else if (r + 1 < m) {
        if (ALIGNED) {
          v = *reinterpret_cast<const double2 *>(src);
        } else {  // column starts are only 8-byte aligned in general (m odd)
          v.x = src[0];
          v.y = src[1];
        }
      } else if (r < m) {
        v.x = src[0];
      }
    }
    reg[i] = v;
  }
}

// registers -> LDS (column-major, leading dimension kGramLd; 16-byte aligned because kGramLd is even)
__device__ __forceinline__ void store_tile(double *lds, int kpad, const double2 (&reg)[kColIters]) {
  const int seg = threadIdx.x >> 4, part = threadIdx.x & 15;
#pragma unroll
  for (int i = 0; i < kColIters; ++i) {
    const int c = seg + kGramSegs * i;
    if (c < kpad) *reinterpret_cast<double2 *>(lds + c * kGramLd + 2 * part) = reg[i];
  }
}

// tile number -> (ti, tj).  SAME (G = S'S, symmetric): only tiles with ti <= tj are computed, the
// reduce kernel mirrors them (the mirrored MFMA sums would be the same bits anyway).
template <bool SAME>
__device__ __forceinline__ void tile_decode(int tile, int tb, int &ti, int &tj) {
  if (SAME) {
    ti = 0;
    int len = tb;
    while (tile >= len) {
      tile -= len;
      ++ti;
      --len;
    }
    tj = ti + tile;
  } else {
    ti = tile / tb;
    tj = tile % tb;
  }
}

// The next tile's global loads are issued into registers BEFORE the MFMA phase of the current tile,
// so HBM latency overlaps the matrix pipe; 3 workgroups per CU (<= 52 KB LDS each) overlap each
// other's barrier phases.
// TPW = output tiles per wave = ceil(ntiles / 8), a compile-time constant so the MFMA phase is
// branch-free and the LDS operand reads of step kk+1 are issued ahead of the MFMAs of step kk; a wave
// whose last slot has no tile recomputes tile (0,0) there and discards it.
template <bool SAME, bool ALIGNED, int TPW>
__global__ __launch_bounds__(kGramThreads) void k_gram(size_t m, int ka, int kb, const double *__restrict__ S,
                                                       const double *__restrict__ T,
                                                       double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ta = (ka + 15) / 16, tb = (kb + 15) / 16;
  const int kapad = ta * 16, kbpad = tb * 16;
  double *ldsS = smem;
  double *ldsT = SAME ? smem : smem + (size_t)kapad * kGramLd;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = SAME ? ta * (ta + 1) / 2 : ta * tb;

  const double *pa[TPW], *pb[TPW];
  int row0[TPW], col0[TPW];
  double4v acc[TPW];
  const int loff = (lane & 15) * kGramLd + (lane >> 4);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    acc[t] = (double4v){0.0, 0.0, 0.0, 0.0};
    int ti = 0, tj = 0;
    if (w + kGramWaves * t < ntiles) tile_decode<SAME>(w + kGramWaves * t, tb, ti, tj);
    row0[t] = ti * 16;
    col0[t] = tj * 16;
    pa[t] = ldsS + ti * 16 * kGramLd + loff;
    pb[t] = ldsT + tj * 16 * kGramLd + loff;
  }

  // interleaved tiles: at step s the grid reads one contiguous band of 32 * gridDim rows of every
  // column (DRAM-page friendly); workgroup b takes tile b of each band
  const size_t band = (size_t)kGramRows * gridDim.x;
  const size_t rb = (size_t)blockIdx.x * kGramRows, re = m;
  double2 ra[kColIters], rt[kColIters];
  if (rb < re) {
    load_tile<ALIGNED>(S, m, ka, rb, ra);
    if (!SAME) load_tile<ALIGNED>(T, m, kb, rb, rt);
  }
  for (size_t r0 = rb; r0 < re; r0 += band) {
    __syncthreads();  // everyone finished reading the previous tile
    store_tile(ldsS, kapad, ra);
    if (!SAME) store_tile(ldsT, kbpad, rt);
    __syncthreads();
    if (r0 + band < re) {
      load_tile<ALIGNED>(S, m, ka, r0 + band, ra);
      if (!SAME) load_tile<ALIGNED>(T, m, kb, r0 + band, rt);
    }
    double av[2][TPW], bv[2][TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      av[0][t] = pa[t][0];
      bv[0][t] = pb[t][0];
    }
#pragma unroll
    for (int kk = 0; kk < kGramRows / 4; ++kk) {
      const int cur = kk & 1, nxt = cur ^ 1;
      if (kk + 1 < kGramRows / 4) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          av[nxt][t] = pa[t][4 * (kk + 1)];
          bv[nxt][t] = pb[t][4 * (kk + 1)];
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[cur][t], bv[cur][t], acc[t], 0, 0, 0);
    }
  }
  // partial Gram of this workgroup, column-major ka x kb; f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 j
  double *out = partial + (size_t)blockIdx.x * ka * kb;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (w + kGramWaves * t < ntiles) {
      const int col = col0[t] + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = row0[t] + (lane >> 4) + 4 * j;
        if (row < ka && col < kb) out[(size_t)col * ka + row] = acc[t][j];
      }
    }
  }
}

// ---- LDS-free Gram for square panels (ka == kb <= 80, m % 4 == 0, 32-byte aligned) ----------------
// A wave owns every nrow-th 16-row step and a compile-time range [LO, HI) of the T x T output tiles
// (all of them for T <= 4; for T = 5 two waves split them 13/12, so that the accumulators plus one
// operand set stay under 256 registers and TWO waves fit on a SIMD).  Lane (i = l & 15, q = l >> 4)
// loads the 4 consecutive rows r0 + 4q .. 4q+3 of column 16 t + i with one 32-byte load: a wave
// instruction reads 16 whole 128-byte lines, every byte used, and those four values are the lane's
// MFMA operands of the next four k-steps (step j contracts over the rows {r0 + 4q + j}; A and B use
// the same row partition, so the sum over the 16 rows is complete).  No LDS, no barrier, no explicit
// double buffering: the overlap of loads and MFMAs comes from the second wave on the SIMD.  (Explicit
// register double-buffering was tried first: at 400+ registers the compiler either spilled or put
// s_waitcnt vmcnt(0) in front of the first MFMA of every step, i.e. waited for the prefetch it had
// just issued.)  Only whole 16-row steps are handled here; the m % 16 leftover rows go through
// k_gram_tail.
typedef double double4l __attribute__((ext_vector_type(4)));
constexpr int kGdH = 1;  // 16-row blocks per pipeline step of k_gram_direct

constexpr int gd_tile_index(int T, bool same, int a, int b) {  // a-major; upper triangle only when same
  if (!same) return a * T + b;
  int idx = 0;
  for (int r = 0; r < a; ++r) idx += T - r;
  return idx + (b - a);
}
constexpr bool gd_in(int T, bool same, int lo, int hi, int a, int b) {
  return (!same || b >= a) && gd_tile_index(T, same, a, b) >= lo && gd_tile_index(T, same, a, b) < hi;
}
constexpr bool gd_need_row(int T, bool same, int lo, int hi, int a) {  // operand block a of S used as A?
  for (int b = 0; b < T; ++b)
    if (gd_in(T, same, lo, hi, a, b)) return true;
  return false;
}
constexpr bool gd_need_col(int T, bool same, int lo, int hi, int b) {  // operand block b used as B?
  for (int a = 0; a < T; ++a)
    if (gd_in(T, same, lo, hi, a, b)) return true;
  return false;
}

// The T panel may come in two pieces (columns [0, k1) at Tm, [k1, k) at T2; T2 == null: one piece): LOBPCG's
// AS = [AX | A(R) A(P)] without assembling it.  Same row partition, same order of contraction: same bits.
template <int T, bool SAME, int LO, int HI>
__device__ __forceinline__ void gram_direct_body(size_t mfull, size_t m, int k, const double *__restrict__ S,
                                                 const double *__restrict__ Tm, const double *__restrict__ T2,
                                                 int k1, size_t rowwave, size_t nrow,
                                                 double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  // columns past k are clamped to column k-1: their products land in rows/columns that are never stored
  const size_t lane_off = 4 * (size_t)(lane >> 4);
  const double *ps[T];
#pragma unroll
  for (int t = 0; t < T; ++t) ps[t] = S + (size_t)std::min(16 * t + (lane & 15), k - 1) * m + lane_off;
  const double *pt[SAME ? 1 : T];  // the T panel's operand columns
  if (!SAME) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int c = std::min(16 * t + (lane & 15), k - 1);
      pt[t] = (T2 && c >= k1 ? T2 + (size_t)(c - k1) * m : Tm + (size_t)c * m) + lane_off;
    }
  }
  double4v acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) acc[a][b] = (double4v){0.0, 0.0, 0.0, 0.0};
  // Software pipeline inside the wave: a step is kGdH x 16 rows; the loads of step s+1 (into na/nb) are
  // issued right after the operands of step s were moved to sa/sb, so they have the whole MFMA block of
  // step s (kGdH x 4 x ntiles MFMAs, ~3.5 us at 13 tiles) to land.  Two identical waves on a SIMD WITHOUT
  // such a pipeline run in lockstep (both wait for memory, then both want the matrix pipe): measured
  // MfmaUtil 44 %, 89 % of wave cycles in s_waitcnt.
  const size_t band = 16 * kGdH * nrow;
  auto load = [&](double4l(&va)[kGdH][T], double4l(&vb)[kGdH][T], size_t r) {
#pragma unroll
    for (int h = 0; h < kGdH; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        // SAME: one operand set serves both sides
        if (gd_need_row(T, SAME, LO, HI, t) || (SAME && gd_need_col(T, SAME, LO, HI, t)))
          va[h][t] = *reinterpret_cast<const double4l *>(ps[t] + r + 16 * h);
        if (!SAME && gd_need_col(T, SAME, LO, HI, t))
          vb[h][t] = *reinterpret_cast<const double4l *>(pt[SAME ? 0 : t] + r + 16 * h);
      }
  };
  auto mma = [&](const double4l(&va)[kGdH][T], const double4l(&vb)[kGdH][T]) {
#pragma unroll
    for (int h = 0; h < kGdH; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < T; ++a)
#pragma unroll
          for (int b = 0; b < T; ++b)
            if (gd_in(T, SAME, LO, HI, a, b))
              acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[h][a][j], SAME ? va[h][b][j] : vb[h][b][j],
                                                              acc[a][b], 0, 0, 0);
  };
  double4l sa[kGdH][T], sb[kGdH][T], na[kGdH][T], nb[kGdH][T];
  size_t r0 = rowwave * 16 * kGdH;
  if (r0 < mfull) load(na, nb, r0);
  for (; r0 < mfull; r0 += band) {
#pragma unroll
    for (int h = 0; h < kGdH; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t) {  // the one wait of the step: loads issued a whole MFMA block ago
        sa[h][t] = na[h][t];
        sb[h][t] = nb[h][t];
      }
    if (r0 + band < mfull) load(na, nb, r0 + band);
    mma(sa, sb);
  }
  // One partial Gram per WAVE, straight from the accumulators (f64 C/D map: col = lane & 15,
  // row = (lane >> 4) + 4 j).  Summing the four waves of a workgroup through LDS first was tried: the
  // extra live ranges pushed the T = 5 kernel into scratch (1700 us instead of 560).
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
      if (gd_in(T, SAME, LO, HI, a, b)) {
        const int col = b * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = a * 16 + (lane >> 4) + 4 * j;
          if (row < k && col < k) out[(size_t)col * k + row] = acc[a][b][j];
        }
      }
}

// amdgpu_waves_per_eu(1, 2): plan for at most two waves per SIMD.  Without a cap the scheduler chases
// occupancy it cannot use: it shortens live ranges by re-loading operands in the middle of the MFMA
// sequence, each reload behind an s_waitcnt vmcnt(0) (measured: 1200 us vs 640).
template <int T, bool SAME>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_gram_direct(
    size_t mfull, size_t m, int k, const double *__restrict__ S, const double *__restrict__ Tm,
    const double *__restrict__ T2, int k1, double *__restrict__ partial) {
  constexpr int NT = SAME ? T * (T + 1) / 2 : T * T;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  gram_direct_body<T, SAME, 0, NT>(mfull, m, k, S, Tm, T2, k1, wave, (size_t)gridDim.x * 4,
                                   partial + wave * (size_t)k * k);
}

// ---- both Grams of a Rayleigh-Ritz step in ONE pass over S (r04) ---------------------------------------------------
// G_A = S' [T1 | T2] (T = A S, the operator is SYMMETRIC: LinearAlgebra/Concepts.h SymmetricLinearOperator, so G_A is
// too) and G_B = S' S (B absent), reference LOBPCG.h:271-272, from one read of S and one of T.  The two separate
// launches are bound by the matrix pipe on PADDED tiles, not by memory (ns = 72 -> 5 x 5 tiles of 16: S'AS 25 tiles at
// 80 % of the measured 71.5 TF/s, S'S 15): here only the upper block triangle of BOTH is formed -- 30 tiles instead of
// 40, the mirror image is filled by the reduction as it always was for S'S -- and S is read once instead of twice
// (2.30 GB instead of 3.45 at ns = 72).  Upper triangle only is also what the reference's eigensolver looks at (Eigen's
// self-adjoint solvers read one triangle of their arguments).  Per wave: 30 x 4 accumulator doubles (240 registers, the
// compiler keeps them in the accumulation half of the 512-entry file) + two operand sets of 2 x T double4 (160): one
// wave per SIMD (amdgpu_waves_per_eu(1, 1)); the overlap of loads and MFMAs is the software pipeline of
// gram_direct_body (the loads of step s + 1 are issued before the 120 MFMAs of step s).
// r05, HALF (k mod 16 in 1 ... 8, e.g. cfg5's ns = 72 = 4.5 tiles): the last tile column of BOTH Grams is at most half
// full, so the two half columns share ONE 16-wide B operand -- lanes 0-7 of the operand hold columns 16 (T-1) + i of
// T = A(S), lanes 8-15 the same columns of S -- and one MFMA per (a, k-step) yields S'A(S)(:, last) in result columns
// 0-7 and S'S(:, last) in columns 8-15: T (T + 1) - T tiles instead of T (T + 1) (25 instead of 30 at T = 5), 8 T
// accumulator registers fewer, no extra loads (the operand is one load per lane either way; which panel a lane reads
// is decided once, in the pointer set-up).  Every product is the product the separate tiles form: same bits.
// PAIR = false: S'T alone (upper block triangle) -- the generalized problem forms S'A(S) and S'B(S) with two launches
// of it (r05); T as column blocks like S, so [AX | A(W) | A(P)] need not be assembled either.
#ifndef MI_GRAM_DEEP
#define MI_GRAM_DEEP 1
#endif
template <int T, bool HALF, bool PAIR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (PAIR ? T >= 4 : T >= 5) ? 1 : 2))) void k_gram_pair_sym(
    size_t mfull, size_t m, int k, ColBlocks S, ColBlocks Tb, double *__restrict__ partialA,
    double *__restrict__ partialB) {
  static_assert(!HALF || PAIR, "the shared last tile column is a property of the pair");
  const size_t rowwave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nrow = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  const size_t lane_off = 4 * (size_t)(lane >> 4);
  const double *ps[T], *pt[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int c = std::min(16 * t + (lane & 15), k - 1);  // (clamped: products of padding columns are never stored)
    ps[t] = S.col(c, m) + lane_off;
    pt[t] = Tb.col(c, m) + lane_off;
    if (HALF && t == T - 1) {  // the shared operand: columns 16 (T-1) + i of T in lanes i < 8, of S in lanes 8 + i
      const int i = lane & 15, c2 = std::min(16 * t + (i & 7), k - 1);
      pt[t] = (i < 8 ? Tb.col(c2, m) : S.col(c2, m)) + lane_off;
    }
  }
  constexpr int TB = HALF ? T - 1 : T;  // tile columns of the second Gram that have tiles of their own
  double4v accA[T][T], accB[PAIR ? T : 1][PAIR ? T : 1];  // (only a <= b is used)
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) {
      accA[a][b] = (double4v){0.0, 0.0, 0.0, 0.0};
      if (PAIR) accB[a][b] = (double4v){0.0, 0.0, 0.0, 0.0};
    }
  const size_t band = 16 * nrow;
  // the m % 16 leftover rows (m % 4 == 0: whole groups of four) are one more step of the wave whose turn it is, the
  // groups past m as zeros (r04: this was a kernel of its own per Gram, k_gram_tail, 5 us each plus the drain between)
  const size_t mend = mfull < m ? mfull + 16 : mfull;
  double4l sa[T], sb[T], na[T], nb[T];
  // DEEP (r05): the operands of TWO steps ahead are in flight (fa / fb).  One step's MFMA block -- 100 MFMAs at T = 5 with
  // the shared last column, ~1.5 us -- is shorter than a loaded memory round trip, so with one step of look-ahead the
  // wave waits for memory every step and the MFMAs the shared column saves buy nothing (measured: 527 us either way).
  constexpr bool DEEP = MI_GRAM_DEEP && PAIR && T >= 4;
  double4l fa[DEEP ? T : 1], fb[DEEP ? T : 1];
  size_t r0 = rowwave * 16;
  auto fetch = [&](double4l(&va)[DEEP ? T : T], double4l(&vb)[DEEP ? T : T], size_t r) {
    const bool ok = r + lane_off < m;  // (false only in the last, partial step)
    const size_t rs = ok ? r : 0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      va[t] = *reinterpret_cast<const double4l *>(ps[t] + rs);
      vb[t] = *reinterpret_cast<const double4l *>(pt[t] + rs);
    }
    if (r == mfull) {  // wave-uniform
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          va[t][j] = ok ? va[t][j] : 0.0;
          vb[t][j] = ok ? vb[t][j] : 0.0;
        }
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = a; b < T; ++b) {
          accA[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[a][j], sb[b][j], accA[a][b], 0, 0, 0);
          if (PAIR && b < TB) accB[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[a][j], sa[b][j], accB[a][b], 0, 0, 0);
        }
  };
  if constexpr (DEEP) {
    if (r0 < mend) fetch(na, nb, r0);
    if (r0 + band < mend) fetch(fa, fb, r0 + band);
    for (; r0 < mend; r0 += band) {
#pragma unroll
      for (int t = 0; t < T; ++t) {  // the one wait of the step: loads issued two MFMA blocks ago
        sa[t] = na[t];
        sb[t] = nb[t];
        na[t] = fa[t];
        nb[t] = fb[t];
      }
      if (r0 + 2 * band < mend) fetch(fa, fb, r0 + 2 * band);
      mma();
    }
  } else {
    if (r0 < mend) fetch(na, nb, r0);
    for (; r0 < mend; r0 += band) {
#pragma unroll
      for (int t = 0; t < T; ++t) {  // the one wait of the step
        sa[t] = na[t];
        sb[t] = nb[t];
      }
      if (r0 + band < mend) fetch(na, nb, r0 + band);
      mma();
    }
  }
  double *outA = partialA + rowwave * (size_t)k * k, *outB = PAIR ? partialB + rowwave * (size_t)k * k : nullptr;
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = a; b < T; ++b) {
      const int i = lane & 15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = a * 16 + (lane >> 4) + 4 * j;
        if (HALF && b == T - 1) {  // the shared tile: result columns 0-7 belong to the first Gram, 8-15 to the second
          const int col = b * 16 + (i & 7);
          if (row < k && col < k) (i < 8 ? outA : outB)[(size_t)col * k + row] = accA[a][b][j];
        } else {
          const int col = b * 16 + i;
          if (row < k && col < k) {
            outA[(size_t)col * k + row] = accA[a][b][j];
            if (PAIR) outB[(size_t)col * k + row] = accB[a][b][j];
          }
        }
      }
    }
}

// rows [r_begin, m) (fewer than 16) of S'T as one more partial Gram: one thread per output element
__global__ __launch_bounds__(256) void k_gram_tail(size_t m, size_t r_begin, int ka, int kb,
                                                   const double *__restrict__ S, const double *__restrict__ Tm,
                                                   const double *__restrict__ T2, int k1,
                                                   double *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ka * kb) return;
  const int row = e % ka, col = e / ka;
  const double *tc = (T2 && col >= k1) ? T2 + (size_t)(col - k1) * m : Tm + (size_t)col * m;
  double s = 0;
  for (size_t r = r_begin; r < m; ++r) s += S[(size_t)row * m + r] * tc[r];
  out[e] = s;
}

// G[e] = sum over workgroups of partial[b][e], fixed order; sym: element (row, col) below the block
// diagonal is read from its mirror (col, row)
// A workgroup handles kRedElems elements; group g of kRedGroups sums the workgroups b == g (mod
// kRedGroups) in ascending order, then the groups are added in ascending order: fixed, launch-independent.
constexpr int kRedElems = 16, kRedGroups = 32;
__global__ __launch_bounds__(kRedElems *kRedGroups) void k_gram_reduce(int nblocks, int ka, int nelem, int sym,
                                                                       const double *__restrict__ partial,
                                                                       double *__restrict__ G,
                                                                       const double *__restrict__ partial2 = nullptr,
                                                                       double *__restrict__ G2 = nullptr) {
  if (blockIdx.y == 1) {  // (the second Gram of a pair, same shape: one launch for both)
    partial = partial2;
    G = G2;
  }
  __shared__ double part[kRedGroups][kRedElems];
  const int ex = threadIdx.x % kRedElems, g = threadIdx.x / kRedElems;
  const int e = blockIdx.x * kRedElems + ex;
  double s = 0;
  // sym: elements below the block diagonal are not summed here (their mirror's thread stores them too): reading them
  // from the mirrored positions was a 576-byte-strided gather, 3x the bytes of the whole reduction
  const int erow = e % ka, ecol = e / ka;
  const bool lower = sym && e < nelem && erow / 16 > ecol / 16;
  if (e < nelem && !lower) {
    const int src = e;
    // eight independent loads in flight, added in the same ascending order as one at a time
    int b = g;
    for (; b + 7 * kRedGroups < nblocks; b += 8 * kRedGroups) {
      double t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = partial[(size_t)(b + i * kRedGroups) * nelem + src];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += t[i];
    }
    for (; b < nblocks; b += kRedGroups) s += partial[(size_t)b * nelem + src];
  }
  part[g][ex] = s;
  __syncthreads();
  if (g == 0 && e < nelem && !lower) {
    double tot = part[0][ex];
#pragma unroll
    for (int q = 1; q < kRedGroups; ++q) tot += part[q][ex];
    G[e] = tot;
    if (sym && erow / 16 < ecol / 16) G[(size_t)erow * ka + ecol] = tot;  // the mirrored element (col, row)
  }
}

// Y[:, c0:c0+KC) = S (m x ks) C[:, c0:c0+KC); one thread per row, KC accumulators in registers, the
// C chunk in LDS (row-major by s, so one s needs KC consecutive broadcast reads).  KC = 24 covers a
// whole nx = 24 block in ONE pass over S (the panel is the traffic: 8 m ks bytes per pass).
// The coefficients are wave-uniform: Ct (this chunk, ks x KC, row-major by s, zero-padded past kc) is
// read with SCALAR loads (s_load_dwordx*, SGPR operand of v_fma_f64), not through LDS: with C in LDS
// every FMA pair cost one broadcast ds_read and the kernel was LDS-issue bound (607 us for 72 x 48).
template <int KC>
__global__ __launch_bounds__(256) void k_panel_update(size_t m, int ks, const double *__restrict__ S,
                                                      const double *__restrict__ Ct, int c0, int kc,
                                                      double *__restrict__ Y, int k1, double *__restrict__ Y2,
                                                      size_t r_begin) {
  const double *__restrict__ smem = Ct;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t r = r_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += stride) {
    double acc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] = 0;
    const double *sp = S + r;
    int s = 0;
    for (; s + 4 <= ks; s += 4) {  // four independent column loads in flight
      const double v0 = sp[(size_t)s * m], v1 = sp[(size_t)(s + 1) * m], v2 = sp[(size_t)(s + 2) * m],
                   v3 = sp[(size_t)(s + 3) * m];
      const double *cr = smem + s * KC;
      // column blocks of 8: 4 x 8 coefficients = 64 SGPRs live at a time (all 4 x KC at once spills)
#pragma unroll
      for (int cb = 0; cb < KC; cb += 8) {
#pragma unroll
        for (int c = cb; c < cb + 8; ++c) acc[c] += v0 * cr[c];
#pragma unroll
        for (int c = cb; c < cb + 8; ++c) acc[c] += v1 * cr[KC + c];
#pragma unroll
        for (int c = cb; c < cb + 8; ++c) acc[c] += v2 * cr[2 * KC + c];
#pragma unroll
        for (int c = cb; c < cb + 8; ++c) acc[c] += v3 * cr[3 * KC + c];
      }
    }
    for (; s < ks; ++s) {
      const double sv = sp[(size_t)s * m];
      const double *cr = smem + s * KC;
#pragma unroll
      for (int c = 0; c < KC; ++c) acc[c] += sv * cr[c];
    }
    // output columns [0, k1) go to Y, [k1, kc) to Y2 (two-destination form; k1 == kc: one panel)
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c0 + c < kc) {
        const int col = c0 + c;
        if (col < k1) Y[(size_t)col * m + r] = acc[c];
        else Y2[(size_t)(col - k1) * m + r] = acc[c];
      }
  }
}

// The 48-column update (X and P of one LOBPCG iteration) on the matrix pipe, operand-stationary: the transposed
// product Y' = C' S' in 16 x 16 tiles, A operand = C' (16 output columns x 4 basis columns per step) held in
// REGISTERS for the whole kernel (KS4 steps x 3 tiles: <= 54 doubles per lane), B operand = 16 rows x 4 basis
// columns of S straight from global memory (four 128-byte row segments per wave load, every byte used once), the next
// 16-row block's operands in flight during the current block's MFMAs.  Lane (i = l & 15, q = l >> 4) of the result
// holds Y[row0 + i][16 t + q + 4 j], so a store instruction writes four 128-byte segments.  The VALU form above
// streams the 27 KB coefficient block through the scalar cache once per 64 rows and sits at 40 % of the fp64 rate.
// Rows are taken in 16-row blocks; the m % 16 leftover rows are one more, masked block.
// The basis may lie in up to three column blocks (mi_panel_blocks): the host cuts it into steps of four columns that
// never straddle two blocks and hands over the address of each step's first column (wave-uniform: scalar registers, the
// per-lane part q m + i is shared by all steps).  A block's last step, when the block's width is not a multiple of
// four, is its LAST four columns, the ones the step before already covered carrying zero coefficients.
struct StepBases {
  const double *p[18];
};
template <int KS4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_panel_update_mfma(
    size_t nblocks, size_t m, StepBases S, const double *__restrict__ Ct, int kc, double *__restrict__ Y, int k1,
    double *__restrict__ Y2) {
  constexpr int NT = 3;  // 48 output columns
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  // coefficients: A operand of step kk, tile t: row 4 kk + q of Ct (4 KS4 x 48, row-major; the host has put the
  // coefficients of the step's four basis columns there), column 16 t + i
  double ca[KS4][NT];
#pragma unroll
  for (int kk = 0; kk < KS4; ++kk)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ca[kk][t] = Ct[(size_t)(4 * kk + q) * 48 + 16 * t + i];
    }
  // B operand of step kk: rows row0 + i of the step's column q
  const size_t lane_off = (size_t)q * m + i;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  // no explicit double buffering: with it the 18-step instance needs 292 registers (one wave per SIMD); at <= 256 two
  // waves share a SIMD and one's operand loads overlap the other's MFMA block
  // PART: the m % 16 leftover rows as one more block -- loads clamped to the last row, stores of rows < m only (r04: this
  // was a kernel of its own, k_panel_update_tail, 8 us behind the product)
  auto block = [&](size_t b, auto part) {
    constexpr bool PART = decltype(part)::value;
    const size_t rowl = b * 16 + i;
    const size_t off = PART ? (rowl < m ? rowl : m - 1) - (size_t)i : b * 16;
    double cur[KS4];
#pragma unroll
    for (int kk = 0; kk < KS4; ++kk) cur[kk] = S.p[kk][lane_off + off];
    double4v acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KS4; ++kk)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[kk][t], cur[kk], acc[t], 0, 0, 0);
    if (PART && rowl >= m) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 16 * t + q + 4 * j;
        if (col < kc) {
          // non-temporal stores (r04): the 0.77 GB of X and P written here are read by nobody before they have left
          // the L2s anyway, and as ordinary stores they push the S lines this kernel is streaming out of them --
          // 449 -> 403 us per call at cfg5 (non-temporal LOADS of S instead: 507 us)
          if (col < k1) __builtin_nontemporal_store(acc[t][j], Y + (size_t)col * m + rowl);
          else __builtin_nontemporal_store(acc[t][j], Y2 + (size_t)(col - k1) * m + rowl);
        }
      }
  };
  for (size_t b = wave; b < nblocks; b += nwaves) block(b, std::false_type{});
  if (nblocks * 16 < m && nblocks % nwaves == wave) block(nblocks, std::true_type{});
}

// r05: the same product in blocks of 32 rows as TWO interleaved tiles -- lane (i, q) owns rows row0 + 2 i and
// row0 + 2 i + 1: one 16-byte load of S(row0 + 2 i .. + 1, column 4 kk + q) is the B operand of the even-row tile (.x) and
// of the odd-row tile (.y), and the two results of a lane for an output column are two consecutive rows: one 16-byte
// store.  Every load and store instruction moves 16 bytes per lane (256-byte segments per 16 lanes) instead of 8:
// tools/microbench/panel_stream.hip measured the written third of this kernel's traffic 12 % faster that way
// (VERDICT r04, item 3 i).  Same MFMAs on the same operands: the bits of k_panel_update_mfma.  m even (16-byte
// aligned column starts); the m % 32 leftover rows are one more, clamped block.
template <int KS4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_panel_update_mfma2(
    size_t nblocks, size_t m, StepBases S, const double *__restrict__ Ct, int kc, double *__restrict__ Y, int k1,
    double *__restrict__ Y2) {
  constexpr int NT = 3;  // 48 output columns
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  double ca[KS4][NT];
#pragma unroll
  for (int kk = 0; kk < KS4; ++kk)
#pragma unroll
    for (int t = 0; t < NT; ++t) ca[kk][t] = Ct[(size_t)(4 * kk + q) * 48 + 16 * t + i];
  const size_t lane_off = (size_t)q * m + 2 * (size_t)i;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  auto block = [&](size_t b, auto part) {
    constexpr bool PART = decltype(part)::value;
    const size_t rowl = b * 32 + 2 * (size_t)i;  // (even; m even: rowl < m implies rowl + 1 < m)
    const size_t off = PART ? (rowl < m ? rowl : m - 2) - 2 * (size_t)i : b * 32;
    double2 cur[KS4];
#pragma unroll
    for (int kk = 0; kk < KS4; ++kk) cur[kk] = *reinterpret_cast<const double2 *>(S.p[kk] + lane_off + off);
    double4v accE[NT], accO[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      accE[t] = (double4v){0.0, 0.0, 0.0, 0.0};
      accO[t] = (double4v){0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int kk = 0; kk < KS4; ++kk)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        accE[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[kk][t], cur[kk].x, accE[t], 0, 0, 0);
        accO[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[kk][t], cur[kk].y, accO[t], 0, 0, 0);
      }
    if (PART && rowl >= m) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 16 * t + q + 4 * j;
        if (col < kc) {
          double *dst = (col < k1 ? Y + (size_t)col * m : Y2 + (size_t)(col - k1) * m) + rowl;
          typedef double double2v __attribute__((ext_vector_type(2)));
          __builtin_nontemporal_store((double2v){accE[t][j], accO[t][j]}, reinterpret_cast<double2v *>(dst));
        }
      }
  };
  for (size_t b = wave; b < nblocks; b += nwaves) block(b, std::false_type{});
  if (nblocks * 32 < m && nblocks % nwaves == wave) block(nblocks, std::true_type{});
}

// columns [c0, c0 + 8): R = AX - BX theta; partial rows of |R_j|^2 (comps 0..7) and |X_j|^2 (comps 8..15)
__global__ __launch_bounds__(kBlock) void k_residual(size_t m, int nx, int c0, const double *__restrict__ AX,
                                                     const double *__restrict__ BX, const double *__restrict__ X,
                                                     const double *__restrict__ theta, double *__restrict__ R,
                                                     double *__restrict__ partials) {
  __shared__ double lds[16 * kWaves];
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 0;
  double th[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) th[c] = (c0 + c < nx) ? theta[c0 + c] : 0.0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t r = (size_t)blockIdx.x * kBlock + threadIdx.x; r < m; r += stride) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c0 + c < nx) {
        const size_t idx = (size_t)(c0 + c) * m + r;
        const double res = __builtin_fma(-BX[idx], th[c], AX[idx]);  // (explicit: k_spmm_colmajor_win<.., RES> does the same)
        const double xv = X[idx];
        R[idx] = res;
        a[c] = __builtin_fma(res, res, a[c]);
        a[8 + c] = __builtin_fma(xv, xv, a[8 + c]);
      }
    }
  }
  block_partials_store<16>(a, lds, partials);
}

// Y = A X for a column-major panel: one thread per row, KC columns per pass over the matrix (the
// matrix stream, 12 B/nnz, is re-read once per pass; X gathers of a wave are contiguous per column
// for banded matrices).  Columns past k are clamped to the last valid one and not stored.
template <int KC>
__global__ __launch_bounds__(256) void k_spmm_colmajor(size_t n, size_t nslices, const long long *__restrict__ sp,
                                                       const int *__restrict__ col, const double *__restrict__ val,
                                                       int k, int c0, const double *__restrict__ X,
                                                       double *__restrict__ Y) {
  const int lane = threadIdx.x & 63;
  // contiguous row ranges per XCD: the neighbour rows a stencil-like matrix gathers then live in
  // the same XCD's L2 instead of being refetched by all eight
  const size_t slice = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const size_t row = slice * 64 + lane;
  double acc[KC];
  const double *xc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    acc[c] = 0;
    xc[c] = X + (size_t)std::min(c0 + c, k - 1) * n;
  }
  const long long b0 = sp[slice], b1 = sp[slice + 1];
  for (long long kk = b0; kk < b1; ++kk) {
    const size_t e = (size_t)kk * 64 + lane;
    // matrix stream and result are touched once: keep them from evicting the X lines that the
    // neighbouring rows (one grid plane = 3 MB of X at 24 columns) are about to gather again
    const double a = __builtin_nontemporal_load(val + e);
    const size_t j = (size_t)__builtin_nontemporal_load(col + e);
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] += a * xc[c][j];
  }
  if (row < n) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c0 + c < k) __builtin_nontemporal_store(acc[c], Y + (size_t)(c0 + c) * n + row);
  }
}

// The same product from the value-indexed packed copy of the matrix (mi_csr::pk: one dword per entry,
// (col - row) << 8 | value index): a third of the matrix bytes per pass, which makes NARROW column chunks
// affordable -- and narrow chunks are what keeps the gathered windows of X (KC columns x the rows a stencil
// reaches, two grid planes for the 7-point operator) inside an XCD's 4 MB L2.
template <int KC>
__global__ __launch_bounds__(256) void k_spmm_colmajor_pk(size_t n, size_t nslices, const long long *__restrict__ sp,
                                                          const uint32_t *__restrict__ pk,
                                                          const double *__restrict__ vtab, int k, int c0,
                                                          const double *__restrict__ X, double *__restrict__ Y) {
  __shared__ double vt[256];
  vt[threadIdx.x] = vtab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t slice = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const size_t row = slice * 64 + lane;
  double acc[KC];
  const double *xc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    acc[c] = 0;
    xc[c] = X + (size_t)std::min(c0 + c, k - 1) * n;
  }
  const long long b0 = sp[slice], b1 = sp[slice + 1];
  for (long long kk = b0; kk < b1; ++kk) {
    const uint32_t w = __builtin_nontemporal_load(pk + (size_t)kk * 64 + lane);
    const double a = vt[w & 255u];
    const size_t j = (size_t)((long long)row + (long long)((int32_t)w >> 8));
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] += a * xc[c][j];
  }
  if (row < n) {
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c0 + c < k) __builtin_nontemporal_store(acc[c], Y + (size_t)(c0 + c) * n + row);
  }
}

// The packed product in WINDOW form (the matrix's wk / wfar arrays, sparse.hip build_window; spmm_core.h): a workgroup
// of 4 waves takes a contiguous run of tiles; the rows of X near its own live in an LDS ring, column-major like the
// panel (KC columns x ring rows; staging a chunk is KC coalesced 512-byte loads and conflict-free ds_write_b64), so
// only the rows' own chunk and the two far rows of every row are loaded from global memory: 3.25 loads per output
// element instead of 7 through the L1-miss path, which is what bounds k_spmm_colmajor_pk (317 us per 24 columns at
// cfg5; 180 us when all seven hit L1).  Far rows are gathered one tile ahead (they are then in flight while their
// owner stages them: one fetch into the XCD's L2) and enter the entry loop from registers, selected per lane, in
// storage order: the same fused multiply-adds in the same order as the kernels above.
// RES (r03): the product is LOBPCG's A(X) of the NEW Ritz block (LOBPCG.h:281) and B is absent: the residual
// R = AX - X diag(theta) (:285) and the partial rows of |R_j|^2, |X_j|^2 (:293,302) are finished in the same pass -- the
// row's own X values are in the ring anyway -- instead of a pass of their own that re-reads AX and X (k_residual).
// Same operations on the same operands as k_residual (one fused multiply-add per element, squares accumulated by fma):
// R has its bits; the norms differ from its by the grouping of their sums only.
// the Ritz values of the residual form travel as kernel arguments (r04: they were a staged upload -- a 4 us copy kernel
// and a dependent launch -- between the panel update and the product)
struct ThetaArg {
  double v[kGramMaxK];
};
template <int KC, int HW, int WC, bool FARD, bool RES = false>
__global__ __launch_bounds__(kWinBlock) void k_spmm_colmajor_win(SellView A, WinView W, int k, int c_first, int nruns,
                                                                 ColBlocks X, double *__restrict__ Y,
                                                                 ThetaArg theta, double *__restrict__ R,
                                                                 double *__restrict__ partials_all) {
  // blockIdx.y: the pass (KC columns each) -- all passes of a panel in ONE launch (r04): no drain / ramp between them
  // and, in the residual form, one reduction behind the lot; blockIdx.x: the run of tiles, gridDim.x = nruns rounded
  // up to whole XCD rounds so that x mod 8 is the XCD in every pass (xcd_remap)
  const int c0 = c_first + (int)blockIdx.y * KC;
  double *const partials = RES ? partials_all + (size_t)blockIdx.y * (2 * KC) * kMaxRows : nullptr;
  extern __shared__ __attribute__((aligned(16))) double ring_dyn[];  // KC x (ring rows + zero row)
  __shared__ double vt[256];
  __shared__ double red[RES ? 2 * KC * kWinWaves : 1];
  double nr[RES ? KC : 1], nxs[RES ? KC : 1], th[RES ? KC : 1];
  if (RES) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      nr[c] = 0;
      nxs[c] = 0;
      th[c] = (c0 + c < k) ? theta.v[c0 + c] : 0.0;  // (c0 is wave-uniform: scalar loads from the argument segment)
    }
  }
  constexpr int NW = kWinWaves;
  static_assert(kWinBlock == 256 && kFarCap == 2, "table fill and far-slot decoding");
  vt[threadIdx.x] = A.vtab[threadIdx.x];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const int nchunks = (int)A.nslices, ntiles = (nchunks + NW - 1) / NW, per = (ntiles + (int)nb - 1) / (int)nb;
  int t0 = (int)lb * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  if (W.bounds) {
    const unsigned lc = lb < (unsigned)nruns ? lb : (unsigned)nruns;  // (a padding workgroup: the empty run at the end)
    t0 = scalar_int(W.bounds, lc);
    t1 = lb < (unsigned)nruns ? scalar_int(W.bounds, lc + 1) : t0;
  }
  if (t0 >= t1) {
    if constexpr (RES) {  // an idle workgroup still owns a partial row: zeros
      double z[2 * KC];
#pragma unroll
      for (int c = 0; c < 2 * KC; ++c) z[c] = 0;
      block_partials_store_nw<2 * KC, kWinWaves>(z, red, partials);
    }
    return;
  }
  // (the half-width is a template parameter: the column offsets c * RR of the ring are then immediates of the LDS
  // instructions instead of eight address additions per entry)
  constexpr int wc = WC, nc = 2 * NW + 2 * WC, zrow = nc * 64;
  constexpr unsigned RR = (unsigned)zrow + 1u;  // ring rows per column (+ the zero row)
  LdsDouble *const L = (LdsDouble *)ring_dyn;
  const size_t m = A.n;
  const unsigned mbytes = (unsigned)(m * 8);
  const unsigned lane8 = (unsigned)lane * 8u, lane4 = (unsigned)lane * 4u;
  const double *xc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) xc[c] = X.col(std::min(c0 + c, k - 1), m);
  auto chunk_load = [&](int q, double (&buf)[KC]) {
    const unsigned off = (unsigned)q * 512u + lane8, offs = off < mbytes ? off : 0u;  // past the last row: any valid address
#pragma unroll
    for (int c = 0; c < KC; ++c)
      buf[c] = pinned_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(xc[c]) + offs));
  };
  auto chunk_store = [&](int slot, const double (&buf)[KC]) {
    LdsDouble *dst = L + (unsigned)slot * 64u + (unsigned)lane;
#pragma unroll
    for (int c = 0; c < KC; ++c) dst[(unsigned)c * RR] = buf[c];
  };
  auto load_words = [&](unsigned (&cw)[HW], int kk) {
    const char *pb = reinterpret_cast<const char *>(W.wk) + (unsigned)kk * 256u + lane4;
#pragma unroll
    for (int j = 0; j < HW; ++j) cw[j] = pinned_load(reinterpret_cast<const unsigned *>(pb + j * 256));
  };
  auto load_far = [&](unsigned (&f)[kFarCap], int sl) {
    if constexpr (FARD) {  // pure far structure (mi_csr::win_far_pure): row + D, row - D, computed
      const unsigned nrows = (unsigned)m, r = (unsigned)sl * 64u + (unsigned)lane, rr = r < nrows ? r : nrows - 1u;
      f[0] = rr + W.far_d < nrows ? rr + W.far_d : rr;
      f[1] = rr >= W.far_d ? rr - W.far_d : rr;
    } else {
      const char *fb = reinterpret_cast<const char *>(W.wfar) + (unsigned)sl * (unsigned)(kFarCap * 256) + lane4;
#pragma unroll
      for (int j = 0; j < kFarCap; ++j) f[j] = pinned_load(reinterpret_cast<const unsigned *>(fb + j * 256));
    }
  };
  // (MI_SPMM_ABLATE_FAR: timing-only experiment builds, WRONG results -- what the two far gathers of a row cost.
  //  1: they read rows next to the wave's own chunk instead (same instructions, lines that are in L1 / L2 anyway: the
  //     time of the pass if the far rows cost no traffic beyond the L2); 2: no far loads at all.  r06, EXPERIMENTS.md)
  auto far_rows = [&](const unsigned (&f)[kFarCap], double (&g)[kFarCap][KC]) {
#pragma unroll
    for (int s_ = 0; s_ < kFarCap; ++s_)
#pragma unroll
      for (int c = 0; c < KC; ++c) {
#if defined(MI_SPMM_ABLATE_FAR) && MI_SPMM_ABLATE_FAR == 2
        g[s_][c] = (double)f[s_];
#elif defined(MI_SPMM_ABLATE_FAR) && MI_SPMM_ABLATE_FAR == 1
        // f[0] = row + D, f[1] = row - D (load_far): back to the row itself, one / two chunks further on
        const unsigned near_ = (s_ == 0 ? f[0] - W.far_d : f[1] + W.far_d) + 64u * (unsigned)(s_ + 1);
        g[s_][c] = pinned_load(xc[c] + (near_ < (unsigned)m ? near_ : 0u));
#else
        g[s_][c] = pinned_load(xc[c] + f[s_]);
#endif
      }
  };

  int slice = t0 * NW + w;
  bool have = slice < nchunks;
  int kq = 0, b1 = 0;
  if (have) {
    kq = (int)slice_bound(A.slice_ptr, slice);
    b1 = (int)slice_bound(A.slice_ptr, slice + 1);
  }
  unsigned f0[kFarCap], cw[HW], fn[kFarCap];
  load_far(f0, have ? slice : 0);
  load_words(cw, kq);
  int slot_own;
  {
    const int lo = t0 * NW - wc;
    int hi = t0 * NW + NW + wc;
    if (hi > nchunks) hi = nchunks;
    constexpr int kFill = (NW + 2 * kMaxWinChunks + NW - 1) / NW;
    double b[kFill][KC];
#pragma unroll
    for (int i = 0; i < kFill; ++i) {
      const int q = lo + w + i * NW;
      chunk_load(q >= 0 && q < hi ? q : 0, b[i]);
    }
#pragma unroll
    for (int i = 0; i < kFill; ++i) {
      const int q = lo + w + i * NW;
      if (q >= 0 && q < hi) chunk_store(q % nc, b[i]);
    }
    if (threadIdx.x < KC) L[(unsigned)threadIdx.x * RR + (unsigned)zrow] = 0.0;
    slot_own = (t0 * NW + w) % nc;
  }
  int slot_job = (t0 * NW + NW + wc + w) % nc;
  double gf[kFarCap][KC];
  {
    const bool more0 = have && t0 + 1 < t1 && slice + NW < nchunks;
    load_far(fn, more0 ? slice + NW : (have ? slice : 0));
    far_rows(f0, gf);
  }
  lds_barrier();

  for (int t = t0; t < t1; ++t) {
    const int qj = (t + 1) * NW + wc + w;
    const bool job = t + 1 < t1 && qj < nchunks;
    double pre[KC];
    chunk_load(job ? qj : 0, pre);
    if (have) {
      const int nslice = slice + NW;
      const bool more = t + 1 < t1 && nslice < nchunks;
      int q0 = kq, q1 = b1;
      if (more) { q0 = (int)slice_bound(A.slice_ptr, nslice); q1 = (int)slice_bound(A.slice_ptr, nslice + 1); }
      unsigned wd[HW];
#pragma unroll
      for (int j = 0; j < HW; ++j) wd[j] = (kq + j < b1) ? cw[j] : W.zw;
      double gn[kFarCap][KC];
      far_rows(fn, gn);
      load_words(cw, q0);
      {
        const int nslice2 = nslice + NW;
        const bool more2 = more && t + 2 < t1 && nslice2 < nchunks;
        load_far(fn, more2 ? nslice2 : (more ? nslice : slice));
      }
      double acc[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) acc[c] = 0;
      // Entry j of a slice is nearly always of one kind for all 64 rows (a stencil's j-th neighbour): one scalar
      // branch per entry on the ballots picks ring / first far row / second far row for the whole wave; only mixed
      // slices (grid boundaries) pay per-lane selects.  (With selects everywhere the loop was ~950 instructions per
      // slice and the kernel VALU-bound.)
      const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
      for (int j = 0; j < HW; ++j) {
        const double a = vt[wd[j] & 255u];
        const int ri0 = (int)(wd[j] >> 8), fr = ri0 - (zrow + 1);
        const bool isfar = fr >= 0, second = ((fr >> 6) & 1) != 0;
        const unsigned long long mfar = __builtin_amdgcn_ballot_w64(isfar);
        const unsigned long long msec = __builtin_amdgcn_ballot_w64(isfar && second);
        if (mfar == 0) {
          const LdsDouble *l = L + (unsigned)ri0;
#pragma unroll
          for (int c = 0; c < KC; ++c) acc[c] = __builtin_fma(a, l[(unsigned)c * RR], acc[c]);
        } else if (mfar == all && msec == 0) {
#pragma unroll
          for (int c = 0; c < KC; ++c) acc[c] = __builtin_fma(a, gf[0][c], acc[c]);
        } else if (mfar == all && msec == all) {
#pragma unroll
          for (int c = 0; c < KC; ++c) acc[c] = __builtin_fma(a, gf[1][c], acc[c]);
        } else {
          const LdsDouble *l = L + (isfar ? (unsigned)zrow : (unsigned)ri0);
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            double v = l[(unsigned)c * RR];
            v = isfar ? (second ? gf[1][c] : gf[0][c]) : v;
            acc[c] = __builtin_fma(a, v, acc[c]);
          }
        }
      }
      if (job) chunk_store(slot_job, pre);
      const size_t row = (size_t)slice * 64 + lane;
      if (row < m) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
          if (c0 + c < k) __builtin_nontemporal_store(acc[c], Y + (size_t)(c0 + c) * m + row);
        if constexpr (RES) {
          const LdsDouble *own = L + (unsigned)slot_own * 64u + (unsigned)lane;
#pragma unroll
          for (int c = 0; c < KC; ++c)
            if (c0 + c < k) {
              const double xv = own[(unsigned)c * RR];
              const double res = __builtin_fma(-xv, th[c], acc[c]);   // AX - BX theta with BX = X, as k_residual
              R[(size_t)(c0 + c) * m + row] = res;  // (non-temporal: no measurable change, r04)
              nr[c] = __builtin_fma(res, res, nr[c]);
              nxs[c] = __builtin_fma(xv, xv, nxs[c]);
            }
        }
      }
#pragma unroll
      for (int s_ = 0; s_ < kFarCap; ++s_)
#pragma unroll
        for (int c = 0; c < KC; ++c) gf[s_][c] = gn[s_][c];
      have = more;
      slice = nslice;
      kq = q0;
      b1 = q1;
    } else if (job) {
      chunk_store(slot_job, pre);
    }
    slot_own += NW;
    slot_own = slot_own >= nc ? slot_own - nc : slot_own;
    slot_job += NW;
    slot_job = slot_job >= nc ? slot_job - nc : slot_job;
    if (t + 1 < t1) lds_barrier();
  }
  if constexpr (RES) {  // components 0..KC-1: |R_j|^2, KC..2KC-1: |X_j|^2 (the layout k_residual leaves)
    double a[2 * KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      a[c] = nr[c];
      a[KC + c] = nxs[c];
    }
    block_partials_store_nw<2 * KC, kWinWaves>(a, red, partials);
  }
}

// ---- the panel product in PLANE-SWEEP form (r06) ---------------------------------------------------
// For matrices whose far structure is pure (mi_csr::win_far_pure = D: every entry is within +-128 rows of its row, or at
// exactly row + D / row - D -- a 3-D stencil on an x-fastest grid, D = one grid plane).  The window form above walks a
// workgroup through CONSECUTIVE tiles, keeps the near rows in an LDS ring and gathers the two far rows of every row from
// global memory: 3.25 loads per output element, and the far rows miss the XCD's L2 in 40 % of the cases (1713 MB read at
// the fabric for 1104 algorithmic on cfg5's 48-column product; with the far rows read from lines that are in the L2
// anyway the same kernel takes 408 us instead of 450, without them 350: profiles/r06_spmm_far_ablation.txt).
// Here a workgroup of 4 waves owns a TILE of kSwRows = 512 consecutive rows and walks it through the rows
// base + j D, j = j0 ... j1 - 1: the rows at +D of step j ARE the tile's own rows of step j + 1 and the rows at -D those of
// step j - 1, so every row of X is loaded ONCE as "the next plane's rows" (two steps ahead, into registers), serves as
// the far operand, then as the own rows (written to LDS from the registers), then as the -D operand (read back from
// LDS): no far gather exists.  What is re-read is the near HALO, 128 rows either side of the tile per step (one row per
// thread, prefetched a step ahead; the neighbouring tiles' workgroups are in the same z-range on the same XCD).
// 1.5 loads per output element.  Entries are decoded from the packed words (mi_csr::pk: (col - row) << 8 | value index)
// in storage order with the same fused multiply-adds as k_spmm_colmajor_pk: the same bits.
// (waves x row groups per wave: 4 x 2 -- 230-245 VGPRs, two workgroups = 8 waves per CU, 496-514 us at cfg5 -- or, with
// -DMI_SWEEP_WAVES=8, 8 x 1: the same tile and window, 128 VGPRs, 16 waves per CU: 518-521 us.  Twice the occupancy
// changes nothing: with every workgroup's step requested in one burst the pass is paced by the memory system serving
// 512 workgroups x 96 KB per round, at 4.3-4.5 TB/s)
constexpr int kSwRows = 512, kSwHalo = 128, kSwWin = kSwRows + 2 * kSwHalo;
#ifndef MI_SWEEP_WAVES
#define MI_SWEEP_WAVES 4
#endif
constexpr int kSwWaves = MI_SWEEP_WAVES, kSwBlock = 64 * kSwWaves, kSwGroups = kSwRows / 64 / kSwWaves;
template <int KC, int HW, bool RES>
__global__ __launch_bounds__(kSwBlock) __attribute__((amdgpu_waves_per_eu(kSwWaves == 8 ? 4 : 2, kSwWaves == 8 ? 4 : 2))) void k_spmm_colmajor_sweep(SellView A, unsigned D, int tiles, int zs, int k,
                                                                   int c_first, ColBlocks X, double *__restrict__ Y,
                                                                   ThetaArg theta, double *__restrict__ R,
                                                                   double *__restrict__ partials_all) {
  static_assert(kSwGroups >= 1 && kSwGroups * kSwWaves * 64 == kSwRows && 2 * kSwHalo == 256, "thread <-> halo row, wave <-> groups");
  const int c0 = c_first + (int)blockIdx.y * KC;
  double *const partials = RES ? partials_all + (size_t)blockIdx.y * (2 * KC) * kMaxRows : nullptr;
  extern __shared__ __attribute__((aligned(16))) double ring_dyn[];  // KC x (kSwWin + 1)
  __shared__ double vt[256];
  __shared__ double red[RES ? 2 * KC * kSwWaves : 1];
  constexpr unsigned RR = (unsigned)kSwWin + 1u;
  LdsDouble *const L = (LdsDouble *)ring_dyn;
  if (threadIdx.x < 256) vt[threadIdx.x] = A.vtab[threadIdx.x];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int zseg = (int)lb / tiles, tile = (int)lb % tiles;
  const long long m = (long long)A.n;
  const int nst = (int)((m + (long long)D - 1) / (long long)D);
  const int j0 = zseg * zs, j1 = j0 + zs < nst ? j0 + zs : nst;
  double nr[RES ? KC : 1], nxs[RES ? KC : 1], th[RES ? KC : 1];
  if (RES) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      nr[c] = 0;
      nxs[c] = 0;
      th[c] = (c0 + c < k) ? theta.v[c0 + c] : 0.0;
    }
  }
  const double *xc[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) xc[c] = X.col(std::min(c0 + c, k - 1), (size_t)m);
  // in-plane offset of this lane's row of group i, and whether it is a row of THIS tile (the last tile of a plane is cut)
  unsigned loc[kSwGroups];
  bool mine[kSwGroups];
#pragma unroll
  for (int i = 0; i < kSwGroups; ++i) {
    loc[i] = (unsigned)(64 * (w + kSwWaves * i) + lane);
    mine[i] = (unsigned)tile * (unsigned)kSwRows + loc[i] < D;
  }
  const long long base = (long long)tile * kSwRows;
  auto rows_load = [&](int j, double (&buf)[kSwGroups][KC]) {   // the tile's rows of step j (clamped: unused when outside)
#pragma unroll
    for (int i = 0; i < kSwGroups; ++i) {
      const long long r = base + (long long)j * (long long)D + (long long)loc[i];
      const size_t rr = (r >= 0 && r < m) ? (size_t)r : 0;
#pragma unroll
      for (int c = 0; c < KC; ++c) buf[i][c] = pinned_load(xc[c] + rr);
    }
  };
  // one halo row per thread (128 below the tile, 128 above); 512 threads: the two halves of the workgroup take half the
  // columns of a row each
  constexpr int HP = kSwBlock / 256, HC = KC / HP;
  static_assert(KC % HP == 0, "halo columns per thread");
  const int hrow = (int)(threadIdx.x & 255), hc0 = (int)(threadIdx.x >> 8) * HC;
  auto halo_load = [&](int j, double (&buf)[HC]) {
    const long long r = base + (long long)j * (long long)D +
                        (hrow < kSwHalo ? (long long)hrow - kSwHalo : (long long)kSwRows + (hrow - kSwHalo));
    const size_t rr = (r >= 0 && r < m) ? (size_t)r : 0;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      const double *col = xc[0];   // (xc[hc0 + c] with a runtime hc0: selected below, the array stays in registers)
#pragma unroll
      for (int cc = 0; cc < KC; ++cc) col = (cc == hc0 + c) ? xc[cc] : col;
      buf[c] = pinned_load(col + rr);
    }
  };
  struct Words {
    unsigned wv[HW];
    int width;
  };
  // The words of a step in TWO stages a step apart, so that no load of a step waits for another load of the same step
  // (loads return in order: a dependent pair at the top of a step drains everything issued before it, i.e. exposes the
  // whole memory latency once per step -- measured: 755 -> 540 us with two workgroups per CU, still 11 us per step):
  // stage 1 the slice bounds of the rows, stage 2 -- a step later -- the packed words themselves.
  struct Where {
    unsigned b0, lane_in_slice;
    int width;
  };
  auto where_load = [&](int j, Where (&wh)[kSwGroups]) {
#pragma unroll
    for (int i = 0; i < kSwGroups; ++i) {
      const long long r = base + (long long)j * (long long)D + (long long)loc[i];
      const bool ok = mine[i] && r < m && j < j1;
      const size_t rr = ok ? (size_t)r : 0, sl = rr >> 6;
      const long long b0 = pinned_load(A.slice_ptr + sl), b1 = pinned_load(A.slice_ptr + sl + 1);
      wh[i].b0 = (unsigned)b0;
      wh[i].width = ok ? (int)(b1 - b0) : 0;
      wh[i].lane_in_slice = (unsigned)(rr & 63);
    }
  };
  auto words_load = [&](const Where (&wh)[kSwGroups], Words (&wd)[kSwGroups]) {
#pragma unroll
    for (int i = 0; i < kSwGroups; ++i) {
      wd[i].width = wh[i].width;
#pragma unroll
      for (int e = 0; e < HW; ++e)
        wd[i].wv[e] = (e < wh[i].width) ? pinned_load(A.pk + ((size_t)(wh[i].b0 + (unsigned)e) * 64 + wh[i].lane_in_slice)) : 0u;
    }
  };
  if (j0 >= j1) {
    if constexpr (RES) {
      double z[2 * KC];
#pragma unroll
      for (int c = 0; c < 2 * KC; ++c) z[c] = 0;
      block_partials_store_nw<2 * KC, kSwWaves>(z, red, partials);
    }
    return;
  }
  // registers: xprev = the tile's rows of step j - 1, xcur = of step j (only until they are in the window), xnext = of
  // step j + 1; the rows of step j + 2 are in flight during the step
  double xprev[kSwGroups][KC], xcur[kSwGroups][KC], xnext[kSwGroups][KC], hal[HC];
  Words wd[kSwGroups];
  Where wh[kSwGroups];
  where_load(j0, wh);
  rows_load(j0 - 1, xprev);
  rows_load(j0, xcur);
  rows_load(j0 + 1, xnext);
  halo_load(j0, hal);
  words_load(wh, wd);
  where_load(j0 + 1, wh);
  for (int j = j0; j < j1; ++j) {
    // the tile's own rows and its halo into the window (the previous step's readers are past their barrier)
#pragma unroll
    for (int i = 0; i < kSwGroups; ++i) {
      LdsDouble *dst = L + (unsigned)kSwHalo + loc[i];
#pragma unroll
      for (int c = 0; c < KC; ++c) dst[(unsigned)c * RR] = xcur[i][c];
    }
    {
      LdsDouble *dst = L + (hrow < kSwHalo ? (unsigned)hrow : (unsigned)kSwRows + (unsigned)hrow) + (unsigned)hc0 * RR;
#pragma unroll
      for (int c = 0; c < HC; ++c) dst[(unsigned)c * RR] = hal[c];
    }
    // two steps ahead: the rows that are this step's +2D (into the registers the own rows just left); next step's halo
    // and words
    const bool more = j + 1 < j1;
    Words wn[kSwGroups];
    words_load(wh, wn);          // step j + 1's words (their bounds came in during the previous step)
    where_load(j + 2, wh);
    rows_load(j + 2, xcur);
    halo_load(more ? j + 1 : j, hal);
    lds_barrier();
#pragma unroll
    for (int i = 0; i < kSwGroups; ++i) {
      double acc[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) acc[c] = 0;
      const unsigned long long all = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
      for (int e = 0; e < HW; ++e) {
        const unsigned word = wd[i].wv[e];
        const bool live = e < wd[i].width;
        const double a = vt[word & 255u];
        const int d = (int)word >> 8;
        const bool up = d == (int)D, dn = d == -(int)D;
        const unsigned long long mup = __builtin_amdgcn_ballot_w64(up), mdn = __builtin_amdgcn_ballot_w64(dn),
                                 mlv = __builtin_amdgcn_ballot_w64(live);
        if (mlv == 0) continue;
        if (mup == 0 && mdn == 0 && mlv == all) {
          const LdsDouble *l = L + (unsigned)((int)kSwHalo + (int)loc[i] + d);
#pragma unroll
          for (int c = 0; c < KC; ++c) acc[c] = __builtin_fma(a, l[(unsigned)c * RR], acc[c]);
        } else if (mup == all && mlv == all) {
#pragma unroll
          for (int c = 0; c < KC; ++c) acc[c] = __builtin_fma(a, xnext[i][c], acc[c]);
        } else if (mdn == all && mlv == all) {
#pragma unroll
          for (int c = 0; c < KC; ++c) acc[c] = __builtin_fma(a, xprev[i][c], acc[c]);
        } else {
          const LdsDouble *l = L + (unsigned)((int)kSwHalo + (int)loc[i] + ((up || dn || !live) ? 0 : d));
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            double v = l[(unsigned)c * RR];
            v = up ? xnext[i][c] : (dn ? xprev[i][c] : v);
            const double t = __builtin_fma(a, v, acc[c]);
            acc[c] = live ? t : acc[c];
          }
        }
      }
      // the own rows come back from the window: they are the next step's -D operand (and the residual's X)
      {
        const LdsDouble *own = L + (unsigned)kSwHalo + loc[i];
#pragma unroll
        for (int c = 0; c < KC; ++c) xprev[i][c] = own[(unsigned)c * RR];
      }
      const long long r = base + (long long)j * (long long)D + (long long)loc[i];
      if (mine[i] && r < m) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
          if (c0 + c < k) __builtin_nontemporal_store(acc[c], Y + (size_t)(c0 + c) * (size_t)m + (size_t)r);
        if constexpr (RES) {
#pragma unroll
          for (int c = 0; c < KC; ++c)
            if (c0 + c < k) {
              const double xv = xprev[i][c];
              const double res = __builtin_fma(-xv, th[c], acc[c]);
              R[(size_t)(c0 + c) * (size_t)m + (size_t)r] = res;
              nr[c] = __builtin_fma(res, res, nr[c]);
              nxs[c] = __builtin_fma(xv, xv, nxs[c]);
            }
        }
      }
    }
    // rotate: (own -> prev happened above) next -> cur, the rows loaded at the top of this step -> next
#pragma unroll
    for (int i = 0; i < kSwGroups; ++i) {
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const double t = xnext[i][c];
        xnext[i][c] = xcur[i][c];
        xcur[i][c] = t;
      }
      wd[i] = wn[i];
    }
    if (more) lds_barrier();
  }
  if constexpr (RES) {
    double a[2 * KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      a[c] = nr[c];
      a[KC + c] = nxs[c];
    }
    block_partials_store_nw<2 * KC, kSwWaves>(a, red, partials);
  }
}

// Y[r, c] = d[r] X[r, c]  (diagonal operators of the reference's LOBPCG tests, tests/LOBPCG_unit_test.cpp:56-74)
__global__ __launch_bounds__(256) void k_rowscale(size_t m, size_t k, const double *__restrict__ d,
                                                  const double *__restrict__ X, double *__restrict__ Y) {
  const size_t total = m * k, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) Y[i] = d[i % m] * X[i];
}

// Row-sharded matrices: kc <= 4 columns of a column-major panel <-> a row-major m x kc field, the layout whose halo
// exchange and SpMM kernels exist (sparse.hip, comm.hip)
__global__ __launch_bounds__(256) void k_cols_to_rows(size_t m, int kc, const double *__restrict__ X,
                                                      double *__restrict__ V) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
    for (int j = 0; j < kc; ++j) V[i * kc + j] = X[i + (size_t)j * m];
}
__global__ __launch_bounds__(256) void k_rows_to_cols(size_t m, int kc, const double *__restrict__ V,
                                                      double *__restrict__ Y) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
    for (int j = 0; j < kc; ++j) Y[i + (size_t)j * m] = V[i * kc + j];
}

// (the host Rayleigh-Ritz solver lives in Optimization/LinearAlgebra/DenseSymmetricEigen.h)
int check_panel(mi_ctx *ctx, size_t m, int k, const mi_vec *P, const char *what) {
  MI_REQUIRE(P, "%s is null", what);
  MI_REQUIRE(P->ctx == ctx, "%s belongs to another context", what);
  MI_REQUIRE(P->n >= m * (size_t)k, "%s holds %zu doubles, needs m*k = %zu*%d", what, P->n, m, k);
  return MI_OK;
}

}  // namespace

extern "C" {

// ---- panels held as column blocks (mi_panel_blocks, include/mi355opt.h) ----------------------------------------------
static ColBlocks one_block(const double *d, int k) { return ColBlocks{{d, d, d}, k, k}; }
// validated kernel-side description of a mi_panel_blocks and its total width
static int blocks_to_cols(mi_ctx *ctx, size_t m, const mi_panel_blocks *B, ColBlocks *cb, int *k) {
  MI_REQUIRE(B && B->nblocks >= 1 && B->nblocks <= 3, "a panel has 1 to 3 column blocks");
  int total = 0;
  for (int i = 0; i < B->nblocks; ++i) {
    MI_REQUIRE(B->cols[i] >= 1, "column block %d is empty", i);
    MI_TRY(check_panel(ctx, m, B->cols[i], B->block[i], "column block"));
    total += B->cols[i];
  }
  MI_REQUIRE(total <= kGramMaxK, "panel width must be in [1,%d]", kGramMaxK);
  const double *b0 = B->block[0]->d, *b1 = B->nblocks > 1 ? B->block[1]->d : b0,
               *b2 = B->nblocks > 2 ? B->block[2]->d : b1;
  const int n0 = B->cols[0], n01 = n0 + (B->nblocks > 1 ? B->cols[1] : 0);
  *cb = ColBlocks{{b0, b1, b2}, n0, B->nblocks > 2 ? n01 : total};
  if (B->nblocks == 1) cb->n0 = cb->n01 = total;
  *k = total;
  return MI_OK;
}
// the blocks copied together into one panel (the fall-back of every *_blocks entry point whose fast kernel does not
// take the shape at hand); the caller destroys *out
static int materialize_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *B, mi_vec **out) {
  int k = 0;
  for (int i = 0; i < B->nblocks; ++i) k += B->cols[i];
  MI_TRY(mi_vec_create(ctx, m * (size_t)k, out));
  size_t off = 0;
  for (int i = 0; i < B->nblocks; ++i) {
    const size_t cnt = m * (size_t)B->cols[i];
    const hipError_t e = hipMemcpyAsync((*out)->d + off, B->block[i]->d, cnt * sizeof(double), hipMemcpyDeviceToDevice,
                                        ctx->stream);
    if (e != hipSuccess) {
      mi_vec_destroy(*out);
      *out = nullptr;
      return hip_fail(e, "panel blocks copy", __FILE__, __LINE__);
    }
    off += cnt;
  }
  return MI_OK;
}

// a Gram whose kernels are enqueued but whose result has not been read back yet (gram_impl with job != null)
struct GramJob {
  void *partial = nullptr, *Gdev = nullptr;
  int nelem = 0;
};
static int gram_finish(mi_ctx *ctx, GramJob *jobs, int njobs, double *const *G_host) {
  const void *dev[4];
  size_t bytes[4];
  void *host[4];
  for (int i = 0; i < njobs && i < 4; ++i) {
    dev[i] = jobs[i].Gdev;
    bytes[i] = (size_t)jobs[i].nelem * sizeof(double);
    host[i] = G_host[i];
  }
  const int st = njobs > 0 ? readback_sync(ctx, njobs, dev, bytes, host) : MI_OK;  // (pinned landing area: no blit kernel)
  for (int i = 0; i < njobs; ++i) {
    pool_free(ctx, jobs[i].partial);
    pool_free(ctx, jobs[i].Gdev);
  }
  return st;
}

// T = [T (k1 columns) | T2 (kb - k1 columns)] when T2 != null (square direct shapes only), else T alone.
// job != null: only enqueue; the caller reads back with gram_finish.
static int gram_impl(mi_ctx *ctx, size_t m, int ka, int kb, const mi_vec *S, const mi_vec *T, const mi_vec *T2v,
                     int k1, double *G_host, GramJob *job = nullptr) {
  const double *T2 = T2v ? T2v->d : nullptr;
  const bool same = !T2 && (S->d == T->d) && ka == kb;
  const bool aligned = (m % 2 == 0) && ((uintptr_t)S->d % 16 == 0) && ((uintptr_t)T->d % 16 == 0);
  const int nelem = ka * kb;
  const int kapad = (ka + 15) / 16 * 16, kbpad = (kb + 15) / 16 * 16;
  // square panels of up to 80 columns on 32-byte-aligned columns: the LDS-free one-wave-per-row-range kernel
  const bool direct = ka == kb && ka <= 80 && m % 4 == 0 && (uintptr_t)S->d % 32 == 0 && (uintptr_t)T->d % 32 == 0 &&
                      (!T2 || (uintptr_t)T2 % 32 == 0) && m >= 16 * kGdH;
  MI_REQUIRE(!T2 || direct, "split panels need the direct Gram kernel");
  size_t nb, nwaves = 0;
  const size_t mfull = m - m % (16 * kGdH);  // rows k_gram_direct covers in whole pipeline steps
  if (direct) {  // one partial per wave + one for the leftover rows
    // two waves per SIMD where the registers allow it (k <= 48, and k <= 80 when S == T), else one
    const int occ = (kapad <= 48 || same) ? 2 : 1;
    nwaves = (std::min<size_t>(4 * (size_t)occ * ctx->num_cu, mfull / (16 * kGdH)) + 3) / 4 * 4;
    nb = nwaves + (mfull < m ? 1 : 0);
  } else {  // rows per workgroup: a multiple of the 32-row tile, ~3 workgroups per CU
    nb = std::min<size_t>(3 * (size_t)ctx->num_cu, (m + kGramRows - 1) / kGramRows);
    if (nb < 1) nb = 1;
  }
  void *partial = nullptr, *Gdev = nullptr;
  MI_TRY(pool_alloc(ctx, nb * (size_t)nelem * sizeof(double), &partial));
  MI_TRY(pool_alloc(ctx, (size_t)nelem * sizeof(double), &Gdev));
  const size_t lds = (size_t)(same ? kapad : kapad + kbpad) * kGramLd * sizeof(double);
  if (direct) {
    KScope ks(ctx, MI_K_LOBPCG_GRAM);
    if (mfull < m)
      hipLaunchKernelGGL(k_gram_tail, dim3((nelem + 255) / 256), dim3(256), 0, ctx->stream, m, mfull, ka, kb,
                         (const double *)S->d, (const double *)T->d, T2, k1,
                         (double *)partial + (nb - 1) * (size_t)nelem);
#define GD(TT, SAME, NS)                                                                                  \
  hipLaunchKernelGGL((k_gram_direct<TT, SAME>), dim3((unsigned)(nwaves / 4)), dim3(256), 0, ctx->stream, \
                     mfull, m, ka, (const double *)S->d, (const double *)T->d, T2, k1, (double *)partial)
#define GD_T(SAME)                   \
  switch (kapad / 16) {              \
    case 1: GD(1, SAME, 1); break;   \
    case 2: GD(2, SAME, 1); break;   \
    case 3: GD(3, SAME, 1); break;   \
    case 4: GD(4, SAME, 1); break;   \
    default: GD(5, SAME, 1); break;  \
  }
    if (same) {
      GD_T(true)
    } else {
      GD_T(false)
    }
#undef GD_T
#undef GD
  } else {
    KScope ks(ctx, MI_K_LOBPCG_GRAM);
    const int ta = kapad / 16, tb = kbpad / 16;
    const int ntiles = same ? ta * (ta + 1) / 2 : ta * tb;
    const int tpw = (ntiles + kGramWaves - 1) / kGramWaves;  // 1..5
#define GRAM(SAME, AL, TPW)                                                                                        \
  hipLaunchKernelGGL((k_gram<SAME, AL, TPW>), dim3((unsigned)nb), dim3(kGramThreads), lds, ctx->stream, m, ka, kb, \
                     (const double *)S->d, (const double *)T->d, (double *)partial)
#define GRAM_T(SAME, AL)        \
  switch (tpw) {                \
    case 1: GRAM(SAME, AL, 1); break; \
    case 2: GRAM(SAME, AL, 2); break; \
    case 3: GRAM(SAME, AL, 3); break; \
    case 4: GRAM(SAME, AL, 4); break; \
    default: GRAM(SAME, AL, 5); break; \
  }
    if (same) {
      if (aligned) {
        GRAM_T(true, true)
      } else {
        GRAM_T(true, false)
      }
    } else {
      if (aligned) {
        GRAM_T(false, true)
      } else {
        GRAM_T(false, false)
      }
    }
#undef GRAM_T
#undef GRAM
  }
  hipLaunchKernelGGL(k_gram_reduce, dim3((nelem + kRedElems - 1) / kRedElems), dim3(kRedElems * kRedGroups), 0,
                     ctx->stream, (int)nb, ka, nelem, (int)same, (const double *)partial, (double *)Gdev);
  if (ctx->comm) {
    for (int off = 0; off < nelem; off += 4096)  // all-reduce in chunks the comm layer accepts
      MI_TRY(comm_allreduce(ctx, (double *)Gdev + off, std::min(4096, nelem - off)));
  }
  GramJob mine{partial, Gdev, nelem};
  if (job) {
    *job = mine;
    return MI_OK;
  }
  double *dst[1] = {G_host};
  return gram_finish(ctx, &mine, 1, dst);
}

// assembles T = [T1 | T2] in a fresh panel (for shapes the split kernel does not take)
static int assemble_panel(mi_ctx *ctx, size_t m, int k, int k1, const mi_vec *T1, const mi_vec *T2, mi_vec **out) {
  MI_TRY(mi_vec_create(ctx, m * (size_t)k, out));
  hipError_t e = hipMemcpyAsync((*out)->d, T1->d, m * (size_t)k1 * sizeof(double), hipMemcpyDeviceToDevice,
                                ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync((*out)->d + m * (size_t)k1, T2->d, m * (size_t)(k - k1) * sizeof(double),
                       hipMemcpyDeviceToDevice, ctx->stream);
  if (e != hipSuccess) {
    mi_vec_destroy(*out);
    *out = nullptr;
    return hip_fail(e, "panel assembly", __FILE__, __LINE__);
  }
  return MI_OK;
}
static bool split_direct_ok(size_t m, int k, const mi_vec *S, const mi_vec *T1, const mi_vec *T2) {
  return k <= 80 && m % 4 == 0 && (uintptr_t)S->d % 32 == 0 && (uintptr_t)T1->d % 32 == 0 &&
         (uintptr_t)T2->d % 32 == 0 && m >= 16 * kGdH;
}

// The two Grams of a Rayleigh-Ritz step with ONE synchronisation: both are enqueued back to back (the device goes
// from the first straight into the second), then both results are read back.  Each right-hand panel is either one
// panel of k columns (T?2 == null) or two pieces [T?1 (k1? columns) | T?2].
int mi_lobpcg_gram_pair(mi_ctx *ctx, size_t m, int k, const mi_vec *S, int k1a, const mi_vec *Ta1, const mi_vec *Ta2,
                        int k1b, const mi_vec *Tb1, const mi_vec *Tb2, double *Ga_host, double *Gb_host) {
  MI_REQUIRE(ctx && S && Ta1 && Tb1 && Ga_host && Gb_host, "null argument");
  MI_REQUIRE(k >= 1 && k <= kGramMaxK, "panel width must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(ctx, m, k, S, "S"));
  const mi_vec *T1[2] = {Ta1, Tb1}, *T2[2] = {Ta2, Tb2};
  const int k1[2] = {k1a, k1b};
  GramJob jobs[2];
  mi_vec *tmp[2] = {nullptr, nullptr};
  int st = MI_OK, started = 0;
  for (int i = 0; i < 2 && st == MI_OK; ++i) {
    if (T2[i]) {
      if (!(k1[i] >= 1 && k1[i] < k)) { set_error("need 1 <= k1 < k"); st = MI_ERR_INVALID_ARGUMENT; break; }
      st = check_panel(ctx, m, k1[i], T1[i], "T1");
      if (st == MI_OK) st = check_panel(ctx, m, k - k1[i], T2[i], "T2");
      if (st != MI_OK) break;
      if (split_direct_ok(m, k, S, T1[i], T2[i])) {
        st = gram_impl(ctx, m, k, k, S, T1[i], T2[i], k1[i], nullptr, &jobs[i]);
      } else {
        st = assemble_panel(ctx, m, k, k1[i], T1[i], T2[i], &tmp[i]);
        if (st == MI_OK) st = gram_impl(ctx, m, k, k, S, tmp[i], nullptr, k, nullptr, &jobs[i]);
      }
    } else {
      st = check_panel(ctx, m, k, T1[i], "T");
      if (st == MI_OK) st = gram_impl(ctx, m, k, k, S, T1[i], nullptr, k, nullptr, &jobs[i]);
    }
    if (st == MI_OK) ++started;
  }
  double *dst[2] = {Ga_host, Gb_host};
  const int fin = gram_finish(ctx, jobs, started, dst);  // (also releases what was enqueued when a later step failed)
  for (int i = 0; i < 2; ++i) mi_vec_destroy(tmp[i]);
  return st != MI_OK ? st : fin;
}

// G_A = S' [Ta1 | Ta2] for a SYMMETRIC product (T = A S with a symmetric operator: G_A is symmetric and only its upper
// block triangle is formed, the mirror image filled in) and G_B = S' S, both from ONE pass over S and T
// (k_gram_pair_sym); one synchronisation.  Ta2 may be null (one panel of k columns).  Shapes the fused kernel does not
// take (k > 80, unaligned panels, fewer than 16 rows) go through mi_lobpcg_gram_pair, which forms all of G_A.
static bool gram_pair_sym_direct_ok(size_t m, int k, const ColBlocks &S, const mi_vec *Ta1, const mi_vec *Ta2);
static int gram_pair_sym_direct(mi_ctx *ctx, size_t m, int k, const ColBlocks &Sb, int k1a, const mi_vec *Ta1,
                                const mi_vec *Ta2, double *Ga_host, double *Gb_host,
                                const ColBlocks *Tab = nullptr, const ColBlocks *Tbb = nullptr);

int mi_lobpcg_gram_pair_sym(mi_ctx *ctx, size_t m, int k, const mi_vec *S, int k1a, const mi_vec *Ta1,
                            const mi_vec *Ta2, double *Ga_host, double *Gb_host) {
  MI_REQUIRE(ctx && S && Ta1 && Ga_host && Gb_host, "null argument");
  MI_REQUIRE(k >= 1 && k <= kGramMaxK, "panel width must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(ctx, m, k, S, "S"));
  if (Ta2) {
    MI_REQUIRE(k1a >= 1 && k1a < k, "need 1 <= k1 < k");
    MI_TRY(check_panel(ctx, m, k1a, Ta1, "T1"));
    MI_TRY(check_panel(ctx, m, k - k1a, Ta2, "T2"));
  } else {
    MI_TRY(check_panel(ctx, m, k, Ta1, "T"));
    k1a = k;
  }
  if (!gram_pair_sym_direct_ok(m, k, one_block(S->d, k), Ta1, Ta2))
    return mi_lobpcg_gram_pair(ctx, m, k, S, k1a, Ta1, Ta2, k, S, nullptr, Ga_host, Gb_host);
  return gram_pair_sym_direct(ctx, m, k, one_block(S->d, k), k1a, Ta1, Ta2, Ga_host, Gb_host);
}

int mi_lobpcg_gram_pair_sym_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, int k1a, const mi_vec *Ta1,
                                   const mi_vec *Ta2, double *Ga_host, double *Gb_host) {
  MI_REQUIRE(ctx && S && Ta1 && Ga_host && Gb_host, "null argument");
  ColBlocks cb;
  int k = 0;
  MI_TRY(blocks_to_cols(ctx, m, S, &cb, &k));
  if (Ta2) {
    MI_REQUIRE(k1a >= 1 && k1a < k, "need 1 <= k1 < k");
    MI_TRY(check_panel(ctx, m, k1a, Ta1, "T1"));
    MI_TRY(check_panel(ctx, m, k - k1a, Ta2, "T2"));
  } else {
    MI_TRY(check_panel(ctx, m, k, Ta1, "T"));
    k1a = k;
  }
  if (gram_pair_sym_direct_ok(m, k, cb, Ta1, Ta2))
    return gram_pair_sym_direct(ctx, m, k, cb, k1a, Ta1, Ta2, Ga_host, Gb_host);
  mi_vec *tmp = nullptr;  // shapes the one-pass kernel does not take: the blocks copied together, the general path
  MI_TRY(materialize_blocks(ctx, m, S, &tmp));
  const int st = mi_lobpcg_gram_pair_sym(ctx, m, k, tmp, k1a, Ta1, Ta2, Ga_host, Gb_host);
  mi_vec_destroy(tmp);
  return st;
}

// (S'A(S), S'S) like mi_lobpcg_gram_pair_sym_blocks with T = A(S) held as column blocks too: what a plain-callable
// operator applied to the blocks of S one by one leaves ([AX | A(W(:, nc:)) | A(P(:, nc:))]).
int mi_lobpcg_gram_pair_sym_tblocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, const mi_panel_blocks *T,
                                    double *Ga_host, double *Gb_host) {
  MI_REQUIRE(ctx && S && T && Ga_host && Gb_host, "null argument");
  ColBlocks cs, ct;
  int k = 0, kt = 0;
  MI_TRY(blocks_to_cols(ctx, m, S, &cs, &k));
  MI_TRY(blocks_to_cols(ctx, m, T, &ct, &kt));
  MI_REQUIRE(kt == k, "A(S) must have the width of S (%d): got %d", k, kt);
  auto aligned = [](const ColBlocks &c) {
    return (uintptr_t)c.base[0] % 32 == 0 && (uintptr_t)c.base[1] % 32 == 0 && (uintptr_t)c.base[2] % 32 == 0;
  };
  if (k <= 80 && m % 4 == 0 && m >= 16 && aligned(cs) && aligned(ct))
    return gram_pair_sym_direct(ctx, m, k, cs, k, nullptr, nullptr, Ga_host, Gb_host, &ct, nullptr);
  mi_vec *s = nullptr, *t = nullptr;
  int st = materialize_blocks(ctx, m, S, &s);
  if (st == MI_OK) st = materialize_blocks(ctx, m, T, &t);
  if (st == MI_OK) st = mi_lobpcg_gram_pair_sym(ctx, m, k, s, k, t, nullptr, Ga_host, Gb_host);
  mi_vec_destroy(s);
  mi_vec_destroy(t);
  return st;
}

// The generalized problem (B present, LOBPCG.h:268,272): upper block triangles of S'A(S) and S'B(S) with S, A(S) and
// B(S) each held as column blocks of the same total width.
int mi_lobpcg_gram_pair_gen_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, const mi_panel_blocks *AS,
                                   const mi_panel_blocks *BS, double *Ga_host, double *Gb_host) {
  MI_REQUIRE(ctx && S && AS && BS && Ga_host && Gb_host, "null argument");
  ColBlocks cs, ca, cb;
  int k = 0, ka = 0, kb = 0;
  MI_TRY(blocks_to_cols(ctx, m, S, &cs, &k));
  MI_TRY(blocks_to_cols(ctx, m, AS, &ca, &ka));
  MI_TRY(blocks_to_cols(ctx, m, BS, &cb, &kb));
  MI_REQUIRE(ka == k && kb == k, "A(S) and B(S) must have the width of S (%d): got %d and %d", k, ka, kb);
  auto aligned = [](const ColBlocks &c) {
    return (uintptr_t)c.base[0] % 32 == 0 && (uintptr_t)c.base[1] % 32 == 0 && (uintptr_t)c.base[2] % 32 == 0;
  };
  if (k <= 80 && m % 4 == 0 && m >= 16 && aligned(cs) && aligned(ca) && aligned(cb))
    return gram_pair_sym_direct(ctx, m, k, cs, k, nullptr, nullptr, Ga_host, Gb_host, &ca, &cb);
  // shapes the one-pass kernels do not take: everything copied together, the general pair
  mi_vec *s = nullptr, *a = nullptr, *b = nullptr;
  int st = materialize_blocks(ctx, m, S, &s);
  if (st == MI_OK) st = materialize_blocks(ctx, m, AS, &a);
  if (st == MI_OK) st = materialize_blocks(ctx, m, BS, &b);
  if (st == MI_OK) st = mi_lobpcg_gram_pair(ctx, m, k, s, k, a, nullptr, k, b, nullptr, Ga_host, Gb_host);
  mi_vec_destroy(s);
  mi_vec_destroy(a);
  mi_vec_destroy(b);
  return st;
}

static bool gram_pair_sym_direct_ok(size_t m, int k, const ColBlocks &S, const mi_vec *Ta1, const mi_vec *Ta2) {
  return k <= 80 && m % 4 == 0 && (uintptr_t)S.base[0] % 32 == 0 && (uintptr_t)S.base[1] % 32 == 0 &&
         (uintptr_t)S.base[2] % 32 == 0 && (uintptr_t)Ta1->d % 32 == 0 && (!Ta2 || (uintptr_t)Ta2->d % 32 == 0) && m >= 16;
}

// S'[Ta1 | Ta2] and S'S, upper block triangles, one pass (k_gram_pair_sym); arguments validated by the callers
// Tab / Tbb (r05, the generalized problem): T = A(S) and B(S) as column blocks; the second Gram is then S'B(S) instead
// of S'S, each Gram one launch of k_gram_pair_sym<., false, PAIR = false> (Ta1 / Ta2 unused)
static int gram_pair_sym_direct(mi_ctx *ctx, size_t m, int k, const ColBlocks &Sb, int k1a, const mi_vec *Ta1,
                                const mi_vec *Ta2, double *Ga_host, double *Gb_host, const ColBlocks *Tab,
                                const ColBlocks *Tbb) {
  const int nelem = k * k, kpad = (k + 15) / 16 * 16;
  const size_t mfull = m - m % 16;
  const int occ = kpad <= 32 ? 2 : 1;  // (resident waves per SIMD: 62+16 / 132+48 registers at 1 / 2 tiles, then > 256)
  const size_t nwaves = (std::min<size_t>(4 * (size_t)occ * ctx->num_cu, mfull / 16) + 3) / 4 * 4;
  const size_t nb = nwaves;  // (the leftover rows are a step of one of the waves)
  GramJob jobs[2] = {{nullptr, nullptr, nelem}, {nullptr, nullptr, nelem}};
  int st = MI_OK;
  // One rank: the reduction kernel writes the two k x k results STRAIGHT into pinned host memory (82 KB over PCIe at
  // ns = 72) -- no device buffer, no copy behind the kernel (a blit kernel of ~15 us per Gram even into pinned memory).
  // Several ranks: the results are all-reduced on the device first.
  const bool zero_copy = ctx->comm == nullptr && !ctx->cfg.no_zero_copy;
  const size_t slot = ((size_t)nelem * sizeof(double) + 255) / 256 * 256;
  char *zc_host = nullptr, *zc_dev = nullptr;
  if (zero_copy) st = readback_area(ctx, 2 * slot, (void **)&zc_host, (void **)&zc_dev);
  for (int i = 0; i < 2 && st == MI_OK; ++i) {
    st = pool_alloc(ctx, nb * (size_t)nelem * sizeof(double), &jobs[i].partial);
    if (st == MI_OK) {
      if (zero_copy) jobs[i].Gdev = zc_dev + (size_t)i * slot;
      else st = pool_alloc(ctx, (size_t)nelem * sizeof(double), &jobs[i].Gdev);
    }
  }
  if (st != MI_OK) {
    for (int i = 0; i < 2; ++i) {
      if (jobs[i].partial) pool_free(ctx, jobs[i].partial);
      if (!zero_copy && jobs[i].Gdev) pool_free(ctx, jobs[i].Gdev);
    }
    return st;
  }
  if (Tab && Tbb) {  // generalized problem: S'A(S) and S'B(S), one single-Gram launch each
    KScope ks(ctx, MI_K_LOBPCG_GRAM);
    for (int i = 0; i < 2; ++i) {
      const ColBlocks &Tc = i == 0 ? *Tab : *Tbb;
#define GS1(TT)                                                                                                      \
  hipLaunchKernelGGL((k_gram_pair_sym<TT, false, false>), dim3((unsigned)(nwaves / 4)), dim3(256), 0, ctx->stream,    \
                     mfull, m, k, Sb, Tc, (double *)jobs[i].partial, (double *)nullptr)
      switch (kpad / 16) {
        case 1: GS1(1); break;
        case 2: GS1(2); break;
        case 3: GS1(3); break;
        case 4: GS1(4); break;
        default: GS1(5); break;
      }
#undef GS1
    }
  } else {
  const double *T2 = Ta2 ? Ta2->d : (Ta1 ? Ta1->d : nullptr);
  const ColBlocks Tcb = Tab ? *Tab : ColBlocks{{Ta1->d, T2, T2}, k1a, k};  // (Tab alone: T = A(S) as column blocks)
  // (the last tile column at most half full: it is shared by the two Grams, k_gram_pair_sym<., HALF>)
  const bool half = k % 16 >= 1 && k % 16 <= 8 && !ctx->cfg.no_gram_half;
  {
    KScope ks(ctx, MI_K_LOBPCG_GRAM);
#define GPS(TT, HF)                                                                                                 \
  hipLaunchKernelGGL((k_gram_pair_sym<TT, HF, true>), dim3((unsigned)(nwaves / 4)), dim3(256), 0, ctx->stream, mfull, \
                     m, k, Sb, Tcb, (double *)jobs[0].partial, (double *)jobs[1].partial)
#define GPSH(TT) \
  if (half) { GPS(TT, true); } else { GPS(TT, false); }
    switch (kpad / 16) {
      case 1: GPSH(1); break;
      case 2: GPSH(2); break;
      case 3: GPSH(3); break;
      case 4: GPSH(4); break;
      default: GPSH(5); break;
    }
#undef GPSH
#undef GPS
  }
  }
  hipLaunchKernelGGL(k_gram_reduce, dim3((nelem + kRedElems - 1) / kRedElems, 2), dim3(kRedElems * kRedGroups), 0,
                     ctx->stream, (int)nb, k, nelem, 1, (const double *)jobs[0].partial, (double *)jobs[0].Gdev,
                     (const double *)jobs[1].partial, (double *)jobs[1].Gdev);
  if (ctx->comm)
    for (int i = 0; i < 2; ++i)
      for (int off = 0; off < nelem && st == MI_OK; off += 4096)
        st = comm_allreduce(ctx, (double *)jobs[i].Gdev + off, std::min(4096, nelem - off));
  double *dst[2] = {Ga_host, Gb_host};
  int fin = MI_OK;
  if (zero_copy) {
    fin = stream_wait(ctx, "gram read-back");
    for (int i = 0; i < 2; ++i) {
      if (fin == MI_OK) memcpy(dst[i], zc_host + (size_t)i * slot, (size_t)nelem * sizeof(double));
      pool_free(ctx, jobs[i].partial);
    }
  } else {
    fin = gram_finish(ctx, jobs, 2, dst);
  }
  if (st == MI_OK) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) st = hip_fail(e, "symmetric Gram pair launch", __FILE__, __LINE__);
  }
  return st != MI_OK ? st : fin;
}

int mi_lobpcg_gram(mi_ctx *ctx, size_t m, int ka, int kb, const mi_vec *S, const mi_vec *T, double *G_host) {
  MI_REQUIRE(ctx && G_host, "null argument");
  MI_REQUIRE(ka >= 1 && ka <= kGramMaxK && kb >= 1 && kb <= kGramMaxK, "panel widths must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(ctx, m, ka, S, "S"));
  MI_TRY(check_panel(ctx, m, kb, T, "T"));
  return gram_impl(ctx, m, ka, kb, S, T, nullptr, kb, G_host);
}

int mi_lobpcg_gram_split(mi_ctx *ctx, size_t m, int k, const mi_vec *S, int k1, const mi_vec *T1, const mi_vec *T2,
                         double *G_host) {
  MI_REQUIRE(ctx && G_host, "null argument");
  MI_REQUIRE(k >= 1 && k <= kGramMaxK && k1 >= 1 && k1 < k, "need 1 <= k1 < k <= %d", kGramMaxK);
  MI_TRY(check_panel(ctx, m, k, S, "S"));
  MI_TRY(check_panel(ctx, m, k1, T1, "T1"));
  MI_TRY(check_panel(ctx, m, k - k1, T2, "T2"));
  // the kernel that takes the two pieces as they lie (square panels of <= 80 columns, 32-byte aligned) ...
  const bool direct = k <= 80 && m % 4 == 0 && (uintptr_t)S->d % 32 == 0 && (uintptr_t)T1->d % 32 == 0 &&
                      (uintptr_t)T2->d % 32 == 0 && m >= 16 * kGdH;
  if (direct) return gram_impl(ctx, m, k, k, S, T1, T2, k1, G_host);
  // ... else T = [T1 | T2] assembled once
  mi_vec *T = nullptr;
  MI_TRY(mi_vec_create(ctx, m * (size_t)k, &T));
  hipError_t e = hipMemcpyAsync(T->d, T1->d, m * (size_t)k1 * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(T->d + m * (size_t)k1, T2->d, m * (size_t)(k - k1) * sizeof(double), hipMemcpyDeviceToDevice,
                       ctx->stream);
  int st = e == hipSuccess ? gram_impl(ctx, m, k, k, S, T, nullptr, k, G_host)
                           : hip_fail(e, "panel assembly", __FILE__, __LINE__);
  mi_vec_destroy(T);
  return st;
}

int mi_lobpcg_update(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S, const double *C_host, int ldc,
                     mi_vec *Y) {
  return mi_lobpcg_update2(ctx, m, ks, kc, S, C_host, ldc, Y, kc, nullptr);
}

static int update2_impl(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S, const mi_panel_blocks *blocks,
                        const double *C_host, int ldc, mi_vec *Y, int k1, mi_vec *Y2);

int mi_lobpcg_update2(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S, const double *C_host, int ldc,
                      mi_vec *Y, int k1, mi_vec *Y2) {
  MI_REQUIRE(ctx && C_host, "null argument");
  MI_REQUIRE(ks >= 1 && ks <= kGramMaxK && kc >= 1 && kc <= kGramMaxK && ldc >= ks, "bad small-matrix shape");
  MI_TRY(check_panel(ctx, m, ks, S, "S"));
  return update2_impl(ctx, m, ks, kc, S, nullptr, C_host, ldc, Y, k1, Y2);
}

int mi_lobpcg_update2_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, int kc, const double *C_host, int ldc,
                             mi_vec *Y, int k1, mi_vec *Y2) {
  MI_REQUIRE(ctx && S && C_host, "null argument");
  ColBlocks cb;
  int ks = 0;
  MI_TRY(blocks_to_cols(ctx, m, S, &cb, &ks));
  MI_REQUIRE(kc >= 1 && kc <= kGramMaxK && ldc >= ks, "bad small-matrix shape");
  return update2_impl(ctx, m, ks, kc, nullptr, S, C_host, ldc, Y, k1, Y2);
}

// S: the basis as one panel, or null with `blocks` (the kernels that do not take blocks then get a copy of them put
// together)
static int update2_impl(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S, const mi_panel_blocks *blocks,
                        const double *C_host, int ldc, mi_vec *Y, int k1, mi_vec *Y2) {
  MI_REQUIRE(k1 >= 1 && k1 <= kc && (k1 == kc || Y2), "bad split of the output columns");
  MI_TRY(check_panel(ctx, m, k1, Y, "Y"));
  touch(Y);
  touch(Y2);
  if (k1 < kc) MI_TRY(check_panel(ctx, m, kc - k1, Y2, "Y2"));
  {  // neither destination may overlap the basis: the update reads S while other rows' results are written
    const double *y0 = Y->d, *y1 = Y->d + (size_t)k1 * m;
    const double *z0 = k1 < kc ? Y2->d : nullptr, *z1 = k1 < kc ? Y2->d + (size_t)(kc - k1) * m : nullptr;
    const int nb = blocks ? blocks->nblocks : 1;
    for (int i = 0; i < nb; ++i) {
      const double *s0 = blocks ? blocks->block[i]->d : S->d;
      const double *s1 = s0 + (size_t)(blocks ? blocks->cols[i] : ks) * m;
      MI_REQUIRE(y1 <= s0 || y0 >= s1, "in-place panel update is not supported");
      if (z0) MI_REQUIRE(z1 <= s0 || z0 >= s1, "in-place panel update is not supported");
    }
    if (z0) MI_REQUIRE(z1 <= y0 || z0 >= y1, "the two destination panels overlap");
  }
  // The matrix-pipe kernel takes the whole-iteration update -- 48 output columns (one chunk) from a basis that cuts into
  // 6..18 steps of four columns, none straddling two blocks (see StepBases); every other shape goes through the VALU
  // kernels, which want the basis in one piece.
  StepBases steps{};
  struct StepCols { int logical[4]; };  // basis column behind each of the step's four coefficient rows, -1: zero row
  std::vector<StepCols> step_cols;
  bool all_mfma = kc > 32 && kc <= 48 && m >= 16 && !ctx->cfg.no_update_mfma;
  if (all_mfma) {
    const int nb = blocks ? blocks->nblocks : 1;
    int start = 0;
    for (int bi = 0; bi < nb && all_mfma; ++bi) {
      const int cb = blocks ? blocks->cols[bi] : ks;
      const double *base = blocks ? blocks->block[bi]->d : S->d;
      if (cb < 4) all_mfma = false;
      for (int j = 0; 4 * j < cb && all_mfma; ++j) {
        const int first = std::min(4 * j, cb - 4);
        if (step_cols.size() == 18) { all_mfma = false; break; }
        steps.p[step_cols.size()] = base + (size_t)first * m;
        StepCols sc;
        for (int q = 0; q < 4; ++q) sc.logical[q] = first + q >= 4 * j ? start + first + q : -1;
        step_cols.push_back(sc);
      }
      start += cb;
    }
    if (step_cols.size() < 6) all_mfma = false;
    for (size_t i = step_cols.size(); i < 18; ++i) steps.p[i] = steps.p[0];
  }
  // chunk plan (widest chunk that fits what is left: one pass over S per chunk) and the chunks' coefficient blocks,
  // each ks x KC row-major by s, zero-padded past kc, packed back to back; matrix-pipe form: 4 K4 x 48, one row per
  // (step, column of the step)
  struct Chunk { int c0, width; size_t off; };
  std::vector<Chunk> chunks;
  size_t total = 0;
  std::vector<double> Ct;
  if (all_mfma) {
    chunks.push_back({0, 48, 0});
    total = step_cols.size() * 4 * 48;
    Ct.assign(total, 0.0);
    for (size_t st_ = 0; st_ < step_cols.size(); ++st_)
      for (int q = 0; q < 4; ++q) {
        const int sidx = step_cols[st_].logical[q];
        if (sidx < 0) continue;
        for (int c = 0; c < kc; ++c) Ct[(st_ * 4 + q) * 48 + c] = C_host[(size_t)c * ldc + sidx];
      }
  } else {
    for (int c0 = 0; c0 < kc;) {
      const int left = kc - c0;
      const int width = left > 32 ? 48 : (left > 16 ? 24 : (left > 8 ? 16 : 8));  // 48 = X and P of one iteration
      chunks.push_back({c0, width, total});
      total += (size_t)ks * width;
      c0 += width;
    }
    Ct.assign(total, 0.0);
    for (const Chunk &ch : chunks)
      for (int s = 0; s < ks; ++s)
        for (int c = 0; c < ch.width && ch.c0 + c < kc; ++c)
          Ct[ch.off + (size_t)s * ch.width + c] = C_host[(size_t)(ch.c0 + c) * ldc + s];
  }
  mi_vec *tmp = nullptr;
  if (!S && !all_mfma) {
    MI_TRY(materialize_blocks(ctx, m, blocks, &tmp));
    S = tmp;
  }
  void *Cdev = nullptr;
  int st = pool_alloc(ctx, total * sizeof(double), &Cdev);
  if (st == MI_OK) {
    if (total * sizeof(double) <= mi_ctx::kStageBytes) {
      st = stage_upload(ctx, Ct.data(), total * sizeof(double), Cdev);  // Ct may die: the bytes are in a pinned slot
    } else {
      hipError_t e = hipMemcpyAsync(Cdev, Ct.data(), total * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // Ct dies with this call
      if (e != hipSuccess) st = hip_fail(e, "coefficient upload", __FILE__, __LINE__);
    }
  }
  if (st != MI_OK) {
    if (Cdev) pool_free(ctx, Cdev);
    if (tmp) mi_vec_destroy(tmp);
    return st;
  }
  const int grid = (int)std::min<size_t>((m + 255) / 256, 2048);
  KScope ksc(ctx, MI_K_LOBPCG_UPDATE);
  for (const Chunk &ch : chunks) {
#define UPD(KC)                                                                                       \
  hipLaunchKernelGGL(k_panel_update<KC>, dim3(grid), dim3(256), 0, ctx->stream, m, ks, (const double *)S->d, \
                     (const double *)Cdev + ch.off, ch.c0, kc, Y->d, k1, Y2 ? Y2->d : (double *)nullptr, r_begin)
    const size_t r_begin = 0;
    if (all_mfma) {  // (the m % 16 leftover rows are the last block of one of the waves)
      // 32-row blocks of two interleaved tiles (16 bytes per lane) when every column start is 16-byte aligned
      bool pair = m % 2 == 0 && !ctx->cfg.no_update_pair && (uintptr_t)Y->d % 16 == 0 && (!Y2 || (uintptr_t)Y2->d % 16 == 0);
      for (size_t i = 0; i < 18 && pair; ++i) pair = (uintptr_t)steps.p[i] % 16 == 0;
      const size_t nblocks = pair ? m / 32 : m / 16;
      const int mgrid = (int)std::min<size_t>((nblocks + 3) / 4, (size_t)2 * ctx->num_cu);
#define UPM(K4)                                                                                                \
  case K4:                                                                                                      \
    if (pair)                                                                                                   \
      hipLaunchKernelGGL(k_panel_update_mfma2<K4>, dim3(mgrid), dim3(256), 0, ctx->stream, nblocks, m, steps,    \
                         (const double *)Cdev + ch.off, kc, Y->d, k1, Y2 ? Y2->d : (double *)nullptr);           \
    else                                                                                                        \
      hipLaunchKernelGGL(k_panel_update_mfma<K4>, dim3(mgrid), dim3(256), 0, ctx->stream, nblocks, m, steps,     \
                         (const double *)Cdev + ch.off, kc, Y->d, k1, Y2 ? Y2->d : (double *)nullptr);           \
    break
      switch ((int)step_cols.size()) {
        UPM(6); UPM(7); UPM(8); UPM(9); UPM(10); UPM(11); UPM(12); UPM(13); UPM(14); UPM(15); UPM(16); UPM(17);
        default: UPM(18);
      }
#undef UPM
      continue;
    }
    switch (ch.width) {
      case 48: UPD(48); break;
      case 24: UPD(24); break;
      case 16: UPD(16); break;
      default: UPD(8); break;
    }
#undef UPD
  }
  {
    const hipError_t e = hipGetLastError();
    st = e == hipSuccess ? MI_OK : hip_fail(e, "panel update launch", __FILE__, __LINE__);
  }
  pool_free(ctx, Cdev);  // stream-ordered reuse: later allocations are enqueued after these kernels
  if (tmp) mi_vec_destroy(tmp);
  return st;
}

int mi_lobpcg_residual(mi_ctx *ctx, size_t m, int nx, const mi_vec *AX, const mi_vec *BX, const mi_vec *X,
                       const double *theta_host, mi_vec *R, double *rnorm, double *xnorm) {
  MI_REQUIRE(ctx && theta_host && rnorm && xnorm, "null argument");
  MI_REQUIRE(nx >= 1 && nx <= kGramMaxK, "block size must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(ctx, m, nx, AX, "AX"));
  MI_TRY(check_panel(ctx, m, nx, BX, "BX"));
  MI_TRY(check_panel(ctx, m, nx, X, "X"));
  touch(R);
  MI_TRY(check_panel(ctx, m, nx, R, "R"));
  void *thdev = nullptr;
  MI_TRY(pool_alloc(ctx, (size_t)nx * sizeof(double), &thdev));
  MI_TRY(stage_upload(ctx, theta_host, (size_t)nx * sizeof(double), thdev));  // (no host wait)
  const int grid = grid_for(ctx, m, 2);
  // 8 columns per launch (16 reduction components); every chunk's sums land in their own 16 doubles of
  // one device buffer, read back with ONE copy + sync after the last chunk
  const int nchunks = (nx + 7) / 8;
  void *sums = nullptr;
  MI_TRY(pool_alloc(ctx, (size_t)nchunks * 16 * sizeof(double), &sums));
  for (int ch = 0; ch < nchunks; ++ch) {
    {
      KScope ks(ctx, MI_K_LOBPCG_RESIDUAL);
      hipLaunchKernelGGL(k_residual, dim3(grid), dim3(kBlock), 0, ctx->stream, m, nx, 8 * ch, (const double *)AX->d,
                         (const double *)BX->d, (const double *)X->d, (const double *)thdev, R->d, ctx->partials2);
    }
    MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, grid, 16, (double *)sums + 16 * ch));
  }
  std::vector<double> out((size_t)nchunks * 16);
  hipError_t e = hipMemcpyAsync(out.data(), sums, out.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  pool_free(ctx, sums);
  pool_free(ctx, thdev);
  if (e != hipSuccess) return hip_fail(e, "residual norms read-back", __FILE__, __LINE__);
  for (int c = 0; c < nx; ++c) {
    rnorm[c] = std::sqrt(out[(size_t)(c / 8) * 16 + c % 8]);
    xnorm[c] = std::sqrt(out[(size_t)(c / 8) * 16 + 8 + c % 8]);
  }
  return MI_OK;
}

int mi_rayleigh_ritz(int n, const double *A, const double *B, double *Theta, double *C) {
  MI_REQUIRE(n >= 1 && A && B && Theta && C, "bad argument");
  // the header-only host solver of the template's generic path, compiled in its own host-only unit
  const int rc = mi::rr_host(n, A, B, Theta, C);
  if (rc == 1) {
    set_error("Rayleigh-Ritz: B has a non-positive diagonal entry");
    return MI_ERR_INVALID_ARGUMENT;
  }
  if (rc == 2) {
    set_error("Rayleigh-Ritz: equilibrated B is not positive definite");
    return MI_ERR_INVALID_ARGUMENT;
  }
  return MI_OK;
}

int mi_rayleigh_ritz_lowest(int n, int k, const double *A, const double *B, double *Theta, double *C) {
  MI_REQUIRE(n >= 1 && k >= 1 && k <= n && A && B && Theta && C, "bad argument");
  const int rc = mi::rr_host_lowest(n, k, A, B, Theta, C);
  if (rc == 1) {
    set_error("Rayleigh-Ritz: B has a non-positive diagonal entry");
    return MI_ERR_INVALID_ARGUMENT;
  }
  if (rc == 2) {
    set_error("Rayleigh-Ritz: equilibrated B is not positive definite");
    return MI_ERR_INVALID_ARGUMENT;
  }
  return MI_OK;
}

int mi_panel_rowscale(mi_ctx *ctx, size_t m, int k, const mi_vec *d, const mi_vec *X, mi_vec *Y) {
  MI_REQUIRE(ctx && d, "null argument");
  MI_REQUIRE(d->n == m, "scaling vector must have m entries");
  MI_TRY(check_panel(ctx, m, k, X, "X"));
  MI_TRY(check_panel(ctx, m, k, Y, "Y"));
  touch(Y);
  const int grid = (int)std::min<size_t>((m * (size_t)k + 255) / 256, 4096);
  hipLaunchKernelGGL(k_rowscale, dim3(grid), dim3(256), 0, ctx->stream, m, (size_t)k, (const double *)d->d,
                     (const double *)X->d, Y->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

namespace {
// the window form of the panel product applies (matrices that qualify, sparse.hip build_window; one context, no halo)
bool spmm_win_ok(const mi_csr *A) {
  const bool no_win = A->ctx->cfg.no_spmm_win;
  return A->pk && A->wk && A->win_chunks > 0 && A->win_chunks <= 2 && !no_win && !A->ctx->uniform_grid &&
         A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0 && A->n * 8 < ((size_t)1 << 32);
}

// Y = A X in window form, 8 columns per pass.  theta_host != nullptr: the fused residual form (k_spmm_colmajor_win<..,
// RES>): R = Y - X diag(theta) and, per pass, the 16 column sums |R_j|^2, |X_j|^2 reduced into sums_dev + 16 * pass.
bool spmm_sweep_ok(const mi_csr *A);
int spmm_sweep_launch(const mi_csr *A, int k, ColBlocks Xd, double *Yd, const double *theta_host, double *Rd,
                      double *sums_dev);
int spmm_win_launch(const mi_csr *A, int k, ColBlocks Xd, double *Yd, const double *theta_host, double *Rd,
                    double *sums_dev) {
  mi_ctx *ctx = A->ctx;
  if (spmm_sweep_ok(A)) return spmm_sweep_launch(A, k, Xd, Yd, theta_host, Rd, sums_dev);
  // 49 KB of ring at a half-width of two chunks, two workgroups per CU at 212-231 VGPRs (4 columns per pass: 313 us
  // per 24 columns, 8: 251 -> 224-236 with the later changes; r04: 12 columns per pass -- 74 KB of ring, 241-244 VGPRs,
  // still two waves per SIMD -- 234-242 us in the same call: the per-pass costs are not the bound, not kept)
  constexpr int kSpmmWinCols = 8;
  const int nc = 2 * kWinWaves + 2 * A->win_chunks;
  const size_t lds = (size_t)kSpmmWinCols * ((size_t)nc * 64 + 1) * sizeof(double);
  const bool hw7 = A->win_head <= 7, wc1 = A->win_chunks == 1, res = theta_host != nullptr;
  const void *fn = nullptr;
  const bool fard = A->win_far_pure > 0 && A->win_far_pure < ((size_t)1 << 31) && !ctx->cfg.no_far_computed;
#define PICK3(HWV, WCV, FV, RV) fn = (const void *)k_spmm_colmajor_win<kSpmmWinCols, HWV, WCV, FV, RV>
#define PICK(HWV, WCV)                                                        \
  if (res) { if (fard) PICK3(HWV, WCV, true, true); else PICK3(HWV, WCV, false, true); } \
  else { if (fard) PICK3(HWV, WCV, true, false); else PICK3(HWV, WCV, false, false); }
  if (wc1) { if (hw7) { PICK(7, 1) } else { PICK(8, 1) } }
  else { if (hw7) { PICK(7, 2) } else { PICK(8, 2) } }
#undef PICK
#undef PICK3
  static int occ_cache[2][2][2][2] = {};
  int &occ = occ_cache[hw7 ? 0 : 1][wc1 ? 0 : 1][fard ? 1 : 0][res ? 1 : 0];
  if (occ == 0) {
    int nbk = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, fn, kWinBlock, lds);
    occ = (e == hipSuccess && nbk > 0) ? nbk : 1;
    (void)hipGetLastError();
  }
  const int ntiles = (int)((A->nslices + kWinWaves - 1) / kWinWaves);
  int wgrid = 0;
  const int *bounds = nullptr;
  MI_TRY(window_bounds(ctx, A, std::min(occ, 4) * ctx->num_cu, ntiles, &wgrid, &bounds));
  SellView view = sell_view(A);
  WinView wv{A->wk, A->wfar, A->win_chunks, nc, A->win_zero, bounds, fard ? (unsigned)A->win_far_pure : 0u,
             nullptr};
  ThetaArg theta_arg{};
  if (res) std::copy(theta_host, theta_host + k, theta_arg.v);
  const int npass = (k + kSpmmWinCols - 1) / kSpmmWinCols;
  const int xgrid = (wgrid + kNumXCD - 1) / kNumXCD * kNumXCD;
  if (!res || (ctx->comm == nullptr && xgrid <= kMaxRows)) {
    // all passes in one launch; residual form: the passes' partial rows side by side, one reduction kernel for all
    void *pr = nullptr;
    if (res) MI_TRY(pool_alloc(ctx, (size_t)npass * 2 * kSpmmWinCols * kMaxRows * sizeof(double), &pr));
    double *partials = (double *)pr;
    int c0 = 0, nruns = wgrid;
    void *args[] = {&view, &wv, &k, &c0, &nruns, &Xd, &Yd, &theta_arg, &Rd, &partials};
    hipError_t e = hipLaunchKernel(fn, dim3(xgrid, npass), dim3(kWinBlock), args, lds, ctx->stream);
    int st = e == hipSuccess ? MI_OK : hip_fail(e, "panel product launch", __FILE__, __LINE__);
    if (st == MI_OK && res)
      st = launch_reduce_rows_to_slots(ctx, partials, xgrid, 2 * kSpmmWinCols, sums_dev, npass,
                                       (size_t)2 * kSpmmWinCols * kMaxRows);
    if (pr) pool_free(ctx, pr);
    MI_TRY(st);
  } else {
    double *partials = ctx->partials2;
    for (int c0 = 0; c0 < k; c0 += kSpmmWinCols) {
      int nruns = wgrid;
      void *args[] = {&view, &wv, &k, &c0, &nruns, &Xd, &Yd, &theta_arg, &Rd, &partials};
      MI_HIP(hipLaunchKernel(fn, dim3(wgrid), dim3(kWinBlock), args, lds, ctx->stream));
      MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, wgrid, 16, sums_dev + 16 * (c0 / kSpmmWinCols)));
    }
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// the plane-sweep form applies: packed matrix with a pure far structure, one context, no halo; opt-out NO_SPMM_SWEEP
bool spmm_sweep_ok(const mi_csr *A) {
  return A->pk && A->win_far_pure >= (size_t)(2 * kSwRows) && A->win_far_pure < ((size_t)1 << 22) && A->win_chunks > 0 &&
         A->win_chunks <= 2 && A->win_head <= 8 && !A->ctx->cfg.no_spmm_sweep && !A->ctx->uniform_grid &&
         A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0 && A->n < ((size_t)1 << 31) && !A->ctx->comm;
}

int spmm_sweep_launch(const mi_csr *A, int k, ColBlocks Xd, double *Yd, const double *theta_host, double *Rd,
                      double *sums_dev) {
  mi_ctx *ctx = A->ctx;
  constexpr int KC = 8;
  const size_t lds = (size_t)KC * ((size_t)kSwWin + 1) * sizeof(double);
  const bool hw7 = A->win_head <= 7, res = theta_host != nullptr;
  const void *fn = nullptr;
  if (res) fn = hw7 ? (const void *)k_spmm_colmajor_sweep<KC, 7, true> : (const void *)k_spmm_colmajor_sweep<KC, 8, true>;
  else fn = hw7 ? (const void *)k_spmm_colmajor_sweep<KC, 7, false> : (const void *)k_spmm_colmajor_sweep<KC, 8, false>;
  const unsigned D = (unsigned)A->win_far_pure;
  const int tiles = (int)((D + kSwRows - 1) / kSwRows);
  const int nst = (int)((A->n + D - 1) / D);
  const int npass = (k + KC - 1) / KC;
  // z-segments: enough workgroups to fill the chip twice over in one launch, at most kMaxRows partial rows, and not so
  // short that a segment's start-up (three plane-tiles loaded for zs computed) weighs more than ~20 %
  int zsegs = ctx->cfg.sweep_zsegs > 0 ? ctx->cfg.sweep_zsegs : (4 * ctx->num_cu + tiles * npass - 1) / (tiles * npass);
  zsegs = std::max(1, std::min(zsegs, std::min(nst / 12 > 0 ? nst / 12 : 1, kMaxRows / tiles)));
  const int zs = (nst + zsegs - 1) / zsegs;
  zsegs = (nst + zs - 1) / zs;
  const int xgrid = tiles * zsegs;
  MI_REQUIRE(xgrid <= kMaxRows, "plane-sweep product: %d workgroups per pass", xgrid);
  SellView view = sell_view(A);
  ThetaArg theta_arg{};
  if (res) std::copy(theta_host, theta_host + k, theta_arg.v);
  void *pr = nullptr;
  if (res) MI_TRY(pool_alloc(ctx, (size_t)npass * 2 * KC * kMaxRows * sizeof(double), &pr));
  double *partials = (double *)pr;
  unsigned Darg = D;
  int tiles_a = tiles, zs_a = zs, c0 = 0;
  void *args[] = {&view, &Darg, &tiles_a, &zs_a, &k, &c0, &Xd, &Yd, &theta_arg, &Rd, &partials};
  hipError_t e = hipLaunchKernel(fn, dim3(xgrid, npass), dim3(kSwBlock), args, lds, ctx->stream);
  int st = e == hipSuccess ? MI_OK : hip_fail(e, "plane-sweep panel product launch", __FILE__, __LINE__);
  if (st == MI_OK && res)
    st = launch_reduce_rows_to_slots(ctx, partials, xgrid, 2 * KC, sums_dev, npass, (size_t)2 * KC * kMaxRows);
  if (pr) pool_free(ctx, pr);
  MI_TRY(st);
  MI_HIP(hipGetLastError());
  return MI_OK;
}
}  // namespace

// AX = A X (LOBPCG.h:281) with R = AX - X diag(theta) (:285, B absent: BX = X) and the column norms of R and X
// (:293,302) in the same pass over X when the matrix takes the window form; otherwise the product followed by
// mi_lobpcg_residual -- the same R bits either way.  sync (one read-back of the 2 nx sums)
int mi_csr_spmm_colmajor_residual(const mi_csr *A, int nx, const mi_vec *X, const double *theta_host, mi_vec *AX,
                                  mi_vec *R, double *rnorm, double *xnorm) {
  MI_REQUIRE(A && X && theta_host && AX && R && rnorm && xnorm, "null argument");
  MI_REQUIRE(nx >= 1 && nx <= kGramMaxK, "block size must be in [1,%d]", kGramMaxK);
  mi_ctx *ctx = A->ctx;
  MI_TRY(check_panel(ctx, A->n, nx, X, "X"));
  MI_TRY(check_panel(ctx, A->n, nx, AX, "AX"));
  MI_TRY(check_panel(ctx, A->n, nx, R, "R"));
  MI_REQUIRE(X->d != AX->d && X->d != R->d && AX->d != R->d, "X, AX and R must not alias");
  if (!spmm_win_ok(A) || A->n == 0) {
    MI_TRY(mi_csr_spmm_colmajor(A, nx, X, AX));
    return mi_lobpcg_residual(ctx, A->n, nx, AX, X, X, theta_host, R, rnorm, xnorm);
  }
  touch(AX);
  touch(R);
  const int nchunks = (nx + 7) / 8;
  void *sums = nullptr;
  // one rank: the 2 nx column sums land straight in pinned host memory (the reduction kernel stores them there), no
  // copy behind the last kernel; several ranks: device buffer + read-back
  const bool zero_copy = ctx->comm == nullptr && !ctx->cfg.no_zero_copy;
  const size_t nsums = (size_t)nchunks * 16;
  void *zc_host = nullptr;
  int st = zero_copy ? readback_area(ctx, nsums * sizeof(double), &zc_host, &sums)
                     : pool_alloc(ctx, nsums * sizeof(double), &sums);
  if (st == MI_OK) {
    KScope ks(ctx, MI_K_SPMM);
    st = spmm_win_launch(A, nx, ColBlocks{{X->d, X->d, X->d}, nx, nx}, AX->d, theta_host, R->d, (double *)sums);
  }
  if (st != MI_OK) {  // (the pool buffer goes back on every path)
    if (sums && !zero_copy) pool_free(ctx, sums);
    return st;
  }
  std::vector<double> out(nsums);
  if (zero_copy) {
    st = stream_wait(ctx, "residual norms read-back");
    if (st == MI_OK) memcpy(out.data(), zc_host, nsums * sizeof(double));
  } else {
    const void *dv[1] = {sums};
    const size_t by[1] = {nsums * sizeof(double)};
    void *hs[1] = {out.data()};
    st = readback_sync(ctx, 1, dv, by, hs);
    pool_free(ctx, sums);
  }
  if (st != MI_OK) return st;
  for (int c = 0; c < nx; ++c) {
    rnorm[c] = std::sqrt(out[(size_t)(c / 8) * 16 + c % 8]);
    xnorm[c] = std::sqrt(out[(size_t)(c / 8) * 16 + 8 + c % 8]);
  }
  return MI_OK;
}

int mi_csr_spmm_colmajor_blocks(const mi_csr *A, const mi_panel_blocks *X, mi_vec *Y) {
  MI_REQUIRE(A && X && Y, "null argument");
  mi_ctx *ctx = A->ctx;
  ColBlocks cb;
  int k = 0;
  MI_TRY(blocks_to_cols(ctx, A->n, X, &cb, &k));
  MI_TRY(check_panel(ctx, A->n, k, Y, "Y"));
  for (int i = 0; i < X->nblocks; ++i) {
    const double *s0 = X->block[i]->d, *s1 = s0 + (size_t)X->cols[i] * A->n;
    MI_REQUIRE(Y->d + (size_t)k * A->n <= s0 || Y->d >= s1, "SpMM input and output must not alias");
  }
  const bool sharded = A->halo_lo + A->halo_hi + A->send_lo + A->send_hi > 0;
  if (!sharded && spmm_win_ok(A) && A->n > 0) {
    touch(Y);
    KScope ks(ctx, MI_K_SPMM);
    return spmm_win_launch(A, k, cb, Y->d, nullptr, nullptr, nullptr);
  }
  mi_vec *tmp = nullptr;  // (matrix forms whose product kernels want one panel: the blocks copied together)
  MI_TRY(materialize_blocks(ctx, A->n, X, &tmp));
  const int st = mi_csr_spmm_colmajor(A, k, tmp, Y);
  mi_vec_destroy(tmp);
  return st;
}

int mi_csr_spmm_colmajor(const mi_csr *A, int k, const mi_vec *X, mi_vec *Y) {
  MI_REQUIRE(A && X && Y, "null argument");
  MI_REQUIRE(k >= 1 && k <= kGramMaxK, "panel width must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(A->ctx, A->n, k, X, "X"));
  MI_TRY(check_panel(A->ctx, A->n, k, Y, "Y"));
  MI_REQUIRE(X->d != Y->d, "SpMM input and output must not alias");
  touch(Y);
  mi_ctx *ctx = A->ctx;
  if (A->halo_lo + A->halo_hi + A->send_lo + A->send_hi > 0) {
    // Row-sharded matrix (8(e): "LOBPCG: row-shard S"): four columns at a time through the row-major sharded
    // product -- its halo exchange is sized for p <= 4 -- at the price of two extra passes over those columns.
    // COLLECTIVE: every rank of the communicator makes the same call.
    void *vin = nullptr, *vout = nullptr;
    MI_TRY(pool_alloc(ctx, A->n * 4 * sizeof(double), &vin));
    int st = pool_alloc(ctx, A->n * 4 * sizeof(double), &vout);
    const int tgrid = (int)std::max<size_t>(1, std::min<size_t>((A->n + 255) / 256, 2048));
    for (int c0 = 0; c0 < k && st == MI_OK; c0 += 4) {
      const int kc = std::min(4, k - c0);
      hipLaunchKernelGGL(k_cols_to_rows, dim3(tgrid), dim3(256), 0, ctx->stream, A->n, kc,
                         (const double *)X->d + (size_t)c0 * A->n, (double *)vin);
      st = comm_halo_exchange(ctx, A, kc, (const double *)vin);
      if (st == MI_OK) st = csr_spmm_launch(A, kc, (const double *)vin, (double *)vout);
      hipLaunchKernelGGL(k_rows_to_cols, dim3(tgrid), dim3(256), 0, ctx->stream, A->n, kc, (const double *)vout,
                         Y->d + (size_t)c0 * A->n);
    }
    pool_free(ctx, vin);
    if (vout) pool_free(ctx, vout);
    MI_TRY(st);
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  const int grid = (int)((A->nslices + 3) / 4);
  KScope ks(ctx, MI_K_SPMM);
  if (spmm_win_ok(A)) return spmm_win_launch(A, k, ColBlocks{{X->d, X->d, X->d}, k, k}, Y->d, nullptr, nullptr, nullptr);
  if (A->pk) {
    constexpr int chunk = 24;  // (narrower column chunks measured slower: DESIGN 7.5)
    for (int c0 = 0; c0 < k;) {
      const int left = k - c0;
#define SPMMPK(KC)                                                                                            \
  hipLaunchKernelGGL(k_spmm_colmajor_pk<KC>, dim3(grid), dim3(256), 0, ctx->stream, A->n, A->nslices,         \
                     (const long long *)A->slice_ptr, (const uint32_t *)A->pk, (const double *)A->vtab, k,    \
                     c0, (const double *)X->d, Y->d)
      const int want = std::min(left, chunk);
      if (want > 16) { SPMMPK(24); c0 += 24; }
      else if (want > 8) { SPMMPK(16); c0 += 16; }
      else if (want > 4) { SPMMPK(8); c0 += 8; }
      else if (want > 2) { SPMMPK(4); c0 += 4; }
      else { SPMMPK(2); c0 += 2; }
#undef SPMMPK
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  for (int c0 = 0; c0 < k;) {
    const int left = k - c0;
#define SPMM(KC)                                                                                             \
  hipLaunchKernelGGL(k_spmm_colmajor<KC>, dim3(grid), dim3(256), 0, ctx->stream, A->n, A->nslices,           \
                     (const long long *)A->slice_ptr, (const int *)A->col, (const double *)A->val, k, c0,    \
                     (const double *)X->d, Y->d)
    if (left > 16) {
      SPMM(24);
      c0 += 24;
    } else if (left > 8) {
      SPMM(16);
      c0 += 16;
    } else {
      SPMM(8);
      c0 += 8;
    }
#undef SPMM
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // extern "C"
