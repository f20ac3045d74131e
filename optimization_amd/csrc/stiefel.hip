// stiefel.hip -- Stiefel manifold St(n,p) (p <= 8, embedded metric) kernels and the Rayleigh-quotient
// problem  f(X) = .5 tr(X' A X):  the Objective / QuadraticModel / RiemannianMetric / Retraction
// callables a TNT client supplies (Riemannian/Concepts.h:44-112).  The reference ships only the
// S^2 = St(3,1) lambdas of tests/TNT_unit_test.cpp:73-117; these are their n x p generalisation:
//     project  P_X(Z) = Z - X sym(X'Z)                      (sphere: V - X.dot(V) X,  :73-75)
//     gradient P_X(A X)                                     (:79-85)
//     Hessian  P_X(A V - V sym(X'AX))                       (:94-97)
//     retract  polar factor of X + V                        ((X+V).normalized(),      :106-108)
//
// Per Hessian application (N = n p doubles, A in sliced-ELL) exactly two kernels run:
//   k_st_spmm_gram  reads A, V (gathered through L2), X; writes Z = A V - V S; one partial row of
//                   sym(X'Z) per workgroup                                  [12 nnz + 4 n + 8 (3N)]
//   k_st_finish     prologue: every workgroup re-reduces those <= 512 rows (deterministic);
//                   body: Hp = Z - X sym(X'Z); partial rows of <V,Hp>, <Hp,Hp>, <V,V>      [8 (4N)]
// so the three curvature inner products of STPCG (IterativeSolvers.h:300,305-306) cost no extra pass
// and no scalar kernel sits between the passes.
#include "comm_ipc.h"
#include "spmm_core.h"
#include "stiefel_core.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

using namespace mi;

namespace {

enum { POST_SYM = 1, POST_INVSQRT = 2 };

// (a scheduling fence behind every column of the three P x P products: without it the scheduler issues all 192 LDS
// reads of a row up front -- 128 more live doubles -- and the kernel loses its second wave per SIMD)
#ifndef MI_WIDE_NO_SCHED
#define MI_WIDE_SCHED() __builtin_amdgcn_sched_barrier(0)
#else
#define MI_WIDE_SCHED()
#endif
#ifndef MI_WIDE_CHUNK
#define MI_WIDE_CHUNK 2
#endif
#ifndef MI_WIDE_CHUNK_FROM
#define MI_WIDE_CHUNK_FROM 7   // rows of >= this many doubles gather in chunks of MI_WIDE_CHUNK entries
#endif
#ifndef MI_WIDE_QUAD_CHUNK
#define MI_WIDE_QUAD_CHUNK 8   // entries per chunk step of the quad-layout pass (16 bytes per entry and lane)
#endif
#ifndef MI_WIDE_QUAD_WAVES
#define MI_WIDE_QUAD_WAVES 2   // waves per SIMD of the quad-layout pass (<= 168 registers: 3 would fit)
#endif
#ifndef MI_WIDE_WAVES
#define MI_WIDE_WAVES 2        // waves per SIMD the kernel is held to (measured: 3 -- 168 registers, p <= 7 -- is 3-6 % slower)
#endif


// ---- small dense helpers (device) --------------------------------------------------------
template <int P>
__device__ void dev_sym_invsqrt(const double *G, double *out) {
  // cyclic Jacobi on the P x P symmetric matrix G; out = Q diag(w^-1/2) Q'
  double M[P * P], Q[P * P];
  for (int i = 0; i < P * P; ++i) { M[i] = G[i]; Q[i] = 0; }
  for (int i = 0; i < P; ++i) Q[i * P + i] = 1;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int i = 0; i < P; ++i)
      for (int j = i + 1; j < P; ++j) off += M[i * P + j] * M[i * P + j];
    if (off == 0) break;
    for (int i = 0; i + 1 < P; ++i)
      for (int j = i + 1; j < P; ++j) {
        const double apq = M[i * P + j];
        if (apq == 0) continue;
        const double tau = (M[j * P + j] - M[i * P + i]) / (2 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
        const double c = 1 / sqrt(1 + t * t), s = t * c;
        for (int k = 0; k < P; ++k) {
          const double a = M[k * P + i], b = M[k * P + j];
          M[k * P + i] = c * a - s * b;
          M[k * P + j] = s * a + c * b;
        }
        for (int k = 0; k < P; ++k) {
          const double a = M[i * P + k], b = M[j * P + k];
          M[i * P + k] = c * a - s * b;
          M[j * P + k] = s * a + c * b;
        }
        for (int k = 0; k < P; ++k) {
          const double a = Q[k * P + i], b = Q[k * P + j];
          Q[k * P + i] = c * a - s * b;
          Q[k * P + j] = s * a + c * b;
        }
      }
  }
  for (int i = 0; i < P; ++i)
    for (int j = 0; j < P; ++j) {
      double acc = 0;
      for (int k = 0; k < P; ++k) acc += Q[i * P + k] * (1.0 / sqrt(M[k * P + k])) * Q[j * P + k];
      out[i * P + j] = acc;
    }
}

// G^-1/2 for rows wider than 4 doubles (r05), by ONE WAVE with the matrices in LDS: the coupled Newton-Schulz iteration
//     Y0 = G / s, Z0 = I;  T = Z Y;  Y <- Y (3 I - T) / 2,  Z <- (3 I - T) Z / 2;   Z -> (G / s)^-1/2,  s = |G|_F
// (eigenvalues of G / s lie in (0, 1], so |I - G / s| < 1: it converges for every SPD G, quadratically; the polar
// retraction's G = (X + V)'(X + V) = I + V'V for a tangent V has its spectrum in [1, s]: log2(s) + ~6 iterations).
// Three 8 x 8 matrix products per iteration, lane (i, j) one entry each: ~1 us per iteration.  dev_sym_invsqrt's
// cyclic Jacobi is strictly serial -- ~300 rotations with three dependent divisions / square roots each, M and Q in
// scratch memory at P = 8: the retraction took 1.7 ms at n = 1e6; as a wave-parallel Jacobi in LDS still 0.65 ms -- and
// stays the p <= 4 path (same rotations as the oracle's).  Here the result agrees with it to rounding (tested at 1e-13).
// T, Y, Z: P x P doubles of LDS each (work[0 .. 3 P P)); out may be LDS.  All 64 lanes of the calling wave call it.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int P>
__device__ void dev_sym_invsqrt_wave(const double *G, double *work, double *out) {
  double *T = work, *Y = work + P * P, *Z = work + 2 * P * P;
  const int l = threadIdx.x & 63, i = (l < P * P ? l : 0) / P, j = (l < P * P ? l : 0) % P;
  const bool mine = l < P * P;
  double s2 = 0;
  for (int k = 0; k < P * P; ++k) s2 += G[k] * G[k];
  const double sc = sqrt(s2);
  if (mine) {
    Y[l] = G[l] / sc;
    Z[l] = (i == j) ? 1.0 : 0.0;
  }
  wave_lds_sync();
  bool last = false;
  for (int it = 0; it < 100; ++it) {
    double t = 0;
    for (int k = 0; k < P; ++k) t += Z[i * P + k] * Y[k * P + j];
    if (mine) T[l] = t;
    wave_lds_sync();
    double err = 0;  // |Z Y - I|_F^2, the same number in every lane
    for (int k = 0; k < P * P; ++k) {
      const double d = T[k] - ((k / P == k % P) ? 1.0 : 0.0);
      err += d * d;
    }
    double yn = 0, zn = 0;
    for (int k = 0; k < P; ++k) {
      const double rkj = ((k == j) ? 3.0 : 0.0) - T[k * P + j], rik = ((i == k) ? 3.0 : 0.0) - T[i * P + k];
      yn += Y[i * P + k] * rkj;
      zn += rik * Z[k * P + j];
    }
    wave_lds_sync();
    if (mine) {
      Y[l] = .5 * yn;
      Z[l] = .5 * zn;
    }
    wave_lds_sync();
    if (last) break;
    if (!(err > 1e-26)) last = true;  // (converged to ~1e-13: one more quadratic step finishes it; NaN ends it too)
  }
  if (mine) out[l] = Z[l] / sqrt(sc);
  wave_lds_sync();
}

// ---- kernels -----------------------------------------------------------------------------

// Z = A V - V S (S may be null => Z = A V); partial row of sym(X'Z)
template <int P>
__global__ __launch_bounds__(kBlock) void k_st_spmm_gram(SellView A, const CgState *__restrict__ st,
                                                         const double *__restrict__ V,
                                                         const double *__restrict__ X,
                                                         const double *__restrict__ S,
                                                         double *__restrict__ Z,
                                                         double *__restrict__ partials) {
  __shared__ double lds[SymIdx<P>::NS * kWaves];
  if (st && st->mode != CG_RUN) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double Sm[P * P];
#pragma unroll
  for (int i = 0; i < P * P; ++i) Sm[i] = S ? S[i] : 0.0;
  double G[P * P];
#pragma unroll
  for (int i = 0; i < P * P; ++i) G[i] = 0;
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
      double x[P], v[P];
#pragma unroll
      for (int c = 0; c < P; ++c) { x[c] = X[row * P + c]; v[c] = V[row * P + c]; }
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#pragma unroll
        for (int a = 0; a < P; ++a) t += v[a] * Sm[a * P + b];
        acc[b] -= t;
        Z[row * P + b] = acc[b];
      }
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) G[a * P + b] += x[a] * acc[b];
    }
  }
  store_sym_partials<P>(G, lds, partials);
}

// out = P_X(A V - V S) and the three curvature partials in ONE pass.  The projection's matrix
// M = sym(X'(A V - V S)) is known before the pass: for symmetric A it equals sym(Y'V - (X'V) S) with
// Y = A X fixed during the inner solve, and STPCG's direction kernel left its partial rows when it formed
// V (mi_op::dirgram).  So no second pass over Z, X, V is needed (k_st_finish: 8 (4N) bytes saved).
// k_st_spmm_gram on the lean pipelined core (packed matrix when there is one); same arithmetic per row
template <int P, bool HALO, bool PK>
__global__ __launch_bounds__(StBlk<P>::threads) void k_st_spmm_gram_stream(SellView A, const CgState *__restrict__ st,
                                                                const double *__restrict__ V,
                                                                const double *__restrict__ X,
                                                                const double *__restrict__ S,
                                                                double *__restrict__ Z,
                                                                double *__restrict__ partials) {
  constexpr int NW = StBlk<P>::waves;
  constexpr bool WIDE = P > 4;  // (rows wider than 4 doubles: 256-thread workgroups, S in LDS: stiefel_core.h StBlk)
  __shared__ double lds[SymIdx<P>::NS * NW];
  __shared__ double vt[PK ? 256 : 1];
  __shared__ double SmL[WIDE ? P * P : 1];
  if (st && st->mode != CG_RUN) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PK) {
    if (threadIdx.x < 256) vt[threadIdx.x] = A.vtab[threadIdx.x];
  }
  if (WIDE && threadIdx.x < P * P) SmL[threadIdx.x] = S ? S[threadIdx.x] : 0.0;
  if (PK || WIDE) __syncthreads();
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = (A.nslices * lb) / nb, s1 = (A.nslices * (lb + 1)) / nb;
  double SmR[WIDE ? 1 : P * P];
  if (!WIDE) {
#pragma unroll
    for (int i = 0; i < P * P; ++i) SmR[i] = S ? S[i] : 0.0;
  }
  double G[P * P];
#pragma unroll
  for (int i = 0; i < P * P; ++i) G[i] = 0;
  struct Epi {
    const SellView &A;
    const double *__restrict__ X, *__restrict__ V;
    double *__restrict__ Z;
    const double (&SmR)[WIDE ? 1 : P * P];  // S in registers (p <= 4) ...
    const double *SmL;                      // ... or in LDS
    double (&G)[P * P];
    int lane;
    double x[P], v[P];
    __device__ __forceinline__ unsigned lane_off(size_t slice) const {
      return (slice * 64 + lane < A.n) ? (unsigned)lane * (unsigned)(P * 8) : 0u;
    }
    __device__ __forceinline__ void begin(size_t slice) {
      const unsigned off = lane_off(slice);
      const double *xs = reinterpret_cast<const double *>(reinterpret_cast<const char *>(X + slice * 64 * P) + off);
      const double *vs = reinterpret_cast<const double *>(reinterpret_cast<const char *>(V + slice * 64 * P) + off);
#pragma unroll
      for (int c = 0; c < P; ++c) { x[c] = xs[c]; v[c] = vs[c]; }
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P]) {
      if (slice * 64 + lane >= A.n) return;
      if (WIDE) asm volatile("" ::: "memory");  // (S is re-read from LDS per row, not parked in 64 registers)
      double *zs = reinterpret_cast<double *>(reinterpret_cast<char *>(Z + slice * 64 * P) + lane_off(slice));
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#pragma unroll
        for (int a = 0; a < P; ++a) t += v[a] * (WIDE ? SmL[a * P + b] : SmR[WIDE ? 0 : a * P + b]);
        acc[b] -= t;
        zs[b] = acc[b];
        if (WIDE) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int a = 0; a < P; ++a)
#pragma unroll
        for (int b = 0; b < P; ++b) G[a * P + b] += x[a] * acc[b];
    }
  } epi{A, X, V, Z, SmR, SmL, G, lane, {}, {}};
  sell_stream<P, HALO, PK, Epi, NW, (P >= MI_WIDE_CHUNK_FROM ? MI_WIDE_CHUNK : MI_SPMM_CHUNK)>(
      A, s0 + (size_t)__builtin_amdgcn_readfirstlane(w), s1, lane, V, vt, epi);
  store_sym_partials<P>(G, lds, partials);
}

// RECUR (mi_op::apply_dir with gram_count < 0): M is read from gdir (packed symmetric, replicated scalars kept
// by STPCG) and the packed symmetric Gram  sym(Y'out - (X'out) S)  of the OUTPUT rides along as components
// 3.. of the partial row (DirComps<P>::value components in all).
// HW > 0 (mi_csr::win_chunks > 0): the window form -- the rows of V near the workgroup's own are staged in an LDS
// ring and every entry's row is read from LDS at an address the host worked out (spmm_core.h sell_window); HW = the
// entries per slice.  Same arithmetic, bit-identical results.
// FAR (window form, unsharded): 0 far columns loaded; 1 computed (pure far structure); 2 computed + 16-bit words
// TWOK (r05, opt-in experiment TWO_KERNEL_STEP, window form only): the pass also reads the residual r and leaves the
// partial rows of <r,out> and <V,r> (components KC-2, KC-1): with them <r+,r+> = <r,r> + 2 alpha <r,Hp> + alpha^2 <Hp,Hp>
// is known without a pass over r+, and the two CG kernels of an iteration merge into one (stpcg.hip k_cg_step2).
template <int P, bool FROM_SLOTS, bool HALO, bool RECUR, bool PK, int HW, int FAR = 0, bool TWOK = false>
__global__ __launch_bounds__(HW > 0 ? kWinBlock : kBlock) void k_st_hess_fused(SellView A, WinView Wv, const CgState *__restrict__ st,
                                                          const double *__restrict__ V,
                                                          const double *__restrict__ X,
                                                          const double *__restrict__ Y,
                                                          const double *__restrict__ S,
                                                          const double *__restrict__ gram_partials, int count,
                                                          const double *__restrict__ slots,
                                                          const double *__restrict__ gdir,
                                                          double *__restrict__ out,
                                                          double *__restrict__ partials, HaloWaitArg<HALO> hwait,
                                                          const double *__restrict__ Rres = nullptr) {
  constexpr bool WIN = HW > 0 && P <= 3;  // (P = 4: ring + parked rows would need > 160 KB of LDS; never dispatched)
  static_assert(!TWOK || (WIN && RECUR), "the two-kernel step exists in the window form only");
  constexpr int NS = SymIdx<P>::NS, KC = TWOK ? 3 + NS + 2 : (RECUR ? DirComps<P>::value : 3);
  constexpr int kLds = (NS * (kWaves + 1) > KC * kWaves) ? NS * (kWaves + 1) : KC * kWaves;
  __shared__ double lds[kLds];
  __shared__ double vt[PK ? 256 : 1];  // PK: the matrix's value table
  __shared__ double ring[WIN ? kWinLdsRows * P : 1];  // WIN: ring, zero row, far slots
  if (st && st->mode != CG_RUN) return;
  // sharded: the neighbours' rows of V were pushed by the kernel that wrote V (comm_ipc.h HaloPush); wait for them here
  if constexpr (HALO) halo_wait(hwait.w);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#ifdef MI_WIN_STAMPS  // both forms: each wave's entry and exit on the constant 100 MHz clock (slots 56, 57)
  const unsigned long long t_entry = wall_clock64();
#endif
  if (PK) {
    if (threadIdx.x < 256) vt[threadIdx.x] = A.vtab[threadIdx.x];
    if (!WIN) __syncthreads();  // WIN: the barrier behind the ring fill publishes the table too
  }
  // a contiguous range of SLICES per workgroup (XCD-aware like group_range, but balanced to one slice)
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = (A.nslices * lb) / nb, s1 = (A.nslices * (lb + 1)) / nb;
  double Sm[P * P];
#pragma unroll
  for (int i = 0; i < P * P; ++i) Sm[i] = S[i];
  double Mm[P * P];
  if (RECUR) {
#pragma unroll
    for (int aa = 0; aa < P; ++aa)
#pragma unroll
      for (int b = aa; b < P; ++b) {
        const double m = gdir[SLOT_GDIR_P + SymIdx<P>::at(aa, b)];
        Mm[aa * P + b] = m;
        Mm[b * P + aa] = m;
      }
  } else {
    load_sym<P, FROM_SLOTS>(gram_partials, count, slots, Mm, lds);
  }
  double a[KC];
#pragma unroll
  for (int i = 0; i < KC; ++i) a[i] = 0;
  struct Epi {
    const SellView &A;
    const double *__restrict__ X, *__restrict__ Y, *__restrict__ V;
    double *__restrict__ out;
    const double (&Sm)[P * P], (&Mm)[P * P];
    double (&a)[KC];
    int lane;
    double x[P], y[P], v[P];
    const double *__restrict__ Rr;
    double rr[TWOK ? P : 1], rn[TWOK ? P : 1];
    // rows of the slice from a scalar base + a 32-bit lane offset (lanes past the last row read its first)
    // (32-bit arithmetic: the fields span < 4 GiB, sell_stream_ok)
    __device__ __forceinline__ unsigned lane_off(size_t slice) const {
      return ((unsigned)slice * 64u + (unsigned)lane < (unsigned)A.n) ? (unsigned)lane * (unsigned)(P * 8) : 0u;
    }
    __device__ __forceinline__ const double *row_of(const double *F, size_t slice, unsigned off) const {
      return reinterpret_cast<const double *>(reinterpret_cast<const char *>(F) + (unsigned)slice * (unsigned)(64 * P * 8) +
                                              off);
    }
    __device__ __forceinline__ void begin(size_t slice) {
      const unsigned off = lane_off(slice);
      const double *xs = row_of(X, slice, off);
#pragma unroll
      for (int c = 0; c < P; ++c) x[c] = xs[c];
      if (!WIN) {  // WIN: the row of V comes from the ring (end(slice, acc, vrow))
        const double *vs = row_of(V, slice, off);
#pragma unroll
        for (int c = 0; c < P; ++c) v[c] = vs[c];
      }
      if (RECUR) {
        const double *ys = row_of(Y, slice, off);
#pragma unroll
        for (int c = 0; c < P; ++c) y[c] = ys[c];
      }
    }
    // sell_window's protocol: request(slice) issues the loads of the slice's rows (pinned: see spmm_core.h),
    // end(slice, acc, vrow) consumes them
    double xn[WIN ? P : 1], yn[WIN ? P : 1];
    __device__ __forceinline__ void request(size_t slice) {
      const unsigned off = lane_off(slice);
      const double *xs = row_of(X, slice, off);
#pragma unroll
      for (int c = 0; c < P; ++c) xn[WIN ? c : 0] = pinned_load(xs + c);
      if (RECUR) {
        const double *ys = row_of(Y, slice, off);
#pragma unroll
        for (int c = 0; c < P; ++c) yn[WIN ? c : 0] = pinned_load(ys + c);
      }
      if (TWOK) {
        const double *rs = row_of(Rr, slice, off);
#pragma unroll
        for (int c = 0; c < P; ++c) rn[TWOK ? c : 0] = pinned_load(rs + c);
      }
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P], const double (&vrow)[P]) {
#pragma unroll
      for (int c = 0; c < P; ++c) {
        v[c] = vrow[c];
        x[c] = xn[WIN ? c : 0];
        if (RECUR) y[c] = yn[WIN ? c : 0];
        if (TWOK) rr[TWOK ? c : 0] = rn[TWOK ? c : 0];
      }
      end(slice, acc);
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P]) {
      if ((unsigned)slice * 64u + (unsigned)lane >= (unsigned)A.n) return;
      double *os = reinterpret_cast<double *>(reinterpret_cast<char *>(out) + (unsigned)slice * (unsigned)(64 * P * 8) +
                                              lane_off(slice));
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += v[aa] * Sm[aa * P + b];
        acc[b] -= t;  // Z = A V - V S
      }
      double o[P];
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += x[aa] * Mm[aa * P + b];
        o[b] = acc[b] - t;  // Z - X M
        os[b] = o[b];
        a[0] += v[b] * o[b]; a[1] += o[b] * o[b]; a[2] += v[b] * v[b];
        if (TWOK) { a[KC - 2] += rr[TWOK ? b : 0] * o[b]; a[KC - 1] += v[b] * rr[TWOK ? b : 0]; }
      }
      if (RECUR) {  // packed sym(y o' - x (o S)'): the Gram of this output row
        double os_[P];
#pragma unroll
        for (int b = 0; b < P; ++b) {
          double t = 0;
#pragma unroll
          for (int aa = 0; aa < P; ++aa) t += o[aa] * Sm[aa * P + b];
          os_[b] = t;
        }
#pragma unroll
        for (int aa = 0; aa < P; ++aa)
#pragma unroll
          for (int b = aa; b < P; ++b) {
            const double gab = y[aa] * o[b] - x[aa] * os_[b];
            const double gba = y[b] * o[aa] - x[b] * os_[aa];
            a[3 + SymIdx<P>::at(aa, b)] += (aa == b) ? gab : .5 * (gab + gba);
          }
      }
    }
  } epi{A, X, Y, V, out, Sm, Mm, a, lane, {}, {}, {}, Rres, {}, {}, {}, {}};
  // the wave index as a scalar: slice bounds then come from scalar loads and the loop control is scalar
  const int wu = __builtin_amdgcn_readfirstlane(w);
  if constexpr (WIN) {
    // a contiguous run of whole TILES (kWinWaves slices) per workgroup: the launch plan's table, or equal runs
    const int ntiles = (int)((A.nslices + kWinWaves - 1) / kWinWaves), per = (ntiles + (int)nb - 1) / (int)nb;
    int t0 = (int)lb * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    if (Wv.bounds) {  // runs cut to the stride of the far entries (window_bounds)
      t0 = scalar_int(Wv.bounds, lb);
      t1 = scalar_int(Wv.bounds, lb + 1);
    }
    if constexpr (HALO)
      sell_window<P, HW, true, (FAR >= 1), false>(A, Wv, t0, t1, wu, lane, V, vt, ring, epi, hwait.halo_lo, hwait.halo_hi);
    else
      // (far rows as coalesced images, sell_window FARC: 26.7-26.9 us either way at p = 3 -- the gathers stay)
      sell_window<P, HW, false, (FAR >= 1), (FAR == 2), Epi, P, (FAR >= 1) && (MI_WIN_FARC >= 2)>(A, Wv, t0, t1, wu, lane, V, vt, ring, epi);
  } else {
    sell_stream<P, HALO, PK>(A, s0 + (size_t)wu, s1, lane, V, vt, epi);
  }
#ifdef MI_WIN_STAMPS
  const unsigned long long t_body = wall_clock64();
#endif
  if constexpr (WIN) block_partials_store_nw<KC, kWinWaves>(a, lds, partials);
  else block_partials_store<KC>(a, lds, partials);
#ifdef MI_WIN_STAMPS
  if (g_stamp_buf && lane == 0) {
    unsigned long long *o = g_stamp_buf + ((size_t)blockIdx.x * (WIN ? kWinWaves : kWaves) + w) * kStampSlots;
    o[56] = t_entry;
    o[57] = wall_clock64();
    o[58] = t_body;
  }
#endif
}

// ---- rows wider than 4 doubles (p = 5 ... 8) ------------------------------------------------------------------------------
// The one-pass Hessian in its recurrence form for wide rows.  Same schedule and the same per-row arithmetic as
// k_st_hess_fused<P, ., ., RECUR = true, ...> -- Z = A V - V S, out = Z - X M with M = G(p) from STPCG's recurrence, the
// three curvature dots and the packed Gram sym(Y'out - (X'out) S) of the output in the epilogue of the only pass -- but
// laid out for what p = 8 costs: a row is 64 bytes, three P x P products per row, 3 + P (P + 1) / 2 <= 39 accumulators
// per lane.  So: 256-thread workgroups (one wave per SIMD, the whole vector register file to itself instead of the 128
// registers a 1024-thread workgroup leaves a wave), S and M in LDS (every lane reads the same address: broadcast reads,
// no bank conflicts) instead of 2 x 64 replicated registers, up to kMaxRows partial rows.
constexpr int kWideBlock = 256, kWideWaves = kWideBlock / 64;
// first slice of workgroup lb's range: the planned runs of whole 4-slice tiles (sparse.hip window_bounds: cut so that
// the far stride of the matrix -- the plane stride of a 3-D stencil -- is a whole number of runs, a far row then is
// some other workgroup's own row at the same moment of the kernel) or equal shares
__device__ __forceinline__ size_t wide_first_slice(size_t nslices, unsigned lb, unsigned nb, const int *bounds) {
  if (bounds) {
    const size_t s = (size_t)bounds[lb] * kWideWaves;
    return s < nslices ? s : nslices;
  }
  return (nslices * lb) / nb;
}
template <int P>
struct WideWaves {
  static constexpr int value = P <= 7 ? MI_WIDE_WAVES : 2;
};
template <int P, bool HALO, bool PK>
__global__ __launch_bounds__(kWideBlock) __attribute__((amdgpu_waves_per_eu(WideWaves<P>::value, WideWaves<P>::value))) void k_st_hess_wide(SellView A, const CgState *__restrict__ st,
                                                             const double *__restrict__ V, const double *__restrict__ X,
                                                             const double *__restrict__ Y, const double *__restrict__ S,
                                                             const double *__restrict__ gdir, double *__restrict__ out,
                                                             double *__restrict__ partials, HaloWaitArg<HALO> hwait,
                                                             const int *__restrict__ bounds) {
  constexpr int NS = SymIdx<P>::NS, KC = 3 + NS;
  __shared__ double lds[KC * kWideWaves];
  __shared__ double vt[PK ? 256 : 1];
  __shared__ double Sm[P * P], Mm[P * P];
#ifndef MI_WIDE_ABLATE_MATH  // (the timing experiments produce wrong iterates: their launches must not turn into no-ops)
  if (st && st->mode != CG_RUN) return;
#endif
  if constexpr (HALO) halo_wait(hwait.w);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PK) vt[threadIdx.x] = A.vtab[threadIdx.x];
  if (threadIdx.x < P * P) {
    const int aa = threadIdx.x / P, b = threadIdx.x % P;
    Sm[threadIdx.x] = S[threadIdx.x];
    Mm[threadIdx.x] = gdir[SLOT_GDIR_P + (aa <= b ? SymIdx<P>::at(aa, b) : SymIdx<P>::at(b, aa))];
  }
  __syncthreads();
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = wide_first_slice(A.nslices, lb, nb, bounds), s1 = wide_first_slice(A.nslices, lb + 1, nb, bounds);
  double a[KC];
#pragma unroll
  for (int i = 0; i < KC; ++i) a[i] = 0;
  struct Epi {
    const SellView &A;
    const double *__restrict__ X, *__restrict__ Y, *__restrict__ V;
    double *__restrict__ out;
    const double *Sm, *Mm;
    double (&a)[KC];
    int lane;
    double x[P], y[P], v[P];
    __device__ __forceinline__ unsigned lane_off(size_t slice) const {
      return ((unsigned)slice * 64u + (unsigned)lane < (unsigned)A.n) ? (unsigned)lane * (unsigned)(P * 8) : 0u;
    }
    __device__ __forceinline__ const double *row_of(const double *F, size_t slice, unsigned off) const {
      return reinterpret_cast<const double *>(reinterpret_cast<const char *>(F) + (unsigned)slice * (unsigned)(64 * P * 8) + off);
    }
    __device__ __forceinline__ void begin(size_t slice) {
      const unsigned off = lane_off(slice);
      const double *xs = row_of(X, slice, off), *vs = row_of(V, slice, off), *ys = row_of(Y, slice, off);
#pragma unroll
#ifdef MI_WIDE_ABLATE_OWN  // (timing experiment only: X and Y rows not read -- wrong results)
      for (int c = 0; c < P; ++c) { v[c] = vs[c]; x[c] = v[c] + 1.0; y[c] = v[c] - 1.0; }
#else
      for (int c = 0; c < P; ++c) { x[c] = xs[c]; v[c] = vs[c]; y[c] = ys[c]; }
#endif
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P]) {
      if ((unsigned)slice * 64u + (unsigned)lane >= (unsigned)A.n) return;
      // (S and M are re-read from LDS for every row: loop-invariant code motion must not park 128 doubles in registers)
      asm volatile("" ::: "memory");
      double *os = reinterpret_cast<double *>(reinterpret_cast<char *>(out) + (unsigned)slice * (unsigned)(64 * P * 8) +
                                              lane_off(slice));
#ifndef MI_WIDE_ABLATE_MATH  // (timing experiment only: wrong results)
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += v[aa] * Sm[aa * P + b];
        acc[b] -= t;  // Z = A V - V S
        MI_WIDE_SCHED();
      }
#endif
      double o[P];
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#ifndef MI_WIDE_ABLATE_MATH
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += x[aa] * Mm[aa * P + b];
#endif
        o[b] = acc[b] - t;  // Z - X M
#ifdef MI_WIDE_ABLATE_STORE  // (timing experiment only: one double of the row stored)
        if (b == 0) os[b] = acc[0] + acc[P - 1];
#else
        os[b] = o[b];
#endif
        a[0] += v[b] * o[b]; a[1] += o[b] * o[b]; a[2] += v[b] * v[b];
        MI_WIDE_SCHED();
      }
      double os_[P];  // packed sym(y o' - x (o S)'): the Gram of this output row
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#ifndef MI_WIDE_ABLATE_MATH
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += o[aa] * Sm[aa * P + b];
#endif
        os_[b] = t;
        MI_WIDE_SCHED();
      }
#ifndef MI_WIDE_ABLATE_GRAM
#pragma unroll
      for (int aa = 0; aa < P; ++aa)
#pragma unroll
        for (int b = aa; b < P; ++b) {
          const double gab = y[aa] * o[b] - x[aa] * os_[b];
          const double gba = y[b] * o[aa] - x[b] * os_[aa];
          a[3 + SymIdx<P>::at(aa, b)] += (aa == b) ? gab : .5 * (gab + gba);
        }
#else
      a[3] += y[0] * os_[0] + x[P - 1];
#endif
    }
  } epi{A, X, Y, V, out, Sm, Mm, a, lane, {}, {}, {}};
  const int wu = __builtin_amdgcn_readfirstlane(w);
  sell_stream<P, HALO, PK, Epi, kWideWaves, (P >= MI_WIDE_CHUNK_FROM ? MI_WIDE_CHUNK : MI_SPMM_CHUNK)>(A, s0 + (size_t)wu, s1, lane, V, vt, epi);
  block_partials_store_nw<KC, kWideWaves>(a, lds, partials);
}

// The wide-row pass in WINDOW form (r06): the same epilogue as k_st_hess_wide (S and M in LDS, 3 + P (P +
// 1) / 2 partial components), but the product A V through spmm_core.h's sell_window instead of sell_stream -- the near
// rows of V come out of an LDS ring that is filled by coalesced 512-byte loads of whole 64-row chunks (a chunk of a
// row-major field is 64 P contiguous doubles), only the two far rows of a row are gathered lane by lane, and the far
// columns are computed (pure far structure).  What the lane-per-row form pays for -- seven neighbour gathers of P
// strided 8-byte loads each, every wave instruction touching 64 half-lines (profiles/r05_wide_ablation.txt) -- is gone;
// the own X and Y rows remain lane-per-row loads.  One context, packed matrix with a window (sparse.hip build_window),
// no halo.  The ring is dynamic LDS sized for the matrix's own window: (nc 64 + 1 + far slots) P doubles.
#ifndef MI_WIDEWIN_COAL
#define MI_WIDEWIN_COAL 0   // 1: own X / Y rows as coalesced memory images, transposed through the wave's far slots (measured slower)
#endif
#ifndef MI_WIDEWIN_X16
#define MI_WIDEWIN_X16 1    // P = 6: own X / Y rows by 16-byte asm loads (k_st_hess_widewin Epi::request)
#endif
#ifndef MI_WIDEWIN_3WAVES_UPTO
#define MI_WIDEWIN_3WAVES_UPTO 3   // (experiment: 4 = three workgroups per CU at p = 4)
#endif
#ifndef MI_WIDEWIN_2WAVES_UPTO
#define MI_WIDEWIN_2WAVES_UPTO 7   // widths held to 256 VGPRs (two workgroups per CU); wider: one workgroup per CU
#endif
// ring rows of the wide window form (spmm_core.h sell_window RS): at P = 8 rows 64 bytes apart put the row-by-row reads
// of a wave on a sixteenth of the LDS banks (218 us; 9 doubles apart: 102 us -- still behind the quad layout's 80, so
// p = 8 does not take this form by default); at P = 6 the padding changes nothing (57.7 us either way) and costs
// 11-14 registers: not applied
template <int P>
struct WideRing {
  static constexpr int stride = (P == 8) ? 9 : P;
};
template <int P, int HW, bool FARD>
__global__ __launch_bounds__(kWinBlock) __attribute__((amdgpu_waves_per_eu(P <= MI_WIDEWIN_3WAVES_UPTO ? 3 : P <= MI_WIDEWIN_2WAVES_UPTO ? 2 : 1, P <= MI_WIDEWIN_3WAVES_UPTO ? 3 : P <= MI_WIDEWIN_2WAVES_UPTO ? 2 : 1))) void k_st_hess_widewin(
    SellView A, WinView Wv, const CgState *__restrict__ st, const double *__restrict__ V, const double *__restrict__ X,
    const double *__restrict__ Y, const double *__restrict__ S, const double *__restrict__ gdir, double *__restrict__ out,
    double *__restrict__ partials) {
  static_assert(kWideWaves == kWinWaves, "one workgroup shape");
  constexpr int KC = DirComps<P>::value;  // 3 + P (P + 1) / 2 (P = 4: padded to 16, the pads stay zero)
  __shared__ double lds[KC * kWideWaves];
  __shared__ double vt[256];
  __shared__ double Sm[P * P], Mm[P * P];
  extern __shared__ __attribute__((aligned(16))) double ring_dyn[];
#if !defined(MI_WIDE_ABLATE_MATH) && !defined(MI_WIDE_ABLATE_OWN) && !defined(MI_WIDE_ABLATE_STORE)
  if (st && st->mode != CG_RUN) return;   // (the timing experiments produce wrong iterates: their launches must not turn into no-ops)
#endif
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  vt[threadIdx.x] = A.vtab[threadIdx.x];
  if (threadIdx.x < P * P) {
    const int aa = threadIdx.x / P, b = threadIdx.x % P;
    Sm[threadIdx.x] = S[threadIdx.x];
    Mm[threadIdx.x] = gdir[SLOT_GDIR_P + (aa <= b ? SymIdx<P>::at(aa, b) : SymIdx<P>::at(b, aa))];
  }
  // (sell_window's barrier behind the ring fill publishes vt, Sm and Mm too)
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  double a[KC];
#pragma unroll
  for (int i = 0; i < KC; ++i) a[i] = 0;
  struct Epi {
    const SellView &A;
    const double *__restrict__ X, *__restrict__ Y;
    double *__restrict__ out;
    const double *Sm, *Mm;
    double (&a)[KC];
    int lane;
    double xn[P], yn[P];
    __device__ __forceinline__ unsigned lane_off(size_t slice) const {
      return ((unsigned)slice * 64u + (unsigned)lane < (unsigned)A.n) ? (unsigned)lane * (unsigned)(P * 8) : 0u;
    }
    __device__ __forceinline__ const double *row_of(const double *F, size_t slice, unsigned off) const {
      return reinterpret_cast<const double *>(reinterpret_cast<const char *>(F) + (unsigned)slice * (unsigned)(64 * P * 8) + off);
    }
#if MI_WIDEWIN_COAL && !defined(MI_WIDE_ABLATE_OWN)
    // The slice's rows of X and Y are 64 P consecutive doubles each: P coalesced 512-byte loads per field (lane l: the
    // doubles l, 64 + l, ...) instead of P strided 8-byte loads per lane -- at 6 doubles a row a wave instruction of
    // those touches 24 lines, and the texture path, not the bytes, then paces the pass (profiles/r06_widewin_ablation.txt:
    // -7 us without them).  The images become rows again through the wave's far slots, free once the entries are done.
    // MEASURED (r06, same box, alternating): p = 5 / 6 / 7: 46.3 / 56.1 / 74.8 us without -> 47.5 / 57.0 / 82.3 us with:
    // the 4 P extra LDS operations per row cost more than the lines save.  Not the default.
    using wants_scratch = void;
    __device__ __forceinline__ void request(size_t slice) {
      const unsigned nPbytes = (unsigned)(A.n * P * 8);
      const unsigned b0 = (unsigned)slice * (unsigned)(64 * P * 8) + (unsigned)lane * 8u;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const unsigned b = b0 + (unsigned)c * 512u;
        xn[c] = pinned_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(X) + (b < nPbytes ? b : 0u)));
      }
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const unsigned b = b0 + (unsigned)c * 512u;
        yn[c] = pinned_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(Y) + (b < nPbytes ? b : 0u)));
      }
    }
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P], const double (&v)[P], LdsDouble *scratch) {
      constexpr int RS = WideRing<P>::stride;
      image_store<P, RS>(scratch, lane, xn);
      image_store<P, RS>(scratch + 64 * RS, lane, yn);
      image_row<P, RS>(scratch, lane, xn);   // (one wave: its LDS operations complete in order)
      image_row<P, RS>(scratch + 64 * RS, lane, yn);
      end(slice, acc, v);
    }
#elif MI_WIDEWIN_X16 && !defined(MI_WIDE_ABLATE_OWN)
    // P = 4, 6: a row is two / three 16-byte pieces -- P / 2 global_load_dwordx4 per field instead of P 8-byte loads (half
    // the texture-path instructions, half the hits on lines still pending: p = 6 57.5 -> 55.8 us, same box, alternating;
    // at P = 8 in this form 100 -> 105 us, not used).  There is no ordered (pinned) 16-byte load to be had from the
    // compiler, so these are asm statements: absent from hipcc's s_waitcnt bookkeeping, waited for by the explicit
    // vmcnt(0) in arrive() (they are the newest loads of the tile: a wait hipcc computes for an older load only gets longer).
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d xq[3], yq[3];
    __device__ __forceinline__ void request(size_t slice) {
      if constexpr (P == 4 || P == 6) {
        const unsigned off = (unsigned)slice * (unsigned)(64 * P * 8) + lane_off(slice);
#pragma unroll
        for (int c = 0; c < P / 2; ++c)
          asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(xq[c]) : "v"(off), "s"(X), "n"(16 * c) : "memory");
#pragma unroll
        for (int c = 0; c < P / 2; ++c)
          asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(yq[c]) : "v"(off), "s"(Y), "n"(16 * c) : "memory");
      } else {
        const unsigned off = lane_off(slice);
        const double *xs = row_of(X, slice, off), *ys = row_of(Y, slice, off);
#pragma unroll
        for (int c = 0; c < P; ++c) xn[c] = pinned_load(xs + c);
#pragma unroll
        for (int c = 0; c < P; ++c) yn[c] = pinned_load(ys + c);
      }
    }
    __device__ __forceinline__ void arrive() {
      if constexpr (P == 6)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(yq[0]), "+v"(yq[1]), "+v"(yq[2])::"memory");
      else if constexpr (P == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(yq[0]), "+v"(yq[1])::"memory");
      if constexpr (P == 4 || P == 6) {
#pragma unroll
        for (int c = 0; c < P / 2; ++c) { xn[2 * c] = xq[c].x; xn[2 * c + 1] = xq[c].y; yn[2 * c] = yq[c].x; yn[2 * c + 1] = yq[c].y; }
      }
    }
#else
    __device__ __forceinline__ void request(size_t slice) {
      const unsigned off = lane_off(slice);
      const double *xs = row_of(X, slice, off), *ys = row_of(Y, slice, off);
#pragma unroll
#ifdef MI_WIDE_ABLATE_OWN   // (timing experiment only: X and Y rows not read -- wrong results)
      for (int c = 0; c < P; ++c) { xn[c] = (double)(off + c); yn[c] = xn[c] - 1.0; }
      (void)xs; (void)ys;
#else
      for (int c = 0; c < P; ++c) xn[c] = pinned_load(xs + c);
#pragma unroll
      for (int c = 0; c < P; ++c) yn[c] = pinned_load(ys + c);
#endif
    }
#endif
    __device__ __forceinline__ void end(size_t slice, double (&acc)[P], const double (&v)[P]) {
#if MI_WIDEWIN_X16 && !MI_WIDEWIN_COAL && !defined(MI_WIDE_ABLATE_OWN)
      arrive();  // (every lane: the wait is not a matter of the lane's row)
#endif
      if ((unsigned)slice * 64u + (unsigned)lane >= (unsigned)A.n) return;
      // (S and M are re-read from LDS for every row: loop-invariant code motion must not park 128 doubles in registers)
      asm volatile("" ::: "memory");
      double *os = reinterpret_cast<double *>(reinterpret_cast<char *>(out) + (unsigned)slice * (unsigned)(64 * P * 8) +
                                              lane_off(slice));
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#ifndef MI_WIDE_ABLATE_MATH
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += v[aa] * Sm[aa * P + b];
#endif
        acc[b] -= t;  // Z = A V - V S
        MI_WIDE_SCHED();
      }
      double o[P];
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#ifndef MI_WIDE_ABLATE_MATH
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += xn[aa] * Mm[aa * P + b];
#endif
        o[b] = acc[b] - t;  // Z - X M
#ifdef MI_WIDE_ABLATE_STORE  // (timing experiment only: one double of the row stored)
        if (b == 0) os[b] = acc[0] + acc[P - 1];
#else
        os[b] = o[b];  // (non-temporal stores / loads of the own rows: 58 -> 63-74 us at p = 6, EXPERIMENTS.md r06)
#endif
        a[0] += v[b] * o[b]; a[1] += o[b] * o[b]; a[2] += v[b] * v[b];
        MI_WIDE_SCHED();
      }
      double os_[P];  // packed sym(y o' - x (o S)'): the Gram of this output row
#pragma unroll
      for (int b = 0; b < P; ++b) {
        double t = 0;
#ifndef MI_WIDE_ABLATE_MATH
#pragma unroll
        for (int aa = 0; aa < P; ++aa) t += o[aa] * Sm[aa * P + b];
#endif
        os_[b] = t;
        MI_WIDE_SCHED();
      }
#pragma unroll
      for (int aa = 0; aa < P; ++aa)
#pragma unroll
        for (int b = aa; b < P; ++b) {
          const double gab = yn[aa] * o[b] - xn[aa] * os_[b];
          const double gba = yn[b] * o[aa] - xn[b] * os_[aa];
          a[3 + SymIdx<P>::at(aa, b)] += (aa == b) ? gab : .5 * (gab + gba);
        }
    }
  } epi{A, X, Y, out, Sm, Mm, a, lane, {}, {}};
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int ntiles = (int)((A.nslices + kWinWaves - 1) / kWinWaves), per = (ntiles + (int)nb - 1) / (int)nb;
  int t0 = (int)lb * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  if (Wv.bounds) {
    t0 = scalar_int(Wv.bounds, lb);
    t1 = scalar_int(Wv.bounds, lb + 1);
  }
  if (t0 >= t1) __syncthreads();  // (sell_window returns at once: the tables above still need their barrier)
  sell_window<P, HW, false, FARD, false, Epi, WideRing<P>::stride, FARD && MI_WIN_FARC>(A, Wv, t0, t1, wu, lane, V, vt, ring_dyn, epi);
  __syncthreads();
  block_partials_store_nw<KC, kWideWaves>(a, lds, partials);
}

// The same pass in the QUAD layout of spmm_core.h (sell_stream_quad): lane (g, q) holds columns 2q, 2q + 1 of the rows
// 16 t + g of its wave's slice, every load and store of a row is 64 contiguous bytes per quad.  A row's full width --
// the left operands of the three products and of the Gram -- comes from the quad's four lanes by DPP broadcasts; S and M
// sit in LDS padded to 8 x 8 with zero columns (a lane reads its two columns of a row of the matrix as one 16-byte
// read); the Gram of the output is accumulated RAW (y_a o_b - x_a (o S)_b for the lane's two b: 2 P accumulators instead
// of P (P + 1) / 2) and symmetrised once, at the end, through LDS.
template <int CTRL>
__device__ __forceinline__ double dpp_quad_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// full[2 i + c] = the value of column 2 i + c held by lane i of the quad
template <int P>
__device__ __forceinline__ void quad_row(const double (&own)[2], double (&full)[8]) {
  full[0] = dpp_quad_f64<0x00>(own[0]); full[1] = dpp_quad_f64<0x00>(own[1]);
  full[2] = dpp_quad_f64<0x55>(own[0]); full[3] = dpp_quad_f64<0x55>(own[1]);
  full[4] = dpp_quad_f64<0xAA>(own[0]);
  if constexpr (P > 5) full[5] = dpp_quad_f64<0xAA>(own[1]);
  if constexpr (P > 6) full[6] = dpp_quad_f64<0xFF>(own[0]);
  if constexpr (P > 7) full[7] = dpp_quad_f64<0xFF>(own[1]);
}
template <int P, bool HALO, bool PK>
__global__ __launch_bounds__(kWideBlock) __attribute__((amdgpu_waves_per_eu(MI_WIDE_QUAD_WAVES, MI_WIDE_QUAD_WAVES))) void k_st_hess_wideq(SellView A, const CgState *__restrict__ st,
                                                             const double *__restrict__ V, const double *__restrict__ X,
                                                             const double *__restrict__ Y, const double *__restrict__ S,
                                                             const double *__restrict__ gdir, double *__restrict__ out,
                                                             double *__restrict__ partials, HaloWaitArg<HALO> hwait,
                                                             const int *__restrict__ bounds) {
  constexpr int NS = SymIdx<P>::NS, KC = 3 + NS;
  __shared__ double lds[3 * kWideWaves];
  __shared__ double graw[kWideWaves][P][8];
  __shared__ double vt[PK ? 256 : 1];
  __shared__ __attribute__((aligned(16))) double Sm[P * 8], Mm[P * 8];
#ifndef MI_WIDEQ_ABLATE_EPI
  if (st && st->mode != CG_RUN) return;
#endif
  if constexpr (HALO) halo_wait(hwait.w);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PK) vt[threadIdx.x] = A.vtab[threadIdx.x];
  if (threadIdx.x < P * 8) {
    const int aa = threadIdx.x >> 3, b = threadIdx.x & 7;
    Sm[threadIdx.x] = b < P ? S[aa * P + b] : 0.0;
    Mm[threadIdx.x] = b < P ? gdir[SLOT_GDIR_P + (aa <= b ? SymIdx<P>::at(aa, b) : SymIdx<P>::at(b, aa))] : 0.0;
  }
  __syncthreads();
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = wide_first_slice(A.nslices, lb, nb, bounds), s1 = wide_first_slice(A.nslices, lb + 1, nb, bounds);
  double a[3] = {0, 0, 0};
  double G[P][2];
#pragma unroll
  for (int i = 0; i < P; ++i) G[i][0] = G[i][1] = 0;
  struct Epi {
    const SellView &A;
    const double *__restrict__ X, *__restrict__ Y, *__restrict__ V;
    double *__restrict__ out;
    const double *Sm, *Mm;
    double (&a)[3];
    double (&G)[P][2];
    int q, g;
    unsigned o0, o1;
    __device__ __forceinline__ unsigned row_off(size_t slice, int t) const {  // (rows past the end re-read the slice's first)
      const unsigned r = (unsigned)slice * 64u + (unsigned)(16 * t + g);
      return (r < (unsigned)A.n ? r : (unsigned)slice * 64u) * (unsigned)(P * 8);
    }
    __device__ __forceinline__ void begin(size_t slice, int t, double (&x)[2], double (&v)[2], double (&y)[2]) const {
      const unsigned off = row_off(slice, t);
      QuadCols<P>::load(reinterpret_cast<const char *>(X), off, o0, o1, x[0], x[1]);
      QuadCols<P>::load(reinterpret_cast<const char *>(V), off, o0, o1, v[0], v[1]);
      QuadCols<P>::load(reinterpret_cast<const char *>(Y), off, o0, o1, y[0], y[1]);
    }
    __device__ __forceinline__ void end(size_t slice, int t, double (&acc)[2], const double (&x)[2], const double (&v)[2],
                                        const double (&y)[2]) {
      const bool c0 = 2 * q < P, c1 = 2 * q + 1 < P;
      const double2 *S2 = reinterpret_cast<const double2 *>(Sm) + q, *M2 = reinterpret_cast<const double2 *>(Mm) + q;
      // (S and M are re-read from LDS for every row: loop-invariant code motion must not park them in registers)
      asm volatile("" ::: "memory");
      const bool live = (unsigned)slice * 64u + (unsigned)(16 * t + g) < (unsigned)A.n;
      const bool u0 = live && c0, u1 = live && c1;
      const double vv[2] = {u0 ? v[0] : 0.0, u1 ? v[1] : 0.0};
      const double xx[2] = {u0 ? x[0] : 0.0, u1 ? x[1] : 0.0};
      const double yy[2] = {u0 ? y[0] : 0.0, u1 ? y[1] : 0.0};
      double z[2] = {u0 ? acc[0] : 0.0, u1 ? acc[1] : 0.0};
#ifdef MI_WIDEQ_ABLATE_EPI  // (timing experiment only: the row epilogue reduced to the store -- wrong results)
      {
        char *ob = reinterpret_cast<char *>(out);
        const unsigned ro = row_off(slice, t);
        if (u0) *reinterpret_cast<double2 *>(ob + (ro + o0)) = make_double2(z[0] + xx[0] + yy[0] + vv[0], z[1] + xx[1] + yy[1] + vv[1]);
        a[0] += z[0];
        return;
      }
#endif
      double vf[8], xf[8], yf[8], of[8];
      quad_row<P>(vv, vf);
      quad_row<P>(xx, xf);
#pragma unroll
      for (int aa = 0; aa < P; ++aa) {  // Z = A V - V S, then out = Z - X M (the lane's two columns)
        const double2 sc = S2[aa * 4];
        z[0] -= vf[aa] * sc.x; z[1] -= vf[aa] * sc.y;
      }
      MI_WIDE_SCHED();
#pragma unroll
      for (int aa = 0; aa < P; ++aa) {
        const double2 mc = M2[aa * 4];
        z[0] -= xf[aa] * mc.x; z[1] -= xf[aa] * mc.y;
      }
      MI_WIDE_SCHED();
      {
        char *ob = reinterpret_cast<char *>(out);
        const unsigned ro = row_off(slice, t);
        if constexpr (P % 2 == 0) {
          if (u0) *reinterpret_cast<double2 *>(ob + (ro + o0)) = make_double2(z[0], z[1]);
        } else {
          if (u0) *reinterpret_cast<double *>(ob + (ro + o0)) = z[0];
          if (u1) *reinterpret_cast<double *>(ob + (ro + o1)) = z[1];
        }
      }
      a[0] += vv[0] * z[0]; a[0] += vv[1] * z[1];
      a[1] += z[0] * z[0]; a[1] += z[1] * z[1];
      a[2] += vv[0] * vv[0]; a[2] += vv[1] * vv[1];
      quad_row<P>(z, of);
      double os_[2] = {0, 0};  // (o S), the lane's two columns
#pragma unroll
      for (int aa = 0; aa < P; ++aa) {
        const double2 sc = S2[aa * 4];
        os_[0] += of[aa] * sc.x; os_[1] += of[aa] * sc.y;
      }
      MI_WIDE_SCHED();
      quad_row<P>(yy, yf);
#pragma unroll
      for (int aa = 0; aa < P; ++aa) {  // raw Gram of this output row: y_a o_b - x_a (o S)_b
        G[aa][0] += yf[aa] * z[0] - xf[aa] * os_[0];
        G[aa][1] += yf[aa] * z[1] - xf[aa] * os_[1];
      }
    }
  } epi{A, X, Y, V, out, Sm, Mm, a, G, lane & 3, lane >> 2, QuadCols<P>::off0(lane & 3), QuadCols<P>::off1(lane & 3)};
  const int wu = __builtin_amdgcn_readfirstlane(w);
  sell_stream_quad<P, HALO, PK, Epi, kWideWaves, MI_WIDE_QUAD_CHUNK>(A, s0 + (size_t)wu, s1, lane, V, vt, epi);
  // the wave's raw Gram: sum over the 16 quads (lane bits 2..5), lanes 0..3 hold columns 2q, 2q + 1
#pragma unroll
  for (int aa = 0; aa < P; ++aa)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double t = G[aa][c];
#pragma unroll
      for (int m = 4; m < 64; m <<= 1) t += __shfl_xor(t, m, 64);
      if (lane < 4) graw[w][aa][2 * lane + c] = t;
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double t = wave_reduce_sum(a[k]);
    if (lane == 0) lds[k * kWideWaves + w] = t;
  }
  __syncthreads();
  if (threadIdx.x < KC) {
    double t[kWideWaves];
    if (threadIdx.x < 3) {
#pragma unroll
      for (int i = 0; i < kWideWaves; ++i) t[i] = lds[threadIdx.x * kWideWaves + i];
    } else {
      int aa = 0, idx = (int)threadIdx.x - 3;  // packed index -> (aa, b), aa <= b
      while (idx >= P - aa) { idx -= P - aa; ++aa; }
      const int b = aa + idx;
#pragma unroll
      for (int i = 0; i < kWideWaves; ++i) t[i] = aa == b ? graw[i][aa][aa] : .5 * (graw[i][aa][b] + graw[i][b][aa]);
    }
#pragma unroll
    for (int span = 1; span < kWideWaves; span *= 2)
#pragma unroll
      for (int i = 0; i + span < kWideWaves; i += 2 * span) t[i] = t[i] + t[i + span];
    partials[(size_t)threadIdx.x * kMaxRows + blockIdx.x] = t[0];
  }
}

// Gram partial rows of two dense n x P fields.
//   VARIANT 0: gram(X,Z)              1: Y = X + Z written to out, gram(Y,Y)
//           2: Z' = dinv_rows .* Z written to out, gram(X,Z')
//   SYM: symmetrised partials (P(P+1)/2 components) else raw (P*P components)
template <int P, int VARIANT, bool SYM>
__global__ __launch_bounds__(StBlk<P>::threads) void k_st_gram(size_t n, const double *__restrict__ X,
                                                    const double *__restrict__ Zin,
                                                    const double *__restrict__ dinv,
                                                    double *__restrict__ out,
                                                    double *__restrict__ partials) {
  constexpr int BLK = StBlk<P>::threads, NW = StBlk<P>::waves;
  __shared__ double lds[P * P * NW];
  double G[P * P];
#pragma unroll
  for (int i = 0; i < P * P; ++i) G[i] = 0;
  const size_t stride = (size_t)gridDim.x * BLK;
  for (size_t row = (size_t)blockIdx.x * BLK + threadIdx.x; row < n; row += stride) {
    double x[P], z[P];
#pragma unroll
    for (int c = 0; c < P; ++c) { x[c] = X[row * P + c]; z[c] = Zin[row * P + c]; }
    if (VARIANT == 1) {
#pragma unroll
      for (int c = 0; c < P; ++c) { x[c] = x[c] + z[c]; z[c] = x[c]; out[row * P + c] = x[c]; }
    } else if (VARIANT == 2) {
      const double d = dinv[row];
#pragma unroll
      for (int c = 0; c < P; ++c) { z[c] = d * z[c]; out[row * P + c] = z[c]; }
    }
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
      for (int b = 0; b < P; ++b) G[a * P + b] += x[a] * z[b];
  }
  if (SYM) store_sym_partials<P>(G, lds, partials);
  else block_partials_store_w<P * P, NW>(G, lds, partials);
}

// prologue: M = sym Gram (re-reduced by every workgroup, or read from all-reduced slots);
// body: out = Z - X M;  DOTS: partial rows of <Vin,out>, <out,out>, <Vin,Vin>;  M_out (nullable): M
template <int P, bool DOTS, bool FROM_SLOTS>
__global__ __launch_bounds__(StBlk<P>::threads) void k_st_finish(size_t n, const CgState *__restrict__ st,
                                                      const double *__restrict__ X,
                                                      const double *__restrict__ Z,
                                                      const double *__restrict__ Vin,
                                                      const double *__restrict__ gram_partials, int count,
                                                      const double *__restrict__ slots,
                                                      double *__restrict__ M_out,
                                                      double *__restrict__ out,
                                                      double *__restrict__ partials) {
  constexpr int BLK = StBlk<P>::threads, NW = StBlk<P>::waves;
  constexpr bool WIDE = P > 4;  // (256-thread workgroups, M in LDS: stiefel_core.h StBlk)
  __shared__ double lds[SymIdx<P>::NS * (kWaves + 1) + 3 * kWaves];
  __shared__ double MmL[WIDE ? P * P : 1];
  if (st && st->mode != CG_RUN) return;
  const size_t stride = (size_t)gridDim.x * BLK;
  const size_t row0 = (size_t)blockIdx.x * BLK + threadIdx.x;
  // prefetch the first grid-stride step before the prologue's reduction (hides its latency)
  double x[P], z[P], vi[P];
#pragma unroll
  for (int c = 0; c < P; ++c) { x[c] = 0; z[c] = 0; vi[c] = 0; }
  if (row0 < n) {
#pragma unroll
    for (int c = 0; c < P; ++c) {
      x[c] = X[row0 * P + c];
      z[c] = Z[row0 * P + c];
      if (DOTS) vi[c] = Vin[row0 * P + c];
    }
  }
  double Mm[P * P];
  load_sym<P, FROM_SLOTS>(gram_partials, count, slots, Mm, lds);
  if (M_out && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < P * P; ++i) M_out[i] = Mm[i];
  }
  if (WIDE) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < P * P; ++i) MmL[i] = Mm[i];
    }
    __syncthreads();
  }
  double a[3] = {0, 0, 0};
  for (size_t row = row0; row < n;) {
    const size_t rnext = row + stride;
    double xn[P], zn[P], vn[P];
#pragma unroll
    for (int c = 0; c < P; ++c) { xn[c] = x[c]; zn[c] = z[c]; vn[c] = vi[c]; }
    if (rnext < n) {
#pragma unroll
      for (int c = 0; c < P; ++c) {
        xn[c] = X[rnext * P + c];
        zn[c] = Z[rnext * P + c];
        if (DOTS) vn[c] = Vin[rnext * P + c];
      }
    }
    if (WIDE) asm volatile("" ::: "memory");  // (M is re-read from LDS per row, not parked in 64 registers)
#pragma unroll
    for (int b = 0; b < P; ++b) {
      double t = 0;
#pragma unroll
      for (int aa = 0; aa < P; ++aa) t += x[aa] * (WIDE ? MmL[aa * P + b] : Mm[aa * P + b]);
      const double o = z[b] - t;
      out[row * P + b] = o;
      if (DOTS) { a[0] += vi[b] * o; a[1] += o * o; a[2] += vi[b] * vi[b]; }
      if (WIDE) MI_WIDE_SCHED();
    }
    row = rnext;
#pragma unroll
    for (int c = 0; c < P; ++c) { x[c] = xn[c]; z[c] = zn[c]; vi[c] = vn[c]; }
  }
  if (DOTS) block_partials_store_w<3, NW>(a, lds, partials);
}

// retraction finish: Y <- Y (Y'Y)^-1/2, the P x P inverse square root computed once per workgroup
template <int P, bool FROM_SLOTS>
__global__ __launch_bounds__(StBlk<P>::threads) void k_st_polar(size_t n, double *__restrict__ Y,
                                                     const double *__restrict__ gram_partials, int count,
                                                     const double *__restrict__ slots) {
  constexpr int BLK = StBlk<P>::threads;
  constexpr bool WIDE = P > 4;
  __shared__ double lds[SymIdx<P>::NS * (kWaves + 1)];
  __shared__ double Minv[P * P];
  __shared__ double jac[WIDE ? 4 * P * P : 1];  // WIDE: the Gram and the three work matrices of dev_sym_invsqrt_wave
  double G[P * P];
  load_sym<P, FROM_SLOTS>(gram_partials, count, slots, G, lds);
  if (WIDE) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < P * P; ++i) jac[i] = G[i];
    }
    __syncthreads();
    if (threadIdx.x < 64) dev_sym_invsqrt_wave<P>(jac, jac + P * P, Minv);
  } else {
    if (threadIdx.x == 0) dev_sym_invsqrt<P>(G, Minv);
  }
  __syncthreads();
  double Mm[WIDE ? 1 : P * P];
  if (!WIDE) {
#pragma unroll
    for (int i = 0; i < P * P; ++i) Mm[i] = Minv[i];
  }
  const size_t stride = (size_t)gridDim.x * BLK;
  for (size_t row = (size_t)blockIdx.x * BLK + threadIdx.x; row < n; row += stride) {
    double y[P];
#pragma unroll
    for (int c = 0; c < P; ++c) y[c] = Y[row * P + c];
    if (WIDE) asm volatile("" ::: "memory");
#pragma unroll
    for (int b = 0; b < P; ++b) {
      double t = 0;
#pragma unroll
      for (int a = 0; a < P; ++a) t += y[a] * (WIDE ? Minv[a * P + b] : Mm[WIDE ? 0 : a * P + b]);
      Y[row * P + b] = t;
    }
  }
}

// ---- launch helpers (dispatch on p) --------------------------------------------------------
#define DISPATCH_P(p, ...)                                          \
  switch (p) {                                                      \
    case 1: { constexpr int P = 1; __VA_ARGS__; } break;            \
    case 2: { constexpr int P = 2; __VA_ARGS__; } break;            \
    case 3: { constexpr int P = 3; __VA_ARGS__; } break;            \
    case 4: { constexpr int P = 4; __VA_ARGS__; } break;            \
    case 5: { constexpr int P = 5; __VA_ARGS__; } break;            \
    case 6: { constexpr int P = 6; __VA_ARGS__; } break;            \
    case 7: { constexpr int P = 7; __VA_ARGS__; } break;            \
    case 8: { constexpr int P = 8; __VA_ARGS__; } break;            \
    default: set_error("p must be in [1,%d], got %d", kMaxP, p); return MI_ERR_INVALID_ARGUMENT; \
  }
// the instantiations tuned for narrow rows (the one-pass Hessian in its p <= 4 forms: per-thread P x P matrices in
// registers, 1024-thread workgroups, the LDS-window forms); p >= 5 takes k_st_hess_wide
#define DISPATCH_P4(p, ...)                                         \
  switch (p) {                                                      \
    case 1: { constexpr int P = 1; __VA_ARGS__; } break;            \
    case 2: { constexpr int P = 2; __VA_ARGS__; } break;            \
    case 3: { constexpr int P = 3; __VA_ARGS__; } break;            \
    case 4: { constexpr int P = 4; __VA_ARGS__; } break;            \
    default: set_error("internal: narrow-row kernel asked for p = %d", p); return MI_ERR_INTERNAL; \
  }

inline int row_grid(const mi_ctx *ctx, size_t n) { return grid_for(ctx, n, 2); }
// rows wider than 4 doubles run 256-thread workgroups (StBlk): up to kMaxRows of them (one partial row each)
inline int row_grid(const mi_ctx *ctx, size_t n, int p) {
  if (p <= 4 || ctx->uniform_grid) return row_grid(ctx, n);
  size_t blocks = std::max<size_t>(1, (n + 511) / 512);
  blocks = std::min<size_t>(blocks, std::min<size_t>(kMaxRows, 2 * (size_t)ctx->max_grid));
  return (int)blocks;
}
inline int nsym(int p) { return p * (p + 1) / 2; }

// sharded only: partial rows -> slots -> all-reduce
int sharded_reduce(mi_ctx *ctx, int count, int k, double *slots) {
  return reduce_rows_allreduce(ctx, ctx->partials2, count, k, slots);
}

int launch_spmm_gram(mi_ctx *ctx, const mi_csr *A, int p, const CgState *st, const double *V,
                     const double *X, const double *S, double *Z, int *count) {
  const size_t ngroups = sell_groups(A);
  int grid = uniform_grid(ctx, ngroups);
  MI_TRY(comm_halo_exchange(ctx, A, p, V));
  SellView view = sell_view(A);  // after the exchange: it selects the halo buffer the rows landed in
  KScope ks(ctx, MI_K_STIEFEL_SPMM_GRAM);
  const bool no_stream = ctx->cfg.no_spmm_stream;
  // (rows wider than 4 doubles exist in the pipelined form only: 256-thread workgroups, stiefel_core.h StBlk)
  MI_REQUIRE(p <= 4 || sell_stream_ok(A, p), "Stiefel rows of %d doubles need fields below 4 GiB", p);
  if ((!no_stream || p > 4) && sell_stream_ok(A, p)) {
    if (!ctx->uniform_grid && grid > 256) grid = 256;  // one workgroup per CU, one round
    if (p > 4 && !ctx->uniform_grid)                   // 256-thread workgroups: two per CU
      grid = (int)std::max<size_t>(1, std::min<size_t>((A->nslices + 3) / 4, std::min<size_t>(2 * (size_t)ctx->num_cu, 2 * (size_t)ctx->max_grid)));
#define SG(HL, PKV)                                                                                            \
  DISPATCH_P(p, hipLaunchKernelGGL((k_st_spmm_gram_stream<P, HL, PKV>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, \
                                   view, st, V, X, S, Z, ctx->partials2))
    if (A->halo) { if (A->pk) { SG(true, true); } else { SG(true, false); } }
    else { if (A->pk) { SG(false, true); } else { SG(false, false); } }
#undef SG
  } else {
    DISPATCH_P4(p, hipLaunchKernelGGL(k_st_spmm_gram<P>, dim3(grid), dim3(kBlock), 0, ctx->stream, view, st, V,
                                      X, S, Z, ctx->partials2));
  }
  *count = grid;
  return MI_OK;
}

// out = Z - X sym(Gram) where the symmetrised Gram partial rows are in ctx->partials2
int launch_finish(mi_ctx *ctx, size_t n, int p, const CgState *st, const double *X, const double *Z,
                  const double *Vin, int count, double *M_out, double *out, bool dots, int *nparts) {
  const int grid = row_grid(ctx, n, p);
  double *slots = ctx->scalars + SLOT_GRAM;
  // several ranks: all-reduce the Gram partial rows themselves and keep the prologue re-reduction
  // (no one-workgroup reduce kernel); the slot variant stays reachable through MI355OPT_FORCE_SLOT_PATH
  const bool sharded = slot_mode(ctx);
  if (rows_mode(ctx)) MI_TRY(comm_allreduce_rows(ctx, ctx->partials2, nsym(p)));
  if (sharded) MI_TRY(sharded_reduce(ctx, count, nsym(p), slots));
  KScope ks(ctx, MI_K_STIEFEL_FINISH_DOTS);
#define FIN(D, F)                                                                                       \
  DISPATCH_P(p, hipLaunchKernelGGL((k_st_finish<P, D, F>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, n, st, \
                                   X, Z, Vin, (const double *)ctx->partials2, count,                    \
                                   (const double *)slots, M_out, out, ctx->partials))
  if (dots) {
    if (sharded) { FIN(true, true); } else { FIN(true, false); }
  } else {
    if (sharded) { FIN(false, true); } else { FIN(false, false); }
  }
#undef FIN
  if (nparts) *nparts = grid;
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int check_np(mi_ctx *ctx, size_t n, int p, const mi_vec *a, const mi_vec *b, const mi_vec *c) {
  MI_REQUIRE(ctx, "ctx is null");
  MI_REQUIRE(p >= 1 && p <= kMaxP, "p must be in [1,%d], got %d", kMaxP, p);
  const mi_vec *vs[3] = {a, b, c};
  for (const mi_vec *v : vs) {
    if (!v) continue;
    MI_REQUIRE(v->ctx == ctx, "vector belongs to another context");
    MI_REQUIRE(v->n == n * (size_t)p, "expected an n x p = %zu x %d field, got %zu doubles", n, p, v->n);
  }
  return MI_OK;
}

}  // namespace

struct mi_stiefel_rq {
  mi_ctx *ctx;
  const mi_csr *A;
  size_t n;
  int p;
  double *S_dev;  // P*P: sym(X'AX) of the last model() call
  mi_vec *Z;      // n x p scratch
  mi_vec *Y;      // A X of the last model() call (fixed during the inner solve; mi_op::dirgram)
  mi_dirgram dg;
  mi_op hess;     // borrowed operator object bound to X
  const mi_vec *X;
  uint64_t X_serial = 0;  // (handles are recycled: "bound to this X" means this handle AND this vector's serial)
  bool bound_to(const mi_vec *x) const { return X == x && x && X_serial == x->serial; }
  // mi_stiefel_rq_trial: what it already worked out at the trial point (A X+, sym(X+'A X+), the gradient), handed
  // to the next mi_stiefel_rq_model call if that call is for the same vector
  double *S_next = nullptr;
  mi_vec *Y_next = nullptr, *grad_next = nullptr, *Hh = nullptr;
  // The cache key is the trial vector's handle AND its contents' identity (mi_vec::serial / gen): handles and pooled
  // device pointers are both recycled, and an in-place write must invalidate the speculation too.
  const mi_vec *trial_X = nullptr;
  const double *trial_d = nullptr;
  uint64_t trial_serial = 0, trial_gen = 0;
  bool warned_unsymmetric = false;
  void remember_trial(const mi_vec *Xt) {
    trial_X = Xt;
    trial_d = Xt->d;
    trial_serial = Xt->serial;
    trial_gen = gen_of(Xt);
  }
  bool is_trial(const mi_vec *X) const {
    return trial_X == X && trial_d == X->d && trial_serial == X->serial && trial_gen == gen_of(X);
  }
};

namespace {

int rq_apply_common(mi_op *self, const mi_vec *in, mi_vec *out, bool dots, int *nparts) {
  mi_stiefel_rq *q = (mi_stiefel_rq *)self->impl;
  mi_ctx *ctx = q->ctx;
  int count = 0;
  MI_TRY(launch_spmm_gram(ctx, q->A, q->p, ctx->cg_live, in->d, q->X->d, q->S_dev, q->Z->d, &count));
  return launch_finish(ctx, q->n, q->p, ctx->cg_live, q->X->d, q->Z->d, in->d, count, nullptr, out->d, dots,
                       nparts);
}
int rq_apply(mi_op *self, const mi_vec *in, mi_vec *out) {
  return rq_apply_common(self, in, out, false, nullptr);
}
int rq_apply_dots(mi_op *self, const mi_vec *in, mi_vec *out, int *nparts) {
  return rq_apply_common(self, in, out, true, nparts);
}

// one-pass Hessian for STPCG: the direction kernel left `gram_count` partial rows of sym(Y'in - (X'in) S)

// resident workgroups per CU of the window instantiation that will run (registers + LDS), asked of the runtime once
// per instantiation: the launch plan must fit one round
int window_occupancy(int p, bool halo, int hw, bool fard) {
  static int cache[4][2][2][2] = {};
  int &slot = cache[p][halo ? 1 : 0][hw == 7 ? 0 : 1][fard ? 1 : 0];
  if (slot == 0) {
    int nb = 0;
    hipError_t e = hipErrorUnknown;
#define OCC(PV, HL, HWV, FV)                                                                                   \
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_st_hess_fused<PV, false, HL, true, true, HWV, FV>,  \
                                                   kWinBlock, 0)
#define OCC_F(PV, HL, HWV) \
  if (fard) OCC(PV, HL, HWV, 1); else OCC(PV, HL, HWV, 0)
#define OCC_P(PV)                                                \
  if (halo) { if (hw == 7) { OCC_F(PV, true, 7); } else { OCC_F(PV, true, 8); } } \
  else { if (hw == 7) { OCC_F(PV, false, 7); } else { OCC_F(PV, false, 8); } }
    if (p == 1) { OCC_P(1) } else if (p == 2) { OCC_P(2) } else { OCC_P(3) }
#undef OCC_P
#undef OCC_F
#undef OCC
    slot = (e == hipSuccess && nb > 0) ? nb : 1;
    (void)hipGetLastError();
  }
  return slot;
}

// r06: the window form where the matrix has one (packed, window of <= 2 chunks, no halo, one context) -- p = 4 ... 7 by
// default (at p = 8 the ring's rows are 64 bytes apart: 16-way LDS bank conflicts, and the quad layout is there);
// MI355OPT_WIDE_WINDOW = 0 / 1 forces it off / on for every width
static bool wide_window_form(const mi_ctx *ctx, const mi_csr *A, int p) {
  return (ctx->cfg.wide_window < 0 ? p <= 7 && !(ctx->cfg.wide_quad > 0) : ctx->cfg.wide_window != 0) && A->pk && A->wk &&
         A->win_chunks > 0 && A->win_chunks <= 2 && !A->halo && !ctx->uniform_grid && !ctx->cfg.no_window &&
         A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0;
}

// p = 5 ... 8 (and p = 4 in window form): the recurrence form through k_st_hess_wide / k_st_hess_widewin
int rq_apply_dir_wide(mi_stiefel_rq *q, const mi_vec *in, mi_vec *out, int gram_count, int *nparts) {
  mi_ctx *ctx = q->ctx;
  const mi_csr *A = q->A;
  const int p = q->p;
  MI_REQUIRE(gram_count < 0, "internal: rows wider than 4 doubles take the one-pass Hessian in its recurrence form only");
  // 256-thread workgroups, as many per CU as the kernel is held to (WideWaves<P>: one wave per SIMD each): one resident
  // round, at most kMaxRows partial rows; several ranks in rows mode need exactly kMaxGrid of them
  const size_t wgs = (A->nslices + kWideWaves - 1) / kWideWaves;
  // the quad layout where it measured faster (St(1e6,p), profiles/r05_wide_quad_ab.txt: p = 8: 82 vs 98 us; p = 5, 6, 7:
  // 80 / 73 / 93 vs 59 / 69 / 83 us for one lane per row); MI355OPT_WIDE_QUAD=0 / 1 forces either form
  const bool quad = ctx->cfg.wide_quad < 0 ? p == 8 : ctx->cfg.wide_quad != 0;
  const int resident = ctx->num_cu * (quad ? MI_WIDE_QUAD_WAVES : p <= 7 ? MI_WIDE_WAVES : 2);
  int grid = ctx->uniform_grid ? kMaxGrid : (int)std::max<size_t>(1, std::min<size_t>(wgs, std::min(resident, kMaxRows)));
  if (!ctx->uniform_grid && ctx->max_grid < kMaxGrid) grid = std::min(grid, ctx->max_grid);
  const int *bounds = nullptr;
  if (!ctx->uniform_grid && A->win_far_stride && !ctx->cfg.no_win_bounds && kWideWaves == kWinWaves)
    MI_TRY(window_bounds(ctx, A, grid, (int)wgs, &grid, &bounds));  // (runs cut to the far stride; tiles = 4 slices)
  const bool winform = wide_window_form(ctx, A, p);
  MI_REQUIRE(p > 4 || winform, "internal: p = 4 takes the wide rows' kernel in its window form only");
  if (winform) {
    const int wc = A->win_chunks, nc = 2 * kWinWaves + 2 * wc;
    const bool fard = A->win_far_pure > 0 && A->win_far_pure < ((size_t)1 << 31) && !ctx->cfg.no_far_computed;
    const size_t lds_bytes = (size_t)(nc * 64 + 1 + kWinFarRows) * (size_t)(p == 8 ? 9 : p) * sizeof(double);
    const void *fn = nullptr;
#define WW(PV, HWV) fn = fard ? (const void *)k_st_hess_widewin<PV, HWV, true> : (const void *)k_st_hess_widewin<PV, HWV, false>
#define WW_P(PV) if (A->win_head <= 7) { WW(PV, 7); } else { WW(PV, 8); }
    switch (p) {
      case 4: WW_P(4); break;
      case 5: WW_P(5); break;
      case 6: WW_P(6); break;
      case 7: WW_P(7); break;
      default: WW_P(8); break;
    }
#undef WW_P
#undef WW
    static bool attr_set[5][2][2] = {};
    bool &done = attr_set[p - 4][A->win_head <= 7 ? 0 : 1][fard ? 1 : 0];
    if (!done) {  // (more than 64 KB of dynamic LDS needs the attribute)
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 8 * 1024));
      (void)hipGetLastError();
      done = true;
    }
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kWinBlock, lds_bytes) != hipSuccess || occ < 1) occ = 1;
    (void)hipGetLastError();
    const int ntiles = (int)((A->nslices + kWinWaves - 1) / kWinWaves);
    int wgs = std::min(std::min(occ, p <= MI_WIDEWIN_3WAVES_UPTO ? 3 : 2) * ctx->num_cu, kMaxRows);
    if (ctx->max_grid < kMaxGrid) wgs = std::min(wgs, ctx->max_grid);
    int wgrid = 0;
    const int *wbounds = nullptr;
    MI_TRY(window_bounds(ctx, A, wgs, ntiles, &wgrid, &wbounds));
    WinView wv{A->wk, A->wfar, wc, nc, A->win_zero, wbounds, fard ? (unsigned)A->win_far_pure : 0u, nullptr};
    SellView view = sell_view(A);
    const CgState *live = ctx->cg_live;
    const double *inp = in->d, *Xp = q->X->d, *Yp = q->Y->d, *Sp = q->S_dev, *gd = ctx->scalars + SLOT_GDIR;
    double *outp = out->d, *parts = ctx->partials;
    void *args[] = {&view, &wv, &live, &inp, &Xp, &Yp, &Sp, &gd, &outp, &parts};
    KScope ks(ctx, MI_K_STIEFEL_HESS_FUSED);
    MI_HIP(hipLaunchKernel(fn, dim3(wgrid), dim3(kWinBlock), args, lds_bytes, ctx->stream));
    *nparts = wgrid;
    return MI_OK;
  }
  HaloWaitArg<true> hw_halo;
  HaloWaitArg<false> hw_none;
  MI_TRY(comm_halo_exchange_or_wait(ctx, A, p, in->d, &hw_halo.w));
  hw_halo.halo_lo = (unsigned)A->halo_lo;
  hw_halo.halo_hi = (unsigned)A->halo_hi;
  SellView view = sell_view(A);  // after the exchange: it selects the halo buffer the rows landed in
  KScope ks(ctx, MI_K_STIEFEL_HESS_FUSED);
#define HWIDE_ARGS(HWARG)                                                                                               \
  dim3(grid), dim3(kWideBlock), 0, ctx->stream, view, (const CgState *)ctx->cg_live, (const double *)in->d,             \
      (const double *)q->X->d, (const double *)q->Y->d, (const double *)q->S_dev,                                       \
      (const double *)(ctx->scalars + SLOT_GDIR), out->d, ctx->partials, HWARG, bounds
#define HWIDE(PV, HL, PKV, HWARG)                                                                   \
  if (quad) hipLaunchKernelGGL((k_st_hess_wideq<PV, HL, PKV>), HWIDE_ARGS(HWARG));                     \
  else hipLaunchKernelGGL((k_st_hess_wide<PV, HL, PKV>), HWIDE_ARGS(HWARG))
#define HWIDE_P(PV)                                                                                  \
  if (A->halo) { if (A->pk) { HWIDE(PV, true, true, hw_halo); } else { HWIDE(PV, true, false, hw_halo); } } \
  else { if (A->pk) { HWIDE(PV, false, true, hw_none); } else { HWIDE(PV, false, false, hw_none); } }
  switch (p) {
    case 5: HWIDE_P(5); break;
    case 6: HWIDE_P(6); break;
    case 7: HWIDE_P(7); break;
    default: HWIDE_P(8); break;
  }
#undef HWIDE_P
#undef HWIDE
#undef HWIDE_ARGS
  *nparts = grid;
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int rq_apply_dir(mi_op *self, const mi_vec *in, mi_vec *out, int gram_count, int *nparts) {
  mi_stiefel_rq *q = (mi_stiefel_rq *)self->impl;
  if (q->p > 4) return rq_apply_dir_wide(q, in, out, gram_count, nparts);
  mi_ctx *ctx = q->ctx;
  const mi_csr *A = q->A;
  const int p = q->p;
  // p = 4 has no window form of its own (ring + far slots of 4 x 4 workgroups: see WIN below); in the recurrence form it
  // takes the wide rows' (k_st_hess_widewin<4>, dynamic LDS sized for the matrix's own window) where that one applies
  if (p == 4 && gram_count == -1 && wide_window_form(ctx, A, p)) return rq_apply_dir_wide(q, in, out, gram_count, nparts);
  // one workgroup per CU and one round (the kernel needs > 64 VGPRs: a second round would only repeat the
  // prologue); the rows mode of several ranks needs the uniform 512-row partial layout instead
  constexpr int cap = 256;
  int grid = uniform_grid(ctx, sell_groups(A));
  if (!ctx->uniform_grid && grid > cap) grid = cap;
  double *slots = ctx->scalars + SLOT_GRAM;
  const bool recur = gram_count < 0;
  const bool sharded = slot_mode(ctx) && !recur;
  const bool halo = A->halo != nullptr;
  // the window form when the matrix qualifies (decided at creation, sparse.hip build_window); p = 4 does not fit
  const bool no_win = ctx->cfg.no_window;
  const int wc = (no_win || p > 3 || !A->wk) ? 0 : A->win_chunks;
  // (computed far columns: matrices whose far entries are all at row +- D, not sharded, D in 32 bits)
  const bool fard = A->win_far_pure > 0 && A->win_far_pure < ((size_t)1 << 31) && !ctx->cfg.no_far_computed;
  // 16-bit words: opt-in (MI355OPT_WORDS16=1).  On cfg2 they shorten the pass by ~1 us (25.9 -> 24.9 us) by halving
  // the matrix stream (28 -> 16 MB of 132 MB); the pass then runs at the same ~5.5 TB/s of a smaller total.
  const bool w16 = fard && !A->halo && A->wk16 && ctx->cfg.words16;
  WinView wv{A->wk, A->wfar, wc, 2 * kWinWaves + 2 * wc, A->win_zero, nullptr, fard ? (unsigned)A->win_far_pure : 0u,
             A->wk16};
  const bool win = recur && wc > 0;
  if (win && !ctx->uniform_grid) {  // planned runs of whole tiles (sparse.hip window_bounds)
    const int ntiles = (int)((A->nslices + kWinWaves - 1) / kWinWaves);
    // the workgroup budget: what is resident at once (one round), at most kMaxRows partial rows
    const int occ = std::min(window_occupancy(p, halo, A->win_head <= 7 ? 7 : 8, fard), kWaves / kWinWaves);
    int wgs = std::min(cap * occ, kMaxRows);
    // (one-GPU rehearsals of several ranks: the pass waits for its neighbours in its prologue, so -- like the CG
    // kernels -- it must leave room for their kernels while it does; mi_internal.h mi_ctx::max_grid)
    if (ctx->max_grid < kMaxGrid) wgs = std::min(wgs, ctx->max_grid);  // (lowered explicitly: MI355OPT_MAX_GRID)
    const int *bounds = nullptr;
    MI_TRY(window_bounds(ctx, A, wgs, ntiles, &grid, &bounds));
    wv.bounds = bounds;
  }
  const int block = win ? kWinBlock : kBlock;
  HaloWaitArg<true> hw_halo;   // a push folded into the kernel that wrote `in`: the pass waits in its prologue
  HaloWaitArg<false> hw_none;
  MI_TRY(comm_halo_exchange_or_wait(ctx, A, p, in->d, &hw_halo.w));
  hw_halo.halo_lo = (unsigned)A->halo_lo;  // (computed far columns of a row shard: spmm_core.h load_far)
  hw_halo.halo_hi = (unsigned)A->halo_hi;
  SellView view = sell_view(A);  // after the exchange: it selects the halo buffer the rows landed in
  if (!recur) {
    if (rows_mode(ctx)) MI_TRY(comm_allreduce_rows(ctx, ctx->partials2, nsym(p)));
    if (sharded) MI_TRY(sharded_reduce(ctx, gram_count, nsym(p), slots));
  }
  KScope ks(ctx, MI_K_STIEFEL_HESS_FUSED);
#define HF3(F, HL, RC, PKV, HWV)                                                                              \
  DISPATCH_P4(p, hipLaunchKernelGGL((k_st_hess_fused<P, F, HL, RC, PKV, HWV>), dim3(grid), dim3(block), 0,     \
                                   ctx->stream, view, wv, (const CgState *)ctx->cg_live,                     \
                                   (const double *)in->d, (const double *)q->X->d, (const double *)q->Y->d,  \
                                   (const double *)q->S_dev, (const double *)ctx->partials2, gram_count,     \
                                   (const double *)slots, (const double *)(ctx->scalars + SLOT_GDIR),        \
                                   out->d, ctx->partials, HW_ARG_##HL))
#define HW_ARG_true hw_halo
#define HW_ARG_false hw_none
#define HF(F, HL, RC)                                                          \
  if (A->pk) { HF3(F, HL, RC, true, 0); }                                      \
  else { HF3(F, HL, RC, false, 0); }
#define HF3D(HWV, FARV)                                                                                       \
  DISPATCH_P4(p, hipLaunchKernelGGL((k_st_hess_fused<P, false, false, true, true, HWV, FARV>), dim3(grid),      \
                                   dim3(block), 0, ctx->stream, view, wv, (const CgState *)ctx->cg_live,      \
                                   (const double *)in->d, (const double *)q->X->d, (const double *)q->Y->d,   \
                                   (const double *)q->S_dev, (const double *)ctx->partials2, gram_count,      \
                                   (const double *)slots, (const double *)(ctx->scalars + SLOT_GDIR),         \
                                   out->d, ctx->partials, hw_none))
#define HF3DH(HWV)                                                                                            \
  DISPATCH_P4(p, hipLaunchKernelGGL((k_st_hess_fused<P, false, true, true, true, HWV, 1>), dim3(grid),         \
                                   dim3(block), 0, ctx->stream, view, wv, (const CgState *)ctx->cg_live,      \
                                   (const double *)in->d, (const double *)q->X->d, (const double *)q->Y->d,   \
                                   (const double *)q->S_dev, (const double *)ctx->partials2, gram_count,      \
                                   (const double *)slots, (const double *)(ctx->scalars + SLOT_GDIR),         \
                                   out->d, ctx->partials, hw_halo))
  if (gram_count == -2) {  // the two-kernel step (opt-in experiment): stpcg.hip asked for it only where twok says it exists
    MI_REQUIRE(win && fard && !halo && !w16 && p == 3 && ctx->twok_r, "internal: two-kernel step without its Hessian form");
#define HF2K(HWV)                                                                                                       \
  hipLaunchKernelGGL((k_st_hess_fused<3, false, false, true, true, HWV, 1, true>), dim3(grid), dim3(block), 0,          \
                     ctx->stream, view, wv, (const CgState *)ctx->cg_live, (const double *)in->d,                       \
                     (const double *)q->X->d, (const double *)q->Y->d, (const double *)q->S_dev,                        \
                     (const double *)ctx->partials2, gram_count, (const double *)slots,                                 \
                     (const double *)(ctx->scalars + SLOT_GDIR), out->d, ctx->partials, hw_none, ctx->twok_r)
    if (A->win_head <= 7) { HF2K(7); } else { HF2K(8); }
#undef HF2K
  } else if (win && w16) {  // the window form with computed far columns and 16-bit words
    if (A->win_head <= 7) { HF3D(7, 2); } else { HF3D(8, 2); }
  } else if (win && fard && halo) {  // ... with computed far columns, some of them halo columns
    if (A->win_head <= 7) { HF3DH(7); } else { HF3DH(8); }
  } else if (win && fard) {  // ... with computed far columns
    if (A->win_head <= 7) { HF3D(7, 1); } else { HF3D(8, 1); }
  } else if (win) {  // the window form (recurrence form only: the unpreconditioned solve)
    if (A->win_head <= 7) {
      if (halo) { HF3(false, true, true, true, 7); } else { HF3(false, false, true, true, 7); }
    } else {
      if (halo) { HF3(false, true, true, true, 8); } else { HF3(false, false, true, true, 8); }
    }
  } else if (recur) {
    if (halo) { HF(false, true, true); } else { HF(false, false, true); }
  } else if (halo) {
    if (sharded) { HF(true, true, false); } else { HF(false, true, false); }
  } else {
    if (sharded) { HF(true, false, false); } else { HF(false, false, false); }
  }
#undef HF3DH
#undef HF3D
#undef HF3
#undef HF
#undef HW_ARG_true
#undef HW_ARG_false
  *nparts = grid;
  MI_HIP(hipGetLastError());
  return MI_OK;
}

struct RqPreconImpl {
  mi_stiefel_rq *q;
  const mi_vec *X;
  const mi_vec *dinv;
};
int rq_precon_apply(mi_precon *self, const mi_vec *r, mi_vec *v) {
  RqPreconImpl *im = (RqPreconImpl *)self->impl;
  mi_stiefel_rq *q = im->q;
  mi_ctx *ctx = q->ctx;
  const int p = q->p;
  const int grid = row_grid(q->ctx, q->n, p);
  DISPATCH_P(p, hipLaunchKernelGGL((k_st_gram<P, 2, true>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, q->n,
                                   (const double *)im->X->d, (const double *)r->d,
                                   (const double *)im->dinv->d, q->Z->d, ctx->partials2));
  return launch_finish(ctx, q->n, q->p, nullptr, im->X->d, q->Z->d, nullptr, grid, nullptr, v->d, false,
                       nullptr);
}
void rq_precon_destroy(mi_precon *self) { delete (RqPreconImpl *)self->impl; }

}  // namespace

extern "C" {

#ifdef MI_WIN_STAMPS
// experiment builds: (de)register the device buffer the window kernel dumps its stamps into
__attribute__((visibility("default"))) int mi_debug_stamp_buffer(void *dev_ptr) {
  MI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(mi::g_stamp_buf), &dev_ptr, sizeof(dev_ptr)));
  return MI_OK;
}
#endif

int mi_stiefel_gram(mi_ctx *ctx, size_t n, int p, const mi_vec *X, const mi_vec *Z, double *G_host) {
  MI_TRY(check_np(ctx, n, p, X, Z, nullptr));
  MI_REQUIRE(X && Z && G_host, "null argument");
  const int grid = row_grid(ctx, n, p);
  DISPATCH_P(p, hipLaunchKernelGGL((k_st_gram<P, 0, false>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, n,
                                   (const double *)X->d, (const double *)Z->d, (const double *)nullptr,
                                   (double *)nullptr, ctx->partials2));
  double *slots = ctx->scalars + SLOT_RAW;
  MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, grid, p * p, slots));
  return read_slots_sync(ctx, SLOT_RAW, p * p, G_host);
}

int mi_stiefel_project(mi_ctx *ctx, size_t n, int p, const mi_vec *X, const mi_vec *Z, mi_vec *out) {
  MI_TRY(check_np(ctx, n, p, X, Z, out));
  MI_REQUIRE(X && Z && out, "null argument");
  touch(out);
  const int grid = row_grid(ctx, n, p);
  DISPATCH_P(p, hipLaunchKernelGGL((k_st_gram<P, 0, true>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, n,
                                   (const double *)X->d, (const double *)Z->d, (const double *)nullptr,
                                   (double *)nullptr, ctx->partials2));
  return launch_finish(ctx, n, p, nullptr, X->d, Z->d, nullptr, grid, nullptr, out->d, false, nullptr);
}

int mi_stiefel_retract(mi_ctx *ctx, size_t n, int p, const mi_vec *X, const mi_vec *V, mi_vec *Y) {
  MI_TRY(check_np(ctx, n, p, X, V, Y));
  MI_REQUIRE(X && V && Y, "null argument");
  touch(Y);
  const int grid = row_grid(ctx, n, p);
  KScope ks(ctx, MI_K_STIEFEL_RETRACT);
  DISPATCH_P(p, hipLaunchKernelGGL((k_st_gram<P, 1, true>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, n,
                                   (const double *)X->d, (const double *)V->d, (const double *)nullptr,
                                   Y->d, ctx->partials2));
  double *slots = ctx->scalars + SLOT_GRAM;
  if (ctx->comm != nullptr || ctx->force_slot_path) {
    MI_TRY(sharded_reduce(ctx, grid, nsym(p), slots));
    DISPATCH_P(p, hipLaunchKernelGGL((k_st_polar<P, true>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, n,
                                     Y->d, (const double *)ctx->partials2, grid, (const double *)slots));
  } else {
    DISPATCH_P(p, hipLaunchKernelGGL((k_st_polar<P, false>), dim3(grid), dim3(StBlk<P>::threads), 0, ctx->stream, n,
                                     Y->d, (const double *)ctx->partials2, grid, (const double *)slots));
  }
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_stiefel_rq_create(mi_ctx *ctx, const mi_csr *A, size_t n, int p, mi_stiefel_rq **out) {
  MI_REQUIRE(ctx && A && out, "null argument");
  MI_REQUIRE(p >= 1 && p <= kMaxP, "p must be in [1,%d], got %d", kMaxP, p);
  MI_REQUIRE(A->n == n && A->ctx == ctx, "matrix has %zu rows, expected %zu", A->n, n);
  mi_stiefel_rq *q = new mi_stiefel_rq();
  q->ctx = ctx;
  q->A = A;
  q->n = n;
  q->p = p;
  q->X = nullptr;
  MI_HIP(hipMalloc((void **)&q->S_dev, kMaxP * kMaxP * sizeof(double)));
  MI_HIP(hipMemsetAsync(q->S_dev, 0, kMaxP * kMaxP * sizeof(double), ctx->stream));
  MI_TRY(mi_vec_create(ctx, n * (size_t)p, &q->Z));
  MI_TRY(mi_vec_create(ctx, n * (size_t)p, &q->Y));
  q->hess.ctx = ctx;
  q->hess.n = n * (size_t)p;
  q->hess.apply = rq_apply;
  q->hess.apply_dots = rq_apply_dots;
  q->hess.apply_dir = rq_apply_dir;
  q->hess.impl = q;
  q->hess.borrowed = true;
  *out = q;
  return MI_OK;
}

int mi_stiefel_rq_destroy(mi_stiefel_rq *q) {
  if (!q) return MI_OK;
  (void)hipStreamSynchronize(q->ctx->stream);
  (void)hipFree(q->S_dev);
  (void)hipFree(q->S_next);
  mi_vec_destroy(q->Z);
  mi_vec_destroy(q->Y);
  mi_vec_destroy(q->Y_next);
  mi_vec_destroy(q->grad_next);
  mi_vec_destroy(q->Hh);
  delete q;
  return MI_OK;
}

int mi_stiefel_rq_objective(mi_stiefel_rq *q, const mi_vec *X, double *f) {
  MI_REQUIRE(q && X && f, "null argument");
  MI_TRY(check_np(q->ctx, q->n, q->p, X, nullptr, nullptr));
  mi_ctx *ctx = q->ctx;
  q->trial_X = nullptr;  // (Z is scratch of both)
  int count = 0;
  MI_TRY(launch_spmm_gram(ctx, q->A, q->p, nullptr, X->d, X->d, nullptr, q->Z->d, &count));
  double *slots = ctx->scalars + SLOT_GRAM;
  const int ns = nsym(q->p);
  MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, count, ns, slots));
  double G[kMaxP * (kMaxP + 1) / 2];
  MI_TRY(read_slots_sync(ctx, SLOT_GRAM, ns, G));
  double tr = 0;
  for (int a = 0, idx = 0; a < q->p; ++a) {  // diagonal entries of the packed symmetric Gram
    tr += G[idx];
    idx += q->p - a;
  }
  *f = .5 * tr;
  return MI_OK;
}

int mi_stiefel_rq_model(mi_stiefel_rq *q, const mi_vec *X, mi_vec *grad, mi_op **hess) {
  MI_REQUIRE(q && X && grad, "null argument");
  MI_TRY(check_np(q->ctx, q->n, q->p, X, grad, nullptr));
  int count = 0;
  touch(grad);
  if (q->is_trial(X)) {
    // X is the point mi_stiefel_rq_trial just evaluated: A X, S and the gradient exist already
    std::swap(q->Y, q->Y_next);
    std::swap(q->S_dev, q->S_next);
    MI_TRY(mi_vec_copy(grad, q->grad_next));
  } else {
    // Y = A X (kept: the direction-Gram identity needs it) ; S = sym(X'AX) (kept on the device for the
    // Hessian) ; grad = Y - X S
    MI_TRY(launch_spmm_gram(q->ctx, q->A, q->p, nullptr, X->d, X->d, nullptr, q->Y->d, &count));
    MI_TRY(launch_finish(q->ctx, q->n, q->p, nullptr, X->d, q->Y->d, nullptr, count, q->S_dev, grad->d, false,
                         nullptr));
  }
  q->trial_X = nullptr;
  q->X = X;
  q->X_serial = X->serial;
  q->dg.p = q->p;
  q->dg.n = q->n;
  q->dg.X = X->d;
  q->dg.Y = q->Y->d;
  q->dg.S = q->S_dev;
  q->dg.halo_A = q->A->halo ? q->A : nullptr;
  // (the opt-in two-kernel step: p = 3, window form with computed far columns, one rank)
  q->dg.twok = q->p == 3 && !q->A->halo && q->A->wk && q->A->win_chunks > 0 && q->A->win_far_pure > 0 &&
               q->A->win_far_pure < ((size_t)1 << 31) && !q->ctx->cfg.no_window && !q->ctx->cfg.no_far_computed &&
               !q->ctx->uniform_grid && !(q->A->wk16 && q->ctx->cfg.words16);
  // the one-pass kernel uses 32-bit byte offsets: fields of 4 GiB or more keep the two-pass operator -- and so does
  // a matrix that is not symmetric (checked at creation): the one-pass form replaces X'(A p) by (A X)'p
  q->hess.dirgram = (sell_stream_ok(q->A, q->p) && q->A->symmetric) ? &q->dg : nullptr;
  if (!q->A->symmetric && !q->warned_unsymmetric) {
    q->warned_unsymmetric = true;
    fprintf(stderr, "mi355opt: the matrix of this Stiefel Rayleigh-quotient problem is not symmetric: the Hessian "
                    "operator keeps its two-pass form (P_X(A V - V S) with A as given)\n");
  }
  if (hess) *hess = &q->hess;
  return MI_OK;
}

// One trial step of a trust-region / line-search method at the point X the model is bound to (reference
// Riemannian/TNT.h:493-512,573-585): step norm, retraction, objective at the trial point, predicted-decrease terms and
// -- speculatively -- the model at the trial point, as ONE launch chain and ONE read-back.  Every number is produced
// by the same kernels, in the same order of summation, as the separate calls (mi_vec_dot_batch, mi_stiefel_retract,
// mi_stiefel_rq_objective, mi_stiefel_rq_model) would: only one sparse product (A X+ serves the objective AND the
// next gradient) and the intermediate synchronisations disappear.
//   out[0] = f(X+), out[1] = <h,h>, out[2] = <g,h>, out[3] = <h, Hess h>, out[4] = |grad f(X+)|^2
int mi_stiefel_rq_trial(mi_stiefel_rq *q, const mi_vec *X, const mi_vec *h, const mi_vec *g, mi_vec *X_trial,
                        double out[5]) {
  MI_REQUIRE(q && X && h && g && X_trial && out, "null argument");
  MI_REQUIRE(q->bound_to(X), "mi_stiefel_rq_trial: the model is not bound to this X (call mi_stiefel_rq_model first)");
  MI_TRY(check_np(q->ctx, q->n, q->p, X, h, g));
  MI_TRY(check_np(q->ctx, q->n, q->p, X_trial, nullptr, nullptr));
  mi_ctx *ctx = q->ctx;
  ctx->fusion.fused_trial_steps++;
  const size_t N = q->n * (size_t)q->p;
  if (!q->Y_next) {
    MI_TRY(mi_vec_create(ctx, N, &q->Y_next));
    MI_TRY(mi_vec_create(ctx, N, &q->grad_next));
    MI_TRY(mi_vec_create(ctx, N, &q->Hh));
    MI_HIP(hipMalloc((void **)&q->S_next, kMaxP * kMaxP * sizeof(double)));
  }
  q->trial_X = nullptr;
  // (a) Hess h, then |h|^2, <g,h>, <h, Hess h> in one pass (as MI355::dot_batch does)
  MI_TRY(rq_apply(&q->hess, h, q->Hh));
  {
    const double *xs[3] = {h->d, g->d, h->d}, *ys[3] = {h->d, h->d, q->Hh->d};
    MI_TRY(dot_batch_to_slots(ctx, 3, xs, ys, N, SLOT_MISC));
  }
  // (b) X+ = polar(X + h)
  MI_TRY(mi_stiefel_retract(ctx, q->n, q->p, X, h, X_trial));
  // (c) A X+ with the Gram rows of sym(X+' A X+): the objective (reduced exactly as mi_stiefel_rq_objective does)
  //     and, through the same rows, S+ and the gradient at X+ (exactly as mi_stiefel_rq_model does)
  int count = 0;
  MI_TRY(launch_spmm_gram(ctx, q->A, q->p, nullptr, X_trial->d, X_trial->d, nullptr, q->Y_next->d, &count));
  const int ns = nsym(q->p);
  MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, count, ns, ctx->scalars + SLOT_GRAM));
  MI_TRY(launch_finish(ctx, q->n, q->p, nullptr, X_trial->d, q->Y_next->d, nullptr, count, q->S_next,
                       q->grad_next->d, false, nullptr));
  {
    const double *xs[1] = {q->grad_next->d}, *ys[1] = {q->grad_next->d};
    MI_TRY(dot_batch_to_slots(ctx, 1, xs, ys, N, SLOT_MISC + 3));
  }
  // (d) one read-back: slots [SLOT_GRAM, SLOT_MISC + 4)
  static_assert(SLOT_GRAM < SLOT_MISC && SLOT_MISC + 4 <= kScalarSlots, "slot map");
  double buf[SLOT_MISC + 4 - SLOT_GRAM];
  MI_TRY(read_slots_sync(ctx, SLOT_GRAM, SLOT_MISC + 4 - SLOT_GRAM, buf));
  double tr = 0;
  for (int a = 0, idx = 0; a < q->p; ++a) {  // diagonal entries of the packed symmetric Gram
    tr += buf[idx];
    idx += q->p - a;
  }
  out[0] = .5 * tr;
  out[1] = buf[SLOT_MISC - SLOT_GRAM + 0];
  out[2] = buf[SLOT_MISC - SLOT_GRAM + 1];
  out[3] = buf[SLOT_MISC - SLOT_GRAM + 2];
  out[4] = buf[SLOT_MISC - SLOT_GRAM + 3];
  q->remember_trial(X_trial);
  return MI_OK;
}

// One Armijo trial of a backtracking line search along -g (reference Riemannian/GradientDescent.h:266-286:
// h = -t g; x_trial = retract(x, h); f(x_trial)) plus, speculatively, the gradient at the trial point and its squared
// norm (:325-327, used if the trial is accepted): one launch chain, one read-back.  Same kernels and summation
// orders as the separate calls.  out[0] = f(X+), out[1] = |grad f(X+)|^2.
int mi_stiefel_rq_armijo_trial(mi_stiefel_rq *q, const mi_vec *X, const mi_vec *g, double t, mi_vec *h_out,
                               mi_vec *X_trial, double out[2]) {
  MI_REQUIRE(q && X && g && h_out && X_trial && out, "null argument");
  MI_REQUIRE(q->bound_to(X),
             "mi_stiefel_rq_armijo_trial: the model is not bound to this X (call mi_stiefel_rq_model first)");
  MI_TRY(check_np(q->ctx, q->n, q->p, X, g, h_out));
  MI_TRY(check_np(q->ctx, q->n, q->p, X_trial, nullptr, nullptr));
  mi_ctx *ctx = q->ctx;
  ctx->fusion.fused_trial_steps++;
  const size_t N = q->n * (size_t)q->p;
  if (!q->Y_next) {
    MI_TRY(mi_vec_create(ctx, N, &q->Y_next));
    MI_TRY(mi_vec_create(ctx, N, &q->grad_next));
    MI_TRY(mi_vec_create(ctx, N, &q->Hh));
    MI_HIP(hipMalloc((void **)&q->S_next, kMaxP * kMaxP * sizeof(double)));
  }
  q->trial_X = nullptr;
  touch(X_trial);
  MI_TRY(mi_vec_scale_to(h_out, -t, g));  // h = -t * g (:276)
  MI_TRY(mi_stiefel_retract(ctx, q->n, q->p, X, h_out, X_trial));
  int count = 0;
  MI_TRY(launch_spmm_gram(ctx, q->A, q->p, nullptr, X_trial->d, X_trial->d, nullptr, q->Y_next->d, &count));
  const int ns = nsym(q->p);
  MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, count, ns, ctx->scalars + SLOT_GRAM));
  MI_TRY(launch_finish(ctx, q->n, q->p, nullptr, X_trial->d, q->Y_next->d, nullptr, count, q->S_next,
                       q->grad_next->d, false, nullptr));
  {
    const double *xs[1] = {q->grad_next->d}, *ys[1] = {q->grad_next->d};
    MI_TRY(dot_batch_to_slots(ctx, 1, xs, ys, N, SLOT_MISC));
  }
  double buf[SLOT_MISC + 1 - SLOT_GRAM];
  MI_TRY(read_slots_sync(ctx, SLOT_GRAM, SLOT_MISC + 1 - SLOT_GRAM, buf));
  double tr = 0;
  for (int a = 0, idx = 0; a < q->p; ++a) {
    tr += buf[idx];
    idx += q->p - a;
  }
  out[0] = .5 * tr;
  out[1] = buf[SLOT_MISC - SLOT_GRAM];
  q->remember_trial(X_trial);
  return MI_OK;
}

int mi_stiefel_rq_precon(mi_stiefel_rq *q, const mi_vec *X, const mi_vec *dinv_rows, mi_precon **out) {
  MI_REQUIRE(q && X && dinv_rows && out, "null argument");
  MI_TRY(check_np(q->ctx, q->n, q->p, X, nullptr, nullptr));
  MI_REQUIRE(dinv_rows->n == q->n, "dinv_rows must have one entry per row");
  mi_precon *P = new mi_precon();
  P->ctx = q->ctx;
  P->n = q->n * (size_t)q->p;
  P->kind = 0;
  P->apply = rq_precon_apply;
  P->destroy = rq_precon_destroy;
  P->impl = new RqPreconImpl{q, X, dinv_rows};
  *out = P;
  return MI_OK;
}

}  // extern "C"
