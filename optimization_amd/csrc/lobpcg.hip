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
// Gram kernel: a workgroup (4 waves) owns a contiguous range of rows; per 32-row tile both panels are
// staged in LDS column-major with leading dimension 34 (== 2 mod 32, so the 16 columns x 2 rows that a
// half-wave reads for one MFMA operand hit 32 distinct 8-byte bank pairs); each wave accumulates its
// share of the (ka/16) x (kb/16) output tiles in registers over the whole row range; per-workgroup
// partial Grams are then summed in fixed order by a second kernel (deterministic).
// Algorithmic bytes: 8 m (ka + kb)  (8 m ka when S == T); flops 2 m ka kb.
#include <algorithm>
#include <cmath>

#include "mi_internal.h"

using namespace mi;

namespace {

typedef double double4v __attribute__((ext_vector_type(4)));

constexpr int kGramRows = 32;       // rows per LDS tile
constexpr int kGramLd = 34;         // LDS leading dimension
constexpr int kGramMaxK = 96;       // max panel width
constexpr int kGramThreads = 256;
constexpr int kMaxTilesPerWave = 9;  // (96/16)^2 / 4

// stage a 32-row x kpad-column tile (rows r0.., zero-filled past m and past k) into LDS
__device__ __forceinline__ void stage_tile(const double *__restrict__ P, size_t m, int k, int kpad, size_t r0,
                                           double *lds) {
  // 16 threads (16 B each) cover one 32-row column segment
  const int seg = threadIdx.x >> 4, part = threadIdx.x & 15;
  for (int c = seg; c < kpad; c += kGramThreads / 16) {
    const size_t r = r0 + 2 * (size_t)part;
    double2 v = make_double2(0.0, 0.0);
    if (c < k) {
      const double *src = P + (size_t)c * m + r;
      if (r + 1 < m) {
        // column starts are only 8-byte aligned in general (m odd): two scalar loads
        v.x = src[0];
        v.y = src[1];
      } else if (r < m) {
        v.x = src[0];
      }
    }
    lds[c * kGramLd + 2 * part] = v.x;
    lds[c * kGramLd + 2 * part + 1] = v.y;
  }
}

template <bool SAME>
__global__ __launch_bounds__(kGramThreads) void k_gram(size_t m, int ka, int kb, const double *__restrict__ S,
                                                       const double *__restrict__ T, size_t rows_per_block,
                                                       double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int ta = (ka + 15) / 16, tb = (kb + 15) / 16;
  const int kapad = ta * 16, kbpad = tb * 16;
  double *ldsS = smem;
  double *ldsT = SAME ? smem : smem + (size_t)kapad * kGramLd;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ntiles = ta * tb;

  double4v acc[kMaxTilesPerWave];
#pragma unroll
  for (int t = 0; t < kMaxTilesPerWave; ++t) acc[t] = (double4v){0.0, 0.0, 0.0, 0.0};

  const size_t rb = (size_t)blockIdx.x * rows_per_block;
  const size_t re = std::min(m, rb + rows_per_block);
  for (size_t r0 = rb; r0 < re; r0 += kGramRows) {
    __syncthreads();
    stage_tile(S, m, ka, kapad, r0, ldsS);
    if (!SAME) stage_tile(T, m, kb, kbpad, r0, ldsT);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kMaxTilesPerWave; ++t) {
      const int tile = w + 4 * t;
      if (tile < ntiles) {
        const int ti = tile / tb, tj = tile % tb;
        const double *pa = ldsS + (ti * 16 + (lane & 15)) * kGramLd + (lane >> 4);
        const double *pb = ldsT + (tj * 16 + (lane & 15)) * kGramLd + (lane >> 4);
#pragma unroll
        for (int kk = 0; kk < kGramRows / 4; ++kk)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk], pb[4 * kk], acc[t], 0, 0, 0);
      }
    }
  }
  // partial Gram of this workgroup, column-major ka x kb; f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 j
  double *out = partial + (size_t)blockIdx.x * ka * kb;
#pragma unroll
  for (int t = 0; t < kMaxTilesPerWave; ++t) {
    const int tile = w + 4 * t;
    if (tile < ntiles) {
      const int ti = tile / tb, tj = tile % tb;
      const int col = tj * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = ti * 16 + (lane >> 4) + 4 * j;
        if (row < ka && col < kb) out[(size_t)col * ka + row] = acc[t][j];
      }
    }
  }
}

// G[e] = sum over workgroups of partial[b][e], fixed order
__global__ void k_gram_reduce(int nblocks, int nelem, const double *__restrict__ partial, double *__restrict__ G) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  double s = 0;
  for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * nelem + e];
  G[e] = s;
}

// Y[:, c0:c0+KC) = S (m x ks) C[:, c0:c0+KC); one thread per row, C chunk in LDS
template <int KC>
__global__ __launch_bounds__(256) void k_panel_update(size_t m, int ks, const double *__restrict__ S,
                                                      const double *__restrict__ Cdev, int ldc, int c0, int kc,
                                                      double *__restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) double smem[];  // ks x KC, row-major by s
  for (int i = threadIdx.x; i < ks * KC; i += blockDim.x) {
    const int s = i / KC, c = i % KC;
    smem[i] = (c0 + c < kc) ? Cdev[(size_t)(c0 + c) * ldc + s] : 0.0;
  }
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < m; r += stride) {
    double acc[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) acc[c] = 0;
    for (int s = 0; s < ks; ++s) {
      const double sv = S[(size_t)s * m + r];
      const double *cr = smem + s * KC;
#pragma unroll
      for (int c = 0; c < KC; ++c) acc[c] += sv * cr[c];
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
      if (c0 + c < kc) Y[(size_t)(c0 + c) * m + r] = acc[c];
  }
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
        const double res = AX[idx] - BX[idx] * th[c];
        const double xv = X[idx];
        R[idx] = res;
        a[c] += res * res;
        a[8 + c] += xv * xv;
      }
    }
  }
  block_partials_store<16>(a, lds, partials);
}

// Y = A X for a column-major panel: one thread per row, columns in chunks of 8
__global__ __launch_bounds__(256) void k_spmm_colmajor(size_t n, size_t nslices, const long long *__restrict__ sp,
                                                       const int *__restrict__ col, const double *__restrict__ val,
                                                       int k, int c0, const double *__restrict__ X,
                                                       double *__restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const size_t slice = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const size_t row = slice * 64 + lane;
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0;
  const long long b0 = sp[slice], b1 = sp[slice + 1];
  for (long long kk = b0; kk < b1; ++kk) {
    const size_t e = (size_t)kk * 64 + lane;
    const double a = val[e];
    const size_t j = (size_t)col[e];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c0 + c < k) acc[c] += a * X[(size_t)(c0 + c) * n + j];
  }
  if (row < n) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c0 + c < k) Y[(size_t)(c0 + c) * n + row] = acc[c];
  }
}

// Y[r, c] = d[r] X[r, c]  (diagonal operators of the reference's LOBPCG tests, tests/LOBPCG_unit_test.cpp:56-74)
__global__ __launch_bounds__(256) void k_rowscale(size_t m, size_t k, const double *__restrict__ d,
                                                  const double *__restrict__ X, double *__restrict__ Y) {
  const size_t total = m * k, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) Y[i] = d[i % m] * X[i];
}

// ---- host: generalized symmetric-definite eigenproblem (LOBPCG.h:53-62) ---------------------
int cholesky_lower(int n, std::vector<double> &A) {
  for (int j = 0; j < n; ++j) {
    double d = A[j + (size_t)j * n];
    for (int k = 0; k < j; ++k) d -= A[j + (size_t)k * n] * A[j + (size_t)k * n];
    if (!(d > 0)) return -1;
    d = std::sqrt(d);
    A[j + (size_t)j * n] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i + (size_t)j * n];
      for (int k = 0; k < j; ++k) s -= A[i + (size_t)k * n] * A[j + (size_t)k * n];
      A[i + (size_t)j * n] = s / d;
    }
    for (int i = 0; i < j; ++i) A[i + (size_t)j * n] = 0;
  }
  return 0;
}

void jacobi_eigh(int n, std::vector<double> &M, std::vector<double> &V, double *w) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[i + (size_t)i * n] = 1;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) (i != j ? off : diag) += M[i + (size_t)j * n] * M[i + (size_t)j * n];
    if (off <= 1e-32 * (diag + off) || off == 0) break;
    for (int p = 0; p + 1 < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = M[p + (size_t)q * n];
        if (apq == 0) continue;
        const double tau = (M[q + (size_t)q * n] - M[p + (size_t)p * n]) / (2 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
        const double c = 1 / std::sqrt(1 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double a = M[k + (size_t)p * n], b = M[k + (size_t)q * n];
          M[k + (size_t)p * n] = c * a - s * b;
          M[k + (size_t)q * n] = s * a + c * b;
        }
        for (int k = 0; k < n; ++k) {
          const double a = M[p + (size_t)k * n], b = M[q + (size_t)k * n];
          M[p + (size_t)k * n] = c * a - s * b;
          M[q + (size_t)k * n] = s * a + c * b;
        }
        for (int k = 0; k < n; ++k) {
          const double a = V[k + (size_t)p * n], b = V[k + (size_t)q * n];
          V[k + (size_t)p * n] = c * a - s * b;
          V[k + (size_t)q * n] = s * a + c * b;
        }
      }
  }
  for (int i = 0; i < n; ++i) w[i] = M[i + (size_t)i * n];
  for (int i = 0; i + 1 < n; ++i) {  // ascending
    int mn = i;
    for (int j = i + 1; j < n; ++j)
      if (w[j] < w[mn]) mn = j;
    if (mn != i) {
      std::swap(w[i], w[mn]);
      for (int k = 0; k < n; ++k) std::swap(V[k + (size_t)i * n], V[k + (size_t)mn * n]);
    }
  }
}

int check_panel(mi_ctx *ctx, size_t m, int k, const mi_vec *P, const char *what) {
  MI_REQUIRE(P, "%s is null", what);
  MI_REQUIRE(P->ctx == ctx, "%s belongs to another context", what);
  MI_REQUIRE(P->n >= m * (size_t)k, "%s holds %zu doubles, needs m*k = %zu*%d", what, P->n, m, k);
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_lobpcg_gram(mi_ctx *ctx, size_t m, int ka, int kb, const mi_vec *S, const mi_vec *T, double *G_host) {
  MI_REQUIRE(ctx && G_host, "null argument");
  MI_REQUIRE(ka >= 1 && ka <= kGramMaxK && kb >= 1 && kb <= kGramMaxK, "panel widths must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(ctx, m, ka, S, "S"));
  MI_TRY(check_panel(ctx, m, kb, T, "T"));
  const bool same = (S->d == T->d) && ka == kb;
  // rows per workgroup: a multiple of the 32-row tile, ~2 workgroups per CU
  size_t nb = std::min<size_t>(2 * (size_t)ctx->num_cu, (m + kGramRows - 1) / kGramRows);
  if (nb < 1) nb = 1;
  size_t rpb = ((m + nb - 1) / nb + kGramRows - 1) / kGramRows * kGramRows;
  nb = (m + rpb - 1) / rpb;
  const int nelem = ka * kb;
  void *partial = nullptr, *Gdev = nullptr;
  MI_TRY(pool_alloc(ctx, nb * (size_t)nelem * sizeof(double), &partial));
  MI_TRY(pool_alloc(ctx, (size_t)nelem * sizeof(double), &Gdev));
  const int kapad = (ka + 15) / 16 * 16, kbpad = (kb + 15) / 16 * 16;
  const size_t lds = (size_t)(same ? kapad : kapad + kbpad) * kGramLd * sizeof(double);
  {
    KScope ks(ctx, MI_K_LOBPCG_GRAM);
    if (same)
      hipLaunchKernelGGL(k_gram<true>, dim3((unsigned)nb), dim3(kGramThreads), lds, ctx->stream, m, ka, kb,
                         (const double *)S->d, (const double *)T->d, rpb, (double *)partial);
    else
      hipLaunchKernelGGL(k_gram<false>, dim3((unsigned)nb), dim3(kGramThreads), lds, ctx->stream, m, ka, kb,
                         (const double *)S->d, (const double *)T->d, rpb, (double *)partial);
  }
  hipLaunchKernelGGL(k_gram_reduce, dim3((nelem + 255) / 256), dim3(256), 0, ctx->stream, (int)nb, nelem,
                     (const double *)partial, (double *)Gdev);
  if (ctx->comm) {
    for (int off = 0; off < nelem; off += 4096)  // all-reduce in chunks the comm layer accepts
      MI_TRY(comm_allreduce(ctx, (double *)Gdev + off, std::min(4096, nelem - off)));
  }
  hipError_t e = hipMemcpyAsync(G_host, Gdev, (size_t)nelem * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  pool_free(ctx, partial);
  pool_free(ctx, Gdev);
  if (e != hipSuccess) return hip_fail(e, "gram read-back", __FILE__, __LINE__);
  return MI_OK;
}

int mi_lobpcg_update(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S, const double *C_host, int ldc,
                     mi_vec *Y) {
  MI_REQUIRE(ctx && C_host, "null argument");
  MI_REQUIRE(ks >= 1 && ks <= kGramMaxK && kc >= 1 && kc <= kGramMaxK && ldc >= ks, "bad small-matrix shape");
  MI_TRY(check_panel(ctx, m, ks, S, "S"));
  MI_TRY(check_panel(ctx, m, kc, Y, "Y"));
  MI_REQUIRE(S->d != Y->d, "in-place panel update is not supported");
  void *Cdev = nullptr;
  MI_TRY(pool_alloc(ctx, (size_t)ldc * kc * sizeof(double), &Cdev));
  MI_HIP(hipMemcpyAsync(Cdev, C_host, (size_t)ldc * kc * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  // the host buffer may be reused by the caller right after we return
  MI_HIP(hipStreamSynchronize(ctx->stream));
  const int grid = (int)std::min<size_t>((m + 255) / 256, 2048);
  KScope ksc(ctx, MI_K_LOBPCG_UPDATE);
  for (int c0 = 0; c0 < kc; c0 += 8)
    hipLaunchKernelGGL(k_panel_update<8>, dim3(grid), dim3(256), (size_t)ks * 8 * sizeof(double), ctx->stream, m,
                       ks, (const double *)S->d, (const double *)Cdev, ldc, c0, kc, Y->d);
  MI_HIP(hipGetLastError());
  pool_free(ctx, Cdev);  // stream-ordered reuse: later allocations are enqueued after these kernels
  return MI_OK;
}

int mi_lobpcg_residual(mi_ctx *ctx, size_t m, int nx, const mi_vec *AX, const mi_vec *BX, const mi_vec *X,
                       const double *theta_host, mi_vec *R, double *rnorm, double *xnorm) {
  MI_REQUIRE(ctx && theta_host && rnorm && xnorm, "null argument");
  MI_REQUIRE(nx >= 1 && nx <= kGramMaxK, "block size must be in [1,%d]", kGramMaxK);
  MI_TRY(check_panel(ctx, m, nx, AX, "AX"));
  MI_TRY(check_panel(ctx, m, nx, BX, "BX"));
  MI_TRY(check_panel(ctx, m, nx, X, "X"));
  MI_TRY(check_panel(ctx, m, nx, R, "R"));
  void *thdev = nullptr;
  MI_TRY(pool_alloc(ctx, (size_t)nx * sizeof(double), &thdev));
  MI_HIP(hipMemcpyAsync(thdev, theta_host, (size_t)nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MI_HIP(hipStreamSynchronize(ctx->stream));
  const int grid = grid_for(m, 2);
  double *slots = ctx->scalars + SLOT_GRAM;
  for (int c0 = 0; c0 < nx; c0 += 8) {
    {
      KScope ks(ctx, MI_K_LOBPCG_RESIDUAL);
      hipLaunchKernelGGL(k_residual, dim3(grid), dim3(kBlock), 0, ctx->stream, m, nx, c0, (const double *)AX->d,
                         (const double *)BX->d, (const double *)X->d, (const double *)thdev, R->d, ctx->partials2);
    }
    MI_TRY(launch_reduce_rows_to_slots(ctx, ctx->partials2, grid, 16, slots));
    MI_TRY(comm_allreduce(ctx, slots, 16));
    double out[16];
    MI_TRY(read_slots_sync(ctx, SLOT_GRAM, 16, out));
    for (int c = 0; c < 8 && c0 + c < nx; ++c) {
      rnorm[c0 + c] = std::sqrt(out[c]);
      xnorm[c0 + c] = std::sqrt(out[8 + c]);
    }
  }
  pool_free(ctx, thdev);
  return MI_OK;
}

int mi_rayleigh_ritz(int n, const double *A, const double *B, double *Theta, double *C) {
  MI_REQUIRE(n >= 1 && A && B && Theta && C, "bad argument");
  std::vector<double> D(n), L((size_t)n * n), M((size_t)n * n), T((size_t)n * n), Y;
  for (int i = 0; i < n; ++i) {
    MI_REQUIRE(B[i + (size_t)i * n] > 0, "B has a non-positive diagonal entry (%d)", i);
    D[i] = 1.0 / std::sqrt(B[i + (size_t)i * n]);  // LOBPCG.h:56
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      L[i + (size_t)j * n] = D[i] * B[i + (size_t)j * n] * D[j];  // D B D  :59
      M[i + (size_t)j * n] = D[i] * A[i + (size_t)j * n] * D[j];  // D A D  :59
    }
  if (cholesky_lower(n, L)) {
    set_error("Rayleigh-Ritz: equilibrated B is not positive definite");
    return MI_ERR_INVALID_ARGUMENT;
  }
  for (int j = 0; j < n; ++j)  // T = L^-1 M
    for (int i = 0; i < n; ++i) {
      double s = M[i + (size_t)j * n];
      for (int k = 0; k < i; ++k) s -= L[i + (size_t)k * n] * T[k + (size_t)j * n];
      T[i + (size_t)j * n] = s / L[i + (size_t)i * n];
    }
  for (int j = 0; j < n; ++j)  // M = T L^-T
    for (int i = 0; i < n; ++i) {
      double s = T[i + (size_t)j * n];
      for (int k = 0; k < j; ++k) s -= M[i + (size_t)k * n] * L[j + (size_t)k * n];
      M[i + (size_t)j * n] = s / L[j + (size_t)j * n];
    }
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) {
      const double a = .5 * (M[i + (size_t)j * n] + M[j + (size_t)i * n]);
      M[i + (size_t)j * n] = a;
      M[j + (size_t)i * n] = a;
    }
  jacobi_eigh(n, M, Y, Theta);
  for (int j = 0; j < n; ++j) {  // x = L^-T y ; C = D x  (:61)
    for (int i = n; i-- > 0;) {
      double s = Y[i + (size_t)j * n];
      for (int k = i + 1; k < n; ++k) s -= L[k + (size_t)i * n] * T[k + (size_t)j * n];
      T[i + (size_t)j * n] = s / L[i + (size_t)i * n];
    }
    for (int i = 0; i < n; ++i) C[i + (size_t)j * n] = D[i] * T[i + (size_t)j * n];
  }
  return MI_OK;
}

int mi_panel_rowscale(mi_ctx *ctx, size_t m, int k, const mi_vec *d, const mi_vec *X, mi_vec *Y) {
  MI_REQUIRE(ctx && d, "null argument");
  MI_REQUIRE(d->n == m, "scaling vector must have m entries");
  MI_TRY(check_panel(ctx, m, k, X, "X"));
  MI_TRY(check_panel(ctx, m, k, Y, "Y"));
  const int grid = (int)std::min<size_t>((m * (size_t)k + 255) / 256, 4096);
  hipLaunchKernelGGL(k_rowscale, dim3(grid), dim3(256), 0, ctx->stream, m, (size_t)k, (const double *)d->d,
                     (const double *)X->d, Y->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_csr_spmm_colmajor(const mi_csr *A, int k, const mi_vec *X, mi_vec *Y) {
  MI_REQUIRE(A && X && Y, "null argument");
  MI_REQUIRE(k >= 1 && k <= kGramMaxK, "panel width must be in [1,%d]", kGramMaxK);
  MI_REQUIRE(A->halo_lo + A->halo_hi == 0, "column-major SpMM does not support sharded matrices yet");
  MI_TRY(check_panel(A->ctx, A->n, k, X, "X"));
  MI_TRY(check_panel(A->ctx, A->n, k, Y, "Y"));
  MI_REQUIRE(X->d != Y->d, "SpMM input and output must not alias");
  mi_ctx *ctx = A->ctx;
  const int grid = (int)((A->nslices + 3) / 4);
  KScope ks(ctx, MI_K_SPMM);
  for (int c0 = 0; c0 < k; c0 += 8)
    hipLaunchKernelGGL(k_spmm_colmajor, dim3(grid), dim3(256), 0, ctx->stream, A->n, A->nslices,
                       (const long long *)A->slice_ptr, (const int *)A->col, (const double *)A->val, k, c0,
                       (const double *)X->d, Y->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // extern "C"
