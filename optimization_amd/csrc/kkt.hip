// kkt.hip -- the constraint preconditioner of the PROJECTED Steihaug-Toint CG (reference
// LinearAlgebra/IterativeSolvers.h:83-85 `STPCGPreconditioner`, applied at :229-253 and :381-405; the reference's
// own cases: tests/IterativeSolvers_unit_test.cpp:316-496, n = 1000, m = 100 dense constraints, diagonal M):
//        [M  A'][v]   [r]
//        [A  0 ][l] = [0]        <=>   l = S^-1 A M^-1 r,   v = M^-1 (r - A' l),   S = A M^-1 A'
// followed (when STPCG was given `At`) by the residual correction r -= A' l of :251,403.
//
// Device form (small m, the case the reference tests): S is formed, Cholesky-factored and inverted ONCE, on the
// device, when the preconditioner is created.  Every application is two launches and no scalar kernel:
//   k_kkt_b       per-workgroup partial rows of b = A (M^-1 r): one wave per constraint row over the workgroup's
//                 column chunk (coalesced reads of A's rows), fixed-shape wave reductions
//   k_kkt_finish  EVERY workgroup re-reduces b from the partial rows and forms l = S^-1 b in its prologue (identical
//                 code on identical data: identical bits everywhere, no atomics, no one-workgroup kernel); body:
//                 w_i = sum_a A[a,i] l_a (coalesced over i), v_i = M^-1_i (r_i - w_i) and, if asked, r_i -= w_i
// A is m x n row-major (a constraint row contiguous), i.e. the column-major n x m panel of A'.
//
// General form (r04, mi_precon_create_constraint_csr: SPARSE constraints, any m): A and A' in CSR, S = A M^-1 A' formed
// as a sparse matrix once (host SpGEMM at creation); an application is three launches and still no host round trip:
//   k_kkt_b_csr     b = A (M^-1 r): one wave per constraint row
//   k_kkt_cg        S l = b by a Jacobi-preconditioned conjugate-gradient iteration that lives entirely inside ONE
//                   workgroup (m is small next to n: its vectors stay in L2, its reductions are workgroup reductions in
//                   a fixed order -- deterministic, no atomics, no launch per inner iteration), run to a relative
//                   residual of 1e-14 by default: the projection must be accurate to rounding or the outer iterates
//                   leave the null space of A
//   k_kkt_finish_csr  w = A' l (CSR of A': one thread per column of A), v = M^-1 (r - w), and r -= w for `At`
#include <algorithm>
#include <map>
#include <vector>

#include "mi_internal.h"

using namespace mi;

namespace {

constexpr int kKktMaxM = 512;  // constraints (the prologue holds b and l in LDS; the one-off inversion is O(m^2) per thread)

struct KktImpl {
  size_t n = 0, m = 0;
  const double *A = nullptr;     // m x n row-major, caller-owned
  const double *Minv = nullptr;  // n, caller-owned
  double *Sinv = nullptr;        // m x m
  double *bpart = nullptr;       // m x kMaxGrid partial sums of b
  double *lambda = nullptr;      // m (the multiplier estimate of the last application)
  // sparse form (mi_precon_create_constraint_csr)
  bool sparse = false;
  int *a_ptr = nullptr, *a_col = nullptr;    // A, CSR, m rows
  double *a_val = nullptr;
  int *t_ptr = nullptr, *t_col = nullptr;    // A', CSR, n rows
  double *t_val = nullptr;
  int *s_ptr = nullptr, *s_col = nullptr;    // S = A M^-1 A', CSR, m rows
  double *s_val = nullptr, *s_dinv = nullptr;
  double *work = nullptr;                    // b, r, z, p, q: 5 m doubles
  double *info = nullptr;                    // [0] inner iterations of the last application, [1] its relative residual,
                                             // [2] the largest relative residual any application ended with,
                                             // [3] STICKY failure word: 0 ok, 1 the inner iteration broke down (p'Sp <= 0
                                             // or not finite: dependent constraint rows, NaN in r), 2 it stopped at
                                             // max_inner short of the tolerance; checked behind the solve's own read-back
  double tol = 1e-14;
  int max_inner = 0;
};

__global__ __launch_bounds__(256) void k_kkt_schur(size_t n, int m, const double *__restrict__ A,
                                                   const double *__restrict__ Minv, double *__restrict__ S) {
  // workgroup t <-> pair (a, b <= a)
  __shared__ double lds[4];
  int a = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((size_t)(a + 1) * (a + 2) / 2 <= blockIdx.x) ++a;
  while ((size_t)a * (a + 1) / 2 > blockIdx.x) --a;
  const int b = (int)(blockIdx.x - (size_t)a * (a + 1) / 2);
  const double *ra = A + (size_t)a * n, *rb = A + (size_t)b * n;
  double acc = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) acc += ra[i] * Minv[i] * rb[i];
  acc = wave_reduce_sum(acc);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    S[(size_t)a * m + b] = s;
    S[(size_t)b * m + a] = s;
  }
}

// one workgroup: S -> L (lower Cholesky, in place), then Sinv column by column (thread c solves L L' x = e_c)
__global__ __launch_bounds__(kBlock) void k_kkt_invert(int m, double *__restrict__ S, double *__restrict__ Sinv,
                                                       int *__restrict__ fail) {
  __shared__ int bad;
  __shared__ double d0[kKktMaxM];  // the diagonal before elimination: a pivot that cancelled to rounding level relative
                                   // to it means linearly dependent constraint rows, not a tiny positive number
  if (threadIdx.x == 0) bad = 0;
  for (int j = threadIdx.x; j < m; j += kBlock) d0[j] = S[(size_t)j * m + j];
  __syncthreads();
  for (int j = 0; j < m; ++j) {
    if (threadIdx.x == 0) {
      const double d = S[(size_t)j * m + j];
      if (!(d > 1e-12 * d0[j]) || !(d0[j] > 0)) bad = 1;
      S[(size_t)j * m + j] = sqrt(d);
    }
    __syncthreads();
    if (bad) break;
    const double ljj = S[(size_t)j * m + j];
    for (int i = j + 1 + threadIdx.x; i < m; i += kBlock) S[(size_t)i * m + j] /= ljj;
    __syncthreads();
    // trailing update of the lower triangle: S[i][k] -= L[i][j] L[k][j], j < k <= i
    const int t = m - j - 1;
    for (int e = threadIdx.x; e < t * t; e += kBlock) {
      const int i = j + 1 + e / t, k = j + 1 + e % t;
      if (k <= i) S[(size_t)i * m + k] -= S[(size_t)i * m + j] * S[(size_t)k * m + j];
    }
    __syncthreads();
  }
  if (bad) {
    if (threadIdx.x == 0) *fail = 1;
    return;
  }
  for (int c = threadIdx.x; c < m; c += kBlock) {
    double *x = Sinv + (size_t)c * m;  // row c of the (symmetric) inverse
    for (int i = 0; i < m; ++i) {      // L y = e_c
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= S[(size_t)i * m + k] * x[k];
      x[i] = s / S[(size_t)i * m + i];
    }
    for (int i = m - 1; i >= 0; --i) {  // L' x = y
      double s = x[i];
      for (int k = i + 1; k < m; ++k) s -= S[(size_t)k * m + i] * x[k];
      x[i] = s / S[(size_t)i * m + i];
    }
  }
}

__device__ __forceinline__ void kkt_chunk(size_t n, size_t &c0, size_t &c1) {
  // contiguous column chunk of this workgroup, a multiple of 64 columns
  const size_t per = ((n + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
  c0 = (size_t)blockIdx.x * per;
  c1 = c0 + per < n ? c0 + per : n;
  if (c0 > n) c0 = n;
}

__global__ __launch_bounds__(kBlock) void k_kkt_b(size_t n, int m, const double *__restrict__ A,
                                                  const double *__restrict__ Minv, const double *__restrict__ r,
                                                  double *__restrict__ bpart) {
  size_t c0, c1;
  kkt_chunk(n, c0, c1);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int a = w; a < m; a += kWaves) {
    const double *row = A + (size_t)a * n;
    double acc = 0;
    for (size_t i = c0 + lane; i < c1; i += 64) acc += row[i] * (Minv[i] * r[i]);
    acc = wave_reduce_sum(acc);
    if (lane == 0) bpart[(size_t)a * kMaxGrid + blockIdx.x] = acc;
  }
}

template <bool SUBTRACT>
__global__ __launch_bounds__(kBlock) void k_kkt_finish(size_t n, int m, int nparts, const double *__restrict__ A,
                                                       const double *__restrict__ Minv,
                                                       const double *__restrict__ Sinv,
                                                       const double *__restrict__ bpart, double *__restrict__ r,
                                                       double *__restrict__ v, double *__restrict__ lambda_out) {
  __shared__ double bs[kKktMaxM], ls[kKktMaxM];
  for (int a = threadIdx.x; a < m; a += kBlock) {
    const double *src = bpart + (size_t)a * kMaxGrid;
    double s = 0;
    for (int j = 0; j < nparts; ++j) s += src[j];
    bs[a] = s;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < m; a += kBlock) {
    const double *row = Sinv + (size_t)a * m;
    double s = 0;
    for (int b = 0; b < m; ++b) s += row[b] * bs[b];
    ls[a] = s;
    if (blockIdx.x == 0) lambda_out[a] = s;
  }
  __syncthreads();
  size_t c0, c1;
  kkt_chunk(n, c0, c1);
  for (size_t i = c0 + threadIdx.x; i < c1; i += kBlock) {
    double w = 0;
    for (int a = 0; a < m; ++a) w += A[(size_t)a * n + i] * ls[a];
    const double ri = r[i], d = ri - w;
    v[i] = Minv[i] * d;
    if (SUBTRACT) r[i] = d;
  }
}

__global__ __launch_bounds__(kBlock) void k_kkt_At(size_t n, int m, const double *__restrict__ A,
                                                   const double *__restrict__ l, double *__restrict__ out) {
  __shared__ double ls[kKktMaxM];
  for (int a = threadIdx.x; a < m; a += kBlock) ls[a] = l[a];
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    double w = 0;
    for (int a = 0; a < m; ++a) w += A[(size_t)a * n + i] * ls[a];
    out[i] = w;
  }
}


// ---- sparse constraints --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kkt_b_csr(int m, const int *__restrict__ ptr, const int *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ Minv,
                                                   const double *__restrict__ r, double *__restrict__ b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  double acc = 0;
  for (int k = ptr[row] + lane; k < ptr[row + 1]; k += 64) {
    const int j = col[k];
    acc += val[k] * (Minv[j] * r[j]);
  }
  acc = wave_reduce_sum(acc);
  if (lane == 0) b[row] = acc;
}

// workgroup-wide sum, the same bits in every thread (fixed shape); lds: kWaves doubles.  Contains barriers.
__device__ __forceinline__ double wg_sum(double v, double *lds) {
  v = wave_reduce_sum(v);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  const double s = sum16(lds);
  __syncthreads();
  return s;
}

// S l = b, Jacobi-preconditioned CG, ONE workgroup.  work: b, r, z, p, q (m doubles each).
__global__ __launch_bounds__(kBlock) void k_kkt_cg(int m, const int *__restrict__ ptr, const int *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ dinv,
                                                   double *__restrict__ work, double *__restrict__ l, double tol,
                                                   int max_inner, double *__restrict__ info) {
  __shared__ double lds[kWaves];
  const double *b = work;
  double *r = work + m, *z = work + 2 * (size_t)m, *p = work + 3 * (size_t)m, *q = work + 4 * (size_t)m;
  const int t = threadIdx.x;
  double a0 = 0, a1 = 0;
  for (int i = t; i < m; i += kBlock) {
    const double bi = b[i], zi = dinv[i] * bi;
    l[i] = 0;
    r[i] = bi;
    z[i] = zi;
    p[i] = zi;
    a0 += bi * zi;
    a1 += bi * bi;
  }
  double rz = wg_sum(a0, lds);
  const double bb = wg_sum(a1, lds);
  double rr = bb;
  int it = 0;
  // (all threads hold the same rz, rr, bb: the loop control is uniform)
  int broke = 0;
  while (it < max_inner && rr > tol * tol * bb && bb > 0) {
    __syncthreads();  // p of the previous step is complete
    double pq = 0;
    for (int i = t; i < m; i += kBlock) {
      double s = 0;
      for (int k = ptr[i]; k < ptr[i + 1]; ++k) s += val[k] * p[col[k]];
      q[i] = s;
      pq += p[i] * s;
    }
    pq = wg_sum(pq, lds);
    // S is positive definite for independent constraint rows: anything else (semi-definite S, NaN / Inf in r) would
    // make alpha garbage and end the loop through a false comparison with a NaN, looking like convergence
    if (!(pq > 0) || !(pq <= 1.79769313486231571e308)) {
      broke = 1;
      break;
    }
    const double alpha = rz / pq;
    a0 = 0;
    a1 = 0;
    for (int i = t; i < m; i += kBlock) {
      l[i] += alpha * p[i];
      const double ri = r[i] - alpha * q[i], zi = dinv[i] * ri;
      r[i] = ri;
      z[i] = zi;
      a0 += ri * zi;
      a1 += ri * ri;
    }
    const double rz_new = wg_sum(a0, lds);
    rr = wg_sum(a1, lds);
    const double beta = rz_new / rz;
    rz = rz_new;
    for (int i = t; i < m; i += kBlock) p[i] = z[i] + beta * p[i];
    ++it;
  }
  if (t == 0) {
    const double rel = bb > 0 ? sqrt(rr / bb) : (bb == 0 ? 0.0 : bb /* NaN */);
    info[0] = (double)it;
    info[1] = rel;
    if (!(rel <= info[2])) info[2] = rel;  // (a NaN is recorded, and stays)
    if (broke || !(bb >= 0))
      info[3] = 1.0;
    else if (!(rr <= tol * tol * bb) && info[3] == 0.0)
      info[3] = 2.0;
  }
}

template <bool SUBTRACT, bool V_OUT>
__global__ __launch_bounds__(kBlock) void k_kkt_finish_csr(size_t n, const int *__restrict__ ptr,
                                                           const int *__restrict__ col, const double *__restrict__ val,
                                                           const double *__restrict__ Minv, const double *__restrict__ l,
                                                           double *__restrict__ r, double *__restrict__ v) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    double w = 0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) w += val[k] * l[col[k]];
    if (V_OUT) {
      const double d = r[i] - w;
      v[i] = Minv[i] * d;
      if (SUBTRACT) r[i] = d;
    } else {
      v[i] = w;  // A' l alone (mi_precon_constraint_At)
    }
  }
}

int kkt_run_csr(mi_precon *self, mi_vec *r, mi_vec *v, int subtract) {
  KktImpl *k = (KktImpl *)self->impl;
  mi_ctx *ctx = self->ctx;
  hipLaunchKernelGGL(k_kkt_b_csr, dim3((unsigned)((k->m + 3) / 4)), dim3(256), 0, ctx->stream, (int)k->m,
                     (const int *)k->a_ptr, (const int *)k->a_col, (const double *)k->a_val, k->Minv,
                     (const double *)r->d, k->work);
  hipLaunchKernelGGL(k_kkt_cg, dim3(1), dim3(kBlock), 0, ctx->stream, (int)k->m, (const int *)k->s_ptr,
                     (const int *)k->s_col, (const double *)k->s_val, (const double *)k->s_dinv, k->work, k->lambda,
                     k->tol, k->max_inner, k->info);
  const int grid = grid_for(ctx, k->n, 1);
  if (subtract)
    hipLaunchKernelGGL((k_kkt_finish_csr<true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, k->n,
                       (const int *)k->t_ptr, (const int *)k->t_col, (const double *)k->t_val, k->Minv,
                       (const double *)k->lambda, r->d, v->d);
  else
    hipLaunchKernelGGL((k_kkt_finish_csr<false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, k->n,
                       (const int *)k->t_ptr, (const int *)k->t_col, (const double *)k->t_val, k->Minv,
                       (const double *)k->lambda, r->d, v->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}
int kkt_apply_csr(mi_precon *self, const mi_vec *r, mi_vec *v) { return kkt_run_csr(self, const_cast<mi_vec *>(r), v, 0); }
void kkt_destroy_csr(mi_precon *self) {
  KktImpl *k = (KktImpl *)self->impl;
  (void)hipStreamSynchronize(self->ctx->stream);
  for (void *q : {(void *)k->a_ptr, (void *)k->a_col, (void *)k->a_val, (void *)k->t_ptr, (void *)k->t_col,
                  (void *)k->t_val, (void *)k->s_ptr, (void *)k->s_col, (void *)k->s_val, (void *)k->s_dinv,
                  (void *)k->work, (void *)k->info, (void *)k->lambda})
    if (q) (void)hipFree(q);
  delete k;
}

int kkt_run(mi_precon *self, mi_vec *r, mi_vec *v, int subtract) {
  KktImpl *k = (KktImpl *)self->impl;
  mi_ctx *ctx = self->ctx;
  const int grid = grid_for(ctx, k->n, 1);
  hipLaunchKernelGGL(k_kkt_b, dim3(grid), dim3(kBlock), 0, ctx->stream, k->n, (int)k->m, k->A, k->Minv,
                     (const double *)r->d, k->bpart);
  if (subtract)
    hipLaunchKernelGGL(k_kkt_finish<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, k->n, (int)k->m, grid, k->A,
                       k->Minv, (const double *)k->Sinv, (const double *)k->bpart, r->d, v->d, k->lambda);
  else
    hipLaunchKernelGGL(k_kkt_finish<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, k->n, (int)k->m, grid, k->A,
                       k->Minv, (const double *)k->Sinv, (const double *)k->bpart, r->d, v->d, k->lambda);
  MI_HIP(hipGetLastError());
  return MI_OK;
}
int kkt_apply(mi_precon *self, const mi_vec *r, mi_vec *v) { return kkt_run(self, const_cast<mi_vec *>(r), v, 0); }
void kkt_destroy(mi_precon *self) {
  KktImpl *k = (KktImpl *)self->impl;
  (void)hipStreamSynchronize(self->ctx->stream);
  (void)hipFree(k->Sinv);
  (void)hipFree(k->bpart);
  (void)hipFree(k->lambda);
  delete k;
}

}  // namespace

extern "C" {

int mi_precon_create_constraint(mi_ctx *ctx, size_t n, size_t m, const mi_vec *A, const mi_vec *Minv,
                                mi_precon **out) {
  MI_REQUIRE(ctx && A && Minv && out, "null argument");
  MI_REQUIRE(A->ctx == ctx && Minv->ctx == ctx, "vector belongs to another context");
  MI_REQUIRE(m >= 1 && m <= (size_t)kKktMaxM, "number of constraints must be in [1,%d], got %zu", kKktMaxM, m);
  MI_REQUIRE(n >= m && A->n == n * m && Minv->n == n, "constraint matrix must be m x n row-major (m <= n), M^-1 of length n");
  // S = A M^-1 A' and b = A M^-1 r are sums over ALL columns: on a row-sharded context they would be rank-local partial
  // sums nobody reduces (a wrong, rank-dependent lambda; the replicated CG scalars would part ways)
  MI_REQUIRE(ctx->world_size <= 1, "the constraint preconditioner is single-rank (context has %d ranks): its Schur "
             "complement and right-hand sides are not reduced across ranks", ctx->world_size);
  MI_TRY(ensure_device());
  KktImpl *k = new KktImpl();
  k->n = n;
  k->m = m;
  k->A = A->d;
  k->Minv = Minv->d;
  double *S = nullptr;
  int *fail = nullptr;
  int st = MI_OK, failed = 0;
  auto hipok = [&](hipError_t e, const char *what) {
    if (e != hipSuccess && st == MI_OK) st = hip_fail(e, what, __FILE__, __LINE__);
  };
  hipok(hipMalloc((void **)&k->Sinv, m * m * sizeof(double)), "hipMalloc Sinv");
  hipok(hipMalloc((void **)&k->bpart, m * kMaxGrid * sizeof(double)), "hipMalloc bpart");
  hipok(hipMalloc((void **)&k->lambda, m * sizeof(double)), "hipMalloc lambda");
  hipok(hipMalloc((void **)&S, m * m * sizeof(double)), "hipMalloc S");
  hipok(hipMalloc((void **)&fail, sizeof(int)), "hipMalloc flag");
  if (st == MI_OK) {
    hipok(hipMemsetAsync(fail, 0, sizeof(int), ctx->stream), "memset");
    hipLaunchKernelGGL(k_kkt_schur, dim3((unsigned)(m * (m + 1) / 2)), dim3(256), 0, ctx->stream, n, (int)m, k->A,
                       k->Minv, S);
    hipLaunchKernelGGL(k_kkt_invert, dim3(1), dim3(kBlock), 0, ctx->stream, (int)m, S, k->Sinv, fail);
    hipok(hipGetLastError(), "kkt setup launch");
    hipok(hipMemcpyAsync(&failed, fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream), "flag copy");
    hipok(hipStreamSynchronize(ctx->stream), "kkt setup");
  }
  (void)hipFree(S);
  (void)hipFree(fail);
  if (st == MI_OK && failed) {
    set_error("A M^-1 A' is not positive definite: the constraint rows are linearly dependent (or M^-1 is not positive)");
    st = MI_ERR_INVALID_ARGUMENT;
  }
  if (st != MI_OK) {
    (void)hipFree(k->Sinv);
    (void)hipFree(k->bpart);
    (void)hipFree(k->lambda);
    delete k;
    return st;
  }
  mi_precon *P = new mi_precon();
  P->ctx = ctx;
  P->n = n;
  P->kind = 3;
  P->apply = kkt_apply;
  P->apply_project = kkt_run;
  P->destroy = kkt_destroy;
  P->impl = k;
  *out = P;
  return MI_OK;
}

int mi_precon_create_constraint_csr(mi_ctx *ctx, size_t n, size_t m, const int32_t *rowptr, const int32_t *col,
                                    const double *val, const mi_vec *Minv, double inner_tol,
                                    size_t inner_max_iterations, mi_precon **out) {
  MI_REQUIRE(ctx && rowptr && Minv && out, "null argument");
  MI_REQUIRE(Minv->ctx == ctx && Minv->n == n, "M^-1 must be a vector of length n on this context");
  MI_REQUIRE(m >= 1 && m <= n && n < (size_t)INT32_MAX, "need 1 <= m <= n < 2^31");
  MI_REQUIRE(rowptr[0] == 0 && rowptr[m] >= 0, "bad row pointers");
  const size_t nnz = (size_t)rowptr[m];
  MI_REQUIRE(nnz == 0 || (col && val), "null argument");
  for (size_t a = 0; a < m; ++a) {
    MI_REQUIRE(rowptr[a + 1] >= rowptr[a], "row pointers must not decrease");
    for (int32_t k = rowptr[a]; k < rowptr[a + 1]; ++k)
      MI_REQUIRE(col[k] >= 0 && (size_t)col[k] < n, "column index %d of constraint %zu out of range", col[k], a);
  }
  MI_REQUIRE(ctx->world_size <= 1, "the constraint preconditioner is single-rank (context has %d ranks)", ctx->world_size);
  MI_REQUIRE(!(inner_tol < 0) && inner_tol < 1, "inner tolerance must be in [0, 1)");
  MI_TRY(ensure_device());
  // --- host: A' (CSR of the transpose) and S = A M^-1 A' (sparse, sorted columns)
  std::vector<double> minv(n);
  MI_HIP(hipMemcpyAsync(minv.data(), Minv->d, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MI_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int> tptr(n + 1, 0), tcol(nnz);
  std::vector<double> tval(nnz);
  for (size_t k = 0; k < nnz; ++k) ++tptr[(size_t)col[k] + 1];
  for (size_t i = 0; i < n; ++i) tptr[i + 1] += tptr[i];
  {
    std::vector<int> fill(tptr.begin(), tptr.end() - 1);
    for (size_t a = 0; a < m; ++a)
      for (int32_t k = rowptr[a]; k < rowptr[a + 1]; ++k) {
        const int pos = fill[(size_t)col[k]]++;
        tcol[pos] = (int)a;
        tval[pos] = val[k];
      }
  }
  std::vector<int> sptr(m + 1, 0), scol;
  std::vector<double> sval, sdinv(m);
  {
    std::map<int, double> acc;
    for (size_t a = 0; a < m; ++a) {
      acc.clear();
      for (int32_t k = rowptr[a]; k < rowptr[a + 1]; ++k) {
        const size_t j = (size_t)col[k];
        const double f = val[k] * minv[j];
        for (int q = tptr[j]; q < tptr[j + 1]; ++q) acc[tcol[q]] += f * tval[q];
      }
      const auto d = acc.find((int)a);
      if (d == acc.end() || !(d->second > 0)) {
        set_error("A M^-1 A' has a non-positive diagonal entry at constraint %zu (an empty row, or M^-1 not positive)", a);
        return MI_ERR_INVALID_ARGUMENT;
      }
      sdinv[a] = 1.0 / d->second;
      for (const auto &e : acc) {
        scol.push_back(e.first);
        sval.push_back(e.second);
      }
      sptr[a + 1] = (int)scol.size();
    }
  }
  KktImpl *k = new KktImpl();
  k->sparse = true;
  k->n = n;
  k->m = m;
  k->Minv = Minv->d;
  k->tol = inner_tol > 0 ? inner_tol : 1e-14;
  k->max_inner = (int)std::min<size_t>(inner_max_iterations ? inner_max_iterations : 10 * m + 100, (size_t)INT32_MAX);
  int st = MI_OK;
  auto up = [&](void **dst, const void *src, size_t bytes) {
    if (st != MI_OK) return;
    hipError_t e = hipMalloc(dst, bytes ? bytes : 8);
    if (e == hipSuccess && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) st = hip_fail(e, "constraint preconditioner upload", __FILE__, __LINE__);
  };
  std::vector<int> aptr(rowptr, rowptr + m + 1), acol(col, col + nnz);
  up((void **)&k->a_ptr, aptr.data(), (m + 1) * sizeof(int));
  up((void **)&k->a_col, acol.data(), nnz * sizeof(int));
  up((void **)&k->a_val, val, nnz * sizeof(double));
  up((void **)&k->t_ptr, tptr.data(), (n + 1) * sizeof(int));
  up((void **)&k->t_col, tcol.data(), nnz * sizeof(int));
  up((void **)&k->t_val, tval.data(), nnz * sizeof(double));
  up((void **)&k->s_ptr, sptr.data(), (m + 1) * sizeof(int));
  up((void **)&k->s_col, scol.data(), scol.size() * sizeof(int));
  up((void **)&k->s_val, sval.data(), sval.size() * sizeof(double));
  up((void **)&k->s_dinv, sdinv.data(), m * sizeof(double));
  const double zeros[4] = {0, 0, 0, 0};
  up((void **)&k->info, zeros, sizeof(zeros));
  if (st == MI_OK) {
    hipError_t e = hipMalloc((void **)&k->work, 5 * m * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&k->lambda, m * sizeof(double));
    if (e != hipSuccess) st = hip_fail(e, "constraint preconditioner work space", __FILE__, __LINE__);
  }
  mi_precon *P = new mi_precon();
  P->ctx = ctx;
  P->n = n;
  P->kind = 3;
  P->apply = kkt_apply_csr;
  P->apply_project = kkt_run_csr;
  P->destroy = kkt_destroy_csr;
  P->impl = k;
  P->fail_word = k->info ? k->info + 3 : nullptr;
  if (st != MI_OK) {
    kkt_destroy_csr(P);
    delete P;
    return st;
  }
  *out = P;
  return MI_OK;
}

int mi_precon_constraint_info(mi_precon *P, size_t *last_inner_iterations, double *last_relative_residual,
                              double *worst_relative_residual, int *status_code) {
  MI_REQUIRE(P && P->kind == 3, "not a constraint preconditioner");
  KktImpl *k = (KktImpl *)P->impl;
  double h[4] = {0, 0, 0, 0};
  if (k->sparse) {
    MI_HIP(hipMemcpyAsync(h, k->info, sizeof(h), hipMemcpyDeviceToHost, P->ctx->stream));
    MI_HIP(hipStreamSynchronize(P->ctx->stream));
    P->ctx->host_syncs++;
  }
  if (last_inner_iterations) *last_inner_iterations = (size_t)h[0];
  if (last_relative_residual) *last_relative_residual = h[1];
  if (worst_relative_residual) *worst_relative_residual = h[2];
  // the getter itself succeeds whatever the inner iteration did: the code is an OUTPUT (1: broke down -- p'Sp <= 0 or not
  // finite; 2: stopped at its iteration limit short of the tolerance), so that a caller can read the residual exactly
  // when it needs it
  if (status_code) *status_code = (int)h[3];
  return MI_OK;
}

int mi_precon_constraint_solve(mi_precon *P, const mi_vec *r, mi_vec *v, mi_vec *lambda) {
  MI_REQUIRE(P && r && v, "null argument");
  MI_REQUIRE(P->kind == 3, "not a constraint preconditioner");
  KktImpl *k = (KktImpl *)P->impl;
  MI_REQUIRE(r->n == k->n && v->n == k->n && (!lambda || lambda->n == k->m), "dimension mismatch");
  touch(v);
  MI_TRY(P->apply_project(P, const_cast<mi_vec *>(r), v, 0));
  if (lambda) {
    touch(lambda);
    MI_HIP(hipMemcpyAsync(lambda->d, k->lambda, k->m * sizeof(double), hipMemcpyDeviceToDevice, P->ctx->stream));
  }
  return MI_OK;
}

int mi_precon_constraint_At(mi_precon *P, const mi_vec *lambda, mi_vec *out) {
  MI_REQUIRE(P && lambda && out, "null argument");
  MI_REQUIRE(P->kind == 3, "not a constraint preconditioner");
  KktImpl *k = (KktImpl *)P->impl;
  MI_REQUIRE(lambda->n == k->m && out->n == k->n, "dimension mismatch");
  touch(out);
  if (k->sparse)
    hipLaunchKernelGGL((k_kkt_finish_csr<false, false>), dim3(grid_for(P->ctx, k->n, 1)), dim3(kBlock), 0, P->ctx->stream,
                       k->n, (const int *)k->t_ptr, (const int *)k->t_col, (const double *)k->t_val, k->Minv,
                       (const double *)lambda->d, (double *)nullptr, out->d);
  else
    hipLaunchKernelGGL(k_kkt_At, dim3(grid_for(P->ctx, k->n, 1)), dim3(kBlock), 0, P->ctx->stream, k->n, (int)k->m,
                       k->A, (const double *)lambda->d, out->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

}  // extern "C"
