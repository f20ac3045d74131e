// blas1.hip -- device vectors and the Vector-concept operators as
// HBM-streaming gfx950 kernels: 16 B/lane loads, grid-stride over <= 512 fat workgroups,
// deterministic two-stage reductions (per-workgroup partial row -> fixed-order sum; no fp64 atomics).
#include "mi_internal.h"

using namespace mi;

namespace {

// MODE 0: z = a*x + b*y ; MODE 1: z = a*x ; MODE 2: z = a ; MODE 3: z = x / a (a true division, like the host code)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_axpby(size_t n, double a, const double *__restrict__ x,
                                                  double b, const double *__restrict__ y,
                                                  double *__restrict__ z) {
  const size_t n2 = n >> 1;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    double2 r;
    if (MODE == 2) {
      r.x = a; r.y = a;
    } else {
      const double2 xv = reinterpret_cast<const double2 *>(x)[i];
      if (MODE == 1) {
        r.x = a * xv.x; r.y = a * xv.y;
      } else if (MODE == 3) {
        r.x = xv.x / a; r.y = xv.y / a;
      } else {
        const double2 yv = reinterpret_cast<const double2 *>(y)[i];
        r.x = a * xv.x + b * yv.x;
        r.y = a * xv.y + b * yv.y;
      }
    }
    reinterpret_cast<double2 *>(z)[i] = r;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const size_t i = n - 1;
    z[i] = (MODE == 2) ? a : (MODE == 1) ? a * x[i] : (MODE == 3) ? x[i] / a : a * x[i] + b * y[i];
  }
}

struct DotArgs {
  const double *x[4];
  const double *y[4];
};

template <int K>
__global__ __launch_bounds__(kBlock) void k_dot(size_t n, DotArgs args, double *__restrict__ partials) {
  __shared__ double lds[K * kWaves];
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0;
  const size_t n2 = n >> 1;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double2 xv = reinterpret_cast<const double2 *>(args.x[k])[i];
      const double2 yv = reinterpret_cast<const double2 *>(args.y[k])[i];
      acc[k] += xv.x * yv.x;
      acc[k] += xv.y * yv.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] += args.x[k][n - 1] * args.y[k][n - 1];
  }
  block_partials_store<K>(acc, lds, partials);
}

template <int K>
__global__ __launch_bounds__(kBlock) void k_reduce_rows_to_slots(const double *__restrict__ partials,
                                                                 int count, double *__restrict__ slots,
                                                                 size_t batch_stride) {
  // (workgroup b: the b-th buffer of a batch, batch_stride doubles apart, into slots[b K .. b K + K))
  __shared__ double lds[K * (kWaves + 1)];
  double out[K];
  reduce_rows<K>(partials + (size_t)blockIdx.x * batch_stride, count, out, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) slots[(size_t)blockIdx.x * K + k] = out[k];
  }
}

// any width (the 15 ... 64 components of the Stiefel kernels with p = 5 ... 8): wave w sums components w, w + 16, ...
// exactly as reduce_rows does (a lane's rows in row order, then the wave reduction)
__global__ __launch_bounds__(kBlock) void k_reduce_rows_to_slots_any(const double *__restrict__ partials, int count, int K,
                                                                     double *__restrict__ slots, size_t batch_stride) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double *base = partials + (size_t)blockIdx.x * batch_stride;
  for (int c = w; c < K; c += kWaves) {
    const double *src = base + (size_t)c * kMaxRows;
    double t[kMaxRows / 64];
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) {
      const int r = lane + 64 * j;
      t[j] = (r < count) ? src[r] : 0.0;
    }
    double v = 0;
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) v += t[j];
    v = wave_reduce_sum(v);
    if (lane == 0) slots[(size_t)blockIdx.x * K + c] = v;
  }
}

int check_same(const mi_vec *a, const mi_vec *b) {
  MI_REQUIRE(a && b, "null vector");
  MI_REQUIRE(a->ctx == b->ctx, "vectors belong to different contexts");
  MI_REQUIRE(a->n == b->n, "vector length mismatch (%zu vs %zu)", a->n, b->n);
  return MI_OK;
}

}  // namespace

namespace mi {

int launch_reduce_rows_to_slots(mi_ctx *ctx, const double *partials, int count, int k, double *slots, int nbatch,
                                size_t batch_stride) {
#define RR(K) \
  hipLaunchKernelGGL(k_reduce_rows_to_slots<K>, dim3(nbatch), dim3(kBlock), 0, ctx->stream, partials, count, slots, \
                     batch_stride)
  switch (k) {
    case 1: RR(1); break;
    case 2: RR(2); break;
    case 3: RR(3); break;
    case 4: RR(4); break;
    case 6: RR(6); break;
    case 9: RR(9); break;
    case 10: RR(10); break;
    case 16: RR(16); break;
    default:
      MI_REQUIRE(k >= 1 && k <= kMaxComps, "unsupported reduction width %d", k);
      hipLaunchKernelGGL(k_reduce_rows_to_slots_any, dim3(nbatch), dim3(kBlock), 0, ctx->stream, partials, count, k, slots,
                         batch_stride);
      break;
  }
#undef RR
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// enqueue k dot products -> ctx->scalars[slot0 .. slot0+k) (device), all-reduced across ranks
int dot_batch_to_slots(mi_ctx *ctx, int k, const double *const *x, const double *const *y, size_t n,
                       int slot0) {
  DotArgs a;
  for (int i = 0; i < 4; ++i) {
    a.x[i] = x[i < k ? i : 0];
    a.y[i] = y[i < k ? i : 0];
  }
  const int grid = grid_for(ctx, n, 4);
  {
    KScope ks(ctx, MI_K_BLAS1);
    switch (k) {
      case 1: hipLaunchKernelGGL(k_dot<1>, dim3(grid), dim3(kBlock), 0, ctx->stream, n, a, ctx->partials_user); break;
      case 2: hipLaunchKernelGGL(k_dot<2>, dim3(grid), dim3(kBlock), 0, ctx->stream, n, a, ctx->partials_user); break;
      case 3: hipLaunchKernelGGL(k_dot<3>, dim3(grid), dim3(kBlock), 0, ctx->stream, n, a, ctx->partials_user); break;
      default: hipLaunchKernelGGL(k_dot<4>, dim3(grid), dim3(kBlock), 0, ctx->stream, n, a, ctx->partials_user); break;
    }
  }
  return reduce_rows_allreduce(ctx, ctx->partials_user, grid, k, ctx->scalars + slot0);
}

// the slots land in the pinned words and the polled flag follows them: no copy engine, no wake-up (stream_wait)
__global__ void k_slots_to_host(double *host, const double *__restrict__ src, int k, unsigned long long *flag,
                                unsigned long long seq) {
  for (int i = threadIdx.x; i < k; i += blockDim.x) host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int read_slots_sync(mi_ctx *ctx, int slot0, int k, double *out) {
  unsigned long long *flag = nullptr;
  double *hdev = nullptr;
  unsigned long long seq = poll_begin(ctx, &flag);
  if (seq && hipHostGetDevicePointer((void **)&hdev, ctx->host_scalars, 0) != hipSuccess) {
    (void)hipGetLastError();
    seq = 0;
  }
  ctx->host_syncs++;
  if (seq) {
    hipLaunchKernelGGL(k_slots_to_host, dim3(1), dim3(64), 0, ctx->stream, hdev, (const double *)(ctx->scalars + slot0),
                       k, flag, seq);
    if (hipGetLastError() != hipSuccess) seq = 0;  // (nothing was enqueued: the copy below does the work)
  }
  if (seq) {
    MI_TRY(poll_finish(ctx, seq, "scalar read-back"));
  } else {
    MI_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + slot0, sizeof(double) * k,
                          hipMemcpyDeviceToHost, ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
  }
  for (int i = 0; i < k; ++i) out[i] = ctx->host_scalars[i];
  return MI_OK;
}
}  // namespace mi

extern "C" {

int mi_vec_create(mi_ctx *ctx, size_t n, mi_vec **out) {
  MI_REQUIRE(ctx && out, "null argument");
  void *p = nullptr;
  MI_TRY(pool_alloc(ctx, n * sizeof(double), &p));
  mi_vec *v = new mi_vec{ctx, n, (double *)p, true, ++ctx->vec_serial, 0, nullptr};
  *out = v;
  return MI_OK;
}

int mi_vec_destroy(mi_vec *v) {
  if (!v) return MI_OK;
  if (v->root) {  // a view: the last one out releases an owner that was destroyed before it
    mi_vec *root = v->root;
    delete v;
    if (--root->views == 0 && root->zombie) {
      if (root->owned) pool_free(root->ctx, root->d);
      delete root;
    }
    return MI_OK;
  }
  if (v->views) {  // live views: keep storage and generation counter until the last of them is destroyed
    v->zombie = true;
    return MI_OK;
  }
  if (v->owned) pool_free(v->ctx, v->d);
  delete v;
  return MI_OK;
}

int mi_vec_view(const mi_vec *base, size_t offset, size_t n, mi_vec **out) {
  MI_REQUIRE(base && out, "null argument");
  MI_REQUIRE(offset + n <= base->n, "view [%zu, %zu) exceeds the base vector (%zu)", offset, offset + n, base->n);
  // an odd offset only costs alignment: gfx950 global 16-byte accesses need 4-byte alignment
  mi_vec *root = base->root ? base->root : const_cast<mi_vec *>(base);
  MI_REQUIRE(!root->zombie, "view of a destroyed vector");
  *out = new mi_vec{base->ctx, n, base->d + offset, false, ++base->ctx->vec_serial, 0, root};
  ++root->views;
  return MI_OK;
}

int mi_vec_touch(mi_vec *v) {
  MI_REQUIRE(v, "null argument");
  touch(v);
  return MI_OK;
}

int mi_vec_len(const mi_vec *v, size_t *n) {
  MI_REQUIRE(v && n, "null argument");
  *n = v->n;
  return MI_OK;
}

int mi_vec_data(const mi_vec *v, void **p) {
  MI_REQUIRE(v && p, "null argument");
  *p = v->d;
  return MI_OK;
}

int mi_vec_upload(mi_vec *v, const double *host, size_t n) {
  MI_REQUIRE(v && (host || n == 0), "null argument");
  MI_REQUIRE(n == v->n, "upload length %zu != vector length %zu", n, v->n);
  touch(v);
  MI_HIP(hipMemcpyAsync(v->d, host, n * sizeof(double), hipMemcpyHostToDevice, v->ctx->stream));
  MI_HIP(hipStreamSynchronize(v->ctx->stream));
  return MI_OK;
}

int mi_vec_download(const mi_vec *v, double *host, size_t n) {
  MI_REQUIRE(v && (host || n == 0), "null argument");
  MI_REQUIRE(n == v->n, "download length %zu != vector length %zu", n, v->n);
  MI_HIP(hipMemcpyAsync(host, v->d, n * sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
  MI_HIP(hipStreamSynchronize(v->ctx->stream));
  return MI_OK;
}

int mi_vec_copy(mi_vec *dst, const mi_vec *src) {
  MI_TRY(check_same(dst, src));
  if (dst->d == src->d) return MI_OK;
  touch(dst);
  MI_HIP(hipMemcpyAsync(dst->d, src->d, src->n * sizeof(double), hipMemcpyDeviceToDevice,
                        dst->ctx->stream));
  return MI_OK;
}

int mi_vec_fill(mi_vec *v, double a) {
  MI_REQUIRE(v, "null vector");
  touch(v);
  if (v->n == 0) return MI_OK;
  KScope ks(v->ctx, MI_K_BLAS1);
  hipLaunchKernelGGL(k_axpby<2>, dim3(grid_for(v->ctx, v->n, 4)), dim3(kBlock), 0, v->ctx->stream, v->n, a,
                     (const double *)nullptr, 0.0, (const double *)nullptr, v->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_vec_scale(mi_vec *v, double a) {
  MI_REQUIRE(v, "null vector");
  touch(v);
  if (v->n == 0) return MI_OK;
  KScope ks(v->ctx, MI_K_BLAS1);
  hipLaunchKernelGGL(k_axpby<1>, dim3(grid_for(v->ctx, v->n, 4)), dim3(kBlock), 0, v->ctx->stream, v->n, a,
                     (const double *)v->d, 0.0, (const double *)nullptr, v->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_vec_scale_to(mi_vec *z, double a, const mi_vec *x) {
  MI_TRY(check_same(z, x));
  touch(z);
  if (z->n == 0) return MI_OK;
  KScope ks(z->ctx, MI_K_BLAS1);
  hipLaunchKernelGGL(k_axpby<1>, dim3(grid_for(z->ctx, z->n, 4)), dim3(kBlock), 0, z->ctx->stream, z->n, a,
                     (const double *)x->d, 0.0, (const double *)nullptr, z->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_vec_div(mi_vec *v, double a) {
  MI_REQUIRE(v, "null vector");
  touch(v);
  if (v->n == 0) return MI_OK;
  KScope ks(v->ctx, MI_K_BLAS1);
  hipLaunchKernelGGL(k_axpby<3>, dim3(grid_for(v->ctx, v->n, 4)), dim3(kBlock), 0, v->ctx->stream, v->n, a,
                     (const double *)v->d, 0.0, (const double *)nullptr, v->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_vec_axpby(mi_vec *z, double a, const mi_vec *x, double b, const mi_vec *y) {
  MI_TRY(check_same(z, x));
  MI_TRY(check_same(z, y));
  touch(z);
  if (z->n == 0) return MI_OK;
  KScope ks(z->ctx, MI_K_BLAS1);
  hipLaunchKernelGGL(k_axpby<0>, dim3(grid_for(z->ctx, z->n, 4)), dim3(kBlock), 0, z->ctx->stream, z->n, a,
                     (const double *)x->d, b, (const double *)y->d, z->d);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

int mi_vec_axpy(mi_vec *y, double a, const mi_vec *x) { return mi_vec_axpby(y, a, x, 1.0, y); }

int mi_vec_dot_batch(mi_ctx *ctx, int k, const mi_vec *const *x, const mi_vec *const *y, double *out) {
  MI_REQUIRE(ctx && x && y && out, "null argument");
  MI_REQUIRE(k >= 1 && k <= 4, "k must be in [1,4], got %d", k);
  const double *xs[4], *ys[4];
  for (int i = 0; i < k; ++i) {
    MI_TRY(check_same(x[i], y[i]));
    MI_REQUIRE(x[i]->n == x[0]->n && x[i]->ctx == ctx, "batched vectors must share length and context");
    xs[i] = x[i]->d;
    ys[i] = y[i]->d;
  }
  MI_TRY(dot_batch_to_slots(ctx, k, xs, ys, x[0]->n, 0));
  ctx->fusion.generic_inner_products += (unsigned long long)k;
  return read_slots_sync(ctx, 0, k, out);
}

int mi_vec_dot(const mi_vec *x, const mi_vec *y, double *out) {
  MI_TRY(check_same(x, y));
  const mi_vec *xs[1] = {x}, *ys[1] = {y};
  return mi_vec_dot_batch(x->ctx, 1, xs, ys, out);
}

}  // extern "C"
