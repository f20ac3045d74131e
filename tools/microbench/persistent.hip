// Is a one-launch, register-resident CG iteration worth building on MI355X?  Measures
//   (1) the cost of a device-wide barrier (256 workgroups, one per CU) followed by the fixed-order
//       re-reduction of 256 partial rows x 3 components that every CG phase boundary needs;
//   (2) the SpMM phase of St(1e6,3) (7-point stencil, sliced-ELL-64) when each lane keeps R rows of the
//       product in registers, at 1024 threads x R=4 and 512 threads x R=8 per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/persistent.hip -o /tmp/persistent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kRows = 512;  // partial-row stride

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned &epoch, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    epoch += 1;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = epoch * nwg;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

template <int K>
__device__ __forceinline__ void reduce_rows(const double *partials, int count, double (&out)[K], double *lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w < K) {
    const double *src = partials + (size_t)w * kRows;
    double t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = lane + 64 * j; t[j] = r < count ? src[r] : 0.0; }
    double v = (t[0] + t[1]) + (t[2] + t[3]);
    v = wave_sum(v);
    if (lane == 0) lds[w] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = lds[k];
  __syncthreads();
}

// (1) barrier + reduction rounds; checks the totals every round
template <int T>
__global__ __launch_bounds__(T) void k_bar(unsigned *ctr, double *partials, int rounds, int with_reduce, int *bad) {
  __shared__ double lds[8];
  unsigned epoch = 0;
  for (int it = 0; it < rounds; ++it) {
    double *buf = partials + (size_t)(it & 1) * 3 * kRows;
    if (threadIdx.x < 3) buf[threadIdx.x * kRows + blockIdx.x] = (double)(it + 1) * (threadIdx.x + 1);
    grid_barrier(ctr, epoch, gridDim.x);
    if (with_reduce) {
      double d[3];
      reduce_rows<3>(buf, gridDim.x, d, lds);
      if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c)
          if (d[c] != (double)(it + 1) * (c + 1) * gridDim.x) atomicAdd(bad, 1);
    }
  }
}


// (1b) barrier and reduction fused: every workgroup publishes {partial, tag = round} as one 16-byte
// write-through store; component wave c of every workgroup polls the 256 tagged partials of component c
// with 16-byte agent-scope loads until all carry this round's tag, then sums them in row order.
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16_agent(d2 *p, d2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ d2 load16_agent(const d2 *p) {
  d2 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int K, int W>
__device__ __forceinline__ void exchange(d2 *buf, unsigned &epoch, const double (&acc)[K], double (&out)[K],
                                         double *lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) lds[k * W + w] = v;
  }
  __syncthreads();
  epoch += 1;
  const double tag = (double)epoch;
  d2 *mine = buf + (size_t)(epoch & 1) * K * kRows;
  if (threadIdx.x < K) {
    double v = 0;
    for (int j = 0; j < W; ++j) v += lds[threadIdx.x * W + j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    d2 pk; pk[0] = v; pk[1] = tag;
    store16_agent(mine + threadIdx.x * kRows + blockIdx.x, pk);
  }
  if (w < K) {
    const d2 *src = mine + (size_t)w * kRows;
    const int count = gridDim.x;
    double t[4];
    bool ok;
    do {
      ok = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = lane + 64 * j;
        if (r < count) {
          const d2 d = load16_agent(src + r);
          t[j] = d[0];
          ok = ok && (d[1] == tag);
        } else t[j] = 0;
      }
    } while (!__all(ok));
    double v = (t[0] + t[1]) + (t[2] + t[3]);
    v = wave_sum(v);
    if (lane == 0) lds[K * W + w] = v;
    if (w == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = lds[K * W + k];
}

template <int T>
__global__ __launch_bounds__(T) void k_bar2(d2 *buf, int rounds, int *bad) {
  __shared__ double lds[3 * (T / 64) + 8];
  unsigned epoch = 0;
  for (int it = 0; it < rounds; ++it) {
    double acc[3];
    for (int c = 0; c < 3; ++c) acc[c] = (threadIdx.x == 5) ? (double)(it + 1) * (c + 1) : 0.0;
    double d[3];
    exchange<3, T / 64>(buf, epoch, acc, d, lds);
    if (threadIdx.x == 0)
      for (int c = 0; c < 3; ++c)
        if (d[c] != (double)(it + 1) * (c + 1) * gridDim.x) atomicAdd(bad, 1);
  }
}

__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
  const unsigned q = nb / 8, r = nb % 8, xcd = b % 8, idx = b / 8;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// (2) SpMM phase with the product held in registers; `iters` phases separated by barriers, V ping-pongs
template <int T, int R, int CH>
__global__ __launch_bounds__(T) void k_spmm(size_t n, size_t nslices, const long long *slice_ptr, const int *col,
                                            const double *val, double *V0, double *V1, unsigned *ctr, int iters,
                                            double *out) {
  constexpr int W = T / 64, P = 3;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t s0 = (nslices * lb) / nb, s1 = (nslices * (lb + 1)) / nb;
  double acc[R][P];
  unsigned epoch = 0;
  double chk = 0;
  for (int it = 0; it < iters; ++it) {
    const double *V = (it & 1) ? V1 : V0;
    double *Vn = (it & 1) ? V0 : V1;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const size_t slice = s0 + w + (size_t)j * W;
#pragma unroll
      for (int c = 0; c < P; ++c) acc[j][c] = 0;
      if (slice < s1) {
        const long long b0 = slice_ptr[slice], b1 = slice_ptr[slice + 1];
        for (long long k = b0; k < b1; k += CH) {
          double a[CH]; size_t ci[CH]; bool ok[CH];
#pragma unroll
          for (int q = 0; q < CH; ++q) {
            ok[q] = k + q < b1;
            const size_t e = (size_t)(ok[q] ? k + q : k) * 64 + lane;
            a[q] = val[e]; ci[q] = (size_t)col[e];
          }
#pragma unroll
          for (int q = 0; q < CH; ++q) {
            const double *src = V + ci[q] * P;
#pragma unroll
            for (int c = 0; c < P; ++c) { const double t = a[q] * src[c]; acc[j][c] += ok[q] ? t : 0.0; }
          }
        }
      }
    }
    // stand-in for the vector phases: the next direction is a scaled copy of the product
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const size_t slice = s0 + w + (size_t)j * W;
      const size_t row = slice * 64 + lane;
      if (slice < s1 && row < n) {
#pragma unroll
        for (int c = 0; c < P; ++c) { Vn[row * P + c] = 0.125 * acc[j][c]; chk += acc[j][c]; }
      }
    }
    grid_barrier(ctr, epoch, nb);
  }
  out[(size_t)blockIdx.x * T + threadIdx.x] = chk;
}

template <class F> float timeit(F f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("%s: %d CUs\n", prop.name, ncu);
  unsigned *ctr; double *partials; int *bad;
  CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&partials, 2 * 3 * kRows * 8)); CK(hipMalloc(&bad, 4));
  CK(hipMemset(bad, 0, 4));
  const int rounds = 2000;
  for (int T : {512, 1024})
    for (int red : {0, 1}) {
      auto run = [&]() {
        CK(hipMemsetAsync(ctr, 0, 4));
        void *args[] = {&ctr, &partials, (void *)&rounds, &red, &bad};
        if (T == 512) CK(hipLaunchCooperativeKernel((void *)k_bar<512>, dim3(ncu), dim3(512), args, 0, 0));
        else CK(hipLaunchCooperativeKernel((void *)k_bar<1024>, dim3(ncu), dim3(1024), args, 0, 0));
      };
      const float ms = timeit(run);
      int hbad; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      printf("barrier T=%d reduce=%d: %.2f us/round (bad=%d)\n", T, red, ms * 1e3 / rounds, hbad);
    }


  {
    d2 *buf; CK(hipMalloc(&buf, 2 * 3 * kRows * 16));
    for (int T : {512, 1024}) {
      auto run = [&]() {
        CK(hipMemsetAsync(buf, 0, 2 * 3 * kRows * 16));
        void *args[] = {&buf, (void *)&rounds, &bad};
        if (T == 512) CK(hipLaunchCooperativeKernel((void *)k_bar2<512>, dim3(ncu), dim3(512), args, 0, 0));
        else CK(hipLaunchCooperativeKernel((void *)k_bar2<1024>, dim3(ncu), dim3(1024), args, 0, 0));
      };
      const float ms = timeit(run);
      int hbad; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      printf("tagged exchange T=%d: %.2f us/round (bad=%d)\n", T, ms * 1e3 / rounds, hbad);
    }
  }

  // 7-point stencil on 100^3 in sliced-ELL-64, width 7
  const int nx = 100; const size_t n = (size_t)nx * nx * nx, nsl = (n + 63) / 64;
  std::vector<long long> sp(nsl + 1); for (size_t s = 0; s <= nsl; ++s) sp[s] = 7 * (long long)s;
  std::vector<int> col(nsl * 7 * 64); std::vector<double> val(nsl * 7 * 64);
  for (size_t s = 0; s < nsl; ++s)
    for (int l = 0; l < 64; ++l) {
      const size_t row = s * 64 + l;
      const long long off[7] = {0, -1, 1, -nx, nx, -(long long)nx * nx, (long long)nx * nx};
      for (int k = 0; k < 7; ++k) {
        const long long c = (long long)row + off[k];
        const bool ok = row < n && c >= 0 && c < (long long)n;
        col[(s * 7 + k) * 64 + l] = ok ? (int)c : (int)(row < n ? row : 0);
        val[(s * 7 + k) * 64 + l] = ok ? (k == 0 ? 6.0 : -1.0) : 0.0;
      }
    }
  long long *dsp; int *dcol; double *dval, *V0, *V1, *out;
  CK(hipMalloc(&dsp, sp.size() * 8)); CK(hipMalloc(&dcol, col.size() * 4)); CK(hipMalloc(&dval, val.size() * 8));
  CK(hipMalloc(&V0, n * 3 * 8)); CK(hipMalloc(&V1, n * 3 * 8)); CK(hipMalloc(&out, (size_t)ncu * 1024 * 8));
  CK(hipMemcpy(dsp, sp.data(), sp.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dval, val.data(), val.size() * 8, hipMemcpyHostToDevice));
  std::vector<double> hv(n * 3); for (size_t i = 0; i < hv.size(); ++i) hv[i] = 1.0 + (i % 7) * 0.01;
  CK(hipMemcpy(V0, hv.data(), hv.size() * 8, hipMemcpyHostToDevice));
  const int iters = 200;
  size_t nn = n, ns = nsl;
  void *args[] = {&nn, &ns, &dsp, &dcol, &dval, &V0, &V1, &ctr, (void *)&iters, &out};
#define RUN(T, R, CH)                                                                                  \
  {                                                                                                    \
    auto run = [&]() {                                                                                 \
      CK(hipMemsetAsync(ctr, 0, 4));                                                                   \
      CK(hipLaunchCooperativeKernel((void *)k_spmm<T, R, CH>, dim3(ncu), dim3(T), args, 0, 0));        \
    };                                                                                                 \
    const float ms = timeit(run);                                                                      \
    printf("spmm phase T=%d R=%d CH=%d: %.2f us/iter  (matrix 84 MB + V 24 MB gather + 24 MB write)\n", T, R, CH, \
           ms * 1e3 / iters);                                                                          \
  }
  RUN(1024, 4, 4)
  RUN(1024, 4, 7)
  RUN(512, 8, 4)
  RUN(512, 8, 7)
  RUN(256, 16, 7)
  return 0;
}
