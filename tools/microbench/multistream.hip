// How fast can MI355X read a column-major m x k fp64 panel when every workgroup walks ALL k columns
// (the access shape of Gram / panel-update kernels), versus a flat stream of the same bytes?
//   A: lane = row, 8 B/lane   (512 B contiguous per column per wave instruction)
//   B: lane = row pair, 16 B/lane (1 KB contiguous per column per wave instruction)
//   C: lane = (column l&15, 4-row group l>>4), 32 B/lane (16 x 128 B lines per instruction; direct-MFMA feed)
//   F: flat: the whole panel as one 1-D array, 16 B/lane grid-stride
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/multistream.hip -o /tmp/multistream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <initializer_list>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void kA(const double *P, size_t m, int k, double *out) {
  double s = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < m; r += stride) {
    int c = 0;
    for (; c + 4 <= k; c += 4) {
      const double a = P[(size_t)c * m + r], b = P[(size_t)(c + 1) * m + r], d = P[(size_t)(c + 2) * m + r], e = P[(size_t)(c + 3) * m + r];
      s += a + b + d + e;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kB(const double *P, size_t m, int k, double *out) {
  double s = 0;
  const size_t stride = (size_t)gridDim.x * 512;
  for (size_t r = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x); r < m; r += stride) {
    int c = 0;
    for (; c + 4 <= k; c += 4) {
      const double2 a = *(const double2 *)(P + (size_t)c * m + r), b = *(const double2 *)(P + (size_t)(c + 1) * m + r),
                    d = *(const double2 *)(P + (size_t)(c + 2) * m + r), e = *(const double2 *)(P + (size_t)(c + 3) * m + r);
      s += a.x + a.y + b.x + b.y + d.x + d.y + e.x + e.y;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int T>
__global__ __launch_bounds__(256) void kC(const double *P, size_t m, int k, double *out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), band = 16 * (size_t)gridDim.x * 4;
  double s = 0;
  for (size_t r0 = wave * 16; r0 + 16 <= m; r0 += band) {
    d4 v[T];
#pragma unroll
    for (int t = 0; t < T; ++t) v[t] = *(const d4 *)(P + (size_t)(16 * t + (lane & 15)) * m + r0 + 4 * (lane >> 4));
#pragma unroll
    for (int t = 0; t < T; ++t) s += v[t][0] + v[t][1] + v[t][2] + v[t][3];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kF(const double *P, size_t n, double *out) {
  double s = 0;
  const size_t stride = (size_t)gridDim.x * 512;
  for (size_t i = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x); i + 1 < n; i += stride) {
    const double2 a = *(const double2 *)(P + i);
    s += a.x + a.y;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const size_t m = 2000376; const int kmax = 160;
  double *P, *out; (void)hipMalloc(&P, m * kmax * 8); (void)hipMalloc(&out, 8192 * 256 * 8); (void)hipMemset(P, 0, m * kmax * 8);
  for (int k : {16, 48, 80, 160}) {
    const double gb = 8.0 * m * k / 1e9;
    for (int grid : {1024, 2048, 4096}) {
      float a = timeit([&] { hipLaunchKernelGGL(kA, dim3(grid), dim3(256), 0, 0, P, m, k, out); });
      float b = timeit([&] { hipLaunchKernelGGL(kB, dim3(grid), dim3(256), 0, 0, P, m, k, out); });
      float f = timeit([&] { hipLaunchKernelGGL(kF, dim3(grid), dim3(256), 0, 0, P, m * (size_t)k, out); });
      float c = 0;
      const int gc = grid / 4;  // kC: few fat waves like the gram kernel
      if (k == 16) c = timeit([&] { hipLaunchKernelGGL(kC<1>, dim3(gc), dim3(256), 0, 0, P, m, k, out); });
      if (k == 48) c = timeit([&] { hipLaunchKernelGGL(kC<3>, dim3(gc), dim3(256), 0, 0, P, m, k, out); });
      if (k == 80) c = timeit([&] { hipLaunchKernelGGL(kC<5>, dim3(gc), dim3(256), 0, 0, P, m, k, out); });
      if (k == 160) c = timeit([&] { hipLaunchKernelGGL(kC<10>, dim3(gc), dim3(256), 0, 0, P, m, k, out); });
      printf("k=%3d grid=%4d: A(8B) %.0f  B(16B) %.0f  C(32B,16col; grid/4) %.0f  flat %.0f  GB/s\n", k, grid, gb / a * 1e3, gb / b * 1e3,
             gb / c * 1e3, gb / f * 1e3);
    }
  }
  return 0;
}
