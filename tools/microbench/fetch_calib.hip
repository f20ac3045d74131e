// fetch_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the
// access widths the Stiefel Hessian kernel mixes: 16 B, 8 B and 4 B per lane streams, 24-byte
// row gathers (row-major n x 3 doubles) and a mix shaped like one Hessian pass (7 matrix words + 7 row gathers +
// 2 own-row reads + 1 row write per row of a 100x100x100 grid).  Every kernel touches each byte of its inputs once;
// the byte counts are printed so that tools/fetch_calib.py can divide the counters by them.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

__global__ __launch_bounds__(256) void calib_stream16(const double2 *__restrict__ a, size_t n, double *sink) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 1.2345e300) *sink = s;
}
__global__ __launch_bounds__(256) void calib_stream8(const double *__restrict__ a, size_t n, double *sink) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 1.2345e300) *sink = s;
}
__global__ __launch_bounds__(256) void calib_stream4(const uint32_t *__restrict__ a, size_t n, double *sink) {
  uint32_t s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 0x12345678u) *sink = s;
}
// lane r reads the 24 bytes of row r
__global__ __launch_bounds__(256) void calib_rows24(const double *__restrict__ a, size_t rows, double *sink) {
  double s = 0;
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256)
    s += a[3 * r] + a[3 * r + 1] + a[3 * r + 2];
  if (s == 1.2345e300) *sink = s;
}
// write-only: 24 bytes per lane (the Hessian's output rows) and 16 bytes per lane
__global__ __launch_bounds__(256) void calib_write24(double *__restrict__ a, size_t rows) {
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    a[3 * r] = 1.0;
    a[3 * r + 1] = 2.0;
    a[3 * r + 2] = 3.0;
  }
}
__global__ __launch_bounds__(256) void calib_write16(double2 *__restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = double2{1.0, 2.0};
}
// One Hessian-shaped pass over a nx x ny x nz grid: per row 7 four-byte words (coalesced, slice-major like sliced
// ELL: word j of the 64 rows of a slice is contiguous), the 24-byte rows of V at the 7 stencil neighbours, the own
// rows of two more fields, one 24-byte row written.
__global__ __launch_bounds__(256) void calib_hess_mix(const uint32_t *__restrict__ words, const double *__restrict__ V,
                                                      const double *__restrict__ X, const double *__restrict__ Y,
                                                      double *__restrict__ W, int nx, int ny, int nz) {
  const size_t n = (size_t)nx * ny * nz;
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (size_t)gridDim.x * 256) {
    const size_t slice = r >> 6, lane = r & 63;
    double acc[3] = {0, 0, 0};
    const long long off[7] = {0, -1, 1, -(long long)nx, nx, -(long long)nx * ny, (long long)nx * ny};
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const uint32_t w = words[(slice * 7 + j) * 64 + lane];
      long long c = (long long)r + off[j];
      if (c < 0 || c >= (long long)n) c = r;
      const double a = (double)(w & 255u);
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[k] += a * V[3 * c + k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) W[3 * r + k] = acc[k] - X[3 * r + k] * Y[3 * r + k];
  }
}

int main() {
  const int nx = 100, ny = 100, nz = 100;
  const size_t n = (size_t)nx * ny * nz, big = 64u << 20;  // 64 Mi elements for the pure streams
  double *sink;
  CK(hipMalloc(&sink, 8));
  void *buf;
  CK(hipMalloc(&buf, big * 16));
  CK(hipMemset(buf, 0, big * 16));
  double *V, *X, *Y, *W;
  uint32_t *words;
  CK(hipMalloc(&V, n * 24));
  CK(hipMalloc(&X, n * 24));
  CK(hipMalloc(&Y, n * 24));
  CK(hipMalloc(&W, n * 24));
  CK(hipMalloc(&words, ((n + 63) / 64) * 7 * 64 * 4));
  CK(hipMemset(V, 0, n * 24));
  CK(hipMemset(X, 0, n * 24));
  CK(hipMemset(Y, 0, n * 24));
  CK(hipMemset(words, 0, ((n + 63) / 64) * 7 * 64 * 4));
  const int grid = 2048, reps = 5;
  printf("{\"calib_stream16\": {\"read\": %zu, \"write\": 0},\n", big * 16);
  printf(" \"calib_stream8\": {\"read\": %zu, \"write\": 0},\n", big * 8);
  printf(" \"calib_stream4\": {\"read\": %zu, \"write\": 0},\n", big * 4);
  printf(" \"calib_rows24\": {\"read\": %zu, \"write\": 0},\n", (big * 16 / 24) * 24);
  printf(" \"calib_write24\": {\"read\": 0, \"write\": %zu},\n", (big * 16 / 24) * 24);
  printf(" \"calib_write16\": {\"read\": 0, \"write\": %zu},\n", big * 16);
  printf(" \"calib_hess_mix\": {\"read\": %zu, \"write\": %zu}}\n", ((n + 63) / 64) * 7 * 64 * 4 + 3 * n * 24, n * 24);
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(calib_stream16, dim3(grid), dim3(256), 0, 0, (const double2 *)buf, big, sink);
    hipLaunchKernelGGL(calib_stream8, dim3(grid), dim3(256), 0, 0, (const double *)buf, big, sink);
    hipLaunchKernelGGL(calib_stream4, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, big, sink);
    hipLaunchKernelGGL(calib_rows24, dim3(grid), dim3(256), 0, 0, (const double *)buf, big * 16 / 24, sink);
    hipLaunchKernelGGL(calib_write24, dim3(grid), dim3(256), 0, 0, (double *)buf, big * 16 / 24);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (double2 *)buf, big);
    hipLaunchKernelGGL(calib_hess_mix, dim3(grid), dim3(256), 0, 0, (const uint32_t *)words, (const double *)V,
                       (const double *)X, (const double *)Y, W, nx, ny, nz);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
