// f64 peak probe: v_mfma_f64_16x16x4_f64 vs v_mfma_f64_4x4x4_4b_f64 vs v_fma_f64, no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/f64_peak.hip -o /tmp/f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 4096, NACC = 8;
__global__ __launch_bounds__(256) void k_mfma16(double *out, double a, double b) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double x = a + threadIdx.x, y = b;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma4(double *out, double a, double b) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double x = a + threadIdx.x, y = b;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double *out, double a, double b) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double x = a + threadIdx.x * 1e-9, y = b;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], x, y);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  double *out; hipMalloc(&out, 256 * 2048 * 8 * 4);
  for (int wgs_per_cu : {1, 2, 4}) {
    const int grid = 256 * wgs_per_cu;
    float t16 = timeit([&] { hipLaunchKernelGGL(k_mfma16, dim3(grid), dim3(256), 0, 0, out, 1.0, 1e-9); });
    float t4 = timeit([&] { hipLaunchKernelGGL(k_mfma4, dim3(grid), dim3(256), 0, 0, out, 1.0, 1e-9); });
    float tf = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, 1.0, 1e-9); });
    const double waves = grid * 4.0;
    printf("wgs/cu %d: mfma16x16x4 %.1f TF/s (%.0f ns/instr/SIMD-wave) | mfma4x4x4 %.1f TF/s | v_fma_f64 %.1f TF/s\n",
           wgs_per_cu, waves * ITERS * NACC * 2048.0 / (t16 * 1e-3) / 1e12,
           t16 * 1e6 / (ITERS * NACC * wgs_per_cu), waves * ITERS * NACC * 512.0 / (t4 * 1e-3) / 1e12,
           waves * ITERS * 16 * 128.0 / (tf * 1e-3) / 1e12);
  }
  return 0;
}
