// Does the 5x5-tile accumulate pattern of k_gram_direct (25 accumulators = 200 AGPRs, operands from 10
// double4 registers) run the f64 MFMA pipe at the same rate as the 8-accumulator peak probe?
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/f64_mma_tiles.hip -o /tmp/f64_mma_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 512;
template <int T, bool SAME>
__global__ __launch_bounds__(256) void k_tiles(double *out, const double *in) {
  d4 acc[T][T];
  for (int a = 0; a < T; ++a) for (int b = 0; b < T; ++b) acc[a][b] = (d4){0, 0, 0, 0};
  d4 sa[T], sb[T];
  for (int t = 0; t < T; ++t) { sa[t] = *(const d4 *)(in + 4 * (threadIdx.x + 256 * t)); sb[t] = *(const d4 *)(in + 4 * (threadIdx.x + 256 * (t + T))); }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = SAME ? a : 0; b < T; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[a][j], SAME ? sa[b][j] : sb[b][j], acc[a][b], 0, 0, 0);
    // keep the loop from being collapsed
    asm volatile("" ::: "memory");
  }
  double s = 0;
  for (int a = 0; a < T; ++a) for (int b = 0; b < T; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  double *out, *in; (void)hipMalloc(&out, 256 * 256 * 8); (void)hipMalloc(&in, 4 * 256 * 12 * 8); (void)hipMemset(in, 0, 4 * 256 * 12 * 8);
  const int grid = 256;
  auto rep = [&](const char *name, float ms, double mfmas_per_iter) {
    printf("%s: %.1f TF/s, %.1f ns per MFMA per SIMD\n", name, grid * 4.0 * ITERS * mfmas_per_iter * 2048.0 / (ms * 1e-3) / 1e12,
           ms * 1e6 / (ITERS * mfmas_per_iter));
  };
  rep("T=5 full (100 MFMA/step)", timeit([&] { hipLaunchKernelGGL((k_tiles<5, false>), dim3(grid), dim3(256), 0, 0, out, in); }), 100);
  rep("T=5 sym  ( 60 MFMA/step)", timeit([&] { hipLaunchKernelGGL((k_tiles<5, true>), dim3(grid), dim3(256), 0, 0, out, in); }), 60);
  rep("T=4 full ( 64 MFMA/step)", timeit([&] { hipLaunchKernelGGL((k_tiles<4, false>), dim3(grid), dim3(256), 0, 0, out, in); }), 64);
  rep("T=2 full ( 16 MFMA/step)", timeit([&] { hipLaunchKernelGGL((k_tiles<2, false>), dim3(grid), dim3(256), 0, 0, out, in); }), 16);
  return 0;
}
