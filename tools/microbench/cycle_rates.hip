// cycle_rates.hip -- the byte pattern of one fused STPCG iteration (cfg2: six 24 MB fields) as three BARE streaming
// kernels in a cycle, each timed by the profiler: does a plain kernel keep its stand-alone rate (stream_rates.hip) when
// its operands were last written by a DIFFERENT kernel two launches ago?  K1 (the Hessian pass's mix, without the
// matrix): hp = p + x + y (3 reads : 1 write); K2 (k_cg_update's): r = r + a hp (2 : 1); K3 (k_cg_pupdate's):
// s = s + a p, p = b p - r (3 : 2).  Run under rocprofv3 --kernel-trace --stats, or stand-alone for wall time per cycle.
// Usage: cycle_rates [MB per field = 24] [cycles = 300]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k1_hess_mix(size_t n2, const v2d *p, const v2d *x, const v2d *y, v2d *hp) {
  for (size_t i = blockIdx.x * 1024ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 1024) hp[i] = p[i] + x[i] + y[i];
}
__global__ __launch_bounds__(1024) void k2_update_mix(size_t n2, double a, const v2d *hp, v2d *r) {
  for (size_t i = blockIdx.x * 1024ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 1024) r[i] = r[i] + a * hp[i];
}
__global__ __launch_bounds__(1024) void k3_pupdate_mix(size_t n2, double a, double b, const v2d *r, v2d *p, v2d *s) {
  for (size_t i = blockIdx.x * 1024ull + threadIdx.x; i < n2; i += (size_t)gridDim.x * 1024) {
    const v2d pi = p[i];
    s[i] = s[i] + a * pi;
    p[i] = b * pi - r[i];
  }
}
int main(int argc, char **argv) {
  const double mb = argc > 1 ? atof(argv[1]) : 24.0;
  const int cycles = argc > 2 ? atoi(argv[2]) : 300;
  const size_t n2 = (size_t)(mb * 1e6 / 16);
  v2d *f[6];
  for (auto &q : f) { CK(hipMalloc(&q, n2 * 16)); CK(hipMemset(q, 0, n2 * 16)); }
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int grid = 2 * pr.multiProcessorCount;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int c = 0; c < cycles; ++c) {
      hipLaunchKernelGGL(k1_hess_mix, dim3(grid), dim3(1024), 0, 0, n2, f[0], f[1], f[2], f[3]);
      hipLaunchKernelGGL(k2_update_mix, dim3(grid), dim3(1024), 0, 0, n2, 1e-3, f[3], f[4]);
      hipLaunchKernelGGL(k3_pupdate_mix, dim3(grid), dim3(1024), 0, 0, n2, 1e-3, 0.5, f[4], f[0], f[5]);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("{\"mb_per_field\": %.0f, \"footprint_mb\": %.0f, \"moved_mb_per_cycle\": %.0f, \"us_per_cycle\": %.2f, \"GBps\": %.0f}\n", mb, 6 * mb,
                    12 * mb, 1e3 * ms / cycles, 12 * mb * 1e6 / (1e3 * ms / cycles) / 1e3);
  }
  return 0;
}
