// Does data written by one kernel survive in the writer XCD's L2 for the NEXT kernel on the same stream?
// Kernel W writes a vector (each workgroup a contiguous chunk, workgroup -> XCD round robin as the hardware does);
// kernel R reads it back either with the SAME chunk -> workgroup map (a reader sits on the XCD that wrote its chunk)
// or with the map rotated by one workgroup (every chunk is read from another XCD).  Sizes from 8 MB (fits the 8 x 4 MB
// of L2 easily) to 64 MB.  If R(same) is not faster than R(rotated), kernel boundaries leave nothing in L2 and an
// XCD-aligned producer/consumer mapping of the CG kernels cannot pay.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/l2_carry.hip -o /tmp/l2_carry
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void kW(double2 *v, size_t per, double a) {
  double2 *p = v + (size_t)blockIdx.x * per;
  for (size_t i = threadIdx.x; i < per; i += 1024) p[i] = make_double2(a + i, a - i);
}
__global__ __launch_bounds__(1024) void kR(const double2 *v, size_t per, int rot, double *out) {
  const unsigned b = (blockIdx.x + rot) % gridDim.x;
  const double2 *p = v + (size_t)b * per;
  double s = 0;
  for (size_t i = threadIdx.x; i < per; i += 1024) { const double2 x = p[i]; s += x.x + x.y; }
  if (s == 1.2345) out[0] = s;
}
int main() {
  const int grid = 512, reps = 50;
  double *out; hipMalloc(&out, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t mb : {8, 16, 24, 32, 64}) {
    const size_t n2 = mb * (1u << 20) / 16, per = n2 / grid;
    double2 *v; hipMalloc(&v, per * grid * 16);
    for (int rot : {0, 1, 8}) {   // 8: rotated by a whole round of XCDs = same XCD, other CU
      float tot = 0;
      for (int r = 0; r < reps + 3; ++r) {
        hipLaunchKernelGGL(kW, dim3(grid), dim3(1024), 0, 0, v, per, (double)r);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kR, dim3(grid), dim3(1024), 0, 0, v, per, rot, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) tot += ms;
      }
      printf("{\"MB\": %zu, \"rot\": %d, \"read_us\": %.2f, \"GBps\": %.0f}\n", mb, rot, 1e3 * tot / reps,
             mb * 1.048576e6 / (1e3 * tot / reps) / 1e3);
    }
    hipFree(v);
  }
  return 0;
}
