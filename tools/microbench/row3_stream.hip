// row3_stream.hip -- does it matter HOW a kernel touches row-major n x 3 fp64 fields (r04)?
//
// The one-pass Hessian kernel of cfg2 (k_st_hess_fused) reads the X and Y rows of its own row and writes its output row
// as three 8-byte accesses per lane, 24 bytes apart across the lanes: a wave instruction covers 1536 bytes at one third
// density and three instructions complete the lines.  The CG kernels walk the same fields as flat arrays of double2.
// This program streams two n x 3 fields in and one out (cfg2's n = 10^6: 24 MB each, Infinity-Cache resident), with a
// trivial combination, in both access forms:
//   rows : lane = row, three 8-byte loads / stores per field and lane (what the Hessian kernel's epilogue does)
//   flat : lane = double2 element of the flat array (what the CG kernels do)
//   lds  : loaded and stored flat, turned into rows through LDS in between (what a fix would look like)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/row3_stream.hip -o tools/microbench/row3_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_rows(size_t n, const double *__restrict__ A, const double *__restrict__ B,
                                              double *__restrict__ O) {
  for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (size_t)gridDim.x * 256) {
    double a[3], b[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] = A[r * 3 + c]; b[c] = B[r * 3 + c]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) O[r * 3 + c] = a[c] + 2.0 * b[(c + 1) % 3];
  }
}
typedef double double2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_flat(size_t n2, const double2v *__restrict__ A, const double2v *__restrict__ B,
                                              double2v *__restrict__ O) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const double2v a = A[i], b = B[i];
    O[i] = a + 2.0 * b;
  }
}
// a wave's 64 rows = 192 doubles = 96 double2: loaded flat (lanes 0..63, then 0..31), through LDS into rows, combined per
// row, back through LDS, stored flat
__global__ __launch_bounds__(256) void k_lds(size_t n, const double *__restrict__ A, const double *__restrict__ B,
                                             double *__restrict__ O) {
  __shared__ double la[4][192], lb[4][192], lo[4][192];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t nslice = n / 64;
  for (size_t s = (size_t)blockIdx.x * 4 + w; s < nslice; s += (size_t)gridDim.x * 4) {
    const double2v *a2 = reinterpret_cast<const double2v *>(A + s * 192), *b2 = reinterpret_cast<const double2v *>(B + s * 192);
    const double2v x0 = a2[lane], y0 = b2[lane];
    double2v x1 = {0, 0}, y1 = {0, 0};
    if (lane < 32) { x1 = a2[64 + lane]; y1 = b2[64 + lane]; }
    reinterpret_cast<double2v *>(la[w])[lane] = x0;
    reinterpret_cast<double2v *>(lb[w])[lane] = y0;
    if (lane < 32) { reinterpret_cast<double2v *>(la[w])[64 + lane] = x1; reinterpret_cast<double2v *>(lb[w])[64 + lane] = y1; }
    double a[3], b[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { a[c] = la[w][lane * 3 + c]; b[c] = lb[w][lane * 3 + c]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) lo[w][lane * 3 + c] = a[c] + 2.0 * b[(c + 1) % 3];
    double2v *o2 = reinterpret_cast<double2v *>(O + s * 192);
    o2[lane] = reinterpret_cast<double2v *>(lo[w])[lane];
    if (lane < 32) o2[64 + lane] = reinterpret_cast<double2v *>(lo[w])[64 + lane];
  }
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 1000000;
  double *A, *B, *O;
  CK(hipMalloc(&A, n * 24));
  CK(hipMalloc(&B, n * 24));
  CK(hipMalloc(&O, n * 24));
  CK(hipMemset(A, 0, n * 24));
  CK(hipMemset(B, 0, n * 24));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double gb = (double)n * 72 / 1e9;
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 20; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %7.2f us  %5.2f TB/s\n", name, 1e3 * ms / reps, gb / (ms / reps));
  };
  for (int wg : {512, 1024, 2048}) {
    printf("n = %zu, %d workgroups\n", n, wg);
    run("  rows (3 x 8 B, stride 24)", [&] { hipLaunchKernelGGL(k_rows, dim3(wg), dim3(256), 0, 0, n, A, B, O); });
    run("  flat double2", [&] { hipLaunchKernelGGL(k_flat, dim3(wg), dim3(256), 0, 0, n * 3 / 2, (const double2v *)A, (const double2v *)B, (double2v *)O); });
    run("  flat + LDS transposition", [&] { hipLaunchKernelGGL(k_lds, dim3(wg), dim3(256), 0, 0, n, A, B, O); });
  }
  return 0;
}
