// sync_latency.hip -- what a host wait for the stream costs on this stack, and whether polling a flag in pinned memory
// (written by the last kernel itself, or by hipStreamWriteValue64 behind it) is cheaper (r04).  LOBPCG waits twice per
// iteration (Gram matrices, residual norms), the TNT loop once per outer iteration.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/sync_latency.hip -o tools/microbench/sync_latency
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_work(double *p, size_t n, int spin) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a = i < n ? p[i] : 0;
  for (int k = 0; k < spin; ++k) a = a * 1.0000001 + 1e-9;
  if (i < n) p[i] = a;
}
__global__ void k_work_flag(double *p, size_t n, int spin, volatile unsigned long long *flag, unsigned long long v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double a = i < n ? p[i] : 0;
  for (int k = 0; k < spin; ++k) a = a * 1.0000001 + 1e-9;
  if (i < n) p[i] = a;
  if (i == 0) {  // (one workgroup only in this test: the kernel's last act)
    __threadfence_system();
    *flag = v;
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *d;
  CK(hipMalloc(&d, 1 << 20));
  unsigned long long *flag;
  CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
  unsigned long long *dflag = nullptr;
  CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
  *flag = 0;
  const int spin = 20000;  // ~ tens of microseconds of device time, one workgroup
  for (int mode = 0; mode < 3; ++mode) {
    double tot = 0, dev = 0;
    const int reps = 200;
    for (int r = 0; r < reps + 20; ++r) {
      const unsigned long long v = (unsigned long long)(mode * 100000 + r + 1);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      const double t0 = now_us();
      CK(hipEventRecord(e0, st));
      if (mode == 0) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, d, (size_t)256, spin);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
      } else if (mode == 1) {
        hipLaunchKernelGGL(k_work_flag, dim3(1), dim3(256), 0, st, d, (size_t)256, spin, dflag, v);
        CK(hipEventRecord(e1, st));
        while (*(volatile unsigned long long *)flag != v) __builtin_ia32_pause();
      } else {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, st, d, (size_t)256, spin);
        CK(hipEventRecord(e1, st));
        hipError_t e = hipStreamWriteValue64(st, dflag, v, 0);
        if (e != hipSuccess) { printf("hipStreamWriteValue64: %s\n", hipGetErrorString(e)); return 0; }
        while (*(volatile unsigned long long *)flag != v) __builtin_ia32_pause();
      }
      const double t1 = now_us();
      CK(hipStreamSynchronize(st));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 20) { tot += t1 - t0; dev += 1e3 * ms; }
      CK(hipEventDestroy(e0));
      CK(hipEventDestroy(e1));
    }
    const char *names[3] = {"hipStreamSynchronize", "flag stored by the kernel, host polls", "hipStreamWriteValue64, host polls"};
    printf("%-40s host %7.1f us per round, kernel (events) %6.1f us, overhead %6.1f us\n", names[mode], tot / reps,
           dev / reps, (tot - dev) / reps);
  }
  return 0;
}
