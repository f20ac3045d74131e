// grid_barrier_probe.hip -- what a PERSISTENT inner solve would pay per grid-wide step on gfx950 (DESIGN.md "what comes
// next": the solver state r, s of St(1e6,3) is 48 MB -- it fits the chip's 128 MB of vector registers, and a persistent
// kernel that kept it there would move ~100 MB per iteration instead of 320 MB; what it needs instead of the three
// kernel boundaries of an iteration is three GRID BARRIERS with agent-scope release / acquire, because the direction p
// is gathered across XCDs whose L2s are not coherent with each other inside a kernel).
//
// Measures, for a grid of G workgroups x 256 threads that is resident all at once (G <= 2 per CU):
//   mode 0   two barriers per round, nothing else                      -> cost of a barrier
//   mode 1   every workgroup writes its slice of a field (bytes/G per round), barrier, reads the slice of the workgroup
//            half a grid away (another XCD) and checks every value, barrier -> barrier + visibility + the traffic itself
//   mode 2, 3  the same two with a barrier of per-workgroup flags (no contended counter)
// Usage: grid_barrier_probe [G=512] [rounds=2000] [MB=24]
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/grid_barrier_probe.hip -o tools/microbench/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned &gen, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    gen += 1;
    // release: this workgroup's writes (the leader's fence covers the CU's L1 write-through and this XCD's L2 write-back)
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = gen * nwg;
    // (bounded: a grid that is not resident all at once must end in an error, not hang the box)
    for (unsigned spin = 0; __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) {
      __builtin_amdgcn_s_sleep(1);
      if (spin > (1u << 22)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // acquire: drop stale lines of other XCDs' writes
  }
  __syncthreads();
}

// the same barrier without a contended counter: every workgroup publishes its generation in a flag of its own (one
// 64-byte line each), every workgroup polls all the flags (thread t: flags t, t + 256, ...)
__device__ __forceinline__ void grid_barrier_flags(unsigned *flags, unsigned *bar, unsigned &gen, unsigned nwg) {
  __syncthreads();
  gen += 1;
  if (threadIdx.x == 0) __hip_atomic_store(flags + 16 * blockIdx.x, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned w = threadIdx.x; w < nwg; w += 256)
    for (unsigned spin = 0; __hip_atomic_load(flags + 16 * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen; ++spin) {
      __builtin_amdgcn_s_sleep(1);
      if (spin > (1u << 22)) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_probe(unsigned *bar, unsigned *flags, double *field, unsigned long long *errors, int rounds, int mode,
                                                size_t per_wg) {
  unsigned gen = 0;
  const unsigned nwg = gridDim.x;
  const size_t mine = (size_t)blockIdx.x * per_wg, theirs = (size_t)((blockIdx.x + nwg / 2) % nwg) * per_wg;
  unsigned long long bad = 0;
  for (int it = 0; it < rounds; ++it) {
    if (mode & 1)
      for (size_t i = threadIdx.x; i < per_wg; i += 256) field[mine + i] = (double)it + 1e-7 * (double)(mine + i);
    if (mode & 2) grid_barrier_flags(flags, bar, gen, nwg); else grid_barrier(bar, gen, nwg);
    if (mode & 1)
      for (size_t i = threadIdx.x; i < per_wg; i += 256) {
        // (a plain load: the acquire fence of the barrier is what makes it see the other XCD's write)
        const double v = field[theirs + i];
        bad += v != (double)it + 1e-7 * (double)(theirs + i);
      }
    if (mode & 2) grid_barrier_flags(flags, bar, gen, nwg); else grid_barrier(bar, gen, nwg);
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 512, rounds = argc > 2 ? atoi(argv[2]) : 2000;
  const double mb = argc > 3 ? atof(argv[3]) : 24.0;
  const size_t per_wg = (size_t)(mb * 1e6 / 8 / G);
  unsigned *bar, *flags;
  double *field;
  unsigned long long *err, herr = 0;
  CK(hipMalloc(&bar, 8)); CK(hipMalloc(&flags, 64 * (size_t)G)); CK(hipMalloc(&field, per_wg * G * 8)); CK(hipMalloc(&err, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_probe, 256, 0));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  printf("device %s: %d CUs x %d resident workgroups of this kernel; grid %d\n", pr.name, pr.multiProcessorCount, nb, G);
  if (G > nb * pr.multiProcessorCount) { fprintf(stderr, "grid not resident at once\n"); return 1; }
  for (int mode = 0; mode < 4; ++mode) {   // bit 0: traffic; bit 1: flag barrier instead of the counter
    for (int rep = 0; rep < 2; ++rep) {   // first repetition: warm-up
      CK(hipMemset(bar, 0, 8)); CK(hipMemset(err, 0, 8)); CK(hipMemset(flags, 0, 64 * (size_t)G));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_probe, dim3(G), dim3(256), 0, 0, bar, flags, field, err, rounds, mode, per_wg);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(&herr, err, 8, hipMemcpyDeviceToHost));
      unsigned hb[2]; CK(hipMemcpy(hb, bar, 8, hipMemcpyDeviceToHost));
      if (hb[1]) { fprintf(stderr, "a barrier timed out (grid not resident?)\n"); return 2; }
      if (rep)
        printf("{\"barrier\": \"%s\", \"mode\": %d, \"grid\": %d, \"rounds\": %d, \"us_per_round\": %.3f, \"barriers_per_round\": 2, \"mb_written_and_read_per_round\": %.1f, "
               "\"wrong_values\": %llu}\n", (mode & 2) ? "flags" : "counter", mode, G, rounds, 1e3 * ms / rounds, (mode & 1) ? mb : 0.0, herr);
    }
  }
  return herr != 0;
}
