// panel_stream.hip -- what the ACCESS PATTERNS of the LOBPCG panel update and Gram kernels can reach on MI355X, without
// their arithmetic (r04).
//
// k_panel_update_mfma (csrc/lobpcg.hip) reads the 72 columns of a column-major m x 72 basis and writes 48 columns of
// X and P: 1.92 GB at m = 2 000 376 in 400-430 us = 4.5-4.8 TB/s, with the matrix pipe ~45 % busy, and neither more
// loads in flight (three operand sets, k_panel_update_mfma_pf) nor cache policies move it.  This program runs the same
// loads and stores with a trivial sum in between, in two lane mappings:
//   seg : lane (i = l & 15, q = l >> 4) touches row r0 + i of column 4 kk + q -- the MFMA operand layout: every wave
//         instruction covers four 128-byte row segments in four columns (what the product kernel does);
//   row : lane l touches row r0 + l of ONE column -- 512 contiguous bytes per wave instruction.
// and with 1 or 2 waves per SIMD.  If "row" is much faster than "seg", a transposing stage through LDS pays; if both sit
// where the product kernel sits, the panel layout itself (120 streams 16 MB apart) is the bound.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/panel_stream.hip -o tools/microbench/panel_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int KS = 72, KC = 48;

template <bool NT>
__global__ __launch_bounds__(256) void k_seg(size_t nblocks, size_t m, const double *__restrict__ S,
                                             double *__restrict__ Y) {
  const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
  const double *sbase = S + (size_t)q * m + i;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t b = wave; b < nblocks; b += nwaves) {
    double cur[KS / 4];
#pragma unroll
    for (int kk = 0; kk < KS / 4; ++kk) cur[kk] = sbase[(size_t)kk * 4 * m + b * 16];
    double acc = 0;
#pragma unroll
    for (int kk = 0; kk < KS / 4; ++kk) acc += cur[kk];
    const size_t row = b * 16 + i;
#pragma unroll
    for (int t = 0; t < KC / 4; ++t) {
      const int col = 4 * t + q;
      if (NT) __builtin_nontemporal_store(acc + t, Y + (size_t)col * m + row);
      else Y[(size_t)col * m + row] = acc + t;
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_row(size_t nblocks64, size_t m, const double *__restrict__ S,
                                             double *__restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t b = wave; b < nblocks64; b += nwaves) {
    const size_t row = b * 64 + lane;
    double cur[KS];
#pragma unroll
    for (int c = 0; c < KS; ++c) cur[c] = S[(size_t)c * m + row];
    double acc = 0;
#pragma unroll
    for (int c = 0; c < KS; ++c) acc += cur[c];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (NT) __builtin_nontemporal_store(acc + c, Y + (size_t)c * m + row);
      else Y[(size_t)c * m + row] = acc + c;
    }
  }
}

// reads only / writes only in the row mapping: the two directions apart
__global__ __launch_bounds__(256) void k_read(size_t nblocks64, size_t m, const double *__restrict__ S,
                                              double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  double acc = 0;
  for (size_t b = wave; b < nblocks64; b += nwaves) {
    const size_t row = b * 64 + lane;
    double cur[KS];
#pragma unroll
    for (int c = 0; c < KS; ++c) cur[c] = S[(size_t)c * m + row];
#pragma unroll
    for (int c = 0; c < KS; ++c) acc += cur[c];
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(size_t nblocks64, size_t m, double *__restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t b = wave; b < nblocks64; b += nwaves) {
    const size_t row = b * 64 + lane;
#pragma unroll
    for (int c = 0; c < KC; ++c) __builtin_nontemporal_store((double)c, Y + (size_t)c * m + row);
  }
}

// writes only, 16 and 32 bytes per lane (1 and 2 KB contiguous per wave instruction) -- does the store width matter?
typedef double double2s __attribute__((ext_vector_type(2)));
typedef double double4s __attribute__((ext_vector_type(4)));
template <int W>
__global__ __launch_bounds__(256) void k_write_wide(size_t nblocks, size_t m, double *__restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t b = wave; b < nblocks; b += nwaves) {
    const size_t row = (b * 64 + lane) * W;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (W == 2) __builtin_nontemporal_store((double2s){(double)c, 1.0}, reinterpret_cast<double2s *>(Y + (size_t)c * m + row));
      else __builtin_nontemporal_store((double4s){(double)c, 1.0, 2.0, 3.0}, reinterpret_cast<double4s *>(Y + (size_t)c * m + row));
    }
  }
}

// the same 16 bytes per lane WITHOUT another row-to-lane mapping: a wave still owns 64 rows, lanes 2p and 2p + 1 swap
// one value each, the even lane then stores rows 2p, 2p + 1 of column c and the odd lane those of column c + 1
// (512 contiguous bytes per column, two columns per store instruction)
__global__ __launch_bounds__(256) void k_write_paired(size_t nblocks64, size_t m, double *__restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  for (size_t b = wave; b < nblocks64; b += nwaves) {
    const size_t row2 = b * 64 + (lane & ~1);
#pragma unroll
    for (int c = 0; c < KC; c += 2) {
      const double own0 = (double)c + lane, own1 = (double)c + 1 + lane;           // this lane's row of columns c, c + 1
      const double give = (lane & 1) ? own0 : own1;
      const double got = __shfl_xor(give, 1);
      const double2s v = (lane & 1) ? (double2s){got, own1} : (double2s){own0, got};
      __builtin_nontemporal_store(v, reinterpret_cast<double2s *>(Y + (size_t)(c + (lane & 1)) * m + row2));
    }
  }
}

// reads only in the Gram kernels' mapping: lane (i = l & 15, q = l >> 4) loads the four rows r0 + 4 q .. + 3 of column
// 16 t + i with one 32-byte load (a wave instruction = 16 columns x one 128-byte segment), T tiles of 16 columns, twice
// (S and A(S)); a wave takes every nwaves-th 16-row step
typedef double double4l __attribute__((ext_vector_type(4)));
template <int T>
__global__ __launch_bounds__(256) void k_gram_read(size_t nsteps, size_t m, const double *__restrict__ S,
                                                   const double *__restrict__ S2, double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  const size_t lane_off = 4 * (size_t)(lane >> 4) + (size_t)(lane & 15) * m;
  double acc = 0;
  for (size_t b = wave; b < nsteps; b += nwaves) {
    double4l a[T], c[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      a[t] = *reinterpret_cast<const double4l *>(S + (size_t)16 * t * m + lane_off + b * 16);
      c[t] = *reinterpret_cast<const double4l *>(S2 + (size_t)16 * t * m + lane_off + b * 16);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) acc += a[t][0] + a[t][3] + c[t][1] + c[t][2];
  }
  if (acc == 12345.678) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t m = argc > 1 ? (size_t)atoll(argv[1]) : 2000376;
  double *S, *Y;
  CK(hipMalloc(&S, m * KS * sizeof(double)));
  CK(hipMalloc(&Y, m * KC * sizeof(double)));
  CK(hipMemset(S, 0, m * KS * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double gb = (double)m * (KS + KC) * 8 / 1e9;
  auto run = [&](const char *name, auto launch, double bytes_gb) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.1f us  %6.2f TB/s\n", name, 1e3 * ms / reps, bytes_gb / (ms / reps));
  };
  const size_t nb16 = m / 16, nb64 = m / 64;
  for (int wg : {256, 512, 1024, 2048}) {
    printf("grid %d workgroups of 4 waves\n", wg);
    run("  seg, plain stores", [&] { hipLaunchKernelGGL(k_seg<false>, dim3(wg), dim3(256), 0, 0, nb16, m, S, Y); }, gb);
    run("  seg, non-temporal stores", [&] { hipLaunchKernelGGL(k_seg<true>, dim3(wg), dim3(256), 0, 0, nb16, m, S, Y); }, gb);
    run("  row, plain stores", [&] { hipLaunchKernelGGL(k_row<false>, dim3(wg), dim3(256), 0, 0, nb64, m, S, Y); }, gb);
    run("  row, non-temporal stores", [&] { hipLaunchKernelGGL(k_row<true>, dim3(wg), dim3(256), 0, 0, nb64, m, S, Y); }, gb);
    run("  row, 72 columns read only", [&] { hipLaunchKernelGGL(k_read, dim3(wg), dim3(256), 0, 0, nb64, m, S, Y); },
        (double)m * KS * 8 / 1e9);
    run("  row, 48 columns written only", [&] { hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, nb64, m, Y); },
        (double)m * KC * 8 / 1e9);
  }
  for (int wg : {512, 2048}) {
    printf("wide stores, %d workgroups\n", wg);
    run("  48 columns written, 16 B per lane", [&] { hipLaunchKernelGGL(k_write_wide<2>, dim3(wg), dim3(256), 0, 0, m / 128, m, Y); },
        (double)m * KC * 8 / 1e9);
    run("  48 columns written, paired 16 B", [&] { hipLaunchKernelGGL(k_write_paired, dim3(wg), dim3(256), 0, 0, nb64, m, Y); },
        (double)m * KC * 8 / 1e9);
    run("  48 columns written, 32 B per lane", [&] { hipLaunchKernelGGL(k_write_wide<4>, dim3(wg), dim3(256), 0, 0, m / 256, m, Y); },
        (double)m * KC * 8 / 1e9);
  }
  // the Gram mapping, read only: 2 x 48 and 2 x 80 columns (two panels of 80 columns: S itself serves as both)
  double *S2;
  CK(hipMalloc(&S2, m * 80 * sizeof(double)));
  CK(hipMemset(S2, 0, m * 80 * sizeof(double)));
  double *S1;
  CK(hipMalloc(&S1, m * 80 * sizeof(double)));
  CK(hipMemset(S1, 0, m * 80 * sizeof(double)));
  for (int wg : {256, 512}) {
    printf("Gram mapping, %d workgroups\n", wg);
    run("  3 tiles x 2 panels, read only", [&] { hipLaunchKernelGGL(k_gram_read<3>, dim3(wg), dim3(256), 0, 0, nb16, m, S1, S2, Y); },
        (double)m * 96 * 8 / 1e9);
    run("  5 tiles x 2 panels, read only", [&] { hipLaunchKernelGGL(k_gram_read<5>, dim3(wg), dim3(256), 0, 0, nb16, m, S1, S2, Y); },
        (double)m * 160 * 8 / 1e9);
  }
  return 0;
}
