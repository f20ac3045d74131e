// ql_chain.hip -- how long the SERIAL part of a device-resident Rayleigh-Ritz solve would take on MI355X (r04).
//
// The r03 verdict asked for the LOBPCG Rayleigh-Ritz solve (reference LOBPCG.h:53-62; ours: DenseSymmetricEigen.h) as
// one workgroup: equilibrate, Cholesky, tridiagonalise, implicit QL in LDS.  The QL phase is a chain of plane
// rotations, each a square root and two divisions whose inputs are the previous rotation's outputs; nothing in it can
// be spread over lanes.  This microbenchmark runs exactly that recurrence -- the scalar part of tql2, no eigenvector
// work at all -- on ONE lane for a 72 x 72 (and 48, 96) symmetric tridiagonal matrix and reports rotations and time:
// a lower bound for ANY device QL, to hold against the host solver's total (0.13-0.19 ms at n = 72 on the GPU box).
// Second figure: one round of a parallel cyclic Jacobi sweep's critical path (rotation set-up + two barriers), times
// the ~8 x (n - 1) rounds such a method needs.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/ql_chain.hip -o tools/microbench/ql_chain && tools/microbench/ql_chain
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_ql_chain(int n, const double *d_in, const double *e_in, double *d_out, int *rotations) {
  if (threadIdx.x != 0) return;
  double d[96], e[96];
  for (int i = 0; i < n; ++i) { d[i] = d_in[i]; e[i] = e_in[i]; }
  int rot = 0;
  double shift = 0, tst = 0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    tst = fmax(tst, fabs(d[l]) + fabs(e[l]));
    int mm = l;
    while (mm < n - 1 && fabs(e[mm]) > eps * tst) ++mm;
    if (mm > l) {
      int guard = 0;
      do {
        double g = d[l];
        double p = (d[l + 1] - g) / (2 * e[l]);
        double r = sqrt(p * p + 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        shift += h;
        p = d[mm];
        double c = 1, c2 = 1, c3 = 1, s = 0, s2 = 0;
        const double el1 = e[l + 1];
        for (int i = mm - 1; i >= l; --i) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = sqrt(p * p + e[i] * e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          ++rot;
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (fabs(e[l]) > eps * tst && ++guard < 200);
    }
    d[l] += shift;
    e[l] = 0;
  }
  for (int i = 0; i < n; ++i) d_out[i] = d[i];
  *rotations = rot;
}

// critical path of ONE round of a parallel Jacobi sweep: every pair's rotation set-up (the same latency on all
// lanes), a barrier, the row / column update of an n x n matrix in LDS by 256 threads, a barrier
__global__ void k_jacobi_round(int n, int rounds, double *M) {
  __shared__ double A[96 * 97];
  __shared__ double cs[96];
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) A[(i / n) * 97 + i % n] = M[i];
  __syncthreads();
  for (int r = 0; r < rounds; ++r) {
    const int t = threadIdx.x;
    if (t < n / 2) {
      const int p = t, q = n - 1 - t;
      const double app = A[p * 97 + p], aqq = A[q * 97 + q], apq = A[p * 97 + q] + 1e-3;
      const double th = (aqq - app) / (2 * apq);
      const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0);
      cs[2 * t] = c;
      cs[2 * t + 1] = tt * c;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < (n / 2) * n; e += blockDim.x) {
      const int pr = e / n, k = e % n, p = pr, q = n - 1 - pr;
      const double c = cs[2 * pr], s = cs[2 * pr + 1];
      const double ap = A[p * 97 + k], aq = A[q * 97 + k];
      A[p * 97 + k] = c * ap - s * aq;
      A[q * 97 + k] = s * ap + c * aq;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n * n; i += blockDim.x) M[i] = A[(i / n) * 97 + i % n];
}

int main() {
  for (int n : {48, 72, 96}) {
    std::vector<double> d(n), e(n);
    unsigned long long lcg = 20260928ull + n;
    auto rnd = [&] { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) { d[i] = 1.0 + rnd(); e[i] = (i + 1 < n) ? 0.5 * rnd() - 0.25 : 0.0; }
    double *dd, *de, *dout, *M;
    int *drot;
    hipMalloc(&dd, n * 8); hipMalloc(&de, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&drot, 4); hipMalloc(&M, 96 * 96 * 8);
    hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(de, e.data(), n * 8, hipMemcpyHostToDevice);
    hipMemset(M, 0, 96 * 96 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    float ms_ql = 0, ms_j = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k_ql_chain, dim3(1), dim3(64), 0, 0, n, dd, de, dout, drot);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms_ql, a, b);
    }
    int rot = 0;
    hipMemcpy(&rot, drot, 4, hipMemcpyDeviceToHost);
    const int rounds = 8 * (n - 1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k_jacobi_round, dim3(1), dim3(256), 0, 0, n, rounds, M);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms_j, a, b);
    }
    std::printf("n = %2d: QL scalar chain alone, one lane: %5d rotations, %7.1f us (%5.1f ns per rotation);  "
                "%d rounds of a parallel Jacobi sweep's critical path, one workgroup: %7.1f us (%4.2f us per round)\n",
                n, rot, 1e3 * ms_ql, 1e6 * ms_ql / (rot ? rot : 1), rounds, 1e3 * ms_j, 1e3 * ms_j / rounds);
    hipFree(dd); hipFree(de); hipFree(dout); hipFree(drot); hipFree(M);
  }
  return 0;
}
