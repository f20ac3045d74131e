// examples/stpcg_user_stencil.hip -- a USER-written HIP Hessian-vector product inside the fused Steihaug-Toint CG.
//
// The reference's STPCG takes the Hessian as a callable (IterativeSolvers.h:166-179, applied at :294; TNT binds the
// caller's QuadraticModel into it at TNT.h:400-426).  Here the callable is a hand-written stencil kernel,
//     (H v)_i = (2 + sigma) v_i - v_{i-1} - v_{i+1},
// registered with mi_op_create_callback_fused: its kernel also leaves the per-workgroup partial sums of <v,Hv>,
// <Hv,Hv>, <v,v> (IterativeSolvers.h:300,305-306), so one STPCG iteration is three launches -- this kernel and the
// library's two CG kernels -- exactly as with the built-in operators.  The same product registered as a plain
// callback (mi_op_create_callback) costs a fourth launch and a second pass over v and Hv per iteration.
//
// build:  hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include -I optimization_amd/include \
//               examples/stpcg_user_stencil.hip -o examples/bin/stpcg_user_stencil -L optimization_amd -lmi355opt
// run:    examples/bin/stpcg_user_stencil [n] [dump.bin]   (dump: n, iterations, then g and s as raw doubles)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/MI355/Device.h"
#include "mi355opt.h"

namespace LA = Optimization::LinearAlgebra;
using Optimization::MI355::check;
using Optimization::MI355::Context;
using Optimization::MI355::DeviceVector;

constexpr int kThreads = 256;

template <bool DOTS>
__global__ __launch_bounds__(kThreads) void k_stencil(size_t n, double sigma, const double *__restrict__ v,
                                                      double *__restrict__ Hv, double *__restrict__ partials,
                                                      size_t partial_stride) {
  double acc[3] = {0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const double x = v[i];
    const double y = (2.0 + sigma) * x - (i > 0 ? v[i - 1] : 0.0) - (i + 1 < n ? v[i + 1] : 0.0);
    Hv[i] = y;
    if (DOTS) { acc[0] += x * y; acc[1] += y * y; acc[2] += x * x; }
  }
  if (!DOTS) return;
  // the contract of mi_apply_fused_fn: a DETERMINISTIC workgroup sum (fixed tree, no atomics), one row per workgroup
  __shared__ double lds[3][kThreads];
  for (int c = 0; c < 3; ++c) lds[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int c = 0; c < 3; ++c) lds[c][threadIdx.x] += lds[c][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 3) partials[threadIdx.x * partial_stride + blockIdx.x] = lds[threadIdx.x][0];
}

struct Stencil {
  mi_ctx *ctx;
  size_t n;
  double sigma;
};
static const double *ptr(const mi_vec *v) { void *p = nullptr; mi_vec_data(v, &p); return (const double *)p; }

static int stencil_apply(void *user, const mi_vec *in, mi_vec *out) {  // plain product (mi_op_apply)
  const Stencil *s = (const Stencil *)user;
  void *stream = nullptr;
  mi_ctx_stream(s->ctx, &stream);
  hipLaunchKernelGGL(k_stencil<false>, dim3(512), dim3(kThreads), 0, (hipStream_t)stream, s->n, s->sigma, ptr(in),
                     (double *)ptr(out), (double *)nullptr, (size_t)0);
  return MI_OK;
}
static int stencil_apply_fused(void *user, const mi_vec *in, mi_vec *out, const mi_fused_args *a, int *rows) {
  const Stencil *s = (const Stencil *)user;
  const int grid = a->required_rows > 0 ? a->required_rows : (a->max_rows < 512 ? a->max_rows : 512);
  hipLaunchKernelGGL(k_stencil<true>, dim3(grid), dim3(kThreads), 0, (hipStream_t)a->stream, s->n, s->sigma, ptr(in),
                     (double *)ptr(out), a->partials, a->partial_stride);
  *rows = grid;
  return MI_OK;
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 22;
  Context ctx(0);
  Stencil st{ctx.get(), n, 0.05};
  std::vector<double> gh(n);
  for (size_t i = 0; i < n; ++i) gh[i] = std::sin(0.001 * (double)i) + 0.5 * std::cos(0.37 * (double)i);
  DeviceVector g(ctx, gh.data(), n);
  LA::InnerProduct<DeviceVector> ip = Optimization::MI355::FrobeniusInnerProduct{};
  mi_op *fused = nullptr, *plain = nullptr;
  check(mi_op_create_callback_fused(ctx.get(), n, stencil_apply, stencil_apply_fused, &st, &fused));
  check(mi_op_create_callback(ctx.get(), n, stencil_apply, &st, &plain));
  std::vector<double> s_host;
  size_t its[2] = {0, 0};
  for (int which = 0; which < 2; ++which) {  // 0: fused callback, 1: plain callback
    LA::SymmetricLinearOperator<DeviceVector> H = Optimization::MI355::DeviceOperator{which ? plain : fused};
    double mnorm = 0, ms = 0;
    size_t iters = 0;
    DeviceVector s;
    for (int rep = 0; rep < 3; ++rep) {  // (the first repetition warms the pool and loads the kernels)
      check(mi_timer_start(ctx.get()));
      s = LA::STPCG<DeviceVector, std::nullptr_t>(g, H, ip, mnorm, iters, 1e9, 200, 1e-8, 1.0);
      check(mi_timer_stop(ctx.get(), &ms));
    }
    its[which] = iters;
    DeviceVector r = H(s) + g;
    std::printf("%-15s %3zu iterations, |Hs + g| / |g| = %.2e, %.1f us per iteration\n",
                which ? "plain callback:" : "fused callback:", iters, r.norm() / g.norm(), 1e3 * ms / (double)iters);
    if (which == 0) s_host = s.to_host();
  }
  if (argc > 2) {
    FILE *f = std::fopen(argv[2], "wb");
    const double hdr[2] = {(double)n, (double)its[0]};
    std::fwrite(hdr, sizeof(double), 2, f);
    std::fwrite(gh.data(), sizeof(double), n, f);
    std::fwrite(s_host.data(), sizeof(double), n, f);
    std::fclose(f);
  }
  mi_op_destroy(fused);
  mi_op_destroy(plain);
  return its[0] == its[1] ? 0 : 1;
}
