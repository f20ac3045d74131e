// harness_sinfit.hip -- the reference's own TNLS problem (tests/TNLS_unit_test.cpp:151-260: fit y = sin(b0 t + b1) to m
// samples; root finding on exact data, least squares on noisy data, with and without the right preconditioner R^-1 of
// J'J = R'R) ON THE DEVICE, as a client that brings HIP kernels of its own would write it:
//
//   F(beta)       one kernel: residual y - sin(b0 t + b1), beta read from device memory
//   J(beta)       one kernel fills the two Jacobian columns; the pair handed back is two FRESH mi_op handles (callback
//                 operators 2 -> m and m -> 2) wrapped as MI355::DeviceHessian -- a Jacobian that CHANGES with every outer
//                 iteration, so Riemannian::TNLS must re-tag its LSQR operators after every linearisation
//                 (TNLS.h:414-462 of the reference; the retag_for_device step of this repository's TNLS.h)
//   (M, M')       R from a one-workgroup Cholesky of J'J on the device, applied by two small kernels, tagged as well
//
// mode 0: tagged callables -> every inner solve must run in the fused mi_lsqr (asserted by the caller through
// mi_ctx_fusion_counters); mode 1: the same kernels behind plain lambdas -> the generic LSQR loop on DeviceVector.
// Built by optimization_amd.build.build_harness() with hipcc into tests/cpp/libharness_sinfit.so.  No oracle code.
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "Optimization/MI355/Device.h"
#include "Optimization/Riemannian/TNLS.h"
#include "mi355opt.h"

using namespace Optimization;
using MI355::check;
using MI355::Context;
using MI355::DeviceVector;
namespace LA = Optimization::LinearAlgebra;
namespace RM = Optimization::Riemannian;

namespace {
constexpr int kT = 256;
const double *cptr(const mi_vec *v) { void *p = nullptr; mi_vec_data(v, &p); return (const double *)p; }
double *mptr(mi_vec *v) { void *p = nullptr; mi_vec_data(v, &p); return (double *)p; }
hipStream_t stream_of(mi_ctx *c) { void *s = nullptr; mi_ctx_stream(c, &s); return (hipStream_t)s; }

__global__ void k_residual(size_t m, const double *t, const double *y, const double *beta, double *out) {
  const double b0 = beta[0], b1 = beta[1];
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < m; i += (size_t)gridDim.x * kT) out[i] = y[i] - sin(b0 * t[i] + b1);
}
// Jacobian columns: d/db1 = -cos(b0 t + b1), d/db0 = that times t
__global__ void k_jacobian(size_t m, const double *t, const double *beta, double *Jt, double *J1) {
  const double b0 = beta[0], b1 = beta[1];
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < m; i += (size_t)gridDim.x * kT) {
    const double c = -cos(b0 * t[i] + b1);
    J1[i] = c;
    Jt[i] = c * t[i];
  }
}
__global__ void k_jv(size_t m, const double *Jt, const double *J1, const double *v, double *out) {
  const double v0 = v[0], v1 = v[1];
  for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < m; i += (size_t)gridDim.x * kT) out[i] = Jt[i] * v0 + J1[i] * v1;
}
// one workgroup, fixed tree: deterministic sums over the m rows
template <int K, class F>
__device__ void block_sums(size_t m, double (&acc)[K], F f) {
  __shared__ double lds[K][kT];
  for (int c = 0; c < K; ++c) acc[c] = 0;
  for (size_t i = threadIdx.x; i < m; i += kT) f(i, acc);
  for (int c = 0; c < K; ++c) lds[c][threadIdx.x] = acc[c];
  __syncthreads();
  for (int s = kT / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int c = 0; c < K; ++c) lds[c][threadIdx.x] += lds[c][threadIdx.x + s];
    __syncthreads();
  }
  for (int c = 0; c < K; ++c) acc[c] = lds[c][0];
}
__global__ __launch_bounds__(kT) void k_jtw(size_t m, const double *Jt, const double *J1, const double *w, double *out) {
  double a[2];
  block_sums<2>(m, a, [&](size_t i, double (&s)[2]) { s[0] += Jt[i] * w[i]; s[1] += J1[i] * w[i]; });
  if (threadIdx.x == 0) { out[0] = a[0]; out[1] = a[1]; }
}
// R'R = J'J, R = [r00 r01; 0 r11] -> R[0], R[1], R[2]
__global__ __launch_bounds__(kT) void k_chol(size_t m, const double *Jt, const double *J1, double *R) {
  double a[3];
  block_sums<3>(m, a, [&](size_t i, double (&s)[3]) { s[0] += Jt[i] * Jt[i]; s[1] += Jt[i] * J1[i]; s[2] += J1[i] * J1[i]; });
  if (threadIdx.x == 0) {
    const double r00 = sqrt(a[0]), r01 = a[1] / r00;
    R[0] = r00; R[1] = r01; R[2] = sqrt(a[2] - r01 * r01);
  }
}
__global__ void k_rinv(const double *R, const double *v, double *o) {   // o = R^-1 v
  const double o1 = v[1] / R[2];
  o[1] = o1;
  o[0] = (v[0] - R[1] * o1) / R[0];
}
__global__ void k_rinvT(const double *R, const double *v, double *o) {  // o = R^-T v
  const double o0 = v[0] / R[0];
  o[0] = o0;
  o[1] = (v[1] - R[1] * o0) / R[2];
}

struct SinFit {
  mi_ctx *ctx;
  size_t m;
  DeviceVector t, y, Jt, J1, R;
  hipStream_t st;
  SinFit(const Context &c, size_t m_, const double *th, const double *yh)
      : ctx(c.get()), m(m_), t(c, th, m_), y(c, yh, m_), Jt(c, m_), J1(c, m_), R(c, std::vector<double>{1.0, 0.0, 1.0}),
        st(stream_of(c.get())) {}
  int grid() const { return (int)((m + kT - 1) / kT < 512 ? (m + kT - 1) / kT : 512); }
  DeviceVector F(const DeviceVector &b) const {
    DeviceVector r = DeviceVector::on(ctx, m);
    hipLaunchKernelGGL(k_residual, dim3(grid()), dim3(kT), 0, st, m, cptr(t.handle()), cptr(y.handle()), cptr(b.handle()),
                       mptr(r.handle()));
    check(mi_vec_touch(r.handle()));
    return r;
  }
  void linearise(const DeviceVector &b) {
    hipLaunchKernelGGL(k_jacobian, dim3(grid()), dim3(kT), 0, st, m, cptr(t.handle()), cptr(b.handle()), mptr(Jt.handle()),
                       mptr(J1.handle()));
    hipLaunchKernelGGL(k_chol, dim3(1), dim3(kT), 0, st, m, cptr(Jt.handle()), cptr(J1.handle()), mptr(R.handle()));
    check(mi_vec_touch(Jt.handle()));
    check(mi_vec_touch(J1.handle()));
    check(mi_vec_touch(R.handle()));
  }
  // the four operators as C-ABI callbacks (enqueue on the context stream, no synchronisation)
  static int cb_jv(void *u, const mi_vec *in, mi_vec *out) {
    const SinFit *s = (const SinFit *)u;
    hipLaunchKernelGGL(k_jv, dim3(s->grid()), dim3(kT), 0, s->st, s->m, cptr(s->Jt.handle()), cptr(s->J1.handle()), cptr(in),
                       mptr(out));
    return MI_OK;
  }
  static int cb_jtw(void *u, const mi_vec *in, mi_vec *out) {
    const SinFit *s = (const SinFit *)u;
    hipLaunchKernelGGL(k_jtw, dim3(1), dim3(kT), 0, s->st, s->m, cptr(s->Jt.handle()), cptr(s->J1.handle()), cptr(in), mptr(out));
    return MI_OK;
  }
  static int cb_rinv(void *u, const mi_vec *in, mi_vec *out) {
    const SinFit *s = (const SinFit *)u;
    hipLaunchKernelGGL(k_rinv, dim3(1), dim3(1), 0, s->st, cptr(s->R.handle()), cptr(in), mptr(out));
    return MI_OK;
  }
  static int cb_rinvT(void *u, const mi_vec *in, mi_vec *out) {
    const SinFit *s = (const SinFit *)u;
    hipLaunchKernelGGL(k_rinvT, dim3(1), dim3(1), 0, s->st, cptr(s->R.handle()), cptr(in), mptr(out));
    return MI_OK;
  }
};

std::shared_ptr<mi_op> make_op(mi_ctx *ctx, size_t n_in, size_t n_out, mi_apply_fn fn, void *user) {
  mi_op *op = nullptr;
  check(mi_op_create_callback_rect(ctx, n_in, n_out, fn, user, &op));
  return std::shared_ptr<mi_op>(op, [](mi_op *o) { mi_op_destroy(o); });
}
thread_local std::string g_msg;
}  // namespace

extern "C" const char *hs_last_error() { return g_msg.c_str(); }

extern "C" int hs_tnls_sinfit(size_t m, const double *t, const double *y, const double *beta0, int with_precon, int mode,
                              double root_tolerance, double gradient_tolerance, double Delta_tolerance,
                              size_t max_iterations, double *beta_out, double *f_out, double *gradnorm_out,
                              int *status_out, size_t *outer_out, size_t *inner_total_out, size_t *jacobians_out,
                              mi_fusion_counters *counters_out) {
  try {
    Context ctx(0);
    SinFit prob(ctx, m, t, y);
    size_t jacobians = 0;
    // every operator handed back is kept alive until the run ends (TNLS holds the callables, not the handles)
    std::vector<std::shared_ptr<mi_op>> keep;
    RM::Mapping<DeviceVector, DeviceVector> F = [&](const DeviceVector &b) { return prob.F(b); };
    RM::JacobianPairFunction<DeviceVector, DeviceVector, DeviceVector> J = [&](const DeviceVector &b) {
      prob.linearise(b);
      ++jacobians;
      RM::Jacobian<DeviceVector, DeviceVector, DeviceVector> dF;
      RM::JacobianAdjoint<DeviceVector, DeviceVector, DeviceVector> dFt;
      // FRESH handles each time: what a client whose Jacobian object is rebuilt per linearisation hands back
      auto a = make_op(ctx.get(), 2, m, &SinFit::cb_jv, &prob), at = make_op(ctx.get(), m, 2, &SinFit::cb_jtw, &prob);
      keep.push_back(a);
      keep.push_back(at);
      if (mode == 0) {
        dF = MI355::DeviceHessian{a.get()};
        dFt = MI355::DeviceHessian{at.get()};
      } else {
        mi_op *ra = a.get(), *rat = at.get();
        dF = [ra](const DeviceVector &, const DeviceVector &v) { return MI355::apply_device_operator(ra, v); };
        dFt = [rat](const DeviceVector &, const DeviceVector &w) { return MI355::apply_device_operator(rat, w); };
      }
      return std::make_pair(dF, dFt);
    };
    auto mop = make_op(ctx.get(), 2, 2, &SinFit::cb_rinv, &prob), mtop = make_op(ctx.get(), 2, 2, &SinFit::cb_rinvT, &prob);
    RM::LinearOperator<DeviceVector, DeviceVector> M, MT;
    RM::RiemannianMetric<DeviceVector, DeviceVector, double> metric;
    LA::InnerProduct<DeviceVector, double> ipY;
    if (mode == 0) {
      M = MI355::DeviceHessian{mop.get()};
      MT = MI355::DeviceHessian{mtop.get()};
      metric = MI355::FrobeniusMetric{};
      ipY = MI355::FrobeniusInnerProduct{};
    } else {
      mi_op *rm = mop.get(), *rmt = mtop.get();
      M = [rm](const DeviceVector &, const DeviceVector &v) { return MI355::apply_device_operator(rm, v); };
      MT = [rmt](const DeviceVector &, const DeviceVector &v) { return MI355::apply_device_operator(rmt, v); };
      metric = [](const DeviceVector &, const DeviceVector &a, const DeviceVector &c) { return a.dot(c); };
      ipY = [](const DeviceVector &a, const DeviceVector &c) { return a.dot(c); };
    }
    RM::Retraction<DeviceVector, DeviceVector> retract = [](const DeviceVector &x, const DeviceVector &v) { return x + v; };
    RM::TNLSParams<double> p;
    p.relative_decrease_tolerance = 0;
    p.stepsize_tolerance = 0;
    p.gradient_tolerance = gradient_tolerance;
    p.root_tolerance = root_tolerance;
    p.Delta_tolerance = Delta_tolerance;
    p.max_iterations = max_iterations;
    std::optional<RM::TNLSPreconditioner<DeviceVector, DeviceVector>> precon;
    if (with_precon) precon = std::make_pair(M, MT);
    DeviceVector b0(ctx, beta0, 2);
    check(mi_ctx_fusion_counters_reset(ctx.get()));
    RM::TNLSResult<DeviceVector, double> r =
        RM::TNLS<DeviceVector, DeviceVector, DeviceVector, double>(F, J, metric, ipY, retract, b0, precon, p);
    check(mi_ctx_fusion_counters(ctx.get(), counters_out));
    const std::vector<double> xh = r.x.to_host();
    beta_out[0] = xh[0];
    beta_out[1] = xh[1];
    *f_out = r.f;
    *gradnorm_out = r.gradfx_norm;
    *status_out = static_cast<int>(r.status);
    *outer_out = r.inner_iterations.size();
    size_t tot = 0;
    for (size_t k : r.inner_iterations) tot += k;
    *inner_total_out = tot;
    *jacobians_out = jacobians;
  } catch (const std::invalid_argument &e) {
    g_msg = e.what();
    return -1;
  } catch (const std::exception &e) {
    g_msg = e.what();
    return -2;
  }
  return 0;
}
