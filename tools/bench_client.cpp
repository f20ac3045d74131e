// bench_client.cpp -- a CLIENT of the drop-in template layer for bench.py's cfg3 / cfg5 legs: the same calls a user of
// the reference writes (Optimization::Riemannian::TNT, Optimization::LinearAlgebra::LOBPCG, reference TNT.h:242-254,
// LOBPCG.h:376-385) on MI355::DeviceVector / MI355::DeviceMatrix, timed on a microsecond clock.  Built by
// optimization_amd.build.build_harness() into tools/libbench_client.so and linked to libmi355opt.so only: nothing under
// oracle/ is included, linked or called here (these legs are measurements, not checks).
#include <chrono>
#include <cstdint>
#include <cstring>
#include <optional>
#include <string>
#include <vector>

#include "Optimization/LinearAlgebra/LOBPCG.h"
#include "Optimization/MI355/Device.h"
#include "Optimization/MI355/Matrix.h"
#include "Optimization/MI355/SO3.h"
#include "Optimization/Riemannian/TNT.h"

using namespace Optimization;
using MI355::Context;
using MI355::DeviceVector;
namespace LA = Optimization::LinearAlgebra;
namespace RM = Optimization::Riemannian;

static thread_local std::string g_msg;
extern "C" const char *bc_last_error() { return g_msg.c_str(); }

#define BC_GUARD_BEGIN try {
#define BC_GUARD_END               \
  }                                \
  catch (const std::exception &e) { \
    g_msg = e.what();              \
    return -1;                     \
  }                                \
  return 0;

struct bc_tnt_report {
  double seconds;        // wall time of the LAST TNT call (device drained before and after)
  double f;              // final objective
  double gradfx_norm;
  size_t outer_iterations;
  size_t inner_iterations_total;
  size_t host_syncs;     // mi_ctx_sync_count over that call
  int status;
};

// TNT on SO(3)^N chordal rotation averaging with the 3x3 block-Jacobi preconditioner (BASELINE cfg3) on the caller's
// context: `repeats` runs from R0, the last one timed (the first fills the memory pool and loads the kernels).
extern "C" int bc_tnt_so3n(void *ctx_handle, size_t N, size_t E, const int32_t *ei, const int32_t *ej, const double *Rt,
                           const double *w, const double *R0, size_t max_outer, size_t max_tpcg, int repeats,
                           bc_tnt_report *rep) {
  BC_GUARD_BEGIN
  Context ctx = Context::adopt(static_cast<mi_ctx *>(ctx_handle));
  MI355::RotationAveraging prob(ctx, N, E, ei, ej, Rt, w);
  DeviceVector x0(ctx, R0, 9 * N);
  RM::TNTParams<double> tp;  // reference defaults (TNT.h:76-130) except the stopping rules, which must not fire
  tp.max_iterations = max_outer;
  tp.max_TPCG_iterations = max_tpcg;
  tp.gradient_tolerance = 1e-12;
  tp.relative_decrease_tolerance = 0.0;
  tp.stepsize_tolerance = 0.0;
  tp.preconditioned_gradient_tolerance = 0.0;
  std::optional<RM::LinearOperator<DeviceVector, DeviceVector>> pc = prob.preconditioner();
  for (int r = 1; r < repeats; ++r)
    (void)RM::TNT<DeviceVector, DeviceVector>(prob.objective(), prob.quadratic_model(), prob.metric(), prob.retraction(),
                                              x0, pc, tp);
  size_t s0 = 0, s1 = 0;
  MI355::check(mi_ctx_sync(ctx.get()));
  MI355::check(mi_ctx_sync_count(ctx.get(), &s0));
  const auto t0 = std::chrono::steady_clock::now();
  RM::TNTResult<DeviceVector, double> res = RM::TNT<DeviceVector, DeviceVector>(
      prob.objective(), prob.quadratic_model(), prob.metric(), prob.retraction(), x0, pc, tp);
  MI355::check(mi_ctx_sync(ctx.get()));
  rep->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  MI355::check(mi_ctx_sync_count(ctx.get(), &s1));
  rep->host_syncs = s1 - s0 - 1;
  rep->f = res.f;
  rep->gradfx_norm = res.gradfx_norm;
  rep->outer_iterations = res.inner_iterations.size();
  rep->inner_iterations_total = 0;
  for (size_t k : res.inner_iterations) rep->inner_iterations_total += k;
  rep->status = static_cast<int>(res.status);
  BC_GUARD_END
}

struct bc_lobpcg_report {
  double seconds_per_iteration;  // user function to user function, first two intervals dropped (basis not yet full)
  double seconds_total;
  double theta0;                 // lowest Ritz value at the end
  double max_residual;           // largest residual norm the last user-function call saw
  size_t iterations;
  size_t iterations_timed;
  size_t nc;
};

// LOBPCG (random-X0 overload, LOBPCG.h:376-385) on the caller's context and sparse matrix: the tagged sparse panel
// operator, B and T absent (BASELINE cfg5).
extern "C" int bc_lobpcg(void *ctx_handle, void *csr_handle, size_t m, size_t nx, size_t nev, size_t max_iters,
                         double tau, bc_lobpcg_report *rep) {
  BC_GUARD_BEGIN
  using MI355::DeviceMatrix;
  using MI355::HostVectorD;
  using Op = LA::SymmetricLinearOperator<DeviceMatrix>;
  Context ctx = Context::adopt(static_cast<mi_ctx *>(ctx_handle));
  MI355::make_current(ctx);
  Op A = MI355::DeviceCsrPanelOperator{static_cast<mi_csr *>(csr_handle)};
  std::vector<std::chrono::steady_clock::time_point> stamps;
  double rmax = 0.0;
  std::optional<LA::LOBPCGUserFunction<HostVectorD, DeviceMatrix>> uf =
      [&](size_t, const Op &, const std::optional<Op> &, const std::optional<Op> &, size_t, const HostVectorD &,
          const DeviceMatrix &, const HostVectorD &r, size_t) {
        stamps.push_back(std::chrono::steady_clock::now());
        rmax = 0.0;
        for (size_t i = 0; i < r.size(); ++i) rmax = r(i) > rmax ? r(i) : rmax;
        return false;
      };
  size_t iters = 0, nc = 0;
  MI355::check(mi_ctx_sync(ctx.get()));
  const auto t0 = std::chrono::steady_clock::now();
  std::pair<HostVectorD, DeviceMatrix> out = LA::LOBPCG<HostVectorD, DeviceMatrix>(
      A, std::optional<Op>(), std::optional<Op>(), m, nx, nev, max_iters, iters, nc, tau, uf);
  MI355::check(mi_ctx_sync(ctx.get()));
  rep->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  rep->iterations = iters;
  rep->nc = nc;
  rep->theta0 = out.first(0);
  rep->max_residual = rmax;
  rep->iterations_timed = stamps.size() > 2 ? stamps.size() - 2 : 0;
  rep->seconds_per_iteration =
      rep->iterations_timed ? std::chrono::duration<double>(stamps.back() - stamps[1]).count() / (double)rep->iterations_timed
                            : 0.0;
  BC_GUARD_END
}
