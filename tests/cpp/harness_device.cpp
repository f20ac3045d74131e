// harness_device.cpp -- the drop-in template API exercised with Vector = MI355::DeviceVector, i.e.
// the way a client of the reference would use the MI355X build: same function templates, same
// std::function callables, vectors living in HBM.  Mirrors the reference's own unit tests
// (tests/IterativeSolvers_unit_test.cpp STPCG cases, tests/TNT_unit_test.cpp sphere cases,
// tests/GradientDescent_unit_test.cpp sphere case) plus the BASELINE cfg2 Stiefel problem.
// Entry points hd_* are called from pytest (-m gpu) and compared with the oracle / golden fixtures.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <optional>
#include <vector>

#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/LinearAlgebra/LOBPCG.h"
#include "Optimization/MI355/Device.h"
#include "Optimization/MI355/Matrix.h"
#include "Optimization/MI355/SO3.h"
#include "Optimization/MI355/Stiefel.h"
#include "Optimization/Riemannian/GradientDescent.h"
#include "Optimization/Riemannian/TNLS.h"
#include "Optimization/Riemannian/TNT.h"
#include "kkt_dense.h"  // host-side dense KKT solves of the projected-STPCG user callables (test infrastructure)
#include "oracle.h"  // result / parameter structs only (plain data)

using namespace Optimization;
using MI355::Context;
using MI355::DeviceVector;
namespace LA = Optimization::LinearAlgebra;
namespace RM = Optimization::Riemannian;

static thread_local std::string g_msg;
extern "C" const char *hd_last_error() { return g_msg.c_str(); }

#define HD_GUARD_BEGIN try {
#define HD_GUARD_END                         \
  }                                          \
  catch (const std::invalid_argument &e) {   \
    g_msg = e.what();                        \
    return -1;                               \
  }                                          \
  catch (const std::exception &e) {          \
    g_msg = e.what();                        \
    return -2;                               \
  }                                          \
  return 0;

static void fill_params(RM::TNTParams<double> &tp, const orc_tnt_params *p) {
  tp.max_iterations = p->max_iterations;
  tp.max_computation_time = p->max_computation_time;
  tp.gradient_tolerance = p->gradient_tolerance;
  tp.relative_decrease_tolerance = p->relative_decrease_tolerance;
  tp.stepsize_tolerance = p->stepsize_tolerance;
  tp.Delta0 = p->Delta0;
  tp.eta1 = p->eta1;
  tp.eta2 = p->eta2;
  tp.alpha1 = p->alpha1;
  tp.alpha2 = p->alpha2;
  tp.max_TPCG_iterations = p->max_TPCG_iterations;
  tp.kappa_fgr = p->kappa_fgr;
  tp.theta = p->theta;
  tp.preconditioned_gradient_tolerance = p->preconditioned_gradient_tolerance;
  tp.Delta_tolerance = p->Delta_tolerance;
}

// mi_ctx_fusion_counters of the context of the last harness call that took a snapshot (FusionSnap below)
static mi_fusion_counters g_last_fusion = {0, 0, 0, 0, 0, 0, 0};
extern "C" void hd_last_fusion_counters(mi_fusion_counters *out) { *out = g_last_fusion; }
struct FusionSnap {  // declared right behind the Context: its destructor runs before the context is destroyed
  mi_ctx *c;
  explicit FusionSnap(const Context &ctx) : c(ctx.get()) {}
  ~FusionSnap() { (void)mi_ctx_fusion_counters(c, &g_last_fusion); }
};

static double g_last_tnt_seconds = 0.0;
static double g_last_tnt_wall_seconds = 0.0;
static size_t g_last_tnt_syncs = 0;
// wall time of the last TNT call itself on a microsecond clock (TNTResult::elapsed_time has the reference
// Stopwatch's millisecond resolution)
extern "C" double hd_last_tnt_wall_seconds() { return g_last_tnt_wall_seconds; }
// benchmarks: run the TNT call this many times in ONE context and report the last (the first run of a fresh context
// pays for its device allocations -- the memory pool is empty -- and for loading every kernel)
static int g_tnt_repeats = 1;
extern "C" void hd_set_tnt_repeats(int n) { g_tnt_repeats = n > 0 ? n : 1; }
// host<->device synchronisations the library made during the last hd_tnt_stiefel run (mi_ctx_sync_count)
extern "C" size_t hd_last_tnt_syncs() { return g_last_tnt_syncs; }
static double g_last_solve_seconds = 0.0;
// wall time of the last LSQR call through this harness (device drained before and after)
extern "C" double hd_last_solve_seconds() { return g_last_solve_seconds; }
// TNTResult::elapsed_time (TNT.h:608) of the last TNT run through this harness
extern "C" double hd_last_tnt_seconds() { return g_last_tnt_seconds; }

static void export_result(const RM::TNTResult<DeviceVector, double> &r, size_t accepted, orc_tnt_result *res) {
  g_last_tnt_seconds = r.elapsed_time;
  const std::vector<double> x = r.x.to_host();
  std::memcpy(res->x, x.data(), x.size() * sizeof(double));
  res->f = r.f;
  res->gradfx_norm = r.gradfx_norm;
  res->preconditioned_gradfx_norm = r.preconditioned_grad_f_x_norm;
  res->status = static_cast<int>(r.status);
  res->outer_iterations = r.inner_iterations.size();
  res->n_trace = r.objective_values.size();
  for (size_t i = 0; i < res->n_trace; ++i) {
    res->objective_values[i] = r.objective_values[i];
    res->gradient_norms[i] = r.gradient_norms[i];
    res->preconditioned_gradient_norms[i] = r.preconditioned_gradient_norms[i];
    res->trust_region_radius[i] = r.trust_region_radius[i];
  }
  for (size_t i = 0; i < res->outer_iterations; ++i) {
    res->inner_iterations[i] = r.inner_iterations[i];
    res->update_step_norms[i] = r.update_step_norms[i];
    res->update_step_M_norms[i] = r.update_step_M_norms[i];
    res->gain_ratios[i] = r.gain_ratios[i];
  }
  res->accepted = accepted;
}

// ------------------------------------------------------------------------------------------------
// STPCG on a diagonal Hessian (tests/IterativeSolvers_unit_test.cpp:86-130,138-310).
//   mode 0: tagged device callables -> fused HIP loop;  mode 1: plain lambdas -> generic loop whose
//   every Vector operator runs on the GPU.  Multiplier = Vector as in the reference's tests.
// ------------------------------------------------------------------------------------------------
extern "C" int hd_stpcg_diag(size_t n, const double *g, const double *D, const double *Minv, double Delta,
                             size_t max_iterations, double kappa, double theta, int mode, double *s_out,
                             double *M_norm, size_t *iterations) {
  HD_GUARD_BEGIN
  Context ctx(0);
  FusionSnap fusion_snap(ctx);
  DeviceVector gd(ctx, g, n), Dd(ctx, D, n);
  mi_op *op = nullptr;
  MI355::check(mi_op_create_diag(ctx.get(), Dd.handle(), &op));
  mi_precon *pc = nullptr;
  std::optional<DeviceVector> Mi;
  if (Minv) {
    Mi = DeviceVector(ctx, Minv, n);
    MI355::check(mi_precon_create_diag(ctx.get(), Mi->handle(), &pc));
  }
  LA::SymmetricLinearOperator<DeviceVector> H;
  LA::InnerProduct<DeviceVector> ip;
  std::optional<LA::STPCGPreconditioner<DeviceVector, DeviceVector>> P;
  if (mode == 0) {
    H = MI355::DeviceOperator{op};
    ip = MI355::FrobeniusInnerProduct{};
    if (pc) P = MI355::DeviceSTPCGPreconditioner<DeviceVector>{pc};
  } else {
    H = [op](const DeviceVector &v) { return MI355::DeviceOperator{op}(v); };
    ip = [](const DeviceVector &a, const DeviceVector &b) { return a.dot(b); };
    if (pc)
      P = [pc](const DeviceVector &v) -> std::pair<DeviceVector, DeviceVector> {
        return MI355::DeviceSTPCGPreconditioner<DeviceVector>{pc}(v);
      };
  }
  double mn = 0;
  size_t it = 0;
  DeviceVector s = LA::STPCG<DeviceVector, DeviceVector>(gd, H, ip, mn, it, Delta, max_iterations, kappa, theta, P);
  const std::vector<double> sh = s.to_host();
  std::memcpy(s_out, sh.data(), n * sizeof(double));
  *M_norm = mn;
  *iterations = it;
  mi_op_destroy(op);
  if (pc) mi_precon_destroy(pc);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// STPCG on DeviceVector with a user function that stops at iteration `stop_at` (IterativeSolvers.h:365-369): the
// device callables are the tagged ones (which would select the fused solver), the user function forces the generic
// loop -- every statement of it still runs on the GPU.
// ------------------------------------------------------------------------------------------------
extern "C" int hd_stpcg_diag_stop(size_t n, const double *g, const double *D, const double *Minv, double Delta,
                                  size_t max_iterations, double kappa, double theta, size_t stop_at, double *s_out,
                                  double *M_norm, size_t *iterations, size_t *calls) {
  HD_GUARD_BEGIN
  Context ctx(0);
  DeviceVector gd(ctx, g, n), Dd(ctx, D, n);
  mi_op *op = nullptr;
  MI355::check(mi_op_create_diag(ctx.get(), Dd.handle(), &op));
  mi_precon *pc = nullptr;
  std::optional<DeviceVector> Mi;
  if (Minv) {
    Mi = DeviceVector(ctx, Minv, n);
    MI355::check(mi_precon_create_diag(ctx.get(), Mi->handle(), &pc));
  }
  LA::SymmetricLinearOperator<DeviceVector> H = MI355::DeviceOperator{op};
  LA::InnerProduct<DeviceVector> ip = MI355::FrobeniusInnerProduct{};
  std::optional<LA::STPCGPreconditioner<DeviceVector, std::nullptr_t>> P;
  if (pc) P = MI355::DeviceSTPCGPreconditioner<std::nullptr_t>{pc};
  size_t ncalls = 0;
  std::optional<LA::STPCGUserFunction<DeviceVector, std::nullptr_t>> uf =
      LA::STPCGUserFunction<DeviceVector, std::nullptr_t>(
          [&](size_t k, const DeviceVector &, const LA::SymmetricLinearOperator<DeviceVector> &,
              const std::optional<LA::STPCGPreconditioner<DeviceVector, std::nullptr_t>> &,
              const std::optional<LA::LinearOperator<std::nullptr_t, DeviceVector>> &, const DeviceVector &,
              const DeviceVector &, const DeviceVector &, const DeviceVector &, double) {
            ++ncalls;
            return k == stop_at;
          });
  double mn = 0;
  size_t it = 0;
  const std::optional<LA::LinearOperator<std::nullptr_t, DeviceVector>> At_none;
  DeviceVector s = LA::STPCG<DeviceVector, std::nullptr_t>(gd, H, ip, mn, it, Delta, max_iterations, kappa, theta, P,
                                                           At_none, uf);
  const std::vector<double> sh = s.to_host();
  std::memcpy(s_out, sh.data(), n * sizeof(double));
  *M_norm = mn;
  *iterations = it;
  *calls = ncalls;
  mi_op_destroy(op);
  if (pc) mi_precon_destroy(pc);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// Projected STPCG on DeviceVector (the `At` + constraint-preconditioner branch, IterativeSolvers.h:229-253,
// 381-405; the reference's cases tests/IterativeSolvers_unit_test.cpp:316-496): diagonal Hessian as a device
// operator, Frobenius inner product, Multiplier = DeviceVector; the user's constraint preconditioner and A'
// do their dense algebra on the host (oracle/kkt_dense.h, the same code as the reference-side driver) and hand
// device vectors back -- the solver's own work (every vector statement, the r -= A'lambda correction) is on the GPU.
// ------------------------------------------------------------------------------------------------
// mode 0: as described above (generic loop, host KKT algebra);  mode 1: the DEVICE constraint preconditioner
// (mi_precon_create_constraint: S = A M^-1 A' formed, factored and inverted on the GPU) as tagged P / At callables ->
// the whole projected solve runs through the fused loop (mi_stpcg with constraint_At);  mode 2: the same device
// callables hidden in plain lambdas -> generic loop, device KKT algebra.
static size_t g_last_kkt_inner = 0;
static double g_last_kkt_worst = 0;
// inner CG of the sparse constraint preconditioner during the last hd_stpcg_projected call: iterations of its last
// application, and the largest relative residual any application ended with
extern "C" size_t hd_last_kkt_inner() { return g_last_kkt_inner; }
extern "C" double hd_last_kkt_worst_residual() { return g_last_kkt_worst; }

extern "C" int hd_stpcg_projected(size_t n, size_t m, const double *g, const double *Pdiag, const double *Mdiag,
                                  const double *A, double Delta, size_t max_iterations, double kappa_fgr,
                                  double theta, int mode, double *s_out, double *M_norm, size_t *iterations) {
  HD_GUARD_BEGIN
  Context ctx(0);
  DeviceVector gd(ctx, g, n), Pd(ctx, Pdiag, n);
  mi_op *op = nullptr;
  MI355::check(mi_op_create_diag(ctx.get(), Pd.handle(), &op));
  LA::SymmetricLinearOperator<DeviceVector> H = MI355::DeviceOperator{op};
  LA::InnerProduct<DeviceVector> ip = MI355::FrobeniusInnerProduct{};
  if (mode != 0) {
    std::vector<double> mi(n);
    for (size_t i = 0; i < n; ++i) mi[i] = 1.0 / Mdiag[i];
    DeviceVector Ad(ctx, A, n * m), Mi(ctx, mi);
    mi_precon *kkt = nullptr;
    if (mode == 3) {  // the SPARSE form of the device KKT object: CSR of A's non-zeros, S l = b by the in-kernel CG
      std::vector<int32_t> rp(m + 1, 0), cl;
      std::vector<double> vl;
      for (size_t a = 0; a < m; ++a) {
        for (size_t j = 0; j < n; ++j)
          if (A[a * n + j] != 0.0) {
            cl.push_back((int32_t)j);
            vl.push_back(A[a * n + j]);
          }
        rp[a + 1] = (int32_t)cl.size();
      }
      MI355::check(mi_precon_create_constraint_csr(ctx.get(), n, m, rp.data(), cl.data(), vl.data(), Mi.handle(), 0.0, 0,
                                                   &kkt));
    } else {
      MI355::check(mi_precon_create_constraint(ctx.get(), n, m, Ad.handle(), Mi.handle(), &kkt));
    }
    const MI355::DeviceConstraintPreconditioner cp{kkt, m};
    const MI355::DeviceConstraintTranspose ct{kkt, n};
    std::optional<LA::STPCGPreconditioner<DeviceVector, DeviceVector>> P = LA::STPCGPreconditioner<DeviceVector, DeviceVector>(cp);
    std::optional<LA::LinearOperator<DeviceVector, DeviceVector>> At = LA::LinearOperator<DeviceVector, DeviceVector>(ct);
    if (mode == 2) {
      P = LA::STPCGPreconditioner<DeviceVector, DeviceVector>([cp](const DeviceVector &r) { return cp(r); });
      At = LA::LinearOperator<DeviceVector, DeviceVector>([ct](const DeviceVector &l) { return ct(l); });
    }
    double mn = 0;
    size_t it = 0, c0 = 0, c1 = 0;
    MI355::check(mi_ctx_sync_count(ctx.get(), &c0));
    DeviceVector s = LA::STPCG<DeviceVector, DeviceVector>(gd, H, ip, mn, it, Delta, max_iterations, kappa_fgr, theta,
                                                           P, At);
    MI355::check(mi_ctx_sync_count(ctx.get(), &c1));
    g_last_tnt_syncs = c1 - c0;
    const std::vector<double> sh = s.to_host();
    std::memcpy(s_out, sh.data(), n * sizeof(double));
    *M_norm = mn;
    *iterations = it;
    g_last_kkt_inner = 0;
    g_last_kkt_worst = 0;
    MI355::check(mi_precon_constraint_info(kkt, &g_last_kkt_inner, nullptr, &g_last_kkt_worst, nullptr));
    mi_precon_destroy(kkt);
    mi_op_destroy(op);
    return 0;
  }
  const KktDense K(n, m, A, Mdiag);
  std::optional<LA::STPCGPreconditioner<DeviceVector, DeviceVector>> P =
      LA::STPCGPreconditioner<DeviceVector, DeviceVector>([&](const DeviceVector &r) {
        const std::vector<double> rh = r.to_host();
        std::vector<double> x(n), l(m);
        K.solve(rh.data(), x.data(), l.data());
        return std::make_pair(DeviceVector(ctx, x.data(), n), DeviceVector(ctx, l.data(), m));
      });
  std::optional<LA::LinearOperator<DeviceVector, DeviceVector>> At =
      LA::LinearOperator<DeviceVector, DeviceVector>([&](const DeviceVector &l) {
        const std::vector<double> lh = l.to_host();
        std::vector<double> out(n);
        K.At(lh.data(), out.data());
        return DeviceVector(ctx, out.data(), n);
      });
  double mn = 0;
  size_t it = 0;
  DeviceVector s = LA::STPCG<DeviceVector, DeviceVector>(gd, H, ip, mn, it, Delta, max_iterations, kappa_fgr, theta,
                                                         P, At);
  const std::vector<double> sh = s.to_host();
  std::memcpy(s_out, sh.data(), n * sizeof(double));
  *M_norm = mn;
  *iterations = it;
  mi_op_destroy(op);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// TNT on St(n,p): f(X) = 1/2 tr(X'AX) (BASELINE cfg2).  mode 0: tagged callables (fused inner loop);
// mode 1: the same callables hidden inside plain lambdas (generic path, all on the GPU).
// ------------------------------------------------------------------------------------------------
extern "C" int hd_tnt_stiefel(size_t n, int p, const int32_t *rowptr, const int32_t *col, const double *val,
                              const double *X0, const orc_tnt_params *params, int mode, orc_tnt_result *res) {
  HD_GUARD_BEGIN
  Context ctx(0);
  FusionSnap fusion_snap(ctx);
  MI355::StiefelRayleighQuotient prob(ctx, n, p, rowptr, col, val);
  DeviceVector x0(ctx, X0, n * (size_t)p);
  RM::TNTParams<double> tp;
  fill_params(tp, params);
  size_t accepted = 0;
  std::optional<RM::TNTUserFunction<DeviceVector, DeviceVector>> uf =
      [&](size_t, double, const DeviceVector &, double, const DeviceVector &,
          const RM::LinearOperator<DeviceVector, DeviceVector> &, double, size_t, const DeviceVector &, double,
          double, bool acc) {
        accepted += acc;
        return false;
      };
  Objective<DeviceVector> f = prob.objective();
  RM::QuadraticModel<DeviceVector, DeviceVector> QM = prob.quadratic_model();
  RM::RiemannianMetric<DeviceVector, DeviceVector> metric = prob.metric();
  RM::Retraction<DeviceVector, DeviceVector> retract = prob.retraction();
  if (mode == 2) retract = prob.plain_retraction();  // tagged model and metric, but no fused trial step
  if (mode == 3) {
    // a client's own objective on top of the problem's (here: + 1) with the problem's tagged model and retraction:
    // TNT must call THIS f (the reference always calls the supplied f), i.e. keep the statement sequence
    auto ft = prob.objective();
    f = [ft](const DeviceVector &X) { return ft(X) + 1.0; };
  }
  if (mode == 4) {
    // r06: ONLY the Hessian the model returns is wrapped in a lambda of the client's -- what a
    // logger or a penalty term does, and what the reference's own adapter lambdas (TNT.h:400-426) look like.  Metric,
    // objective and retraction stay tagged.  The run must give the same answers; the fusion counters must say that it
    // ran on the generic side of the boundary.
    auto QMt = prob.quadratic_model();
    QM = [QMt](const DeviceVector &X, DeviceVector &g, RM::LinearOperator<DeviceVector, DeviceVector> &Hs) {
      RM::LinearOperator<DeviceVector, DeviceVector> tagged;
      QMt(X, g, tagged);
      Hs = [tagged](const DeviceVector &Y, const DeviceVector &V) { return tagged(Y, V); };
    };
  }
  if (mode == 1) {  // hide the tags
    auto QMt = prob.quadratic_model();
    QM = [QMt](const DeviceVector &X, DeviceVector &g, RM::LinearOperator<DeviceVector, DeviceVector> &Hs) {
      RM::LinearOperator<DeviceVector, DeviceVector> tagged;
      QMt(X, g, tagged);
      Hs = [tagged](const DeviceVector &Y, const DeviceVector &V) { return tagged(Y, V); };
    };
    metric = [](const DeviceVector &, const DeviceVector &a, const DeviceVector &b) { return a.dot(b); };
  }
  size_t s0 = 0, s1 = 0;
  MI355::check(mi_ctx_sync_count(ctx.get(), &s0));
  for (int rep = 1; rep < g_tnt_repeats; ++rep)
    (void)RM::TNT<DeviceVector, DeviceVector>(f, QM, metric, retract, x0,
                                              std::optional<RM::LinearOperator<DeviceVector, DeviceVector>>(), tp, uf);
  accepted = 0;
  MI355::check(mi_ctx_sync_count(ctx.get(), &s0));
  const auto wall0 = std::chrono::steady_clock::now();
  RM::TNTResult<DeviceVector, double> r =
      RM::TNT<DeviceVector, DeviceVector>(f, QM, metric, retract, x0,
                                          std::optional<RM::LinearOperator<DeviceVector, DeviceVector>>(), tp, uf);
  g_last_tnt_wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
  MI355::check(mi_ctx_sync_count(ctx.get(), &s1));
  g_last_tnt_syncs = s1 - s0;
  export_result(r, accepted, res);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// TNT on SO(3)^N chordal rotation averaging (BASELINE cfg3), optional 3x3 block-Jacobi preconditioner.
// ------------------------------------------------------------------------------------------------
extern "C" int hd_tnt_so3n(size_t N, size_t E, const int32_t *ei, const int32_t *ej, const double *Rt,
                           const double *w, const double *R0, const orc_tnt_params *params, int with_precon,
                           orc_tnt_result *res) {
  HD_GUARD_BEGIN
  Context ctx(0);
  FusionSnap fusion_snap(ctx);
  MI355::RotationAveraging prob(ctx, N, E, ei, ej, Rt, w);
  DeviceVector x0(ctx, R0, 9 * N);
  RM::TNTParams<double> tp;
  fill_params(tp, params);
  size_t accepted = 0;
  std::optional<RM::TNTUserFunction<DeviceVector, DeviceVector>> uf =
      [&](size_t, double, const DeviceVector &, double, const DeviceVector &,
          const RM::LinearOperator<DeviceVector, DeviceVector> &, double, size_t, const DeviceVector &, double,
          double, bool acc) {
        accepted += acc;
        return false;
      };
  // with_precon: bit 0 = 3x3 block-Jacobi preconditioner, bit 1 = plain (untagged) retraction: no fused trial step
  std::optional<RM::LinearOperator<DeviceVector, DeviceVector>> pc;
  if (with_precon & 1) pc = prob.preconditioner();
  RM::Retraction<DeviceVector, DeviceVector> retract = prob.retraction();
  if (with_precon & 2) retract = prob.plain_retraction();
  size_t s0 = 0, s1 = 0;
  MI355::check(mi_ctx_sync_count(ctx.get(), &s0));
  for (int rep = 1; rep < g_tnt_repeats; ++rep)
    (void)RM::TNT<DeviceVector, DeviceVector>(prob.objective(), prob.quadratic_model(), prob.metric(), retract, x0, pc,
                                              tp, uf);
  accepted = 0;
  MI355::check(mi_ctx_sync_count(ctx.get(), &s0));
  const auto wall0 = std::chrono::steady_clock::now();
  RM::TNTResult<DeviceVector, double> r = RM::TNT<DeviceVector, DeviceVector>(
      prob.objective(), prob.quadratic_model(), prob.metric(), retract, x0, pc, tp, uf);
  g_last_tnt_wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
  MI355::check(mi_ctx_sync_count(ctx.get(), &s1));
  g_last_tnt_syncs = s1 - s0;
  export_result(r, accepted, res);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// The reference's sphere problem (tests/TNT_unit_test.cpp:63-122) with DeviceVector and the
// extra-argument pack Args = {DeviceVector} (the fixed point P), written with Vector operators only.
// ------------------------------------------------------------------------------------------------
namespace {
DeviceVector sphere_project(const DeviceVector &X, const DeviceVector &V) { return V - X.dot(V) * X; }
struct SphereProblem {
  Objective<DeviceVector, double, DeviceVector> F;
  RM::VectorField<DeviceVector, DeviceVector, DeviceVector> gradF;
  RM::LinearOperatorConstructor<DeviceVector, DeviceVector, DeviceVector> HessCon;
  RM::RiemannianMetric<DeviceVector, DeviceVector, double, DeviceVector> metric;
  RM::Retraction<DeviceVector, DeviceVector, DeviceVector> retract;
  RM::LinearOperator<DeviceVector, DeviceVector, DeviceVector> precon;
  SphereProblem(const Context &ctx) {
    F = [](const DeviceVector &X, DeviceVector &P) { return (X - P).squaredNorm(); };
    gradF = [](const DeviceVector &X, DeviceVector &P) { return sphere_project(X, 2 * (X - P)); };
    auto gradFc = gradF;
    HessCon = [gradFc](const DeviceVector &, DeviceVector &) {
      RM::LinearOperator<DeviceVector, DeviceVector, DeviceVector> Hs =
          [gradFc](const DeviceVector &X, const DeviceVector &Xdot, DeviceVector &P) -> DeviceVector {
        return sphere_project(X, 2 * Xdot) - X.dot(gradFc(X, P)) * Xdot;
      };
      return Hs;
    };
    metric = [](const DeviceVector &, const DeviceVector &a, const DeviceVector &b, DeviceVector &) {
      return a.dot(b);
    };
    retract = [](const DeviceVector &X, const DeviceVector &V, DeviceVector &) {
      DeviceVector Y = X + V;
      return Y / Y.norm();
    };
    // preconditioner diag(1,2,3) (tests/TNT_unit_test.cpp:111-117) as a device operator
    auto keep = std::make_shared<DeviceVector>(ctx, std::vector<double>{1.0, 2.0, 3.0});
    mi_op *raw = nullptr;
    MI355::check(mi_op_create_diag(ctx.get(), keep->handle(), &raw));
    std::shared_ptr<mi_op> op2(raw, [](mi_op *o) { mi_op_destroy(o); });
    precon = [keep, op2](const DeviceVector &, const DeviceVector &V, DeviceVector &) {
      return MI355::DeviceOperator{op2.get()}(V);
    };
  }
};
}  // namespace

extern "C" int hd_tnt_sphere(int with_precon, const double *x0, const orc_tnt_params *params,
                             orc_tnt_result *res) {
  HD_GUARD_BEGIN
  Context ctx(0);
  SphereProblem sp(ctx);
  DeviceVector P(ctx, std::vector<double>{0.0, 0.0, 1.0});
  DeviceVector X0(ctx, x0, 3);
  RM::TNTParams<double> tp;
  fill_params(tp, params);
  size_t accepted = 0;
  std::optional<RM::TNTUserFunction<DeviceVector, DeviceVector, double, DeviceVector>> uf =
      [&](size_t, double, const DeviceVector &, double, const DeviceVector &,
          const RM::LinearOperator<DeviceVector, DeviceVector, DeviceVector> &, double, size_t,
          const DeviceVector &, double, double, bool acc, DeviceVector &) {
        accepted += acc;
        return false;
      };
  std::optional<RM::LinearOperator<DeviceVector, DeviceVector, DeviceVector>> pc;
  if (with_precon) pc = sp.precon;
  RM::TNTResult<DeviceVector, double> r = RM::TNT<DeviceVector, DeviceVector, double, DeviceVector>(
      sp.F, sp.gradF, sp.HessCon, sp.metric, sp.retract, X0, P, pc, tp, uf);
  export_result(r, accepted, res);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// BASELINE cfg1: chained Rosenbrock (n = 100) through EuclideanTNT<DeviceVector> (reference
// Riemannian/TNT.h:757-805).  The user callables evaluate f, grad f and the tridiagonal Hessian product on the HOST
// from a downloaded copy of x (the same statements as the oracle's problem definition, oracle/problems.c --
// user code may do what it likes), everything TNT and STPCG do with the vectors runs on the GPU.
//   mode 0: Hessian handed over as a device operator (tridiagonal CSR rebuilt at every outer iterate)
//   mode 1: Hessian product as a plain host lambda (download v, multiply, upload)
// ------------------------------------------------------------------------------------------------
extern "C" int hd_tnt_rosenbrock(size_t n, int precon_kind, const double *x0h, const orc_tnt_params *params, int mode,
                                 orc_tnt_result *res) {
  HD_GUARD_BEGIN
  Context ctx(0);
  DeviceVector x0(ctx, x0h, n);
  RM::TNTParams<double> tp;
  fill_params(tp, params);
  size_t accepted = 0;
  Objective<DeviceVector> f = [n](const DeviceVector &X) {
    const std::vector<double> x = X.to_host();
    double s = 0;
    for (size_t i = 0; i + 1 < n; ++i) {
      const double a = 1 - x[i], b = x[i + 1] - x[i] * x[i];
      s += a * a + 100 * b * b;
    }
    return s;
  };
  struct Tri {  // the Hessian at one iterate: CSR on the device, kept alive by the operator that uses it
    mi_csr *A = nullptr;
    mi_op *op = nullptr;
    ~Tri() {
      if (op) mi_op_destroy(op);
      if (A) mi_csr_destroy(A);
    }
  };
  std::shared_ptr<Tri> current;  // the Hessian of the latest model (TNT drops the previous one when it asks again)
  RM::EuclideanQuadraticModel<DeviceVector> QM = [n, mode, &ctx, &current](
                                                     const DeviceVector &X, DeviceVector &g,
                                                     RM::EuclideanLinearOperator<DeviceVector> &Hs) {
    auto xh = std::make_shared<std::vector<double>>(X.to_host());
    const std::vector<double> &x = *xh;
    std::vector<double> gh(n, 0.0);
    for (size_t i = 0; i + 1 < n; ++i) {
      const double b = x[i + 1] - x[i] * x[i];
      gh[i] += -2 * (1 - x[i]) - 400 * x[i] * b;
      gh[i + 1] += 200 * b;
    }
    g = DeviceVector(ctx, gh.data(), n);
    if (mode == 1) {
      Hs = [xh, n, &ctx](const DeviceVector &, const DeviceVector &V) {
        const std::vector<double> &x = *xh;
        const std::vector<double> v = V.to_host();
        std::vector<double> hv(n, 0.0);
        for (size_t i = 0; i + 1 < n; ++i) {
          const double dii = 2 + 1200 * x[i] * x[i] - 400 * x[i + 1];
          const double off = -400 * x[i];
          hv[i] += dii * v[i] + off * v[i + 1];
          hv[i + 1] += off * v[i] + 200 * v[i + 1];
        }
        return DeviceVector(ctx, hv.data(), n);
      };
      return;
    }
    // tridiagonal CSR: row i = [off_{i-1}, (200 if i > 0) + dii (if i < n-1), off_i]
    std::vector<int32_t> rp(n + 1, 0), ci;
    std::vector<double> va;
    for (size_t i = 0; i < n; ++i) {
      double d = 0;
      if (i > 0) {
        ci.push_back((int32_t)(i - 1));
        va.push_back(-400 * x[i - 1]);
        d += 200;
      }
      if (i + 1 < n) d += 2 + 1200 * x[i] * x[i] - 400 * x[i + 1];
      ci.push_back((int32_t)i);
      va.push_back(d);
      if (i + 1 < n) {
        ci.push_back((int32_t)(i + 1));
        va.push_back(-400 * x[i]);
      }
      rp[i + 1] = (int32_t)ci.size();
    }
    auto tri = std::make_shared<Tri>();
    MI355::check(mi_csr_create(ctx.get(), n, ci.size(), rp.data(), ci.data(), va.data(), &tri->A));
    MI355::check(mi_op_create_csr(ctx.get(), tri->A, 1, &tri->op));
    Hs = MI355::DeviceHessian{tri->op};
    current = tri;
  };
  std::optional<RM::EuclideanLinearOperator<DeviceVector>> pc;
  if (precon_kind) {
    pc = [n, &ctx](const DeviceVector &X, const DeviceVector &V) {
      const std::vector<double> x = X.to_host(), v = V.to_host();
      std::vector<double> pv(n);
      for (size_t i = 0; i < n; ++i) pv[i] = v[i] / (std::fabs(2 + 1200 * x[i] * x[i]) + 200);
      return DeviceVector(ctx, pv.data(), n);
    };
  }
  std::optional<RM::EuclideanTNTUserFunction<DeviceVector>> uf =
      [&](size_t, double, const DeviceVector &, double, const DeviceVector &,
          const RM::EuclideanLinearOperator<DeviceVector> &, double, size_t, const DeviceVector &, double, double,
          bool acc) {
        accepted += acc;
        return false;
      };
  RM::TNTResult<DeviceVector, double> r = RM::EuclideanTNT<DeviceVector>(f, QM, x0, pc, tp, uf);
  export_result(r, accepted, res);
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// LOBPCG (tests/LOBPCG_unit_test.cpp:106-225): diagonal operators A, B, T on column-major device
// panels, or a sparse A (BASELINE cfg5).  Adiag/Bdiag/Tdiag may be null; if rowptr != null the
// operator A is the CSR matrix instead of diag(Adiag).  X0 == null -> the random-X0 overload.
// ------------------------------------------------------------------------------------------------
extern "C" int hd_gaussian_probe(size_t m, size_t nx, double *out) {
  HD_GUARD_BEGIN
  Context ctx(0);
  MI355::DeviceMatrix like(ctx, 1, 1);
  MI355::DeviceMatrix Om = MI355::gaussian_probe(like, m, nx);
  const std::vector<double> h = Om.to_host();
  std::memcpy(out, h.data(), h.size() * sizeof(double));
  HD_GUARD_END
}

static std::vector<std::chrono::steady_clock::time_point> g_lobpcg_stamps;
// per-iteration Ritz values / residual norms of the last hd_lobpcg call (nx per iteration, as the user function saw them)
static std::vector<double> g_lobpcg_theta_trace, g_lobpcg_r_trace;
extern "C" size_t hd_lobpcg_trace(double *theta, double *r, size_t cap) {
  const size_t n = g_lobpcg_theta_trace.size() < cap ? g_lobpcg_theta_trace.size() : cap;
  if (theta) std::memcpy(theta, g_lobpcg_theta_trace.data(), n * sizeof(double));
  if (r) std::memcpy(r, g_lobpcg_r_trace.data(), n * sizeof(double));
  return g_lobpcg_theta_trace.size();
}

extern "C" int hd_lobpcg(size_t m, size_t nx, size_t nev, const double *Adiag, const int32_t *rowptr,
                         const int32_t *col, const double *val, const double *Bdiag, const double *Tdiag,
                         const double *X0, size_t max_iters, double tau, double *Theta_out, double *X_out,
                         size_t *num_iters, size_t *nc_out, double *resid_out) {
  HD_GUARD_BEGIN
  using MI355::DeviceMatrix;
  using MI355::HostVectorD;
  using Op = LA::SymmetricLinearOperator<DeviceMatrix>;
  Context ctx(0);
  MI355::make_current(ctx);
  auto diag_op = [&](const double *d) -> Op {
    auto dv = std::make_shared<DeviceVector>(ctx, d, m);
    mi_ctx *c = ctx.get();
    return [dv, c, m](const DeviceMatrix &X) {
      DeviceMatrix Y(c, m, X.cols());
      MI355::check(mi_panel_rowscale(c, m, (int)X.cols(), dv->handle(), X.handle(), Y.handle()));
      return Y;
    };
  };
  Op A;
  mi_csr *csr = nullptr;
  if (rowptr) {
    MI355::check(mi_csr_create(ctx.get(), m, (size_t)rowptr[m], rowptr, col, val, &csr));
    // the tagged sparse operator (MI355/Matrix.h): the loop fuses A(X) with the residual and its norms; with
    // HD_LOBPCG_PLAIN_OPERATOR=1 the same product hidden in a plain lambda (the reference's statement sequence)
    const char *plain = getenv("HD_LOBPCG_PLAIN_OPERATOR");
    if (plain && plain[0] == '1') {
      mi_ctx *c = ctx.get();
      A = [csr, c, m](const DeviceMatrix &X) {
        DeviceMatrix Y(c, m, X.cols());
        MI355::check(mi_csr_spmm_colmajor(csr, (int)X.cols(), X.handle(), Y.handle()));
        return Y;
      };
    } else {
      A = MI355::DeviceCsrPanelOperator{csr};
    }
  } else {
    A = diag_op(Adiag);
  }
  std::optional<Op> B, T;
  if (Bdiag) B = diag_op(Bdiag);
  if (Tdiag) T = diag_op(Tdiag);
  size_t iters = 0, nc = 0;
  std::vector<double> resid;
  g_lobpcg_theta_trace.clear();
  g_lobpcg_r_trace.clear();
  std::optional<LA::LOBPCGUserFunction<HostVectorD, DeviceMatrix>> uf =
      [&](size_t, const Op &, const std::optional<Op> &, const std::optional<Op> &, size_t, const HostVectorD &th,
          const DeviceMatrix &, const HostVectorD &r, size_t) {
        resid.assign(r.data(), r.data() + r.size());
        g_lobpcg_theta_trace.insert(g_lobpcg_theta_trace.end(), th.data(), th.data() + th.size());
        g_lobpcg_r_trace.insert(g_lobpcg_r_trace.end(), r.data(), r.data() + r.size());
        g_lobpcg_stamps.push_back(std::chrono::steady_clock::now());
        return false;
      };
  g_lobpcg_stamps.clear();
  std::pair<HostVectorD, DeviceMatrix> out;
  if (X0) {
    DeviceMatrix X0d(ctx, m, nx, X0);
    out = LA::LOBPCG<HostVectorD, DeviceMatrix>(A, B, T, X0d, nev, max_iters, iters, nc, tau, uf);
  } else {
    out = LA::LOBPCG<HostVectorD, DeviceMatrix>(A, B, T, m, nx, nev, max_iters, iters, nc, tau, uf);
  }
  for (size_t i = 0; i < nev; ++i) Theta_out[i] = out.first(i);
  const std::vector<double> xh = out.second.to_host();
  std::memcpy(X_out, xh.data(), xh.size() * sizeof(double));
  *num_iters = iters;
  *nc_out = nc;
  if (resid_out)
    for (size_t i = 0; i < resid.size() && i < nx; ++i) resid_out[i] = resid[i];
  if (csr) mi_csr_destroy(csr);
  HD_GUARD_END
}

// LOBPCG on a context and a CSR matrix made by the caller -- e.g. a rank of a communicator with its row shard of A
// (8(e): panels row-sharded, Gram and residual norms all-reduced, Rayleigh-Ritz replicated).  m = local rows.
// COLLECTIVE when the context carries a communicator.
extern "C" int hd_lobpcg_on(void *ctx_handle, void *csr_handle, size_t m, size_t nx, size_t nev, const double *X0,
                            size_t max_iters, double tau, double *Theta_out, double *X_out, size_t *num_iters,
                            size_t *nc_out, double *resid_out) {
  HD_GUARD_BEGIN
  using MI355::DeviceMatrix;
  using MI355::HostVectorD;
  using Op = LA::SymmetricLinearOperator<DeviceMatrix>;
  Context ctx = Context::adopt(static_cast<mi_ctx *>(ctx_handle));
  MI355::make_current(ctx);
  mi_ctx *c = ctx.get();
  const mi_csr *csr = static_cast<const mi_csr *>(csr_handle);
  Op A = [csr, c, m](const DeviceMatrix &X) {
    DeviceMatrix Y(c, m, X.cols());
    MI355::check(mi_csr_spmm_colmajor(csr, (int)X.cols(), X.handle(), Y.handle()));
    return Y;
  };
  size_t iters = 0, nc = 0;
  std::vector<double> resid;
  std::optional<LA::LOBPCGUserFunction<HostVectorD, DeviceMatrix>> uf =
      [&](size_t, const Op &, const std::optional<Op> &, const std::optional<Op> &, size_t, const HostVectorD &,
          const DeviceMatrix &, const HostVectorD &r, size_t) {
        resid.assign(r.data(), r.data() + r.size());
        return false;
      };
  DeviceMatrix X0d(ctx, m, nx, X0);
  std::pair<HostVectorD, DeviceMatrix> out = LA::LOBPCG<HostVectorD, DeviceMatrix>(
      A, std::optional<Op>(), std::optional<Op>(), X0d, nev, max_iters, iters, nc, tau, uf);
  for (size_t i = 0; i < nev; ++i) Theta_out[i] = out.first(i);
  const std::vector<double> xh = out.second.to_host();
  std::memcpy(X_out, xh.data(), xh.size() * sizeof(double));
  *num_iters = iters;
  *nc_out = nc;
  if (resid_out)
    for (size_t i = 0; i < resid.size() && i < nx; ++i) resid_out[i] = resid[i];
  HD_GUARD_END
}

// mean wall time of one LOBPCG iteration of the last hd_lobpcg call, seconds (user-function to user-function,
// first interval dropped); 0 if fewer than 3 iterations ran
extern "C" double hd_lobpcg_seconds_per_iteration() {
  if (g_lobpcg_stamps.size() < 3) return 0.0;
  const std::chrono::duration<double> d = g_lobpcg_stamps.back() - g_lobpcg_stamps[1];
  return d.count() / (double)(g_lobpcg_stamps.size() - 2);
}

// RiemannianGradientDescentSphere (tests/GradientDescent_unit_test.cpp:76-130) on the device
extern "C" int hd_gd_sphere(const double *x0, double *x_out, double *f_out, double *gradnorm_out,
                            int *status_out, size_t *iterations_out, size_t cap, double *objective_values,
                            size_t *linesearch_iterations) {
  HD_GUARD_BEGIN
  Context ctx(0);
  SphereProblem sp(ctx);
  DeviceVector P(ctx, std::vector<double>{0.0, 0.0, 1.0});
  DeviceVector X0(ctx, x0, 3);
  RM::GradientDescentParams<double> gp;
  gp.gradient_tolerance = 1e-6;
  gp.relative_decrease_tolerance = 0;
  gp.stepsize_tolerance = 0;
  gp.max_iterations = 1000000;
  RM::GradientDescentResult<DeviceVector, double> r = RM::GradientDescent<DeviceVector, DeviceVector, double, DeviceVector>(
      sp.F, sp.gradF, sp.metric, sp.retract, X0, P, gp);
  const std::vector<double> x = r.x.to_host();
  std::memcpy(x_out, x.data(), 3 * sizeof(double));
  *f_out = r.f;
  *gradnorm_out = r.gradfx_norm;
  *status_out = static_cast<int>(r.status);
  *iterations_out = r.linesearch_iterations.size();
  for (size_t i = 0; i < r.objective_values.size() && i < cap; ++i) objective_values[i] = r.objective_values[i];
  for (size_t i = 0; i < r.linesearch_iterations.size() && i < cap; ++i)
    linesearch_iterations[i] = r.linesearch_iterations[i];
  HD_GUARD_END
}

// GradientDescent<DeviceVector, DeviceVector> on f(X) = 1/2 tr(X'AX) over St(n,p).
//   mode 0: tagged retraction + Frobenius metric -> each Armijo trial is one fused launch chain with one
//           read-back (mi_stiefel_rq_armijo_trial);  mode 1: plain retraction -> the reference's statement sequence
// Per-iteration objective values and line-search counts (capacity cap), and the host<->device synchronisations.
extern "C" int hd_gd_stiefel(size_t n, int p, const int32_t *rowptr, const int32_t *col, const double *val,
                             const double *X0, size_t max_iterations, double gradient_tolerance, double alpha,
                             double beta, double sigma, size_t max_ls_iterations, int mode, double *x_out,
                             double *f_out, double *gradnorm_out, int *status_out, size_t *iterations_out, size_t cap,
                             double *objective_values, size_t *linesearch_iterations, size_t *syncs_out) {
  HD_GUARD_BEGIN
  Context ctx(0);
  FusionSnap fusion_snap(ctx);
  MI355::StiefelRayleighQuotient prob(ctx, n, p, rowptr, col, val);
  DeviceVector x0(ctx, X0, n * (size_t)p);
  RM::GradientDescentParams<double> gp;
  gp.max_iterations = max_iterations;
  gp.gradient_tolerance = gradient_tolerance;
  gp.relative_decrease_tolerance = 0;
  gp.stepsize_tolerance = 0;
  gp.alpha = alpha;
  gp.beta = beta;
  gp.sigma = sigma;
  gp.max_ls_iterations = max_ls_iterations;
  Objective<DeviceVector> f = prob.objective();
  RM::VectorField<DeviceVector, DeviceVector> grad = prob.gradient();
  RM::RiemannianMetric<DeviceVector, DeviceVector> metric = prob.metric();
  RM::Retraction<DeviceVector, DeviceVector> retract = prob.retraction();
  if (mode == 1) retract = prob.plain_retraction();
  size_t s0 = 0, s1 = 0;
  MI355::check(mi_ctx_sync_count(ctx.get(), &s0));
  RM::GradientDescentResult<DeviceVector, double> r =
      RM::GradientDescent<DeviceVector, DeviceVector, double>(f, grad, metric, retract, x0, gp);
  MI355::check(mi_ctx_sync_count(ctx.get(), &s1));
  *syncs_out = s1 - s0;
  const std::vector<double> x = r.x.to_host();
  std::memcpy(x_out, x.data(), x.size() * sizeof(double));
  *f_out = r.f;
  *gradnorm_out = r.gradfx_norm;
  *status_out = static_cast<int>(r.status);
  *iterations_out = r.linesearch_iterations.size();
  for (size_t i = 0; i < r.objective_values.size() && i < cap; ++i) objective_values[i] = r.objective_values[i];
  for (size_t i = 0; i < r.linesearch_iterations.size() && i < cap; ++i)
    linesearch_iterations[i] = r.linesearch_iterations[i];
  HD_GUARD_END
}

// ------------------------------------------------------------------------------------------------
// LSQR (IterativeSolvers.h:552-855) and TNLS (TNLS.h:265-729) on DeviceVector: the generic loops of the
// drop-in headers running through the Vector concept on the GPU.  A and A' are two CSR matrices.
// ------------------------------------------------------------------------------------------------
namespace {
struct CsrPair {
  mi_csr *A = nullptr, *At = nullptr;
  CsrPair(Context &ctx, size_t n, const int32_t *rp, const int32_t *cl, const double *vl, const int32_t *rpt,
          const int32_t *clt, const double *vlt) {
    MI355::check(mi_csr_create(ctx.get(), n, (size_t)rp[n], rp, cl, vl, &A));
    MI355::check(mi_csr_create(ctx.get(), n, (size_t)rpt[n], rpt, clt, vlt, &At));
  }
  ~CsrPair() {
    mi_csr_destroy(A);
    mi_csr_destroy(At);
  }
  static DeviceVector apply(const mi_csr *M, const DeviceVector &x) {
    DeviceVector y = DeviceVector::like(x);
    MI355::check(mi_csr_spmm(M, 1, x.handle(), y.handle()));
    return y;
  }
};
}  // namespace

// mode 0: operators / inner products as the tagged device callables => the template hands the solve to the
// fused mi_lsqr; mode 1: plain lambdas => the generic loop of the template through the Vector concept
extern "C" int hd_lsqr_csr(size_t n, const int32_t *rp, const int32_t *cl, const double *vl, const int32_t *rpt,
                           const int32_t *clt, const double *vlt, const double *b, size_t max_iterations,
                           double lambda, double btol, double Atol, double Acond_limit, double Delta, int mode,
                           double *x_out, double *xnorm_out, size_t *iterations_out) {
  HD_GUARD_BEGIN
  Context ctx(0);
  FusionSnap fusion_snap(ctx);
  CsrPair M(ctx, n, rp, cl, vl, rpt, clt, vlt);
  LA::LinearOperator<DeviceVector, DeviceVector> Aop, Atop;
  LA::InnerProduct<DeviceVector, double> ip;
  mi_op *opA = nullptr, *opAt = nullptr;
  if (mode == 0) {
    MI355::check(mi_op_create_csr(ctx.get(), M.A, 1, &opA));
    MI355::check(mi_op_create_csr(ctx.get(), M.At, 1, &opAt));
    Aop = MI355::DeviceOperator{opA};
    Atop = MI355::DeviceOperator{opAt};
    ip = MI355::FrobeniusInnerProduct{};
  } else {
    Aop = [&](const DeviceVector &x) { return CsrPair::apply(M.A, x); };
    Atop = [&](const DeviceVector &y) { return CsrPair::apply(M.At, y); };
    ip = [](const DeviceVector &a, const DeviceVector &c) { return a.dot(c); };
  }
  DeviceVector bv(ctx, b, n);
  double xnorm = 0;
  size_t iters = 0;
  (void)bv.dot(bv);  // drain uploads before the clock starts
  const auto t0 = std::chrono::steady_clock::now();
  DeviceVector x = LA::LSQR<DeviceVector, double>(Aop, Atop, bv, ip, xnorm, iters, max_iterations, lambda, btol, Atol,
                                                 Acond_limit, Delta);
  (void)x.dot(x);    // ... and the solve before it stops
  g_last_solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const std::vector<double> xh = x.to_host();
  for (size_t j = 0; j < n; ++j) x_out[j] = j < xh.size() ? xh[j] : 0.0;
  *xnorm_out = xnorm;
  *iterations_out = iters;
  mi_op_destroy(opA);
  mi_op_destroy(opAt);
  HD_GUARD_END
}

// mode 0: Jacobian pair / metric / inner product as tagged device callables => TNLS hands every inner solve to the
// fused mi_lsqr; mode 1: plain lambdas => generic loops
extern "C" int hd_tnls_affine(size_t n, const int32_t *rp, const int32_t *cl, const double *vl, const int32_t *rpt,
                              const int32_t *clt, const double *vlt, const double *b, const double *x0,
                              double root_tolerance, double gradient_tolerance, size_t max_iterations,
                              size_t max_LSQR_iterations, int mode, double *x_out, double *f_out,
                              double *gradnorm_out, int *status_out, size_t *outer_out, size_t *inner_total_out) {
  HD_GUARD_BEGIN
  Context ctx(0);
  FusionSnap fusion_snap(ctx);
  CsrPair M(ctx, n, rp, cl, vl, rpt, clt, vlt);
  DeviceVector bv(ctx, b, n);
  mi_op *opA = nullptr, *opAt = nullptr;
  MI355::check(mi_op_create_csr(ctx.get(), M.A, 1, &opA));
  MI355::check(mi_op_create_csr(ctx.get(), M.At, 1, &opAt));
  RM::Mapping<DeviceVector, DeviceVector> F = [&](const DeviceVector &x) { return CsrPair::apply(M.A, x) - bv; };
  RM::JacobianPairFunction<DeviceVector, DeviceVector, DeviceVector> J = [&](const DeviceVector &) {
    RM::Jacobian<DeviceVector, DeviceVector, DeviceVector> dF;
    RM::JacobianAdjoint<DeviceVector, DeviceVector, DeviceVector> dFt;
    if (mode == 0) {
      dF = MI355::DeviceHessian{opA};
      dFt = MI355::DeviceHessian{opAt};
    } else {
      dF = [&](const DeviceVector &, const DeviceVector &v) { return CsrPair::apply(M.A, v); };
      dFt = [&](const DeviceVector &, const DeviceVector &w) { return CsrPair::apply(M.At, w); };
    }
    return std::make_pair(dF, dFt);
  };
  RM::RiemannianMetric<DeviceVector, DeviceVector, double> metric;
  LA::InnerProduct<DeviceVector, double> ipY;
  if (mode == 0) {
    metric = MI355::FrobeniusMetric{};
    ipY = MI355::FrobeniusInnerProduct{};
  } else {
    metric = [](const DeviceVector &, const DeviceVector &a, const DeviceVector &c) { return a.dot(c); };
    ipY = [](const DeviceVector &a, const DeviceVector &c) { return a.dot(c); };
  }
  RM::Retraction<DeviceVector, DeviceVector> retract = [](const DeviceVector &x, const DeviceVector &v) {
    return x + v;
  };
  RM::TNLSParams<double> p;
  p.relative_decrease_tolerance = 0;
  p.stepsize_tolerance = 0;
  p.gradient_tolerance = gradient_tolerance;
  p.root_tolerance = root_tolerance;
  p.max_iterations = max_iterations;
  p.max_LSQR_iterations = max_LSQR_iterations;
  DeviceVector xs(ctx, x0, n);
  RM::TNLSResult<DeviceVector, double> r = RM::TNLS<DeviceVector, DeviceVector, DeviceVector, double>(
      F, J, metric, ipY, retract, xs, std::optional<RM::TNLSPreconditioner<DeviceVector, DeviceVector>>(), p);
  const std::vector<double> xh = r.x.to_host();
  std::memcpy(x_out, xh.data(), n * sizeof(double));
  *f_out = r.f;
  *gradnorm_out = r.gradfx_norm;
  *status_out = static_cast<int>(r.status);
  *outer_out = r.inner_iterations.size();
  size_t tot = 0;
  for (size_t k : r.inner_iterations) tot += k;
  *inner_total_out = tot;
  mi_op_destroy(opA);
  mi_op_destroy(opAt);
  HD_GUARD_END
}
