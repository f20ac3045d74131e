// Optimization/Riemannian/TNLS.h -- drop-in for the reference header of the same path: Riemannian
// truncated-Newton trust-region method for nonlinear least squares  min_x |F(x)|, with trust-region
// LSQR on the Jacobian as sub-problem solver.
//
//   reference: include/Optimization/Riemannian/TNLS.h
//              TNLSPreconditioner :60-63, TNLSUserFunction :94-101, TNLSParams :107-169,
//              TNLSStatus :173-203, TNLSResult :207-226, TNLS :265-729, EuclideanTNLS :747-765
//
// MI355X build, written from scratch against that interface.  Generic over the variable / tangent /
// residual types; with MI355::DeviceVector all vector work runs on the GPU through the Vector
// operators (LSQR: two operator applications, 3-5 norms and 4 AXPY-type updates per pass).
// As in the reference, the Jacobian pair function is typed WITHOUT the Args pack (reference :269), so
// the template is usable with Args = {} only.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/Concepts.h"
#include "Optimization/Util/Stopwatch.h"

namespace Optimization {
namespace Riemannian {

// Right preconditioner (M, M') applied as  A = dF_x o M,  A' = M' o dF_x^*       (reference :60-63)
template <typename VariableX, typename TangentX, typename... Args>
using TNLSPreconditioner =
    std::pair<LinearOperator<VariableX, TangentX, Args...>, LinearOperator<VariableX, TangentX, Args...>>;

// Observer called once per outer iteration before the step is applied; true stops   (reference :94-101)
template <typename VariableX, typename TangentX, typename VectorY, typename Scalar = double, typename... Args>
using TNLSUserFunction = std::function<bool(
    size_t i, double t, const VariableX &x, VectorY Fx,
    const Jacobian<VariableX, TangentX, VectorY, Args...> &gradFx,
    const JacobianAdjoint<VariableX, TangentX, VectorY, Args...> &gradFxT, Scalar Delta,
    size_t num_LSQR_iters, const TangentX &h, Scalar dL, Scalar rho, bool accepted, Args &...args)>;

template <typename Scalar = double>
struct TNLSParams : public SmoothOptimizerParams<Scalar> {  // reference :107-169
  Scalar Delta0 = 1;
  Scalar eta1 = .05;
  Scalar eta2 = .9;
  Scalar alpha1 = .25;
  Scalar alpha2 = 2.5;
  size_t max_LSQR_iterations = 1000;
  Scalar kappa_fgr = .1;      // LSQR btol = min(|F|^theta, kappa_fgr)
  Scalar theta = .5;
  Scalar lambda = 0;          // Tikhonov damping of the LSQR sub-problem
  Scalar Atol = 1e-6;
  Scalar Acond_limit = 1e8;
  Scalar root_tolerance = 1e-6;   // on |F(x)|
  Scalar Delta_tolerance = 1e-6;
};

enum class TNLSStatus {  // reference :173-203
  Root,
  Gradient,
  RelativeDecrease,
  Stepsize,
  TrustRegion,
  IterationLimit,
  ElapsedTime,
  UserFunction
};

template <typename Variable, typename Scalar = double>
struct TNLSResult : public SmoothOptimizerResult<Variable, Scalar> {  // reference :207-226
  TNLSStatus status;
  std::vector<size_t> inner_iterations;
  std::vector<Scalar> rho;
  std::vector<Scalar> trust_region_radius;
};

template <typename VariableX, typename TangentX, typename VectorY, typename Scalar = double, typename... Args>
TNLSResult<VariableX, Scalar>
TNLS(const Mapping<VariableX, VectorY, Args...> &F, const JacobianPairFunction<VariableX, TangentX, VectorY> &J,
     const RiemannianMetric<VariableX, TangentX, Scalar, Args...> &metric_X,
     const LinearAlgebra::InnerProduct<VectorY, Scalar, Args...> &inner_product_Y,
     const Retraction<VariableX, TangentX, Args...> &retract_X, const VariableX &x0, Args &...args,
     const std::optional<TNLSPreconditioner<VariableX, TangentX, Args...>> &precon = std::nullopt,
     const TNLSParams<Scalar> &params = TNLSParams<Scalar>(),
     const std::optional<TNLSUserFunction<VariableX, TangentX, VectorY, Scalar, Args...>> &user_function =
         std::nullopt) {
  auto require = [](bool ok, const char *msg) {
    if (!ok) throw std::invalid_argument(msg);
  };
  // reference :286-352
  require(!(params.max_computation_time < 0), "Maximum computation time must be a nonnegative real value");
  require(!(params.root_tolerance < 0), "Root tolerance must be a nonnegative real value");
  require(!(params.gradient_tolerance < 0), "Gradient tolerance must be a nonnegative real value");
  require(!(params.relative_decrease_tolerance < 0), "Relative decrease tolerance must be a nonnegative real value");
  require(!(params.stepsize_tolerance < 0), "Stepsize tolerance must be a nonnegative real value");
  require(!(params.Delta_tolerance < 0), "Trust-region radius tolerance must be a nonnegative real value");
  require(!(params.Delta0 <= 0), "Initial trust-region radius must be a positive real value");
  require(!(params.eta1 <= 0 || params.eta1 >= 1),
          "Threshold on gain ratio for a successful iteration (eta1) must satisfy 0 < eta1 < 1");
  require(!(params.eta1 > params.eta2 || params.eta2 >= 1),
          "Threshold on gain ratio for a very successful iteration (eta2) must satisfy eta1 <= eta2 < 1");
  require(!(params.alpha1 <= 0 || params.alpha1 >= 1),
          "Multiplicative factor for decreasing trust-region radius (alpha1) must satisfy 0 < alpha1 < 1");
  require(!(params.alpha2 <= 1),
          "Multiplicative factor for increasing trust-region radius (alpha1) must satisfy alpha2 > 1");
  require(!(params.kappa_fgr <= 0 || params.kappa_fgr >= 1),
          "Target relative decrease in predicted residual for inexact update step computation (kappa_fgr) "
          "must satisfy 0 < kappa_fgr < 1");
  require(!(params.theta < 0),
          "Target superlinear convergence rate parameter (theta) must be a nonnegative real number");
  require(!(params.Atol < 0), "Relative norm stopping tolerance Atol must be a nonnegative real number");
  require(!(params.Acond_limit <= 0), "Stopping criterion Acond_limit must be a positive real number");

  namespace LA = Optimization::LinearAlgebra;
  const Scalar sqrt_eps = sqrt(std::numeric_limits<Scalar>::epsilon());

  TNLSResult<VariableX, Scalar> result;
  result.status = TNLSStatus::IterationLimit;

  VariableX x, x_trial;
  VectorY Fx, Fx_trial;
  Scalar F_norm, F_sq, F_trial_norm, F_trial_sq;
  Scalar Delta;
  TangentX h;
  Scalar h_norm = 0, h_M_norm = 0, relative_decrease = 0;
  TangentX gradL;   // gradient of L(x) = |F(x)|
  Scalar gradL_norm;
  Jacobian<VariableX, TangentX, VectorY, Args...> dF;
  JacobianAdjoint<VariableX, TangentX, VectorY, Args...> dFt;

  const size_t outer_width = floor(log10(params.max_iterations)) + 1;
  const size_t inner_width = floor(log10(params.max_LSQR_iterations)) + 1;

  // linearise at x: Jacobian pair and grad L = dF' F / |F|                       :414-426, :634-639
  auto linearise = [&]() {
    std::tie(dF, dFt) = J(x, args...);
    gradL = dFt(x, Fx, args...) / F_norm;
    gradL_norm = sqrt(metric_X(x, gradL, gradL, args...));
  };

  x = x0;
  Fx = F(x, args...);
  F_sq = inner_product_Y(Fx, Fx, args...);
  F_norm = sqrt(F_sq);
  linearise();

  // operators of the LSQR sub-problem, following x / dF as they change           :432-462
  LA::LinearOperator<TangentX, VectorY, Args...> A_generic;
  LA::LinearOperator<VectorY, TangentX, Args...> At_generic;
  if (precon) {
    A_generic = [&x, &dF, &precon](const TangentX &v, Args &...a) -> VectorY {
      return dF(x, precon->first(x, v, a...), a...);
    };
    At_generic = [&x, &dFt, &precon](const VectorY &w, Args &...a) -> TangentX {
      return precon->second(x, dFt(x, w, a...), a...);
    };
  } else {
    A_generic = [&x, &dF](const TangentX &v, Args &...a) -> VectorY { return dF(x, v, a...); };
    At_generic = [&x, &dFt](const VectorY &w, Args &...a) -> TangentX { return dFt(x, w, a...); };
  }
  const LA::InnerProduct<TangentX, Scalar, Args...> inner_product_X_generic =
      [&x, &metric_X](const TangentX &a, const TangentX &b, Args &...aa) -> Scalar {
    return metric_X(x, a, b, aa...);
  };
  LA::LinearOperator<TangentX, VectorY, Args...> A = A_generic;
  LA::LinearOperator<VectorY, TangentX, Args...> At = At_generic;
  LA::InnerProduct<TangentX, Scalar, Args...> inner_product_X = inner_product_X_generic;

#if OPTIMIZATION_HAVE_MI355
  // Device fast path: when the Jacobian pair handed back by J consists of tagged device operators (an
  // mi_op each, bound to x: MI355::DeviceHessian) and the metric is the Frobenius one, LSQR receives the
  // tagged callables it recognises and the whole inner solve runs in the fused mi_lsqr.  Redone after
  // every linearisation because J may hand back new operators.
  constexpr bool device_types = MI355::is_device_vector<TangentX>::value &&
                                MI355::is_device_vector<VectorY>::value && sizeof...(Args) == 0;
  // (with a right preconditioner whose two halves are tagged device operators as well, A = dF o M and A' = M' o dF^*
  // (:432-447) are composed on the device -- mi_op_create_compose -- so that the preconditioned solve stays fused too)
  std::shared_ptr<mi_op> composed_A, composed_At;
  auto retag_for_device = [&]() {
    if constexpr (device_types) {
      // back to the generic views first (J may hand back tagged device operators at one iterate and plain callables
      // at the next: a stale tagged view would point at an operator bound to an old base point)
      A = A_generic;
      At = At_generic;
      inner_product_X = inner_product_X_generic;
      composed_A.reset();
      composed_At.reset();
      const auto *a = dF.template target<MI355::DeviceHessian>();
      const auto *at = dFt.template target<MI355::DeviceHessian>();
      if (!a || !at || !a->op || !at->op || !metric_X.template target<MI355::FrobeniusMetric>()) return;
      if (precon) {
        const auto *m = precon->first.template target<MI355::DeviceHessian>();
        const auto *mt = precon->second.template target<MI355::DeviceHessian>();
        if (!m || !mt || !m->op || !mt->op) return;
        mi_op *ca = nullptr, *cat = nullptr;
        MI355::check(mi_op_create_compose(x.context(), a->op, m->op, &ca));
        composed_A.reset(ca, [](mi_op *o) { mi_op_destroy(o); });
        MI355::check(mi_op_create_compose(x.context(), mt->op, at->op, &cat));
        composed_At.reset(cat, [](mi_op *o) { mi_op_destroy(o); });
        A = MI355::DeviceOperator{ca};
        At = MI355::DeviceOperator{cat};
      } else {
        A = MI355::DeviceOperator{a->op};
        At = MI355::DeviceOperator{at->op};
      }
      inner_product_X = MI355::FrobeniusInnerProduct{};
    }
  };
  retag_for_device();
#else
  auto retag_for_device = []() {};
#endif

  Delta = params.Delta0;
  if (params.verbose) {
    std::cout << std::scientific;
    std::cout.precision(params.precision);
    std::cout << "Truncated-Newton trust-region nonlinear least-squares optimization: " << std::endl
              << std::endl;
  }

  const auto clock_start = Stopwatch::tick();
  for (size_t iteration = 0; iteration < params.max_iterations; ++iteration) {  // :478
    const double elapsed = Stopwatch::tock(clock_start);
    if (elapsed > params.max_computation_time) {
      result.status = TNLSStatus::ElapsedTime;
      break;
    }
    result.time.push_back(elapsed);
    result.objective_values.push_back(F_norm);
    result.gradient_norms.push_back(gradL_norm);
    result.trust_region_radius.push_back(Delta);
    if (params.log_iterates) result.iterates.push_back(x);

    if (params.verbose) {
      std::cout << "Iter: ";
      std::cout.width(outer_width);
      std::cout << iteration << ", time: " << elapsed << ", |F|: ";
      std::cout.width(params.precision + 7);
      std::cout << F_norm << ", |g|: " << gradL_norm;
    }

    if (F_norm < params.root_tolerance) {  // :508
      result.status = TNLSStatus::Root;
      break;
    }
    if (gradL_norm < params.gradient_tolerance) {  // :513
      result.status = TNLSStatus::Gradient;
      break;
    }

    // inexact Gauss-Newton step from trust-region LSQR                           :525-537
    const Scalar etak = std::min(std::pow(F_norm, params.theta), params.kappa_fgr);
    size_t inner_iterations;
    h = LA::LSQR<TangentX, VectorY, Scalar, Args...>(A, At, -Fx, inner_product_X, inner_product_Y, args...,
                                                     h_M_norm, inner_iterations, params.max_LSQR_iterations,
                                                     params.lambda, etak, params.Atol, params.Acond_limit,
                                                     Delta);
    if (precon) h = precon->first(x, h, args...);
    h_norm = sqrt(metric_X(x, h, h, args...));

    if (params.verbose) {
      std::cout << ", Delta: " << Delta << ", inner iters: ";
      std::cout.width(inner_width);
      std::cout << inner_iterations << ", |h|: " << h_norm;
    }

    x_trial = retract_X(x, h, args...);                       // :552
    Fx_trial = F(x_trial, args...);
    F_trial_sq = inner_product_Y(Fx_trial, Fx_trial, args...);
    F_trial_norm = sqrt(F_trial_sq);

    // gain ratio on SQUARED norms: actual vs. linearised decrease                 :565-583
    const VectorY lin = dF(x, h, args...) + Fx;
    const Scalar lin_sq = inner_product_Y(lin, lin, args...);
    const Scalar dq = F_sq - lin_sq;
    const Scalar dL = F_norm - F_trial_norm;
    const Scalar df2 = F_sq - F_trial_sq;
    relative_decrease = dL / (sqrt_eps + F_norm);
    const Scalar rho = df2 / dq;

    if (params.verbose) {
      std::cout << ", dL: " << dL << ", rho: " << rho << ". ";
    }
    const bool accepted = (!std::isnan(rho) && rho > params.eta1);  // :594
    if (params.verbose) std::cout << (accepted ? "Step accepted" : "Step REJECTED!");

    result.inner_iterations.push_back(inner_iterations);
    result.update_step_norms.push_back(h_norm);
    result.rho.push_back(rho);

    if (user_function && (*user_function)(iteration, elapsed, x, Fx, dF, dFt, Delta, inner_iterations, h, dL,
                                          rho, accepted, args...)) {  // :606-614
      result.status = TNLSStatus::UserFunction;
      break;
    }

    if (accepted) {  // :617-640
      x = std::move(x_trial);
      Fx = std::move(Fx_trial);
      F_sq = F_trial_sq;
      F_norm = F_trial_norm;
      if (relative_decrease < params.relative_decrease_tolerance) {
        result.status = TNLSStatus::RelativeDecrease;
        break;
      }
      if (h_norm < params.stepsize_tolerance) {
        result.status = TNLSStatus::Stepsize;
        break;
      }
      linearise();
      retag_for_device();
    }

    if ((!std::isnan(rho)) && (rho >= params.eta2)) {  // :644-657
      Delta = std::max<Scalar>(params.alpha2 * h_M_norm, Delta);
    } else if (std::isnan(rho) || (rho < params.eta1)) {
      Delta = params.alpha1 * h_M_norm;
      if (Delta < params.Delta_tolerance) {
        result.status = TNLSStatus::TrustRegion;
        break;
      }
    }
    if (params.verbose) std::cout << std::endl;
  }

  result.elapsed_time = Stopwatch::tock(clock_start);
  result.x = x;
  result.f = F_norm;
  result.gradfx_norm = gradL_norm;

  if (params.verbose) {
    std::cout << std::endl << std::endl << "Optimization finished!" << std::endl;
    switch (result.status) {
      case TNLSStatus::Root:
        std::cout << "Found root! (Residual norm: " << F_norm << ")" << std::endl;
        break;
      case TNLSStatus::Gradient:
        std::cout << "Found first-order critical point! (Gradient norm: " << gradL_norm << ")" << std::endl;
        break;
      case TNLSStatus::RelativeDecrease:
        std::cout << "Algorithm terminated due to insufficient relative decrease: " << relative_decrease
                  << " < " << params.relative_decrease_tolerance << std::endl;
        break;
      case TNLSStatus::Stepsize:
        std::cout << "Algorithm terminated due to excessively small step size: |h| = " << h_norm << " < "
                  << params.stepsize_tolerance << std::endl;
        break;
      case TNLSStatus::TrustRegion:
        std::cout << "Algorithm terminated due to excessively small trust region radius: " << Delta << " < "
                  << params.Delta_tolerance << std::endl;
        break;
      case TNLSStatus::IterationLimit:
        std::cout << "Algorithm exceeded maximum number of outer iterations" << std::endl;
        break;
      case TNLSStatus::ElapsedTime:
        std::cout << "Algorithm exceeded maximum allowed computation time: (" << result.elapsed_time << " > "
                  << params.max_computation_time << " seconds)" << std::endl;
        break;
      case TNLSStatus::UserFunction:
        std::cout << "Algorithm terminated due to user-supplied stopping criterion" << std::endl;
        break;
    }
    std::cout << "Final residual norm: " << result.f << std::endl;
    std::cout << "Norm of Riemannian gradient: " << result.gradfx_norm << std::endl;
    std::cout << "Total elapsed computation time: " << result.elapsed_time << " seconds" << std::endl
              << std::endl;
    std::cout << std::defaultfloat;
    std::cout.precision(6);
  }
  return result;
}

// flat metric, dot product on the residual space, R_X(V) = X + V                (reference :747-765)
template <typename Vector, typename Scalar = double, typename... Args>
TNLSResult<Vector, Scalar>
EuclideanTNLS(const Mapping<Vector, Vector, Args...> &F, const JacobianPairFunction<Vector, Vector, Vector> &J,
              const Vector &x0, Args &...args,
              const std::optional<TNLSPreconditioner<Vector, Vector, Args...>> &precon = std::nullopt,
              const TNLSParams<Scalar> &params = TNLSParams<Scalar>(),
              const std::optional<TNLSUserFunction<Vector, Vector, Vector, Scalar, Args...>> &user_function =
                  std::nullopt) {
  const RiemannianMetric<Vector, Vector, Scalar, Args...> metric = EuclideanMetric<Vector, Scalar, Args...>;
  const LinearAlgebra::InnerProduct<Vector, Scalar, Args...> ip = EuclideanInnerProduct<Vector, Scalar, Args...>;
  const Retraction<Vector, Vector, Args...> retract = EuclideanRetraction<Vector, Args...>;
  return TNLS<Vector, Vector, Vector, Scalar, Args...>(F, J, metric, ip, retract, x0, args..., precon, params,
                                                       user_function);
}

}  // namespace Riemannian
}  // namespace Optimization
