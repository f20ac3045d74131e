// Optimization/Riemannian/TNT.h -- drop-in for the reference header of the same path: the Riemannian
// truncated-Newton trust-region method (Conn-Gould-Toint Alg. 6.1.1 on a manifold; Absil-Baker-
// Gallivan), with the Steihaug-Toint truncated preconditioned CG as sub-problem solver.
//
//   reference: include/Optimization/Riemannian/TNT.h
//              TNTUserFunction :64-71, TNTParams :76-130, TNTStatus :134-164, TNTResult :168-194,
//              TNT (quadratic-model form) :242-689, TNT (gradient + Hessian-constructor form) :704-736,
//              EuclideanTNT :757-773 and :778-805
//
// MI355X build, written from scratch against that interface (same names, template parameters,
// argument order, defaults, status values, exceptions, trace layout; the acceptance / radius-update
// arithmetic follows the reference statement by statement -- see the :line tags).  With
// Variable = Tangent = MI355::DeviceVector and the tagged callables of Optimization/MI355/Device.h
// the inner solver is the fused HIP STPCG; otherwise everything runs through the Vector operators.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <iostream>
#include <limits>
#include <optional>
#include <stdexcept>
#include <vector>

#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/Concepts.h"
#include "Optimization/Util/Stopwatch.h"

namespace Optimization {
namespace Riemannian {

// Observer called once per outer iteration, after the step h and its gain ratio are known and
// BEFORE h is applied; returning true stops the optimizer with status UserFunction.  (reference :64-71)
template <typename Variable, typename Tangent, typename Scalar = double, typename... Args>
using TNTUserFunction = std::function<bool(
    size_t i, double t, const Variable &x, Scalar f, const Tangent &g,
    const LinearOperator<Variable, Tangent, Args...> &HessOp, Scalar Delta, size_t num_STPCG_iters,
    const Tangent &h, Scalar df, Scalar rho, bool accepted, Args &...args)>;

template <typename Scalar = double>
struct TNTParams : public SmoothOptimizerParams<Scalar> {  // reference :76-130
  // trust-region control
  Scalar Delta0 = 1;     // initial radius
  Scalar eta1 = .05;     // accept the step if rho > eta1           (0 < eta1 <= eta2)
  Scalar eta2 = .9;      // "very successful" if rho >= eta2        (eta1 <= eta2 < 1)
  Scalar alpha1 = .25;   // radius shrink factor                    (0 < alpha1 < 1)
  Scalar alpha2 = 2.5;   // radius growth factor                    (alpha2 > 1)
  // truncated CG control (Trust-Region Methods, Sec. 7.5.1)
  size_t max_TPCG_iterations = 1000;
  Scalar kappa_fgr = .1;  // residual reduction target
  Scalar theta = .5;      // superlinear-rate exponent
  // extra stopping rules
  Scalar preconditioned_gradient_tolerance = 1e-6;  // on |M^-1 g|
  Scalar Delta_tolerance = 1e-6;                    // on the radius
};

enum class TNTStatus {  // reference :134-164 -- the order is part of the interface
  Gradient,
  PreconditionedGradient,
  RelativeDecrease,
  Stepsize,
  TrustRegion,
  IterationLimit,
  ElapsedTime,
  UserFunction
};

template <typename Variable, typename Scalar = double>
struct TNTResult : public SmoothOptimizerResult<Variable, Scalar> {  // reference :168-194
  Scalar preconditioned_grad_f_x_norm;
  TNTStatus status;
  std::vector<Scalar> preconditioned_gradient_norms;
  std::vector<size_t> inner_iterations;     // STPCG passes per outer iteration
  std::vector<Scalar> update_step_M_norms;  // |h|_M per outer iteration
  std::vector<Scalar> gain_ratios;          // rho per outer iteration
  std::vector<Scalar> trust_region_radius;  // radius at the START of each iteration (+ final)
};

namespace detail {
inline void tnt_require(bool ok, const char *msg) {
  if (!ok) throw std::invalid_argument(msg);
}
}  // namespace detail

template <typename Variable, typename Tangent, typename Scalar = double, typename... Args>
TNTResult<Variable, Scalar>
TNT(const Objective<Variable, Scalar, Args...> &f, const QuadraticModel<Variable, Tangent, Args...> &QM,
    const RiemannianMetric<Variable, Tangent, Scalar, Args...> &metric,
    const Retraction<Variable, Tangent, Args...> &retract, const Variable &x0, Args &...args,
    const std::optional<LinearOperator<Variable, Tangent, Args...>> &precon = std::nullopt,
    const TNTParams<Scalar> &params = TNTParams<Scalar>(),
    const std::optional<TNTUserFunction<Variable, Tangent, Scalar, Args...>> &user_function =
        std::nullopt) {
  using detail::tnt_require;
  // parameter checks, reference :260-318 (theta > 1 is rejected later, inside STPCG)
  tnt_require(!(params.max_computation_time < 0), "Maximum computation time must be a nonnegative real value");
  tnt_require(!(params.gradient_tolerance < 0), "Gradient tolerance must be a nonnegative real value");
  tnt_require(!(params.preconditioned_gradient_tolerance < 0),
              "Preconditioned gradient tolerance must be a nonnegative real value");
  tnt_require(!(params.relative_decrease_tolerance < 0),
              "Relative decrease tolerance must be a nonnegative real value");
  tnt_require(!(params.stepsize_tolerance < 0), "Stepsize tolerance must be a nonnegative real value");
  tnt_require(!(params.Delta_tolerance < 0), "Trust-region radius tolerance must be a nonnegative real value");
  tnt_require(!(params.Delta0 <= 0), "Initial trust-region radius must be a positive real value");
  tnt_require(!(params.eta1 <= 0 || params.eta1 >= 1),
              "Threshold on gain ratio for a successful iteration (eta1) must satisfy 0 < eta1 < 1");
  tnt_require(!(params.eta1 > params.eta2 || params.eta2 >= 1),
              "Threshold on gain ratio for a very successful iteration (eta2) must satisfy eta1 <= eta2 < 1");
  tnt_require(!(params.alpha1 <= 0 || params.alpha1 >= 1),
              "Multiplicative factor for decreasing trust-region radius (alpha1) must satisfy 0 < alpha1 < 1");
  tnt_require(!(params.alpha2 <= 1),
              "Multiplicative factor for increasing trust-region radius (alpha1) must satisfy alpha2 > 1");
  tnt_require(!(params.kappa_fgr <= 0 || params.kappa_fgr >= 1),
              "Target relative decrease in predicted residual for inexact update step computation "
              "(kappa_fgr) must satisfy 0 < kappa_fgr < 1");
  tnt_require(!(params.theta < 0),
              "Target superlinear convergence rate parameter (theta) must be a nonnegative real number");

  namespace LA = Optimization::LinearAlgebra;
  using Multiplier = std::nullptr_t;  // unconstrained sub-problems                 :397

  const Scalar sqrt_eps = sqrt(std::numeric_limits<Scalar>::epsilon());  // :323
  TNTResult<Variable, Scalar> result;
  result.status = TNTStatus::IterationLimit;  // :327

  Variable x, x_trial;
  Scalar fx, fx_trial;
  Scalar grad_norm, precon_grad_norm;
  Scalar Delta;
  Scalar h_norm = 0, h_M_norm = 0;
  Scalar relative_decrease = 0;
  Tangent grad;
  LinearOperator<Variable, Tangent, Args...> Hess;

  const size_t outer_width = floor(log10(params.max_iterations)) + 1;          // :356
  const size_t inner_width = floor(log10(params.max_TPCG_iterations)) + 1;    // :359

  result.time.reserve(params.max_iterations + 1);
  result.objective_values.reserve(params.max_iterations + 1);
  result.gradient_norms.reserve(params.max_iterations + 1);
  result.preconditioned_gradient_norms.reserve(params.max_iterations + 1);
  result.trust_region_radius.reserve(params.max_iterations + 1);
  if (params.log_iterates) result.iterates.reserve(params.max_iterations + 1);

  // |g| and |M^-1 g| at the current point                                          :382-392, :575-585
  auto measure_gradient = [&]() {
    grad_norm = sqrt(metric(x, grad, grad, args...));
    if (precon) {
      Tangent Pg = (*precon)(x, grad, args...);
      precon_grad_norm = sqrt(metric(x, Pg, Pg, args...));
    } else {
      precon_grad_norm = grad_norm;
    }
  };

  x = x0;                       // :375
  fx = f(x, args...);           // :377
  QM(x, grad, Hess, args...);   // :380
  measure_gradient();

  // x-free views of the model for the inner solver; they follow x / Hess as those change  :400-426
  const LA::SymmetricLinearOperator<Tangent, Args...> H_generic = [&x, &Hess](const Tangent &v,
                                                                               Args &...a) -> Tangent {
    return Hess(x, v, a...);
  };
  const LA::InnerProduct<Tangent, Scalar, Args...> inner_product_generic =
      [&x, &metric](const Tangent &a, const Tangent &b, Args &...aa) -> Scalar { return metric(x, a, b, aa...); };
  std::optional<LA::STPCGPreconditioner<Tangent, Multiplier, Args...>> Pop_generic;
  if (precon)
    Pop_generic = [&x, &precon](const Tangent &v, Args &...a) -> std::pair<Tangent, Multiplier> {
      return std::pair<Tangent, Multiplier>((*precon)(x, v, a...), Multiplier());
    };
  LA::SymmetricLinearOperator<Tangent, Args...> H = H_generic;
  LA::InnerProduct<Tangent, Scalar, Args...> inner_product = inner_product_generic;
  std::optional<LA::STPCGPreconditioner<Tangent, Multiplier, Args...>> Pop = Pop_generic;

#if OPTIMIZATION_HAVE_MI355
  // Device fast path: swap the generic lambdas for the tagged function objects that STPCG recognises.
  // H and Pop are rebuilt after every QM call because the user's QuadraticModel may hand back a new
  // device operator each time.
  constexpr bool device_types = MI355::is_device_vector<Tangent>::value && sizeof...(Args) == 0;
  auto retag_for_device = [&]() {
    if constexpr (device_types) {
      // back to the generic views first: a QuadraticModel may hand back a tagged device operator at one iterate and
      // a plain callable at the next, and a stale tagged view would point at an operator bound to an old base point
      H = H_generic;
      inner_product = inner_product_generic;
      Pop = Pop_generic;
      if (metric.template target<MI355::FrobeniusMetric>()) inner_product = MI355::FrobeniusInnerProduct{};
      if (const auto *dh = Hess.template target<MI355::DeviceHessian>()) H = MI355::DeviceOperator{dh->op};
      if (precon)
        if (const auto *dp = precon->template target<MI355::DevicePreconditioner>())
          Pop = MI355::DeviceSTPCGPreconditioner<Multiplier>{dp->P};
    }
  };
  retag_for_device();
#else
  auto retag_for_device = []() {};
#endif

  Delta = params.Delta0;  // :429

  if (params.verbose) {
    std::cout << std::scientific;
    std::cout.precision(params.precision);
    std::cout << "Truncated-Newton trust-region optimization: " << std::endl << std::endl;
  }

  const auto clock_start = Stopwatch::tick();  // :440

  for (size_t iteration = 0; iteration < params.max_iterations; ++iteration) {  // :446
    const double elapsed = Stopwatch::tock(clock_start);
    if (elapsed > params.max_computation_time) {
      result.status = TNTStatus::ElapsedTime;
      break;
    }

    // traces: one entry per STARTED iteration                                      :455-462
    result.time.push_back(elapsed);
    result.objective_values.push_back(fx);
    result.gradient_norms.push_back(grad_norm);
    result.preconditioned_gradient_norms.push_back(precon_grad_norm);
    result.trust_region_radius.push_back(Delta);
    if (params.log_iterates) result.iterates.push_back(x);

    if (params.verbose) {
      std::cout << "Iter: ";
      std::cout.width(outer_width);
      std::cout << iteration << ", time: " << elapsed << ", f: ";
      std::cout.width(params.precision + 7);
      std::cout << fx << ", |g|: " << grad_norm << ", |M^{-1}g|: " << precon_grad_norm;
    }

    if (grad_norm < params.gradient_tolerance) {  // :474
      result.status = TNTStatus::Gradient;
      break;
    }
    if (precon_grad_norm < params.preconditioned_gradient_tolerance) {  // :478
      result.status = TNTStatus::PreconditionedGradient;
      break;
    }

    // Device path: a retraction whose owner evaluates the whole trial step (|h|, x_trial, f(x_trial), <g,h>,
    // <h,Hess h> and the gradient norm(s) at x_trial) as one launch chain with one read-back -- :493-512 and :573-585
    // in one call.  Taken only when the objective, the Hessian the model set and the retraction all name the SAME
    // problem object as owner (the chain evaluates the owner's f and Hessian: a wrapped / penalised / logging
    // objective, or callables of two problems, must keep the statement sequence below, which calls what was
    // supplied), with the Frobenius metric.  A preconditioner does not matter to the chain; if it is the owner's,
    // the chain also delivers |M^-1 grad| at the trial point.
    bool step_scalars_batched = false, trial_done = false;
    Scalar gh = 0, hHh = 0, trial_grad_sqnorm = 0, trial_precon_grad_sqnorm = -1;
#if OPTIMIZATION_HAVE_MI355
    const MI355::DeviceTrialRetraction *fused_trial = nullptr;
    bool owners_precon = false;
    if constexpr (device_types) {
      const auto *tr = retract.template target<MI355::DeviceTrialRetraction>();
      const auto *fo = f.template target<MI355::DeviceObjective>();
      const auto *dh = Hess.template target<MI355::DeviceHessian>();
      if (tr && tr->trial && tr->owner && fo && fo->owner == tr->owner && dh && dh->owner == tr->owner &&
          metric.template target<MI355::FrobeniusMetric>())
        fused_trial = tr;
      if (fused_trial && precon)
        if (const auto *dp = precon->template target<MI355::DevicePreconditioner>())
          owners_precon = dp->owner == fused_trial->owner;
      if (!fused_trial && !grad.empty())  // the statement sequence below on device vectors: make it observable
        (void)mi_ctx_note_generic(
            grad.context(), MI_GENERIC_TRIAL,
            !tr ? "the retraction is not a MI355::DeviceTrialRetraction (a plain callable, or a tagged one wrapped in a lambda)"
            : !fo ? "the objective is not the MI355::DeviceObjective of the retraction's problem (a plain callable, or wrapped)"
            : !dh ? "the Hessian the quadratic model returned is not a MI355::DeviceHessian (a plain callable, or wrapped)"
                  : "objective, Hessian and retraction belong to different problems, or the metric is not "
                    "MI355::FrobeniusMetric");
    }
    // with a fused trial step right behind it, the fused inner solve does not wait for the device either: one
    // read-back serves both (MI355::DeferScope)
    MI355::DeferScope defer(fused_trial != nullptr);
#endif

    // inner solve                                                                   :488-493
    size_t inner_iterations;
    Tangent h = LA::STPCG<Tangent, Multiplier, Scalar, Args...>(
        grad, H, inner_product, args..., h_M_norm, inner_iterations, Delta, params.max_TPCG_iterations,
        params.kappa_fgr, params.theta, Pop);
#if OPTIMIZATION_HAVE_MI355
    if constexpr (device_types) {
      if (fused_trial) {
        auto t = fused_trial->trial(x, h, grad, owners_precon);
        x_trial = std::move(t.x_trial);
        fx_trial = t.f_trial;
        h_norm = sqrt(t.hh);
        gh = t.gh;
        hHh = t.hHh;
        trial_grad_sqnorm = t.grad_trial_sqnorm;
        trial_precon_grad_sqnorm = t.precon_grad_trial_sqnorm;
        step_scalars_batched = trial_done = true;
      }
      if (defer.taken()) {  // |h|_M and the pass count of the inner solve (no wait: the trial's read-back was behind it)
        const mi_stpcg_result sr = defer.collect();
        h_M_norm = sr.update_step_M_norm;
        inner_iterations = sr.num_iterations;
      }
      // Without such a retraction (Frobenius metric, tagged Hessian): the three step scalars |h|^2, <g,h>,
      // <h,Hess h> of :496,:511-512 come from ONE pass and ONE synchronisation instead of three (the Hessian
      // application is a pure device operation, so evaluating it before the retraction is unobservable).
      if (!trial_done && metric.template target<MI355::FrobeniusMetric>() &&
          Hess.template target<MI355::DeviceHessian>()) {
        const Tangent Hh = Hess(x, h, args...);
        const std::vector<double> d = MI355::dot_batch({{&h, &h}, {&grad, &h}, {&h, &Hh}});
        h_norm = sqrt(d[0]);
        gh = d[1];
        hHh = d[2];
        step_scalars_batched = true;
      }
    }
#endif
    if (!step_scalars_batched) h_norm = sqrt(metric(x, h, h, args...));

    if (params.verbose) {
      std::cout << ", Delta: " << Delta << ", inner iters: ";
      std::cout.width(inner_width);
      std::cout << inner_iterations << ", |h|: " << h_norm << ", |h|_M: " << h_M_norm;
    }

    if (!trial_done) {
      x_trial = retract(x, h, args...);   // :505
      fx_trial = f(x_trial, args...);     // :508
    }

    // predicted vs. actual decrease                                                 :511-521
    const Scalar dm = step_scalars_batched
                          ? -gh - .5 * hHh
                          : -metric(x, grad, h, args...) - .5 * metric(x, h, Hess(x, h, args...), args...);
    const Scalar df = fx - fx_trial;
    relative_decrease = df / (sqrt_eps + fabs(fx));
    const Scalar rho = df / dm;

    if (params.verbose) {
      std::cout << ", df: ";
      std::cout.width(params.precision + 7);
      std::cout << df << ", rho: ";
      std::cout.width(params.precision + 7);
      std::cout << rho << ". ";
    }

    const bool accepted = (!std::isnan(rho) && rho > params.eta1);  // :532
    if (params.verbose) std::cout << (accepted ? "Step accepted" : "Step REJECTED!");

    result.inner_iterations.push_back(inner_iterations);  // :538-541
    result.update_step_norms.push_back(h_norm);
    result.update_step_M_norms.push_back(h_M_norm);
    result.gain_ratios.push_back(rho);

    if (user_function && (*user_function)(iteration, elapsed, x, fx, grad, Hess, Delta, inner_iterations, h,
                                          df, rho, accepted, args...)) {  // :545-552
      result.status = TNTStatus::UserFunction;
      break;
    }

    if (accepted) {  // :555-587
      x = std::move(x_trial);
      fx = fx_trial;
      if (relative_decrease < params.relative_decrease_tolerance) {
        result.status = TNTStatus::RelativeDecrease;
        break;
      }
      if (h_norm < params.stepsize_tolerance) {
        result.status = TNTStatus::Stepsize;
        break;
      }
      QM(x, grad, Hess, args...);
      retag_for_device();
      if (trial_done) {  // |grad f(x_trial)|^2 came with the trial step                  :575-585
        grad_norm = sqrt(trial_grad_sqnorm);
        if (!precon) {
          precon_grad_norm = grad_norm;
        } else if (trial_precon_grad_sqnorm >= 0) {  // ... and so did |M^-1 grad|^2 (the owner's preconditioner)
          precon_grad_norm = sqrt(trial_precon_grad_sqnorm);
        } else {
          Tangent Pg = (*precon)(x, grad, args...);
          precon_grad_norm = sqrt(metric(x, Pg, Pg, args...));
        }
      } else {
        measure_gradient();
      }
    }

    // radius update: both branches scale |h|_M, not Delta                          :590-603
    if ((!std::isnan(rho)) && (rho >= params.eta2)) {
      Delta = std::max<Scalar>(params.alpha2 * h_M_norm, Delta);
    } else if (std::isnan(rho) || (rho < params.eta1)) {
      Delta = params.alpha1 * h_M_norm;
      if (Delta < params.Delta_tolerance) {
        result.status = TNTStatus::TrustRegion;
        break;
      }
    }

    if (params.verbose) std::cout << std::endl;
  }

  result.elapsed_time = Stopwatch::tock(clock_start);  // :608
  result.x = x;
  result.f = fx;
  result.gradfx_norm = grad_norm;
  result.preconditioned_grad_f_x_norm = precon_grad_norm;
  // the final state is appended once more                                          :617-624
  result.time.push_back(result.elapsed_time);
  result.objective_values.push_back(fx);
  result.gradient_norms.push_back(grad_norm);
  result.preconditioned_gradient_norms.push_back(precon_grad_norm);
  result.trust_region_radius.push_back(Delta);
  if (params.log_iterates) result.iterates.push_back(x);

  if (params.verbose) {  // :626-686
    std::cout << std::endl << std::endl << "Optimization finished!" << std::endl;
    switch (result.status) {
      case TNTStatus::Gradient:
        std::cout << "Found first-order critical point! (Gradient norm: " << grad_norm << ")" << std::endl;
        break;
      case TNTStatus::PreconditionedGradient:
        std::cout << "Found first-order critical point! (Preconditioned gradient norm: " << precon_grad_norm
                  << ")" << std::endl;
        break;
      case TNTStatus::RelativeDecrease:
        std::cout << "Algorithm terminated due to insufficient relative decrease: " << relative_decrease
                  << " < " << params.relative_decrease_tolerance << std::endl;
        break;
      case TNTStatus::Stepsize:
        std::cout << "Algorithm terminated due to excessively small step size: |h| = " << h_norm << " < "
                  << params.stepsize_tolerance << std::endl;
        break;
      case TNTStatus::TrustRegion:
        std::cout << "Algorithm terminated due to excessively small trust region radius: " << Delta << " < "
                  << params.Delta_tolerance << std::endl;
        break;
      case TNTStatus::IterationLimit:
        std::cout << "Algorithm exceeded maximum number of outer iterations" << std::endl;
        break;
      case TNTStatus::ElapsedTime:
        std::cout << "Algorithm exceeded maximum allowed computation time: (" << result.elapsed_time << " > "
                  << params.max_computation_time << " seconds)" << std::endl;
        break;
      case TNTStatus::UserFunction:
        std::cout << "Algorithm terminated due to user-supplied stopping criterion" << std::endl;
        break;
    }
    std::cout << "Final objective value: " << result.f << std::endl;
    std::cout << "Norm of Riemannian gradient: " << result.gradfx_norm << std::endl;
    std::cout << "Norm of preconditioned Riemannian gradient: " << result.preconditioned_grad_f_x_norm
              << std::endl;
    std::cout << "Total elapsed computation time: " << result.elapsed_time << " seconds" << std::endl
              << std::endl;
    std::cout << std::defaultfloat;
    std::cout.precision(6);
  }
  return result;
}

// Same method, model given as separate gradient field and Hessian constructor  (reference :704-736)
template <typename Variable, typename Tangent, typename Scalar = double, typename... Args>
TNTResult<Variable, Scalar>
TNT(const Objective<Variable, Scalar, Args...> &f, const VectorField<Variable, Tangent, Args...> &grad_f,
    const LinearOperatorConstructor<Variable, Tangent, Args...> &HessianConstructor,
    const RiemannianMetric<Variable, Tangent, Scalar, Args...> &metric,
    const Retraction<Variable, Tangent, Args...> &retract, const Variable &x0, Args &...args,
    const std::optional<LinearOperator<Variable, Tangent, Args...>> &precon = std::nullopt,
    const TNTParams<Scalar> &params = TNTParams<Scalar>(),
    const std::optional<TNTUserFunction<Variable, Tangent, Scalar, Args...>> &user_function =
        std::nullopt) {
  QuadraticModel<Variable, Tangent, Args...> QM = [&grad_f, &HessianConstructor](
                                                      const Variable &X, Tangent &g,
                                                      LinearOperator<Variable, Tangent, Args...> &Hs,
                                                      Args &...a) {
    g = grad_f(X, a...);
    Hs = HessianConstructor(X, a...);
  };
  return TNT<Variable, Tangent, Scalar, Args...>(f, QM, metric, retract, x0, args..., precon, params,
                                                 user_function);
}

/// Euclidean conveniences: flat metric, R_X(V) = X + V, one Vector type          (reference :753-805)

template <typename Vector, typename Scalar = double, typename... Args>
using EuclideanTNTUserFunction = TNTUserFunction<Vector, Vector, Scalar, Args...>;

template <typename Vector, typename Scalar = double, typename... Args>
TNTResult<Vector, Scalar>
EuclideanTNT(const Objective<Vector, Scalar, Args...> &f, const EuclideanQuadraticModel<Vector, Args...> &QM,
             const Vector &x0, Args &...args,
             const std::optional<EuclideanLinearOperator<Vector, Args...>> &precon = std::nullopt,
             const TNTParams<Scalar> &params = TNTParams<Scalar>(),
             const std::optional<EuclideanTNTUserFunction<Vector, Scalar, Args...>> &user_function =
                 std::nullopt) {
  // explicit std::function objects (the reference passes the bare function templates, which
  // clang -- hence hipcc -- rejects in front of a parameter pack)
  const RiemannianMetric<Vector, Vector, Scalar, Args...> metric = EuclideanMetric<Vector, Scalar, Args...>;
  const Retraction<Vector, Vector, Args...> retract = EuclideanRetraction<Vector, Args...>;
  return TNT<Vector, Vector, Scalar, Args...>(f, QM, metric, retract, x0, args..., precon, params,
                                              user_function);
}

template <typename Vector, typename Scalar = double, typename... Args>
TNTResult<Vector, Scalar>
EuclideanTNT(const Objective<Vector, Scalar, Args...> &f, const EuclideanVectorField<Vector, Args...> &nabla_f,
             const EuclideanLinearOperatorConstructor<Vector, Args...> &HessianConstructor, const Vector &x0,
             Args &...args, const std::optional<EuclideanLinearOperator<Vector, Args...>> &precon = std::nullopt,
             const TNTParams<Scalar> &params = TNTParams<Scalar>(),
             const std::optional<EuclideanTNTUserFunction<Vector, Scalar, Args...>> &user_function =
                 std::nullopt) {
  EuclideanQuadraticModel<Vector, Args...> QM = [&nabla_f, &HessianConstructor](
                                                    const Vector &X, Vector &g,
                                                    EuclideanLinearOperator<Vector, Args...> &Hs, Args &...a) {
    g = nabla_f(X, a...);
    Hs = HessianConstructor(X, a...);
  };
  return EuclideanTNT<Vector, Scalar, Args...>(f, QM, x0, args..., precon, params, user_function);
}

}  // namespace Riemannian
}  // namespace Optimization
