// Optimization/Riemannian/GradientDescent.h -- drop-in for the reference header of the same path:
// Riemannian gradient descent with Armijo backtracking line search.
//
//   reference: include/Optimization/Riemannian/GradientDescent.h
//              GradientDescentUserFunction :36-40, GradientDescentParams :44-58,
//              GradientDescentStatus :62-85, GradientDescentResult :89-99, GradientDescent :124-398,
//              EuclideanGradientDescent :420-434
//
// MI355X build, written from scratch against that interface.  Generic over Variable / Tangent: with
// MI355::DeviceVector every vector operation (the scaled step `-t grad`, the user's retraction and
// objective) is enqueued on the GPU through the Vector operators.
#pragma once

#include <cmath>
#include <cstddef>
#include <functional>
#include <iostream>
#include <limits>
#include <optional>
#include <stdexcept>
#include <vector>

#include "Optimization/Riemannian/Concepts.h"
#include "Optimization/Util/Stopwatch.h"

#if __has_include("mi355opt.h")
#include "Optimization/MI355/Device.h"
#define OPTIMIZATION_GD_HAVE_MI355 1
#else
#define OPTIMIZATION_GD_HAVE_MI355 0
#endif

namespace Optimization {
namespace Riemannian {

// Observer called after each successful line search, before the step is applied   (reference :36-40)
template <typename Variable, typename Tangent, typename Scalar = double, typename... Args>
using GradientDescentUserFunction = std::function<void(size_t i, double t, const Variable &x, Scalar f,
                                                       const Tangent &g, const Tangent &h, Scalar df,
                                                       Args &...args)>;

template <typename Scalar = double>
struct GradientDescentParams : public SmoothOptimizerParams<Scalar> {  // reference :44-58
  Scalar alpha = 1.0;              // first trial stepsize of every line search
  Scalar beta = .5;                // stepsize shrink factor, in (0,1)
  Scalar sigma = .5;               // Armijo sufficient-decrease fraction, in (0,1)
  size_t max_ls_iterations = 100;  // trial steps per line search
};

enum class GradientDescentStatus {  // reference :62-85
  Gradient,
  RelativeDecrease,
  Stepsize,
  LineSearch,
  IterationLimit,
  ElapsedTime,
};

template <typename Variable, typename Scalar = double>
struct GradientDescentResult : public SmoothOptimizerResult<Variable, Scalar> {  // reference :89-99
  GradientDescentStatus status;
  std::vector<size_t> linesearch_iterations;
};

template <typename Variable, typename Tangent, typename Scalar = double, typename... Args>
GradientDescentResult<Variable, Scalar> GradientDescent(
    const Objective<Variable, Scalar, Args...> &f, const VectorField<Variable, Tangent, Args...> &grad_f,
    const RiemannianMetric<Variable, Tangent, Scalar, Args...> &metric,
    const Retraction<Variable, Tangent, Args...> &retract, const Variable &x0, Args &...args,
    const GradientDescentParams<Scalar> &params = GradientDescentParams<Scalar>(),
    const std::optional<GradientDescentUserFunction<Variable, Tangent, Scalar, Args...>> &user_function =
        std::nullopt) {
  // reference :141-161
  if (params.max_computation_time < 0)
    throw std::invalid_argument("Maximum computation time must be a nonnegative real value");
  if (params.gradient_tolerance < 0)
    throw std::invalid_argument("Gradient tolerance must be a nonnegative real value");
  if (params.alpha <= 0)
    throw std::invalid_argument("Initial stepsize for backtracking line-search must be a positive real value");
  if (params.beta <= 0 || params.beta >= 1)
    throw std::invalid_argument("Multiplicative shrinkage factor for stepsize in backtracking line-search "
                                "must be a value in the range (0, 1)");
  if (params.sigma <= 0 || params.sigma >= 1)
    throw std::invalid_argument("Sufficient fractional decrease parameter for step acceptance in "
                                "backtracking line search must be a value in the range (0, 1)");

  const Scalar sqrt_eps = sqrt(std::numeric_limits<Scalar>::epsilon());
  GradientDescentResult<Variable, Scalar> result;
  result.status = GradientDescentStatus::IterationLimit;

  Variable x = x0, x_trial;
  Scalar fx = f(x, args...), fx_trial = 0;                       // :212-213
  Tangent g = grad_f(x, args...);                                // :216
  Scalar g_norm = sqrt(metric(x, g, g, args...));                // :217
  Tangent h;
  Scalar h_norm = 0, df = 0, relative_decrease = 0;
  size_t ls_iters = 0;

  const size_t outer_width = floor(log10(params.max_iterations)) + 1;
  const size_t ls_width = floor(log10(params.max_ls_iterations)) + 1;
  if (params.verbose) {
    std::cout << std::scientific;
    std::cout.precision(params.precision);
    std::cout << "Gradient descent optimization: " << std::endl << std::endl;
  }

  const auto clock_start = Stopwatch::tick();
  for (size_t iteration = 0; iteration < params.max_iterations; iteration++) {  // :231
    const double elapsed = Stopwatch::tock(clock_start);
    if (elapsed > params.max_computation_time) {
      result.status = GradientDescentStatus::ElapsedTime;
      break;
    }
    result.time.push_back(elapsed);
    result.objective_values.push_back(fx);
    result.gradient_norms.push_back(g_norm);
    if (params.log_iterates) result.iterates.push_back(x);

    if (params.verbose) {
      std::cout << "Iter: ";
      std::cout.width(outer_width);
      std::cout << iteration << ", time: " << elapsed << ", f: ";
      std::cout.width(params.precision + 7);
      std::cout << fx << ", |g|: " << g_norm;
    }

    if (g_norm < params.gradient_tolerance) {  // :259
      result.status = GradientDescentStatus::Gradient;
      break;
    }

    // Armijo backtracking: t = alpha, alpha beta, alpha beta^2, ...               :266-286
    Scalar t = params.alpha / params.beta;
    ls_iters = 0;
    bool sufficient = false;
    // Device path: a tagged retraction (MI355::DeviceTrialRetraction, Frobenius metric) evaluates a whole Armijo
    // trial -- h = -t g, the retraction, f at the trial point and, speculatively, the gradient norm there -- as one
    // launch chain with one read-back.  Only when the objective and the gradient field name the SAME problem object
    // as the retraction's owner: the chain evaluates the owner's f and gradient, and a wrapped / penalised / logging
    // objective must get the calls the reference makes.
    bool fused_trial = false;
    Scalar trial_grad_sqnorm = 0;
#if OPTIMIZATION_GD_HAVE_MI355
    const MI355::DeviceTrialRetraction *armijo = nullptr;
    if constexpr (MI355::is_device_vector<Tangent>::value && MI355::is_device_vector<Variable>::value &&
                  sizeof...(Args) == 0) {
      if (metric.template target<MI355::FrobeniusMetric>()) {
        armijo = retract.template target<MI355::DeviceTrialRetraction>();
        const auto *fo = f.template target<MI355::DeviceObjective>();
        const auto *go = grad_f.template target<MI355::DeviceGradientField>();
        if (armijo && !(armijo->armijo && armijo->owner && fo && fo->owner == armijo->owner && go &&
                        go->owner == armijo->owner))
          armijo = nullptr;
      }
    }
#endif
    do {
      ls_iters++;
      t *= params.beta;
#if OPTIMIZATION_GD_HAVE_MI355
      if constexpr (MI355::is_device_vector<Tangent>::value && MI355::is_device_vector<Variable>::value &&
                    sizeof...(Args) == 0) {
        if (armijo) {
          auto a = armijo->armijo(x, g, t);
          h = std::move(a.h);
          x_trial = std::move(a.x_trial);
          fx_trial = a.f_trial;
          trial_grad_sqnorm = a.grad_trial_sqnorm;
          fused_trial = true;
        }
      }
#endif
      if (!fused_trial) {
#if OPTIMIZATION_GD_HAVE_MI355
        if constexpr (MI355::is_device_vector<Tangent>::value && MI355::is_device_vector<Variable>::value)
          if (!g.empty())  // the statement sequence on device vectors: observable (mi_ctx_fusion_counters)
            (void)mi_ctx_note_generic(g.context(), MI_GENERIC_TRIAL,
                                      "GradientDescent: metric, objective, gradient field and retraction are not the "
                                      "tagged callables of ONE device problem (plain callables, or wrapped in lambdas)");
#endif
        h = -t * g;
        x_trial = retract(x, h, args...);
        fx_trial = f(x_trial, args...);
      }
      df = fx - fx_trial;
      sufficient = (df > params.sigma * t * g_norm * g_norm);
    } while ((!sufficient) && (ls_iters < params.max_ls_iterations));

    if (params.verbose) {
      std::cout << ", ls iters: ";
      std::cout.width(ls_width);
      std::cout << ls_iters;
    }
    if (!sufficient) {  // :295
      result.status = GradientDescentStatus::LineSearch;
      break;
    }

    h_norm = t * g_norm;                                // :302
    relative_decrease = df / (fabs(fx) + sqrt_eps);     // :305
    result.linesearch_iterations.push_back(ls_iters);
    result.update_step_norms.push_back(h_norm);

    if (user_function) (*user_function)(iteration, elapsed, x, fx, g, h, df, args...);  // :312

    if (params.verbose) {
      std::cout << ", |h|: " << h_norm << ", df: " << df << std::endl;
    }

    if (fused_trial) {  // :323-327 (x_trial keeps its device handle: the gradient callable recognises the point)
      x = std::move(x_trial);
      fx = fx_trial;
      g = grad_f(x, args...);
      g_norm = sqrt(trial_grad_sqnorm);
    } else {
      x = x_trial;
      fx = fx_trial;
      g = grad_f(x, args...);
      g_norm = sqrt(metric(x, g, g, args...));
    }

    if (relative_decrease < params.relative_decrease_tolerance) {  // :330
      result.status = GradientDescentStatus::RelativeDecrease;
      break;
    }
    if (h_norm < params.stepsize_tolerance) {  // :336
      result.status = GradientDescentStatus::Stepsize;
      break;
    }
  }

  result.elapsed_time = Stopwatch::tock(clock_start);
  result.x = x;
  result.f = fx;
  result.gradfx_norm = g_norm;

  if (params.verbose) {
    std::cout << std::endl << std::endl << "Optimization finished!" << std::endl;
    switch (result.status) {
      case GradientDescentStatus::Gradient:
        std::cout << "Found first-order critical point! (Gradient norm: " << g_norm << ")" << std::endl;
        break;
      case GradientDescentStatus::RelativeDecrease:
        std::cout << "Algorithm terminated due to insufficient relative decrease: " << relative_decrease
                  << " < " << params.relative_decrease_tolerance << std::endl;
        break;
      case GradientDescentStatus::Stepsize:
        std::cout << "Algorithm terminated due to excessively small step size: |h| = " << h_norm << " < "
                  << params.stepsize_tolerance << std::endl;
        break;
      case GradientDescentStatus::LineSearch:
        std::cout << "Algorithm terminated due to linesearch's inability to find a stepsize with "
                     "sufficient decrease"
                  << std::endl;
        break;
      case GradientDescentStatus::IterationLimit:
        std::cout << "Algorithm exceeded maximum number of outer iterations" << std::endl;
        break;
      case GradientDescentStatus::ElapsedTime:
        std::cout << "Algorithm exceeded maximum allowed computation time: (" << result.elapsed_time << " > "
                  << params.max_computation_time << " seconds)" << std::endl;
        break;
    }
    std::cout << "Final objective value: " << result.f << std::endl;
    std::cout << "Norm of Riemannian gradient: " << result.gradfx_norm << std::endl;
    std::cout << "Total elapsed computation time: " << result.elapsed_time << " seconds" << std::endl
              << std::endl;
    std::cout << std::defaultfloat;
    std::cout.precision(6);
  }
  return result;
}

template <typename Vector, typename Scalar = double, typename... Args>
using EuclideanGradientDescentUserFunction = GradientDescentUserFunction<Vector, Vector, Scalar, Args...>;

// flat metric and R_X(V) = X + V                                               (reference :420-434)
template <typename Vector, typename Scalar = double, typename... Args>
GradientDescentResult<Vector, Scalar> EuclideanGradientDescent(
    const Objective<Vector, Scalar, Args...> &f, const EuclideanVectorField<Vector, Args...> grad_f,
    const Vector &x0, Args &...args,
    const GradientDescentParams<Scalar> &params = GradientDescentParams<Scalar>(),
    const std::optional<EuclideanGradientDescentUserFunction<Vector, Scalar, Args...>> &user_function =
        std::nullopt) {
  const RiemannianMetric<Vector, Vector, Scalar, Args...> metric = EuclideanMetric<Vector, Scalar, Args...>;
  const Retraction<Vector, Vector, Args...> retract = EuclideanRetraction<Vector, Args...>;
  return GradientDescent<Vector, Vector, Scalar, Args...>(f, grad_f, metric, retract, x0, args..., params,
                                                          user_function);
}

}  // namespace Riemannian
}  // namespace Optimization
