// Optimization/Riemannian/Concepts.h -- drop-in for the reference header of the same path
// (reference Riemannian/Concepts.h:44-190): the function-object vocabulary of the Riemannian
// optimizers (vector fields, linear operators on tangent spaces, quadratic models, metrics,
// retractions, Jacobians), the parameter/result records of smooth optimizers, and the Euclidean
// conveniences.  Written from scratch; names, signatures and defaults follow the reference so that
// client code recompiles unchanged.
#pragma once

#include <functional>
#include <utility>
#include <vector>

#include "Optimization/Base/Concepts.h"

namespace Optimization {
namespace Riemannian {

// X |-> V(X) in T_X(M)                                          (reference :44-45)
template <typename Variable, typename Tangent, typename... Args>
using VectorField = std::function<Tangent(const Variable &X, Args &...args)>;

// (X, V) |-> A_X[V], a linear map on T_X(M)                     (reference :48-50)
template <typename Variable, typename Tangent, typename... Args>
using LinearOperator = std::function<Tangent(const Variable &X, const Tangent &V, Args &...args)>;

// X |-> A_X                                                     (reference :53-56)
template <typename Variable, typename Tangent, typename... Args>
using LinearOperatorConstructor =
    std::function<LinearOperator<Variable, Tangent, Args...>(const Variable &X, Args &...args)>;

// Fills in grad f(X) and Hess f(X) at once                      (reference :59-63)
template <typename Variable, typename Tangent, typename... Args>
using QuadraticModel = std::function<void(const Variable &X, Tangent &gradient,
                                          LinearOperator<Variable, Tangent, Args...> &Hessian,
                                          Args &...args)>;

// F : X -> Y between manifolds                                  (reference :66-67)
template <typename VariableX, typename VariableY, typename... Args>
using Mapping = std::function<VariableY(const VariableX &X, Args &...args)>;

// dF_X : T_X -> T_F(X) and its adjoint                          (reference :70-79)
template <typename VariableX, typename TangentX, typename TangentY, typename... Args>
using Jacobian = std::function<TangentY(const VariableX &X, const TangentX &V, Args &...args)>;

template <typename VariableX, typename TangentX, typename TangentY, typename... Args>
using JacobianAdjoint = std::function<TangentX(const VariableX &X, const TangentY &W, Args &...args)>;

// X |-> (dF_X, dF_X^*)                                          (reference :82-87)
template <typename VariableX, typename TangentX, typename TangentY, typename... Args>
using JacobianPairFunction =
    std::function<std::pair<Jacobian<VariableX, TangentX, TangentY, Args...>,
                            JacobianAdjoint<VariableX, TangentX, TangentY, Args...>>(
        const VariableX &X, Args &...args)>;

// g_X(V1, V2)                                                   (reference :101-104)
template <typename Variable, typename Tangent, typename Scalar = double, typename... Args>
using RiemannianMetric =
    std::function<Scalar(const Variable &X, const Tangent &V1, const Tangent &V2, Args &...args)>;

// R_X(V) in M                                                   (reference :109-112)
template <typename Variable, typename Tangent, typename... Args>
using Retraction = std::function<Variable(const Variable &X, const Tangent &update, Args &...args)>;

// Stopping tolerances of smooth optimizers                      (reference :116-131)
template <typename Scalar = double>
struct SmoothOptimizerParams : public OptimizerParams {
  Scalar gradient_tolerance = 1e-6;           // on |grad f(x)|
  Scalar relative_decrease_tolerance = 1e-6;  // on (f_prev - f) / (sqrt(eps) + |f_prev|)
  Scalar stepsize_tolerance = 1e-6;           // on |h|
};

// ... and what they return on top of OptimizerResult           (reference :135-148)
template <typename Variable, typename Scalar = double>
struct SmoothOptimizerResult : public OptimizerResult<Variable, Scalar> {
  Scalar gradfx_norm;                     // |grad f| at the returned point
  std::vector<Scalar> gradient_norms;     // per started iteration (+ final)
  std::vector<Scalar> update_step_norms;  // |h| of every computed step
};

/// Euclidean spaces: points and tangent vectors share one type   (reference :162-190)

template <typename Vector, typename... Args>
using EuclideanVectorField = VectorField<Vector, Vector, Args...>;

template <typename Vector, typename... Args>
using EuclideanLinearOperator = LinearOperator<Vector, Vector, Args...>;

template <typename Vector, typename... Args>
using EuclideanLinearOperatorConstructor = LinearOperatorConstructor<Vector, Vector, Args...>;

template <typename Vector, typename... Args>
using EuclideanQuadraticModel = QuadraticModel<Vector, Vector, Args...>;

// <V1, V2> := V1.dot(V2)
template <typename Vector, typename Scalar = double, typename... Args>
Scalar EuclideanInnerProduct(const Vector &V1, const Vector &V2, Args &...) {
  return V1.dot(V2);
}

// the flat metric ignores the base point
template <typename Vector, typename Scalar = double, typename... Args>
Scalar EuclideanMetric(const Vector &, const Vector &V1, const Vector &V2, Args &...args) {
  return EuclideanInnerProduct<Vector, Scalar, Args...>(V1, V2, args...);
}

// R_X(V) = X + V
template <typename Vector, typename... Args>
Vector EuclideanRetraction(const Vector &X, const Vector &V, Args &...) {
  return X + V;
}

}  // namespace Riemannian
}  // namespace Optimization
