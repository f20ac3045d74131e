// Optimization/LinearAlgebra/Concepts.h -- drop-in for the reference header of the same path
// (reference LinearAlgebra/Concepts.h:16-26): callable aliases used by the Krylov solvers.
#pragma once

#include <functional>

namespace Optimization {
namespace LinearAlgebra {

// y = A(x)
template <typename X, typename Y, typename... Args>
using LinearOperator = std::function<Y(const X &x, Args &...args)>;

// A : X -> X, self-adjoint with respect to the inner product in use
template <typename X, typename... Args>
using SymmetricLinearOperator = LinearOperator<X, X, Args...>;

// <x, y>
template <typename Vector, typename Scalar = double, typename... Args>
using InnerProduct = std::function<Scalar(const Vector &X, const Vector &Y, Args &...args)>;

}  // namespace LinearAlgebra
}  // namespace Optimization
