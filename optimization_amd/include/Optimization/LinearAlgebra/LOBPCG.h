// Optimization/LinearAlgebra/LOBPCG.h -- drop-in for the reference header of the same path: Knyazev's
// locally optimal block preconditioned conjugate gradient method for the smallest eigenpairs of
// A x = lambda B x, with soft locking.
//
//   reference: include/Optimization/LinearAlgebra/LOBPCG.h
//              RayleighRitz :53-62, LOBPCGUserFunction :86-93, LOBPCG (given X0) :131-337,
//              LOBPCG (random X0) :376-390
//
// MI355X build, written from scratch against that interface (same template parameters, argument
// order, defaults, exceptions, iteration structure -- see the :line tags).  The reference is written
// against Eigen's dense API; here the panel algebra goes through a handful of free functions found
// by argument-dependent lookup on the Matrix type (Optimization/MI355/Matrix.h provides them for
// MI355::DeviceMatrix: column-major panels in HBM, Gram products on fp64 MFMA):
//     gram(S, T) -> small host matrix S'T          times_small(S, C, row0, kc) -> S C[row0:, :kc]
//     residual_and_norms(AX, BX, X, theta, r, xn)   gaussian_probe(like, m, nx)   rayleigh_ritz(A, B)
//     ritz_update(S, C, nx, X, P) -> X = S C[:, :nx] and P = S[:, nx:] C[nx:, :nx] in one pass over S
//   plus the members rows(), cols(), leftCols(k), middleCols(j,k), rightCols(k), set_cols(...),
//   truncate_cols(k), norm().
// As in the reference, the operators are invoked WITHOUT the Args pack (reference :213-219,247,267-282).
#pragma once

#include <cmath>
#include <cstddef>
#include <functional>
#include <optional>
#include <stdexcept>
#include <tuple>
#include <utility>

#include "Optimization/LinearAlgebra/Concepts.h"
#include "Optimization/MI355/Matrix.h"

namespace Optimization {
namespace LinearAlgebra {

// Basic Rayleigh-Ritz step: for symmetric A and SPD B returns (Theta, C) with Theta ascending,
// C'AC = diag(Theta), C'BC = I; B is diagonally equilibrated first.           (reference :53-62)
template <typename Vector, typename Matrix>
std::pair<Vector, Matrix> RayleighRitz(const Matrix &A, const Matrix &B) {
  auto tc = rayleigh_ritz(A, B);  // ADL on the small-matrix type
  return std::pair<Vector, Matrix>(std::move(tc.first), std::move(tc.second));
}

// Observer called once per iteration after the Ritz pairs and residuals are updated; true stops.
// (reference :86-93)
template <typename Vector, typename Matrix, typename Scalar = double, typename... Args>
using LOBPCGUserFunction = std::function<bool(
    size_t i, const SymmetricLinearOperator<Matrix, Args...> &A,
    const std::optional<SymmetricLinearOperator<Matrix, Args...>> &B,
    const std::optional<SymmetricLinearOperator<Matrix, Args...>> &T, size_t nev, const Vector &Theta,
    const Matrix &X, const Vector &residuals, size_t nc, Args &...args)>;

// nev smallest eigenpairs from the block X0 (m x nx, nev <= nx <= m).  Eigenpair i counts as
// converged when |A x_i - theta_i B x_i| <= tau (|A|_2 + |theta_i| |B|_2) |x_i| with the 2-norms
// estimated from a Gaussian probe (Sec. 4.3 of "A Robust and Efficient Implementation of LOBPCG");
// only the leading run of converged pairs is soft-locked.                       (reference :131-337)
template <typename Vector, typename Matrix, typename Scalar = double, typename... Args>
std::pair<Vector, Matrix>
LOBPCG(const SymmetricLinearOperator<Matrix, Args...> &A,
       const std::optional<SymmetricLinearOperator<Matrix, Args...>> &B,
       const std::optional<SymmetricLinearOperator<Matrix, Args...>> &T, const Matrix &X0, size_t nev,
       size_t max_iters, size_t &num_iters, size_t &nc, Args &...args, Scalar tau = 1e-6,
       const std::optional<LOBPCGUserFunction<Vector, Matrix, Scalar, Args...>> &user_function =
           std::nullopt) {
  const size_t m = X0.rows();
  const size_t nx = X0.cols();
  if (nev > nx)
    throw std::invalid_argument("Block size nx must be greater than or equal to the number nev of "
                                "desired eigenpairs");  // :150
  if (nx > m)
    throw std::invalid_argument("Block size nx must be less than or equal to the dimension m of the "
                                "problem");  // :155

  Matrix X = X0;  // :162
  Matrix AX, BX, R, W, P;
  // search basis [X, W, P], allocated at full width (:185) -- twice: iteration k reads one panel and writes the
  // X, R and P of iteration k+1 straight into the blocks of the other, so that while no pair is locked (nc == 0,
  // no preconditioner) the column copies of :254-259 disappear (2.3 GB per iteration at cfg5)
  Matrix Sa = empty_panel(X0, m, 3 * nx), Sb = empty_panel(X0, m, 3 * nx);
  Matrix *Scur = &Sa, *Snext = &Sb;
  bool in_basis = false;  // X, R, P are views of *Scur's blocks 0, 1, 2
  Vector Theta, r, xnorm;
  size_t ns = 0;

  // 2-norm estimates from one Gaussian probe                                      :205-214
  Scalar A2normest, B2normest;
  {
    const Matrix Omega = gaussian_probe(X0, m, nx);
    const Scalar om = Omega.norm();
    A2normest = A(Omega).norm() / om;
    B2normest = B ? (*B)(Omega).norm() / om : 1.0;
  }

  // B-orthonormalise the start block: rotate AX and BX (X itself is NOT rotated)   :218-230
  AX = A(X);
  BX = B ? (*B)(X) : X;
  {
    auto tc = rayleigh_ritz(gram(X, AX), gram(X, BX));
    Theta = Vector(std::move(tc.first));
    AX = times_small(AX, tc.second, 0, nx);
    BX = times_small(BX, tc.second, 0, nx);
  }
  R = residual_and_norms(AX, BX, X, Theta, r, xnorm);
  nc = 0;  // :233

  for (num_iters = 1; num_iters < max_iters; ++num_iters) {  // :237
    if (T) W = (*T)(R);  // preconditioned residuals (T absent: W is R itself, no copy)   :247

    // S = [X, W(not converged), P(not converged)]  (soft locking drops the FIRST nc columns)  :254-264
    Matrix &S = *Scur;
    const bool in_place = in_basis && nc == 0;  // [X | R | P] already sit where the basis wants them
    if (!in_basis) S.set_cols(0, X, 0, nx);
    if (T) {
      S.set_cols(nx, W, nc, nx - nc);
    } else if (!in_place) {
      if (in_basis) {  // R is a view of S itself: shift its unlocked columns through a copy
        const Matrix Rc = R;
        S.set_cols(nx, Rc, nc, nx - nc);
      } else {
        S.set_cols(nx, R, nc, nx - nc);
      }
    }
    if (num_iters > 1) {
      if (!in_place) {
        if (in_basis) {
          const Matrix Pc = P;
          S.set_cols(2 * nx - nc, Pc, nc, nx - nc);
        } else {
          S.set_cols(2 * nx - nc, P, nc, nx - nc);
        }
      }
      ns = 3 * nx - 2 * nc;
    } else {
      ns = 2 * nx - nc;
    }
    const Matrix Sns = S.leftCols(ns);  // view

    const Matrix AS = A(Sns);                       // :267
    const Matrix BS = B ? (*B)(Sns) : Matrix();     // :268 (B absent: S'BS = S'S, no copy)
    auto tc = rayleigh_ritz(gram(Sns, AS), B ? gram(Sns, BS) : gram(Sns, Sns));  // :271-275
    Theta = Vector(std::move(tc.first));
    const auto &C = tc.second;

    // X = S C[:, :nx] (:278) and P = S[:, nx:ns] C[nx:ns, :nx] (:288) read the same basis: one pass
    // ... written straight into the other panel's first and third block
    X = Snext->leftCols(nx);
    P = Snext->middleCols(2 * nx, nx);
    ritz_update_into(Sns, C, nx, X, P);
    AX = A(X);                       // operators re-applied, not AS C               :281-282
    if (B) BX = (*B)(X);             // B absent: BX is X itself, no copy
    R = Snext->middleCols(nx, nx);   // residuals into its second block
    residual_and_norms_into(R, AX, B ? BX : X, X, Theta.head(nx), r, xnorm);  // :285,293
    in_basis = true;
    std::swap(Scur, Snext);

    // leading run of converged pairs among the first nev                           :298-318
    for (nc = 0; nc < nev; ++nc) {
      const Scalar tol = tau * (A2normest + B2normest * std::fabs(Theta(nc))) * xnorm(nc);
      if (!(r(nc) <= tol)) break;
    }

    if (user_function && (*user_function)(num_iters, A, B, T, nev, Theta.head(nx), X, r, nc, args...))
      break;  // :322-324
    if (nc == nev) break;  // :327
  }

  Theta.conservativeResize(nev);  // :333-334
  X.truncate_cols(nev);
  Matrix Xout = X;  // own storage: X may be a view that keeps a whole 3 nx panel alive
  return std::make_pair(Theta, std::move(Xout));
}

// Same, starting from a random m x nx block                                       (reference :376-390)
template <typename Vector, typename Matrix, typename Scalar = double, typename... Args>
std::pair<Vector, Matrix>
LOBPCG(const SymmetricLinearOperator<Matrix, Args...> &A,
       const std::optional<SymmetricLinearOperator<Matrix, Args...>> &B,
       const std::optional<SymmetricLinearOperator<Matrix, Args...>> &T, size_t m, size_t nx, size_t nev,
       size_t max_iters, size_t &num_iters, size_t &nc, Args &...args, Scalar tau = 1e-6,
       const std::optional<LOBPCGUserFunction<Vector, Matrix, Scalar, Args...>> &user_function =
           std::nullopt) {
  Matrix X0 = Matrix::Random(m, nx);
  return LOBPCG<Vector, Matrix, Scalar, Args...>(A, B, T, X0, nev, max_iters, num_iters, nc, args..., tau,
                                                 user_function);
}

}  // namespace LinearAlgebra
}  // namespace Optimization
