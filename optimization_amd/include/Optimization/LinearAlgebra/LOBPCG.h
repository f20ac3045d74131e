// Optimization/LinearAlgebra/LOBPCG.h -- drop-in for the reference header of the same path: Knyazev's
// locally optimal block preconditioned conjugate gradient method for the smallest eigenpairs of
// A x = lambda B x, with soft locking.
//
//   reference: include/Optimization/LinearAlgebra/LOBPCG.h
//              RayleighRitz :53-62, LOBPCGUserFunction :86-93, LOBPCG (given X0) :131-337,
//              LOBPCG (random X0) :376-390
//
// MI355X build, written from scratch against that interface (same template parameters, argument
// order, defaults, exceptions, iteration structure -- see the :line tags).  Two bodies behind the one template:
//
//   * Matrix = MI355::DeviceMatrix (column-major panels in HBM, Optimization/MI355/Matrix.h, included when the
//     C ABI header is on the include path): the panel algebra goes through a handful of free functions found by
//     argument-dependent lookup -- gram (fp64 MFMA), times_small, ritz_update_into, residual_and_norms, ... -- and
//     the search basis lives in two panels that are written in place (detail::lobpcg_device).
//   * any other dense Matrix / Vector pair (the reference's use with Eigen::MatrixXd / VectorXd; anything with
//     Matrix(rows, cols), rows(), cols(), operator()(i, j) and Vector(n), operator()(i), size()): a host
//     implementation written with element access only (detail::lobpcg_dense), statement by statement the
//     reference's iteration.  It needs neither Eigen nor the device library.
//
// Both use the same host Rayleigh-Ritz (DenseSymmetricEigen.h), so a host run and a device run of one problem
// can be compared iteration by iteration.  As in the reference, the operators are invoked WITHOUT the Args pack
// (reference :213-219,247,267-282).
#pragma once

#include <cmath>
#include <cstddef>
#include <functional>
#include <optional>
#include <random>
#include <stdexcept>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "Optimization/LinearAlgebra/Concepts.h"
#include "Optimization/LinearAlgebra/DenseSymmetricEigen.h"
#if __has_include("mi355opt.h") && __has_include("Optimization/MI355/Matrix.h")
#include "Optimization/MI355/Matrix.h"
#define OPTIMIZATION_LOBPCG_HAS_DEVICE_PANELS 1
#endif

namespace Optimization {
namespace LinearAlgebra {

namespace detail {
template <typename M>
struct is_device_panel : std::false_type {};
template <typename M>
struct is_mi355_small : std::false_type {};
#ifdef OPTIMIZATION_LOBPCG_HAS_DEVICE_PANELS
template <>
struct is_device_panel<MI355::DeviceMatrix> : std::true_type {};
template <>
struct is_mi355_small<MI355::HostMatrix> : std::true_type {};
#endif
}  // namespace detail

// Basic Rayleigh-Ritz step: for symmetric A and SPD B returns (Theta, C) with Theta ascending,
// C'AC = diag(Theta), C'BC = I; B is diagonally equilibrated first.           (reference :53-62)
template <typename Vector, typename Matrix>
std::pair<Vector, Matrix> RayleighRitz(const Matrix &A, const Matrix &B) {
  if constexpr (detail::is_mi355_small<Matrix>::value) {
    auto tc = rayleigh_ritz(A, B);  // ADL on the small-matrix type
    return std::pair<Vector, Matrix>(std::move(tc.first), std::move(tc.second));
  } else {
    const size_t n = A.rows();
    std::vector<double> a(n * n), b(n * n), th(n), c(n * n);
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < n; ++i) {
        a[i + j * n] = A(i, j);
        b[i + j * n] = B(i, j);
      }
    if (dense::generalized_symmetric_eig((int)n, a.data(), b.data(), th.data(), c.data()))
      throw std::invalid_argument("RayleighRitz: B must be symmetric positive definite");
    Vector Theta(n);
    Matrix C(n, n);
    for (size_t j = 0; j < n; ++j) {
      Theta(j) = th[j];
      for (size_t i = 0; i < n; ++i) C(i, j) = c[i + j * n];
    }
    return std::make_pair(std::move(Theta), std::move(C));
  }
}

// Observer called once per iteration after the Ritz pairs and residuals are updated; true stops.
// (reference :86-93)
template <typename Vector, typename Matrix, typename Scalar = double, typename... Args>
using LOBPCGUserFunction = std::function<bool(
    size_t i, const SymmetricLinearOperator<Matrix, Args...> &A,
    const std::optional<SymmetricLinearOperator<Matrix, Args...>> &B,
    const std::optional<SymmetricLinearOperator<Matrix, Args...>> &T, size_t nev, const Vector &Theta,
    const Matrix &X, const Vector &residuals, size_t nc, Args &...args)>;

namespace detail {

#ifdef OPTIMIZATION_LOBPCG_HAS_DEVICE_PANELS
// ---- device panels (MI355::DeviceMatrix) -----------------------------------------------------------------------
template <typename Vector, typename Matrix, typename Scalar, typename... Args>
std::pair<Vector, Matrix>
lobpcg_device(const SymmetricLinearOperator<Matrix, Args...> &A,
              const std::optional<SymmetricLinearOperator<Matrix, Args...>> &B,
              const std::optional<SymmetricLinearOperator<Matrix, Args...>> &T, const Matrix &X0, size_t nev,
              size_t max_iters, size_t &num_iters, size_t &nc, Args &...args, Scalar tau,
              const std::optional<LOBPCGUserFunction<Vector, Matrix, Scalar, Args...>> &user_function) {
  const size_t m = X0.rows();
  const size_t nx = X0.cols();

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

  // The basis is never assembled.  S = [X, W(:, nc:), P(:, nc:)] (:254-264) goes to the Gram, update and product
  // kernels as the three column blocks lie (MI355::PanelBlocks) -- with locked pairs (nc > 0) the reference's middleCols
  // assignments would move 2 (nx - nc) columns together every iteration (15 % of a cfg5 run's kernel time).  r05: with a
  // B operator too (the generalized problem took the assembled basis and full Grams until r04).
  const auto *fused_op = B ? nullptr : A.template target<MI355::DeviceCsrPanelOperator>();
  for (num_iters = 1; num_iters < max_iters; ++num_iters) {  // :237
    if (T) W = (*T)(R);  // preconditioned residuals (T absent: W is R itself, no copy)   :247

    MI355::PanelBlocks Sblk;  // the basis, as its blocks
    auto gram_of_blocks = [&]() {
      Sblk.add(X.leftCols(nx));
      Sblk.add((T ? W : R).middleCols(nc, nx - nc));
      if (num_iters > 1) Sblk.add(P.middleCols(nc, nx - nc));
      ns = Sblk.cols();
      // A(S) (:267) column by column is [A(X) | A([W P])], and A(X) is the AX of the previous iteration (:281; the
      // start block's AX was rotated, :226, so the first iteration applies A to all of S); likewise B(S) (:268,282)
      const bool reuse_x = num_iters > 1 && ns > nx;
      const MI355::PanelBlocks rest = reuse_x ? Sblk.from(1) : Sblk.from(0);
      if (fused_op) {  // B absent, A a tagged sparse operator: all of A([W P]) from one panel product
        const Matrix AS = (*fused_op)(rest);  // :267
        const Matrix none;
        // S'A(S) and S'S from ONE pass over S, the upper block triangle of each (A is a SymmetricLinearOperator, so
        // both are symmetric, and one triangle is all the reference's eigensolver reads); one read-back (:271-275)
        return reuse_x ? gram_pair_sym(Sblk, AX, AS) : gram_pair_sym(Sblk, AS, none);
      }
      // A (and B) plain callables on a Matrix.  They are applied to the blocks of S as they lie (a block is a
      // contiguous view) -- to [R | P] in one call while nothing is locked and they sit next to each other in the basis
      // panel -- and the Grams are formed from the blocks of S, A(S) (and B(S)): upper block triangles on the matrix
      // pipe, one read-back (:271-275).  Column for column the same operator results as A(S), B(S) on the assembled
      // basis, which is never formed (r05: also without B; r04 copied the blocks together for a plain callable).
      // NOTE for clients: the reference calls each operator ONCE per iteration on the assembled S
      // (:267-268); here a plain callable is called once on [R | P] or once per block, on views of nx - nc columns, and
      // A(X) is carried over -- same columns out, but another call count and other panel widths than under the reference.
      MI355::PanelBlocks ASb, BSb;
      if (reuse_x) {
        ASb.add(AX.leftCols(nx));
        if (B) BSb.add(BX.leftCols(nx));
      }
      if (reuse_x && rest.blocks() == 2 && in_basis && !T && nc == 0) {
        const Matrix RP = Scur->middleCols(nx, 2 * nx);  // view: R and P are blocks 1 and 2 of the current panel
        ASb.add(A(RP));
        if (B) BSb.add((*B)(RP));
      } else {
        for (size_t i = 0; i < rest.blocks(); ++i) {
          ASb.add(A(rest.block(i)));
          if (B) BSb.add((*B)(rest.block(i)));
        }
      }
      if (!B) return gram_pair_sym(Sblk, ASb);
      return gram_pair_gen(Sblk, ASb, BSb);
    };
    auto gg = gram_of_blocks();
    // only the nx lowest Ritz pairs are read below (:278,288,293-318): the solver that forms just those columns
    auto tc = rayleigh_ritz_lowest(gg.first, gg.second, nx);
    Theta = Vector(std::move(tc.first));
    const auto &C = tc.second;

    // X = S C[:, :nx] (:278) and P = S[:, nx:ns] C[nx:ns, :nx] (:288) read the same basis: one pass
    // ... written straight into the other panel's first and third block
    // (the blocks keep the storage of the old X, R, P alive while their names move on to the other panel)
    X = Snext->leftCols(nx);
    P = Snext->middleCols(2 * nx, nx);
    ritz_update_into(Sblk, C, nx, X, P);
    R = Snext->middleCols(nx, nx);   // residuals into its second block
    if (fused_op) {
      // B absent and A a tagged sparse operator: A(X) (:281), the residual (:285) and the norms (:293,302) in one pass
      AX = apply_with_residual_into(*fused_op, X, Theta.head(nx), R, r, xnorm);
    } else {
      AX = A(X);                       // operators re-applied, not AS C               :281-282
      if (B) BX = (*B)(X);             // B absent: BX is X itself, no copy
      residual_and_norms_into(R, AX, B ? BX : X, X, Theta.head(nx), r, xnorm);  // :285,293
    }
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
#endif  // OPTIMIZATION_LOBPCG_HAS_DEVICE_PANELS

// ---- any dense Matrix / Vector pair, host, element access only -----------------------------------------------
template <typename Vector, typename Matrix, typename Scalar, typename... Args>
std::pair<Vector, Matrix>
lobpcg_dense(const SymmetricLinearOperator<Matrix, Args...> &A,
             const std::optional<SymmetricLinearOperator<Matrix, Args...>> &B,
             const std::optional<SymmetricLinearOperator<Matrix, Args...>> &T, const Matrix &X0, size_t nev,
             size_t max_iters, size_t &num_iters, size_t &nc, Args &...args, Scalar tau,
             const std::optional<LOBPCGUserFunction<Vector, Matrix, Scalar, Args...>> &user_function) {
  const size_t m = X0.rows();
  const size_t nx = X0.cols();
  // the handful of dense operations the iteration needs
  auto cols = [m](const Matrix &M, size_t j0, size_t k) {  // M.middleCols(j0, k)
    Matrix out(m, k);
    for (size_t j = 0; j < k; ++j)
      for (size_t i = 0; i < m; ++i) out(i, j) = M(i, j0 + j);
    return out;
  };
  auto set_cols = [m](Matrix &D, size_t j0, const Matrix &S, size_t i0, size_t k) {  // D.middleCols(j0,k) = S.middleCols(i0,k)
    for (size_t j = 0; j < k; ++j)
      for (size_t i = 0; i < m; ++i) D(i, j0 + j) = S(i, i0 + j);
  };
  auto gram = [m](const Matrix &S, const Matrix &Tm) {  // S' T
    Matrix G(S.cols(), Tm.cols());
    for (size_t b = 0; b < (size_t)Tm.cols(); ++b)
      for (size_t a = 0; a < (size_t)S.cols(); ++a) {
        Scalar s = 0;
        for (size_t i = 0; i < m; ++i) s += S(i, a) * Tm(i, b);
        G(a, b) = s;
      }
    return G;
  };
  auto times = [m](const Matrix &S, size_t s0, size_t ks, const Matrix &C, size_t r0, size_t kc) {
    Matrix Y(m, kc);  // S[:, s0 : s0 + ks) * C[r0 : r0 + ks, 0 : kc)
    for (size_t j = 0; j < kc; ++j) {
      for (size_t i = 0; i < m; ++i) Y(i, j) = 0;
      for (size_t r = 0; r < ks; ++r) {
        const Scalar c = C(r0 + r, j);
        for (size_t i = 0; i < m; ++i) Y(i, j) += S(i, s0 + r) * c;
      }
    }
    return Y;
  };
  auto fro = [](const Matrix &M) {
    Scalar s = 0;
    for (size_t j = 0; j < (size_t)M.cols(); ++j)
      for (size_t i = 0; i < (size_t)M.rows(); ++i) s += M(i, j) * M(i, j);
    return std::sqrt(s);
  };
  auto col_norm = [m](const Matrix &M, size_t j) {
    Scalar s = 0;
    for (size_t i = 0; i < m; ++i) s += M(i, j) * M(i, j);
    return std::sqrt(s);
  };
  auto residuals = [m, nx](const Matrix &AX, const Matrix &BX, const Vector &Theta) {  // AX - BX diag(Theta(:nx))
    Matrix R(m, nx);
    for (size_t j = 0; j < nx; ++j)
      for (size_t i = 0; i < m; ++i) R(i, j) = AX(i, j) - BX(i, j) * Theta(j);
    return R;
  };

  Matrix X = X0;                    // :162
  Matrix AX, BX, R, W, P;
  Matrix S(m, 3 * nx);              // :185
  Vector Theta, r(nx);
  size_t ns = 0;

  Scalar A2normest, B2normest;      // :205-214
  {
    std::default_random_engine gen;
    std::normal_distribution<double> normal(0, 1.0);
    Matrix Omega(m, nx);
    for (size_t i = 0; i < m; ++i)
      for (size_t j = 0; j < nx; ++j) Omega(i, j) = normal(gen);
    const Scalar om = fro(Omega);
    A2normest = fro(A(Omega)) / om;
    B2normest = B ? fro((*B)(Omega)) / om : 1.0;
  }

  AX = A(X);                        // :218
  BX = B ? (*B)(X) : X;             // :219
  {
    auto tc = RayleighRitz<Vector, Matrix>(gram(X, AX), gram(X, BX));  // :222-223
    Theta = std::move(tc.first);
    AX = times(AX, 0, nx, tc.second, 0, nx);  // :226 (X itself is NOT rotated)
    BX = times(BX, 0, nx, tc.second, 0, nx);  // :227
  }
  R = residuals(AX, BX, Theta);     // :230
  nc = 0;                           // :233

  for (num_iters = 1; num_iters < max_iters; ++num_iters) {  // :237
    W = T ? (*T)(R) : R;                                     // :247
    set_cols(S, 0, X, 0, nx);                                // :254
    set_cols(S, nx, W, nc, nx - nc);                         // :255 (soft locking drops the FIRST nc columns)
    if (num_iters > 1) {
      set_cols(S, 2 * nx - nc, P, nc, nx - nc);              // :259
      ns = 3 * nx - 2 * nc;
    } else {
      ns = 2 * nx - nc;                                      // :263
    }
    const Matrix Sns = cols(S, 0, ns);
    const Matrix AS = A(Sns);                                // :267
    const Matrix BS = B ? (*B)(Sns) : Sns;                   // :268
    auto tc = RayleighRitz<Vector, Matrix>(gram(Sns, AS), gram(Sns, BS));  // :271-275
    Theta = std::move(tc.first);
    const Matrix &C = tc.second;
    X = times(Sns, 0, ns, C, 0, nx);                         // :278
    AX = A(X);                                               // :281
    BX = B ? (*B)(X) : X;                                    // :282
    R = residuals(AX, BX, Theta);                            // :285
    P = times(Sns, nx, ns - nx, C, nx, nx);                  // :288
    for (size_t j = 0; j < nx; ++j) r(j) = col_norm(R, j);   // :293
    for (nc = 0; nc < nev; ++nc) {                           // :298-318 leading run of converged pairs
      const Scalar tol = tau * (A2normest + B2normest * std::fabs(Theta(nc))) * col_norm(X, nc);
      if (!(r(nc) <= tol)) break;
    }
    if (user_function) {                                     // :322-324
      Vector th(nx);
      for (size_t j = 0; j < nx; ++j) th(j) = Theta(j);
      if ((*user_function)(num_iters, A, B, T, nev, th, X, r, nc, args...)) break;
    }
    if (nc == nev) break;                                    // :327
  }

  Vector Tout(nev);                                          // :333-334
  for (size_t j = 0; j < nev; ++j) Tout(j) = Theta(j);
  return std::make_pair(std::move(Tout), cols(X, 0, nev));
}

}  // namespace detail

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
#ifdef OPTIMIZATION_LOBPCG_HAS_DEVICE_PANELS
  if constexpr (detail::is_device_panel<Matrix>::value)
    return detail::lobpcg_device<Vector, Matrix, Scalar, Args...>(A, B, T, X0, nev, max_iters, num_iters, nc, args...,
                                                                   tau, user_function);
  else
#endif
    return detail::lobpcg_dense<Vector, Matrix, Scalar, Args...>(A, B, T, X0, nev, max_iters, num_iters, nc, args...,
                                                                  tau, user_function);
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
