// harness_host.cpp -- the MI355X build's template layer (optimization_amd/include/Optimization/...)
// instantiated on a plain HOST vector through exactly the same driver code as the real reference
// (oracle/template_driver.inc, compiled into oracle/_ref/libref.so with the reference's headers).
// pytest compares hz_* with ref_* bit for bit: same problems, same call sequence, only the template
// implementations differ.  Compiled WITHOUT the C-ABI header on the include path, i.e. it also
// proves the template layer is self-contained, generic C++17 (no device dependency).
#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/GradientDescent.h"
#include "Optimization/Riemannian/TNLS.h"
#include "Optimization/Riemannian/TNT.h"

#define DRV(name) hz_##name
#include "template_driver.inc"

// ------------------------------------------------------------------------------------------------------------------
// LOBPCG through the template layer's GENERIC path (any dense Matrix / Vector pair, element access only): compiled
// here without the C-ABI header on the include path, on a minimal dense matrix type -- what a client holding
// Eigen::MatrixXd / VectorXd gets.  Diagonal operators A, B, T as in the reference's tests/LOBPCG_unit_test.cpp:56-74,
// or a CSR matrix for A.  theta_trace / r_trace (trace_cap x nx, row per iteration) record the Ritz values and
// residual norms the user function sees.
// ------------------------------------------------------------------------------------------------------------------
#include "Optimization/LinearAlgebra/LOBPCG.h"

namespace {
struct DenseM {  // column-major
  size_t r_ = 0, c_ = 0;
  std::vector<double> d;
  DenseM() = default;
  DenseM(size_t r, size_t c) : r_(r), c_(c), d(r * c, 0.0) {}
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  double &operator()(size_t i, size_t j) { return d[i + j * r_]; }
  double operator()(size_t i, size_t j) const { return d[i + j * r_]; }
};
struct DenseV {
  std::vector<double> d;
  DenseV() = default;
  explicit DenseV(size_t n) : d(n, 0.0) {}
  size_t size() const { return d.size(); }
  double &operator()(size_t i) { return d[i]; }
  double operator()(size_t i) const { return d[i]; }
};
}  // namespace

extern "C" int hz_lobpcg_dense(size_t m, size_t nx, size_t nev, const double *Adiag, const int *rowptr, const int *col,
                               const double *val, const double *Bdiag, const double *Tdiag, const double *X0,
                               size_t max_iters, double tau, double *Theta_out, double *X_out, size_t *num_iters,
                               size_t *nc_out, double *theta_trace, double *r_trace, size_t trace_cap) {
  namespace LA = Optimization::LinearAlgebra;
  using Op = LA::SymmetricLinearOperator<DenseM>;
  try {
    auto diag_op = [m](const double *dg) -> Op {
      return [dg, m](const DenseM &X) {
        DenseM Y(m, X.cols());
        for (size_t j = 0; j < X.cols(); ++j)
          for (size_t i = 0; i < m; ++i) Y(i, j) = dg[i] * X(i, j);
        return Y;
      };
    };
    Op A;
    if (rowptr)
      A = [=](const DenseM &X) {
        DenseM Y(m, X.cols());
        for (size_t j = 0; j < X.cols(); ++j)
          for (size_t i = 0; i < m; ++i) {
            double s = 0;
            for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) s += val[k] * X((size_t)col[k], j);
            Y(i, j) = s;
          }
        return Y;
      };
    else
      A = diag_op(Adiag);
    std::optional<Op> B, T;
    if (Bdiag) B = diag_op(Bdiag);
    if (Tdiag) T = diag_op(Tdiag);
    DenseM X0m(m, nx);
    std::memcpy(X0m.d.data(), X0, m * nx * sizeof(double));
    size_t iters = 0, nc = 0;
    std::optional<LA::LOBPCGUserFunction<DenseV, DenseM>> uf = LA::LOBPCGUserFunction<DenseV, DenseM>(
        [&](size_t i, const Op &, const std::optional<Op> &, const std::optional<Op> &, size_t, const DenseV &Theta,
            const DenseM &, const DenseV &r, size_t) {
          if (theta_trace && i - 1 < trace_cap)
            for (size_t j = 0; j < nx; ++j) {
              theta_trace[(i - 1) * nx + j] = Theta(j);
              r_trace[(i - 1) * nx + j] = r(j);
            }
          return false;
        });
    auto res = LA::LOBPCG<DenseV, DenseM>(A, B, T, X0m, nev, max_iters, iters, nc, tau, uf);
    for (size_t j = 0; j < nev; ++j) Theta_out[j] = res.first(j);
    std::memcpy(X_out, res.second.d.data(), m * nev * sizeof(double));
    *num_iters = iters;
    *nc_out = nc;
  } catch (const std::invalid_argument &) {
    return -1;
  } catch (const std::exception &) {
    return -2;
  }
  return 0;
}

// The header-only Rayleigh-Ritz solver as THIS translation unit compiles it (g++ -O2 -march=x86-64-v3
// -ffp-contract=off): pytest compares it bit for bit with the library's own unit (csrc/rr_host.cpp: g++ -O3 -mavx2
// -ffp-contract=off) -- vector width and optimisation level must not change a bit as long as nothing is fused.
extern "C" int hz_rayleigh_ritz(int n, const double *A, const double *B, double *Theta, double *C) {
  return Optimization::LinearAlgebra::dense::generalized_symmetric_eig(n, A, B, Theta, C);
}
