// kkt_dense.h -- the constraint preconditioner of the reference's projected-STPCG tests
// (tests/IterativeSolvers_unit_test.cpp:348-405,437-470) without Eigen/UMFPACK: for a DIAGONAL M and a dense
// m x n constraint matrix A (row-major) the KKT system
//        [M A'][x]   [r]
//        [A 0 ][l] = [0]
// is solved through its Schur complement S = A M^-1 A' (Cholesky, computed once):
//        l = S^-1 A M^-1 r,   x = M^-1 (r - A' l).
// Plain loops in a fixed order, so that every build that includes this file (the real-reference driver, the host
// and the device harness of the MI355X template layer) performs bit-identical arithmetic.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <vector>

struct KktDense {
  size_t n = 0, m = 0;
  std::vector<double> A;     // m x n, row-major
  std::vector<double> Minv;  // n
  std::vector<double> L;     // m x m lower Cholesky factor of S, row-major
  KktDense(size_t n_, size_t m_, const double *A_, const double *Mdiag) : n(n_), m(m_), A(A_, A_ + n_ * m_), Minv(n_), L(m_ * m_, 0.0) {
    for (size_t i = 0; i < n; ++i) Minv[i] = 1.0 / Mdiag[i];
    std::vector<double> S(m * m, 0.0);
    for (size_t a = 0; a < m; ++a)
      for (size_t b = 0; b <= a; ++b) {
        double s = 0;
        for (size_t i = 0; i < n; ++i) s += A[a * n + i] * Minv[i] * A[b * n + i];
        S[a * m + b] = s;
      }
    for (size_t j = 0; j < m; ++j) {
      double d = S[j * m + j];
      for (size_t k = 0; k < j; ++k) d -= L[j * m + k] * L[j * m + k];
      if (!(d > 0)) throw std::runtime_error("KktDense: Schur complement not positive definite");
      L[j * m + j] = std::sqrt(d);
      for (size_t i = j + 1; i < m; ++i) {
        double s = S[i * m + j];
        for (size_t k = 0; k < j; ++k) s -= L[i * m + k] * L[j * m + k];
        L[i * m + j] = s / L[j * m + j];
      }
    }
  }
  // out (n) = A' l
  void At(const double *l, double *out) const {
    for (size_t i = 0; i < n; ++i) out[i] = 0;
    for (size_t a = 0; a < m; ++a)
      for (size_t i = 0; i < n; ++i) out[i] += A[a * n + i] * l[a];
  }
  // out (m) = A x
  void Ax(const double *x, double *out) const {
    for (size_t a = 0; a < m; ++a) {
      double s = 0;
      for (size_t i = 0; i < n; ++i) s += A[a * n + i] * x[i];
      out[a] = s;
    }
  }
  // (x, l) = Mc(r)
  void solve(const double *r, double *x, double *l) const {
    std::vector<double> t(n), b(m), w(n);
    for (size_t i = 0; i < n; ++i) t[i] = Minv[i] * r[i];
    Ax(t.data(), b.data());
    for (size_t i = 0; i < m; ++i) {  // L y = b
      double s = b[i];
      for (size_t k = 0; k < i; ++k) s -= L[i * m + k] * b[k];
      b[i] = s / L[i * m + i];
    }
    for (size_t ii = m; ii-- > 0;) {  // L' l = y
      double s = b[ii];
      for (size_t k = ii + 1; k < m; ++k) s -= L[k * m + ii] * l[k];
      l[ii] = s / L[ii * m + ii];
    }
    At(l, w.data());
    for (size_t i = 0; i < n; ++i) x[i] = Minv[i] * (r[i] - w[i]);
  }
};
