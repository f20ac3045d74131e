// Optimization/LinearAlgebra/DenseSymmetricEigen.h -- small dense symmetric-definite eigenproblem on the host,
// header-only and dependency-free: what Optimization::LinearAlgebra::RayleighRitz needs (reference
// LinearAlgebra/LOBPCG.h:53-62, where Eigen's GeneralizedSelfAdjointEigenSolver does this job).
//
//   generalized_symmetric_eig(n, A, B, Theta, C):  A, B column-major n x n, B positive definite;
//       Theta ascending, C'AC = diag(Theta), C'BC = I;  B is diagonally equilibrated first (:56);
//       returns 0, 1 (non-positive diagonal of B) or 2 (equilibrated B not positive definite).
//
// Used by the generic (any dense Matrix type) path of LOBPCG.h and by the C ABI's mi_rayleigh_ritz
// (optimization_amd/csrc/lobpcg.hip), so the device path and a host run of the same template share one
// Rayleigh-Ritz, bit for bit.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

namespace Optimization {
namespace LinearAlgebra {
namespace dense {

// Left-looking in COLUMN (axpy) form: every inner loop runs down a column of the column-major array, unit stride,
// so the compiler vectorises it (the row-oriented dot form cost 4x more at n = 72).
inline int cholesky_lower(int n, std::vector<double> &A) {
  for (int j = 0; j < n; ++j) {
    double *aj = &A[(size_t)j * n];
    for (int k = 0; k < j; ++k) {  // column j -= L[j,k] * column k   (rows j..n-1)
      const double *ak = &A[(size_t)k * n];
      const double ljk = ak[j];
      for (int i = j; i < n; ++i) aj[i] -= ljk * ak[i];
    }
    const double d = aj[j];
    if (!(d > 0)) return -1;
    const double r = std::sqrt(d);
    aj[j] = r;
    for (int i = j + 1; i < n; ++i) aj[i] /= r;
    for (int i = 0; i < j; ++i) aj[i] = 0;
  }
  return 0;
}

// X <- L^-1 X for the n x n column-major X (forward substitution, axpy form: unit stride down L's columns).
// Pivot loop OUTSIDE, right-hand sides inside: the n divisions of a pivot step are independent of each other (one
// column at a time they form a chain -- divide, short axpy, divide ... -- and the solve is bound by the divider's
// latency: 57 us of the 270 us pencil at n = 72), and every element still receives the same updates in the same order,
// so the result has the same bits.
inline void lower_solve_inplace(int n, const std::vector<double> &L, std::vector<double> &X) {
  std::vector<double> piv((size_t)n);
  for (int k = 0; k < n; ++k) {
    const double *lk = &L[(size_t)k * n];
    const double lkk = lk[k];
    for (int j = 0; j < n; ++j) {
      const double xk = X[(size_t)j * n + k] / lkk;
      X[(size_t)j * n + k] = xk;
      piv[j] = xk;
    }
    for (int j = 0; j < n; ++j) {
      double *x = &X[(size_t)j * n];
      const double xk = piv[j];
      for (int i = k + 1; i < n; ++i) x[i] -= lk[i] * xk;
    }
  }
}

// Symmetric eigen-decomposition of the ns x ns projected pencil (ns <= 96): Householder reduction to
// tridiagonal form followed by implicit-shift QL with accumulated transformations (the classic
// EISPACK tred2/tql2 pair).  O(n^3) with a small constant: 72 x 72 takes well under a millisecond
// on one host core, so the device never waits long for the Ritz coefficients (a cyclic Jacobi here
// cost 60 ms per LOBPCG iteration, 10x the whole device side of the iteration).
// M is destroyed; on return V(:, j) is the eigenvector of w[j], ascending.
// sqrt(a^2 + b^2): std::hypot's overflow/underflow care costs ~40 ns a call and tql2 makes ~10^4 of them per
// 72 x 72 problem (a third of its time); the plain form is exact enough whenever neither square can overflow or
// vanish, which holds for the equilibrated pencils this is fed -- the guarded branch keeps the general case
static inline double fast_hypot(double a, double b) {
  const double aa = std::fabs(a), ab = std::fabs(b), big = aa > ab ? aa : ab;
  if (big < 1e150 && big > 1e-150) return std::sqrt(a * a + b * b);
  return std::hypot(a, b);
}

// dot product on four independent accumulators (a single one is a chain of dependent fmas, ~4 cycles each)
static inline double dot4(const double *a, const double *b, int n) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int k = 0;
  for (; k + 4 <= n; k += 4) {
    s0 += a[k] * b[k];
    s1 += a[k + 1] * b[k + 1];
    s2 += a[k + 2] * b[k + 2];
    s3 += a[k + 3] * b[k + 3];
  }
  for (; k < n; ++k) s0 += a[k] * b[k];
  return (s0 + s1) + (s2 + s3);
}

inline void sym_eigh(int n, std::vector<double> &M, std::vector<double> &V, double *w) {
  V = M;
  std::vector<double> ev((size_t)n, 0.0);
  double *d = w, *e = ev.data();
#define VV(i, j) V[(size_t)(i) + (size_t)(j) * n]
  // --- Householder tridiagonalisation, last row first
  for (int j = 0; j < n; ++j) d[j] = VV(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0, h = 0;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = VV(i - 1, j);
        VV(i, j) = 0;
        VV(j, i) = 0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = f > 0 ? -std::sqrt(h) : std::sqrt(h);
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0;
      for (int j = 0; j < i; ++j) {  // e = (A u) / h, using the lower triangle only
        f = d[j];
        VV(j, i) = f;
        g = e[j] + VV(j, j) * f;
        for (int k = j + 1; k < i; ++k) {
          g += VV(k, j) * d[k];
          e[k] += VV(k, j) * f;
        }
        e[j] = g;
      }
      f = 0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {  // rank-2 update of the leading block
        f = d[j];
        g = e[j];
        for (int k = j; k < i; ++k) VV(k, j) -= f * e[k] + g * d[k];
        d[j] = VV(i - 1, j);
        VV(i, j) = 0;
      }
    }
    d[i] = h;
  }
  // --- accumulate the reflectors
  for (int i = 0; i + 1 < n; ++i) {
    VV(n - 1, i) = VV(i, i);
    VV(i, i) = 1;
    const double h = d[i + 1];
    if (h != 0) {
      for (int k = 0; k <= i; ++k) d[k] = VV(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        const double g = dot4(&VV(0, i + 1), &VV(0, j), i + 1);
        for (int k = 0; k <= i; ++k) VV(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) VV(k, i + 1) = 0;
  }
  for (int j = 0; j < n; ++j) {
    d[j] = VV(n - 1, j);
    VV(n - 1, j) = 0;
  }
  VV(n - 1, n - 1) = 1;
  // --- implicit QL on (d, e)
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0;
  double shift = 0, tst = 0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    tst = std::max(tst, std::fabs(d[l]) + std::fabs(e[l]));
    int mm = l;
    while (mm < n - 1 && std::fabs(e[mm]) > eps * tst) ++mm;
    if (mm > l) {
      int guard = 0;
      do {
        double g = d[l];
        double p = (d[l + 1] - g) / (2 * e[l]);
        double r = fast_hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        shift += h;
        p = d[mm];
        double c = 1, c2 = 1, c3 = 1, s = 0, s2 = 0;
        const double el1 = e[l + 1];
        // The rotations' scalar recurrence (a chain of a square root and two divisions per step) does not depend on V,
        // so rotation i is applied to V one step LATE, next to the scalar work of rotation i - 1: the two are then
        // independent inside one loop body and overlap in the core, instead of the vector update waiting for its own
        // c and s.  Same operations on every element in the same order: same bits.
        double ca = 0, sa = 0;  // the rotation waiting to be applied (to columns ia, ia + 1)
        int ia = -1;
        for (int i = mm - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = fast_hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          if (ia >= 0) {
            double *va = &VV(0, ia), *vb = &VV(0, ia + 1);
            for (int k = 0; k < n; ++k) {
              const double hk = vb[k];
              vb[k] = sa * va[k] + ca * hk;
              va[k] = ca * va[k] - sa * hk;
            }
          }
          ca = c;
          sa = s;
          ia = i;
        }
        if (ia >= 0) {
          double *va = &VV(0, ia), *vb = &VV(0, ia + 1);
          for (int k = 0; k < n; ++k) {
            const double hk = vb[k];
            vb[k] = sa * va[k] + ca * hk;
            va[k] = ca * va[k] - sa * hk;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst && ++guard < 200);
    }
    d[l] += shift;
    e[l] = 0;
  }
#undef VV
  for (int i = 0; i + 1 < n; ++i) {  // ascending
    int mn = i;
    for (int j = i + 1; j < n; ++j)
      if (w[j] < w[mn]) mn = j;
    if (mn != i) {
      std::swap(w[i], w[mn]);
      for (int k = 0; k < n; ++k) std::swap(V[k + (size_t)i * n], V[k + (size_t)mn * n]);
    }
  }
}

inline int generalized_symmetric_eig(int n, const double *A, const double *B, double *Theta, double *C) {
  std::vector<double> D(n), D2(n), L((size_t)n * n), M((size_t)n * n), T((size_t)n * n), Y;
  for (int i = 0; i < n; ++i) {
    if (!(B[i + (size_t)i * n] > 0)) return 1;
    D[i] = 1.0 / std::sqrt(B[i + (size_t)i * n]);  // LOBPCG.h:56
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      L[i + (size_t)j * n] = D[i] * B[i + (size_t)j * n] * D[j];  // D B D  :59
      M[i + (size_t)j * n] = D[i] * A[i + (size_t)j * n] * D[j];  // D A D  :59
    }
  if (cholesky_lower(n, L)) return 2;
  // M <- L^-1 M L^-T as two forward substitutions with a transposition in between:
  // T = L^-1 M, then (T L^-T)' = L^-1 T'
  lower_solve_inplace(n, L, M);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) T[i + (size_t)j * n] = M[j + (size_t)i * n];
  lower_solve_inplace(n, L, T);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) M[i + (size_t)j * n] = T[j + (size_t)i * n];
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) {
      const double a = .5 * (M[i + (size_t)j * n] + M[j + (size_t)i * n]);
      M[i + (size_t)j * n] = a;
      M[j + (size_t)i * n] = a;
    }
  sym_eigh(n, M, Y, Theta);
  // x = L^-T y ; C = D x  (:61): backward substitution in axpy form on U = L' (unit stride down U's columns; the
  // dot form is a chain of dependent fmas, 4x slower)
  for (int k = 0; k < n; ++k)
    for (int i = 0; i < n; ++i) T[i + (size_t)k * n] = L[k + (size_t)i * n];  // T = U = L'
  // (pivot loop outside, eigenvectors inside: as lower_solve_inplace, same bits)
  for (int k = n; k-- > 0;) {
    const double *uk = &T[(size_t)k * n];
    const double ukk = uk[k];
    for (int j = 0; j < n; ++j) {
      const double xk = Y[(size_t)j * n + k] / ukk;
      Y[(size_t)j * n + k] = xk;
      D2[j] = xk;
    }
    for (int j = 0; j < n; ++j) {
      double *x = &Y[(size_t)j * n];
      const double xk = D2[j];
      for (int i = 0; i < k; ++i) x[i] -= uk[i] * xk;
    }
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) C[i + (size_t)j * n] = D[i] * Y[i + (size_t)j * n];
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// The k LOWEST eigenpairs only (r04).  An LOBPCG iteration uses nx of the ns <= 3 nx Ritz pairs of its projected
// pencil (reference LOBPCG.h:278,288 read C(:, :nx), :293-318 Theta(:nx)); the full solver above spends two thirds of
// its time on eigenvector columns nobody reads.  Same mathematics, same reflectors, same rotations:
//   * Householder reduction as in sym_eigh, but the reflectors are KEPT (not accumulated into an n x n matrix);
//   * implicit QL on (d, e) with the SAME scalar recurrence -- so the eigenvalues have the bits of the full solver --
//     while every plane rotation (c, s, position) is recorded instead of being applied to n rows;
//   * the wanted eigenvectors of the tridiagonal matrix are the wanted columns of the rotations' product, obtained by
//     applying the recorded rotations in REVERSE to unit vectors (a rotation touches two entries of a vector: 6 k flops
//     instead of 6 n), then the reflectors, then L^-T and D, all on an n x k block.
// No inverse iteration, no re-orthogonalisation: the vectors are as orthogonal as tql2's, clusters or not (the converged
// Ritz values of LOBPCG ARE clusters).  They agree with the full solver's columns to rounding, not to the bit.
// 72 x 72, k = 24: 2.4 instead of 5 Mflop.
// ---------------------------------------------------------------------------------------------------------------------
struct PlaneRotation {
  double c, s;
  int i;  // acts on entries (i, i + 1)
};

inline int generalized_symmetric_eig_lowest(int n, int k, const double *A, const double *B, double *Theta, double *C) {
  if (k > n) k = n;
  std::vector<double> D(n), L((size_t)n * n), M((size_t)n * n), T((size_t)n * n);
  for (int i = 0; i < n; ++i) {
    if (!(B[i + (size_t)i * n] > 0)) return 1;
    D[i] = 1.0 / std::sqrt(B[i + (size_t)i * n]);
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      L[i + (size_t)j * n] = D[i] * B[i + (size_t)j * n] * D[j];
      M[i + (size_t)j * n] = D[i] * A[i + (size_t)j * n] * D[j];
    }
  if (cholesky_lower(n, L)) return 2;
  lower_solve_inplace(n, L, M);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) T[i + (size_t)j * n] = M[j + (size_t)i * n];
  lower_solve_inplace(n, L, T);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) M[i + (size_t)j * n] = T[j + (size_t)i * n];
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) {
      const double a = .5 * (M[i + (size_t)j * n] + M[j + (size_t)i * n]);
      M[i + (size_t)j * n] = a;
      M[j + (size_t)i * n] = a;
    }
  // --- Householder tridiagonalisation, last row first: the statements of sym_eigh; reflector i stays in column i of V
  // (rows 0 .. i-1) with its h in hs[i]
  std::vector<double> &V = M;
  std::vector<double> dv((size_t)n, 0.0), ev((size_t)n, 0.0), hs((size_t)n, 0.0);
  double *d = dv.data(), *e = ev.data();
#define VV(i, j) V[(size_t)(i) + (size_t)(j) * n]
  for (int j = 0; j < n; ++j) d[j] = VV(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0, h = 0;
    for (int kk = 0; kk < i; ++kk) scale += std::fabs(d[kk]);
    if (scale == 0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = VV(i - 1, j);
        VV(i, j) = 0;
        VV(j, i) = 0;
      }
    } else {
      for (int kk = 0; kk < i; ++kk) {
        d[kk] /= scale;
        h += d[kk] * d[kk];
      }
      double f = d[i - 1];
      double g = f > 0 ? -std::sqrt(h) : std::sqrt(h);
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        VV(j, i) = f;
        g = e[j] + VV(j, j) * f;
        for (int kk = j + 1; kk < i; ++kk) {
          g += VV(kk, j) * d[kk];
          e[kk] += VV(kk, j) * f;
        }
        e[j] = g;
      }
      f = 0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int kk = j; kk < i; ++kk) VV(kk, j) -= f * e[kk] + g * d[kk];
        d[j] = VV(i - 1, j);
        VV(i, j) = 0;
      }
    }
    hs[i] = h;
  }
  // diagonal of the tridiagonal matrix (what sym_eigh parks in row n-1 while it accumulates)
  for (int i = 0; i < n; ++i) d[i] = VV(i, i);
  // --- implicit QL on (d, e): the scalar recurrence of sym_eigh, rotations recorded
  // (a plain array with a running count: push_back's capacity test sits on the recurrence's critical path otherwise.
  // tql2 needs ~1.2 n^2 rotations on these pencils; 200 guard iterations of n rotations per eigenvalue is the hard cap)
  std::vector<PlaneRotation> rot((size_t)4 * n * n + 64);
  size_t nrot = 0;
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0;
  double shift = 0, tst = 0;
  const double eps = 2.220446049250313e-16;
  for (int l = 0; l < n; ++l) {
    tst = std::max(tst, std::fabs(d[l]) + std::fabs(e[l]));
    int mm = l;
    while (mm < n - 1 && std::fabs(e[mm]) > eps * tst) ++mm;
    if (mm > l) {
      int guard = 0;
      do {
        double g = d[l];
        double p = (d[l + 1] - g) / (2 * e[l]);
        double r = fast_hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        shift += h;
        p = d[mm];
        double c = 1, c2 = 1, c3 = 1, s = 0, s2 = 0;
        const double el1 = e[l + 1];
        for (int i = mm - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = fast_hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          if (nrot == rot.size()) rot.resize(2 * rot.size());
          rot[nrot].c = c;  // V(:, i), V(:, i+1) <- V(:, i) c - V(:, i+1) s,  V(:, i) s + V(:, i+1) c
          rot[nrot].s = s;
          rot[nrot].i = i;
          ++nrot;
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst && ++guard < 200);
    }
    d[l] += shift;
    e[l] = 0;
  }
  // --- the k smallest, ascending (ties by position)
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
  // W (row-major n x k: a rotation then updates two contiguous rows): column j = e_{order[j]}, rotations in reverse.
  // With V_final = G_1 G_2 ... G_K (each G acting on two columns from the right), V_final e_p = G_1 ( ... (G_K e_p)).
  std::vector<double> W((size_t)n * k, 0.0);
  for (int j = 0; j < k; ++j) {
    Theta[j] = d[order[j]];
    W[(size_t)order[j] * k + j] = 1.0;
  }
  for (size_t q = nrot; q-- > 0;) {
    const double c = rot[q].c, s = rot[q].s;
    double *wa = &W[(size_t)rot[q].i * k], *wb = wa + k;
    for (int j = 0; j < k; ++j) {
      const double a = wa[j], b = wb[j];
      wa[j] = c * a + s * b;
      wb[j] = c * b - s * a;
    }
  }
  // --- the reflectors: Z = P_{n-1} ... P_1 with P_i = I - u_i u_i' / h_i on the leading i entries (the order in which
  // sym_eigh accumulates them), so Z W = P_{n-1}( ... (P_1 W))
  std::vector<double> gj((size_t)k);
  for (int i = 1; i < n; ++i) {
    const double h = hs[i];
    if (h == 0) continue;
    const double *u = &VV(0, i);
    for (int j = 0; j < k; ++j) gj[j] = 0;
    for (int r = 0; r < i; ++r) {
      const double ur = u[r];
      const double *wr = &W[(size_t)r * k];
      for (int j = 0; j < k; ++j) gj[j] += ur * wr[j];
    }
    for (int r = 0; r < i; ++r) {
      const double ur = u[r] / h;
      double *wr = &W[(size_t)r * k];
      for (int j = 0; j < k; ++j) wr[j] -= gj[j] * ur;
    }
  }
#undef VV
  // --- x = L^-T y (backward substitution, row-major block: row r -= L(i, r) * row i), C = D x
  for (int i = n; i-- > 0;) {
    double *wi = &W[(size_t)i * k];
    const double lii = L[i + (size_t)i * n];
    for (int j = 0; j < k; ++j) wi[j] /= lii;
    for (int r = 0; r < i; ++r) {
      const double lir = L[i + (size_t)r * n];  // L(i, r)
      double *wr = &W[(size_t)r * k];
      for (int j = 0; j < k; ++j) wr[j] -= lir * wi[j];
    }
  }
  for (int j = 0; j < k; ++j)
    for (int i = 0; i < n; ++i) C[i + (size_t)j * n] = D[i] * W[(size_t)i * k + j];
  return 0;
}

}  // namespace dense
}  // namespace LinearAlgebra
}  // namespace Optimization
