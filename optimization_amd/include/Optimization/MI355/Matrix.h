// Optimization/MI355/Matrix.h -- dense types for the LOBPCG drop-in:
//   HostMatrix     small column-major dense matrix on the host (Gram matrices, Ritz coefficients)
//   HostVectorD    small host vector (Ritz values, residual norms)
//   DeviceMatrix   tall-skinny m x k COLUMN-MAJOR panel in HBM (ld = m), the layout of the reference's
//                  Eigen dense matrices, with zero-copy column-block views (leftCols / middleCols /
//                  rightCols) and the panel operations LOBPCG needs as free functions found by
//                  argument-dependent lookup (gram, times_small, residual_and_norms, ...), each one
//                  C-ABI call into optimization_amd/csrc/lobpcg.hip.
#pragma once

#include <cmath>
#include <cstddef>
#include <memory>
#include <random>
#include <stdexcept>
#include <utility>
#include <vector>

#include "Optimization/MI355/Device.h"

namespace Optimization {
namespace MI355 {

class HostVectorD {
 public:
  HostVectorD() = default;
  explicit HostVectorD(size_t n, double v = 0.0) : d_(n, v) {}
  size_t size() const { return d_.size(); }
  void resize(size_t n) { d_.resize(n); }
  void conservativeResize(size_t n) { d_.resize(n); }
  double &operator()(size_t i) { return d_[i]; }
  double operator()(size_t i) const { return d_[i]; }
  double &operator[](size_t i) { return d_[i]; }
  double operator[](size_t i) const { return d_[i]; }
  HostVectorD head(size_t k) const {
    HostVectorD h(k);
    for (size_t i = 0; i < k; ++i) h.d_[i] = d_[i];
    return h;
  }
  double *data() { return d_.data(); }
  const double *data() const { return d_.data(); }

 private:
  std::vector<double> d_;
};

class HostMatrix {
 public:
  HostMatrix() = default;
  HostMatrix(size_t r, size_t c, double v = 0.0) : r_(r), c_(c), d_(r * c, v) {}
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  double &operator()(size_t i, size_t j) { return d_[i + j * r_]; }
  double operator()(size_t i, size_t j) const { return d_[i + j * r_]; }
  double *data() { return d_.data(); }
  const double *data() const { return d_.data(); }

 private:
  size_t r_ = 0, c_ = 0;
  std::vector<double> d_;
};

// The context used by DeviceMatrix::Random(m, nx), whose reference counterpart Matrix::Random
// (LOBPCG.h:386) takes no context argument.
inline mi_ctx *&current_context_slot() {
  static mi_ctx *c = nullptr;
  return c;
}
inline void make_current(const Context &ctx) { current_context_slot() = ctx.get(); }

class DeviceMatrix {
 public:
  DeviceMatrix() = default;
  DeviceMatrix(mi_ctx *ctx, size_t rows, size_t cols)
      : buf_(std::make_shared<DeviceVector>(DeviceVector::on(ctx, rows * cols))), rows_(rows), cols_(cols) {
    bind();
  }
  DeviceMatrix(const Context &ctx, size_t rows, size_t cols) : DeviceMatrix(ctx.get(), rows, cols) {}
  // from a column-major host array
  DeviceMatrix(const Context &ctx, size_t rows, size_t cols, const double *host) : DeviceMatrix(ctx, rows, cols) {
    check(mi_vec_upload(buf_->handle(), host, rows * cols));
  }
  DeviceMatrix(const DeviceMatrix &o) : rows_(o.rows_), cols_(o.cols_) {  // deep copy (value semantics)
    if (o.view_) {
      buf_ = std::make_shared<DeviceVector>(DeviceVector::on(o.context(), rows_ * cols_));
      bind();
      check(mi_vec_copy(view_, o.view_));
    }
  }
  DeviceMatrix(DeviceMatrix &&o) noexcept
      : buf_(std::move(o.buf_)), off_(o.off_), rows_(o.rows_), cols_(o.cols_), view_(o.view_) {
    o.view_ = nullptr;
    o.rows_ = o.cols_ = 0;
  }
  DeviceMatrix &operator=(const DeviceMatrix &o) {
    if (this != &o) {
      DeviceMatrix tmp(o);
      *this = std::move(tmp);
    }
    return *this;
  }
  DeviceMatrix &operator=(DeviceMatrix &&o) noexcept {
    if (this != &o) {
      unbind();
      buf_ = std::move(o.buf_);
      off_ = o.off_;
      rows_ = o.rows_;
      cols_ = o.cols_;
      view_ = o.view_;
      o.view_ = nullptr;
      o.rows_ = o.cols_ = 0;
    }
    return *this;
  }
  ~DeviceMatrix() { unbind(); }

  size_t rows() const { return rows_; }
  size_t cols() const { return cols_; }
  bool empty() const { return view_ == nullptr; }
  mi_vec *handle() const { return view_; }
  mi_ctx *context() const { return buf_ ? buf_->context() : nullptr; }

  // columns [j0, j0 + k) as a view sharing this matrix's storage
  DeviceMatrix cols_view(size_t j0, size_t k) const {
    DeviceMatrix v;
    v.buf_ = buf_;
    v.off_ = off_ + j0 * rows_;
    v.rows_ = rows_;
    v.cols_ = k;
    v.bind();
    return v;
  }
  DeviceMatrix leftCols(size_t k) const { return cols_view(0, k); }
  DeviceMatrix middleCols(size_t j0, size_t k) const { return cols_view(j0, k); }
  DeviceMatrix rightCols(size_t k) const { return cols_view(cols_ - k, k); }
  // this[:, j0 : j0+count) = src[:, i0 : i0+count)
  void set_cols(size_t j0, const DeviceMatrix &src, size_t i0, size_t count) {
    if (count == 0) return;
    DeviceMatrix d = cols_view(j0, count), s = src.cols_view(i0, count);
    check(mi_vec_copy(d.view_, s.view_));
  }
  void truncate_cols(size_t k) {  // X.conservativeResize(NoChange, nev)  LOBPCG.h:334
    cols_ = k;
    unbind();
    bind();
  }
  double norm() const {  // Frobenius
    double s = 0;
    check(mi_vec_dot(view_, view_, &s));
    return std::sqrt(s);
  }
  std::vector<double> to_host() const {  // column-major
    std::vector<double> h(rows_ * cols_);
    if (view_) check(mi_vec_download(view_, h.data(), h.size()));
    return h;
  }
  // uniform [-1, 1] entries like Eigen's Matrix::Random, generated on the host (LOBPCG.h:386)
  static DeviceMatrix Random(size_t m, size_t k) {
    mi_ctx *ctx = current_context_slot();
    if (!ctx) throw std::runtime_error("DeviceMatrix::Random: call MI355::make_current(ctx) first");
    std::vector<double> h(m * k);
    std::mt19937_64 gen(20260928);
    std::uniform_real_distribution<double> u(-1.0, 1.0);
    for (auto &x : h) x = u(gen);
    DeviceMatrix M(ctx, m, k);
    check(mi_vec_upload(M.buf_->handle(), h.data(), h.size()));
    return M;
  }

 private:
  void bind() {
    if (buf_ && rows_ * cols_ > 0) check(mi_vec_view(buf_->handle(), off_, rows_ * cols_, &view_));
  }
  void unbind() {
    if (view_) mi_vec_destroy(view_);
    view_ = nullptr;
  }
  std::shared_ptr<DeviceVector> buf_;
  size_t off_ = 0, rows_ = 0, cols_ = 0;
  mi_vec *view_ = nullptr;
};

// A panel as up to three column blocks that need not be adjacent (mi_panel_blocks): LOBPCG's search basis
// S = [X, W(:, nc:), P(:, nc:)] (LOBPCG.h:254-264) as the three blocks lie, instead of a copy that moves the unlocked
// columns together every iteration.  Holds VIEWS (they share the storage of the matrices they were cut from and keep
// it alive).
class PanelBlocks {
 public:
  void add(DeviceMatrix &&view) {
    if (view.cols() == 0) return;
    if (n_ == 3) throw std::invalid_argument("PanelBlocks: at most three column blocks");
    if (n_ > 0 && view.rows() != blk_[0].rows()) throw std::invalid_argument("PanelBlocks: row counts differ");
    blk_[n_++] = std::move(view);
  }
  size_t blocks() const { return n_; }
  size_t rows() const { return n_ ? blk_[0].rows() : 0; }
  size_t cols() const {
    size_t k = 0;
    for (size_t i = 0; i < n_; ++i) k += blk_[i].cols();
    return k;
  }
  mi_ctx *context() const { return n_ ? blk_[0].context() : nullptr; }
  const DeviceMatrix &block(size_t i) const { return blk_[i]; }
  // blocks [first, blocks()) (new views of the same storage)
  PanelBlocks from(size_t first) const {
    PanelBlocks r;
    for (size_t i = first; i < n_; ++i) r.add(blk_[i].leftCols(blk_[i].cols()));
    return r;
  }
  mi_panel_blocks raw() const {
    mi_panel_blocks b{};
    b.nblocks = (int)n_;
    for (size_t i = 0; i < n_; ++i) {
      b.block[i] = blk_[i].handle();
      b.cols[i] = (int)blk_[i].cols();
    }
    return b;
  }
  // the blocks copied together into one panel (for operators that are plain callables on a Matrix)
  DeviceMatrix assembled() const {
    DeviceMatrix M(context(), rows(), cols());
    size_t j = 0;
    for (size_t i = 0; i < n_; ++i) {
      M.set_cols(j, blk_[i], 0, blk_[i].cols());
      j += blk_[i].cols();
    }
    return M;
  }

 private:
  DeviceMatrix blk_[3];
  size_t n_ = 0;
};

// ---- panel operations used by LOBPCG (found by ADL) ------------------------------------------------

// G = S' T   (LOBPCG.h:223,271-272) -- fp64 MFMA kernel
inline HostMatrix gram(const DeviceMatrix &S, const DeviceMatrix &T) {
  HostMatrix G(S.cols(), T.cols());
  check(mi_lobpcg_gram(S.context(), S.rows(), (int)S.cols(), (int)T.cols(), S.handle(), T.handle(), G.data()));
  return G;
}
// G = S' [T1 | T2] for a panel held in two pieces (S.cols() == T1.cols() + T2.cols()): same bits as gram(S, T)
inline HostMatrix gram_split(const DeviceMatrix &S, const DeviceMatrix &T1, const DeviceMatrix &T2) {
  HostMatrix G(S.cols(), S.cols());
  check(mi_lobpcg_gram_split(S.context(), S.rows(), (int)S.cols(), S.handle(), (int)T1.cols(), T1.handle(),
                             T2.handle(), G.data()));
  return G;
}
// (S' [A1 | A2], S' [B1 | B2]) with one synchronisation; A2 / B2 may be empty (then A1 / B1 has S.cols() columns)
inline std::pair<HostMatrix, HostMatrix> gram_pair(const DeviceMatrix &S, const DeviceMatrix &A1,
                                                   const DeviceMatrix &A2, const DeviceMatrix &B1,
                                                   const DeviceMatrix &B2) {
  HostMatrix GA(S.cols(), S.cols()), GB(S.cols(), S.cols());
  const bool sa = A2.cols() > 0, sb = B2.cols() > 0;
  check(mi_lobpcg_gram_pair(S.context(), S.rows(), (int)S.cols(), S.handle(), (int)A1.cols(), A1.handle(),
                            sa ? A2.handle() : nullptr, (int)B1.cols(), B1.handle(), sb ? B2.handle() : nullptr,
                            GA.data(), GB.data()));
  return {std::move(GA), std::move(GB)};
}
// (S' [A1 | A2], S' S) for a SYMMETRIC product A1|A2 = A(S) (LOBPCG.h:271-272 without a B operator): one pass over S and
// A(S), the upper block triangle of both formed and mirrored (mi_lobpcg_gram_pair_sym); A2 may be empty
inline std::pair<HostMatrix, HostMatrix> gram_pair_sym(const DeviceMatrix &S, const DeviceMatrix &A1,
                                                       const DeviceMatrix &A2) {
  HostMatrix GA(S.cols(), S.cols()), GB(S.cols(), S.cols());
  check(mi_lobpcg_gram_pair_sym(S.context(), S.rows(), (int)S.cols(), S.handle(), (int)A1.cols(), A1.handle(),
                                A2.cols() > 0 ? A2.handle() : nullptr, GA.data(), GB.data()));
  return {std::move(GA), std::move(GB)};
}
// the same for a basis held as column blocks (mi_lobpcg_gram_pair_sym_blocks)
inline std::pair<HostMatrix, HostMatrix> gram_pair_sym(const PanelBlocks &S, const DeviceMatrix &A1,
                                                       const DeviceMatrix &A2) {
  const size_t k = S.cols();
  HostMatrix GA(k, k), GB(k, k);
  const mi_panel_blocks b = S.raw();
  check(mi_lobpcg_gram_pair_sym_blocks(S.context(), S.rows(), &b, (int)A1.cols(), A1.handle(),
                                       A2.cols() > 0 ? A2.handle() : nullptr, GA.data(), GB.data()));
  return {std::move(GA), std::move(GB)};
}
// (S'A(S), S'S) with A(S) as column blocks too (mi_lobpcg_gram_pair_sym_tblocks)
inline std::pair<HostMatrix, HostMatrix> gram_pair_sym(const PanelBlocks &S, const PanelBlocks &AS) {
  const size_t k = S.cols();
  HostMatrix GA(k, k), GB(k, k);
  const mi_panel_blocks s = S.raw(), a = AS.raw();
  check(mi_lobpcg_gram_pair_sym_tblocks(S.context(), S.rows(), &s, &a, GA.data(), GB.data()));
  return {std::move(GA), std::move(GB)};
}
// the generalized problem: (S'A(S), S'B(S)), everything as column blocks (mi_lobpcg_gram_pair_gen_blocks)
inline std::pair<HostMatrix, HostMatrix> gram_pair_gen(const PanelBlocks &S, const PanelBlocks &AS, const PanelBlocks &BS) {
  const size_t k = S.cols();
  HostMatrix GA(k, k), GB(k, k);
  const mi_panel_blocks s = S.raw(), a = AS.raw(), b = BS.raw();
  check(mi_lobpcg_gram_pair_gen_blocks(S.context(), S.rows(), &s, &a, &b, GA.data(), GB.data()));
  return {std::move(GA), std::move(GB)};
}
// Y = S C[row0 : row0+S.cols(), 0 : kc]   (LOBPCG.h:226-227,278,288)
inline DeviceMatrix times_small(const DeviceMatrix &S, const HostMatrix &C, size_t row0, size_t kc) {
  DeviceMatrix Y(S.context(), S.rows(), kc);
  check(mi_lobpcg_update(S.context(), S.rows(), (int)S.cols(), (int)kc, S.handle(), C.data() + row0,
                         (int)C.rows(), Y.handle()));
  return Y;
}
// X = S C[:, 0:nx]  and  P = S[:, nx:] C[nx:, 0:nx]  (LOBPCG.h:278 and :288) in ONE pass over the
// search basis S: the two products become one m x 2nx panel Y = S [C(:, :nx) | C0(:, :nx)], C0 = C with
// its first nx rows zeroed (the zero rows add exact +0.0 terms first, so P has the bits of the separate
// product).  X and P are returned as the two column-block views of Y.
inline void ritz_update(const DeviceMatrix &S, const HostMatrix &C, size_t nx, DeviceMatrix &X, DeviceMatrix &P) {
  const size_t ns = S.cols();
  HostMatrix C2(ns, 2 * nx);
  for (size_t j = 0; j < nx; ++j)
    for (size_t i = 0; i < ns; ++i) {
      C2(i, j) = C(i, j);
      C2(i, nx + j) = i < nx ? 0.0 : C(i, j);
    }
  DeviceMatrix Y(S.context(), S.rows(), 2 * nx);
  check(mi_lobpcg_update(S.context(), S.rows(), (int)ns, (int)(2 * nx), S.handle(), C2.data(), (int)ns, Y.handle()));
  X = Y.leftCols(nx);
  P = Y.middleCols(nx, nx);
}
// The same straight into two caller-supplied panels (views of the NEXT search basis): no copy of X and P into the
// basis afterwards (LOBPCG.h:254-259)
inline void ritz_update_into(const DeviceMatrix &S, const HostMatrix &C, size_t nx, DeviceMatrix &X, DeviceMatrix &P) {
  const size_t ns = S.cols();
  HostMatrix C2(ns, 2 * nx);
  for (size_t j = 0; j < nx; ++j)
    for (size_t i = 0; i < ns; ++i) {
      C2(i, j) = C(i, j);
      C2(i, nx + j) = i < nx ? 0.0 : C(i, j);
    }
  check(mi_lobpcg_update2(S.context(), S.rows(), (int)ns, (int)(2 * nx), S.handle(), C2.data(), (int)ns, X.handle(),
                          (int)nx, P.handle()));
}
// ... and for a basis held as column blocks (mi_lobpcg_update2_blocks)
inline void ritz_update_into(const PanelBlocks &S, const HostMatrix &C, size_t nx, DeviceMatrix &X, DeviceMatrix &P) {
  const size_t ns = S.cols();
  HostMatrix C2(ns, 2 * nx);
  for (size_t j = 0; j < nx; ++j)
    for (size_t i = 0; i < ns; ++i) {
      C2(i, j) = C(i, j);
      C2(i, nx + j) = i < nx ? 0.0 : C(i, j);
    }
  const mi_panel_blocks b = S.raw();
  check(mi_lobpcg_update2_blocks(S.context(), S.rows(), &b, (int)(2 * nx), C2.data(), (int)ns, X.handle(), (int)nx,
                                 P.handle()));
}
// The sparse operator of a LOBPCG client as a TAGGED callable (put it into the SymmetricLinearOperator<DeviceMatrix>
// argument `A`): used as a plain callable it is Y = A X (mi_csr_spmm_colmajor); the device LOBPCG loop recognises it
// (std::function::target) and, when B is absent, asks it for A(X) of the new Ritz block TOGETHER with the residual
// and its norms (LOBPCG.h:281,285,293,302; mi_csr_spmm_colmajor_residual: one pass over X instead of two).
struct DeviceCsrPanelOperator {
  mi_csr *A = nullptr;
  DeviceMatrix operator()(const DeviceMatrix &X) const {
    DeviceMatrix Y(X.context(), X.rows(), X.cols());
    check(mi_csr_spmm_colmajor(A, (int)X.cols(), X.handle(), Y.handle()));
    return Y;
  }
  // A [block 0 | block 1 | ...] of a panel held as column blocks (mi_csr_spmm_colmajor_blocks)
  DeviceMatrix operator()(const PanelBlocks &X) const {
    DeviceMatrix Y(X.context(), X.rows(), X.cols());
    const mi_panel_blocks b = X.raw();
    check(mi_csr_spmm_colmajor_blocks(A, &b, Y.handle()));
    return Y;
  }
};
// AX = A X and R = AX - X diag(theta) with the column norms of R and X, R into a caller-supplied panel
inline DeviceMatrix apply_with_residual_into(const DeviceCsrPanelOperator &op, const DeviceMatrix &X,
                                             const HostVectorD &theta, DeviceMatrix &R, HostVectorD &rnorm,
                                             HostVectorD &xnorm) {
  const size_t nx = X.cols();
  rnorm.resize(nx);
  xnorm.resize(nx);
  DeviceMatrix AX(X.context(), X.rows(), nx);
  check(mi_csr_spmm_colmajor_residual(op.A, (int)nx, X.handle(), theta.data(), AX.handle(), R.handle(), rnorm.data(),
                                      xnorm.data()));
  return AX;
}
// R = AX - BX diag(theta) into a caller-supplied panel (a view of the next basis' second block)
inline void residual_and_norms_into(DeviceMatrix &R, const DeviceMatrix &AX, const DeviceMatrix &BX,
                                    const DeviceMatrix &X, const HostVectorD &theta, HostVectorD &rnorm,
                                    HostVectorD &xnorm) {
  const size_t nx = X.cols();
  rnorm.resize(nx);
  xnorm.resize(nx);
  check(mi_lobpcg_residual(X.context(), X.rows(), (int)nx, AX.handle(), BX.handle(), X.handle(), theta.data(),
                           R.handle(), rnorm.data(), xnorm.data()));
}
// R = AX - BX diag(theta); returns R and fills the column norms of R and X  (LOBPCG.h:230,285,293,302)
inline DeviceMatrix residual_and_norms(const DeviceMatrix &AX, const DeviceMatrix &BX, const DeviceMatrix &X,
                                       const HostVectorD &theta, HostVectorD &rnorm, HostVectorD &xnorm) {
  const size_t nx = X.cols();
  DeviceMatrix R(X.context(), X.rows(), nx);
  rnorm.resize(nx);
  xnorm.resize(nx);
  check(mi_lobpcg_residual(X.context(), X.rows(), (int)nx, AX.handle(), BX.handle(), X.handle(), theta.data(),
                           R.handle(), rnorm.data(), xnorm.data()));
  return R;
}
// N(0,1) probe matrix drawn exactly like LOBPCG.h:205-211: default-seeded std::default_random_engine,
// filled row by row (i outer, j inner) on the host, then uploaded
inline DeviceMatrix gaussian_probe(const DeviceMatrix &like, size_t m, size_t nx) {
  std::default_random_engine gen;
  std::normal_distribution<double> normal(0, 1.0);
  std::vector<double> h(m * nx);
  for (size_t i = 0; i < m; ++i)
    for (size_t j = 0; j < nx; ++j) h[i + j * m] = normal(gen);
  DeviceMatrix Om(like.context(), m, nx);
  check(mi_vec_upload(Om.handle(), h.data(), h.size()));
  return Om;
}
inline DeviceMatrix empty_panel(const DeviceMatrix &like, size_t m, size_t k) {
  return DeviceMatrix(like.context(), m, k);
}

// Rayleigh-Ritz on host matrices (LOBPCG.h:53-62)
inline std::pair<HostVectorD, HostMatrix> rayleigh_ritz(const HostMatrix &A, const HostMatrix &B) {
  const size_t n = A.rows();
  HostVectorD theta(n);
  HostMatrix C(n, n);
  check(mi_rayleigh_ritz((int)n, A.data(), B.data(), theta.data(), C.data()));
  return std::make_pair(std::move(theta), std::move(C));
}

// the k lowest Ritz pairs only (what an LOBPCG iteration reads: LOBPCG.h:278,288,293-318); Theta has k entries, C is
// n x k.  Same reduction and QL recurrence as rayleigh_ritz: the Ritz values have its bits, the vectors agree to rounding.
inline std::pair<HostVectorD, HostMatrix> rayleigh_ritz_lowest(const HostMatrix &A, const HostMatrix &B, size_t k) {
  const size_t n = A.rows();
  if (k > n) k = n;
  HostVectorD theta(k);
  HostMatrix C(n, k);
  check(mi_rayleigh_ritz_lowest((int)n, (int)k, A.data(), B.data(), theta.data(), C.data()));
  return std::make_pair(std::move(theta), std::move(C));
}

}  // namespace MI355
}  // namespace Optimization
