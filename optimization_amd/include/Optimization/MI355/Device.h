// Optimization/MI355/Device.h -- the device-side "Vector" of the MI355X build and the tagged
// callables through which the generic templates (TNT, STPCG, ...) recognise that they can run
// their fused HIP path.  Thin C++17 over the C ABI of include/mi355opt.h; g++- and clang-compilable;
// no HIP headers needed by client code.
//
//   DeviceVector     satisfies the implicit Vector concept of the reference's templates (SURVEY.md
//                    Appendix A: `0 * g`, copy, unary minus, `*=`, `+=`, `-=`, `a * v`, `+`, `-`,
//                    `/ a`, `.dot`) with every operator ENQUEUED on the context stream; only
//                    `.dot()` / `.norm()` / host copies synchronise.
//   FrobeniusMetric / FrobeniusInnerProduct / DeviceHessian / DeviceOperator / DevicePreconditioner
//                    function objects to put inside the reference-style std::function arguments.
//                    The templates detect them with std::function::target<>() and then hand the
//                    whole inner loop to mi_stpcg (device-resident scalars, no host read-backs);
//                    any other callable still works through the generic (operator-by-operator)
//                    path, on the GPU, just with one synchronisation per inner product.
#pragma once

#include <cmath>
#include <cstddef>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <initializer_list>
#include <utility>
#include <vector>

#include "mi355opt.h"

namespace Optimization {
namespace MI355 {

// status -> exception: invalid arguments keep the reference's exception type
inline void check(int status) {
  if (status == MI_OK) return;
  const std::string msg = std::string(mi_status_string(status)) + ": " + mi_last_error();
  if (status == MI_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

// One GPU, one stream, one memory pool.  Shared by every DeviceVector created from it.
class Context {
 public:
  explicit Context(int device = 0) {
    mi_ctx *c = nullptr;
    check(mi_ctx_create(device, &c));
    h_ = std::shared_ptr<mi_ctx>(c, [](mi_ctx *p) { mi_ctx_destroy(p); });
  }
  // a context made elsewhere (e.g. one that already carries a communicator, mi_comm_init): not destroyed here
  static Context adopt(mi_ctx *c) {
    Context r{Adopt{}};
    r.h_ = std::shared_ptr<mi_ctx>(c, [](mi_ctx *) {});
    return r;
  }
  mi_ctx *get() const { return h_.get(); }
  void synchronize() const { check(mi_ctx_sync(h_.get())); }

 private:
  struct Adopt {};
  explicit Context(Adopt) {}
  std::shared_ptr<mi_ctx> h_;
};

class DeviceVector {
 public:
  DeviceVector() = default;  // `Vector v;` -- empty until assigned (IterativeSolvers.h:217-226)
  DeviceVector(const Context &ctx, size_t n) : ctx_(ctx.get()) { check(mi_vec_create(ctx_, n, &v_)); }
  DeviceVector(const Context &ctx, const std::vector<double> &host) : DeviceVector(ctx, host.size()) {
    check(mi_vec_upload(v_, host.data(), host.size()));
  }
  DeviceVector(const Context &ctx, const double *host, size_t n) : DeviceVector(ctx, n) {
    check(mi_vec_upload(v_, host, n));
  }
  DeviceVector(const DeviceVector &o) : ctx_(o.ctx_) {
    if (o.v_) {
      check(mi_vec_create(ctx_, o.size(), &v_));
      check(mi_vec_copy(v_, o.v_));
    }
  }
  DeviceVector(DeviceVector &&o) noexcept : ctx_(o.ctx_), v_(o.v_), borrowed_(o.borrowed_) {
    o.v_ = nullptr;
    o.borrowed_ = false;
  }
  ~DeviceVector() { release(); }

  DeviceVector &operator=(const DeviceVector &o) {
    if (this == &o) return *this;
    if (!o.v_) {
      release();
      ctx_ = o.ctx_;
      return *this;
    }
    if (!v_ || size() != o.size() || borrowed_) {
      release();
      ctx_ = o.ctx_;
      check(mi_vec_create(ctx_, o.size(), &v_));
    }
    check(mi_vec_copy(v_, o.v_));
    return *this;
  }
  DeviceVector &operator=(DeviceVector &&o) noexcept {
    if (this != &o) {
      release();
      ctx_ = o.ctx_;
      v_ = o.v_;
      borrowed_ = o.borrowed_;
      o.v_ = nullptr;
      o.borrowed_ = false;
    }
    return *this;
  }

  // non-owning view of a handle owned by the C layer (operator callbacks)
  static DeviceVector view(mi_ctx *ctx, mi_vec *v) {
    DeviceVector d;
    d.ctx_ = ctx;
    d.v_ = v;
    d.borrowed_ = true;
    return d;
  }

  bool empty() const { return v_ == nullptr; }
  size_t size() const {
    size_t n = 0;
    if (v_) check(mi_vec_len(v_, &n));
    return n;
  }
  mi_vec *handle() const { return v_; }
  mi_ctx *context() const { return ctx_; }

  std::vector<double> to_host() const {
    std::vector<double> h(size());
    if (v_) check(mi_vec_download(v_, h.data(), h.size()));
    return h;
  }

  // --- the Vector concept -----------------------------------------------------------------
  double dot(const DeviceVector &o) const {  // Riemannian/Concepts.h:178
    double out = 0;
    check(mi_vec_dot(v_, o.v_, &out));
    return out;
  }
  double squaredNorm() const { return dot(*this); }
  double norm() const;
  DeviceVector &operator+=(const DeviceVector &o) {
    check(mi_vec_axpy(v_, 1.0, o.v_));
    return *this;
  }
  DeviceVector &operator-=(const DeviceVector &o) {
    check(mi_vec_axpy(v_, -1.0, o.v_));
    return *this;
  }
  DeviceVector &operator*=(double a) {
    check(mi_vec_scale(v_, a));
    return *this;
  }
  DeviceVector &operator/=(double a) {  // `u /= beta` IterativeSolvers.h:653: a true division (mi_vec_div)
    check(mi_vec_div(v_, a));
    return *this;
  }
  // z = a x + b y without temporaries
  static DeviceVector axpby(double a, const DeviceVector &x, double b, const DeviceVector &y) {
    DeviceVector z = like(x);
    check(mi_vec_axpby(z.v_, a, x.v_, b, y.v_));
    return z;
  }
  static DeviceVector like(const DeviceVector &x) { return on(x.ctx_, x.size()); }
  // uninitialised vector of n doubles on the given raw context handle
  static DeviceVector on(mi_ctx *ctx, size_t n) {
    DeviceVector z;
    z.ctx_ = ctx;
    check(mi_vec_create(ctx, n, &z.v_));
    return z;
  }

 private:
  void release() {
    if (v_ && !borrowed_) mi_vec_destroy(v_);
    v_ = nullptr;
    borrowed_ = false;
  }
  mi_ctx *ctx_ = nullptr;
  mi_vec *v_ = nullptr;
  bool borrowed_ = false;
};

inline DeviceVector operator*(double a, const DeviceVector &v) {  // `0 * g`, `alpha * p`: one product per element
  DeviceVector z = DeviceVector::like(v);                        // (a*v + 0*v would turn Inf into NaN and -0.0 into +0.0)
  check(mi_vec_scale_to(z.handle(), a, v.handle()));
  return z;
}
inline DeviceVector operator*(const DeviceVector &v, double a) { return a * v; }
inline DeviceVector operator/(const DeviceVector &v, double a) {
  DeviceVector z(v);
  z /= a;
  return z;
}
inline DeviceVector operator+(const DeviceVector &a, const DeviceVector &b) {
  return DeviceVector::axpby(1.0, a, 1.0, b);
}
inline DeviceVector operator-(const DeviceVector &a, const DeviceVector &b) {
  return DeviceVector::axpby(1.0, a, -1.0, b);
}
inline DeviceVector operator-(const DeviceVector &a) { return DeviceVector::axpby(-1.0, a, 0.0, a); }
// rvalue overloads reuse the temporary's storage: `s + alpha * p`, `-v + beta * p`
inline DeviceVector operator+(const DeviceVector &a, DeviceVector &&b) {
  check(mi_vec_axpy(b.handle(), 1.0, a.handle()));
  return std::move(b);
}
inline DeviceVector operator+(DeviceVector &&a, const DeviceVector &b) {
  check(mi_vec_axpy(a.handle(), 1.0, b.handle()));
  return std::move(a);
}
inline DeviceVector operator+(DeviceVector &&a, DeviceVector &&b) {
  check(mi_vec_axpy(a.handle(), 1.0, b.handle()));
  return std::move(a);
}
inline DeviceVector operator-(DeviceVector &&a) {
  check(mi_vec_scale(a.handle(), -1.0));
  return std::move(a);
}
inline double DeviceVector::norm() const { return std::sqrt(squaredNorm()); }

// up to 4 inner products <x_i, y_i> with ONE device pass and ONE synchronisation (mi_vec_dot_batch)
inline std::vector<double> dot_batch(std::initializer_list<std::pair<const DeviceVector *, const DeviceVector *>> pairs) {
  std::vector<const mi_vec *> xs, ys;
  mi_ctx *ctx = nullptr;
  for (const auto &pr : pairs) {
    xs.push_back(pr.first->handle());
    ys.push_back(pr.second->handle());
    ctx = pr.first->context();
  }
  std::vector<double> out(xs.size(), 0.0);
  check(mi_vec_dot_batch(ctx, (int)xs.size(), xs.data(), ys.data(), out.data()));
  return out;
}

// A caller that is about to enqueue more device work right behind an inner solve and will read everything back with
// one wait (TNT's fused trial step) opens a DeferScope around its STPCG call: the fused solver (mi_stpcg) then returns
// without waiting for the device -- the step is valid in stream order -- and the scope's collect() delivers
// |s|_M and the iteration count afterwards (mi_stpcg_collect; no wait if something else has synchronised since).
// Thread-local, so concurrent optimizers on different contexts/threads do not see each other's requests.
struct DeferScope {
  explicit DeferScope(bool want) : prev_(slot()) { slot() = want ? this : nullptr; }
  ~DeferScope() { slot() = prev_; }
  DeferScope(const DeferScope &) = delete;
  DeferScope &operator=(const DeferScope &) = delete;
  static DeferScope *active() { return slot(); }
  void taken_on(mi_ctx *ctx) { ctx_ = ctx; }
  bool taken() const { return ctx_ != nullptr; }
  mi_stpcg_result collect() {
    mi_stpcg_result r{};
    check(mi_stpcg_collect(ctx_, &r));
    ctx_ = nullptr;
    return r;
  }

 private:
  static DeferScope *&slot() {
    static thread_local DeferScope *s = nullptr;
    return s;
  }
  DeferScope *prev_;
  mi_ctx *ctx_ = nullptr;
};

template <typename T>
struct is_device_vector : std::false_type {};
template <>
struct is_device_vector<DeviceVector> : std::true_type {};

// ---------------------------------------------------------------------------------------------
// tagged callables
// ---------------------------------------------------------------------------------------------

// metric(X, V1, V2) = <V1, V2>_F -- the embedded metric of Stiefel / product manifolds and the
// coordinate metric of so(3)^N.  Put it into RiemannianMetric<DeviceVector, DeviceVector, double, ...>.
struct FrobeniusMetric {
  template <typename... A>
  double operator()(const DeviceVector &, const DeviceVector &V1, const DeviceVector &V2, A &...) const {
    return V1.dot(V2);
  }
};
// inner_product(a, b) = <a, b>_F for LinearAlgebra::InnerProduct<DeviceVector, double, ...>
struct FrobeniusInnerProduct {
  template <typename... A>
  double operator()(const DeviceVector &a, const DeviceVector &b, A &...) const {
    return a.dot(b);
  }
};

// LinearAlgebra::SymmetricLinearOperator<DeviceVector, ...> backed by an mi_op
// (a rectangular operator -- a Jacobian, n_in -> n_out -- returns a vector of ITS output length)
inline DeviceVector apply_device_operator(mi_op *op, const DeviceVector &v) {
  size_t n_out = 0;
  check(mi_op_dims(op, nullptr, &n_out));
  DeviceVector out = n_out == v.size() ? DeviceVector::like(v) : DeviceVector::on(v.context(), n_out);
  check(mi_op_apply(op, v.handle(), out.handle()));
  return out;
}
struct DeviceOperator {
  mi_op *op = nullptr;
  template <typename... A>
  DeviceVector operator()(const DeviceVector &v, A &...) const {
    return apply_device_operator(op, v);
  }
};
// Callables handed out by a problem object (MI355/Stiefel.h, MI355/SO3.h) carry the address of that object as
// `owner`.  TNT / GradientDescent replace their statement-by-statement trial step by the owner's fused chain only
// when the objective, the model (through the Hessian it sets), the gradient field and the retraction ALL name the
// same owner -- a wrapped, penalised or logging objective, or callables of two different problem objects, keep the
// reference's statement sequence (which always calls the supplied f and metric).
struct DeviceObjective {
  const void *owner = nullptr;
  std::function<double(const DeviceVector &)> f;
  template <typename... A>
  double operator()(const DeviceVector &X, A &...) const {
    return f(X);
  }
};
struct DeviceGradientField {
  const void *owner = nullptr;
  std::function<DeviceVector(const DeviceVector &)> grad;
  template <typename... A>
  DeviceVector operator()(const DeviceVector &X, A &...) const {
    return grad(X);
  }
};

// Riemannian::Retraction whose owner can evaluate a WHOLE TRIAL STEP of a trust-region method in one launch chain
// with one read-back (reference Riemannian/TNT.h:493-512,573-585): the step norm, x_trial = retract(x, h), f(x_trial),
// <g,h>, <h, Hess h> and, speculatively, the gradient norm at x_trial.  TNT recognises it (std::function::target) and
// calls trial() instead of retract / f / metric one by one; used as a plain Retraction it retracts.
struct DeviceTrialRetraction {
  struct Trial {
    DeviceVector x_trial;
    double f_trial, hh, gh, hHh, grad_trial_sqnorm;
    // |M^-1 grad f(x_trial)|^2 with the OWNER's preconditioner rebuilt at x_trial (TNT.h:578-580), or < 0 when the
    // owner has none / it was not requested
    double precon_grad_trial_sqnorm = -1;
  };
  // the same for a backtracking line search along -g (Riemannian/GradientDescent.h:266-286): h = -t g, x_trial,
  // f(x_trial) and the squared gradient norm at x_trial, one read-back per trial
  struct ArmijoTrial {
    DeviceVector h, x_trial;
    double f_trial, grad_trial_sqnorm;
  };
  const void *owner = nullptr;
  std::function<DeviceVector(const DeviceVector &, const DeviceVector &)> retract;
  // with_precon: the caller runs TNT with the owner's preconditioner and wants |M^-1 grad|^2 at x_trial as well
  std::function<Trial(const DeviceVector &x, const DeviceVector &h, const DeviceVector &grad, bool with_precon)> trial;
  std::function<ArmijoTrial(const DeviceVector &x, const DeviceVector &grad, double t)> armijo;
  template <typename... A>
  DeviceVector operator()(const DeviceVector &X, const DeviceVector &V, A &...) const {
    return retract(X, V);
  }
};
// Riemannian::LinearOperator<DeviceVector, DeviceVector, ...> (the Hessian set by a QuadraticModel)
// backed by an mi_op that is already bound to the base point X
struct DeviceHessian {
  mi_op *op = nullptr;
  const void *owner = nullptr;
  template <typename... A>
  DeviceVector operator()(const DeviceVector &, const DeviceVector &v, A &...) const {
    return apply_device_operator(op, v);
  }
};
// Riemannian::LinearOperator used as TNT's `precon`, backed by an mi_precon bound to X
struct DevicePreconditioner {
  mi_precon *P = nullptr;
  const void *owner = nullptr;
  template <typename... A>
  DeviceVector operator()(const DeviceVector &, const DeviceVector &r, A &...) const {
    DeviceVector out = DeviceVector::like(r);
    check(mi_precon_apply(P, r.handle(), out.handle()));
    return out;
  }
};
// STPCGPreconditioner<DeviceVector, Multiplier, ...>: P(r) = (M^-1 r, Multiplier())
template <typename Multiplier>
struct DeviceSTPCGPreconditioner {
  mi_precon *P = nullptr;
  template <typename... A>
  std::pair<DeviceVector, Multiplier> operator()(const DeviceVector &r, A &...) const {
    DeviceVector out = DeviceVector::like(r);
    check(mi_precon_apply(P, r.handle(), out.handle()));
    return std::make_pair(std::move(out), Multiplier());
  }
};

// The constraint preconditioner of the PROJECTED solve and the transposed constraint operator handed to STPCG as
// `P` and `At` (reference IterativeSolvers.h:83-85,178, used at :229-253,381-405), both backed by ONE mi_precon made
// by mi_precon_create_constraint (diagonal M, dense m x n constraints): STPCG<DeviceVector, DeviceVector> recognises
// the pair and runs the whole projected solve through the fused device loop (mi_stpcg with constraint_At); used as
// plain callables they do what the reference's callables do.
struct DeviceConstraintPreconditioner {
  mi_precon *P = nullptr;
  size_t multipliers = 0;  // m
  template <typename... A>
  std::pair<DeviceVector, DeviceVector> operator()(const DeviceVector &r, A &...) const {
    DeviceVector v = DeviceVector::like(r), l = DeviceVector::on(r.context(), multipliers);
    check(mi_precon_constraint_solve(P, r.handle(), v.handle(), l.handle()));
    return std::make_pair(std::move(v), std::move(l));
  }
};
struct DeviceConstraintTranspose {
  mi_precon *P = nullptr;
  size_t n = 0;
  template <typename... A>
  DeviceVector operator()(const DeviceVector &l, A &...) const {
    DeviceVector out = DeviceVector::on(l.context(), n);
    check(mi_precon_constraint_At(P, l.handle(), out.handle()));
    return out;
  }
};

}  // namespace MI355
}  // namespace Optimization
