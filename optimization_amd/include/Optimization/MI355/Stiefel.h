// Optimization/MI355/Stiefel.h -- ready-made device callables for optimisation on the Stiefel
// manifold St(n,p) (p <= 8, embedded metric, polar retraction), in the shape the reference's
// templates expect (Objective, QuadraticModel, RiemannianMetric, Retraction of
// Optimization/Riemannian/Concepts.h).  The reference ships no manifold code beyond the S^2 lambdas
// of its tests (tests/TNT_unit_test.cpp:73-117); this is their n x p, GPU-resident generalisation,
// backed by the kernels of optimization_amd/csrc/stiefel.hip through the C ABI.
//
//   StiefelRayleighQuotient prob(ctx, n, p, rowptr, col, val);          // f(X) = 1/2 tr(X' A X)
//   auto result = Optimization::Riemannian::TNT<DeviceVector, DeviceVector>(
//       prob.objective(), prob.quadratic_model(), prob.metric(), prob.retraction(), X0,
//       std::optional<Optimization::Riemannian::LinearOperator<DeviceVector, DeviceVector>>(), params);
//   (as with the reference, an empty Args pack needs a TYPED empty optional for `precon`)
#pragma once

#include <cstdint>

#include "Optimization/MI355/Device.h"
#include "Optimization/Riemannian/Concepts.h"

namespace Optimization {
namespace MI355 {

class StiefelRayleighQuotient {
 public:
  using Vector = DeviceVector;

  StiefelRayleighQuotient(const Context &ctx, size_t n, int p, const int32_t *rowptr, const int32_t *col,
                          const double *val)
      : ctx_(ctx), n_(n), p_(p) {
    check(mi_csr_create(ctx_.get(), n, (size_t)rowptr[n], rowptr, col, val, &A_));
    check(mi_stiefel_rq_create(ctx_.get(), A_, n, p, &prob_));
  }
  StiefelRayleighQuotient(const StiefelRayleighQuotient &) = delete;
  StiefelRayleighQuotient &operator=(const StiefelRayleighQuotient &) = delete;
  ~StiefelRayleighQuotient() {
    if (jacobi_) mi_precon_destroy(jacobi_);
    if (prob_) mi_stiefel_rq_destroy(prob_);
    if (A_) mi_csr_destroy(A_);
  }

  size_t rows() const { return n_; }
  int cols() const { return p_; }
  const Context &context() const { return ctx_; }

  // f(X) = 1/2 tr(X' A X)
  // (tagged with this object as owner: TNT / GradientDescent fuse the trial step only when objective, model,
  // gradient field and retraction all come from the same problem object -- MI355/Device.h)
  Objective<Vector, double> objective() {
    return DeviceObjective{this, [this](const Vector &X) {
                             double f = 0;
                             check(mi_stiefel_rq_objective(prob_, X.handle(), &f));
                             return f;
                           }};
  }
  // grad f(X) = A X - X sym(X'AX);  Hess f(X)[V] = P_X(A V - V sym(X'AX)) as a device operator
  Riemannian::QuadraticModel<Vector, Vector> quadratic_model() {
    return [this](const Vector &X, Vector &grad, Riemannian::LinearOperator<Vector, Vector> &Hess) {
      if (grad.empty() || grad.size() != n_ * (size_t)p_) grad = Vector(ctx_, n_ * (size_t)p_);
      mi_op *op = nullptr;
      check(mi_stiefel_rq_model(prob_, X.handle(), grad.handle(), &op));
      Hess = DeviceHessian{op, this};
    };
  }
  Riemannian::RiemannianMetric<Vector, Vector, double> metric() { return FrobeniusMetric{}; }
  // polar retraction (X + V) ((X+V)'(X+V))^-1/2 -- tagged: TNT evaluates a whole trial step (retraction, f at the
  // trial point, the predicted-decrease terms, the next gradient) through mi_stiefel_rq_trial, one read-back
  Riemannian::Retraction<Vector, Vector> retraction() {
    DeviceTrialRetraction r;
    r.owner = this;
    r.retract = [this](const Vector &X, const Vector &V) {
      Vector Y = Vector::like(X);
      check(mi_stiefel_retract(ctx_.get(), n_, p_, X.handle(), V.handle(), Y.handle()));
      return Y;
    };
    r.trial = [this](const Vector &X, const Vector &h, const Vector &g, bool /*with_precon: none owned*/) {
      DeviceTrialRetraction::Trial t;
      t.x_trial = Vector::like(X);
      double out[5];
      check(mi_stiefel_rq_trial(prob_, X.handle(), h.handle(), g.handle(), t.x_trial.handle(), out));
      t.f_trial = out[0];
      t.hh = out[1];
      t.gh = out[2];
      t.hHh = out[3];
      t.grad_trial_sqnorm = out[4];
      return t;
    };
    r.armijo = [this](const Vector &X, const Vector &g, double t) {
      DeviceTrialRetraction::ArmijoTrial a;
      a.h = Vector::like(X);
      a.x_trial = Vector::like(X);
      double out[2];
      check(mi_stiefel_rq_armijo_trial(prob_, X.handle(), g.handle(), t, a.h.handle(), a.x_trial.handle(), out));
      a.f_trial = out[0];
      a.grad_trial_sqnorm = out[1];
      return a;
    };
    return r;
  }
  // grad f(X) as a VectorField (GradientDescent's interface); after a fused trial at X it is already there
  Riemannian::VectorField<Vector, Vector> gradient() {
    return DeviceGradientField{this, [this](const Vector &X) {
                                 Vector grad(ctx_, n_ * (size_t)p_);
                                 check(mi_stiefel_rq_model(prob_, X.handle(), grad.handle(), nullptr));
                                 return grad;
                               }};
  }
  // the same without the tag (one call per statement of the reference's loop)
  Riemannian::Retraction<Vector, Vector> plain_retraction() {
    return [this](const Vector &X, const Vector &V) {
      Vector Y = Vector::like(X);
      check(mi_stiefel_retract(ctx_.get(), n_, p_, X.handle(), V.handle(), Y.handle()));
      return Y;
    };
  }
  // tangent-space projection P_X(Z) = Z - X sym(X'Z)
  Vector project(const Vector &X, const Vector &Z) {
    Vector out = Vector::like(Z);
    check(mi_stiefel_project(ctx_.get(), n_, p_, X.handle(), Z.handle(), out.handle()));
    return out;
  }

 private:
  Context ctx_;
  size_t n_;
  int p_;
  mi_csr *A_ = nullptr;
  mi_stiefel_rq *prob_ = nullptr;
  mi_precon *jacobi_ = nullptr;
};

}  // namespace MI355
}  // namespace Optimization
