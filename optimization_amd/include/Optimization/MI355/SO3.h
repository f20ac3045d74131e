// Optimization/MI355/SO3.h -- ready-made device callables for chordal rotation averaging on SO(3)^N
//     f(R) = 1/2 sum_e w_e | R_j - R_i Rt_e |_F^2
// in the shape the reference's templates expect (Objective, QuadraticModel, RiemannianMetric,
// Retraction, and a LinearOperator preconditioner; Optimization/Riemannian/Concepts.h).  Variable =
// N row-major 3x3 blocks (9N doubles), Tangent = so(3)^N coordinates (3N doubles), metric =
// coordinate dot product, retraction R_i exp(hat(xi_i)), preconditioner = 3x3 block-Jacobi built from
// the Hessian's diagonal blocks.  Kernels: optimization_amd/csrc/so3.hip.
#pragma once

#include <cstdint>

#include "Optimization/MI355/Device.h"
#include "Optimization/Riemannian/Concepts.h"

namespace Optimization {
namespace MI355 {

class RotationAveraging {
 public:
  using Vector = DeviceVector;

  RotationAveraging(const Context &ctx, size_t N, size_t n_edges, const int32_t *ei, const int32_t *ej,
                    const double *Rt, const double *w)
      : ctx_(ctx), N_(N) {
    check(mi_so3n_create(ctx_.get(), N, n_edges, ei, ej, Rt, w, &prob_));
  }
  RotationAveraging(const RotationAveraging &) = delete;
  RotationAveraging &operator=(const RotationAveraging &) = delete;
  ~RotationAveraging() {
    if (prob_) mi_so3n_destroy(prob_);
  }

  size_t rotations() const { return N_; }

  Objective<Vector, double> objective() {
    return DeviceObjective{this, [this](const Vector &R) {
                             double f = 0;
                             check(mi_so3n_objective(prob_, R.handle(), &f));
                             return f;
                           }};
  }
  // gradient in so(3)^N coordinates + the 3x3-block sparse Hessian assembled at R; also refreshes the
  // block-Jacobi preconditioner returned by preconditioner()
  Riemannian::QuadraticModel<Vector, Vector> quadratic_model() {
    return [this](const Vector &R, Vector &grad, Riemannian::LinearOperator<Vector, Vector> &Hess) {
      if (grad.empty() || grad.size() != 3 * N_) grad = Vector(ctx_, 3 * N_);
      mi_op *op = nullptr;
      check(mi_so3n_model(prob_, R.handle(), grad.handle(), &op, &bj_));
      Hess = DeviceHessian{op, this};
    };
  }
  Riemannian::RiemannianMetric<Vector, Vector, double> metric() { return FrobeniusMetric{}; }
  // R_i exp(hat(xi_i)) -- tagged: TNT evaluates a whole trial step (retraction, f at the trial point, the
  // predicted-decrease terms, the model and both gradient norms at the trial point) through mi_so3n_trial, one read-back
  Riemannian::Retraction<Vector, Vector> retraction() {
    DeviceTrialRetraction r;
    r.owner = this;
    r.retract = [this](const Vector &R, const Vector &xi) {
      Vector Y = Vector::like(R);
      check(mi_so3n_retract(prob_, R.handle(), xi.handle(), Y.handle()));
      return Y;
    };
    r.trial = [this](const Vector &R, const Vector &h, const Vector &g, bool with_precon) {
      DeviceTrialRetraction::Trial t;
      t.x_trial = Vector::like(R);
      double out[6];
      check(mi_so3n_trial(prob_, R.handle(), h.handle(), g.handle(), with_precon ? 1 : 0, t.x_trial.handle(), out));
      t.f_trial = out[0];
      t.hh = out[1];
      t.gh = out[2];
      t.hHh = out[3];
      t.grad_trial_sqnorm = out[4];
      t.precon_grad_trial_sqnorm = out[5];
      return t;
    };
    return r;
  }
  // the same without the tag (one call per statement of the reference's loop)
  Riemannian::Retraction<Vector, Vector> plain_retraction() {
    return [this](const Vector &R, const Vector &xi) {
      Vector Y = Vector::like(R);
      check(mi_so3n_retract(prob_, R.handle(), xi.handle(), Y.handle()));
      return Y;
    };
  }
  // 3x3 block-Jacobi; valid after the first quadratic_model() call (TNT calls QM before precon)
  Riemannian::LinearOperator<Vector, Vector> preconditioner() {
    if (!bj_) {
      // bind the (problem-owned) handle now; its inverse blocks are refreshed by every model call
      Vector tmpR(ctx_, 9 * N_), tmpg(ctx_, 3 * N_);
      check(mi_vec_fill(tmpR.handle(), 0.0));
      mi_op *op = nullptr;
      check(mi_so3n_model(prob_, tmpR.handle(), tmpg.handle(), &op, &bj_));
    }
    return DevicePreconditioner{bj_, this};
  }

 private:
  Context ctx_;
  size_t N_;
  mi_so3n *prob_ = nullptr;
  mi_precon *bj_ = nullptr;
};

}  // namespace MI355
}  // namespace Optimization
