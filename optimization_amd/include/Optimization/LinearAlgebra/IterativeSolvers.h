// Optimization/LinearAlgebra/IterativeSolvers.h -- drop-in for the reference header of the same
// path: Steihaug-Toint truncated preconditioned projected CG (STPCG) and trust-region LSQR.
//
//   reference: include/Optimization/LinearAlgebra/IterativeSolvers.h
//              STPCGUserFunction :50-59, STPCGPreconditioner :83-85, STPCG :166-426,
//              LSQRUserFunction :450-456, LSQR :552-855, same-type LSQR sugar :859-875
//
// MI355X build.  Written from scratch against that interface: same template parameters, argument
// order, defaults, exceptions and -- statement for statement -- the same floating-point recurrences,
// so that a host Vector reproduces the reference's iterates bit for bit (tests/ compare with the
// reference compiled from its own headers).  When Vector is MI355::DeviceVector and the callables
// are the tagged device function objects of Optimization/MI355/Device.h, STPCG hands the whole loop
// to the fused HIP implementation (mi_stpcg: 4 streaming kernels per iteration, device-resident
// scalars, no host read-back inside the loop); every other combination runs the generic loop below
// through the Vector's operators.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <optional>
#include <stdexcept>
#include <tuple>
#include <type_traits>
#include <utility>

#include "Optimization/LinearAlgebra/Concepts.h"

#if __has_include("mi355opt.h")
#include "Optimization/MI355/Device.h"
#define OPTIMIZATION_HAVE_MI355 1
#else
#define OPTIMIZATION_HAVE_MI355 0
#endif

namespace Optimization {
namespace LinearAlgebra {

// Observer invoked once per completed STPCG pass, after alpha_k is known and before the update is
// applied; returning true stops the solver with s_k un-updated.   (reference :50-59, :365-369)
template <typename Vector, typename Multiplier, typename Scalar = double, typename... Args>
using STPCGUserFunction = std::function<bool(
    size_t k, const Vector &g, const SymmetricLinearOperator<Vector, Args...> &H,
    const std::optional<LinearOperator<Vector, std::pair<Vector, Multiplier>, Args...>> &P,
    const std::optional<LinearOperator<Multiplier, Vector, Args...>> &At, const Vector &sk,
    const Vector &rk, const Vector &vk, const Vector &pk, Scalar alpha_k, Args &...args)>;

// Constraint preconditioner: P(s) = (v, lambda) with [M A'; A 0] [v; lambda] = [s; 0]
// (reference :83-85).  Without constraints this is v = M^-1 s and lambda is ignored.
template <typename Vector, typename Multiplier, typename... Args>
using STPCGPreconditioner = LinearOperator<Vector, std::pair<Vector, Multiplier>, Args...>;

namespace detail {

// Step to the trust-region boundary along p from s, in the M-norm, from the three running
// M-inner-products (reference :330-332 and :355-357 -- one formula, used twice).
template <typename Scalar>
inline Scalar boundary_steplength(Scalar s_M_p, Scalar p_M_p, Scalar s_M_s, Scalar Delta_sq) {
  return (-s_M_p + sqrt(s_M_p * s_M_p + p_M_p * (Delta_sq - s_M_s))) / p_M_p;
}

#if OPTIMIZATION_HAVE_MI355
// Fused device path.  Returns true (and fills the outputs) if the operands qualify.
template <typename Vector, typename Multiplier, typename Scalar, typename... Args>
bool stpcg_on_device(const Vector &g, const SymmetricLinearOperator<Vector, Args...> &H,
                     const InnerProduct<Vector, Scalar, Args...> &inner_product,
                     const std::optional<STPCGPreconditioner<Vector, Multiplier, Args...>> &P,
                     const std::optional<LinearOperator<Multiplier, Vector, Args...>> &At,
                     Scalar Delta, size_t max_iterations, Scalar kappa_fgr, Scalar theta, Scalar epsilon,
                     Vector &s_out, Scalar &update_step_M_norm, size_t &num_iterations, const char *&why) {
  // `why`: which probe sent the solve to the generic loop (reported through mi_ctx_note_generic by the caller)
  if constexpr (!MI355::is_device_vector<Vector>::value || !std::is_same<Scalar, double>::value) {
    why = "Scalar is not double";
    return false;
  } else {
    using namespace MI355;
    auto no = [&why](const char *w) {
      why = w;
      return false;
    };
    if (g.empty()) return no("the gradient is an empty vector");
    if (!inner_product.template target<FrobeniusInnerProduct>())
      return no("the inner product is not MI355::FrobeniusInnerProduct (a plain callable, or a tagged one wrapped in a "
                "lambda)");
    const DeviceOperator *dop = H.template target<DeviceOperator>();
    if (!dop || !dop->op)
      return no("the operator H is not a MI355::DeviceOperator (a plain callable, or a tagged one wrapped in a lambda)");
    mi_precon *prec = nullptr;
    bool constraint_At = false;
    if (P) {
      if (const auto *dp = P->template target<DeviceSTPCGPreconditioner<Multiplier>>()) {
        // (an ordinary preconditioner has no multipliers for `At` to act on)
        if (!dp->P || At) return no("an ordinary device preconditioner was passed together with At");
        prec = dp->P;
      } else if constexpr (is_device_vector<Multiplier>::value) {
        // projected solve (:229-253,381-405): constraint preconditioner and A' of ONE device KKT object
        const auto *cp = P->template target<DeviceConstraintPreconditioner>();
        if (!cp || !cp->P)
          return no("the preconditioner is neither a MI355::DeviceSTPCGPreconditioner nor a "
                    "MI355::DeviceConstraintPreconditioner");
        if (At) {
          const auto *ct = At->template target<DeviceConstraintTranspose>();
          if (!ct || ct->P != cp->P)
            return no("At is not the MI355::DeviceConstraintTranspose of the constraint preconditioner's KKT object");
          constraint_At = true;
        }
        prec = cp->P;
      } else {
        return no("the preconditioner is not a MI355::DeviceSTPCGPreconditioner");
      }
    } else if (At) {
      return no("At was passed without a constraint preconditioner");
    }
    mi_stpcg_params prm;
    mi_stpcg_default_params(&prm);
    prm.Delta = Delta;
    prm.max_iterations = max_iterations;
    prm.kappa_fgr = kappa_fgr;
    prm.theta = theta;
    prm.epsilon = epsilon;
    prm.constraint_At = constraint_At ? 1 : 0;
    // a caller that reads the result together with its own next chain (TNT's fused trial step): no wait here
    DeferScope *defer = DeferScope::active();
    prm.defer_result = defer ? 1 : 0;
    mi_stpcg_result res;
    s_out = DeviceVector::like(g);
    check(mi_stpcg(g.context(), g.handle(), dop->op, prec, &prm, s_out.handle(), &res, nullptr));
    if (defer) defer->taken_on(g.context());
    update_step_M_norm = res.update_step_M_norm;  // (placeholders in deferred mode: DeferScope::collect)
    num_iterations = res.num_iterations;
    return true;
  }
}

// Fused device LSQR (mi_lsqr): both operators tagged DeviceOperator, both inner products Frobenius.
template <typename VectorX, typename VectorY, typename Scalar, typename... Args>
bool lsqr_on_device(const LinearOperator<VectorX, VectorY, Args...> &A,
                    const LinearOperator<VectorY, VectorX, Args...> &At, const VectorY &b,
                    const InnerProduct<VectorX, Scalar, Args...> &ipx,
                    const InnerProduct<VectorY, Scalar, Args...> &ipy, size_t max_iterations, Scalar lambda,
                    Scalar btol, Scalar Atol, Scalar Abar_cond_limit, Scalar Delta, VectorX &x_out, Scalar &xnorm,
                    size_t &num_iterations, const char *&why) {
  if constexpr (!MI355::is_device_vector<VectorX>::value || !MI355::is_device_vector<VectorY>::value ||
                !std::is_same<Scalar, double>::value || sizeof...(Args) != 0) {
    why = "extra arguments (Args...), a Scalar other than double, or a host vector type";
    return false;
  } else {
    using namespace MI355;
    auto no = [&why](const char *w) {
      why = w;
      return false;
    };
    if (b.empty()) return no("b is an empty vector");
    if (!ipx.template target<FrobeniusInnerProduct>() || !ipy.template target<FrobeniusInnerProduct>())
      return no("an inner product is not MI355::FrobeniusInnerProduct (a plain callable, or a tagged one wrapped in a "
                "lambda)");
    const DeviceOperator *da = A.template target<DeviceOperator>();
    const DeviceOperator *dat = At.template target<DeviceOperator>();
    if (!da || !da->op || !dat || !dat->op)
      return no("A or At is not a MI355::DeviceOperator (a plain callable, or a tagged one wrapped in a lambda)");
    size_t nx = 0;
    check(mi_op_dims(dat->op, nullptr, &nx));
    mi_lsqr_params prm;
    mi_lsqr_default_params(&prm);
    prm.max_iterations = max_iterations;
    prm.lambda = lambda;
    prm.btol = btol;
    prm.Atol = Atol;
    prm.Acond_limit = Abar_cond_limit;
    prm.Delta = Delta;
    mi_lsqr_result res;
    x_out = DeviceVector::on(b.context(), nx);
    check(mi_lsqr(b.context(), da->op, dat->op, b.handle(), &prm, x_out.handle(), &res));
    xnorm = res.xnorm;
    num_iterations = res.num_iterations;
    return true;
  }
}
#endif

}  // namespace detail

// Approximately solves   min_s <g,s> + 1/2 <s,Hs>   s.t.  A s = 0,  |s|_M <= Delta
// by the Steihaug-Toint truncated preconditioned projected conjugate-gradient method
// (Conn-Gould-Toint Alg. 5.4.2 / 7.5.1; Gould-Hribar-Nocedal Alg. 6.1 / 6.3).       reference :166-426
//
//   update_step_M_norm  out: |s|_M of the returned step (= Delta on boundary exits)
//   num_iterations      out: completed passes (a boundary exit inside pass k reports k)
//   termination         sqrt(<r,v>) <= |r0|_P min(kappa_fgr, |r0|_P^theta), negative curvature /
//                       boundary, p in ker(H) up to epsilon, max_iterations, or the user function
//   throws std::invalid_argument for Delta <= 0, kappa_fgr not in [0,1), theta not in [0,1],
//   epsilon not in (0,1)                                                              reference :183-205
template <typename Vector, typename Multiplier, typename Scalar = double, typename... Args>
Vector STPCG(const Vector &g, const SymmetricLinearOperator<Vector, Args...> &H,
             const InnerProduct<Vector, Scalar, Args...> &inner_product, Args &...args,
             Scalar &update_step_M_norm, size_t &num_iterations, Scalar Delta,
             size_t max_iterations = 1000, Scalar kappa_fgr = .1, Scalar theta = .5,
             const std::optional<STPCGPreconditioner<Vector, Multiplier, Args...>> &P = std::nullopt,
             const std::optional<LinearOperator<Multiplier, Vector, Args...>> &At = std::nullopt,
             const std::optional<STPCGUserFunction<Vector, Multiplier, Scalar, Args...>> &user_function =
                 std::nullopt,
             Scalar epsilon = 1e-8) {
  if (Delta <= 0)
    throw std::invalid_argument("Trust-region radius (Delta) must be a positive real value");
  if ((kappa_fgr < 0) || (kappa_fgr >= 1))
    throw std::invalid_argument("Target fractional reduction of the gradient norm (kappa_fgr) must "
                                "be a real value in the range [0,1)");
  if ((theta < 0) || (theta > 1))
    throw std::invalid_argument("Target superlinear convergence rate (theta) must be a real value "
                                "in the range [0,1]");
  if ((epsilon <= 0) || (epsilon >= 1))
    throw std::invalid_argument("Relative norm tolerance for declaring a vector to lie in the kernel "
                                "of H (epsilon) should be a small positive number in the range (0,1)");

#if OPTIMIZATION_HAVE_MI355
  if constexpr (MI355::is_device_vector<Vector>::value) {
    const char *why = "extra arguments (Args...) are passed to the callables";
    if constexpr (sizeof...(Args) == 0) {
      if (!user_function) {
        Vector s_dev;
        if (detail::stpcg_on_device<Vector, Multiplier, Scalar>(g, H, inner_product, P, At, Delta, max_iterations,
                                                                kappa_fgr, theta, epsilon, s_dev,
                                                                update_step_M_norm, num_iterations, why))
          return s_dev;
      } else {
        why = "a user function is supplied (it observes every iteration's vectors)";
      }
    }
    // the generic loop below on device vectors: make the fall observable (mi_ctx_fusion_counters, MI355OPT_WARN_GENERIC)
    if (!g.empty()) (void)mi_ctx_note_generic(g.context(), MI_GENERIC_STPCG, why);
  }
#endif

  // ---- generic loop (any Vector with the operators of SURVEY.md Appendix A) --------------------
  Vector s = 0 * g;  // estimate of the step; also fixes the dimension          :211
  Vector r = g;      // model gradient at s:  r = g + H s                        :214
  Vector v;          // preconditioned (projected) residual
  Vector p;          // search direction
  Vector Hp;
  Multiplier lambda;  // multiplier estimate returned by the constraint preconditioner

  // v = P(r) (or r itself); with constraints also strip the A' lambda component from r  :229-253
  auto precondition = [&]() {
    if (!P) {
      v = r;
    } else {
      std::tie(v, lambda) = (*P)(r, args...);
      if (At) r -= (*At)(lambda, args...);
    }
  };
  precondition();
  p = -v;  // :256

  // running M-inner-products; never recomputed from the vectors                  :259-266
  Scalar s_M_p = 0;
  Scalar s_M_s = 0;
  Scalar p_M_p = inner_product(r, v, args...);

  const Scalar Delta_sq = Delta * Delta;                                         // :271
  const Scalar r0_norm = sqrt(inner_product(r, v, args...));                     // :275
  const Scalar target = r0_norm * std::min(kappa_fgr, std::pow(r0_norm, theta));  // :278-279

  Scalar alpha, beta, kappa;
  for (num_iterations = 0; num_iterations < max_iterations; ++num_iterations) {  // :285
    if (std::sqrt(inner_product(r, v, args...)) <= target) break;                // :290

    Hp = H(p, args...);                  // :294
    kappa = inner_product(p, Hp, args...);  // curvature along p                  :300

    // p (numerically) in the kernel of H: ride it to the boundary, downhill      :305-337
    if (sqrt(inner_product(Hp, Hp, args...)) / sqrt(inner_product(p, p, args...)) < epsilon) {
      if (inner_product(p, r, args...) < 0) {
        p *= -1;
        s_M_p *= -1;
      }
      const Scalar sigma = detail::boundary_steplength(s_M_p, p_M_p, s_M_s, Delta_sq);
      update_step_M_norm = Delta;
      s += sigma * p;
      return s;
    }

    alpha = inner_product(r, v, args...) / kappa;                                // :341
    const Scalar next_s_M_s = s_M_s + 2 * alpha * s_M_p + alpha * alpha * p_M_p;  // :344-345

    // negative curvature, or the full step leaves the region: stop on the boundary  :347-362
    if ((kappa <= 0) || (next_s_M_s > Delta_sq)) {
      const Scalar sigma = detail::boundary_steplength(s_M_p, p_M_p, s_M_s, Delta_sq);
      update_step_M_norm = Delta;
      s += sigma * p;
      return s;
    }

    if (user_function &&
        (*user_function)(num_iterations, g, H, P, At, s, r, v, p, alpha, args...))  // :365-369
      break;

    s = s + alpha * p;  // :374
    r += alpha * Hp;    // :377
    precondition();     // :381-405

    const Scalar rv = inner_product(r, v, args...);  // :408
    beta = rv / (alpha * kappa);                     // :412

    s_M_s = next_s_M_s;                       // :415
    s_M_p = beta * (s_M_p + alpha * p_M_p);   // :416
    p_M_p = rv + beta * beta * p_M_p;         // :417

    p = -v + beta * p;  // :420
  }

  update_step_M_norm = sqrt(s_M_s);  // :424
  return s;
}

// Observer invoked at the end of every LSQR pass (all quantities already updated); returning true
// stops the solver.                                                              (reference :450-456)
template <typename VectorX, typename VectorY, typename Scalar = double, typename... Args>
using LSQRUserFunction = std::function<bool(
    size_t k, const LinearOperator<VectorX, VectorY, Args...> &A,
    const LinearOperator<VectorY, VectorX, Args...> &At, const VectorY &b, const VectorX &xk,
    Scalar xk_norm, Scalar rbar_norm, Scalar Abar_rbar_norm, Scalar Abar_norm_est, Scalar Abar_cond_est,
    Args &...args)>;

// LSQR (Paige & Saunders) for   min_x |A x - b|^2 + lambda |x|^2   s.t. |x| <= Delta,
// i.e. the damped system Abar = [A; sqrt(lambda) I], bbar = [b; 0], with the stopping rules
//   S1 |rbar| <= btol |b| + Atol |Abar| |x|,   S2 |Abar' rbar| <= Atol |Abar| |rbar|,
//   S3 cond(Abar) estimate >= Abar_cond_limit, S4 |x| reaches Delta (the last step is shortened to
//   end exactly on the boundary), S5 the user function.                          reference :552-855
template <typename VectorX, typename VectorY, typename Scalar = double, typename... Args>
VectorX LSQR(const LinearOperator<VectorX, VectorY, Args...> &A,
             const LinearOperator<VectorY, VectorX, Args...> &At, const VectorY &b,
             const InnerProduct<VectorX, Scalar, Args...> &inner_product_x,
             const InnerProduct<VectorY, Scalar, Args...> &inner_product_y, Args &...args, Scalar &xnorm,
             size_t &num_iterations, size_t max_iterations = 1000, Scalar lambda = 0, Scalar btol = 1e-6,
             Scalar Atol = 1e-6, Scalar Abar_cond_limit = 1e8,
             Scalar Delta = sqrt(std::numeric_limits<Scalar>::max()),
             const std::optional<LSQRUserFunction<VectorX, VectorY, Scalar, Args...>> &user_function =
                 std::nullopt) {
  if (lambda < 0)
    throw std::invalid_argument("Tikhonov regularization parameter (lambda) must be a nonnegative "
                                "real value");
  if (btol < 0) throw std::invalid_argument("Stopping tolerance btol must be a nonnegative real number");
  if (Atol < 0) throw std::invalid_argument("Stopping tolerance Atol must be a nonnegative real number");
  if (Abar_cond_limit <= 0)
    throw std::invalid_argument("Stopping tolerance Abar_cond_limit must be a positive real number");
  if (Delta <= 0)
    throw std::invalid_argument("Trust-region radius (Delta) must be a positive real value");

#if OPTIMIZATION_HAVE_MI355
  {
    const char *why = "a user function is supplied (it observes every iteration's vectors)";
    if (!user_function) {
      VectorX x_dev;
      if (detail::lsqr_on_device<VectorX, VectorY, Scalar, Args...>(A, At, b, inner_product_x, inner_product_y,
                                                                    max_iterations, lambda, btol, Atol, Abar_cond_limit,
                                                                    Delta, x_dev, xnorm, num_iterations, why))
        return x_dev;
    }
    if constexpr (MI355::is_device_vector<VectorY>::value)
      if (!b.empty()) (void)mi_ctx_note_generic(b.context(), MI_GENERIC_LSQR, why);
  }
#endif

  VectorX x;
  xnorm = 0;
  num_iterations = 0;

  Scalar xx = 0;             // running |x|^2 of the QR-based estimate            :601
  Scalar Anorm = 0;          // estimate of |Abar|                                :604
  Scalar Acond = 0;          // estimate of cond(Abar)                            :607
  Scalar D_frob_sq = 0;      // |D|_F^2 of the direction matrix, eq. (4.9)        :612
  Scalar bnorm, rbar_norm;   //                                                   :615,620
  Scalar Arnorm = 0;         // |Abar' rbar|                                      :626
  const Scalar sqrt_lambda = sqrt(lambda);

  // Golub-Kahan start: beta u = b, alpha v = A' u                                :635-667
  Scalar alpha = 0, beta = 0;
  VectorX v, w;
  VectorY u;
  u = b;
  v = At(u, args...);
  x = 0 * v;
  alpha = sqrt(inner_product_x(v, v, args...));
  beta = sqrt(inner_product_y(u, u, args...));
  if (beta > 0) u /= beta;
  if (alpha > 0) {
    v /= alpha;
    alpha /= beta;  // v was built from b rather than from the unit vector u
    w = v;
  }

  Arnorm = alpha * beta;  // :670
  if (Arnorm == 0) return x;  // x = 0 already solves the least-squares problem    :671-674

  bnorm = beta;
  rbar_norm = beta;

  Scalar rhobar = alpha, phibar = beta;           // :686-687
  Scalar cs2 = -1, sn2 = 0, z = 0, res2 = 0;      // :689-692

  for (num_iterations = 0; num_iterations < max_iterations; ++num_iterations) {  // :696
    // next bidiagonalisation step: beta u = A v - alpha u ; alpha v = A' u - beta v   :707-724
    u = A(v, args...) - alpha * u;
    beta = sqrt(inner_product_y(u, u, args...));
    if (beta > 0) {
      u /= beta;
      Anorm = sqrt(Anorm * Anorm + alpha * alpha + beta * beta + lambda);
      v = At(u, args...) - beta * v;
      alpha = sqrt(inner_product_x(v, v, args...));
      if (alpha > 0) v /= alpha;
    }

    // rotation removing the damping term                                           :729-735
    const Scalar rhobar1 = sqrt(rhobar * rhobar + lambda);
    const Scalar cs1 = rhobar / rhobar1;
    const Scalar sn1 = sqrt_lambda / rhobar1;
    const Scalar psi = sn1 * phibar;
    phibar *= cs1;

    // rotation removing the sub-diagonal beta                                      :740-747
    const Scalar rho = sqrt(rhobar1 * rhobar1 + beta * beta);
    const Scalar cs = rhobar1 / rho;
    const Scalar sn = beta / rho;
    const Scalar theta = sn * alpha;
    rhobar = -cs * alpha;
    const Scalar phi = cs * phibar;
    phibar *= sn;
    const Scalar tau = sn * phi;

    // right rotation removing the super-diagonal theta -> estimate of |x|          :753-760
    const Scalar delta = sn2 * rho;
    const Scalar gammabar = -cs2 * rho;
    const Scalar rhs = phi - delta * z;
    const Scalar zbar = rhs / gammabar;
    const Scalar gamma = sqrt(gammabar * gammabar + theta * theta);
    cs2 = gammabar / gamma;
    sn2 = theta / gamma;
    z = rhs / gamma;

    const Scalar w_sq = inner_product_x(w, w, args...);  // :765
    const Scalar d_sq = w_sq / (rho * rho);              // :766

    xnorm = sqrt(xx + zbar * zbar);  // |x| after the full update                  :769-770
    xx += z * z;

    const Scalar t2 = -theta / rho;  // :772
    Scalar t1;
    if (xnorm <= Delta) {
      t1 = phi / rho;  // :779
    } else {
      // shorten the step so that x + t1 w lands on the trust-region boundary      :785-793
      const Scalar xtx = inner_product_x(x, x, args...);
      const Scalar wtx = inner_product_x(w, x, args...);
      t1 = (-wtx + sqrt(wtx * wtx + w_sq * (Delta * Delta - xtx))) / w_sq;
      xnorm = Delta;
    }

    x += t1 * w;     // :798
    w = v + t2 * w;  // :799

    D_frob_sq += d_sq;                  // :802
    Acond = Anorm * sqrt(D_frob_sq);    // eq. (5.10)                              :808
    const Scalar res1 = phibar * phibar;
    res2 += psi * psi;
    rbar_norm = sqrt(res1 + res2);      // :812-814
    Arnorm = alpha * fabs(tau);         // :818

    if (rbar_norm <= btol * bnorm + Atol * Anorm * xnorm) break;  // S1            :825
    if (Arnorm <= Atol * Anorm * rbar_norm) break;                // S2            :829
    if (Acond >= Abar_cond_limit) break;                          // S3            :833
    if (xnorm >= Delta) break;                                    // S4            :837
    if (user_function && (*user_function)(num_iterations, A, At, b, x, xnorm, rbar_norm, Arnorm, Anorm,
                                          Acond, args...))
      break;                                                      // S5            :845-851
  }
  return x;
}

// Same-type convenience: domain and codomain of A share the vector type and inner product
// (reference :859-875)
template <typename Vector, typename Scalar = double, typename... Args>
Vector LSQR(const LinearOperator<Vector, Vector, Args...> &A,
            const LinearOperator<Vector, Vector, Args...> &At, const Vector &b,
            const InnerProduct<Vector, Scalar, Args...> &inner_product, Args &...args, Scalar &xnorm,
            size_t &num_iterations, size_t max_iterations = 1000, Scalar lambda = 0, Scalar btol = 1e-6,
            Scalar Atol = 1e-6, Scalar Abar_cond_limit = 1e8,
            Scalar Delta = sqrt(std::numeric_limits<Scalar>::max()),
            const std::optional<LSQRUserFunction<Vector, Vector, Scalar, Args...>> &user_function =
                std::nullopt) {
  return LSQR<Vector, Vector, Scalar, Args...>(A, At, b, inner_product, inner_product, args..., xnorm,
                                              num_iterations, max_iterations, lambda, btol, Atol,
                                              Abar_cond_limit, Delta, user_function);
}

}  // namespace LinearAlgebra
}  // namespace Optimization
