// ref_driver.cpp -- builds oracle/_ref/libref.so: the REAL reference templates
// (Optimization::LinearAlgebra::STPCG, Optimization::Riemannian::TNT), compiled from the headers
// where they lie under /root/reference/include (never copied into this repo), instantiated on a
// plain host vector type and exposed through the same C signatures as the oracle (oracle.h).
//
// TEST INFRASTRUCTURE ONLY.  Built only where /root/reference exists (this container); the built
// .so is git-ignored but travels to the GPU box, where it is used as a checker and as the
// "reference" CPU baseline.  The reference's TNT.h / IterativeSolvers.h include no third-party
// header and are generic over the vector type, so instantiating them on RefVec is
// ordinary use of their template parameter, not a stand-in for a missing dependency.  LOBPCG.h
// hard-requires Eigen (absent) and is NOT built here.
#include "oracle.h"

#include <cstring>
#include <optional>
#include <stdexcept>
#include <vector>

#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/TNT.h"

namespace {

// Minimal host vector satisfying the implicit Vector concept.
struct RefVec {
  std::vector<double> d;
  RefVec() = default;
  explicit RefVec(size_t n) : d(n) {}
  RefVec(const double *p, size_t n) : d(p, p + n) {}
  size_t size() const { return d.size(); }
  double dot(const RefVec &o) const {
    double s = 0;
    for (size_t i = 0; i < d.size(); ++i) s += d[i] * o.d[i];
    return s;
  }
  RefVec &operator+=(const RefVec &o) {
    for (size_t i = 0; i < d.size(); ++i) d[i] += o.d[i];
    return *this;
  }
  RefVec &operator-=(const RefVec &o) {
    for (size_t i = 0; i < d.size(); ++i) d[i] -= o.d[i];
    return *this;
  }
  RefVec &operator*=(double a) {
    for (auto &x : d) x *= a;
    return *this;
  }
};
RefVec operator*(double a, const RefVec &v) {
  RefVec r(v.size());
  for (size_t i = 0; i < v.size(); ++i) r.d[i] = a * v.d[i];
  return r;
}
RefVec operator+(const RefVec &a, const RefVec &b) {
  RefVec r(a.size());
  for (size_t i = 0; i < a.size(); ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
RefVec operator-(const RefVec &a) {
  RefVec r(a.size());
  for (size_t i = 0; i < a.size(); ++i) r.d[i] = -a.d[i];
  return r;
}

using Scalar = double;
using Mult = std::nullptr_t;

} // namespace

extern "C" int ref_stpcg(size_t n, const double *g, orc_apply_fn H, void *H_user, orc_inner_fn ip,
                         void *ip_user, orc_apply_fn P, void *P_user, double Delta,
                         size_t max_iterations, double kappa_fgr, double theta, double epsilon,
                         double *s_out, double *update_step_M_norm, size_t *num_iterations,
                         int *exit_reason, orc_stpcg_trace *trace) {
  namespace LA = Optimization::LinearAlgebra;
  LA::SymmetricLinearOperator<RefVec> Hop = [&](const RefVec &v) {
    RefVec out(n);
    H(H_user, v.d.data(), out.d.data());
    return out;
  };
  LA::InnerProduct<RefVec, Scalar> inner = [&](const RefVec &a, const RefVec &b) {
    return ip(ip_user, a.d.data(), b.d.data());
  };
  std::optional<LA::STPCGPreconditioner<RefVec, Mult>> Pop;
  if (P)
    Pop = [&](const RefVec &v) {
      RefVec out(n);
      P(P_user, v.d.data(), out.d.data());
      return std::make_pair(out, Mult());
    };
  if (trace) trace->len = 0;
  std::optional<LA::STPCGUserFunction<RefVec, Mult, Scalar>> uf;
  if (trace)
    uf = [&](size_t, const RefVec &, const LA::SymmetricLinearOperator<RefVec> &,
             const std::optional<LA::STPCGPreconditioner<RefVec, Mult>> &,
             const std::optional<LA::LinearOperator<Mult, RefVec>> &, const RefVec &,
             const RefVec &, const RefVec &, const RefVec &, Scalar alpha) {
      if (trace->len < trace->cap && trace->alpha) trace->alpha[trace->len] = alpha;
      trace->len++;
      return false;
    };
  try {
    RefVec gv(g, n);
    Scalar mnorm = 0;
    size_t iters = 0;
    const std::optional<LA::LinearOperator<Mult, RefVec>> At_none; // unconstrained
    RefVec s = LA::STPCG<RefVec, Mult, Scalar>(gv, Hop, inner, mnorm, iters, Delta, max_iterations,
                                                kappa_fgr, theta, Pop, At_none, uf, epsilon);
    std::memcpy(s_out, s.d.data(), n * sizeof(double));
    *update_step_M_norm = mnorm;
    *num_iterations = iters;
    if (exit_reason) *exit_reason = -1; // not observable through the reference interface
    if (trace && trace->len > trace->cap) trace->len = trace->cap;
  } catch (const std::invalid_argument &) {
    return -1;
  }
  return 0;
}

extern "C" int ref_tnt(orc_problem *prob, const double *x0, const orc_tnt_params *params,
                       orc_tnt_result *res) {
  namespace R = Optimization::Riemannian;
  const size_t nv = prob->nvar, nt = prob->ntan;
  prob->n_f = prob->n_grad = prob->n_hess = prob->n_metric = prob->n_retract = prob->n_precon = 0;

  Optimization::Objective<RefVec, Scalar> f = [&](const RefVec &x) {
    prob->n_f++;
    return prob->f(prob->user, x.d.data());
  };
  R::QuadraticModel<RefVec, RefVec> QM = [&](const RefVec &x, RefVec &grad,
                                             R::LinearOperator<RefVec, RefVec> &Hess) {
    prob->n_grad++;
    grad = RefVec(nt);
    prob->grad(prob->user, x.d.data(), grad.d.data());
    Hess = [prob, nt](const RefVec &xx, const RefVec &v) {
      prob->n_hess++;
      RefVec out(nt);
      prob->hess(prob->user, xx.d.data(), v.d.data(), out.d.data());
      return out;
    };
  };
  R::RiemannianMetric<RefVec, RefVec, Scalar> metric = [&](const RefVec &x, const RefVec &a,
                                                          const RefVec &b) {
    prob->n_metric++;
    return prob->metric(prob->user, x.d.data(), a.d.data(), b.d.data());
  };
  R::Retraction<RefVec, RefVec> retract = [&](const RefVec &x, const RefVec &v) {
    prob->n_retract++;
    RefVec y(nv);
    prob->retract(prob->user, x.d.data(), v.d.data(), y.d.data());
    return y;
  };
  std::optional<R::LinearOperator<RefVec, RefVec>> precon;
  if (prob->precon)
    precon = [&](const RefVec &x, const RefVec &v) {
      prob->n_precon++;
      RefVec out(nt);
      prob->precon(prob->user, x.d.data(), v.d.data(), out.d.data());
      return out;
    };

  R::TNTParams<Scalar> tp;
  tp.max_iterations = params->max_iterations;
  tp.max_computation_time = params->max_computation_time;
  tp.gradient_tolerance = params->gradient_tolerance;
  tp.relative_decrease_tolerance = params->relative_decrease_tolerance;
  tp.stepsize_tolerance = params->stepsize_tolerance;
  tp.Delta0 = params->Delta0;
  tp.eta1 = params->eta1;
  tp.eta2 = params->eta2;
  tp.alpha1 = params->alpha1;
  tp.alpha2 = params->alpha2;
  tp.max_TPCG_iterations = params->max_TPCG_iterations;
  tp.kappa_fgr = params->kappa_fgr;
  tp.theta = params->theta;
  tp.preconditioned_gradient_tolerance = params->preconditioned_gradient_tolerance;
  tp.Delta_tolerance = params->Delta_tolerance;

  size_t accepted = 0;
  std::optional<R::TNTUserFunction<RefVec, RefVec, Scalar>> uf =
      [&](size_t, double, const RefVec &, Scalar, const RefVec &,
          const R::LinearOperator<RefVec, RefVec> &, Scalar, size_t, const RefVec &, Scalar, Scalar,
          bool acc) {
        if (acc) accepted++;
        return false;
      };

  try {
    RefVec xv(x0, nv);
    R::TNTResult<RefVec, Scalar> r =
        R::TNT<RefVec, RefVec, Scalar>(f, QM, metric, retract, xv, precon, tp, uf);
    std::memcpy(res->x, r.x.d.data(), nv * sizeof(double));
    res->f = r.f;
    res->gradfx_norm = r.gradfx_norm;
    res->preconditioned_gradfx_norm = r.preconditioned_grad_f_x_norm;
    res->status = static_cast<int>(r.status);
    res->outer_iterations = r.inner_iterations.size();
    res->n_trace = r.objective_values.size();
    for (size_t i = 0; i < res->n_trace; ++i) {
      res->objective_values[i] = r.objective_values[i];
      res->gradient_norms[i] = r.gradient_norms[i];
      res->preconditioned_gradient_norms[i] = r.preconditioned_gradient_norms[i];
      res->trust_region_radius[i] = r.trust_region_radius[i];
    }
    for (size_t i = 0; i < res->outer_iterations; ++i) {
      res->inner_iterations[i] = r.inner_iterations[i];
      res->update_step_norms[i] = r.update_step_norms[i];
      res->update_step_M_norms[i] = r.update_step_M_norms[i];
      res->gain_ratios[i] = r.gain_ratios[i];
    }
    res->accepted = accepted;
  } catch (const std::invalid_argument &) {
    return -1;
  }
  return 0;
}
