/* oracle.c -- CPU restatement of the reference algorithms (see oracle.h).  TEST INFRASTRUCTURE ONLY.
 *
 * Compile with -ffp-contract=off.  All file:line citations are relative to
 * /root/reference/include/Optimization.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* Optional all-cores leg (liboracle_omp.so, -fopenmp -DORC_OMP): the vector statements of STPCG run as OpenMP
 * loops.  The default liboracle.so is built WITHOUT it and stays the sequential restatement that is compared
 * bit for bit with the real reference. */
#if defined(ORC_OMP) && defined(_OPENMP)
#include <omp.h>
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static)")
void orc_set_threads(int n) { omp_set_num_threads(n); }
int orc_max_threads(void) { return omp_get_max_threads(); }
#else
#define ORC_PAR_FOR
void orc_set_threads(int n) { (void)n; }
int orc_max_threads(void) { return 1; }
#endif

/* ------------------------------------------------------------------------------------------- */
void orc_problem_free(orc_problem *p) {
  if (!p) return;
  if (p->destroy) p->destroy(p->user);
  free(p);
}

static double *dalloc(size_t n) { return (double *)malloc((n ? n : 1) * sizeof(double)); }

/* =============================================================================================
 * STPCG -- LinearAlgebra/IterativeSolvers.h:166-426 (At absent)
 * ========================================================================================== */
int orc_stpcg(size_t n, const double *g, orc_apply_fn H, void *H_user, orc_inner_fn ip,
              void *ip_user, orc_apply_fn P, void *P_user, double Delta, size_t max_iterations,
              double kappa_fgr, double theta, double epsilon, double *s, double *update_step_M_norm,
              size_t *num_iterations, int *exit_reason, orc_stpcg_trace *trace) {
  /* argument checks :183-205 */
  if (Delta <= 0) return -1;
  if ((kappa_fgr < 0) || (kappa_fgr >= 1)) return -1;
  if ((theta < 0) || (theta > 1)) return -1;
  if ((epsilon <= 0) || (epsilon >= 1)) return -1;

  double *r = dalloc(n), *v = dalloc(n), *p = dalloc(n), *Hp = dalloc(n);
  size_t i;
  int reason = ORC_STPCG_EXIT_MAXIT;
  if (trace) trace->len = 0;

  ORC_PAR_FOR
  for (i = 0; i < n; ++i) s[i] = 0 * g[i]; /* :211 */
  memcpy(r, g, n * sizeof(double));        /* :214 */
  if (!P)
    memcpy(v, r, n * sizeof(double)); /* :231 */
  else
    P(P_user, r, v); /* :234 */

  ORC_PAR_FOR
  for (i = 0; i < n; ++i) p[i] = -v[i]; /* :256 */

  double sk_M_pk = 0;                   /* :259 */
  double sk_M_2 = 0;                    /* :263 */
  double pk_M_2 = ip(ip_user, r, v);    /* :266 */
  double Delta_2 = Delta * Delta;       /* :271 */
  double r0_norm = sqrt(ip(ip_user, r, v)); /* :275 */
  double target_rk_norm = r0_norm * fmin(kappa_fgr, pow(r0_norm, theta)); /* :278-279 */
  /* std::min(a,b) returns a unless b < a; fmin differs only for NaN operands, where the
   * subsequent comparison at :290 is false either way. */

  double alpha_k, beta_k, kappa_k;
  size_t k;
  for (k = 0; k < max_iterations; ++k) {
    if (sqrt(ip(ip_user, r, v)) <= target_rk_norm) { /* :290 */
      reason = ORC_STPCG_EXIT_RESIDUAL;
      break;
    }
    H(H_user, p, Hp);              /* :294 */
    kappa_k = ip(ip_user, p, Hp);  /* :300 */

    if (sqrt(ip(ip_user, Hp, Hp)) / sqrt(ip(ip_user, p, p)) < epsilon) { /* :305-307 */
      if (ip(ip_user, p, r) < 0) {                                       /* :320 */
        for (i = 0; i < n; ++i) p[i] *= -1;                              /* :324 */
        sk_M_pk *= -1;                                                   /* :325 */
      }
      double sigma_k =
          (-sk_M_pk + sqrt(sk_M_pk * sk_M_pk + pk_M_2 * (Delta_2 - sk_M_2))) / pk_M_2; /* :330 */
      *update_step_M_norm = Delta;                                                      /* :334 */
      for (i = 0; i < n; ++i) s[i] += sigma_k * p[i];                                   /* :336 */
      *num_iterations = k;
      if (exit_reason) *exit_reason = ORC_STPCG_EXIT_KERNEL;
      free(r); free(v); free(p); free(Hp);
      return 0;
    }

    alpha_k = ip(ip_user, r, v) / kappa_k;                                          /* :341 */
    double skplus1_M_2 = sk_M_2 + 2 * alpha_k * sk_M_pk + alpha_k * alpha_k * pk_M_2; /* :344 */

    if ((kappa_k <= 0) || (skplus1_M_2 > Delta_2)) { /* :347 */
      double sigma_k =
          (-sk_M_pk + sqrt(sk_M_pk * sk_M_pk + pk_M_2 * (Delta_2 - sk_M_2))) / pk_M_2; /* :355 */
      *update_step_M_norm = Delta;                                                      /* :359 */
      for (i = 0; i < n; ++i) s[i] += sigma_k * p[i];                                   /* :360 */
      *num_iterations = k;
      if (exit_reason) *exit_reason = ORC_STPCG_EXIT_BOUNDARY;
      free(r); free(v); free(p); free(Hp);
      return 0;
    }

    ORC_PAR_FOR
    for (i = 0; i < n; ++i) s[i] = s[i] + alpha_k * p[i]; /* :374 */
    ORC_PAR_FOR
    for (i = 0; i < n; ++i) r[i] += alpha_k * Hp[i];      /* :377 */
    if (!P)
      memcpy(v, r, n * sizeof(double)); /* :383 */
    else
      P(P_user, r, v); /* :386 */

    double rk_vk = ip(ip_user, r, v);     /* :408 */
    beta_k = rk_vk / (alpha_k * kappa_k); /* :412 */

    sk_M_2 = skplus1_M_2;                           /* :415 */
    sk_M_pk = beta_k * (sk_M_pk + alpha_k * pk_M_2); /* :416 */
    pk_M_2 = rk_vk + beta_k * beta_k * pk_M_2;       /* :417 */

    ORC_PAR_FOR
    for (i = 0; i < n; ++i) p[i] = -v[i] + beta_k * p[i]; /* :420 */

    if (trace && trace->len < trace->cap) {
      size_t t = trace->len++;
      if (trace->alpha) trace->alpha[t] = alpha_k;
      if (trace->beta) trace->beta[t] = beta_k;
      if (trace->kappa) trace->kappa[t] = kappa_k;
      if (trace->rv) trace->rv[t] = rk_vk;
    }
  }
  *num_iterations = k;
  *update_step_M_norm = sqrt(sk_M_2); /* :424 */
  if (exit_reason) *exit_reason = reason;
  free(r); free(v); free(p); free(Hp);
  return 0;
}

/* =============================================================================================
 * TNT -- Riemannian/TNT.h:242-689
 * ========================================================================================== */
void orc_tnt_default_params(orc_tnt_params *p) {
  p->max_iterations = 100;             /* Base/Concepts.h:45 */
  p->max_computation_time = DBL_MAX;   /* :48 */
  p->gradient_tolerance = 1e-6;        /* Riemannian/Concepts.h:120 */
  p->relative_decrease_tolerance = 1e-6;
  p->stepsize_tolerance = 1e-6;
  p->Delta0 = 1;                       /* TNT.h:81 */
  p->eta1 = .05;
  p->eta2 = .9;
  p->alpha1 = .25;
  p->alpha2 = 2.5;
  p->max_TPCG_iterations = 1000;
  p->kappa_fgr = .1;
  p->theta = .5;
  p->preconditioned_gradient_tolerance = 1e-6;
  p->Delta_tolerance = 1e-6;
}

typedef struct tnt_ctx {
  orc_problem *prob;
  const double *x;
} tnt_ctx;

/* TNT.h:400-403 */
static void tnt_H(void *u, const double *v, double *hv) {
  tnt_ctx *c = (tnt_ctx *)u;
  c->prob->n_hess++;
  c->prob->hess(c->prob->user, c->x, v, hv);
}
/* TNT.h:406-410 */
static double tnt_ip(void *u, const double *a, const double *b) {
  tnt_ctx *c = (tnt_ctx *)u;
  c->prob->n_metric++;
  return c->prob->metric(c->prob->user, c->x, a, b);
}
/* TNT.h:413-419 */
static void tnt_P(void *u, const double *v, double *pv) {
  tnt_ctx *c = (tnt_ctx *)u;
  c->prob->n_precon++;
  c->prob->precon(c->prob->user, c->x, v, pv);
}

static double now_ms_quantised(const struct timespec *t0) {
  /* Util/Stopwatch.h:22-29: whole milliseconds / 1000 */
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  long long ns = (long long)(t1.tv_sec - t0->tv_sec) * 1000000000LL + (t1.tv_nsec - t0->tv_nsec);
  return (double)(ns / 1000000LL) / 1000.0;
}

int orc_tnt(orc_problem *prob, const double *x0, const orc_tnt_params *params, orc_tnt_result *res) {
  /* :260-318 */
  if (params->max_computation_time < 0) return -1;
  if (params->gradient_tolerance < 0) return -1;
  if (params->preconditioned_gradient_tolerance < 0) return -1;
  if (params->relative_decrease_tolerance < 0) return -1;
  if (params->stepsize_tolerance < 0) return -1;
  if (params->Delta_tolerance < 0) return -1;
  if (params->Delta0 <= 0) return -1;
  if (params->eta1 <= 0 || params->eta1 >= 1) return -1;
  if (params->eta1 > params->eta2 || params->eta2 >= 1) return -1;
  if (params->alpha1 <= 0 || params->alpha1 >= 1) return -1;
  if (params->alpha2 <= 1) return -1;
  if (params->kappa_fgr <= 0 || params->kappa_fgr >= 1) return -1;
  if (params->theta < 0) return -1;

  const size_t nv = prob->nvar, nt = prob->ntan;
  double sqrt_eps = sqrt(DBL_EPSILON); /* :323 */
  prob->n_f = prob->n_grad = prob->n_hess = prob->n_metric = prob->n_retract = prob->n_precon = 0;

  double *x = dalloc(nv), *x_prop = dalloc(nv), *grad = dalloc(nt), *h = dalloc(nt),
         *tmp = dalloc(nt);
  tnt_ctx ctx;
  ctx.prob = prob;
  ctx.x = x;

  res->status = ORC_TNT_ITERATION_LIMIT; /* :327 */
  res->n_trace = 0;
  res->outer_iterations = 0;
  res->accepted = 0;

  double fx, fx_prop, gradfx_norm, pgradfx_norm, Delta, h_norm = 0, h_M_norm = 0,
                                                        relative_decrease = 0;

  memcpy(x, x0, nv * sizeof(double)); /* :375 */
  prob->n_f++;
  fx = prob->f(prob->user, x); /* :377 */
  prob->n_grad++;
  prob->grad(prob->user, x, grad); /* :380 */
  prob->n_metric++;
  gradfx_norm = sqrt(prob->metric(prob->user, x, grad, grad)); /* :382 */
  if (prob->precon) {                                          /* :383-387 */
    prob->n_precon++;
    prob->precon(prob->user, x, grad, tmp);
    prob->n_metric++;
    pgradfx_norm = sqrt(prob->metric(prob->user, x, tmp, tmp));
  } else {
    pgradfx_norm = gradfx_norm; /* :391 */
  }

  Delta = params->Delta0; /* :429 */
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0); /* :440 */

  size_t iteration;
  for (iteration = 0; iteration < params->max_iterations; ++iteration) { /* :446 */
    double elapsed = now_ms_quantised(&t0);
    if (elapsed > params->max_computation_time) { /* :449 */
      res->status = ORC_TNT_ELAPSED_TIME;
      break;
    }
    /* :455-459 */
    res->objective_values[res->n_trace] = fx;
    res->gradient_norms[res->n_trace] = gradfx_norm;
    res->preconditioned_gradient_norms[res->n_trace] = pgradfx_norm;
    res->trust_region_radius[res->n_trace] = Delta;
    res->n_trace++;

    if (gradfx_norm < params->gradient_tolerance) { /* :474 */
      res->status = ORC_TNT_GRADIENT;
      break;
    }
    if (pgradfx_norm < params->preconditioned_gradient_tolerance) { /* :478 */
      res->status = ORC_TNT_PRECONDITIONED_GRADIENT;
      break;
    }

    /* :489-492 */
    size_t inner = 0;
    int rc = orc_stpcg(nt, grad, tnt_H, &ctx, tnt_ip, &ctx, prob->precon ? tnt_P : NULL, &ctx,
                       Delta, params->max_TPCG_iterations, params->kappa_fgr, params->theta, 1e-8,
                       h, &h_M_norm, &inner, NULL, NULL);
    if (rc) { /* theta > 1 throws from inside STPCG */
      free(x); free(x_prop); free(grad); free(h); free(tmp);
      return -1;
    }
    prob->n_metric++;
    h_norm = sqrt(prob->metric(prob->user, x, h, h)); /* :493 */

    prob->n_retract++;
    prob->retract(prob->user, x, h, x_prop); /* :505 */
    prob->n_f++;
    fx_prop = prob->f(prob->user, x_prop); /* :508 */

    /* :511-512: dm = -metric(x,grad,h) - .5*metric(x,h,Hess(x,h)).  C++ leaves the evaluation order
     * of the two metric() operands unspecified; they are pure, so only the counters could differ. */
    prob->n_metric++;
    double m1 = prob->metric(prob->user, x, grad, h);
    prob->n_hess++;
    prob->hess(prob->user, x, h, tmp);
    prob->n_metric++;
    double m2 = prob->metric(prob->user, x, h, tmp);
    double dm = -m1 - .5 * m2;

    double df = fx - fx_prop;                         /* :515 */
    relative_decrease = df / (sqrt_eps + fabs(fx));   /* :518 */
    double rho = df / dm;                             /* :521 */
    int step_accepted = (!isnan(rho) && rho > params->eta1); /* :532 */

    res->inner_iterations[res->outer_iterations] = inner; /* :538-541 */
    res->update_step_norms[res->outer_iterations] = h_norm;
    res->update_step_M_norms[res->outer_iterations] = h_M_norm;
    res->gain_ratios[res->outer_iterations] = rho;
    res->outer_iterations++;

    if (step_accepted) { /* :555 */
      res->accepted++;
      memcpy(x, x_prop, nv * sizeof(double)); /* :557 */
      fx = fx_prop;
      if (relative_decrease < params->relative_decrease_tolerance) { /* :561 */
        res->status = ORC_TNT_RELATIVE_DECREASE;
        break;
      }
      if (h_norm < params->stepsize_tolerance) { /* :567 */
        res->status = ORC_TNT_STEPSIZE;
        break;
      }
      prob->n_grad++;
      prob->grad(prob->user, x, grad); /* :573 */
      prob->n_metric++;
      gradfx_norm = sqrt(prob->metric(prob->user, x, grad, grad)); /* :575 */
      if (prob->precon) {                                          /* :576-580 */
        prob->n_precon++;
        prob->precon(prob->user, x, grad, tmp);
        prob->n_metric++;
        pgradfx_norm = sqrt(prob->metric(prob->user, x, tmp, tmp));
      } else {
        pgradfx_norm = gradfx_norm;
      }
    }

    if ((!isnan(rho)) && (rho >= params->eta2)) { /* :590 */
      double a = params->alpha2 * h_M_norm;
      Delta = (a < Delta) ? Delta : a; /* std::max(a, Delta) :593 -> returns a unless a < Delta */
    } else if (isnan(rho) || (rho < params->eta1)) { /* :594 */
      Delta = params->alpha1 * h_M_norm;             /* :597 */
      if (Delta < params->Delta_tolerance) {         /* :599 */
        res->status = ORC_TNT_TRUST_REGION;
        break;
      }
    }
  }

  /* :611-621 */
  memcpy(res->x, x, nv * sizeof(double));
  res->f = fx;
  res->gradfx_norm = gradfx_norm;
  res->preconditioned_gradfx_norm = pgradfx_norm;
  res->objective_values[res->n_trace] = fx;
  res->gradient_norms[res->n_trace] = gradfx_norm;
  res->preconditioned_gradient_norms[res->n_trace] = pgradfx_norm;
  res->trust_region_radius[res->n_trace] = Delta;
  res->n_trace++;

  free(x); free(x_prop); free(grad); free(h); free(tmp);
  return 0;
}

/* =============================================================================================
 * Rayleigh-Ritz -- LinearAlgebra/LOBPCG.h:53-62
 *
 * The reference calls Eigen::GeneralizedSelfAdjointEigenSolver (Eigen3 >= 3.3.3, not vendored,
 * absent here) on the diagonally equilibrated pair (DAD, DBD), D = diag(B)^-1/2.  Eigen's published
 * algorithm for ABx_lx: Cholesky DBD = L L', standard problem L^-1 (DAD) L^-T y = lambda y solved by
 * Householder tridiagonalisation + implicit QR, eigenvalues ascending, x = L^-T y (so x' (DBD) x = I).
 * Restated here with the same reduction and a cyclic Jacobi eigensolver for the standard problem
 * (eigenvector signs are not defined by either method -- compare subspaces, not columns).
 * ========================================================================================== */
static int cholesky_lower(size_t n, double *A) { /* column-major, in place, lower */
  size_t i, j, k;
  for (j = 0; j < n; ++j) {
    double d = A[j + j * n];
    for (k = 0; k < j; ++k) d -= A[j + k * n] * A[j + k * n];
    if (!(d > 0)) return -1;
    d = sqrt(d);
    A[j + j * n] = d;
    for (i = j + 1; i < n; ++i) {
      double s = A[i + j * n];
      for (k = 0; k < j; ++k) s -= A[i + k * n] * A[j + k * n];
      A[i + j * n] = s / d;
    }
    for (i = 0; i < j; ++i) A[i + j * n] = 0;
  }
  return 0;
}

/* cyclic Jacobi: M symmetric n x n column-major (destroyed), V eigenvectors (columns), w eigenvalues */
static void jacobi_eigh(size_t n, double *M, double *V, double *w) {
  size_t i, j, k, sweep;
  for (i = 0; i < n * n; ++i) V[i] = 0;
  for (i = 0; i < n; ++i) V[i + i * n] = 1;
  for (sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (j = 0; j < n; ++j)
      for (i = 0; i < n; ++i) {
        if (i != j) off += M[i + j * n] * M[i + j * n];
        else diag += M[i + j * n] * M[i + j * n];
      }
    if (off <= 1e-32 * (diag + off) || off == 0) break;
    for (i = 0; i + 1 < n; ++i)
      for (j = i + 1; j < n; ++j) {
        double apq = M[i + j * n];
        if (apq == 0) continue;
        double app = M[i + i * n], aqq = M[j + j * n];
        double tau = (aqq - app) / (2 * apq);
        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
        double c = 1 / sqrt(1 + t * t), s = t * c;
        for (k = 0; k < n; ++k) { /* columns i,j of M */
          double mki = M[k + i * n], mkj = M[k + j * n];
          M[k + i * n] = c * mki - s * mkj;
          M[k + j * n] = s * mki + c * mkj;
        }
        for (k = 0; k < n; ++k) { /* rows i,j of M */
          double mik = M[i + k * n], mjk = M[j + k * n];
          M[i + k * n] = c * mik - s * mjk;
          M[j + k * n] = s * mik + c * mjk;
        }
        for (k = 0; k < n; ++k) {
          double vki = V[k + i * n], vkj = V[k + j * n];
          V[k + i * n] = c * vki - s * vkj;
          V[k + j * n] = s * vki + c * vkj;
        }
      }
  }
  for (i = 0; i < n; ++i) w[i] = M[i + i * n];
  /* sort ascending (selection sort, swapping columns) */
  for (i = 0; i + 1 < n; ++i) {
    size_t m = i;
    for (j = i + 1; j < n; ++j)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      double t = w[i];
      w[i] = w[m];
      w[m] = t;
      for (k = 0; k < n; ++k) {
        t = V[k + i * n];
        V[k + i * n] = V[k + m * n];
        V[k + m * n] = t;
      }
    }
  }
}

int orc_rayleigh_ritz(size_t n, const double *A, const double *B, double *Theta, double *C) {
  size_t i, j, k;
  double *D = dalloc(n), *L = dalloc(n * n), *M = dalloc(n * n), *T = dalloc(n * n),
         *Y = dalloc(n * n);
  for (i = 0; i < n; ++i) D[i] = 1.0 / sqrt(B[i + i * n]); /* :56 */
  for (j = 0; j < n; ++j)
    for (i = 0; i < n; ++i) {
      L[i + j * n] = D[i] * B[i + j * n] * D[j]; /* DBD :59 */
      M[i + j * n] = D[i] * A[i + j * n] * D[j]; /* DAD :59 */
    }
  if (cholesky_lower(n, L)) {
    free(D); free(L); free(M); free(T); free(Y);
    return -1;
  }
  /* T = L^-1 M  (forward substitution on each column) */
  for (j = 0; j < n; ++j)
    for (i = 0; i < n; ++i) {
      double s = M[i + j * n];
      for (k = 0; k < i; ++k) s -= L[i + k * n] * T[k + j * n];
      T[i + j * n] = s / L[i + i * n];
    }
  /* M = T L^-T : solve X L' = T, row by row -> column j of X: X[:,j] = (T[:,j] - sum_{k<j} X[:,k] L[j,k]) / L[j,j] */
  for (j = 0; j < n; ++j)
    for (i = 0; i < n; ++i) {
      double s = T[i + j * n];
      for (k = 0; k < j; ++k) s -= M[i + k * n] * L[j + k * n];
      M[i + j * n] = s / L[j + j * n];
    }
  /* symmetrise against roundoff */
  for (j = 0; j < n; ++j)
    for (i = j + 1; i < n; ++i) {
      double a = .5 * (M[i + j * n] + M[j + i * n]);
      M[i + j * n] = a;
      M[j + i * n] = a;
    }
  jacobi_eigh(n, M, Y, Theta);
  /* x = L^-T y (back substitution), then C = D x  (:61) */
  for (j = 0; j < n; ++j) {
    for (i = n; i-- > 0;) {
      double s = Y[i + j * n];
      for (k = i + 1; k < n; ++k) s -= L[k + i * n] * T[k + j * n];
      T[i + j * n] = s / L[i + i * n];
    }
    for (i = 0; i < n; ++i) C[i + j * n] = D[i] * T[i + j * n];
  }
  free(D); free(L); free(M); free(T); free(Y);
  return 0;
}

/* =============================================================================================
 * LOBPCG -- LinearAlgebra/LOBPCG.h:131-337.  Column-major m x k panels.
 * ========================================================================================== */
static void gemm_tn(size_t m, size_t ka, size_t kb, const double *A, const double *B, double *C) {
  /* C (ka x kb) = A' (m x ka)' * B (m x kb) */
  size_t i, j, r;
  for (j = 0; j < kb; ++j)
    for (i = 0; i < ka; ++i) {
      double s = 0;
      const double *a = A + i * m, *b = B + j * m;
      for (r = 0; r < m; ++r) s += a[r] * b[r];
      C[i + j * ka] = s;
    }
}
static void gemm_nn(size_t m, size_t k, size_t n, const double *A, const double *B, size_t ldb,
                    double *C) {
  /* C (m x n) = A (m x k) * B (k x n, leading dim ldb) */
  size_t i, j, r;
  for (j = 0; j < n; ++j) {
    double *c = C + j * m;
    for (i = 0; i < m; ++i) c[i] = 0;
    for (r = 0; r < k; ++r) {
      double b = B[r + j * ldb];
      const double *a = A + r * m;
      for (i = 0; i < m; ++i) c[i] += a[i] * b;
    }
  }
}
static double fro_norm(size_t n, const double *a) {
  double s = 0;
  size_t i;
  for (i = 0; i < n; ++i) s += a[i] * a[i];
  return sqrt(s);
}

int orc_lobpcg(size_t m, size_t nx, size_t nev, orc_matop_fn A, void *A_user, orc_matop_fn B,
               void *B_user, orc_matop_fn T, void *T_user, const double *X0, const double *Omega,
               size_t max_iters, double tau, double *Theta_out, double *X_out, size_t *num_iters_out,
               size_t *nc_out, double *resid_out) {
  if (nev > nx) return -1; /* :150 */
  if (nx > m) return -1;   /* :155 */
  size_t i, j, ns = 0, nc = 0, num_iters = 0;
  const size_t nsmax = 3 * nx;
  double *X = dalloc(m * nx), *AX = dalloc(m * nx), *BX = dalloc(m * nx), *R = dalloc(m * nx),
         *W = dalloc(m * nx), *P = dalloc(m * nx), *S = dalloc(m * nsmax), *AS = dalloc(m * nsmax),
         *BS = dalloc(m * nsmax), *StAS = dalloc(nsmax * nsmax), *StBS = dalloc(nsmax * nsmax),
         *C = dalloc(nsmax * nsmax), *Theta = dalloc(nsmax), *tmp = dalloc(m * nx),
         *r = dalloc(nx), *Ccut = dalloc(nsmax * nx);
  memcpy(X, X0, m * nx * sizeof(double)); /* :162 */

  /* :213-214 */
  A(A_user, m, nx, Omega, tmp);
  double A2normest = fro_norm(m * nx, tmp) / fro_norm(m * nx, Omega);
  double B2normest = 1.0;
  if (B) {
    B(B_user, m, nx, Omega, tmp);
    B2normest = fro_norm(m * nx, tmp) / fro_norm(m * nx, Omega);
  }

  A(A_user, m, nx, X, AX); /* :218 */
  if (B) B(B_user, m, nx, X, BX); else memcpy(BX, X, m * nx * sizeof(double)); /* :219 */
  gemm_tn(m, nx, nx, X, AX, StAS);
  gemm_tn(m, nx, nx, X, BX, StBS);
  int rc = orc_rayleigh_ritz(nx, StAS, StBS, Theta, C); /* :222-223 */
  if (rc) goto done;
  gemm_nn(m, nx, nx, AX, C, nx, tmp); memcpy(AX, tmp, m * nx * sizeof(double)); /* :226 */
  gemm_nn(m, nx, nx, BX, C, nx, tmp); memcpy(BX, tmp, m * nx * sizeof(double)); /* :227 */
  for (j = 0; j < nx; ++j)
    for (i = 0; i < m; ++i) R[i + j * m] = AX[i + j * m] - BX[i + j * m] * Theta[j]; /* :230 */
  nc = 0; /* :233 */

  for (num_iters = 1; num_iters < max_iters; ++num_iters) { /* :237 */
    if (T) T(T_user, m, nx, R, W); else memcpy(W, R, m * nx * sizeof(double)); /* :247 */
    memcpy(S, X, m * nx * sizeof(double));                                     /* :254 */
    memcpy(S + m * nx, W + m * nc, m * (nx - nc) * sizeof(double));            /* :255 */
    if (num_iters > 1) {
      memcpy(S + m * (2 * nx - nc), P + m * nc, m * (nx - nc) * sizeof(double)); /* :259 */
      ns = 3 * nx - 2 * nc;
    } else {
      ns = 2 * nx - nc; /* :263 */
    }
    A(A_user, m, ns, S, AS); /* :267 */
    if (B) B(B_user, m, ns, S, BS); else memcpy(BS, S, m * ns * sizeof(double)); /* :268 */
    gemm_tn(m, ns, ns, S, AS, StAS); /* :271 */
    gemm_tn(m, ns, ns, S, BS, StBS); /* :272 */
    rc = orc_rayleigh_ritz(ns, StAS, StBS, Theta, C); /* :275 */
    if (rc) break;
    gemm_nn(m, ns, nx, S, C, ns, X); /* :278 */
    A(A_user, m, nx, X, AX);         /* :281 */
    if (B) B(B_user, m, nx, X, BX); else memcpy(BX, X, m * nx * sizeof(double)); /* :282 */
    for (j = 0; j < nx; ++j)
      for (i = 0; i < m; ++i) R[i + j * m] = AX[i + j * m] - BX[i + j * m] * Theta[j]; /* :285 */
    /* :288 P = S[:, nx:ns] * C[nx:ns, :nx] */
    for (j = 0; j < nx; ++j)
      for (i = 0; i < ns - nx; ++i) Ccut[i + j * (ns - nx)] = C[(nx + i) + j * ns];
    gemm_nn(m, ns - nx, nx, S + m * nx, Ccut, ns - nx, P);
    /* :293-307 */
    for (j = 0; j < nx; ++j) r[j] = fro_norm(m, R + j * m);
    for (nc = 0; nc < nev; ++nc) {
      double tol = tau * (A2normest + B2normest * fabs(Theta[nc])) * fro_norm(m, X + nc * m);
      if (!(r[nc] <= tol)) break; /* :316-318 */
    }
    if (nc == nev) break; /* :327 */
  }
done:
  memcpy(Theta_out, Theta, nev * sizeof(double)); /* :333 */
  memcpy(X_out, X, m * nev * sizeof(double));     /* :334 */
  if (resid_out) memcpy(resid_out, r, nx * sizeof(double));
  *num_iters_out = num_iters;
  *nc_out = nc;
  free(X); free(AX); free(BX); free(R); free(W); free(P); free(S); free(AS); free(BS);
  free(StAS); free(StBS); free(C); free(Theta); free(tmp); free(r); free(Ccut);
  return rc;
}
