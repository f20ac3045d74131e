/* problems.c -- the user-supplied callables (objective, quadratic model, metric, retraction,
 * preconditioner) of the test / benchmark problems, in plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference ships no manifold code beyond the S^2 lambdas in its tests/examples; these are
 * the harness's problem definitions, shared verbatim by the oracle (oracle.c) and by the real
 * reference templates (ref_driver.cpp wraps these same functions in std::function objects), so that
 * a difference between the two can only come from the algorithms.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Optional all-cores leg (liboracle_omp.so, -fopenmp -DORC_OMP; see oracle.c): the row loops of the Stiefel problem and
 * the inner products run as OpenMP loops (the sums are then re-associated: the last bits differ from the sequential
 * restatement, which is why the default build keeps them sequential). */
#if defined(ORC_OMP) && defined(_OPENMP)
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static)")
#define ORC_PAR_SUM(var) _Pragma("omp parallel for schedule(static) reduction(+ : s)")
#else
#define ORC_PAR_FOR
#define ORC_PAR_SUM(var)
#endif

static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }
static double dotn(size_t n, const double *a, const double *b) {
  double s = 0;
  size_t i;
  ORC_PAR_SUM(s)
  for (i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
static orc_problem *new_problem(size_t nvar, size_t ntan, void *user) {
  orc_problem *p = (orc_problem *)calloc(1, sizeof(orc_problem));
  p->nvar = nvar;
  p->ntan = ntan;
  p->user = user;
  return p;
}
static double euclid_metric_n(size_t n, const double *a, const double *b) { return dotn(n, a, b); }

/* =============================================================================================
 * Sphere S^2: tests/TNT_unit_test.cpp:63-122 (same expressions, evaluated left to right)
 * ========================================================================================== */
typedef struct { double P[3]; } sphere_t;

static void sphere_project(const double *X, const double *V, double *out) { /* :73-75 */
  double d = X[0] * V[0] + X[1] * V[1] + X[2] * V[2];
  int i;
  for (i = 0; i < 3; ++i) out[i] = V[i] - d * X[i];
}
static double sphere_f(void *u, const double *X) { /* :77 */
  sphere_t *s = (sphere_t *)u;
  double a = X[0] - s->P[0], b = X[1] - s->P[1], c = X[2] - s->P[2];
  return a * a + b * b + c * c;
}
static void sphere_grad(void *u, const double *X, double *g) { /* :79-85 */
  sphere_t *s = (sphere_t *)u;
  double nf[3];
  int i;
  for (i = 0; i < 3; ++i) nf[i] = 2 * (X[i] - s->P[i]);
  sphere_project(X, nf, g);
}
static void sphere_hess(void *u, const double *X, const double *Xdot, double *hv) { /* :94-97 */
  double eh[3], g[3], pr[3];
  int i;
  for (i = 0; i < 3; ++i) eh[i] = 2 * Xdot[i];
  sphere_project(X, eh, pr);
  sphere_grad(u, X, g);
  double d = X[0] * g[0] + X[1] * g[1] + X[2] * g[2];
  for (i = 0; i < 3; ++i) hv[i] = pr[i] - d * Xdot[i];
}
static double sphere_metric(void *u, const double *X, const double *a, const double *b) { /* :102 */
  (void)u; (void)X;
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static void sphere_retract(void *u, const double *X, const double *V, double *Y) { /* :106-108 */
  (void)u;
  double y[3] = {X[0] + V[0], X[1] + V[1], X[2] + V[2]};
  double nrm = sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
  Y[0] = y[0] / nrm; Y[1] = y[1] / nrm; Y[2] = y[2] / nrm;
}
static void sphere_precon(void *u, const double *X, const double *V, double *out) { /* :111-117 */
  (void)u; (void)X;
  out[0] = 1.0 * V[0]; out[1] = 2.0 * V[1]; out[2] = 3.0 * V[2];
}
orc_problem *orc_problem_sphere(const double P[3], int with_precon) {
  sphere_t *s = (sphere_t *)calloc(1, sizeof(sphere_t));
  memcpy(s->P, P, sizeof(s->P));
  orc_problem *p = new_problem(3, 3, s);
  p->f = sphere_f; p->grad = sphere_grad; p->hess = sphere_hess; p->metric = sphere_metric;
  p->retract = sphere_retract; p->precon = with_precon ? sphere_precon : NULL; p->destroy = free;
  return p;
}

/* =============================================================================================
 * Chained Rosenbrock, Euclidean
 * ========================================================================================== */
typedef struct { size_t n; int precon_kind; } rosen_t;

static double rosen_f(void *u, const double *x) {
  rosen_t *r = (rosen_t *)u;
  double s = 0;
  size_t i;
  for (i = 0; i + 1 < r->n; ++i) {
    double a = 1 - x[i], b = x[i + 1] - x[i] * x[i];
    s += a * a + 100 * b * b;
  }
  return s;
}
static void rosen_grad(void *u, const double *x, double *g) {
  rosen_t *r = (rosen_t *)u;
  size_t i;
  for (i = 0; i < r->n; ++i) g[i] = 0;
  for (i = 0; i + 1 < r->n; ++i) {
    double b = x[i + 1] - x[i] * x[i];
    g[i] += -2 * (1 - x[i]) - 400 * x[i] * b;
    g[i + 1] += 200 * b;
  }
}
static void rosen_hess(void *u, const double *x, const double *v, double *hv) {
  rosen_t *r = (rosen_t *)u;
  size_t i;
  for (i = 0; i < r->n; ++i) hv[i] = 0;
  for (i = 0; i + 1 < r->n; ++i) {
    double dii = 2 + 1200 * x[i] * x[i] - 400 * x[i + 1];
    double off = -400 * x[i];
    hv[i] += dii * v[i] + off * v[i + 1];
    hv[i + 1] += off * v[i] + 200 * v[i + 1];
  }
}
static double rosen_metric(void *u, const double *x, const double *a, const double *b) {
  (void)x;
  return euclid_metric_n(((rosen_t *)u)->n, a, b);
}
static void rosen_retract(void *u, const double *x, const double *v, double *y) {
  size_t i, n = ((rosen_t *)u)->n;
  for (i = 0; i < n; ++i) y[i] = x[i] + v[i]; /* Riemannian/Concepts.h:188-190 */
}
static void rosen_precon(void *u, const double *x, const double *v, double *pv) {
  size_t i, n = ((rosen_t *)u)->n;
  for (i = 0; i < n; ++i) pv[i] = v[i] / (fabs(2 + 1200 * x[i] * x[i]) + 200);
}
orc_problem *orc_problem_rosenbrock(size_t n, int precon_kind) {
  rosen_t *r = (rosen_t *)calloc(1, sizeof(rosen_t));
  r->n = n; r->precon_kind = precon_kind;
  orc_problem *p = new_problem(n, n, r);
  p->f = rosen_f; p->grad = rosen_grad; p->hess = rosen_hess; p->metric = rosen_metric;
  p->retract = rosen_retract; p->precon = precon_kind ? rosen_precon : NULL; p->destroy = free;
  return p;
}

/* =============================================================================================
 * Diagonal quadratic, Euclidean (the operators of tests/IterativeSolvers_unit_test.cpp:86-130)
 * ========================================================================================== */
typedef struct { size_t n; double *D, *g, *Minv; } diag_t;
static void diag_destroy(void *u) {
  diag_t *d = (diag_t *)u;
  free(d->D); free(d->g); free(d->Minv); free(d);
}
static double diag_f(void *u, const double *x) {
  diag_t *d = (diag_t *)u;
  double s = 0;
  size_t i;
  for (i = 0; i < d->n; ++i) s += d->g[i] * x[i] + .5 * x[i] * d->D[i] * x[i];
  return s;
}
static void diag_grad(void *u, const double *x, double *g) {
  diag_t *d = (diag_t *)u;
  size_t i;
  for (i = 0; i < d->n; ++i) g[i] = d->g[i] + d->D[i] * x[i];
}
static void diag_hess(void *u, const double *x, const double *v, double *hv) {
  diag_t *d = (diag_t *)u;
  size_t i;
  (void)x;
  for (i = 0; i < d->n; ++i) hv[i] = d->D[i] * v[i];
}
static double diag_metric(void *u, const double *x, const double *a, const double *b) {
  (void)x;
  return euclid_metric_n(((diag_t *)u)->n, a, b);
}
static void diag_retract(void *u, const double *x, const double *v, double *y) {
  size_t i, n = ((diag_t *)u)->n;
  for (i = 0; i < n; ++i) y[i] = x[i] + v[i];
}
static void diag_precon(void *u, const double *x, const double *v, double *pv) {
  diag_t *d = (diag_t *)u;
  size_t i;
  (void)x;
  for (i = 0; i < d->n; ++i) pv[i] = d->Minv[i] * v[i];
}
orc_problem *orc_problem_diag_quadratic(size_t n, const double *D, const double *g,
                                        const double *Minv) {
  diag_t *d = (diag_t *)calloc(1, sizeof(diag_t));
  d->n = n;
  d->D = dalloc(n); memcpy(d->D, D, n * sizeof(double));
  d->g = dalloc(n); memcpy(d->g, g, n * sizeof(double));
  if (Minv) { d->Minv = dalloc(n); memcpy(d->Minv, Minv, n * sizeof(double)); }
  orc_problem *p = new_problem(n, n, d);
  p->f = diag_f; p->grad = diag_grad; p->hess = diag_hess; p->metric = diag_metric;
  p->retract = diag_retract; p->precon = Minv ? diag_precon : NULL; p->destroy = diag_destroy;
  return p;
}

/* =============================================================================================
 * Stiefel Rayleigh quotient  f(X) = .5 tr(X' A X),  X in St(n,p) row-major n x p  (BASELINE cfg2)
 * ========================================================================================== */
void orc_csr_spmm(size_t n, size_t p, const int *rowptr, const int *col, const double *val,
                  const double *V, double *W) {
  size_t i, c;
  int k;
#if defined(ORC_OMP) && defined(_OPENMP)
#pragma omp parallel for schedule(static) private(c, k)
#endif
  for (i = 0; i < n; ++i) {
    double acc[16];
    for (c = 0; c < p; ++c) acc[c] = 0;
    for (k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const double a = val[k];
      const double *v = V + (size_t)col[k] * p;
      for (c = 0; c < p; ++c) acc[c] += a * v[c];
    }
    for (c = 0; c < p; ++c) W[i * p + c] = acc[c];
  }
}

/* p x p symmetric Jacobi eigen-decomposition, row-major, p <= 16 */
static void sym_jacobi(size_t p, const double *Gin, double *evec, double *eval) {
  double M[256];
  size_t i, j, k, sweep;
  memcpy(M, Gin, p * p * sizeof(double));
  for (i = 0; i < p * p; ++i) evec[i] = 0;
  for (i = 0; i < p; ++i) evec[i * p + i] = 1;
  for (sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (i = 0; i < p; ++i)
      for (j = i + 1; j < p; ++j) off += M[i * p + j] * M[i * p + j];
    if (off == 0) break;
    for (i = 0; i + 1 < p; ++i)
      for (j = i + 1; j < p; ++j) {
        double apq = M[i * p + j];
        if (apq == 0) continue;
        double tau = (M[j * p + j] - M[i * p + i]) / (2 * apq);
        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
        double c = 1 / sqrt(1 + t * t), s = t * c;
        for (k = 0; k < p; ++k) {
          double a = M[k * p + i], b = M[k * p + j];
          M[k * p + i] = c * a - s * b;
          M[k * p + j] = s * a + c * b;
        }
        for (k = 0; k < p; ++k) {
          double a = M[i * p + k], b = M[j * p + k];
          M[i * p + k] = c * a - s * b;
          M[j * p + k] = s * a + c * b;
        }
        for (k = 0; k < p; ++k) {
          double a = evec[k * p + i], b = evec[k * p + j];
          evec[k * p + i] = c * a - s * b;
          evec[k * p + j] = s * a + c * b;
        }
      }
  }
  for (i = 0; i < p; ++i) eval[i] = M[i * p + i];
}
static void sym_invsqrt(size_t p, const double *G, double *out) {
  double Q[256], w[16];
  size_t i, j, k;
  sym_jacobi(p, G, Q, w);
  for (i = 0; i < p; ++i)
    for (j = 0; j < p; ++j) {
      double s = 0;
      for (k = 0; k < p; ++k) s += Q[i * p + k] * (1.0 / sqrt(w[k])) * Q[j * p + k];
      out[i * p + j] = s;
    }
}
void orc_sym3_invsqrt(const double G[9], double out[9]) { sym_invsqrt(3, G, out); }

typedef struct {
  size_t n, p;
  int *rowptr, *col;
  double *val, *dinv;
  double *S;    /* p x p, sym(X' A X) cached by grad (the QM call) */
  double *W, *Z; /* n x p scratch */
} stiefel_t;

static void stiefel_destroy(void *u) {
  stiefel_t *s = (stiefel_t *)u;
  free(s->rowptr); free(s->col); free(s->val); free(s->dinv); free(s->S); free(s->W); free(s->Z);
  free(s);
}
/* G (p x p) = X' Z */
static void gram(size_t n, size_t p, const double *X, const double *Z, double *G) {
  size_t i, a, b;
  for (a = 0; a < p * p; ++a) G[a] = 0;
#if defined(ORC_OMP) && defined(_OPENMP)
  {
    double Gl[256];
    const size_t pp = p * p;
    for (a = 0; a < 256; ++a) Gl[a] = 0;
#pragma omp parallel for schedule(static) private(a, b) reduction(+ : Gl[:pp])
    for (i = 0; i < n; ++i)
      for (a = 0; a < p; ++a)
        for (b = 0; b < p; ++b) Gl[a * p + b] += X[i * p + a] * Z[i * p + b];
    for (a = 0; a < pp; ++a) G[a] = Gl[a];
    return;
  }
#endif
  for (i = 0; i < n; ++i)
    for (a = 0; a < p; ++a)
      for (b = 0; b < p; ++b) G[a * p + b] += X[i * p + a] * Z[i * p + b];
}
static void symmetrize(size_t p, double *G) {
  size_t a, b;
  for (a = 0; a < p; ++a)
    for (b = a + 1; b < p; ++b) {
      double m = .5 * (G[a * p + b] + G[b * p + a]);
      G[a * p + b] = m;
      G[b * p + a] = m;
    }
}
/* out = Z - X M */
static void sub_XM(size_t n, size_t p, const double *Z, const double *X, const double *M,
                   double *out) {
  size_t i, a, b;
#if defined(ORC_OMP) && defined(_OPENMP)
#pragma omp parallel for schedule(static) private(a, b)
#endif
  for (i = 0; i < n; ++i)
    for (b = 0; b < p; ++b) {
      double s = 0;
      for (a = 0; a < p; ++a) s += X[i * p + a] * M[a * p + b];
      out[i * p + b] = Z[i * p + b] - s;
    }
}
static double stiefel_f(void *u, const double *X) {
  stiefel_t *s = (stiefel_t *)u;
  orc_csr_spmm(s->n, s->p, s->rowptr, s->col, s->val, X, s->W);
  return .5 * dotn(s->n * s->p, X, s->W);
}
static void stiefel_grad(void *u, const double *X, double *g) {
  stiefel_t *s = (stiefel_t *)u;
  orc_csr_spmm(s->n, s->p, s->rowptr, s->col, s->val, X, s->W); /* Euclidean gradient A X */
  gram(s->n, s->p, X, s->W, s->S);
  symmetrize(s->p, s->S);               /* S = sym(X' A X), cached for hess */
  sub_XM(s->n, s->p, s->W, X, s->S, g); /* grad = AX - X S */
}
static void stiefel_hess(void *u, const double *X, const double *V, double *hv) {
  stiefel_t *s = (stiefel_t *)u;
  size_t i, a, b, n = s->n, p = s->p;
  double M[256];
  orc_csr_spmm(n, p, s->rowptr, s->col, s->val, V, s->W);
#if defined(ORC_OMP) && defined(_OPENMP)
#pragma omp parallel for schedule(static) private(a, b)
#endif
  for (i = 0; i < n; ++i) /* Z = A V - V S */
    for (b = 0; b < p; ++b) {
      double t = 0;
      for (a = 0; a < p; ++a) t += V[i * p + a] * s->S[a * p + b];
      s->Z[i * p + b] = s->W[i * p + b] - t;
    }
  gram(n, p, X, s->Z, M);
  symmetrize(p, M);
  sub_XM(n, p, s->Z, X, M, hv); /* P_X(Z) */
}
static double stiefel_metric(void *u, const double *X, const double *a, const double *b) {
  stiefel_t *s = (stiefel_t *)u;
  (void)X;
  return dotn(s->n * s->p, a, b);
}
static void stiefel_retract(void *u, const double *X, const double *V, double *Y) {
  stiefel_t *s = (stiefel_t *)u;
  size_t i, a, b, n = s->n, p = s->p;
  double G[256], Gi[256];
  for (i = 0; i < n * p; ++i) s->W[i] = X[i] + V[i];
  gram(n, p, s->W, s->W, G);
  symmetrize(p, G);
  sym_invsqrt(p, G, Gi);
  for (i = 0; i < n; ++i)
    for (b = 0; b < p; ++b) {
      double t = 0;
      for (a = 0; a < p; ++a) t += s->W[i * p + a] * Gi[a * p + b];
      Y[i * p + b] = t;
    }
}
static void stiefel_precon(void *u, const double *X, const double *V, double *pv) {
  stiefel_t *s = (stiefel_t *)u;
  size_t i, b, n = s->n, p = s->p;
  double M[256];
  for (i = 0; i < n; ++i)
    for (b = 0; b < p; ++b) s->Z[i * p + b] = s->dinv[i] * V[i * p + b];
  gram(n, p, X, s->Z, M);
  symmetrize(p, M);
  sub_XM(n, p, s->Z, X, M, pv);
}
orc_problem *orc_problem_stiefel_rq(size_t n, size_t p, const int *rowptr, const int *col,
                                    const double *val, const double *dinv) {
  if (p > 16) return NULL;
  stiefel_t *s = (stiefel_t *)calloc(1, sizeof(stiefel_t));
  size_t nnz = (size_t)rowptr[n];
  s->n = n; s->p = p;
  s->rowptr = (int *)malloc((n + 1) * sizeof(int)); memcpy(s->rowptr, rowptr, (n + 1) * sizeof(int));
  s->col = (int *)malloc((nnz ? nnz : 1) * sizeof(int)); memcpy(s->col, col, nnz * sizeof(int));
  s->val = dalloc(nnz); memcpy(s->val, val, nnz * sizeof(double));
  if (dinv) { s->dinv = dalloc(n); memcpy(s->dinv, dinv, n * sizeof(double)); }
  s->S = dalloc(p * p); s->W = dalloc(n * p); s->Z = dalloc(n * p);
  orc_problem *pr = new_problem(n * p, n * p, s);
  pr->f = stiefel_f; pr->grad = stiefel_grad; pr->hess = stiefel_hess; pr->metric = stiefel_metric;
  pr->retract = stiefel_retract; pr->precon = dinv ? stiefel_precon : NULL;
  pr->destroy = stiefel_destroy;
  return pr;
}

/* =============================================================================================
 * Chordal rotation averaging on SO(3)^N (BASELINE cfg3)
 *   f(R) = .5 sum_e w_e | R_j - R_i Rt_e |_F^2 ,  e = (i -> j)
 *   Euclidean gradient  EG = L R (connection Laplacian):  EG_j += w (R_j - R_i Rt),
 *                                                         EG_i += w (R_i - R_j Rt')
 *   tangent at R_i: R_i hat(xi_i); metric <xi,eta> = xi . eta (= half the Frobenius metric)
 *   grad_i = vee(Q_i - Q_i'),  Q_i = R_i' EG_i
 *   Hess[xi]_i = vee(T_i - T_i'),  T_i = R_i' (L V)_i - hat(xi_i) sym(Q_i),  V_i = R_i hat(xi_i)
 *   (embedded-submanifold formula for the rotation group with the bi-invariant metric)
 *   retraction: R_i exp(hat(xi_i)) (Rodrigues)
 * ========================================================================================== */
typedef struct {
  size_t N, E;
  int *ei, *ej;
  double *Rt, *w;
  int precon_kind;
  double *EG;   /* 9N scratch */
  double *C;    /* 9N: sym(Q_i) cached by grad */
  double *degw; /* N: sum of incident weights */
  double *V;    /* 9N scratch */
} so3n_t;

static void so3n_destroy(void *u) {
  so3n_t *s = (so3n_t *)u;
  free(s->ei); free(s->ej); free(s->Rt); free(s->w); free(s->EG); free(s->C); free(s->degw);
  free(s->V); free(s);
}
static void mat3_mul(const double *A, const double *B, double *C) { /* C = A B */
  int i, j, k;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      double s = 0;
      for (k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
static void mat3_mul_bt(const double *A, const double *B, double *C) { /* C = A B' */
  int i, j, k;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      double s = 0;
      for (k = 0; k < 3; ++k) s += A[i * 3 + k] * B[j * 3 + k];
      C[i * 3 + j] = s;
    }
}
static void mat3_mul_at(const double *A, const double *B, double *C) { /* C = A' B */
  int i, j, k;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      double s = 0;
      for (k = 0; k < 3; ++k) s += A[k * 3 + i] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
static void hat3(const double *x, double *K) {
  K[0] = 0; K[1] = -x[2]; K[2] = x[1];
  K[3] = x[2]; K[4] = 0; K[5] = -x[0];
  K[6] = -x[1]; K[7] = x[0]; K[8] = 0;
}
/* vee(T - T') = (T32-T23, T13-T31, T21-T12) */
static void vee_skew2(const double *T, double *out) {
  out[0] = T[7] - T[5];
  out[1] = T[2] - T[6];
  out[2] = T[3] - T[1];
}
void orc_so3_exp(const double xi[3], double R[9]) {
  double th2 = xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2];
  double th = sqrt(th2), a, b;
  double K[9], K2[9];
  int i;
  if (th < 1e-4) { /* series: sin(t)/t, (1-cos t)/t^2 */
    a = 1 - th2 / 6 + th2 * th2 / 120;
    b = .5 - th2 / 24 + th2 * th2 / 720;
  } else {
    a = sin(th) / th;
    b = (1 - cos(th)) / th2;
  }
  hat3(xi, K);
  mat3_mul(K, K, K2);
  for (i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
  R[0] += 1; R[4] += 1; R[8] += 1;
}
/* OUT = L * Y for a field Y of 3x3 blocks (row-major), the connection-Laplacian action */
static void so3n_laplacian(so3n_t *s, const double *Y, double *OUT) {
  size_t e, k;
  double T[9];
  for (k = 0; k < 9 * s->N; ++k) OUT[k] = 0;
  for (e = 0; e < s->E; ++e) {
    const size_t i = (size_t)s->ei[e], j = (size_t)s->ej[e];
    const double w = s->w[e], *Rt = s->Rt + 9 * e;
    mat3_mul(Y + 9 * i, Rt, T); /* Y_i Rt */
    for (k = 0; k < 9; ++k) OUT[9 * j + k] += w * (Y[9 * j + k] - T[k]);
    mat3_mul_bt(Y + 9 * j, Rt, T); /* Y_j Rt' */
    for (k = 0; k < 9; ++k) OUT[9 * i + k] += w * (Y[9 * i + k] - T[k]);
  }
}
static double so3n_f(void *u, const double *R) {
  so3n_t *s = (so3n_t *)u;
  size_t e, k;
  double T[9], acc = 0;
  for (e = 0; e < s->E; ++e) {
    const size_t i = (size_t)s->ei[e], j = (size_t)s->ej[e];
    double q = 0;
    mat3_mul(R + 9 * i, s->Rt + 9 * e, T);
    for (k = 0; k < 9; ++k) {
      double d = R[9 * j + k] - T[k];
      q += d * d;
    }
    acc += s->w[e] * q;
  }
  return .5 * acc;
}
static void so3n_grad(void *u, const double *R, double *g) {
  so3n_t *s = (so3n_t *)u;
  size_t i;
  double Q[9];
  so3n_laplacian(s, R, s->EG);
  for (i = 0; i < s->N; ++i) {
    double *C = s->C + 9 * i;
    int a, b;
    mat3_mul_at(R + 9 * i, s->EG + 9 * i, Q);
    vee_skew2(Q, g + 3 * i);
    for (a = 0; a < 3; ++a)
      for (b = 0; b < 3; ++b) C[a * 3 + b] = .5 * (Q[a * 3 + b] + Q[b * 3 + a]);
  }
}
static void so3n_hess(void *u, const double *R, const double *xi, double *h) {
  so3n_t *s = (so3n_t *)u;
  size_t i;
  int k;
  double K[9], T[9], KC[9];
  for (i = 0; i < s->N; ++i) {
    hat3(xi + 3 * i, K);
    mat3_mul(R + 9 * i, K, s->V + 9 * i);
  }
  so3n_laplacian(s, s->V, s->EG);
  for (i = 0; i < s->N; ++i) {
    hat3(xi + 3 * i, K);
    mat3_mul_at(R + 9 * i, s->EG + 9 * i, T);
    mat3_mul(K, s->C + 9 * i, KC);
    for (k = 0; k < 9; ++k) T[k] -= KC[k];
    vee_skew2(T, h + 3 * i);
  }
}
static double so3n_metric(void *u, const double *R, const double *a, const double *b) {
  (void)R;
  return dotn(3 * ((so3n_t *)u)->N, a, b);
}
static void so3n_retract(void *u, const double *R, const double *xi, double *Y) {
  so3n_t *s = (so3n_t *)u;
  size_t i;
  double Ex[9];
  for (i = 0; i < s->N; ++i) {
    orc_so3_exp(xi + 3 * i, Ex);
    mat3_mul(R + 9 * i, Ex, Y + 9 * i);
  }
}
/* 3x3 block-Jacobi: D_i = 2 degw_i I - (tr(C_i) I - C_i)  (diagonal block of the coordinate Hessian) */
static void so3n_precon(void *u, const double *R, const double *v, double *pv) {
  so3n_t *s = (so3n_t *)u;
  size_t i;
  (void)R;
  for (i = 0; i < s->N; ++i) {
    const double *C = s->C + 9 * i;
    const double tr = C[0] + C[4] + C[8], d = 2 * s->degw[i] - tr;
    double a = d + C[0], b = C[1], c = C[2], e = d + C[4], f = C[5], g = d + C[8];
    /* symmetric 3x3 inverse via cofactors: D = [a b c; b e f; c f g] */
    double c00 = e * g - f * f, c01 = c * f - b * g, c02 = b * f - c * e;
    double c11 = a * g - c * c, c12 = b * c - a * f, c22 = a * e - b * b;
    double det = a * c00 + b * c01 + c * c02;
    const double *x = v + 3 * i;
    pv[3 * i + 0] = (c00 * x[0] + c01 * x[1] + c02 * x[2]) / det;
    pv[3 * i + 1] = (c01 * x[0] + c11 * x[1] + c12 * x[2]) / det;
    pv[3 * i + 2] = (c02 * x[0] + c12 * x[1] + c22 * x[2]) / det;
  }
}
orc_problem *orc_problem_so3n(size_t N, size_t E, const int *ei, const int *ej, const double *Rt,
                              const double *w, int precon_kind) {
  so3n_t *s = (so3n_t *)calloc(1, sizeof(so3n_t));
  size_t e;
  s->N = N; s->E = E; s->precon_kind = precon_kind;
  s->ei = (int *)malloc((E ? E : 1) * sizeof(int)); memcpy(s->ei, ei, E * sizeof(int));
  s->ej = (int *)malloc((E ? E : 1) * sizeof(int)); memcpy(s->ej, ej, E * sizeof(int));
  s->Rt = dalloc(9 * E); memcpy(s->Rt, Rt, 9 * E * sizeof(double));
  s->w = dalloc(E); memcpy(s->w, w, E * sizeof(double));
  s->EG = dalloc(9 * N); s->C = dalloc(9 * N); s->V = dalloc(9 * N); s->degw = dalloc(N);
  for (e = 0; e < E; ++e) {
    s->degw[ei[e]] += w[e];
    s->degw[ej[e]] += w[e];
  }
  orc_problem *p = new_problem(9 * N, 3 * N, s);
  p->f = so3n_f; p->grad = so3n_grad; p->hess = so3n_hess; p->metric = so3n_metric;
  p->retract = so3n_retract; p->precon = precon_kind ? so3n_precon : NULL;
  p->destroy = so3n_destroy;
  return p;
}
