/* oracle.h -- CPU restatement ("oracle") of the reference's TNT / Steihaug-Toint CG / LOBPCG path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under optimization_amd/ or include/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker.
 *
 * Plain C on raw double arrays; every function cites the reference file:line (relative to
 * /root/reference/include/Optimization) whose arithmetic it restates.  Summation order of every
 * inner product is strictly sequential (i = 0..n-1), and the file is compiled with
 * -ffp-contract=off, so that it reproduces, bit for bit, the reference templates instantiated on
 * a sequential-dot host vector (oracle/ref_driver.cpp -> oracle/_ref/libref.so).
 *
 * Pinning (see oracle/README.md): SURVEY.md Appendix C golden values (JSON files under tests/golden), the
 * known-answer cases of the reference's own unit tests, and libref.so outputs on every problem.
 * LOBPCG iterates: parity unpinned (reference needs Eigen, absent here); eigenvalue answers and
 * the Rayleigh-Ritz identities of tests/LOBPCG_unit_test.cpp are pinned.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------
 * Problem description: the reference's user-supplied callables (Riemannian/Concepts.h:44-112)
 * flattened to C function pointers on double arrays.  `grad` plays the role of the
 * QuadraticModel call (TNT.h:380,573): it returns the Riemannian gradient at x and may cache
 * x-dependent state in `user` that `hess` (the LinearOperator returned by QM) then uses.
 * ------------------------------------------------------------------------------------------- */
typedef struct orc_problem {
  size_t nvar; /* doubles in a Variable (point on the manifold) */
  size_t ntan; /* doubles in a Tangent vector */
  void *user;
  double (*f)(void *user, const double *x);
  void (*grad)(void *user, const double *x, double *g);
  void (*hess)(void *user, const double *x, const double *v, double *hv);
  double (*metric)(void *user, const double *x, const double *a, const double *b);
  void (*retract)(void *user, const double *x, const double *v, double *y);
  void (*precon)(void *user, const double *x, const double *v, double *pv); /* may be NULL */
  void (*destroy)(void *user);
  /* call counters (filled by orc_tnt / ref_tnt wrappers) */
  size_t n_f, n_grad, n_hess, n_metric, n_retract, n_precon;
} orc_problem;

void orc_problem_free(orc_problem *p);

/* ---------------------------------------------------------------------------------------------
 * STPCG  (LinearAlgebra/IterativeSolvers.h:166-426), unconstrained form (At == nullopt).
 * ------------------------------------------------------------------------------------------- */
typedef void (*orc_apply_fn)(void *user, const double *in, double *out);
typedef double (*orc_inner_fn)(void *user, const double *a, const double *b);

typedef struct orc_stpcg_trace {
  size_t cap;    /* capacity of the arrays below (entries beyond cap are dropped) */
  size_t len;    /* completed iterations recorded */
  double *alpha; /* alpha_k   (IterativeSolvers.h:341) */
  double *beta;  /* beta_k    (:412) */
  double *kappa; /* <p,Hp>    (:300) */
  double *rv;    /* <r,v> after the update (:408) */
} orc_stpcg_trace;

enum {
  ORC_STPCG_EXIT_RESIDUAL = 0,  /* :290 break */
  ORC_STPCG_EXIT_MAXIT = 1,     /* loop exhausted */
  ORC_STPCG_EXIT_KERNEL = 2,    /* :305-337 p in ker H -> boundary */
  ORC_STPCG_EXIT_BOUNDARY = 3   /* :347-361 negative curvature or step leaves region */
};

/* returns 0, or -1 on an argument the reference rejects with std::invalid_argument (:183-205) */
int orc_stpcg(size_t n, const double *g, orc_apply_fn H, void *H_user, orc_inner_fn ip,
              void *ip_user, orc_apply_fn P /*nullable*/, void *P_user, double Delta,
              size_t max_iterations, double kappa_fgr, double theta, double epsilon, double *s_out,
              double *update_step_M_norm, size_t *num_iterations, int *exit_reason,
              orc_stpcg_trace *trace /*nullable*/);

/* ---------------------------------------------------------------------------------------------
 * TNT  (Riemannian/TNT.h:242-689)
 * ------------------------------------------------------------------------------------------- */
typedef struct orc_tnt_params {
  /* OptimizerParams (Base/Concepts.h:42-60) */
  size_t max_iterations;       /* 100 */
  double max_computation_time; /* DBL_MAX */
  /* SmoothOptimizerParams (Riemannian/Concepts.h:116-131) */
  double gradient_tolerance;          /* 1e-6 */
  double relative_decrease_tolerance; /* 1e-6 */
  double stepsize_tolerance;          /* 1e-6 */
  /* TNTParams (TNT.h:76-130) */
  double Delta0;                            /* 1 */
  double eta1, eta2, alpha1, alpha2;        /* .05 .9 .25 2.5 */
  size_t max_TPCG_iterations;               /* 1000 */
  double kappa_fgr, theta;                  /* .1 .5 */
  double preconditioned_gradient_tolerance; /* 1e-6 */
  double Delta_tolerance;                   /* 1e-6 */
} orc_tnt_params;

void orc_tnt_default_params(orc_tnt_params *p);

/* TNTStatus order (TNT.h:134-164) */
enum {
  ORC_TNT_GRADIENT = 0,
  ORC_TNT_PRECONDITIONED_GRADIENT,
  ORC_TNT_RELATIVE_DECREASE,
  ORC_TNT_STEPSIZE,
  ORC_TNT_TRUST_REGION,
  ORC_TNT_ITERATION_LIMIT,
  ORC_TNT_ELAPSED_TIME,
  ORC_TNT_USER_FUNCTION
};

typedef struct orc_tnt_result {
  /* caller allocates: x[nvar]; the trace arrays with capacity max_iterations + 2 */
  double *x;
  double f, gradfx_norm, preconditioned_gradfx_norm;
  int status;
  size_t outer_iterations;  /* number of started-and-completed outer iterations (= len of inner_iterations) */
  size_t n_trace;           /* entries in objective_values/gradient_norms/... (TNT.h:455-459,617-621) */
  double *objective_values, *gradient_norms, *preconditioned_gradient_norms, *trust_region_radius;
  size_t *inner_iterations;
  double *update_step_norms, *update_step_M_norms, *gain_ratios;
  size_t accepted;
} orc_tnt_result;

/* returns 0, or -1 for parameters the reference rejects (TNT.h:260-318) */
int orc_tnt(orc_problem *prob, const double *x0, const orc_tnt_params *params, orc_tnt_result *res);

/* ---------------------------------------------------------------------------------------------
 * Rayleigh-Ritz + LOBPCG  (LinearAlgebra/LOBPCG.h:53-62, 131-337).  Column-major dense storage.
 * ------------------------------------------------------------------------------------------- */
/* A, B: n x n symmetric (B SPD), column-major.  Theta[n] ascending, C n x n with C'AC = Theta,
 * C'BC = I.  Returns 0, or -1 if B is not positive definite after equilibration. */
int orc_rayleigh_ritz(size_t n, const double *A, const double *B, double *Theta, double *C);

/* Matrix operator: Y (m x k, column-major) = Op * X (m x k) */
typedef void (*orc_matop_fn)(void *user, size_t m, size_t k, const double *X, double *Y);

/* Omega: the m x nx Gaussian probe of LOBPCG.h:205-214, column-major, supplied by the caller
 * (the reference draws it from a default-seeded std::default_random_engine; the harness generates
 * it on the host and hands the same array to every implementation). */
int orc_lobpcg(size_t m, size_t nx, size_t nev, orc_matop_fn A, void *A_user, orc_matop_fn B,
               void *B_user, orc_matop_fn T, void *T_user, const double *X0, const double *Omega,
               size_t max_iters, double tau, double *Theta_out /*nev*/, double *X_out /*m x nev*/,
               size_t *num_iters, size_t *nc, double *resid_out /*nx, nullable*/);

/* ---------------------------------------------------------------------------------------------
 * Problems (oracle/problems.c)
 * ------------------------------------------------------------------------------------------- */
/* f(X) = |X - P|^2 on S^2, tests/TNT_unit_test.cpp:63-122; with_precon: diag(1,2,3) (:111-117) */
orc_problem *orc_problem_sphere(const double P[3], int with_precon);
/* chained Rosenbrock, Euclidean; precon_kind 0 none, 1 Jacobi-like
 * 1/(|2+1200 x_i^2|+200) */
orc_problem *orc_problem_rosenbrock(size_t n, int precon_kind);
/* f(x) = <g,x> + .5 <x, D x>, Euclidean, diagonal Hessian D, optional Jacobi precon Minv */
orc_problem *orc_problem_diag_quadratic(size_t n, const double *D, const double *g,
                                        const double *Minv /*nullable*/);
/* Rayleigh quotient f(X) = .5 tr(X' A X) on St(n,p); X row-major n x p; A CSR (int32 col) SPD;
 * embedded metric, projection P_X(Z) = Z - X sym(X'Z), polar retraction;
 * Hess[V] = P_X(A V - V sym(X' A X)).  dinv (nullable): diagonal (Jacobi) preconditioner values,
 * applied as P_X(dinv .* V). */
orc_problem *orc_problem_stiefel_rq(size_t n, size_t p, const int *rowptr, const int *col,
                                    const double *val, const double *dinv);
/* Chordal rotation averaging on SO(3)^N: f(R) = .5 sum_e w_e |R_j - R_i Rt_e|_F^2; variable = N
 * row-major 3x3 blocks (9N doubles); tangent = so(3)^N coordinates (3N doubles), xi_i <-> R_i hat(xi_i);
 * metric = Euclidean dot of coordinates; retraction R_i exp(hat(xi_i)); precon_kind 0 none, 1 = 3x3
 * block-Jacobi from the Hessian's diagonal blocks at x. */
orc_problem *orc_problem_so3n(size_t N, size_t n_edges, const int *ei, const int *ej,
                              const double *Rt /*9 per edge*/, const double *w /*per edge*/,
                              int precon_kind);

/* Small dense helpers exported for tests */
void orc_csr_spmm(size_t n, size_t p, const int *rowptr, const int *col, const double *val,
                  const double *V, double *W);
void orc_sym3_invsqrt(const double G[9], double out[9]); /* G SPD 3x3 row-major -> G^{-1/2} */
void orc_so3_exp(const double xi[3], double R[9]);

#ifdef __cplusplus
}
#endif
#endif
