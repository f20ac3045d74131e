/* mi355opt.h -- C ABI of libmi355opt.so: the MI355X (gfx950) device layer underneath the
 * Optimization::Riemannian::TNT / Optimization::LinearAlgebra::STPCG / LOBPCG function templates.
 *
 * The reference (david-m-rosen/Optimization) is a header-only C++ template library with no FFI:
 * its "interface" for this path is the implicit Vector concept (operators used at
 * LinearAlgebra/IterativeSolvers.h:211-420, Riemannian/TNT.h:375-557) plus user-supplied
 * std::function callables (Riemannian/Concepts.h:44-112).  This header is what a C++ maintainer
 * binds underneath those templates: every entry point below names the
 * reference expression(s) it realises on the device.  File:line citations are relative to
 * /root/reference/include/Optimization.
 *
 * Conventions
 *   - plain C, opaque handles, no C++/torch types; every function returns an mi_status (0 = ok);
 *     mi_last_error() gives the message of the calling thread's last failure.
 *   - all work is enqueued on the context's HIP stream and is asynchronous unless documented as
 *     synchronous ("sync").  One context = one GPU = one stream; not thread-safe per context (the
 *     reference is single-threaded).
 *   - vectors are flat fp64 arrays in HBM.  A Stiefel / Euclidean n x p matrix is row-major
 *     (p contiguous), SO(3)^N variables are N row-major 3x3 blocks, so(3)^N tangents are 3N doubles.
 *   - there is NO CPU fallback: without a GPU every compute entry point fails with
 *     MI_ERR_NO_DEVICE.
 */
#ifndef MI355OPT_H
#define MI355OPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_API __attribute__((visibility("default")))

typedef enum mi_status {
  MI_OK = 0,
  MI_ERR_INVALID_ARGUMENT = 1, /* the checks the reference answers with std::invalid_argument */
  MI_ERR_HIP = 2,
  MI_ERR_OOM = 3,
  MI_ERR_NO_DEVICE = 4,
  MI_ERR_COMM = 5,
  MI_ERR_INTERNAL = 6
} mi_status;

typedef struct mi_ctx mi_ctx;       /* device context: stream, memory pool, scalar file, communicator */
typedef struct mi_vec mi_vec;       /* fp64 device vector */
typedef struct mi_csr mi_csr;       /* sparse matrix (CSR on input, sliced-ELL-64 in HBM) */
typedef struct mi_bsr3 mi_bsr3;     /* 3x3-block sparse matrix (pose-graph Hessians) */
typedef struct mi_op mi_op;         /* symmetric linear operator on tangent vectors (the HVP) */
typedef struct mi_precon mi_precon; /* preconditioner v = M^-1 r */
typedef struct mi_stiefel_rq mi_stiefel_rq; /* Rayleigh-quotient problem on St(n,p) */
typedef struct mi_so3n mi_so3n;             /* chordal rotation-averaging problem on SO(3)^N */

/* ---------------------------------------------------------------------------------------------
 * (1) context / stream / pool
 * ------------------------------------------------------------------------------------------- */
MI_API const char *mi_version(void);
MI_API const char *mi_last_error(void);
MI_API const char *mi_status_string(int status);
MI_API int mi_device_count(int *count);
MI_API int mi_ctx_create(int device, mi_ctx **out);
MI_API int mi_ctx_destroy(mi_ctx *ctx);
MI_API int mi_ctx_sync(mi_ctx *ctx);                   /* sync: hipStreamSynchronize */
/* number of times the library waited for the device on this context (scalar read-backs, solver exits, explicit
 * syncs; uploads/downloads of whole vectors not counted): how often the host sits on the critical path of e.g. one
 * TNT outer iteration (tools/bench_tnt.py) */
MI_API int mi_ctx_sync_count(mi_ctx *ctx, size_t *count);
/* Which side of the fusion boundary the work of this context ran on (r06).  The template layer selects the fused
 * entry points by probing its std::function arguments with target<>() (Riemannian/TNT.h, LinearAlgebra/
 * IterativeSolvers.h of this repository): a client that wraps a tagged callable in a lambda of its own -- a logger, a
 * penalty term, the adapter style of the reference's own TNT.h:400-426 -- silently gets the generic loop (one host
 * synchronisation per inner product).  These counters make that visible: the fused_* fields count calls of the fused
 * entry points (mi_stpcg, mi_lsqr, mi_*_trial), generic_inner_products counts host-synchronising inner products
 * (mi_vec_dot, mi_vec_dot_batch), and the generic_* solver fields are reported by the template layer through
 * mi_ctx_note_generic when a solve on MI355::DeviceVector falls to its generic loop.  With MI355OPT_WARN_GENERIC=1 in the
 * environment the first such fall of each kind also prints one line on stderr saying which probe failed. */
typedef struct mi_fusion_counters {
  unsigned long long fused_stpcg_solves;
  unsigned long long generic_stpcg_solves;
  unsigned long long fused_lsqr_solves;
  unsigned long long generic_lsqr_solves;
  unsigned long long fused_trial_steps;      /* TNT / GradientDescent trial points evaluated by one fused chain */
  unsigned long long generic_trial_steps;    /* ... by the reference's statement sequence on DeviceVector */
  unsigned long long generic_inner_products; /* host-synchronising inner products */
} mi_fusion_counters;
enum { MI_GENERIC_STPCG = 0, MI_GENERIC_LSQR = 1, MI_GENERIC_TRIAL = 2 };
MI_API int mi_ctx_fusion_counters(mi_ctx *ctx, mi_fusion_counters *out);
MI_API int mi_ctx_fusion_counters_reset(mi_ctx *ctx);
/* template layer -> library: a solve / trial step of kind `what` (MI_GENERIC_*) on device vectors ran the generic loop;
 * `why` (may be NULL) names the probe that failed */
MI_API int mi_ctx_note_generic(mi_ctx *ctx, int what, const char *why);
MI_API int mi_ctx_stream(mi_ctx *ctx, void **stream);  /* the hipStream_t all work is enqueued on */
MI_API int mi_ctx_device_name(mi_ctx *ctx, char *buf, size_t buflen);
MI_API int mi_ctx_pool_bytes(mi_ctx *ctx, size_t *bytes_reserved);

/* HIP-event timing of whole regions and of individual kernels (bench.py's roofline leg).
 * Kernel ids are the MI_K_* values; when timing is enabled for an id every launch of that kernel is
 * bracketed by an event pair on the context stream. */
enum {
  MI_K_NONE = 0,
  MI_K_CG_INIT, MI_K_CG_DOT3, MI_K_CG_SCALAR_A, MI_K_CG_UPDATE, MI_K_CG_SCALAR_B, MI_K_CG_PUPDATE,
  MI_K_SPMM, MI_K_STIEFEL_SPMM_GRAM, MI_K_STIEFEL_GRAM_REDUCE, MI_K_STIEFEL_FINISH_DOTS,
  MI_K_STIEFEL_RETRACT, MI_K_BSR3_SPMV_DOTS, MI_K_BLAS1, MI_K_LOBPCG_GRAM, MI_K_LOBPCG_UPDATE,
  MI_K_LOBPCG_RESIDUAL, MI_K_STIEFEL_HESS_FUSED,
  MI_K_COMM_ALLREDUCE, /* a scalar exchange across ranks that is a launch of its own: one RCCL all-reduce (group) or
                          one exchange kernel of the peer-memory layer; folded exchanges have none */
  MI_K_COMM_HALO,      /* a halo exchange that is a launch of its own: ncclSend/ncclRecv group or push kernel */
  MI_K_COUNT
};
/* The library's switches (A/B forms of its kernels, verification hooks of the multi-rank path) are read from the
 * environment ONCE per context, in mi_ctx_create: MI355OPT_<NAME>=<integer> for NAME in FORCE_SLOT_PATH,
 * FORCE_LOCKSTEP, FORCE_UNIFORM_GRID, MAX_GRID, NO_DIRGRAM, DIRGRAM_DIRECT, IPC_TIMEOUT_MS, NO_FOLD, HALO_PUSH_LATE,
 * NO_PACKED, NO_WINDOW, NO_WIN_BOUNDS, NO_FAR_COMPUTED, WORDS16, NO_SPMM_STREAM, NO_SPMM_WIN, NO_UPDATE_MFMA,
 * NO_ZERO_COPY, NO_GRAM_HALF, NO_UPDATE_PAIR, SO3_NO_QUAT, SO3_NO_RQUAT, SO3_SORT_NBR, HALO_RPRIME, TWO_KERNEL_STEP, WIDE_QUAD,
 * NO_POLLED_SYNC, WARN_GENERIC, REANCHOR, NO_SPMM_SWEEP, SWEEP_ZSEGS, WIDE_WINDOW, EARLY_S (DESIGN.md / INTEGRATION.md say what each selects).  For the boolean switches a value that
 * is not an integer counts as 1 unless it is "no" / "false" / "off"; the integer-valued ones (MAX_GRID, IPC_TIMEOUT_MS,
 * WIDE_QUAD, WIDE_WINDOW, SO3_SORT_NBR, REANCHOR, SWEEP_ZSEGS) take integers only -- anything else is ignored with a warning and the default stays.  Any other
 * MI355OPT_* variable found in the environment (a removed or misspelt switch) gets one warning on stderr.  mi_ctx_set_option changes one on a live context (name with or without the MI355OPT_
 * prefix); format switches (NO_PACKED) act when a matrix is created, the communication ones before the layer they
 * concern is brought up.  Nothing else in the library reads the environment. */
MI_API int mi_ctx_set_option(mi_ctx *ctx, const char *name, long value);
MI_API int mi_ktime_enable(mi_ctx *ctx, int kernel_id, int on);
MI_API int mi_ktime_reset(mi_ctx *ctx);
/* sync: resolves recorded event pairs; returns launches and total milliseconds for kernel_id */
MI_API int mi_ktime_read(mi_ctx *ctx, int kernel_id, size_t *launches, double *total_ms);
MI_API const char *mi_kernel_name(int kernel_id);
/* Profiler ranges (SURVEY.md 8(b) group 9): roctx markers, visible to `rocprofv3 --marker-trace`.  The library
 * brackets its own solves (mi_stpcg, mi_lsqr) with ranges of those names; callers may nest their own (e.g. one per
 * TNT outer iteration, TNT.h:436-683).  No-ops (still MI_OK) when no ROCTx library (rocprofiler-sdk-roctx, else roctracer's libroctx64) is installed. */
MI_API int mi_range_push(const char *name);
MI_API int mi_range_pop(void);
MI_API int mi_timer_start(mi_ctx *ctx);             /* records an event on the stream */
MI_API int mi_timer_stop(mi_ctx *ctx, double *ms);  /* sync: records, waits, returns elapsed ms */

/* ---------------------------------------------------------------------------------------------
 * (2) vectors -- the implicit Vector concept
 * ------------------------------------------------------------------------------------------- */
MI_API int mi_vec_create(mi_ctx *ctx, size_t n, mi_vec **out); /* `Vector v;` + sizing; pooled */
MI_API int mi_vec_destroy(mi_vec *v);                           /* returns storage to the pool */
/* non-owning window [offset, offset+n) of `base` (e.g. a column block of a column-major panel:
 * `S.leftCols(ns)`, `S.middleCols(..)` LOBPCG.h:254-268); destroy with mi_vec_destroy.  Destruction order is free:
 * an owner destroyed while views of it are alive keeps its storage until the last view is destroyed (a handle to the
 * destroyed owner itself must of course not be used again). */
MI_API int mi_vec_view(const mi_vec *base, size_t offset, size_t n, mi_vec **out);
MI_API int mi_vec_len(const mi_vec *v, size_t *n);
MI_API int mi_vec_data(const mi_vec *v, void **device_ptr);
/* Announce a write made OUTSIDE the library through the pointer of mi_vec_data (torch interop, a user kernel): problem
 * objects cache speculative results keyed on a vector's contents (the trial point of mi_stiefel_rq_trial /
 * mi_so3n_trial), and every library entry point that writes a vector advances its generation stamp itself -- a view
 * shares the stamp of the vector that owns the storage -- but a raw-pointer write is invisible to it. */
MI_API int mi_vec_touch(mi_vec *v);
MI_API int mi_vec_upload(mi_vec *v, const double *host, size_t n);         /* sync */
MI_API int mi_vec_download(const mi_vec *v, double *host, size_t n);       /* sync */
MI_API int mi_vec_copy(mi_vec *dst, const mi_vec *src);   /* `r_k = g;` IterativeSolvers.h:214,231,383 */
MI_API int mi_vec_fill(mi_vec *v, double a);
MI_API int mi_vec_scale(mi_vec *v, double a);             /* `p_k *= -1;` :324 ; `u /= beta` :653 */
MI_API int mi_vec_div(mi_vec *v, double a);               /* `u /= beta` :653,712, `v /= alpha` :658,723: a true
                                                             division, rounded like the reference's */
MI_API int mi_vec_scale_to(mi_vec *z, double a, const mi_vec *x); /* z = a * x: `0 * g` :211 (0 * Inf = NaN, -0.0
                                                             kept), `sigma_k * p_k`, unary minus :256 */
MI_API int mi_vec_axpy(mi_vec *y, double a, const mi_vec *x); /* `s_k += sigma_k * p_k;` :336,360,377 */
/* z = a*x + b*y (z may alias x or y): `s_k = s_k + alpha_k*p_k` :374, `p_k = -v_k + beta_k*p_k` :420,
 * `0 * g` :211, unary minus :256, `X + V` Riemannian/Concepts.h:189 */
MI_API int mi_vec_axpby(mi_vec *z, double a, const mi_vec *x, double b, const mi_vec *y);
/* sync: `V1.dot(V2)` Riemannian/Concepts.h:178 -- deterministic two-stage reduction */
MI_API int mi_vec_dot(const mi_vec *x, const mi_vec *y, double *out);
/* k <= 4 inner products in one pass, results to host (sync) */
MI_API int mi_vec_dot_batch(mi_ctx *ctx, int k, const mi_vec *const *x, const mi_vec *const *y,
                            double *out);

/* ---------------------------------------------------------------------------------------------
 * (3) sparse operators (user HVP building blocks; the reference has none -- its HVP is a user
 *     callable invoked at IterativeSolvers.h:294 and TNT.h:512)
 * ------------------------------------------------------------------------------------------- */
MI_API int mi_csr_create(mi_ctx *ctx, size_t n, size_t nnz, const int32_t *rowptr,
                         const int32_t *col, const double *val, mi_csr **out); /* sync (upload) */
MI_API int mi_csr_destroy(mi_csr *A);
/* W (n x p row-major) = A V ; 1 <= p <= 8 */
MI_API int mi_csr_spmm(const mi_csr *A, int p, const mi_vec *V, mi_vec *W);

/* ---------------------------------------------------------------------------------------------
 * (4) linear operators and preconditioners handed to STPCG
 *     mi_op     <-> SymmetricLinearOperator<Vector> H   (LinearAlgebra/Concepts.h:20-21; called at
 *                   IterativeSolvers.h:294)
 *     mi_precon <-> STPCGPreconditioner<Vector,Multiplier> P with Multiplier = nullptr_t
 *                   (IterativeSolvers.h:83-85; called at :234,386; TNT.h:413-419)
 * ------------------------------------------------------------------------------------------- */
/* callback: must ENQUEUE out = Op(in) on the context stream and return MI_OK without synchronising */
typedef int (*mi_apply_fn)(void *user, const mi_vec *in, mi_vec *out);
MI_API int mi_op_create_callback(mi_ctx *ctx, size_t n, mi_apply_fn fn, void *user, mi_op **out);
/* rectangular operator (n_in -> n_out), e.g. a Jacobian and its adjoint for LSQR / TNLS */
MI_API int mi_op_create_callback_rect(mi_ctx *ctx, size_t n_in, size_t n_out, mi_apply_fn fn, void *user,
                                      mi_op **out);
/* A USER operator with the solver's reductions inside its own pass (r04).  The whole point of the reference is the
 * user-supplied Hessian (`H(p)` at IterativeSolvers.h:294, bound from the caller's QuadraticModel at TNT.h:400-426); a
 * plain callback operator pays a separate 2N-byte pass for the three curvature inner products of :300,305-306.  A
 * fused callback leaves them behind itself, like the built-in operators do: next to out = Op(in) the user's kernel
 * writes, for each of its workgroups b in [0, rows), the workgroup's partial sums of
 *     component 0: <in, out>     component 1: <out, out>     component 2: <in, in>
 * at args->partials[c * args->partial_stride + b] (fp64, EVERY row in [0, rows) written, zeros for idle workgroups) and
 * reports `rows`.  Contract: 1 <= rows <= args->max_rows; if args->required_rows > 0 (several ranks with RCCL: every
 * rank must leave the same number of rows) rows must equal it; the partial sums must be deterministic (fixed-order
 * reduction inside the workgroup, no atomics) because every workgroup of the consuming kernel re-reduces the rows in a
 * fixed order and all of them must obtain the same bits.  Everything is enqueued on args->stream (the context's
 * stream), nothing may synchronise.  With such an operator a fused STPCG iteration is three launches for ANY HIP
 * operator: the user's pass, k_cg_update, k_cg_pupdate.  `fn` is the same product without the sums (mi_op_apply, the
 * `dm` product of TNT.h:512).  examples/stpcg_user_stencil.hip is a complete client. */
typedef struct mi_fused_args {
  double *partials;      /* device; component-major partial rows */
  size_t partial_stride; /* doubles between components */
  int max_rows;          /* at most this many workgroups may leave a row */
  int required_rows;     /* > 0: exactly this many rows */
  void *stream;          /* hipStream_t of the context */
} mi_fused_args;
typedef int (*mi_apply_fused_fn)(void *user, const mi_vec *in, mi_vec *out, const mi_fused_args *args, int *rows);
MI_API int mi_op_create_callback_fused(mi_ctx *ctx, size_t n, mi_apply_fn fn, mi_apply_fused_fn fused, void *user,
                                       mi_op **out);
MI_API int mi_op_create_diag(mi_ctx *ctx, const mi_vec *d, mi_op **out);           /* Hp = d .* p */
MI_API int mi_op_create_csr(mi_ctx *ctx, const mi_csr *A, int p, mi_op **out);     /* Hp = A p    */
/* out = outer(inner(in)): e.g. TNLS's right-preconditioned LSQR operators A = dF o M, A' = M' o dF^* (reference
 * TNLS.h:60-63,432-447) as ONE device operator each, so that the inner solve stays in mi_lsqr.  Borrows both operands
 * (they must outlive the composition); owns one pooled intermediate vector. */
MI_API int mi_op_create_compose(mi_ctx *ctx, mi_op *outer, mi_op *inner, mi_op **out);
MI_API int mi_op_apply(mi_op *op, const mi_vec *in, mi_vec *out);
MI_API int mi_op_dims(const mi_op *op, size_t *n_in, size_t *n_out); /* either pointer may be null */
MI_API int mi_op_destroy(mi_op *op);

MI_API int mi_precon_create_callback(mi_ctx *ctx, size_t n, mi_apply_fn fn, void *user,
                                     mi_precon **out);
MI_API int mi_precon_create_diag(mi_ctx *ctx, const mi_vec *dinv, mi_precon **out); /* v = dinv .* r */
/* 3x3 block-Jacobi: blocks holds N row-major 3x3 INVERSE diagonal blocks (9N doubles) */
MI_API int mi_precon_create_block3(mi_ctx *ctx, const mi_vec *inv_blocks, mi_precon **out);
/* Constraint preconditioner of the PROJECTED solve (IterativeSolvers.h:83-85, applied at :229-253 and :381-405;
 * tests/IterativeSolvers_unit_test.cpp:316-496): P(r) = (v, lambda) with [M A'; A 0][v; lambda] = [r; 0] for a
 * DIAGONAL M (Minv = its inverse, n doubles) and a dense m x n constraint matrix A (row-major, m <= 512 <= n).
 * S = A M^-1 A' is formed, factored and inverted once on the device (sync); every application is two launches.
 * mi_precon_apply gives v; mi_precon_constraint_solve also lambda; mi_precon_constraint_At is A' (the `At` argument
 * of STPCG).  Handed to mi_stpcg with params.constraint_At != 0 the solve takes the reference's `At` branch
 * (r -= A' lambda after every preconditioner application, :251,403) inside the same pass.  A and Minv must outlive P. */
MI_API int mi_precon_create_constraint(mi_ctx *ctx, size_t n, size_t m, const mi_vec *A_rowmajor,
                                       const mi_vec *Minv_diag, mi_precon **out);
/* The same for SPARSE constraints and any number of them (r04): A (m x n) in CSR on the host (int32), M still diagonal.
 * A, A' and S = A M^-1 A' (formed once, on the host, as a sparse matrix) live on the device; an application is three
 * launches and no host round trip: b = A M^-1 r, then S lambda = b by a Jacobi-preconditioned CG iteration running
 * inside ONE workgroup (fixed-order reductions: deterministic) to a relative residual of inner_tol (0: 1e-14 -- the
 * projection must be accurate to rounding or the outer iterates drift out of the null space of A) or
 * inner_max_iterations (0: 10 m + 100), then v = M^-1 (r - A' lambda) and, with constraint_At, r -= A' lambda.
 * mi_precon_constraint_info reports what the inner iteration needed and left (sync).  Minv must outlive P.
 * Failure semantics (r06): a BREAKDOWN of the inner iteration (p'Sp <= 0 or not finite: dependent constraint rows, NaN /
 * Inf in the residual; status code 1) makes the mi_stpcg / mi_stpcg_collect that used the preconditioner return
 * MI_ERR_INTERNAL -- its result is not a projected step.  Stopping at inner_max_iterations short of inner_tol (status
 * code 2) is NOT an error: the caller who caps the iteration count gets the inexact projection it asked for, the solve
 * returns MI_OK with mi_stpcg_result::precon_status = 2, and mi_precon_constraint_info -- which always returns MI_OK on
 * a valid handle -- gives the code and the worst relative residual so far.  The status word is per solve (reset at the
 * start of each mi_stpcg). */
MI_API int mi_precon_create_constraint_csr(mi_ctx *ctx, size_t n, size_t m, const int32_t *rowptr, const int32_t *col,
                                           const double *val, const mi_vec *Minv, double inner_tol,
                                           size_t inner_max_iterations, mi_precon **out);
MI_API int mi_precon_constraint_info(mi_precon *P, size_t *last_inner_iterations, double *last_relative_residual,
                                     double *worst_relative_residual, int *status_code /* 0, 1 or 2; any pointer may
                                     be null */);
MI_API int mi_precon_constraint_solve(mi_precon *P, const mi_vec *r, mi_vec *v, mi_vec *lambda /*nullable*/);
MI_API int mi_precon_constraint_At(mi_precon *P, const mi_vec *lambda, mi_vec *out);
MI_API int mi_precon_apply(mi_precon *P, const mi_vec *r, mi_vec *v);
MI_API int mi_precon_destroy(mi_precon *P);

/* ---------------------------------------------------------------------------------------------
 * (5) fused Steihaug-Toint truncated preconditioned CG -- LinearAlgebra::STPCG
 *     (IterativeSolvers.h:166-426, unconstrained form: At == nullopt), metric = Euclidean/Frobenius
 *     dot of the coordinate arrays.  Device-resident scalar recurrences (:259-279,330-345,412-417);
 *     the host only polls a pinned status word and keeps `run_ahead` iterations enqueued.
 * ------------------------------------------------------------------------------------------- */
typedef struct mi_stpcg_params {
  double Delta;          /* :171 */
  size_t max_iterations; /* :172 (1000) */
  double kappa_fgr;      /* :172 (.1) */
  double theta;          /* :172 (.5) */
  double epsilon;        /* :179 (1e-8) */
  int run_ahead;         /* iterations the host may enqueue beyond the last one known complete (0 -> default 3) */
  int constraint_At;     /* nonzero: P is a constraint preconditioner (mi_precon_create_constraint) and the solve was given
                            `At` (IterativeSolvers.h:178): r -= A' lambda after every application of P (:251,403) */
  int defer_result;      /* nonzero: do NOT wait for the device at the exit.  s_out is valid in stream order (anything
                            enqueued afterwards sees it); `result` only receives hvp_calls, the other fields arrive
                            through mi_stpcg_collect.  Lets a caller put its next launch chain (e.g. the trust-region
                            trial step, Riemannian/TNT.h:493-512) behind the solve and read both back with ONE wait. */
} mi_stpcg_params;

enum {
  MI_STPCG_EXIT_RESIDUAL = 0, /* :290 */
  MI_STPCG_EXIT_MAXIT = 1,    /* :285 loop exhausted */
  MI_STPCG_EXIT_KERNEL = 2,   /* :305-337 */
  MI_STPCG_EXIT_BOUNDARY = 3  /* :347-361 */
};

typedef struct mi_stpcg_result {
  double update_step_M_norm; /* :334,359,424 */
  size_t num_iterations;     /* :285 (not incremented on boundary exits) */
  int exit_reason;
  size_t hvp_calls;          /* operator applications enqueued (incl. speculative ones past the exit) */
  double rv_final;           /* last <r,v> */
  int precon_status;         /* 0, or 2: a constraint preconditioner's inner iteration stopped at its iteration limit short
                                of its tolerance during this solve (an INEXACT projection; mi_precon_constraint_info has
                                the residual left).  A breakdown (code 1) makes mi_stpcg return MI_ERR_INTERNAL instead */
} mi_stpcg_result;

typedef struct mi_stpcg_trace { /* optional per-iteration scalars, host arrays of capacity cap */
  size_t cap, len;
  double *alpha, *beta, *kappa, *rv;
} mi_stpcg_trace;

MI_API void mi_stpcg_default_params(mi_stpcg_params *p);
/* sync at exit.  s_out must have g's length.  Returns MI_ERR_INVALID_ARGUMENT for the argument
 * ranges the reference rejects (:183-205). */
MI_API int mi_stpcg(mi_ctx *ctx, const mi_vec *g, mi_op *H, mi_precon *P /*nullable*/,
                    const mi_stpcg_params *params, mi_vec *s_out, mi_stpcg_result *result,
                    mi_stpcg_trace *trace /*nullable*/);
/* Result of the last mi_stpcg call made with defer_result on this context (update_step_M_norm :334,359,424,
 * num_iterations :285, exit reason, last <r,v>).  Waits only if the state copy enqueued at the solve's exit has not
 * completed yet -- after any later read-back on the same context it has.  (trace is not available in deferred mode.) */
MI_API int mi_stpcg_collect(mi_ctx *ctx, mi_stpcg_result *result);

/* ---------------------------------------------------------------------------------------------
 * (5b) fused LSQR  <->  LinearAlgebra::LSQR  IterativeSolvers.h:552-855 (Paige & Saunders, damped,
 *      with the trust-region radius of :779-793 and the stopping rules S1-S4 of :825-837; the user
 *      function S5 is only available through the generic template loop).  Device-resident like
 *      mi_stpcg: no scalar reaches the host inside the loop.  A: n_x -> n_y, At: n_y -> n_x.
 * ------------------------------------------------------------------------------------------- */
typedef struct mi_lsqr_params {
  size_t max_iterations; /* :558 (1000) */
  double lambda;         /* :558 (0)    */
  double btol, Atol;     /* :558-559 (1e-6) */
  double Acond_limit;    /* :559 (1e8)  */
  double Delta;          /* :559 (sqrt(DBL_MAX)) */
  int run_ahead;         /* as in mi_stpcg_params */
} mi_lsqr_params;
enum {
  MI_LSQR_EXIT_MAXIT = 0,   /* loop exhausted (:696) */
  MI_LSQR_EXIT_S1 = 1,      /* :825 */
  MI_LSQR_EXIT_S2 = 2,      /* :829 */
  MI_LSQR_EXIT_S3 = 3,      /* :833 */
  MI_LSQR_EXIT_S4 = 4,      /* :837 */
  MI_LSQR_EXIT_TRIVIAL = 5  /* A'b = 0: x = 0 returned before the loop (:671-674) */
};
typedef struct mi_lsqr_result {
  double xnorm;          /* :769-770,793 */
  size_t num_iterations; /* :696 (not advanced by the pass that breaks) */
  int exit_reason;
  double rbar_norm, Arnorm, Anorm, Acond; /* :814,818,711,808 */
  size_t operator_applications;           /* enqueued, incl. speculative ones past the exit */
} mi_lsqr_result;
MI_API void mi_lsqr_default_params(mi_lsqr_params *p);
MI_API int mi_lsqr(mi_ctx *ctx, mi_op *A, mi_op *At, const mi_vec *b, const mi_lsqr_params *params,
                   mi_vec *x_out, mi_lsqr_result *result); /* sync at exit */


/* ---------------------------------------------------------------------------------------------
 * (6) Stiefel manifold St(n,p), 1 <= p <= 8, embedded metric -- the callables a client of
 *     TNT supplies (Objective, QuadraticModel, RiemannianMetric, Retraction; sphere analogue in
 *     the reference: tests/TNT_unit_test.cpp:73-117).  p <= 4: rows in registers, 1024-thread workgroups, the
 *     LDS-window form of the one-pass Hessian for p <= 3; p = 5 ... 8: the wide-row one-pass Hessian (256-thread
 *     workgroups, the p x p matrices in LDS) in STPCG's recurrence form, the two-pass operator with a preconditioner
 * ------------------------------------------------------------------------------------------- */
MI_API int mi_stiefel_gram(mi_ctx *ctx, size_t n, int p, const mi_vec *X, const mi_vec *Z,
                           double *G_host /* p*p row-major, sync */);
MI_API int mi_stiefel_project(mi_ctx *ctx, size_t n, int p, const mi_vec *X, const mi_vec *Z,
                              mi_vec *out); /* Z - X sym(X'Z) */
MI_API int mi_stiefel_retract(mi_ctx *ctx, size_t n, int p, const mi_vec *X, const mi_vec *V,
                              mi_vec *Y);   /* polar: (X+V) ((X+V)'(X+V))^-1/2 */
/* f(X) = .5 tr(X' A X) */
MI_API int mi_stiefel_rq_create(mi_ctx *ctx, const mi_csr *A, size_t n, int p, mi_stiefel_rq **out);
MI_API int mi_stiefel_rq_destroy(mi_stiefel_rq *prob);
MI_API int mi_stiefel_rq_objective(mi_stiefel_rq *prob, const mi_vec *X, double *f); /* sync */
/* QuadraticModel: grad = AX - X sym(X'AX); caches S = sym(X'AX) on the device and binds the
 * Hessian operator  Hess[V] = P_X(A V - V S)  to X (X must outlive the operator's use). */
MI_API int mi_stiefel_rq_model(mi_stiefel_rq *prob, const mi_vec *X, mi_vec *grad, mi_op **hess);
/* One trial step at the point X the model is bound to -- the statements between the inner solve and the
 * accept/reject decision of Riemannian/TNT.h:493-512 (|h|, x_trial = retract(x, h), f(x_trial), <g,h>, <h, Hess h>)
 * plus, speculatively, the model at the trial point (:573-585: gradient and its norm) -- as one launch chain with ONE
 * read-back (sync).  out[5] = {f(X+), <h,h>, <g,h>, <h,Hess h>, |grad f(X+)|^2}.  A following
 * mi_stiefel_rq_model(prob, X_trial, ...) reuses A X+, S+ and the gradient instead of recomputing them.  Every number
 * has the bits the separate calls would produce. */
MI_API int mi_stiefel_rq_trial(mi_stiefel_rq *prob, const mi_vec *X, const mi_vec *h, const mi_vec *g,
                               mi_vec *X_trial, double out[5]);
/* One Armijo trial of a backtracking line search along -g (Riemannian/GradientDescent.h:266-286: h = -t g, x_trial =
 * retract(x, h), f(x_trial)) plus, speculatively, the gradient at the trial point and its squared norm (:325-327):
 * one launch chain, ONE read-back (sync).  out[2] = {f(X+), |grad f(X+)|^2}; h_out receives -t g.  A following
 * mi_stiefel_rq_model(prob, X_trial, ...) reuses what was computed. */
MI_API int mi_stiefel_rq_armijo_trial(mi_stiefel_rq *prob, const mi_vec *X, const mi_vec *g, double t, mi_vec *h_out,
                                      mi_vec *X_trial, double out[2]);
/* row-scaling (Jacobi) preconditioner projected to the tangent space: v = P_X(dinv_rows .* r) */
MI_API int mi_stiefel_rq_precon(mi_stiefel_rq *prob, const mi_vec *X, const mi_vec *dinv_rows,
                                mi_precon **out);

/* ---------------------------------------------------------------------------------------------
 * (7) SO(3)^N chordal rotation averaging  f(R) = .5 sum_e w_e |R_j - R_i Rt_e|_F^2
 * ------------------------------------------------------------------------------------------- */
MI_API int mi_so3n_create(mi_ctx *ctx, size_t N, size_t n_edges, const int32_t *ei,
                          const int32_t *ej, const double *Rt /*9 per edge*/, const double *w,
                          mi_so3n **out); /* sync (upload) */
MI_API int mi_so3n_destroy(mi_so3n *prob);
MI_API int mi_so3n_objective(mi_so3n *prob, const mi_vec *R, double *f); /* sync */
/* QuadraticModel: gradient in so(3)^N coordinates, Hessian operator as a symmetric 3x3-block
 * sparse matrix rebuilt at R, optional 3x3 block-Jacobi preconditioner from its diagonal blocks */
MI_API int mi_so3n_model(mi_so3n *prob, const mi_vec *R, mi_vec *grad, mi_op **hess,
                         mi_precon **block_jacobi /*nullable*/);
MI_API int mi_so3n_retract(mi_so3n *prob, const mi_vec *R, const mi_vec *xi, mi_vec *Y);
/* One trial step at the point R the model is bound to (Riemannian/TNT.h:493-512: |h|, R_trial = retract(R, h),
 * f(R_trial), <g,h>, <h, Hess h>) plus, speculatively, the model at the trial point (:573-585: gradient, Hessian blocks,
 * block-Jacobi blocks, |grad|^2 and -- with_precon -- |M^-1 grad|^2 with the preconditioner rebuilt there): one launch
 * chain, ONE read-back (sync).  out[6] = {f(R+), <h,h>, <g,h>, <h,Hess h>, |grad f(R+)|^2, |M+^-1 grad f(R+)|^2 or -1}.
 * A following mi_so3n_model(prob, R_trial, ...) swaps the speculative model in instead of rebuilding it.  Every number
 * has the bits the separate calls would produce. */
MI_API int mi_so3n_trial(mi_so3n *prob, const mi_vec *R, const mi_vec *h, const mi_vec *g, int with_precon,
                         mi_vec *R_trial, double out[6]);

/* ---------------------------------------------------------------------------------------------
 * (8) LOBPCG building blocks (LinearAlgebra/LOBPCG.h:131-337).  Panels are column-major m x k
 *     (ld = m), matching the reference's Eigen dense layout.
 * ------------------------------------------------------------------------------------------- */
/* G (ka x kb, column-major, host, sync) = S' * AS  -- LOBPCG.h:223,271-272 ; fp64 MFMA */
MI_API int mi_lobpcg_gram(mi_ctx *ctx, size_t m, int ka, int kb, const mi_vec *S, const mi_vec *AS,
                          double *G_host);
/* The same for T = [T1 (m x k1) | T2 (m x (k - k1))] held in two panels, G = S' T (k x k): the reference's S' A(S)
 * (LOBPCG.h:267,271) with A(S) = [A(X) | A([W P])] where A(X) is the panel of the previous iteration (:281) -- the same
 * columns, hence the same bits, one operator application fewer.  sync */
MI_API int mi_lobpcg_gram_split(mi_ctx *ctx, size_t m, int k, const mi_vec *S, int k1, const mi_vec *T1,
                                const mi_vec *T2, double *G_host);
/* Both Grams of a Rayleigh-Ritz step (LOBPCG.h:271-272: S'AS and S'BS) with ONE synchronisation: enqueued back to
 * back, read back together.  Each right-hand panel is one panel of k columns (T?2 == NULL, k1? ignored) or two
 * pieces [T?1 (k1? columns) | T?2].  Same kernels, same bits as the separate calls.  sync */
MI_API int mi_lobpcg_gram_pair(mi_ctx *ctx, size_t m, int k, const mi_vec *S, int k1a, const mi_vec *Ta1,
                               const mi_vec *Ta2, int k1b, const mi_vec *Tb1, const mi_vec *Tb2, double *Ga_host,
                               double *Gb_host);
/* The same for the case every LOBPCG iteration without a B operator is in: G_a = S' [Ta1 | Ta2] with T = A S for a
 * SYMMETRIC operator A (the reference's SymmetricLinearOperator, LOBPCG.h:131-134) and G_b = S' S, LOBPCG.h:271-272.
 * Both are symmetric; only the upper block triangle of each is formed (16 x 16 tiles; the lower block triangle is its
 * mirror image, which is also all the reference's eigensolver reads), from ONE pass over S and T.  Ta2 may be null. */
MI_API int mi_lobpcg_gram_pair_sym(mi_ctx *ctx, size_t m, int k, const mi_vec *S, int k1a, const mi_vec *Ta1,
                                   const mi_vec *Ta2, double *Ga_host, double *Gb_host);
/* A panel held as 1 to 3 column blocks that need not be adjacent: block i = the first cols[i] columns of block[i] (a
 * panel, or a mi_vec_view of some columns of one).  LOBPCG's search basis S = [X, W(:, nc:), P(:, nc:)]
 * (LOBPCG.h:254-264) is handed over like this -- as its three blocks lie -- instead of being copied together every
 * iteration once pairs are locked (nc > 0).  The *_blocks entry points below compute what their namesakes compute on
 * the concatenation of the blocks; shapes their one-pass kernels do not take are copied together internally. */
typedef struct mi_panel_blocks {
  int nblocks;
  const mi_vec *block[3];
  int cols[3];
} mi_panel_blocks;
MI_API int mi_lobpcg_gram_pair_sym_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, int k1a, const mi_vec *Ta1,
                                          const mi_vec *Ta2, double *Ga_host, double *Gb_host);
/* mi_lobpcg_gram_pair_sym_blocks with T = A(S) held as column blocks as well (a plain-callable operator applied to the
 * blocks of S one by one). */
MI_API int mi_lobpcg_gram_pair_sym_tblocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, const mi_panel_blocks *T,
                                           double *Ga_host, double *Gb_host);
/* The generalized problem (a B operator, LOBPCG.h:131-140,268,272; tests/LOBPCG_unit_test.cpp:178-225): the upper block
 * triangles of S' A(S) and S' B(S) -- both symmetric, SymmetricLinearOperator -- with S, A(S) and B(S) each held as column
 * blocks of the same total width ([X | W(:, nc:) | P(:, nc:)], [AX | A(W..) | A(P..)], [BX | B(W..) | B(P..)]): nothing is
 * copied together (r05).  One launch per Gram on the matrix pipe, one reduction, one read-back. */
MI_API int mi_lobpcg_gram_pair_gen_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, const mi_panel_blocks *AS,
                                          const mi_panel_blocks *BS, double *Ga_host, double *Gb_host);
MI_API int mi_lobpcg_update2_blocks(mi_ctx *ctx, size_t m, const mi_panel_blocks *S, int kc, const double *C_host,
                                    int ldc, mi_vec *Y, int k1, mi_vec *Y2);
MI_API int mi_csr_spmm_colmajor_blocks(const mi_csr *A, const mi_panel_blocks *X, mi_vec *Y);
/* Y (m x kc) = S (m x ks) * C (ks x kc column-major host) -- LOBPCG.h:226-227,278,288 */
MI_API int mi_lobpcg_update(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S,
                            const double *C_host, int ldc, mi_vec *Y);
/* The same with two destinations: output columns [0, k1) -> Y (m x k1), [k1, kc) -> Y2 (m x (kc - k1)).  Lets the
 * LOBPCG loop write X and P (LOBPCG.h:278,288) straight into the first and third block of the NEXT search basis
 * instead of copying them there (:254-259). */
MI_API int mi_lobpcg_update2(mi_ctx *ctx, size_t m, int ks, int kc, const mi_vec *S, const double *C_host, int ldc,
                             mi_vec *Y, int k1, mi_vec *Y2);
/* R = AX - BX diag(theta); rnorm[j] = |R_j|, xnorm[j] = |X_j| (host, sync) -- LOBPCG.h:230,285,293,302 */
MI_API int mi_lobpcg_residual(mi_ctx *ctx, size_t m, int nx, const mi_vec *AX, const mi_vec *BX,
                              const mi_vec *X, const double *theta_host, mi_vec *R, double *rnorm,
                              double *xnorm);
/* Rayleigh-Ritz on the host (ns <= 96): LOBPCG.h:53-62.  Theta ascending, C'AC = Theta, C'BC = I */
MI_API int mi_rayleigh_ritz(int n, const double *A, const double *B, double *Theta, double *C);
/* The k LOWEST Ritz pairs only: Theta[k] ascending, C n x k column-major with C'AC = diag(Theta), C'BC = I.  An LOBPCG
 * iteration reads nx of the ns <= 3 nx pairs (LOBPCG.h:278,288,293-318); same reduction, same QL recurrence (the Ritz
 * values have the bits of mi_rayleigh_ritz), the vectors from the recorded rotations applied to k columns instead of n
 * (they agree with the full solver's columns to rounding).  Host, like mi_rayleigh_ritz. */
MI_API int mi_rayleigh_ritz_lowest(int n, int k, const double *A, const double *B, double *Theta, double *C);
/* Y (n x k column-major) = A X : the sparse operator of LOBPCG clients (called at LOBPCG.h:213,218,267,281) */
MI_API int mi_csr_spmm_colmajor(const mi_csr *A, int k, const mi_vec *X, mi_vec *Y);
/* AX (n x nx) = A X (LOBPCG.h:281) together with R = AX - X diag(theta) (:285, B absent so BX = X) and the column norms
 * of R and X (:293,302; host, sync): when the matrix takes the LDS-window form the residual is finished in the
 * product's own pass (the row's X values are in the ring), otherwise this is mi_csr_spmm_colmajor followed by
 * mi_lobpcg_residual.  R has the same bits either way. */
MI_API int mi_csr_spmm_colmajor_residual(const mi_csr *A, int nx, const mi_vec *X, const double *theta_host,
                                         mi_vec *AX, mi_vec *R, double *rnorm, double *xnorm);
/* Y[r,c] = d[r] X[r,c] : diagonal operator / Jacobi preconditioner on a column-major panel
 * (the operators of tests/LOBPCG_unit_test.cpp:56-74) */
MI_API int mi_panel_rowscale(mi_ctx *ctx, size_t m, int k, const mi_vec *d, const mi_vec *X, mi_vec *Y);

/* ---------------------------------------------------------------------------------------------
 * (9) multi-GPU: one process per GPU, tangent vectors row-sharded, inner products completed by an
 *     in-stream RCCL all-reduce of the scalar slots.  uid = ncclUniqueId bytes
 *     obtained on rank 0 with mi_comm_unique_id and broadcast by the launcher.
 * ------------------------------------------------------------------------------------------- */
#define MI_COMM_UID_BYTES 128
MI_API int mi_comm_unique_id(unsigned char uid[MI_COMM_UID_BYTES]);
MI_API int mi_comm_init(mi_ctx *ctx, int world_size, int rank,
                        const unsigned char uid[MI_COMM_UID_BYTES]); /* sync */
MI_API int mi_comm_finalize(mi_ctx *ctx);
MI_API int mi_comm_info(mi_ctx *ctx, int *world_size, int *rank);
/* ranks of the RCCL communicator as RCCL itself reports them (ncclCommCount); 0: no RCCL communicator attached */
MI_API int mi_comm_rccl_count(mi_ctx *ctx, int *nranks);
/* Peer-memory layer for the tiny, latency-bound exchanges of the STPCG path (scalar all-reduces, halo
 * rows): every rank exports one fine-grained device arena (hipIpcGetMemHandle), the launcher gathers
 * the handles, every rank maps all of them (hipIpcOpenMemHandle: plain xGMI peer stores), runs the
 * collective self-test, and the layer is enabled only if EVERY rank passed (the launcher reduces the
 * verdicts); otherwise RCCL keeps doing those exchanges.  Also works without RCCL (mi_comm_init not
 * called): that is how several ranks on ONE GPU are tested (RCCL refuses duplicate devices). */
#define MI_COMM_IPC_HANDLE_BYTES 64
MI_API int mi_comm_ipc_export(mi_ctx *ctx, unsigned char handle[MI_COMM_IPC_HANDLE_BYTES]);
MI_API int mi_comm_ipc_attach(mi_ctx *ctx, int world_size, int rank,
                              const unsigned char *handles /* world_size x MI_COMM_IPC_HANDLE_BYTES */);
/* Collective, sync.  Known-value all-reduces and all-gathers through the exchange kernel, then (world_size > 1) the
 * FOLDED forms between the real peers: the scalar exchange inside a streaming kernel's prologue and all three forms of
 * the halo push (early fold, late fold, separate kernel) over 24 rounds with a consumer that checks every halo double.
 * The caller must combine the ranks' verdicts (a collective: no rank may go on before every rank is through) and pass
 * the result to mi_comm_ipc_enable on every rank. */
MI_API int mi_comm_ipc_selftest(mi_ctx *ctx, int *ok);
MI_API int mi_comm_ipc_enable(mi_ctx *ctx, int on);
MI_API int mi_comm_ipc_error(mi_ctx *ctx, int *err);        /* nonzero: a bounded wait timed out */
/* exchanges folded into their producer / consumer kernels (default on; env MI355OPT_NO_FOLD=1: off) switched at run
 * time -- every rank must make the same call between the same two solves.  Results are bit-identical either way. */
MI_API int mi_comm_ipc_fold(mi_ctx *ctx, int on);
/* Kernels the peer-memory layer launched on its own so far: out[0] scalar-exchange kernels, out[1] halo-push kernels,
 * out[2] halo pushes that rode in the kernel producing the vector instead (no launch), out[3] those of out[2] in the
 * EARLY form (the producer walks its vector starting at the neighbours' rows and signals after its first step).
 * With folding on (the default) a sharded fused STPCG iteration adds nothing to out[0] and out[1]: it is the three
 * kernels of the single-GPU step. */
MI_API int mi_comm_kernel_launches(mi_ctx *ctx, unsigned long long out[4]);
/* Verification hooks (used by tests/test_gpu_comm.py on a ONE-GPU box): pretend to be `rank` of
 * `world_size` WITHOUT a communicator, and fill a sharded matrix's halo rows by hand, so that the halo
 * addressing of the sparse kernels can be checked against the global product.  Not part of the drop-in
 * surface; with a communicator the halo is filled by the in-stream ncclSend/ncclRecv exchange. */
/* which form of the sparse kernels a matrix got: out = {window half-width in 64-row chunks (0: none), widest slice, far
 * stride D when every entry outside the window is at row +- D (the kernels then compute those columns, halo columns
 * of a row shard included) else 0, halo rows} */
MI_API int mi_debug_csr_window_info(const mi_csr *A, size_t out[4]);
/* host-only: the run plan of the LDS-window kernels (first tile of every run + the end) for `ntiles` tiles of 256 rows,
 * a workgroup budget, a CU count and the matrix's far stride in rows (0: none); needs no GPU */
MI_API int mi_debug_window_runs(int ntiles, int max_wgs, int num_cu, size_t far_stride, int *bounds_out, int cap,
                                int *nb_out);
MI_API int mi_debug_set_rank(mi_ctx *ctx, int world_size, int rank);
MI_API int mi_debug_csr_set_halo(mi_csr *A, int p, const double *halo_rows_host); /* (need_lo+need_hi) x p */
/* Measurement hook: average microseconds of `reps` back-to-back applications of the operator in the fused form
 * mi_stpcg uses (the user HVP of IterativeSolvers.h:294 together with the dots of :300,305-306), timed with
 * events on the context stream.  sync.  Not part of the drop-in surface (tools/time_op.py). */
MI_API int mi_debug_time_fused_apply(mi_op *op, const mi_vec *in, mi_vec *out, int reps, double *us_per_call);
/* halo description for a row-sharded sparse operator: rows [row_begin,row_end) of a global n x n
 * matrix are local; columns outside are fetched from the owning neighbour before each SpMM */
/* host-only planning step of mi_csr_create_sharded (no GPU needed; also used by the CPU gloo tests):
 * halo extents + local column indices ([0,n) local, then halo from rank-1, then halo from rank+1) */
MI_API int mi_csr_shard_plan(size_t n_global, int world_size, int rank, const size_t *row_starts,
                             size_t nnz_local, const int64_t *col_global, int32_t *col_local, size_t *need_lo,
                             size_t *need_hi);
MI_API int mi_csr_create_sharded(mi_ctx *ctx, size_t n_global, size_t row_begin, size_t row_end,
                                 size_t nnz_local, const int32_t *rowptr, const int64_t *col_global,
                                 const double *val, const size_t *row_starts /*world_size+1*/,
                                 mi_csr **out);

#ifdef __cplusplus
}
#endif
#endif /* MI355OPT_H */
