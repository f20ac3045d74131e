// ops.hip -- generic linear operators and preconditioners handed to mi_stpcg:
//   mi_op     <-> SymmetricLinearOperator<Vector> (LinearAlgebra/Concepts.h:20-21)
//   mi_precon <-> STPCGPreconditioner<Vector, nullptr_t> (IterativeSolvers.h:83-85)
#include "mi_internal.h"

using namespace mi;

namespace {

struct CallbackImpl {
  mi_apply_fn fn;
  void *user;
};

int op_callback_apply(mi_op *self, const mi_vec *in, mi_vec *out) {
  CallbackImpl *c = (CallbackImpl *)self->impl;
  int s = c->fn(c->user, in, out);
  if (s != MI_OK) set_error("operator callback returned status %d", s);
  return s;
}
void op_callback_destroy(mi_op *self) { delete (CallbackImpl *)self->impl; }

// Hp = d .* p fused with the three curvature dots (diagonal test operator of
// tests/IterativeSolvers_unit_test.cpp:99,107,112)
__global__ __launch_bounds__(kBlock) void k_diag_apply_dots(size_t n, const double *__restrict__ d,
                                                            const double *__restrict__ p,
                                                            double *__restrict__ Hp,
                                                            double *__restrict__ partials) {
  __shared__ double lds[3 * kWaves];
  double a[3] = {0, 0, 0};
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 dv = reinterpret_cast<const double2 *>(d)[i];
    const double2 pv = reinterpret_cast<const double2 *>(p)[i];
    const double2 hv = make_double2(dv.x * pv.x, dv.y * pv.y);
    reinterpret_cast<double2 *>(Hp)[i] = hv;
    a[0] += pv.x * hv.x; a[0] += pv.y * hv.y;
    a[1] += hv.x * hv.x; a[1] += hv.y * hv.y;
    a[2] += pv.x * pv.x; a[2] += pv.y * pv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double pv = p[n - 1], hv = d[n - 1] * pv;
    Hp[n - 1] = hv;
    a[0] += pv * hv; a[1] += hv * hv; a[2] += pv * pv;
  }
  if (!partials) return;
  block_partials_store<3>(a, lds, partials);
}

struct DiagImpl {
  const double *d;
};
int op_diag_apply_dots(mi_op *self, const mi_vec *in, mi_vec *out, int *nparts) {
  mi_ctx *ctx = self->ctx;
  const int grid = grid_for(ctx, self->n, 4);
  hipLaunchKernelGGL(k_diag_apply_dots, dim3(grid), dim3(kBlock), 0, ctx->stream, self->n,
                     ((DiagImpl *)self->impl)->d, (const double *)in->d, out->d, ctx->partials);
  *nparts = grid;
  return MI_OK;
}
int op_diag_apply(mi_op *self, const mi_vec *in, mi_vec *out) {
  mi_ctx *ctx = self->ctx;
  hipLaunchKernelGGL(k_diag_apply_dots, dim3(grid_for(ctx, self->n, 4)), dim3(kBlock), 0, ctx->stream,
                     self->n, ((DiagImpl *)self->impl)->d, (const double *)in->d, out->d,
                     (double *)nullptr);
  return MI_OK;
}
void op_diag_destroy(mi_op *self) { delete (DiagImpl *)self->impl; }

struct CsrOpImpl {
  const mi_csr *A;
  int p;
};
int op_csr_apply(mi_op *self, const mi_vec *in, mi_vec *out) {
  CsrOpImpl *c = (CsrOpImpl *)self->impl;
  return mi_csr_spmm(c->A, c->p, in, out);
}
int op_csr_apply_dots(mi_op *self, const mi_vec *in, mi_vec *out, int *nparts) {
  CsrOpImpl *c = (CsrOpImpl *)self->impl;
  bool unsupported = false;
  MI_TRY(csr_spmm_dots(c->A, c->p, in, out, nparts, &unsupported));
  if (!unsupported) return MI_OK;
  MI_TRY(mi_csr_spmm(c->A, c->p, in, out));  // fields of 4 GiB or more: product, then the generic dot kernel
  return launch_dot3_partials(self->ctx, in->n, in->d, out->d, nparts);
}
int op_csr_apply_sub_scaled(mi_op *self, const mi_vec *in, const double *scale, const int *mode, const int *gate,
                            mi_vec *inout, double *partials, int *nparts) {
  CsrOpImpl *c = (CsrOpImpl *)self->impl;
  return csr_spmv_sub_scaled(c->A, in, scale, mode, gate, inout, partials, nparts);
}
void op_csr_destroy(mi_op *self) { delete (CsrOpImpl *)self->impl; }

struct FusedCallbackImpl {
  mi_apply_fn fn;
  mi_apply_fused_fn fused;
  void *user;
};
int op_fused_callback_apply(mi_op *self, const mi_vec *in, mi_vec *out) {
  FusedCallbackImpl *c = (FusedCallbackImpl *)self->impl;
  int s = c->fn(c->user, in, out);
  if (s != MI_OK) set_error("operator callback returned status %d", s);
  return s;
}
// out = Op(in) with the three curvature partials left by the USER's kernel (mi355opt.h mi_apply_fused_fn)
int op_fused_callback_apply_dots(mi_op *self, const mi_vec *in, mi_vec *out, int *nparts) {
  FusedCallbackImpl *c = (FusedCallbackImpl *)self->impl;
  mi_ctx *ctx = self->ctx;
  const mi_fused_args a{ctx->partials, (size_t)kMaxRows, ctx->uniform_grid ? kMaxGrid : kMaxRows,
                        ctx->uniform_grid ? kMaxGrid : 0, (void *)ctx->stream};
  int rows = 0;
  int s = c->fused(c->user, in, out, &a, &rows);
  if (s != MI_OK) {
    set_error("fused operator callback returned status %d", s);
    return s;
  }
  MI_REQUIRE(rows >= 1 && rows <= a.max_rows, "fused operator callback reported %d partial rows (allowed: 1..%d)", rows,
             a.max_rows);
  MI_REQUIRE(a.required_rows == 0 || rows == a.required_rows,
             "fused operator callback reported %d partial rows, this context needs exactly %d", rows, a.required_rows);
  *nparts = rows;
  return MI_OK;
}
void op_fused_callback_destroy(mi_op *self) { delete (FusedCallbackImpl *)self->impl; }

int precon_callback_apply(mi_precon *self, const mi_vec *r, mi_vec *v) {
  CallbackImpl *c = (CallbackImpl *)self->impl;
  int s = c->fn(c->user, r, v);
  if (s != MI_OK) set_error("preconditioner callback returned status %d", s);
  return s;
}
void precon_callback_destroy(mi_precon *self) { delete (CallbackImpl *)self->impl; }

__global__ __launch_bounds__(kBlock) void k_block3_apply(size_t nb, const double *__restrict__ M,
                                                         const double *__restrict__ r,
                                                         double *__restrict__ v) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t b = (size_t)blockIdx.x * kBlock + threadIdx.x; b < nb; b += stride) {
    const double r0 = r[3 * b], r1 = r[3 * b + 1], r2 = r[3 * b + 2];
    const double *m = M + 9 * b;
    v[3 * b] = m[0] * r0 + m[1] * r1 + m[2] * r2;
    v[3 * b + 1] = m[3] * r0 + m[4] * r1 + m[5] * r2;
    v[3 * b + 2] = m[6] * r0 + m[7] * r1 + m[8] * r2;
  }
}

int precon_diag_apply(mi_precon *self, const mi_vec *r, mi_vec *v) {
  hipLaunchKernelGGL(k_diag_apply_dots, dim3(grid_for(self->ctx, self->n, 4)), dim3(kBlock), 0, self->ctx->stream,
                     self->n, self->data, (const double *)r->d, v->d, (double *)nullptr);
  return MI_OK;
}
int precon_block3_apply(mi_precon *self, const mi_vec *r, mi_vec *v) {
  const size_t nb = self->n / 3;
  hipLaunchKernelGGL(k_block3_apply, dim3(grid_for(self->ctx, nb, 2)), dim3(kBlock), 0, self->ctx->stream, nb,
                     self->data, (const double *)r->d, v->d);
  return MI_OK;
}

}  // namespace

extern "C" {

int mi_op_create_callback(mi_ctx *ctx, size_t n, mi_apply_fn fn, void *user, mi_op **out) {
  MI_REQUIRE(ctx && fn && out, "null argument");
  mi_op *op = new mi_op();
  op->ctx = ctx;
  op->n = n;
  op->apply = op_callback_apply;
  op->destroy = op_callback_destroy;
  op->impl = new CallbackImpl{fn, user};
  *out = op;
  return MI_OK;
}

int mi_op_create_callback_fused(mi_ctx *ctx, size_t n, mi_apply_fn fn, mi_apply_fused_fn fused, void *user,
                                mi_op **out) {
  MI_REQUIRE(ctx && fn && fused && out, "null argument");
  mi_op *op = new mi_op();
  op->ctx = ctx;
  op->n = n;
  op->apply = op_fused_callback_apply;
  op->apply_dots = op_fused_callback_apply_dots;
  op->destroy = op_fused_callback_destroy;
  op->impl = new FusedCallbackImpl{fn, fused, user};
  *out = op;
  return MI_OK;
}

int mi_op_create_callback_rect(mi_ctx *ctx, size_t n_in, size_t n_out, mi_apply_fn fn, void *user, mi_op **out) {
  MI_TRY(mi_op_create_callback(ctx, n_in, fn, user, out));
  (*out)->n_out = n_out;
  return MI_OK;
}

// out = outer(inner(in)) through one pooled intermediate vector, which lives as long as the operator (stream order makes
// its reuse across applications safe).  Neither operand is owned.
namespace {
struct ComposeImpl {
  mi_op *outer, *inner;
  mi_vec *mid;
};
int op_compose_apply(mi_op *self, const mi_vec *in, mi_vec *out) {
  ComposeImpl *c = static_cast<ComposeImpl *>(self->impl);
  MI_TRY(mi_op_apply(c->inner, in, c->mid));
  return mi_op_apply(c->outer, c->mid, out);
}
void op_compose_destroy(mi_op *self) {
  ComposeImpl *c = static_cast<ComposeImpl *>(self->impl);
  mi_vec_destroy(c->mid);
  delete c;
}
}  // namespace

int mi_op_create_compose(mi_ctx *ctx, mi_op *outer, mi_op *inner, mi_op **out) {
  MI_REQUIRE(ctx && outer && inner && out, "null argument");
  MI_REQUIRE(outer->ctx == ctx && inner->ctx == ctx, "operators belong to another context");
  const size_t mid_n = inner->n_out ? inner->n_out : inner->n;
  MI_REQUIRE(outer->n == mid_n, "compose: inner produces %zu values, outer takes %zu", mid_n, outer->n);
  mi_vec *mid = nullptr;
  MI_TRY(mi_vec_create(ctx, mid_n, &mid));
  mi_op *op = new mi_op();
  op->ctx = ctx;
  op->n = inner->n;
  const size_t n_out = outer->n_out ? outer->n_out : outer->n;
  op->n_out = n_out == op->n ? 0 : n_out;
  op->apply = op_compose_apply;
  op->destroy = op_compose_destroy;
  op->impl = new ComposeImpl{outer, inner, mid};
  *out = op;
  return MI_OK;
}

int mi_op_create_diag(mi_ctx *ctx, const mi_vec *d, mi_op **out) {
  MI_REQUIRE(ctx && d && out, "null argument");
  MI_REQUIRE(d->ctx == ctx, "vector belongs to another context");
  mi_op *op = new mi_op();
  op->ctx = ctx;
  op->n = d->n;
  op->apply = op_diag_apply;
  op->apply_dots = op_diag_apply_dots;
  op->destroy = op_diag_destroy;
  op->impl = new DiagImpl{d->d};
  *out = op;
  return MI_OK;
}

int mi_op_create_csr(mi_ctx *ctx, const mi_csr *A, int p, mi_op **out) {
  MI_REQUIRE(ctx && A && out, "null argument");
  MI_REQUIRE(p >= 1 && p <= kMaxP, "p must be in [1,%d]", kMaxP);
  const size_t n = A->n;
  mi_op *op = new mi_op();
  op->ctx = ctx;
  op->n = n * (size_t)p;
  op->apply = op_csr_apply;
  op->apply_dots = op_csr_apply_dots;
  if (p == 1) op->apply_sub_scaled = op_csr_apply_sub_scaled;
  op->destroy = op_csr_destroy;
  op->impl = new CsrOpImpl{A, p};
  *out = op;
  return MI_OK;
}

int mi_op_apply(mi_op *op, const mi_vec *in, mi_vec *out) {
  MI_REQUIRE(op && in && out, "null argument");
  MI_REQUIRE(in->n == op->n && out->n == (op->n_out ? op->n_out : op->n), "operator dimension mismatch");
  MI_REQUIRE(in->d != out->d, "operator input and output must not alias");
  touch(out);
  return op->apply(op, in, out);
}

// Measurement hook (tools/time_op.py): `reps` back-to-back applications of the operator's FUSED STPCG form --
// apply_dir in its recurrence form when the operator has one (the projection matrix is whatever the scalar file
// holds: timing only), else apply_dots, else apply -- between two events on the context stream.
int mi_debug_time_fused_apply(mi_op *op, const mi_vec *in, mi_vec *out, int reps, double *us_per_call) {
  MI_REQUIRE(op && in && out && us_per_call && reps > 0, "bad argument");
  MI_REQUIRE(in->n == op->n && out->n == (op->n_out ? op->n_out : op->n), "operator dimension mismatch");
  mi_ctx *ctx = op->ctx;
  hipEvent_t e0, e1;
  MI_HIP(hipEventCreate(&e0));
  MI_HIP(hipEventCreate(&e1));
  int nparts = 0;
  auto once = [&]() -> int {
    if (op->dirgram && op->apply_dir) return op->apply_dir(op, in, out, -1, &nparts);
    if (op->apply_dots) return op->apply_dots(op, in, out, &nparts);
    return op->apply(op, in, out);
  };
  for (int i = 0; i < 3; ++i) MI_TRY(once());
  MI_HIP(hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < reps; ++i) MI_TRY(once());
  MI_HIP(hipEventRecord(e1, ctx->stream));
  MI_HIP(hipEventSynchronize(e1));
  float ms = 0;
  MI_HIP(hipEventElapsedTime(&ms, e0, e1));
  *us_per_call = 1e3 * (double)ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return MI_OK;
}

int mi_op_dims(const mi_op *op, size_t *n_in, size_t *n_out) {
  MI_REQUIRE(op, "op is null");
  if (n_in) *n_in = op->n;
  if (n_out) *n_out = op->n_out ? op->n_out : op->n;
  return MI_OK;
}

int mi_op_destroy(mi_op *op) {
  if (!op || op->borrowed) return MI_OK;
  if (op->destroy) op->destroy(op);
  delete op;
  return MI_OK;
}

int mi_precon_create_callback(mi_ctx *ctx, size_t n, mi_apply_fn fn, void *user, mi_precon **out) {
  MI_REQUIRE(ctx && fn && out, "null argument");
  mi_precon *P = new mi_precon();
  P->ctx = ctx;
  P->n = n;
  P->kind = 0;
  P->apply = precon_callback_apply;
  P->destroy = precon_callback_destroy;
  P->impl = new CallbackImpl{fn, user};
  *out = P;
  return MI_OK;
}

int mi_precon_create_diag(mi_ctx *ctx, const mi_vec *dinv, mi_precon **out) {
  MI_REQUIRE(ctx && dinv && out, "null argument");
  MI_REQUIRE(dinv->ctx == ctx, "vector belongs to another context");
  mi_precon *P = new mi_precon();
  P->ctx = ctx;
  P->n = dinv->n;
  P->kind = 1;
  P->data = dinv->d;
  P->apply = precon_diag_apply;
  *out = P;
  return MI_OK;
}

int mi_precon_create_block3(mi_ctx *ctx, const mi_vec *inv_blocks, mi_precon **out) {
  MI_REQUIRE(ctx && inv_blocks && out, "null argument");
  MI_REQUIRE(inv_blocks->ctx == ctx, "vector belongs to another context");
  MI_REQUIRE(inv_blocks->n % 9 == 0, "inverse blocks vector must hold 9 doubles per block");
  mi_precon *P = new mi_precon();
  P->ctx = ctx;
  P->n = inv_blocks->n / 3;
  P->kind = 2;
  P->data = inv_blocks->d;
  P->apply = precon_block3_apply;
  *out = P;
  return MI_OK;
}

int mi_precon_apply(mi_precon *P, const mi_vec *r, mi_vec *v) {
  MI_REQUIRE(P && r && v, "null argument");
  MI_REQUIRE(r->n == P->n && v->n == P->n, "preconditioner dimension mismatch");
  touch(v);
  return P->apply(P, r, v);
}

int mi_precon_destroy(mi_precon *P) {
  if (!P || P->borrowed) return MI_OK;
  if (P->destroy) P->destroy(P);
  delete P;
  return MI_OK;
}

}  // extern "C"
