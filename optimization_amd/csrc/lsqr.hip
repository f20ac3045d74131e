// lsqr.hip -- fused, device-resident LSQR (reference: LinearAlgebra/IterativeSolvers.h:552-855; the
// statement sequence is the one of optimization_amd/include/Optimization/LinearAlgebra/IterativeSolvers.h).
//
// Same construction as the fused STPCG (stpcg.hip): no scalar ever travels to the host inside the loop.
// Each global reduction is left as <= 512 component-major partial rows by the producing kernel and
// re-reduced in the prologue of the consuming kernel by every workgroup (identical code on identical
// data => identical bits, no atomics, no grid barrier); every workgroup then advances the reference's
// scalar recurrences itself (fp contraction off) and workgroup 0 persists the state.  The host enqueues
// speculatively with bounded run-ahead, watching the pinned progress word.
//
// One pass (:696-851) is 2 operator applications + 5 vector kernels:
//   t_y = A v                              (user operator)
//   k_lsqr_u      u = t_y - alpha u ; partial |u|^2                                   :707-708
//   k_lsqr_unorm  [beta, |Abar| estimate] u /= beta                                   :709-711
//   t_x = A' u                             (user operator)
//   k_lsqr_v      v = t_x - beta v ; partial |v|^2                                    :712
//   k_lsqr_vnorm  [alpha] v /= alpha ; partials <w,w>, <x,x>, <w,x>                   :713-714,765,785-786
//   k_lsqr_xw     [both plane rotations, |x| estimate, step lengths, S1-S4] x += t1 w ; w = v + t2 w   :729-837
// Algorithmic bytes per pass (n_x = n_y = N, operators excluded): 3N + 2N + 3N + 4N + 5N = 17 N * 8.
#include <cmath>
#include <utility>

#include "comm_ipc.h"
#include "mi_internal.h"

using namespace mi;

namespace {

enum { LSQR_RUN = 0, LSQR_DONE = 1 };

struct LsqrConst {  // per-solve constants: kernel arguments
  double lambda, sqrt_lambda, btol, Atol, Acond_limit, Delta;
  unsigned long long max_iterations;
};

struct LsqrState {
  double alpha, beta, Anorm, Acond, D_frob_sq, bnorm, rbar_norm, Arnorm;
  double rhobar, phibar, cs2, sn2, z, res2, xx, xnorm;
  double t1, t2;
  unsigned long long k, launches;
  int mode, exit_reason, beta_pos, pad;
};

#define LSQR_FIELDS(X)                                                                                  \
  X(alpha) X(beta) X(Anorm) X(Acond) X(D_frob_sq) X(bnorm) X(rbar_norm) X(Arnorm) X(rhobar) X(phibar) \
  X(cs2) X(sn2) X(z) X(res2) X(xx) X(xnorm) X(t1) X(t2) X(k) X(launches) X(mode) X(exit_reason) X(beta_pos)
__device__ __forceinline__ LsqrState ld(const LsqrState *__restrict__ s) {
  LsqrState r;
#define X(f) r.f = s->f;
  LSQR_FIELDS(X)
#undef X
  r.pad = 0;
  return r;
}
__device__ __forceinline__ void st(LsqrState *__restrict__ d, const LsqrState &s) {
#define X(f) d->f = s.f;
  LSQR_FIELDS(X)
#undef X
}

__device__ __forceinline__ void publish(HostStatus *hs, unsigned long long launches, int done) {
  __hip_atomic_store(&hs->word, (uint64_t)(launches << 1) | (uint64_t)(done ? 1 : 0), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- start-up (:635-674) -------------------------------------------------------------------------
// u = b ; partial |u|^2
__global__ __launch_bounds__(kBlock) void k_lsqr_init_u(size_t ny, const double *__restrict__ b,
                                                        double *__restrict__ u, double *__restrict__ partials) {
  __shared__ double lds[kWaves + 1];
  double acc[1] = {0};
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ny; i += stride) {
    const double v = b[i];
    u[i] = v;
    acc[0] += v * v;
  }
  block_partials_store<1>(acc, lds, partials);
}
// v = t_x (= A' b) ; x = 0 * v ; partial |v|^2
__global__ __launch_bounds__(kBlock) void k_lsqr_init_v(size_t nx, const double *__restrict__ tx,
                                                        double *__restrict__ v, double *__restrict__ x,
                                                        double *__restrict__ partials) {
  __shared__ double lds[kWaves + 1];
  double acc[1] = {0};
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nx; i += stride) {
    const double t = tx[i];
    v[i] = t;
    x[i] = 0 * t;  // `x = 0 * v` (:645): NaN/Inf in A'b propagate exactly like in the reference
    acc[0] += t * t;
  }
  block_partials_store<1>(acc, lds, partials);
}
// [beta = |u|, alpha = |v|]  u /= beta ;  (one launch over max(nx, ny) elements does both vectors)
// then v /= alpha, alpha /= beta, w = v ; state initialisation
// (FOLD: several ranks through the peer-memory layer -- every sum over the ranks completes in the prologue of the kernel
// that consumes it, comm_ipc.h; NoFold: one rank, or RCCL has already summed the partial rows over the ranks)
template <class FOLD>
__global__ __launch_bounds__(kBlock) void k_lsqr_init_scale(size_t nx, size_t ny, LsqrConst c, LsqrState *s0,
                                                            const double *__restrict__ pu, int nu,
                                                            const double *__restrict__ pv, int nv,
                                                            double *__restrict__ u, double *__restrict__ v,
                                                            double *__restrict__ w, HostStatus *hs, FOLD fu,
                                                            FOLD fv) {
#pragma clang fp contract(off)
  __shared__ double lds[kWaves + 1];
  double r1[1], r2[1];
  reduce_rows<1>(pu, nu, r1, lds);
  fold_maybe<1>(r1, fu, lds);
  reduce_rows<1>(pv, nv, r2, lds);
  fold_maybe<1>(r2, fv, lds);
  const double beta = sqrt(r1[0]);
  double alpha = sqrt(r2[0]);
  const bool bpos = beta > 0, apos = alpha > 0;
  const size_t stride = (size_t)gridDim.x * kBlock, i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (bpos)
    for (size_t i = i0; i < ny; i += stride) u[i] = u[i] / beta;
  if (apos)
    for (size_t i = i0; i < nx; i += stride) {
      const double t = v[i] / alpha;
      v[i] = t;
      w[i] = t;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (apos) alpha = alpha / beta;  // :660 (v was built from b, not from the unit vector u)
    LsqrState s;
    s.alpha = alpha;
    s.beta = beta;
    s.Anorm = 0;
    s.Acond = 0;
    s.D_frob_sq = 0;
    s.bnorm = beta;
    s.rbar_norm = beta;
    s.Arnorm = alpha * beta;  // :670
    s.rhobar = alpha;
    s.phibar = beta;
    s.cs2 = -1;
    s.sn2 = 0;
    s.z = 0;
    s.res2 = 0;
    s.xx = 0;
    s.xnorm = 0;
    s.t1 = s.t2 = 0;
    s.k = 0;
    s.launches = 0;
    s.exit_reason = MI_LSQR_EXIT_MAXIT;
    s.beta_pos = 1;
    s.mode = LSQR_RUN;
    if (s.Arnorm == 0) {  // x = 0 already solves the problem (:671-674)
      s.mode = LSQR_DONE;
      s.exit_reason = MI_LSQR_EXIT_TRIVIAL;
    } else if (c.max_iterations == 0) {
      s.mode = LSQR_DONE;
    }
    s.pad = 0;
    st(s0, s);
    publish(hs, 0, s.mode == LSQR_DONE);
  }
}

// ---- one pass ------------------------------------------------------------------------------------
// u = t_y - alpha u ; partial |u|^2                                                     :707-708
__global__ __launch_bounds__(kBlock) void k_lsqr_u(size_t ny, const LsqrState *__restrict__ s_in,
                                                   const double *__restrict__ ty, double *__restrict__ u,
                                                   double *__restrict__ partials) {
  __shared__ double lds[kWaves + 1];
  if (s_in->mode != LSQR_RUN) return;
  const double alpha = s_in->alpha;
  double acc[1] = {0};
  const size_t n2 = ny >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 t = reinterpret_cast<const double2 *>(ty)[i];
    double2 uu = reinterpret_cast<double2 *>(u)[i];
    uu.x = t.x - alpha * uu.x;
    uu.y = t.y - alpha * uu.y;
    reinterpret_cast<double2 *>(u)[i] = uu;
    acc[0] += uu.x * uu.x;
    acc[0] += uu.y * uu.y;
  }
  if ((ny & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double uu = ty[ny - 1] - alpha * u[ny - 1];
    u[ny - 1] = uu;
    acc[0] += uu * uu;
  }
  block_partials_store<1>(acc, lds, partials);
}

// [beta = |u| ; |Abar| estimate] u /= beta                                              :708-711
template <class FOLD>
// (80 scalar registers: the FoldArgs instantiation would take 84; see k_lsqr_xw)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_lsqr_unorm(size_t ny, LsqrConst c, const LsqrState *__restrict__ s_in,
                                                       LsqrState *__restrict__ s_out,
                                                       const double *__restrict__ partials, int nparts,
                                                       double *__restrict__ u, FOLD fold) {
#pragma clang fp contract(off)
  __shared__ double lds[kWaves + 1];
  LsqrState s = ld(s_in);
  if (s.mode != LSQR_RUN) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st(s_out, s);
    return;
  }
  double r[1];
  reduce_rows<1>(partials, nparts, r, lds);
  fold_maybe<1>(r, fold, lds);
  const double beta = sqrt(r[0]);
  s.beta = beta;
  s.beta_pos = beta > 0;
  if (s.beta_pos) s.Anorm = sqrt(s.Anorm * s.Anorm + s.alpha * s.alpha + beta * beta + c.lambda);  // :711
  if (blockIdx.x == 0 && threadIdx.x == 0) st(s_out, s);
  if (!s.beta_pos) return;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < ny; i += stride) u[i] = u[i] / beta;
}

// v = t_x - beta v ; partial |v|^2                                                      :712-713
__global__ __launch_bounds__(kBlock) void k_lsqr_v(size_t nx, const LsqrState *__restrict__ s_in,
                                                   const double *__restrict__ tx, double *__restrict__ v,
                                                   double *__restrict__ partials) {
  __shared__ double lds[kWaves + 1];
  if (s_in->mode != LSQR_RUN || !s_in->beta_pos) return;
  const double beta = s_in->beta;
  double acc[1] = {0};
  const size_t n2 = nx >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 t = reinterpret_cast<const double2 *>(tx)[i];
    double2 vv = reinterpret_cast<double2 *>(v)[i];
    vv.x = t.x - beta * vv.x;
    vv.y = t.y - beta * vv.y;
    reinterpret_cast<double2 *>(v)[i] = vv;
    acc[0] += vv.x * vv.x;
    acc[0] += vv.y * vv.y;
  }
  if ((nx & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double vv = tx[nx - 1] - beta * v[nx - 1];
    v[nx - 1] = vv;
    acc[0] += vv * vv;
  }
  block_partials_store<1>(acc, lds, partials);
}

// [alpha = |v|] v /= alpha ; partials <w,w>, <x,x>, <w,x>                                :713-714,765,785-786
template <class FOLD>
// (80 scalar registers: the FoldArgs instantiation would take 82; see k_lsqr_xw)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_lsqr_vnorm(size_t nx, const LsqrState *__restrict__ s_in,
                                                       LsqrState *__restrict__ s_out,
                                                       const double *__restrict__ partials_v, int nparts,
                                                       double *__restrict__ v, const double *__restrict__ w,
                                                       const double *__restrict__ x,
                                                       double *__restrict__ partials3, FOLD fold) {
#pragma clang fp contract(off)
  __shared__ double lds[3 * (kWaves + 1)];
  LsqrState s = ld(s_in);
  if (s.mode != LSQR_RUN) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st(s_out, s);
    return;
  }
  double alpha = s.alpha;
  if (s.beta_pos) {
    double r[1];
    reduce_rows<1>(partials_v, nparts, r, lds);
    fold_maybe<1>(r, fold, lds);  // (beta_pos is replicated: every rank takes this branch or none does)
    alpha = sqrt(r[0]);
    s.alpha = alpha;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) st(s_out, s);
  const bool scale = s.beta_pos && alpha > 0;
  double acc[3] = {0, 0, 0};
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nx; i += stride) {
    if (scale) v[i] = v[i] / alpha;
    const double wv = w[i], xv = x[i];
    acc[0] += wv * wv;
    acc[1] += xv * xv;
    acc[2] += wv * xv;
  }
  block_partials_store<3>(acc, lds, partials3);
}

// [rotations, norms, step lengths, stopping rules] x += t1 w ; w = v + t2 w               :729-837
// (at most 80 scalar registers: with the 82..96 it would take, a 1024-thread workgroup gets the CU to itself instead
// of sharing it with a second one -- stpcg.hip, k_cg_update_s80)
template <class FOLD>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_lsqr_xw(size_t nx, LsqrConst c, const LsqrState *__restrict__ s_in,
                                                    LsqrState *__restrict__ s_out,
                                                    const double *__restrict__ partials3, int nparts,
                                                    const double *__restrict__ v, double *__restrict__ w,
                                                    double *__restrict__ x, HostStatus *hs, FOLD fold) {
#pragma clang fp contract(off)
  __shared__ double lds[3 * (kWaves + 1)];
  LsqrState s = ld(s_in);
  const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
  if (s.mode != LSQR_RUN) {
    if (leader) st(s_out, s);
    return;
  }
  double d[3];
  reduce_rows<3>(partials3, nparts, d, lds);
  fold_maybe<3>(d, fold, lds);
  const double w_sq = d[0], xtx = d[1], wtx = d[2];
  const double alpha = s.alpha, beta = s.beta;
  // rotation removing the damping term                                             :729-735
  const double rhobar1 = sqrt(s.rhobar * s.rhobar + c.lambda);
  const double cs1 = s.rhobar / rhobar1;
  const double sn1 = c.sqrt_lambda / rhobar1;
  const double psi = sn1 * s.phibar;
  double phibar = s.phibar * cs1;
  // rotation removing the sub-diagonal beta                                        :740-747
  const double rho = sqrt(rhobar1 * rhobar1 + beta * beta);
  const double cs = rhobar1 / rho;
  const double sn = beta / rho;
  const double theta = sn * alpha;
  s.rhobar = -cs * alpha;
  const double phi = cs * phibar;
  phibar = phibar * sn;
  const double tau = sn * phi;
  // right rotation removing the super-diagonal theta -> estimate of |x|            :753-760
  const double delta = s.sn2 * rho;
  const double gammabar = -s.cs2 * rho;
  const double rhs = phi - delta * s.z;
  const double zbar = rhs / gammabar;
  const double gamma = sqrt(gammabar * gammabar + theta * theta);
  s.cs2 = gammabar / gamma;
  s.sn2 = theta / gamma;
  s.z = rhs / gamma;
  const double d_sq = w_sq / (rho * rho);  // :766
  double xnorm = sqrt(s.xx + zbar * zbar);  // :769
  s.xx = s.xx + s.z * s.z;
  const double t2 = -theta / rho;  // :772
  double t1;
  if (xnorm <= c.Delta) {
    t1 = phi / rho;  // :779
  } else {           // :785-793
    t1 = (-wtx + sqrt(wtx * wtx + w_sq * (c.Delta * c.Delta - xtx))) / w_sq;
    xnorm = c.Delta;
  }
  s.xnorm = xnorm;
  s.t1 = t1;
  s.t2 = t2;
  s.phibar = phibar;
  s.D_frob_sq = s.D_frob_sq + d_sq;          // :802
  s.Acond = s.Anorm * sqrt(s.D_frob_sq);     // :808
  const double res1 = phibar * phibar;
  s.res2 = s.res2 + psi * psi;
  s.rbar_norm = sqrt(res1 + s.res2);         // :812-814
  s.Arnorm = alpha * fabs(tau);              // :818
  s.k = s.k + 1;
  s.launches = s.launches + 1;
  // stopping rules; num_iterations is NOT advanced when the loop is left through a break (:696)
  int exit_reason = -1;
  if (s.rbar_norm <= c.btol * s.bnorm + c.Atol * s.Anorm * xnorm) exit_reason = MI_LSQR_EXIT_S1;  // :825
  else if (s.Arnorm <= c.Atol * s.Anorm * s.rbar_norm) exit_reason = MI_LSQR_EXIT_S2;             // :829
  else if (s.Acond >= c.Acond_limit) exit_reason = MI_LSQR_EXIT_S3;                               // :833
  else if (xnorm >= c.Delta) exit_reason = MI_LSQR_EXIT_S4;                                       // :837
  if (exit_reason >= 0) {
    s.k = s.k - 1;
    s.exit_reason = exit_reason;
    s.mode = LSQR_DONE;
  } else if (s.k >= c.max_iterations) {
    s.exit_reason = MI_LSQR_EXIT_MAXIT;
    s.mode = LSQR_DONE;
  }
  if (leader) {
    st(s_out, s);
    publish(hs, s.launches, s.mode == LSQR_DONE);
  }
  // x += t1 w (:798) ; w = v + t2 w (:799)   -- applied also on the pass that decides the exit
  const size_t n2 = nx >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 vv = reinterpret_cast<const double2 *>(v)[i];
    double2 ww = reinterpret_cast<double2 *>(w)[i];
    double2 xx = reinterpret_cast<double2 *>(x)[i];
    xx.x += t1 * ww.x;
    xx.y += t1 * ww.y;
    ww.x = vv.x + t2 * ww.x;
    ww.y = vv.y + t2 * ww.y;
    reinterpret_cast<double2 *>(x)[i] = xx;
    reinterpret_cast<double2 *>(w)[i] = ww;
  }
  if ((nx & 1) && leader) {
    x[nx - 1] += t1 * w[nx - 1];
    w[nx - 1] = v[nx - 1] + t2 * w[nx - 1];
  }
}

inline void cpu_relax() { __builtin_ia32_pause(); }

}  // namespace

extern "C" {

void mi_lsqr_default_params(mi_lsqr_params *p) {
  if (!p) return;
  p->max_iterations = 1000;  // :558
  p->lambda = 0;
  p->btol = 1e-6;
  p->Atol = 1e-6;
  p->Acond_limit = 1e8;
  p->Delta = std::sqrt(1.7976931348623157e308);  // sqrt(numeric_limits<double>::max())  :559
  p->run_ahead = 3;
}

int mi_lsqr(mi_ctx *ctx, mi_op *A, mi_op *At, const mi_vec *b, const mi_lsqr_params *prm, mi_vec *x_out,
            mi_lsqr_result *result) {
  MI_REQUIRE(ctx && A && At && b && prm && x_out && result, "null argument");
  touch(x_out);
  ctx->fusion.fused_lsqr_solves++;
  RangeScope range("mi_lsqr");
  // the reference's own argument checks (:573-590), same messages
  MI_REQUIRE(!(prm->lambda < 0), "Tikhonov regularization parameter (lambda) must be a nonnegative real value");
  MI_REQUIRE(!(prm->btol < 0), "Stopping tolerance btol must be a nonnegative real number");
  MI_REQUIRE(!(prm->Atol < 0), "Stopping tolerance Atol must be a nonnegative real number");
  MI_REQUIRE(prm->Acond_limit > 0, "Stopping tolerance Abar_cond_limit must be a positive real number");
  MI_REQUIRE(prm->Delta > 0, "Trust-region radius (Delta) must be a positive real value");
  const size_t nx = A->n, ny = A->n_out ? A->n_out : A->n;
  MI_REQUIRE(At->n == ny && (At->n_out ? At->n_out : At->n) == nx, "A and A' have inconsistent dimensions");
  MI_REQUIRE(b->n == ny && x_out->n == nx, "b / x have the wrong length");
  // Several ranks (r03): x, b and every work vector are the rank's ROW SLABS; the operators do their own exchange (a
  // callback's business; the built-in CSR operators of a row-sharded SYMMETRIC matrix exchange halos themselves, and
  // A' is then the same operator); the five reductions of a pass complete across the ranks either inside the
  // consumer's prologue (peer-memory layer, comm_ipc.h) or by an in-stream RCCL all-reduce of the partial rows.
  // (the predicates of mi_stpcg: a size-1 communicator takes the same code path as 8 ranks -- that is how a 1-GPU
  // box exercises the RCCL calls, tests/test_gpu_comm.py)
  const bool multi = (ctx->comm != nullptr && ctx->world_size > 1) || ctx->force_lockstep;
  const bool folded = comm_ipc_enabled(ctx);
  const bool rows = rows_mode(ctx);
  // a communicator of several ranks whose reductions would stay rank-local (RCCL with the slot path forced: neither
  // folded nor rows) must not run: every norm would be a partial sum, the ranks would leave the loop at different passes
  MI_REQUIRE(!(ctx->comm != nullptr && ctx->world_size > 1) || folded || rows,
             "mi_lsqr on %d ranks needs the peer-memory layer or the RCCL rows mode (MI355OPT_FORCE_SLOT_PATH is set?)",
             ctx->world_size);
  static_assert(sizeof(LsqrState) <= 256, "state slots are 256 bytes apart");

  mi_vec *u = nullptr, *v = nullptr, *w = nullptr, *ty = nullptr, *tx = nullptr;
  MI_TRY(mi_vec_create(ctx, ny, &u));
  MI_TRY(mi_vec_create(ctx, nx, &v));
  MI_TRY(mi_vec_create(ctx, nx, &w));
  MI_TRY(mi_vec_create(ctx, ny, &ty));
  MI_TRY(mi_vec_create(ctx, nx, &tx));
  void *sraw = nullptr;
  MI_TRY(pool_alloc(ctx, 512, &sraw));
  // two state copies; a kernel reads one and writes the other (u:a  unorm:a->b  v:b  vnorm:b->a  xw:a->b),
  // so a pass that starts in `a` ends in `b` and the roles swap from pass to pass
  LsqrState *sa = (LsqrState *)sraw, *sb = (LsqrState *)((char *)sraw + 256);
  int ret = MI_OK;
  hipStream_t stq = ctx->stream;
  const LsqrConst c{prm->lambda, std::sqrt(prm->lambda), prm->btol, prm->Atol, prm->Acond_limit, prm->Delta,
                    (unsigned long long)prm->max_iterations};
  const int run_ahead = prm->run_ahead > 0 ? prm->run_ahead : 3;
  const int gx = grid_for(ctx, nx, 4), gy = grid_for(ctx, ny, 4), gmax = gx > gy ? gx : gy;
  double *pa = ctx->partials, *pb = ctx->partials_b, *p3 = ctx->partials2;
  size_t applies = 0;
  ctx->epoch++;
  ctx->status->word = 0;
  ctx->status->epoch = ctx->epoch;
  ctx->cg_live = nullptr;

#define LQ_CHECK(expr) \
  do {                 \
    int _s = (expr);   \
    if (_s != MI_OK) { \
      ret = _s;        \
      goto cleanup;    \
    }                  \
  } while (0)

  // --- start-up: beta u = b, alpha v = A' u (:635-667) --------------------------------------------
  MI_HIP(hipMemsetAsync(w->d, 0, nx * sizeof(double), stq));
  hipLaunchKernelGGL(k_lsqr_init_u, dim3(gy), dim3(kBlock), 0, stq, ny, (const double *)b->d, u->d, pa);
  LQ_CHECK(At->apply(At, u, tx));
  ++applies;
  hipLaunchKernelGGL(k_lsqr_init_v, dim3(gx), dim3(kBlock), 0, stq, nx, (const double *)tx->d, v->d, x_out->d, pb);
  if (folded) {
    const FoldArgs fu = comm_fold_next_always(ctx), fv = comm_fold_next_always(ctx);
    hipLaunchKernelGGL(k_lsqr_init_scale<FoldArgs>, dim3(gmax), dim3(kBlock), 0, stq, nx, ny, c, sa, (const double *)pa,
                       gy, (const double *)pb, gx, u->d, v->d, w->d, ctx->status_dev, fu, fv);
  } else {
    if (rows) {
      LQ_CHECK(comm_allreduce_rows(ctx, pa, 1));
      LQ_CHECK(comm_allreduce_rows(ctx, pb, 1));
    }
    hipLaunchKernelGGL(k_lsqr_init_scale<NoFold>, dim3(gmax), dim3(kBlock), 0, stq, nx, ny, c, sa, (const double *)pa,
                       gy, (const double *)pb, gx, u->d, v->d, w->d, ctx->status_dev, NoFold{}, NoFold{});
  }
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) LQ_CHECK(hip_fail(e, "lsqr init launch", __FILE__, __LINE__));
  }

  // --- passes: speculative enqueue with bounded run-ahead --------------------------------------------
  for (size_t k = 0; k < prm->max_iterations; ++k) {
    uint64_t wd = ctx->status->word;
    while (!(wd & 1) && k > (wd >> 1) + (uint64_t)run_ahead) {
      cpu_relax();
      wd = ctx->status->word;
    }
    // (several ranks: every enqueued pass carries exchanges, so all ranks enqueue the SAME number -- launches at the
    // exit + run_ahead, a function of the replicated device state only; cf. mi_stpcg)
    if ((wd & 1) && (!multi || k >= (wd >> 1) + (uint64_t)run_ahead)) break;
    // operators that can fold `out = Op(in) - scale out` and |out|^2 into their own pass (built-in CSR) save a
    // kernel and a round trip of the product through memory per application
    int nu = gy, nv = gx;
    if (A->apply_sub_scaled) {
      LQ_CHECK(A->apply_sub_scaled(A, v, &sa->alpha, &sa->mode, nullptr, u, pa, &nu));
    } else {
      LQ_CHECK(A->apply(A, v, ty));
      hipLaunchKernelGGL(k_lsqr_u, dim3(gy), dim3(kBlock), 0, stq, ny, (const LsqrState *)sa, (const double *)ty->d,
                         u->d, pa);
    }
    if (folded) {
      hipLaunchKernelGGL(k_lsqr_unorm<FoldArgs>, dim3(gy), dim3(kBlock), 0, stq, ny, c, (const LsqrState *)sa, sb,
                         (const double *)pa, nu, u->d, comm_fold_next_always(ctx));
    } else {
      if (rows) LQ_CHECK(comm_allreduce_rows(ctx, pa, 1));
      hipLaunchKernelGGL(k_lsqr_unorm<NoFold>, dim3(gy), dim3(kBlock), 0, stq, ny, c, (const LsqrState *)sa, sb,
                         (const double *)pa, nu, u->d, NoFold{});
    }
    if (At->apply_sub_scaled) {
      LQ_CHECK(At->apply_sub_scaled(At, u, &sb->beta, &sb->mode, &sb->beta_pos, v, pb, &nv));
    } else {
      LQ_CHECK(At->apply(At, u, tx));
      hipLaunchKernelGGL(k_lsqr_v, dim3(gx), dim3(kBlock), 0, stq, nx, (const LsqrState *)sb, (const double *)tx->d,
                         v->d, pb);
    }
    applies += 2;
    if (folded) {
      hipLaunchKernelGGL(k_lsqr_vnorm<FoldArgs>, dim3(gx), dim3(kBlock), 0, stq, nx, (const LsqrState *)sb, sa,
                         (const double *)pb, nv, v->d, (const double *)w->d, (const double *)x_out->d, p3,
                         comm_fold_next_always(ctx));
      hipLaunchKernelGGL(k_lsqr_xw<FoldArgs>, dim3(gx), dim3(kBlock), 0, stq, nx, c, (const LsqrState *)sa, sb,
                         (const double *)p3, gx, (const double *)v->d, w->d, x_out->d, ctx->status_dev,
                         comm_fold_next_always(ctx));
    } else {
      if (rows) LQ_CHECK(comm_allreduce_rows(ctx, pb, 1));
      hipLaunchKernelGGL(k_lsqr_vnorm<NoFold>, dim3(gx), dim3(kBlock), 0, stq, nx, (const LsqrState *)sb, sa,
                         (const double *)pb, nv, v->d, (const double *)w->d, (const double *)x_out->d, p3, NoFold{});
      if (rows) LQ_CHECK(comm_allreduce_rows(ctx, p3, 3));
      hipLaunchKernelGGL(k_lsqr_xw<NoFold>, dim3(gx), dim3(kBlock), 0, stq, nx, c, (const LsqrState *)sa, sb,
                         (const double *)p3, gx, (const double *)v->d, w->d, x_out->d, ctx->status_dev, NoFold{});
    }
    std::swap(sa, sb);
  }
  {
    LsqrState h;
    hipError_t e = hipMemcpyAsync(&h, sa, sizeof(LsqrState), hipMemcpyDeviceToHost, stq);
    if (e == hipSuccess) e = hipStreamSynchronize(stq);
    ctx->host_syncs++;
    if (e != hipSuccess) LQ_CHECK(hip_fail(e, "lsqr read-back", __FILE__, __LINE__));
    {
      int ipc_err = 0;
      (void)mi_comm_ipc_error(ctx, &ipc_err);
      if (ipc_err) {
        set_error("a wait in the peer-memory exchange layer timed out: the result of this solve is invalid");
        ret = MI_ERR_COMM;
        goto cleanup;
      }
    }
    result->xnorm = h.xnorm;
    result->num_iterations = (size_t)h.k;
    result->exit_reason = h.exit_reason;
    result->rbar_norm = h.rbar_norm;
    result->Arnorm = h.Arnorm;
    result->Anorm = h.Anorm;
    result->Acond = h.Acond;
    result->operator_applications = applies;
  }
cleanup:
#undef LQ_CHECK
  mi_vec_destroy(u);
  mi_vec_destroy(v);
  mi_vec_destroy(w);
  mi_vec_destroy(ty);
  mi_vec_destroy(tx);
  pool_free(ctx, sraw);
  return ret;
}

}  // extern "C"
