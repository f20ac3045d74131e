// stpcg.hip -- fused device realisation of Optimization::LinearAlgebra::STPCG
// (reference: LinearAlgebra/IterativeSolvers.h:166-426, unconstrained form).
//
// Design (MI355X-first):
//   * every vector phase between two global reductions is ONE streaming kernel (16 B/lane loads,
//     <= 2048 workgroups, grid-stride), producing per-workgroup partial sums;
//   * the scalar recurrences (:259-283,330-345,412-417) live in a device-resident CgState and are
//     advanced by one-workgroup "scalar" kernels that also sum the partials in a fixed order
//     (deterministic; no fp64 atomics) and take every branch decision of the reference loop;
//   * the host never reads a scalar back inside the loop: it enqueues iterations speculatively,
//     bounded by `run_ahead`, and watches a pinned progress word written by the scalar kernels;
//     kernels enqueued past the exit see mode != CG_RUN and return immediately;
//   * row-sharded multi-GPU: partials -> scalar slots -> in-stream RCCL all-reduce -> scalar kernel
//     (the recurrences are replicated; all ranks take identical decisions).
//
// Algorithmic HBM traffic per completed iteration (N fp64 per vector), excluding the operator:
//   dots 2N (or fused into the operator's last kernel), update 6N, direction 3N  => 88*N bytes;
//   +2N with a diagonal preconditioner (104*N), +4N with 3x3 block-Jacobi (120*N).
#include "mi_internal.h"

#include <cmath>

using namespace mi;

namespace {

enum { PRE_NONE = 0, PRE_DIAG = 1, PRE_BLOCK3 = 2, PRE_EXTERNAL = 3 };

// ---------------------------------------------------------------------------------------------
// vector kernels
// ---------------------------------------------------------------------------------------------

// r = g; s = 0*g (:211,214); v = P r (:231/234); p = -v (:256); partial <r,v> (:266)
template <int PRE>
__global__ __launch_bounds__(kBlock) void k_cg_init(size_t n, const double *__restrict__ g,
                                                    const double *__restrict__ pre,
                                                    double *__restrict__ r, double *__restrict__ v,
                                                    double *__restrict__ p, double *__restrict__ s,
                                                    double *__restrict__ partials) {
  __shared__ double lds[8];
  double acc = 0;
  const size_t stride = (size_t)gridDim.x * kBlock;
  if (PRE == PRE_BLOCK3) {
    const size_t nb = n / 3;
    for (size_t b = (size_t)blockIdx.x * kBlock + threadIdx.x; b < nb; b += stride) {
      const double g0 = g[3 * b], g1 = g[3 * b + 1], g2 = g[3 * b + 2];
      const double *M = pre + 9 * b;
      const double v0 = M[0] * g0 + M[1] * g1 + M[2] * g2;
      const double v1 = M[3] * g0 + M[4] * g1 + M[5] * g2;
      const double v2 = M[6] * g0 + M[7] * g1 + M[8] * g2;
      r[3 * b] = g0; r[3 * b + 1] = g1; r[3 * b + 2] = g2;
      s[3 * b] = 0 * g0; s[3 * b + 1] = 0 * g1; s[3 * b + 2] = 0 * g2;
      v[3 * b] = v0; v[3 * b + 1] = v1; v[3 * b + 2] = v2;
      p[3 * b] = -v0; p[3 * b + 1] = -v1; p[3 * b + 2] = -v2;
      acc += g0 * v0; acc += g1 * v1; acc += g2 * v2;
    }
  } else {
    const size_t n2 = n >> 1;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
      const double2 gv = reinterpret_cast<const double2 *>(g)[i];
      reinterpret_cast<double2 *>(r)[i] = gv;
      reinterpret_cast<double2 *>(s)[i] = make_double2(0 * gv.x, 0 * gv.y);
      if (PRE == PRE_EXTERNAL) continue;
      double2 vv = gv;
      if (PRE == PRE_DIAG) {
        const double2 d = reinterpret_cast<const double2 *>(pre)[i];
        vv.x = d.x * gv.x; vv.y = d.y * gv.y;
        reinterpret_cast<double2 *>(v)[i] = vv;
      }
      reinterpret_cast<double2 *>(p)[i] = make_double2(-vv.x, -vv.y);
      acc += gv.x * vv.x; acc += gv.y * vv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const size_t i = n - 1;
      const double gv = g[i];
      r[i] = gv; s[i] = 0 * gv;
      if (PRE != PRE_EXTERNAL) {
        double vv = gv;
        if (PRE == PRE_DIAG) { vv = pre[i] * gv; v[i] = vv; }
        p[i] = -vv;
        acc += gv * vv;
      }
    }
  }
  if (PRE == PRE_EXTERNAL) return;
  const double t = block_reduce_sum(acc, lds);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.x * kPartialStride] = t;
}

// external preconditioner path: partial <r,v>; optionally p = -v (initialisation, :256)
template <bool NEG_P>
__global__ __launch_bounds__(kBlock) void k_cg_dot_rv(size_t n, const CgState *__restrict__ st,
                                                      const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ p,
                                                      double *__restrict__ partials) {
  __shared__ double lds[8];
  if (!NEG_P && st->mode != CG_RUN) return;
  double acc = 0;
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 rv = reinterpret_cast<const double2 *>(r)[i];
    const double2 vv = reinterpret_cast<const double2 *>(v)[i];
    if (NEG_P) reinterpret_cast<double2 *>(p)[i] = make_double2(-vv.x, -vv.y);
    acc += rv.x * vv.x; acc += rv.y * vv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    if (NEG_P) p[n - 1] = -v[n - 1];
    acc += r[n - 1] * v[n - 1];
  }
  const double t = block_reduce_sum(acc, lds);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.x * kPartialStride] = t;
}

// partials of <x,y>, <y,y>, <x,x>  (kappa :300, kernel test :305-306) for operators without fused dots
__global__ __launch_bounds__(kBlock) void k_cg_dot3(size_t n, const double *__restrict__ x,
                                                    const double *__restrict__ y,
                                                    double *__restrict__ partials) {
  __shared__ double lds[8];
  double a0 = 0, a1 = 0, a2 = 0;
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[i];
    const double2 yv = reinterpret_cast<const double2 *>(y)[i];
    a0 += xv.x * yv.x; a0 += xv.y * yv.y;
    a1 += yv.x * yv.x; a1 += yv.y * yv.y;
    a2 += xv.x * xv.x; a2 += xv.y * xv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double xv = x[n - 1], yv = y[n - 1];
    a0 += xv * yv; a1 += yv * yv; a2 += xv * xv;
  }
  const double t0 = block_reduce_sum(a0, lds);
  const double t1 = block_reduce_sum(a1, lds);
  const double t2 = block_reduce_sum(a2, lds);
  if (threadIdx.x == 0) {
    double *o = partials + (size_t)blockIdx.x * kPartialStride;
    o[0] = t0; o[1] = t1; o[2] = t2;
  }
}

// CG_RUN:            s = s + alpha p (:374); r += alpha Hp (:377); v = P r (:383/386); partial <r,v> (:408)
// CG_APPLY_SIGMA:    s += sigma p (:360)
// CG_KERNEL_PENDING: partial <p,r> (:320)
template <int PRE>
__global__ __launch_bounds__(kBlock) void k_cg_update(size_t n, const CgState *__restrict__ st,
                                                      const double *__restrict__ p,
                                                      const double *__restrict__ Hp,
                                                      const double *__restrict__ pre,
                                                      double *__restrict__ s, double *__restrict__ r,
                                                      double *__restrict__ v,
                                                      double *__restrict__ partials) {
  __shared__ double lds[8];
  const int mode = st->mode;
  if (mode == CG_DONE || mode == CG_APPLY_SIGMA_LATE) return;
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  double acc = 0;
  if (mode == CG_APPLY_SIGMA) {
    const double sigma = st->sigma;
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      double2 sv = reinterpret_cast<double2 *>(s)[i];
      sv.x += sigma * pv.x; sv.y += sigma * pv.y;
      reinterpret_cast<double2 *>(s)[i] = sv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s[n - 1] += sigma * p[n - 1];
    return;
  }
  if (mode == CG_KERNEL_PENDING) {
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      const double2 rv = reinterpret_cast<const double2 *>(r)[i];
      acc += pv.x * rv.x; acc += pv.y * rv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += p[n - 1] * r[n - 1];
  } else {  // CG_RUN
    const double alpha = st->alpha;
    if (PRE == PRE_BLOCK3) {
      const size_t nb = n / 3;
      for (size_t b = i0; b < nb; b += stride) {
        double rr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const size_t i = 3 * b + c;
          s[i] = s[i] + alpha * p[i];
          rr[c] = r[i] + alpha * Hp[i];
          r[i] = rr[c];
        }
        const double *M = pre + 9 * b;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double vv = M[3 * c] * rr[0] + M[3 * c + 1] * rr[1] + M[3 * c + 2] * rr[2];
          v[3 * b + c] = vv;
          acc += rr[c] * vv;
        }
      }
    } else {
      for (size_t i = i0; i < n2; i += stride) {
        const double2 pv = reinterpret_cast<const double2 *>(p)[i];
        const double2 hv = reinterpret_cast<const double2 *>(Hp)[i];
        double2 sv = reinterpret_cast<double2 *>(s)[i];
        double2 rv = reinterpret_cast<double2 *>(r)[i];
        sv.x = sv.x + alpha * pv.x; sv.y = sv.y + alpha * pv.y;
        rv.x += alpha * hv.x; rv.y += alpha * hv.y;
        reinterpret_cast<double2 *>(s)[i] = sv;
        reinterpret_cast<double2 *>(r)[i] = rv;
        if (PRE == PRE_EXTERNAL) continue;
        double2 vv = rv;
        if (PRE == PRE_DIAG) {
          const double2 d = reinterpret_cast<const double2 *>(pre)[i];
          vv.x = d.x * rv.x; vv.y = d.y * rv.y;
          reinterpret_cast<double2 *>(v)[i] = vv;
        }
        acc += rv.x * vv.x; acc += rv.y * vv.y;
      }
      if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const size_t i = n - 1;
        s[i] = s[i] + alpha * p[i];
        const double rv = r[i] + alpha * Hp[i];
        r[i] = rv;
        if (PRE != PRE_EXTERNAL) {
          double vv = rv;
          if (PRE == PRE_DIAG) { vv = pre[i] * rv; v[i] = vv; }
          acc += rv * vv;
        }
      }
    }
    if (PRE == PRE_EXTERNAL) return;
  }
  const double t = block_reduce_sum(acc, lds);
  if (threadIdx.x == 0) partials[(size_t)blockIdx.x * kPartialStride] = t;
}

// CG_RUN: p = -v + beta p (:420);  CG_APPLY_SIGMA_LATE: s += sigma p (:336)
__global__ __launch_bounds__(kBlock) void k_cg_pupdate(size_t n, const CgState *__restrict__ st,
                                                       const double *__restrict__ v,
                                                       double *__restrict__ p, double *__restrict__ s) {
  const int mode = st->mode;
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (mode == CG_RUN) {
    const double beta = st->beta;
    for (size_t i = i0; i < n2; i += stride) {
      const double2 vv = reinterpret_cast<const double2 *>(v)[i];
      double2 pv = reinterpret_cast<double2 *>(p)[i];
      pv.x = -vv.x + beta * pv.x; pv.y = -vv.y + beta * pv.y;
      reinterpret_cast<double2 *>(p)[i] = pv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) p[n - 1] = -v[n - 1] + beta * p[n - 1];
  } else if (mode == CG_APPLY_SIGMA_LATE) {
    const double sigma = st->sigma;
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      double2 sv = reinterpret_cast<double2 *>(s)[i];
      sv.x += sigma * pv.x; sv.y += sigma * pv.y;
      reinterpret_cast<double2 *>(s)[i] = sv;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) s[n - 1] += sigma * p[n - 1];
  }
}

// ---------------------------------------------------------------------------------------------
// scalar kernels (one workgroup).  fp contraction is off so the recurrences round exactly like
// the reference's scalar C++ on the host.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void publish(HostStatus *hs, unsigned long long k, int done) {
  __hip_atomic_store(&hs->iters_done, (uint64_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (done) __hip_atomic_store(&hs->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct CgSetup {
  double Delta, kappa_fgr, theta, epsilon;
  unsigned long long max_iterations;
  unsigned int epoch;
};

// input values come either from per-workgroup partials (single GPU) or from all-reduced slots
template <bool FROM_SLOTS>
__device__ __forceinline__ double fetch(const double *partials, int nparts, const double *slots, int c,
                                        double *lds) {
  if (FROM_SLOTS) return slots[c];
  return reduce_partials(partials, nparts, c, lds);
}

template <bool FROM_SLOTS>
__global__ __launch_bounds__(kBlock) void k_cg_scalar_init(CgState *st, CgSetup cfg,
                                                           const double *partials, int nparts,
                                                           const double *slots, HostStatus *hs) {
#pragma clang fp contract(off)
  __shared__ double lds[8];
  const double rv0 = fetch<FROM_SLOTS>(partials, nparts, slots, 0, lds);
  if (threadIdx.x != 0) return;
  st->sk_M_pk = 0;                 // :259
  st->sk_M_2 = 0;                  // :263
  st->pk_M_2 = rv0;                // :266
  st->Delta_2 = cfg.Delta * cfg.Delta;  // :271
  const double r0_norm = sqrt(rv0);     // :275
  const double pw = pow(r0_norm, cfg.theta);
  st->target_rk_norm = r0_norm * ((pw < cfg.kappa_fgr) ? pw : cfg.kappa_fgr);  // :278-279
  st->rv = rv0;
  st->alpha = st->beta = st->kappa = st->sigma = 0;
  st->M_norm = 0;
  st->kappa_fgr = cfg.kappa_fgr; st->theta = cfg.theta; st->epsilon = cfg.epsilon;
  st->k = 0;
  st->max_iterations = cfg.max_iterations;
  st->epoch = cfg.epoch;
  st->exit_reason = MI_STPCG_EXIT_MAXIT;
  int mode = CG_RUN;
  if (cfg.max_iterations == 0) {  // :285 loop body never runs
    mode = CG_DONE;
    st->M_norm = sqrt(0.0);  // :424
  } else if (sqrt(rv0) <= st->target_rk_norm) {  // :290
    mode = CG_DONE;
    st->exit_reason = MI_STPCG_EXIT_RESIDUAL;
    st->M_norm = sqrt(0.0);
  }
  st->mode = mode;
  publish(hs, 0, mode == CG_DONE);
}

// after the operator: kappa, kernel test, alpha, boundary test (:300-362)
template <bool FROM_SLOTS>
__global__ __launch_bounds__(kBlock) void k_cg_scalar_a(CgState *st, const double *partials, int nparts,
                                                        const double *slots, HostStatus *hs) {
#pragma clang fp contract(off)
  __shared__ double lds[8];
  const int mode = st->mode;
  if (mode == CG_DONE) return;
  if (mode == CG_APPLY_SIGMA_LATE) {  // the late boundary step has been applied by k_cg_pupdate
    if (threadIdx.x == 0) {
      st->mode = CG_DONE;
      publish(hs, st->k, 1);
    }
    return;
  }
  const double pHp = fetch<FROM_SLOTS>(partials, nparts, slots, 0, lds);
  const double HpHp = fetch<FROM_SLOTS>(partials, nparts, slots, 1, lds);
  const double pp = fetch<FROM_SLOTS>(partials, nparts, slots, 2, lds);
  if (threadIdx.x != 0) return;
  const double kappa = pHp;  // :300
  st->kappa = kappa;
  if (sqrt(HpHp) / sqrt(pp) < st->epsilon) {  // :305-307
    st->mode = CG_KERNEL_PENDING;            // needs <p,r> (:320) -> k_cg_update
    return;
  }
  const double alpha = st->rv / kappa;  // :341
  const double skplus1_M_2 =
      st->sk_M_2 + 2 * alpha * st->sk_M_pk + alpha * alpha * st->pk_M_2;  // :344-345
  if ((kappa <= 0) || (skplus1_M_2 > st->Delta_2)) {                     // :347
    const double sk_M_pk = st->sk_M_pk;
    st->sigma = (-sk_M_pk + sqrt(sk_M_pk * sk_M_pk + st->pk_M_2 * (st->Delta_2 - st->sk_M_2))) /
                st->pk_M_2;        // :355-357
    st->mode = CG_APPLY_SIGMA;
    st->exit_reason = MI_STPCG_EXIT_BOUNDARY;
    return;
  }
  st->alpha = alpha;
  st->beta = skplus1_M_2;  // parked until k_cg_scalar_b consumes it
}

// after the update: beta and the M-norm recurrences (:408-417), loop control (:285,290)
template <bool FROM_SLOTS>
__global__ __launch_bounds__(kBlock) void k_cg_scalar_b(CgState *st, double Delta, const double *partials,
                                                        int nparts, const double *slots, HostStatus *hs,
                                                        double *trace, size_t trace_cap) {
#pragma clang fp contract(off)
  __shared__ double lds[8];
  const int mode = st->mode;
  if (mode == CG_DONE || mode == CG_APPLY_SIGMA_LATE) return;
  if (mode == CG_APPLY_SIGMA) {  // boundary step applied by k_cg_update (:359-361)
    if (threadIdx.x == 0) {
      st->M_norm = Delta;
      st->mode = CG_DONE;
      publish(hs, st->k, 1);
    }
    return;
  }
  const double red = fetch<FROM_SLOTS>(partials, nparts, slots, 0, lds);
  if (threadIdx.x != 0) return;
  if (mode == CG_KERNEL_PENDING) {  // :320-337
    double sk_M_pk = st->sk_M_pk;
    const bool flip = red < 0;  // <p,r> < 0
    if (flip) sk_M_pk *= -1;    // :325
    const double sigma =
        (-sk_M_pk + sqrt(sk_M_pk * sk_M_pk + st->pk_M_2 * (st->Delta_2 - st->sk_M_2))) / st->pk_M_2;
    st->sk_M_pk = sk_M_pk;
    st->sigma = flip ? -sigma : sigma;  // s += sigma * (-p)  ==  s += (-sigma) * p
    st->M_norm = Delta;                 // :334
    st->exit_reason = MI_STPCG_EXIT_KERNEL;
    st->mode = CG_APPLY_SIGMA_LATE;
    return;
  }
  // CG_RUN
  const double rk_vk = red;                         // :408
  const double alpha = st->alpha, kappa = st->kappa;
  const double beta = rk_vk / (alpha * kappa);      // :412
  const double skplus1_M_2 = st->beta;              // parked by k_cg_scalar_a
  st->sk_M_2 = skplus1_M_2;                         // :415
  st->sk_M_pk = beta * (st->sk_M_pk + alpha * st->pk_M_2);  // :416
  st->pk_M_2 = rk_vk + beta * beta * st->pk_M_2;            // :417
  st->rv = rk_vk;
  st->beta = beta;
  const unsigned long long k = st->k;
  if (trace && k < trace_cap) {
    trace[k] = alpha;
    trace[trace_cap + k] = beta;
    trace[2 * trace_cap + k] = kappa;
    trace[3 * trace_cap + k] = rk_vk;
  }
  st->k = k + 1;
  int done = 0;
  if (k + 1 >= st->max_iterations) {  // :285
    done = 1;
    st->exit_reason = MI_STPCG_EXIT_MAXIT;
  } else if (sqrt(rk_vk) <= st->target_rk_norm) {  // :290 (evaluated at the top of the next pass)
    done = 1;
    st->exit_reason = MI_STPCG_EXIT_RESIDUAL;
  }
  if (done) {
    st->M_norm = sqrt(st->sk_M_2);  // :424
    st->mode = CG_DONE;
  }
  publish(hs, k + 1, done);
}

__global__ __launch_bounds__(kBlock) void k_reduce_partials_to_slots(const double *__restrict__ partials,
                                                                     int count, int k,
                                                                     double *__restrict__ slots) {
  __shared__ double lds[8];
  for (int c = 0; c < k; ++c) {
    const double t = reduce_partials(partials, count, c, lds);
    if (threadIdx.x == 0) slots[c] = t;
  }
}

inline void cpu_relax() { __builtin_ia32_pause(); }

}  // namespace

namespace mi {
int launch_dot3_partials(mi_ctx *ctx, size_t n, const double *x, const double *y, int *nparts) {
  const int grid = grid_for(n, 8);
  KScope ks(ctx, MI_K_CG_DOT3);
  hipLaunchKernelGGL(k_cg_dot3, dim3(grid), dim3(kBlock), 0, ctx->stream, n, x, y, ctx->partials);
  *nparts = grid;
  return MI_OK;
}
}  // namespace mi

extern "C" {

void mi_stpcg_default_params(mi_stpcg_params *p) {
  if (!p) return;
  p->Delta = 1.0;
  p->max_iterations = 1000;  // :172
  p->kappa_fgr = .1;
  p->theta = .5;
  p->epsilon = 1e-8;  // :179
  p->run_ahead = 3;
}

int mi_stpcg(mi_ctx *ctx, const mi_vec *g, mi_op *H, mi_precon *P, const mi_stpcg_params *prm,
             mi_vec *s_out, mi_stpcg_result *result, mi_stpcg_trace *trace) {
  MI_REQUIRE(ctx && g && H && prm && s_out && result, "null argument");
  MI_REQUIRE(g->ctx == ctx && s_out->ctx == ctx && H->ctx == ctx, "objects belong to another context");
  MI_REQUIRE(g->n == s_out->n && g->n == H->n, "dimension mismatch: g %zu, s %zu, H %zu", g->n,
             s_out->n, H->n);
  MI_REQUIRE(!P || (P->ctx == ctx && P->n == g->n), "preconditioner dimension/context mismatch");
  // reference argument checks, IterativeSolvers.h:183-205
  MI_REQUIRE(prm->Delta > 0, "Trust-region radius (Delta) must be a positive real value");
  MI_REQUIRE(prm->kappa_fgr >= 0 && prm->kappa_fgr < 1,
             "Target fractional reduction of the gradient norm (kappa_fgr) must be a real value in "
             "the range [0,1)");
  MI_REQUIRE(prm->theta >= 0 && prm->theta <= 1,
             "Target superlinear convergence rate (theta) must be a real value in the range [0,1]");
  MI_REQUIRE(prm->epsilon > 0 && prm->epsilon < 1,
             "Relative norm tolerance for declaring a vector to lie in the kernel of H (epsilon) "
             "should be a small positive number in the range (0,1)");

  const size_t n = g->n;
  const int pre = !P ? PRE_NONE : (P->kind == 1 ? PRE_DIAG : (P->kind == 2 ? PRE_BLOCK3 : PRE_EXTERNAL));
  MI_REQUIRE(pre != PRE_BLOCK3 || n % 3 == 0, "block-Jacobi preconditioner needs n divisible by 3");
  const bool sharded = ctx->world_size > 1;
  const int run_ahead = prm->run_ahead > 0 ? prm->run_ahead : 3;

  mi_vec *r = nullptr, *v = nullptr, *p = nullptr, *Hp = nullptr;
  MI_TRY(mi_vec_create(ctx, n, &r));
  MI_TRY(mi_vec_create(ctx, n, &p));
  MI_TRY(mi_vec_create(ctx, n, &Hp));
  if (pre != PRE_NONE) MI_TRY(mi_vec_create(ctx, n, &v));
  double *vd = (pre == PRE_NONE) ? r->d : v->d;  // v aliases r when P is absent (:231,383)
  const double *pred = P ? P->data : nullptr;

  // trace storage
  size_t tcap = 0;
  if (trace && trace->cap) {
    tcap = trace->cap;
    if (ctx->trace_cap < tcap) {
      if (ctx->trace_dev) MI_HIP(hipFree(ctx->trace_dev));
      MI_HIP(hipMalloc((void **)&ctx->trace_dev, 4 * tcap * sizeof(double)));
      ctx->trace_cap = tcap;
    }
    tcap = ctx->trace_cap;  // layout stride
  }

  ctx->epoch++;
  ctx->status->iters_done = 0;
  ctx->status->done = 0;
  ctx->status->epoch = ctx->epoch;

  CgSetup cfg{prm->Delta, prm->kappa_fgr, prm->theta, prm->epsilon,
              (unsigned long long)prm->max_iterations, ctx->epoch};
  hipStream_t st = ctx->stream;
  const int grid = grid_for(n, 8);
  double *slots = ctx->scalars + SLOT_CG;
  int ret = MI_OK;

#define CG_CHECK(expr)              \
  do {                              \
    int _s = (expr);                \
    if (_s != MI_OK) {              \
      ret = _s;                     \
      goto cleanup;                 \
    }                               \
  } while (0)

  // --- initialisation -----------------------------------------------------------------------
  {
    KScope ks(ctx, MI_K_CG_INIT);
    switch (pre) {
      case PRE_NONE:
        hipLaunchKernelGGL(k_cg_init<PRE_NONE>, dim3(grid), dim3(kBlock), 0, st, n, g->d, pred, r->d,
                           vd, p->d, s_out->d, ctx->partials);
        break;
      case PRE_DIAG:
        hipLaunchKernelGGL(k_cg_init<PRE_DIAG>, dim3(grid), dim3(kBlock), 0, st, n, g->d, pred, r->d,
                           vd, p->d, s_out->d, ctx->partials);
        break;
      case PRE_BLOCK3:
        hipLaunchKernelGGL(k_cg_init<PRE_BLOCK3>, dim3(grid), dim3(kBlock), 0, st, n, g->d, pred, r->d,
                           vd, p->d, s_out->d, ctx->partials);
        break;
      default:
        hipLaunchKernelGGL(k_cg_init<PRE_EXTERNAL>, dim3(grid), dim3(kBlock), 0, st, n, g->d, pred,
                           r->d, vd, p->d, s_out->d, ctx->partials);
        break;
    }
  }
  if (pre == PRE_EXTERNAL) {
    CG_CHECK(P->apply(P, r, v));
    hipLaunchKernelGGL(k_cg_dot_rv<true>, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg, r->d, v->d,
                       p->d, ctx->partials);
  }
  if (sharded) {
    hipLaunchKernelGGL(k_reduce_partials_to_slots, dim3(1), dim3(kBlock), 0, st, ctx->partials, grid, 1,
                       slots);
    CG_CHECK(comm_allreduce(ctx, slots, 1));
    hipLaunchKernelGGL(k_cg_scalar_init<true>, dim3(1), dim3(kBlock), 0, st, ctx->cg, cfg,
                       ctx->partials, grid, slots, ctx->status_dev);
  } else {
    hipLaunchKernelGGL(k_cg_scalar_init<false>, dim3(1), dim3(kBlock), 0, st, ctx->cg, cfg,
                       ctx->partials, grid, slots, ctx->status_dev);
  }
  if (hipGetLastError() != hipSuccess) CG_CHECK(hip_fail(hipErrorLaunchFailure, "stpcg init", __FILE__, __LINE__));

  // --- main loop: speculative enqueue with bounded run-ahead -----------------------------------
  {
    size_t hvp = 0;
    for (size_t k = 0; k < prm->max_iterations; ++k) {
      if (ctx->status->done) break;
      while (!ctx->status->done && k > ctx->status->iters_done + (uint64_t)run_ahead) cpu_relax();
      if (ctx->status->done) break;

      // Hp = H(p) (:294) + partials of <p,Hp>, <Hp,Hp>, <p,p>
      int nparts = 0;
      if (H->apply_dots) {
        CG_CHECK(H->apply_dots(H, p, Hp, &nparts));
      } else {
        CG_CHECK(H->apply(H, p, Hp));
        CG_CHECK(launch_dot3_partials(ctx, n, p->d, Hp->d, &nparts));
      }
      ++hvp;
      {
        KScope ks(ctx, MI_K_CG_SCALAR_A);
        if (sharded) {
          hipLaunchKernelGGL(k_reduce_partials_to_slots, dim3(1), dim3(kBlock), 0, st, ctx->partials,
                             nparts, 3, slots);
          CG_CHECK(comm_allreduce(ctx, slots, 3));
          hipLaunchKernelGGL(k_cg_scalar_a<true>, dim3(1), dim3(kBlock), 0, st, ctx->cg, ctx->partials,
                             nparts, slots, ctx->status_dev);
        } else {
          hipLaunchKernelGGL(k_cg_scalar_a<false>, dim3(1), dim3(kBlock), 0, st, ctx->cg, ctx->partials,
                             nparts, slots, ctx->status_dev);
        }
      }
      {
        KScope ks(ctx, MI_K_CG_UPDATE);
        switch (pre) {
          case PRE_NONE:
            hipLaunchKernelGGL(k_cg_update<PRE_NONE>, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg, p->d,
                               Hp->d, pred, s_out->d, r->d, vd, ctx->partials);
            break;
          case PRE_DIAG:
            hipLaunchKernelGGL(k_cg_update<PRE_DIAG>, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg, p->d,
                               Hp->d, pred, s_out->d, r->d, vd, ctx->partials);
            break;
          case PRE_BLOCK3:
            hipLaunchKernelGGL(k_cg_update<PRE_BLOCK3>, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg,
                               p->d, Hp->d, pred, s_out->d, r->d, vd, ctx->partials);
            break;
          default:
            hipLaunchKernelGGL(k_cg_update<PRE_EXTERNAL>, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg,
                               p->d, Hp->d, pred, s_out->d, r->d, vd, ctx->partials);
            break;
        }
      }
      if (pre == PRE_EXTERNAL) {
        // v = P(r) (:386) then <r,v>; in the (rare) non-RUN modes the update kernel has already
        // written the partial it needs and k_cg_dot_rv<false> leaves it untouched.
        CG_CHECK(P->apply(P, r, v));
        hipLaunchKernelGGL(k_cg_dot_rv<false>, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg, r->d, v->d,
                           p->d, ctx->partials);
      }
      {
        KScope ks(ctx, MI_K_CG_SCALAR_B);
        if (sharded) {
          hipLaunchKernelGGL(k_reduce_partials_to_slots, dim3(1), dim3(kBlock), 0, st, ctx->partials,
                             grid, 1, slots);
          CG_CHECK(comm_allreduce(ctx, slots, 1));
          hipLaunchKernelGGL(k_cg_scalar_b<true>, dim3(1), dim3(kBlock), 0, st, ctx->cg, prm->Delta,
                             ctx->partials, grid, slots, ctx->status_dev, tcap ? ctx->trace_dev : nullptr,
                             tcap);
        } else {
          hipLaunchKernelGGL(k_cg_scalar_b<false>, dim3(1), dim3(kBlock), 0, st, ctx->cg, prm->Delta,
                             ctx->partials, grid, slots, ctx->status_dev, tcap ? ctx->trace_dev : nullptr,
                             tcap);
        }
      }
      {
        KScope ks(ctx, MI_K_CG_PUPDATE);
        hipLaunchKernelGGL(k_cg_pupdate, dim3(grid), dim3(kBlock), 0, st, n, ctx->cg, vd, p->d,
                           s_out->d);
      }
    }
    result->hvp_calls = hvp;
  }

  // --- read back the final state ----------------------------------------------------------------
  {
    hipError_t e = hipMemcpyAsync(ctx->cg_host, ctx->cg, sizeof(CgState), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) CG_CHECK(hip_fail(e, "stpcg read-back", __FILE__, __LINE__));
    const CgState &f = *ctx->cg_host;
    result->update_step_M_norm = f.M_norm;
    result->num_iterations = (size_t)f.k;
    result->exit_reason = f.exit_reason;
    result->rv_final = f.rv;
    if (trace && trace->cap) {
      const size_t len = f.k < trace->cap ? (size_t)f.k : trace->cap;
      trace->len = len;
      double *dst[4] = {trace->alpha, trace->beta, trace->kappa, trace->rv};
      for (int c = 0; c < 4; ++c)
        if (dst[c] && len) {
          e = hipMemcpy(dst[c], ctx->trace_dev + (size_t)c * tcap, len * sizeof(double),
                        hipMemcpyDeviceToHost);
          if (e != hipSuccess) CG_CHECK(hip_fail(e, "stpcg trace copy", __FILE__, __LINE__));
        }
    }
  }

cleanup:
#undef CG_CHECK
  if (ret != MI_OK) (void)hipStreamSynchronize(st);
  mi_vec_destroy(r);
  mi_vec_destroy(p);
  mi_vec_destroy(Hp);
  mi_vec_destroy(v);
  return ret;
}

}  // extern "C"
