// stpcg.hip -- fused device realisation of Optimization::LinearAlgebra::STPCG
// (reference: LinearAlgebra/IterativeSolvers.h:166-426, unconstrained form).
//
// Design (MI355X-first):
//   * every vector phase between two global reductions is ONE streaming kernel (16 B/lane loads,
//     <= 512 fat workgroups, grid-stride) that ends by writing one partial-sum row per workgroup;
//   * there are NO scalar kernels in the loop: the next streaming kernel's workgroups each re-reduce
//     those <= 512 rows in their prologue (identical code on identical data => bit-identical totals
//     in every workgroup, deterministic, no fp64 atomics) and each advance the reference's scalar
//     recurrences (:259-283,330-345,412-417) and take its branch decisions themselves; workgroup 0
//     persists the new state.  The state is double-buffered (read st_in, write st_out) so no
//     workgroup can observe a half-updated state;
//   * the host never reads a scalar back inside the loop: it enqueues iterations speculatively,
//     bounded by `run_ahead`, and watches a pinned progress word; kernels enqueued past the exit see
//     mode == CG_DONE and return immediately;
//   * row-sharded multi-GPU: partial rows -> one-workgroup reduce -> in-stream RCCL all-reduce ->
//     the consumers read the all-reduced slots instead (the recurrences are replicated; every rank
//     takes identical decisions because ncclAllReduce returns identical bits everywhere).
//
// Per completed iteration the CG part launches 2 kernels:
//   k_cg_update   A-step prologue (kappa, kernel test, alpha, boundary test :300-362) +
//                 s += alpha p, r += alpha Hp, v = M^-1 r, partial <r,v> (:374-408)      [6N (+2N/+4N)]
//   k_cg_pupdate  B-step prologue (beta, M-norm recurrences, loop control :408-417,285,290) +
//                 p = -v + beta p (:420)                                                [3N]
// and the operator supplies <p,Hp>, <Hp,Hp>, <p,p> partials from its last pass (else k_cg_dot3, 2N).
#include "comm_ipc.h"
#include "mi_internal.h"
#include "stiefel_core.h"

#include <cmath>
#include <type_traits>

using namespace mi;

namespace {

enum { PRE_NONE = 0, PRE_DIAG = 1, PRE_BLOCK3 = 2, PRE_EXTERNAL = 3 };

__device__ __forceinline__ void publish(HostStatus *hs, unsigned long long launches, int done) {
  // one relaxed system-scope store: the host only looks at this word (results are read after a stream
  // synchronisation), so no release/write-back of the L2 is needed here
  __hip_atomic_store(&hs->word, (uint64_t)(launches << 1) | (uint64_t)(done ? 1 : 0), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}

// Field-by-field state copies.  An aggregate copy (`*dst = cs`) makes the compiler keep the
// per-thread struct in memory (promoted to 56 B/thread of LDS + scratch: +7-10 us per kernel,
// tools/microbench/prologue.hip); explicit fields let SROA keep everything in registers.
#define CG_FIELDS(X)                                                                               \
  X(sk_M_pk) X(sk_M_2) X(pk_M_2) X(target_rk_norm) X(rv) X(alpha) X(beta)                          \
  X(kappa) X(sigma) X(skplus1_M_2) X(M_norm) X(k) X(launches) X(mode) X(exit_reason)
__device__ __forceinline__ CgState load_state(const CgState *__restrict__ src) {
  CgState s;
#define X(f) s.f = src->f;
  CG_FIELDS(X)
#undef X
  return s;
}
__device__ __forceinline__ void store_state(CgState *__restrict__ dst, const CgState &s) {
#define X(f) dst->f = s.f;
  CG_FIELDS(X)
#undef X
}

// ---------------------------------------------------------------------------------------------
// scalar steps.  fp contraction is off so the recurrences round exactly like the reference's
// scalar C++ on the host.
// ---------------------------------------------------------------------------------------------

// after the operator: kappa, kernel test, alpha, boundary test (:300-362)
__device__ __forceinline__ void step_a(CgState &s, const CgConst &c, double pHp, double HpHp, double pp) {
#pragma clang fp contract(off)
  const double kappa = pHp;  // :300
  s.kappa = kappa;
  if (sqrt(HpHp) / sqrt(pp) < c.epsilon) {  // :305-307
    s.mode = CG_KERNEL_PENDING;              // needs <p,r> (:320): supplied by k_cg_update's body
    return;
  }
  const double alpha = s.rv / kappa;                                               // :341
  const double skplus1 = s.sk_M_2 + 2 * alpha * s.sk_M_pk + alpha * alpha * s.pk_M_2;  // :344-345
  if ((kappa <= 0) || (skplus1 > c.Delta_2)) {                                     // :347
    s.sigma = (-s.sk_M_pk + sqrt(s.sk_M_pk * s.sk_M_pk + s.pk_M_2 * (c.Delta_2 - s.sk_M_2))) /
              s.pk_M_2;  // :355-357
    s.mode = CG_APPLY_SIGMA;
    s.exit_reason = MI_STPCG_EXIT_BOUNDARY;
    return;
  }
  s.alpha = alpha;
  s.skplus1_M_2 = skplus1;
}

// after the update: resolves every mode; returns true if a direction update p = -v + beta p follows
__device__ __forceinline__ void step_b(CgState &s, const CgConst &c, double red) {
#pragma clang fp contract(off)
  if (s.mode == CG_APPLY_SIGMA) {  // boundary step applied by k_cg_update (:359-361)
    s.M_norm = c.Delta;
    s.mode = CG_DONE;
    return;
  }
  if (s.mode == CG_KERNEL_PENDING) {  // :320-337, red = <p,r>
    double sk_M_pk = s.sk_M_pk;
    const bool flip = red < 0;
    if (flip) sk_M_pk *= -1;  // :325
    const double sigma =
        (-sk_M_pk + sqrt(sk_M_pk * sk_M_pk + s.pk_M_2 * (c.Delta_2 - s.sk_M_2))) / s.pk_M_2;  // :330
    s.sk_M_pk = sk_M_pk;
    s.sigma = flip ? -sigma : sigma;  // s += sigma * (-p)  ==  s += (-sigma) * p   (:324,336)
    s.M_norm = c.Delta;               // :334
    s.exit_reason = MI_STPCG_EXIT_KERNEL;
    s.mode = CG_DONE;                 // k_cg_pupdate's body applies the step
    return;
  }
  // CG_RUN, red = <r,v> after the update
  const double rk_vk = red;                                  // :408
  const double beta = rk_vk / (s.alpha * s.kappa);           // :412
  s.sk_M_2 = s.skplus1_M_2;                                  // :415
  s.sk_M_pk = beta * (s.sk_M_pk + s.alpha * s.pk_M_2);       // :416
  s.pk_M_2 = rk_vk + beta * beta * s.pk_M_2;                 // :417
  s.rv = rk_vk;
  s.beta = beta;
  s.k = s.k + 1;
  if (s.k >= c.max_iterations) {  // :285
    s.exit_reason = MI_STPCG_EXIT_MAXIT;
    s.M_norm = sqrt(s.sk_M_2);  // :424
    s.mode = CG_DONE;
  } else if (sqrt(rk_vk) <= s.target_rk_norm) {  // :290 (top of the next pass)
    s.exit_reason = MI_STPCG_EXIT_RESIDUAL;
    s.M_norm = sqrt(s.sk_M_2);
    s.mode = CG_DONE;
  }
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// r'-halo form of the sharded solve (comm_ipc.h comm_rprime_*): halo(p') = -halo(r') + beta halo(p), the expression
// k_cg_pupdate evaluates for the rows it owns (:420) -- same TU, same contraction, same bits -- applied by every rank to
// the halo rows it holds.  `st` is the state k_cg_pupdate left: a direction update happened iff it still says CG_RUN.
__global__ __launch_bounds__(256) void k_halo_dir(const CgState *__restrict__ st, const double *__restrict__ hr,
                                                  double *__restrict__ hp, size_t count) {
  if (st->mode != CG_RUN) return;
  const double beta = st->beta;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    hp[i] = -hr[i] + beta * hp[i];
}


// r = g; s = 0*g (:211,214); v = P r (:231/234); p = -v (:256); partial <r,v> (:266)
template <int PRE>
__global__ __launch_bounds__(kBlock) void k_cg_init(size_t n, const double *__restrict__ g,
                                                    const double *__restrict__ pre,
                                                    double *__restrict__ r, double *__restrict__ v,
                                                    double *__restrict__ p, double *__restrict__ s,
                                                    double *__restrict__ partials) {
  __shared__ double lds[kWaves];
  double acc[1] = {0};
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
      acc[0] += g0 * v0; acc[0] += g1 * v1; acc[0] += g2 * v2;
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
      acc[0] += gv.x * vv.x; acc[0] += gv.y * vv.y;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const size_t i = n - 1;
      const double gv = g[i];
      r[i] = gv; s[i] = 0 * gv;
      if (PRE != PRE_EXTERNAL) {
        double vv = gv;
        if (PRE == PRE_DIAG) { vv = pre[i] * gv; v[i] = vv; }
        p[i] = -vv;
        acc[0] += gv * vv;
      }
    }
  }
  if (PRE == PRE_EXTERNAL) return;
  block_partials_store<1>(acc, lds, partials);
}

// external preconditioner path: partial <r,v>; NEG_P: also p = -v (initialisation, :256)
template <bool NEG_P>
__global__ __launch_bounds__(kBlock) void k_cg_dot_rv(size_t n, const CgState *__restrict__ st,
                                                      const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ p,
                                                      double *__restrict__ partials) {
  __shared__ double lds[kWaves];
  if (!NEG_P && st->mode != CG_RUN) return;  // the update kernel already left the partial it needs
  double acc[1] = {0};
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 rv = reinterpret_cast<const double2 *>(r)[i];
    const double2 vv = reinterpret_cast<const double2 *>(v)[i];
    if (NEG_P) reinterpret_cast<double2 *>(p)[i] = make_double2(-vv.x, -vv.y);
    acc[0] += rv.x * vv.x; acc[0] += rv.y * vv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    if (NEG_P) p[n - 1] = -v[n - 1];
    acc[0] += r[n - 1] * v[n - 1];
  }
  block_partials_store<1>(acc, lds, partials);
}

// partials of <x,y>, <y,y>, <x,x>  (kappa :300, kernel test :305-306) for operators without fused dots
__global__ __launch_bounds__(kBlock) void k_cg_dot3(size_t n, const CgState *__restrict__ st,
                                                    const double *__restrict__ x,
                                                    const double *__restrict__ y,
                                                    double *__restrict__ partials) {
  __shared__ double lds[3 * kWaves];
  if (st && st->mode != CG_RUN) return;
  double a[3] = {0, 0, 0};
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 xv = reinterpret_cast<const double2 *>(x)[i];
    const double2 yv = reinterpret_cast<const double2 *>(y)[i];
    a[0] += xv.x * yv.x; a[0] += xv.y * yv.y;
    a[1] += yv.x * yv.x; a[1] += yv.y * yv.y;
    a[2] += xv.x * xv.x; a[2] += xv.y * xv.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double xv = x[n - 1], yv = y[n - 1];
    a[0] += xv * yv; a[1] += yv * yv; a[2] += xv * xv;
  }
  block_partials_store<3>(a, lds, partials);
}

// G(p0) (packed symmetric, gns components, wave w sums the `count` partial rows of components w, w + 16, ... in fixed order)
// into gdir[SLOT_GDIR_P..), and G(r0) = -G(p0) (p0 = -r0) into gdir[0..)
__device__ __forceinline__ void gdir_from_rows(const double *__restrict__ partials, int count, int gns,
                                               double *__restrict__ gdir) {
  const int lane = threadIdx.x & 63;
  for (int w = threadIdx.x >> 6; w < gns; w += kWaves) {  // (gns > kWaves: p >= 6, in rounds)
    const double *src = partials + (size_t)w * kMaxRows;
    double t[kMaxRows / 64];
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) {
      const int r = lane + 64 * j;
      t[j] = (r < count) ? src[r] : 0.0;
    }
    double v = 0;
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) v += t[j];
    v = wave_reduce_sum(v);
    if (lane == 0) {
      gdir[SLOT_GDIR_P + w] = v;
      gdir[w] = -v;
    }
  }
}

// the same fixed-order reduction into ONE destination (re-anchoring, r06)
__device__ __forceinline__ void gdir_rows_to(const double *__restrict__ partials, int count, int gns,
                                             double *__restrict__ dst) {
  const int lane = threadIdx.x & 63;
  for (int w = threadIdx.x >> 6; w < gns; w += kWaves) {
    const double *src = partials + (size_t)w * kMaxRows;
    double t[kMaxRows / 64];
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) {
      const int r = lane + 64 * j;
      t[j] = (r < count) ? src[r] : 0.0;
    }
    double v = 0;
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) v += t[j];
    v = wave_reduce_sum(v);
    if (lane == 0) dst[w] = v;
  }
}

struct CgSetup {
  double Delta, kappa_fgr, theta, epsilon;
  unsigned long long max_iterations;
};

// one workgroup, once per solve: :259-279 and the first pass of :285-290
// gdir != nullptr: k_cg_gdir_init's work rides along (one rank, recurrence form: its kernel has nothing to wait for):
// G(p0) from the `gcount` Gram rows k_cg_init_dirgram left, G(r0) = -G(p0)
template <bool FROM_SLOTS>
__global__ __launch_bounds__(kBlock) void k_cg_scalar_init(CgState *st, CgSetup cfg,
                                                           const double *partials, int nparts,
                                                           const double *slots, HostStatus *hs,
                                                           const double *__restrict__ gpartials, int gcount, int gns,
                                                           double *__restrict__ gdir) {
#pragma clang fp contract(off)
  __shared__ double lds[kWaves + 1];
  if (gdir) gdir_from_rows(gpartials, gcount, gns, gdir);
  double red[1];
  if (FROM_SLOTS) red[0] = slots[0];
  else reduce_rows<1>(partials, nparts, red, lds);
  if (threadIdx.x != 0) return;
  CgState s;
  const double rv0 = red[0];
  s.sk_M_pk = 0;                        // :259
  s.sk_M_2 = 0;                         // :263
  s.pk_M_2 = rv0;                       // :266
  const double r0_norm = sqrt(rv0);     // :275
  const double pw = pow(r0_norm, cfg.theta);
  s.target_rk_norm = r0_norm * ((pw < cfg.kappa_fgr) ? pw : cfg.kappa_fgr);  // :278-279 (std::min)
  s.rv = rv0;
  s.alpha = s.beta = s.kappa = s.sigma = s.skplus1_M_2 = 0;
  s.M_norm = 0;
  s.k = 0;
  s.launches = 0;
  s.exit_reason = MI_STPCG_EXIT_MAXIT;
  s.mode = CG_RUN;
  if (cfg.max_iterations == 0) {  // :285 body never runs
    s.mode = CG_DONE;
    s.M_norm = sqrt(0.0);  // :424
  } else if (sqrt(rv0) <= s.target_rk_norm) {  // :290
    s.mode = CG_DONE;
    s.exit_reason = MI_STPCG_EXIT_RESIDUAL;
    s.M_norm = sqrt(0.0);
  }
  store_state(st, s);
  publish(hs, 0, s.mode == CG_DONE);
}

// mi_op::dirgram plumbing.  Direct form: k_cg_pupdate<.,SP> forms the Gram rows of the new direction from X, Y.
// Recurrence form (gdir != null, no preconditioner): the operator pass leaves the rows of G(Hp) as components
// 3.. of its partial row; workgroup 0 of k_cg_update advances G(r) += alpha G(Hp) and workgroup 0 of
// k_cg_pupdate G(p) = -G(r) + beta G(p) -- packed symmetric, ns = P(P+1)/2 doubles each at gdir[0..) / gdir[SLOT_GDIR_P..).
struct DirGramArgs {
  const double *X, *Y, *S;
  double *gpartials;
  double *gdir;
  int ns;
};

// A-step prologue + body:
//   CG_RUN:            r += alpha Hp (:377); v = P r (:383/386); partial <r,v> (:408)   [s += alpha p: see k_cg_pupdate]
//   CG_APPLY_SIGMA:    s += sigma p (:360)
//   CG_KERNEL_PENDING: partial <p,r> (:320)
// (the kernels follow the DirGramLds helper below: stpcg_kernels.inc)
#ifdef MI_FOLD_STAMPS
}  // namespace
extern "C" __attribute__((visibility("default"))) int mi_debug_fold_stamp_buffer(void *dev_ptr) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mi::g_fold_stamps), &dev_ptr, sizeof(dev_ptr));
}
namespace {
#endif

// B-step prologue + body:  CG_RUN: s = s + alpha p (:374), p = -v + beta p (:420);  kernel exit: s += sigma p (:336)
//   SP > 0 (mi_op::dirgram): the fields are rows of SP doubles and the kernel also leaves the partial rows of
//   sym(Y'p - (X'p) S) of the NEW direction in dg.gpartials -- the projection matrix of the next Hessian pass.
template <int SP>
struct DirGramLds {
  static constexpr int value = (SymIdx<SP>::NS * kWaves > kWaves + 1) ? SymIdx<SP>::NS * kWaves : kWaves + 1;
};
template <>
struct DirGramLds<0> {
  static constexpr int value = kWaves + 1;  // (>= kIpcVals: the folded exchange's staging)
};
static_assert(kWaves + 1 >= kIpcVals, "LDS of the folded exchange");

#define CG_KERNEL_ATTR
#define CG_UPDATE_NAME k_cg_update
#define CG_PUPDATE_NAME k_cg_pupdate
#include "stpcg_kernels.inc"
#undef CG_KERNEL_ATTR
#undef CG_UPDATE_NAME
#undef CG_PUPDATE_NAME
// The same two kernels with at most 80 scalar registers.  Measured on gfx950 (tools/fold_stamps.py, in-kernel clock
// stamps): a kernel that is allocated 96 SGPRs gets 7 waves per SIMD, not the 8 the compiler reports -- and these
// kernels run 1024-thread workgroups, two per CU, which needs all 8: with 81..96 SGPRs the second half of the grid only
// starts when the first half has finished (k_cg_update: +3.7 us of 15).  The single-GPU instantiations of the hot loop
// sit at exactly 80; the ones that carry the exchange arguments (FoldArgs, FoldPush), read reduced slots, or keep 16
// components need 82..100 and are launched from this pair instead: a few scalar spills to vector lanes.
#define CG_KERNEL_ATTR __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8)))
#define CG_UPDATE_NAME k_cg_update_s80
#define CG_PUPDATE_NAME k_cg_pupdate_s80
#include "stpcg_kernels.inc"
#undef CG_KERNEL_ATTR
#undef CG_UPDATE_NAME
#undef CG_PUPDATE_NAME

// EARLY S (late r06): the direction kernel of the single-context solve without a preconditioner (v = r, plain vectors,
// G(p) by recurrence).  alpha is the A-step's -- it is in the state this kernel STARTS from -- so s += alpha p (:374)
// does not depend on this kernel's own reduction; only beta does.  Where a thread's whole walk is at most three
// elements (cfg2: two grid-stride steps + the ragged one; the host checks) every operand of the kernel is requested at
// once, behind the reduction's row loads; s is updated and stored while the other operands are still on their way;
// what is left behind the barrier is p = -v + beta p from registers.  The read phase no longer waits for the
// prologue and the prologue no longer idles the memory system (tools/microbench/stream_rates.hip: the bare stream of
// this byte mix takes 15.8 us per launch, k_cg_pupdate 18.1).  Same walk, same expressions in the same translation
// unit (same contraction), same bits as k_cg_pupdate<false, 0, NoFold>; every other form keeps that kernel.
// MEASURED (same box, alternating, rocprofv3): 17.95 us against k_cg_pupdate's 18.10, the step 55.8 against 56.1 us -- and
// with the reduction and its barrier compiled out altogether (timing build, wrong results) still 20.1-20.3 us by event
// pairs against 20.0-20.1: the prologue is NOT what separates this kernel from the bare stream (EXPERIMENTS.md r06).
// Opt-in (MI355OPT_EARLY_S=1): not the default for 0.3 us.
template <class F>
__device__ __forceinline__ void reduce_rows_1x512(const double *__restrict__ partials, int count, double &out, double *lds,
                                                  F &&after_issue) {
  // reduce_rows<1> for count <= 512: the same lane -> rows map and the same order of additions (its further eight
  // terms per lane are zeros there)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = 0.0;
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = lane + 64 * j;
      if (r < count) t[j] = partials[r];
    }
  }
  after_issue();
  if (w == 0) {
    double v = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) v += t[j];
    v = wave_reduce_sum(v);
    if (lane == 0) lds[0] = v;
  }
  __syncthreads();
  out = lds[0];
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_cg_pupdate_early(
    size_t n, CgConst cc, const CgState *__restrict__ st_in, CgState *__restrict__ st_out,
    const double *__restrict__ partials_b, int nparts_b, const double *__restrict__ v, double *__restrict__ p,
    double *__restrict__ s, HostStatus *hs, double *__restrict__ trace, size_t trace_cap, DirGramArgs dg) {
  __shared__ double lds[kWaves + 1];
  // (the walk of k_cg_pupdate, stpcg_kernels.inc, in 32-bit element indices: the host launches this kernel only where
  // n / 2 <= 3 x 512 x 1024)
  const unsigned n2 = (unsigned)(n >> 1), stride = gridDim.x * kBlock, i0 = blockIdx.x * kBlock + threadIdx.x;
  const unsigned tail0 = (unsigned)cc.rag0, per = (unsigned)cc.rag_per;
  const unsigned irem = (threadIdx.x < per) ? tail0 + blockIdx.x * per + threadIdx.x : n2;
  auto next_of = [&](unsigned i_) -> unsigned {
    const unsigned j_ = i_ + stride;
    return j_ < tail0 ? j_ : ((i_ < tail0 && irem < n2) ? irem : n2);
  };
  const unsigned ifirst = i0 < tail0 ? i0 : (irem < n2 ? irem : n2);
  unsigned Le[3];
  {
    unsigned L = ifirst;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Le[j] = L;
      if (L < n2) L = next_of(L);
    }
  }
  double2 pe[3], ve[3], se[3];
  CgState cs = load_state(st_in);
  const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
  if (cs.mode == CG_DONE) {
    if (leader) store_state(st_out, cs);
    return;
  }
  const int mode_in = cs.mode;
  auto early_s = [&] {
    if (mode_in != CG_RUN) return;
    const double alpha = cs.alpha;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (Le[j] < n2) {
        pe[j] = reinterpret_cast<double2 *>(p)[Le[j]];
        se[j] = reinterpret_cast<double2 *>(s)[Le[j]];
      }
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (Le[j] < n2) ve[j] = reinterpret_cast<const double2 *>(v)[Le[j]];
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (Le[j] < n2) {
        double2 sv = se[j];
        const double2 pv = pe[j];
        sv.x = sv.x + alpha * pv.x; sv.y = sv.y + alpha * pv.y;
        reinterpret_cast<double2 *>(s)[Le[j]] = sv;
      }
  };
  double red = 0;
  if (mode_in != CG_APPLY_SIGMA) reduce_rows_1x512(partials_b, nparts_b, red, lds, early_s);
  step_b(cs, cc, red);
  cs.launches = cs.launches + 1;
  if (leader) {
    store_state(st_out, cs);
    if (mode_in == CG_RUN && trace && cs.k - 1 < trace_cap) {
      const size_t k = (size_t)(cs.k - 1);
      trace[k] = cs.alpha;
      trace[trace_cap + k] = cs.beta;
      trace[2 * trace_cap + k] = cs.kappa;
      trace[3 * trace_cap + k] = cs.rv;
    }
    publish(hs, cs.launches, cs.mode == CG_DONE);
    if (dg.gdir && mode_in == CG_RUN && cs.mode == CG_RUN) {  // G(p) = -G(r) + beta G(p)  (:420)
      for (int i = 0; i < dg.ns; ++i) dg.gdir[SLOT_GDIR_P + i] = -dg.gdir[i] + cs.beta * dg.gdir[SLOT_GDIR_P + i];
    }
  }
  if (mode_in == CG_KERNEL_PENDING) {
    const double sigma = cs.sigma;
    for (unsigned i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      double2 sv = reinterpret_cast<double2 *>(s)[i];
      sv.x += sigma * pv.x; sv.y += sigma * pv.y;
      reinterpret_cast<double2 *>(s)[i] = sv;
    }
    if ((n & 1) && leader) s[n - 1] += sigma * p[n - 1];
  } else if (mode_in == CG_RUN) {
    const double alpha = cs.alpha, beta = cs.beta;
    const bool dir = cs.mode == CG_RUN;
    if (dir) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (Le[j] < n2) {
          double2 pv = pe[j];
          const double2 vv = ve[j];
          pv.x = -vv.x + beta * pv.x; pv.y = -vv.y + beta * pv.y;
          reinterpret_cast<double2 *>(p)[Le[j]] = pv;
        }
    }
    if ((n & 1) && leader) {
      s[n - 1] = s[n - 1] + alpha * p[n - 1];
      if (dir) p[n - 1] = -v[n - 1] + beta * p[n - 1];
    }
  }
}


// ---- opt-in experiment (r05; Config::two_kernel_step) -------------------------------------------------
// ONE kernel for the A-step and the B-step of an unpreconditioned iteration over rows of Stiefel(n,3) (v == r).  Without
// a preconditioner <r+,r+> = <r,r> + 2 alpha <r,Hp> + alpha^2 <Hp,Hp> is known once the Hessian pass has also left
// <r,Hp> (k_st_hess_fused<..., TWOK>: it reads r, 8 N more bytes there), so r+ = r + alpha Hp (:377), s += alpha p (:374)
// and p+ = -r+ + beta p (:420) are one pass: reads r, Hp, p, s and writes r, p, s = 56 N bytes instead of 24 N + 40 N, and
// one kernel boundary fewer.  It CHANGES THE ROUNDING of :408 -- beta comes from a three-term recurrence that cancels
// as the residual falls, not from a sum over the new residual -- which is why it is opt-in and never the default; the
// distance from the reference it costs is measured in EXPERIMENTS.md.  The kernel test (:305-337) is served by a fifth
// extra partial, <p,r>, so every exit of STPCG is taken inside this kernel: an iteration is two launches.
template <int NS>
__global__ __launch_bounds__(kBlock) void k_cg_step2(size_t n, CgConst cc, const CgState *__restrict__ st_in,
                                                     CgState *__restrict__ st_out, const double *__restrict__ partials_a,
                                                     int nparts_a, const double *__restrict__ Hp, double *__restrict__ r,
                                                     double *__restrict__ p, double *__restrict__ s, HostStatus *hs,
                                                     double *__restrict__ trace, size_t trace_cap, DirGramArgs dg) {
  constexpr int KC = 3 + NS + 2;
  __shared__ double lds[3 * (kWaves + 1)];
  static_assert(KC <= kWaves, "one wave per component");
  CgState cs = load_state(st_in);
  const bool leader = blockIdx.x == 0 && threadIdx.x == 0;
  if (cs.mode == CG_DONE) {
    if (leader) store_state(st_out, cs);
    return;
  }
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock, i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  double2 hv0 = make_double2(0, 0), rv0 = hv0, pv0 = hv0, sv0 = hv0;
  auto prefetch = [&] {
    if (i0 < n2) {
      hv0 = reinterpret_cast<const double2 *>(Hp)[i0];
      rv0 = reinterpret_cast<double2 *>(r)[i0];
      pv0 = reinterpret_cast<double2 *>(p)[i0];
      sv0 = reinterpret_cast<double2 *>(s)[i0];
    }
  };
  double d[KC];
  reduce_rows<KC>(partials_a, nparts_a, d, lds, prefetch);
  step_a(cs, cc, d[0], d[1], d[2]);
  const int mode_a = cs.mode;
  double red = 0;
  {
#pragma clang fp contract(off)
    if (mode_a == CG_RUN) red = cs.rv + 2 * cs.alpha * d[KC - 2] + cs.alpha * cs.alpha * d[1];  // <r+,r+>  (:408)
    else if (mode_a == CG_KERNEL_PENDING) red = d[KC - 1];                                        // <p,r>   (:320)
  }
  step_b(cs, cc, red);
  cs.launches = cs.launches + 1;
  if (leader) {
    store_state(st_out, cs);
    if (mode_a == CG_RUN && trace && cs.k - 1 < trace_cap) {
      const size_t k = (size_t)(cs.k - 1);
      trace[k] = cs.alpha;
      trace[trace_cap + k] = cs.beta;
      trace[2 * trace_cap + k] = cs.kappa;
      trace[3 * trace_cap + k] = cs.rv;
    }
    publish(hs, cs.launches, cs.mode == CG_DONE);
    if (mode_a == CG_RUN) {
      for (int i = 0; i < NS; ++i) dg.gdir[i] = dg.gdir[i] + cs.alpha * d[3 + i];  // G(r) += alpha G(Hp)
      if (cs.mode == CG_RUN)
        for (int i = 0; i < NS; ++i) dg.gdir[SLOT_GDIR_P + i] = -dg.gdir[i] + cs.beta * dg.gdir[SLOT_GDIR_P + i];
    }
  }
  if (mode_a != CG_RUN) {  // boundary step (:359-361) or kernel step (:324-336): s += sigma p, and the solve is over
    const double sigma = cs.sigma;
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      double2 sv = reinterpret_cast<double2 *>(s)[i];
      sv.x += sigma * pv.x; sv.y += sigma * pv.y;
      reinterpret_cast<double2 *>(s)[i] = sv;
    }
    if ((n & 1) && leader) s[n - 1] += sigma * p[n - 1];
    return;
  }
  const double alpha = cs.alpha, beta = cs.beta;
  const bool dir = cs.mode == CG_RUN;
  double2 hv = hv0, rv = rv0, pv = pv0, sv = sv0;
  for (size_t i = i0; i < n2;) {
    const size_t inext = i + stride;
    double2 hn = hv, rn = rv, pn = pv, sn = sv;
    if (inext < n2) {
      hn = reinterpret_cast<const double2 *>(Hp)[inext];
      rn = reinterpret_cast<double2 *>(r)[inext];
      pn = reinterpret_cast<double2 *>(p)[inext];
      sn = reinterpret_cast<double2 *>(s)[inext];
    }
    rv.x += alpha * hv.x; rv.y += alpha * hv.y;
    reinterpret_cast<double2 *>(r)[i] = rv;
    sv.x = sv.x + alpha * pv.x; sv.y = sv.y + alpha * pv.y;
    reinterpret_cast<double2 *>(s)[i] = sv;
    if (dir) {
      pv.x = -rv.x + beta * pv.x; pv.y = -rv.y + beta * pv.y;
      reinterpret_cast<double2 *>(p)[i] = pv;
    }
    i = inext; hv = hn; rv = rn; pv = pn; sv = sn;
  }
  if ((n & 1) && leader) {
    const size_t i = n - 1;
    const double rn_ = r[i] + alpha * Hp[i];
    r[i] = rn_;
    s[i] = s[i] + alpha * p[i];
    if (dir) p[i] = -rn_ + beta * p[i];
  }
}

// partial rows of sym(Y'p - (X'p) S) for the FIRST direction (p = -v, :256); later ones come from k_cg_pupdate
// (rows wider than 4 doubles, r05: 256-thread workgroups and S in LDS -- stiefel_core.h StBlk; as 1024-thread
// instantiations these two kernels spilled 250 ... 350 bytes per lane at p = 8)
template <int SP>
__global__ __launch_bounds__(StBlk<SP>::threads) void k_cg_dirgram(size_t nrows, const double *__restrict__ p, DirGramArgs dg) {
  constexpr int BLK = StBlk<SP>::threads, NW = StBlk<SP>::waves;
  constexpr bool WIDE = SP > 4;
  __shared__ double lds[SymIdx<SP>::NS * NW];
  __shared__ double SmL[WIDE ? SP * SP : 1];
  double Sm[WIDE ? 1 : SP * SP], G[SP * SP];
#pragma unroll
  for (int i = 0; i < SP * SP; ++i) G[i] = 0;
  if (WIDE) {
    if (threadIdx.x < SP * SP) SmL[threadIdx.x] = dg.S[threadIdx.x];
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < SP * SP; ++i) Sm[WIDE ? 0 : i] = dg.S[i];
  }
  const size_t stride = (size_t)gridDim.x * BLK;
  for (size_t row = (size_t)blockIdx.x * BLK + threadIdx.x; row < nrows; row += stride) {
    double x[SP], y[SP], pv[SP];
#pragma unroll
    for (int c = 0; c < SP; ++c) { x[c] = dg.X[row * SP + c]; y[c] = dg.Y[row * SP + c]; pv[c] = p[row * SP + c]; }
    if (WIDE) asm volatile("" ::: "memory");
#pragma unroll
    for (int b = 0; b < SP; ++b) {
      double t = 0;
#pragma unroll
      for (int a = 0; a < SP; ++a) t += pv[a] * (WIDE ? SmL[a * SP + b] : Sm[WIDE ? 0 : a * SP + b]);
#pragma unroll
      for (int a = 0; a < SP; ++a) G[a * SP + b] += y[a] * pv[b] - x[a] * t;
    }
  }
  store_sym_partials<SP>(G, lds, dg.gpartials);
}

// the same for TWO vectors in one pass (re-anchoring, r06; rows of <= 4 doubles): partial rows of G(a) as components
// [0, NS) and of G(b) as [NS, 2 NS); X and Y are read once.  Row for row and thread for thread the sums of k_cg_dirgram.
template <int SP>
__global__ __launch_bounds__(kBlock) void k_cg_dirgram2(size_t nrows, const CgState *__restrict__ st,
                                                        const double *__restrict__ va, const double *__restrict__ vb,
                                                        DirGramArgs dg) {
  static_assert(SP >= 1 && SP <= 4, "narrow rows");
  __shared__ double lds[SymIdx<SP>::NS * kWaves];
  if (st->mode != CG_RUN) return;
  double Sm[SP * SP], Ga[SP * SP], Gb[SP * SP];
#pragma unroll
  for (int i = 0; i < SP * SP; ++i) { Ga[i] = 0; Gb[i] = 0; Sm[i] = dg.S[i]; }
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t row = (size_t)blockIdx.x * kBlock + threadIdx.x; row < nrows; row += stride) {
    double x[SP], y[SP], a[SP], b[SP];
#pragma unroll
    for (int c = 0; c < SP; ++c) {
      x[c] = dg.X[row * SP + c]; y[c] = dg.Y[row * SP + c]; a[c] = va[row * SP + c]; b[c] = vb[row * SP + c];
    }
#pragma unroll
    for (int j = 0; j < SP; ++j) {
      double ta = 0, tb = 0;
#pragma unroll
      for (int i = 0; i < SP; ++i) { ta += a[i] * Sm[i * SP + j]; tb += b[i] * Sm[i * SP + j]; }
#pragma unroll
      for (int i = 0; i < SP; ++i) {
        Ga[i * SP + j] += y[i] * a[j] - x[i] * ta;
        Gb[i * SP + j] += y[i] * b[j] - x[i] * tb;
      }
    }
  }
  store_sym_partials<SP>(Ga, lds, dg.gpartials);
  __syncthreads();
  store_sym_partials<SP>(Gb, lds, dg.gpartials + (size_t)SymIdx<SP>::NS * kMaxRows);
}

// k_cg_init<PRE_NONE> and k_cg_dirgram in one pass for the unpreconditioned solve over rows of SP doubles (the cfg2 /
// cfg4 hot path: v = r, p = -g): g is read once, p is not read back, one launch fewer per solve (16.4 + 14.3 us as two
// kernels at cfg2).  Row r of every field belongs to the thread k_cg_dirgram gives it, so the Gram rows have its bits;
// the partial rows of <r,v> group their terms by rows instead of by pairs of elements.
template <int SP>
__global__ __launch_bounds__(StBlk<SP>::threads) void k_cg_init_dirgram(size_t nrows, const double *__restrict__ g,
                                                            double *__restrict__ r, double *__restrict__ p,
                                                            double *__restrict__ s, double *__restrict__ partials,
                                                            DirGramArgs dg) {
  constexpr int BLK = StBlk<SP>::threads, NW = StBlk<SP>::waves;
  constexpr bool WIDE = SP > 4;
  __shared__ double lds[SymIdx<SP>::NS * NW];
  __shared__ double SmL[WIDE ? SP * SP : 1];
  double Sm[WIDE ? 1 : SP * SP], G[SP * SP];
#pragma unroll
  for (int i = 0; i < SP * SP; ++i) G[i] = 0;
  if (WIDE) {
    if (threadIdx.x < SP * SP) SmL[threadIdx.x] = dg.S[threadIdx.x];
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < SP * SP; ++i) Sm[WIDE ? 0 : i] = dg.S[i];
  }
  double acc[1] = {0};
  const size_t stride = (size_t)gridDim.x * BLK;
  for (size_t row = (size_t)blockIdx.x * BLK + threadIdx.x; row < nrows; row += stride) {
    double x[SP], y[SP], gv[SP], pv[SP];
#pragma unroll
    for (int c = 0; c < SP; ++c) { x[c] = dg.X[row * SP + c]; y[c] = dg.Y[row * SP + c]; gv[c] = g[row * SP + c]; }
#pragma unroll
    for (int c = 0; c < SP; ++c) {
      pv[c] = -gv[c];                      // p = -v, v = r = g  (:211,231,256)
      r[row * SP + c] = gv[c];
      s[row * SP + c] = 0 * gv[c];         // :214
      p[row * SP + c] = pv[c];
      acc[0] += gv[c] * gv[c];             // <r,v>  (:266)
    }
    if (WIDE) asm volatile("" ::: "memory");
#pragma unroll
    for (int b = 0; b < SP; ++b) {
      double t = 0;
#pragma unroll
      for (int a = 0; a < SP; ++a) t += pv[a] * (WIDE ? SmL[a * SP + b] : Sm[WIDE ? 0 : a * SP + b]);
#pragma unroll
      for (int a = 0; a < SP; ++a) G[a * SP + b] += y[a] * pv[b] - x[a] * t;
    }
  }
  store_sym_partials<SP>(G, lds, dg.gpartials);
  __syncthreads();
  block_partials_store_w<1, NW>(acc, lds, partials);
}

// recurrence form: G(p0) from the rows k_cg_dirgram left (or the all-reduced slots); G(r0) = -G(p0) (p0 = -r0)
__global__ __launch_bounds__(kBlock) void k_cg_gdir_init(const double *__restrict__ partials, int count, int ns,
                                                         const double *__restrict__ slots, int from_slots,
                                                         double *__restrict__ gdir) {
  if (!from_slots) {
    gdir_from_rows(partials, count, ns, gdir);
    return;
  }
  if (threadIdx.x < ns) {
    const double v = slots[threadIdx.x];
    gdir[SLOT_GDIR_P + threadIdx.x] = v;
    gdir[threadIdx.x] = -v;
  }
}

// Re-anchoring of the recurrence form (r06): dst = G(vector) from the rows k_cg_dirgram just left (or the all-reduced
// slots).  The recurrences G(r') = G(r) + alpha G(Hp), G(p') = -G(r') + beta G(p) carry an ABSOLUTE error of the size
// rounding had at the scale of the residual they started from; on identical inputs a 200-iteration solve ends 2.6e-13
// from the reference with them and 8e-15 with the two-pass operator, and follows the reference's alpha trace for 354
// instead of 395 iterations (tests/test_gpu_long_solves.py).  Every mi_ctx::cfg.reanchor iterations both are replaced
// by their direct values.  A solve that has left CG_RUN keeps what it has (its kernels do nothing any more).
// dst2 != nullptr: components [ns, 2 ns) go there (the two-vector pass k_cg_dirgram2)
__global__ __launch_bounds__(kBlock) void k_cg_gdir_anchor(const double *__restrict__ partials, int count, int ns,
                                                           const double *__restrict__ slots, int from_slots,
                                                           const CgState *__restrict__ st, double *__restrict__ dst,
                                                           double *__restrict__ dst2) {
  if (st->mode != CG_RUN) return;
  if (!from_slots) {
    gdir_rows_to(partials, count, ns, dst);
    if (dst2) gdir_rows_to(partials + (size_t)ns * kMaxRows, count, ns, dst2);
    return;
  }
  if (threadIdx.x < ns) dst[threadIdx.x] = slots[threadIdx.x];
  if (dst2 && threadIdx.x < ns) dst2[threadIdx.x] = slots[ns + threadIdx.x];
}

inline void cpu_relax() { __builtin_ia32_pause(); }

}  // namespace

namespace mi {
int launch_dot3_partials(mi_ctx *ctx, size_t n, const double *x, const double *y, int *nparts) {
  const int grid = grid_for(ctx, n, 4);
  KScope ks(ctx, MI_K_CG_DOT3);
  hipLaunchKernelGGL(k_cg_dot3, dim3(grid), dim3(kBlock), 0, ctx->stream, n, ctx->cg_live, x, y,
                     ctx->partials);
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
  p->run_ahead = 0;  // 0: the library default (3, or MI355OPT_RUN_AHEAD)
  p->constraint_At = 0;
  p->defer_result = 0;
}

// the sticky failure word of the solve's preconditioner (mi_precon::fail_word) travels right behind the state copy
static inline double *precon_fail_host(mi_ctx *ctx) { return reinterpret_cast<double *>(ctx->cg_host + 1); }
// Code 1 (the preconditioner's inner iteration BROKE DOWN: p'Sp <= 0, NaN / Inf) is an error of the solve: its result is
// not a projected step.  Code 2 (the inner iteration stopped at its iteration limit short of its tolerance) is NOT: a
// caller that caps inner_max_iterations to bound the cost asks for an inexact projection -- the solve
// returns MI_OK, mi_stpcg_result::precon_status says 2, and mi_precon_constraint_info has the residual that was left.
static int precon_fail_check(mi_ctx *ctx, mi_stpcg_result *result) {
  const double w = *precon_fail_host(ctx);
  if (result) result->precon_status = (int)w;
  if (w != 1.0) return MI_OK;
  set_error("the preconditioner of this solve reported a failed application (code 1: its inner iteration broke down -- "
            "dependent constraint rows, or NaN / Inf in the residual): the result is not a projected step");
  return MI_ERR_INTERNAL;
}

// The final state of a solve lands in the pinned words by a kernel store, the polled sequence number behind it
// (context.hip: stream_wait): no copy engine and no wake-up between a solve and its caller.
__global__ void k_cg_result_to_host(unsigned long long *host, const unsigned long long *__restrict__ st,
                                    const double *__restrict__ fail_dev, unsigned long long *flag,
                                    unsigned long long seq) {
  constexpr int kWords = (int)(sizeof(CgState) / 8);
  static_assert(sizeof(CgState) % 8 == 0, "CgState is copied in 8-byte words");
  if (threadIdx.x < kWords) host[threadIdx.x] = st[threadIdx.x];
  if (threadIdx.x == kWords) reinterpret_cast<double *>(host)[kWords] = fail_dev ? *fail_dev : 0.0;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// enqueue that kernel; returns the sequence number to poll for, 0 when polling is off (the caller copies instead)
static unsigned long long result_to_host(mi_ctx *ctx, const CgState *st_final, const mi_precon *P) {
  unsigned long long *flag = nullptr, *hdev = nullptr;
  unsigned long long seq = poll_begin(ctx, &flag);
  if (!seq) return 0;
  if (hipHostGetDevicePointer((void **)&hdev, ctx->cg_host, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;  // (the sequence number stays unused: the next one is larger)
  }
  hipLaunchKernelGGL(k_cg_result_to_host, dim3(1), dim3(64), 0, ctx->stream, hdev,
                     reinterpret_cast<const unsigned long long *>(st_final),
                     (const double *)(P ? P->fail_word : nullptr), flag, seq);
  if (hipGetLastError() != hipSuccess) return 0;  // (nothing was enqueued: the caller copies)
  return seq;
}

int mi_stpcg_collect(mi_ctx *ctx, mi_stpcg_result *result) {
  MI_REQUIRE(ctx && result, "null argument");
  MI_REQUIRE(ctx->cg_deferred, "mi_stpcg_collect: no deferred solve is pending on this context");
  ctx->cg_deferred = false;
  if (ctx->cg_deferred_seq) {
    const unsigned long long seq = ctx->cg_deferred_seq;
    ctx->cg_deferred_seq = 0;
    if (__atomic_load_n(ctx->poll_flag, __ATOMIC_ACQUIRE) < seq) {
      ctx->host_syncs++;
      MI_TRY(poll_finish(ctx, seq, "stpcg deferred read-back"));
    }
  } else {
    hipError_t e = hipEventQuery(ctx->cg_deferred_ev);
    if (e == hipErrorNotReady) {
      e = hipEventSynchronize(ctx->cg_deferred_ev);
      ctx->host_syncs++;
    }
    if (e != hipSuccess) return hip_fail(e, "stpcg deferred read-back", __FILE__, __LINE__);
  }
  int ipc_err = 0;
  (void)mi_comm_ipc_error(ctx, &ipc_err);
  if (ipc_err) {
    set_error("a wait in the peer-memory exchange layer timed out: the result of this solve is invalid");
    return MI_ERR_COMM;
  }
  const CgState &f = *ctx->cg_host;
  result->update_step_M_norm = f.M_norm;
  result->num_iterations = (size_t)f.k;
  result->exit_reason = f.exit_reason;
  result->rv_final = f.rv;
  result->hvp_calls = ctx->cg_deferred_hvp;
  return precon_fail_check(ctx, result);
}

int mi_stpcg(mi_ctx *ctx, const mi_vec *g, mi_op *H, mi_precon *P, const mi_stpcg_params *prm,
             mi_vec *s_out, mi_stpcg_result *result, mi_stpcg_trace *trace) {
  MI_REQUIRE(ctx && g && H && prm && s_out && result, "null argument");
  MI_REQUIRE(g->ctx == ctx && s_out->ctx == ctx && H->ctx == ctx, "objects belong to another context");
  MI_REQUIRE(g->n == s_out->n && g->n == H->n, "dimension mismatch: g %zu, s %zu, H %zu", g->n,
             s_out->n, H->n);
  MI_REQUIRE(g->d != s_out->d, "g and s_out must not alias");
  touch(s_out);
  ctx->fusion.fused_stpcg_solves++;
  MI_REQUIRE(!P || (P->ctx == ctx && P->n == g->n), "preconditioner dimension/context mismatch");
  MI_REQUIRE(!prm->constraint_At || (P && P->apply_project),
             "constraint_At needs a constraint preconditioner (mi_precon_create_constraint)");
  MI_REQUIRE(!(P && P->kind == 3 && ctx->world_size > 1),
             "the constraint preconditioner is single-rank: its reductions are not completed across ranks");
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

  RangeScope range("mi_stpcg");
  comm_halo_fold_drop(ctx);  // (nothing may be pending from an earlier solve: see cleanup)
  const size_t n = g->n;
  const int pre = !P ? PRE_NONE : (P->kind == 1 ? PRE_DIAG : (P->kind == 2 ? PRE_BLOCK3 : PRE_EXTERNAL));
  MI_REQUIRE(pre != PRE_BLOCK3 || n % 3 == 0, "block-Jacobi preconditioner needs n divisible by 3");
  // several ranks: the partial rows themselves are all-reduced (rows_mode) and the consumers keep
  // their prologue re-reduction; the slot variants stay reachable through MI355OPT_FORCE_SLOT_PATH
  const bool sharded = slot_mode(ctx);
  const bool rows = rows_mode(ctx);
  // direction-Gram fusion (mi_op::dirgram): one-pass Hessian, the Gram rows come from the direction kernel
  const mi_dirgram *dgp = (!ctx->no_dirgram && H->dirgram && H->apply_dir) ? H->dirgram : nullptr;
  // rows wider than 4 doubles (Stiefel p = 5 ... 8) have the one-pass operator in its recurrence form only (no
  // preconditioner); with one, the operator keeps its two passes and the direction kernel its flat form
  if (dgp && dgp->p > 4 && (pre != PRE_NONE || ctx->dirgram_direct)) dgp = nullptr;
  MI_REQUIRE(!dgp || (dgp->p >= 1 && dgp->p <= kMaxP && dgp->n * (size_t)dgp->p == g->n),
             "operator's direction-Gram description does not match the problem dimension");
  // (default depth 3: depth 2 gives the same cfg2 step and -0.6 % on a cfg3 TNT run)
  const int run_ahead = prm->run_ahead > 0 ? prm->run_ahead : 3;
  const bool lockstep = (ctx->comm != nullptr && ctx->world_size > 1) || ctx->force_lockstep;

  // the failure word is per solve: a NaN residual in one solve must not condemn the object for the next
  if (P && P->fail_word) MI_HIP(hipMemsetAsync(P->fail_word, 0, sizeof(double), ctx->stream));

  mi_vec *r = nullptr, *v = nullptr, *p = nullptr, *Hp = nullptr;
  MI_TRY(mi_vec_create(ctx, n, &r));
  MI_TRY(mi_vec_create(ctx, n, &p));
  MI_TRY(mi_vec_create(ctx, n, &Hp));
  if (pre != PRE_NONE) MI_TRY(mi_vec_create(ctx, n, &v));
  double *vd = (pre == PRE_NONE) ? r->d : v->d;  // v aliases r when P is absent (:231,383)
  const double *pred = P ? P->data : nullptr;

  size_t tcap = 0;
  if (trace && trace->cap) {
    if (ctx->trace_cap < trace->cap) {
      if (ctx->trace_dev) MI_HIP(hipFree(ctx->trace_dev));
      MI_HIP(hipMalloc((void **)&ctx->trace_dev, 4 * trace->cap * sizeof(double)));
      ctx->trace_cap = trace->cap;
    }
    tcap = ctx->trace_cap;  // layout stride
  }

  ctx->cg_deferred = false;  // (a result nobody collected is superseded)
  ctx->cg_deferred_seq = 0;
  ctx->epoch++;
  ctx->status->word = 0;
  ctx->status->epoch = ctx->epoch;

  CgSetup cfg{prm->Delta, prm->kappa_fgr, prm->theta, prm->epsilon,
              (unsigned long long)prm->max_iterations};
  CgConst cc{prm->Delta, prm->Delta * prm->Delta, prm->epsilon, (unsigned long long)prm->max_iterations, 0ull, 0ull};
  hipStream_t st = ctx->stream;
  const int grid = (pre == PRE_BLOCK3) ? grid_for(ctx, n / 3, 2) : grid_for(ctx, n, 4);
  {
    const size_t n2 = n >> 1, step = (size_t)grid * kBlock;
    cc.rag0 = (n2 / step) * step;
    cc.rag_per = (n2 - cc.rag0 + (size_t)grid - 1) / (size_t)grid;
  }
  // recurrence form of the direction Gram: needs v == r (no preconditioner)
  const bool recur = dgp && pre == PRE_NONE && !ctx->dirgram_direct;
  const int gns = dgp ? dgp->p * (dgp->p + 1) / 2 : 0;
  const DirGramArgs dga = dgp ? DirGramArgs{dgp->X, dgp->Y, dgp->S, ctx->partials2,
                                            recur ? ctx->scalars + SLOT_GDIR : nullptr, gns}
                              : DirGramArgs{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  const int sp = (dgp && !recur) ? dgp->p : 0;  // row form of k_cg_pupdate only in the direct form
  const int kc = recur ? dir_comps(dgp->p) : 3;  // components of the operator's partial rows
  double *slots_g = ctx->scalars + SLOT_GRAM;    // slot file of the recurrence form (kc <= 16 doubles)
  double *slots_a = ctx->scalars + SLOT_CG, *slots_b = ctx->scalars + SLOT_CG + 4;
  CgState *st0 = ctx->cg, *st1 = ctx->cg1;
  // Several ranks through the peer-memory layer, unpreconditioned recurrence form (the cfg2 / cfg4 hot loop): the two
  // scalar exchanges of an iteration are folded into the prologues of their consumers (no exchange kernels).  Every
  // other combination keeps the separate exchange kernels.
  // (the folded exchange carries at most kIpcVals = 16 values: p <= 4)
  const bool folded = sharded && recur && kc <= kIpcVals && comm_fold_enabled(ctx) && !ctx->force_slot_path;
  // r'-halo form (Config::halo_rprime): the halo rows of r' travel with the <r,v> all-reduce, halo(p') is formed locally
  bool rprime = false;
  if (recur && !folded && dgp->halo_A) {
    // (allocates the residual-halo buffers on first use; a rank that cannot must fail HERE, loudly, rather than take
    // another collective sequence than its peers -- ADVICE r05)
    const int st_rp = comm_rprime_prepare(ctx, dgp->halo_A, &rprime);
    if (st_rp != MI_OK) {
      mi_vec_destroy(r);
      mi_vec_destroy(p);
      mi_vec_destroy(Hp);
      if (v) mi_vec_destroy(v);
      return st_rp;
    }
  }
  // unpreconditioned recurrence form: initialisation and the first direction's Gram rows in one pass
  // (k_cg_init_dirgram); one rank, nothing to exchange: G(p0), G(r0) set by k_cg_scalar_init itself
  const bool init_fused = recur && n == dgp->n * (size_t)dgp->p;
  const bool gdir_in_scalar_init = recur && !sharded && !rows;
  // opt-in experiment: the two-kernel step (k_cg_step2; one rank, Stiefel(n,3) in the window form)
  const bool twok = ctx->cfg.two_kernel_step && recur && dgp->twok && dgp->p == 3 && !sharded && !rows && !lockstep &&
                    init_fused && gdir_in_scalar_init;
  CgState *st_final = st0;
  double *tr = tcap ? ctx->trace_dev : nullptr;
  int ret = MI_OK;
  ctx->cg_live = st0;

#define CG_CHECK(expr)   \
  do {                   \
    int _s = (expr);     \
    if (_s != MI_OK) {   \
      ret = _s;          \
      goto cleanup;      \
    }                    \
  } while (0)
#define LAUNCH_PRE(KERNEL, ...)                                                                   \
  switch (pre) {                                                                                  \
    case PRE_NONE: hipLaunchKernelGGL((KERNEL<PRE_NONE>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__); break;     \
    case PRE_DIAG: hipLaunchKernelGGL((KERNEL<PRE_DIAG>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__); break;     \
    case PRE_BLOCK3: hipLaunchKernelGGL((KERNEL<PRE_BLOCK3>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__); break; \
    default: hipLaunchKernelGGL((KERNEL<PRE_EXTERNAL>), dim3(grid), dim3(kBlock), 0, st, __VA_ARGS__); break;       \
  }
#define LAUNCH_UPDATE(FS)                                                                              \
  switch (pre) {                                                                                       \
    case PRE_NONE: hipLaunchKernelGGL((k_cg_update<PRE_NONE, FS>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break;     \
    case PRE_DIAG: hipLaunchKernelGGL((k_cg_update<PRE_DIAG, FS>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break;     \
    case PRE_BLOCK3: hipLaunchKernelGGL((k_cg_update<PRE_BLOCK3, FS>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break; \
    default: hipLaunchKernelGGL((k_cg_update<PRE_EXTERNAL, FS>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break;       \
  }

  // --- initialisation -----------------------------------------------------------------------
  if (init_fused) {
    KScope ks(ctx, MI_K_CG_INIT);
    const size_t nrows = dgp->n;
#define INIT_DG(SPV)                                                                                       \
  hipLaunchKernelGGL(k_cg_init_dirgram<SPV>, dim3(grid), dim3(StBlk<SPV>::threads), 0, st, nrows, (const double *)g->d, \
                     r->d, p->d, s_out->d, ctx->partials_b, dga)
    switch (dgp->p) {
      case 1: INIT_DG(1); break;
      case 2: INIT_DG(2); break;
      case 3: INIT_DG(3); break;
      case 4: INIT_DG(4); break;
      case 5: INIT_DG(5); break;
      case 6: INIT_DG(6); break;
      case 7: INIT_DG(7); break;
      default: INIT_DG(8); break;
    }
#undef INIT_DG
  } else {
    KScope ks(ctx, MI_K_CG_INIT);
    LAUNCH_PRE(k_cg_init, n, (const double *)g->d, pred, r->d, vd, p->d, s_out->d, ctx->partials_b);
  }
  if (pre == PRE_EXTERNAL) {
    if (P->apply_project) CG_CHECK(P->apply_project(P, r, v, prm->constraint_At));  // (v, lambda) = P(r); r -= A' lambda
    else CG_CHECK(P->apply(P, r, v));
    hipLaunchKernelGGL(k_cg_dot_rv<true>, dim3(grid), dim3(kBlock), 0, st, n, (const CgState *)st0,
                       (const double *)r->d, (const double *)v->d, p->d, ctx->partials_b);
  }
  if (dgp) {
    const size_t nrows = dgp->n;
    if (!init_fused) {
      switch (dgp->p) {
        case 1: hipLaunchKernelGGL(k_cg_dirgram<1>, dim3(grid), dim3(kBlock), 0, st, nrows, (const double *)p->d, dga); break;
        case 2: hipLaunchKernelGGL(k_cg_dirgram<2>, dim3(grid), dim3(kBlock), 0, st, nrows, (const double *)p->d, dga); break;
        case 3: hipLaunchKernelGGL(k_cg_dirgram<3>, dim3(grid), dim3(kBlock), 0, st, nrows, (const double *)p->d, dga); break;
        case 4: hipLaunchKernelGGL(k_cg_dirgram<4>, dim3(grid), dim3(kBlock), 0, st, nrows, (const double *)p->d, dga); break;
        case 5: hipLaunchKernelGGL(k_cg_dirgram<5>, dim3(grid), dim3(StBlk<5>::threads), 0, st, nrows, (const double *)p->d, dga); break;
        case 6: hipLaunchKernelGGL(k_cg_dirgram<6>, dim3(grid), dim3(StBlk<6>::threads), 0, st, nrows, (const double *)p->d, dga); break;
        case 7: hipLaunchKernelGGL(k_cg_dirgram<7>, dim3(grid), dim3(StBlk<7>::threads), 0, st, nrows, (const double *)p->d, dga); break;
        default: hipLaunchKernelGGL(k_cg_dirgram<8>, dim3(grid), dim3(StBlk<8>::threads), 0, st, nrows, (const double *)p->d, dga); break;
      }
    }
    if (recur && !gdir_in_scalar_init) {
      if (sharded) CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials2, grid, gns, slots_g));
      else if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials2, gns));
      hipLaunchKernelGGL(k_cg_gdir_init, dim3(1), dim3(kBlock), 0, st, (const double *)ctx->partials2, grid, gns,
                         (const double *)slots_g, sharded ? 1 : 0, dga.gdir);
    }
  }
  if (sharded) {
    CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials_b, grid, 1, slots_b));
    hipLaunchKernelGGL(k_cg_scalar_init<true>, dim3(1), dim3(kBlock), 0, st, st0, cfg,
                       (const double *)ctx->partials_b, grid, (const double *)slots_b, ctx->status_dev,
                       (const double *)nullptr, 0, 0, (double *)nullptr);
  } else {
    if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials_b, 1));
    hipLaunchKernelGGL(k_cg_scalar_init<false>, dim3(1), dim3(kBlock), 0, st, st0, cfg,
                       (const double *)ctx->partials_b, grid, (const double *)slots_b, ctx->status_dev,
                       (const double *)ctx->partials2, grid, gns, gdir_in_scalar_init ? dga.gdir : (double *)nullptr);
  }
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) CG_CHECK(hip_fail(e, "stpcg init launch", __FILE__, __LINE__));
  }

  // --- main loop: speculative enqueue with bounded run-ahead -----------------------------------
  {
    size_t hvp = 0;
    for (size_t k = 0; k < prm->max_iterations; ++k) {
      // throttle: at most run_ahead launches ahead of the device
      uint64_t w = ctx->status->word;
      while (!(w & 1) && k > (w >> 1) + (uint64_t)run_ahead) {
        cpu_relax();
        w = ctx->status->word;
      }
      if (w & 1) {
        // One rank: stop as soon as the exit is seen.  Several ranks: every enqueued iteration carries
        // collectives, so all ranks must enqueue the SAME number -- exactly (launches at exit) + run_ahead,
        // a function of the replicated device state only, not of when this host happened to look.  (No
        // rank can get past that count unknowingly: the throttle holds it until the exit is published.)
        if (!lockstep || k >= (w >> 1) + (uint64_t)run_ahead) break;
      }

      // r06: re-anchor the recurrence form's G(p), G(r) by the direct form every cfg.reanchor iterations (the count is
      // the loop index: the same on every rank, so the exchanges it carries stay matched across ranks)
      if (recur && !twok && ctx->cfg.reanchor > 0 && k > 0 && k % (size_t)ctx->cfg.reanchor == 0) {
        const size_t nrows = dgp->n;
        if (dgp->p <= 4) {
          // rows of <= 4 doubles: both Grams from ONE pass over X, Y, p, r and one (all-)reduction of 2 gns components
          switch (dgp->p) {
            case 1: hipLaunchKernelGGL(k_cg_dirgram2<1>, dim3(grid), dim3(kBlock), 0, st, nrows, (const CgState *)st0, (const double *)p->d, (const double *)r->d, dga); break;
            case 2: hipLaunchKernelGGL(k_cg_dirgram2<2>, dim3(grid), dim3(kBlock), 0, st, nrows, (const CgState *)st0, (const double *)p->d, (const double *)r->d, dga); break;
            case 3: hipLaunchKernelGGL(k_cg_dirgram2<3>, dim3(grid), dim3(kBlock), 0, st, nrows, (const CgState *)st0, (const double *)p->d, (const double *)r->d, dga); break;
            default: hipLaunchKernelGGL(k_cg_dirgram2<4>, dim3(grid), dim3(kBlock), 0, st, nrows, (const CgState *)st0, (const double *)p->d, (const double *)r->d, dga); break;
          }
          if (sharded) CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials2, grid, 2 * gns, slots_g));
          else if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials2, 2 * gns));
          hipLaunchKernelGGL(k_cg_gdir_anchor, dim3(1), dim3(kBlock), 0, st, (const double *)ctx->partials2, grid, gns,
                             (const double *)slots_g, sharded ? 1 : 0, (const CgState *)st0, dga.gdir + SLOT_GDIR_P,
                             dga.gdir);
        } else {
          for (int which = 0; which < 2; ++which) {
            const double *vec = which == 0 ? p->d : r->d;
            switch (dgp->p) {
              case 5: hipLaunchKernelGGL(k_cg_dirgram<5>, dim3(grid), dim3(StBlk<5>::threads), 0, st, nrows, vec, dga); break;
              case 6: hipLaunchKernelGGL(k_cg_dirgram<6>, dim3(grid), dim3(StBlk<6>::threads), 0, st, nrows, vec, dga); break;
              case 7: hipLaunchKernelGGL(k_cg_dirgram<7>, dim3(grid), dim3(StBlk<7>::threads), 0, st, nrows, vec, dga); break;
              default: hipLaunchKernelGGL(k_cg_dirgram<8>, dim3(grid), dim3(StBlk<8>::threads), 0, st, nrows, vec, dga); break;
            }
            if (sharded) CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials2, grid, gns, slots_g));
            else if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials2, gns));
            hipLaunchKernelGGL(k_cg_gdir_anchor, dim3(1), dim3(kBlock), 0, st, (const double *)ctx->partials2, grid, gns,
                               (const double *)slots_g, sharded ? 1 : 0, (const CgState *)st0,
                               dga.gdir + (which == 0 ? SLOT_GDIR_P : 0), (double *)nullptr);
          }
        }
      }

      // Hp = H(p) (:294) + partial rows of <p,Hp>, <Hp,Hp>, <p,p> in ctx->partials
      int nparts = 0;
      if (twok) {  // two launches per iteration; the state ping-pongs between st0 and st1 ACROSS iterations
        CgState *cur = (k & 1) ? st1 : st0, *nxt = (k & 1) ? st0 : st1;
        ctx->cg_live = cur;
        ctx->twok_r = r->d;
        CG_CHECK(H->apply_dir(H, p, Hp, -2, &nparts));
        ctx->twok_r = nullptr;
        ++hvp;
        {
          KScope ks(ctx, MI_K_CG_UPDATE);
          hipLaunchKernelGGL(k_cg_step2<6>, dim3(grid), dim3(kBlock), 0, st, n, cc, (const CgState *)cur, nxt,
                             (const double *)ctx->partials, nparts, (const double *)Hp->d, r->d, p->d, s_out->d,
                             ctx->status_dev, tr, tcap, dga);
        }
        st_final = nxt;
        continue;
      }
      if (dgp) {
        CG_CHECK(H->apply_dir(H, p, Hp, recur ? -1 : grid, &nparts));
      } else if (H->apply_dots) {
        CG_CHECK(H->apply_dots(H, p, Hp, &nparts));
      } else {
        CG_CHECK(H->apply(H, p, Hp));
        CG_CHECK(launch_dot3_partials(ctx, n, p->d, Hp->d, &nparts));
      }
      ++hvp;
#define UPD_ARGS                                                                                      \
  n, cc, (const CgState *)st0, st1, (const double *)ctx->partials, nparts,                              \
      (const double *)(recur ? slots_g : slots_a), (const double *)p->d, (const double *)Hp->d, pred,  \
      s_out->d, r->d, vd, ctx->partials_b, dga
      if (recur) {
        // 3 dots + the Gram rows of Hp in one reduction (and one exchange across ranks)
        // KN: k_cg_update, or its 80-SGPR twin for the instantiations that need more (see above; 16 components always do)
#define UPD_RECUR(KN, FS, FT, FV)                                                                                  \
  switch (kc) {                                                                                                    \
    case 4: hipLaunchKernelGGL((KN<PRE_NONE, FS, 4, FT>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, FV); break;   \
    case 6: hipLaunchKernelGGL((KN<PRE_NONE, FS, 6, FT>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, FV); break;   \
    case 9: hipLaunchKernelGGL((KN<PRE_NONE, FS, 9, FT>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, FV); break;   \
    default: hipLaunchKernelGGL((k_cg_update_s80<PRE_NONE, FS, 16, FT>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, FV); break; \
  }
  // p = 5 ... 8: 18, 24, 31, 39 components (3 + p (p + 1) / 2).  Never folded (the host keeps the separate exchange for
  // them), and launched from the uncapped kernel: 39 reduced values per thread do not fit the 64 vector registers the
  // two-workgroups-per-CU twins are held to
#define UPD_WIDE(FS)                                                                                                        \
  switch (kc) {                                                                                                             \
    case 18: hipLaunchKernelGGL((k_cg_update<PRE_NONE, FS, 18, NoFold>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break; \
    case 24: hipLaunchKernelGGL((k_cg_update<PRE_NONE, FS, 24, NoFold>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break; \
    case 31: hipLaunchKernelGGL((k_cg_update<PRE_NONE, FS, 31, NoFold>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break; \
    default: hipLaunchKernelGGL((k_cg_update<PRE_NONE, FS, 39, NoFold>), dim3(grid), dim3(kBlock), 0, st, UPD_ARGS, NoFold{}); break; \
  }
        if (kc > 16) {
          if (sharded) CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials, nparts, kc, slots_g));
          else if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials, kc));
          KScope ks(ctx, MI_K_CG_UPDATE);
          if (sharded) { UPD_WIDE(true); } else { UPD_WIDE(false); }
        } else if (sharded && folded) {
          // the sum over the ranks completes in the kernel's own prologue (comm_ipc.h): no exchange kernel
          const FoldArgs fold_a = comm_fold_next(ctx);
          KScope ks(ctx, MI_K_CG_UPDATE);
          UPD_RECUR(k_cg_update_s80, false, FoldArgs, fold_a);
        } else if (sharded) {
          CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials, nparts, kc, slots_g));
          KScope ks(ctx, MI_K_CG_UPDATE);
          UPD_RECUR(k_cg_update_s80, true, NoFold, NoFold{});
        } else {
          if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials, kc));
          KScope ks(ctx, MI_K_CG_UPDATE);
          UPD_RECUR(k_cg_update, false, NoFold, NoFold{});
        }
#undef UPD_RECUR
#undef UPD_WIDE
      } else if (sharded) {
        CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials, nparts, 3, slots_a));
        KScope ks(ctx, MI_K_CG_UPDATE);
        LAUNCH_UPDATE(true);
      } else {
        if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials, 3));
        KScope ks(ctx, MI_K_CG_UPDATE);
        LAUNCH_UPDATE(false);
      }
#undef UPD_ARGS
      if (pre == PRE_EXTERNAL) {
        if (P->apply_project) CG_CHECK(P->apply_project(P, r, v, prm->constraint_At));  // :386,403
        else CG_CHECK(P->apply(P, r, v));  // v = P(r) (:386)
        hipLaunchKernelGGL(k_cg_dot_rv<false>, dim3(grid), dim3(kBlock), 0, st, n, (const CgState *)st1,
                           (const double *)r->d, (const double *)v->d, p->d, ctx->partials_b);
      }
#define PUPD_ARGS                                                                                          \
  n, cc, (const CgState *)st1, st0, (const double *)ctx->partials_b, grid, (const double *)slots_b,        \
      (const double *)vd, p->d, s_out->d, ctx->status_dev, tr, tcap, dga
#define PUPD(KN, FS, SPV) \
  hipLaunchKernelGGL((KN<FS, SPV>), dim3(grid), dim3(kBlock), 0, st, PUPD_ARGS, NoFold{})
  // (KN0 / KN1: the kernel for rows of 0 / 1 doubles, the two forms that count on 8 waves per SIMD; wider rows are
  // bound by their vector registers)
#define LAUNCH_PUPD(KN0, KN1, FS)               \
  switch (sp) {                                 \
    case 0: PUPD(KN0, FS, 0); break;            \
    case 1: PUPD(KN1, FS, 1); break;            \
    case 2: PUPD(k_cg_pupdate, FS, 2); break;   \
    case 3: PUPD(k_cg_pupdate, FS, 3); break;   \
    default: PUPD(k_cg_pupdate, FS, 4); break;  \
  }
      if (sharded && folded) {
        const FoldArgs fold_b = comm_fold_next(ctx);  // (folded <=> recurrence form <=> sp == 0)
        // ... and the halo exchange of the NEXT Hessian pass rides in this kernel's stores of the new direction
        FoldPush fp{fold_b, HaloPush{}};
        const bool push = dgp && dgp->halo_A && k + 1 < prm->max_iterations &&
                          comm_halo_fold_next(ctx, dgp->halo_A, dgp->p, p->d, &fp.h);
        KScope ks(ctx, MI_K_CG_PUPDATE);
        if (push)
          hipLaunchKernelGGL((k_cg_pupdate_s80<false, 0, FoldPush>), dim3(grid), dim3(kBlock), 0, st, PUPD_ARGS, fp);
        else
          hipLaunchKernelGGL((k_cg_pupdate_s80<false, 0, FoldArgs>), dim3(grid), dim3(kBlock), 0, st, PUPD_ARGS, fold_b);
      } else if (sharded) {
        CG_CHECK(reduce_rows_allreduce(ctx, ctx->partials_b, grid, 1, slots_b));
        if (rprime) CG_CHECK(comm_rprime_exchange(ctx, dgp->halo_A, dgp->p, r->d, nullptr, 0));
        KScope ks(ctx, MI_K_CG_PUPDATE);
        LAUNCH_PUPD(k_cg_pupdate_s80, k_cg_pupdate_s80, true);
      } else {
        if (rows && rprime) CG_CHECK(comm_rprime_exchange(ctx, dgp->halo_A, dgp->p, r->d, ctx->partials_b, 1));
        else if (rows) CG_CHECK(comm_allreduce_rows(ctx, ctx->partials_b, 1));
        KScope ks(ctx, MI_K_CG_PUPDATE);
        // (the early-s form: single context, v = r or any plain vector, at most three elements per thread, <= 512 rows)
        const bool early = sp == 0 && !rows && ctx->cfg.early_s && grid <= 512 && cc.rag0 <= 2ull * (size_t)grid * kBlock &&
                           (n >> 1) < ((size_t)1 << 31);
        if (early)
          hipLaunchKernelGGL(k_cg_pupdate_early, dim3(grid), dim3(kBlock), 0, st, n, cc, (const CgState *)st1, st0,
                             (const double *)ctx->partials_b, grid, (const double *)vd, p->d, s_out->d, ctx->status_dev, tr,
                             tcap, dga);
        else
          LAUNCH_PUPD(k_cg_pupdate, k_cg_pupdate_s80, false);
      }
      if (rprime && (sharded || rows)) {
        const double *hr = nullptr;
        double *hp = nullptr;
        size_t cnt = 0;
        comm_rprime_buffers(ctx, dgp->halo_A, dgp->p, &hr, &hp, &cnt);
        if (cnt) {
          const int hg = (int)std::max<size_t>(1, std::min<size_t>((cnt + 255) / 256, 512));
          hipLaunchKernelGGL(k_halo_dir, dim3(hg), dim3(256), 0, st, (const CgState *)st0, hr, hp, cnt);
        }
        comm_rprime_mark(ctx, dgp->halo_A, dgp->p, p->d);
      }
#undef LAUNCH_PUPD
#undef PUPD
#undef PUPD_ARGS
    }
    result->hvp_calls = hvp;
    comm_halo_fold_drop(ctx);  // (a push behind the last enqueued iteration has no reader)
  }

  // --- read back the final state ----------------------------------------------------------------
  if (prm->defer_result && !(trace && trace->cap)) {
    // the copy travels behind the solve; whoever waits for the stream next (or mi_stpcg_collect) completes it
    hipError_t e = hipSuccess;
    *precon_fail_host(ctx) = 0.0;
    ctx->cg_deferred_seq = result_to_host(ctx, st_final, P);
    if (!ctx->cg_deferred_seq) {
      if (!ctx->cg_deferred_ev) e = hipEventCreateWithFlags(&ctx->cg_deferred_ev, hipEventDisableTiming);
      if (e == hipSuccess) e = hipMemcpyAsync(ctx->cg_host, st_final, sizeof(CgState), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess && P && P->fail_word)
        e = hipMemcpyAsync(precon_fail_host(ctx), P->fail_word, sizeof(double), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipEventRecord(ctx->cg_deferred_ev, st);
    }
    if (e != hipSuccess) CG_CHECK(hip_fail(e, "stpcg deferred read-back", __FILE__, __LINE__));
    ctx->cg_deferred = true;
    ctx->cg_deferred_hvp = result->hvp_calls;
    result->update_step_M_norm = 0;
    result->num_iterations = 0;
    result->exit_reason = -1;
    result->rv_final = 0;
    result->precon_status = 0;
  } else {
    hipError_t e = hipSuccess;
    *precon_fail_host(ctx) = 0.0;
    ctx->host_syncs++;
    if (const unsigned long long seq = result_to_host(ctx, st_final, P)) {
      CG_CHECK(poll_finish(ctx, seq, "stpcg read-back"));
    } else {
      e = hipMemcpyAsync(ctx->cg_host, st_final, sizeof(CgState), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess && P && P->fail_word)
        e = hipMemcpyAsync(precon_fail_host(ctx), P->fail_word, sizeof(double), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (e != hipSuccess) CG_CHECK(hip_fail(e, "stpcg read-back", __FILE__, __LINE__));
    }
    CG_CHECK(precon_fail_check(ctx, result));
    {
      int ipc_err = 0;
      (void)mi_comm_ipc_error(ctx, &ipc_err);
      if (ipc_err) {
        set_error("a wait in the peer-memory exchange layer timed out: the result of this solve is invalid");
        ret = MI_ERR_COMM;
        goto cleanup;
      }
    }
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
#undef LAUNCH_PRE
#undef LAUNCH_UPDATE
  ctx->cg_live = nullptr;
  // a folded halo push nobody consumed must not outlive the solve on ANY exit: p is a recycled pool pointer, and the
  // next solve's first Hessian pass would match the stale (A, p, V) entry, skip the real exchange and wait for a flag
  // that never comes (r04, ADVICE: an error exit between comm_halo_fold_next and the pass used to leave it set)
  comm_halo_fold_drop(ctx);
  if (ret != MI_OK) (void)hipStreamSynchronize(st);
  mi_vec_destroy(r);
  mi_vec_destroy(p);
  mi_vec_destroy(Hp);
  mi_vec_destroy(v);
  return ret;
}

}  // extern "C"
