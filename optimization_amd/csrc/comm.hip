// comm.hip -- multi-GPU layer: one process per GPU, RCCL over xGMI.
//
// The TNT / STPCG path shards by rows of the tangent vector.  Every vector update
// is local; each inner product is a local partial + ONE in-stream all-reduce of 1-9 fp64 scalars
// (latency-bound, not link-bandwidth-bound); a row-sharded sparse HVP needs one nearest-neighbour
// halo exchange (ncclSend/ncclRecv of boundary rows, ~1 MB).  The scalar recurrences are replicated:
// ncclAllReduce returns identical bits on every rank, so all ranks take identical branch decisions.
// The reference has no communication layer at all -- nothing to mirror.
#include <rccl/rccl.h>

#include <cstdlib>

#include "comm_ipc.h"
#include "mi_internal.h"

using namespace mi;

namespace {
// ---- peer-memory ("IPC") exchange layer ------------------------------------------------------
// RCCL's all-reduce of a handful of scalars costs a kernel launch plus ~15-25 us of protocol on an
// 8-GPU xGMI node; the STPCG path needs three of them and one halo exchange per 85 us iteration.  For
// these tiny, latency-bound exchanges every rank instead maps every other rank's ARENA (fine-grained
// device memory, exported with hipIpcGetMemHandle / opened with hipIpcOpenMemHandle: plain xGMI peer
// stores) and one workgroup does: reduce partial rows -> store my K values + a sequence-numbered flag
// into every peer's mailbox -> wait for every peer's flag in MY mailbox -> sum in rank order.  Same
// summation order on every rank => replicated, deterministic results.  Waits are bounded (timeout ->
// error word, never a hang); the layer is switched on only after a collective self-test passed on
// every rank, else RCCL stays in charge.
// (mailbox layout and the device-side wait / folded exchange: comm_ipc.h)
constexpr size_t kIpcArenaBytes = 64u << 20;   // mailbox + halo regions (r05: 16 -> 64 MB: halo buffers hold kMaxP = 8 columns, and
                                               // the r'-halo form keeps a second pair per matrix: 2 x 10.2 MB at cfg4's 200 x 200 plane)
constexpr uint64_t kIpcTimeoutTicks = 2000000000ull;  // default: 20 s of the 100 MHz wall clock (MI355OPT_IPC_TIMEOUT_MS)

struct Comm {
  ncclComm_t nccl = nullptr;
  double *scratch = nullptr;  // device, small all-gather buffer
  // IPC layer
  bool ipc_mapped = false, ipc_enabled = false;
  bool fold = true;                      // exchanges folded into the consumers' prologues (MI355OPT_NO_FOLD=1: off)
  char *arena = nullptr;                 // my arena (fine-grained device memory)
  char *peer[kIpcMaxRanks] = {nullptr};  // mapped arenas, peer[rank] == arena
  char **peer_dev = nullptr;             // device copy of peer[]
  uint64_t seq = 0, halo_seq = 0;
  size_t arena_top = kIpcMailboxBytes;   // bump allocator for halo regions
  unsigned int *err_host = nullptr, *err_dev = nullptr;  // pinned, device-visible error word
  double *vals_dev = nullptr;            // staging for value exchanges (kIpcMaxRanks * kIpcVals)
  uint64_t timeout = kIpcTimeoutTicks;   // bound of every device-side wait, in 100 MHz ticks
  // a halo push folded into the kernel that wrote the vector (comm_halo_fold_next), not yet consumed
  struct { const mi_csr *A = nullptr; const double *V = nullptr; int p = 0; uint64_t seq = 0; } pushed;
  // r'-halo form: the halo of this (A, p, V) was formed locally (comm_rprime_mark), not yet consumed
  struct { const mi_csr *A = nullptr; const double *V = nullptr; int p = 0; } formed;
  // kernels this layer launched itself: [0] scalar exchanges, [1] halo pushes, and [2] halo pushes that rode in the
  // producer kernel instead (mi_comm_kernel_launches)
  unsigned long long launched[4] = {0, 0, 0, 0};  // ([3]: of the folded pushes, those in the early form)
};

int nccl_fail(ncclResult_t r, const char *what) {
  set_error("RCCL error %d (%s) in %s", (int)r, ncclGetErrorString(r), what);
  return MI_ERR_COMM;
}
#define MI_NCCL(expr)                                   \
  do {                                                  \
    ncclResult_t _r = (expr);                           \
    if (_r != ncclSuccess) return nccl_fail(_r, #expr); \
  } while (0)

// One workgroup.  vals: K values per rank, either the fixed-order sums of `count` partial rows
// (partials != null) or in_vals[0..K).  SUM: out[0..K) = sum over ranks in rank order; else
// out[r * K + k] = value k of rank r (all-gather).
template <int K, bool SUM>
__global__ __launch_bounds__(kBlock) void k_ipc_exchange(const double *__restrict__ partials, int count,
                                                         const double *__restrict__ in_vals, char *const *peers,
                                                         int P, int rank, uint64_t seq, double *__restrict__ out,
                                                         unsigned int *err, uint64_t timeout) {
  __shared__ double lds[K * (kWaves + 1)];
  double v[K];
  if (partials) {
    reduce_rows<K>(partials, count, v, lds);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = in_vals[k];
  }
  const int q = (int)(seq % kIpcRing);
  const int t = threadIdx.x;
  if (t < P) {  // thread t serves peer t: push my values, then the flag (release)
    IpcMailbox *mb = reinterpret_cast<IpcMailbox *>(peers[t]);
#pragma unroll
    for (int k = 0; k < K; ++k)
      __hip_atomic_store(&mb->val[q][rank][k], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->flag[q][rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ... and wait for peer t's flag in MY mailbox
    IpcMailbox *mine = reinterpret_cast<IpcMailbox *>(peers[rank]);
    ipc_wait(&mine->flag[q][t], seq, err, timeout);
  }
  __syncthreads();
  if (t < K) {
    const IpcMailbox *mine = reinterpret_cast<const IpcMailbox *>(peers[rank]);
    if (SUM) {
      double s = 0;
      for (int r = 0; r < P; ++r)
        s += __hip_atomic_load(&mine->val[q][r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      out[t] = s;
    } else {
      for (int r = 0; r < P; ++r)
        out[r * K + t] = __hip_atomic_load(&mine->val[q][r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Halo exchange by peer stores: my first `send_lo` rows go into rank-1's upper halo, my last `send_hi`
// rows into rank+1's lower halo (both inside the neighbour's arena); the last workgroup to finish
// raises the neighbours' flags and waits for theirs, so the kernel's completion means "my halo is in".
__global__ __launch_bounds__(256) void k_ipc_halo_push(const double *__restrict__ V, size_t n_doubles,
                                                       size_t lo_doubles, size_t hi_doubles, char *const *peers,
                                                       int P, int rank, size_t dst_lo_off, size_t dst_hi_off,
                                                       int expect_lo, int expect_hi, uint64_t seq,
                                                       unsigned int *err, uint64_t timeout) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (rank > 0 && lo_doubles) {
    double *dst = reinterpret_cast<double *>(peers[rank - 1] + dst_lo_off);
    for (size_t i = i0; i < lo_doubles; i += stride) dst[i] = V[i];
  }
  if (rank + 1 < P && hi_doubles) {
    double *dst = reinterpret_cast<double *>(peers[rank + 1] + dst_hi_off);
    const double *src = V + (n_doubles - hi_doubles);
    for (size_t i = i0; i < hi_doubles; i += stride) dst[i] = src[i];
  }
  // one system-scope release per workgroup, behind the barrier that orders the other waves' stores before it (a
  // release per thread writes the whole L2 back once per wave: comm_ipc.h halo_push_first_step_done); every wave has
  // waited for its own stores first (wave_stores_done)
  wave_stores_done();
  __syncthreads();
  __shared__ bool last;
  IpcMailbox *mine = reinterpret_cast<IpcMailbox *>(peers[rank]);
  if (threadIdx.x == 0) {
    __threadfence_system();
    const unsigned int done = __hip_atomic_fetch_add(&mine->halo_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (done == gridDim.x - 1);
  }
  __syncthreads();
  if (!last) return;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&mine->halo_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Both ranks of a pair that exchanges anything raise each other's flag and wait for it, also in the direction
    // in which no rows travel (expect_* = "this pair is active", the same on both of its ranks): a rank that only
    // sends could otherwise run exchanges ahead of the neighbour that still reads the halo buffer it would
    // overwrite (two buffers, so two exchanges ahead) whenever no collective separates two exchanges.
    if (rank > 0 && expect_lo)  // I am rank-1's upper neighbour: flag[1] there
      __hip_atomic_store(&reinterpret_cast<IpcMailbox *>(peers[rank - 1])->halo_flag[1], seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (rank + 1 < P && expect_hi)
      __hip_atomic_store(&reinterpret_cast<IpcMailbox *>(peers[rank + 1])->halo_flag[0], seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
    if (rank > 0 && expect_lo) ipc_wait(&mine->halo_flag[0], seq, err, timeout);
    if (rank + 1 < P && expect_hi) ipc_wait(&mine->halo_flag[1], seq, err, timeout);
  }
}


// ---- self-test of the FOLDED forms (r04, ADVICE: the check bench.py made must be the library's own) ---------------
// Stand-ins for k_cg_pupdate<FoldPush> and for the Hessian pass that consumes its halo, built from the SAME helpers in
// the same places: a folded scalar exchange in the prologue, halo_push_store next to the stores of a (rotated)
// grid-stride walk, the first-step signal of the early form, halo_push_finish of the late one; the consumer waits in
// its prologue and checks every halo double.  Values are exact functions of (rank, round, index), so a row that
// arrives late, never, or from the wrong exchange is counted.
__device__ __forceinline__ double selftest_val(int rank, int round, size_t idx) {
  return (double)(rank + 1) * 1099511627776.0 /* 2^40 */ + (double)round * 16777216.0 /* 2^24 */ + (double)idx;
}
template <class FOLD>
__global__ __launch_bounds__(kBlock) void k_ipc_selftest_producer(size_t n2, double2 *__restrict__ vec, int rank,
                                                                  int round, FOLD fold, double *__restrict__ sums) {
  __shared__ double lds[kIpcVals];
  double d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = (double)((rank + 1) * (k + 1)) + (double)round;
  fold_maybe<3>(d, fold, lds);
  if (blockIdx.x == 0 && threadIdx.x < 3) sums[threadIdx.x] = d[threadIdx.x];
  const size_t stride = (size_t)gridDim.x * kBlock, i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  bool pushed = false;
  for (size_t L = i0; L < n2; L += stride) {
    const size_t q = halo_phys(fold, L);
    const double2 v = make_double2(selftest_val(rank, round, 2 * q), selftest_val(rank, round, 2 * q + 1));
    vec[q] = v;
    pushed |= halo_push_store(fold, q, v);
    if (L == i0) halo_push_first_step_done(fold);
  }
  halo_push_finish(fold, pushed);
}
__global__ __launch_bounds__(256) void k_ipc_selftest_consumer(HaloWait w, const double *halo, size_t lo,
                                                                size_t hi, size_t n, int rank, int round,
                                                                unsigned int *__restrict__ bad) {
  halo_wait(w);
  const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned int mine = 0;
  // [0, lo): the LAST lo doubles of rank-1; [lo, lo + hi): the FIRST hi doubles of rank+1.  PLAIN loads on purpose:
  // that is how the Hessian pass reads its halo, and the buffer was read two rounds ago -- a line that survived in a
  // cache through the wait's acquire would show up here
  for (size_t i = i0; i < lo; i += stride) mine += halo[i] != selftest_val(rank - 1, round, n - lo + i);
  for (size_t i = i0; i < hi; i += stride) mine += halo[lo + i] != selftest_val(rank + 1, round, i);
  if (mine) atomicAdd(bad, mine);
}

int ipc_exchange(mi_ctx *ctx, Comm *c, const double *partials, int count, const double *in_vals, int k, bool sum,
                 double *out) {
  const uint64_t seq = ++c->seq;
  ++c->launched[0];
  KScope ks(ctx, MI_K_COMM_ALLREDUCE);
#define IX(K)                                                                                              \
  if (sum)                                                                                                 \
    hipLaunchKernelGGL((k_ipc_exchange<K, true>), dim3(1), dim3(kBlock), 0, ctx->stream, partials, count,  \
                       in_vals, (char *const *)c->peer_dev, ctx->world_size, ctx->rank, seq, out, c->err_dev, c->timeout); \
  else                                                                                                     \
    hipLaunchKernelGGL((k_ipc_exchange<K, false>), dim3(1), dim3(kBlock), 0, ctx->stream, partials, count, \
                       in_vals, (char *const *)c->peer_dev, ctx->world_size, ctx->rank, seq, out, c->err_dev, c->timeout)
  switch (k) {
    case 1: IX(1); break;
    case 2: IX(2); break;
    case 3: IX(3); break;
    case 4: IX(4); break;
    case 6: IX(6); break;
    case 9: IX(9); break;
    case 10: IX(10); break;
    case 16: IX(16); break;
    default: set_error("unsupported exchange width %d", k); return MI_ERR_INTERNAL;
  }
#undef IX
  MI_HIP(hipGetLastError());
  return MI_OK;
}
}  // namespace

namespace mi {

bool comm_ipc_enabled(const mi_ctx *ctx) { return ctx->comm && ((const Comm *)ctx->comm)->ipc_enabled; }

FoldArgs comm_fold_next(mi_ctx *ctx) {
  FoldArgs f;
  Comm *c = (Comm *)ctx->comm;
  if (!c || !c->ipc_enabled || !c->fold) return f;
  f.peers = (char *const *)c->peer_dev;
  f.err = c->err_dev;
  f.seq = ++c->seq;
  f.timeout = c->timeout;
  f.P = ctx->world_size;
  f.rank = ctx->rank;
  return f;
}
FoldArgs comm_fold_next_always(mi_ctx *ctx) {
  FoldArgs f;
  Comm *c = (Comm *)ctx->comm;
  if (!c || !c->ipc_enabled) return f;
  f.peers = (char *const *)c->peer_dev;
  f.err = c->err_dev;
  f.seq = ++c->seq;
  f.timeout = c->timeout;
  f.P = ctx->world_size;
  f.rank = ctx->rank;
  return f;
}
bool comm_fold_enabled(const mi_ctx *ctx) {
  const Comm *c = (const Comm *)ctx->comm;
  return c && c->ipc_enabled && c->fold;
}

int comm_allreduce(mi_ctx *ctx, double *buf, int count) {
  // a communicator of size 1 still goes through RCCL: the single-GPU box then exercises exactly the
  // calls the 8-GPU node makes (tests/test_gpu_comm.py)
  if (!ctx->comm) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  if (c->nccl) {
    KScope ks(ctx, MI_K_COMM_ALLREDUCE);
    MI_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->nccl, ctx->stream));
    return MI_OK;
  }
  // peer-memory layer only (no RCCL communicator): chunks of <= 16 values through the mailbox
  MI_REQUIRE(c->ipc_enabled, "no RCCL communicator and the peer-memory layer is not enabled");
  for (int off = 0; off < count; off += kIpcVals) {
    const int k = std::min(kIpcVals, count - off);
    // widths the exchange kernel is instantiated for: pad up through the staging buffer
    const int kk = k <= 4 ? k : (k <= 6 ? 6 : (k <= 10 ? (k <= 9 ? 9 : 10) : 16));
    MI_HIP(hipMemsetAsync(c->vals_dev, 0, sizeof(double) * kIpcVals, ctx->stream));
    MI_HIP(hipMemcpyAsync(c->vals_dev, buf + off, sizeof(double) * k, hipMemcpyDeviceToDevice, ctx->stream));
    MI_TRY(ipc_exchange(ctx, c, nullptr, 0, c->vals_dev, kk, true, c->vals_dev + kIpcVals));
    MI_HIP(hipMemcpyAsync(buf + off, c->vals_dev + kIpcVals, sizeof(double) * k, hipMemcpyDeviceToDevice,
                          ctx->stream));
  }
  return MI_OK;
}

int comm_allreduce_rows(mi_ctx *ctx, double *partials, int k) {
  // A uniform-grid context (every multi-rank context) leaves exactly kMaxGrid partial rows per component; rows
  // kMaxGrid..kMaxRows of the component-major buffer are never written there.  Reduce the live half only: k strided
  // segments of kMaxGrid doubles in one RCCL group (one launch), half the payload of the per-iteration exchanges.
  Comm *c = (Comm *)ctx->comm;
  if (c && c->nccl && ctx->uniform_grid && k > 0) {
    if (k == 1) return comm_allreduce(ctx, partials, kMaxGrid);
    KScope ks(ctx, MI_K_COMM_ALLREDUCE);
    MI_NCCL(ncclGroupStart());
    for (int j = 0; j < k; ++j) {
      double *seg = partials + (size_t)j * kMaxRows;
      ncclResult_t r = ncclAllReduce(seg, seg, (size_t)kMaxGrid, ncclDouble, ncclSum, c->nccl, ctx->stream);
      if (r != ncclSuccess) {
        (void)ncclGroupEnd();
        MI_NCCL(r);
      }
    }
    MI_NCCL(ncclGroupEnd());
    return MI_OK;
  }
  return comm_allreduce(ctx, partials, k * kMaxRows);
}

int reduce_rows_allreduce(mi_ctx *ctx, const double *partials, int count, int k, double *slots) {
  Comm *c = (Comm *)ctx->comm;
  // (the exchange kernel re-reduces the rows itself for the widths it is instantiated for; any other width -- the
  // Stiefel components of p >= 5 -- is reduced to slots first and exchanged in padded chunks by comm_allreduce)
  const bool ix = k == 1 || k == 2 || k == 3 || k == 4 || k == 6 || k == 9 || k == 10 || k == 16;
  if (c && c->ipc_enabled && ix)
    return ipc_exchange(ctx, c, partials, count, nullptr, k, true, slots);
  MI_TRY(launch_reduce_rows_to_slots(ctx, partials, count, k, slots));
  return comm_allreduce(ctx, slots, k);
}

bool comm_halo_fold_next(mi_ctx *ctx, const mi_csr *A, int p, const double *V, HaloPush *push) {
  *push = HaloPush{};
  if (ctx->world_size <= 1 || !ctx->comm || !A) return false;
  Comm *c = (Comm *)ctx->comm;
  if (!(c->ipc_enabled && c->fold && A->halo_in_arena)) return false;
  if (A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0) return false;
  const int rk = ctx->rank, ws = ctx->world_size;
  const size_t lo = rk > 0 ? A->send_lo * p : 0, hi = rk + 1 < ws ? A->send_hi * p : 0, nd = A->n * (size_t)p;
  // the pushing kernel stores double2: every boundary of the exchange must fall on one
  if ((lo | hi | nd | (A->peer_lo_rows * p) | A->halo_stride) & 1) return false;
  c->pushed = {};  // (an unconsumed one: its counters stay advanced, like an exchange nobody read)
  const size_t buf_off = A->halo_off + (++A->halo_exchanges & 1) * A->halo_stride * sizeof(double);
  const bool act_lo = rk > 0 && A->halo_lo + A->send_lo > 0, act_hi = rk + 1 < ws && A->halo_hi + A->send_hi > 0;
  push->seq = ++c->halo_seq;
  push->mine = reinterpret_cast<IpcMailbox *>(c->peer[rk]);
  if (lo) push->dst_lo = reinterpret_cast<double2 *>(c->peer[rk - 1] + buf_off + A->peer_lo_rows * p * sizeof(double));
  if (hi) push->dst_hi = reinterpret_cast<double2 *>(c->peer[rk + 1] + buf_off);
  push->lo2 = lo / 2;
  push->hi_from = (nd - hi) / 2;
  if (act_lo) push->mb_lo = reinterpret_cast<IpcMailbox *>(c->peer[rk - 1]);
  if (act_hi) push->mb_hi = reinterpret_cast<IpcMailbox *>(c->peer[rk + 1]);
  // the early form (comm_ipc.h): both boundary regions inside the pushing kernel's FIRST grid-stride step
  push->n2 = nd / 2;
  push->rot = push->hi_from;
  {
    const size_t first = (hi + lo) / 2, step = (size_t)grid_for(ctx, nd, 4) * kBlock;
    const bool late = ctx->cfg.halo_push_late;
    // (step <= n2: the kernel's first step is a whole grid-stride step)
    push->early_waves = (!late && first > 0 && first <= step && step <= push->n2) ? (unsigned int)((first + 63) / 64) : 0u;
  }
  c->pushed.A = A; c->pushed.V = V; c->pushed.p = p; c->pushed.seq = push->seq;
  ++c->launched[2];
  if (push->early_waves) ++c->launched[3];
  return true;
}

void comm_halo_fold_drop(mi_ctx *ctx) {
  if (ctx->comm) {
    ((Comm *)ctx->comm)->pushed = {};
    ((Comm *)ctx->comm)->formed = {};
  }
}

int comm_halo_exchange_or_wait(mi_ctx *ctx, const mi_csr *A, int p, const double *V, HaloWait *w) {
  *w = HaloWait{};
  Comm *c = (Comm *)ctx->comm;
  if (c && c->formed.A && c->formed.A == A && c->formed.V == V && c->formed.p == p) {
    c->formed = {};  // r'-halo form: the rows are in A->halo_cur() already (stream order), nothing to wait for
    return MI_OK;
  }
  if (c && c->pushed.seq && c->pushed.A == A && c->pushed.V == V && c->pushed.p == p) {
    const int rk = ctx->rank, ws = ctx->world_size;
    w->mine = reinterpret_cast<const IpcMailbox *>(c->peer[rk]);
    w->err = c->err_dev;
    w->seq = c->pushed.seq;
    w->timeout = c->timeout;
    w->expect_lo = rk > 0 && A->halo_lo + A->send_lo > 0;
    w->expect_hi = rk + 1 < ws && A->halo_hi + A->send_hi > 0;
    c->pushed = {};
    return MI_OK;
  }
  return comm_halo_exchange(ctx, A, p, V);
}

int comm_halo_exchange(mi_ctx *ctx, const mi_csr *A, int p, const double *V) {
  if (ctx->world_size <= 1 || !ctx->comm) return MI_OK;
  if (A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  c->pushed = {};
  c->formed = {};
  const int rk = ctx->rank, ws = ctx->world_size;
  if (c->ipc_enabled && A->halo_in_arena) {
    // this exchange's buffer (mi_csr::halo_stride); every rank counts the exchanges of a matrix identically
    const size_t buf_off = A->halo_off + (++A->halo_exchanges & 1) * A->halo_stride * sizeof(double);
    const size_t lo = A->send_lo * p, hi = A->send_hi * p;
    const size_t work = std::max(lo, hi);
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((work + 255) / 256, 64));
    KScope ks(ctx, MI_K_COMM_HALO);
    hipLaunchKernelGGL(k_ipc_halo_push, dim3(grid), dim3(256), 0, ctx->stream, V, A->n * (size_t)p, lo, hi,
                       (char *const *)c->peer_dev, ws, rk,
                       buf_off + A->peer_lo_rows * p * sizeof(double),  // behind rank-1's lower halo
                       buf_off,                                          // rank+1's lower halo
                       (int)(rk > 0 && A->halo_lo + A->send_lo > 0), (int)(rk + 1 < ws && A->halo_hi + A->send_hi > 0),
                       ++c->halo_seq,
                       c->err_dev, c->timeout);
    ++c->launched[1];
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  MI_REQUIRE(c->nccl, "halo exchange needs RCCL or the enabled peer-memory layer");
  KScope ks(ctx, MI_K_COMM_HALO);
  MI_NCCL(ncclGroupStart());
  if (rk > 0) {
    if (A->send_lo) MI_NCCL(ncclSend(V, A->send_lo * p, ncclDouble, rk - 1, c->nccl, ctx->stream));
    if (A->halo_lo) MI_NCCL(ncclRecv(const_cast<double *>(A->halo_cur()), A->halo_lo * p, ncclDouble, rk - 1, c->nccl, ctx->stream));
  }
  if (rk + 1 < ws) {
    if (A->send_hi)
      MI_NCCL(ncclSend(V + (A->n - A->send_hi) * p, A->send_hi * p, ncclDouble, rk + 1, c->nccl,
                       ctx->stream));
    if (A->halo_hi)
      MI_NCCL(ncclRecv(const_cast<double *>(A->halo_cur()) + A->halo_lo * p, A->halo_hi * p, ncclDouble, rk + 1, c->nccl,
                       ctx->stream));
  }
  MI_NCCL(ncclGroupEnd());
  return MI_OK;
}

// ---- r'-halo form (comm_ipc.h) -------------------------------------------------------------------------------------------
static int rprime_alloc(mi_ctx *ctx, const mi_csr *A) {
  if (A->halo_r) return MI_OK;
  // (every rank gets here at the same point of the same solve: the arena's bump allocator stays in step)
  return comm_halo_alloc(ctx, 2 * A->halo_stride * sizeof(double), &A->halo_r, &A->halo_r_in_arena, &A->halo_r_off);
}

// Whether this solve takes the r'-halo form.  The decision is made from replicated facts only (the switch, the layer in
// use, the halo extents' non-emptiness -- all the same on every rank), and the buffer pair is allocated HERE with the
// return code propagated (it used to be a side effect of a const query that answered "no" on a rank whose
// allocation failed -- that rank then enqueued the 3-collective form while its peers enqueued the r'-halo group, and the
// run hung instead of reporting anything).  *enabled is only meaningful when MI_OK is returned.
int comm_rprime_prepare(mi_ctx *ctx, const mi_csr *A, bool *enabled) {
  *enabled = false;
  if (!ctx->cfg.halo_rprime || ctx->world_size <= 1 || !ctx->comm || !A || !A->halo) return MI_OK;
  // (a rank without halo rows of its own still takes part in its neighbours' exchanges: every rank of a slab partition
  // with more than one slab sends or receives something, so this test is the same everywhere)
  if (A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0) return MI_OK;
  // the r' rows need a buffer pair of their own where the layer in use can reach it: inside the arena for peer stores
  // (the same decision on every rank: same sizes, same allocation order), anywhere for RCCL
  const Comm *c = (const Comm *)ctx->comm;
  const int st = rprime_alloc(ctx, A);
  if (st != MI_OK) {
    set_error("r'-halo form: this rank could not allocate its residual-halo buffers (%zu bytes); refusing to continue "
              "with a collective sequence that would differ from its peers'", 2 * A->halo_stride * sizeof(double));
    return MI_ERR_OOM;
  }
  if (c->ipc_enabled && A->halo_in_arena) *enabled = A->halo_r_in_arena;
  else *enabled = c->nccl != nullptr;
  return MI_OK;
}

void comm_rprime_buffers(const mi_ctx *, const mi_csr *A, int p, const double **halo_r, double **halo_p, size_t *count) {
  *halo_r = A->halo_r + (A->halo_r_exchanges & 1) * A->halo_stride;
  *halo_p = const_cast<double *>(A->halo_cur());
  *count = (A->halo_lo + A->halo_hi) * (size_t)p;
}

void comm_rprime_mark(mi_ctx *ctx, const mi_csr *A, int p, const double *V) {
  Comm *c = (Comm *)ctx->comm;
  c->pushed = {};
  c->formed.A = A; c->formed.V = V; c->formed.p = p;
}

int comm_rprime_exchange(mi_ctx *ctx, const mi_csr *A, int p, const double *R, double *partials, int k) {
  Comm *c = (Comm *)ctx->comm;
  MI_TRY(rprime_alloc(ctx, A));
  const int rk = ctx->rank, ws = ctx->world_size;
  if (c->ipc_enabled && A->halo_in_arena && A->halo_r_in_arena) {
    // peer-memory layer, separate kernels: the scalar exchange is the caller's own kernel; the rows by peer stores
    MI_REQUIRE(!partials, "r'-halo form: rows all-reduce and the peer-memory layer do not go together");
    const size_t buf_off = A->halo_r_off + (++A->halo_r_exchanges & 1) * A->halo_stride * sizeof(double);
    const size_t lo = A->send_lo * p, hi = A->send_hi * p;
    const size_t work = std::max(lo, hi);
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((work + 255) / 256, 64));
    KScope ks(ctx, MI_K_COMM_HALO);
    hipLaunchKernelGGL(k_ipc_halo_push, dim3(grid), dim3(256), 0, ctx->stream, R, A->n * (size_t)p, lo, hi,
                       (char *const *)c->peer_dev, ws, rk, buf_off + A->peer_lo_rows * p * sizeof(double), buf_off,
                       (int)(rk > 0 && A->halo_lo + A->send_lo > 0), (int)(rk + 1 < ws && A->halo_hi + A->send_hi > 0),
                       ++c->halo_seq, c->err_dev, c->timeout);
    ++c->launched[1];
    MI_HIP(hipGetLastError());
    return MI_OK;
  }
  MI_REQUIRE(c->nccl, "r'-halo form needs RCCL or the enabled peer-memory layer");
  double *dst = A->halo_r + (A->halo_r_exchanges & 1) * A->halo_stride;  // (in-stream collectives: one buffer will do)
  KScope ks(ctx, partials ? MI_K_COMM_ALLREDUCE : MI_K_COMM_HALO);
  MI_NCCL(ncclGroupStart());
  ncclResult_t r = ncclSuccess;
  for (int j = 0; j < k && partials && r == ncclSuccess; ++j) {
    double *seg = partials + (size_t)j * kMaxRows;
    r = ncclAllReduce(seg, seg, (size_t)(ctx->uniform_grid ? kMaxGrid : kMaxRows), ncclDouble, ncclSum, c->nccl, ctx->stream);
  }
  if (rk > 0) {
    if (A->send_lo && r == ncclSuccess) r = ncclSend(R, A->send_lo * p, ncclDouble, rk - 1, c->nccl, ctx->stream);
    if (A->halo_lo && r == ncclSuccess) r = ncclRecv(dst, A->halo_lo * p, ncclDouble, rk - 1, c->nccl, ctx->stream);
  }
  if (rk + 1 < ws) {
    if (A->send_hi && r == ncclSuccess)
      r = ncclSend(R + (A->n - A->send_hi) * p, A->send_hi * p, ncclDouble, rk + 1, c->nccl, ctx->stream);
    if (A->halo_hi && r == ncclSuccess)
      r = ncclRecv(dst + A->halo_lo * p, A->halo_hi * p, ncclDouble, rk + 1, c->nccl, ctx->stream);
  }
  if (r != ncclSuccess) {
    (void)ncclGroupEnd();
    MI_NCCL(r);
  }
  MI_NCCL(ncclGroupEnd());
  return MI_OK;
}

// Every rank learns how many rows its neighbours need from it: all-gather of (need_lo, need_hi).
int comm_exchange_halo_counts(mi_ctx *ctx, size_t need_lo, size_t need_hi, size_t *send_lo,
                              size_t *send_hi, size_t *peer_lo_rows, size_t *max_halo_rows) {
  *send_lo = *send_hi = *peer_lo_rows = 0;
  *max_halo_rows = need_lo + need_hi;
  if (ctx->world_size <= 1 || !ctx->comm) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  const int ws = ctx->world_size, rk = ctx->rank;
  std::vector<double> host(2 * (size_t)ws, 0.0);
  double mine[2] = {(double)need_lo, (double)need_hi};
  if (c->nccl) {
    MI_HIP(hipMemcpyAsync(c->scratch + 2 * rk, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    MI_NCCL(ncclAllGather(c->scratch + 2 * rk, c->scratch, 2, ncclDouble, c->nccl, ctx->stream));
    MI_HIP(hipMemcpyAsync(host.data(), c->scratch, host.size() * sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
  } else {
    MI_REQUIRE(c->ipc_enabled, "no RCCL communicator and the peer-memory layer is not enabled");
    MI_HIP(hipMemcpyAsync(c->vals_dev, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    MI_TRY(ipc_exchange(ctx, c, nullptr, 0, c->vals_dev, 2, false, c->vals_dev + kIpcVals));
    MI_HIP(hipMemcpyAsync(host.data(), c->vals_dev + kIpcVals, host.size() * sizeof(double), hipMemcpyDeviceToHost,
                          ctx->stream));
  }
  MI_HIP(hipStreamSynchronize(ctx->stream));
  if (rk > 0) {
    *send_lo = (size_t)host[2 * (rk - 1) + 1];       // rank-1 needs rows above its range
    *peer_lo_rows = (size_t)host[2 * (rk - 1) + 0];  // ... and stores them behind its own lower halo
  }
  if (rk + 1 < ws) *send_hi = (size_t)host[2 * (rk + 1) + 0];  // rank+1 needs rows below its range
  for (int r = 0; r < ws; ++r) *max_halo_rows = std::max(*max_halo_rows, (size_t)(host[2 * r] + host[2 * r + 1]));
  return MI_OK;
}

int comm_halo_alloc(mi_ctx *ctx, size_t bytes, double **ptr, bool *in_arena, size_t *arena_off) {
  *in_arena = false;
  *arena_off = 0;
  Comm *c = (Comm *)ctx->comm;
  bytes = (std::max<size_t>(bytes, 8) + 255) / 256 * 256;
  if (c && c->ipc_mapped && c->arena_top + bytes <= kIpcArenaBytes) {
    *ptr = reinterpret_cast<double *>(c->arena + c->arena_top);
    *arena_off = c->arena_top;
    *in_arena = true;
    c->arena_top += bytes;  // never reclaimed: a context holds a handful of sharded matrices
    // NOT zeroed here: the whole arena was zeroed before its handle was exported and a region is handed out once --
    // and a neighbour that is ahead may ALREADY have pushed its rows of the first exchange into this region (it needs
    // nothing from this rank to do so: the offset is the same everywhere), which a memset enqueued now would wipe
    // (seen once in ~100 runs of the 2-process test under CPU contention: a slab product with a zero halo).
    return MI_OK;
  }
  MI_HIP(hipMalloc((void **)ptr, bytes));
  MI_HIP(hipMemsetAsync(*ptr, 0, bytes, ctx->stream));
  return MI_OK;
}

void comm_halo_free(mi_ctx *, double *ptr, bool in_arena) {
  if (!in_arena) (void)hipFree(ptr);
}

}  // namespace mi

extern "C" {

int mi_comm_unique_id(unsigned char uid[MI_COMM_UID_BYTES]) {
  MI_REQUIRE(uid, "uid is null");
  static_assert(sizeof(ncclUniqueId) <= MI_COMM_UID_BYTES, "ncclUniqueId larger than MI_COMM_UID_BYTES");
  ncclUniqueId id;
  MI_NCCL(ncclGetUniqueId(&id));
  memset(uid, 0, MI_COMM_UID_BYTES);
  memcpy(uid, &id, sizeof(id));
  return MI_OK;
}

int mi_comm_init(mi_ctx *ctx, int world_size, int rank, const unsigned char uid[MI_COMM_UID_BYTES]) {
  MI_REQUIRE(ctx && uid, "null argument");
  MI_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size/rank %d/%d", world_size, rank);
  MI_REQUIRE(!ctx->comm, "communicator already initialised");
  MI_HIP(hipSetDevice(ctx->device));
  Comm *c = new Comm();
  ncclUniqueId id;
  memcpy(&id, uid, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->nccl, world_size, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return nccl_fail(r, "ncclCommInitRank");
  }
  MI_HIP(hipMalloc((void **)&c->scratch, sizeof(double) * 2 * (size_t)world_size + 64));
  ctx->comm = c;
  ctx->world_size = world_size;
  ctx->rank = rank;
  ctx->uniform_grid = world_size > 1 || ctx->cfg.force_uniform_grid;
  // warm the communicator (first collective builds the rings) with one tiny all-reduce
  MI_HIP(hipMemsetAsync(ctx->scalars + SLOT_MISC, 0, sizeof(double), ctx->stream));
  MI_TRY(comm_allreduce(ctx, ctx->scalars + SLOT_MISC, 1));
  MI_HIP(hipStreamSynchronize(ctx->stream));
  return MI_OK;
}

// ---- peer-memory layer: setup ---------------------------------------------------------------------
int mi_comm_ipc_export(mi_ctx *ctx, unsigned char handle[MI_COMM_IPC_HANDLE_BYTES]) {
  MI_REQUIRE(ctx && handle, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) <= MI_COMM_IPC_HANDLE_BYTES, "hipIpcMemHandle_t too large");
  Comm *c = (Comm *)ctx->comm;
  if (!c) {  // peer-memory layer without RCCL (several ranks on ONE GPU in tests: RCCL refuses that)
    c = new Comm();
    ctx->comm = c;
  }
  MI_REQUIRE(!c->arena, "arena already exported");
  if (ctx->cfg.ipc_timeout_ms > 0) c->timeout = (uint64_t)ctx->cfg.ipc_timeout_ms * 100000ull;
  c->fold = !ctx->cfg.no_fold;
  MI_HIP(hipSetDevice(ctx->device));
  MI_HIP(hipExtMallocWithFlags((void **)&c->arena, kIpcArenaBytes, hipDeviceMallocFinegrained));
  MI_HIP(hipMemset(c->arena, 0, kIpcArenaBytes));
  // (the zeros must be there before any peer can write: hipMemset may return early, and the context's stream does not
  // order itself behind the null stream)
  MI_HIP(hipDeviceSynchronize());
  hipIpcMemHandle_t h;
  MI_HIP(hipIpcGetMemHandle(&h, c->arena));
  memset(handle, 0, MI_COMM_IPC_HANDLE_BYTES);
  memcpy(handle, &h, sizeof(h));
  return MI_OK;
}

int mi_comm_ipc_attach(mi_ctx *ctx, int world_size, int rank, const unsigned char *handles) {
  MI_REQUIRE(ctx && handles, "null argument");
  MI_REQUIRE(world_size >= 1 && world_size <= kIpcMaxRanks, "the peer-memory layer supports at most %d ranks",
             kIpcMaxRanks);
  MI_REQUIRE(rank >= 0 && rank < world_size, "bad rank");
  Comm *c = (Comm *)ctx->comm;
  MI_REQUIRE(c && c->arena, "call mi_comm_ipc_export first");
  MI_REQUIRE(!c->ipc_mapped, "peer arenas already mapped");
  if (c->nccl) MI_REQUIRE(ctx->world_size == world_size && ctx->rank == rank, "rank/size differ from the communicator");
  MI_HIP(hipSetDevice(ctx->device));
  for (int r = 0; r < world_size; ++r) {
    if (r == rank) {
      c->peer[r] = c->arena;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)r * MI_COMM_IPC_HANDLE_BYTES, sizeof(h));
    void *ptr = nullptr;
    MI_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    c->peer[r] = (char *)ptr;
  }
  MI_HIP(hipMalloc((void **)&c->peer_dev, sizeof(char *) * kIpcMaxRanks));
  MI_HIP(hipMemcpy(c->peer_dev, c->peer, sizeof(char *) * kIpcMaxRanks, hipMemcpyHostToDevice));
  MI_HIP(hipHostMalloc((void **)&c->err_host, sizeof(unsigned int), hipHostMallocMapped));
  *c->err_host = 0;
  MI_HIP(hipHostGetDevicePointer((void **)&c->err_dev, c->err_host, 0));
  MI_HIP(hipMalloc((void **)&c->vals_dev, sizeof(double) * (kIpcVals + kIpcMaxRanks * kIpcVals)));
  c->ipc_mapped = true;
  ctx->world_size = world_size;
  ctx->rank = rank;
  return MI_OK;
}

// The folded exchange and the three forms of the halo push (early fold, late fold, separate kernel) between real
// peers, the form changing from round to round the way a solve mixes them: 12 rounds on a boundary that fits the
// producer's first grid-stride step (the early form's condition) and 12 on a boundary the size of cfg4's (one 200 x 200
// plane of a 3-column field per neighbour: late form and separate kernel).  The producer runs 32 workgroups whatever
// the context's grid cap is: every workgroup waits for its peers in its prologue, so with all ranks on ONE GPU (the
// rehearsals) (ranks - 1) x grid + 1 workgroups must be resident together.  Collective.  *good stays true only if every
// halo double and every folded sum of every round was right on THIS rank and no wait timed out.
static int ipc_selftest_folded_b(mi_ctx *ctx, Comm *c, size_t b, int rounds, int round0, bool *good) {
  const int P = ctx->world_size, rk = ctx->rank;
  const size_t n = (size_t)3 << 20;  // doubles per rank
  mi_csr A;                          // only the halo description of a sharded matrix (p = 1: rows = doubles)
  A.ctx = ctx;
  A.n = n;
  A.halo_lo = rk > 0 ? b : 0;
  A.halo_hi = rk + 1 < P ? b : 0;
  A.send_lo = A.halo_lo;
  A.send_hi = A.halo_hi;
  A.peer_lo_rows = rk > 1 ? b : 0;  // halo_lo of rank-1
  A.halo_stride = 2 * b;
  bool in_arena = false;
  size_t off = 0;
  double *halo = nullptr;
  MI_TRY(comm_halo_alloc(ctx, 2 * A.halo_stride * sizeof(double), &halo, &in_arena, &off));
  if (!in_arena) {  // (the arena is exhausted: nothing to test with, and nothing would fold either)
    comm_halo_free(ctx, halo, in_arena);
    return MI_OK;
  }
  A.halo = halo;
  A.halo_in_arena = true;
  A.halo_off = off;
  double2 *vec = nullptr;
  double *sums = nullptr;
  MI_TRY(pool_alloc(ctx, n * sizeof(double), (void **)&vec));
  int rc = pool_alloc(ctx, 256, (void **)&sums);
  if (rc != MI_OK) {
    pool_free(ctx, vec);
    return rc;
  }
  unsigned int *bad = reinterpret_cast<unsigned int *>(sums + 3);
  const bool was_enabled = c->ipc_enabled, was_fold = c->fold;
  c->ipc_enabled = true;
  const int grid = std::min(grid_for(ctx, n, 4), 32);
  bool ok = hipMemsetAsync(sums, 0, 256, ctx->stream) == hipSuccess;
  for (int i = 0; i < rounds && rc == MI_OK && ok; ++i) {
    const int round = round0 + i;
    const int form = was_fold ? i % 3 : 2;  // 0: folded, early where the geometry allows; 1: folded, late; 2: separate
    FoldPush fp{};
    HaloWait w{};
    double want[3];
    for (int k = 0; k < 3; ++k) {
      want[k] = 0;
      for (int r = 0; r < P; ++r) want[k] += (double)((r + 1) * (k + 1)) + (double)round;
    }
    if (form < 2) {
      c->fold = true;
      fp.f = comm_fold_next(ctx);
      const bool folded = comm_halo_fold_next(ctx, &A, 1, reinterpret_cast<const double *>(vec), &fp.h);
      // (the early form's condition for THIS kernel's grid: both boundary regions inside its first grid-stride step)
      const size_t first = (A.send_lo + A.send_hi) / 2, step = (size_t)grid * kBlock;
      fp.h.early_waves = (form == 0 && first > 0 && first <= step && step <= fp.h.n2) ? (unsigned int)((first + 63) / 64) : 0u;
      hipLaunchKernelGGL((k_ipc_selftest_producer<FoldPush>), dim3(grid), dim3(kBlock), 0, ctx->stream, n / 2, vec, rk,
                         round, fp, sums);
      if (!folded) ok = false;  // (b and n are even and in the arena: it must fold)
      rc = comm_halo_exchange_or_wait(ctx, &A, 1, reinterpret_cast<const double *>(vec), &w);
    } else {
      FoldArgs fa = comm_fold_next_always(ctx);
      hipLaunchKernelGGL((k_ipc_selftest_producer<FoldArgs>), dim3(grid), dim3(kBlock), 0, ctx->stream, n / 2, vec, rk,
                         round, fa, sums);
      rc = comm_halo_exchange(ctx, &A, 1, reinterpret_cast<const double *>(vec));
    }
    if (rc != MI_OK) break;
    hipLaunchKernelGGL(k_ipc_selftest_consumer, dim3(64), dim3(256), 0, ctx->stream, w, A.halo_cur(), A.halo_lo, A.halo_hi,
                       n, rk, round, bad);
    double got[4];
    if (hipMemcpyAsync(got, sums, sizeof(got), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      rc = MI_ERR_HIP;
      set_error("peer-memory self-test: HIP error in round %d", round);
      break;
    }
    unsigned int nbad = 0;
    memcpy(&nbad, &got[3], sizeof(nbad));  // (`bad` lives right behind sums[0..3): one copy)
    if (nbad || *c->err_host) ok = false;
    for (int k = 0; k < 3; ++k)
      if (got[k] != want[k]) ok = false;
  }
  c->fold = was_fold;
  c->ipc_enabled = was_enabled;
  c->pushed = {};
  (void)hipStreamSynchronize(ctx->stream);
  pool_free(ctx, vec);
  pool_free(ctx, sums);
  if (rc != MI_OK) return rc;
  if (!ok) *good = false;
  return MI_OK;
}
static int ipc_selftest_folded(mi_ctx *ctx, Comm *c, bool *good) {
  const size_t top0 = c->arena_top;
  int rc = ipc_selftest_folded_b(ctx, c, 16384, 12, 0, good);   // early / late / separate
  if (rc == MI_OK && *good) rc = ipc_selftest_folded_b(ctx, c, 120000, 12, 12, good);  // cfg4's boundary: late / separate
  // The test's halo regions go back to the arena, zeroed as they were handed out.  Safe: every rank has consumed its
  // last round (so every push into this rank's regions has landed), and no rank can start pushing rows of a real
  // matrix before the caller has combined the ranks' verdicts -- a collective, i.e. after every rank is through here.
  if (c->arena_top > top0) {
    hipError_t e = hipMemsetAsync(c->arena + top0, 0, c->arena_top - top0, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess && rc == MI_OK) rc = hip_fail(e, "peer-memory self-test: arena reset", __FILE__, __LINE__);
    c->arena_top = top0;
  }
  return rc;
}

// Collective: a few exchanges with known per-rank values; *ok = 1 iff every sum / gather came back
// right and no wait timed out on THIS rank.  The caller combines the ranks' verdicts (min) and passes
// the result to mi_comm_ipc_enable on every rank.
int mi_comm_ipc_selftest(mi_ctx *ctx, int *ok) {
  MI_REQUIRE(ctx && ok, "null argument");
  Comm *c = (Comm *)ctx->comm;
  MI_REQUIRE(c && c->ipc_mapped, "peer arenas not mapped");
  *ok = 0;
  const int P = ctx->world_size, rk = ctx->rank;
  bool good = true;
  for (int round = 0; round < 6 && good; ++round) {
    double in[kIpcVals], out[kIpcMaxRanks * kIpcVals];
    for (int k = 0; k < kIpcVals; ++k) in[k] = (rk + 1) * 1000.0 + round * 16 + k;
    MI_HIP(hipMemcpyAsync(c->vals_dev, in, sizeof(in), hipMemcpyHostToDevice, ctx->stream));
    const bool sum = (round % 2 == 0);
    MI_TRY(ipc_exchange(ctx, c, nullptr, 0, c->vals_dev, sum ? 16 : 4, sum, c->vals_dev + kIpcVals));
    MI_HIP(hipMemcpyAsync(out, c->vals_dev + kIpcVals, sizeof(double) * (sum ? 16 : 4 * P), hipMemcpyDeviceToHost,
                          ctx->stream));
    MI_HIP(hipStreamSynchronize(ctx->stream));
    if (*c->err_host) good = false;
    if (sum) {
      for (int k = 0; k < kIpcVals && good; ++k) {
        double want = 0;
        for (int r = 0; r < P; ++r) want += (r + 1) * 1000.0 + round * 16 + k;
        if (out[k] != want) good = false;
      }
    } else {
      for (int r = 0; r < P && good; ++r)
        for (int k = 0; k < 4; ++k)
          if (out[r * 4 + k] != (r + 1) * 1000.0 + round * 16 + k) good = false;
    }
  }
  if (good && P > 1) MI_TRY(ipc_selftest_folded(ctx, c, &good));
  *ok = good ? 1 : 0;
  return MI_OK;
}

int mi_comm_ipc_enable(mi_ctx *ctx, int on) {
  MI_REQUIRE(ctx, "ctx is null");
  Comm *c = (Comm *)ctx->comm;
  MI_REQUIRE(c && c->ipc_mapped, "peer arenas not mapped");
  MI_REQUIRE(on || c->nccl, "no RCCL communicator to fall back to");
  c->ipc_enabled = on != 0;
  if (!c->ipc_enabled && c->err_host) *c->err_host = 0;  // a failed layer must not poison the RCCL path
  // slot path (peer-memory): rows never cross ranks; RCCL rows mode needs the same row count everywhere
  ctx->uniform_grid = !c->ipc_enabled && (ctx->world_size > 1 || ctx->cfg.force_uniform_grid);
  return MI_OK;
}

// Exchanges folded into their producer / consumer kernels (comm_ipc.h) on or off at run time; the same on every rank.
int mi_comm_ipc_fold(mi_ctx *ctx, int on) {
  MI_REQUIRE(ctx, "ctx is null");
  Comm *c = (Comm *)ctx->comm;
  MI_REQUIRE(c, "no communicator");
  c->fold = on != 0;
  c->pushed = {};
  return MI_OK;
}

// 0: no error; nonzero: a wait inside the peer-memory layer timed out (results since then are invalid)
int mi_comm_ipc_error(mi_ctx *ctx, int *err) {
  MI_REQUIRE(ctx && err, "null argument");
  Comm *c = (Comm *)ctx->comm;
  *err = (c && c->ipc_enabled && c->err_host) ? (int)*c->err_host : 0;
  return MI_OK;
}

int mi_comm_kernel_launches(mi_ctx *ctx, unsigned long long out[4]) {
  MI_REQUIRE(ctx && out, "null argument");
  Comm *c = (Comm *)ctx->comm;
  for (int i = 0; i < 4; ++i) out[i] = c ? c->launched[i] : 0ull;
  return MI_OK;
}

int mi_comm_finalize(mi_ctx *ctx) {
  if (!ctx || !ctx->comm) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  (void)hipStreamSynchronize(ctx->stream);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  for (int r = 0; r < kIpcMaxRanks; ++r)
    if (c->peer[r] && c->peer[r] != c->arena) (void)hipIpcCloseMemHandle(c->peer[r]);
  if (c->arena) (void)hipFree(c->arena);
  if (c->peer_dev) (void)hipFree(c->peer_dev);
  if (c->err_host) (void)hipHostFree(c->err_host);
  if (c->vals_dev) (void)hipFree(c->vals_dev);
  (void)hipFree(c->scratch);
  delete c;
  ctx->comm = nullptr;
  ctx->world_size = 1;
  ctx->rank = 0;
  ctx->uniform_grid = false;
  return MI_OK;
}

int mi_debug_set_rank(mi_ctx *ctx, int world_size, int rank) {
  MI_REQUIRE(ctx, "ctx is null");
  MI_REQUIRE(!ctx->comm, "a communicator is attached: rank and size come from it");
  MI_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bad world_size/rank %d/%d", world_size, rank);
  ctx->world_size = world_size;
  ctx->rank = rank;
  return MI_OK;
}

int mi_debug_csr_set_halo(mi_csr *A, int p, const double *halo_rows_host) {
  MI_REQUIRE(A && halo_rows_host, "null argument");
  MI_REQUIRE(p >= 1 && p <= kMaxP, "halo buffers hold at most %d columns", kMaxP);
  const size_t rows = A->halo_lo + A->halo_hi;
  if (rows)
    MI_HIP(hipMemcpy(const_cast<double *>(A->halo_cur()), halo_rows_host, rows * (size_t)p * sizeof(double), hipMemcpyHostToDevice));
  return MI_OK;
}

int mi_comm_rccl_count(mi_ctx *ctx, int *nranks) {
  MI_REQUIRE(ctx && nranks, "null argument");
  *nranks = 0;
  Comm *c = (Comm *)ctx->comm;
  if (c && c->nccl) MI_NCCL(ncclCommCount(c->nccl, nranks));
  return MI_OK;
}

int mi_comm_info(mi_ctx *ctx, int *world_size, int *rank) {
  MI_REQUIRE(ctx, "ctx is null");
  if (world_size) *world_size = ctx->world_size;
  if (rank) *rank = ctx->rank;
  return MI_OK;
}

}  // extern "C"
