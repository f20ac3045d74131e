// comm_ipc.h -- device side of the peer-memory exchange layer (comm.hip), shared with the kernels that FOLD an
// exchange into their own prologue (stpcg.hip): mailbox layout, bounded flag wait, and the folded all-reduce.
#pragma once

#include "mi_internal.h"

namespace mi {

constexpr int kIpcMaxRanks = 8;
constexpr int kIpcRing = 4;            // mailbox slots in flight (2 would do: a rank is never >1 exchange ahead)
constexpr int kIpcVals = 16;           // doubles per rank per exchange
constexpr size_t kIpcMailboxBytes = 64 * 1024;

struct IpcMailbox {  // lives at offset 0 of every arena
  uint64_t flag[kIpcRing][kIpcMaxRanks];             // flag[q][r] == seq: rank r's values of exchange seq are in
  double val[kIpcRing][kIpcMaxRanks][kIpcVals];
  uint64_t halo_flag[2];                              // [0]: from rank-1, [1]: from rank+1
  unsigned int halo_count;                            // "last workgroup" counter of k_ipc_halo_push
  unsigned int pad;
};
static_assert(sizeof(IpcMailbox) <= kIpcMailboxBytes, "mailbox too large");

// What a consumer kernel needs to complete a sum over the ranks in its own prologue (peers == nullptr: no exchange).
struct FoldArgs {
  char *const *peers = nullptr;  // device array of the mapped arenas, peers[rank] = mine
  unsigned int *err = nullptr;   // sticky error word (a bounded wait timed out)
  uint64_t seq = 0;              // sequence number of THIS exchange (the same on every rank)
  uint64_t timeout = 0;          // bound of the wait, 100 MHz ticks
  int P = 1, rank = 0;
};

// Once ANY wait of this rank has timed out the error word is sticky and every later wait gives up after 1 ms instead
// of its full bound: a broken exchange fails the solve in about one timeout, not in (exchanges left) x timeout.
__device__ __forceinline__ bool ipc_give_up(uint64_t waited, unsigned int *err, uint64_t timeout) {
  if (waited > timeout) {
    __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  return waited > 100000ull && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
}
__device__ __forceinline__ bool ipc_wait(const uint64_t *flag, uint64_t want, unsigned int *err, uint64_t timeout) {
  const uint64_t t0 = wall_clock64();
  // sequence numbers only grow: ">= want", so that a peer that is already one exchange further on (it may raise
  // this flag again before a descheduled waiter has looked) still releases the wait
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (ipc_give_up(wall_clock64() - t0, err, timeout)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}
// The same without the acquire's cache invalidation, for waits inside a streaming kernel.  What is read afterwards are
// the mailbox VALUES, and only with system-scope atomic loads (fold_exchange_sum): those are issued behind this loop
// (program order; from the other threads behind the workgroup barrier that follows it) and are served by memory, not
// by a cache, so there is no stale line an acquire would have to invalidate -- the peer wrote the values before its
// release of the flag.  A formal acquire was tried (r04, ADVICE: one system-scope acquire fence per workgroup behind
// the successful poll = `buffer_inv sc0 sc1`): it invalidates the XCD's L2 under the streaming kernel that is running,
// measured on the sharded step at one rank k_cg_update 17.3 -> 20.1 us and k_cg_pupdate 21.9 -> 25.4 us, the step
// 60.8 -> 66.1 us, more than the separate exchange kernels cost (63.9).  Not kept; the cross-device behaviour of this
// form is what the self-test of mi_comm_ipc_selftest checks between the real peers before the layer is enabled.
__device__ __forceinline__ bool ipc_wait_relaxed(const uint64_t *flag, uint64_t want, unsigned int *err,
                                                 uint64_t timeout) {
  const uint64_t t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (ipc_give_up(wall_clock64() - t0, err, timeout)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
// Every wave, behind its last peer store and in front of the workgroup barrier that precedes the ONE system-scope
// release of its workgroup: wait until this wave's own vector-memory operations have completed.  A workgroup barrier
// does not do that (one CU per workgroup: the compiler puts `s_waitcnt lgkmcnt(0)` only in front of s_barrier), and the
// release thread 0 issues behind the barrier drains wave 0's queue, nobody else's -- on a real xGMI link the other
// waves' rows could still be in flight when the flag goes up (r04, ADVICE).  No cache write-back involved (that is what
// made one release PER WAVE expensive): once every wave's stores have reached the L2 the single write-back of thread 0,
// which runs on the same XCD, covers them all.
__device__ __forceinline__ void wave_stores_done() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // compiler: no store sinks below this point
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0); expcnt / lgkmcnt untouched (gfx9 encoding)
}

// Folded all-reduce: every thread of every workgroup holds the SAME local sums d[0..K) (the prologue re-reduction of
// the partial rows).  Workgroup 0 pushes them into every rank's mailbox (thread t serves rank t: K relaxed stores, then
// the sequence-numbered flag with release); every workgroup waits for the P flags in ITS OWN rank's mailbox and sums
// the P contributions in rank order -- identical code on identical data on every workgroup of every rank: identical
// bits everywhere, no separate exchange kernel.  lds: >= kIpcVals doubles.  Contains barriers.
#ifdef MI_FOLD_STAMPS  // experiment builds: where the folded exchange spends its time (8 stamps per workgroup)
static __device__ unsigned long long *g_fold_stamps = nullptr;
#define FOLD_STAMP(i) do { if (g_fold_stamps && threadIdx.x == 0) g_fold_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define FOLD_STAMP_K(i) do { if (K > 1) FOLD_STAMP(i); } while (0)  // (k_cg_update's exchange only)
#else
#define FOLD_STAMP_K(i) do { } while (0)
#define FOLD_STAMP(i) do { } while (0)
#endif
template <int K>
__device__ __forceinline__ void fold_exchange_sum(double (&d)[K], const FoldArgs &f, double *lds) {
  static_assert(K <= kIpcVals, "exchange width");
  const int q = (int)(f.seq % kIpcRing);
  const int t = threadIdx.x;
  IpcMailbox *mine = reinterpret_cast<IpcMailbox *>(f.peers[f.rank]);
  FOLD_STAMP_K(1);
  if (blockIdx.x == 0 && t < f.P) {
    IpcMailbox *mb = reinterpret_cast<IpcMailbox *>(f.peers[t]);
#pragma unroll
    for (int k = 0; k < K; ++k)
      __hip_atomic_store(&mb->val[q][f.rank][k], d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->flag[q][f.rank], f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  FOLD_STAMP_K(2);
  if (t < f.P) ipc_wait_relaxed(&mine->flag[q][t], f.seq, f.err, f.timeout);
  FOLD_STAMP_K(3);
  __syncthreads();
  FOLD_STAMP_K(4);
  if (t < K) {
    double s = 0;
    for (int r = 0; r < f.P; ++r)
      s += __hip_atomic_load(&mine->val[q][r][t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    lds[t] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) d[k] = lds[k];
  __syncthreads();
  FOLD_STAMP_K(5);
}

// The kernels that can fold an exchange are instantiated twice: with FoldArgs (several ranks, peer-memory layer) and
// with NoFold, an EMPTY parameter whose fold_maybe() is nothing at all -- the single-GPU kernels must not carry the
// extra kernel arguments and the branch (measured: +3 us on k_cg_update when they did).
struct NoFold {};
template <int K>
__device__ __forceinline__ void fold_maybe(double (&)[K], const NoFold &, double *) {}
template <int K>
__device__ __forceinline__ void fold_maybe(double (&d)[K], const FoldArgs &f, double *lds) {
  // (wider sums -- the Stiefel components of p >= 5 -- are never folded: the host keeps the separate exchange, stpcg.hip)
  if constexpr (K <= kIpcVals) {
    if (f.peers) fold_exchange_sum<K>(d, f, lds);
  }
}
struct FoldPush;
template <int K>
__device__ __forceinline__ void fold_maybe(double (&d)[K], const FoldPush &fp, double *lds);

// The HALO exchange folded the same way (STPCG's recurrence form over a sharded Stiefel problem): the PUSH half rides
// in the kernel that writes the next direction -- k_cg_pupdate stores every element a neighbour needs a second time,
// into that neighbour's halo buffer, and the last workgroup to finish raises the neighbours' flags -- and the WAIT half
// sits in the prologue of the Hessian pass that reads the halo.  Same flags, sequence numbers and double buffering as
// k_ipc_halo_push (comm.hip), so a rank whose matrix does not qualify keeps the separate kernel next to ranks that
// fold.  Buffer reuse: between two pushes every rank passes a folded scalar exchange (k_cg_update), which no rank leaves
// before every rank has entered it, i.e. has finished the Hessian pass that read the buffer written two pushes ago.
struct HaloPush {
  double2 *dst_lo = nullptr, *dst_hi = nullptr;  // rank-1's / rank+1's halo buffer of this exchange (null: no rows)
  size_t lo2 = 0, hi_from = 0;                   // double2 [0, lo2) go to dst_lo, [hi_from, n2) to dst_hi
  IpcMailbox *mine = nullptr, *mb_lo = nullptr, *mb_hi = nullptr;  // flags to raise (null: pair not active)
  uint64_t seq = 0;                              // 0: nothing folded into this launch
  // EARLY form (early_waves > 0): the pushing kernel walks its vector rotated by `rot` double2 (logical element L is
  // physical element (L + rot) mod n2, rot = hi_from), so that the rows the neighbours need -- the LAST hi rows, then
  // the FIRST lo rows -- are the first thing its first grid-stride step touches (in `early_waves` 64-element chunks);
  // the workgroups that hold them signal as soon as that step is stored and fenced, two thirds of the kernel before
  // its end, and the xGMI transfer hides behind the rest of the kernel instead of standing between it and the
  // neighbour's Hessian pass.
  size_t rot = 0, n2 = 0;
  unsigned int early_waves = 0;
};
struct FoldPush {  // a folded scalar exchange in the prologue AND a folded halo push in the body (k_cg_pupdate)
  FoldArgs f;
  HaloPush h;
};
struct HaloWait {
  const IpcMailbox *mine = nullptr;
  unsigned int *err = nullptr;
  uint64_t seq = 0, timeout = 0;  // seq 0: the halo is already in (separate exchange kernel, or not sharded)
  int expect_lo = 0, expect_hi = 0;
};
template <bool HALO>
struct HaloWaitArg {};  // kernels without a halo carry no argument for it
template <>
struct HaloWaitArg<true> {
  HaloWait w;
  unsigned int halo_lo = 0, halo_hi = 0;  // the matrix's halo extents (computed far columns: spmm_core.h load_far)
};

__device__ __forceinline__ bool halo_push_store(const NoFold &, size_t, const double2 &) { return false; }
__device__ __forceinline__ bool halo_push_store(const FoldArgs &, size_t, const double2 &) { return false; }
__device__ __forceinline__ bool halo_push_store(const FoldPush &fp, size_t i, const double2 &v) {
  const HaloPush &h = fp.h;
  bool any = false;
  if (i < h.lo2) { h.dst_lo[i] = v; any = true; }
  if (h.dst_hi && i >= h.hi_from) { h.dst_hi[i - h.hi_from] = v; any = true; }
  return any;
}
// where logical element L of the pushing kernel's walk lives (L itself for every kernel but the early form)
__device__ __forceinline__ size_t halo_phys(const NoFold &, size_t L) { return L; }
__device__ __forceinline__ size_t halo_phys(const FoldArgs &, size_t L) { return L; }
__device__ __forceinline__ size_t halo_phys(const FoldPush &fp, size_t L) {
  if (!fp.h.early_waves) return L;
  const size_t q = L + fp.h.rot;
  return q >= fp.h.n2 ? q - fp.h.n2 : q;
}
// early form: every WORKGROUP, once, right behind the stores of its first grid-stride step (all of its threads are in
// that step: the host takes the early form only when the vector is at least one step long).  The workgroups that hold
// the neighbours' rows count themselves and the last of them raises the flags.  One system-scope release per workgroup,
// behind a barrier (the other waves' stores happen before it), not one per wave: such a release writes the whole L2
// back, in the middle of a kernel that dirties 48 MB -- per wave (~470 of them) it cost the 2-rank rehearsal 6 % and
// the 4-rank one 10 % per step.  Contains a barrier (uniform: every thread of the workgroup calls it or none does).
__device__ __forceinline__ void halo_push_first_step_done(const NoFold &) {}
__device__ __forceinline__ void halo_push_first_step_done(const FoldArgs &) {}
__device__ __forceinline__ void halo_push_first_step_done(const FoldPush &fp) {
  const HaloPush &h = fp.h;
  if (!h.seq || !h.early_waves) return;
  // (early_waves counts 64-element chunks: this workgroup holds some of them iff its first chunk is one)
  const unsigned int wpb = blockDim.x >> 6, first_chunk = blockIdx.x * wpb;
  if (first_chunk >= h.early_waves) return;
  wave_stores_done();
  __syncthreads();
  if (threadIdx.x != 0) return;
  __threadfence_system();  // (replaced by a plain wait for the stores -- they are uncached peer stores -- nothing changes)
  const unsigned int nblocks = (h.early_waves + wpb - 1) / wpb;
  const unsigned int done = __hip_atomic_fetch_add(&h.mine->halo_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (done != nblocks - 1) return;
  __hip_atomic_store(&h.mine->halo_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (h.mb_lo) __hip_atomic_store(&h.mb_lo->halo_flag[1], h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (h.mb_hi) __hip_atomic_store(&h.mb_hi->halo_flag[0], h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// every workgroup, once, behind its last halo_push_store (pushed: this thread stored something).  Contains a barrier.
// (Nothing in the early form: its workgroups have signalled already.)
__device__ __forceinline__ void halo_push_finish(const NoFold &, bool) {}
__device__ __forceinline__ void halo_push_finish(const FoldArgs &, bool) {}
__device__ __forceinline__ void halo_push_finish(const FoldPush &fp, bool pushed) {
  const HaloPush &h = fp.h;
  if (!h.seq || h.early_waves) return;
  (void)pushed;
  wave_stores_done();
  __syncthreads();
  if (threadIdx.x != 0) return;
  __threadfence_system();  // (one release per workgroup behind the barrier, as in halo_push_first_step_done)
  const unsigned int done = __hip_atomic_fetch_add(&h.mine->halo_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (done != gridDim.x - 1) return;
  __hip_atomic_store(&h.mine->halo_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (h.mb_lo) __hip_atomic_store(&h.mb_lo->halo_flag[1], h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (h.mb_hi) __hip_atomic_store(&h.mb_hi->halo_flag[0], h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// prologue of the kernel that reads the halo.  Contains a barrier.
__device__ __forceinline__ void halo_wait(const HaloWait &w) {
  if (!w.seq) return;
  if (threadIdx.x == 0) {
    if (w.expect_lo) ipc_wait(&w.mine->halo_flag[0], w.seq, w.err, w.timeout);
    if (w.expect_hi) ipc_wait(&w.mine->halo_flag[1], w.seq, w.err, w.timeout);
  }
  __syncthreads();
}

template <int K>
__device__ __forceinline__ void fold_maybe(double (&d)[K], const FoldPush &fp, double *lds) {
  if constexpr (K <= kIpcVals) {
    if (fp.f.peers) fold_exchange_sum<K>(d, fp.f, lds);
  }
}

// host (comm.hip): fold the NEXT halo exchange of the n x p field `V` over A's pattern.  True: `push` goes to the
// kernel that writes V (launched next on the stream), and the next comm_halo_exchange_or_wait(ctx, A, p, V, &w)
// launches nothing and hands out the wait instead.  False: nothing changed; the exchange stays a kernel.
bool comm_halo_fold_next(mi_ctx *ctx, const mi_csr *A, int p, const double *V, HaloPush *push);
// comm_halo_exchange for a kernel that can wait in its own prologue: *w is the wait of a folded push of exactly this
// (A, p, V), or seq 0 after launching the separate exchange kernel (or when nothing is exchanged at all)
int comm_halo_exchange_or_wait(mi_ctx *ctx, const mi_csr *A, int p, const double *V, HaloWait *w);
// forget a folded push nobody consumed (the solve ended behind it; the next exchange is a whole one again)
void comm_halo_fold_drop(mi_ctx *ctx);
// ---- r'-halo form (Config::halo_rprime; IterativeSolvers.h:377,408-420) ------------------------------------------------
// The halo rows of the next direction p' = -r' + beta p (:420) can only travel once beta is known, i.e. behind the SECOND
// all-reduce of an iteration: three dependent collectives per iteration on the RCCL layer.  r' (:377) is final one kernel
// earlier: its boundary rows can ride in the same RCCL group as the all-reduce of <r',v'> (:408), and every rank then forms
// halo(p') = -halo(r') + beta halo(p) from the halo rows of p it already holds -- the SAME expression, on the same bits,
// as the owner of those rows evaluates for them (k_cg_pupdate), so the result is bit-identical to exchanging p'.
// usable for this matrix on this context right now? (several ranks, a halo, the switch on)
int comm_rprime_prepare(mi_ctx *ctx, const struct mi_csr *A, bool *enabled);
// all-reduce of the k component rows of `partials` (rows mode) AND the boundary rows of the n x p field R into A's r'-halo
// buffer: one RCCL group; with the peer-memory layer's separate kernels, the scalar exchange is the caller's and this
// only pushes the rows (partials == null)
int comm_rprime_exchange(mi_ctx *ctx, const struct mi_csr *A, int p, const double *R, double *partials, int k);
// the buffers k_halo_dir combines: the r' rows just received, the halo of p the next Hessian pass will read, and their
// length in doubles
void comm_rprime_buffers(const mi_ctx *ctx, const struct mi_csr *A, int p, const double **halo_r, double **halo_p,
                         size_t *count);
// the halo of the n x p field V over A's pattern is in place (formed locally): the next comm_halo_exchange_or_wait of
// exactly (A, p, V) exchanges nothing
void comm_rprime_mark(mi_ctx *ctx, const struct mi_csr *A, int p, const double *V);

// host: arguments of the next folded exchange on this context (comm.hip).  peers == nullptr when the peer-memory
// layer is not carrying the exchanges or folding is switched off (MI355OPT_NO_FOLD=1): separate exchange kernels.
FoldArgs comm_fold_next(mi_ctx *ctx);
// the same whenever the peer-memory layer carries the exchanges, MI355OPT_NO_FOLD or not (solvers that have no
// separate-kernel variant of their exchanges: the fused LSQR)
FoldArgs comm_fold_next_always(mi_ctx *ctx);

}  // namespace mi
