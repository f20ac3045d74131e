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
// The same without the acquire's cache invalidation, for waits inside a streaming kernel: what is read afterwards
// (the mailbox values) is read with system-scope atomic loads, which do not come out of a cache, and loads issue and
// return in program order behind the flag load.
__device__ __forceinline__ bool ipc_wait_relaxed(const uint64_t *flag, uint64_t want, unsigned int *err,
                                                 uint64_t timeout) {
  const uint64_t t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (ipc_give_up(wall_clock64() - t0, err, timeout)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

// Folded all-reduce: every thread of every workgroup holds the SAME local sums d[0..K) (the prologue re-reduction of
// the partial rows).  Workgroup 0 pushes them into every rank's mailbox (thread t serves rank t: K relaxed stores, then
// the sequence-numbered flag with release); every workgroup waits for the P flags in ITS OWN rank's mailbox and sums
// the P contributions in rank order -- identical code on identical data on every workgroup of every rank: identical
// bits everywhere, no separate exchange kernel.  lds: >= kIpcVals doubles.  Contains barriers.
template <int K>
__device__ __forceinline__ void fold_exchange_sum(double (&d)[K], const FoldArgs &f, double *lds) {
  static_assert(K <= kIpcVals, "exchange width");
  const int q = (int)(f.seq % kIpcRing);
  const int t = threadIdx.x;
  IpcMailbox *mine = reinterpret_cast<IpcMailbox *>(f.peers[f.rank]);
  if (blockIdx.x == 0 && t < f.P) {
    IpcMailbox *mb = reinterpret_cast<IpcMailbox *>(f.peers[t]);
#pragma unroll
    for (int k = 0; k < K; ++k)
      __hip_atomic_store(&mb->val[q][f.rank][k], d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&mb->flag[q][f.rank], f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t < f.P) ipc_wait_relaxed(&mine->flag[q][t], f.seq, f.err, f.timeout);
  __syncthreads();
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
}

// The kernels that can fold an exchange are instantiated twice: with FoldArgs (several ranks, peer-memory layer) and
// with NoFold, an EMPTY parameter whose fold_maybe() is nothing at all -- the single-GPU kernels must not carry the
// extra kernel arguments and the branch (measured: +3 us on k_cg_update when they did).
struct NoFold {};
template <int K>
__device__ __forceinline__ void fold_maybe(double (&)[K], const NoFold &, double *) {}
template <int K>
__device__ __forceinline__ void fold_maybe(double (&d)[K], const FoldArgs &f, double *lds) {
  if (f.peers) fold_exchange_sum<K>(d, f, lds);
}

// host: arguments of the next folded exchange on this context (comm.hip).  peers == nullptr when the peer-memory
// layer is not carrying the exchanges or folding is switched off (MI355OPT_NO_FOLD=1): separate exchange kernels.
FoldArgs comm_fold_next(mi_ctx *ctx);
// the same whenever the peer-memory layer carries the exchanges, MI355OPT_NO_FOLD or not (solvers that have no
// separate-kernel variant of their exchanges: the fused LSQR)
FoldArgs comm_fold_next_always(mi_ctx *ctx);

}  // namespace mi
