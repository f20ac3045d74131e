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

#include "mi_internal.h"

using namespace mi;

namespace {
struct Comm {
  ncclComm_t nccl = nullptr;
  double *scratch = nullptr;  // device, small all-gather buffer
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
}  // namespace

namespace mi {

int comm_allreduce(mi_ctx *ctx, double *buf, int count) {
  // a communicator of size 1 still goes through RCCL: the single-GPU box then exercises exactly the
  // calls the 8-GPU node makes (tests/test_gpu_comm.py)
  if (!ctx->comm) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  MI_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->nccl, ctx->stream));
  return MI_OK;
}

int comm_allreduce_rows(mi_ctx *ctx, double *partials, int k) {
  return comm_allreduce(ctx, partials, k * kMaxRows);
}

int comm_halo_exchange(mi_ctx *ctx, const mi_csr *A, int p, const double *V) {
  if (ctx->world_size <= 1 || !ctx->comm) return MI_OK;
  if (A->halo_lo + A->halo_hi + A->send_lo + A->send_hi == 0) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  const int rk = ctx->rank, ws = ctx->world_size;
  MI_NCCL(ncclGroupStart());
  if (rk > 0) {
    if (A->send_lo) MI_NCCL(ncclSend(V, A->send_lo * p, ncclDouble, rk - 1, c->nccl, ctx->stream));
    if (A->halo_lo) MI_NCCL(ncclRecv(A->halo, A->halo_lo * p, ncclDouble, rk - 1, c->nccl, ctx->stream));
  }
  if (rk + 1 < ws) {
    if (A->send_hi)
      MI_NCCL(ncclSend(V + (A->n - A->send_hi) * p, A->send_hi * p, ncclDouble, rk + 1, c->nccl,
                       ctx->stream));
    if (A->halo_hi)
      MI_NCCL(ncclRecv(A->halo + A->halo_lo * p, A->halo_hi * p, ncclDouble, rk + 1, c->nccl,
                       ctx->stream));
  }
  MI_NCCL(ncclGroupEnd());
  return MI_OK;
}

// Every rank learns how many rows its neighbours need from it: all-gather of (need_lo, need_hi).
int comm_exchange_halo_counts(mi_ctx *ctx, size_t need_lo, size_t need_hi, size_t *send_lo,
                              size_t *send_hi) {
  *send_lo = *send_hi = 0;
  if (ctx->world_size <= 1 || !ctx->comm) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  const int ws = ctx->world_size, rk = ctx->rank;
  std::vector<double> host(2 * (size_t)ws, 0.0);
  double mine[2] = {(double)need_lo, (double)need_hi};
  MI_HIP(hipMemcpyAsync(c->scratch + 2 * rk, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
  MI_NCCL(ncclAllGather(c->scratch + 2 * rk, c->scratch, 2, ncclDouble, c->nccl, ctx->stream));
  MI_HIP(hipMemcpyAsync(host.data(), c->scratch, host.size() * sizeof(double), hipMemcpyDeviceToHost,
                        ctx->stream));
  MI_HIP(hipStreamSynchronize(ctx->stream));
  if (rk > 0) *send_lo = (size_t)host[2 * (rk - 1) + 1];       // rank-1 needs rows above its range
  if (rk + 1 < ws) *send_hi = (size_t)host[2 * (rk + 1) + 0];  // rank+1 needs rows below its range
  return MI_OK;
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
  g_uniform_grid = world_size > 1 || getenv("MI355OPT_FORCE_UNIFORM_GRID") != nullptr;
  // warm the communicator (first collective builds the rings) with one tiny all-reduce
  MI_HIP(hipMemsetAsync(ctx->scalars + SLOT_MISC, 0, sizeof(double), ctx->stream));
  MI_TRY(comm_allreduce(ctx, ctx->scalars + SLOT_MISC, 1));
  MI_HIP(hipStreamSynchronize(ctx->stream));
  return MI_OK;
}

int mi_comm_finalize(mi_ctx *ctx) {
  if (!ctx || !ctx->comm) return MI_OK;
  Comm *c = (Comm *)ctx->comm;
  (void)hipStreamSynchronize(ctx->stream);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  (void)hipFree(c->scratch);
  delete c;
  ctx->comm = nullptr;
  ctx->world_size = 1;
  ctx->rank = 0;
  g_uniform_grid = false;
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
  MI_REQUIRE(p >= 1 && p <= 4, "halo buffers hold at most 4 columns");
  const size_t rows = A->halo_lo + A->halo_hi;
  if (rows)
    MI_HIP(hipMemcpy(A->halo, halo_rows_host, rows * (size_t)p * sizeof(double), hipMemcpyHostToDevice));
  return MI_OK;
}

int mi_comm_info(mi_ctx *ctx, int *world_size, int *rank) {
  MI_REQUIRE(ctx, "ctx is null");
  if (world_size) *world_size = ctx->world_size;
  if (rank) *rank = ctx->rank;
  return MI_OK;
}

}  // extern "C"
