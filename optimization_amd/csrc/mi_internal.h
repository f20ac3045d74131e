// mi_internal.h -- shared internals of libmi355opt.so (not installed; the public surface is
// include/mi355opt.h).  gfx950 only: wave = 64 lanes, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <set>
#include <vector>

#include "mi355opt.h"

namespace mi {

// Launch geometry of the streaming kernels.  Workgroups are FAT (1024 threads = 16 waves) and FEW
// (<= 512 = 2 per CU, i.e. all 32 wave slots of every CU) on purpose: every global reduction is a
// per-workgroup partial row that the NEXT kernel's workgroups all re-reduce in their prologue
// (deterministic, no atomics, no separate one-workgroup kernel -- a dependent one-workgroup kernel
// costs ~10 us on MI355X: boundary + cross-XCD read of the partials).  Few rows keep that prologue
// at a few KB per workgroup.
constexpr int kWave = 64;
constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / kWave;  // 16
constexpr int kMaxP = 8;                // widest Stiefel / tall-skinny field the templated kernels are instantiated for
constexpr int kMaxGrid = 512;
constexpr int kMaxRows = 1024;          // partial rows per component (the streaming kernels leave <= kMaxGrid; the
                                        // window kernels, with their smaller workgroups, up to kMaxRows)
constexpr int kMaxComps = 64;           // components per partial buffer (component-major layout): 3 curvature dots + the 36
                                        // packed entries of an 8 x 8 symmetric Gram (Stiefel p = 8); the raw 8 x 8 Gram of mi_stiefel_gram: 64
constexpr int kNumXCD = 8;
constexpr int kScalarSlots = 256;       // device scalar file (doubles)

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define MI_HIP(expr)                                                   \
  do {                                                                 \
    hipError_t _e = (expr);                                            \
    if (_e != hipSuccess) return ::mi::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define MI_TRY(expr)          \
  do {                        \
    int _s = (expr);          \
    if (_s != MI_OK) return _s; \
  } while (0)

#define MI_REQUIRE(cond, ...)              \
  do {                                     \
    if (!(cond)) {                         \
      ::mi::set_error(__VA_ARGS__);        \
      return MI_ERR_INVALID_ARGUMENT;      \
    }                                      \
  } while (0)

// Host-visible progress word written by the device (fine-grained pinned memory).
struct HostStatus {
  // (B-step launches that did work in the current solve) << 1 | done.  ONE word, written with one store,
  // so the host can never see "done" without the launch count it belongs to: with several ranks the
  // number of speculatively enqueued iterations (each carries collectives) is derived from it and must
  // be the same on every rank.
  volatile uint64_t word;
  volatile uint32_t epoch;
};

// Device-resident state of one STPCG solve (IterativeSolvers.h:259-283).  Two copies ping-pong:
// a kernel reads st[i] and its workgroup 0 writes st[i^1] (no intra-kernel read/write race).
struct CgState {
  double sk_M_pk, sk_M_2, pk_M_2;  // :259,263,266
  double target_rk_norm;           // :278 (depends on the initial residual: computed on the device)
  double rv;                       // current <r,v>
  double alpha, beta, kappa, sigma;
  double skplus1_M_2;              // :344, carried from the A-step to the B-step
  double M_norm;                   // update_step_M_norm
  unsigned long long k;            // num_iterations
  unsigned long long launches;     // B-step launches that found the solve still running (progress word)
  int mode;                        // CgMode
  int exit_reason;
};
static_assert(sizeof(CgState) <= 128, "the CG state is meant to fit one 128-byte line");
// Per-solve constants known on the host travel as kernel arguments (SGPRs), not through the state.
struct CgConst {
  double Delta, Delta_2, epsilon;  // :171,271,179
  unsigned long long max_iterations;
  // the walk of the two streaming kernels over their n/2 double2 elements (stpcg_kernels.inc): whole grid-stride steps
  // up to rag0, then the ragged rest in one contiguous piece of rag_per elements per workgroup
  unsigned long long rag0, rag_per;
};
enum CgMode { CG_RUN = 0, CG_KERNEL_PENDING = 1, CG_APPLY_SIGMA = 2, CG_DONE = 3 };

// Every MI355OPT_* switch of the library.  Parsed ONCE per context, in mi_ctx_create (context.hip config_from_env):
// nothing on an enqueue path reads the environment.  mi_ctx_set_option changes one of them on a live context (the A/B
// tests that compare two forms of a kernel inside one process); matrix-format switches (no_packed) act when a
// matrix is created.
struct Config {
  bool force_uniform_grid = false;  // FORCE_UNIFORM_GRID: kMaxGrid workgroups in every reduction-producing kernel
  long ipc_timeout_ms = 0;          // IPC_TIMEOUT_MS: bound of the peer-memory layer's waits (0: the default, 20 s)
  bool no_fold = false;             // NO_FOLD: separate exchange kernels instead of the folded forms (comm_ipc.h)
  bool halo_push_late = false;      // HALO_PUSH_LATE: the folded halo push signals at the END of the direction kernel
  bool no_packed = false;           // NO_PACKED: no value-indexed 4-byte copy of a matrix (generic 12-byte path)
  bool no_window = false;           // NO_WINDOW: the one-pass Hessian in streaming form
  bool no_win_bounds = false;       // NO_WIN_BOUNDS: equal runs instead of the planned ones (window kernels)
  bool no_far_computed = false;     // NO_FAR_COMPUTED: far columns loaded from wfar
  bool words16 = false;             // WORDS16: 16-bit window words (opt-in)
  bool no_spmm_stream = false;      // NO_SPMM_STREAM: the straight sparse core instead of the pipelined one
  bool no_spmm_sweep = true;        // NO_SPMM_SWEEP: the LOBPCG panel product of a pure-far-structure matrix keeps the window
                                    // form (r06: the plane-sweep form is opt-in until it measures faster: NO_SPMM_SWEEP=0)
  int sweep_zsegs = 0;              // SWEEP_ZSEGS: z-segments of the plane-sweep product (0: chosen by the launcher)
  bool no_spmm_win = false;         // NO_SPMM_WIN: the LOBPCG panel product in gather form
  bool no_zero_copy = false;        // NO_ZERO_COPY: LOBPCG Gram / residual results through a device buffer + read-back
  int wide_quad = -1;               // WIDE_QUAD: the one-pass Hessian of rows of 5 ... 8 doubles in the quad layout (4 lanes per
                                    // row): -1 = where it measured faster (p = 8), 0 = never (one lane per row), 1 = always
  int wide_window = -1;             // WIDE_WINDOW: the one-pass Hessian of rows of 5 ... 8 doubles in WINDOW form (r06): -1 = p <= 7
                                    // when the matrix has a window, 0 = never, 1 = every width
  bool no_polled_sync = false;      // NO_POLLED_SYNC: stream_wait is hipStreamSynchronize (no flag kernel + host poll)
  bool two_kernel_step = false;     // TWO_KERNEL_STEP: opt-in experiment (r05): <r+,r+> by recurrence, the two CG kernels of an
                                    // unpreconditioned Stiefel(n,3) iteration merged (changes the rounding of IterativeSolvers.h:408)
  // REANCHOR: every this many iterations the recurrence form of the Stiefel Hessian's projection matrix (stpcg.hip) is
  // re-anchored -- G(p) and G(r) recomputed by the direct form (two passes over X, Y and the vector) -- so that the
  // absolute error the recurrences carry stays at the scale of the CURRENT residual instead of the initial one
  // (r06, tests/test_gpu_long_solves.py).  0: never (the r03-r05 behaviour).
  int reanchor = 25;  // (25: the alpha trace of a 1000-iteration-budget solve is followed as long as with the direct form --
                      //  p = 8: 305 iterations, 50: 261, never: 256, two-pass 303; tools/anchor_probe.py; +2.1 % per iteration
                      //  in long solves, nothing in solves of <= 25 iterations)
  bool no_update_pair = false;      // NO_UPDATE_PAIR: the matrix-pipe panel update in 16-row blocks, 8 bytes per lane (r04 form)
  bool no_gram_half = false;        // NO_GRAM_HALF: the fused Gram pair with a tile column of its own per Gram (r04 form)
  bool no_update_mfma = false;      // NO_UPDATE_MFMA: the 48-column panel update on the vector pipe
  bool halo_rprime = false;         // HALO_RPRIME: sharded STPCG exchanges the halo of r' (with the <r,v> all-reduce) and every
                                    // rank forms halo(p') = -halo(r') + beta halo(p) itself: 2 collectives per iteration
                                    // instead of 3 on the RCCL layer (`--comm rccl2`)
  bool early_s = false;             // EARLY_S: the direction kernel of the single-context recurrence solve applies s += alpha p AHEAD of its reduction (k_cg_pupdate_early; opt-in experiment, same bits, same time)
  bool so3_no_rquat = false;        // SO3_NO_RQUAT: the SO(3)^N model assembly gathers the neighbours' rotations as 72-byte matrices
                                    // (r05 form) instead of 32-byte quaternions written by the retraction (r06; creation-time)
  bool so3_no_quat = false;         // SO3_NO_QUAT: the measurements of mi_so3n stay 3 x 3 matrices (r04 form; creation-time)
  int so3_sort_nbr = 0;             // SO3_SORT_NBR: mi_so3n_create orders a node's incidences by neighbour (experiment)
};

struct KTimer {
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  size_t launches = 0;
  double total_ms = 0;
};

}  // namespace mi

struct mi_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  char device_name[128] = {0};
  int num_cu = 256;
  bool uniform_grid = false;  // see uniform_grid() below
  // cap of the streaming kernels' grids (default kMaxGrid; MI355OPT_MAX_GRID).  Several processes rehearsing a
  // multi-GPU run on ONE GPU need it: the consumer kernels WAIT for their peers in their prologue (comm_ipc.h), so
  // the kernels of all ranks must be resident at the same time.
  int max_grid = 512;
  // memory pool: free lists keyed by byte size
  std::multimap<size_t, void *> pool_free;
  std::map<void *, size_t> pool_all;
  size_t pool_bytes = 0;
  // reductions: component-major partial buffers, kMaxComps x kMaxRows doubles each
  void *control_slab = nullptr;     // one allocation holding everything below up to `cg`
  double *partials = nullptr;       // operator -> CG (curvature dots)
  double *partials_b = nullptr;     // CG update -> CG direction (<r,v>)
  double *partials2 = nullptr;      // operator-internal (e.g. Stiefel Gram)
  double *partials_user = nullptr;  // mi_vec_dot* (safe to call from callbacks)
  double *scalars = nullptr;        // kScalarSlots doubles
  double *host_scalars = nullptr;   // pinned staging for scalar read-backs
  mi::CgState *cg = nullptr;        // device, copy 0
  mi::CgState *cg1 = nullptr;       // device, copy 1
  mi::CgState *cg_host = nullptr;   // pinned copy for read-back
  const double *twok_r = nullptr;   // two-kernel step: the residual the Hessian pass reads (set by mi_stpcg around apply_dir)
  const mi::CgState *cg_live = nullptr;  // state copy operators may consult to skip work after exit
  // mi_stpcg with defer_result: the state copy into cg_host is in flight behind this event (mi_stpcg_collect)
  hipEvent_t cg_deferred_ev = nullptr;
  bool cg_deferred = false;
  size_t cg_deferred_hvp = 0;
  unsigned long long cg_deferred_seq = 0;  // polled form of the deferred result (0: the event form)
  mi::HostStatus *status = nullptr;      // pinned, device-visible
  mi::HostStatus *status_dev = nullptr;  // device pointer of the same memory
  double *trace_dev = nullptr;           // 4 x trace_cap doubles
  size_t trace_cap = 0;
  unsigned int epoch = 0;
  uint64_t vec_serial = 0;  // mi_vec::serial source
  size_t host_syncs = 0;  // stream synchronisations the library made on this context (mi_ctx_sync_count)
  mi_fusion_counters fusion = {0, 0, 0, 0, 0, 0, 0};  // mi_ctx_fusion_counters
  bool warn_generic = false;      // env MI355OPT_WARN_GENERIC=1: one stderr line per kind of generic fall
  bool warned_generic[3] = {false, false, false};
  // pinned staging ring for small host -> device uploads that must not stall the host (stage_upload, context.hip)
  static constexpr int kStageSlots = 4;
  static constexpr size_t kStageBytes = 96 * 96 * sizeof(double);
  void *stage_host[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[kStageSlots] = {nullptr, nullptr, nullptr, nullptr};
  int stage_next = 0;
  // pinned landing area for small device -> host read-backs (readback_sync, context.hip): a copy into PAGEABLE memory
  // goes through the runtime's own staging buffers with a blit kernel and an internal wait (15 us per 41 KB Gram)
  void *readback_host = nullptr;
  size_t readback_bytes = 0;
  // stream_wait (context.hip): a one-thread kernel behind the work stores the next sequence number into this coherent
  // pinned word and the host polls it -- the wake-up of hipStreamSynchronize costs more than the kernel
  unsigned long long *poll_flag = nullptr;
  unsigned long long poll_seq = 0;
  // timing
  mi::KTimer ktime[MI_K_COUNT];
  std::vector<hipEvent_t> event_pool;
  hipEvent_t t_start = nullptr, t_stop = nullptr;
  // comm (RCCL); opaque here, defined in comm.hip
  void *comm = nullptr;
  int world_size = 1, rank = 0;
  bool force_slot_path = false;  // env MI355OPT_FORCE_SLOT_PATH=1: use the multi-GPU (reduce-kernel + slots) path on one GPU
  bool force_lockstep = false;   // env MI355OPT_FORCE_LOCKSTEP=1: multi-rank enqueue rule of mi_stpcg on one GPU
  bool no_dirgram = false;       // env MI355OPT_NO_DIRGRAM=1: STPCG ignores mi_op::dirgram (two-pass Stiefel Hessian)
  bool dirgram_direct = false;   // env MI355OPT_DIRGRAM_DIRECT=1: direction-kernel Gram rows even without a preconditioner
  mi::Config cfg;                // the other switches (parsed with the five above at creation; mi_ctx_set_option)
};

struct mi_vec {
  mi_ctx *ctx;
  size_t n;
  double *d;
  bool owned;  // storage belongs to the pool
  // Identity of the CONTENTS, for caches keyed on "this vector as it was when I looked" (the speculative trial point
  // of mi_stiefel_rq_trial): `serial` is unique per mi_vec_create/mi_vec_view on a context (handles and pooled
  // device pointers are both recycled), `gen` is bumped by every entry point that writes the vector (mi::touch).
  // A VIEW (mi_vec_view) shares the generation counter of the vector that owns the storage (`root`; null for an owning
  // vector, the owner itself for a view of a view): a write through the base, through a sibling view or through the
  // view itself invalidates whatever was cached on any of them (r04, ADVICE).  Writes through the raw pointer of
  // mi_vec_data are invisible to the library: the caller announces them with mi_vec_touch.
  uint64_t serial = 0;
  uint64_t gen = 0;
  mi_vec *root = nullptr;
  // Lifetime across the C ABI (r05, ADVICE): an owner counts its live views; mi_vec_destroy on an owner that still has
  // views only marks it `zombie` — storage and generation counter stay until the last view goes — so a plain C client
  // that destroys in the wrong order gets neither a host use-after-free in touch() nor a recycled device pointer.
  uint32_t views = 0;
  bool zombie = false;
};
namespace mi {
inline void touch(mi_vec *v) {
  if (v) ++(v->root ? v->root : v)->gen;
}
inline uint64_t gen_of(const mi_vec *v) { return (v->root ? v->root : v)->gen; }
}  // namespace mi

namespace mi {

int pool_alloc(mi_ctx *ctx, size_t bytes, void **out);
void pool_free(mi_ctx *ctx, void *p);
// Asynchronous upload of <= mi_ctx::kStageBytes from pageable host memory: copied into a pinned slot of a small ring
// and from there in-stream; the caller's buffer may die at once, the host does not wait for the device (a slot is
// reused only after the copy that read it has completed).
int stage_upload(mi_ctx *ctx, const void *src, size_t bytes, void *dst_dev);
// n device buffers -> n host buffers behind ONE stream synchronisation, through the context's pinned landing area
// (counted in mi_ctx::host_syncs)
int readback_sync(mi_ctx *ctx, int n, const void *const *dev, const size_t *bytes, void *const *host);
// The landing area itself, for kernels that write their (small) result straight into host memory: at least `bytes`
// of pinned memory, as a host pointer and as the device pointer of the same memory.  One user at a time: whoever asks
// next may get the same bytes.
int readback_area(mi_ctx *ctx, size_t bytes, void **host, void **dev);
// wait until everything enqueued on the context's stream so far has finished (counts as one host synchronisation)
int stream_wait(mi_ctx *ctx, const char *what);
// the two halves of stream_wait for a caller whose own last kernel stores the flag: the sequence number to store
// (0: polling is off) and the device address of the word; then the bounded poll with hipStreamSynchronize behind it
unsigned long long poll_begin(mi_ctx *ctx, unsigned long long **flag_dev);
int poll_finish(mi_ctx *ctx, unsigned long long seq, const char *what);
int ensure_device();

// workgroups for an n-element streaming kernel in which each thread handles `per_thread` elements
// per grid-stride step
// mi_ctx::uniform_grid is set by mi_comm_init when world_size > 1 (one process per GPU): every
// reduction-producing kernel of that context then runs with exactly kMaxGrid workgroups on EVERY rank, so that all
// ranks leave the same number of partial rows and the component-major partial buffers can be
// all-reduced row by row (comm_allreduce_rows) -- idle workgroups store zero partials.
inline int uniform_grid(const mi_ctx *ctx, size_t blocks) {
  if (ctx->uniform_grid) return kMaxGrid;
  if (blocks < 1) blocks = 1;
  if (blocks > (size_t)ctx->max_grid) blocks = ctx->max_grid;
  return (int)blocks;
}

inline int grid_for(const mi_ctx *ctx, size_t n, int per_thread) {
  if (ctx->uniform_grid) return kMaxGrid;
  size_t blocks = (n + (size_t)kBlock * per_thread - 1) / ((size_t)kBlock * per_thread);
  if (blocks < 1) blocks = 1;
  if (blocks > (size_t)ctx->max_grid) blocks = ctx->max_grid;
  return (int)blocks;
}

// Profiler range (roctx) for the lifetime of the object: one per solve
struct RangeScope {
  explicit RangeScope(const char *name) { (void)mi_range_push(name); }
  ~RangeScope() { (void)mi_range_pop(); }
};

// Kernel-timing scope: brackets a launch with an event pair when enabled for `id`.
struct KScope {
  mi_ctx *ctx;
  int id;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  KScope(mi_ctx *c, int kid);
  ~KScope();
};
hipEvent_t event_get(mi_ctx *ctx);

// all-reduce (sum) of `count` doubles at device pointer `buf`, in-stream; no-op when world_size==1
int comm_allreduce(mi_ctx *ctx, double *buf, int count);
// Multi-rank reduction of k components of a component-major partial buffer WITHOUT the one-workgroup
// reduce kernel: all-reduce all k x kMaxRows doubles in place; the consumers keep the prologue
// re-reduction of the single-GPU path (sum over ranks first, then over rows: fixed order, same bits on
// every rank).  rows_mode(): a communicator is attached and the slot path is not forced.
int comm_allreduce_rows(mi_ctx *ctx, double *partials, int k);
bool comm_ipc_enabled(const mi_ctx *ctx);
// the peer-memory layer carries the exchanges AND they are folded into the consumers' prologues (comm_ipc.h)
bool comm_fold_enabled(const mi_ctx *ctx);
inline bool rows_mode(const mi_ctx *ctx) {
  return ctx->comm != nullptr && !ctx->force_slot_path && !comm_ipc_enabled(ctx);
}
// slot variants (FROM_SLOTS kernels): forced, or the peer-memory layer delivers the sums into slots
inline bool slot_mode(const mi_ctx *ctx) { return ctx->force_slot_path || comm_ipc_enabled(ctx); }

// k <= 4 dot products -> ctx->scalars[slot0..slot0+k) on the device (all-reduced across ranks)
int dot_batch_to_slots(mi_ctx *ctx, int k, const double *const *x, const double *const *y, size_t n,
                       int slot0);
int read_slots_sync(mi_ctx *ctx, int slot0, int k, double *out);
// per-workgroup partials of <x,y>,<y,y>,<x,x> into ctx->partials components 0,1,2
int launch_dot3_partials(mi_ctx *ctx, size_t n, const double *x, const double *y, int *nparts);
// one-workgroup kernel: sum `count` partial rows of components [0,k) -> slots[0..k)
int launch_reduce_rows_to_slots(mi_ctx *ctx, const double *partials, int count, int k, double *slots, int nbatch = 1,
                                size_t batch_stride = 0);
// the same followed by the sum over ranks (one kernel with the peer-memory layer, else reduce kernel +
// RCCL all-reduce); no-op beyond the local reduction without a communicator
int reduce_rows_allreduce(mi_ctx *ctx, const double *partials, int count, int k, double *slots);

// scalar-file slot map
// SLOT_GDIR: packed symmetric Gram of the CG residual [0,NS) and of the direction [40, 40+NS) (recurrence form of
// mi_op::dirgram, stpcg.hip; NS = p(p+1)/2 <= 36 for p <= 8); SLOT_GRAM doubles as the 3+NS-component slot file of
// that form (<= 39 slots); SLOT_RAW: the raw p x p Gram of mi_stiefel_gram (<= 64 slots)
enum { SLOT_USER = 0, SLOT_CG = 8, SLOT_GRAM = 16, SLOT_GDIR_P = 40 /* offset of G(p) inside SLOT_GDIR */, SLOT_MISC = 56, SLOT_GDIR = 64,
       SLOT_RAW = 192 };
static_assert(SLOT_GRAM + 3 + 36 <= SLOT_MISC && SLOT_GDIR + SLOT_GDIR_P + 36 <= SLOT_RAW && SLOT_RAW + 64 <= kScalarSlots,
              "slot map");

// ---- device helpers ---------------------------------------------------------------------
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// fixed-order sum of the 16 per-wave values lds[0..15]
__device__ __forceinline__ double sum16(const double *l) {
  return (((l[0] + l[1]) + (l[2] + l[3])) + ((l[4] + l[5]) + (l[6] + l[7]))) +
         (((l[8] + l[9]) + (l[10] + l[11])) + ((l[12] + l[13]) + (l[14] + l[15])));
}

// Workgroup-wide sums of K per-thread accumulators, written as this workgroup's partial row
// (component-major: partials[c * kMaxRows + blockIdx.x]).  Fixed shape => deterministic.
// lds: >= K * 16 doubles.  Contains barriers: call from uniform control flow.
template <int K>
__device__ __forceinline__ void block_partials_store(const double (&acc)[K], double *lds,
                                                     double *__restrict__ partials) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_reduce_sum(acc[k]);
    if (lane == 0) lds[k * kWaves + w] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) partials[(size_t)threadIdx.x * kMaxRows + blockIdx.x] = sum16(lds + threadIdx.x * kWaves);
  __syncthreads();
}

// The same for a workgroup of NW (a power of two <= 16) waves (the window kernels): fixed-order pairwise sum.
template <int K, int NW>
__device__ __forceinline__ void block_partials_store_nw(const double (&acc)[K], double *lds,
                                                        double *__restrict__ partials) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double v = wave_reduce_sum(acc[k]);
    if (lane == 0) lds[k * NW + w] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double t[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) t[i] = lds[threadIdx.x * NW + i];
#pragma unroll
    for (int span = 1; span < NW; span *= 2)
#pragma unroll
      for (int i = 0; i + span < NW; i += 2 * span) t[i] = t[i] + t[i + span];
    partials[(size_t)threadIdx.x * kMaxRows + blockIdx.x] = t[0];
  }
  __syncthreads();
}

// Every thread of the workgroup obtains the fixed-order sums over `count` (<= kMaxRows) partial rows
// of components [0,K).  All workgroups of a kernel run identical code on identical data, so they
// all obtain bit-identical totals.  lds: >= K * 17 doubles.  Contains barriers.
// One wave per component (K <= 16 waves): each lane sums its <= 8 rows in row order from 8
// independent coalesced loads, then a single wave reduction; one barrier publishes the K totals.
// (The earlier all-waves form cost K x 16 fp64 wave reductions = hundreds of ds_bpermute per
// workgroup, ~2-4 us at the head of every consumer kernel.)
// after_issue(): called once the row loads are in the queue and before they are waited for -- the place for a
// consumer's own first loads (they then return BEHIND the rows, in order, and do not delay the reduction).
// K > 16 (the 3 + 15 ... 36 components of a Stiefel solve with p = 5 ... 8): wave w takes components w, w + 16, ...
// in rounds -- same row order, same wave reduction per component; lds: >= K doubles.
template <int K, class F>
__device__ __forceinline__ void reduce_rows_wide(const double *__restrict__ partials, int count, double (&out)[K],
                                                 double *lds, F &&after_issue) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int R = (K + kWaves - 1) / kWaves;
  double v[R];
#pragma unroll
  for (int q = 0; q < R; ++q) {
    const int c = w + q * kWaves;
    double t[kMaxRows / 64];
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) {
      const int r = lane + 64 * j;
      t[j] = (c < K && r < count) ? partials[(size_t)c * kMaxRows + r] : 0.0;
    }
    if (q == 0) after_issue();
    double a = 0;
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) a += t[j];
    v[q] = wave_reduce_sum(a);
  }
#pragma unroll
  for (int q = 0; q < R; ++q)
    if (lane == 0 && w + q * kWaves < K) lds[w + q * kWaves] = v[q];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = lds[k];
  __syncthreads();
}
template <int K, class F>
__device__ __forceinline__ void reduce_rows(const double *__restrict__ partials, int count,
                                            double (&out)[K], double *lds, F &&after_issue) {
  if constexpr (K > kWaves) {
    reduce_rows_wide<K>(partials, count, out, lds, after_issue);
    return;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double t[kMaxRows / 64];
#pragma unroll
  for (int j = 0; j < kMaxRows / 64; ++j) t[j] = 0.0;
  if (w < K) {
    const double *src = partials + (size_t)w * kMaxRows;
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) {
      const int r = lane + 64 * j;
      if (r < count) t[j] = src[r];
    }
  }
  after_issue();
  if (w < K) {
    double v = 0;
#pragma unroll
    for (int j = 0; j < kMaxRows / 64; ++j) v += t[j];
    v = wave_reduce_sum(v);
    if (lane == 0) lds[w] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = lds[k];
  __syncthreads();
}
template <int K>
__device__ __forceinline__ void reduce_rows(const double *__restrict__ partials, int count,
                                            double (&out)[K], double *lds) {
  reduce_rows<K>(partials, count, out, lds, [] {});
}
// The same for a workgroup of NW waves (r05: the 256-thread kernels of Stiefel rows wider than 4 doubles): wave w sums
// components w, w + NW, ... in rounds; a lane's rows in row order, then the wave reduction.  lds: >= K doubles.
template <int K, int NW>
__device__ __forceinline__ void reduce_rows_nw(const double *__restrict__ partials, int count, double (&out)[K],
                                               double *lds) {
  if constexpr (NW == kWaves) {
    reduce_rows<K>(partials, count, out, lds);
  } else {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = w; c < K; c += NW) {
      const double *src = partials + (size_t)c * kMaxRows;
      double t[kMaxRows / 64];
#pragma unroll
      for (int j = 0; j < kMaxRows / 64; ++j) {
        const int r = lane + 64 * j;
        t[j] = (r < count) ? src[r] : 0.0;
      }
      double a = 0;
#pragma unroll
      for (int j = 0; j < kMaxRows / 64; ++j) a += t[j];
      a = wave_reduce_sum(a);
      if (lane == 0) lds[c] = a;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = lds[k];
    __syncthreads();
  }
}
// block_partials_store for a workgroup of NW waves (NW == 16: the 1024-thread form, same bits as ever)
template <int K, int NW>
__device__ __forceinline__ void block_partials_store_w(const double (&acc)[K], double *lds, double *__restrict__ partials) {
  if constexpr (NW == kWaves) block_partials_store<K>(acc, lds, partials);
  else block_partials_store_nw<K, NW>(acc, lds, partials);
}

// XCD-aware workgroup remap (guide T1): consecutive logical tiles land on the same XCD so that a
// contiguous row range shares one L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
  const unsigned q = nb / kNumXCD, r = nb % kNumXCD, xcd = b % kNumXCD, idx = b / kNumXCD;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace mi

// Sparse matrix in sliced-ELL-64 (one slice = one wavefront of rows): element (slice s, k, lane)
// lives at (slice_ptr[s] + k) * 64 + lane, so that a wave's loads of values and column indices are
// perfectly coalesced.  Padding entries have val = 0 and col = own row.
struct mi_csr {
  mi_ctx *ctx = nullptr;
  size_t n = 0;        // local rows
  size_t ncols = 0;    // local columns incl. halo (== n when not sharded)
  size_t nnz = 0;      // true non-zeros (algorithmic byte accounting uses this)
  // A == A' entry by entry (same bits), checked once at creation (sparse.hip csr_is_symmetric; a row shard checks its
  // diagonal block).  The one-pass Stiefel Hessian rests on X'(A p) = (A X)'p: without symmetry the operator keeps
  // its two-pass form (stiefel.hip mi_stiefel_rq_model).
  bool symmetric = false;
  size_t padded = 0;   // stored entries
  size_t nslices = 0;
  long long *slice_ptr = nullptr;  // device, nslices + 1
  int *col = nullptr;              // device, padded
  double *val = nullptr;           // device, padded
  // Value-indexed packed copy (lossless; built when the matrix has <= 256 distinct stored values and every
  // stored column lies within +-2^23 of its row): entry = (col - row) << 8 | index into vtab.  4 bytes per stored entry instead of 12:
  // stencil / unit-weight Laplacian operators stream a third of the bytes (spmm_core.h sell_stream<.,.,true>).
  uint32_t *pk = nullptr;          // device, padded (null: not representable)
  double *vtab = nullptr;          // device, 256 doubles
  int nvtab = 0;
  // LDS-window form of the sparse kernels (spmm_core.h sell_window): entries whose column lies within
  // 64 * win_chunks rows of their row are gathered from an LDS ring of the workgroup's rows of V instead of
  // through L1/L2.  0: the matrix does not qualify (decided at creation, sparse.hip build_window).
  //   wk    one dword per stored entry (positions as pk): (LDS row index << 8) | value index
  //   wfar  two columns per row, (slice * 2 + slot) * 64 + lane: the row's far entries (else the row itself)
  int win_chunks = 0;
  int win_head = 0;        // widest slice (<= kWinHead)
  uint32_t win_zero = 0;   // word of a non-entry: zero row, index of 0.0 in vtab
  uint32_t *wk = nullptr;  // device, padded + kWinHead * 64
  int32_t *wfar = nullptr; // device, (nslices + 1) * 2 * 64
  // 16-bit form of the words (matrices with <= 32 table values incl. 0.0 and LDS rows < 2048): entry = (LDS row << 5)
  // | value index, EIGHT entries per row in four dwords, (slice * 4 + q) * 64 + lane -- fixed 16 bytes per row,
  // non-entries are the zero word, no slice bounds needed.  Null when the matrix does not qualify.
  uint32_t *wk16 = nullptr;
  size_t win_far_stride = 0;       // |column - row| shared by >= 80 % of the far entries, or 0
  size_t win_far_pure = 0;         // D when EVERY far entry is at row +- D (slot 0: +D, slot 1: -D), else 0
  // workgroup -> first tile table of the window kernels (stiefel.hip window_bounds), built on first use for one
  // workgroup budget: win_bounds_n + 1 device ints
  // One plan per workgroup budget (the Hessian pass and the column-major panel product ask with different budgets on
  // the same matrix: a single slot would be rebuilt -- hipFree + hipMalloc + blocking copy -- on every alternation);
  // plans live as long as the matrix, an entry appears only after its table is on the device.
  struct WinPlan {
    int *bounds = nullptr;  // device, n + 1 ints
    int n = 0;
  };
  mutable std::map<int, WinPlan> win_plans;
  // row-sharded operation (world_size > 1).  Local column index c < n addresses the local rows of
  // V; c >= n addresses the halo buffer: [n, n+halo_lo) = last halo_lo rows of rank-1,
  // [n+halo_lo, n+halo_lo+halo_hi) = first halo_hi rows of rank+1.
  size_t halo_lo = 0, halo_hi = 0;  // rows received from rank-1 / rank+1
  size_t send_lo = 0, send_hi = 0;  // rows sent to rank-1 (our first rows) / rank+1 (our last rows)
  double *halo = nullptr;           // device, (halo_lo + halo_hi) * kMaxP doubles (p <= kMaxP = 8)
  // peer-memory (IPC) exchange: the halo lives in this rank's arena at byte offset halo_off (the same
  // offset on every rank); peer_lo_rows = halo_lo of rank-1 (our first rows land behind them there)
  bool halo_in_arena = false;
  size_t halo_off = 0, peer_lo_rows = 0;
  // The halo storage holds TWO buffers of halo_stride doubles; exchange number e of this matrix lands in
  // buffer e & 1.  A peer-store push only waits for the neighbour's PUSH of the same exchange, not for the
  // neighbour's kernels that read the halo afterwards, so with one buffer a fast rank's next push could
  // overwrite rows a slow rank is still reading (two sharded SpMMs back to back, no scalar exchange in between:
  // seen once in ~60 runs of tests/ipc_worker.py).  With two, a rank would have to be two pushes ahead, which
  // the flag wait of the push in between rules out.
  size_t halo_stride = 0;
  mutable unsigned long long halo_exchanges = 0;
  const double *halo_cur() const { return halo ? halo + (halo_exchanges & 1) * halo_stride : nullptr; }
  // r'-halo form of the sharded STPCG (Config::halo_rprime, comm.hip comm_rprime_*): where the neighbours' boundary rows of
  // the new RESIDUAL land (same layout and double-buffering as `halo`); allocated on first use, by every rank alike
  mutable double *halo_r = nullptr;
  mutable bool halo_r_in_arena = false;
  mutable size_t halo_r_off = 0;
  mutable unsigned long long halo_r_exchanges = 0;
};

namespace mi {
// W = A V - (*scale) W with partials of |W|^2 (p = 1): see mi_op::apply_sub_scaled
int csr_spmv_sub_scaled(const mi_csr *A, const mi_vec *V, const double *scale, const int *mode, const int *gate,
                        mi_vec *W, double *partials, int *nparts);
// W = A V with the curvature partials <V,W>, <W,W>, <V,V> in ctx->partials (see mi_op::apply_dots)
int csr_spmm_dots(const mi_csr *A, int p, const mi_vec *V, mi_vec *W, int *nparts, bool *unsupported);
// In-stream halo exchange of the n x p field V into A->halo (no-op when not sharded).
int comm_halo_exchange(mi_ctx *ctx, const mi_csr *A, int p, const double *V);
int comm_exchange_halo_counts(mi_ctx *ctx, size_t need_lo, size_t need_hi, size_t *send_lo,
                              size_t *send_hi, size_t *peer_lo_rows, size_t *max_halo_rows);
// halo storage for a sharded matrix: inside the IPC arena when the peer-memory layer is mapped (bytes
// must be the same on every rank), else an ordinary device allocation
int comm_halo_alloc(mi_ctx *ctx, size_t bytes, double **ptr, bool *in_arena, size_t *arena_off);
void comm_halo_free(mi_ctx *ctx, double *ptr, bool in_arena);
// W = A V without the halo exchange (callers that fuse do it themselves)
int csr_spmm_launch(const mi_csr *A, int p, const double *V, double *W);
// launch plan of the LDS-window kernels for a workgroup budget (sparse.hip): grid and, when the runs are cut by the
// far stride, the workgroup -> first tile table on the device (cached on the matrix)
int window_bounds(mi_ctx *ctx, const mi_csr *A, int wgs, int ntiles, int *grid, const int **bounds_out);
}  // namespace mi

// operator / preconditioner objects --------------------------------------------------------
struct mi_op {
  mi_ctx *ctx = nullptr;
  size_t n = 0;      // input length
  size_t n_out = 0;  // output length (0: same as n)
  // out = Op(in)
  int (*apply)(mi_op *self, const mi_vec *in, mi_vec *out) = nullptr;
  // out = Op(in) and per-workgroup partials of <in,out>, <out,out>, <in,in> in ctx->partials
  // components 0,1,2; *nparts = number of partial rows written.  nullptr => generic dot3 kernel.
  // Implementations may consult ctx->cg_live (device CgState, may be null) to skip work once the
  // solve has left CG_RUN.
  int (*apply_dots)(mi_op *self, const mi_vec *in, mi_vec *out, int *nparts) = nullptr;
  // inout = Op(in) - (*scale) inout, with per-workgroup partials of <inout,inout> in partials component 0
  // (the bidiagonalisation step of LSQR, IterativeSolvers.h:707,712, folded into the operator's last pass).
  // scale / mode / gate are DEVICE pointers: the kernel does nothing when *mode != 0 or (gate && !*gate).
  // nullptr => the solver applies the operator and runs its own update kernel.
  int (*apply_sub_scaled)(mi_op *self, const mi_vec *in, const double *scale, const int *mode, const int *gate,
                          mi_vec *inout, double *partials, int *nparts) = nullptr;
  // Direction-Gram fusion (Stiefel Rayleigh-quotient Hessian): when `dirgram` is set, STPCG's direction
  // kernel also leaves, in ctx->partials2, the partial rows of  sym(Y'p - (X'p) S)  of the NEW direction p
  // (= sym(X'(A p - p S)) for symmetric A, Y = A X), and the solver calls apply_dir instead of apply_dots:
  // out = Op(in) + the three curvature partials in ONE pass over the matrix, the projection's P x P
  // matrix being re-reduced from those `gram_count` rows in the kernel's prologue.
  // gram_count < 0 selects the recurrence form (no preconditioner): the Gram is LINEAR in the direction, and
  // p' = -r' + beta p, r' = r + alpha Hp, so STPCG carries G(r), G(p) as 2 x P(P+1)/2 replicated scalars
  // (ctx->scalars + SLOT_GDIR) and the operator pass itself leaves the rows of G(Hp) as components 3.. of
  // ctx->partials next to the three dots; its prologue reads M = G(p) from there -- no Gram rows to reduce,
  // no X/Y reads in the direction kernel, and one exchange fewer per iteration across ranks.
  const struct mi_dirgram *dirgram = nullptr;
  int (*apply_dir)(mi_op *self, const mi_vec *in, mi_vec *out, int gram_count, int *nparts) = nullptr;
  void (*destroy)(mi_op *self) = nullptr;
  void *impl = nullptr;
  bool borrowed = false;  // owned by a problem object; mi_op_destroy is a no-op
};

struct mi_dirgram {
  int p = 0;                  // columns of the n x p fields
  size_t n = 0;               // rows
  const double *X = nullptr;  // n x p, row-major
  const double *Y = nullptr;  // A X
  const double *S = nullptr;  // device, p x p
  // the matrix whose halo exchange precedes every apply_dir (null: none).  STPCG may fold the push half of the NEXT
  // exchange into its direction kernel (comm_ipc.h comm_halo_fold_next); apply_dir then only waits.
  const struct mi_csr *halo_A = nullptr;
  bool twok = false;  // apply_dir(gram_count = -2) exists: the Hessian pass of the two-kernel step (Config::two_kernel_step)
};

struct mi_precon {
  mi_ctx *ctx = nullptr;
  size_t n = 0;
  int kind = 0;  // 0 callback/generic, 1 diag (fusable), 2 block3 (fusable), 3 constraint (kkt.hip)
  const double *data = nullptr;  // dinv (n) or inverse blocks (9 * n/3)
  int (*apply)(mi_precon *self, const mi_vec *r, mi_vec *v) = nullptr;
  // constraint preconditioner (IterativeSolvers.h:229-253,381-405): (v, lambda) = P(r) and, if subtract, the
  // residual correction r -= A' lambda of the `At` branch in the same pass
  int (*apply_project)(mi_precon *self, mi_vec *r, mi_vec *v, int subtract) = nullptr;
  void (*destroy)(mi_precon *self) = nullptr;
  void *impl = nullptr;
  bool borrowed = false;
  // device address of a sticky failure word (double; 0 = every application since mi_stpcg last cleared it did what it claims), or null.
  // mi_stpcg copies it behind the solve's own state read-back and answers MI_ERR_INTERNAL if it is set.
  double *fail_word = nullptr;
};
