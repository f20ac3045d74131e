// context.hip -- context, stream, pooled allocator, event timing, error reporting.
#include "mi_internal.h"
#include <chrono>
#include <cstring>
#include <strings.h>
#include <unistd.h>

#include <dlfcn.h>

namespace mi {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line) {
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  (void)hipGetLastError();
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return MI_ERR_NO_DEVICE;
  if (e == hipErrorOutOfMemory) return MI_ERR_OOM;
  return MI_ERR_HIP;
}

static int readback_reserve(mi_ctx *ctx, size_t total) {
  if (total > ctx->readback_bytes) {
    MI_HIP(hipStreamSynchronize(ctx->stream));  // (nobody may still be writing into the old area)
    if (ctx->readback_host) (void)hipHostFree(ctx->readback_host);
    ctx->readback_host = nullptr;
    ctx->readback_bytes = 0;
    const size_t cap = std::max<size_t>(total, 256 * 1024);
    // (Mapped | Coherent like poll_flag and host_scalars: kernels store Gram blocks and residual norms straight into this
    // area and the host reads them behind a polled flag -- with hipHostMallocDefault the coherence of such stores hangs
    // on the HIP_HOST_COHERENT environment variable)
    MI_HIP(hipHostMalloc(&ctx->readback_host, cap, hipHostMallocMapped | hipHostMallocCoherent));
    ctx->readback_bytes = cap;
  }
  return MI_OK;
}
int readback_area(mi_ctx *ctx, size_t bytes, void **host, void **dev) {
  MI_TRY(readback_reserve(ctx, bytes));
  *host = ctx->readback_host;
  MI_HIP(hipHostGetDevicePointer(dev, ctx->readback_host, 0));
  return MI_OK;
}
__global__ void k_host_flag(unsigned long long *flag, unsigned long long seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// device pointers of the polled word and of a coherent pinned area; 0 if polling is off (or the mapping failed)
unsigned long long poll_begin(mi_ctx *ctx, unsigned long long **flag_dev) {
  if (!ctx->poll_flag || ctx->cfg.no_polled_sync) return 0;
  if (hipHostGetDevicePointer((void **)flag_dev, ctx->poll_flag, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return ++ctx->poll_seq;
}
// Poll for `seq` (stored by a kernel the caller enqueued behind the work -- and whose LAUNCH the caller has checked:
// seq = 0 means "no flag will come").  Bounded: a kernel that faulted never stores the flag -- the synchronisation
// below then reports the error -- and a long queue is better slept on than spun on.
static inline void cpu_relax() {
#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(__i386__))
  __builtin_ia32_pause();
#endif
}
int poll_finish(mi_ctx *ctx, unsigned long long seq, const char *what) {
  if (seq) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
      if (__atomic_load_n(ctx->poll_flag, __ATOMIC_ACQUIRE) >= seq) return MI_OK;  // (the sequence only grows, in stream order)
      cpu_relax();  // (spin-wait hint: a sibling hyper-thread / an oversubscribed host gets the core's slots)
      if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return hip_fail(e, what, __FILE__, __LINE__);
  return MI_OK;
}
int stream_wait(mi_ctx *ctx, const char *what) {
  ctx->host_syncs++;
  unsigned long long *dev = nullptr;
  const unsigned long long seq = poll_begin(ctx, &dev);
  if (seq) {
    hipLaunchKernelGGL(k_host_flag, dim3(1), dim3(1), 0, ctx->stream, dev, seq);
    if (hipGetLastError() != hipSuccess) return poll_finish(ctx, 0, what);  // (no flag will come: the blocking wait)
  }
  return poll_finish(ctx, seq, what);
}
int readback_sync(mi_ctx *ctx, int n, const void *const *dev, const size_t *bytes, void *const *host) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += (bytes[i] + 63) / 64 * 64;
  MI_TRY(readback_reserve(ctx, total));
  size_t off = 0;
  hipError_t e = hipSuccess;
  for (int i = 0; i < n && e == hipSuccess; ++i) {
    e = hipMemcpyAsync((char *)ctx->readback_host + off, dev[i], bytes[i], hipMemcpyDeviceToHost, ctx->stream);
    off += (bytes[i] + 63) / 64 * 64;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  ctx->host_syncs++;
  if (e != hipSuccess) return hip_fail(e, "read-back", __FILE__, __LINE__);
  off = 0;
  for (int i = 0; i < n; ++i) {
    memcpy(host[i], (const char *)ctx->readback_host + off, bytes[i]);
    off += (bytes[i] + 63) / 64 * 64;
  }
  return MI_OK;
}

int ensure_device() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (hipGetDeviceCount: %s, count=%d); libmi355opt has no CPU "
              "fallback",
              hipGetErrorString(e), count);
    return MI_ERR_NO_DEVICE;
  }
  return MI_OK;
}

int pool_alloc(mi_ctx *ctx, size_t bytes, void **out) {
  if (bytes == 0) bytes = 8;
  bytes = (bytes + 255) & ~(size_t)255;
  auto it = ctx->pool_free.find(bytes);
  if (it != ctx->pool_free.end()) {
    *out = it->second;
    ctx->pool_free.erase(it);
    return MI_OK;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    // release cached blocks and retry once
    for (auto &kv : ctx->pool_free) {
      (void)hipFree(kv.second);
      ctx->pool_all.erase(kv.second);
      ctx->pool_bytes -= kv.first;
    }
    ctx->pool_free.clear();
    (void)hipGetLastError();
    e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  }
  ctx->pool_all[p] = bytes;
  ctx->pool_bytes += bytes;
  *out = p;
  return MI_OK;
}

void pool_free(mi_ctx *ctx, void *p) {
  if (!p) return;
  auto it = ctx->pool_all.find(p);
  if (it == ctx->pool_all.end()) return;
  ctx->pool_free.insert({it->second, p});
}

int stage_upload(mi_ctx *ctx, const void *src, size_t bytes, void *dst_dev) {
  MI_REQUIRE(bytes <= mi_ctx::kStageBytes, "staged upload of %zu bytes exceeds the slot size", bytes);
  const int slot = ctx->stage_next;
  ctx->stage_next = (slot + 1) % mi_ctx::kStageSlots;
  if (!ctx->stage_host[slot]) {
    MI_HIP(hipHostMalloc(&ctx->stage_host[slot], mi_ctx::kStageBytes, hipHostMallocDefault));
    MI_HIP(hipEventCreateWithFlags(&ctx->stage_ev[slot], hipEventDisableTiming));
  } else {
    MI_HIP(hipEventSynchronize(ctx->stage_ev[slot]));  // the copy that last read this slot (long done in practice)
  }
  memcpy(ctx->stage_host[slot], src, bytes);
  MI_HIP(hipMemcpyAsync(dst_dev, ctx->stage_host[slot], bytes, hipMemcpyHostToDevice, ctx->stream));
  MI_HIP(hipEventRecord(ctx->stage_ev[slot], ctx->stream));
  return MI_OK;
}

hipEvent_t event_get(mi_ctx *ctx) {
  if (!ctx->event_pool.empty()) {
    hipEvent_t e = ctx->event_pool.back();
    ctx->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

KScope::KScope(mi_ctx *c, int kid) : ctx(c), id(kid) {
  if (ctx->ktime[id].enabled) {
    e0 = event_get(ctx);
    e1 = event_get(ctx);
    if (e0) (void)hipEventRecord(e0, ctx->stream);
  }
}
KScope::~KScope() {
  if (e0 && e1) {
    (void)hipEventRecord(e1, ctx->stream);
    ctx->ktime[id].pending.push_back({e0, e1});
  }
}

}  // namespace mi

using namespace mi;

// The one place the library reads its environment: every MI355OPT_<NAME> switch, once per context.
namespace {
struct OptionDesc {
  const char *name;
  int (*set)(mi_ctx *, long);
  bool integer = false;  // an integer-valued option (a count, a time): only an integer is a value for it
};
// (plain functions, not lambdas: a captureless lambda does not convert to a function pointer in hipcc's device pass)
#define OPT_BOOL(FN, FIELD) \
  int FN(mi_ctx *c, long v) { c->FIELD = v != 0; return MI_OK; }
OPT_BOOL(opt_force_slot_path, force_slot_path)
OPT_BOOL(opt_force_lockstep, force_lockstep)
OPT_BOOL(opt_no_dirgram, no_dirgram)
OPT_BOOL(opt_dirgram_direct, dirgram_direct)
OPT_BOOL(opt_force_uniform_grid, cfg.force_uniform_grid)
OPT_BOOL(opt_no_fold, cfg.no_fold)
OPT_BOOL(opt_halo_push_late, cfg.halo_push_late)
OPT_BOOL(opt_no_packed, cfg.no_packed)
OPT_BOOL(opt_no_window, cfg.no_window)
OPT_BOOL(opt_no_win_bounds, cfg.no_win_bounds)
OPT_BOOL(opt_no_far_computed, cfg.no_far_computed)
OPT_BOOL(opt_words16, cfg.words16)
OPT_BOOL(opt_no_spmm_stream, cfg.no_spmm_stream)
OPT_BOOL(opt_no_spmm_win, cfg.no_spmm_win)
OPT_BOOL(opt_no_spmm_sweep, cfg.no_spmm_sweep)
OPT_BOOL(opt_no_zero_copy, cfg.no_zero_copy)
OPT_BOOL(opt_no_polled_sync, cfg.no_polled_sync)
OPT_BOOL(opt_no_update_mfma, cfg.no_update_mfma)
OPT_BOOL(opt_halo_rprime, cfg.halo_rprime)
OPT_BOOL(opt_no_gram_half, cfg.no_gram_half)
OPT_BOOL(opt_so3_no_quat, cfg.so3_no_quat)
OPT_BOOL(opt_so3_no_rquat, cfg.so3_no_rquat)
OPT_BOOL(opt_early_s, cfg.early_s)
OPT_BOOL(opt_no_update_pair, cfg.no_update_pair)
OPT_BOOL(opt_two_kernel_step, cfg.two_kernel_step)
OPT_BOOL(opt_warn_generic, warn_generic)
#undef OPT_BOOL
int opt_max_grid(mi_ctx *c, long v) {
  c->max_grid = (int)std::min<long>(kMaxGrid, std::max<long>(1, v));
  return MI_OK;
}
int opt_wide_quad(mi_ctx *c, long v) {
  c->cfg.wide_quad = v < 0 ? -1 : v != 0;
  return MI_OK;
}
int opt_wide_window(mi_ctx *c, long v) {
  c->cfg.wide_window = v < 0 ? -1 : v != 0;
  return MI_OK;
}
int opt_so3_sort_nbr(mi_ctx *c, long v) {
  c->cfg.so3_sort_nbr = (int)v;
  return MI_OK;
}
int opt_sweep_zsegs(mi_ctx *c, long v) {
  c->cfg.sweep_zsegs = (int)std::max<long>(0, std::min<long>(v, 1024));
  return MI_OK;
}
int opt_reanchor(mi_ctx *c, long v) {
  c->cfg.reanchor = (int)std::max<long>(0, std::min<long>(v, 1 << 20));
  return MI_OK;
}
int opt_ipc_timeout_ms(mi_ctx *c, long v) {
  c->cfg.ipc_timeout_ms = std::max<long>(0, v);
  return MI_OK;
}
const OptionDesc kOptions[] = {
    {"FORCE_SLOT_PATH", opt_force_slot_path}, {"FORCE_LOCKSTEP", opt_force_lockstep},
    {"NO_DIRGRAM", opt_no_dirgram}, {"DIRGRAM_DIRECT", opt_dirgram_direct}, {"MAX_GRID", opt_max_grid, true},
    {"FORCE_UNIFORM_GRID", opt_force_uniform_grid}, {"IPC_TIMEOUT_MS", opt_ipc_timeout_ms, true},
    {"NO_FOLD", opt_no_fold}, {"HALO_PUSH_LATE", opt_halo_push_late}, {"NO_PACKED", opt_no_packed},
    {"NO_WINDOW", opt_no_window}, {"NO_WIN_BOUNDS", opt_no_win_bounds}, {"NO_FAR_COMPUTED", opt_no_far_computed},
    {"WORDS16", opt_words16}, {"NO_SPMM_STREAM", opt_no_spmm_stream}, {"NO_SPMM_WIN", opt_no_spmm_win},
    {"NO_SPMM_SWEEP", opt_no_spmm_sweep}, {"SWEEP_ZSEGS", opt_sweep_zsegs, true},
    {"NO_ZERO_COPY", opt_no_zero_copy}, {"NO_POLLED_SYNC", opt_no_polled_sync}, {"WIDE_QUAD", opt_wide_quad, true}, {"WIDE_WINDOW", opt_wide_window, true},
    {"NO_UPDATE_MFMA", opt_no_update_mfma}, {"HALO_RPRIME", opt_halo_rprime}, {"NO_GRAM_HALF", opt_no_gram_half}, {"SO3_NO_QUAT", opt_so3_no_quat}, {"SO3_NO_RQUAT", opt_so3_no_rquat}, {"EARLY_S", opt_early_s}, {"NO_UPDATE_PAIR", opt_no_update_pair}, {"TWO_KERNEL_STEP", opt_two_kernel_step}, {"SO3_SORT_NBR", opt_so3_sort_nbr, true},
    {"WARN_GENERIC", opt_warn_generic}, {"REANCHOR", opt_reanchor, true},
};
// value of a BOOLEAN switch: an integer; anything else ("yes", "true", "on" -- and the presence-only `MI355OPT_X=` of
// the r01-r03 scripts) means 1, so that no spelling that used to switch something on is silently off.  An INTEGER-valued
// option (MAX_GRID, IPC_TIMEOUT_MS, WIDE_QUAD, SO3_SORT_NBR) takes integers only: `MI355OPT_IPC_TIMEOUT_MS=` or `=yes`
// used to mean a 1 ms timeout and `MI355OPT_MAX_GRID=` one workgroup -- now a warning, and the default stays.
bool option_value(const OptionDesc &o, const char *e, long *v) {
  char *end = nullptr;
  *v = strtol(e, &end, 0);
  if (end != e && *end == '\0') return true;
  if (o.integer) {
    fprintf(stderr, "[mi355opt] warning: MI355OPT_%s=\"%s\" is not an integer; ignored (the default stays)\n", o.name, e);
    return false;
  }
  *v = (!strcasecmp(e, "no") || !strcasecmp(e, "false") || !strcasecmp(e, "off")) ? 0 : 1;
  return true;
}
void config_from_env(mi_ctx *ctx) {
  for (const OptionDesc &o : kOptions) {
    char name[64];
    snprintf(name, sizeof(name), "MI355OPT_%s", o.name);
    long v = 0;
    if (const char *e = getenv(name))
      if (option_value(o, e, &v)) (void)o.set(ctx, v);
  }
  // A removed or misspelt switch must not turn an A/B script into two runs of the default path without a word: warn
  // once per process about every MI355OPT_* variable that is neither a switch of the library nor one of the names the
  // harness around it owns (MI355OPT_LIB, _COMM, _BUILD_TAG, _EXTRA_CFLAGS, _NO_UNITY, _BENCH_*).
  static bool warned = false;
  if (warned || !environ) return;
  warned = true;
  for (char **ep = environ; *ep; ++ep) {
    if (strncmp(*ep, "MI355OPT_", 9) != 0) continue;
    const char *nm = *ep + 9, *eq = strchr(nm, '=');
    const size_t len = eq ? (size_t)(eq - nm) : strlen(nm);
    bool known = false;
    for (const OptionDesc &o : kOptions) known |= strlen(o.name) == len && strncmp(o.name, nm, len) == 0;
    for (const char *h : {"LIB", "COMM", "BUILD_TAG", "EXTRA_CFLAGS", "NO_UNITY"})
      known |= strlen(h) == len && strncmp(h, nm, len) == 0;
    known |= len >= 6 && strncmp(nm, "BENCH_", 6) == 0;
    if (!known)
      fprintf(stderr, "[mi355opt] warning: environment variable MI355OPT_%.*s is not a switch of this build (removed "
                      "or misspelt?); it changes nothing\n", (int)len, nm);
  }
}
}  // namespace

extern "C" {

const char *mi_version(void) { return "mi355opt 0.1.0 (gfx950)"; }
const char *mi_last_error(void) { return g_err; }
const char *mi_status_string(int s) {
  switch (s) {
    case MI_OK: return "ok";
    case MI_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MI_ERR_HIP: return "HIP runtime error";
    case MI_ERR_OOM: return "out of device memory";
    case MI_ERR_NO_DEVICE: return "no MI355X/HIP device (no CPU fallback)";
    case MI_ERR_COMM: return "RCCL communicator error";
    case MI_ERR_INTERNAL: return "internal error";
  }
  return "unknown status";
}

static const char *kKernelNames[MI_K_COUNT] = {
    "none", "cg_init", "cg_dot3", "cg_scalar_a", "cg_update", "cg_scalar_b", "cg_pupdate",
    "csr_spmm", "stiefel_spmm_gram", "stiefel_gram_reduce", "stiefel_finish_dots",
    "stiefel_retract", "bsr3_spmv_dots", "blas1", "lobpcg_gram", "lobpcg_update",
    "lobpcg_residual", "stiefel_hess_fused", "comm_allreduce", "comm_halo"};
const char *mi_kernel_name(int id) {
  if (id < 0 || id >= MI_K_COUNT) return "?";
  return kKernelNames[id];
}

int mi_ctx_set_option(mi_ctx *ctx, const char *name, long value) {
  MI_REQUIRE(ctx && name, "null argument");
  if (strncmp(name, "MI355OPT_", 9) == 0) name += 9;
  for (const OptionDesc &o : kOptions)
    if (strcmp(o.name, name) == 0) return o.set(ctx, value);
  set_error("unknown option '%s'", name);
  return MI_ERR_INVALID_ARGUMENT;
}

int mi_device_count(int *count) {
  MI_REQUIRE(count, "count is null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    c = 0;
  }
  *count = c;
  return MI_OK;
}

// everything of mi_ctx_create that can fail; the caller destroys the half-built context on error
static int ctx_init(mi_ctx *ctx, int device) {
  ctx->device = device;
  hipDeviceProp_t prop;
  MI_HIP(hipGetDeviceProperties(&prop, device));
  snprintf(ctx->device_name, sizeof(ctx->device_name), "%s (%s, %d CUs)",
           prop.name[0] ? prop.name : "AMD Instinct", prop.gcnArchName, prop.multiProcessorCount);
  ctx->num_cu = prop.multiProcessorCount;
  MI_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  // All small control buffers (partial rows, scalar slots, the two CG states) come out of ONE allocation.  (r05: 4 MB
  // -- the component-major partial buffers grew from 16 to 64 components for Stiefel p <= 8; what a p <= 4 prologue
  // touches, components 0 ... 15 of each buffer, lies where it lay.)
  const size_t pbytes = sizeof(double) * kMaxComps * kMaxRows;
  const size_t slab_bytes = 4u << 20;
  MI_HIP(hipMalloc((void **)&ctx->control_slab, slab_bytes));
  // (on the context's own stream: it is a non-blocking stream, which does not order itself behind the null stream)
  MI_HIP(hipMemsetAsync(ctx->control_slab, 0, slab_bytes, ctx->stream));
  {
    char *top = (char *)ctx->control_slab;
    auto carve = [&](size_t bytes) {
      char *p = top;
      top += (bytes + 255) / 256 * 256;
      return p;
    };
    ctx->cg = (CgState *)carve(2 * sizeof(CgState));
    ctx->cg1 = ctx->cg + 1;
    ctx->scalars = (double *)carve(sizeof(double) * kScalarSlots);
    ctx->partials = (double *)carve(pbytes);
    ctx->partials_b = (double *)carve(pbytes);
    ctx->partials2 = (double *)carve(pbytes);
    ctx->partials_user = (double *)carve(pbytes);
    if ((size_t)(top - (char *)ctx->control_slab) > slab_bytes) {
      set_error("control slab overflow");
      return MI_ERR_INTERNAL;
    }
  }
  MI_HIP(hipHostMalloc((void **)&ctx->host_scalars, sizeof(double) * kScalarSlots,
                       hipHostMallocMapped | hipHostMallocCoherent));  // (written by k_slots_to_host, blas1.hip)
  MI_HIP(hipHostMalloc((void **)&ctx->cg_host, sizeof(CgState) + 16, hipHostMallocMapped | hipHostMallocCoherent)  /* + a preconditioner's failure word; written by k_cg_result_to_host */);
  MI_HIP(hipHostMalloc((void **)&ctx->status, sizeof(HostStatus),
                       hipHostMallocMapped | hipHostMallocCoherent));
  MI_HIP(hipHostMalloc((void **)&ctx->poll_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *ctx->poll_flag = 0;
  memset((void *)ctx->status, 0, sizeof(HostStatus));
  MI_HIP(hipHostGetDevicePointer((void **)&ctx->status_dev, (void *)ctx->status, 0));
  config_from_env(ctx);
  MI_HIP(hipEventCreate(&ctx->t_start));
  MI_HIP(hipEventCreate(&ctx->t_stop));
  return MI_OK;
}

int mi_comm_finalize(mi_ctx *ctx);
int mi_ctx_destroy(mi_ctx *ctx);

int mi_ctx_create(int device, mi_ctx **out) {
  MI_REQUIRE(out, "out is null");
  MI_TRY(ensure_device());
  MI_HIP(hipSetDevice(device));
  mi_ctx *ctx = new mi_ctx();
  const int st = ctx_init(ctx, device);
  if (st != MI_OK) {
    (void)mi_ctx_destroy(ctx);  // tolerates the members that were never created
    return st;
  }
  *out = ctx;
  return MI_OK;
}

int mi_ctx_destroy(mi_ctx *ctx) {
  if (!ctx) return MI_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  (void)mi_comm_finalize(ctx);
  for (auto &kv : ctx->pool_all) (void)hipFree(kv.first);
  for (int i = 0; i < MI_K_COUNT; ++i)
    for (auto &pr : ctx->ktime[i].pending) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < mi_ctx::kStageSlots; ++i) {
    if (ctx->stage_ev[i]) (void)hipEventDestroy(ctx->stage_ev[i]);
    if (ctx->stage_host[i]) (void)hipHostFree(ctx->stage_host[i]);
  }
  (void)hipFree(ctx->control_slab);
  (void)hipFree(ctx->trace_dev);
  (void)hipHostFree(ctx->host_scalars);
  if (ctx->readback_host) (void)hipHostFree(ctx->readback_host);
  if (ctx->poll_flag) (void)hipHostFree(ctx->poll_flag);
  if (ctx->cg_deferred_ev) (void)hipEventDestroy(ctx->cg_deferred_ev);
  (void)hipHostFree(ctx->cg_host);
  (void)hipHostFree((void *)ctx->status);
  if (ctx->t_start) (void)hipEventDestroy(ctx->t_start);
  if (ctx->t_stop) (void)hipEventDestroy(ctx->t_stop);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  (void)hipGetLastError();
  delete ctx;
  return MI_OK;
}

int mi_ctx_sync(mi_ctx *ctx) {
  MI_REQUIRE(ctx, "ctx is null");
  MI_HIP(hipStreamSynchronize(ctx->stream));
  ctx->host_syncs++;
  return MI_OK;
}

int mi_ctx_sync_count(mi_ctx *ctx, size_t *count) {
  MI_REQUIRE(ctx && count, "null argument");
  *count = ctx->host_syncs;
  return MI_OK;
}

int mi_ctx_fusion_counters(mi_ctx *ctx, mi_fusion_counters *out) {
  MI_REQUIRE(ctx && out, "null argument");
  *out = ctx->fusion;
  return MI_OK;
}

int mi_ctx_fusion_counters_reset(mi_ctx *ctx) {
  MI_REQUIRE(ctx, "ctx is null");
  ctx->fusion = mi_fusion_counters{0, 0, 0, 0, 0, 0, 0};
  return MI_OK;
}

int mi_ctx_note_generic(mi_ctx *ctx, int what, const char *why) {
  MI_REQUIRE(ctx, "ctx is null");
  MI_REQUIRE(what >= 0 && what <= 2, "bad kind %d", what);
  static const char *const kKind[3] = {"STPCG", "LSQR", "TNT / GradientDescent trial step"};
  if (what == MI_GENERIC_STPCG) ctx->fusion.generic_stpcg_solves++;
  if (what == MI_GENERIC_LSQR) ctx->fusion.generic_lsqr_solves++;
  if (what == MI_GENERIC_TRIAL) ctx->fusion.generic_trial_steps++;
  if (ctx->warn_generic && !ctx->warned_generic[what]) {
    ctx->warned_generic[what] = true;
    fprintf(stderr, "[mi355opt] note (MI355OPT_WARN_GENERIC): %s on MI355::DeviceVector runs the GENERIC loop (every inner "
                    "product waits for the device): %s\n", kKind[what], why ? why : "a target<>() probe failed");
  }
  return MI_OK;
}

int mi_ctx_stream(mi_ctx *ctx, void **stream) {
  MI_REQUIRE(ctx && stream, "null argument");
  *stream = (void *)ctx->stream;
  return MI_OK;
}

int mi_ctx_device_name(mi_ctx *ctx, char *buf, size_t buflen) {
  MI_REQUIRE(ctx && buf && buflen, "null argument");
  snprintf(buf, buflen, "%s", ctx->device_name);
  return MI_OK;
}

int mi_ctx_pool_bytes(mi_ctx *ctx, size_t *bytes) {
  MI_REQUIRE(ctx && bytes, "null argument");
  *bytes = ctx->pool_bytes;
  return MI_OK;
}

int mi_ktime_enable(mi_ctx *ctx, int id, int on) {
  MI_REQUIRE(ctx && id > 0 && id < MI_K_COUNT, "bad kernel id %d", id);
  ctx->ktime[id].enabled = on != 0;
  return MI_OK;
}

static int ktime_resolve(mi_ctx *ctx, int id) {
  KTimer &t = ctx->ktime[id];
  if (t.pending.empty()) return MI_OK;
  MI_HIP(hipStreamSynchronize(ctx->stream));
  for (auto &pr : t.pending) {
    float ms = 0;
    MI_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
    t.total_ms += ms;
    t.launches++;
    ctx->event_pool.push_back(pr.first);
    ctx->event_pool.push_back(pr.second);
  }
  t.pending.clear();
  return MI_OK;
}

int mi_ktime_reset(mi_ctx *ctx) {
  MI_REQUIRE(ctx, "ctx is null");
  for (int i = 1; i < MI_K_COUNT; ++i) {
    MI_TRY(ktime_resolve(ctx, i));
    ctx->ktime[i].launches = 0;
    ctx->ktime[i].total_ms = 0;
  }
  return MI_OK;
}

int mi_ktime_read(mi_ctx *ctx, int id, size_t *launches, double *total_ms) {
  MI_REQUIRE(ctx && id > 0 && id < MI_K_COUNT, "bad kernel id %d", id);
  MI_TRY(ktime_resolve(ctx, id));
  if (launches) *launches = ctx->ktime[id].launches;
  if (total_ms) *total_ms = ctx->ktime[id].total_ms;
  return MI_OK;
}

namespace {
// roctx is looked up at run time: the library must load (and time nothing) where the tracer is absent
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    // rocprofv3 traces the markers of rocprofiler-sdk's ROCTx; the roctracer one (libroctx64) is the fallback
    void *h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
  }
};
Roctx &roctx() {
  static Roctx r;
  return r;
}
}  // namespace

int mi_range_push(const char *name) {
  MI_REQUIRE(name, "range name is null");
  if (roctx().push) (void)roctx().push(name);
  return MI_OK;
}

int mi_range_pop(void) {
  if (roctx().pop) (void)roctx().pop();
  return MI_OK;
}

int mi_timer_start(mi_ctx *ctx) {
  MI_REQUIRE(ctx, "ctx is null");
  MI_HIP(hipEventRecord(ctx->t_start, ctx->stream));
  return MI_OK;
}

int mi_timer_stop(mi_ctx *ctx, double *ms) {
  MI_REQUIRE(ctx && ms, "null argument");
  MI_HIP(hipEventRecord(ctx->t_stop, ctx->stream));
  MI_HIP(hipEventSynchronize(ctx->t_stop));
  float f = 0;
  MI_HIP(hipEventElapsedTime(&f, ctx->t_start, ctx->t_stop));
  *ms = f;
  return MI_OK;
}

}  // extern "C"
