// spmm_core.h -- sliced-ELL-64 sparse x tall-skinny (n x P, row-major) product core, shared by the
// plain SpMM kernel (sparse.hip) and the fused Stiefel Hessian kernel (stiefel.hip).
#pragma once
#include "mi_internal.h"

namespace mi {

struct SellView {
  size_t n, nslices;
  const long long *__restrict__ slice_ptr;
  const int *__restrict__ col;
  const double *__restrict__ val;
  const double *__restrict__ halo;  // may be null
  const uint32_t *__restrict__ pk;  // value-indexed packed entries (mi_csr::pk), may be null
  const double *__restrict__ vtab;
};

inline SellView sell_view(const mi_csr *A) {
  return SellView{A->n, A->nslices, A->slice_ptr, A->col, A->val, A->halo_cur(), A->pk, A->vtab};
}

// Slice bounds through the CONSTANT address space: the matrix structure is immutable while a kernel runs, but the
// compiler cannot prove that the kernel's own stores do not alias it, so a plain load of a wave-uniform
// slice_ptr[i] becomes a VECTOR load + v_readfirstlane behind an s_waitcnt vmcnt(0) -- the wave drains its whole
// memory queue once per slice.  From address space 4 it is an s_load that is waited for on its own counter.
typedef __attribute__((address_space(4))) const long long ConstLL;
__device__ __forceinline__ long long slice_bound(const long long *sp, size_t i) {
  return ((ConstLL *)sp)[i];
}

#ifndef MI_SPMM_CHUNK
#define MI_SPMM_CHUNK 4  // measured on cfg2: 4 -> 33.4 us, 8 -> 34.1 us, entry-at-a-time -> 35 us
#endif
constexpr int kSlicesPerGroup = kWaves;  // one workgroup pass covers 16 slices = 1024 rows

// acc[0..P) = row `row` of A times V.  One lane per row; a wave owns one slice, so the loads of
// val/col at (base + k) * 64 + lane are contiguous across the wave (512 B + 256 B per k).
template <int P>
__device__ __forceinline__ void sell_row_times(const SellView &A, size_t slice, int lane,
                                               const double *__restrict__ V, double (&acc)[P]) {
  const long long b0 = A.slice_ptr[slice], b1 = A.slice_ptr[slice + 1];
#pragma unroll
  for (int c = 0; c < P; ++c) acc[c] = 0;
  // Chunks of MI_SPMM_CHUNK entries: all (value, column) pairs of a chunk are loaded first, then the
  // CH x P gathers are issued back to back (memory-level parallelism instead of a load -> gather
  // dependency per entry).  Entries beyond the slice width are predicated off by a select.
  // (Non-temporal loads of the matrix stream were measured SLOWER, 40 vs 33 us: the 87 MB matrix is
  // Infinity-Cache resident across iterations and `nt` forfeits that.)
  constexpr int CH = MI_SPMM_CHUNK;
  for (long long k = b0; k < b1; k += CH) {
    double a[CH];
    size_t cidx[CH];
    bool valid[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      valid[j] = k + j < b1;
      const size_t e = (size_t)(valid[j] ? k + j : k) * 64 + lane;
      a[j] = A.val[e];
      cidx[j] = (size_t)A.col[e];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#ifdef MI_DEBUG_NO_GATHER  // timing experiment only: every entry reads the lane's own row
      const double *src = V + (slice * 64 + lane < A.n ? slice * 64 + lane : 0) * P + 0 * cidx[j];
#else
      const double *src = (cidx[j] < A.n) ? (V + cidx[j] * P) : (A.halo + (cidx[j] - A.n) * P);
#endif
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const double t = a[j] * src[c];
        acc[c] += valid[j] ? t : 0.0;  // select, not multiply-by-zero: never manufactures a NaN
      }
    }
  }
}

// Lean, software-pipelined form of the same product for a wave that owns the slices first, first + kWaves,
// ... (< end); used by the one-pass Stiefel Hessian.
//   * 32-bit byte offsets from scalar bases (global_load with saddr) instead of 64-bit multiply-adds per
//     gathered entry: the caller guarantees that V, the halo, the value and the column arrays each span
//     < 4 GiB (sell_stream_ok);
//   * predication once per entry (the VALUE is zeroed), then one multiply and one add per component;
//   * HALO is a template flag, so unsharded matrices carry no column-range select;
//   * the value/column operands of the NEXT chunk -- of this slice or of the wave's next slice -- are
//     requested right after the current chunk's gathers (unconditionally: behind a branch the compiler can no
//     longer count outstanding loads and drains the queue), and the row epilogue's operands one slice ahead:
//       epi.begin(slice)      issue the loads the epilogue of `slice` will need
//       epi.end(slice, acc)   consume acc[0..P) = (A V)(row slice*64+lane, :)
// `first` must be wave-uniform (readfirstlane): slice bounds then come from scalar loads.
// What this does NOT buy on cfg2 is time: with wall_clock64 stamps inside the kernel the
// main loop runs at ~6.4 TB/s, the practical HBM rate, in every variant tried -- ~3x fewer instructions (this
// form), deeper prefetch, chunk sizes 2..8, L1-bypassing matrix loads, an LDS copy of the workgroup's rows of
// V, a column-major V -- and the rest of the 33 us is the prologue reduction and the ramp-down.
// PK: the matrix is read from its value-indexed packed copy (one dword per entry: (col - row) << 8 | value
// index; `vt` = the 256-entry value table, staged in LDS by the caller) -- 4 instead of 12 bytes per entry.
template <int P, bool HALO, bool PK, class Epi>
__device__ __forceinline__ void sell_stream(const SellView &A, size_t first, size_t end, int lane,
                                            const double *__restrict__ V, const double *vt, Epi &epi) {
  constexpr int CH = MI_SPMM_CHUNK;
  if (first >= end) return;
  size_t slice = first;
  long long k = slice_bound(A.slice_ptr, slice), b1 = slice_bound(A.slice_ptr, slice + 1);
  const unsigned lane8 = (unsigned)lane * 8u, lane4 = (unsigned)lane * 4u;
  const unsigned nloc = (unsigned)A.n;
  // operands of a chunk as loaded: PK: packed words (decoded when the chunk becomes current); else value + column
  struct Ops {
    double a[PK ? 1 : CH];
    unsigned c[CH];
  };
  // entries beyond the slice width re-read the chunk's first entry -- or, for an empty slice, entry 0 of
  // the matrix (mi_csr always stores >= 64 entries); their value is zeroed when the chunk is consumed
  auto load_chunk = [&](Ops &o, long long kk, long long bb) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const long long kj = (kk + j < bb) ? kk + j : ((kk < bb) ? kk : 0);
      if (PK) {
        const char *pb = reinterpret_cast<const char *>(A.pk + (size_t)kj * 64);  // scalar base
        o.c[j] = *reinterpret_cast<const unsigned *>(pb + lane4);
      } else {
        const char *vb = reinterpret_cast<const char *>(A.val + (size_t)kj * 64);
        const char *cb = reinterpret_cast<const char *>(A.col + (size_t)kj * 64);
        o.a[j] = *reinterpret_cast<const double *>(vb + lane8);
        o.c[j] = *reinterpret_cast<const unsigned *>(cb + lane4);
      }
    }
  };
  // bounds of the wave's NEXT slice, requested a whole slice ahead: a scalar load issued where its result is
  // needed costs a full memory latency per slice while the vector queue is saturated (measured: a young
  // wave waited 4-6 us for its first pair)
  size_t cand = slice + kWaves;
  long long q0 = k, q1 = b1;
  if (cand < end) { q0 = slice_bound(A.slice_ptr, cand); q1 = slice_bound(A.slice_ptr, cand + 1); }
  Ops cur;
  load_chunk(cur, k, b1);
  epi.begin(slice);
  double acc[P];
#pragma unroll
  for (int c = 0; c < P; ++c) acc[c] = 0;
  for (;;) {
    // the wave's next chunk: all selects on scalars, no branches
    const bool row_done = k + CH >= b1;
    const bool more_slices = cand < end;
    const bool have_next = !row_done || more_slices;
    const size_t nslice = (row_done && more_slices) ? cand : slice;
    const long long nk = row_done ? (more_slices ? q0 : k) : k + CH;
    const long long nb1 = (row_done && more_slices) ? q1 : b1;
    double a[CH];
    unsigned ci[CH];
    if (PK) {
      const unsigned row = (unsigned)(slice * 64) + (unsigned)lane;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        a[j] = vt[cur.c[j] & 255u];
        ci[j] = row + (unsigned)((int)cur.c[j] >> 8);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) { a[j] = cur.a[j]; ci[j] = cur.c[j]; }
    }
    double g[CH][P];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const char *base = reinterpret_cast<const char *>(V);
      unsigned boff = ci[j] * (unsigned)(P * 8);
      if (HALO && ci[j] >= nloc) {
        base = reinterpret_cast<const char *>(A.halo);
        boff = (ci[j] - nloc) * (unsigned)(P * 8);
      }
      const double *src = reinterpret_cast<const double *>(base + boff);
#pragma unroll
      for (int c = 0; c < P; ++c) g[j][c] = src[c];
    }
    Ops nxt;
    load_chunk(nxt, nk, nb1);  // (re-reads the first chunk of the last slice when nothing follows)
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const double aj = (k + j < b1) ? a[j] : 0.0;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        // product and sum rounded separately, entry by entry in storage order: bit-identical to a plain CSR
        // loop on the host (the checks of the sharded products rely on that)
#pragma clang fp contract(off)
        const double t = aj * g[j][c];
        acc[c] = acc[c] + t;
      }
    }
    if (row_done) {
      epi.end(slice, acc);
#pragma unroll
      for (int c = 0; c < P; ++c) acc[c] = 0;
      if (have_next) epi.begin(nslice);
    }
    if (!have_next) break;
    if (row_done) {  // moved on to `cand`: request the bounds of the slice after it
      cand += kWaves;
      if (cand < end) { q0 = slice_bound(A.slice_ptr, cand); q1 = slice_bound(A.slice_ptr, cand + 1); }
    }
    slice = nslice; k = nk; b1 = nb1;
    cur = nxt;
  }
}

// ---- LDS-window form of sell_stream ---------------------------------------------------------------------------
// What bounds the one-pass Hessian on cfg2 (in-kernel shader-clock stamps, tools/stamps.py): not
// bytes and not the gathers themselves but EXPOSED LATENCY.  A wave of sell_stream keeps one 4-entry chunk of
// matrix words in flight; they take ~2100 cycles from request to arrival and the wave needs them twice per
// slice, and because vector loads return in order every wait for a just-issued gather also waits for every
// prefetch issued before it.  This form removes the dependent global loads from the steady state:
//   * the rows of V NEAR the workgroup's own are staged once per workgroup in an LDS ring (coalesced 512-byte
//     loads) and near entries are gathered from there (conflict-free ds_reads, no vector-memory queue);
//   * each row's few FAR entries (the +-nx*ny neighbours of a 3-D stencil, halo columns; at most kFarCap per
//     row on the fast path) are gathered first thing in a slice -- and only THEN are the next slice's matrix
//     words and epilogue rows requested, so the wait for the gathers (s_waitcnt vmcnt(#later loads)) never
//     waits for a prefetch, and every prefetched operand has a whole slice time (~10k cycles at the bandwidth
//     bound) to arrive;
//   * the per-slice code is straight-line (branch-free selects between the gathered and the staged value), so
//     the compiler's s_waitcnt counts are exact instead of the conservative vmcnt(0) a loop back-edge forces.
// The arithmetic per row is unchanged: same entries, same order, same separately rounded products and sums --
// results are bit-identical to sell_stream / sell_row_times.
//
//   ring     kRingChunks chunks of 64 rows x P doubles; chunk q (rows 64q..64q+63) lives at slot q & 63, so
//            row c is at ring[(c & 4095) * P].
//   tiles    the workgroup's slices [s0, s1) are taken 16 at a time (wave w owns slice s0 + 16 t + w of tile t);
//            tile t needs the chunks [s0 + 16 t - wc, s0 + 16 t + 16 + wc).  Behind tile t every wave stages ONE
//            more chunk (loads issued before its slice, LDS write after it), then one LDS-only barrier
//            (s_waitcnt lgkmcnt + s_barrier: global loads stay in flight across it).  Waves that wait at the
//            barrier have their next slice's 6 KB in flight, so the memory system does not run dry meanwhile.
//   hazards  the write behind tile t overwrites chunk q - 64 with q < s0 + 16 (t + 2) + wc, i.e. below
//            s0 + 16 t - wc as long as wc <= kMaxWinChunks = 16: nothing a wave still in tile t can need.
//   near     entry (row r, column c) is served from the ring iff |c - r| <= 64 wc (then chunk(c) is within wc
//            chunks of the wave's own slice, hence staged) and c is a local row (halo columns are not in V).
//   eligible only matrices whose slices are all at most kWinHead entries wide, whose rows have at most kFarCap far
//            entries each, and which have a packed copy (mi_csr::pk) take this form (decided at creation,
//            sparse.hip window_chunks): that keeps a slice's code free of loops and branches, and the two sets
//            of prefetched operands small enough for 128 registers.  Every other matrix keeps sell_stream.
// `w` must be wave-uniform (readfirstlane).  Epilogue protocol: epi.request(slice) issues the loads the epilogue
// of `slice` needs (one slice ahead), epi.park() moves them from registers to wave-private LDS at the top of that
// slice, epi.end(slice, acc, vrow) consumes them together with acc = (A V)(row,:) and vrow = V(row,:) (read from
// the ring: no global read of the row).
#ifndef MI_WIN_LDS_BATCH
#define MI_WIN_LDS_BATCH 4
#endif
constexpr int kRingChunks = 64;
constexpr int kMaxWinChunks = 16;
constexpr int kFarCap = 2;
constexpr int kWinHead = 8;

// In-kernel timeline (experiment builds only, -DMI_WIN_STAMPS): shader-clock stamps taken once a given value has
// arrived, kept in LDS per wave and dumped by the kernel at its end (tools/stamps.py).
#ifdef MI_WIN_STAMPS
constexpr int kStampSlots = 64;
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ __forceinline__ void stamp_put(unsigned long long *area, int slot, double dep) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  if (slot < kStampSlots && (threadIdx.x & 63) == 0) area[slot] = t;
}
#define MI_STAMP(slot, dep) stamp_put(stamp_area, (slot), (double)(dep))
__device__ __forceinline__ void stamp_real(unsigned long long *area, int slot) {  // the constant 100 MHz clock
  const unsigned long long t = wall_clock64();
  if ((threadIdx.x & 63) == 0) area[slot] = t;
}
#else
#define MI_STAMP(slot, dep) ((void)0)
#endif

// A load the compiler keeps where it is written: a relaxed atomic load at wavefront scope is a plain global_load
// (no cache-policy bits at that scope) but "ordered" -- neither the IR passes (which sink ordinary loads to their
// first use and hoist address-ready ones) nor the machine scheduler move memory operations across it.  The window
// kernels' queue discipline (what is issued before what, so that s_waitcnt vmcnt(N) never waits for a prefetch)
// rests on it.
template <class T>
__device__ __forceinline__ T pinned_load(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

typedef __attribute__((address_space(3))) double LdsDouble;
typedef __attribute__((address_space(1))) double GlobalDouble;

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int P, bool HALO, bool PK, class Epi>
__device__ __forceinline__ void sell_window(const SellView &A, int wc, size_t s0_, size_t s1_, int w, int lane,
                                            const double *__restrict__ V, const double *vt, double *ring,
                                            Epi &epi) {
  static_assert(PK, "the window form reads the packed matrix copy");
#ifdef MI_WIN_DEBUG  // timing experiments only (wrong results): flags in the high bits of wc
  const int dbg = wc >> 8;
  wc &= 255;
#else
  constexpr int dbg = 0;
#endif
  constexpr int HW = kWinHead;  // entries per slice (raw operands of TWO slices are held in registers)
  // 32-bit indices throughout (sell_stream_ok: < 2^29 stored entries, < 2^26 slices): 64-bit slice numbers and
  // bounds cost a register pair each, and the scalar file is what overflows first in this kernel
  const int nchunks = (int)A.nslices;
  const int s0 = (int)s0_, s1 = (int)s1_;
  const int ntiles = (s1 - s0 + kWaves - 1) / kWaves;
  const unsigned nloc = (unsigned)A.n;
  const unsigned nPbytes = (unsigned)(A.n * P * 8);
  const unsigned lane8 = (unsigned)lane * 8u, lane4 = (unsigned)lane * 4u;
  const unsigned wr = 64u * (unsigned)wc;
  constexpr unsigned kRingMask = (unsigned)(kRingChunks * 64 - 1);
#ifdef MI_WIN_STAMPS
  __shared__ unsigned long long stamp_lds[kWaves * kStampSlots];
  unsigned long long *stamp_area = stamp_lds + w * kStampSlots;
  if (lane == 0)
    for (int i = 0; i < kStampSlots; ++i) stamp_area[i] = 0;
#endif
  MI_STAMP(0, 0.0);
#ifdef MI_WIN_STAMPS
  stamp_real(stamp_area, kStampSlots - 3);
#endif
  // chunk q of V: lane l takes the doubles l, 64 + l, ... of its 64 P doubles (P coalesced 512-byte loads)
  auto chunk_load = [&](int q, double (&buf)[P]) {
    const unsigned b0 = (unsigned)q * (unsigned)(64 * P * 8) + lane8;
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const unsigned b = b0 + (unsigned)c * 512u;
      const unsigned bs = b < nPbytes ? b : 0u;  // past the last row: any valid address (never referenced)
      buf[c] = pinned_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(V) + bs));
    }
  };
  auto chunk_store = [&](int q, const double (&buf)[P]) {
    double *dst = ring + ((unsigned)q & (unsigned)(kRingChunks - 1)) * (unsigned)(64 * P) + lane;
#pragma unroll
    for (int c = 0; c < P; ++c) dst[c * 64] = buf[c];
  };
  // chunk staged behind tile i (tile i + 1 needs it); -1: none for this wave
  auto job_chunk = [&](int i) -> int {
    const int q = s0 + kWaves * (i + 1) + wc + w;
    return (i + 1 < ntiles && q < nchunks && q < s1 + wc) ? q : -1;
  };
  // raw operands of a slice's head entries; entries beyond the slice width re-read its first entry (or entry 0
  // of the matrix for an empty slice: mi_csr always stores >= 64 entries) and are zeroed when consumed
  struct Head {
    unsigned c[HW];
  };
  // ONE scalar base + immediate offsets j * 256: the words behind a narrower slice belong to the next slice (or to
  // the kWinHead rows of zero padding mi_csr keeps behind the packed copy) and are masked off when consumed
  auto load_head = [&](Head &h, int kk) {
    const char *pb = reinterpret_cast<const char *>(A.pk) + (unsigned)kk * 256u + lane4;
#pragma unroll
    for (int j = 0; j < HW; ++j) h.c[j] = pinned_load(reinterpret_cast<const unsigned *>(pb + j * 256));
  };
  // row `c` of V (or of the halo) through the vector-memory path
  auto far_row = [&](unsigned c, double (&out)[P]) {
    const char *base = reinterpret_cast<const char *>(V);
    unsigned boff = c * (unsigned)(P * 8);
    if (HALO && c >= nloc) {
      base = reinterpret_cast<const char *>(A.halo);
      boff = (c - nloc) * (unsigned)(P * 8);
    }
    const GlobalDouble *src = (const GlobalDouble *)reinterpret_cast<const double *>(base + boff);
#pragma unroll
    for (int cc = 0; cc < P; ++cc) out[cc] = src[cc];
  };
  // the same with pinned loads (the far gathers of the slice loop)
  auto far_row_pinned = [&](unsigned c, double (&out)[P]) {
    const char *base = reinterpret_cast<const char *>(V);
    unsigned boff = c * (unsigned)(P * 8);
    if (HALO && c >= nloc) {
      base = reinterpret_cast<const char *>(A.halo);
      boff = (c - nloc) * (unsigned)(P * 8);
    }
    const double *src = reinterpret_cast<const double *>(base + boff);
#pragma unroll
    for (int cc = 0; cc < P; ++cc) out[cc] = pinned_load(src + cc);
  };
  auto near_row = [&](unsigned c, double (&out)[P]) {
    const LdsDouble *l = (const LdsDouble *)ring + (c & kRingMask) * (unsigned)P;
#pragma unroll
    for (int cc = 0; cc < P; ++cc) out[cc] = l[cc];
  };

  // the wave's first slice: matrix operands and epilogue rows requested before the ring is filled
  int slice = s0 + w;
  bool have = slice < s1;
  int k = 0, b1 = 0;
  if (have) {
    k = (int)slice_bound(A.slice_ptr, slice);
    b1 = (int)slice_bound(A.slice_ptr, slice + 1);
  }
  // ONE register set for the prefetched operands (raw matrix words here, epilogue rows in the Epi), loop-carried:
  // at the top of a slice they are turned into something else (columns + packed value indices; the rows are parked
  // in wave-private LDS by epi.park()) -- work that needs the data and therefore stays at the top, where the loads
  // are a slice old -- and the same registers then receive the next slice's loads.  (A plain copy `cur = next`
  // is placed by the compiler at the BOTTOM of the previous slice, right behind the loads' issue: a wait for the
  // whole prefetch in every slice.)
  Head h;
  load_head(h, k);  // (no slice: entry 0 of the matrix, never consumed)
  epi.request(have ? slice : s0);
  {
    const int lo = s0 - wc;
    int hi = s0 + kWaves + wc;
    if (hi > nchunks) hi = nchunks;
    if (hi > s1 + wc) hi = s1 + wc;
    for (int q = lo + w; q < hi; q += kWaves) {
      if (q < 0) continue;
      double b[P];
      chunk_load(q, b);
      chunk_store(q, b);
    }
  }
  lds_barrier();
  MI_STAMP(1, 0.0);

  for (int t = 0; t < ntiles; ++t) {
    MI_STAMP(2 + 8 * t, 0.0);
    const int qj = job_chunk(t);
    double pre[P] = {};
    if (!(dbg & 16)) chunk_load(qj >= 0 ? qj : 0, pre);  // unconditional: the s_waitcnt counts below stay exact on every path
    if (have) {
      const int nslice = slice + kWaves;
      const bool more = nslice < s1;
      int q0 = k, q1 = b1;  // nothing follows: the prefetch re-reads this slice
      if (more) { q0 = (int)slice_bound(A.slice_ptr, nslice); q1 = (int)slice_bound(A.slice_ptr, nslice + 1); }
      const unsigned row = (unsigned)(slice * 64) + (unsigned)lane;
      // ---- the slice's words -> columns, packed value indices, this lane's (<= kFarCap) far entries ----------
      // what stays of the eight words: per entry a 16-bit row offset into the ring (0 for far entries and beyond
      // the slice width: the lane's own row, read and discarded / multiplied by a zero value) and an 8-bit value
      // index, packed two and four to a register; the columns of the (<= 2) far entries
      unsigned dpk[HW / 2], vix[HW / 4];
      int f0 = HW, f1 = HW, nf = 0;
      unsigned cf0 = row < nloc ? row : 0u, cf1 = cf0;  // lanes without a far entry re-read their own row
#pragma unroll
      for (int j = 0; j < HW / 4; ++j) vix[j] = 0;
#pragma unroll
      for (int j = 0; j < HW / 2; ++j) dpk[j] = 0;
      static_assert(kFarCap == 2, "two far slots per row (f0, f1)");
#pragma unroll
      for (int j = 0; j < HW; ++j) {
        const int d = (int)h.c[j] >> 8;
        vix[j / 4] |= (h.c[j] & 255u) << (8 * (j % 4));
        const bool valid = k + j < b1;  // wave-uniform
        const bool nearj = (unsigned)(d + (int)wr) <= 2u * wr && (!HALO || row + (unsigned)d < nloc);
        const bool far = valid && !nearj;
        const bool first = far && nf == 0, second = far && nf == 1;
        f0 = first ? j : f0;
        f1 = second ? j : f1;
        cf0 = first ? row + (unsigned)d : cf0;
        cf1 = second ? row + (unsigned)d : cf1;
        nf += far ? 1 : 0;
        const unsigned dn = (valid && nearj) ? ((unsigned)d & 0xffffu) : 0u;  // |d| <= 64 wc <= 1024
        dpk[j / 2] |= dn << (16 * (j % 2));
      }
      epi.park();  // the epilogue rows of THIS slice: registers -> wave-private LDS
      if (dbg & 1) { cf0 = row; cf1 = row; }  // (experiment: no far gathers)
      double gf0[P], gf1[P];
      far_row_pinned(cf0, gf0);
      far_row_pinned(cf1, gf1);
      MI_STAMP(2 + 8 * t + 1, (double)vix[0]);  // this slice's matrix words were there
      // ---- only now the next slice's operands (newer than the gathers: waiting for those never waits for these)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 8)) load_head(h, q0);
      if (!(dbg & 2)) epi.request(more ? nslice : slice);
      // ---- consume the entries in storage order (straight-line: no branch, no loop up to the epilogue) -----
      double acc[P];
#pragma unroll
      for (int c = 0; c < P; ++c) acc[c] = 0;
      // Every product below depends on the gathers (through the selects), so the compiler would issue ALL ring
      // reads and value lookups first and only then wait (64 registers: the accumulators spill).  Tie the ring
      // addresses to the gathers' arrival instead: an opaque zero that exists once both have landed.  The ring
      // reads are then issued MI_WIN_LDS_BATCH entries at a time, right where they are consumed.
      unsigned rowz;
      asm volatile("v_mov_b32 %0, %1" : "=v"(rowz) : "v"(row), "v"(gf0[P - 1]), "v"(gf1[P - 1]));
#pragma unroll
      for (int j = 0; j < HW; ++j) {
        const double aj = (k + j < b1) ? vt[(vix[j / 4] >> (8 * (j % 4))) & 255u] : 0.0;
        double g[P];
        near_row(rowz + (unsigned)(int)(short)(dpk[j / 2] >> (16 * (j % 2))), g);
#pragma unroll
        for (int c = 0; c < P; ++c) {
          g[c] = (j == f0) ? gf0[c] : g[c];
          g[c] = (j == f1) ? gf1[c] : g[c];
        }
#pragma unroll
        for (int c = 0; c < P; ++c) {
#pragma clang fp contract(off)
          const double tt = aj * g[c];
          acc[c] = acc[c] + tt;
        }
        if (j == 0) MI_STAMP(2 + 8 * t + 2, acc[0]);  // the far gathers have arrived
        if ((j % MI_WIN_LDS_BATCH) == MI_WIN_LDS_BATCH - 1 && j + 1 < HW) {
          // the next batch's ring addresses "depend" on this batch's sums
          asm volatile("v_mov_b32 %0, %1" : "=v"(rowz) : "v"(rowz), "v"(acc[0]), "v"(acc[P - 1]));
        }
      }
      MI_STAMP(2 + 8 * t + 4, acc[0]);
      // the staged chunk goes into the ring here: its loads are the oldest in the queue (eighteen newer ones),
      // and its slot holds a chunk no wave still in this tile can need (hazards, above)
      if (qj >= 0) chunk_store(qj, pre);
      {
        double vrow[P];
        near_row(row, vrow);
        if (!(dbg & 4)) epi.end(slice, acc, vrow);
        MI_STAMP(2 + 8 * t + 5, acc[0]);  // epilogue done, stores issued
      }
      have = more;
      slice = nslice;
      k = q0;
      b1 = q1;
    }
    else if (qj >= 0) {
      chunk_store(qj, pre);  // a wave without a slice in this tile
    }
    if (t + 1 < ntiles) {
      MI_STAMP(2 + 8 * t + 6, 0.0);
      lds_barrier();
      MI_STAMP(2 + 8 * t + 7, 0.0);  // barrier passed
    }
  }
#ifdef MI_WIN_STAMPS
  MI_STAMP(kStampSlots - 1, 0.0);
  stamp_real(stamp_area, kStampSlots - 2);
  if (g_stamp_buf && (lane < 56 || lane > 58))
    g_stamp_buf[((size_t)blockIdx.x * kWaves + w) * kStampSlots + lane] = stamp_area[lane];
#endif
}

// sell_stream's 32-bit offsets: every array it indexes must span < 4 GiB
inline bool sell_stream_ok(const mi_csr *A, int p) {
  const size_t lim = (size_t)1 << 32;
  return (A->n + A->halo_lo + A->halo_hi + 64) * (size_t)p * 8 < lim && A->padded * 8 < lim;
}

// Workgroup -> contiguous range of slice groups, XCD-aware: the workgroups of one XCD cover one
// contiguous row range, so the gathers of V (rows i +- 1, +- nx, +- nx*ny of a stencil) stay inside
// that XCD's L2.
__device__ __forceinline__ void group_range(size_t ngroups, size_t &g0, size_t &g1) {
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  g0 = (ngroups * lb) / nb;
  g1 = (ngroups * (lb + 1)) / nb;
}

inline size_t sell_groups(const mi_csr *A) { return (A->nslices + kSlicesPerGroup - 1) / kSlicesPerGroup; }

}  // namespace mi
