// spmm_core.h -- sliced-ELL-64 sparse x tall-skinny (n x P, row-major) product core, shared by the
// plain SpMM kernel (sparse.hip) and the fused Stiefel Hessian kernel (stiefel.hip).
#pragma once
#include "mi_internal.h"
#include <type_traits>

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
typedef __attribute__((address_space(4))) const int ConstInt;
__device__ __forceinline__ int scalar_int(const int *p, size_t i) { return ((ConstInt *)p)[i]; }

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
// NW: the waves of the workgroup (the wave takes every NW-th slice of [first, end)).
// CHK: entries per chunk (rows of 8 doubles gather 64 bytes per entry: two of them in flight per lane keep the
// register cost of a chunk what four 24-byte gathers cost).
template <int P, bool HALO, bool PK, class Epi, int NW = kWaves, int CHK = MI_SPMM_CHUNK>
__device__ __forceinline__ void sell_stream(const SellView &A, size_t first, size_t end, int lane,
                                            const double *__restrict__ V, const double *vt, Epi &epi) {
  constexpr int CH = CHK;
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
  size_t cand = slice + NW;
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
        // (an EMPTY slice parks its loads on entry 0 of the matrix, whose column offsets belong to other rows: the own row)
        ci[j] = k < b1 ? row + (unsigned)((int)cur.c[j] >> 8) : (row < nloc ? row : nloc - 1u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) { a[j] = cur.a[j]; ci[j] = cur.c[j]; }
    }
    double g[CH][P];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const char *base = reinterpret_cast<const char *>(V);
#ifdef MI_ABLATE_GATHER_OWN  // (timing experiment only: every entry reads the lane's own row -- wrong results)
      ci[j] = (unsigned)(slice * 64) + (unsigned)lane < nloc ? (unsigned)(slice * 64) + (unsigned)lane : 0u;
#endif
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
      cand += NW;
      if (cand < end) { q0 = slice_bound(A.slice_ptr, cand); q1 = slice_bound(A.slice_ptr, cand + 1); }
    }
    slice = nslice; k = nk; b1 = nb1;
    cur = nxt;
  }
}

// ---- quad layout of sell_stream for rows of 5 ... 8 doubles ----------------------------------------------------
// With one lane per row a 64-byte row makes every 16-byte load of a wave touch 64 different half-lines: measured on
// St(1e6,8) (profiles/r05_wide_ablation.txt) the L1 then delivers about one LANE per clock -- a pass that only loads
// the direction's rows eight times, all of them L1 hits, takes 51 us of the wide Hessian's 101.  Here a QUAD of lanes
// owns a row: lane (g, q) = (lane >> 2, lane & 3) holds columns 2q, 2q + 1 of row 16 t + g of the wave's slice, and
// the wave goes through its slice in four UNITS t = 0..3 of 16 rows, so a quad reads 64 contiguous bytes and a wave
// instruction covers 16 whole rows -- consecutive rows (own rows, +-1 neighbours) are one contiguous kilobyte.  The
// matrix words of a unit are 16 consecutive words of the sliced-ELL chunk (one 64-byte line, each read by the four
// lanes of its quad).  Columns >= P (P < 8) are loaded from a valid address and never used: the epilogue zeroes them.
//       epi.begin(slice, t, x, v, y)        issue the loads the epilogue of the unit will need (into x, v, y)
//       epi.end(slice, t, acc, x, v, y)     consume acc[c] = (A V)(row slice*64 + 16 t + g, column 2 q + c)
// Same arithmetic per element as sell_stream: product and sum rounded separately, entries in storage order.
template <int P>
struct QuadCols {
  // byte offsets of the lane's two columns inside a row (a column that does not exist re-reads the lane's first,
  // or column 0 for a lane without columns)
  __device__ __forceinline__ static unsigned off0(int q) { return (unsigned)((2 * q < P ? 2 * q : 0) * 8); }
  __device__ __forceinline__ static unsigned off1(int q) { return (unsigned)((2 * q + 1 < P ? 2 * q + 1 : (2 * q < P ? 2 * q : 0)) * 8); }
  // (one 32-bit byte offset from the field's base -- a scalar -- per access: global_load with saddr, no 64-bit adds)
  __device__ __forceinline__ static void load(const char *base, unsigned row_off, unsigned o0, unsigned o1, double &d0,
                                              double &d1) {
    if constexpr (P % 2 == 0) {  // rows are 16-byte aligned and the two columns adjacent (or both unused)
      const double2 d = *reinterpret_cast<const double2 *>(base + (row_off + o0));
      d0 = d.x; d1 = d.y;
    } else {
      d0 = *reinterpret_cast<const double *>(base + (row_off + o0));
      d1 = *reinterpret_cast<const double *>(base + (row_off + o1));
    }
  }
};

// Software pipeline over the wave's CHUNK STEPS (unit by unit, CH entries each; a stencil row is one step at CH = 8):
// while step s is consumed, the row pieces of step s + 1 (its gathers and the unit's own rows of X, V, Y) and the
// matrix words of step s + 2 are in flight.  Every step issues the SAME loads in the same order -- own rows too, again
// for every further chunk of a long row -- because the wait counters are in-order and counted statically: loads behind
// a branch make the compiler drain the queue where the paths meet (the lane-per-row form does, at every slice
// boundary, and loses the prefetch of its epilogue's operands there).
template <int P, bool HALO, bool PK, class Epi, int NW, int CHK>
__device__ __forceinline__ void sell_stream_quad(const SellView &A, size_t first, size_t end, int lane,
                                                 const double *__restrict__ V, const double *vt, Epi &epi) {
  constexpr int CH = CHK;
  if (first >= end) return;
  const int q = lane & 3, g = lane >> 2;
  const unsigned o0 = QuadCols<P>::off0(q), o1 = QuadCols<P>::off1(q);
  const unsigned nloc = (unsigned)A.n;
  struct Pos {  // one chunk step: entries [k, k + CH) of unit t of `slice` (whose entries are [k0, b1))
    size_t slice;
    int t;
    long long k, k0, b1;
    bool valid;
  };
  struct Words {
    double a[PK ? 1 : CH];
    unsigned c[CH];
  };
  struct Step {  // what a step has in flight: its gathered row pieces, the own rows of its unit, its matrix values
    double gv[CH][2];
    double x[2], v[2], y[2];
    Words w;
  };
  size_t cand = first + NW;
  long long q0 = 0, q1 = 0;
  if (cand < end) { q0 = slice_bound(A.slice_ptr, cand); q1 = slice_bound(A.slice_ptr, cand + 1); }
  // the step after p; past the last one: the same position again, marked invalid (its loads re-read valid addresses)
  auto advance = [&](const Pos &p) {
    Pos n = p;
    if (!p.valid) return n;
    if (p.k + CH < p.b1) { n.k = p.k + CH; return n; }
    if (p.t < 3) { n.t = p.t + 1; n.k = p.k0; return n; }
    if (cand < end) {
      n.slice = cand; n.t = 0; n.k0 = n.k = q0; n.b1 = q1;
      cand += NW;  // (the bounds of the slice after it: requested a whole slice ahead)
      if (cand < end) { q0 = slice_bound(A.slice_ptr, cand); q1 = slice_bound(A.slice_ptr, cand + 1); }
      return n;
    }
    n.valid = false;
    return n;
  };
  // (entries beyond the slice width re-read the chunk's first entry -- or entry 0 of the matrix for an empty slice --
  // and are zeroed when the chunk is consumed: sell_stream)
  auto load_words = [&](Words &o, const Pos &p) {
    const unsigned r4 = (unsigned)(16 * p.t + g) * 4u, r8 = (unsigned)(16 * p.t + g) * 8u;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const long long kj = (p.k + j < p.b1) ? p.k + j : ((p.k < p.b1) ? p.k : 0);
      if (PK) {
        const char *pb = reinterpret_cast<const char *>(A.pk + (size_t)kj * 64);
        o.c[j] = *reinterpret_cast<const unsigned *>(pb + r4);
      } else {
        const char *vb = reinterpret_cast<const char *>(A.val + (size_t)kj * 64);
        const char *cb = reinterpret_cast<const char *>(A.col + (size_t)kj * 64);
        o.a[j] = *reinterpret_cast<const double *>(vb + r8);
        o.c[j] = *reinterpret_cast<const unsigned *>(cb + r4);
      }
    }
  };
  // the words of step p have arrived in `wd`: issue its gathers and the own rows of its unit into `s`
  auto issue = [&](Step &s, const Words &wd, const Pos &p) {
    s.w = wd;
    const unsigned row = (unsigned)(p.slice * 64) + (unsigned)(16 * p.t + g);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      // (an EMPTY slice parks its loads on entry 0 of the matrix, whose column offsets belong to other rows: the own row)
      const unsigned ci = PK ? (p.k < p.b1 ? row + (unsigned)((int)wd.c[j] >> 8) : (row < nloc ? row : nloc - 1u)) : wd.c[j];
      const char *base = reinterpret_cast<const char *>(V);
      unsigned boff = ci * (unsigned)(P * 8);
      if (HALO && ci >= nloc) {
        base = reinterpret_cast<const char *>(A.halo);
        boff = (ci - nloc) * (unsigned)(P * 8);
      }
      QuadCols<P>::load(base, boff, o0, o1, s.gv[j][0], s.gv[j][1]);
    }
    epi.begin(p.slice, p.t, s.x, s.v, s.y);
  };
  double acc[2] = {0, 0};
  auto consume = [&](Step &s, const Pos &p) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const double av = PK ? vt[s.w.c[j] & 255u] : s.w.a[j];
      const double aj = (p.k + j < p.b1) ? av : 0.0;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma clang fp contract(off)
        const double pr = aj * s.gv[j][c];
        acc[c] = acc[c] + pr;
      }
    }
    if (p.k + CH >= p.b1) {  // the unit's last chunk
      epi.end(p.slice, p.t, acc, s.x, s.v, s.y);
      acc[0] = acc[1] = 0;
    }
  };
  Pos pc;
  pc.slice = first; pc.t = 0; pc.valid = true;
  pc.k0 = pc.k = slice_bound(A.slice_ptr, first);
  pc.b1 = slice_bound(A.slice_ptr, first + 1);
  Words wd;
  Step sa, sb;
  load_words(wd, pc);
  issue(sa, wd, pc);
  Pos pn = advance(pc);
  load_words(wd, pn);
  for (;;) {
    {  // consume sa (step pc) while sb (step pn) and the words of the step after it are in flight
      issue(sb, wd, pn);
      const Pos pw = advance(pn);
      load_words(wd, pw);
      consume(sa, pc);
      if (!pn.valid) break;
      pc = pn; pn = pw;
    }
    {
      issue(sa, wd, pn);
      const Pos pw = advance(pn);
      load_words(wd, pw);
      consume(sb, pc);
      if (!pn.valid) break;
      pc = pn; pn = pw;
    }
  }
}

// ---- LDS-window form of sell_stream ---------------------------------------------------------------------------
// What bounds the one-pass Hessian on cfg2 (in-kernel stamps, tools/stamps.py) is not bytes but
// EXPOSED LATENCY: a wave of sell_stream keeps one 4-entry chunk of matrix words in flight, they take ~2100 cycles
// from request to arrival, the wave needs them twice per slice, and because vector loads return in order every wait
// for a just-issued gather also waits for every prefetch issued before it.  The first window form (gathers from an
// LDS ring, decoded on the fly) removed those waits and ran SLOWER: with the loads out of the way it was bound by its
// own ~700 VALU instructions per slice (classifying entries, unpacking, selecting between gathered and staged
// values).  This form therefore moves every decision to the HOST, once per matrix (sparse.hip build_window):
//
//   wk      one dword per stored entry, the same sliced-ELL positions as mi_csr::pk: (LDS row index << 8) | value
//           index.  The row index says where the kernel will find the entry's row of V in LDS: a ring position for
//           a NEAR entry (|col - row| <= 64 wc, a local column), one of the wave's two FAR SLOTS otherwise.
//   wfar    two columns per row (slice-major, 64 lanes contiguous): the far entries' columns (the row itself when
//           it has fewer), gathered first thing in a slice and written to the far slots.
//   per entry the kernel does: word -> two LDS addresses (5 integer instructions), one value read, one row read,
//           P multiplies and P adds.  No compare, no select, no per-lane case distinction.
//
//   waves    a workgroup of the window kernels has kWinWaves = 4 waves (256 threads) and four of them share a CU:
//            the per-tile barrier keeps a workgroup's waves in the same phase (all waiting for memory, then all
//            computing), so with ONE workgroup per CU loads and arithmetic add up instead of overlapping (measured:
//            17.8 us with every load removed + 11.5 us of loads = the 29.4 us of the 16-wave version; 8 waves x 2:
//            26.3 us; 4 waves x 4: 26.0 us, and 210 instead of 235 us on St(8e6,3), beyond the Infinity Cache).
//   ring     nc = 2 kWinWaves + 2 wc chunks of 64 rows; chunk q lives at slot q % nc.  Row index nc * 64 is the ZERO ROW
//            (zeros): the word of an entry beyond a slice's width is replaced by `zw` (zero row, index of 0.0).
//   far      row index nc * 64 + 1 + (slice % kWinWaves) * 128 + slot * 64 + lane: wave-private, so no barrier is
//            needed between writing and reading them.
//   tiles    aligned groups of NW = kWinWaves slices; wave w owns slice NW T + w of tile T, a workgroup owns a
//            contiguous run of whole tiles (every wave of every workgroup but the last does the same number of
//            slices).  Tile T needs the chunks [NW T - wc, NW T + NW + wc); during tile T every wave stages ONE chunk of tile T + 1
//            (pinned loads at the top, LDS write behind the entries), then one LDS-only barrier (s_waitcnt lgkmcnt
//            + s_barrier: global loads stay in flight across it).
//   hazards  staging chunk q <= NW T + 2 NW - 1 + wc overwrites chunk q - nc <= NW T - 1 - wc, below everything a wave
//            in tile T can need (all waves are in tile T between the barriers).
//   queue    per slice, in this order and kept so by pinned (wavefront-scope atomic = plain, but ordered) loads:
//            far gathers; next slice's words and far columns; this slice's epilogue rows (epi.request); s_waitcnt
//            for the gathers only (vmcnt(#later loads)); entries; staged chunk -> ring; epilogue.  No wait covers a
//            load issued after the one it needs, except the epilogue's (the rows it waits for are the newest).
//   eligible matrices with a packed copy, every slice at most kWinHead entries wide, every row at most kFarCap far
//            entries, wc <= kMaxWinChunks (decided at creation); every other matrix keeps sell_stream.
// Same entries, same order, same separately rounded products and sums as sell_stream / sell_row_times: results are
// bit-identical.  `w` must be wave-uniform (readfirstlane).  Epilogue protocol: epi.request(slice) issues the loads
// the epilogue of `slice` needs, epi.end(slice, acc, vrow) consumes them with acc = (A V)(row,:), vrow = V(row,:).
#ifdef MI_WIN_STAMPS
constexpr int kMaxWinChunks = 2;  // (the stamp area needs the LDS of the wider rings)
#else
constexpr int kMaxWinChunks = 4;
#endif
constexpr int kFarCap = 2;
constexpr int kWinHead = 8;
#ifndef MI_WIN_WAVES
#define MI_WIN_WAVES 4
#endif
constexpr int kWinWaves = MI_WIN_WAVES;  // waves per workgroup of the window kernels = slices per tile
constexpr int kWinBlock = kWinWaves * 64;
constexpr int kWinRingRowsMax = (2 * kWinWaves + 2 * kMaxWinChunks) * 64;  // ring rows at the widest window
constexpr int kWinFarRows = kWinWaves * kFarCap * 64;                       // far slots of a workgroup
constexpr int kWinLdsRows = kWinRingRowsMax + 1 + kWinFarRows;   // ring, zero row, far slots

// In-kernel timeline (experiment builds only, -DMI_WIN_STAMPS): shader-clock stamps taken once a given value has
// arrived, kept in LDS per wave and dumped by the kernel at its end (tools/stamps.py).
#ifdef MI_WIN_STAMPS
constexpr int kStampSlots = 64;     // per wave in the dump buffer
constexpr int kStampLdsSlots = 48;  // of which the window core uses the first ones (kept in LDS until the end)
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ __forceinline__ void stamp_put(unsigned long long *area, int slot, double dep) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory");
  if (slot < kStampLdsSlots && (threadIdx.x & 63) == 0) area[slot] = t;
}
#define MI_STAMP(slot, dep) stamp_put(stamp_area, (slot), (double)(dep))
#else
#define MI_STAMP(slot, dep) ((void)0)
#endif

// A load the compiler keeps where it is written: a relaxed atomic load at wavefront scope is a plain global_load
// (no cache-policy bits at that scope) but "ordered" -- neither the IR passes (which sink ordinary loads to their
// first use and hoist address-ready ones) nor the machine scheduler move memory operations across it.  The window
// kernels' queue discipline rests on it.
template <class T>
__device__ __forceinline__ T pinned_load(const T *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

typedef __attribute__((address_space(3))) double LdsDouble;

__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// 64 consecutive rows of a row-major field as the wave holds them after P coalesced 512-byte loads (lane l: the
// doubles l, 64 + l, ... of the 64 P): stored as rows RS doubles apart / one lane's row read back
template <int P, int RS>
__device__ __forceinline__ void image_store(LdsDouble *rows, int lane, const double (&buf)[P]) {
  if constexpr (RS == P) {
#pragma unroll
    for (int c = 0; c < P; ++c) rows[c * 64 + lane] = buf[c];
  } else {
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const unsigned e = (unsigned)lane + 64u * (unsigned)c;  // element of the chunk: row e / P, column e % P
      rows[(e / (unsigned)P) * (unsigned)RS + e % (unsigned)P] = buf[c];
    }
  }
}
template <int P, int RS>
__device__ __forceinline__ void image_row(const LdsDouble *rows, int lane, double (&out)[P]) {
  const LdsDouble *l = rows + (unsigned)lane * (unsigned)RS;
#pragma unroll
  for (int c = 0; c < P; ++c) out[c] = l[c];
}
// An epilogue that declares `using wants_scratch = void;` is handed the wave's far slots (2 x 64 rows, RS doubles apart) as
// scratch space: end(slice, acc, vrow, scratch) -- free once the slice's entries are done.
template <class E, class = void>
struct EpiScratch : std::false_type {};
template <class E>
struct EpiScratch<E, std::void_t<typename E::wants_scratch>> : std::true_type {};

#ifndef MI_WIN_FARC
#define MI_WIN_FARC 1   // pure far structure, one context: far rows as coalesced images (sell_window FARC) -- 1: the wide rows
                        // (p = 5 / 6 / 7: 46.7 / 57.3 / 78.0 -> 46.3 / 56.1 / 74.8 us), 2: also p <= 4 (no change there)
#endif
// the window-form matrix arrays (mi_csr::wk, wfar) and constants
struct WinView {
  const uint32_t *__restrict__ wk;
  const int32_t *__restrict__ wfar;
  int wc, nc;     // window half-width in chunks; ring chunks (2 kWinWaves + 2 wc)
  uint32_t zw;    // word of a non-entry: zero row, index of 0.0
  const int *__restrict__ bounds;  // first tile of every workgroup (+ end), or null: equal runs
  unsigned far_d;                  // FARD kernels: the far stride D (far columns = row + D, row - D; wfar unused)
  const uint32_t *__restrict__ wk16;  // W16 kernels: the 16-bit form of the words (mi_csr::wk16)
};

// Tiles [t0, t1) of kWinWaves slices for this workgroup; `lds_rows` = kWinLdsRows x P doubles (ring, zero row, far slots).
// FARD: the matrix's far structure is pure (mi_csr::win_far_pure): the far columns of a row are row + D and row - D
// (slot 0, slot 1) and are computed, not loaded (with HALO: possibly a halo column, see halo_lo / halo_hi)
// W16: the words come in the 16-bit form (mi_csr::wk16: eight entries per row in four dwords, indexed by slice; a
// non-entry is the zero word, so no slice bounds are consulted)
// halo_lo / halo_hi (HALO && FARD): rows of rank - 1 / rank + 1 in the halo buffer; a computed far column that leaves
// the local rows is the halo column of that row (mi_csr: column n + h, h < halo_lo: row h - halo_lo of the slab below,
// else row h - halo_lo of the slab above)
// RS (r06): doubles between consecutive rows of the ring -- P (the ring is the memory image of its chunks) or, for the
// wide rows, P padded to an odd number: with RS = P = 6 or 8 the row-by-row reads of 64 lanes fall on half / a sixteenth
// of the LDS banks (k_st_hess_widewin).  RS > P scatters a staged chunk's doubles to (row, column) addresses.
// FARC (r06): FARD without a halo -- the far rows of a slice are the 64 CONSECUTIVE rows at slice 64 +- D: loaded as P
// coalesced 512-byte loads per slot (a memory image, like a ring chunk) instead of P strided 8-byte loads per lane (at 6
// doubles a row a wave instruction of those touches 24 lines).  A row without that neighbour has no word for the slot, so
// whatever lies at the clamped address is never referenced.
template <int P, int HW, bool HALO, bool FARD, bool W16, class Epi, int RS = P, bool FARC = false>
__device__ __forceinline__ void sell_window(const SellView &A, const WinView &W, int t0, int t1, int w, int lane,
                                            const double *__restrict__ V, const double *vt, double *lds_rows,
                                            Epi &epi, unsigned halo_lo = 0, unsigned halo_hi = 0) {
  static_assert(HW <= kWinHead, "head width");
  static_assert(!FARC || (FARD && !HALO), "coalesced far rows: pure far structure, one context");
  constexpr int NW = kWinWaves;
  const int nchunks = (int)A.nslices;
#ifdef MI_WIN_DEBUG  // timing experiments only (wrong results): flags in the high bits of wc
  const int dbg = W.wc >> 8;
  const int wc = W.wc & 255, nc = W.nc;
#else
  constexpr int dbg = 0;
  const int wc = W.wc, nc = W.nc;
#endif
  const unsigned nloc = (unsigned)A.n;
  const unsigned nPbytes = (unsigned)(A.n * P * 8);
  const unsigned lane8 = (unsigned)lane * 8u, lane4 = (unsigned)lane * 4u;
  LdsDouble *const L = (LdsDouble *)lds_rows;
#ifdef MI_WIN_STAMPS
  __shared__ unsigned long long stamp_lds[NW * kStampLdsSlots];
  unsigned long long *stamp_area = stamp_lds + w * kStampLdsSlots;
  if (lane < kStampLdsSlots) stamp_area[lane] = 0;
#endif
  MI_STAMP(0, 0.0);
  if (t0 >= t1) return;
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
  auto chunk_store = [&](int slot, const double (&buf)[P]) {
    image_store<P, RS>(L + (unsigned)slot * (unsigned)(64 * RS), lane, buf);
  };
  // raw operands of a slice: its words (ONE scalar base + immediate offsets j * 256: the words behind a narrower
  // slice belong to the next slice or to the padding mi_csr keeps behind the array, and are replaced by zw when
  // consumed) and its two far columns
  constexpr int NWORD = W16 ? 4 : HW;
  struct Head {
    unsigned c[NWORD];
    unsigned f[kFarCap];
  };
  auto load_words = [&](Head &h, int kk, int sl) {
    const char *pb = W16 ? reinterpret_cast<const char *>(W.wk16) + (unsigned)sl * 1024u + lane4
                         : reinterpret_cast<const char *>(W.wk) + (unsigned)kk * 256u + lane4;
#pragma unroll
    for (int j = 0; j < NWORD; ++j) h.c[j] = pinned_load(reinterpret_cast<const unsigned *>(pb + j * 256));
  };
  // entry j of the slice as (LDS row << 8 | value index)
  auto entry_word = [&](const Head &h, int j, int kcur, int bcur) -> unsigned {
    if constexpr (W16) {
      const unsigned e = (h.c[j >> 1] >> (16 * (j & 1))) & 0xffffu;
      return ((e >> 5) << 8) | (e & 31u);
    } else {
      return (kcur + j < bcur) ? h.c[j] : W.zw;  // (wave-uniform condition)
    }
  };
  auto load_far = [&](unsigned (&f)[kFarCap], int sl) {
    if constexpr (FARD) {
      static_assert(!FARD || kFarCap == 2, "computed far columns: two slots");
      const unsigned r = (unsigned)sl * 64u + (unsigned)lane, rr = r < nloc ? r : nloc - 1u, D = W.far_d;
      // (a row without that neighbour has no word for the slot: any valid column will do -- its own)
      if constexpr (HALO) {
        const unsigned up = rr + D;  // local row, or row up - nloc of the slab above = halo column nloc + halo_lo + that
        f[0] = up < nloc ? up : (up - nloc < halo_hi ? up + halo_lo : rr);
        // row rr - D, or row -(D - rr) of the slab below = halo column nloc + halo_lo - (D - rr)
        f[1] = rr >= D ? rr - D : (D - rr <= halo_lo ? nloc + halo_lo - (D - rr) : rr);
      } else {
        f[0] = rr + D < nloc ? rr + D : rr;
        f[1] = rr >= D ? rr - D : rr;
      }
    } else {
      const char *fb = reinterpret_cast<const char *>(W.wfar) + (unsigned)sl * (unsigned)(kFarCap * 256) + lane4;
#pragma unroll
      for (int j = 0; j < kFarCap; ++j) f[j] = pinned_load(reinterpret_cast<const unsigned *>(fb + j * 256));
    }
  };
  auto load_head = [&](Head &h, int kk, int sl) {
    load_words(h, kk, sl);
    load_far(h.f, sl);
  };
  auto far_row = [&](unsigned c, double (&out)[P]) {
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

  // FARC: slot s of slice sl as a memory image (32-bit wrap-around below row 0 lands past the field or on a valid
  // address: clamped or unreferenced either way)
  auto far_chunk = [&](int sl, int s, double (&out)[P]) {
    const unsigned r0 = (unsigned)sl * 64u + (s == 0 ? W.far_d : 0u - W.far_d);
    const unsigned b0 = r0 * (unsigned)(P * 8) + lane8;
#pragma unroll
    for (int cc = 0; cc < P; ++cc) {
      const unsigned b = b0 + (unsigned)cc * 512u;
      const unsigned bs = b < nPbytes ? b : 0u;
      out[cc] = pinned_load(reinterpret_cast<const double *>(reinterpret_cast<const char *>(V) + bs));
    }
  };

  const int zrow = nc * 64;
  const unsigned far_rows0 = (unsigned)(zrow + 1 + w * (kFarCap * 64)) * (unsigned)RS;          // the wave's slot 0, row 0
  const unsigned far_base = (unsigned)(zrow + 1 + w * (kFarCap * 64) + lane) * (unsigned)RS;  // this lane's slot 0
  int slice = t0 * NW + w;               // (tiles are aligned: slice % NW == w)
  bool have = slice < nchunks;
  int k = 0, b1 = 0;
  if (have) {
    k = (int)slice_bound(A.slice_ptr, slice);
    b1 = (int)slice_bound(A.slice_ptr, slice + 1);
  }
  Head h;
#ifndef MI_WIN_NO_FARAHEAD
  // Far rows are gathered ONE TILE AHEAD: during a tile the registers gf hold this slice's far rows (gathered while
  // the previous tile ran), h.c its words and h.f the far columns of the NEXT slice.  The entry loop then waits for
  // no memory at all; the tile's only waits are for the staged chunk and for the X/Y rows at its end.
  unsigned f0[kFarCap];
  if constexpr (!FARC) load_far(f0, have ? slice : 0);
  load_words(h, k, have ? slice : 0);
#else
  load_head(h, k, have ? slice : 0);
#endif
  // ring: the first tile's window, the zero row
  int slot_own;  // ring slot of the wave's own chunk (= slice % nc), advanced by 16 per tile
  {
    const int lo = t0 * NW - wc;
    int hi = t0 * NW + NW + wc;
    if (hi > nchunks) hi = nchunks;
    // all of this wave's chunks requested before the first is stored: one memory round trip, not one per chunk
    constexpr int kFill = (NW + 2 * kMaxWinChunks + NW - 1) / NW;
    double b[kFill][P];
#pragma unroll
    for (int i = 0; i < kFill; ++i) {
      const int q = lo + w + i * NW;
      chunk_load(q >= 0 && q < hi ? q : 0, b[i]);  // (unconditional: chunk 0 when there is none)
    }
#pragma unroll
    for (int i = 0; i < kFill; ++i) {
      const int q = lo + w + i * NW;
      if (q >= 0 && q < hi) chunk_store(q % nc, b[i]);
    }
    if (threadIdx.x < P) L[zrow * RS + threadIdx.x] = 0.0;
    slot_own = (t0 * NW + w) % nc;
  }
  int slot_job = (t0 * NW + NW + wc + w) % nc;  // slot of the chunk this wave stages during the first tile
#ifndef MI_WIN_NO_FARAHEAD
  double gf[kFarCap][P];
  {
    const bool more0 = have && t0 + 1 < t1 && slice + NW < nchunks;
    if constexpr (FARC) {
#pragma unroll
      for (int s = 0; s < kFarCap; ++s) far_chunk(have ? slice : 0, s, gf[s]);
    } else {
      load_far(h.f, more0 ? slice + NW : (have ? slice : 0));
#pragma unroll
      for (int s = 0; s < kFarCap; ++s) far_row(f0[s], gf[s]);
    }
  }
#endif
  lds_barrier();
  MI_STAMP(1, 0.0);

  for (int t = t0; t < t1; ++t) {
    MI_STAMP(2 + 8 * (t - t0), 0.0);
    const int qj = (t + 1) * NW + wc + w;  // staged for tile t + 1 (its last wc chunks included)
    const bool job = t + 1 < t1 && qj < nchunks;
    double pre[P] = {};
    if (!(dbg & 16)) chunk_load(job ? qj : 0, pre);  // unconditional: the s_waitcnt counts below stay exact on every path
    if (have) {
      const int nslice = slice + NW;
      const bool more = t + 1 < t1 && nslice < nchunks;
      int q0 = k, q1 = b1;  // nothing follows: the prefetch re-reads this slice
      if (more) { q0 = (int)slice_bound(A.slice_ptr, nslice); q1 = (int)slice_bound(A.slice_ptr, nslice + 1); }
      // ---- this slice's operands become LDS contents and addresses; the registers take the next slice's -------
#ifndef MI_WIN_NO_FARAHEAD
      unsigned wd[HW];
#pragma unroll
      for (int j = 0; j < HW; ++j) wd[j] = entry_word(h, j, k, b1);
      MI_STAMP(2 + 8 * (t - t0) + 1, (double)wd[0]);  // this slice's matrix words were there
      double gn[kFarCap][P];  // the NEXT slice's far rows (h.f: its far columns, loaded a tile ago)
      if constexpr (FARC) {
#pragma unroll
        for (int s = 0; s < kFarCap; ++s) far_chunk(more ? nslice : slice, s, gn[s]);
      } else {
#pragma unroll
        for (int s = 0; s < kFarCap; ++s) far_row(h.f[s], gn[s]);
      }
      load_words(h, q0, more ? nslice : slice);
      if constexpr (!FARC) {
        const int nslice2 = nslice + NW;
        const bool more2 = more && t + 2 < t1 && nslice2 < nchunks;
        load_far(h.f, more2 ? nslice2 : (more ? nslice : slice));
      }
#else
      double gf[kFarCap][P];
#pragma unroll
      for (int s = 0; s < kFarCap; ++s) far_row((dbg & 1) ? (unsigned)(slice * 64 + lane) : h.f[s], gf[s]);
      unsigned wd[HW];
#pragma unroll
      for (int j = 0; j < HW; ++j) wd[j] = entry_word(h, j, k, b1);
      MI_STAMP(2 + 8 * (t - t0) + 1, (double)wd[0]);  // this slice's matrix words were there
      if (!(dbg & 8)) load_head(h, q0, more ? nslice : slice);
#endif
      if (!(dbg & 2)) epi.request(slice);  // the epilogue rows of THIS slice: the entries' work lies between here and their use
      // the gathers (and only they: 2 P + HW + 2 + 2 P pinned loads were issued behind them) -> far slots
      if constexpr (FARC) {
#pragma unroll
        for (int s = 0; s < kFarCap; ++s) image_store<P, RS>(L + far_rows0 + (unsigned)(s * 64 * RS), lane, gf[s]);
      } else {
#pragma unroll
        for (int s = 0; s < kFarCap; ++s)
#pragma unroll
          for (int c = 0; c < P; ++c) L[far_base + (unsigned)(s * 64 * RS + c)] = gf[s][c];
      }
      MI_STAMP(2 + 8 * (t - t0) + 2, gf[kFarCap - 1][P - 1]);
      // ---- entries in storage order ----------------------------------------------------------------------------
      double acc[P];
#pragma unroll
      for (int c = 0; c < P; ++c) acc[c] = 0;
#pragma unroll
      for (int j = 0; j < HW; ++j) {
        const double aj = vt[wd[j] & 255u];
        const LdsDouble *l = L + (wd[j] >> 8) * (unsigned)RS;
#pragma unroll
        for (int c = 0; c < P; ++c) {
#pragma clang fp contract(off)
          const double tt = aj * l[c];
          acc[c] = acc[c] + tt;
        }
      }
      MI_STAMP(2 + 8 * (t - t0) + 4, acc[0]);
      // the staged chunk goes into the ring here: its loads are the oldest in the queue
      if (job) chunk_store(slot_job, pre);
      {
        double vrow[P];
        const LdsDouble *l = L + (unsigned)(slot_own * 64 + lane) * (unsigned)RS;
#pragma unroll
        for (int c = 0; c < P; ++c) vrow[c] = l[c];
        if constexpr (EpiScratch<Epi>::value) {
          if (!(dbg & 4)) epi.end(slice, acc, vrow, L + far_rows0);
        } else {
          if (!(dbg & 4)) epi.end(slice, acc, vrow);
        }
        MI_STAMP(2 + 8 * (t - t0) + 5, acc[0]);  // epilogue done, stores issued
      }
#ifndef MI_WIN_NO_FARAHEAD
#pragma unroll
      for (int s = 0; s < kFarCap; ++s)
#pragma unroll
        for (int c = 0; c < P; ++c) gf[s][c] = gn[s][c];
#endif
      have = more;
      slice = nslice;
      k = q0;
      b1 = q1;
    } else if (job) {
      chunk_store(slot_job, pre);  // a wave without a slice in this tile
    }
    slot_own += NW;
    slot_own = slot_own >= nc ? slot_own - nc : slot_own;
    slot_job += NW;
    slot_job = slot_job >= nc ? slot_job - nc : slot_job;
    if (t + 1 < t1) {
      MI_STAMP(2 + 8 * (t - t0) + 6, 0.0);
      lds_barrier();
      MI_STAMP(2 + 8 * (t - t0) + 7, 0.0);  // barrier passed
    }
  }
#ifdef MI_WIN_STAMPS
  if (g_stamp_buf && lane < kStampLdsSlots)
    g_stamp_buf[((size_t)blockIdx.x * NW + w) * kStampSlots + lane] = stamp_area[lane];
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
