// stiefel_core.h -- packed symmetric P x P helpers shared by the Stiefel kernels (stiefel.hip) and the
// direction-Gram variant of STPCG's direction kernel (stpcg.hip).
#pragma once
#include "mi_internal.h"

namespace mi {

template <int P>
struct SymIdx {
  static constexpr int NS = P * (P + 1) / 2;
  __host__ __device__ static constexpr int at(int a, int b) {  // a <= b
    return a * P - a * (a - 1) / 2 + (b - a);
  }
};

// components the one-pass Hessian leaves per workgroup when the Gram of its OUTPUT rides along (recurrence
// form of mi_op::dirgram): 3 curvature dots + NS packed symmetric entries, padded to a supported row count
template <int P>
struct DirComps {
  static constexpr int NS = SymIdx<P>::NS;
  // (p <= 4: the widths the exchange kernels are instantiated for; p = 5 ... 8: 18, 24, 31, 39 -- exactly 3 + NS)
  static constexpr int value = (3 + NS <= 4) ? 4 : (3 + NS <= 6) ? 6 : (3 + NS <= 9) ? 9 : (3 + NS <= 16) ? 16 : 3 + NS;
};
inline int dir_comps(int p) { return p == 1 ? 4 : p == 2 ? 6 : p == 3 ? 9 : p == 4 ? 16 : 3 + p * (p + 1) / 2; }

// Workgroup shape of the row-wise Stiefel kernels: rows of up to 4 doubles keep their p x p matrices in registers and run
// 1024-thread workgroups (128 registers per wave); wider rows (r05) run 256-thread workgroups -- one wave per SIMD, the
// register file to itself -- with the uniform p x p matrices in LDS.  (As 1024-thread instantiations the p = 8 kernels
// spilled 130 ... 1500 bytes per lane: the retraction took 1.7 ms at n = 1e6, the model assembly 450 us.)
template <int P>
struct StBlk {
  static constexpr int threads = P > 4 ? 256 : kBlock;
  static constexpr int waves = threads / 64;
};

// per-thread raw Gram accumulators -> this workgroup's partial row of the SYMMETRISED Gram
template <int P>
__device__ __forceinline__ void store_sym_partials(const double (&G)[P * P], double *lds,
                                                   double *__restrict__ partials) {
  constexpr int NS = SymIdx<P>::NS;
  double Gs[NS];
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int b = a; b < P; ++b)
      Gs[SymIdx<P>::at(a, b)] = (a == b) ? G[a * P + a] : .5 * (G[a * P + b] + G[b * P + a]);
  block_partials_store_w<NS, StBlk<P>::waves>(Gs, lds, partials);
}

// every thread: full symmetric P x P matrix M from the reduced rows (or all-reduced slots)
template <int P, bool FROM_SLOTS>
__device__ __forceinline__ void load_sym(const double *__restrict__ partials, int count,
                                         const double *__restrict__ slots, double (&M)[P * P],
                                         double *lds) {
  constexpr int NS = SymIdx<P>::NS;
  double s[NS];
  if (FROM_SLOTS) {
#pragma unroll
    for (int i = 0; i < NS; ++i) s[i] = slots[i];
  } else {
    reduce_rows_nw<NS, StBlk<P>::waves>(partials, count, s, lds);
  }
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int b = a; b < P; ++b) {
      M[a * P + b] = s[SymIdx<P>::at(a, b)];
      M[b * P + a] = s[SymIdx<P>::at(a, b)];
    }
}

}  // namespace mi
