// rr_host.cpp -- the small dense Rayleigh-Ritz problem of LOBPCG (reference LinearAlgebra/LOBPCG.h:53-62) as a
// host-only translation unit.  The solver is the header-only one the generic path of the template uses
// (Optimization/LinearAlgebra/DenseSymmetricEigen.h); it is compiled HERE by g++ with contraction off
// (-ffp-contract=off: no fused multiply-adds, no reassociation) and as TWO clones of the flattened solver -- an AVX2
// one (wider vectors for the unit-stride loops) and a baseline one, picked at load time by the CPU's features, so the
// library never executes an instruction the host lacks.  Both keep every result bit-identical to the scalar build; the
// AVX2 clone is 1.5-2x faster than the same source through hipcc's host pass: the solve sits between two Gram read-backs and the panel update of every LOBPCG iteration, nothing on
// the device can overlap it.
#include "Optimization/LinearAlgebra/DenseSymmetricEigen.h"

namespace mi {
__attribute__((target_clones("avx2", "default"), flatten))
int rr_host(int n, const double *A, const double *B, double *Theta, double *C) {
  return Optimization::LinearAlgebra::dense::generalized_symmetric_eig(n, A, B, Theta, C);
}
// the nx lowest Ritz pairs only (what an LOBPCG iteration reads; DenseSymmetricEigen.h generalized_symmetric_eig_lowest)
__attribute__((target_clones("avx2", "default"), flatten))
int rr_host_lowest(int n, int k, const double *A, const double *B, double *Theta, double *C) {
  return Optimization::LinearAlgebra::dense::generalized_symmetric_eig_lowest(n, k, A, B, Theta, C);
}
}  // namespace mi
