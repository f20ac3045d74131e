// rr_host.cpp -- the small dense Rayleigh-Ritz problem of LOBPCG (reference LinearAlgebra/LOBPCG.h:53-62) as a
// host-only translation unit.  The solver is the header-only one the generic path of the template uses
// (Optimization/LinearAlgebra/DenseSymmetricEigen.h); it is compiled HERE by g++ with AVX2 and with contraction off
// (-mavx2 -ffp-contract=off: wider vectors for the unit-stride loops, no fused multiply-adds, no reassociation), which
// keeps every result bit-identical to the scalar build and is 1.5-2x faster than the same source through hipcc's
// host pass: the solve sits between two Gram read-backs and the panel update of every LOBPCG iteration, nothing on
// the device can overlap it.
#include "Optimization/LinearAlgebra/DenseSymmetricEigen.h"

namespace mi {
int rr_host(int n, const double *A, const double *B, double *Theta, double *C) {
  return Optimization::LinearAlgebra::dense::generalized_symmetric_eig(n, A, B, Theta, C);
}
}  // namespace mi
