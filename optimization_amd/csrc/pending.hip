// pending.hip -- entry points declared in include/mi355opt.h whose kernels have not landed yet.
// They fail loudly (never silently fall back to the CPU).
#include "mi_internal.h"

using namespace mi;

#define MI_PENDING(name)                                   \
  do {                                                     \
    set_error(name ": kernels not implemented yet");       \
    return MI_ERR_INTERNAL;                                \
  } while (0)

extern "C" {

int mi_lobpcg_gram(mi_ctx *, size_t, int, int, const mi_vec *, const mi_vec *, double *) { MI_PENDING("mi_lobpcg_gram"); }
int mi_lobpcg_update(mi_ctx *, size_t, int, int, const mi_vec *, const double *, int, mi_vec *) { MI_PENDING("mi_lobpcg_update"); }
int mi_lobpcg_residual(mi_ctx *, size_t, int, const mi_vec *, const mi_vec *, const mi_vec *, const double *,
                       mi_vec *, double *, double *) { MI_PENDING("mi_lobpcg_residual"); }
int mi_rayleigh_ritz(int, const double *, const double *, double *, double *) { MI_PENDING("mi_rayleigh_ritz"); }

}  // extern "C"
