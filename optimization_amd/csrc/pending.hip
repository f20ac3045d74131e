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

int mi_so3n_create(mi_ctx *, size_t, size_t, const int32_t *, const int32_t *, const double *,
                   const double *, mi_so3n **) { MI_PENDING("mi_so3n_create"); }
int mi_so3n_destroy(mi_so3n *) { return MI_OK; }
int mi_so3n_objective(mi_so3n *, const mi_vec *, double *) { MI_PENDING("mi_so3n_objective"); }
int mi_so3n_model(mi_so3n *, const mi_vec *, mi_vec *, mi_op **, mi_precon **) { MI_PENDING("mi_so3n_model"); }
int mi_so3n_retract(mi_so3n *, const mi_vec *, const mi_vec *, mi_vec *) { MI_PENDING("mi_so3n_retract"); }

int mi_lobpcg_gram(mi_ctx *, size_t, int, int, const mi_vec *, const mi_vec *, double *) { MI_PENDING("mi_lobpcg_gram"); }
int mi_lobpcg_update(mi_ctx *, size_t, int, int, const mi_vec *, const double *, int, mi_vec *) { MI_PENDING("mi_lobpcg_update"); }
int mi_lobpcg_residual(mi_ctx *, size_t, int, const mi_vec *, const mi_vec *, const mi_vec *, const double *,
                       mi_vec *, double *, double *) { MI_PENDING("mi_lobpcg_residual"); }
int mi_rayleigh_ritz(int, const double *, const double *, double *, double *) { MI_PENDING("mi_rayleigh_ritz"); }

}  // extern "C"
