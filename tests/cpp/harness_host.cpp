// harness_host.cpp -- the MI355X build's template layer (optimization_amd/include/Optimization/...)
// instantiated on a plain HOST vector through exactly the same driver code as the real reference
// (oracle/template_driver.inc, compiled into oracle/_ref/libref.so with the reference's headers).
// pytest compares hz_* with ref_* bit for bit: same problems, same call sequence, only the template
// implementations differ.  Compiled WITHOUT the C-ABI header on the include path, i.e. it also
// proves the template layer is self-contained, generic C++17 (no device dependency).
#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/GradientDescent.h"
#include "Optimization/Riemannian/TNLS.h"
#include "Optimization/Riemannian/TNT.h"

#define DRV(name) hz_##name
#include "template_driver.inc"
