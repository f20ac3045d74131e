// ref_driver.cpp -- builds oracle/_ref/libref.so: the REAL reference templates
// (Optimization::LinearAlgebra::{STPCG, LSQR}, Optimization::Riemannian::{TNT, GradientDescent, TNLS}),
// compiled from the headers where they lie under /root/reference/include (never copied into this
// repo), instantiated on a plain host vector type and exposed as ref_* C entry points.
//
// TEST INFRASTRUCTURE ONLY.  Built only where /root/reference exists (this container); the built
// .so is git-ignored but travels to the GPU box, where it is used as a checker and as the
// "reference" CPU baseline.  The reference's TNT.h / IterativeSolvers.h / GradientDescent.h / TNLS.h
// include no third-party header and are generic over the vector type, so
// instantiating them on a host vector is ordinary use of their template parameter, not a stand-in
// for a missing dependency.  LOBPCG.h hard-requires Eigen (absent) and is NOT built here.
//
// The driver body is shared with tests/cpp/harness_host.cpp, which compiles the SAME code against
// the MI355X build's own headers (hz_* entry points) -- see template_driver.inc.
#include "Optimization/LinearAlgebra/IterativeSolvers.h"
#include "Optimization/Riemannian/GradientDescent.h"
#include "Optimization/Riemannian/TNLS.h"
#include "Optimization/Riemannian/TNT.h"

#define DRV(name) ref_##name
#include "template_driver.inc"
