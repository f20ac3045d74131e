"""optimization_amd -- MI355X-native Riemannian TNT / Steihaug-Toint CG / LOBPCG hot path.

Layout:
  csrc/      hand-written HIP kernels (gfx950) behind the C ABI of include/mi355opt.h
  include/   C++17 host layer mirroring the reference's function templates
             (Optimization::Riemannian::TNT, Optimization::LinearAlgebra::STPCG, ...)
  capi.py    ctypes binding of the C ABI (harness plumbing for tests/ and bench.py)
  build.py   hipcc build of libmi355opt.so (in-tree)
  workloads.py  seeded synthetic inputs of the BASELINE.json configurations
"""
__all__ = ["capi", "build", "workloads"]
