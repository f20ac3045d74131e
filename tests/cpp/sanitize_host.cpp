// sanitize_host.cpp -- the drop-in template layer under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md 5:
// the reference ships no sanitizer target; this is the host-hardening check of the MI355X build's headers).
// The same translation unit as the host harness (harness_host.cpp: TNT / STPCG / GradientDescent / LSQR / TNLS / LOBPCG
// instantiated on a plain host vector) plus a main() that drives every template once on small problems; built by
// optimization_amd/build.py build_sanitize() with g++ -fsanitize=address,undefined -fno-sanitize-recover=all and run
// by __graft_entry__.build() and tests/test_cpu_host_logic.py.  Any report aborts with a non-zero exit code.
#include "harness_host.cpp"

#include <cstdio>

extern "C" {
#include "oracle.h"
}

static int fail(const char *what) {
  std::fprintf(stderr, "sanitize_host: %s failed\n", what);
  return 1;
}

int main() {
  // --- TNT (with and without preconditioner) and GradientDescent on the oracle's problems
  for (int precon = 0; precon < 2; ++precon) {
    orc_problem *pr = orc_problem_rosenbrock(30, precon);
    orc_tnt_params prm;
    orc_tnt_default_params(&prm);
    prm.max_iterations = 1000;
    prm.gradient_tolerance = 1e-8;
    prm.relative_decrease_tolerance = prm.stepsize_tolerance = 0;   // (cfg1's parameters)
    std::vector<double> x0(30, 0.1), x(30);
    const size_t cap = prm.max_iterations + 2;
    std::vector<double> a(cap), b(cap), c(cap), d(cap), e(cap), f(cap), g(cap);
    std::vector<size_t> inner(cap);
    orc_tnt_result res{};
    res.x = x.data();
    res.objective_values = a.data(); res.gradient_norms = b.data(); res.preconditioned_gradient_norms = c.data();
    res.trust_region_radius = d.data(); res.inner_iterations = inner.data(); res.update_step_norms = e.data();
    res.update_step_M_norms = f.data(); res.gain_ratios = g.data();
    if (hz_tnt(pr, x0.data(), &prm, &res) != 0 || (res.status != ORC_TNT_GRADIENT && res.status != ORC_TNT_PRECONDITIONED_GRADIENT)) {
      std::fprintf(stderr, "status %d f %g outer %zu\n", res.status, res.f, res.outer_iterations);
      return fail("TNT Rosenbrock");
    }
    orc_problem_free(pr);
  }
  {
    const double P[3] = {0.0, 0.0, 1.0};
    orc_problem *pr = orc_problem_sphere(P, 0);
    const double x0[3] = {0.6, 0.0, 0.8};
    double x[3], fv = 0, gn = 0;
    int st = -1;
    size_t it = 0;
    std::vector<double> ov(4096);
    std::vector<size_t> ls(4096);
    if (hz_gd(pr, x0, 1000, 1e-6, 0.0, 0.0, 1.0, 0.5, 0.5, 100, x, &fv, &gn, &st, &it, 4096, ov.data(), ls.data()) != 0)
      return fail("GradientDescent sphere");
    orc_problem_free(pr);
  }
  // --- LSQR / TNLS
  {
    const size_t m = 12, n = 5;
    std::vector<double> A(m * n), bb(m), x(n);
    for (size_t i = 0; i < m; ++i) {
      bb[i] = 1.0 + 0.1 * (double)i;
      for (size_t j = 0; j < n; ++j) A[i * n + j] = (i == j ? 2.0 : 0.0) + 0.05 * (double)((i * 7 + j * 3) % 11);
    }
    double xn = 0;
    size_t it = 0;
    if (hz_lsqr_dense(m, n, A.data(), bb.data(), 100, 0.0, 1e-10, 1e-10, 1e8, 1e9, x.data(), &xn, &it) != 0)
      return fail("LSQR");
    std::vector<double> t(40), y(40);
    for (size_t i = 0; i < 40; ++i) { t[i] = 0.1 * (double)i; y[i] = std::sin(1.3 * t[i] + 0.4); }
    const double beta0[2] = {1.2, 0.3};
    double beta[2], fv = 0, gn = 0;
    int st = -1;
    size_t outer = 0, inner = 0;
    for (int precon = 0; precon < 2; ++precon)
      if (hz_tnls_sinfit(40, t.data(), y.data(), beta0, precon, 1e-9, 0.0, 1e-12, 50, beta, &fv, &gn, &st, &outer, &inner))
        return fail("TNLS");
  }
  // --- LOBPCG (generic dense path) and the Rayleigh-Ritz solver
  {
    const size_t m = 60, nx = 6, nev = 3;
    std::vector<double> Ad(m), X0(m * nx), Th(nev), X(m * nev);
    for (size_t i = 0; i < m; ++i) Ad[i] = 1.0 + (double)i;
    unsigned long long lcg = 12345;
    for (size_t k = 0; k < m * nx; ++k) {
      lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
      X0[k] = (double)(lcg >> 11) / 9007199254740992.0 - 0.5;
    }
    size_t it = 0, nc = 0;
    if (hz_lobpcg_dense(m, nx, nev, Ad.data(), nullptr, nullptr, nullptr, nullptr, nullptr, X0.data(), 500, 1e-8, Th.data(),
                        X.data(), &it, &nc, nullptr, nullptr, 0) != 0 || nc != nev || std::fabs(Th[0] - 1.0) > 1e-6) {
      std::fprintf(stderr, "iterations %zu nc %zu theta0 %.12g\n", it, nc, Th[0]);
      return fail("LOBPCG");
    }
  }
  // --- the Rayleigh-Ritz solvers (full, and the k lowest pairs) on a small SPD pencil
  {
    const int n = 40, k = 13;
    std::vector<double> A((size_t)n * n), B((size_t)n * n), th(n), C((size_t)n * n), tl(k), Cl((size_t)n * k);
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        A[i + (size_t)j * n] = (i == j ? 3.0 + 0.1 * i : 0.0) + 0.01 * std::cos(0.3 * (i + j));
        B[i + (size_t)j * n] = (i == j ? 2.0 : 0.0) + 0.01 * std::sin(0.2 * (i + j) + 1.0) * (i == j ? 0 : 1);
      }
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < j; ++i) B[i + (size_t)j * n] = B[j + (size_t)i * n];
    namespace dn = Optimization::LinearAlgebra::dense;
    if (dn::generalized_symmetric_eig(n, A.data(), B.data(), th.data(), C.data()) != 0) return fail("RayleighRitz");
    if (dn::generalized_symmetric_eig_lowest(n, k, A.data(), B.data(), tl.data(), Cl.data()) != 0 || tl[0] != th[0])
      return fail("RayleighRitz (lowest pairs)");
  }
  std::printf("sanitize_host: ok\n");
  return 0;
}
