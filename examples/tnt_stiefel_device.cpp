// examples/tnt_stiefel_device.cpp -- the reference's client pattern (examples/Riemannian_optimization_example.cpp:
// build callables, fill TNTParams, call Optimization::Riemannian::TNT, read the TNTResult) with
// Vector = Optimization::MI355::DeviceVector: the p lowest eigenvectors of a sparse SPD matrix by minimising the
// Rayleigh quotient f(X) = 1/2 tr(X'AX) over the Stiefel manifold St(n,p).
//
//   (built by optimization_amd/build.py: build_harness() into examples/bin/; by hand:)
//   g++ -std=c++17 -O2 -I optimization_amd/include -I include examples/tnt_stiefel_device.cpp \
//       -L optimization_amd -lmi355opt -Wl,-rpath,$PWD/optimization_amd -o tnt_stiefel_device
//   ./tnt_stiefel_device [grid]        (grid^3 unknowns per column, default 60)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <optional>
#include <random>
#include <vector>

#include "Optimization/MI355/Stiefel.h"
#include "Optimization/Riemannian/TNT.h"

using namespace Optimization;
using MI355::DeviceVector;

int main(int argc, char **argv) {
  const int g = argc > 1 ? std::atoi(argv[1]) : 60, p = 3;
  const size_t n = (size_t)g * g * g;
  // 7-point Laplacian + 0.1 I in CSR
  std::vector<int32_t> rowptr(n + 1, 0), col;
  std::vector<double> val;
  for (int z = 0; z < g; ++z)
    for (int y = 0; y < g; ++y)
      for (int x = 0; x < g; ++x) {
        const size_t i = ((size_t)z * g + y) * g + x;
        auto put = [&](size_t j, double v) {
          col.push_back((int32_t)j);
          val.push_back(v);
        };
        if (z > 0) put(i - (size_t)g * g, -1);
        if (y > 0) put(i - g, -1);
        if (x > 0) put(i - 1, -1);
        put(i, 6.1);
        if (x + 1 < g) put(i + 1, -1);
        if (y + 1 < g) put(i + g, -1);
        if (z + 1 < g) put(i + (size_t)g * g, -1);
        rowptr[i + 1] = (int32_t)col.size();
      }
  // a random point of St(n,p): Gram-Schmidt of a Gaussian block (row-major n x p)
  std::mt19937_64 rng(7);
  std::normal_distribution<double> N01;
  std::vector<double> X0(n * p);
  for (double &v : X0) v = N01(rng);
  for (int a = 0; a < p; ++a) {
    for (int b = 0; b < a; ++b) {
      double d = 0;
      for (size_t i = 0; i < n; ++i) d += X0[i * p + a] * X0[i * p + b];
      for (size_t i = 0; i < n; ++i) X0[i * p + a] -= d * X0[i * p + b];
    }
    double s = 0;
    for (size_t i = 0; i < n; ++i) s += X0[i * p + a] * X0[i * p + a];
    s = std::sqrt(s);
    for (size_t i = 0; i < n; ++i) X0[i * p + a] /= s;
  }

  try {
    MI355::Context ctx(0);
    MI355::StiefelRayleighQuotient prob(ctx, n, p, rowptr.data(), col.data(), val.data());
    DeviceVector x0(ctx, X0.data(), n * p);
    Riemannian::TNTParams<double> params;
    params.max_iterations = 200;
    params.max_TPCG_iterations = 50;
    params.gradient_tolerance = 1e-7;
    params.relative_decrease_tolerance = 1e-14;
    params.stepsize_tolerance = 1e-12;
    auto result = Riemannian::TNT<DeviceVector, DeviceVector>(
        prob.objective(), prob.quadratic_model(), prob.metric(), prob.retraction(), x0,
        std::optional<Riemannian::LinearOperator<DeviceVector, DeviceVector>>(), params);
    size_t inner = 0;
    for (size_t k : result.inner_iterations) inner += k;
    // sum of the p smallest eigenvalues of the grid operator, for comparison
    double exact = 0;
    const int modes[3][3] = {{1, 1, 1}, {1, 1, 2}, {1, 2, 1}};
    for (int a = 0; a < p; ++a) {
      double lam = 0.1;
      for (int d = 0; d < 3; ++d) lam += 4 * std::pow(std::sin(M_PI * modes[a][d] / (2.0 * (g + 1))), 2);
      exact += lam;
    }
    std::printf("St(%zu,%d): status %d after %zu outer / %zu inner iterations, %.3f s\n", n, p, (int)result.status,
                result.inner_iterations.size(), inner, result.elapsed_time);
    std::printf("2 f(X) = %.12f   sum of the %d smallest eigenvalues = %.12f   |grad| = %.2e\n", 2 * result.f, p,
                exact, result.gradfx_norm);
    return std::fabs(2 * result.f - exact) < 1e-6 * exact ? 0 : 1;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
