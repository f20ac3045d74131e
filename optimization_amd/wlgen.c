/* wlgen.c -- machine-independent synthetic inputs for the BASELINE.json configurations (harness side; not part of the
 * C ABI and not on the product path: optimization_amd/workloads.py calls it through ctypes).
 *
 * SURVEY.md 8(d) specifies the cfg2 start as "X0 = QR of mt19937_64(seed=20260928) U(-1,1) n x 3 (host-generated)"
 * so that the same bits come out on every host.  r01-r05 generated these arrays with numpy (PCG64 + LAPACK QR, libm
 * sin), whose last bits follow the CPU model; a full-size SHA-256 fixture then only held on the machine it was made
 * on.  Everything here is integer arithmetic or IEEE-754 +,-,*,/,sqrt in a fixed sequential order (compile with
 * -ffp-contract=off, no -ffast-math): the same bytes on any x86-64 host.
 *
 *   - mt19937_64: Matsumoto & Nishimura's 64-bit Mersenne twister (the generator std::mt19937_64 names), seeded by
 *     init_genrand64(seed).  U(-1,1) := 2 * ((x >> 11) * 2^-53) - 1, exact in double.
 *   - sin(pi * num / den): integer reduction to the first quadrant, then a fixed Taylor polynomial in Horner form.
 *     Not correctly rounded, but deterministic (no libm).
 *   - thin QR: modified Gram-Schmidt applied twice, column sums accumulated in `long double` in row order.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define NN 312
#define MM 156
#define MATRIX_A 0xB5026F5AA96619E9ULL
#define UM 0xFFFFFFFF80000000ULL
#define LM 0x7FFFFFFFULL

typedef struct { uint64_t mt[NN]; int mti; } mt64;

static void mt_seed(mt64 *g, uint64_t seed) {
  g->mt[0] = seed;
  for (g->mti = 1; g->mti < NN; g->mti++)
    g->mt[g->mti] = 6364136223846793005ULL * (g->mt[g->mti - 1] ^ (g->mt[g->mti - 1] >> 62)) + (uint64_t)g->mti;
}

static uint64_t mt_next(mt64 *g) {
  static const uint64_t mag01[2] = {0ULL, MATRIX_A};
  uint64_t x;
  if (g->mti >= NN) {
    int i;
    for (i = 0; i < NN - MM; i++) {
      x = (g->mt[i] & UM) | (g->mt[i + 1] & LM);
      g->mt[i] = g->mt[i + MM] ^ (x >> 1) ^ mag01[(int)(x & 1ULL)];
    }
    for (; i < NN - 1; i++) {
      x = (g->mt[i] & UM) | (g->mt[i + 1] & LM);
      g->mt[i] = g->mt[i + (MM - NN)] ^ (x >> 1) ^ mag01[(int)(x & 1ULL)];
    }
    x = (g->mt[NN - 1] & UM) | (g->mt[0] & LM);
    g->mt[NN - 1] = g->mt[MM - 1] ^ (x >> 1) ^ mag01[(int)(x & 1ULL)];
    g->mti = 0;
  }
  x = g->mt[g->mti++];
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}

/* the raw 64-bit outputs (known-answer test against the published first outputs of the generator) */
void wl_mt19937_64_raw(uint64_t seed, size_t count, uint64_t *out) {
  mt64 g;
  mt_seed(&g, seed);
  for (size_t i = 0; i < count; i++) out[i] = mt_next(&g);
}

void wl_uniform_pm1(uint64_t seed, size_t count, double *out) {
  mt64 g;
  mt_seed(&g, seed);
  for (size_t i = 0; i < count; i++) out[i] = 2.0 * ((double)(mt_next(&g) >> 11) * 0x1.0p-53) - 1.0;
}

/* sin(pi * num / den), num >= 0, den > 0 */
double wl_sin_pi_ratio(int64_t num, int64_t den) {
  /* pi in two pieces so that pi * t keeps ~1e-17 of relative accuracy; t = r / den with 0 <= r <= den / 2 */
  static const double PI_HI = 3.141592653589793116, PI_LO = 1.2246467991473532e-16;
  int64_t r = num % (2 * den);
  double sign = 1.0;
  if (r >= den) { r -= den; sign = -1.0; }
  if (2 * r > den) r = den - r;              /* sin(pi - x) = sin x */
  double t = (double)r / (double)den;       /* in [0, 1/2] */
  double x = PI_HI * t + PI_LO * t;
  double x2 = x * x;
  /* Taylor to x^29 (|x| <= pi/2: the first dropped term is below 2e-26) */
  static const double c[] = {
      -1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0, 1.0 / 6227020800.0,
      -1.0 / 1307674368000.0, 1.0 / 355687428096000.0, -1.0 / 121645100408832000.0,
      1.0 / 51090942171709440000.0, -1.0 / 25852016738884976640000.0, 1.0 / 15511210043330985984000000.0,
      -1.0 / 10888869450418352160768000000.0, 1.0 / 8841761993739701954543616000000.0};
  double s = c[13];
  for (int i = 12; i >= 0; i--) s = s * x2 + c[i];
  return sign * (x + x * x2 * s);
}

/* unit-norm eigenvector of the 7-point Dirichlet Laplacian on nx x ny x nz (x fastest) for mode (kx, ky, kz) >= 1:
 * the tensor product of sqrt(2 / (n + 1)) sin(pi k j / (n + 1)), j = 1..n (sum of squares of each factor: exactly 1
 * in exact arithmetic) */
int wl_grid_mode(int64_t nx, int64_t ny, int64_t nz, int64_t kx, int64_t ky, int64_t kz, double *out, size_t stride) {
  double *sx = (double *)malloc(sizeof(double) * (size_t)(nx + ny + nz));
  if (!sx) return 1;
  double *sy = sx + nx, *sz = sy + ny;
  for (int64_t j = 0; j < nx; j++) sx[j] = sqrt(2.0 / (double)(nx + 1)) * wl_sin_pi_ratio(kx * (j + 1), nx + 1);
  for (int64_t j = 0; j < ny; j++) sy[j] = sqrt(2.0 / (double)(ny + 1)) * wl_sin_pi_ratio(ky * (j + 1), ny + 1);
  for (int64_t j = 0; j < nz; j++) sz[j] = sqrt(2.0 / (double)(nz + 1)) * wl_sin_pi_ratio(kz * (j + 1), nz + 1);
  size_t i = 0;
  for (int64_t c = 0; c < nz; c++)
    for (int64_t b = 0; b < ny; b++) {
      double zy = sz[c] * sy[b];
      for (int64_t a = 0; a < nx; a++, i++) out[i * stride] = zy * sx[a];
    }
  free(sx);
  return 0;
}

/* in place: the n x p row-major matrix M becomes the Q factor of its thin QR with a positive diagonal of R (modified
 * Gram-Schmidt, every projection applied twice).  Returns 1 on a (numerically) dependent column. */
int wl_thin_qr(size_t n, size_t p, double *M) {
  for (size_t j = 0; j < p; j++) {
    for (int pass = 0; pass < 2; pass++)
      for (size_t k = 0; k < j; k++) {
        long double d = 0.0L;
        for (size_t i = 0; i < n; i++) d += (long double)M[i * p + k] * (long double)M[i * p + j];
        double dd = (double)d;
        for (size_t i = 0; i < n; i++) M[i * p + j] -= dd * M[i * p + k];
      }
    long double q = 0.0L;
    for (size_t i = 0; i < n; i++) q += (long double)M[i * p + j] * (long double)M[i * p + j];
    double nrm = sqrt((double)q);
    if (!(nrm > 0.0)) return 1;
    for (size_t i = 0; i < n; i++) M[i * p + j] /= nrm;
    /* a second normalisation takes out the rounding of the first quotient pass */
    q = 0.0L;
    for (size_t i = 0; i < n; i++) q += (long double)M[i * p + j] * (long double)M[i * p + j];
    nrm = sqrt((double)q);
    for (size_t i = 0; i < n; i++) M[i * p + j] /= nrm;
  }
  return 0;
}

/* M += scale * U(-1,1) from mt19937_64(seed), in storage order */
void wl_add_uniform_pm1(uint64_t seed, size_t count, double scale, double *M) {
  mt64 g;
  mt_seed(&g, seed);
  for (size_t i = 0; i < count; i++) M[i] += scale * (2.0 * ((double)(mt_next(&g) >> 11) * 0x1.0p-53) - 1.0);
}
