// so3.hip -- chordal rotation averaging on SO(3)^N (BASELINE cfg3):
//     f(R) = 1/2 sum_e w_e | R_j - R_i Rt_e |_F^2 ,  e = (i -> j)
// as device callables for TNT (Objective, QuadraticModel, Retraction; metric = coordinate dot).
// The reference has no manifold code beyond S^2 (tests/TNT_unit_test.cpp:73-117); problem definition
// and formulas are shared with the CPU oracle (oracle/problems.c, "SO(3)^N"):
//     tangent at R_i: R_i hat(xi_i), xi in R^3;  grad_i = vee(Q_i - Q_i'),  Q_i = R_i' (L R)_i
//     Hess[xi]_i = vee(T_i - T_i'),  T_i = R_i' (L V)_i - hat(xi_i) sym(Q_i),  V_i = R_i hat(xi_i)
// with L the connection Laplacian.  Hess is LINEAR in xi with 3x3 blocks, so the quadratic model is
// assembled ONCE per outer iteration as a symmetric 3x3-block sparse matrix
//     h_i = D_i xi_i - sum_{inc (i,j)} w B_ij xi_j ,   D_i = 2 degw_i I - (tr(C_i) I - C_i),  C_i = sym(Q_i)
// and every STPCG pass is ONE kernel: block-ELL SpMV fused with the three curvature inner products
// <xi,h>, <h,h>, <xi,xi> (IterativeSolvers.h:300,305-306).  D_i^-1 is the 3x3 block-Jacobi
// preconditioner (fused into k_cg_update, PRE_BLOCK3).
//
// Layout: incidences in SELL-64-sigma over nodes (sigma = 1024: nodes degree-sorted inside each
// workgroup window, perm[slice*64+lane] = node); block component c of entry (slice s, k, lane) at
// ((slice_ptr[s] + k) * 9 + c) * 64 + lane  => each of the 9 component streams is coalesced; the
// diagonal blocks D_i are kept in the same slice order (Dsl).
// Algorithmic bytes per HVP: 72 (nnzb + N) + 4 nnzb + 8 (3N read + 3N written)  [nnzb = incidences].
#include <algorithm>

#include "mi_internal.h"

using namespace mi;

namespace {

constexpr bool kBsr3NtDefault = true;  // 123.5 -> 112.3 us per block-Jacobi step at N = 5e5 (r03)

struct IncView {
  size_t N, nslices;
  const long long *__restrict__ slice_ptr;
  const int *__restrict__ perm;   // node owned by (slice, lane), -1 for the padding lanes of the last window
  const int *__restrict__ nbr;    // neighbour node j (padding: own node)
  // (padding entries carry weight 0 in winc, which is what the kernels test: the edge ids and the head/tail flags the
  // packing uses stay on the host, 4 + 1 bytes per incidence entry nobody streams)
};

__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_mul_at(const double *A, const double *B, double *C) {  // A' B
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
__device__ __forceinline__ void vee_skew2(const double *T, double *o) {
  o[0] = T[7] - T[5];
  o[1] = T[2] - T[6];
  o[2] = T[3] - T[1];
}
// M = Q hat(e_m) : column operations on Q (hat(e_m) has two non-zeros)
__device__ __forceinline__ void q_hat_basis(const double *Q, int m, double *M) {
  // hat(e0) = [0 0 0; 0 0 -1; 0 1 0], hat(e1) = [0 0 1; 0 0 0; -1 0 0], hat(e2) = [0 -1 0; 1 0 0; 0 0 0]
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double q0 = Q[r * 3], q1 = Q[r * 3 + 1], q2 = Q[r * 3 + 2];
    if (m == 0) { M[r * 3] = 0; M[r * 3 + 1] = q2; M[r * 3 + 2] = -q1; }
    else if (m == 1) { M[r * 3] = -q2; M[r * 3 + 1] = 0; M[r * 3 + 2] = q0; }
    else { M[r * 3] = q1; M[r * 3 + 1] = -q0; M[r * 3 + 2] = 0; }
  }
}

// One thread per node: gradient, C_i, D_i, D_i^-1, and the off-diagonal blocks w B_ij of the Hessian.
// Sinc / winc: the measurement of every incidence as ITS node uses it (Rt_e at the head, Rt_e' at the tail) and its
// weight, in the incidences' own component-major slot order -- static, built once by mi_so3n_create.  Gathered from
// the edge arrays they were 72 bytes at a random place per incidence (1.56 cache lines on average, each edge fetched
// from both ends: ~600 MB per assembly at N = 5e5); as a stream they are 216 MB.
// r05: the OBJECTIVE rides along.  f(R) = 1/2 sum_e w_e |R_j - R_i Rt_e|^2 is 1/4 of the same sum over INCIDENCES (every
// edge is seen from both ends, |R_j - R_i Rt| = |R_i - R_j Rt'|), and the term R_i - R_j S is formed here anyway for the
// gradient: its squared norm costs 9 multiply-adds per incidence instead of an edge pass of its own with two more
// 72-byte gathers per edge (k_so3_objective: 61 us and 416 MB at the fabric per trial step at N = 5e5).  MODEL = false:
// the objective alone (mi_so3n_objective) -- the same loop, the same partial sums, the same bits.
// fpartials: one partial row per workgroup (<= kMaxRows workgroups: a workgroup walks groups of 4 slices).
// SQ (r05): the measurements are stored as unit quaternions (w, x, y, z) -- 32 instead of 72 bytes per incidence of the
// assembly's largest read stream (227 -> 101 MB at N = 5e5) -- and expanded here (12 products).  Chosen at creation, only
// when every measurement is a rotation to rounding (|S'S - I| <= 1e-13, det > 0): the expansion re-orthonormalises what it
// is given, which must not change the problem.
__device__ __forceinline__ void quat_to_mat(double w, double x, double y, double z, double *S) {
  const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  S[0] = 1 - 2 * (yy + zz); S[1] = 2 * (xy - wz);     S[2] = 2 * (xz + wy);
  S[3] = 2 * (xy + wz);     S[4] = 1 - 2 * (xx + zz); S[5] = 2 * (yz - wx);
  S[6] = 2 * (xz - wy);     S[7] = 2 * (yz + wx);     S[8] = 1 - 2 * (xx + yy);
}
// Rotation -> unit quaternion (w, x, y, z), Shepperd's branch on the largest of trace, R00, R11, R22 (no cancellation in
// the pivot), contraction off: the retraction's fused form and the conversion pass below must give the same bits for
// the same matrix whatever kernel they are inlined into.
__device__ __forceinline__ void mat_to_quat(const double *R, double *q) {
#pragma clang fp contract(off)
  const double tr = R[0] + R[4] + R[8];
  double w, x, y, z;
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2;  // 4 w
    w = .25 * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;  // 4 x
    w = (R[7] - R[5]) / s; x = .25 * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;  // 4 y
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = .25 * s; z = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;  // 4 z
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = .25 * s;
  }
  const double nrm = sqrt(w * w + x * x + y * y + z * z);
  q[0] = w / nrm; q[1] = x / nrm; q[2] = y / nrm; q[3] = z / nrm;
}
// Rq[4 i ...] = quaternion of R_i (r06): the 32-byte gather record of the model assembly
__global__ __launch_bounds__(256) void k_so3_quat(size_t N, const double *__restrict__ R, double *__restrict__ Rq) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += stride) {
    double Ri[9], q[4];
#pragma unroll
    for (int c = 0; c < 9; ++c) Ri[c] = R[9 * i + c];
    mat_to_quat(Ri, q);
    *reinterpret_cast<double2 *>(Rq + 4 * i) = make_double2(q[0], q[1]);
    *reinterpret_cast<double2 *>(Rq + 4 * i + 2) = make_double2(q[2], q[3]);
  }
}
// GQ (r06): the neighbour's rotation R_j is gathered as its unit quaternion from Rq (32 bytes, never
// across a 128-byte line; two 16-byte loads) and expanded, instead of as nine doubles at a 72-byte stride (1.56 lines on
// average).  Rq is written by the retraction of the trial step (k_so3_retract<true>) or, for a point the library did not
// produce, by k_so3_quat right before the assembly.  The own rotation R_i is read from R itself.
template <bool MODEL, bool SQ, bool GQ>
__device__ __forceinline__ void so3_model_slice(const IncView &inc, const double *__restrict__ R,
                                                const double *__restrict__ Rq,
                                                const double *__restrict__ Sinc, const double *__restrict__ winc,
                                                double *__restrict__ grad, double *__restrict__ Dinv,
                                                double *__restrict__ Bblk, double *__restrict__ Dsl, size_t slice,
                                                int lane, double &facc) {
  if (slice >= inc.nslices) return;
  const int node = inc.perm[slice * 64 + lane];
  const bool live = node >= 0;
  const size_t i = live ? (size_t)node : 0;
  double Ri[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) Ri[c] = live ? R[9 * i + c] : (c % 4 == 0 ? 1.0 : 0.0);
  double EG[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double degw = 0;
  const long long b0 = inc.slice_ptr[slice], b1 = inc.slice_ptr[slice + 1];
  for (long long k = b0; k < b1; ++k) {
    const size_t e0 = (size_t)k * 64 + lane;
    const double we = winc[e0];
    double Bk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live && we != 0) {
      const size_t j = (size_t)inc.nbr[e0];
      double S[9], Rj[9];
#if defined(MI_SO3_ABLATE_GATHER4)  // (timing experiment only: a 32-byte record per neighbour -- wrong results)
      {
        const double2 q0 = *reinterpret_cast<const double2 *>(R + 8 * (j & ~(size_t)1)), q1 = *reinterpret_cast<const double2 *>(R + 8 * (j & ~(size_t)1) + 2);
        Rj[0] = q0.x; Rj[1] = q0.y; Rj[2] = q1.x; Rj[3] = q1.y;
#pragma unroll
        for (int c = 4; c < 9; ++c) Rj[c] = Rj[c - 4] + 1.0;
      }
#elif defined(MI_SO3_ABLATE_GATHER0)  // (no neighbour gather at all)
#pragma unroll
      for (int c = 0; c < 9; ++c) Rj[c] = Ri[c] + we;
#else
      if (GQ) {
        const double2 qa = *reinterpret_cast<const double2 *>(Rq + 4 * j), qb = *reinterpret_cast<const double2 *>(Rq + 4 * j + 2);
        quat_to_mat(qa.x, qa.y, qb.x, qb.y, Rj);
      } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) Rj[c] = R[9 * j + c];
      }
#endif
      // head: term R_i - R_j Rt (j = tail); tail: R_i - R_j Rt'  (the transposition is in Sinc)
      if (SQ) {
        const double *sq = Sinc + (size_t)k * 4 * 64 + lane;
        quat_to_mat(sq[0], sq[64], sq[128], sq[192], S);
      } else {
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] = Sinc[((size_t)k * 9 + c) * 64 + lane];
      }
      double RjS[9];
      mat3_mul(Rj, S, RjS);
      double q = 0;
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const double d = Ri[c] - RjS[c];
        EG[c] += we * d;
        q += d * d;
      }
      facc += we * q;
      degw += we;
      if (!MODEL) continue;
      double Q[9];
      mat3_mul_at(Ri, Rj, Q);  // R_i' R_j
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double M[9], MS[9], col[3];
        q_hat_basis(Q, m, M);
        mat3_mul(M, S, MS);
        vee_skew2(MS, col);
        Bk[0 * 3 + m] = we * col[0];
        Bk[1 * 3 + m] = we * col[1];
        Bk[2 * 3 + m] = we * col[2];
      }
    }
    if (MODEL) {
#pragma unroll
      for (int c = 0; c < 9; ++c) Bblk[((size_t)k * 9 + c) * 64 + lane] = Bk[c];
    }
  }
  if (!MODEL) return;
  if (!live) {
#pragma unroll
    for (int c = 0; c < 9; ++c) Dsl[(slice * 9 + c) * 64 + lane] = 0.0;
    return;
  }
  double Q[9], C[9];
  mat3_mul_at(Ri, EG, Q);
  vee_skew2(Q, grad + 3 * i);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) C[a * 3 + b] = .5 * (Q[a * 3 + b] + Q[b * 3 + a]);
  const double tr = C[0] + C[4] + C[8], d = 2 * degw - tr;
  const double a = d + C[0], b = C[1], c = C[2], e = d + C[4], f = C[5], g = d + C[8];
  // (the block is kept component-major per slice only, for the coalesced SpMV stream: a second, node-ordered copy --
  // nine scattered 8-byte stores per node, read by nobody -- was written here until late r03)
  {
    const double dd[9] = {a, b, c, b, e, f, c, f, g};
#pragma unroll
    for (int q = 0; q < 9; ++q) Dsl[(slice * 9 + q) * 64 + lane] = dd[q];
  }
  const double c00 = e * g - f * f, c01 = c * f - b * g, c02 = b * f - c * e;
  const double c11 = a * g - c * c, c12 = b * c - a * f, c22 = a * e - b * b;
  const double det = a * c00 + b * c01 + c * c02;
  double *Di = Dinv + 9 * i;
  Di[0] = c00 / det; Di[1] = c01 / det; Di[2] = c02 / det;
  Di[3] = c01 / det; Di[4] = c11 / det; Di[5] = c12 / det;
  Di[6] = c02 / det; Di[7] = c12 / det; Di[8] = c22 / det;
}

template <bool MODEL, bool SQ, bool GQ>
__global__ __launch_bounds__(256) void k_so3_model(IncView inc, const double *__restrict__ R, const double *__restrict__ Rq,
                                                   const double *__restrict__ Sinc, const double *__restrict__ winc,
                                                   double *__restrict__ grad,
                                                   double *__restrict__ Dinv, double *__restrict__ Bblk,
                                                   double *__restrict__ Dsl, double *__restrict__ fpartials) {
  __shared__ double flds[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double facc = 0;
  const size_t ngroups = (inc.nslices + 3) / 4;
  for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x)
    so3_model_slice<MODEL, SQ, GQ>(inc, R, Rq, Sinc, winc, grad, Dinv, Bblk, Dsl, grp * 4 + w, lane, facc);
  // the workgroup's partial of the objective: workgroup b -> row b % kMaxRows of component b / kMaxRows (one workgroup
  // per group of slices keeps the dynamic balance of ~2000 short workgroups: a grid capped at kMaxRows rows cost the
  // assembly 30 us at N = 5e5); the components are added in fixed order by k_sum_slots
  facc = wave_reduce_sum(facc);
  if (lane == 0) flds[w] = facc;
  __syncthreads();
  if (threadIdx.x == 0) {
    fpartials[(size_t)(blockIdx.x / kMaxRows) * kMaxRows + blockIdx.x % kMaxRows] = (flds[0] + flds[1]) + (flds[2] + flds[3]);
    // the rows of the last component past the last workgroup read as zero (the buffer is shared with other reductions)
    const size_t total = (size_t)((gridDim.x + kMaxRows - 1) / kMaxRows) * kMaxRows, z = (size_t)gridDim.x + blockIdx.x;
    if (z < total) fpartials[z] = 0.0;
  }
}
// dst[0] = src[0] + ... + src[k-1] in index order (one thread)
__global__ void k_sum_slots(const double *__restrict__ src, int k, double *__restrict__ dst) {
  double s = 0;
  for (int i = 0; i < k; ++i) s += src[i];
  dst[0] = s;
}

// h = D xi - sum B_ij xi_j, fused with the three curvature dots (one 256-thread workgroup = 4 slices)
// NT: the matrix streams (diagonal blocks, off-diagonal blocks, neighbour indices -- read exactly once per pass) are
// loaded with the non-temporal policy, so that they do not evict the 12 MB of xi that the random chord gathers re-visit
// from the XCD's 4 MB L2 (A/B switch MI355OPT_BSR3_NT, measured in DESIGN 5.1).
template <bool NT>
__device__ __forceinline__ double ld_stream(const double *p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ int ld_stream(const int *p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool DOTS, bool NT>
__global__ __launch_bounds__(kBlock) void k_bsr3_spmv(IncView inc, const CgState *__restrict__ st,
                                                      const double *__restrict__ Dsl,
                                                      const double *__restrict__ Bblk,
                                                      const double *__restrict__ xi, double *__restrict__ h,
                                                      double *__restrict__ partials) {
  __shared__ double lds[3 * kWaves];
  if (st && st->mode != CG_RUN) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a[3] = {0, 0, 0};
  const size_t ngroups = (inc.nslices + kWaves - 1) / kWaves;
  const unsigned nb = gridDim.x, lb = xcd_remap(blockIdx.x, nb);
  const size_t g0 = (ngroups * lb) / nb, g1 = (ngroups * (lb + 1)) / nb;
  for (size_t g = g0; g < g1; ++g) {
    const size_t slice = g * kWaves + w;
    if (slice >= inc.nslices) continue;
    const int node = inc.perm[slice * 64 + lane];
    if (node < 0) continue;
    const size_t i = (size_t)node;  // degree-sorted inside the workgroup's 1024-node window
    const double x0 = xi[3 * i], x1 = xi[3 * i + 1], x2 = xi[3 * i + 2];
    const double *D = Dsl + slice * 9 * 64 + lane;
    double h0 = ld_stream<NT>(D) * x0 + ld_stream<NT>(D + 64) * x1 + ld_stream<NT>(D + 128) * x2;
    double h1 = ld_stream<NT>(D + 192) * x0 + ld_stream<NT>(D + 256) * x1 + ld_stream<NT>(D + 320) * x2;
    double h2 = ld_stream<NT>(D + 384) * x0 + ld_stream<NT>(D + 448) * x1 + ld_stream<NT>(D + 512) * x2;
    const long long b0 = inc.slice_ptr[slice], b1 = inc.slice_ptr[slice + 1];
    for (long long k = b0; k < b1; ++k) {
      const size_t e0 = (size_t)k * 64 + lane;
      const size_t j = (size_t)ld_stream<NT>(inc.nbr + e0);
      const double y0 = xi[3 * j], y1 = xi[3 * j + 1], y2 = xi[3 * j + 2];
      const double *B = Bblk + (size_t)k * 9 * 64 + lane;
      h0 -= ld_stream<NT>(B) * y0 + ld_stream<NT>(B + 64) * y1 + ld_stream<NT>(B + 128) * y2;
      h1 -= ld_stream<NT>(B + 192) * y0 + ld_stream<NT>(B + 256) * y1 + ld_stream<NT>(B + 320) * y2;
      h2 -= ld_stream<NT>(B + 384) * y0 + ld_stream<NT>(B + 448) * y1 + ld_stream<NT>(B + 512) * y2;
    }
    h[3 * i] = h0; h[3 * i + 1] = h1; h[3 * i + 2] = h2;
    if (DOTS) {
      a[0] += x0 * h0; a[0] += x1 * h1; a[0] += x2 * h2;
      a[1] += h0 * h0; a[1] += h1 * h1; a[1] += h2 * h2;
      a[2] += x0 * x0; a[2] += x1 * x1; a[2] += x2 * x2;
    }
  }
  if (DOTS) block_partials_store<3>(a, lds, partials);
}

// Y_i = R_i exp(hat(xi_i))  (Rodrigues; same series switch as oracle/problems.c: orc_so3_exp)
// (80 scalar registers: see stpcg.hip, k_cg_update_s80)
// WQ (r06): also the quaternion of every Y_i into Yq (the model assembly's gather record at the trial point)
template <bool WQ>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_num_sgpr(80))) void k_so3_retract(size_t N, const double *__restrict__ R,
                                                        const double *__restrict__ xi, double *__restrict__ Y,
                                                        double *__restrict__ Yq) {
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
    const double x0 = xi[3 * i], x1 = xi[3 * i + 1], x2 = xi[3 * i + 2];
    const double th2 = x0 * x0 + x1 * x1 + x2 * x2, th = sqrt(th2);
    double a, b;
    if (th < 1e-4) {
      a = 1 - th2 / 6 + th2 * th2 / 120;
      b = .5 - th2 / 24 + th2 * th2 / 720;
    } else {
      a = sin(th) / th;
      b = (1 - cos(th)) / th2;
    }
    const double K[9] = {0, -x2, x1, x2, 0, -x0, -x1, x0, 0};
    double K2[9], Ex[9], Ri[9], Yi[9];
    mat3_mul(K, K, K2);
#pragma unroll
    for (int c = 0; c < 9; ++c) Ex[c] = a * K[c] + b * K2[c];
    Ex[0] += 1; Ex[4] += 1; Ex[8] += 1;
#pragma unroll
    for (int c = 0; c < 9; ++c) Ri[c] = R[9 * i + c];
    mat3_mul(Ri, Ex, Yi);
#pragma unroll
    for (int c = 0; c < 9; ++c) Y[9 * i + c] = Yi[c];
    if (WQ) {
      double q[4];
      mat_to_quat(Yi, q);
      *reinterpret_cast<double2 *>(Yq + 4 * i) = make_double2(q[0], q[1]);
      *reinterpret_cast<double2 *>(Yq + 4 * i + 2) = make_double2(q[2], q[3]);
    }
  }
}

int upload(void **dst, const void *src, size_t bytes) {
  MI_HIP(hipMalloc(dst, bytes ? bytes : 8));
  if (bytes) MI_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return MI_OK;
}

}  // namespace

struct mi_so3n {
  mi_ctx *ctx = nullptr;
  size_t N = 0, E = 0, nslices = 0, nnzb = 0, padded = 0;
  long long *slice_ptr = nullptr;
  int *perm = nullptr, *nbr = nullptr;
  mi_vec *Dinv = nullptr;  // 9N (node order): the inverse diagonal blocks, the block-Jacobi preconditioner
  double *Bblk = nullptr;                   // padded * 9
  double *Sinc = nullptr, *winc = nullptr;  // padded * 9 (or * 4: sinc_quat), padded: per-incidence measurement and weight
  bool sinc_quat = false;                   // Sinc holds unit quaternions (all measurements are rotations to rounding)
  double *Dsl = nullptr;                    // nslices * 9 * 64: diagonal blocks in slice order
  double *Rq = nullptr;                     // 4 N: unit quaternions of the point the NEXT assembly gathers from (r06; null:
                                            // SO3_NO_RQUAT -- the assembly gathers R itself)
  mi_op hess;
  mi_precon bj;
  const mi_vec *R = nullptr;                // the point the model is bound to (mi_so3n_model) ...
  uint64_t R_serial = 0;                    // ... by handle AND serial (handles are recycled)
  // mi_so3n_trial: the model assembled speculatively at the trial point (a second set of the arrays above plus the
  // gradient), swapped in by the next mi_so3n_model call if that call is for the same vector -- keyed on the
  // handle AND the identity of its contents (mi_vec::serial / gen), as in stiefel.hip
  mi_vec *Dinv_next = nullptr, *grad_next = nullptr, *Hh = nullptr, *Pg = nullptr;
  double *Bblk_next = nullptr, *Dsl_next = nullptr;
  const mi_vec *trial_R = nullptr;
  const double *trial_d = nullptr;
  uint64_t trial_serial = 0, trial_gen = 0;
  bool is_trial(const mi_vec *X) const {
    return trial_R == X && trial_d == X->d && trial_serial == X->serial && trial_gen == gen_of(X);
  }
};

namespace {

// workgroups of the model assembly: one per group of 4 slices while their objective partials fit 8 components of
// kMaxRows rows (N <= 2.1e6 rotations), a grid-stride walk beyond
constexpr int kModelComps = 8;
int model_grid(const mi_so3n *q) { return (int)std::min<size_t>((q->nslices + 3) / 4, (size_t)kModelComps * kMaxRows); }
IncView view(const mi_so3n *q) {
  return IncView{q->N, q->nslices, q->slice_ptr, q->perm, q->nbr};
}
// the assembly (model = true) or its objective alone, objective partials into ctx->partials2
// have_quat: q->Rq already holds the quaternions of R (the trial step's retraction wrote them); otherwise they are formed
// here first (one pass over R: a point the library did not produce -- the start, or the statement sequence's calls)
void launch_model(mi_so3n *q, bool model, int grid, const double *R, double *grad, double *Dinv, double *Bblk, double *Dsl,
                  bool have_quat = false) {
  mi_ctx *ctx = q->ctx;
  const bool gq = q->Rq != nullptr;
  if (gq && !have_quat)
    hipLaunchKernelGGL(k_so3_quat, dim3(grid_for(ctx, q->N, 1)), dim3(256), 0, ctx->stream, q->N, R, q->Rq);
#define SO3M(MV, SQV, GQV)                                                                                      \
  hipLaunchKernelGGL((k_so3_model<MV, SQV, GQV>), dim3(grid), dim3(256), 0, ctx->stream, view(q), R,             \
                     (const double *)q->Rq, (const double *)q->Sinc, (const double *)q->winc, grad, Dinv, Bblk, Dsl, \
                     ctx->partials2)
#define SO3M_G(MV, SQV) if (gq) SO3M(MV, SQV, true); else SO3M(MV, SQV, false)
  if (model) { if (q->sinc_quat) { SO3M_G(true, true); } else { SO3M_G(true, false); } }
  else { if (q->sinc_quat) { SO3M_G(false, true); } else { SO3M_G(false, false); } }
#undef SO3M_G
#undef SO3M
}
// the objective partials the assembly left in ctx->partials2 -> one (all-reduced) sum in slots[0]
int model_objective_to_slot(mi_so3n *q, int grid, double *slot) {
  mi_ctx *ctx = q->ctx;
  const int comps = (grid + kMaxRows - 1) / kMaxRows;
  if (comps == 1) return reduce_rows_allreduce(ctx, ctx->partials2, grid, 1, slot);
  // (the rows of the last component past the last workgroup were zeroed by the kernel itself)
  double *tmp = ctx->scalars + SLOT_RAW;  // (free outside mi_stiefel_gram)
  MI_TRY(reduce_rows_allreduce(ctx, ctx->partials2, kMaxRows, comps, tmp));
  hipLaunchKernelGGL(k_sum_slots, dim3(1), dim3(1), 0, ctx->stream, (const double *)tmp, comps, slot);
  MI_HIP(hipGetLastError());
  return MI_OK;
}


int so3_apply_common(mi_op *self, const mi_vec *in, mi_vec *out, bool dots, int *nparts) {
  mi_so3n *q = (mi_so3n *)self->impl;
  mi_ctx *ctx = q->ctx;
  const size_t ngroups = (q->nslices + kWaves - 1) / kWaves;
  const int grid = uniform_grid(ctx, ngroups);
  KScope ks(ctx, MI_K_BSR3_SPMV_DOTS);
  constexpr bool nt = kBsr3NtDefault;
#define BSR3(DV, NTV, STATE, PART)                                                                               \
  hipLaunchKernelGGL((k_bsr3_spmv<DV, NTV>), dim3(grid), dim3(kBlock), 0, ctx->stream, view(q), STATE,          \
                     (const double *)q->Dsl, (const double *)q->Bblk, (const double *)in->d, out->d, PART)
  if (dots) {
    if (nt) BSR3(true, true, ctx->cg_live, ctx->partials); else BSR3(true, false, ctx->cg_live, ctx->partials);
  } else {
    if (nt) BSR3(false, true, (const CgState *)nullptr, (double *)nullptr);
    else BSR3(false, false, (const CgState *)nullptr, (double *)nullptr);
  }
#undef BSR3
  if (nparts) *nparts = grid;
  MI_HIP(hipGetLastError());
  return MI_OK;
}
int so3_apply(mi_op *self, const mi_vec *in, mi_vec *out) { return so3_apply_common(self, in, out, false, nullptr); }
int so3_apply_dots(mi_op *self, const mi_vec *in, mi_vec *out, int *np) {
  return so3_apply_common(self, in, out, true, np);
}
int so3_bj_apply(mi_precon *self, const mi_vec *r, mi_vec *v) {
  mi_precon *tmp = nullptr;
  MI_TRY(mi_precon_create_block3(self->ctx, ((mi_so3n *)self->impl)->Dinv, &tmp));
  const int s = mi_precon_apply(tmp, r, v);
  mi_precon_destroy(tmp);
  return s;
}

}  // namespace

extern "C" {

int mi_so3n_create(mi_ctx *ctx, size_t N, size_t E, const int32_t *ei, const int32_t *ej, const double *Rt,
                   const double *w, mi_so3n **out) {
  MI_REQUIRE(ctx && out && (E == 0 || (ei && ej && Rt && w)), "null argument");
  MI_REQUIRE(N > 0 && N < (size_t)INT32_MAX, "bad number of rotations");
  for (size_t e = 0; e < E; ++e)
    MI_REQUIRE(ei[e] >= 0 && (size_t)ei[e] < N && ej[e] >= 0 && (size_t)ej[e] < N && ei[e] != ej[e],
               "edge %zu has invalid endpoints", e);
  // incidence lists: node -> (neighbour, edge, direction), in edge order
  std::vector<std::vector<int>> lists(N);
  for (size_t e = 0; e < E; ++e) {
    lists[(size_t)ej[e]].push_back((int)e + 1);      // head: +(e+1)
    lists[(size_t)ei[e]].push_back(-((int)e + 1));   // tail: -(e+1)
  }
  // Experiment switch SO3_SORT_NBR (r04, VERDICT item 3: "incidences ordered in column phases"): every node's incidences
  // in ascending order of the NEIGHBOUR instead of edge order.  The whole product is one resident set of waves (N / 64
  // waves, one node per lane) that start together and walk their incidence slots at the same pace, so slot k of every
  // row then points into the same quantile region of xi at about the same time -- the chance that a chord gather finds
  // its 128-byte line in the XCD's 4 MB L2 should rise.  Measured in DESIGN 5.1; changes the order of a row's sum.
  if (ctx->cfg.so3_sort_nbr)
    for (size_t i = 0; i < N; ++i)
      std::stable_sort(lists[i].begin(), lists[i].end(), [&](int a, int b) {
        const int ea = std::abs(a) - 1, eb = std::abs(b) - 1;
        const int ja = a > 0 ? ei[ea] : ej[ea], jb = b > 0 ? ei[eb] : ej[eb];
        return ja < jb;
      });
  // SELL-64-sigma, sigma = kBlock = the 1024 nodes one workgroup pass owns: inside each window the
  // nodes are ordered by descending degree (stable), so a slice's 64 nodes have near-equal degree
  // and the padding of an irregular pose graph drops from ~1.66x to ~1.05x of the incidences.  The
  // window is a contiguous node range, so x_i loads / h_i stores stay inside one 24 KB span per
  // workgroup pass; tangent vectors keep the caller's node order.
  const size_t nslices = (N + 63) / 64;
  std::vector<int> perm(nslices * 64, -1);
  for (size_t w0 = 0; w0 < N; w0 += kBlock) {
    const size_t w1 = std::min(N, w0 + kBlock);
    std::vector<int> ids(w1 - w0);
    for (size_t i = w0; i < w1; ++i) ids[i - w0] = (int)i;
    std::stable_sort(ids.begin(), ids.end(),
                     [&](int a, int b) { return lists[(size_t)a].size() > lists[(size_t)b].size(); });
    for (size_t i = w0; i < w1; ++i) perm[i] = ids[i - w0];
  }
  std::vector<long long> sp(nslices + 1, 0);
  for (size_t s = 0; s < nslices; ++s) {
    size_t wmax = 0;
    for (size_t l = s * 64; l < (s + 1) * 64; ++l)
      if (perm[l] >= 0) wmax = std::max(wmax, lists[(size_t)perm[l]].size());
    sp[s + 1] = sp[s] + (long long)wmax;
  }
  const size_t padded = (size_t)sp[nslices] * 64;
  std::vector<int> nbr(padded), edge(padded, -1);
  std::vector<signed char> dir(padded, 0);
  size_t nnzb = 0;
  for (size_t s = 0; s < nslices; ++s)
    for (int lane = 0; lane < 64; ++lane) {
      const bool owned = perm[s * 64 + lane] >= 0;
      const size_t i = owned ? (size_t)perm[s * 64 + lane] : N - 1;
      for (long long k = 0; k < sp[s + 1] - sp[s]; ++k) {
        const size_t e0 = (size_t)(sp[s] + k) * 64 + lane;
        nbr[e0] = (int)i;
        if (owned && (size_t)k < lists[i].size()) {
          const int code = lists[i][(size_t)k];
          const int e = std::abs(code) - 1;
          edge[e0] = e;
          dir[e0] = code > 0 ? 1 : -1;
          nbr[e0] = code > 0 ? ei[e] : ej[e];
          ++nnzb;
        }
      }
    }
  // the measurement and weight of every incidence in slot order (k_so3_model streams them)
  // ... the measurements as unit quaternions when every one of them is a rotation to rounding (k_so3_model<., SQ>)
  bool all_rot = !ctx->cfg.so3_no_quat;
  for (size_t e = 0; e < E && all_rot; ++e) {
    const double *M = Rt + 9 * e;
    for (int a = 0; a < 3 && all_rot; ++a)
      for (int b = 0; b < 3; ++b) {
        const double d = M[a] * M[b] + M[3 + a] * M[3 + b] + M[6 + a] * M[6 + b] - (a == b ? 1.0 : 0.0);
        if (!(std::fabs(d) <= 1e-13)) all_rot = false;
      }
    const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
    if (!(det > 0)) all_rot = false;
  }
  auto to_quat = [](const double *M, double *qv) {  // Shepperd's branch on the largest of (trace, m00, m11, m22)
    const double tr = M[0] + M[4] + M[8];
    double w, x, y, z;
    if (tr >= M[0] && tr >= M[4] && tr >= M[8]) {
      w = 1 + tr; x = M[7] - M[5]; y = M[2] - M[6]; z = M[3] - M[1];
    } else if (M[0] >= M[4] && M[0] >= M[8]) {
      w = M[7] - M[5]; x = 1 + M[0] - M[4] - M[8]; y = M[1] + M[3]; z = M[2] + M[6];
    } else if (M[4] >= M[8]) {
      w = M[2] - M[6]; x = M[1] + M[3]; y = 1 - M[0] + M[4] - M[8]; z = M[5] + M[7];
    } else {
      w = M[3] - M[1]; x = M[2] + M[6]; y = M[5] + M[7]; z = 1 - M[0] - M[4] + M[8];
    }
    const double nrm = std::sqrt(w * w + x * x + y * y + z * z);
    qv[0] = w / nrm; qv[1] = x / nrm; qv[2] = y / nrm; qv[3] = z / nrm;
  };
  const int scomp = all_rot ? 4 : 9;
  std::vector<double> sinc(std::max<size_t>(1, padded * scomp), 0.0), winc(std::max<size_t>(1, padded), 0.0);
  for (size_t s = 0; s < nslices; ++s)
    for (long long k = sp[s]; k < sp[s + 1]; ++k)
      for (int lane = 0; lane < 64; ++lane) {
        const size_t e0 = (size_t)k * 64 + lane;
        const int e = edge[e0];
        if (e < 0) continue;
        winc[e0] = w[e];
        double Sm[9];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) Sm[r * 3 + c] = dir[e0] > 0 ? Rt[9 * (size_t)e + r * 3 + c] : Rt[9 * (size_t)e + c * 3 + r];
        if (all_rot) {
          double qv[4];
          to_quat(Sm, qv);
          for (int c = 0; c < 4; ++c) sinc[((size_t)k * 4 + c) * 64 + lane] = qv[c];
        } else {
          for (int c = 0; c < 9; ++c) sinc[((size_t)k * 9 + c) * 64 + lane] = Sm[c];
        }
      }
  mi_so3n *q = new mi_so3n();
  q->ctx = ctx;
  q->N = N;
  q->E = E;
  q->nslices = nslices;
  q->nnzb = nnzb;
  q->padded = padded;
  q->sinc_quat = all_rot;
  MI_TRY(upload((void **)&q->slice_ptr, sp.data(), sp.size() * sizeof(long long)));
  MI_TRY(upload((void **)&q->perm, perm.data(), perm.size() * sizeof(int)));
  MI_HIP(hipMalloc((void **)&q->Dsl, nslices * 9 * 64 * sizeof(double)));
  if (!ctx->cfg.so3_no_rquat) MI_HIP(hipMalloc((void **)&q->Rq, std::max<size_t>(1, N) * 4 * sizeof(double)));
  MI_TRY(upload((void **)&q->nbr, nbr.data(), padded * sizeof(int)));
  MI_HIP(hipMalloc((void **)&q->Bblk, std::max<size_t>(1, padded * 9) * sizeof(double)));
  MI_TRY(upload((void **)&q->Sinc, sinc.data(), sinc.size() * sizeof(double)));
  MI_TRY(upload((void **)&q->winc, winc.data(), winc.size() * sizeof(double)));
  MI_TRY(mi_vec_create(ctx, 9 * N, &q->Dinv));
  q->hess.ctx = ctx;
  q->hess.n = 3 * N;
  q->hess.apply = so3_apply;
  q->hess.apply_dots = so3_apply_dots;
  q->hess.impl = q;
  q->hess.borrowed = true;
  q->bj.ctx = ctx;
  q->bj.n = 3 * N;
  q->bj.kind = 2;  // fusable 3x3 block-Jacobi: data = inverse blocks
  q->bj.data = q->Dinv->d;
  q->bj.apply = so3_bj_apply;
  q->bj.impl = q;
  q->bj.borrowed = true;
  *out = q;
  return MI_OK;
}

int mi_so3n_destroy(mi_so3n *q) {
  if (!q) return MI_OK;
  (void)hipStreamSynchronize(q->ctx->stream);
  (void)hipFree(q->slice_ptr); (void)hipFree(q->nbr);
  (void)hipFree(q->Bblk); (void)hipFree(q->perm); (void)hipFree(q->Dsl);
  if (q->Rq) (void)hipFree(q->Rq);
  (void)hipFree(q->Sinc); (void)hipFree(q->winc);
  (void)hipFree(q->Bblk_next); (void)hipFree(q->Dsl_next);
  mi_vec_destroy(q->Dinv);
  mi_vec_destroy(q->Dinv_next);
  mi_vec_destroy(q->grad_next);
  mi_vec_destroy(q->Hh);
  mi_vec_destroy(q->Pg);
  delete q;
  return MI_OK;
}

int mi_so3n_objective(mi_so3n *q, const mi_vec *R, double *f) {
  MI_REQUIRE(q && R && f, "null argument");
  MI_REQUIRE(R->ctx == q->ctx && R->n == 9 * q->N, "R must hold N row-major 3x3 blocks");
  mi_ctx *ctx = q->ctx;
  // the incidence form of the sum (r05): the loop of the model assembly without its outputs -- what mi_so3n_trial gets
  // from the assembly at the trial point itself, bit for bit
  const int grid = model_grid(q);
  launch_model(q, false, grid, R->d, nullptr, nullptr, nullptr, nullptr);
  MI_HIP(hipGetLastError());
  MI_TRY(model_objective_to_slot(q, grid, ctx->scalars + SLOT_MISC));
  double s = 0;
  MI_TRY(read_slots_sync(ctx, SLOT_MISC, 1, &s));
  *f = .25 * s;  // (every edge from both ends)
  return MI_OK;
}

int mi_so3n_model(mi_so3n *q, const mi_vec *R, mi_vec *grad, mi_op **hess, mi_precon **block_jacobi) {
  MI_REQUIRE(q && R && grad, "null argument");
  MI_REQUIRE(R->ctx == q->ctx && R->n == 9 * q->N, "R must hold N row-major 3x3 blocks");
  MI_REQUIRE(grad->n == 3 * q->N, "gradient must hold 3N doubles");
  touch(grad);
  mi_ctx *ctx = q->ctx;
  if (q->is_trial(R)) {
    // R is the point mi_so3n_trial just evaluated: its model exists already
    std::swap(q->Dinv, q->Dinv_next);
    std::swap(q->Bblk, q->Bblk_next);
    std::swap(q->Dsl, q->Dsl_next);
    q->bj.data = q->Dinv->d;
    MI_TRY(mi_vec_copy(grad, q->grad_next));
  } else {
    launch_model(q, true, model_grid(q), R->d, grad->d, q->Dinv->d, q->Bblk, q->Dsl);
    MI_HIP(hipGetLastError());
  }
  q->trial_R = nullptr;
  q->R = R;
  q->R_serial = R->serial;
  if (hess) *hess = &q->hess;
  if (block_jacobi) *block_jacobi = &q->bj;
  return MI_OK;
}

int mi_so3n_retract(mi_so3n *q, const mi_vec *R, const mi_vec *xi, mi_vec *Y) {
  MI_REQUIRE(q && R && xi && Y, "null argument");
  MI_REQUIRE(R->n == 9 * q->N && Y->n == 9 * q->N && xi->n == 3 * q->N, "dimension mismatch");
  touch(Y);
  hipLaunchKernelGGL(k_so3_retract<false>, dim3(grid_for(q->ctx, q->N, 1)), dim3(kBlock), 0, q->ctx->stream, q->N,
                     (const double *)R->d, (const double *)xi->d, Y->d, (double *)nullptr);
  MI_HIP(hipGetLastError());
  return MI_OK;
}

// One trial step of a trust-region method at the point R the model is bound to (reference Riemannian/TNT.h:493-512,
// 573-585): Hess h with |h|^2, <g,h>, <h,Hess h>; R+ = retract(R, h); f(R+); and -- speculatively -- the model at R+
// (gradient, Hessian blocks, block-Jacobi inverse blocks) with |grad f(R+)|^2 and, if asked, |M+^-1 grad f(R+)|^2: ONE
// launch chain, ONE read-back.  Every number comes from the kernels, grids and summation orders the separate calls
// (mi_op_apply + mi_vec_dot_batch, mi_so3n_retract, mi_so3n_objective, mi_so3n_model, mi_vec_dot, mi_precon_apply)
// use.  out[6] = {f(R+), <h,h>, <g,h>, <h,Hess h>, |grad f(R+)|^2, |M+^-1 grad f(R+)|^2 (-1 unless with_precon)}.
int mi_so3n_trial(mi_so3n *q, const mi_vec *R, const mi_vec *h, const mi_vec *g, int with_precon, mi_vec *R_trial,
                  double out[6]) {
  MI_REQUIRE(q && R && h && g && R_trial && out, "null argument");
  MI_REQUIRE(q->R == R && q->R_serial == R->serial,
             "mi_so3n_trial: the model is not bound to this R (call mi_so3n_model first)");
  MI_REQUIRE(R->n == 9 * q->N && R_trial->n == 9 * q->N && h->n == 3 * q->N && g->n == 3 * q->N, "dimension mismatch");
  MI_REQUIRE(R_trial->d != R->d, "the trial point must not alias the current one");
  mi_ctx *ctx = q->ctx;
  ctx->fusion.fused_trial_steps++;
  const size_t N3 = 3 * q->N;
  if (!q->grad_next) {
    MI_TRY(mi_vec_create(ctx, 9 * q->N, &q->Dinv_next));
    MI_TRY(mi_vec_create(ctx, N3, &q->grad_next));
    MI_TRY(mi_vec_create(ctx, N3, &q->Hh));
    MI_TRY(mi_vec_create(ctx, N3, &q->Pg));
    MI_HIP(hipMalloc((void **)&q->Bblk_next, std::max<size_t>(1, q->padded * 9) * sizeof(double)));
    MI_HIP(hipMalloc((void **)&q->Dsl_next, q->nslices * 9 * 64 * sizeof(double)));
  }
  q->trial_R = nullptr;
  // (a) Hess h, then |h|^2, <g,h>, <h, Hess h> in one pass (as MI355::dot_batch does)
  MI_TRY(so3_apply(&q->hess, h, q->Hh));
  {
    const double *xs[3] = {h->d, g->d, h->d}, *ys[3] = {h->d, h->d, q->Hh->d};
    MI_TRY(dot_batch_to_slots(ctx, 3, xs, ys, N3, SLOT_MISC));
  }
  // (b) R+ = R exp(hat h) (+ the quaternions of R+, the assembly's gather records: the bits k_so3_quat would give)
  if (q->Rq) {
    touch(R_trial);
    hipLaunchKernelGGL(k_so3_retract<true>, dim3(grid_for(ctx, q->N, 1)), dim3(kBlock), 0, ctx->stream, q->N,
                       (const double *)R->d, (const double *)h->d, R_trial->d, q->Rq);
    MI_HIP(hipGetLastError());
  } else {
    MI_TRY(mi_so3n_retract(q, R, h, R_trial));
  }
  // (c) + (d) the model at R+ into the second set of arrays, with f(R+) from the same pass (r05: the objective was an
  // edge pass of its own, 61 us and 416 MB at the fabric) -- reduced exactly as mi_so3n_objective does
  {
    const int grid = model_grid(q);
    launch_model(q, true, grid, R_trial->d, q->grad_next->d, q->Dinv_next->d, q->Bblk_next, q->Dsl_next, q->Rq != nullptr);
    MI_HIP(hipGetLastError());
    MI_TRY(model_objective_to_slot(q, grid, ctx->scalars + SLOT_MISC + 3));
  }
  {
    const double *xs[1] = {q->grad_next->d}, *ys[1] = {q->grad_next->d};
    MI_TRY(dot_batch_to_slots(ctx, 1, xs, ys, N3, SLOT_MISC + 4));
  }
  if (with_precon) {
    mi_precon *tmp = nullptr;
    MI_TRY(mi_precon_create_block3(ctx, q->Dinv_next, &tmp));
    const int s = mi_precon_apply(tmp, q->grad_next, q->Pg);
    mi_precon_destroy(tmp);
    MI_TRY(s);
    const double *xs[1] = {q->Pg->d}, *ys[1] = {q->Pg->d};
    MI_TRY(dot_batch_to_slots(ctx, 1, xs, ys, N3, SLOT_MISC + 5));
  }
  // (e) one read-back
  static_assert(SLOT_MISC + 6 <= SLOT_GDIR, "slot map");
  double buf[6];
  MI_TRY(read_slots_sync(ctx, SLOT_MISC, with_precon ? 6 : 5, buf));
  out[0] = .25 * buf[3];
  out[1] = buf[0];
  out[2] = buf[1];
  out[3] = buf[2];
  out[4] = buf[4];
  out[5] = with_precon ? buf[5] : -1.0;
  q->trial_R = R_trial;
  q->trial_d = R_trial->d;
  q->trial_serial = R_trial->serial;
  q->trial_gen = gen_of(R_trial);
  return MI_OK;
}

}  // extern "C"
