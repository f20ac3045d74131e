// prologue.hip -- microbenchmark: what does a "re-reduce the partial rows in every workgroup's
// prologue" cost compared with reading pre-reduced scalars, for an update-like streaming kernel that
// follows a producer kernel?  Build: hipcc --offload-arch=gfx950 -O3 -I include -I optimization_amd/csrc
// tools/microbench/prologue.hip -o gpurun_out/prologue && ./gpurun_out/prologue
#include <cstdio>
#include <vector>

#include "mi_internal.h"

using namespace mi;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// producer: Hp = 1.0001 * Z ; partial rows of 3 dots
__global__ __launch_bounds__(kBlock) void producer(size_t n, const double *__restrict__ Z,
                                                   const double *__restrict__ p, double *__restrict__ Hp,
                                                   double *__restrict__ partials) {
  __shared__ double lds[3 * kWaves];
  double a[3] = {0, 0, 0};
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 z = reinterpret_cast<const double2 *>(Z)[i];
    const double2 pv = reinterpret_cast<const double2 *>(p)[i];
    const double2 h = make_double2(1.0001 * z.x, 1.0001 * z.y);
    reinterpret_cast<double2 *>(Hp)[i] = h;
    a[0] += pv.x * h.x + pv.y * h.y; a[1] += h.x * h.x + h.y * h.y; a[2] += pv.x * pv.x + pv.y * pv.y;
  }
  block_partials_store<3>(a, lds, partials);
}

// MODE 0: alpha from slots (no prologue reduce)        1: reduce_rows<3> prologue
//      2: reduce prologue, but first body loads issued BEFORE the prologue (software prefetch)
//      3: like 1 but only wave 0 of each workgroup reduces (others wait at one barrier)
template <int MODE>
__global__ __launch_bounds__(kBlock) void consumer(size_t n, const double *__restrict__ partials, int count,
                                                   const double *__restrict__ slots,
                                                   const double *__restrict__ p, const double *__restrict__ Hp,
                                                   double *__restrict__ s, double *__restrict__ r,
                                                   double *__restrict__ partials_b) {
  __shared__ double lds[3 * (kWaves + 1)];
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  double2 pv0, hv0, sv0, rv0;
  if (MODE == 2 && i < n2) {
    pv0 = reinterpret_cast<const double2 *>(p)[i];
    hv0 = reinterpret_cast<const double2 *>(Hp)[i];
    sv0 = reinterpret_cast<double2 *>(s)[i];
    rv0 = reinterpret_cast<double2 *>(r)[i];
  }
  double d[3];
  if (MODE == 0) {
    d[0] = slots[0]; d[1] = slots[1]; d[2] = slots[2];
  } else if (MODE == 3) {
    if (threadIdx.x < 64) {
      double t[3] = {0, 0, 0};
      for (int rr = threadIdx.x; rr < count; rr += 64)
        for (int k = 0; k < 3; ++k) t[k] += partials[(size_t)k * kMaxRows + rr];
      for (int k = 0; k < 3; ++k) {
        const double v = wave_reduce_sum(t[k]);
        if (threadIdx.x == 0) lds[k] = v;
      }
    }
    __syncthreads();
    d[0] = lds[0]; d[1] = lds[1]; d[2] = lds[2];
  } else {
    reduce_rows<3>(partials, count, d, lds);
  }
  const double alpha = 1e-3 * d[0] / (d[1] + d[2] + 1.0);
  double acc[1] = {0};
  bool first = (MODE == 2);
  for (; i < n2; i += stride) {
    double2 pv, hv, sv, rv;
    if (first) { pv = pv0; hv = hv0; sv = sv0; rv = rv0; first = false; }
    else {
      pv = reinterpret_cast<const double2 *>(p)[i];
      hv = reinterpret_cast<const double2 *>(Hp)[i];
      sv = reinterpret_cast<double2 *>(s)[i];
      rv = reinterpret_cast<double2 *>(r)[i];
    }
    sv.x += alpha * pv.x; sv.y += alpha * pv.y;
    rv.x += alpha * hv.x; rv.y += alpha * hv.y;
    reinterpret_cast<double2 *>(s)[i] = sv;
    reinterpret_cast<double2 *>(r)[i] = rv;
    acc[0] += rv.x * rv.x + rv.y * rv.y;
  }
  block_partials_store<1>(acc, lds, partials_b);
}

// closer replica of k_cg_update: state struct, scalar step with sqrt/div, leader write-back,
// mode-dependent bodies.  VAR bit0: leader writes st_out; bit1: mode branches; bit2: real step math
template <int VAR>
__global__ __launch_bounds__(kBlock) void consumer2(size_t n, const CgState *__restrict__ st_in,
                                                    CgState *__restrict__ st_out,
                                                    const double *__restrict__ partials, int count,
                                                    const double *__restrict__ p, const double *__restrict__ Hp,
                                                    double *__restrict__ s, double *__restrict__ r,
                                                    double *__restrict__ partials_b) {
  __shared__ double lds[3 * (kWaves + 1)];
  CgState cs = *st_in;
  if (cs.mode == CG_DONE) return;
  double d[3];
  reduce_rows<3>(partials, count, d, lds);
  if (VAR & 4) {
    cs.kappa = d[0];
    if (sqrt(d[1]) / sqrt(d[2]) < cs.epsilon) cs.mode = CG_KERNEL_PENDING;
    else {
      const double alpha = cs.rv / d[0];
      const double sk = cs.sk_M_2 + 2 * alpha * cs.sk_M_pk + alpha * alpha * cs.pk_M_2;
      if (d[0] <= 0 || sk > cs.Delta_2) {
        cs.sigma = (-cs.sk_M_pk + sqrt(cs.sk_M_pk * cs.sk_M_pk + cs.pk_M_2 * (cs.Delta_2 - cs.sk_M_2))) / cs.pk_M_2;
        cs.mode = CG_APPLY_SIGMA;
      } else { cs.alpha = alpha; cs.skplus1_M_2 = sk; }
    }
  } else {
    cs.alpha = 1e-3 * d[0] / (d[1] + d[2] + 1.0);
  }
  if ((VAR & 1) && !(VAR & 2) && blockIdx.x == 0 && threadIdx.x == 0) *st_out = cs;   // early, one thread
  if ((VAR & 1) && (VAR & 2) && blockIdx.x == 0 && threadIdx.x < sizeof(CgState) / 8) {  // early, cooperative
    const double *src = reinterpret_cast<const double *>(&cs);
    reinterpret_cast<double *>(st_out)[threadIdx.x] = src[threadIdx.x];
  }
  const size_t n2 = n >> 1, stride = (size_t)gridDim.x * kBlock;
  const size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x;
  double acc[1] = {0};
  const int mode = cs.mode;
  if (mode == CG_APPLY_SIGMA) {
    const double sigma = cs.sigma;
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      double2 sv = reinterpret_cast<double2 *>(s)[i];
      sv.x += sigma * pv.x; sv.y += sigma * pv.y;
      reinterpret_cast<double2 *>(s)[i] = sv;
    }
    return;
  }
  if (mode == CG_KERNEL_PENDING) {
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      const double2 rv = reinterpret_cast<const double2 *>(r)[i];
      acc[0] += pv.x * rv.x + pv.y * rv.y;
    }
  } else {
    const double alpha = cs.alpha;
    for (size_t i = i0; i < n2; i += stride) {
      const double2 pv = reinterpret_cast<const double2 *>(p)[i];
      const double2 hv = reinterpret_cast<const double2 *>(Hp)[i];
      double2 sv = reinterpret_cast<double2 *>(s)[i];
      double2 rv = reinterpret_cast<double2 *>(r)[i];
      sv.x += alpha * pv.x; sv.y += alpha * pv.y;
      rv.x += alpha * hv.x; rv.y += alpha * hv.y;
      reinterpret_cast<double2 *>(s)[i] = sv;
      reinterpret_cast<double2 *>(r)[i] = rv;
      acc[0] += rv.x * rv.x + rv.y * rv.y;
    }
  }
  block_partials_store<1>(acc, lds, partials_b);
  if ((VAR & 8) && blockIdx.x == 0 && threadIdx.x == 0) *st_out = cs;  // late, one thread
}

__global__ __launch_bounds__(kBlock) void reduce3(const double *__restrict__ partials, int count,
                                                  double *__restrict__ slots) {
  __shared__ double lds[3 * (kWaves + 1)];
  double d[3];
  reduce_rows<3>(partials, count, d, lds);
  if (threadIdx.x == 0) { slots[0] = d[0]; slots[1] = d[1]; slots[2] = d[2]; }
}

int main() {
  const size_t n = 3000000;
  double *Z, *p, *Hp, *s, *r, *pa, *pb, *slots;
  CK(hipMalloc(&Z, n * 8)); CK(hipMalloc(&p, n * 8)); CK(hipMalloc(&Hp, n * 8)); CK(hipMalloc(&s, n * 8));
  CK(hipMalloc(&r, n * 8)); CK(hipMalloc(&pa, kMaxComps * kMaxRows * 8)); CK(hipMalloc(&pb, kMaxComps * kMaxRows * 8));
  CK(hipMalloc(&slots, 64));
  std::vector<double> h(n, 1e-3);
  for (double *d : {Z, p, Hp, s, r}) CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(pa, 0, kMaxComps * kMaxRows * 8)); CK(hipMemset(slots, 0, 64));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = grid_for(n, 4), reps = 300;
  {
    CgState *stt;
    CK(hipMalloc(&stt, 64 * sizeof(CgState)));
    CgState hs{};
    hs.sk_M_pk = 0.1; hs.sk_M_2 = 0.2; hs.pk_M_2 = 0.3; hs.Delta = 1e3; hs.Delta_2 = 1e6; hs.rv = 1e-3;
    hs.epsilon = 1e-8; hs.max_iterations = 1000; hs.mode = CG_RUN;
    CK(hipMemcpy(stt, &hs, sizeof(hs), hipMemcpyHostToDevice));
    CK(hipMemcpy(stt + 32, &hs, sizeof(hs), hipMemcpyHostToDevice));
    for (int var : {0, 1, 3, 4, 5, 7, 8, 12}) {
      for (int it = 0; it < reps + 20; ++it) {
        if (it == 20) CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(producer, dim3(489), dim3(kBlock), 0, st, n, Z, p, Hp, pa);
#define C2(V) case V: hipLaunchKernelGGL(consumer2<V>, dim3(grid), dim3(kBlock), 0, st, n, (const CgState *)stt, stt + 32, pa, 489, p, Hp, s, r, pb); break;
        switch (var) { C2(0) C2(1) C2(3) C2(4) C2(5) C2(7) C2(8) C2(12) }
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("consumer2 var %2d (early-write %d, cooperative %d, real-step %d, late-write %d): %.2f us per pair\n", var, var & 1,
             (var >> 1) & 1, (var >> 2) & 1, (var >> 3) & 1, 1e3 * ms / reps);
    }
  }
  for (int gridp : {489, 512}) {
    for (int mode = 0; mode < 5; ++mode) {
      for (int it = 0; it < reps + 20; ++it) {
        if (it == 20) CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(producer, dim3(gridp), dim3(kBlock), 0, st, n, Z, p, Hp, pa);
        switch (mode) {
          case 0: hipLaunchKernelGGL(consumer<0>, dim3(grid), dim3(kBlock), 0, st, n, pa, gridp, slots, p, Hp, s, r, pb); break;
          case 1: hipLaunchKernelGGL(consumer<1>, dim3(grid), dim3(kBlock), 0, st, n, pa, gridp, slots, p, Hp, s, r, pb); break;
          case 2: hipLaunchKernelGGL(consumer<2>, dim3(grid), dim3(kBlock), 0, st, n, pa, gridp, slots, p, Hp, s, r, pb); break;
          case 3: hipLaunchKernelGGL(consumer<3>, dim3(grid), dim3(kBlock), 0, st, n, pa, gridp, slots, p, Hp, s, r, pb); break;
          case 4:
            hipLaunchKernelGGL(reduce3, dim3(1), dim3(kBlock), 0, st, pa, gridp, slots);
            hipLaunchKernelGGL(consumer<0>, dim3(grid), dim3(kBlock), 0, st, n, pa, gridp, slots, p, Hp, s, r, pb);
            break;
        }
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const char *names[] = {"slots(no reduce; not a valid pipeline)", "fused reduce_rows prologue", "fused + prefetch before prologue",
                             "fused, wave-0-only reduce", "separate 1-WG reduce kernel + slots"};
      printf("producer grid %d  mode %d %-42s : %.2f us per producer+consumer pair\n", gridp, mode, names[mode],
             1e3 * ms / reps);
    }
  }
  return 0;
}
