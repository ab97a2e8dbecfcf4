// Probe 3: sustained fp64 MFMA throughput (long kernels, random operands) with and without LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

// pure register MFMA: 16 accumulators, VGPR form (512-thread block)
__global__ void __launch_bounds__(512) mfma_reg(double* out, const double* in, int iters) {
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = in[threadIdx.x], b = in[threadIdx.x + 512];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// operands re-read from LDS every k-substep (1 A + 16 B fragments per 16 MFMAs), like the predict kernel
__global__ void __launch_bounds__(512) mfma_lds(double* out, const double* in, int iters) {
  __shared__ double As[2 * 16 * 80], Bs[2 * 16 * 272];
  for (int i = threadIdx.x; i < 2 * 16 * 80; i += 512) As[i] = in[i % 1024];
  for (int i = threadIdx.x; i < 2 * 16 * 272; i += 512) Bs[i] = in[(i * 7) % 1024];
  __syncthreads();
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3, fk = lane >> 4, fr = lane & 15;
  for (int it = 0; it < iters; ++it) {
    const double* pa = As + (it & 1) * 16 * 80 + w * 16 + fr;
    const double* pb = Bs + (it & 1) * 16 * 272 + fr;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      const double af = pa[(kk + fk) * 80];
#pragma unroll
      for (int h = 0; h < 16; h += 4) {
        double bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = pb[(kk + fk) * 272 + (h + j) * 16];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[h + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af, acc[h + j], 0, 0, 0);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the register-blocked pattern of rownorm2_kernel / rownormp_kernel<4,4>: 4 A + 4 B fragments per 16 MFMAs (round 4: what is the ceiling of THAT pattern?)
// BAR: an s_barrier per 64 MFMAs as in the real k-step
template <bool BAR>
__global__ void __launch_bounds__(512) mfma_lds44(double* out, const double* in, int iters) {
  __shared__ double As[2 * 16 * 144], Bs[2 * 16 * 272];
  for (int i = threadIdx.x; i < 2 * 16 * 144; i += 512) As[i] = in[i % 1024];
  for (int i = threadIdx.x; i < 2 * 16 * 272; i += 512) Bs[i] = in[(i * 7) % 1024];
  __syncthreads();
  d4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 2, wn = w & 3, fk = lane >> 4, fr = lane & 15;
  for (int it = 0; it < iters; ++it) {
    const double* pa = As + (it & 1) * 16 * 144 + wm * 64 + fr;
    const double* pb = Bs + (it & 1) * 16 * 272 + wn * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = pa[(kk + fk) * 144 + i * 16];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = pb[(kk + fk) * 272 + j * 64];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *out, *in; hipMalloc(&out, 256 * 512 * 8 * 2); hipMalloc(&in, 1024 * 8);
  double h[1024]; srand(1); for (int i = 0; i < 1024; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep)
  for (int iters : {2000, 20000, 100000}) {
    for (int which = 0; which < 2; ++which) {
      int per_iter = which == 0 ? 16 : 64;
      int it = which == 0 ? iters : iters / 4;
      hipEventRecord(e0);
      if (which == 0) mfma_reg<<<256, 512>>>(out, in, it); else mfma_lds<<<256, 512>>>(out, in, it);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 256.0 * 8 * (double)it * per_iter * 2048.0;
      printf("%s iters %6d: %8.3f ms  %.2f TFLOP/s\n", which == 0 ? "mfma_reg (2 waves/SIMD)" : "mfma_lds (2 waves/SIMD)", it, ms, fl / ms / 1e9);
    }
  }
  for (int which = 0; which < 2; ++which) {
    const int it = 25000;
    hipEventRecord(e0);
    if (which == 0) mfma_lds44<false><<<256, 512>>>(out, in, it); else mfma_lds44<true><<<256, 512>>>(out, in, it);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mfma_lds44 (4 A + 4 B fragments per 16 MFMAs, 2 waves/SIMD%s) iters %6d: %8.3f ms  %.2f TFLOP/s\n", which ? ", a barrier per 64 MFMAs" : "", it, ms, 256.0 * 8 * (double)it * 64 * 2048.0 / ms / 1e9);
  }
  return 0;
}
