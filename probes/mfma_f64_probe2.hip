// Probe 2: can fp64 MFMA and fp64 VALU FMA overlap on gfx950?  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void __launch_bounds__(256) mfma444(double* out, int iters) {
  double acc[8]; for (int i = 0; i < 8; ++i) acc[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave interleaves NM mfma(16x16x4) with NF v_fma_f64 per iteration
template <int NM, int NF>
__global__ void __launch_bounds__(256) mixed(double* out, int iters) {
  d4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
  double f[32]; for (int i = 0; i < 32; ++i) f[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3, m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NM; ++r) {
      acc[r % 8] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r % 8], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NF / NM; ++q) { const int idx = (r * (NF / NM) + q) % 32; f[idx] = fma(f[idx], m, c); }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 32; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512-thread WG: waves 0-3 MFMA only, waves 4-7 FMA only (two waves per SIMD with different roles)
__global__ void __launch_bounds__(512) split(double* out, int it_m, int it_f) {
  const int w = threadIdx.x >> 6;
  double s = 0;
  if (w < 4) {
    d4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < it_m; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double f[16]; for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 1e-3 + i;
    double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < it_f; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = fma(f[i], m, c);
    }
    for (int i = 0; i < 16; ++i) s += f[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float time_ms(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  int ncu = 256; double* out; CK(hipMalloc(&out, (size_t)ncu * 16 * 512 * 8));
  int iters = 2000;
  {
    float ms = time_ms([&] { mfma444<<<ncu, 256>>>(out, iters); }, 5);
    double fl = (double)ncu * 4 * iters * 8 * 512.0;
    printf("mfma_f64 4x4x4(4b) 1 wave/SIMD: %.3f ms %.2f TFLOP/s (%.1f cyc/inst @2.4GHz)\n", ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 8.0));
  }
#define MIX(NM, NF) { float ms = time_ms([&] { mixed<NM, NF><<<ncu, 256>>>(out, iters); }, 5); \
    double fm = (double)ncu * 4 * iters * NM * 2048.0, ff = (double)ncu * 256 * iters * NF * 2.0; \
    printf("mixed 1 wave/SIMD  %d mfma + %3d fma / iter: %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", NM, NF, ms, fm / ms / 1e9, ff / ms / 1e9, (fm + ff) / ms / 1e9); }
  MIX(8, 0) MIX(8, 64) MIX(8, 128) MIX(8, 192) MIX(8, 256) MIX(8, 384)
#define MIX2(NM, NF) { float ms = time_ms([&] { mixed<NM, NF><<<ncu * 2, 256>>>(out, iters); }, 5); \
    double fm = (double)ncu * 2 * 4 * iters * NM * 2048.0, ff = (double)ncu * 2 * 256 * iters * NF * 2.0; \
    printf("mixed 2 waves/SIMD %d mfma + %3d fma / iter: %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", NM, NF, ms, fm / ms / 1e9, ff / ms / 1e9, (fm + ff) / ms / 1e9); }
  MIX2(8, 128) MIX2(8, 256) MIX2(8, 384)
  for (int itf : {0, 1000, 2000, 4000, 6000}) {
    int itm = 1000;
    float ms = time_ms([&] { split<<<ncu, 512>>>(out, itm, itf); }, 5);
    double fm = (double)ncu * 4 * itm * 8 * 2048.0, ff = (double)ncu * 256 * (double)itf * 16 * 2.0;
    printf("split roles: mfma iters %d, fma iters %d: %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", itm, itf, ms, fm / ms / 1e9, ff / ms / 1e9, (fm + ff) / ms / 1e9);
  }
  {
    float ms = time_ms([&] { split<<<ncu, 512>>>(out, 0, 4000); }, 5);
    double ff = (double)ncu * 256 * 4000.0 * 16 * 2.0;
    printf("split roles: fma only (1 wave/SIMD, 16 chains): %.3f ms valu %.2f TFLOP/s\n", ms, ff / ms / 1e9);
  }
  return 0;
}
