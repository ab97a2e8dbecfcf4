// Probe 4: one MFMA wave + one fp64-VALU wave per SIMD (512-thread block): how much VALU work rides along for free?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NV, bool BARRIER, bool SWAP>
__global__ void __launch_bounds__(512) pingpong(double* out, const double* in, int iters) {
  __shared__ double As[2 * 16 * 80], Bs[2 * 16 * 272];
  for (int i = threadIdx.x; i < 2 * 16 * 80; i += 512) As[i] = in[i % 1024];
  for (int i = threadIdx.x; i < 2 * 16 * 272; i += 512) Bs[i] = in[(i * 7) % 1024];
  __syncthreads();
  const int grp = threadIdx.x >> 8;
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3, fk = lane >> 4, fr = lane & 15;
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  double f[8]; for (int i = 0; i < 8; ++i) f[i] = in[threadIdx.x + i];
  const double m = 1.0000001, c = 1e-9;
  auto mfma_phase = [&](int it) {
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
  };
  auto valu_phase = [&]() {
#pragma unroll
    for (int q = 0; q < NV / 8; ++q)
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = fma(f[i], m, c);
  };
  if (SWAP) {
    // roles alternate every half-step (like the predict kernel)
    if (grp == 0) for (int it = 0; it < iters; ++it) { mfma_phase(it); if (BARRIER) __syncthreads(); valu_phase(); if (BARRIER) __syncthreads(); }
    else          for (int it = 0; it < iters; ++it) { valu_phase(); if (BARRIER) __syncthreads(); mfma_phase(it); if (BARRIER) __syncthreads(); }
  } else {
    // fixed roles: group 0 only MFMA (2 phases per iteration), group 1 only VALU
    if (grp == 0) for (int it = 0; it < iters; ++it) { mfma_phase(it); if (BARRIER) __syncthreads(); mfma_phase(it + 1); if (BARRIER) __syncthreads(); }
    else          for (int it = 0; it < iters; ++it) { valu_phase(); if (BARRIER) __syncthreads(); valu_phase(); if (BARRIER) __syncthreads(); }
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, bool BARRIER, bool SWAP> void run(double* out, const double* in, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  pingpong<NV, BARRIER, SWAP><<<256, 512>>>(out, in, 200);
  hipEventRecord(e0);
  pingpong<NV, BARRIER, SWAP><<<256, 512>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // MFMA flops: per iteration both groups together execute 2 mfma phases of 64 MFMAs per wave on 4 waves
  double fl = 256.0 * 4 * (double)iters * 2 * 64 * 2048.0;
  double cyc = ms * 1e-3 * 2.4e9 / iters / 2;  // cycles per half-step
  printf("%-28s NV=%4d: %8.3f ms  %.2f TFLOP/s (MFMA)  %.0f cyc/half-step (ideal 4096)\n", name, NV, ms, fl / ms / 1e9, cyc);
}
int main() {
  double *out, *in; hipMalloc(&out, 256 * 512 * 8 * 2); hipMalloc(&in, 2048 * 8);
  double h[2048]; srand(1); for (int i = 0; i < 2048; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, true, true>(out, in, "swap+barrier");
  run<64, true, true>(out, in, "swap+barrier");
  run<128, true, true>(out, in, "swap+barrier");
  run<256, true, true>(out, in, "swap+barrier");
  run<512, true, true>(out, in, "swap+barrier");
  run<0, false, true>(out, in, "swap, no barrier");
  run<128, false, true>(out, in, "swap, no barrier");
  run<256, false, true>(out, in, "swap, no barrier");
  run<0, true, false>(out, in, "fixed roles+barrier");
  run<128, true, false>(out, in, "fixed roles+barrier");
  run<256, true, false>(out, in, "fixed roles+barrier");
  return 0;
}
