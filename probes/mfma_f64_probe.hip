// Probe: fp64 MFMA (v_mfma_f64_16x16x4_f64) fragment layout and issue rate on gfx950,
// plus fp64 exp() and v_fma_f64 throughput. Diagnostic only (not part of the product path).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void layout_kernel(const double* A, const double* B, double* Draw) {
  // A: 16x4 row-major, B: 4x16 row-major. lane l: a = A[l&15][l>>4], b = B[l>>4][l&15]
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Draw[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void __launch_bounds__(256) rate_kernel(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) exp_kernel(double* out, int iters) {
  double x0 = -1e-3 * threadIdx.x, x1 = x0 - 0.5, x2 = x0 - 1.5, x3 = x0 - 2.5;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int it = 0; it < iters; ++it) {
    s0 += exp(x0); s1 += exp(x1); s2 += exp(x2); s3 += exp(x3);
    x0 -= 1e-4; x1 -= 1e-4; x2 -= 1e-4; x3 -= 1e-4;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3;
}

__global__ void __launch_bounds__(256) fma_kernel(double* out, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = fma(a[i], m, c);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // ---- layout
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;     // asymmetric
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 5 + j * 11 + (j * j) % 7;
  double *dA, *dB, *dD;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  double C[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; C[i][j] = s; }
  int okA = 0, okB = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    double v = D[l * 4 + r];
    if (v == C[(l >> 4) + 4 * r][l & 15]) okA++;        // guide's f64 map: row=(lane>>4)+4*reg
    if (v == C[(l >> 4) * 4 + r][l & 15]) okB++;        // f32-style map
  }
  printf("layout: map row=(lane>>4)+4*reg matches %d/256 ; map row=(lane>>4)*4+reg matches %d/256\n", okA, okB);
  // ---- rates
  int nblk = p.multiProcessorCount * 1;  // one WG (4 waves) per CU => 1 wave per SIMD
  double* out; CK(hipMalloc(&out, (size_t)nblk * 8 * 256 * 8));
  int iters = 2000;
  {
    float ms = time_ms([&] { rate_kernel<4><<<nblk, 256>>>(out, iters); }, 5);
    double fl = (double)nblk * 4 * iters * 4 * 2048.0;
    printf("mfma_f64 16x16x4, 4 acc, 1 wave/SIMD: %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA @2.4GHz)\n", ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 4.0));
  }
  {
    float ms = time_ms([&] { rate_kernel<16><<<nblk, 256>>>(out, iters); }, 5);
    double fl = (double)nblk * 4 * iters * 16 * 2048.0;
    printf("mfma_f64 16x16x4, 16 acc, 1 wave/SIMD: %.3f ms  %.2f TFLOP/s  (%.1f cyc/MFMA @2.4GHz)\n", ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 16.0));
  }
  {
    float ms = time_ms([&] { rate_kernel<16><<<nblk * 2, 256>>>(out, iters); }, 5);
    double fl = (double)nblk * 2 * 4 * iters * 16 * 2048.0;
    printf("mfma_f64 16x16x4, 16 acc, 2 waves/SIMD: %.3f ms  %.2f TFLOP/s\n", ms, fl / ms / 1e9);
  }
  {
    int it2 = 500;
    float ms = time_ms([&] { exp_kernel<<<nblk * 8, 256>>>(out, it2); }, 5);
    double n = (double)nblk * 8 * 256 * it2 * 4;
    printf("fp64 exp(): %.3f ms  %.3f Texp/s  (%.1f cyc per wave-exp per SIMD @2.4GHz)\n", ms, n / ms / 1e9, ms * 1e-3 * 2.4e9 / (it2 * 4.0 * 8 /*waves per SIMD*/));
  }
  {
    int it2 = 4000;
    float ms = time_ms([&] { fma_kernel<<<nblk * 8, 256>>>(out, it2); }, 5);
    double fl = (double)nblk * 8 * 256 * it2 * 8 * 2.0;
    printf("v_fma_f64: %.3f ms  %.2f TFLOP/s\n", ms, fl / ms / 1e9);
  }
  return 0;
}
