// gple_deriv.hip — kernels of the hyper-parameter derivative path (reference hot loop 5, kernel.cpp:337-400).
//
// The reference forms InverseDerivatives = -W dK W (two N^3 GEMMs per length parameter) but only ever consumes its
// diagonal and its product with the labels (kernel.cpp:365-400).  Here: dK/dl_d is materialised once (zero diagonal),
// diag(W dK W) comes from one MFMA GEMM  C = dK W  plus a column dot  sum_j W(j,i) C(j,i),  and (dW) y = -W (dK v) is
// two mat-vecs.
#include "gple_kernels.h"

namespace gple
{
	namespace
	{
		__device__ __forceinline__ double wave_sum(double x)
		{
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
			return x;
		}
		template <int NT>
		__device__ __forceinline__ double block_sum(double x, double* red)
		{
			x = wave_sum(x);
			const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
			__syncthreads();
			if (lane == 0) red[w] = x;
			__syncthreads();
			double tot = 0.0;
			if (threadIdx.x == 0)
				for (int i = 0; i < NT / 64; ++i) tot += red[i];
			return tot;
		}

		// bit-faithful restatement of gaussian_derivative_over_char_length on the training set (kernel.cpp:99-160, 185-198)
		__global__ void __launch_bounds__(256) deriv_gram_kernel(const double* __restrict__ Xt, int N, int n, SEParam p, double* __restrict__ D0,
			double* __restrict__ D1)
		{
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			const bool vi = i < N;
			double a0 = 0.0, a1 = 0.0;
			if (vi) a0 = Xt[2 * i], a1 = Xt[2 * i + 1];
#pragma unroll
			for (int e = 0; e < 4; ++e)
			{
				const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + e;
				double v0 = 0.0, v1 = 0.0;
				if (vi && j < N && i != j)
				{
					const double b0 = Xt[2 * j], b1 = Xt[2 * j + 1];
					const double d0 = __ddiv_rn(__dsub_rn(a0, b0), p.l0), d1 = __ddiv_rn(__dsub_rn(a1, b1), p.l1);
					const double sum = __dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1));
					const double k = __dmul_rn(p.amp, exp(__ddiv_rn(-sum, 2.0))); // K_ij off the diagonal (noise removed, :190)
					v0 = __dmul_rn(k, __ddiv_rn(__dmul_rn(d0, d0), p.l0));
					v1 = __dmul_rn(k, __ddiv_rn(__dmul_rn(d1, d1), p.l1));
				}
				D0[i + static_cast<long>(j) * n] = v0;
				D1[i + static_cast<long>(j) * n] = v1;
			}
		}

		__global__ void __launch_bounds__(256) gemv_partial_kernel(const double* __restrict__ A, long lda, int n, const double* __restrict__ x,
			double* __restrict__ part)
		{
			__shared__ double xs[256];
			const int kc = blockIdx.y;
			xs[threadIdx.x] = x[kc * 256 + threadIdx.x];
			__syncthreads();
			const int i = blockIdx.x * 256 + threadIdx.x;
			const double* __restrict__ a = A + i + static_cast<long>(kc) * 256 * lda;
			double acc = 0.0;
#pragma unroll 8
			for (int k = 0; k < 256; ++k) acc = fma(a[static_cast<long>(k) * lda], xs[k], acc);
			part[static_cast<long>(kc) * n + i] = acc;
		}
		// The same product for small matrices (n <= 1024): gemv_partial_kernel gives a 256-row block ONE workgroup per 256 columns, every thread a chain of
		// 256 dependent loads and FMAs — 13 us for the 256 x 256 product of the objective's gradient at the size the reference runs, on one CU of 256.
		// Here a workgroup takes 64 rows, its four waves a quarter of the 256-column chunk each (chains of 64), the quarters meet in LDS in a fixed order;
		// with one chunk (n = 256) the result is written at once and there is no second kernel.
		__global__ void __launch_bounds__(256) gemv_small_kernel(const double* __restrict__ A, long lda, int n, const double* __restrict__ x, double alpha,
			double* __restrict__ part, double* __restrict__ y)
		{
			__shared__ double xs[256];
			__shared__ double red[4][64];
			const int kc = blockIdx.y, r = threadIdx.x & 63, sl = threadIdx.x >> 6;
			xs[threadIdx.x] = x[kc * 256 + threadIdx.x];
			__syncthreads();
			const int i = blockIdx.x * 64 + r;
			const double* __restrict__ a = A + i + (static_cast<long>(kc) * 256 + sl * 64) * lda;
			double acc = 0.0;
#pragma unroll 16
			for (int k = 0; k < 64; ++k) acc = fma(a[static_cast<long>(k) * lda], xs[sl * 64 + k], acc);
			red[sl][r] = acc;
			__syncthreads();
			if (sl == 0)
			{
				const double v = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
				if (gridDim.y == 1) y[i] = alpha * v;
				else part[static_cast<long>(kc) * n + i] = v;
			}
		}
		// Up to three of these products / column dots / scalings in one launch (blockIdx.z picks the item): at the sizes the reference runs a derivative
		// fit is a string of 4-5 us launches, and the items of a batch do not depend on each other (launch_*_batch; the arithmetic per item is unchanged)
		struct Batch3
		{
			const double* A[3];
			const double* x[3]; // gemv: the vector; coldot: the second matrix; scale: unused
			double alpha[3];
			double* y[3];
		};
		__global__ void __launch_bounds__(256) gemv_small_batch_kernel(const Batch3 b, long lda, int n, double* __restrict__ part)
		{
			__shared__ double xs[256];
			__shared__ double red[4][64];
			const int z = blockIdx.z, kc = blockIdx.y, r = threadIdx.x & 63, sl = threadIdx.x >> 6;
			xs[threadIdx.x] = b.x[z][kc * 256 + threadIdx.x];
			__syncthreads();
			const int i = blockIdx.x * 64 + r;
			const double* __restrict__ a = b.A[z] + i + (static_cast<long>(kc) * 256 + sl * 64) * lda;
			double acc = 0.0;
#pragma unroll 16
			for (int k = 0; k < 64; ++k) acc = fma(a[static_cast<long>(k) * lda], xs[sl * 64 + k], acc);
			red[sl][r] = acc;
			__syncthreads();
			if (sl == 0)
			{
				const double v = (red[0][r] + red[1][r]) + (red[2][r] + red[3][r]);
				if (gridDim.y == 1) b.y[z][i] = b.alpha[z] * v;
				else part[(static_cast<long>(z) * gridDim.y + kc) * n + i] = v;
			}
		}
		__global__ void __launch_bounds__(256) gemv_reduce_batch_kernel(const Batch3 b, const double* __restrict__ part, int n)
		{
			const int z = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x, nc = n / 256;
			double acc = 0.0;
			for (int kc = 0; kc < nc; ++kc) acc += part[(static_cast<long>(z) * nc + kc) * n + i];
			b.y[z][i] = b.alpha[z] * acc;
		}
		__global__ void __launch_bounds__(256) coldot_batch_kernel(const Batch3 b, long lda, long ldb, int n, int shift)
		{
			const int z = blockIdx.z, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
			if (i + shift >= n) return;
			const double* __restrict__ a = b.A[z] + static_cast<long>(i) * lda;
			const double* __restrict__ c = b.x[z] + static_cast<long>(i + shift) * ldb;
			double acc = 0.0;
			for (int j = lane; j < n; j += 64) acc = fma(a[j], c[j], acc);
			acc = wave_sum(acc);
			if (lane == 0) b.y[z][i] = b.alpha[z] * acc;
		}
		__global__ void __launch_bounds__(256) scale_batch_kernel(const Batch3 b, int n0, int n1, int n2)
		{
			const int z = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x, n = z == 0 ? n0 : (z == 1 ? n1 : n2);
			if (i < n) b.y[z][i] = b.alpha[z] * b.A[z][i];
		}
		__global__ void __launch_bounds__(256) gemv_reduce_kernel(const double* __restrict__ part, int n, double alpha, double* __restrict__ y)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			double acc = 0.0;
			for (int kc = 0; kc < n / 256; ++kc) acc += part[static_cast<long>(kc) * n + i];
			y[i] = alpha * acc;
		}
		__global__ void __launch_bounds__(256) coldot_kernel(const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, int n,
			int shift, double alpha, double* __restrict__ out)
		{
			const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
			if (i + shift >= n) return;
			const double* __restrict__ a = A + static_cast<long>(i) * lda;
			const double* __restrict__ b = B + static_cast<long>(i + shift) * ldb;
			double acc = 0.0;
			for (int j = lane; j < n; j += 64) acc = fma(a[j], b[j], acc);
			acc = wave_sum(acc);
			if (lane == 0) out[i] = alpha * acc;
		}
		__global__ void __launch_bounds__(256) scale_kernel(const double* __restrict__ x, double alpha, int n, double* __restrict__ y)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i < n) y[i] = alpha * x[i];
		}

		__global__ void __launch_bounds__(1024) real_deriv_sums_kernel(const double* __restrict__ v, const double* __restrict__ w,
			const double* __restrict__ dv, const double* __restrict__ dwd, int N, int ld, double* __restrict__ out)
		{
			__shared__ double red[16];
			double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (int i = threadIdx.x; i < N; i += 1024)
			{
				const double invd = w[i], diff = v[i] / invd;
#pragma unroll
				for (int ip = 0; ip < 4; ++ip)
				{
					const double dvi = dv[ip * ld + i];
					s[ip] += diff / invd * (dvi - diff * dwd[ip * ld + i]); // kernel.cpp:394
					s[4 + ip] += dvi;
				}
			}
			for (int q = 0; q < 8; ++q)
			{
				const double tot = block_sum<1024>(s[q], red);
				if (threadIdx.x == 0) out[q] = q < 4 ? 2.0 * tot : tot;
			}
		}

		// result[ip] = 2 * PredictionDifference . (dK*_ip v + K* dv_ip), PredictionDifference = Cutoff * s - s t  (kernel.cpp:527-536)
		__global__ void __launch_bounds__(256) predict_deriv_finish_real_kernel(const double* __restrict__ acc, int m_rows, const double* __restrict__ q,
			int M, double self, double sf, const double* __restrict__ s_dev, const double* __restrict__ labels, double* __restrict__ part)
		{
			__shared__ double red[4];
			const int i = blockIdx.x * 256 + threadIdx.x;
			const double s = *s_dev;
			double t[4] = {0, 0, 0, 0};
			if (i < M)
			{
				const double mu = acc[i], var = self - q[i];
				const double cf = cutoff_value(mu * mu, fabs(mu), var);
				const double diff = (mu * cf / s) * s - labels[i] * s;
				t[0] = diff * ((2.0 / sf) * mu + acc[1L * m_rows + i]);
				t[1] = diff * (acc[5L * m_rows + i] + acc[2L * m_rows + i]);
				t[2] = diff * (acc[6L * m_rows + i] + acc[3L * m_rows + i]);
				t[3] = diff * acc[4L * m_rows + i];
			}
			for (int ip = 0; ip < 4; ++ip)
			{
				const double tot = block_sum<256>(t[ip], red);
				if (threadIdx.x == 0) part[ip * gridDim.x + blockIdx.x] = 2.0 * tot;
			}
		}
		__global__ void __launch_bounds__(256) sum4_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out)
		{
			__shared__ double red[4];
			for (int ip = 0; ip < 4; ++ip)
			{
				double s = 0.0;
				for (int i = threadIdx.x; i < nblk; i += 256) s += part[ip * nblk + i];
				const double tot = block_sum<256>(s, red);
				if (threadIdx.x == 0) out[ip] = tot;
			}
		}
	} // namespace

	hipError_t launch_deriv_gram(hipStream_t s, const double* Xt, int N, int n, SEParam p, double* D0, double* D1)
	{
		hipLaunchKernelGGL(deriv_gram_kernel, dim3(n / 64, n / 16), dim3(256), 0, s, Xt, N, n, p, D0, D1);
		return hipGetLastError();
	}
	hipError_t launch_gemv(hipStream_t s, const double* A, long lda, int n, const double* x, double alpha, double* part, double* y)
	{
		if (n <= 1024) // few workgroups of long chains otherwise (gemv_small_kernel)
		{
			hipLaunchKernelGGL(gemv_small_kernel, dim3(n / 64, n / 256), dim3(256), 0, s, A, lda, n, x, alpha, part, y);
			if (n > 256) hipLaunchKernelGGL(gemv_reduce_kernel, dim3(n / 256), dim3(256), 0, s, part, n, alpha, y);
			return hipGetLastError();
		}
		hipLaunchKernelGGL(gemv_partial_kernel, dim3(n / 256, n / 256), dim3(256), 0, s, A, lda, n, x, part);
		hipLaunchKernelGGL(gemv_reduce_kernel, dim3(n / 256), dim3(256), 0, s, part, n, alpha, y);
		return hipGetLastError();
	}
	// cnt <= 3 independent items in one launch; n <= 1024 (the small product kernel); part: cnt * (n / 256) * n doubles
	hipError_t launch_gemv_batch(hipStream_t s, int n, int cnt, const double* const* A, long lda, const double* const* x, const double* alpha, double* part,
		double* const* y)
	{
		if (cnt < 1 || cnt > 3 || n > 1024 || n % 256) return hipErrorInvalidValue;
		Batch3 b{};
		for (int z = 0; z < cnt; ++z) b.A[z] = A[z], b.x[z] = x[z], b.alpha[z] = alpha[z], b.y[z] = y[z];
		hipLaunchKernelGGL(gemv_small_batch_kernel, dim3(n / 64, n / 256, cnt), dim3(256), 0, s, b, lda, n, part);
		if (n > 256) hipLaunchKernelGGL(gemv_reduce_batch_kernel, dim3(n / 256, 1, cnt), dim3(256), 0, s, b, part, n);
		return hipGetLastError();
	}
	hipError_t launch_coldot_batch(hipStream_t s, int n, int cnt, const double* const* A, long lda, const double* const* B, long ldb, int shift, const double* alpha,
		double* const* out)
	{
		if (cnt < 1 || cnt > 3) return hipErrorInvalidValue;
		Batch3 b{};
		for (int z = 0; z < cnt; ++z) b.A[z] = A[z], b.x[z] = B[z], b.alpha[z] = alpha[z], b.y[z] = out[z];
		hipLaunchKernelGGL(coldot_batch_kernel, dim3(n / 4, 1, cnt), dim3(256), 0, s, b, lda, ldb, n, shift);
		return hipGetLastError();
	}
	hipError_t launch_scale_batch(hipStream_t s, int cnt, const double* const* x, const double* alpha, const int* n, double* const* y)
	{
		if (cnt < 1 || cnt > 3) return hipErrorInvalidValue;
		Batch3 b{};
		int nn[3] = {0, 0, 0}, nmax = 0;
		for (int z = 0; z < cnt; ++z) b.A[z] = x[z], b.alpha[z] = alpha[z], b.y[z] = y[z], nn[z] = n[z], nmax = n[z] > nmax ? n[z] : nmax;
		hipLaunchKernelGGL(scale_batch_kernel, dim3((nmax + 255) / 256, 1, cnt), dim3(256), 0, s, b, nn[0], nn[1], nn[2]);
		return hipGetLastError();
	}
	hipError_t launch_coldot(hipStream_t s, const double* A, long lda, const double* B, long ldb, int n, int shift, double alpha, double* out)
	{
		hipLaunchKernelGGL(coldot_kernel, dim3(n / 4), dim3(256), 0, s, A, lda, B, ldb, n, shift, alpha, out);
		return hipGetLastError();
	}
	hipError_t launch_scale(hipStream_t s, const double* x, double alpha, int n, double* y)
	{
		hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, alpha, n, y);
		return hipGetLastError();
	}
	hipError_t launch_real_deriv_sums(hipStream_t s, const double* v, const double* w, const double* dv, const double* dwd, int N, int ld,
		double* out)
	{
		hipLaunchKernelGGL(real_deriv_sums_kernel, dim3(1), dim3(1024), 0, s, v, w, dv, dwd, N, ld, out);
		return hipGetLastError();
	}
	hipError_t launch_predict_deriv_finish_real(hipStream_t s, const double* acc, int m_rows, const double* q, int M, double self, double sf,
		const double* s_dev, const double* labels, double* part, double* out4)
	{
		const int nblk = (M + 255) / 256;
		hipLaunchKernelGGL(predict_deriv_finish_real_kernel, dim3(nblk), dim3(256), 0, s, acc, m_rows, q, M, self, sf, s_dev, labels, part);
		hipLaunchKernelGGL(sum4_kernel, dim3(1), dim3(256), 0, s, part, nblk, out4);
		return hipGetLastError();
	}
} // namespace gple
