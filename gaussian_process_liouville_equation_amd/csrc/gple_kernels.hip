// gple_kernels.hip — element-wise / reduction kernels of the GPR hot path (fp64 VALU work, gfx950).
//
// Reference loops restated here (paths relative to /root/reference/gaussian_process_liouville_equation/):
//   gaussian_kernel / delta_kernel / derivatives   kernel.cpp:8-215        -> se_exact(), gram_* kernels
//   label rescaling                                 kernel.cpp:279-280, complex_kernel.cpp:262-263 -> prep_labels
//   K^-1 y, diag(K^-1)                               kernel.cpp:282-283      -> trmv_lower + colpass on T = chol(K)^-1
//   LOOCV error, population, <r>, purity             kernel.cpp:285-335      -> real_fit_sums, quadform
//   cutoff_factor                                    kernel.h:301-332        -> cutoff_value()
#include "gple_kernels.h"

namespace gple
{
	namespace
	{
		// exp(-(((a0-b0)/l0)^2 + ((a1-b1)/l1)^2)/2) with exactly the reference's operation order and no FMA
		// contraction (kernel.cpp:46-47): subtract, divide, square, sum, negate, /2, exp.
		__device__ __forceinline__ double se_exact(double a0, double a1, double b0, double b1, double l0, double l1)
		{
			const double d0 = __ddiv_rn(__dsub_rn(a0, b0), l0);
			const double d1 = __ddiv_rn(__dsub_rn(a1, b1), l1);
			const double sum = __dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1));
			return exp(__ddiv_rn(-sum, 2.0));
		}

		__device__ __forceinline__ double wave_sum(double x)
		{
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
			return x;
		}
		__device__ __forceinline__ double wave_max(double x)
		{
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
			return x;
		}
		// deterministic block sum (fixed tree); result valid in thread 0
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

		// ------------------------------------------------------------------------------------------------------
		// One launch prepares every input of a fit: the zero-padded copy of the training points, the cleared scalar block
		// (s_out[0 .. nscal), info word included) and the rescaled, zero-padded labels with the rescale factor in s_out[0].
		__global__ void __launch_bounds__(1024) prep_labels_kernel(const double* __restrict__ y, int stride, int complex_abs, int N,
			int Np, double* __restrict__ ys, double* __restrict__ s_out, const double* __restrict__ X, double* __restrict__ Xt, int nscal)
		{
			__shared__ double red[16];
			__shared__ double s_sh;
			for (int i = threadIdx.x; i < 2 * Np; i += 1024) Xt[i] = i < 2 * N ? X[i] : 0.0;
			for (int i = threadIdx.x; i < nscal; i += 1024) s_out[i] = 0.0;
			double m = 0.0;
			for (int i = threadIdx.x; i < N; i += 1024)
			{
				const double a = complex_abs ? hypot(y[2 * i], y[2 * i + 1]) : fabs(y[static_cast<long>(i) * stride]);
				m = fmax(m, a);
			}
			m = wave_max(m);
			if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
			__syncthreads();
			if (threadIdx.x == 0)
			{
				double mm = 0.0;
				for (int i = 0; i < 16; ++i) mm = fmax(mm, red[i]);
				s_sh = 10.0 / mm; // RescaleMaximum / maxCoeff, kernel.h:37
				*s_out = s_sh;
			}
			__syncthreads();
			const double s = s_sh;
			for (int i = threadIdx.x; i < Np; i += 1024)
			{
				if (complex_abs)
				{
					ys[i] = i < N ? y[2 * i] * s : 0.0;
					ys[Np + i] = i < N ? y[2 * i + 1] * s : 0.0;
				}
				else
				{
					ys[i] = i < N ? y[static_cast<long>(i) * stride] * s : 0.0;
				}
			}
		}

		// typed training Gram, padded with the identity (the whole square for the getters, the lower block triangle for the factorisation)
		// blockIdx.x == n_total / 64 (launched only when ys != nullptr): the label row below the matrix, row n_total = ys, then zeros
		__global__ void __launch_bounds__(256) gram_train_kernel(const double* __restrict__ Xt, int N, int Np, int n_total, SEParamSet ps,
			double* __restrict__ K, long ld, const double* __restrict__ ys)
		{
			// ys != nullptr: the matrix is about to be factored — only the 64-blocks on and below the diagonal are read from then on
			if (ys != nullptr && blockIdx.y / 4 > blockIdx.x) return;
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			if (i >= n_total)
			{
#pragma unroll
				for (int e = 0; e < 4; ++e)
				{
					const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + e;
					K[i + static_cast<long>(j) * ld] = i == n_total ? ys[j] : 0.0;
				}
				return;
			}
			const int ti = i >= Np, pi = ti ? i - Np : i;
			const bool vi = pi < N;
			double xi0 = 0.0, xi1 = 0.0;
			if (vi) xi0 = Xt[2 * pi], xi1 = Xt[2 * pi + 1];
#pragma unroll
			for (int e = 0; e < 4; ++e)
			{
				const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + e;
				const int tj = j >= Np, pj = tj ? j - Np : j;
				double val;
				if (vi && pj < N)
				{
					const SEParam& p = ps.p[ti + tj];
					const double g = se_exact(xi0, xi1, Xt[2 * pj], Xt[2 * pj + 1], p.l0, p.l1);
					val = __dmul_rn(p.amp, __dadd_rn(g, __dmul_rn(p.n2, i == j ? 1.0 : 0.0))); // kernel.cpp:227
				}
				else
				{
					val = i == j ? 1.0 : 0.0;
				}
				K[i + static_cast<long>(j) * ld] = val;
			}
		}

		// partial sums of u = T ys over 256-wide k chunks
		__global__ void __launch_bounds__(256) trmv_partial_kernel(const double* __restrict__ T, long ldt, int n, const double* __restrict__ ys,
			double* __restrict__ part)
		{
			const int rb = blockIdx.x, kc = blockIdx.y;
			if (kc > rb) return;
			__shared__ double ysh[256];
			ysh[threadIdx.x] = ys[kc * 256 + threadIdx.x];
			__syncthreads();
			const int i = rb * 256 + threadIdx.x;
			const double* __restrict__ t = T + i + static_cast<long>(kc) * 256 * ldt;
			double acc = 0.0;
#pragma unroll 8
			for (int k = 0; k < 256; ++k) acc = fma(t[static_cast<long>(k) * ldt], ysh[k], acc);
			part[static_cast<long>(kc) * n + i] = acc;
		}
		__global__ void __launch_bounds__(256) trmv_reduce_kernel(const double* __restrict__ part, int n, double* __restrict__ u)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			double acc = 0.0;
			for (int kc = 0; kc <= i / 256; ++kc) acc += part[static_cast<long>(kc) * n + i];
			u[i] = acc;
		}

		// one wave per column of the lower-triangular T
		// info != nullptr and *info < 0 (a wave of the one-launch factorisation gave up waiting, gple_chol.hip): T is then unfinished — finite, but
		// wrong — and every product of this pass is replaced by NaN, so that whatever is enqueued behind the fit yields NaN, never a plausible number
		__global__ void __launch_bounds__(256) colpass_kernel(const double* __restrict__ T, long ldt, int n, const double* __restrict__ u,
			double* __restrict__ v, double* __restrict__ w, int shift, double* __restrict__ wx, const int* __restrict__ info)
		{
			const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
			const double* __restrict__ c = T + static_cast<long>(k) * ldt;
			const bool cross = wx != nullptr && k + shift < n;
			const double* __restrict__ c2 = T + static_cast<long>(cross ? k + shift : k) * ldt;
			double av = 0.0, aw = 0.0, ax = 0.0;
			const int i2 = (k + shift) & ~63; // column k + shift starts at its own diagonal block: the blocks above are never written
			for (int i = (k & ~63) + lane; i < n; i += 64)
			{
				const double tv = c[i];
				av = fma(tv, u[i], av);
				aw = fma(tv, tv, aw);
				if (cross && i >= i2) ax = fma(tv, c2[i], ax);
			}
			av = wave_sum(av), aw = wave_sum(aw), ax = wave_sum(ax);
			if (info != nullptr && *info < 0) av = aw = ax = __builtin_nan("");
			if (lane == 0)
			{
				v[k] = av;
				w[k] = aw;
				if (cross) wx[k] = ax;
			}
		}

		// qpart != nullptr: also the final sum of the purity quadratic form's per-block partials (quadform_kernel) -> qout
		__global__ void __launch_bounds__(1024) real_fit_sums_kernel(const double* __restrict__ Xt, const double* __restrict__ ys,
			const double* __restrict__ v, const double* __restrict__ w, int N, double* __restrict__ out, const double* __restrict__ qpart, int nq,
			double* __restrict__ qout)
		{
			__shared__ double red[16];
			double s[6] = {0, 0, 0, 0, 0, 0};
			for (int i = threadIdx.x; i < N; i += 1024)
			{
				const double vi = v[i], r = vi / w[i];
				s[0] += r * r;          // (InvLbl / Inverse.diagonal())^2, kernel.cpp:285
				s[1] += vi;             // v.sum(), :293
				s[2] += Xt[2 * i] * vi; // feature * v, :308
				s[3] += Xt[2 * i + 1] * vi;
				s[4] += ys[i] * vi; // Label.dot(InvLbl), kernel.h:169
			}
			if (qpart != nullptr)
				for (int i = threadIdx.x; i < nq; i += 1024) s[5] += qpart[i]; // same order of additions as sum_kernel
			for (int q = 0; q < 5; ++q)
			{
				const double tot = block_sum<1024>(s[q], red);
				if (threadIdx.x == 0) out[q] = tot;
			}
			if (qpart != nullptr)
			{
				const double tot = block_sum<1024>(s[5], red);
				if (threadIdx.x == 0) *qout = tot;
			}
		}

		// fdim < 0: sum a_i k_ij b_j ; fdim = d: sum a_i (k_ij ((x_i,d - x_j,d)/l_d)^2 / l_d) b_j  (kernel.cpp:99-160)
		__global__ void __launch_bounds__(256) quadform_kernel(const double* __restrict__ Xt, int N, SEParam p, const double* __restrict__ a,
			const double* __restrict__ b, int fdim, double* __restrict__ part)
		{
			__shared__ double red[4];
			__shared__ double xj[64 * 2], bj[64];
			// a == b: the form is symmetric — blocks above the diagonal are left to their mirror images, which count twice
			const bool sym = a == b;
			if (sym && blockIdx.y > blockIdx.x)
			{
				if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = 0.0;
				return;
			}
			const int j0 = blockIdx.y * 64;
			if (threadIdx.x < 64)
			{
				const int j = j0 + threadIdx.x;
				xj[2 * threadIdx.x] = j < N ? Xt[2 * j] : 0.0;
				xj[2 * threadIdx.x + 1] = j < N ? Xt[2 * j + 1] : 0.0;
				bj[threadIdx.x] = j < N ? b[j] : 0.0;
			}
			__syncthreads();
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			double acc = 0.0;
			if (i < N)
			{
				const double x0 = Xt[2 * i], x1 = Xt[2 * i + 1], ai = a[i];
				const int jb = (threadIdx.x >> 6) * 16;
				for (int e = 0; e < 16; ++e)
				{
					const int jl = jb + e;
					if (j0 + jl < N)
					{
						const double g = se_exact(x0, x1, xj[2 * jl], xj[2 * jl + 1], p.l0, p.l1);
						double kij = p.amp * g;
						if (fdim >= 0)
						{
							const double l = fdim == 0 ? p.l0 : p.l1;
							const double d = ((fdim == 0 ? x0 : x1) - xj[2 * jl + fdim]) / l;
							kij *= d * d / l;
						}
						acc += ai * kij * bj[jl];
					}
				}
			}
			const double tot = block_sum<256>(acc, red);
			if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = sym && blockIdx.y < blockIdx.x ? 2.0 * tot : tot;
		}
		__global__ void __launch_bounds__(1024) sum_kernel(const double* __restrict__ part, int n, double* __restrict__ out)
		{
			__shared__ double red[16];
			double s = 0.0;
			for (int i = threadIdx.x; i < n; i += 1024) s += part[i];
			const double tot = block_sum<1024>(s, red);
			if (threadIdx.x == 0) *out = tot;
		}

		// KernelBase: K and its 4 parameter derivatives (kernel.cpp:168-242)
		__global__ void __launch_bounds__(256) gram_rect_kernel(const double* __restrict__ L, int R, const double* __restrict__ Rt, int C, int same,
			SEParam p, double sf, double sn, double* __restrict__ K, double* __restrict__ dK)
		{
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			if (i >= R) return;
			const double a0 = L[2 * i], a1 = L[2 * i + 1];
			for (int e = 0; e < 4; ++e)
			{
				const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + e;
				if (j >= C) continue;
				const double b0 = Rt[2 * j], b1 = Rt[2 * j + 1];
				const double delta = same ? (i == j ? 1.0 : 0.0) : ((a0 == b0 && a1 == b1) ? 1.0 : 0.0);
				const double g = se_exact(a0, a1, b0, b1, p.l0, p.l1);
				const double k = __dmul_rn(p.amp, __dadd_rn(g, __dmul_rn(p.n2, delta)));
				const long idx = i + static_cast<long>(j) * R;
				K[idx] = k;
				if (dK != nullptr)
				{
					const long sz = static_cast<long>(R) * C;
					dK[idx] = __dmul_rn(k, 2.0 / sf); // :181
					const double noise = __dmul_rn(sf, sn);
					const double base = same ? __dsub_rn(k, __dmul_rn(__dmul_rn(noise, noise), delta)) : k; // :190
					const double d0 = __ddiv_rn(__dsub_rn(a0, b0), p.l0), d1 = __ddiv_rn(__dsub_rn(a1, b1), p.l1);
					const bool zero_diag = same && i == j;
					dK[sz + idx] = zero_diag ? 0.0 : __dmul_rn(base, __ddiv_rn(__dmul_rn(d0, d0), p.l0)); // :109,131
					dK[2 * sz + idx] = zero_diag ? 0.0 : __dmul_rn(base, __ddiv_rn(__dmul_rn(d1, d1), p.l1));
					dK[3 * sz + idx] = same ? __dmul_rn(__dmul_rn(__dmul_rn(2.0, p.amp), sn), delta) : 0.0; // :207
				}
			}
		}

		__global__ void __launch_bounds__(256) cutoff_kernel(const double* __restrict__ pred, int is_complex, const double* __restrict__ var, int M,
			double* __restrict__ factor)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i >= M) return;
			if (is_complex)
			{
				const double re = pred[2 * i], im = pred[2 * i + 1];
				factor[i] = cutoff_value(re * re + im * im, hypot(re, im), var[i]);
			}
			else
			{
				factor[i] = cutoff_value(pred[i] * pred[i], fabs(pred[i]), var[i]);
			}
		}

		// the fit's scalar block: [0] rescale factor, [31] info of the factorisation (an int in the double's slot; gple_capi.hip SDEV_INFO)
		// PredictiveKernel epilogue (kernel.cpp:496-522)
		__global__ void __launch_bounds__(256) predict_finish_real_kernel(const double* __restrict__ q, const double* __restrict__ mu, int M,
			double self, const double* __restrict__ s_dev, const double* __restrict__ labels, double* __restrict__ mean,
			double* __restrict__ var, double* __restrict__ cut, double* __restrict__ err_part)
		{
			__shared__ double red[4];
			const int i = blockIdx.x * 256 + threadIdx.x;
			const double s = *s_dev;
			const bool bad = fit_gave_up(s_dev); // the mean is NaN already (colpass_kernel); the variance comes from the unfinished T: NaN too
			double e = 0.0;
			if (i < M)
			{
				const double m = mu[i], vv = bad ? __builtin_nan("") : self - q[i];
				const double cf = cutoff_value(m * m, fabs(m), vv);
				if (mean) mean[i] = m;
				if (var) var[i] = vv;
				if (cut) cut[i] = m * cf / s;
				if (labels)
				{
					const double d = m - labels[i] * s;
					e = d * d;
				}
			}
			if (err_part != nullptr)
			{
				const double tot = block_sum<256>(e, red);
				if (threadIdx.x == 0) err_part[blockIdx.x] = tot;
			}
		}

		// PredictiveComplexKernel epilogue (complex_kernel.cpp:608-646) in the [Re; Im] basis
		__global__ void __launch_bounds__(256) predict_finish_complex_kernel(const double* __restrict__ q, const double* __restrict__ mu, int M,
			int m_split, double self, const double* __restrict__ s_dev, const double* __restrict__ labels, double* __restrict__ mean,
			double* __restrict__ var, double* __restrict__ cut, double* __restrict__ err_part)
		{
			__shared__ double red[4];
			const int i = blockIdx.x * 256 + threadIdx.x;
			const double s = *s_dev;
			const bool bad = fit_gave_up(s_dev);
			double e = 0.0;
			if (i < M)
			{
				const double re = mu[i], im = mu[m_split + i];
				const double vv = bad ? __builtin_nan("") : self - (q[i] + q[m_split + i]);
				const double cf = cutoff_value(re * re + im * im, hypot(re, im), vv);
				if (mean) mean[2 * i] = re, mean[2 * i + 1] = im;
				if (var) var[i] = vv;
				if (cut) cut[2 * i] = re * cf / s, cut[2 * i + 1] = im * cf / s;
				if (labels)
				{
					const double dr = re - labels[2 * i] * s, di = im - labels[2 * i + 1] * s;
					e = dr * dr + di * di;
				}
			}
			if (err_part != nullptr)
			{
				const double tot = block_sum<256>(e, red);
				if (threadIdx.x == 0) err_part[blockIdx.x] = tot;
			}
		}
	} // namespace

	hipError_t launch_prep_labels(hipStream_t s, const double* y, int stride, int complex_abs, int N, int Np, double* ys, double* s_out,
		const double* X, double* Xt, int nscal)
	{
		hipLaunchKernelGGL(prep_labels_kernel, dim3(1), dim3(1024), 0, s, y, stride, complex_abs, N, Np, ys, s_out, X, Xt, nscal);
		return hipGetLastError();
	}
	hipError_t launch_gram_train(hipStream_t s, const double* Xt, int N, int Np, int n_total, SEParamSet ps, double* K, long ld, const double* ys)
	{
		hipLaunchKernelGGL(gram_train_kernel, dim3(n_total / 64 + (ys ? 1 : 0), n_total / 16), dim3(256), 0, s, Xt, N, Np, n_total, ps, K, ld, ys);
		return hipGetLastError();
	}
	hipError_t launch_trmv_lower(hipStream_t s, const double* T, long ldt, int n, const double* ys, double* part, double* u)
	{
		hipLaunchKernelGGL(trmv_partial_kernel, dim3(n / 256, n / 256), dim3(256), 0, s, T, ldt, n, ys, part);
		hipLaunchKernelGGL(trmv_reduce_kernel, dim3(n / 256), dim3(256), 0, s, part, n, u);
		return hipGetLastError();
	}
	hipError_t launch_colpass(hipStream_t s, const double* T, long ldt, int n, const double* u, double* v, double* w, int shift, double* wx, const int* info)
	{
		hipLaunchKernelGGL(colpass_kernel, dim3(n / 4), dim3(256), 0, s, T, ldt, n, u, v, w, shift, wx, info);
		return hipGetLastError();
	}
	hipError_t launch_real_fit_sums(hipStream_t s, const double* Xt, const double* ys, const double* v, const double* w, int N, double* out,
		const double* qpart, int nq, double* qout)
	{
		hipLaunchKernelGGL(real_fit_sums_kernel, dim3(1), dim3(1024), 0, s, Xt, ys, v, w, N, out, qpart, nq, qout);
		return hipGetLastError();
	}
	hipError_t launch_quadform_partials(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, int fdim, double* part)
	{
		const int g = (N + 63) / 64;
		hipLaunchKernelGGL(quadform_kernel, dim3(g, g), dim3(256), 0, s, Xt, N, p, a, b, fdim, part);
		return hipGetLastError();
	}
	hipError_t launch_quadform(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, int fdim, double* part,
		double* out)
	{
		const int g = (N + 63) / 64;
		hipLaunchKernelGGL(quadform_kernel, dim3(g, g), dim3(256), 0, s, Xt, N, p, a, b, fdim, part);
		hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, part, g * g, out);
		return hipGetLastError();
	}
	hipError_t launch_gram_rect(hipStream_t s, const double* L, int R, const double* Rt, int C, int same, SEParam p, double sf, double sn,
		double* K, double* dK)
	{
		if (R == 0 || C == 0) return hipSuccess;
		hipLaunchKernelGGL(gram_rect_kernel, dim3((R + 63) / 64, (C + 15) / 16), dim3(256), 0, s, L, R, Rt, C, same, p, sf, sn, K, dK);
		return hipGetLastError();
	}
	hipError_t launch_cutoff(hipStream_t s, const double* pred, int is_complex, const double* var, int M, double* factor)
	{
		if (M == 0) return hipSuccess;
		hipLaunchKernelGGL(cutoff_kernel, dim3((M + 255) / 256), dim3(256), 0, s, pred, is_complex, var, M, factor);
		return hipGetLastError();
	}
	hipError_t launch_sum(hipStream_t s, const double* part, int n, double* out)
	{
		hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, part, n, out);
		return hipGetLastError();
	}
	hipError_t launch_predict_finish_real(hipStream_t s, const double* q, const double* mu, int M, double self, const double* s_dev,
		const double* labels, double* mean, double* var, double* cut, double* err_out)
	{
		if (M == 0) return hipSuccess;
		hipLaunchKernelGGL(predict_finish_real_kernel, dim3((M + 255) / 256), dim3(256), 0, s, q, mu, M, self, s_dev, labels, mean, var, cut,
			err_out);
		return hipGetLastError();
	}
	hipError_t launch_predict_finish_complex(hipStream_t s, const double* q, const double* mu, int M, int m_split, double self,
		const double* s_dev, const double* labels, double* mean, double* var, double* cut, double* err_out)
	{
		if (M == 0) return hipSuccess;
		hipLaunchKernelGGL(predict_finish_complex_kernel, dim3((M + 255) / 256), dim3(256), 0, s, q, mu, M, m_split, self, s_dev, labels, mean,
			var, cut, err_out);
		return hipGetLastError();
	}
} // namespace gple
