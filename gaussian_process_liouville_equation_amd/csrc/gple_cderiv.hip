// gple_cderiv.hip — derivative members of the complex kernel (complex_kernel.cpp:20-132, 379-590, 648-668) in the real
// [Re; Im] embedding.
//
// The reference differentiates K (real) and K~ (complex) per parameter ("dK", "dK~", with its quirks: no s^2 factor on the
// sub-kernel and noise derivatives) and pushes them through P, Q.  With C the real covariance of [Re; Im] and M = C^-1:
//     dC_xx = (dK + Re dK~)/2,  dC_yy = (dK - Re dK~)/2,  dC_xy = Im dK~ / 2,     dM = -M dC M,
//     dP_ii = (dM_xx + dM_yy)_ii / 4,  dQ_ii = ((dM_xx - dM_yy)_ii - 2i dM_xy,ii) / 4,  dv = (dw_x + i dw_y)/2, dw = dM ys.
// Every block of every dC is  amp G(l) (c0 + c1 f_d)  with one squared-exponential G — a DSpec.
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
		// amp G (c0 + c1 f_dim) with the reference's Gram arithmetic (kernel.cpp:46-47, 106-110)
		__device__ __forceinline__ double dspec_value(const DSpec& sp, double a0, double a1, double b0, double b1)
		{
			const double d0 = __ddiv_rn(__dsub_rn(a0, b0), sp.l0), d1 = __ddiv_rn(__dsub_rn(a1, b1), sp.l1);
			const double g = exp(__ddiv_rn(-__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), 2.0));
			const double f = sp.dim == 0 ? d0 * d0 / sp.l0 : d1 * d1 / sp.l1;
			return sp.amp * g * (sp.c0 + sp.c1 * f);
		}

		__global__ void __launch_bounds__(256) typed_deriv_gram_kernel(const double* __restrict__ Xt, int N, int Np, int n, DSpecSet spec,
			double* __restrict__ D)
		{
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			const int ti = i >= Np, pi = ti ? i - Np : i;
			const bool vi = pi < N;
			double a0 = 0.0, a1 = 0.0;
			if (vi) a0 = Xt[2 * pi], a1 = Xt[2 * pi + 1];
#pragma unroll
			for (int e = 0; e < 4; ++e)
			{
				const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + e;
				const int tj = j >= Np, pj = tj ? j - Np : j;
				double val = 0.0;
				const DSpec& sp = spec.b[ti + tj];
				if (vi && pj < N && sp.active) val = dspec_value(sp, a0, a1, Xt[2 * pj], Xt[2 * pj + 1]);
				D[i + static_cast<long>(j) * n] = val;
			}
		}

		// complex_kernel.cpp:444-474
		__global__ void __launch_bounds__(1024) complex_deriv_sums_kernel(const double* __restrict__ w, const double* __restrict__ wd,
			const double* __restrict__ wx, const double* __restrict__ dw, const double* __restrict__ dwd, const double* __restrict__ dwx, int N, int Np,
			double* __restrict__ out8)
		{
			__shared__ double red[16];
			const int n = 2 * Np;
			double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			for (int i = threadIdx.x; i < N; i += 1024)
			{
				const double mxx = wd[i], myy = wd[Np + i], mxy = wx[i];
				const double P = 0.25 * (mxx + myy), qr = 0.25 * (mxx - myy), qi = -0.5 * mxy; // P_ii (real), Q_ii
				const double vr = 0.5 * w[i], vi = 0.5 * w[Np + i];
				const double sqd = P * P - (qr * qr + qi * qi); // square_diff
				// diff = (P v - conj(Q v)) / square_diff
				const double qvr = qr * vr - qi * vi, qvi = qr * vi + qi * vr;
				const double dfr = (P * vr - qvr) / sqd, dfi = (P * vi + qvi) / sqd;
				const double dabs2 = dfr * dfr + dfi * dfi;
#pragma unroll
				for (int ip = 0; ip < 8; ++ip)
				{
					const double dmxx = dwd[ip * n + i], dmyy = dwd[ip * n + Np + i], dmxy = dwx[ip * Np + i];
					const double dP = 0.25 * (dmxx + dmyy), dqr = 0.25 * (dmxx - dmyy), dqi = -0.5 * dmxy;
					const double dvr = 0.5 * dw[ip * n + i], dvi = 0.5 * dw[ip * n + Np + i];
					// A = dP v + P dv  (complex), B = dQ v + Q dv (complex); numerator_deriv = conj(diff) (A - conj(B))
					const double Ar = dP * vr + P * dvr, Ai = dP * vi + P * dvi;
					const double Br = (dqr * vr - dqi * vi) + (qr * dvr - qi * dvi), Bi = (dqr * vi + dqi * vr) + (qr * dvi + qi * dvr);
					const double Er = Ar - Br, Ei = Ai + Bi;
					const double num_r = dfr * Er + dfi * Ei; // Re(conj(diff) * E)
					// denominator_deriv = -2 |diff|^2 (P dP - Re(conj(Q) dQ))   (real)
					const double den = -2.0 * dabs2 * (P * dP - (qr * dqr + qi * dqi));
					s[ip] += (num_r + den) / sqd;
				}
			}
			for (int ip = 0; ip < 8; ++ip)
			{
				const double tot = block_sum<1024>(s[ip], red);
				if (threadIdx.x == 0) out8[ip] = 2.0 * tot;
			}
		}

		__global__ void __launch_bounds__(256) multi_quadform_kernel(const double* __restrict__ Xt, int N, SEParam p, const double* __restrict__ a,
			const double* __restrict__ b, double* __restrict__ part)
		{
			__shared__ double red[4];
			__shared__ double xj[64 * 2], aj[64], bj[64];
			const int j0 = blockIdx.y * 64;
			if (threadIdx.x < 64)
			{
				const int j = j0 + threadIdx.x;
				xj[2 * threadIdx.x] = j < N ? Xt[2 * j] : 0.0;
				xj[2 * threadIdx.x + 1] = j < N ? Xt[2 * j + 1] : 0.0;
				aj[threadIdx.x] = j < N ? a[j] : 0.0;
				bj[threadIdx.x] = j < N ? b[j] : 0.0;
			}
			__syncthreads();
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
			if (i < N)
			{
				const double x0 = Xt[2 * i], x1 = Xt[2 * i + 1], ai = a[i], bi = b[i];
				const int jb = (threadIdx.x >> 6) * 16;
				for (int e = 0; e < 16; ++e)
				{
					const int jl = jb + e;
					if (j0 + jl < N)
					{
						const double d0 = __ddiv_rn(__dsub_rn(x0, xj[2 * jl]), p.l0), d1 = __ddiv_rn(__dsub_rn(x1, xj[2 * jl + 1]), p.l1);
						const double k = p.amp * exp(__ddiv_rn(-__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), 2.0));
						const double f[3] = {1.0, d0 * d0 / p.l0, d1 * d1 / p.l1};
						const double pr[3] = {ai * aj[jl], bi * bj[jl], ai * bj[jl]};
#pragma unroll
						for (int q = 0; q < 3; ++q)
#pragma unroll
							for (int v = 0; v < 3; ++v) acc[3 * q + v] += pr[q] * k * f[v];
					}
				}
			}
			const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
			for (int q = 0; q < 9; ++q)
			{
				const double tot = block_sum<256>(acc[q], red);
				if (threadIdx.x == 0) part[q * nblk + blk] = tot;
			}
		}
		__global__ void __launch_bounds__(256) sum_planes_kernel(const double* __restrict__ part, int nblk, int planes, double* __restrict__ out)
		{
			__shared__ double red[4];
			for (int q = 0; q < planes; ++q)
			{
				double s = 0.0;
				for (int i = threadIdx.x; i < nblk; i += 256) s += part[q * nblk + i];
				const double tot = block_sum<256>(s, red);
				if (threadIdx.x == 0) out[q] = tot;
			}
		}

		__global__ void __launch_bounds__(256) aux_matvec_partial_kernel(const double* __restrict__ Xt, int N, SEParam p, const double* __restrict__ a,
			const double* __restrict__ b, double* __restrict__ part)
		{
			__shared__ double xj[256 * 2], aj[256], bj[256];
			const int j0 = blockIdx.y * 256;
			{
				const int j = j0 + threadIdx.x;
				xj[2 * threadIdx.x] = j < N ? Xt[2 * j] : 0.0;
				xj[2 * threadIdx.x + 1] = j < N ? Xt[2 * j + 1] : 0.0;
				aj[threadIdx.x] = j < N ? a[j] : 0.0;
				bj[threadIdx.x] = j < N ? b[j] : 0.0;
			}
			__syncthreads();
			const int i = blockIdx.x * 256 + threadIdx.x;
			double sa = 0.0, sb = 0.0;
			if (i < N)
			{
				const double x0 = Xt[2 * i], x1 = Xt[2 * i + 1];
				const int jn = min(256, N - j0);
				for (int jl = 0; jl < jn; ++jl)
				{
					const double d0 = __ddiv_rn(__dsub_rn(x0, xj[2 * jl]), p.l0), d1 = __ddiv_rn(__dsub_rn(x1, xj[2 * jl + 1]), p.l1);
					const double k = p.amp * exp(__ddiv_rn(-__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), 2.0));
					sa = fma(k, aj[jl], sa);
					sb = fma(k, bj[jl], sb);
				}
				const long nrows = static_cast<long>(gridDim.x) * 256;
				part[(2L * blockIdx.y) * nrows + i] = sa;
				part[(2L * blockIdx.y + 1) * nrows + i] = sb;
			}
		}
		__global__ void __launch_bounds__(256) aux_matvec_reduce_kernel(const double* __restrict__ part, int N, int nchunk, double* __restrict__ ya,
			double* __restrict__ yb)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i >= N) return;
			const long nrows = static_cast<long>(gridDim.x) * 256;
			double sa = 0.0, sb = 0.0;
			for (int c = 0; c < nchunk; ++c)
			{
				sa += part[(2L * c) * nrows + i];
				sb += part[(2L * c + 1) * nrows + i];
			}
			ya[i] = sa;
			yb[i] = sb;
		}
		__global__ void __launch_bounds__(256) aux_dots_kernel(const double* __restrict__ ga, const double* __restrict__ gb, const double* __restrict__ dw,
			int N, int Np, int n, double* __restrict__ out32)
		{
			__shared__ double red[4];
			const int ip = blockIdx.x;
			const double* __restrict__ dx = dw + static_cast<long>(ip) * n;
			const double* __restrict__ dy = dx + Np;
			double s[4] = {0, 0, 0, 0};
			for (int i = threadIdx.x; i < N; i += 256)
			{
				s[0] += ga[i] * dx[i];
				s[1] += gb[i] * dy[i];
				s[2] += ga[i] * dy[i];
				s[3] += gb[i] * dx[i];
			}
			for (int q = 0; q < 4; ++q)
			{
				const double tot = block_sum<256>(s[q], red);
				if (threadIdx.x == 0) out32[4 * ip + q] = tot;
			}
		}

		// complex_kernel.cpp:648-668: result[ip] = 2 Re( PredictionDifference^H (dK* v + K* dv + dK~* conj(v) + K~* conj(dv)) )
		// = 2 sum_i (diff_x z_x + diff_y z_y) with z = dc_ip w + c dw_ip in the [Re; Im] basis
		__global__ void __launch_bounds__(256) predict_deriv_finish_complex_kernel(const double* __restrict__ acc, int m_rows, int m_split,
			const double* __restrict__ q, int M, double self, double s0, const double* __restrict__ s_dev, const double* __restrict__ labels,
			double* __restrict__ part)
		{
			__shared__ double red[4];
			const int i = blockIdx.x * 256 + threadIdx.x;
			const double s = *s_dev;
			double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
			if (i < M)
			{
				const long ix = i, iy = m_split + i;
				const double re = acc[ix], im = acc[iy];
				const double var = self - (q[ix] + q[iy]);
				const double cf = cutoff_value(re * re + im * im, hypot(re, im), var);
				const double dx = (re * cf / s) * s - labels[2 * i] * s, dy = (im * cf / s) * s - labels[2 * i + 1] * s;
				auto plane = [&](int pl, long row) { return acc[static_cast<long>(pl) * m_rows + row]; };
				// planes: 0 = c w, 1..8 = c dw_0..7, 9..14 = dc_1..6 w
				t[0] = dx * ((2.0 / s0) * re + plane(1, ix)) + dy * ((2.0 / s0) * im + plane(1, iy));
#pragma unroll
				for (int ip = 1; ip <= 6; ++ip) t[ip] = dx * (plane(8 + ip, ix) + plane(1 + ip, ix)) + dy * (plane(8 + ip, iy) + plane(1 + ip, iy));
				t[7] = dx * plane(8, ix) + dy * plane(8, iy);
			}
			for (int ip = 0; ip < 8; ++ip)
			{
				const double tot = block_sum<256>(t[ip], red);
				if (threadIdx.x == 0) part[ip * gridDim.x + blockIdx.x] = 2.0 * tot;
			}
		}
	} // namespace

	namespace
	{
		// one wave per column i: three column dots over the Np rows of M_a
		// (blockIdx.z: one parameter of a batch — E, F and the outputs advance by their strides; a single parameter has gridDim.z = 1)
		__global__ void __launch_bounds__(256) cderiv_diag_kernel(const double* __restrict__ M, long ldm, int roff, const double* __restrict__ E,
			const double* __restrict__ F, long lde, int Np, int n, double alpha, double* __restrict__ out_diag, double* __restrict__ out_off, long ef_stride,
			long diag_stride, long off_stride)
		{
			const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
			if (i >= n) return;
			E += blockIdx.z * ef_stride, F += blockIdx.z * ef_stride, out_diag += blockIdx.z * diag_stride, out_off += blockIdx.z * off_stride;
			const double* __restrict__ mi = M + roff + static_cast<long>(i) * ldm;
			const double* __restrict__ ei = E + static_cast<long>(i) * lde;
			const double* __restrict__ fi = F + static_cast<long>(i) * lde;
			double d = 0.0, o = 0.0;
			if (i < Np)
			{
				const double* __restrict__ mp = M + roff + static_cast<long>(Np + i) * ldm;
				const double* __restrict__ ep = E + static_cast<long>(Np + i) * lde;
				const double* __restrict__ fp = F + static_cast<long>(Np + i) * lde;
				for (int j = lane; j < Np; j += 64)
				{
					const double m = mi[j], f = fi[j];
					d = fma(m, ei[j] + 2.0 * f, d);
					o = fma(m, ep[j] + fp[j], o);
					o = fma(mp[j], f, o);
				}
			}
			else
				for (int j = lane; j < Np; j += 64) d = fma(mi[j], ei[j] + 2.0 * fi[j], d);
#pragma unroll
			for (int sft = 32; sft > 0; sft >>= 1) d += __shfl_xor(d, sft), o += __shfl_xor(o, sft);
			if (lane == 0)
			{
				out_diag[i] = alpha * d;
				if (i < Np) out_off[i] = alpha * o;
			}
		}
	} // namespace
	hipError_t launch_cderiv_diag(hipStream_t s, const double* M, long ldm, int roff, const double* E, const double* F, long lde, int Np, int n, double alpha,
		double* out_diag, double* out_off)
	{
		hipLaunchKernelGGL(cderiv_diag_kernel, dim3((n + 3) / 4), dim3(256), 0, s, M, ldm, roff, E, F, lde, Np, n, alpha, out_diag, out_off, 0L, 0L, 0L);
		return hipGetLastError();
	}
	hipError_t launch_cderiv_diag_batch(hipStream_t s, int cnt, const double* M, long ldm, int roff, const double* E, const double* F, long lde, int Np, int n,
		double alpha, double* out_diag, double* out_off, long ef_stride, long diag_stride, long off_stride)
	{
		hipLaunchKernelGGL(cderiv_diag_kernel, dim3((n + 3) / 4, 1, cnt), dim3(256), 0, s, M, ldm, roff, E, F, lde, Np, n, alpha, out_diag, out_off, ef_stride,
			diag_stride, off_stride);
		return hipGetLastError();
	}
	hipError_t launch_typed_deriv_gram(hipStream_t s, const double* Xt, int N, int Np, int n, DSpecSet spec, double* D)
	{
		hipLaunchKernelGGL(typed_deriv_gram_kernel, dim3(n / 64, n / 16), dim3(256), 0, s, Xt, N, Np, n, spec, D);
		return hipGetLastError();
	}
	hipError_t launch_complex_deriv_sums(hipStream_t s, const double* w, const double* wd, const double* wx, const double* dw, const double* dwd,
		const double* dwx, int N, int Np, double* out8)
	{
		hipLaunchKernelGGL(complex_deriv_sums_kernel, dim3(1), dim3(1024), 0, s, w, wd, wx, dw, dwd, dwx, N, Np, out8);
		return hipGetLastError();
	}
	hipError_t launch_multi_quadform(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, double* part, double* out9)
	{
		const int g = (N + 63) / 64;
		hipLaunchKernelGGL(multi_quadform_kernel, dim3(g, g), dim3(256), 0, s, Xt, N, p, a, b, part);
		hipLaunchKernelGGL(sum_planes_kernel, dim3(1), dim3(256), 0, s, part, g * g, 9, out9);
		return hipGetLastError();
	}
	hipError_t launch_aux_matvec(hipStream_t s, const double* Xt, int N, SEParam p, const double* a, const double* b, double* part, double* ya,
		double* yb)
	{
		const int g = (N + 255) / 256;
		hipLaunchKernelGGL(aux_matvec_partial_kernel, dim3(g, g), dim3(256), 0, s, Xt, N, p, a, b, part);
		hipLaunchKernelGGL(aux_matvec_reduce_kernel, dim3(g), dim3(256), 0, s, part, N, g, ya, yb);
		return hipGetLastError();
	}
	hipError_t launch_aux_dots(hipStream_t s, const double* ga, const double* gb, const double* dw, int N, int Np, int n, double* out32)
	{
		hipLaunchKernelGGL(aux_dots_kernel, dim3(8), dim3(256), 0, s, ga, gb, dw, N, Np, n, out32);
		return hipGetLastError();
	}
	hipError_t launch_predict_deriv_finish_complex(hipStream_t s, const double* acc, int m_rows, int m_split, const double* q, int M, double self,
		double s0, const double* s_dev, const double* labels, double* part, double* out8)
	{
		const int nblk = (M + 255) / 256;
		hipLaunchKernelGGL(predict_deriv_finish_complex_kernel, dim3(nblk), dim3(256), 0, s, acc, m_rows, m_split, q, M, self, s0, s_dev, labels, part);
		hipLaunchKernelGGL(sum_planes_kernel, dim3(1), dim3(256), 0, s, part, nblk, 8, out8);
		return hipGetLastError();
	}
} // namespace gple
