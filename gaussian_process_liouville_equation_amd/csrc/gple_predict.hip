// gple_predict.hip — fused GP prediction kernel for gfx950 (the dominant kernel of the fit+predict step).
//
// Reference hot loop 4 (kernel.cpp:495-518, complex_kernel.cpp:608-642):
//     mu_i  = K*_i v                      (M x N Gram row times weights)
//     var_i = k(x*,x*) - K*_i K^-1 K*_i^T   (one N x N GEMV + dot per test point, 2 M N^2 flops)
// The reference materialises the M x N matrix K* and the explicit inverse.  Here
//   * K^-1 = T^T T with T = chol(K)^-1 lower triangular, so  K*_i K^-1 K*_i^T = || T K*_i^T ||^2  — a triangular
//     contraction (M N^2 flops) whose result is a sum of squares (no cancellation inside the quadratic form);
//   * K* is never written to HBM: every 128 x 16 slab of it is generated straight into LDS (exp on the VALU) and
//     consumed as the MFMA operand; the only HBM/L2 stream is T (read once per 128-row tile of test points);
//   * the row norms are accumulated in registers across all N-tiles, the mean is accumulated while the slab for the
//     last N-tile (which spans every k) is generated.
// One workgroup = 8 waves owns 128 test rows and loops over 256-wide N-tiles; K advances 16 per step through a
// double-buffered LDS stage (A: generated 16 x 128, B: 16 x 256 of T).  Each wave owns 16 rows x all 256 columns
// (16 accumulator tiles): every wave then sees the same triangular structure, so the all-zero 16 x 16 blocks of T
// (k > n) are skipped by all waves alike and nobody idles behind the per-step barrier (a 2 x 4 wave grid left whole
// SIMDs idle in the diagonal region: measured 51 % -> see profiles/).  v_mfma_f64_16x16x4_f64 with the result rows
// on n and the result columns (lane & 15) on the test row m, so that the squared row sums stay lane-local.
//
// "Typed" rows/columns implement the complex GP as a real GP on [Re; Im] (see gple_kernels.h, SEParamSet).
#include <type_traits>

#include "gple_kernels.h"

namespace gple
{
	typedef double d4 __attribute__((ext_vector_type(4)));
	typedef double d2 __attribute__((ext_vector_type(2)));

	namespace
	{
		constexpr int BM = 128, BN = 256, BK = 16;
		constexpr int AS = BM + 16, BS = BN + 16;
		constexpr int NTHREADS = 512;

		__global__ void __launch_bounds__(NTHREADS) predict_q_kernel(const PredictArgs a)
		{
			__shared__ __attribute__((aligned(16))) double lds[2 * BK * AS + 2 * BK * BS];
			double* const As = lds;
			double* const Bs = lds + 2 * BK * AS;

			const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
			const int fk = lane >> 4, fr = lane & 15;
			const int m0 = blockIdx.x * BM;
			// this thread's test point (fixed for the whole kernel); rows beyond M are clamped and never stored
			// kq is the same for all lanes of a wave; readfirstlane tells the compiler, so the training point and weight
			// loads of gen_a become scalar (SMEM) loads
			const int ml = t & 127, kq = __builtin_amdgcn_readfirstlane(t >> 7);
			const int gm = m0 + ml;
			const int type_m = m0 >= a.m_split; // uniform per workgroup (m_split is a multiple of BM)
			int pidx = type_m ? gm - a.m_split : gm;
			pidx = pidx < a.M ? pidx : a.M - 1;
			const double xm = a.Xs[2 * pidx], pm = a.Xs[2 * pidx + 1];

			double rsq = 0.0;
			double mu_acc = 0.0;
			const int ntiles = a.n_total / BN;

			// generate the A slab for k-tile k0 into buffer S; accumulate the mean when asked
			auto gen_a = [&](double* __restrict__ S, int k0, bool with_mean) {
				const int type_k = k0 >= a.n_split; // uniform per k-tile (n_split is a multiple of BK)
				const SEParam& p = a.ps.p[type_m + type_k];
				const double amp = p.amp, n2 = (type_m == type_k) ? p.n2 : 0.0, rl0 = p.rl0, rl1 = p.rl1;
				const int kbase = k0 + kq * 4;
#pragma unroll 2
				for (int e = 0; e < 4; ++e)
				{
					const int k = kbase + e;
					const int pk = type_k ? k - a.n_split : k;
					double val = 0.0;
					if (pk < a.N)
					{
						const double xk = a.Xt[2 * pk], pkv = a.Xt[2 * pk + 1];
						const double d0 = (xm - xk) * rl0, d1 = (pm - pkv) * rl1;
						const double g = exp(-0.5 * (d0 * d0 + d1 * d1));
						const double delta = (xm == xk && pm == pkv) ? n2 : 0.0; // delta_kernel: exact equality, kernel.cpp:26
						val = amp * (g + delta);
						if (with_mean) mu_acc = fma(val, a.v[k], mu_acc);
					}
					S[(kq * 4 + e) * AS + ml] = val;
				}
			};
			d2 breg[4];
			auto load_b = [&](int n0, int k0) {
				const double* __restrict__ base = a.T + n0 + static_cast<long>(k0) * a.ldt;
#pragma unroll
				for (int q = 0; q < 4; ++q)
				{
					const int i = t + NTHREADS * q;
					const int r2 = (i & 127) * 2, k = i >> 7;
					breg[q] = *reinterpret_cast<const d2*>(base + r2 + static_cast<long>(k) * a.ldt);
				}
			};
			auto store_b = [&](double* __restrict__ S) {
#pragma unroll
				for (int q = 0; q < 4; ++q)
				{
					const int i = t + NTHREADS * q;
					const int r2 = (i & 127) * 2, k = i >> 7;
					*reinterpret_cast<d2*>(S + k * BS + r2) = breg[q];
				}
			};

			for (int jt = 0; jt < ntiles; ++jt)
			{
				const int n0 = jt * BN;
				const int nk = (n0 + BN) / BK; // T(n,k) = 0 for k > n: k-tiles beyond the N-tile's last column are skipped
				const bool with_mean = jt == ntiles - 1;
				d4 acc[16];
#pragma unroll
				for (int j = 0; j < 16; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};

				__syncthreads(); // the previous tile's last compute is done before the stage is refilled
				gen_a(As, 0, with_mean);
				load_b(n0, 0);
				store_b(Bs);
				__syncthreads();
				// One k-step: MFMAs of slab `it` against the column blocks j >= JMIN, then stage slab it + 1.
				// JMIN is a compile-time constant per loop: any branch that merges around the accumulators makes hipcc spill
				// hundreds of VGPRs, so the triangular skipping is expressed as four consecutive loops instead.
				auto kstep = [&](auto jmin_tag, int it) {
					constexpr int JMIN = decltype(jmin_tag)::value;
					const int cur = it & 1, k0 = it * BK;
					if (it + 1 < nk) load_b(n0, k0 + BK);
					const double* __restrict__ pa = As + cur * BK * AS + w * 16 + fr;
					const double* __restrict__ pb = Bs + cur * BK * BS + fr;
#pragma unroll
					for (int kk = 0; kk < BK; kk += 4)
					{
						const double af = pa[(kk + fk) * AS];
#pragma unroll
						for (int h = JMIN; h < 16; h += 4)
						{
							double bf[4];
#pragma unroll
							for (int j = 0; j < 4; ++j) bf[j] = pb[(kk + fk) * BS + (h + j) * 16];
#pragma unroll
							for (int j = 0; j < 4; ++j) acc[h + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af, acc[h + j], 0, 0, 0);
						}
					}
					if (it + 1 < nk)
					{
						gen_a(As + (cur ^ 1) * BK * AS, k0 + BK, with_mean);
						store_b(Bs + (cur ^ 1) * BK * BS);
					}
					__syncthreads();
				};
				// k-steps below the diagonal N-tile and its first 64 columns see every column block; afterwards the column
				// blocks left of the current k (T(n,k) = 0 for k > n) drop out, 64 columns at a time
				const int nd = n0 / BK;
				int it = 0;
				for (; it < nd + 4; ++it) kstep(std::integral_constant<int, 0>{}, it);
				for (; it < nd + 8; ++it) kstep(std::integral_constant<int, 4>{}, it);
				for (; it < nd + 12; ++it) kstep(std::integral_constant<int, 8>{}, it);
				for (; it < nd + 16; ++it) kstep(std::integral_constant<int, 12>{}, it);
				// result element [n = 16 j + fk + 4 r][m = 16 w + fr]: the row index m is lane-local
#pragma unroll
				for (int j = 0; j < 16; ++j)
#pragma unroll
					for (int r = 0; r < 4; ++r) rsq = fma(acc[j][r], acc[j][r], rsq);
			}

			// reduce over the four lane groups (same fr); the four k-quarters of the mean go through LDS
			rsq += __shfl_xor(rsq, 16);
			rsq += __shfl_xor(rsq, 32);
			__syncthreads();
			double* const red_q = lds;        // [128]
			double* const red_mu = lds + 128; // [4 (kq)][128]
			if (lane < 16) red_q[w * 16 + lane] = rsq;
			red_mu[kq * 128 + ml] = mu_acc;
			__syncthreads();
			if (t < 128)
			{
				const int row = m0 + t;
				const int prow = type_m ? row - a.m_split : row;
				if (prow < a.M)
				{
					a.q[row] = red_q[t];
					a.mu[row] = (red_mu[t] + red_mu[128 + t]) + (red_mu[256 + t] + red_mu[384 + t]);
				}
			}
		}
	} // namespace

	hipError_t launch_predict_q(hipStream_t s, const PredictArgs& a)
	{
		if (a.M <= 0) return hipSuccess;
		if (a.m_rows % BM || a.n_total % BN || a.m_split % BM || a.n_split % BN) return hipErrorInvalidValue;
		hipLaunchKernelGGL(predict_q_kernel, dim3(a.m_rows / BM), dim3(NTHREADS), 0, s, a);
		return hipGetLastError();
	}
} // namespace gple
