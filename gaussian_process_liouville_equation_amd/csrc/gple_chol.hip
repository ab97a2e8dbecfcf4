// gple_chol.hip — blocked fp64 Cholesky, triangular inverse and T^T T on gfx950.
//
// Replaces the reference's Eigen::LDLT + solve(Identity) (kernel.cpp:281-283; complex_kernel.cpp:264-266).
// The kernel matrices of this path are SPD by construction (sf^2 sn^2 ridge, opt.cpp:27), and Eigen's LDLT picks
// its pivots from the not-yet-updated diagonal, which is constant here — i.e. the reference itself runs
// unpivoted.  We therefore factor K = L L^T (right-looking, 64-wide panels), form T = L^-1 by a pairwise merge
// tree of MFMA GEMMs, and only on request W = K^-1 = T^T T.  A non-positive pivot does not abort: sqrt() yields
// NaN which propagates into every output, and *info records the first offending column (reference behaviour:
// LDLT::info() is never checked, NaN/Inf are clamped later by opt.cpp:420-431).
#include <vector>

#include "gple_internal.h"

namespace gple
{
	namespace
	{
		constexpr int NB = CHOL_NB; // 64

		// One wave factors the NB x NB diagonal block and inverts the factor, register-resident:
		//   Cholesky: lane i owns row i of the block (64 doubles in VGPRs); column k is scaled lane-locally, published
		//             to LDS and broadcast back for the rank-1 update of the lane's row (fully unrolled, so every row
		//             element stays a named register);
		//   inverse : lane j owns column j of X = L^-1 and runs its own forward substitution; row i of L is an LDS
		//             broadcast.
		// ~2 x 2016 dependent FMAs with LDS broadcasts in between instead of 128 block-wide barrier rounds.
		// A (global, column-major, lda): in = SPD block (lower used), out = L (lower), strictly upper zeroed.
		// Tinv (global, ldt): out = L^-1 (lower), strictly upper zeroed.
		constexpr int LR = NB + 2; // LDS row stride (doubles): 16-byte aligned rows, rows 4 banks apart
		__device__ __forceinline__ double readlane_f64(double v, int lane)
		{
			const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
			return __hiloint2double(hi, lo);
		}
		// 1/sqrt(d) to full precision: hardware estimate + two Newton steps (a lone wave pays the full latency of every fp64
		// instruction, and sqrt + divide cost ~40 of them per column)
		__device__ __forceinline__ double rsqrt_newton(double d)
		{
			double r = __builtin_amdgcn_rsq(d);
			r = r * fma(-0.5 * d * r, r, 1.5);
			r = r * fma(-0.5 * d * r, r, 1.5);
			return r;
		}
		// The block is a single wave; the __syncthreads() below cost next to nothing at run time but are what keeps hipcc from
		// hoisting the ~2000 LDS broadcasts of the unrolled loops and spilling kilobytes per lane (measured: 7.8 KB without).
		__global__ void __launch_bounds__(64) potrf_diag_kernel(double* __restrict__ A, long lda, double* __restrict__ Tinv, long ldt,
			int* __restrict__ info, int j0)
		{
			__shared__ __attribute__((aligned(16))) double col[NB];
			__shared__ __attribute__((aligned(16))) double Ls[NB * LR];
			const int i = threadIdx.x;
			double a[NB];
#pragma unroll
			for (int j = 0; j < NB; ++j) a[j] = A[i + static_cast<long>(j) * lda]; // coalesced: lanes = consecutive rows
			bool bad = false;
			double my_rinv = 0.0; // 1 / L(i,i), kept by lane i
#pragma unroll
			for (int k = 0; k < NB; ++k)
			{
				const double d = readlane_f64(a[k], k);
				if (!bad && !(d > 0.0))
				{
					bad = true;
					if (i == 0) atomicCAS(info, 0, j0 + k + 1); // first offending column wins (info starts at 0)
				}
				const double r = rsqrt_newton(d); // NaN for d <= 0: propagates, like sqrt of a negative pivot
				double sd = d * r;
				sd = fma(fma(-sd, sd, d), 0.5 * r, sd); // sqrt(d), correctly rounded up to the last bit
				const double l = (i == k) ? sd : a[k] * r;
				a[k] = (i >= k) ? l : 0.0;
				col[i] = l;
				my_rinv = (i == k) ? r : my_rinv;
				__syncthreads();
#pragma unroll
				for (int j = k + 1; j < NB; ++j) a[j] = fma(-l, col[j], a[j]);
				__syncthreads();
			}
#pragma unroll
			for (int j = 0; j < NB; ++j)
			{
				A[i + static_cast<long>(j) * lda] = a[j];
				Ls[i * LR + j] = (j == i) ? my_rinv : a[j]; // the diagonal slot carries the reciprocal (a second LDS array for it
				                                             // makes hipcc hoist 64 loads and spill)
			}
			__syncthreads();
			// lane j: column j of X = L^-1.  x[r] = (delta_rj - sum_{k<r} L(r,k) x[k]) / L(r,r); entries above the diagonal are 0,
			// so the sum may start at k = 0 for every lane.  (A right-looking sweep with independent FMAs per step would
			// shorten the dependency chains, but hipcc then spills 6.6 KB per lane.)
			double x[NB];
			const int j = threadIdx.x;
#pragma unroll
			for (int r = 0; r < NB; ++r)
			{
				double s = (r == j) ? 1.0 : 0.0;
#pragma unroll
				for (int k = 0; k < r; ++k) s = fma(-Ls[r * LR + k], x[k], s);
				x[r] = (r >= j) ? s * Ls[r * LR + r] : 0.0;
			}
			__syncthreads();
			// transpose through LDS for coalesced stores: Ls[c][r] <- X(r, c)
#pragma unroll
			for (int r = 0; r < NB; ++r) Ls[j * LR + r] = x[r];
			__syncthreads();
#pragma unroll
			for (int c = 0; c < NB; ++c) Tinv[i + static_cast<long>(c) * ldt] = Ls[c * LR + i];
		}

		// upper(i<j) = lower(j,i) for a full symmetric result
		__global__ void __launch_bounds__(256) mirror_lower_kernel(double* __restrict__ W, long ldw, int n)
		{
			__shared__ double tile[32][33];
			const int bi = blockIdx.x, bj = blockIdx.y;
			if (bj > bi) return;
			const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
			for (int r = ty; r < 32; r += 8) tile[r][tx] = W[(bi * 32 + tx) + static_cast<long>(bj * 32 + r) * ldw];
			__syncthreads();
			// element (row bj*32 + r', col bi*32 + c') = lower(bi*32 + c', bj*32 + r')
			for (int r = ty; r < 32; r += 8)
			{
				const int row = bj * 32 + tx, col = bi * 32 + r;
				if (row < col) W[row + static_cast<long>(col) * ldw] = tile[tx][r];
			}
		}

		int pick_tile(long m, long n, long batch)
		{
			if (m % 128 || n % 128) return 64;
			return (m / 128) * (n / 128) * batch >= 256 ? 128 : 64;
		}
	} // namespace

	hipError_t potrf_lower(hipStream_t s, double* A, long lda, int n, double* T, long ldt, int* info)
	{
		if (n % NB) return hipErrorInvalidValue;
		for (int j0 = 0; j0 < n; j0 += NB)
		{
			hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(64), 0, s, A + j0 + static_cast<long>(j0) * lda, lda,
				T + j0 + static_cast<long>(j0) * ldt, ldt, info, j0);
			const int m = n - j0 - NB;
			if (m <= 0) break;
			double* P = A + (j0 + NB) + static_cast<long>(j0) * lda;
			GemmDesc g{};
			// panel: P <- P * inv(L_jj)^T   (in place: one 64-wide n-tile per workgroup covers the whole panel width)
			g.A = P, g.lda = lda, g.B = T + j0 + static_cast<long>(j0) * ldt, g.ldb = ldt, g.C = P, g.ldc = lda;
			g.M = m, g.N = NB, g.K = NB, g.batch = 1, g.alpha = 1.0, g.beta = 0.0, g.krange = K_FULL, g.lower_only = 0;
			g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
			hipError_t e = launch_gemm(s, g, 64);
			if (e != hipSuccess) return e;
			// trailing update: A22 -= P P^T (lower tiles)
			g.A = P, g.B = P, g.ldb = lda, g.C = A + (j0 + NB) + static_cast<long>(j0 + NB) * lda;
			g.M = m, g.N = m, g.K = NB, g.alpha = -1.0, g.beta = 1.0, g.lower_only = 1;
			e = launch_gemm(s, g, pick_tile(m, m, 1) == 128 && m >= 2048 ? 128 : 64);
			if (e != hipSuccess) return e;
		}
		return hipGetLastError();
	}

	hipError_t trtri_lower_from_diag(hipStream_t s, const double* L, long ldl, double* T, long ldt, int n, double* work)
	{
		struct Blk
		{
			int start, size;
		};
		std::vector<Blk> blocks;
		for (int j0 = 0; j0 < n; j0 += NB) blocks.push_back({j0, NB});
		while (blocks.size() > 1)
		{
			std::vector<Blk> next;
			size_t i = 0;
			while (i + 1 < blocks.size())
			{
				// a run of consecutive pairs with identical sizes becomes one batched launch
				const int s1 = blocks[i].size, s2 = blocks[i + 1].size, start = blocks[i].start;
				int count = 0;
				size_t j = i;
				while (j + 1 < blocks.size() && blocks[j].size == s1 && blocks[j + 1].size == s2
					&& blocks[j].start == start + count * (s1 + s2))
				{
					next.push_back({blocks[j].start, s1 + s2});
					++count;
					j += 2;
				}
				const long step = s1 + s2;
				GemmDesc g{};
				// W_b (s2 x s1) = L21_b * T11_b
				g.A = L + (start + s1) + static_cast<long>(start) * ldl, g.lda = ldl, g.strideA = step * (ldl + 1);
				g.B = T + start + static_cast<long>(start) * ldt, g.ldb = ldt, g.strideB = step * (ldt + 1);
				g.C = work, g.ldc = s2, g.strideC = static_cast<long>(s1) * s2;
				g.M = s2, g.N = s1, g.K = s1, g.batch = count, g.alpha = 1.0, g.beta = 0.0;
				g.krange = K_GE_N, g.lower_only = 0, g.a_kmajor = false, g.b_kmajor = true, g.c_trans = false;
				const int tile = pick_tile(s2, s1, count);
				hipError_t e = launch_gemm(s, g, tile);
				if (e != hipSuccess) return e;
				// T21_b = - T22_b * W_b
				g.A = T + (start + s1) + static_cast<long>(start + s1) * ldt, g.lda = ldt, g.strideA = step * (ldt + 1);
				g.B = work, g.ldb = s2, g.strideB = static_cast<long>(s1) * s2;
				g.C = T + (start + s1) + static_cast<long>(start) * ldt, g.ldc = ldt, g.strideC = step * (ldt + 1);
				g.K = s2, g.alpha = -1.0, g.krange = K_LE_M;
				e = launch_gemm(s, g, tile);
				if (e != hipSuccess) return e;
				i = j;
			}
			if (i < blocks.size()) next.push_back(blocks[i]);
			blocks.swap(next);
		}
		return hipGetLastError();
	}

	hipError_t lauum_full(hipStream_t s, const double* T, long ldt, double* W, long ldw, int n)
	{
		GemmDesc g{};
		g.A = T, g.lda = ldt, g.B = T, g.ldb = ldt, g.C = W, g.ldc = ldw;
		g.M = n, g.N = n, g.K = n, g.batch = 1, g.alpha = 1.0, g.beta = 0.0;
		g.krange = K_GE_MAX_MN, g.lower_only = 1, g.a_kmajor = true, g.b_kmajor = true, g.c_trans = false;
		hipError_t e = launch_gemm(s, g, pick_tile(n, n, 1));
		if (e != hipSuccess) return e;
		hipLaunchKernelGGL(mirror_lower_kernel, dim3(n / 32, n / 32), dim3(256), 0, s, W, ldw, n);
		return hipGetLastError();
	}
} // namespace gple
