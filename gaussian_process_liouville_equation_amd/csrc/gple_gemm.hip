// gple_gemm.hip — fp64 MFMA (v_mfma_f64_16x16x4_f64) GEMM family for gfx950.
//
// Used by the dense part of the fit (reference hot loops 2,3,5: Eigen LDLT / solve(Identity) / -W dK W,
// kernel.cpp:281-283, 354-358): Cholesky trailing updates and panel solves, the triangular inverse and T^T T.
//
//   C(m,n) = alpha * sum_k A(m,k) B(n,k) + beta * C(m,n)
//
// One workgroup = 4 waves (2 x 2) computes a BM x BN tile (64x64 or 128x128); K advances in steps of 16 through a
// double-buffered LDS stage; every wave owns a (BM/2) x (BN/2) sub-tile held in VGPR-form MFMA accumulators
// (launch bound 256 threads x 2 waves/SIMD caps the budget at 256 VGPRs, which keeps hipcc from bouncing the
// accumulators between AGPRs and VGPRs every iteration — measured 35 vs 78 TFLOP/s on MI355X).
//
// Fragment maps of v_mfma_f64_16x16x4_f64 (verified on hardware, probes/mfma_f64_probe.hip):
//   first operand  X[i = lane & 15][k = lane >> 4]   (16 x 4),
//   second operand Y[k = lane >> 4][j = lane & 15]   (4 x 16),
//   result         D[i = (lane >> 4) + 4 * reg][j = lane & 15].
// The operand that should end up contiguous in memory is therefore fed as the SECOND operand.
#include <cstdlib>

#include "gple_internal.h"
#include <cstdio>
#include <mutex>

namespace gple
{
	typedef double d4 __attribute__((ext_vector_type(4)));
	typedef double d2 __attribute__((ext_vector_type(2)));

	namespace
	{
		constexpr int BK = 16;
		constexpr int LPAD = 16; // row stride (R + 16) doubles: the 4 k-rows of a fragment read hit disjoint bank halves

		// Stages an R x 16 operand tile (rows r, depth k) from HBM into LDS as S[k][r].
		template <int R, bool KMAJOR>
		struct TileLoader
		{
			static constexpr int NV = R * BK / 2 / 256; // double2 per thread
			static constexpr int RS = R + LPAD;
			d2 v[NV];
			__device__ __forceinline__ void load(const double* __restrict__ base, long ld, int t)
			{
				if constexpr (!KMAJOR)
				{
#pragma unroll
					for (int q = 0; q < NV; ++q)
					{
						const int i = t + 256 * q;
						const int r2 = (i % (R / 2)) * 2, k = i / (R / 2);
						v[q] = *reinterpret_cast<const d2*>(base + r2 + static_cast<long>(k) * ld);
					}
				}
				else
				{
					constexpr int TPR = 8 / NV; // threads per row
					const int r = t / TPR, kofs = (t % TPR) * (2 * NV);
#pragma unroll
					for (int q = 0; q < NV; ++q)
						v[q] = *reinterpret_cast<const d2*>(base + kofs + 2 * q + static_cast<long>(r) * ld);
				}
			}
			__device__ __forceinline__ void store(double* __restrict__ S, int t) const
			{
				if constexpr (!KMAJOR)
				{
#pragma unroll
					for (int q = 0; q < NV; ++q)
					{
						const int i = t + 256 * q;
						const int r2 = (i % (R / 2)) * 2, k = i / (R / 2);
						*reinterpret_cast<d2*>(S + k * RS + r2) = v[q];
					}
				}
				else
				{
					constexpr int TPR = 8 / NV;
					const int r = t / TPR, kofs = (t % TPR) * (2 * NV);
#pragma unroll
					for (int q = 0; q < NV; ++q)
					{
						S[(kofs + 2 * q) * RS + r] = v[q].x;
						S[(kofs + 2 * q + 1) * RS + r] = v[q].y;
					}
				}
			}
		};

		template <int BM, int BN, bool AK, bool BKM, bool CT>
		__global__ void __launch_bounds__(256, 2) gemm_f64_kernel(const GemmDesc g)
		{
			constexpr int WTM = BM / 2, WTN = BN / 2;
			constexpr int TM = WTM / 16, TN = WTN / 16;
			constexpr int AS = BM + LPAD, BS = BN + LPAD;
			__shared__ __attribute__((aligned(16))) double lds[2 * BK * AS + 2 * BK * BS];
			double* const As = lds;
			double* const Bs = lds + 2 * BK * AS;

			const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
			// K_LE_M: the k-range of a tile grows with its row block; workgroups are dispatched in blockIdx.x order, so the row
			// blocks are walked from the bottom up — the longest tiles start first and the short ones fill the tail
			int bx = g.krange == K_LE_M ? gridDim.x - 1 - blockIdx.x : blockIdx.x, by = blockIdx.y;
			// Dense products: workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), each XCD with its own 4 MB L2.
			// In plain order the ~64 workgroups resident on one XCD are every 8th row tile of a few columns: 64 operand panels for 64
			// tiles, and the n = 8192 derivative GEMM fetched 15 x its algorithmic bytes from HBM (profiles/r03_traffic.json, C4opt_only1
			// v1).  Here XCD x works through its own strip of N-tiles in row-major order: the resident workgroups form a block of
			// (64 / strip width) x (strip width) tiles that moves through k together.  Placement is a speed matter only (the tile -> result
			// map is the same), and dispatch order is not promised: if it ever changes the kernel is merely back to the old traffic.
			if (g.krange == K_FULL && !g.lower_only && gridDim.y % 8 == 0 && gridDim.z == 1)
			{
				const int L = blockIdx.x + gridDim.x * blockIdx.y, sw = gridDim.y / 8, j = L >> 3;
				by = (L & 7) * sw + j % sw, bx = j / sw;
			}
			const int m0 = bx * BM, n0 = by * BN;
			if (g.lower_only && n0 >= m0 + BM) return;
			int kb = 0, ke = g.K;
			if (g.krange == K_GE_N) kb = n0;
			else if (g.krange == K_LE_M) ke = min(g.K, m0 + BM);
			else if (g.krange == K_GE_MAX_MN) kb = max(m0, n0);
			kb = kb / BK * BK;
			const int nk = (ke - kb + BK - 1) / BK;

			const double* __restrict__ A = g.A + blockIdx.z * g.strideA;
			const double* __restrict__ B = g.B + blockIdx.z * g.strideB;
			double* __restrict__ C = g.C + blockIdx.z * g.strideC;
			// address of operand element (r0, k)
			auto a_at = [&](int k) { return AK ? A + k + static_cast<long>(m0) * g.lda : A + m0 + static_cast<long>(k) * g.lda; };
			auto b_at = [&](int k) { return BKM ? B + k + static_cast<long>(n0) * g.ldb : B + n0 + static_cast<long>(k) * g.ldb; };

			d4 acc[TM][TN];
#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int j = 0; j < TN; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

			TileLoader<BM, AK> la;
			TileLoader<BN, BKM> lb;
			if (nk > 0)
			{
				la.load(a_at(kb), g.lda, t);
				lb.load(b_at(kb), g.ldb, t);
				la.store(As, t);
				lb.store(Bs, t);
			}
			__syncthreads();
			const int fk = lane >> 4, fr = lane & 15;
			for (int it = 0; it < nk; ++it)
			{
				const int cur = it & 1;
				if (it + 1 < nk)
				{
					la.load(a_at(kb + (it + 1) * BK), g.lda, t);
					lb.load(b_at(kb + (it + 1) * BK), g.ldb, t);
				}
				const double* __restrict__ a = As + cur * BK * AS + wm * WTM + fr;
				const double* __restrict__ b = Bs + cur * BK * BS + wn * WTN + fr;
#pragma unroll
				for (int kk = 0; kk < BK; kk += 4)
				{
					double af[TM], bf[TN];
#pragma unroll
					for (int i = 0; i < TM; ++i) af[i] = a[(kk + fk) * AS + i * 16];
#pragma unroll
					for (int j = 0; j < TN; ++j) bf[j] = b[(kk + fk) * BS + j * 16];
#pragma unroll
					for (int i = 0; i < TM; ++i)
#pragma unroll
						for (int j = 0; j < TN; ++j)
						{
							if constexpr (CT) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
							else acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
						}
				}
				if (it + 1 < nk)
				{
					la.store(As + (cur ^ 1) * BK * AS, t);
					lb.store(Bs + (cur ^ 1) * BK * BS, t);
				}
				__syncthreads();
			}

			const double alpha = g.alpha, beta = g.beta;
#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int j = 0; j < TN; ++j)
#pragma unroll
					for (int r = 0; r < 4; ++r)
					{
						long idx;
						if constexpr (CT)
						{
							const int m = m0 + wm * WTM + i * 16 + fk + 4 * r, n = n0 + wn * WTN + j * 16 + fr;
							idx = n + static_cast<long>(m) * g.ldc;
						}
						else
						{
							const int m = m0 + wm * WTM + i * 16 + fr, n = n0 + wn * WTN + j * 16 + fk + 4 * r;
							idx = m + static_cast<long>(n) * g.ldc;
						}
						double val = alpha * acc[i][j][r];
						if (beta != 0.0) val += beta * C[idx];
						C[idx] = val;
					}
		}

		// The 64 x 64 tile for the latency-bound GEMMs of the fit (Cholesky trailing updates with K = 64, merge-tree levels
		// with K = 64 .. 512 on a few dozen workgroups): with one slab of 16 in flight every k-step pays a full HBM/L2 round
		// trip (measured 10.9 us for a K = 64 update whose MFMAs take 1.7 us).  Here four slabs are in flight: a 4-deep
		// register prefetch feeding a 4-slab LDS ring, and the C tile of a beta != 0 update is requested up front as well.
		template <bool AK, bool BKM, bool CT>
		__global__ void __launch_bounds__(256, 2) gemm_f64_deep_kernel(const GemmDesc g)
		{
			constexpr int BM = 64, BN = 64, DEPTH = 4;
			constexpr int WTM = BM / 2, WTN = BN / 2;
			constexpr int TM = WTM / 16, TN = WTN / 16;
			constexpr int AS = BM + LPAD, BS = BN + LPAD;
			__shared__ __attribute__((aligned(16))) double lds[DEPTH * BK * AS + DEPTH * BK * BS];
			double* const As = lds;
			double* const Bs = lds + DEPTH * BK * AS;

			const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
			// K_LE_M: the k-range of a tile grows with its row block; workgroups are dispatched in blockIdx.x order, so the row
			// blocks are walked from the bottom up — the longest tiles start first and the short ones fill the tail
			const int bx = g.krange == K_LE_M ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
			const int m0 = bx * BM, n0 = blockIdx.y * BN;
			if (g.lower_only && n0 >= m0 + BM) return;
			int kb = 0, ke = g.K;
			if (g.krange == K_GE_N) kb = n0;
			else if (g.krange == K_LE_M) ke = min(g.K, m0 + BM);
			else if (g.krange == K_GE_MAX_MN) kb = max(m0, n0);
			kb = kb / BK * BK;
			const int nk = (ke - kb + BK - 1) / BK;

			const double* __restrict__ A = g.A + blockIdx.z * g.strideA;
			const double* __restrict__ B = g.B + blockIdx.z * g.strideB;
			double* __restrict__ C = g.C + blockIdx.z * g.strideC;
			auto a_at = [&](int k) { return AK ? A + k + static_cast<long>(m0) * g.lda : A + m0 + static_cast<long>(k) * g.lda; };
			auto b_at = [&](int k) { return BKM ? B + k + static_cast<long>(n0) * g.ldb : B + n0 + static_cast<long>(k) * g.ldb; };
			const int fk = lane >> 4, fr = lane & 15;
			auto c_index = [&](int i, int j, int r) -> long {
				if constexpr (CT)
				{
					const int m = m0 + wm * WTM + i * 16 + fk + 4 * r, n = n0 + wn * WTN + j * 16 + fr;
					return n + static_cast<long>(m) * g.ldc;
				}
				else
				{
					const int m = m0 + wm * WTM + i * 16 + fr, n = n0 + wn * WTN + j * 16 + fk + 4 * r;
					return m + static_cast<long>(n) * g.ldc;
				}
			};

			TileLoader<BM, AK> la[DEPTH];
			TileLoader<BN, BKM> lb[DEPTH];
#pragma unroll
			for (int p = 0; p < DEPTH; ++p)
				if (p < nk)
				{
					la[p].load(a_at(kb + p * BK), g.lda, t);
					lb[p].load(b_at(kb + p * BK), g.ldb, t);
				}
			const double alpha = g.alpha, beta = g.beta;
			double cold[TM][TN][4];
			if (beta != 0.0)
#pragma unroll
				for (int i = 0; i < TM; ++i)
#pragma unroll
					for (int j = 0; j < TN; ++j)
#pragma unroll
						for (int r = 0; r < 4; ++r) cold[i][j][r] = C[c_index(i, j, r)];

			d4 acc[TM][TN];
#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int j = 0; j < TN; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

			for (int it0 = 0; it0 < nk; it0 += DEPTH)
			{
#pragma unroll
				for (int p = 0; p < DEPTH; ++p)
				{
					const int it = it0 + p;
					if (it < nk) // uniform
					{
						// ring slot p was last read in step it - DEPTH; every wave has passed DEPTH - 1 barriers since
						la[p].store(As + p * BK * AS, t);
						lb[p].store(Bs + p * BK * BS, t);
						if (it + DEPTH < nk)
						{
							la[p].load(a_at(kb + (it + DEPTH) * BK), g.lda, t);
							lb[p].load(b_at(kb + (it + DEPTH) * BK), g.ldb, t);
						}
						__syncthreads();
						const double* __restrict__ a = As + p * BK * AS + wm * WTM + fr;
						const double* __restrict__ b = Bs + p * BK * BS + wn * WTN + fr;
#pragma unroll
						for (int kk = 0; kk < BK; kk += 4)
						{
							double af[TM], bf[TN];
#pragma unroll
							for (int i = 0; i < TM; ++i) af[i] = a[(kk + fk) * AS + i * 16];
#pragma unroll
							for (int j = 0; j < TN; ++j) bf[j] = b[(kk + fk) * BS + j * 16];
#pragma unroll
							for (int i = 0; i < TM; ++i)
#pragma unroll
								for (int j = 0; j < TN; ++j)
								{
									if constexpr (CT) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
									else acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
								}
						}
					}
				}
			}

#pragma unroll
			for (int i = 0; i < TM; ++i)
#pragma unroll
				for (int j = 0; j < TN; ++j)
#pragma unroll
					for (int r = 0; r < 4; ++r)
					{
						double val = alpha * acc[i][j][r];
						if (beta != 0.0) val += beta * cold[i][j][r];
						C[c_index(i, j, r)] = val;
					}
		}

		// Latency-bound GEMMs (merge-tree levels on a few dozen 64-tiles, K = 64 trailing updates): a 64 x 64 tile with a long
		// k-range is MFMA-bound on ONE compute unit (0.43 us per k-step of 16) while most of the chip idles.  Here a workgroup
		// owns a 32 x 32 tile and its four waves split the k-range between them (four times the workgroups, a quarter of the
		// dependent k-steps); the partial sums meet in LDS and are added in a fixed order, so the result does not depend on
		// timing.  Operands go straight from L2 into MFMA fragments (no LDS staging, no barrier in the k-loop): with the k index
		// of sub-step s taken as k0 + 4 (lane >> 4) + s, a k-major operand is one 32-byte load per lane and 16 k.
		template <bool KMAJOR>
		struct FragLoader
		{
			double v[2][4]; // [16-row half][sub-step]
			__device__ __forceinline__ void load(const double* __restrict__ base, long ld, int r0, int k0, int fr, int fk)
			{
#pragma unroll
				for (int i = 0; i < 2; ++i)
				{
					if constexpr (KMAJOR)
					{
						const d4 x = *reinterpret_cast<const d4*>(base + k0 + 4 * fk + static_cast<long>(r0 + 16 * i + fr) * ld);
						v[i][0] = x.x, v[i][1] = x.y, v[i][2] = x.z, v[i][3] = x.w;
					}
					else
					{
#pragma unroll
						for (int s = 0; s < 4; ++s) v[i][s] = base[r0 + 16 * i + fr + static_cast<long>(k0 + 4 * fk + s) * ld];
					}
				}
			}
		};
		template <bool AK, bool BKM>
		__global__ void __launch_bounds__(256, 2) gemm_f64_splitk_kernel(const GemmDesc g)
		{
			constexpr int BT = 32;
			__shared__ __attribute__((aligned(16))) double red[4][BT * BT];
			const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
			const int fr = lane & 15, fk = lane >> 4;
			const int m0 = blockIdx.x * BT, n0 = blockIdx.y * BT;
			if (g.lower_only && n0 >= m0 + BT) return;
			int kb = 0, ke = g.K;
			if (g.krange == K_GE_N) kb = n0;
			else if (g.krange == K_LE_M) ke = min(g.K, m0 + BT);
			else if (g.krange == K_GE_MAX_MN) kb = max(m0, n0);
			kb = kb / BK * BK;
			const int ng = (ke - kb + BK - 1) / BK;
			const int g_begin = w * ng / 4, g_end = (w + 1) * ng / 4; // this wave's k-groups of 16

			const double* __restrict__ A = g.A + blockIdx.z * g.strideA;
			const double* __restrict__ B = g.B + blockIdx.z * g.strideB;
			double* __restrict__ C = g.C + blockIdx.z * g.strideC;
			const double alpha = g.alpha, beta = g.beta;
			double cold[4];
			if (beta != 0.0)
#pragma unroll
				for (int q = 0; q < 4; ++q)
				{
					const int e = t + 256 * q;
					cold[q] = C[m0 + (e & 31) + static_cast<long>(n0 + (e >> 5)) * g.ldc];
				}

			d4 acc[2][2];
#pragma unroll
			for (int i = 0; i < 2; ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
			FragLoader<AK> fa[2];
			FragLoader<BKM> fb[2];
			auto compute = [&](const FragLoader<AK>& a, const FragLoader<BKM>& b) {
#pragma unroll
				for (int s = 0; s < 4; ++s)
#pragma unroll
					for (int i = 0; i < 2; ++i)
#pragma unroll
						for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(b.v[j][s], a.v[i][s], acc[i][j], 0, 0, 0);
			};
			if (g_begin < g_end)
			{
				fa[0].load(A, g.lda, m0, kb + g_begin * BK, fr, fk);
				fb[0].load(B, g.ldb, n0, kb + g_begin * BK, fr, fk);
			}
			for (int gi = g_begin; gi < g_end; gi += 2)
			{
				if (gi + 1 < g_end)
				{
					fa[1].load(A, g.lda, m0, kb + (gi + 1) * BK, fr, fk);
					fb[1].load(B, g.ldb, n0, kb + (gi + 1) * BK, fr, fk);
				}
				compute(fa[0], fb[0]);
				if (gi + 1 < g_end)
				{
					if (gi + 2 < g_end)
					{
						fa[0].load(A, g.lda, m0, kb + (gi + 2) * BK, fr, fk);
						fb[0].load(B, g.ldb, n0, kb + (gi + 2) * BK, fr, fk);
					}
					compute(fa[1], fb[1]);
				}
			}
			// result layout of the MFMA (first operand = B rows): D[n = (lane >> 4) + 4 r][m = lane & 15]
#pragma unroll
			for (int i = 0; i < 2; ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j)
#pragma unroll
					for (int r = 0; r < 4; ++r) red[w][(16 * j + fk + 4 * r) * BT + 16 * i + fr] = acc[i][j][r];
			__syncthreads();
#pragma unroll
			for (int q = 0; q < 4; ++q)
			{
				const int e = t + 256 * q;
				double val = alpha * (((red[0][e] + red[1][e]) + red[2][e]) + red[3][e]);
				if (beta != 0.0) val += beta * cold[q];
				C[m0 + (e & 31) + static_cast<long>(n0 + (e >> 5)) * g.ldc] = val;
			}
		}
		hipError_t launch_splitk(hipStream_t s, const GemmDesc& d)
		{
			if (d.M % 32 || d.N % 32 || d.K % BK || d.M <= 0 || d.N <= 0 || d.batch <= 0 || d.c_trans) return hipErrorInvalidValue;
			const dim3 grid(d.M / 32, d.N / 32, d.batch), block(256);
			if (!d.a_kmajor && !d.b_kmajor) hipLaunchKernelGGL((gemm_f64_splitk_kernel<false, false>), grid, block, 0, s, d);
			else if (!d.a_kmajor && d.b_kmajor) hipLaunchKernelGGL((gemm_f64_splitk_kernel<false, true>), grid, block, 0, s, d);
			else return hipErrorInvalidValue;
			return hipGetLastError();
		}

		template <int T>
		hipError_t launch_tile(hipStream_t s, const GemmDesc& d)
		{
			if (d.M % T || d.N % T || d.K % BK || d.M <= 0 || d.N <= 0 || d.batch <= 0) return hipErrorInvalidValue;
			const dim3 grid(d.M / T, d.N / T, d.batch), block(256);
#define GPLE_GEMM_CASE(ak, bk, ct)                                                           \
	if (d.a_kmajor == ak && d.b_kmajor == bk && d.c_trans == ct)                              \
	{                                                                                         \
		if constexpr (T == 64) hipLaunchKernelGGL((gemm_f64_deep_kernel<ak, bk, ct>), grid, block, 0, s, d); \
		else hipLaunchKernelGGL((gemm_f64_kernel<T, T, ak, bk, ct>), grid, block, 0, s, d);    \
		return hipGetLastError();                                                             \
	}
			GPLE_GEMM_CASE(false, false, false)
			GPLE_GEMM_CASE(false, true, false)
			GPLE_GEMM_CASE(true, true, false)
			GPLE_GEMM_CASE(false, false, true)
			GPLE_GEMM_CASE(false, true, true)
#undef GPLE_GEMM_CASE
			return hipErrorInvalidValue;
		}
	} // namespace

	int gemm_pick_tile(long m, long n, long batch, bool triangular)
	{
		// 128-tiles (4 waves, 16 accumulator tiles each) need at least two workgroups per CU to cover their LDS latency;
		// below that the 64-tile kernel with its deeper prefetch and 4x more workgroups is faster.  Triangular work (lower
		// tiles only, or a k-range that depends on the tile) is uneven per tile, so small tiles stay ahead much longer.
		static const long min_dense = [] {
			const char* e = getenv("GPLE_GEMM_128_MIN_TILES");
			return e ? atol(e) : 512L;
		}();
		static const long min_tri = [] {
			const char* e = getenv("GPLE_GEMM_128_MIN_TILES_TRI");
			return e ? atol(e) : 4096L;
		}();
		// few 64-tiles: the split-k kernel (32 x 32 tiles, four waves on the k-range) gets four times as many workgroups going
		static const long max_splitk = [] {
			const char* e = getenv("GPLE_GEMM_SPLITK_MAX_TILES");
			return e ? atol(e) : 256L;
		}();
		if ((m / 64) * (n / 64) * batch <= max_splitk) return 32;
		if (m % 128 || n % 128) return 64;
		return (m / 128) * (n / 128) * batch >= (triangular ? min_tri : min_dense) ? 128 : 64;
	}

	// GPLE_GEMM_LOG=<file>: one line per launch — stream, tile, M N K batch krange lower_only and the algorithmic flops of the computed
	// tiles' k-ranges — in launch order.  probes/fit_kernel_table.py matches the lines with a rocprofv3 kernel trace of the same run
	// (per stream the order is the same) and prints TFLOP/s per kernel family.  Diagnostic only: nothing is logged without the variable.
	static void log_gemm(hipStream_t s, const GemmDesc& d, int tile)
	{
		static FILE* const f = [] {
			const char* path = getenv("GPLE_GEMM_LOG");
			return path ? fopen(path, "w") : nullptr;
		}();
		if (!f) return;
		// k-range of the tile at (m0, n0), as the kernels take it (rounded to whole tiles)
		double flops = 0.0;
		for (int m0 = 0; m0 < d.M; m0 += tile)
			for (int n0 = 0; n0 < d.N; n0 += tile)
			{
				if (d.lower_only && m0 + tile <= n0) continue;
				int k0 = 0, k1 = d.K;
				if (d.krange == K_GE_N) k0 = n0;
				else if (d.krange == K_LE_M) k1 = std::min(d.K, m0 + tile);
				else if (d.krange == K_GE_MAX_MN) k0 = std::max(m0, n0);
				if (k1 > k0) flops += 2.0 * tile * tile * (k1 - k0);
			}
		static std::mutex mu;
		std::lock_guard<std::mutex> lk(mu);
		fprintf(f, "%p %d %d %d %d %d %d %d %.0f\n", static_cast<void*>(s), tile, d.M, d.N, d.K, d.batch, d.krange, d.lower_only, flops * d.batch);
		fflush(f);
	}

	hipError_t launch_gemm(hipStream_t s, const GemmDesc& d, int tile)
	{
		if (tile == 32 && (d.c_trans || d.a_kmajor)) tile = 64; // operand layouts the split-k kernel is not instantiated for
		log_gemm(s, d, tile);
		if (tile == 32) return launch_splitk(s, d);
		return tile == 128 ? launch_tile<128>(s, d) : launch_tile<64>(s, d);
	}
} // namespace gple
