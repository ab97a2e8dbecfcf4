// gple_chol.hip — blocked fp64 Cholesky, triangular inverse and T^T T on gfx950.
//
// Replaces the reference's Eigen::LDLT + solve(Identity) (kernel.cpp:281-283; complex_kernel.cpp:264-266).
// The kernel matrices of this path are SPD by construction (sf^2 sn^2 ridge, opt.cpp:27), and Eigen's LDLT picks
// its pivots from the not-yet-updated diagonal, which is constant here — i.e. the reference itself runs
// unpivoted.  We therefore factor K = L L^T in 64-wide panels whose diagonal blocks leave their panel already inverted — by default all panels of
// an outer block in ONE launch (potrf_dag_kernel: a spine workgroup walking down the diagonal + tile tasks from a work queue, handed over by
// flags; GPLE_CHOL_SCHEME=step: right-looking, one launch per panel, potrf_step_kernel) —, form T = L^-1 in that same launch for matrices of one
// outer block, by block rows beside the factorisation for larger ones (GEMMs on a side stream; the diagonal part of a row block by its own
// launch or by a merge tree of MFMA GEMMs), and only on request W = K^-1 = T^T T.  A non-positive pivot does not abort: sqrt() yields
// NaN which propagates into every output, and *info records the first offending column (reference behaviour:
// LDLT::info() is never checked, NaN/Inf are clamped later by opt.cpp:420-431).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "gple_internal.h"

namespace gple
{
	namespace
	{
		constexpr int NB = CHOL_NB; // 64

		// 1/sqrt(d) to full precision: hardware estimate (about 23 bits) + one Halley step, r (1 + e/2 + 3 e^2 / 8) with e = 1 - d r^2
		// (cubic: the remaining error is 5/16 e^3; five dependent fp64 instructions where two Newton steps take seven — this sits
		// on the pivot chain of the factorisation, at 8 cycles per instruction)
		__device__ __forceinline__ double rsqrt_newton(double d)
		{
			const double r = __builtin_amdgcn_rsq(d);
			const double e = fma(-(d * r), r, 1.0);
			return fma(r, e * fma(0.375, e, 0.5), r);
		}

		// ---- diagonal-block kernel: L_jj = chol(A_jj), T_jj = L_jj^-1 and the panel rows below, one launch ----------------------
		// The panel step is the critical path of the fit.  Its first form (rounds 1-2: one sweep over the diagonal block and the rows
		// below it, the scaled column published through LDS with a barrier per column) took ~320 ns per column, 20 us per panel.
		// Here the 64 x 64 diagonal block is factored 16 columns at a time by ONE wave with a matrix row per lane and no LDS or
		// barrier on the chain: every 16-lane DPP row also carries a replica of the 16 x 16 diagonal tile, so that the pivot and
		// every multiplier L(base + j, k) is a row_newbcast of the replica — one v_fmac_f64_dpp per rank-1 entry instead of two
		// v_readlane and an FMA (the first version: 240 cycles per column, issue-bound).  Between sub-panels the trailing
		// 16 x 16 tiles are updated with MFMAs by all four waves.  The inverse is built in the same launch, off the chain: the
		// 16 x 16 diagonal inverses by substitution (one wave each, again DPP broadcasts, overlapped with the next sub-panel's
		// chain), the rest by block rows, T(i, b) = -T(i, i) sum_k L(i, k) T(k, b), as MFMA tile products on the idle waves —
		// everything of block row 3 except the last product is done before its diagonal tile is even factored.  The rows below
		// the diagonal block then need no substitution: L21 = A21 T_jj^T, 40 MFMAs per wave.
		constexpr int DLS = 68; // LDS row stride (doubles): 16-byte aligned rows, fragment reads of 16 rows x 4 k mostly conflict-free
		typedef double d4v __attribute__((ext_vector_type(4)));
		// acc(16 x 16, MFMA result layout) +-= X(16 x K) Y(K x 16); X row-major at xs, Y row-major (YT = false) or given as its
		// transpose (YT = true: Y[k][j] = ys[j * DLS + k]).  All operands are requested before the first MFMA; two accumulators.
		template <bool YT, bool NEG, int K>
		__device__ __forceinline__ d4v tile_mac(d4v acc, const double* xs, const double* ys, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
			double x[K / 4], y[K / 4];
#pragma unroll
			for (int q = 0; q < K / 4; ++q)
			{
				x[q] = xs[fr * DLS + 4 * q + fk];
				y[q] = YT ? ys[fr * DLS + 4 * q + fk] : ys[(4 * q + fk) * DLS + fr];
			}
			d4v a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
			for (int q = 0; q < K / 4; ++q)
			{
				const double xv = NEG ? -x[q] : x[q];
				if ((q & 1) && K > 4) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, y[q], a1, 0, 0, 0);
				else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, y[q], acc, 0, 0, 0);
			}
			return K > 4 ? acc + a1 : acc;
		}
		// Operands kept k-major (entry (row, k) at [k * DLS + row] — the way a column-major block arrives from global memory, so that
		// staging it is a copy with the lanes along LDS rows: no transposing stores, which cost an 8-way bank conflict each at this stride).
		// acc -= X Y^T over K = 64, X rows at xs, Y rows at ys
		__device__ __forceinline__ d4v tile_mac_kk_neg64(d4v acc, const double* xs, const double* ys, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
			double x[16], y[16];
#pragma unroll
			for (int q = 0; q < 16; ++q) x[q] = -xs[(4 * q + fk) * DLS + fr], y[q] = ys[(4 * q + fk) * DLS + fr];
			d4v a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
			for (int q = 0; q < 16; q += 2)
			{
				acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[q], y[q], acc, 0, 0, 0);
				a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[q + 1], y[q + 1], a1, 0, 0, 0);
			}
			return acc + a1;
		}
		// two products with a common Y: acc0 -= X0 Y^T, acc1 -= X1 Y^T; one round of operand reads, four MFMA chains
		__device__ __forceinline__ void tile_mac2_kk_neg64(d4v& acc0, d4v& acc1, const double* xs0, const double* xs1, const double* ys, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
			double x0[16], x1[16], y[16];
#pragma unroll
			for (int q = 0; q < 16; ++q)
			{
				x0[q] = -xs0[(4 * q + fk) * DLS + fr];
				x1[q] = -xs1[(4 * q + fk) * DLS + fr];
				y[q] = ys[(4 * q + fk) * DLS + fr];
			}
			d4v b0 = {0.0, 0.0, 0.0, 0.0}, b1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
			for (int q = 0; q < 16; q += 2)
			{
				acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], y[q], acc0, 0, 0, 0);
				acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], y[q], acc1, 0, 0, 0);
				b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q + 1], y[q + 1], b0, 0, 0, 0);
				b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q + 1], y[q + 1], b1, 0, 0, 0);
			}
			acc0 += b0, acc1 += b1;
		}
		// a tile held transposed (accumulator element q of lane (fk, fr) = entry (row fr, column fk + 4 q)) goes to its place in a row-major LDS block
		__device__ __forceinline__ d4v tile_load_t(const double* src, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
			d4v a;
#pragma unroll
			for (int r = 0; r < 4; ++r) a[r] = src[fr * DLS + fk + 4 * r];
			return a;
		}
		__device__ __forceinline__ void tile_store_t(double* dst, d4v a, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
			for (int r = 0; r < 4; ++r) dst[fr * DLS + fk + 4 * r] = a[r];
		}
		__device__ __forceinline__ d4v tile_load(const double* src, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
			d4v a;
#pragma unroll
			for (int r = 0; r < 4; ++r) a[r] = src[(fk + 4 * r) * DLS + fr];
			return a;
		}
		__device__ __forceinline__ void tile_store(double* dst, d4v a, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
			for (int r = 0; r < 4; ++r) dst[(fk + 4 * r) * DLS + fr] = a[r];
		}
		// DPP helpers (gfx90a+: 64-bit DPP exists for v_mov_b64 / v_fmac_f64 with row_newbcast only).  A VGPR written by a VALU
		// instruction needs two wait states before a DPP instruction may read it as its source; hipcc does not look into inline
		// asm, so values that will be broadcast are produced by dpp_mul (asm) and fenced by dpp_fence (s_nop 1).  volatile asm
		// statements keep their program order, which is all the ordering this relies on.
		template <int LANE>
		__device__ __forceinline__ double dpp_bcast(double v)
		{
			double r;
			asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(LANE));
			return r;
		}
		template <int LANE> // acc += bcast_LANE(src) * mul
		__device__ __forceinline__ void dpp_fmac(double& acc, double src, double mul)
		{
			asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(LANE));
		}
		__device__ __forceinline__ double dpp_mul(double a, double b)
		{
			double r;
			asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
			return r;
		}
		__device__ __forceinline__ void dpp_fence() { asm volatile("s_nop 1"); }
		__device__ __forceinline__ void dpp_fence(double& v) { asm volatile("s_nop 1" : "+v"(v)); }

		// columns 16 SP .. 16 SP + 15 of the block.  lane -> matrix row base + lane in p[] (lanes past the last row repeat row 63 and
		// store nothing) and row base + (lane & 15) of the diagonal tile in q[] (the replica of its DPP row).  Nothing is masked:
		// entries above the diagonal carry garbage that no valid entry ever reads, the diagonal of L is never needed (1 / L_kk
		// goes to rinv) and positivity is checked on the reciprocals after the sweep.  Software-pipelined by hand: iteration k
		// runs the 1/sqrt chain of column k while the rank-1 update of column k - 1 is applied to the columns right of k; only
		// the update of column k + 1 by column k sits between two pivots.
		// INV_LAST (sub-panels 1-3, which leave lanes without a row): the sweep also yields the INVERSE of its 16 x 16 diagonal tile.  Rows below a
		// diagonal tile leave a sweep as A21 L^-T; with the identity in the free lanes that is L^-T, whose row i is column i of T = L^-1: written to
		// tile (SP, SP) of TIout (row-major, zeros above the diagonal) — instead of a substitution of 16 dependent steps (diag_inv16, 1.2 us) after the
		// sweep, which for the last sub-panel sat on the panel's critical path.
		template <int SP, bool INV_LAST = false>
		__device__ __forceinline__ void diag_chain(double* S, double* rinv, int lane, int& first_bad, double* TIout = nullptr)
		{
			static_assert(!INV_LAST || SP >= 1, "the first sub-panel has no free lanes below its rows");
			constexpr int FREE0 = NB - 16 * SP; // first lane without a row of its own (sub-panel SP has 64 - 16 SP rows from its diagonal tile down)
			constexpr int base = 16 * SP;
			const int own = min(base + lane, NB - 1);
			double p[16], q[16], rr[16];
			double* const prow = S + own * DLS + base;
			const double* const qrow = S + (base + (lane & 15)) * DLS + base;
#pragma unroll
			for (int k = 0; k < 16; ++k) p[k] = prow[k], q[k] = qrow[k];
			if constexpr (INV_LAST)
				if (lane >= FREE0)
#pragma unroll
					for (int k = 0; k < 16; ++k) p[k] = k == (lane & 15) ? 1.0 : 0.0;
			// the broadcast carries -L(base + j, k); the per-lane factor is the lane's own multiplier (l for p, lq for q)
			double nlq = 0.0, lq = 0.0, l = 0.0;
			constexpr bool ROWS_BELOW = SP < 3 || INV_LAST; // the last sub-panel is its diagonal tile only: p == q in the one DPP row that counts
			[&]<int... Ks>(std::integer_sequence<int, Ks...>)
			{
				(
					[&] {
						constexpr int k = Ks;
						dpp_fence(q[k]); // q[k] was last written by the fmac at the end of the previous iteration
						const double d = dpp_bcast<k>(q[k]);
						const double r = rsqrt_newton(d);
						rr[k] = r;
						if constexpr (k > 0)
							[&]<int... Js>(std::integer_sequence<int, Js...>)
							{
								((Js > k ? (dpp_fmac<Js>(q[Js], nlq, lq), (ROWS_BELOW ? dpp_fmac<Js>(p[Js], nlq, l) : (void)0)) : (void)0), ...);
							}
							(std::make_integer_sequence<int, 16>{});
						nlq = dpp_mul(q[k], -r);
						lq = q[k] * r;
						if constexpr (ROWS_BELOW) l = p[k] * r, p[k] = l;
						else q[k] = lq;
						dpp_fence();
						if constexpr (k + 1 < 16)
						{
							dpp_fmac<k + 1>(q[k + 1], nlq, lq);
							if constexpr (ROWS_BELOW) dpp_fmac<k + 1>(p[k + 1], nlq, l);
						}
						__builtin_amdgcn_sched_barrier(0);
					}(),
					...);
			}
			(std::make_integer_sequence<int, 16>{});
			if constexpr (!ROWS_BELOW)
#pragma unroll
				for (int k = 0; k < 16; ++k) p[k] = q[k]; // lanes 0..15 hold rows 48..63 either way
			if (base + lane < NB)
#pragma unroll
				for (int k = 0; k < 16; ++k) prow[k] = p[k];
			if constexpr (INV_LAST)
				if (lane >= FREE0 && lane < FREE0 + 16)
#pragma unroll
					for (int k = 0; k < 16; ++k) TIout[(base + k) * DLS + base + (lane & 15)] = p[k]; // T(k, i) = (L^-T)(i, k); zero for k < i
			int fb = 0;
#pragma unroll
			for (int k = 15; k >= 0; --k) fb = (rr[k] > 0.0 && rr[k] < __builtin_inf()) ? fb : base + k + 1;
			fb = __builtin_amdgcn_readfirstlane(fb);
			first_bad = first_bad == 0 ? fb : first_bad;
			if (lane == 0)
#pragma unroll
				for (int k = 0; k < 16; ++k) rinv[base + k] = rr[k];
		}
		// TI(b, b) = L(b, b)^-1 (16 x 16): lane & 15 = row i of the tile, x[] = row i of the inverse.  acc_i[c] collects
		// sum_{c <= k < i} L(i, k) X(k, c); row k of the inverse, X(k, c) = -acc_k[c] / L(k, k), is broadcast from lane k at step k.
		__device__ __forceinline__ void diag_inv16(const double* S, const double* rinv, double* TI, int b, int lane)
		{
			const int i = lane & 15;
			const double* const Lrow = S + (b * 16 + i) * DLS + b * 16;
			double Lm[16], acc[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) Lm[k] = i > k ? Lrow[k] : 0.0, acc[k] = 0.0;
			const double ri = rinv[b * 16 + i], nri = -ri;
			[&]<int... Ks>(std::integer_sequence<int, Ks...>)
			{
				(
					[&] {
						constexpr int k = Ks;
						double tmp[16];
#pragma unroll
						for (int c = 0; c < k; ++c) tmp[c] = dpp_mul(nri, acc[c]);
						tmp[k] = ri;
						dpp_fence();
						[&]<int... Cs>(std::integer_sequence<int, Cs...>)
						{
							((Cs <= k ? dpp_fmac<k>(acc[Cs], tmp[Cs], Lm[k]) : (void)0), ...);
						}
						(std::make_integer_sequence<int, 16>{});
					}(),
					...);
			}
			(std::make_integer_sequence<int, 16>{});
			if (lane < 16)
			{
				double* const Trow = TI + (b * 16 + i) * DLS + b * 16;
#pragma unroll
				for (int c = 0; c < 16; ++c) Trow[c] = c < i ? nri * acc[c] : (c == i ? ri : 0.0);
			}
		}
		// A points at block (j0, j0) of the working matrix (column-major, lower part valid); T_jj (ldt) receives inv(L_jj) as a full
		// 64 x 64 block (zeros above the diagonal).  L_jj itself is not kept: nothing downstream reads a diagonal block of the factor.
		// Every workgroup factors and inverts the diagonal block for itself (no workgroup ever waits for another) and then turns
		// its own 64 rows of the panel below, P = A(j0 + 64 + 64 b .., j0 .. j0 + 63), into L21 = P T_jj^T in place; the rows are
		// requested from HBM before the factorisation starts.  Workgroup 0 also stores T_jj.
		template <bool PROBE>
		__global__ void __launch_bounds__(256) potrf_diag_kernel(const double* __restrict__ A, long lda, double* __restrict__ T, long ldt, int* __restrict__ info,
			int j0, long long* __restrict__ stamps, double* __restrict__ P, int below, double* __restrict__ uvec)
		{
			int stamp_i = 0;
			auto stamp = [&]() {
				if constexpr (PROBE)
					if (threadIdx.x == 0) stamps[stamp_i++] = static_cast<long long>(__builtin_readcyclecounter());
			};
			stamp();
			__shared__ __attribute__((aligned(16))) double S[NB * DLS];  // A_jj -> L_jj (strictly lower tiles); later the panel rows
			__shared__ __attribute__((aligned(16))) double TI[NB * DLS]; // T_jj; tile (i, b), b < i, holds V(i, b) until T(i, b) replaces it
			__shared__ double rinv[NB];
			const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
			{
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 16; ++q)
				{
					const int c = (t >> 6) + 4 * q;
					S[r * DLS + c] = A[r + static_cast<long>(c) * lda];
				}
			}
			const bool has_rows = static_cast<int>(blockIdx.x) * NB < below; // uniform
			double* const Pb = P + static_cast<long>(blockIdx.x) * NB;
			double prow[16];
			if (has_rows)
			{
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 16; ++q) prow[q] = Pb[r + static_cast<long>((t >> 6) + 4 * q) * lda];
			}
			int first_bad = 0;
			auto Sx = [&](int ti, int tj) { return S + ti * 16 * DLS + tj * 16; };
			auto Tx = [&](int ti, int tj) { return TI + ti * 16 * DLS + tj * 16; };
			// trailing tile (ti, tj) -= L(ti, sp) L(tj, sp)^T
			auto upd = [&](int ti, int tj, int sp) {
				d4v acc = tile_load(Sx(ti, tj), lane);
				acc = tile_mac<true, true, 16>(acc, Sx(ti, sp), Sx(tj, sp), lane);
				tile_store(Sx(ti, tj), acc, lane);
			};
			// V(i, b) (+)= L(i, k) T(k, b), kept in TI's tile (i, b)
			auto v_acc = [&](int i, int b, int k, bool first) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				if (!first) acc = tile_load(Tx(i, b), lane);
				acc = tile_mac<false, false, 16>(acc, Sx(i, k), Tx(k, b), lane);
				tile_store(Tx(i, b), acc, lane);
			};
			// T(i, b) = -T(i, i) V(i, b), in place
			auto t_fin = [&](int i, int b) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				acc = tile_mac<false, true, 16>(acc, Tx(i, i), Tx(i, b), lane);
				tile_store(Tx(i, b), acc, lane);
			};
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<0>(S, rinv, lane, first_bad);
			__syncthreads();
			stamp();
			if (w == 0) upd(1, 1, 0);
			else if (w == 1) upd(2, 1, 0);
			else if (w == 2) upd(3, 1, 0);
			else upd(2, 2, 0);
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<1>(S, rinv, lane, first_bad);
			else if (w == 1) diag_inv16(S, rinv, TI, 0, lane);
			else if (w == 2) upd(3, 2, 0);
			else upd(3, 3, 0);
			__syncthreads();
			stamp();
			if (w == 0) upd(2, 2, 1);
			else if (w == 1) upd(3, 2, 1);
			else if (w == 2) upd(3, 3, 1);
			else v_acc(1, 0, 0, true);
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<2>(S, rinv, lane, first_bad);
			else if (w == 1) diag_inv16(S, rinv, TI, 1, lane);
			else if (w == 2) v_acc(2, 0, 0, true);
			else v_acc(3, 0, 0, true);
			__syncthreads();
			stamp();
			if (w == 0) upd(3, 3, 2);
			else if (w == 1) t_fin(1, 0);
			else if (w == 2) v_acc(2, 1, 1, true);
			else v_acc(3, 1, 1, true);
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<3>(S, rinv, lane, first_bad);
			else if (w == 1) diag_inv16(S, rinv, TI, 2, lane);
			else if (w == 2) v_acc(2, 0, 1, false);
			else v_acc(3, 0, 1, false);
			__syncthreads();
			stamp();
			if (w == 0) diag_inv16(S, rinv, TI, 3, lane);
			else if (w == 1) t_fin(2, 0), v_acc(3, 0, 2, false); // LDS operations of one wave complete in order
			else if (w == 2) t_fin(2, 1), v_acc(3, 1, 2, false);
			else v_acc(3, 2, 2, true);
			__syncthreads();
			stamp();
			if (w < 3) t_fin(3, w);
			if (has_rows) // L_jj is dead since the last barrier: the panel rows take its place
			{
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 16; ++q) S[r * DLS + (t >> 6) + 4 * q] = prow[q];
			}
			__syncthreads();
			stamp();
			if (blockIdx.x == 0)
			{
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 16; ++q)
				{
					const int c = (t >> 6) + 4 * q;
					T[r + static_cast<long>(c) * ldt] = c <= r ? TI[r * DLS + c] : 0.0;
					if (j0 & NB) T[r - NB + static_cast<long>(c) * ldt] = 0.0; // see potrf_step_kernel
				}
				if (first_bad != 0 && t == 0) atomicCAS(info, 0, j0 + first_bad); // info starts at 0
			}
			stamp();
			if (has_rows)
			{
				// L21(16 w .. 16 w + 15, 16 j ..) = sum_{k <= 16 j + 15} P(., k) T_jj(16 j .., k): wave w owns row tile w of the 64 rows
				d4v out[4];
				out[0] = tile_mac<true, false, 16>((d4v){0.0, 0.0, 0.0, 0.0}, S + w * 16 * DLS, TI, lane);
				out[1] = tile_mac<true, false, 32>((d4v){0.0, 0.0, 0.0, 0.0}, S + w * 16 * DLS, TI + 16 * DLS, lane);
				out[2] = tile_mac<true, false, 48>((d4v){0.0, 0.0, 0.0, 0.0}, S + w * 16 * DLS, TI + 32 * DLS, lane);
				out[3] = tile_mac<true, false, 64>((d4v){0.0, 0.0, 0.0, 0.0}, S + w * 16 * DLS, TI + 48 * DLS, lane);
				// the wave's own rows of S are dead once its MFMAs have read them: reuse them to transpose the result for coalesced stores
#pragma unroll
				for (int j = 0; j < 4; ++j) tile_store(S + w * 16 * DLS + j * 16, out[j], lane);
				__syncthreads();
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 16; ++q)
				{
					const int c = (t >> 6) + 4 * q;
					Pb[r + static_cast<long>(c) * lda] = S[r * DLS + c];
				}
				// the label row carried below the matrix (chol_inverse_factor, `uvec`): its factor entries are u = L^-1 y, 64 per panel
				if (uvec != nullptr && blockIdx.x == gridDim.x - 1 && t < NB) uvec[j0 + t] = S[t];
			}
			stamp();
		}

		// ---- the whole panel step in one launch ----------------------------------------------------------------------------------
		// potrf_diag_kernel followed by the rank-64 update of the block column's strip is two dependent launches per panel: the
		// update (5 us, latency-bound) sits on the critical path only because the NEXT panel's own 64 columns are among what it
		// updates.  Here the panel applies the previous panel's update to its own columns itself (left-looking by exactly one
		// step: diagonal block before the chain starts, its 64 rows below on four extra waves while wave 0 runs the first two
		// chains), and the rest of that update — every column right of this panel — is done by further workgroups of the SAME
		// launch, next to the panel workgroups instead of in front of them.  One launch per panel; no workgroup waits for another:
		//   workgroups [0, ndt):      panel j0 (as potrf_diag_kernel), `pend`: first subtract L_prev(rows) L_prev(diag rows)^T
		//   workgroups [ndt, ndt + .): four 64 x 64 tiles (r >= c) each of  A(c0 + 64 r .., c0 + 64 c ..) -= L_prev(rows r) L_prev(rows c)^T,
		//                              c0 = j0 + 64, L_prev = A(., j0 - 64 .. j0 - 1); straight from L2 into MFMA fragments, no LDS
		// 8 waves: waves 4-7 hold the panel rows as MFMA accumulators from the start (no register copy of them on the chain wave)
		// and share the final product L21 = P T_jj^T with waves 0-3.
		// Schedule of the pending update of the panel rows (16 tile products of K = 64, tile k = 4 * row tile + column tile).  On gfx950 the
		// fp64 MFMA runs on the SIMD's fp64 vector ALUs (matrix and vector fp64 peaks are equal): an MFMA on the chain wave's SIMD stalls
		// the chain (measured: chain stage 3.8k -> 5.8k cycles), and a SIMD gets through about two such products (2 x 1024 MFMA cycles +
		// LDS round trips) per chain stage of 3.7k.  So wave 4 (SIMD 0, like the chain wave) never multiplies, the first chain stage takes
		// the six pending diagonal tiles on waves 1-3, waves 6 / 7 (SIMDs 2 / 3) do two row products in each of the other three chain
		// stages, and the stage of the last 16 x 16 inverse takes the remaining four (wave 5 two: SIMD 1 carries the other inverses
		// until then).  RU_TILE[wave - 5][slot] = tile, RU_STAGE = 1..3: chain stage, 4: the stage after.
		constexpr int RU_TILE[3][8] = {{0, 1, -1, -1, -1, -1, -1, -1}, {2, 3, 4, 5, 6, 7, 8, -1}, {9, 10, 12, 13, 14, 15, 11, -1}};
		constexpr int RU_STAGE[3][8] = {{4, 4, -1, -1, -1, -1, -1, -1}, {1, 1, 2, 2, 3, 3, 4, -1}, {1, 1, 2, 2, 3, 3, 4, -1}};
		constexpr int ru_find(int wv, int st, int nth) // slot of the nth product wave 5 + wv does in stage st, or -1
		{
			int c = 0;
			for (int sl = 0; sl < 8; ++sl)
				if (RU_TILE[wv][sl] >= 0 && RU_STAGE[wv][sl] == st)
				{
					if (c == nth) return sl;
					++c;
				}
			return -1;
		}
		template <int WV, typename F>
		__device__ __forceinline__ void ru_slots(F&& f)
		{
			[&]<int... Sl>(std::integer_sequence<int, Sl...>) { ((RU_TILE[WV][Sl] >= 0 ? f(std::integral_constant<int, Sl>{}) : (void)0), ...); }
			(std::make_integer_sequence<int, 8>{});
		}

		template <bool PROBE>
		__global__ void __launch_bounds__(512) potrf_step_kernel(double* __restrict__ A, long lda, double* __restrict__ T, long ldt, int* __restrict__ info,
			int j0, int below, int ndt, int pend, int sy_nc, int sy_nr, double* __restrict__ uvec, long long* __restrict__ stamps)
		{
			int stamp_i = 0;
			auto stamp = [&]() {
				if constexpr (PROBE)
					if (threadIdx.x == 0 && blockIdx.x == 0) stamps[stamp_i++] = static_cast<long long>(__builtin_readcyclecounter());
			};
			stamp();
			const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
			const int fr = lane & 15, fk = lane >> 4;
			if (static_cast<int>(blockIdx.x) >= ndt)
			{
				// four 64 x 64 tiles per workgroup, a wave pair each (tile id -> column block c, row block r >= c of the strip); a wave owns
				// 2 x 4 of its tile's 16 x 16 blocks: 6 operand fragments per 8 MFMAs, K in two halves of 32
				int id = 4 * (static_cast<int>(blockIdx.x) - ndt) + (w >> 1), c = 0;
				if (id >= sy_nr * sy_nc - sy_nc * (sy_nc - 1) / 2) return; // wave-uniform; no barrier on this path
				while (id >= sy_nr - c) id -= sy_nr - c, ++c;
				const int r = c + id;
				const long c0 = j0 + NB;
				const double* __restrict__ Lr = A + (c0 + static_cast<long>(r) * NB) + static_cast<long>(j0 - NB) * lda; // rows of the result tile
				const double* __restrict__ Lc = A + (c0 + static_cast<long>(c) * NB) + static_cast<long>(j0 - NB) * lda; // its columns, as rows of L_prev
				double* __restrict__ C = A + (c0 + static_cast<long>(r) * NB) + (c0 + static_cast<long>(c) * NB) * lda;
				// accumulator element [i = fk + 4 q][j = fr] = C(row 16 a + j, column 16 b + i): the lanes of a quarter run along a column of C
				const int a0 = 2 * (w & 1);
				d4v acc[2][4];
#pragma unroll
				for (int u = 0; u < 2; ++u)
#pragma unroll
					for (int b = 0; b < 4; ++b)
#pragma unroll
						for (int q = 0; q < 4; ++q) acc[u][b][q] = C[(16 * (a0 + u) + fr) + static_cast<long>(16 * b + fk + 4 * q) * lda];
#pragma unroll
				for (int kh = 0; kh < 2; ++kh)
				{
					double y[2][8], x[4][8];
#pragma unroll
					for (int q = 0; q < 8; ++q)
					{
						const long kcol = static_cast<long>(32 * kh + 4 * q + fk) * lda;
#pragma unroll
						for (int u = 0; u < 2; ++u) y[u][q] = Lr[(16 * (a0 + u) + fr) + kcol];
#pragma unroll
						for (int b = 0; b < 4; ++b) x[b][q] = -Lc[(16 * b + fr) + kcol];
					}
#pragma unroll
					for (int q = 0; q < 8; ++q)
#pragma unroll
						for (int u = 0; u < 2; ++u)
#pragma unroll
							for (int b = 0; b < 4; ++b) acc[u][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[b][q], y[u][q], acc[u][b], 0, 0, 0);
				}
#pragma unroll
				for (int u = 0; u < 2; ++u)
#pragma unroll
					for (int b = 0; b < 4; ++b)
#pragma unroll
						for (int q = 0; q < 4; ++q) C[(16 * (a0 + u) + fr) + static_cast<long>(16 * b + fk + 4 * q) * lda] = acc[u][b][q];
				return;
			}
			__shared__ __attribute__((aligned(16))) double S[NB * DLS];  // A_jj -> L_jj (strictly lower tiles); later the panel rows
			__shared__ __attribute__((aligned(16))) double TI[NB * DLS]; // T_jj; tile (i, b), b < i, holds V(i, b) until T(i, b) replaces it
			__shared__ __attribute__((aligned(16))) double U[NB * DLS];  // pend: this workgroup's 64 rows of L_prev, k-major; at the end the result tiles on their way out
			__shared__ __attribute__((aligned(16))) double D[NB * DLS];  // pend: the diagonal rows of L_prev, k-major
			__shared__ double rinv[NB];
			const double* __restrict__ Ajj = A + j0 + static_cast<long>(j0) * lda;
			const bool has_rows = static_cast<int>(blockIdx.x) * NB < below; // uniform
			double* const Pb = A + (j0 + NB + static_cast<long>(blockIdx.x) * NB) + static_cast<long>(j0) * lda;
			{
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 8; ++q)
				{
					const int c = (t >> 6) + 8 * q;
					S[r * DLS + c] = Ajj[r + static_cast<long>(c) * lda];
				}
				if (pend)
				{
					const double* __restrict__ Ld = Ajj - static_cast<long>(NB) * lda;
#pragma unroll
					for (int q = 0; q < 8; ++q)
					{
						const int c = (t >> 6) + 8 * q;
						D[c * DLS + r] = Ld[r + static_cast<long>(c) * lda];
					}
				}
			}
			d4v pacc[8];
			int first_bad = 0;
			auto Sx = [&](int ti, int tj) { return S + ti * 16 * DLS + tj * 16; };
			auto Tx = [&](int ti, int tj) { return TI + ti * 16 * DLS + tj * 16; };
			auto upd = [&](int ti, int tj, int sp) {
				d4v acc = tile_load(Sx(ti, tj), lane);
				acc = tile_mac<true, true, 16>(acc, Sx(ti, sp), Sx(tj, sp), lane);
				tile_store(Sx(ti, tj), acc, lane);
			};
			// the previous panel's update of diagonal tile (ti, tj)
			auto pend_upd = [&](int ti, int tj) {
				d4v acc = tile_load(Sx(ti, tj), lane);
				acc = tile_mac_kk_neg64(acc, D + ti * 16, D + tj * 16, lane);
				tile_store(Sx(ti, tj), acc, lane);
			};
			// tiles (a, c) and (b, c) of the diagonal block at once; first_t: the first one is (c, a) held transposed instead
			auto pend_upd2 = [&](int a, int c0, int b, int c, bool first_t) {
				d4v acc0 = first_t ? tile_load_t(Sx(a, c0), lane) : tile_load(Sx(a, c0), lane);
				d4v acc1 = tile_load(Sx(b, c), lane);
				tile_mac2_kk_neg64(acc0, acc1, D + (first_t ? c0 : a) * 16, D + b * 16, D + (first_t ? a : c) * 16, lane);
				if (first_t) tile_store_t(Sx(a, c0), acc0, lane);
				else tile_store(Sx(a, c0), acc0, lane);
				tile_store(Sx(b, c), acc1, lane);
			};
			// the panel rows: pick-up, the pending update stage by stage, hand-over to S (waves 5-7, RU_TILE / RU_STAGE)
			// (waves 5-7 hold their tiles TRANSPOSED — lanes run along the rows of the panel, so that the loads below are 128-byte segments;
			// an 8-byte gather costs the address unit ~60 cycles per instruction — and the products are formed as D_j U_i^T accordingly)
			auto rows_pick = [&]<int WV>(std::integral_constant<int, WV>) {
				ru_slots<WV>([&](auto sl) {
					constexpr int k = RU_TILE[WV][decltype(sl)::value];
#pragma unroll
					for (int q = 0; q < 4; ++q) pacc[decltype(sl)::value][q] = Pb[(16 * (k >> 2) + fr) + static_cast<long>(16 * (k & 3) + fk + 4 * q) * lda];
				});
			};
			auto rows_mac = [&]<int WV, int ST>(std::integral_constant<int, WV>, std::integral_constant<int, ST>) {
				constexpr int s0 = ru_find(WV, ST, 0), s1 = ru_find(WV, ST, 1);
				if constexpr (s0 >= 0 && s1 >= 0)
				{
					constexpr int k0 = RU_TILE[WV][s0], k1 = RU_TILE[WV][s1];
					static_assert((k0 >> 2) == (k1 >> 2), "the two products of a stage share their row tile");
					tile_mac2_kk_neg64(pacc[s0], pacc[s1], D + (k0 & 3) * 16, D + (k1 & 3) * 16, U + (k0 >> 2) * 16, lane);
				}
				else if constexpr (s0 >= 0)
				{
					constexpr int k0 = RU_TILE[WV][s0];
					pacc[s0] = tile_mac_kk_neg64(pacc[s0], D + (k0 & 3) * 16, U + (k0 >> 2) * 16, lane);
				}
			};
			auto rows_put = [&]<int WV>(std::integral_constant<int, WV>) {
				ru_slots<WV>([&](auto sl) {
					constexpr int k = RU_TILE[WV][decltype(sl)::value];
					tile_store_t(Sx(k >> 2, k & 3), pacc[decltype(sl)::value], lane);
				});
			};
			auto rows_stage = [&]<int ST>(std::integral_constant<int, ST> st) {
				if (w == 5) rows_mac(std::integral_constant<int, 0>{}, st);
				else if (w == 6) rows_mac(std::integral_constant<int, 1>{}, st);
				else if (w == 7) rows_mac(std::integral_constant<int, 2>{}, st);
			};
			auto v_acc = [&](int i, int b, int k, bool first) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				if (!first) acc = tile_load(Tx(i, b), lane);
				acc = tile_mac<false, false, 16>(acc, Sx(i, k), Tx(k, b), lane);
				tile_store(Tx(i, b), acc, lane);
			};
			auto t_fin = [&](int i, int b) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				acc = tile_mac<false, true, 16>(acc, Tx(i, i), Tx(i, b), lane);
				tile_store(Tx(i, b), acc, lane);
			};
			__syncthreads();
			stamp();
			// what is not needed before the second chain is requested here and arrives during the first: the panel rows (waves 5-7, straight
			// into their accumulators) and, pending, this workgroup's rows of L_prev (waves 4-7 -> U, published by the barrier after the chain)
			const int tt = t & 255;
			double uv[16];
			if (has_rows && pend && w >= 4)
			{
				const double* __restrict__ Lb = Pb - static_cast<long>(NB) * lda;
#pragma unroll
				for (int q = 0; q < 16; ++q) uv[q] = Lb[(tt & 63) + static_cast<long>((tt >> 6) + 4 * q) * lda];
			}
			if (pend)
			{
				// what the first chain reads: tile column 0; the other six lower tiles follow during the first chain (SIMDs 1-3).
				// The barrier orders LDS only (s_waitcnt lgkmcnt(0)): __syncthreads() would also wait for the loads just issued.
				if (w < 4) pend_upd(w, 0);
				asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
				stamp();
			}
			if (has_rows)
			{
				if (w == 5) rows_pick(std::integral_constant<int, 0>{});
				else if (w == 6) rows_pick(std::integral_constant<int, 1>{});
				else if (w == 7) rows_pick(std::integral_constant<int, 2>{});
				if (pend && w >= 4)
#pragma unroll
					for (int q = 0; q < 16; ++q) U[((tt >> 6) + 4 * q) * DLS + (tt & 63)] = uv[q];
			}
			if (w == 0) diag_chain<0>(S, rinv, lane, first_bad);
			else if (pend)
			{
				// two tiles per wave with a common second factor; (3, 1) is formed transposed (D_1 D_3^T) to share D_3 with (3, 3)
				if (w == 1) pend_upd2(1, 1, 2, 1, false);
				else if (w == 2) pend_upd2(2, 2, 3, 2, false);
				else if (w == 3) pend_upd2(3, 1, 3, 3, true);
			}
			__syncthreads();
			stamp();
			if (w == 0) upd(1, 1, 0);
			else if (w == 1) upd(2, 1, 0);
			else if (w == 2) upd(3, 1, 0);
			else if (w == 3) upd(2, 2, 0);
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<1>(S, rinv, lane, first_bad);
			else if (w == 1) diag_inv16(S, rinv, TI, 0, lane);
			else if (w == 2) upd(3, 2, 0);
			else if (w == 3) upd(3, 3, 0);
			else if (pend && has_rows) rows_stage(std::integral_constant<int, 1>{});
			__syncthreads();
			stamp();
			if (w == 0) upd(2, 2, 1);
			else if (w == 1) upd(3, 2, 1);
			else if (w == 2) upd(3, 3, 1);
			else if (w == 3) v_acc(1, 0, 0, true);
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<2>(S, rinv, lane, first_bad);
			else if (w == 1) diag_inv16(S, rinv, TI, 1, lane);
			else if (w == 2) v_acc(2, 0, 0, true);
			else if (w == 3) v_acc(3, 0, 0, true);
			else if (pend && has_rows) rows_stage(std::integral_constant<int, 2>{});
			__syncthreads();
			stamp();
			if (w == 0) upd(3, 3, 2);
			else if (w == 1) t_fin(1, 0);
			else if (w == 2) v_acc(2, 1, 1, true);
			else if (w == 3) v_acc(3, 1, 1, true);
			__syncthreads();
			stamp();
			if (w == 0) diag_chain<3>(S, rinv, lane, first_bad);
			else if (w == 1) diag_inv16(S, rinv, TI, 2, lane);
			else if (w == 2) v_acc(2, 0, 1, false);
			else if (w == 3) v_acc(3, 0, 1, false);
			else if (pend && has_rows) rows_stage(std::integral_constant<int, 3>{});
			__syncthreads();
			stamp();
			if (w == 0) diag_inv16(S, rinv, TI, 3, lane);
			else if (w == 1) t_fin(2, 0), v_acc(3, 0, 2, false); // LDS operations of one wave complete in order
			else if (w == 2) t_fin(2, 1), v_acc(3, 1, 2, false);
			else if (w == 3) v_acc(3, 2, 2, true);
			if (w >= 5 && pend && has_rows) rows_stage(std::integral_constant<int, 4>{});
			__syncthreads();
			stamp();
			if (w < 3) t_fin(3, w);
			if (has_rows) // L_jj is dead since the last barrier: the panel rows take its place
			{
				if (w == 5) rows_put(std::integral_constant<int, 0>{});
				else if (w == 6) rows_put(std::integral_constant<int, 1>{});
				else if (w == 7) rows_put(std::integral_constant<int, 2>{});
			}
			__syncthreads();
			stamp();
			if (blockIdx.x == 0)
			{
				const int r = t & 63;
				double* __restrict__ Tjj = T + j0 + static_cast<long>(j0) * ldt;
#pragma unroll
				for (int q = 0; q < 8; ++q)
				{
					const int c = (t >> 6) + 8 * q;
					Tjj[r + static_cast<long>(c) * ldt] = c <= r ? TI[r * DLS + c] : 0.0;
					// the one block above the diagonal that a reader may touch: 128-tile GEMMs with a triangular k-range start at their
					// 128-aligned diagonal tile, which takes in the block above the second 64-block of the pair.  Nothing else above the
					// diagonal blocks of T is ever written or read.
					if (j0 & NB) Tjj[r - NB + static_cast<long>(c) * ldt] = 0.0;
				}
				if (first_bad != 0 && t == 0) atomicCAS(info, 0, j0 + first_bad); // info starts at 0; first_bad is wave 0's
			}
			if (has_rows)
			{
				// L21(row tile, 16 j ..) = sum_{k <= 16 j + 15} P(., k) T_jj(16 j .., k): waves w and w + 4 share row tile w & 3 — column tiles
				// {0, 3} and {1, 2}, 20 MFMAs each; the results meet in U (dead by now), transposed for coalesced stores
				const int rt = w & 3;
				if (w < 4)
				{
					const d4v o0 = tile_mac<true, false, 16>((d4v){0.0, 0.0, 0.0, 0.0}, S + rt * 16 * DLS, TI, lane);
					const d4v o3 = tile_mac<true, false, 64>((d4v){0.0, 0.0, 0.0, 0.0}, S + rt * 16 * DLS, TI + 48 * DLS, lane);
					tile_store(U + rt * 16 * DLS, o0, lane);
					tile_store(U + rt * 16 * DLS + 48, o3, lane);
				}
				else
				{
					const d4v o1 = tile_mac<true, false, 32>((d4v){0.0, 0.0, 0.0, 0.0}, S + rt * 16 * DLS, TI + 16 * DLS, lane);
					const d4v o2 = tile_mac<true, false, 48>((d4v){0.0, 0.0, 0.0, 0.0}, S + rt * 16 * DLS, TI + 32 * DLS, lane);
					tile_store(U + rt * 16 * DLS + 16, o1, lane);
					tile_store(U + rt * 16 * DLS + 32, o2, lane);
				}
				__syncthreads();
			stamp();
				const int r = t & 63;
#pragma unroll
				for (int q = 0; q < 8; ++q)
				{
					const int c = (t >> 6) + 8 * q;
					Pb[r + static_cast<long>(c) * lda] = U[r * DLS + c];
				}
				// the label row carried below the matrix (chol_inverse_factor, `uvec`): its factor entries are u = L^-1 y, 64 per panel
				if (uvec != nullptr && static_cast<int>(blockIdx.x) == ndt - 1 && t < NB) uvec[j0 + t] = U[t];
			}
			stamp();
		}

		// ---- the panels of an outer block without a launch between them -------------------------------------------------------------
		// One launch per panel costs the spine of the factorisation ~16 us per 64 columns, of which the dependency chain itself (the four
		// 16-column chains, the updates between them, the last 16 x 16 inverse) is ~9: the rest is the launch, reloading what the previous
		// launch had in LDS, and waiting for the slowest workgroup of the panel before the next panel may start.  Here a whole range of
		// panels is ONE launch.  Workgroup 0 (the spine) walks down the diagonal and never waits for the panel rows below it; everything else
		// is done by single waves (four per workgroup, no LDS, no barrier) that own a 16-row quarter of one 64 x 64 tile at a time,
		// left-looking:
		//   tile (r, c), c <= r - 2 or r outside the outer block:   L(r, c) = (A(r, c) - sum_{i < c} L(r, i) L(c, i)^T) T_c^T    after T_c is out
		//   tiles (r, r - 1) and (r, r) of a row the spine will reach: A~ = A - sum_{i <= r - 2} L(r, i) L(., i)^T, stored in place ("pre" tasks);
		//   the spine finishes them itself: L(k, k - 1) = A~(k, k - 1) T_{k-1}^T with T_{k-1} still in its LDS, A_kk = A~(k, k) - L(k, k - 1) L(k, k - 1)^T,
		//   so that no other workgroup sits between T_{k-1} and the first chain of panel k.
		// Hand-over is by flags in global memory, one int per tile quarter (= epoch of the fit that made it final; the buffer is never
		// cleared), data and flags as agent-scope relaxed atomics (sc1: write-through stores, loads that do not trust another XCD's L2) with
		// s_waitcnt vmcnt(0) between a quarter's stores and its flag (probes/hop_probe.hip: 1.9 us per hand-over of a tile, 0.55 for a flag alone).
		// Tasks are drawn from ONE ticket counter in an order in which every task depends on earlier ones only (column by column, the rows the
		// spine needs next first): whatever a task waits for was drawn before it, by a workgroup that is running, so the launch makes progress with
		// any number of its worker workgroups resident.  What the argument assumes is that workgroup 0 — the spine, which draws no ticket — is
		// dispatched no later than the workers that wait for it; HIP dispatches the workgroups of a grid in index order, but does not promise to.
		// The bounded wait backs that up: after a.poll_limit polls (DAG_POLL_LIMIT_DEFAULT, about 5-8 s) a wave raises the error flag, everybody
		// leaves, and *info becomes -1; the host then repeats the factorisation with one launch per panel (no waits between workgroups:
		// gple_capi.hip, recover_fit) — a give-up costs time, never a wrong result.  (2^21 polls, 1-2 s, until round 4: two PROCESSES on one GPU, the
		// other one's contraction holding every CU with one 110 KB workgroup each for seconds on end, starved a launch's workers past that once in a
		// rehearsal — the first give-up seen outside the tests that force one.  A wait is better than a give-up while the other side makes progress.)
		constexpr int DAG_POLL_LIMIT_DEFAULT = 1 << 23;
		// the ticket floor of a launch is (epoch * DAG_MAX_LAUNCHES + launch number) << 32: a factorisation may have up to DAG_MAX_LAUNCHES launches
		// (n = 8192: 12; more than 64, the packing of round 3, from n ~ 29k on or with GPLE_CHOL_OUTER=256 above n = 16384 — launch 64 of epoch e then
		// had the floor of launch 0 of epoch e + 1), and epochs stay below 2^31 / DAG_MAX_LAUNCHES (dag_state)
		constexpr unsigned long long DAG_MAX_LAUNCHES = 4096;
#ifndef DAG_POLL_SLEEP
#define DAG_POLL_SLEEP 8
#endif
		struct DagArgs
		{
			double* A;
			long lda;
			double* T;
			long ldt;
			int* info;
			double* uvec;
			int* flags; // 4 ints per tile (r * FS + c), then per column: T_k, pre-tile (k, k - 1), pre-tile (k, k); then the error word (4 ints in front: the ticket counter)
			int FS, R;  // block columns of the matrix; block rows of A (one more than FS with the label row)
			int c0, C1; // the panels of this launch, an outer block of the factorisation: sums start at c0, rows below C1 are spine rows
			int epoch, nunits, seq; // units of work (tiles: four quarter tasks each); number of this launch within the factorisation
			int poll_limit;         // polls after which a waiting wave gives up (DAG_POLL_LIMIT_DEFAULT; lowered by the give-up test)
			long long* stamps; // probe (GPLE_CHOL_DAG_STAMPS): 8 wall-clock stamps per panel of the spine, or nullptr
			double* pa;        // scratch, one 64 x 64 tile per block row: A~(r, r - 2) before its multiplication by T_{r-2}^T (what the pre-tiles of row r need)
			int tt_ld;         // tiles per row of Tt (= panels of the launch): tile (r, c) at ((r - c0) * tt_ld + (c - c0)) * 4096
			double* Tt;        // the launch also forms T = L^-1 below the diagonal blocks (matrices of one outer block): scratch of FS x FS tiles of
			                   // 64 x 64, tile (r, c) = T(r, c)^T (what the tiles below it multiply with), or nullptr; for the launch's own columns: the whole
			                   // inverse of a one-block matrix, the diagonal row block of T of a launch that is one row block of a larger inverse
		};
		__device__ __forceinline__ double ldc(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		__device__ __forceinline__ void stc(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		__device__ __forceinline__ int ldf(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		__device__ __forceinline__ void stf(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
		__device__ __forceinline__ int* dag_tile(const DagArgs& a, int r, int c) { return a.flags + 4 * (static_cast<long>(r) * a.FS + c); }
		__device__ __forceinline__ int* dag_t(const DagArgs& a, int k) { return a.flags + 4 * (static_cast<long>(a.R) * a.FS + k); }
		__device__ __forceinline__ int* dag_pb(const DagArgs& a, int k) { return a.flags + 4 * (static_cast<long>(a.R + 1) * a.FS + k); }
		__device__ __forceinline__ int* dag_pc(const DagArgs& a, int k) { return a.flags + 4 * (static_cast<long>(a.R + 2) * a.FS + k); }
		__device__ __forceinline__ int* dag_err(const DagArgs& a) { return a.flags + 4 * static_cast<long>(a.R + 3) * a.FS; }
		__device__ __forceinline__ int* dag_pb2(const DagArgs& a, int k) { return a.flags + 4 * (static_cast<long>(a.R + 3) * a.FS + 1 + static_cast<long>(a.FS) * a.FS + a.FS + k); }
		__device__ __forceinline__ int* dag_pa(const DagArgs& a, int r) { return a.flags + 4 * (static_cast<long>(a.R + 3) * a.FS + 1 + static_cast<long>(a.FS) * a.FS + r); }
		__device__ __forceinline__ int* dag_tt(const DagArgs& a, int r, int c) { return a.flags + 4 * (static_cast<long>(a.R + 3) * a.FS + 1 + static_cast<long>(r) * a.FS + c); }
		// the ticket counter of the work queue: in front of the flags, at the same place whatever the matrix size — its high word outgrows every epoch
		// and must never be read as a flag
		__device__ __forceinline__ unsigned long long* dag_tickets(const DagArgs& a) { return reinterpret_cast<unsigned long long*>(a.flags - 4); }
		// the whole wave waits until the four quarter flags at f4 and (if given) the four at g4 and the single flag at f1 carry this epoch
		__device__ __forceinline__ bool dag_wait(const DagArgs& a, const int* f4, const int* g4, const int* f1, int lane, bool patient = false, const int* h4 = nullptr)
		{
			const int* p = f4 + (lane & 3);
			if (g4 != nullptr && (lane & 4)) p = g4 + (lane & 3);
			if (h4 != nullptr && (lane & 8)) p = h4 + (lane & 3);
			if (f1 != nullptr && lane >= 16) p = f1;
			const int* const err = dag_err(a);
			for (int it = 0; it < a.poll_limit; ++it)
			{
				const int v = ldf(p);
				if (__all(v - a.epoch >= 0))
				{
					asm volatile("" ::: "memory");
					return true;
				}
				if ((it & 31) == 31 && ldf(err) - a.epoch >= 0) return false;
				// a task nobody will wait for soon looks less often: the polls of a thousand waiting waves go through the same memory system as
				// the tiles the spine waits for
				if (patient) __builtin_amdgcn_s_sleep(48);
				else __builtin_amdgcn_s_sleep(DAG_POLL_SLEEP);
			}
			if (lane == 0) stf(dag_err(a), a.epoch), atomicExch(a.info, -1);
			return false;
		}
		// one worker task: quarter qa (rows 16 qa ..) of tile (r, cs); sums over i in [c0, iend) of L(r, i) L(xr, i)^T are taken off, then
		// (fin) the result is multiplied by T_cs^T; stored in place; *done = epoch.  Accumulator element acc[b][q] of lane (fr, fk) is entry
		// (row 16 qa + fr, column 16 b + fk + 4 q) of the tile — also the layout of the MFMA operand "row fr, k = 4 (4 b + q) + fk", so the
		// product with T_cs^T needs no transposition.
		__device__ __forceinline__ bool dag_task(const DagArgs& a, int r, int xr, int cs, int iend, bool fin, int qa, int* done, int lane, bool patient, bool publish = false)
		{
			const int fr = lane & 15, fk = lane >> 4;
			const long lda = a.lda;
			double* const Ct = a.A + static_cast<long>(r) * NB + static_cast<long>(cs) * NB * lda + 16 * qa + fr;
			d4v acc[4];
#pragma unroll
			for (int b = 0; b < 4; ++b)
#pragma unroll
				for (int q = 0; q < 4; ++q) acc[b][q] = Ct[static_cast<long>(16 * b + fk + 4 * q) * lda]; // as the previous launches left it
			for (int i = a.c0; i < iend; ++i)
			{
				if (!dag_wait(a, dag_tile(a, xr, i), nullptr, dag_tile(a, r, i) + qa, lane, patient)) return false;
				const double* const Ly = a.A + static_cast<long>(r) * NB + static_cast<long>(i) * NB * lda + 16 * qa + fr;
				const double* const Lx = a.A + static_cast<long>(xr) * NB + static_cast<long>(i) * NB * lda + fr;
				// four chunks of 16 columns, the next one requested before the current one is multiplied (two operand sets of 40 registers)
				double y[2][4], x[2][4][4];
				auto fetch = [&](int ch, int buf) {
#pragma unroll
					for (int q = 0; q < 4; ++q)
					{
						const long kcol = static_cast<long>(16 * ch + 4 * q + fk) * lda;
						y[buf][q] = ldc(Ly + kcol);
#pragma unroll
						for (int b = 0; b < 4; ++b) x[buf][b][q] = ldc(Lx + 16 * b + kcol);
					}
				};
				fetch(0, 0);
#pragma unroll
				for (int ch = 0; ch < 4; ++ch)
				{
					if (ch + 1 < 4) fetch(ch + 1, (ch + 1) & 1);
#pragma unroll
					for (int q = 0; q < 4; ++q)
#pragma unroll
						for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[ch & 1][b][q], y[ch & 1][q], acc[b], 0, 0, 0);
				}
			}
			if (publish) // tile (r, r - 2) of a row the spine will reach: the pre-tiles of that row form L(r, r - 2) = A~ T^T themselves, a hand-over earlier
			{
				double* const Pt = a.pa + static_cast<long>(r) * (NB * NB) + 16 * qa + fr;
#pragma unroll
				for (int b = 0; b < 4; ++b)
#pragma unroll
					for (int q = 0; q < 4; ++q) stc(Pt + (16 * b + fk + 4 * q) * NB, acc[b][q]);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (lane == 0) stf(dag_pa(a, r) + qa, a.epoch);
			}
			if (fin)
			{
				if (!dag_wait(a, dag_t(a, cs), nullptr, nullptr, lane, patient)) return false;
				const double* const Tt = a.T + static_cast<long>(cs) * NB * (a.ldt + 1) + fr;
				double xt[40];
				[&]<int... Bs>(std::integer_sequence<int, Bs...>)
				{
					(
						[&] {
							constexpr int b = Bs, off = 2 * b * (b + 1); // 4 + 8 + .. fragments before block row b
#pragma unroll
							for (int q = 0; q < 4 * (b + 1); ++q) xt[off + q] = ldc(Tt + 16 * b + static_cast<long>(4 * q + fk) * a.ldt);
						}(),
						...);
				}
				(std::make_integer_sequence<int, 4>{});
				d4v out[4];
				[&]<int... Bs>(std::integer_sequence<int, Bs...>)
				{
					(
						[&] {
							constexpr int b = Bs, off = 2 * b * (b + 1);
							out[b] = (d4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
							for (int q = 0; q < 4 * (b + 1); ++q) out[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[off + q], acc[q >> 2][q & 3], out[b], 0, 0, 0);
						}(),
						...);
				}
				(std::make_integer_sequence<int, 4>{});
#pragma unroll
				for (int b = 0; b < 4; ++b) acc[b] = out[b];
				// the label row below the matrix: its factor entries are u = L^-1 y
				if (a.uvec != nullptr && r == a.R - 1 && qa == 0 && fr == 0)
#pragma unroll
					for (int b = 0; b < 4; ++b)
#pragma unroll
						for (int q = 0; q < 4; ++q) a.uvec[cs * NB + 16 * b + fk + 4 * q] = acc[b][q];
			}
#pragma unroll
			for (int b = 0; b < 4; ++b)
#pragma unroll
				for (int q = 0; q < 4; ++q) stc(Ct + static_cast<long>(16 * b + fk + 4 * q) * lda, acc[b][q]);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (lane == 0) stf(done, a.epoch);
			return true;
		}
		// Quarter qa of a pre-tile of row r: (r, r - 1) (diag = false) or (r, r) — A~ = A - sum_{i <= r-2} L(r, i) L(., i)^T, stored in place for
		// the spine.  The terms up to r - 3 as in dag_task; the last one needs L(r, r - 2) = A~(r, r - 2) T_{r-2}^T, which the wave forms itself from the
		// published A~(r, r - 2) (its own rows; for the diagonal tile the other row blocks as well, one after the other) as soon as T_{r-2} is out
		// instead of waiting for the tile's own task to store it: T_{r-2} -> pre-tiles -> spine is two hand-overs instead of three.
		// The off-diagonal tile comes in two halves of 32 columns (half = 0 / 1; -1: the diagonal tile, whole): its last term can only start when the
		// spine's L(r - 1, r - 2) is out, and half the columns are half the operand loads and MFMAs between that flag and the tile's own.
		template <int HALF>
		__device__ __forceinline__ bool dag_pre_task(const DagArgs& a, int r, int qa, int lane, double* xch)
		{
			constexpr bool diag = HALF < 0;
			constexpr int B0 = diag ? 0 : 2 * HALF, NBK = diag ? 4 : 2; // the tile's 16-column blocks B0 .. B0 + NBK - 1
			const int fr = lane & 15, fk = lane >> 4;
			const long lda = a.lda;
			const int cs = diag ? r : r - 1, xr = cs;
			double* const Ct = a.A + static_cast<long>(r) * NB + static_cast<long>(cs) * NB * lda + 16 * qa + fr;
			d4v acc[NBK];
#pragma unroll
			for (int b = 0; b < NBK; ++b)
#pragma unroll
				for (int q = 0; q < 4; ++q) acc[b][q] = Ct[static_cast<long>(16 * (B0 + b) + fk + 4 * q) * lda];
			for (int i = a.c0; i < r - 2; ++i)
			{
				if (!dag_wait(a, dag_tile(a, xr, i), nullptr, dag_tile(a, r, i) + qa, lane)) return false;
				const double* const Ly = a.A + static_cast<long>(r) * NB + static_cast<long>(i) * NB * lda + 16 * qa + fr;
				const double* const Lx = a.A + static_cast<long>(xr) * NB + static_cast<long>(i) * NB * lda + 16 * B0 + fr;
				double y[2][4], x[2][NBK][4];
				auto fetch = [&](int ch, int buf) {
#pragma unroll
					for (int q = 0; q < 4; ++q)
					{
						const long kcol = static_cast<long>(16 * ch + 4 * q + fk) * lda;
						y[buf][q] = ldc(Ly + kcol);
#pragma unroll
						for (int b = 0; b < NBK; ++b) x[buf][b][q] = ldc(Lx + 16 * b + kcol);
					}
				};
				fetch(0, 0);
#pragma unroll
				for (int ch = 0; ch < 4; ++ch)
				{
					if (ch + 1 < 4) fetch(ch + 1, (ch + 1) & 1);
#pragma unroll
					for (int q = 0; q < 4; ++q)
#pragma unroll
						for (int b = 0; b < NBK; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[ch & 1][b][q], y[ch & 1][q], acc[b], 0, 0, 0);
				}
			}
			// the term of column r - 2
			{
				const double* const Pa = a.pa + static_cast<long>(r) * (NB * NB) + fr;
				// rows 16 rb .. of A~(r, r - 2) in accumulator layout = operand layout (row fr, k = 16 b + fk + 4 q)
				auto load_pa = [&](int rb, d4v (&t)[4]) {
#pragma unroll
					for (int b = 0; b < 4; ++b)
#pragma unroll
						for (int q = 0; q < 4; ++q) t[b][q] = ldc(Pa + 16 * rb + (16 * b + fk + 4 * q) * NB);
				};
				if (!dag_wait(a, dag_pa(a, r), nullptr, nullptr, lane)) return false; // (all four quarters; the off-diagonal tile needs its own only)
				d4v pa_own[4];
				load_pa(qa, pa_own);
				if (a.stamps != nullptr && lane == 0 && qa == 0 && diag) a.stamps[16 * r + 12 + 1] = wall_clock64();
				if (!dag_wait(a, dag_t(a, r - 2), nullptr, nullptr, lane)) return false;
				if (a.stamps != nullptr && lane == 0 && qa == 0 && diag) a.stamps[16 * r + 12 + 2] = wall_clock64();
				const double* const Tt = a.T + static_cast<long>(r - 2) * NB * (a.ldt + 1) + fr;
				double xt[40];
				[&]<int... Bs>(std::integer_sequence<int, Bs...>)
				{
					(
						[&] {
							constexpr int b = Bs, off = 2 * b * (b + 1);
#pragma unroll
							for (int q = 0; q < 4 * (b + 1); ++q) xt[off + q] = ldc(Tt + 16 * b + static_cast<long>(4 * q + fk) * a.ldt);
						}(),
						...);
				}
				(std::make_integer_sequence<int, 4>{});
				// rows of L(r, r - 2) from rows of A~: out(row, j) = sum_{k <= j} in(row, k) T(j, k)
				auto times_tt = [&](const d4v (&in)[4], d4v (&out)[4]) {
					[&]<int... Bs>(std::integer_sequence<int, Bs...>)
					{
						(
							[&] {
								constexpr int b = Bs, off = 2 * b * (b + 1);
								out[b] = (d4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
								for (int q = 0; q < 4 * (b + 1); ++q) out[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[off + q], in[q >> 2][q & 3], out[b], 0, 0, 0);
							}(),
							...);
					}
					(std::make_integer_sequence<int, 4>{});
				};
				d4v la_own[4];
				times_tt(pa_own, la_own);
				if constexpr (diag)
				{
					// A~(r, r)(rows qa, columns = rows of block b) -= L(r, r - 2)(rows qa) L(r, r - 2)(rows b)^T: the four waves of the workgroup hold the four
					// row blocks of L(r, r - 2) and meet in LDS (a unit's four quarters run on the four waves of one workgroup, in step; an fp64 MFMA
					// is 64 cycles — forming the other three row blocks again would cost each wave 3 us)
#pragma unroll
					for (int b = 0; b < 4; ++b)
#pragma unroll
						for (int q = 0; q < 4; ++q) xch[(16 * qa + fr) * DLS + 16 * b + fk + 4 * q] = la_own[b][q];
					asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
					for (int b = 0; b < 4; ++b)
					{
						double x[16];
#pragma unroll
						for (int q = 0; q < 16; ++q) x[q] = xch[(16 * b + fr) * DLS + 4 * q + fk];
#pragma unroll
						for (int q = 0; q < 16; ++q) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[q], la_own[q >> 2][q & 3], acc[b], 0, 0, 0);
					}
				}
				else
				{
					if (!dag_wait(a, dag_tile(a, r - 1, r - 2), nullptr, nullptr, lane)) return false; // the spine's L(r - 1, r - 2)
					const double* const Lx = a.A + static_cast<long>(r - 1) * NB + static_cast<long>(r - 2) * NB * lda + 16 * B0 + fr;
					double x[2][NBK][4];
					auto fetch = [&](int ch, int buf) {
#pragma unroll
						for (int q = 0; q < 4; ++q)
#pragma unroll
							for (int b = 0; b < NBK; ++b) x[buf][b][q] = ldc(Lx + 16 * b + static_cast<long>(16 * ch + 4 * q + fk) * lda);
					};
					fetch(0, 0);
#pragma unroll
					for (int ch = 0; ch < 4; ++ch)
					{
						if (ch + 1 < 4) fetch(ch + 1, (ch + 1) & 1);
#pragma unroll
						for (int q = 0; q < 4; ++q)
#pragma unroll
							for (int b = 0; b < NBK; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[ch & 1][b][q], la_own[ch][q], acc[b], 0, 0, 0);
					}
				}
			}
#pragma unroll
			for (int b = 0; b < NBK; ++b)
#pragma unroll
				for (int q = 0; q < 4; ++q) stc(Ct + static_cast<long>(16 * (B0 + b) + fk + 4 * q) * lda, acc[b][q]);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (lane == 0) stf((diag ? dag_pc(a, r) : HALF == 0 ? dag_pb(a, r) : dag_pb2(a, r)) + qa, a.epoch);
			if (a.stamps != nullptr && lane == 0 && qa == 0 && diag) a.stamps[16 * r + 15] = wall_clock64();
			return true;
		}
		// Quarter qa of the inverse's tile (r, j), j < r, held transposed: W = V^T with V = sum_{m = j}^{r-1} L(r, m) T(m, j), then
		// T(r, j)^T = -W T_rr^T — the same two steps as dag_task with the transposed tiles above in the place of the rows of L: rows of the
		// accumulator = columns of T(r, j), so that every operand is read with 16 rows of 8 bytes side by side.  Out: the scratch tile (for the
		// rows below) and the matrix T.
		__device__ __forceinline__ bool dag_ttask(const DagArgs& a, int r, int j, int qa, int lane)
		{
			const int fr = lane & 15, fk = lane >> 4;
			d4v acc[4];
#pragma unroll
			for (int b = 0; b < 4; ++b) acc[b] = (d4v){0.0, 0.0, 0.0, 0.0};
			for (int m = j; m < r; ++m) // (j >= c0: the launch's own columns)
			{
				// the tile above: the diagonal one comes from the spine (all four words of its own flag), the others from their own quarter tasks
				if (!dag_wait(a, dag_tile(a, r, m), m == j ? dag_tt(a, j, j) : nullptr, m == j ? nullptr : dag_tt(a, m, j) + qa, lane, true)) return false;
				const double* const Ty = a.Tt + (static_cast<long>(m - a.c0) * a.tt_ld + (j - a.c0)) * (NB * NB) + 16 * qa + fr;
				const double* const Lx = a.A + static_cast<long>(r) * NB + static_cast<long>(m) * NB * a.lda + fr;
				double y[2][4], x[2][4][4];
				auto fetch = [&](int ch, int buf) {
#pragma unroll
					for (int q = 0; q < 4; ++q)
					{
						const int kk = 16 * ch + 4 * q + fk;
						y[buf][q] = ldc(Ty + static_cast<long>(kk) * NB);
#pragma unroll
						for (int b = 0; b < 4; ++b) x[buf][b][q] = ldc(Lx + 16 * b + static_cast<long>(kk) * a.lda);
					}
				};
				fetch(0, 0);
#pragma unroll
				for (int ch = 0; ch < 4; ++ch)
				{
					if (ch + 1 < 4) fetch(ch + 1, (ch + 1) & 1);
#pragma unroll
					for (int q = 0; q < 4; ++q)
#pragma unroll
						for (int b = 0; b < 4; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ch & 1][b][q], y[ch & 1][q], acc[b], 0, 0, 0);
				}
			}
			if (!dag_wait(a, dag_t(a, r), nullptr, nullptr, lane, r + 1 < a.C1)) return false; // (the last row's tiles are what the launch ends with: they look often)
			const double* const Tt = a.T + static_cast<long>(r) * NB * (a.ldt + 1) + fr;
			double xt[40];
			[&]<int... Bs>(std::integer_sequence<int, Bs...>)
			{
				(
					[&] {
						constexpr int b = Bs, off = 2 * b * (b + 1);
#pragma unroll
						for (int q = 0; q < 4 * (b + 1); ++q) xt[off + q] = ldc(Tt + 16 * b + static_cast<long>(4 * q + fk) * a.ldt);
					}(),
					...);
			}
			(std::make_integer_sequence<int, 4>{});
			d4v out[4];
			[&]<int... Bs>(std::integer_sequence<int, Bs...>)
			{
				(
					[&] {
						constexpr int b = Bs, off = 2 * b * (b + 1);
						out[b] = (d4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
						for (int q = 0; q < 4 * (b + 1); ++q) out[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xt[off + q], acc[q >> 2][q & 3], out[b], 0, 0, 0);
					}(),
					...);
			}
			(std::make_integer_sequence<int, 4>{});
			// out[b][q] of lane (fr, fk) = T(r, j)(row 16 b + fk + 4 q, column 16 qa + fr)
			double* const So = a.Tt + (static_cast<long>(r - a.c0) * a.tt_ld + (j - a.c0)) * (NB * NB) + 16 * qa + fr;
			double* const To = a.T + static_cast<long>(r) * NB + (static_cast<long>(j) * NB + 16 * qa + fr) * a.ldt;
#pragma unroll
			for (int b = 0; b < 4; ++b)
#pragma unroll
				for (int q = 0; q < 4; ++q)
				{
					stc(So + static_cast<long>(16 * b + fk + 4 * q) * NB, out[b][q]);
					To[16 * b + fk + 4 * q] = out[b][q]; // read by later launches only
				}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			if (lane == 0) stf(dag_tt(a, r, j) + qa, a.epoch);
			return true;
		}
		// tasks of the launch, in dependency order; host and device count them the same way.  Column c of the block [c0, C1): the tiles below
		// row c + 1 (the spine finishes (c + 1, c) itself), with the two pre-tiles of row c + 2 right after tile (c + 2, c), which they need
		__host__ __device__ inline int dag_column_units(int c, int C1, int R, bool& has_pre, int& lo)
		{
			lo = c + 1 < C1 ? c + 2 : c + 1;
			has_pre = c + 2 < C1;
			const int ng = R - lo > 0 ? R - lo : 0;
			return ng + (has_pre ? 3 : 0);
		}
		__host__ inline int dag_count_units(int c0, int C1, int R, bool inverse)
		{
			int n = 0;
			for (int c = c0; c < C1; ++c)
			{
				bool hp;
				int lo;
				n += dag_column_units(c, C1, R, hp, lo) + (inverse ? c - c0 : 0); // the inverse's row c (tiles (c, c - 1) .. (c, c0)) waits for T_c like column c
			}
			return n;
		}

		__global__ void __launch_bounds__(256) potrf_dag_kernel(const DagArgs a)
		{
			const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
			__shared__ __attribute__((aligned(16))) double S[NB * DLS];  // spine: A_kk -> L_kk (strictly lower tiles); workers: the exchange tile of a diagonal pre-tile
			__shared__ __attribute__((aligned(16))) double TI[NB * DLS]; // spine: T_k; before the chains of panel k: L(k, k - 1), k-major (D)
			if (blockIdx.x != 0)
			{
				// Units are handed out in their dependency order from one counter: whatever a unit waits for was handed out before it, to a workgroup
				// that is running, so the launch makes progress with any number of its workgroups resident (workgroup 0, dispatched first, is
				// the spine).  The counter is never cleared: every launch raises it to its own floor (epoch, launch number) before drawing.
				__shared__ int cur_unit;
				const unsigned long long floor = (static_cast<unsigned long long>(a.epoch) * DAG_MAX_LAUNCHES + static_cast<unsigned long long>(a.seq)) << 32;
				for (;;)
				{
					if (t == 0)
					{
						atomicMax(dag_tickets(a), floor);
						cur_unit = static_cast<int>(atomicAdd(dag_tickets(a), 1ull) - floor);
					}
					__syncthreads();
					int u = cur_unit;
					__syncthreads();
					if (u >= a.nunits) return;
					const int qa = w;
					// kind 0: tile (r, c) to the end; 1, 2: halves of pre-tile (r, r - 1); 3: pre-tile (r, r); 4: a tile of the inverse
					int kind = -1, r = 0, c = 0;
					for (int cc = a.c0; kind < 0 && cc < a.C1; ++cc)
					{
						bool hp;
						int lo;
						const int nu = dag_column_units(cc, a.C1, a.R, hp, lo);
						if (u < nu)
						{
							c = cc;
							if (!hp) kind = 0, r = lo + u;
							else if (u == 0) kind = 0, r = cc + 2;
							else if (u <= 3) kind = u, r = cc + 2; // 1, 2: the halves of pre-tile (r, r - 1); 3: pre-tile (r, r)
							else kind = 0, r = cc + u - 1;
						}
						else if (a.Tt != nullptr && u < nu + cc - a.c0) kind = 4, r = cc, c = cc - 1 - (u - nu); // the inverse's tile (cc, c), nearest the diagonal first
						u -= nu + (a.Tt != nullptr ? cc - a.c0 : 0);
					}
					if (kind < 0) return;
					bool ok;
					if (kind == 4) ok = dag_ttask(a, r, c, qa, lane);
					else if (kind == 0) ok = dag_task(a, r, c, c, c, true, qa, dag_tile(a, r, c) + qa, lane, r > c + 3 && r >= a.C1 ? true : r > c + 4, r == c + 2 && r < a.C1);
					else if (kind == 1) ok = dag_pre_task<0>(a, r, qa, lane, S);
					else if (kind == 2) ok = dag_pre_task<1>(a, r, qa, lane, S);
					else ok = dag_pre_task<-1>(a, r, qa, lane, S);
					if (!ok) return;
				}
			}
			// ---- the spine: four waves, the schedule of potrf_step_kernel's panel waves
			__shared__ double rinv[NB];
			double* const D = TI;
			const int fr = lane & 15, fk = lane >> 4;
			auto Sx = [&](int ti, int tj) { return S + ti * 16 * DLS + tj * 16; };
			auto Tx = [&](int ti, int tj) { return TI + ti * 16 * DLS + tj * 16; };
			auto upd = [&](int ti, int tj, int sp) {
				d4v acc = tile_load(Sx(ti, tj), lane);
				acc = tile_mac<true, true, 16>(acc, Sx(ti, sp), Sx(tj, sp), lane);
				tile_store(Sx(ti, tj), acc, lane);
			};
			auto pend_upd = [&](int ti, int tj) {
				d4v acc = tile_load(Sx(ti, tj), lane);
				acc = tile_mac_kk_neg64(acc, D + ti * 16, D + tj * 16, lane);
				tile_store(Sx(ti, tj), acc, lane);
			};
			auto pend_upd2 = [&](int aa, int ca, int b, int c, bool first_t) {
				d4v acc0 = first_t ? tile_load_t(Sx(aa, ca), lane) : tile_load(Sx(aa, ca), lane);
				d4v acc1 = tile_load(Sx(b, c), lane);
				tile_mac2_kk_neg64(acc0, acc1, D + (first_t ? ca : aa) * 16, D + b * 16, D + (first_t ? aa : c) * 16, lane);
				if (first_t) tile_store_t(Sx(aa, ca), acc0, lane);
				else tile_store(Sx(aa, ca), acc0, lane);
				tile_store(Sx(b, c), acc1, lane);
			};
			auto v_acc = [&](int i, int b, int k, bool first) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				if (!first) acc = tile_load(Tx(i, b), lane);
				acc = tile_mac<false, false, 16>(acc, Sx(i, k), Tx(k, b), lane);
				tile_store(Tx(i, b), acc, lane);
			};
			auto t_fin = [&](int i, int b) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				acc = tile_mac<false, true, 16>(acc, Tx(i, i), Tx(i, b), lane);
				tile_store(Tx(i, b), acc, lane);
			};
			// No barrier of the spine waits for memory: LDS only (s_waitcnt lgkmcnt(0); s_barrier).  What the workers wait for leaves as early as it
			// exists — L(k, k - 1) straight from the product's registers, the last tile row of T_k from those of its last products — and is flagged
			// by the wave that stored it after a wait of its own, placed where the stores are a stage old; what comes in for the next panel is
			// requested as soon as wave 1 has seen its flags.
			auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
			// rows 16 q4 .. 16 q4 + 15, columns [c_lo, c_lo + 16 nct) of T_k from TI to the diagonal block of T (zeros above the diagonal)
			auto store_t_rows = [&](int k, int q4, int c_lo, int nct) {
				double* __restrict__ Tkk = a.T + static_cast<long>(k) * NB * (a.ldt + 1);
				const int rr = 16 * q4 + fr;
				for (int q = 0; q < 4 * nct; ++q)
				{
					const int c = c_lo + fk + 4 * q;
					stc(Tkk + rr + static_cast<long>(c) * a.ldt, c <= rr ? TI[rr * DLS + c] : 0.0);
				}
			};
			// columns [c_lo, c_lo + nc) of T_k^T (= rows of T_k: complete with their row tile) into the scratch of the inverse's tiles
			auto store_tt_cols = [&](int k, int c_lo, int nc) {
				if (a.Tt == nullptr) return;
				double* __restrict__ Td = a.Tt + (static_cast<long>(k - a.c0) * (a.tt_ld + 1)) * (NB * NB) + lane;
				for (int c = c_lo; c < c_lo + nc; ++c) stc(Td + static_cast<long>(c) * NB, lane <= c ? TI[c * DLS + lane] : 0.0);
			};
			// the block above an odd diagonal block of T: see potrf_step_kernel
			auto zero_above = [&](int k) {
				double* __restrict__ Tab = a.T + static_cast<long>(k) * NB * (a.ldt + 1) - NB;
#pragma unroll
				for (int q = 0; q < 64; ++q) stc(Tab + lane + static_cast<long>(q) * a.ldt, 0.0);
			};
			// T(3, b) = -T(3, 3) V(3, b): to TI for the next panel's first product and straight to memory
			auto t_fin_out = [&](int k, int b) {
				d4v acc = {0.0, 0.0, 0.0, 0.0};
				acc = tile_mac<false, true, 16>(acc, Tx(3, 3), Tx(3, b), lane);
				tile_store(Tx(3, b), acc, lane);
				double* __restrict__ Tkk = a.T + static_cast<long>(k) * NB * (a.ldt + 1);
#pragma unroll
				for (int r = 0; r < 4; ++r) stc(Tkk + 48 + fk + 4 * r + static_cast<long>(16 * b + fr) * a.ldt, acc[r]);
			};
			__builtin_amdgcn_s_setprio(3);
			// A~(k, k): waves 1-3, a row per lane (512 contiguous bytes per load: anything less is a gather to the address unit, ~60 cycles an instruction),
			// 21 or 22 columns per wave; the 8-way bank conflict on the way into S is the lesser evil.  Wave 0 loads none: its issue slots belong to the
			// chains.  A~(k, k - 1): row tile w as operand fragments, every wave (gathers: 16 rows x 4 columns each).
			double sreg[14], bx[16];
			bool have = false;        // ... already requested for the coming panel
			__shared__ int nxt_ready; // wave 1 watches the flags of the coming panel's pre-tiles for everybody
			if (t == 0) nxt_ready = -1;
			// A~(k, k), lower tiles only, by waves 1-3: tile column 0 — all the first chain needs — a 64-row column per instruction (column c belongs to
			// wave 1 + c % 3: sreg[0 .. sj16)), written to S before the chain; and each wave's own two tiles of the stage beside the first chain
			// (wave 1: (1,1), (2,1); wave 2: (2,2), (3,2); wave 3: (3,1), (3,3): sreg[6 .. 14)), which it writes at the start of that stage itself —
			// nobody else reads them before the barrier that ends it.  L(k, k - 1) leaves by the same column ownership (all 64 columns).
			const int sjn = w == 1 ? 22 : 21, sj16 = w == 1 ? 6 : 5;
			// element j (0 .. 7) of the wave's two tiles: row and column in the block
			auto own_row = [&](int j) { const int e = lane + 64 * j; return w == 3 ? 48 + (e & 15) : 16 * w + (e & 31); };
			auto own_col = [&](int j) {
				const int e = lane + 64 * j;
				if (w != 3) return 16 * w + (e >> 5);
				const int cc = e >> 4;
				return cc < 16 ? 16 + cc : 32 + cc;
			};
			// A~(k, k - 1) is asked for a stage or two ahead (the panel's first product needs it at once); A~(k, k) at the top of the panel — it is not
			// needed before that product and two barriers are through, and 44 registers less live through the chains
			auto request_inputs = [&](int k) {
				if (k > a.c0)
				{
					const double* __restrict__ Bt = a.A + static_cast<long>(k) * NB * (a.lda + 1) - static_cast<long>(NB) * a.lda + 16 * w + fr;
#pragma unroll
					for (int q = 0; q < 16; ++q) bx[q] = ldc(Bt + static_cast<long>(4 * q + fk) * a.lda);
				}
			};
			auto request_diag = [&](int k) {
				const double* __restrict__ Akk = a.A + static_cast<long>(k) * NB * (a.lda + 1);
				if (w > 0)
				{
#pragma unroll
					for (int q = 0; q < 6; ++q)
						if (q < sj16) sreg[q] = ldc(Akk + lane + static_cast<long>(w - 1 + 3 * q) * a.lda);
#pragma unroll
					for (int j = 0; j < 8; ++j) sreg[6 + j] = ldc(Akk + own_row(j) + static_cast<long>(own_col(j)) * a.lda);
				}
			};
			// the twelve flags of the pre-tiles of row k (two halves of (k, k - 1), (k, k)), one per lane (the rest repeat them)
			auto pre_flag_ptr = [&](int k) { return (lane & 8 ? dag_pc(a, k) : lane & 4 ? dag_pb2(a, k) : dag_pb(a, k)) + (lane & 3); };
			for (int k = a.c0; k < a.C1; ++k)
			{
				const int j0 = k * NB;
				const bool pend = k > a.c0;
				auto stamp = [&](int i) {
					if (a.stamps != nullptr && t == 0) a.stamps[16 * k + i] = wall_clock64();
				};
				stamp(0);
				if (!have)
				{
					if (k >= a.c0 + 2 && nxt_ready != k && !dag_wait(a, dag_pb(a, k), dag_pb2(a, k), nullptr, lane, false, dag_pc(a, k))) break; // (seen by a watcher too late to ask ahead)
					request_inputs(k);
				}
				have = false;
				request_diag(k);
				stamp(1);
				int first_bad = 0;
				if (pend)
				{
					// L(k, k - 1) = A~(k, k - 1) T_{k-1}^T: row tile w, column tile j over k <= 16 j + 15
					d4v lo[4];
					[&]<int... Js>(std::integer_sequence<int, Js...>)
					{
						(
							[&] {
								constexpr int j = Js;
								d4v o = {0.0, 0.0, 0.0, 0.0};
								double y[4 * (j + 1)];
#pragma unroll
								for (int q = 0; q < 4 * (j + 1); ++q) y[q] = TI[(16 * j + fr) * DLS + 4 * q + fk];
#pragma unroll
								for (int q = 0; q < 4 * (j + 1); ++q) o = __builtin_amdgcn_mfma_f64_16x16x4f64(bx[q], y[q], o, 0, 0, 0);
								lo[j] = o;
							}(),
							...);
					}
					(std::make_integer_sequence<int, 4>{});
					store_tt_cols(k - 1, 32 + 8 * w, 8); // rows 32 .. 63 of T_{k-1}, final since the last two barriers of the round before; flagged below
					stamp(2);
					lds_barrier(); // everybody has read T_{k-1}: its place takes L(k, k - 1), k-major
#pragma unroll
					for (int j = 0; j < 4; ++j) tile_store_t(D + j * 16 * DLS + w * 16, lo[j], lane);
				}
				if (w > 0) // tile column 0 now, the rest in the next stage (nothing before the first chain reads it)
#pragma unroll
					for (int q = 0; q < 6; ++q)
						if (q < sj16) S[lane * DLS + w - 1 + 3 * q] = sreg[q];
				lds_barrier();
				if (pend)
				{
					// T_{k-1} (and its transposed copy): the wave's stores are a product, two barriers and more old, the loads of A~(k, k) — the only
					// younger accesses — have been used: no waiting here
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					if (lane == 0) stf(dag_t(a, k - 1) + w, a.epoch);
					if (a.Tt != nullptr && lane == 0) stf(dag_tt(a, k - 1, k - 1) + w, a.epoch);
					// L(k, k - 1) leaves from D, a column (64 contiguous rows) per instruction, waves 1-3; each flags its columns after the next stage, after a
					// wait of its own that finds the stores a chain old — the tile is only ever read whole (all four words), as the second factor of a product
					if (w > 0)
					{
						double* __restrict__ Lg = a.A + j0 + lane + static_cast<long>(j0 - NB) * a.lda;
						for (int q = 0; q < sjn; ++q) stc(Lg + static_cast<long>(w - 1 + 3 * q) * a.lda, D[(w - 1 + 3 * q) * DLS + lane]);
					}
					pend_upd(w, 0);
					lds_barrier();
				}
				stamp(3);
				if (w > 0) // the wave's own two tiles of this stage
#pragma unroll
					for (int j = 0; j < 8; ++j) S[own_row(j) * DLS + own_col(j)] = sreg[6 + j];
				if (w == 0) diag_chain<0>(S, rinv, lane, first_bad);
				else if (pend)
				{
					if (w == 1) pend_upd2(1, 1, 2, 1, false);
					else if (w == 2) pend_upd2(2, 2, 3, 2, false);
					else pend_upd2(3, 1, 3, 3, true);
				}
				lds_barrier();
				stamp(4);
				if (w > 0 && pend)
				{
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					if (lane == 0) stf(dag_tile(a, k, k - 1) + w, a.epoch);
					if (w == 1 && lane == 1) stf(dag_tile(a, k, k - 1), a.epoch);
				}
				if (w == 0) upd(1, 1, 0);
				else if (w == 1) upd(2, 1, 0);
				else if (w == 2) upd(3, 1, 0);
				else upd(2, 2, 0);
				lds_barrier();
				stamp(5);
				if (w == 0) diag_chain<1, true>(S, rinv, lane, first_bad, TI); // (with the inverse of its diagonal tile: the 16 lanes below its 48 rows)
				else if (w == 1) diag_inv16(S, rinv, TI, 0, lane);
				else if (w == 2) upd(3, 2, 0);
				else
				{
					upd(3, 3, 0);
					if (k & 1) zero_above(k);
				}
				lds_barrier();
				stamp(6);
				if (w == 0) upd(2, 2, 1);
				else if (w == 1) upd(3, 2, 1);
				else if (w == 2) upd(3, 3, 1);
				else v_acc(1, 0, 0, true);
				lds_barrier();
				stamp(7);
				// the coming panel's inputs: wave 1 looks at its pre-tiles' flags once per stage from here on (a load whose answer the next stage reads)
				// and tells the others through LDS; the tiles are requested a stage after the flags are seen — wave 0 not before its last chain is done
				const bool next_in = k + 1 < a.C1;
				const bool next_flags = k + 1 >= a.c0 + 2;
				int pf = a.epoch - 1;
				bool asked = false;
				auto poll = [&] { // wave 1
					if (!next_in || !next_flags || nxt_ready == k + 1) return;
					if (asked && __all(pf - a.epoch >= 0))
					{
						if (lane == 0) nxt_ready = k + 1;
						return;
					}
					pf = ldf(pre_flag_ptr(k + 1)), asked = true;
				};
				auto look = [&] {
					if (next_in && !have && (!next_flags || nxt_ready == k + 1)) request_inputs(k + 1), have = true;
				};
				// wave 3 has most of the last two long stages to spare: one more look each, answer awaited, so that flags that come up during a
				// stage are acted upon at the start of the next one
				auto watch = [&] {
					if (!next_in || !next_flags || have || nxt_ready == k + 1) return;
					const int v = ldf(pre_flag_ptr(k + 1));
					if (__all(v - a.epoch >= 0))
					{
						if (lane == 0) nxt_ready = k + 1;
						request_inputs(k + 1), have = true;
					}
				};
				// With every diagonal tile's inverse coming out of its own sweep (the first one's from wave 1 beside the second sweep), the block rows of T_k
				// are a stage ahead of where a substitution per tile put them: row 3 is complete but for its last factor when the last sweep ends.
				if (w == 1) poll();
				if (w == 0) diag_chain<2, true>(S, rinv, lane, first_bad, TI);
				else if (w == 1) t_fin(1, 0);
				else if (w == 2) v_acc(2, 0, 0, true);
				else v_acc(3, 0, 0, true), store_tt_cols(k, 0, 16);
				lds_barrier();
				stamp(8);
				if (w == 1) poll();
				if (w == 0) look();
				if (w == 0) upd(3, 3, 2);
				else if (w == 1) v_acc(2, 0, 1, false);
				else if (w == 2) v_acc(2, 1, 1, true);
				else v_acc(3, 0, 1, false), v_acc(3, 1, 1, true);
				lds_barrier();
				stamp(9);
				if (w == 1) poll();
				if (w > 0) look();
				if (w == 0) diag_chain<3, true>(S, rinv, lane, first_bad, TI);
				else if (w == 1) t_fin(2, 0), v_acc(3, 0, 2, false); // LDS operations of one wave complete in order
				else if (w == 2) t_fin(2, 1), v_acc(3, 1, 2, false), store_tt_cols(k, 16, 16);
				else
				{
					v_acc(3, 2, 2, true);
					store_t_rows(k, 0, 0, 4), store_t_rows(k, 1, 0, 4); // complete since the stage before the last
					watch();
				}
				lds_barrier();
				stamp(10);
				look();
				if (w < 3) t_fin_out(k, w);
				else store_t_rows(k, 2, 0, 4), store_t_rows(k, 3, 48, 1);
				if (first_bad != 0 && t == 0) atomicCAS(a.info, 0, j0 + first_bad);
				lds_barrier();
				stamp(11);
				if (k + 1 == a.C1)
				{
					store_tt_cols(k, 32 + 8 * w, 8);
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					if (lane == 0) stf(dag_t(a, k) + w, a.epoch);
					if (a.Tt != nullptr && lane == 0) stf(dag_tt(a, k, k) + w, a.epoch);
				}
			}
			if (t == 0 && ldf(dag_err(a)) - a.epoch >= 0) atomicExch(a.info, -1);
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

	} // namespace

	// scheme of the factorisations issued by this host thread: -1 = the environment's, 0 = step, 1 = dag (CholSchemeScope: the recovery from a
	// give-up of the one-launch scheme, and the contexts the tests switch by gple_debug_chol_knobs)
	static thread_local int tl_chol_scheme = -1;
	CholSchemeScope::CholSchemeScope(int scheme): saved(tl_chol_scheme)
	{
		if (scheme >= 0) tl_chol_scheme = scheme;
	}
	CholSchemeScope::~CholSchemeScope() { tl_chol_scheme = saved; }
	static bool chol_dag_scheme()
	{
		static const bool v = [] {
			const char* e = getenv("GPLE_CHOL_SCHEME"); // "step": one launch per panel (rounds 2-3); "dag" (default): one launch per outer block
			return e == nullptr || std::string(e) != "step";
		}();
		return tl_chol_scheme >= 0 ? tl_chol_scheme == 1 : v;
	}
	// every layout function below is a pure function of (n, scheme): its cache is keyed by both, so that a factorisation repeated under the other
	// scheme in the same process lays its work out — and rounds — exactly like a process started with that scheme
	static long layout_key(int n) { return 2L * n + (chol_dag_scheme() ? 1 : 0); }
	// Two-level blocking.  A 64-wide panel step that updates the WHOLE trailing matrix reads and writes it once per panel:
	// 8 n^3 / (3 * 64) bytes in total, 2.9 GB at n = 4096 — the K = 64 updates run at HBM speed, not MFMA speed.  With outer blocks
	// the panel steps only update the rest of their own block (a strip), and the matrix right of the block gets ONE update with
	// K = the block's width per outer block.  In the one-launch panel step the strip's tiles run beside the panel workgroups for
	// free as long as they are done before the panel is (~14 us: about GPLE_CHOL_TILE_BUDGET = 800 tiles of 2-3 us on the CUs the
	// panel leaves idle), so every outer block is made as wide as that budget allows at its first panel — the blocks widen as the
	// trailing matrix shrinks (n = 4096: 832 + 1088 + 2176; below n ~ 2400 the whole matrix is one block), which also makes the
	// K of the separate updates large.  GPLE_CHOL_OUTER = <width> forces equal blocks (0: a single one) for A/B runs.
	static const std::vector<int>& chol_block_bounds(int n)
	{
		static const int forced = [] {
			const char* e = getenv("GPLE_CHOL_OUTER");
			return e ? atoi(e) : -1;
		}();
		static const int budget_env = [] {
			const char* e = getenv("GPLE_CHOL_TILE_BUDGET");
			return e && atoi(e) > 0 ? atoi(e) : 0;
		}();
		// one launch per outer block: the tile tasks of a block are not bound to a panel's duration, wider blocks save trailing updates
		// (n = 4096: 1.88 / 1.75 / 1.80 ms with 800 / 1600 / 2000; n = 8192: 8.80 / 8.43 / 8.57; one block up to n = 3648)
		const int budget = budget_env ? budget_env : (chol_dag_scheme() ? 1600 : 800);
		static const bool fused = [] {
			const char* e = getenv("GPLE_CHOL_FUSED");
			return e == nullptr || atoi(e) != 0;
		}();
		static std::mutex mu;
		static std::map<long, std::vector<int>> cache;
		std::lock_guard<std::mutex> lk(mu);
		auto it = cache.find(layout_key(n));
		if (it != cache.end()) return it->second;
		std::vector<int> b{0};
		if (forced >= 0 || !fused)
		{
			const int OB = forced >= NB ? forced / NB * NB : (forced < 0 && n >= 2048 ? 256 : 0);
			if (OB)
				for (int j = OB; j < n; j += OB) b.push_back(j);
		}
		else
			for (int J0 = 0; J0 < n;)
			{
				// widest block whose first strip (nc column blocks right of the panel, nr row blocks below it) stays within the budget
				const int nr = (n - J0) / NB - 1;
				int nc = 0;
				while (nc < nr && (nc + 1) * nr - (nc + 1) * nc / 2 <= budget) ++nc;
				int w = (nc + 1) * NB;
				if (w < 256) w = 256;
				if (n - (J0 + w) < 256) w = n - J0; // no sliver at the end
				J0 += w;
				if (J0 < n) b.push_back(J0);
			}
		b.push_back(n);
		return cache.emplace(layout_key(n), std::move(b)).first->second;
	}

	// panel steps of the block columns [j_begin, j_end) (multiples of NB; j_begin on an outer-block boundary or 0) of the n x n matrix
	// uvec != nullptr: A carries one more block row (rows n .. n + NB - 1, row n = the scaled labels y, the rest zero); it is factored
	// along as part of every panel, which leaves u = L^-1 y in its first row — collected into uvec — at no extra launch
	// on_final(j): called right after the launch that makes the columns [0, j) of the factor (and their T_jj) final, for every j in `marks`
	// flags of the one-launch-per-panel-range scheme (potrf_dag_kernel): the context's buffer, sized for the largest matrix seen so far, and
	// the epoch of this factorisation
	struct DagState
	{
		int* flags;
		int epoch;
		double* tt; // scratch for the inverse's tiles when the launch forms them (chol_dag_inverse_inside), n * n doubles
		double* pa; // scratch for the published pre-multiplication tiles (r, r - 2), 64 n doubles
		int poll_limit = 0, max_blocks = 0; // the context's debug knobs (0: defaults)
	};
	namespace
	{
		__global__ void spin_stamp_kernel(long long* out) { if (threadIdx.x == 0) *out = wall_clock64(); }
	} // namespace
	// Matrices of one outer block: the launch that factors them also forms T = L^-1 below the diagonal blocks — tile tasks of the same queue, every
	// sum complete but for its last factor when the row's T_rr arrives — instead of a merge tree of GEMM launches after the last panel
	// (n = 256 / 1024 / 2048: 25 / 50 / 90 us of the fit) and, from n = 1024 on, a second panel launch and the side stream's hand-overs.
	static bool chol_dag_inverse_inside(int n)
	{
		static const bool on = [] {
			const char* e = getenv("GPLE_CHOL_DAG_INVERSE"); // 0: the merge tree after the panels (A/B)
			return e == nullptr || atoi(e) != 0;
		}();
		return on && chol_dag_scheme() && n >= 2 * NB && chol_block_bounds(n).size() == 2;
	}
	static int chol_dag_max_blocks()
	{
		static const int v = [] {
			if (const char* e = getenv("GPLE_CHOL_DAG_BLOCKS"))
				if (atoi(e) >= 2) return atoi(e);
			int dev = 0, cus = 0;
			if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 2) cus = 64;
			return cus; // one workgroup per CU: more would only queue behind the resident ones (the ticket order needs no particular number resident)
		}();
		return v;
	}
	static size_t chol_dag_flag_ints(int n) { return 4 * (static_cast<size_t>(2 * (n / NB) + 6) * (n / NB) + 2); } // ticket counter, tile flags, three per column, error word, the inverse's tiles, one per row

	static const std::vector<int>& chol_marks(int n);
	constexpr int DAG_MAX_LATE_BLOCK = 32 * NB; // columns
	// the launches of the one-launch scheme: outer block boundaries and marks (a block boundary closer than 256 columns to a mark gives way to it)
	static std::vector<int> chol_dag_cuts(int n, const std::vector<int>* marks)
	{
		std::vector<int> cuts(chol_block_bounds(n));
		if (marks)
		{
			cuts.erase(std::remove_if(cuts.begin(), cuts.end(),
						   [&](int b) {
							   if (b == 0 || b == n) return false;
							   for (int m : *marks)
								   if (b != m && std::abs(b - m) < 256) return true;
							   // behind the first mark a row block of the inverse is one launch if it is not too wide: fewer rows below, lighter tile
							   // tasks, and the launch can form its row block's diagonal part of T (chol_block_inverse_inside)
							   if (!marks->empty() && b > marks->front())
							   {
								   int g0 = marks->front(), g1 = n;
								   for (int m : *marks)
								   {
									   if (m <= b) g0 = std::max(g0, m);
									   if (m > b) g1 = std::min(g1, m);
								   }
								   if (g1 - g0 <= DAG_MAX_LATE_BLOCK) return true;
							   }
							   return false;
						   }),
				cuts.end());
			cuts.insert(cuts.end(), marks->begin(), marks->end());
		}
		std::sort(cuts.begin(), cuts.end());
		cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
		return cuts;
	}
	// a launch [g0, g1) that is not the first and covers a whole row block of the inverse forms that row block's diagonal part of T itself (tile tasks,
	// as for one-block matrices) instead of leaving a merge tree of GEMM launches to the side stream / to the main stream behind the last panel
	static bool chol_block_inverse_inside(int n, int g0, int g1, const std::vector<int>& cuts)
	{
		static const bool on = [] {
			const char* e = getenv("GPLE_CHOL_DAG_BLOCK_INVERSE"); // 0: merge trees (A/B)
			return e == nullptr || atoi(e) != 0;
		}();
		if (!on || g0 == 0 || g1 - g0 > DAG_MAX_LATE_BLOCK) return false;
		const std::vector<int>& marks = chol_marks(n);
		const bool starts = std::find(marks.begin(), marks.end(), g0) != marks.end();
		const bool ends = g1 == n || std::find(marks.begin(), marks.end(), g1) != marks.end();
		if (!starts || !ends) return false;
		for (int c : cuts)
			if (c > g0 && c < g1) return false; // (cannot happen for consecutive cuts; kept for callers that pass row blocks)
		for (int m : marks)
			if (m > g0 && m < g1) return false;
		return true;
	}
	static hipError_t potrf_columns(hipStream_t s, double* A, long lda, int n, double* T, long ldt, int* info, int j_begin, int j_end, double* uvec,
		const std::vector<int>* marks = nullptr, const std::function<hipError_t(int)>* on_final = nullptr, const DagState* dag = nullptr)
	{
		const std::vector<int>& bounds = chol_block_bounds(n);
		auto at = [&](int r, int c) { return A + r + static_cast<long>(c) * lda; };
		// C(r0.., c0..) -= A(r0.., k0..k0+K) A(c0.., k0..k0+K)^T on the lower tiles of an m x ncols result whose (0,0) lies on the diagonal
		auto syrk_update = [&](int r0, int m, int ncols, int k0, int K) -> hipError_t {
			GemmDesc g{};
			g.A = at(r0, k0), g.lda = lda, g.B = at(r0, k0), g.ldb = lda, g.C = at(r0, r0), g.ldc = lda;
			g.M = m, g.N = ncols, g.K = K, g.batch = 1, g.alpha = -1.0, g.beta = 1.0, g.krange = K_FULL, g.lower_only = 1;
			g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
			// (128-tiles by the dense rule — every computed tile has the same k-range here — measured slower: 2.73 vs 2.67 ms at
			// n = 4096, 12.25 vs 11.16 at 8192; K = 256 is too short for that kernel's two-slab pipeline)
			return launch_gemm(s, g, gemm_pick_tile(m, ncols, 1, true));
		};
		static const bool fused = [] {
			const char* e = getenv("GPLE_CHOL_FUSED");
			return e == nullptr || atoi(e) != 0;
		}();
		if (dag != nullptr)
		{
			// outer block by outer block — the marks are block boundaries too, so that a launch never continues sums another launch began —:
			// its panels in one launch, then the matrix right of it in one update
			const int FS = n / NB, R = FS + (uvec ? 1 : 0);
			const std::vector<int> cuts = chol_dag_cuts(n, marks);
			if (cuts.size() > DAG_MAX_LAUNCHES) return hipErrorInvalidValue; // (n > 256k columns: the ticket floor packs the launch number into 12 bits)
			for (size_t bi = 0; bi + 1 < cuts.size(); ++bi)
			{
				const int J0 = cuts[bi], Jend = cuts[bi + 1];
				if (J0 < j_begin || Jend > j_end) continue;
				DagArgs g{};
				g.A = A, g.lda = lda, g.T = T, g.ldt = ldt, g.info = info, g.uvec = uvec, g.flags = dag->flags + 4, g.FS = FS, g.R = R;
				g.c0 = J0 / NB, g.C1 = Jend / NB, g.epoch = dag->epoch;
				g.poll_limit = dag->poll_limit > 0 ? dag->poll_limit : DAG_POLL_LIMIT_DEFAULT;
				g.pa = dag->pa;
				// the whole inverse of a one-block matrix; otherwise the diagonal row block of T of a launch that is exactly one row block of the
				// inverse — except the first, whose tile tasks are busy enough (chol_block_inverse_inside)
				g.Tt = dag->tt != nullptr && ((J0 == 0 && Jend == n) || chol_block_inverse_inside(n, J0, Jend, cuts)) ? dag->tt : nullptr;
				g.tt_ld = (Jend - J0) / NB;
				g.nunits = dag_count_units(g.c0, g.C1, R, g.Tt != nullptr);
				g.seq = static_cast<int>(bi);
				// once the side stream is at work (from the first mark on) the launch leaves part of the chip to its GEMMs: a workgroup of this kernel
				// holds 70 KB of LDS on its CU whether it works or waits, which halves the GEMM workgroups that fit beside it
				static const int late_blocks = [] {
					const char* e = getenv("GPLE_CHOL_DAG_LATE_BLOCKS");
					return e ? atoi(e) : 64; // 0: no limit.  n = 4096: 1.85 / 1.85 / 1.79 / 1.81 ms with 256 / 128 / 64 / 32 workgroups
				}();
				const bool side_busy = marks != nullptr && !marks->empty() && J0 >= marks->front();
				const int all_blocks = dag->max_blocks >= 2 ? dag->max_blocks : chol_dag_max_blocks();
				const int max_blocks = side_busy && late_blocks >= 2 ? std::min(late_blocks, all_blocks) : all_blocks;
				const int helpers = std::min(max_blocks - 1, g.nunits);
				constexpr int LAST_STAMP = 11;
				static const bool want_stamps = getenv("GPLE_CHOL_DAG_STAMPS") != nullptr;
				static long long* stamp_buf = nullptr;
				if (want_stamps && stamp_buf == nullptr && hipMalloc(reinterpret_cast<void**>(&stamp_buf), 16 * 1024 * sizeof(long long)) != hipSuccess) stamp_buf = nullptr;
				g.stamps = want_stamps && FS <= 1000 ? stamp_buf : nullptr;
				if (g.stamps)
				{
					const hipError_t e = hipMemsetAsync(stamp_buf, 0, 16 * 1024 * sizeof(long long), s);
					if (e != hipSuccess) return e;
					hipLaunchKernelGGL(spin_stamp_kernel, dim3(1), dim3(64), 0, s, stamp_buf + 16 * 1023);
				}
				hipLaunchKernelGGL(potrf_dag_kernel, dim3(1 + helpers), dim3(256), 0, s, g);
				if (g.stamps)
				{
					hipLaunchKernelGGL(spin_stamp_kernel, dim3(1), dim3(64), 0, s, stamp_buf + 16 * 1023 + 1);
					std::vector<long long> h(16 * 1024);
					if (hipStreamSynchronize(s) == hipSuccess && hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess)
					{
						fprintf(stderr, "dag launch: panels [%d, %d), %d units, %d workgroups: %.2f us between the stamps around it\n", g.c0, g.C1, g.nunits, 1 + helpers,
							(h[16 * 1023 + 1] - h[16 * 1023]) * 0.01);
						for (int k = g.c0; k < g.C1; ++k)
						{
							const long long* q = h.data() + 16 * k;
							fprintf(stderr, "  panel %3d: start %8.2f |", k, (q[0] - h[16 * 1023]) * 0.01);
							for (int i = 1; i <= LAST_STAMP; ++i) fprintf(stderr, " %5.2f", (q[i] - q[i - 1]) * 0.01);
							fprintf(stderr, " | step %6.2f", (q[LAST_STAMP] - q[0]) * 0.01);
							if (k >= g.c0 + 2 && q[15] != 0)
							{
								const long long* pq = h.data() + 16 * (k - 1);
								fprintf(stderr, " | diagonal pre-tile of this row, after the start of the panel before: published tile seen %6.2f, T seen %6.2f, done %6.2f", (q[13] - pq[0]) * 0.01,
									(q[14] - pq[0]) * 0.01, (q[15] - pq[0]) * 0.01);
							}
							fprintf(stderr, "\n");
						}
					}
				}
				if (marks && std::find(marks->begin(), marks->end(), Jend) != marks->end())
				{
					const hipError_t e = (*on_final)(Jend);
					if (e != hipSuccess) return e;
				}
				if (n - Jend > 0)
				{
					const int extra = uvec ? NB : 0;
					const hipError_t e = syrk_update(Jend, n + extra - Jend, n - Jend, J0, Jend - J0);
					if (e != hipSuccess) return e;
				}
			}
			return hipGetLastError();
		}
		bool pend = false; // the rank-64 update by the previous panel has not been applied yet (fused scheme: it rides in the next launch)
		for (int j0 = j_begin; j0 < j_end; j0 += NB)
		{
			size_t bi = 0;
			while (bounds[bi + 1] <= j0) ++bi;
			const int J0 = bounds[bi], Jend = bounds[bi + 1]; // outer block of this panel
			const int extra = uvec ? NB : 0;
			const int m = n + extra - j0; // rows of the panel including the diagonal block
			const int below = m - NB;
			const int ndt = below > 0 ? below / NB : 1;
			const int strip = Jend - (j0 + NB); // columns of this block column right of the panel
			if (fused)
			{
				// one launch: the panel (with the previous panel's update of its own columns, if pending) + the rest of that update
				const int sy_nc = pend ? strip / NB : 0, sy_nr = pend ? below / NB : 0;
				const int ntiles = sy_nc * sy_nr - sy_nc * (sy_nc - 1) / 2;
				hipLaunchKernelGGL(potrf_step_kernel<false>, dim3(ndt + (ntiles + 3) / 4), dim3(512), 0, s, A, lda, T, ldt, info, j0, below, ndt, pend ? 1 : 0, sy_nc, sy_nr, uvec,
					static_cast<long long*>(nullptr));
				if (marks && std::find(marks->begin(), marks->end(), j0 + NB) != marks->end())
				{
					const hipError_t e = (*on_final)(j0 + NB);
					if (e != hipSuccess) return e;
				}
				pend = strip > 0;
				if (pend && j0 + NB >= j_end) // nobody comes after this panel in this call: apply its update now
				{
					const hipError_t e = syrk_update(j0 + NB, below, strip, j0, NB);
					if (e != hipSuccess) return e;
					pend = false;
				}
			}
			else
			{
				// diagonal block + the rows below it: one launch, one workgroup per 64 panel rows (each re-does the diagonal block)
				double* Tjj = T + j0 + static_cast<long>(j0) * ldt;
				hipLaunchKernelGGL(potrf_diag_kernel<false>, dim3(ndt), dim3(256), 0, s, at(j0, j0), lda, Tjj, ldt, info, j0, static_cast<long long*>(nullptr),
					at(j0 + (below > 0 ? NB : 0), j0), below, uvec);
				if (marks && std::find(marks->begin(), marks->end(), j0 + NB) != marks->end())
				{
					const hipError_t e = (*on_final)(j0 + NB);
					if (e != hipSuccess) return e;
				}
				// the rest of this block column: rows j0 + NB .. n, columns j0 + NB .. Jend
				if (strip > 0)
				{
					const hipError_t e = syrk_update(j0 + NB, below, strip, j0, NB);
					if (e != hipSuccess) return e;
				}
			}
			// last panel of an outer block: everything right of the block column, once, with K = the block's width
			if (j0 + NB == Jend && n - Jend > 0)
			{
				const hipError_t e = syrk_update(Jend, n + extra - Jend, n - Jend, J0, Jend - J0);
				if (e != hipSuccess) return e;
			}
		}
		return hipGetLastError();
	}

	// probe entry (probes/diag_probe.py): one instrumented launch of the diagonal-block kernel on device buffers
	hipError_t debug_potrf_diag(hipStream_t s, const double* A, double* T, int* info, long long* stamps)
	{
		hipLaunchKernelGGL(potrf_diag_kernel<true>, dim3(1), dim3(256), 0, s, A, 64L, T, 64L, info, 0, stamps, const_cast<double*>(A), 0, static_cast<double*>(nullptr));
		return hipGetLastError();
	}

	// probe entry: one instrumented launch of the one-launch panel step at j0 = 64 of an n x n matrix (n = 128 + below)
	hipError_t debug_potrf_step(hipStream_t s, double* A, long lda, double* T, long ldt, int* info, long long* stamps, int pend, int below)
	{
		const int ndt = below > 0 ? below / NB : 1;
		hipLaunchKernelGGL(potrf_step_kernel<true>, dim3(ndt), dim3(512), 0, s, A, lda, T, ldt, info, NB, below, ndt, pend, 0, 0, static_cast<double*>(nullptr), stamps);
		return hipGetLastError();
	}

	// the context's flag buffer for the one-launch scheme, grown to the matrix at hand, and a fresh epoch
	static hipError_t dag_state(Ctx* ctx, hipStream_t s, int n, DagState& st)
	{
		const size_t need = chol_dag_flag_ints(n);
		if (ctx->dag_flags_ints < need)
		{
			if (ctx->dag_flags) (void)hipFree(ctx->dag_flags);
			ctx->dag_flags = nullptr, ctx->dag_flags_ints = 0;
			hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->dag_flags), need * sizeof(int));
			if (e != hipSuccess) return e;
			if ((e = hipMemsetAsync(ctx->dag_flags, 0, need * sizeof(int), s)) != hipSuccess) return e;
			ctx->dag_flags_ints = need;
			ctx->dag_epoch = 0;
		}
		// epochs are compared as ints and multiplied by DAG_MAX_LAUNCHES in the high word of the ticket counter: long before either runs out (2^18
		// factorisations of one context) the buffer is cleared — in stream order, like everything that uses it — and the count starts again
		static const int epoch_limit = [] {
			const char* e = getenv("GPLE_CHOL_DAG_EPOCH_LIMIT"); // (tests)
			return e && atoi(e) > 0 ? std::min(atoi(e), 1 << 18) : 1 << 18;
		}();
		if (ctx->dag_epoch >= epoch_limit)
		{
			const hipError_t e = hipMemsetAsync(ctx->dag_flags, 0, ctx->dag_flags_ints * sizeof(int), s);
			if (e != hipSuccess) return e;
			ctx->dag_epoch = 0;
		}
		st.flags = ctx->dag_flags;
		st.epoch = ++ctx->dag_epoch;
		st.poll_limit = ctx->dag_poll_limit, st.max_blocks = ctx->dag_blocks;
		return hipSuccess;
	}

	hipError_t potrf_lower(hipStream_t s, double* A, long lda, int n, double* T, long ldt, int* info, double* uvec, Ctx* ctx, double* tt, double* pa)
	{
		if (n % NB) return hipErrorInvalidValue;
		DagState st{};
		st.tt = tt, st.pa = pa;
		const bool use_dag = ctx != nullptr && pa != nullptr && chol_dag_scheme();
		if (use_dag)
		{
			const hipError_t e = dag_state(ctx, s, n, st);
			if (e != hipSuccess) return e;
		}
		const hipError_t e = potrf_columns(s, A, lda, n, T, ldt, info, 0, n, uvec, nullptr, nullptr, use_dag ? &st : nullptr);
		if (e != hipSuccess) return e;
		return hipGetLastError();
	}

	hipError_t trtri_lower_from_diag(hipStream_t s, const double* L, long ldl, double* T, long ldt, int n, double* work, int origin)
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
				int tile = gemm_pick_tile(s2, s1, count, true);
				if (tile == 128 && (origin + start) % 128) tile = 64; // see tri_tile
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

	// Smallest n for which the inverse runs beside the factorisation on the side stream (an event hand-over costs the main stream ~6 us)
	static int chol_overlap_min_n()
	{
		static const int v = [] {
			const char* e = getenv("GPLE_CHOL_OVERLAP_MIN_N");
			return e ? atoi(e) : 1024;
		}();
		return v;
	}
	// The inverse by block rows, beside the factorisation.  T = L^-1 row block by row block: once the panels of the column range
	// [g0, g1) are done, everything its row block of T needs is final — T_gg from the merge tree over its diagonal blocks, then
	// T(g, 0..g0) = -T_gg (L(g, 0..g0) T(0..g0, 0..g0)), two triangular-k GEMMs — so the side stream of the context works through the row
	// blocks while the main stream keeps factoring, and only the last row block is left when the last panel is done.  The first form of
	// this (one split in the middle, both halves' trees + a 2048-wide join after the last panel at n = 4096) left 0.5 ms behind the last
	// panel; the work of a row block grows with g0^2, so the groups shrink towards the end: fork points at 60 % and 80 % of n, from n = 8192 on also at 90 %
	// (GPLE_CHOL_FORKS = comma-separated percentages for A/B runs), none closer than 256 columns to its neighbours.
	static const std::vector<int>& chol_fork_points(int n)
	{
		static const std::vector<int> pct = [] {
			std::vector<int> v;
			if (const char* e = getenv("GPLE_CHOL_FORKS"))
			{
				for (const char* p = e; *p;)
				{
					v.push_back(atoi(p));
					while (*p && *p != ',') ++p;
					if (*p == ',') ++p;
				}
			}
			else v = {60, 80};
			return v;
		}();
		static std::mutex mu;
		static std::map<long, std::vector<int>> cache;
		std::lock_guard<std::mutex> lk(mu);
		auto it = cache.find(layout_key(n));
		if (it != cache.end()) return it->second;
		std::vector<int> f;
		std::vector<int> use = pct;
		if (getenv("GPLE_CHOL_FORKS") == nullptr && n >= 8192 && !chol_dag_scheme()) use.push_back(90); // launch per panel: a third, late fork pays from here on (9.20 vs 9.34 ms at n = 8192; one launch per outer block: 8.96 vs 8.78)
		int next = n; // from the last one down: the late fork matters most
		for (auto it2 = use.rbegin(); it2 != use.rend(); ++it2)
		{
			const int j = static_cast<int>(static_cast<long>(n) * *it2 / 100) / NB * NB;
			if (*it2 <= 0 || *it2 >= 100 || next - j < 256 || j < 256) continue;
			f.insert(f.begin(), j), next = j;
		}
		return cache.emplace(layout_key(n), std::move(f)).first->second;
	}
	// Workspace of chol_inverse_factor, from the fork list actually in use (GPLE_CHOL_FORKS may put the forks anywhere): W = L(g, 0..g0) T(0..g0)
	// of the widest row block product, the merge tree of the widest side job, the merge tree of the last row block (a tree over b columns needs
	// b^2 / 4 doubles: count * s1 * s2 per level); the unsplit path uses one tree of n^2 / 4 at the front
	// The row blocks of the inverse = where the side stream is handed finished columns: the fork points (GPLE_CHOL_MARKS=cuts: every outer block
	// boundary of the one-launch scheme as well — a launch ends there anyway —, measured slower: the side stream's GEMMs run at a fraction of
	// their speed beside a panel launch, and more, smaller ones do worse).  Empty: no overlap (small matrices).
	static const std::vector<int>& chol_marks(int n)
	{
		static std::mutex mu;
		static std::map<long, std::vector<int>> cache;
		std::lock_guard<std::mutex> lk(mu);
		auto it = cache.find(layout_key(n));
		if (it != cache.end()) return it->second;
		std::vector<int> m;
		const std::vector<int>& forks = chol_fork_points(n);
		if (!chol_dag_inverse_inside(n) && n >= chol_overlap_min_n() && n / NB >= 4 && !forks.empty())
		{
			m = forks;
			static const bool all_cuts = [] {
				const char* e = getenv("GPLE_CHOL_MARKS"); // "cuts": every outer block boundary as well (A/B: 1.86 vs 1.83 ms at n = 4096, 9.15 vs 8.91 at 8192)
				return e != nullptr && std::string(e) == "cuts";
			}();
			if (chol_dag_scheme() && all_cuts)
			{
				for (int b : chol_block_bounds(n))
				{
					bool keep = b > 0 && b < n;
					for (int f : forks) keep = keep && std::abs(b - f) >= 256;
					if (keep) m.push_back(b);
				}
				std::sort(m.begin(), m.end());
			}
		}
		return cache.emplace(layout_key(n), std::move(m)).first->second;
	}
	struct InvWork
	{
		size_t prod, side, main, tt;
	};
	// prod: W(rows below the first mark, columns left of the last one), accumulated block row by block row (chol_inverse_factor); side / main: the
	// merge trees of the widest side job / of the last row block (b^2 / 4 doubles for b columns); without marks one tree over all n columns
	static InvWork chol_inverse_work_split(int n)
	{
		const auto sq4 = [](size_t b) { return b * b / 4; };
		InvWork w{0, 0, 0, 0};
		const std::vector<int>& marks = chol_marks(n);
		if (marks.empty())
		{
			w.prod = chol_dag_inverse_inside(n) ? static_cast<size_t>(n) * n : sq4(static_cast<size_t>(n)); // the transposed tiles of the inverse / one tree
			return w;
		}
		size_t done = 0;
		for (int j : marks)
		{
			w.side = std::max(w.side, sq4(static_cast<size_t>(j) - done));
			done = static_cast<size_t>(j);
		}
		w.prod = (static_cast<size_t>(n) - marks.front()) * static_cast<size_t>(marks.back());
		w.main = sq4(static_cast<size_t>(n) - done);
		if (chol_dag_scheme()) // row blocks whose launch forms their diagonal part of T: transposed tiles of the widest one
		{
			const std::vector<int> cuts = chol_dag_cuts(n, &marks);
			int g0 = 0;
			for (size_t i = 0; i <= marks.size(); ++i)
			{
				const int g1 = i < marks.size() ? marks[i] : n;
				if (chol_block_inverse_inside(n, g0, g1, cuts)) w.tt = std::max(w.tt, static_cast<size_t>(g1 - g0) * (g1 - g0));
				g0 = g1;
			}
		}
		return w;
	}
	size_t chol_inverse_work_doubles(int n)
	{
		const InvWork w = chol_inverse_work_split(n);
		return w.prod + w.side + w.main + 64 + static_cast<size_t>(NB) * n + w.tt; // 64 n: potrf_dag_kernel's published tiles; then its transposed tiles
	}
	static double* chol_work_tt(double* work, int n)
	{
		const InvWork w = chol_inverse_work_split(n);
		return w.tt ? work + w.prod + w.side + w.main + 64 + static_cast<size_t>(NB) * n : nullptr;
	}
	static double* chol_work_pa(double* work, int n)
	{
		const InvWork w = chol_inverse_work_split(n);
		return work + w.prod + w.side + w.main + 64;
	}
	// 128-tile GEMMs with a triangular k-range start at their 128-aligned diagonal tile and take in the block above the second 64-block of the
	// pair, which the panel step zeroes for GLOBALLY odd 64-blocks only: a sub-matrix whose origin is an odd multiple of 64 stays on 64-tiles
	static int tri_tile(long m, long ncols, long batch, int origin)
	{
		const int t = gemm_pick_tile(m, ncols, batch, true);
		return t == 128 && origin % 128 ? 64 : t;
	}
	// The side stream must sit on another hardware queue than the main stream.  HIP binds its streams to a handful of hardware queues
	// (GPU_MAX_HW_QUEUES, 4 by default) as they are created, by use count: whether the stream created here shares the main stream's queue
	// depends on how many streams the process made before — torch's, RCCL's — and when it does, the block-row inverse runs after the panels
	// instead of beside them, or worse: N = 4096 fit 2.0 -> 2.4 ms (default priority) / 3.9 ms (low priority) with an RCCL communicator
	// created first (probes/hwqueue_fit_probe.py, profiles/r03_notes.md).  So candidates are tried: a short spin kernel on each of the two
	// streams, started together; on one queue they take twice as long as on two.  Rejected candidates are kept alive until one is accepted
	// (their queues stay taken, so the next candidate goes elsewhere), then destroyed.  Once per context, ~0.3 ms per candidate.
	namespace
	{
		__global__ void spin_kernel(long long ticks)
		{
			const long long t0 = wall_clock64();
			while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
		}
	} // namespace
	static hipError_t pick_side_stream(Ctx* ctx, hipStream_t main_stream)
	{
		// the side stream's GEMMs fill whatever the panel launches leave idle and must not be dispatched ahead of them: lowest priority
		static const int side_prio = [] {
			const char* ev = getenv("GPLE_CHOL_SIDE_PRIORITY"); // 0: default priority (A/B)
			int lo = 0, hi = 0;
			if ((ev && atoi(ev) == 0) || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return 0;
			return lo;
		}();
		static const bool probe = [] {
			const char* ev = getenv("GPLE_CHOL_SIDE_PROBE"); // 0: take the first stream whatever queue it is on (A/B)
			return ev == nullptr || atoi(ev) != 0;
		}();
		constexpr int MAX_CANDIDATES = 8;
		constexpr long long SPIN_TICKS = 10000; // 100 us of the 100 MHz wall clock
		hipError_t e;
		hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
		for (hipEvent_t& x : ev)
			if ((e = hipEventCreate(&x)) != hipSuccess) return e;
		std::vector<hipStream_t> rejected;
		hipStream_t chosen = nullptr;
		for (int attempt = 0; attempt < MAX_CANDIDATES && !chosen; ++attempt)
		{
			hipStream_t cand = nullptr;
			if ((e = hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, side_prio)) != hipSuccess) break;
			ctx->side_attempts = attempt + 1;
			if (!probe)
			{
				chosen = cand;
				break;
			}
			// both spins start at ev[0]; ev[1] / ev[2] close the main / the candidate's one
			float t_main = 0.f, t_side = 0.f, t_ref = 0.f;
			bool ok = hipEventRecord(ev[0], main_stream) == hipSuccess && hipStreamWaitEvent(cand, ev[0], 0) == hipSuccess;
			if (ok)
			{
				hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, main_stream, SPIN_TICKS);
				hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cand, SPIN_TICKS);
				ok = hipEventRecord(ev[1], main_stream) == hipSuccess && hipEventRecord(ev[2], cand) == hipSuccess && hipStreamSynchronize(cand) == hipSuccess
					&& hipStreamSynchronize(main_stream) == hipSuccess && hipEventElapsedTime(&t_main, ev[0], ev[1]) == hipSuccess
					&& hipEventElapsedTime(&t_side, ev[0], ev[2]) == hipSuccess;
			}
			// one spin alone, for the scale (launch latency included)
			if (ok)
			{
				ok = hipEventRecord(ev[0], main_stream) == hipSuccess;
				hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, main_stream, SPIN_TICKS);
				ok = ok && hipEventRecord(ev[1], main_stream) == hipSuccess && hipStreamSynchronize(main_stream) == hipSuccess
					&& hipEventElapsedTime(&t_ref, ev[0], ev[1]) == hipSuccess;
			}
			if (!ok || std::max(t_main, t_side) < 1.6f * t_ref)
			{
				chosen = cand; // (a failed measurement is not a reason to go without a side stream)
				ctx->side_overlaps = ok;
			}
			else rejected.push_back(cand);
		}
		if (!chosen && !rejected.empty()) // every candidate shared the main stream's queue: the last one will have to do
		{
			chosen = rejected.back();
			rejected.pop_back();
		}
		for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
		for (hipEvent_t x : ev) (void)hipEventDestroy(x);
		(void)hipGetLastError();
		if (!chosen) return hipErrorOutOfMemory;
		ctx->side_stream = chosen;
		return hipSuccess;
	}

	hipError_t chol_inverse_factor(Ctx* ctx, hipStream_t s, double* A, long lda, int n, double* T, long ldt, int* info, double* work, double* uvec)
	{
		if (n % NB) return hipErrorInvalidValue;
		ctx->fit_early_rows = 0;
		const std::vector<int>& marks = chol_marks(n);
		if (marks.empty())
		{
			if (chol_dag_inverse_inside(n)) return potrf_lower(s, A, lda, n, T, ldt, info, uvec, ctx, work, chol_work_pa(work, n)); // T complete when the launch ends
			hipError_t e = potrf_lower(s, A, lda, n, T, ldt, info, uvec, ctx, nullptr, chol_work_pa(work, n));
			if (e != hipSuccess) return e;
			return trtri_lower_from_diag(s, A, lda, T, ldt, n, work);
		}
		hipError_t e;
		if (!ctx->side_stream)
		{
			if ((e = pick_side_stream(ctx, s)) != hipSuccess) return e;
			if ((e = hipEventCreateWithFlags(&ctx->side_join, hipEventDisableTiming)) != hipSuccess) return e;
		}
		while (ctx->side_forks.size() < marks.size())
		{
			hipEvent_t ev;
			if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return e;
			ctx->side_forks.push_back(ev);
		}
		hipStream_t side = ctx->side_stream;
		const InvWork iw = chol_inverse_work_split(n);
		// T = L^-1 by block rows, right-looking in W: T(g, 0..g0) = -T_gg W(g, 0..g0) with W(g, .) = sum over the row blocks m above g of
		// L(g, m) T(m, .).  As soon as row block m of T is complete its term goes into the W of EVERY row below — so what is left behind the
		// last panel is the last row block's own tree and one product, not the whole history of its W (round 3: W of the last rows was formed in
		// one piece after the last fork, 0.2 ms of side stream work the main stream waited for at n = 4096).
		const int f1 = marks.front();
		const long ldw = n - f1;
		double* w_all = work;                      // W(r, c) at (r - f1) + c * ldw, rows below the first mark
		double* w_side = work + iw.prod;           // merge tree of a side job
		double* w_main = work + iw.prod + iw.side; // merge tree of the last row block (main stream, beside the side's last job)
		auto tree = [&](hipStream_t st, int g0, int g1, double* w_tree) -> hipError_t {
			return trtri_lower_from_diag(st, A + g0 + static_cast<long>(g0) * lda, lda, T + g0 + static_cast<long>(g0) * ldt, ldt, g1 - g0, w_tree, g0);
		};
		// W(rows j .. n, .) (+)= L(rows, g0 .. g1) T(g0 .. g1, .): columns left of the block dense (T(g, 0..g0), just completed), the block's own
		// columns against its triangular diagonal block (their first term)
		auto w_accumulate = [&](hipStream_t st, int g0, int g1) -> hipError_t {
			GemmDesc g{};
			g.A = A + g1 + static_cast<long>(g0) * lda, g.lda = lda, g.ldb = ldt, g.ldc = ldw;
			g.M = n - g1, g.K = g1 - g0, g.batch = 1, g.alpha = 1.0;
			g.lower_only = 0, g.a_kmajor = false, g.b_kmajor = true, g.c_trans = false;
			if (g0 > 0)
			{
				g.B = T + g0, g.C = w_all + (g1 - f1), g.N = g0, g.beta = 1.0, g.krange = K_FULL;
				const hipError_t er = launch_gemm(st, g, gemm_pick_tile(g.M, g.N, 1, false));
				if (er != hipSuccess) return er;
			}
			g.B = T + g0 + static_cast<long>(g0) * ldt, g.C = w_all + (g1 - f1) + static_cast<long>(g0) * ldw, g.N = g1 - g0, g.beta = 0.0, g.krange = K_GE_N;
			return launch_gemm(st, g, tri_tile(g.M, g.N, 1, g0));
		};
		auto t_product = [&](hipStream_t st, int g0, int g1) -> hipError_t {
			GemmDesc g{};
			g.A = T + g0 + static_cast<long>(g0) * ldt, g.lda = ldt, g.B = w_all + (g0 - f1), g.ldb = ldw, g.C = T + g0, g.ldc = ldt;
			g.M = g1 - g0, g.N = g0, g.K = g1 - g0, g.batch = 1, g.alpha = -1.0, g.beta = 0.0;
			g.krange = K_LE_M, g.lower_only = 0, g.a_kmajor = false, g.b_kmajor = true, g.c_trans = false;
			return launch_gemm(st, g, tri_tile(g1 - g0, g0, 1, g0));
		};
		int done = 0;
		size_t nfork = 0;
		const bool use_dag = chol_dag_scheme();
		const std::vector<int> dag_cuts = use_dag ? chol_dag_cuts(n, &marks) : std::vector<int>{};
		const std::function<hipError_t(int)> on_final = [&](int j) -> hipError_t {
			hipError_t er;
			hipEvent_t ev = ctx->side_forks[nfork++];
			if ((er = hipEventRecord(ev, s)) != hipSuccess) return er;
			if ((er = hipStreamWaitEvent(side, ev, 0)) != hipSuccess) return er;
			const bool inside = use_dag && chol_block_inverse_inside(n, done, j, dag_cuts);
			if (!inside && (er = tree(side, done, j, w_side)) != hipSuccess) return er; // (else: formed by the block's launch)
			// The first row block of T is complete here: T(0 .. j, .) — behind the fork where its launch formed it, behind its merge tree on the side stream
			// otherwise.  A predict may start on those rows (Ctx::fit_early_*).
			if (done == 0)
			{
				if (inside) ctx->fit_early_event = ev;
				else
				{
					if (!ctx->fit_early_own && (er = hipEventCreateWithFlags(&ctx->fit_early_own, hipEventDisableTiming)) != hipSuccess) return er;
					if ((er = hipEventRecord(ctx->fit_early_own, side)) != hipSuccess) return er;
					ctx->fit_early_event = ctx->fit_early_own;
				}
				ctx->fit_early_rows = j;
			}
			if (done > 0 && (er = t_product(side, done, j)) != hipSuccess) return er;
			if ((er = w_accumulate(side, done, j)) != hipSuccess) return er;
			done = j;
			return hipSuccess;
		};
		DagState dst{};
		dst.pa = chol_work_pa(work, n);
		dst.tt = chol_work_tt(work, n);
		if (use_dag && (e = dag_state(ctx, s, n, dst)) != hipSuccess) return e;
		if ((e = potrf_columns(s, A, lda, n, T, ldt, info, 0, n, uvec, &marks, &on_final, use_dag ? &dst : nullptr)) != hipSuccess) return e;
		if ((e = hipEventRecord(ctx->side_join, side)) != hipSuccess) return e;
		// the last row block: its tree (unless its launch formed it) does not need the side's results, the last product does
		if (!(use_dag && chol_block_inverse_inside(n, done, n, dag_cuts)) && (e = tree(s, done, n, w_main)) != hipSuccess) return e;
		if ((e = hipStreamWaitEvent(s, ctx->side_join, 0)) != hipSuccess) return e;
		return t_product(s, done, n);
	}

	// host-only view of the factorisation's layout for n columns (tests/test_host_logic.py through gple_debug_chol_layout): outer block
	// bounds, fork points of the inverse, workspace doubles
	void chol_layout(int n, std::vector<int>& bounds, std::vector<int>& forks, size_t& work_doubles)
	{
		bounds = chol_block_bounds(n);
		forks = chol_marks(n);
		work_doubles = chol_inverse_work_doubles(n);
	}

	hipError_t lauum_full(hipStream_t s, const double* T, long ldt, double* W, long ldw, int n)
	{
		GemmDesc g{};
		g.A = T, g.lda = ldt, g.B = T, g.ldb = ldt, g.C = W, g.ldc = ldw;
		g.M = n, g.N = n, g.K = n, g.batch = 1, g.alpha = 1.0, g.beta = 0.0;
		g.krange = K_GE_MAX_MN, g.lower_only = 1, g.a_kmajor = true, g.b_kmajor = true, g.c_trans = false;
		// the k-range of tile (m0, n0) starts at max(m0, n0): the work per tile is very uneven and the first tile column
		// carries the full K, so small tiles (4x shorter critical path) win until the matrix is large
		hipError_t e = launch_gemm(s, g, gemm_pick_tile(n, n, 1, true));
		if (e != hipSuccess) return e;
		hipLaunchKernelGGL(mirror_lower_kernel, dim3(n / 32, n / 32), dim3(256), 0, s, W, ldw, n);
		return hipGetLastError();
	}
} // namespace gple
