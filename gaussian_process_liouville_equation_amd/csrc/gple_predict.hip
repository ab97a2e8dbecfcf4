// gple_predict.hip — GP prediction kernels for gfx950 (the dominant part of the fit+predict step).
//
// Reference hot loop 4 (kernel.cpp:495-518, complex_kernel.cpp:608-642):
//     mu_i  = K*_i v                        (M x N Gram row times weights)
//     var_i = k(x*,x*) - K*_i K^-1 K*_i^T   (one N x N GEMV + dot per test point, 2 M N^2 flops)
// The reference materialises the whole M x N matrix K* and the explicit inverse.  Here
//   * K^-1 = T^T T with T = chol(K)^-1 lower triangular, so  K*_i K^-1 K*_i^T = || T K*_i^T ||^2  — a triangular
//     contraction (M N^2 flops) whose result is a sum of squares (no cancellation inside the quadratic form);
//   * kstar_gen_kernel writes one bounded chunk of K* (<= PREDICT_SCRATCH_BYTES, column-major chunk_rows x n) to a
//     rolling HBM scratch and reduces the mean on the way (pure VALU: one exp per element, once);
//   * rownorm_kernel streams that chunk and T through LDS into v_mfma_f64_16x16x4_f64 and keeps only the squared
//     row norms.
// Why two kernels (measured on MI355X, see profiles/ and probes/): fp64 MFMA and fp64 VALU share one pipe, every
// VALU instruction of a co-resident wave adds ~4.4 cycles to the MFMA stream, and a fused kernel has to re-generate
// each K* element once per 256-column tile of T it meets (N/512 times on average, 32 fp64 instructions each).
// Fused variants (lock-step, interleaved, wave-specialised ping-pong) all stalled at 52 TFLOP/s at N=4096; the plain
// lock-step MFMA loop with LDS operand reads sustains 72.6 TFLOP/s on its own.
//
// The contraction kernels, oldest first — all on the same 128 x 256 tile, K advancing 16 per barrier, the same virtual-group sums, the newer ones
// bit-identical to their predecessor with the same blocking: rownorm_kernel (round 1; GPLE_ROWNORM_VARIANT=0), rownorm2_kernel (round 2: 4 x 4 / 2 x 8
// fragments per wave, slabs by LDS-DMA; GPLE_ROWNORM_PIPE=0), rownormp_kernel (round 4, the default: the k-steps of a unit as one pipeline pinned
// around the barrier), rownorm3_kernel (n <= 512 with few row blocks) and predict_fused256_kernel (round 4: the whole predict of a real fit with
// N <= 256 in one launch, K* generated inside).
//
// rownorm_kernel: one workgroup = 8 waves owns 128 test rows and loops over 256-wide N-tiles of T; K advances 16 per
// step through a double-buffered LDS stage (global -> registers -> LDS, loads issued before the MFMAs of the current
// step).  Every wave owns 16 rows x all 256 columns (16 accumulator tiles), so the all-zero blocks of T (k > n) are
// skipped by all waves alike; the skipping is expressed as consecutive loops with a compile-time block range because
// any branch that merges around the accumulators makes hipcc spill hundreds of VGPRs.  The result rows sit on n and
// the result columns (lane & 15) on the test row m, so the squared row sums stay lane-local.
//
// "Typed" rows/columns implement the complex GP as a real GP on [Re; Im] (see gple_kernels.h, SEParamSet).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "gple_kernels.h"

namespace gple
{
	typedef double d4 __attribute__((ext_vector_type(4)));
	typedef double d2 __attribute__((ext_vector_type(2)));

	namespace
	{
		constexpr int BM = 128, BN = 256;
		constexpr int BS = BN + 16; // LDS row stride of the T tile [k][n]
		constexpr int NTHREADS = 512;
		constexpr int GEN_KSPLIT_MIN = 8, GEN_KSPLIT_MAX = 64; // k-ranges per row in the generation kernel (partial means)
		// few test rows: split k further so that the generation still fills the chip (>= ~512 workgroups of 128 rows)
		int gen_ksplit(int m_rows)
		{
			int ks = GEN_KSPLIT_MIN;
			while (ks < GEN_KSPLIT_MAX && static_cast<long>(m_rows / 128) * ks < 512) ks *= 2;
			return ks;
		}

		// K*(row, k) = amp (exp(-((dx rl0)^2 + (dp rl1)^2)/2) + n2 [x* == x_k])  for the rows [row0, row0 + rows) of the
		// typed test set -> Ks (column-major, ld = rows); partial means mu_part[ky][row] = sum_{k in range ky} K* v[k].
		// One thread per row, blockIdx.y selects the k-range; the training point of each k is a scalar load.
		// DERIV (real GP only): also accumulates K* dv_ip (ip = 0..3) and (dK*/dl_d) v (d = 0, 1) per row for
		// PredictiveKernel::ErrorDerivatives (kernel.cpp:524-542); mu_part then holds 7 planes of gridDim.y x m_rows.
		// DERIV = 2 (complex GP in the [Re; Im] embedding): 15 planes [c w, c dw_0..7, dc_1..6 w] for
		// PredictiveComplexKernel::ErrorDerivatives (complex_kernel.cpp:648-668); dc_p comes from a.dspec[p - 1].
		// MODE 0: K* of the rows [row0, row0 + rows) and the partial means (the full predict).  Far-row pruning runs it twice:
		// MODE 1 over all rows without storing K* (partial means and the partial sums of K*^2 that decide which rows are live), then
		// MODE 2 over the compacted list of live rows (row0 = offset into the list; K* only).
		template <int DERIV, int MODE>
		__global__ void __launch_bounds__(128) kstar_gen_kernel(const PredictArgs a, int row0, int rows, double* __restrict__ Ks,
			double* __restrict__ mu_part, double* __restrict__ nrm_part, const int* __restrict__ live_list, const int* __restrict__ n_live)
		{
			const int r = blockIdx.x * 128 + threadIdx.x; // row inside the chunk
			int gm = row0 + r;                             // row of the typed test set
			int type_m = (row0 + blockIdx.x * 128) >= a.m_split; // MODE 0, 1: uniform per block (m_split multiple of 128)
			// MODE 3: K* of the rows [row0, row0 + rows) and nothing else (a generation beside the fit: its weights do not exist yet)
			if constexpr (MODE == 2)
			{
				const int nl = *n_live;
				if (row0 + static_cast<int>(blockIdx.x) * 128 >= nl) return; // uniform: nothing live in this block of the list
				gm = live_list[gm < nl ? gm : nl - 1];                        // the tail of the last block repeats the last live row (never read back)
				type_m = gm >= a.m_split;                                      // per lane: the list mixes the two row types of a complex GP
			}
			int pidx = type_m ? gm - a.m_split : gm;
			pidx = pidx < a.M ? pidx : a.M - 1; // rows beyond M are clamped (their results are never read)
			const double xm = a.Xs[2 * pidx], pm = a.Xs[2 * pidx + 1];
			const int ksplit = gridDim.y;
			const int kper = a.n_total / ksplit; // multiple of 4 (n_total is a multiple of 256, ksplit <= 64)
			const int kbeg = blockIdx.y * kper;
			double mu = 0.0, nrm = 0.0;
			constexpr int NACC = DERIV == 2 ? 14 : 6;
			double dacc[NACC];
#pragma unroll
			for (int ip = 0; ip < NACC; ++ip) dacc[ip] = 0.0;
			double* __restrict__ out = Ks + r + static_cast<long>(kbeg) * rows;
			for (int k0 = kbeg; k0 < kbeg + kper; k0 += 4)
			{
				const int type_k = k0 >= a.n_split;
				const SEParam &pa = a.ps.p[type_k], &pb = a.ps.p[1 + type_k]; // row type 0 / 1 against this column type
				const double amp = type_m ? pb.amp : pa.amp, rl0 = type_m ? pb.rl0 : pa.rl0, rl1 = type_m ? pb.rl1 : pa.rl1;
				const double n2 = (type_m == type_k) ? (type_m ? pb.n2 : pa.n2) : 0.0;
#pragma unroll
				for (int e = 0; e < 4; ++e)
				{
					const int k = k0 + e;
					const int pk = type_k ? k - a.n_split : k;
					const bool valid = pk < a.N;
					const int pc = valid ? pk : 0;
					const double xk = a.Xt[2 * pc], pkv = a.Xt[2 * pc + 1];
					// (every product and sum of a K* entry explicitly rounded or fused: left to the compiler, d0 * d0 + d1 * d1 contracts into an fma
					// in the instantiations where d0 * d0 has no second use and does not in the derivative ones — K* then differs by an ulp between
					// the passes of one predict)
					const double d0 = __dmul_rn(xm - xk, rl0), d1 = __dmul_rn(pm - pkv, rl1);
					const double g = exp_nonpos(__dmul_rn(-0.5, fma(d0, d0, __dmul_rn(d1, d1))));
					const double delta = (xm == xk && pm == pkv) ? n2 : 0.0; // delta_kernel: exact equality, kernel.cpp:26
					const double val = valid ? __dmul_rn(amp, g + delta) : 0.0;
					if constexpr (MODE != 2 && MODE != 3) mu = fma(val, a.v[k], mu);
					if constexpr (MODE == 1) nrm = fma(val, val, nrm);
					if constexpr (DERIV == 2)
					{
#pragma unroll
						for (int ip = 0; ip < 8; ++ip) dacc[ip] = fma(val, a.dv[static_cast<long>(ip) * a.n_total + k], dacc[ip]);
						// (explicitly rounded products and fused multiply-adds: the planes must not depend on which MODE instantiation the
						// compiler is contracting — the lean and the full predict are compared bit for bit)
						const double gw = valid ? __dmul_rn(g, a.v[k]) : 0.0;
						const double f0 = __dmul_rn(__dmul_rn(d0, d0), rl0), f1 = __dmul_rn(__dmul_rn(d1, d1), rl1);
#pragma unroll
						for (int ip = 0; ip < 6; ++ip)
						{
							const DSpec& sp = a.dspec[ip].b[type_m + type_k];
							const double fac = sp.active ? __dmul_rn(sp.amp, fma(sp.c1, sp.dim == 0 ? f0 : f1, sp.c0)) : 0.0;
							dacc[8 + ip] = fma(gw, fac, dacc[8 + ip]);
						}
					}
					else if constexpr (DERIV == 1)
					{
#pragma unroll
						for (int ip = 0; ip < 4; ++ip) dacc[ip] = fma(val, a.dv[static_cast<long>(ip) * a.n_total + k], dacc[ip]);
						// dK*/dl_d = K* ((x*_d - x_d)/l_d)^2 / l_d  (test-set branch: K* with its noise delta, kernel.cpp:196)
						const double vk = __dmul_rn(val, a.v[k]);
						dacc[4] = fma(vk, __dmul_rn(__dmul_rn(d0, d0), rl0), dacc[4]);
						dacc[5] = fma(vk, __dmul_rn(__dmul_rn(d1, d1), rl1), dacc[5]);
					}
					if constexpr (MODE != 1) out[static_cast<long>(e) * rows] = val;
				}
				out += 4L * rows;
			}
			if constexpr (MODE != 2 && MODE != 3) mu_part[static_cast<long>(blockIdx.y) * a.m_rows + gm] = mu;
			if constexpr (MODE == 1) nrm_part[static_cast<long>(blockIdx.y) * a.m_rows + gm] = nrm;
			if constexpr (DERIV != 0)
#pragma unroll
				for (int ip = 0; ip < NACC; ++ip) mu_part[(static_cast<long>(ip + 1) * ksplit + blockIdx.y) * a.m_rows + gm] = dacc[ip];
		}

		// the k-steps that cross the diagonal 256-block of an N-tile: step D starts at k = n0 + KB D, where the column blocks
		// j < KB D / 16 are identically zero
		template <int KB, class Step, int... D>
		__device__ __forceinline__ void diag_steps(Step& kstep, int nd, std::integer_sequence<int, D...>)
		{
			(kstep(std::integral_constant<int, (D * KB) / 16>{}, nd + D), ...);
		}

		// Rows whose K* is so small that the contraction cannot move the variance: q = k*^T K^-1 k* <= |k*|^2 / lambda_min(K) and
		// lambda_min(K) >= sf^2 sn^2 (the ridge), so |k*|^2 < thr = 2^-56 sf^2 sn^2 k(x*,x*) leaves q below a quarter of the
		// half-ulp of k(x*,x*): k(x*,x*) - q rounds to k(x*,x*) with or without it — grid points more than ~7 length scales away
		// from every training point: most of a phase-space grid.  Such rows get q = 0 and are not contracted; the others are
		// compacted into a list, K* is generated for the list only, and the row-norm kernel walks the compacted rows.  The output is
		// bit-identical to the full contraction (a row's q depends on nothing but its own K* row and T).
		//
		// Work queue.  The dispatcher hands workgroups to (XCD, CU) slots in a fixed round-robin (measured: 512 live blocks
		// alternating with 512 dead ones take as long as 1024 live ones; 384 live blocks take 12.7 ms when they are contiguous and
		// 21-26 ms when they are scattered, probes/live_pattern_probe.py), and the number of live rows is only known on the device:
		// a pruned launch is therefore one resident workgroup per CU pulling (row block, tile group) units from a counter until it
		// runs past the last live block — every workgroup reaches that exit.
		struct Prune
		{
			int* queue;        // device counter, zeroed before the launch
			const int* n_live; // device: number of live rows (the compacted list's length)
			int row0;          // offset of this chunk in the compacted list
			int nblocks, G;    // row blocks of the chunk (upper bound: the live ones are the first ceil((n_live - row0) / 128)), tile groups
			// A contraction in two launches (rownormp_kernel only; launch_predict_overlapped): the N-tiles [jt_lo, jt_hi) of every unit (jt_hi = 0: all of them).
			// A virtual group's sum runs over tiles of both launches, lane by lane, before anything is reduced — so the first launch SAVES the per-lane
			// partial sums of every group (state_mode 1: [row block][VG][wave][fragment][lane], nothing is written to q) and the second STARTS from them
			// (state_mode 2; a group has saved sums iff it owns a tile below jt_lo, i.e. iff vg < jt_lo): the bits of the unsplit launch.
			int jt_lo = 0, jt_hi = 0;
			double* state = nullptr;
			int state_mode = 0;
		};
		// the unit this workgroup works on next: false when there is none (queue mode: the counter ran past the last unit;
		// static mode: the one unit of the workgroup is done)
		template <bool QUEUE>
		__device__ __forceinline__ bool next_unit(const Prune& pr, int it, int& mblock, int& g, int& G)
		{
			if constexpr (!QUEUE)
			{
				mblock = blockIdx.x, g = blockIdx.y, G = gridDim.y;
				return it == 0;
			}
			__shared__ int s_unit, s_nb;
			if (threadIdx.x == 0)
			{
				s_unit = atomicAdd(pr.queue, 1);
				const int left = *pr.n_live - pr.row0;
				const int nb = left <= 0 ? 0 : (left + BM - 1) / BM;
				s_nb = nb < pr.nblocks ? nb : pr.nblocks;
			}
			__syncthreads();
			const int unit = __builtin_amdgcn_readfirstlane(s_unit), nb = __builtin_amdgcn_readfirstlane(s_nb); // uniform: SGPRs
			__syncthreads();
			G = pr.G;
			if (unit >= nb * pr.G) return false;
			mblock = unit % nb, g = unit / nb;
			return true;
		}
		// live rows -> list (any order: rows are independent) and row -> position in the list (-1: dead)
		__global__ void __launch_bounds__(256) compact_rows_kernel(const double* __restrict__ nrm_part, int planes, int m_rows, double thr,
			int* __restrict__ list, int* __restrict__ pos, int* __restrict__ n_live)
		{
			const int row = blockIdx.x * 256 + threadIdx.x;
			bool live = false;
			if (row < m_rows)
			{
				double sq = 0.0;
				for (int ky = 0; ky < planes; ++ky) sq += nrm_part[static_cast<long>(ky) * m_rows + row];
				live = !(sq < thr); // NaN counts as live
			}
			const unsigned long long mask = __ballot(live);
			const int lane = threadIdx.x & 63;
			int base = 0;
			if (lane == 0 && mask) base = atomicAdd(n_live, __popcll(mask));
			base = __shfl(base, 0);
			if (row < m_rows)
			{
				const int p = live ? base + __popcll(mask & ((1ULL << lane) - 1)) : -1;
				pos[row] = p;
				if (live) list[p] = row;
			}
		}
		// The same list for the cut-off-certain skip (PredictArgs::cut_thr): a POINT is live when |mu|^2 < thr (NaN counts as live); both typed
		// rows of a complex point go together.  mu: the summed means (m_rows); rows at or beyond M (padding) are dead.
		__global__ void __launch_bounds__(256) compact_by_mean_kernel(const double* __restrict__ mu, int m_rows, int m_split, int M, double thr,
			int* __restrict__ list, int* __restrict__ pos, int* __restrict__ n_live)
		{
			const int row = blockIdx.x * 256 + threadIdx.x;
			bool live = false;
			if (row < m_rows)
			{
				const int pt = row >= m_split ? row - m_split : row;
				if (pt < M)
				{
					const double re = mu[pt], im = m_split < m_rows ? mu[m_split + pt] : 0.0;
					live = !(re * re + im * im >= thr);
				}
			}
			const unsigned long long mask = __ballot(live);
			const int lane = threadIdx.x & 63;
			int base = 0;
			if (lane == 0 && mask) base = atomicAdd(n_live, __popcll(mask));
			base = __shfl(base, 0);
			if (row < m_rows)
			{
				const int p = live ? base + __popcll(mask & ((1ULL << lane) - 1)) : -1;
				pos[row] = p;
				if (live) list[p] = row;
			}
		}
		// A row's squared norm is the sum over the N-tiles of T; which workgroup adds which tiles depends on how many groups the launch
		// splits a row block into (rownorm_split: by the number of row blocks).  So that the VALUE does not depend on that choice,
		// the tiles are dealt out to VG fixed "virtual groups" in snake order (tile jt costs jt + 1 units), every virtual group
		// keeps its own partial sum q[vg][row], and the planes are added in the fixed order 0 .. VG - 1 afterwards.  A launch with
		// G groups gives group g the virtual groups snake_G(vg) == g.  Same bits for G = 1, 2, 4, 8 — a batch, its pieces, the
		// pruned and the full predict all agree.
		constexpr int VG = 8;
		__device__ __forceinline__ int snake(int v, int G)
		{
			const int p = v % (2 * G);
			return p < G ? p : 2 * G - 1 - p;
		}
		// q[row] = sum of the VG planes at the row's list position (fixed order), 0 for a dead row; statistics
		__global__ void __launch_bounds__(256) scatter_q_kernel(const double* __restrict__ qpart, long qstride, const int* __restrict__ pos, int m_rows,
			double* __restrict__ q, const int* __restrict__ n_live, unsigned long long* __restrict__ stats)
		{
			const int row = blockIdx.x * 256 + threadIdx.x;
			if (row == 0 && stats != nullptr)
			{
				atomicAdd(stats, static_cast<unsigned long long>((*n_live + BM - 1) / BM));
				atomicAdd(stats + 1, static_cast<unsigned long long>(m_rows / BM));
			}
			if (row >= m_rows) return;
			const int p = pos[row];
			double v = 0.0;
			if (p >= 0)
				for (int vg = 0; vg < VG; ++vg) v += qpart[static_cast<long>(vg) * qstride + p];
			q[row] = v;
		}

		// q[row] = sum_n ( sum_{k <= n} K*(row, k) T(n, k) )^2 for one chunk of rows.
		// WAVES waves x 16 rows per workgroup, K advances KB per barrier.  <8, 16> (the one launched): one workgroup fills a CU
		// (2 waves per SIMD).  GPLE_ROWNORM_VARIANT selects between this kernel and rownorm2_kernel for A/B runs (launch_predict_q).
		template <int WAVES, int KB, bool QUEUE>
		__global__ void __launch_bounds__(WAVES * 64, 8 / WAVES) rownorm_kernel(const double* __restrict__ Ks, int rows, const double* __restrict__ T,
			long ldt, int n_total, double* __restrict__ q, long qstride, const Prune pr)
		{
			constexpr int TM = WAVES * 16, NT = WAVES * 64;
			constexpr int ASr = TM + 16;
			constexpr int ASL = KB * ASr, BSL = KB * BS;
			constexpr int NA = TM * KB / 2 / NT, NBv = BN * KB / 2 / NT; // double2 per thread and slab
			static_assert(NA >= 1 && NBv >= 1, "tile too small for the thread count");
			__shared__ __attribute__((aligned(16))) double lds[2 * ASL + 2 * BSL];
			double* const As = lds;
			double* const Bs = lds + 2 * ASL;
			const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
			const int fk = lane >> 4, fr = lane & 15;
			int m0 = 0; // row block of the current unit (the loaders below read it)
			const int ntiles = n_total / BN;
			double rsq = 0.0;

			d2 areg[NA], breg[NBv];
			auto load_ab = [&](int n0, int k0) {
				const double* __restrict__ abase = Ks + m0 + static_cast<long>(k0) * rows;
#pragma unroll
				for (int qq = 0; qq < NA; ++qq)
				{
					const int i = t + NT * qq;
					const int r2 = (i % (TM / 2)) * 2, k = i / (TM / 2);
					areg[qq] = *reinterpret_cast<const d2*>(abase + r2 + static_cast<long>(k) * rows);
				}
				const double* __restrict__ bbase = T + n0 + static_cast<long>(k0) * ldt;
#pragma unroll
				for (int qq = 0; qq < NBv; ++qq)
				{
					const int i = t + NT * qq;
					const int r2 = (i & 127) * 2, k = i >> 7;
					breg[qq] = *reinterpret_cast<const d2*>(bbase + r2 + static_cast<long>(k) * ldt);
				}
			};
			auto store_ab = [&](int buf) {
				double* __restrict__ sa = As + buf * ASL;
#pragma unroll
				for (int qq = 0; qq < NA; ++qq)
				{
					const int i = t + NT * qq;
					const int r2 = (i % (TM / 2)) * 2, k = i / (TM / 2);
					*reinterpret_cast<d2*>(sa + k * ASr + r2) = areg[qq];
				}
				double* __restrict__ sb = Bs + buf * BSL;
#pragma unroll
				for (int qq = 0; qq < NBv; ++qq)
				{
					const int i = t + NT * qq;
					const int r2 = (i & 127) * 2, k = i >> 7;
					*reinterpret_cast<d2*>(sb + k * BS + r2) = breg[qq];
				}
			};

			// gridDim.y > 1 splits the N-tiles of one row block over several workgroups (few row blocks: fill the chip anyway).
			// N-tile jt costs jt + 1 units, so the tiles are dealt out in snake order: group g of G takes the tiles whose
			// position in a period of 2 G is g or 2 G - 1 - g.
			int mblock, g, G;
			for (int it = 0; next_unit<QUEUE>(pr, it, mblock, g, G); ++it)
			{
			m0 = mblock * TM;
			for (int vg = 0; vg < VG; ++vg)
			{
			if (snake(vg, G) != g) continue; // uniform
			rsq = 0.0;
			for (int jt = 0; jt < ntiles; ++jt)
			{
				if (snake(jt, VG) != vg) continue; // uniform; no accumulator is live here
				const int n0 = jt * BN;
				const int nk = (n0 + BN) / KB; // T(n,k) = 0 for k > n: k-slabs beyond the N-tile's last column are skipped
				d4 acc[16];
#pragma unroll
				for (int j = 0; j < 16; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};

				__syncthreads(); // the previous tile's last MFMAs have finished reading the stage
				load_ab(n0, 0);
				store_ab(0);
				__syncthreads();

				// one k-step: prefetch slab s + 1, MFMAs of slab s against the column blocks j >= JMIN, commit the prefetch
				auto kstep = [&](auto jmin_tag, int s) {
					constexpr int JMIN = decltype(jmin_tag)::value;
					if (s + 1 < nk) load_ab(n0, (s + 1) * KB);
					const double* __restrict__ pa = As + (s & 1) * ASL + w * 16 + fr;
					const double* __restrict__ pb = Bs + (s & 1) * BSL + fr;
#pragma unroll
					for (int kk = 0; kk < KB; kk += 4)
					{
						const double af = pa[(kk + fk) * ASr];
						double bf[16];
#pragma unroll
						for (int j = JMIN; j < 16; ++j) bf[j] = pb[(kk + fk) * BS + j * 16];
#pragma unroll
						for (int j = JMIN; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af, acc[j], 0, 0, 0);
					}
					if (s + 1 < nk) store_ab((s + 1) & 1);
					__syncthreads();
				};
				// k-slabs below the diagonal N-tile see every column block; inside the diagonal 256-block the column blocks left
				// of the current k (T(n,k) = 0 for k > n) drop out one at a time: 16 columns per 16 k
				const int nd = n0 / KB;
				int s = 0;
				for (; s < nd; ++s) kstep(std::integral_constant<int, 0>{}, s);
				diag_steps<KB>(kstep, nd, std::make_integer_sequence<int, BN / KB>{});

				// result element [n = 16 j + fk + 4 r][m = 16 w + fr]: the row index m is lane-local
#pragma unroll
				for (int j = 0; j < 16; ++j)
#pragma unroll
					for (int r = 0; r < 4; ++r) rsq = fma(acc[j][r], acc[j][r], rsq);
			}
			rsq += __shfl_xor(rsq, 16);
			rsq += __shfl_xor(rsq, 32);
			if (lane < 16) q[static_cast<long>(vg) * qstride + m0 + w * 16 + lane] = rsq; // partial sums of virtual group vg
			}
			}
		}

		// ---- register-blocked variant ------------------------------------------------------------------------------------
		// rownorm_kernel gives every wave 16 rows x all 256 columns: 1 + 16 operand fragments from LDS per 16 MFMAs.  The bare
		// MFMA loop with that operand pattern sustains 72.6 TFLOP/s against 78.4 from registers (probes/mfma_f64_sustained):
		// the LDS operand reads cost ~8 %.  Here a wave owns AF x BF fragments (AF * BF = 16 accumulators as before): AF + BF
		// reads per 16 MFMAs — 8 for 4 x 4, 10 for 2 x 8.  The 8 waves form a (128 / (16 AF)) x WN grid over the same 128 x 256
		// workgroup tile; a wave's BF column blocks are interleaved (block wn + WN t), so that inside the diagonal 256-block,
		// where the column blocks left of k drop out one per k-step, all waves keep the same number of live blocks (+-1).
		// The live range depends on the wave's column index, which must be a compile-time constant for the skipping to stay
		// straight-line code (branches around live accumulators make hipcc spill): the tile loop is instantiated per wn.
		template <int AF, int BF, int WNI>
		__device__ __forceinline__ void rownorm2_tiles(const double* __restrict__ Ks, int rows, const double* __restrict__ T, long ldt, int n_total,
			double* lds, double (&rsq)[AF], int m0, int wm, int vg)
		{
			constexpr int KB = 16, TM = BM, NT = NTHREADS, WN = 16 / BF;
			constexpr int ASr = TM + 16;
			constexpr int ASL = KB * ASr, BSL = KB * BS;
			constexpr int NA = TM * KB / 2 / NT, NBv = BN * KB / 2 / NT;
			double* const As = lds;
			double* const Bs = lds + 2 * ASL;
			const int t = threadIdx.x, lane = t & 63;
			const int fk = lane >> 4, fr = lane & 15;
			const int ntiles = n_total / BN;
			// Slabs go global memory -> LDS directly (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPR destination).  Staged
			// through registers (12 x 16 B per lane, written to LDS between the last MFMA and the barrier) the hand-over cost 1.1k of a k-step's
			// 9.9k cycles with every wave of the CU standing in it at once (probes/rownorm_stamps_probe.py); the DMA lands beside the MFMAs.
			// A wave instruction covers one k-row of the K* slab (128 doubles) or half a k-row of the T slab: contiguous in LDS, as the
			// instruction requires (destination = wave-uniform base + 16 lane).
			// Issued from inline asm: behind the builtin hipcc cannot tell the buffer the DMA fills from the buffer the operand reads of the
			// step come from (same __shared__ array) and puts s_waitcnt vmcnt(0) in front of the first ds_read of every k-step — the whole
			// DMA latency back on the critical path.  The asm form is invisible to its counters; nothing else in the k-loop touches vmcnt,
			// and every step ends in an explicit vmcnt(0) in front of the barrier (lds_barrier_dma), so none is ever outstanding when the
			// compiler's own memory operations run.  M0 = LDS byte address of the wave's 1 KB destination.
			const int w = __builtin_amdgcn_readfirstlane(t >> 6);
			auto lds_addr = [](const double* p) { return static_cast<unsigned>(reinterpret_cast<unsigned long>((__attribute__((address_space(3))) const double*)p)); };
			auto glds16 = [](const double* gsrc, unsigned lds_dst) {
				unsigned keep;
				asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
							 : "=&s"(keep)
							 : "v"(gsrc), "s"(lds_dst)
							 : "memory");
			};
			auto lds_barrier_dma = []() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
			const unsigned As_addr = lds_addr(As), Bs_addr = lds_addr(Bs);
			auto stage_ab = [&](int n0, int k0, int buf) {
				const double* __restrict__ abase = Ks + m0 + static_cast<long>(k0) * rows + 2 * lane;
#pragma unroll
				for (int qq = 0; qq < NA; ++qq)
				{
					const int k = w + (NT / 64) * qq; // i = t + NT qq, k = i / 64
					glds16(abase + static_cast<long>(k) * rows, As_addr + 8u * static_cast<unsigned>(buf * ASL + k * ASr));
				}
				const double* __restrict__ bbase = T + n0 + static_cast<long>(k0) * ldt + 128 * (w & 1) + 2 * lane;
#pragma unroll
				for (int qq = 0; qq < NBv; ++qq)
				{
					const int k = (w >> 1) + (NT / 128) * qq; // i = t + NT qq, k = i >> 7, columns (i & 127) * 2
					glds16(bbase + static_cast<long>(k) * ldt, Bs_addr + 8u * static_cast<unsigned>(buf * BSL + k * BS + 128 * (w & 1)));
				}
			};
			for (int jt = 0; jt < ntiles; ++jt)
			{
				if (snake(jt, VG) != vg) continue; // uniform; no accumulator is live here
				const int n0 = jt * BN;
				const int nk = (n0 + BN) / KB;
				d4 acc[AF][BF];
#pragma unroll
				for (int i = 0; i < AF; ++i)
#pragma unroll
					for (int j = 0; j < BF; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
				__syncthreads();
				stage_ab(n0, 0, 0);
				lds_barrier_dma();
				// one k-step against the column blocks wn + WN t, t >= TMIN
				auto kstep = [&](auto tmin_tag, int s) {
					constexpr int TMIN = decltype(tmin_tag)::value;
					const double* __restrict__ pa = As + (s & 1) * ASL + wm * (16 * AF) + fr;
					const double* __restrict__ pb = Bs + (s & 1) * BSL + WNI * 16 + fr;
					// the first group's operands are requested before the next slab's DMA goes out (that buffer was last read in step s - 1,
					// behind a barrier): their LDS latency runs under the six DMA issues instead of after them
					double af0[AF], bf0[BF];
#pragma unroll
					for (int i = 0; i < AF; ++i) af0[i] = pa[fk * ASr + i * 16];
#pragma unroll
					for (int j = TMIN; j < BF; ++j) bf0[j] = pb[fk * BS + j * (16 * WN)];
					if (s + 1 < nk) stage_ab(n0, (s + 1) * KB, (s + 1) & 1);
#pragma unroll
					for (int kk = 0; kk < KB; kk += 4)
					{
						double af[AF], bf[BF];
#pragma unroll
						for (int i = 0; i < AF; ++i) af[i] = kk == 0 ? af0[i] : pa[(kk + fk) * ASr + i * 16];
#pragma unroll
						for (int j = TMIN; j < BF; ++j) bf[j] = kk == 0 ? bf0[j] : pb[(kk + fk) * BS + j * (16 * WN)];
#pragma unroll
						for (int i = 0; i < AF; ++i)
#pragma unroll
							for (int j = TMIN; j < BF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
					}
					lds_barrier_dma(); // the DMA of this step was issued a whole step ago
				};
				const int nd = n0 / KB;
				for (int s = 0; s < nd; ++s) kstep(std::integral_constant<int, 0>{}, s);
				// diagonal 256-block: step D starts at k = n0 + 16 D, the column blocks jb < D are zero; jb = WNI + WN t >= D  <=>  t >= ceil((D - WNI) / WN)
				[&]<int... D>(std::integer_sequence<int, D...>) {
					(kstep(std::integral_constant<int, (D > WNI ? (D - WNI + WN - 1) / WN : 0)>{}, nd + D), ...);
				}(std::make_integer_sequence<int, BN / KB>{});
#pragma unroll
				for (int i = 0; i < AF; ++i)
#pragma unroll
					for (int j = 0; j < BF; ++j)
#pragma unroll
						for (int r = 0; r < 4; ++r) rsq[i] = fma(acc[i][j][r], acc[i][j][r], rsq[i]);
			}
		}
		template <int AF, int BF, bool QUEUE>
		__global__ void __launch_bounds__(NTHREADS, 1) rownorm2_kernel(const double* __restrict__ Ks, int rows, const double* __restrict__ T, long ldt,
			int n_total, double* __restrict__ q, long qstride, const Prune pr)
		{
			constexpr int KB = 16, WN = 16 / BF, ASr = BM + 16;
			__shared__ __attribute__((aligned(16))) double lds[2 * KB * ASr + 2 * KB * BS];
			const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int wm = w / WN, wn = w % WN;
			int mblock, g, G;
			for (int it = 0; next_unit<QUEUE>(pr, it, mblock, g, G); ++it)
			{
			const int m0 = mblock * BM;
			for (int vg = 0; vg < VG; ++vg)
			{
				if (snake(vg, G) != g) continue; // uniform
				double rsq[AF];
#pragma unroll
				for (int i = 0; i < AF; ++i) rsq[i] = 0.0;
				// the tile loop once per column index of the wave (uniform branch, taken once; no accumulator is live across it)
				if constexpr (WN == 4)
				{
					if (wn == 0) rownorm2_tiles<AF, BF, 0>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
					else if (wn == 1) rownorm2_tiles<AF, BF, 1>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
					else if (wn == 2) rownorm2_tiles<AF, BF, 2>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
					else rownorm2_tiles<AF, BF, 3>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
				}
				else
				{
					if (wn == 0) rownorm2_tiles<AF, BF, 0>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
					else rownorm2_tiles<AF, BF, 1>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
				}
				// rows of fragment i: m0 + wm * 16 AF + 16 i + (lane & 15); partial sums of the WN column groups meet in LDS
				__syncthreads();
#pragma unroll
				for (int i = 0; i < AF; ++i)
				{
					double v = rsq[i];
					v += __shfl_xor(v, 16);
					v += __shfl_xor(v, 32);
					if (lane < 16) lds[wn * BM + wm * (16 * AF) + 16 * i + lane] = v;
				}
				__syncthreads();
				if (threadIdx.x < BM)
				{
					double v = 0.0;
#pragma unroll
					for (int c = 0; c < WN; ++c) v += lds[c * BM + threadIdx.x];
					q[static_cast<long>(vg) * qstride + m0 + threadIdx.x] = v;
				}
				__syncthreads(); // the next virtual group's first tile stages operands over these sums
			}
			}
		}


		// ---- the register-blocked contraction as ONE pipeline over all k-steps of a unit ----------------------------------------
		// rownorm2_kernel's k-step, as hipcc schedules it: ... 16 MFMAs | barrier | 16 MFMAs (the last group, operands already in registers) |
		// first operand reads of the next step | ~80 scalar instructions of DMA addressing around six LDS-DMA issues | wait for the operands |
		// MFMAs.  Both waves of a SIMD come out of the barrier together and stay in step, so they reach the scalar stretch together and the
		// MFMA pipe idles through it and through the LDS latency behind it: ~900 of a step's 8600 cycles at C4r (MFMA pipe busy 89.8 %).
		// Here the order around the barrier is pinned (sched_barrier): two MFMAs of the last group straddle it (their operands were read before
		// it), then the next step's first operands are requested, then the six DMA issues go out between the group's other MFMAs — nothing that
		// follows the barrier waits for anything but the barrier.  The DMA addressing is two SGPR bases per slab and six constant VGPR offsets
		// (saddr form).  A slab is requested one step ahead of its use, as before (issued behind the barrier that ends step s - 1, needed
		// behind the barrier that ends step s) — but the sequence of slabs runs across N-tiles and virtual groups: the first slabs of the next
		// tile are requested during the last steps of the current one, so a tile no longer starts with an exposed DMA round trip (16 per row
		// block at n = 4096, 4 of 160 steps' worth at n = 1024); only a unit does.  The tiles of the unit are listed in LDS in the order
		// (virtual group, tile) they had before; per accumulator the same MFMAs in the same order, the same sums behind them: same bits as
		// rownorm2_kernel (test_rownorm_variants_agree_bit_for_bit).
		constexpr int TL_MAX = 256; // N-tiles of a factor (n <= 65536)
		template <int AF, int BF, int WNI>
		__device__ __forceinline__ void rownormp_unit(const double* __restrict__ Ks, int rows, const double* __restrict__ T, long ldt, double* lds,
			const int* tl, int ntl, double* __restrict__ q, long qstride, int m0, int wm, double* __restrict__ su, int mode, int jlo)
		{
			constexpr int KB = 16, WN = 16 / BF, ASr = BM + 16;
			constexpr int ASL = KB * ASr, BSL = KB * BS;
			static_assert(BM * KB / 2 / NTHREADS == 2 && BN * KB / 2 / NTHREADS == 4, "six DMA instructions per wave and slab");
			double* const As = lds;
			double* const Bs = lds + 2 * ASL;
			double* const red = Bs + 2 * BSL;
			const int t = threadIdx.x, lane = t & 63;
			const int fk = lane >> 4, fr = lane & 15;
			const int w = __builtin_amdgcn_readfirstlane(t >> 6);
			auto lds_addr = [](const double* p) { return static_cast<unsigned>(reinterpret_cast<unsigned long>((__attribute__((address_space(3))) const double*)p)); };
			// one LDS-DMA wave instruction: 1 KB from (base + voff) to LDS byte address ldst + poff.  M0 is written, never restored: nothing else in
			// this kernel reads it (hipcc treats M0 as reserved and sets it in front of each of its own uses; checked on the disassembly)
			auto dma = [](const double* base, unsigned voff, unsigned ldst, unsigned poff) {
				asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3" ::"v"(voff), "s"(ldst), "s"(poff), "s"(base) : "memory", "scc");
			};
			// wave w fills k-rows w and w + 8 of the K* slab and, of the T slab, half (w & 1) of the k-rows (w >> 1) + 4 qq
			unsigned voff[6], ldst[6];
			voff[0] = 16u * lane, voff[1] = voff[0] + 64u * static_cast<unsigned>(rows);
			ldst[0] = lds_addr(As) + 8u * static_cast<unsigned>(w * ASr), ldst[1] = ldst[0] + 8u * (8 * ASr);
#pragma unroll
			for (int qq = 0; qq < 4; ++qq)
			{
				voff[2 + qq] = voff[0] + 32u * static_cast<unsigned>(ldt) * qq;
				ldst[2 + qq] = lds_addr(Bs) + 8u * static_cast<unsigned>(((w >> 1) + 4 * qq) * BS + 128 * (w & 1));
			}
			const double* const KsW = Ks + m0 + static_cast<long>(w) * rows;
			const double* const TW = T + 128 * (w & 1) + static_cast<long>(w >> 1) * ldt;
			auto entry = [&](int i) { return __builtin_amdgcn_readfirstlane(tl[i]); };
			auto tile_n0 = [&](int i) { return __builtin_amdgcn_readfirstlane(tl[i]) & 0xffff; };
			auto tile_vg = [&](int i) { return __builtin_amdgcn_readfirstlane(tl[i]) >> 16; };

			// the slab the next DMA fetches: tile ta (first column n0a), k-rows [ka, ka + 16)
			int ta = 0, n0a = tile_n0(0) * BN, ka = 0;
			auto advance = [&](int next_entry) { // next_entry = tl[ta + 1] (-1 behind the last tile)
				ka += KB;
				const bool wrap = ka == n0a + BN, has = next_entry >= 0;
				// behind the last slab of the unit the iterator stays where it is: the two requests past the end re-fetch the last slab (never read)
				ta = wrap && has ? ta + 1 : ta;
				ka = wrap ? (has ? 0 : ka - KB) : ka;
				n0a = wrap && has ? (next_entry & 0xffff) * BN : n0a;
			};
			auto stage_all = [&](int buf) {
				const double* pA = KsW + static_cast<long>(ka) * rows;
				const double* pB = TW + n0a + static_cast<long>(ka) * ldt;
#pragma unroll
				for (int i = 0; i < 6; ++i) dma(i < 2 ? pA : pB, voff[i], ldst[i], 8u * static_cast<unsigned>(buf * (i < 2 ? ASL : BSL)));
			};
			__syncthreads(); // the stage is free (previous unit)
			stage_all(0);
			advance(entry(1));
			stage_all(1);
			advance(entry(ta + 1));
			asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // slab 0 has landed (the six requests of slab 1 may still be out)
			int p = 0; // buffer of the current step
			double cA[AF], cB[BF]; // operands of the next MFMA group
			auto read_ops = [&](int buf, int kk, double (&af)[AF], double (&bf)[BF]) {
				const double* __restrict__ pa = As + buf * ASL + wm * (16 * AF) + fr;
				const double* __restrict__ pb = Bs + buf * BSL + WNI * 16 + fr;
#pragma unroll
				for (int i = 0; i < AF; ++i) af[i] = pa[(kk + fk) * ASr + i * 16];
#pragma unroll
				for (int j = 0; j < BF; ++j) bf[j] = pb[(kk + fk) * BS + j * (16 * WN)];
			};
			read_ops(0, 0, cA, cB);
			// su: this unit's saved per-lane sums (Prune::state), entry (vg, wave, fragment, lane)
			auto su_at = [&](int vg, int i) { return su + ((static_cast<long>(vg) * 8 + w) * AF + i) * 64 + lane; };
			double rsq[AF];
			{
				const int vg0 = tile_vg(0);
#pragma unroll
				for (int i = 0; i < AF; ++i) rsq[i] = (mode == 2 && vg0 < jlo) ? *su_at(vg0, i) : 0.0;
			}
			for (int ti = 0; ti < ntl; ++ti)
			{
				const int n0 = tile_n0(ti) * BN, vg = tile_vg(ti);
				d4 acc[AF][BF];
#pragma unroll
				for (int i = 0; i < AF; ++i)
#pragma unroll
					for (int j = 0; j < BF; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
				// one k-step against the column blocks wn + WN t, t >= TMIN
				auto kstep = [&](auto tmin_tag) {
					constexpr int TMIN = decltype(tmin_tag)::value;
					const int next_entry_v = tl[ta + 1]; // consumed in the middle of the step
					// the slab this step's DMA fetches and where it goes (the buffer this step reads): scalar work, done while the MFMAs behind the last
					// barrier still run — between a wave's last MFMA of the step and the barrier there is nothing but the barrier
					const double* pA = KsW + static_cast<long>(ka) * rows;
					const double* pB = TW + n0a + static_cast<long>(ka) * ldt;
					unsigned poA = 8u * static_cast<unsigned>(p * ASL), poB = 8u * static_cast<unsigned>(p * BSL);
					asm volatile("" : "+s"(pA), "+s"(pB), "+s"(poA), "+s"(poB)); // computed HERE (hipcc sinks them to their use behind the barrier otherwise)
					double nA[AF], nB[BF];
#pragma unroll
					for (int kk = 4; kk < KB; kk += 4)
					{
						// Inside the three groups in front of the barrier hipcc schedules (it requests a group's operands two MFMAs before the end of the
						// group before: 0.923 of the peak at C4r against 0.911 with the requests pinned to the start of the group, 0.918 in its middle —
						// same box); what is pinned is the barrier's neighbourhood: unpinned, hipcc gathers the reads of ALL groups in front of the barrier
						// and sinks three groups of MFMAs behind it
						read_ops(p, kk, nA, nB);
						if (kk == 8) advance(__builtin_amdgcn_readfirstlane(next_entry_v));
#pragma unroll
						for (int i = 0; i < AF; ++i)
#pragma unroll
							for (int j = TMIN; j < BF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(cB[j], cA[i], acc[i][j], 0, 0, 0);
						if (kk == 12) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int i = 0; i < AF; ++i) cA[i] = nA[i];
#pragma unroll
						for (int j = 0; j < BF; ++j) cB[j] = nB[j];
					}
					asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // the next slab has landed; every wave holds its last operands of this one
					p ^= 1;
					// behind the barrier: LEAD MFMAs of the step's last group at once (their operands were read in front of the barrier), then the requests
					// for the next step's first operands, then the six DMA issues of the slab after the next (into the buffer this step read) between
					// the group's other MFMAs.  LEAD: measured on one box, 4 x 4: 2 leads 0 by 0.6 % at C4r; 2 x 8 (ten operand requests): 0 leads 2 by 0.5 % at C2
					constexpr int NM = AF * (BF - TMIN), LEAD = AF == 4 ? 2 : 0;
					int issued = 0;
#pragma unroll
					for (int i = 0; i < AF; ++i)
#pragma unroll
						for (int j = TMIN; j < BF; ++j)
						{
							const int m = i * (BF - TMIN) + (j - TMIN);
							if (m == LEAD)
							{
								read_ops(p, 0, nA, nB);
								__builtin_amdgcn_sched_barrier(0);
							}
							if (m >= LEAD && (m - LEAD) % 2 == 0 && issued < 6)
							{
								dma(issued < 2 ? pA : pB, voff[issued], ldst[issued], issued < 2 ? poA : poB);
								++issued;
							}
							acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(cB[j], cA[i], acc[i][j], 0, 0, 0);
							if (m % 2 == 1) __builtin_amdgcn_sched_barrier(0);
						}
					if constexpr (NM <= LEAD)
					{
						__builtin_amdgcn_sched_barrier(0);
						read_ops(p, 0, nA, nB);
						__builtin_amdgcn_sched_barrier(0);
					}
#pragma unroll
					for (int r = NM > LEAD ? ((NM - LEAD + 1) / 2 < 6 ? (NM - LEAD + 1) / 2 : 6) : 0; r < 6; ++r) dma(r < 2 ? pA : pB, voff[r], ldst[r], r < 2 ? poA : poB);
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int i = 0; i < AF; ++i) cA[i] = nA[i];
#pragma unroll
					for (int j = 0; j < BF; ++j) cB[j] = nB[j];
				};
				const int nd = n0 / KB;
				for (int s = 0; s < nd; ++s) kstep(std::integral_constant<int, 0>{});
				// diagonal 256-block: step D starts at k = n0 + 16 D, the column blocks jb < D are zero; jb = WNI + WN t >= D  <=>  t >= ceil((D - WNI) / WN)
				[&]<int... D>(std::integer_sequence<int, D...>) {
					(kstep(std::integral_constant<int, (D > WNI ? (D - WNI + WN - 1) / WN : 0)>{}), ...);
				}(std::make_integer_sequence<int, BN / KB>{});
#pragma unroll
				for (int i = 0; i < AF; ++i)
#pragma unroll
					for (int j = 0; j < BF; ++j)
#pragma unroll
						for (int r = 0; r < 4; ++r) rsq[i] = fma(acc[i][j][r], acc[i][j][r], rsq[i]);
				// the last tile of a virtual group: its plane of partial sums (the WN column groups meet in LDS, beside the stage)
				if (ti + 1 == ntl || tile_vg(ti + 1) != vg)
				{
					const int vgn = ti + 1 < ntl ? tile_vg(ti + 1) : VG; // the next group of this unit
					if (mode == 1) // the early tiles: the group's sums wait, lane by lane, for its late tiles
					{
#pragma unroll
						for (int i = 0; i < AF; ++i) *su_at(vg, i) = rsq[i], rsq[i] = 0.0;
					}
					else
					{
#pragma unroll
						for (int i = 0; i < AF; ++i)
						{
							double v = rsq[i];
							v += __shfl_xor(v, 16);
							v += __shfl_xor(v, 32);
							if (lane < 16) red[WNI * BM + wm * (16 * AF) + 16 * i + lane] = v;
							rsq[i] = (mode == 2 && vgn < jlo) ? *su_at(vgn, i) : 0.0;
						}
						__syncthreads();
						if (threadIdx.x < BM)
						{
							double v = 0.0;
#pragma unroll
							for (int c = 0; c < WN; ++c) v += red[c * BM + threadIdx.x];
							q[static_cast<long>(vg) * qstride + m0 + threadIdx.x] = v;
						}
						__syncthreads();
					}
				}
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the two requests past the end
		}
		template <int AF, int BF, bool QUEUE>
		__global__ void __launch_bounds__(NTHREADS, 1) rownormp_kernel(const double* __restrict__ Ks, int rows, const double* __restrict__ T, long ldt,
			int n_total, double* __restrict__ q, long qstride, const Prune pr)
		{
			constexpr int KB = 16, WN = 16 / BF, ASr = BM + 16;
			__shared__ __attribute__((aligned(16))) double lds[2 * KB * ASr + 2 * KB * BS + WN * BM];
			__shared__ int tl[TL_MAX + 2];
			__shared__ int s_ntl;
			const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			// wave -> (row group wm, column group wn).  Waves w and w + 4 share a SIMD, and inside the diagonal 256-block a column group's live blocks
			// thin out at its own pace (group wn keeps the blocks wn + WN t >= D): with wn = w % WN both waves of a SIMD belong to the same column
			// group and the SIMD of the last group works up to a block per step longer than the first one's while all meet at the barrier.  Here
			// a SIMD's two waves take column groups that thin out in opposite phase (wn and WN - 1 - wn); which wave computes what changes nothing in the sums
			const int simd = w & 3, half = w >> 2;
			const int wm = WN == 4 ? half : simd, wn = WN == 4 ? (half ? 3 - simd : simd) : (half ^ (simd & 1));
			const int ntiles = n_total / BN;
			const int jlo = pr.jt_lo, jhi = pr.jt_hi > 0 ? pr.jt_hi : ntiles; // the N-tiles of this launch
			int mblock, g, G;
			for (int it = 0; next_unit<QUEUE>(pr, it, mblock, g, G); ++it)
			{
				const int m0 = mblock * BM;
				__syncthreads(); // the previous unit's list has been read to the end
				if (threadIdx.x == 0)
				{
					// the unit's tiles in the order (virtual group, tile): entry = tile | virtual group << 16, -1 behind the last
					int c = 0;
					for (int vg = 0; vg < VG; ++vg)
					{
						if (snake(vg, G) != g) continue;
						for (int jt = jlo; jt < jhi; ++jt)
							if (snake(jt, VG) == vg) tl[c++] = jt | (vg << 16);
					}
					tl[c] = -1, tl[c + 1] = -1;
					s_ntl = c;
				}
				// a virtual group without a tile in this launch still owns its plane of the partial sums: zero (n < 2048: it has no tile at all), or —
				// the late launch of a split contraction — its saved sums, reduced here.  (The early launch writes no plane.)
				double* const su = pr.state ? pr.state + static_cast<long>(mblock) * (VG * 8 * AF * 64) : nullptr;
				for (int vg = 0; vg < VG; ++vg)
				{
					if (snake(vg, G) != g) continue; // uniform
					bool has = false;
					for (int jt = jlo; jt < jhi; ++jt) has = has || snake(jt, VG) == vg;
					if (has || pr.state_mode == 1) continue;
					if (pr.state_mode == 2 && vg < jlo)
					{
						double* const red = lds + 2 * KB * ASr + 2 * KB * BS;
						const int lane = threadIdx.x & 63;
#pragma unroll
						for (int i = 0; i < AF; ++i)
						{
							double v = su[((static_cast<long>(vg) * 8 + w) * AF + i) * 64 + lane];
							v += __shfl_xor(v, 16);
							v += __shfl_xor(v, 32);
							if (lane < 16) red[wn * BM + wm * (16 * AF) + 16 * i + lane] = v;
						}
						__syncthreads();
						if (threadIdx.x < BM)
						{
							double v = 0.0;
#pragma unroll
							for (int c = 0; c < WN; ++c) v += red[c * BM + threadIdx.x];
							q[static_cast<long>(vg) * qstride + m0 + threadIdx.x] = v;
						}
						__syncthreads();
					}
					else if (threadIdx.x < BM) q[static_cast<long>(vg) * qstride + m0 + threadIdx.x] = 0.0;
				}
				__syncthreads();
				const int ntl = __builtin_amdgcn_readfirstlane(s_ntl);
				if (ntl == 0) continue; // uniform
				if constexpr (WN == 4)
				{
					if (wn == 0) rownormp_unit<AF, BF, 0>(Ks, rows, T, ldt, lds, tl, ntl, q, qstride, m0, wm, su, pr.state_mode, jlo);
					else if (wn == 1) rownormp_unit<AF, BF, 1>(Ks, rows, T, ldt, lds, tl, ntl, q, qstride, m0, wm, su, pr.state_mode, jlo);
					else if (wn == 2) rownormp_unit<AF, BF, 2>(Ks, rows, T, ldt, lds, tl, ntl, q, qstride, m0, wm, su, pr.state_mode, jlo);
					else rownormp_unit<AF, BF, 3>(Ks, rows, T, ldt, lds, tl, ntl, q, qstride, m0, wm, su, pr.state_mode, jlo);
				}
				else
				{
					if (wn == 0) rownormp_unit<AF, BF, 0>(Ks, rows, T, ldt, lds, tl, ntl, q, qstride, m0, wm, su, pr.state_mode, jlo);
					else rownormp_unit<AF, BF, 1>(Ks, rows, T, ldt, lds, tl, ntl, q, qstride, m0, wm, su, pr.state_mode, jlo);
				}
			}
		}

		// ---- the same contraction for SHORT factors (n <= 512: one or two N-tiles) and few row blocks — C1: N = 256, 128 x 128 grid ----------
		// rownorm2_kernel<2,8> gives such a launch 128 workgroups of 16 k-steps for 256 CUs, and every step waits for its slab's DMA round trip
		// (~2.5 us against ~1 us of MFMA work in the diagonal tile): 47 us for 1.1 GFLOP, 0.29 of the peak.  Here a workgroup takes 64 rows (twice
		// the workgroups) and keeps THREE slabs in flight (a slab is requested two steps ahead; 64-row slabs make three stages fit: 3 x 43 KB),
		// so a step waits for a DMA issued two steps ago.  The per-row arithmetic is that of <2,8> — one 16-row fragment per wave against the same
		// 8 interleaved column blocks, the same k order, the same order of the partial sums — so a row's q has the same bits whichever of the
		// two kernels a launch gets (test_short_factor_kernel_has_the_bits_of_the_general_one).
		// A slab in LDS: 1 KB chunks (what one wave instruction of the LDS-DMA fills: lanes 0-31 one k-row of 64 doubles, lanes 32-63 another),
		// chunk stride 144 doubles; k-row k sits in chunk 2 (k / 4) + (k & 1), half (k >> 1) & 1: the four k-rows of a fragment read then fall on
		// disjoint bank halves in each of the instruction's two passes, as with the 128-row layout.
		template <int WNI>
		__device__ __forceinline__ void rownorm3_tiles(const double* __restrict__ Ks, int rows, const double* __restrict__ T, long ldt, int n_total, double* lds,
			double& rsq, int m0, int wm, int vg)
		{
			constexpr int KB = 16, NT = NTHREADS, BF = 8, WN = 2, NST = 3; // 64 rows per workgroup
			constexpr int ACH = 144;              // chunk stride of the A slab (doubles)
			constexpr int ASL = (KB / 2) * ACH;   // 8 chunks per slab
			constexpr int BSL = KB * BS;
			constexpr int NBv = BN * KB / 2 / NT; // B: 4 wave instructions per wave and slab; A: one
			double* const As = lds;
			double* const Bs = lds + NST * ASL;
			const int t = threadIdx.x, lane = t & 63;
			const int fk = lane >> 4, fr = lane & 15;
			const int ntiles = n_total / BN;
			const int w = __builtin_amdgcn_readfirstlane(t >> 6);
			auto lds_addr = [](const double* p) { return static_cast<unsigned>(reinterpret_cast<unsigned long>((__attribute__((address_space(3))) const double*)p)); };
			auto glds16 = [](const double* gsrc, unsigned lds_dst) {
				unsigned keep;
				asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
							 : "=&s"(keep)
							 : "v"(gsrc), "s"(lds_dst)
							 : "memory");
			};
			const unsigned As_addr = lds_addr(As), Bs_addr = lds_addr(Bs);
			// A: wave w fills chunk w of the slab: k-rows ka = 4 (w / 2) + (w & 1) (lanes 0-31) and ka + 2 (lanes 32-63)
			const int ka = 4 * (w >> 1) + (w & 1) + 2 * (lane >> 5);
			auto stage_ab = [&](int n0, int k0, int buf) {
				glds16(Ks + m0 + static_cast<long>(k0 + ka) * rows + 2 * (lane & 31), As_addr + 8u * static_cast<unsigned>(buf * ASL + w * ACH));
				const double* __restrict__ bbase = T + n0 + static_cast<long>(k0) * ldt + 128 * (w & 1) + 2 * lane;
#pragma unroll
				for (int qq = 0; qq < NBv; ++qq)
				{
					const int k = (w >> 1) + (NT / 128) * qq;
					glds16(bbase + static_cast<long>(k) * ldt, Bs_addr + 8u * static_cast<unsigned>(buf * BSL + k * BS + 128 * (w & 1)));
				}
			};
			// k-row (kk + fk) of the A slab, row fragment of this wave
			auto a_at = [&](const double* slab, int k) { return slab[(2 * (k >> 2) + (k & 1)) * ACH + ((k >> 1) & 1) * 64 + wm * 16 + fr]; };
			for (int jt = 0; jt < ntiles; ++jt)
			{
				if (snake(jt, VG) != vg) continue; // uniform; no accumulator is live here
				const int n0 = jt * BN;
				const int nk = (n0 + BN) / KB; // >= 16
				d4 acc[BF];
#pragma unroll
				for (int j = 0; j < BF; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
				__syncthreads();
				stage_ab(n0, 0, 0);
				stage_ab(n0, KB, 1);
				asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // slab 0 has landed (the 5 requests of slab 1 may still be out)
				auto kstep = [&](auto tmin_tag, int s) {
					constexpr int TMIN = decltype(tmin_tag)::value;
					const double* __restrict__ pa = As + (s % NST) * ASL;
					const double* __restrict__ pb = Bs + (s % NST) * BSL + WNI * 16 + fr;
					// slab s + 2 goes into the buffer step s - 1 read (every wave is past the barrier that ended it)
					if (s + 2 < nk) stage_ab(n0, (s + 2) * KB, (s + 2) % NST);
#pragma unroll
					for (int kk = 0; kk < KB; kk += 4)
					{
						const double af = a_at(pa, kk + fk);
						double bf[BF];
#pragma unroll
						for (int j = TMIN; j < BF; ++j) bf[j] = pb[(kk + fk) * BS + j * (16 * WN)];
#pragma unroll
						for (int j = TMIN; j < BF; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af, acc[j], 0, 0, 0);
					}
					// the next step reads slab s + 1: everything but the requests of slab s + 2 (if any were made) must have landed
					if (s + 2 < nk) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
					else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
				};
				const int nd = n0 / KB;
				for (int s = 0; s < nd; ++s) kstep(std::integral_constant<int, 0>{}, s);
				[&]<int... D>(std::integer_sequence<int, D...>) {
					(kstep(std::integral_constant<int, (D > WNI ? (D - WNI + WN - 1) / WN : 0)>{}, nd + D), ...);
				}(std::make_integer_sequence<int, BN / KB>{});
#pragma unroll
				for (int j = 0; j < BF; ++j)
#pragma unroll
					for (int r = 0; r < 4; ++r) rsq = fma(acc[j][r], acc[j][r], rsq);
			}
		}
		__global__ void __launch_bounds__(NTHREADS, 1) rownorm3_kernel(const double* __restrict__ Ks, int rows, const double* __restrict__ T, long ldt, int n_total,
			double* __restrict__ q, long qstride)
		{
			constexpr int KB = 16, TM = 64, WN = 2, NST = 3;
			__shared__ __attribute__((aligned(16))) double lds[NST * (KB / 2) * 144 + NST * KB * BS];
			const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int wm = w / WN, wn = w % WN;
			const int m0 = blockIdx.x * TM, g = blockIdx.y, G = gridDim.y;
			for (int vg = 0; vg < VG; ++vg)
			{
				if (snake(vg, G) != g) continue; // uniform
				double rsq = 0.0;
				if (wn == 0) rownorm3_tiles<0>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
				else rownorm3_tiles<1>(Ks, rows, T, ldt, n_total, lds, rsq, m0, wm, vg);
				__syncthreads();
				double v = rsq;
				v += __shfl_xor(v, 16);
				v += __shfl_xor(v, 32);
				if (lane < 16) lds[wn * TM + wm * 16 + lane] = v;
				__syncthreads();
				if (threadIdx.x < TM)
				{
					double sum = 0.0;
#pragma unroll
					for (int c = 0; c < WN; ++c) sum += lds[c * TM + threadIdx.x];
					q[static_cast<long>(vg) * qstride + m0 + threadIdx.x] = sum;
				}
				__syncthreads();
			}
		}

		// ---- ONE launch for the real GP with n = 256 (N <= 256: the size the reference itself runs, C1) ----------------------------------------
		// kstar_gen_kernel + rownorm3_kernel + two sum_mu_kernel launches are 15 + 35 + 5 + 5 us at C1 (128 x 128 grid) with the gaps between them,
		// for 14 us of MFMA work.  With ONE N-tile nothing of K* is ever used twice, so the argument against generating it inside the contraction
		// (a fused kernel regenerates every entry once per N-tile) does not apply: here a workgroup of 64 rows generates the K* slab
		// of the NEXT k-step in registers (two entries per thread: rows t & 63, k-rows w and w + 8 of the slab) while the MFMAs of the current one
		// run, writes it to LDS in front of the step's barrier, and streams the slabs of T by LDS-DMA two steps ahead (three stages).  The fp64
		// exponentials and the MFMAs share the SIMD's pipe, so their times add (7.5 + 14.5 us) — but nothing goes through HBM in between and
		// nothing waits for a launch.  The mean rides along: wave 0 walks its 64 rows' k-chain over the slab in LDS.
		// Same bits as the kernels it replaces, by construction: K* entries by the expression of kstar_gen_kernel, the partial means as chains
		// over the same k-ranges (gen_ksplit) added in the same order (sum_mu_kernel), the contraction with rownorm3_kernel's fragments, k order
		// and sums (test_fused_small_predict_has_the_bits_of_the_unfused_path).
		template <int WNI>
		__device__ __forceinline__ void fused256_wave(const PredictArgs& a, int ksplit, double* lds, int m0, int wm)
		{
			constexpr int KB = 16, BF = 8, WN = 2, NST = 3, TM = 64;
			constexpr int ASr = TM + 16, ASL = KB * ASr, BSL = KB * BS;
			double* const As = lds;
			double* const Bs = lds + 2 * ASL;
			double* const red = Bs + NST * BSL;
			double* const vs = red + WN * TM;
			const int t = threadIdx.x, lane = t & 63;
			const int fk = lane >> 4, fr = lane & 15;
			const int w = __builtin_amdgcn_readfirstlane(t >> 6);
			auto lds_addr = [](const double* p) { return static_cast<unsigned>(reinterpret_cast<unsigned long>((__attribute__((address_space(3))) const double*)p)); };
			auto dma = [](const double* base, unsigned voff, unsigned ldst) {
				asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(ldst), "s"(base) : "memory");
			};
			// T slab s (k-rows 16 s ..): wave w fills half (w & 1) of the k-rows (w >> 1) + 4 qq
			const double* const TW = a.T + 128 * (w & 1) + static_cast<long>(w >> 1) * a.ldt;
			auto stage_b = [&](int s) {
#pragma unroll
				for (int qq = 0; qq < 4; ++qq)
					dma(TW + static_cast<long>(s) * KB * a.ldt, 16u * lane + 32u * static_cast<unsigned>(a.ldt) * qq,
						lds_addr(Bs) + 8u * static_cast<unsigned>((s % NST) * BSL + ((w >> 1) + 4 * qq) * BS + 128 * (w & 1)));
			};
			// this thread's test point (rows beyond M are clamped: their results are never read) and its two K* entries of slab s
			const int gm = m0 + lane;
			const int pidx = gm < a.M ? gm : a.M - 1;
			const double xm = a.Xs[2 * pidx], pm = a.Xs[2 * pidx + 1];
			const SEParam& sp = a.ps.p[0];
			auto kstar = [&](int k) {
				const bool valid = k < a.N;
				const int pc = valid ? k : 0;
				const double xk = a.Xt[2 * pc], pkv = a.Xt[2 * pc + 1]; // uniform: scalar loads
				const double d0 = __dmul_rn(xm - xk, sp.rl0), d1 = __dmul_rn(pm - pkv, sp.rl1);
				const double g = exp_nonpos(__dmul_rn(-0.5, fma(d0, d0, __dmul_rn(d1, d1))));
				const double delta = (xm == xk && pm == pkv) ? sp.n2 : 0.0; // delta_kernel: exact equality, kernel.cpp:26
				return valid ? __dmul_rn(sp.amp, g + delta) : 0.0;
			};
			if (t < BN) vs[t] = a.v[t];
			stage_b(0);
			stage_b(1);
			As[w * ASr + lane] = kstar(w), As[(w + 8) * ASr + lane] = kstar(w + 8);
			asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); // slab 0 of T has landed, slab 0 of K* is written
			d4 acc[BF];
#pragma unroll
			for (int j = 0; j < BF; ++j) acc[j] = (d4){0.0, 0.0, 0.0, 0.0};
			const int kper = BN / ksplit; // k-range of one partial mean (kstar_gen_kernel): 32 ... 4
			double mu = 0.0, msum = 0.0;
			auto kstep = [&](auto d_tag) {
				constexpr int D = decltype(d_tag)::value;
				constexpr int TMIN = D > WNI ? (D - WNI + WN - 1) / WN : 0;
				if constexpr (D + 2 < BN / KB) stage_b(D + 2); // into the stage step D - 1 read
				double e0 = 0.0, e1 = 0.0;
				if constexpr (D + 1 < BN / KB) e0 = kstar(KB * (D + 1) + w), e1 = kstar(KB * (D + 1) + w + 8);
				const double* __restrict__ pa = As + (D & 1) * ASL;
				const double* __restrict__ pb = Bs + (D % NST) * BSL + WNI * 16 + fr;
				if (w == 0)
				{
					// the mean of rows m0 .. m0 + 63: one chain per k-range of kper entries, the ranges added in order (sum_mu_kernel)
#pragma unroll
					for (int k = 0; k < KB; ++k)
					{
						const bool first = ((KB * D + k) & (kper - 1)) == 0;
						msum = first ? msum + mu : msum;
						mu = first ? 0.0 : mu;
						mu = fma(pa[k * ASr + lane], vs[KB * D + k], mu);
					}
				}
#pragma unroll
				for (int kk = 0; kk < KB; kk += 4)
				{
					const double af = pa[(kk + fk) * ASr + wm * 16 + fr];
					double bf[BF];
#pragma unroll
					for (int j = TMIN; j < BF; ++j) bf[j] = pb[(kk + fk) * BS + j * (16 * WN)];
#pragma unroll
					for (int j = TMIN; j < BF; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af, acc[j], 0, 0, 0);
				}
				if constexpr (D + 1 < BN / KB)
				{
					double* __restrict__ pn = As + ((D + 1) & 1) * ASL; // read in step D - 1: every wave is past that step's barrier
					pn[w * ASr + lane] = e0, pn[(w + 8) * ASr + lane] = e1;
				}
				if constexpr (D + 2 < BN / KB) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
				else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
			};
			[&]<int... D>(std::integer_sequence<int, D...>) { (kstep(std::integral_constant<int, D>{}), ...); }(std::make_integer_sequence<int, BN / KB>{});
			double rsq = 0.0;
#pragma unroll
			for (int j = 0; j < BF; ++j)
#pragma unroll
				for (int r = 0; r < 4; ++r) rsq = fma(acc[j][r], acc[j][r], rsq);
			double v = rsq;
			v += __shfl_xor(v, 16);
			v += __shfl_xor(v, 32);
			if (lane < 16) red[WNI * TM + wm * 16 + lane] = v;
			__syncthreads();
			if (t < TM)
			{
				double sum = 0.0;
#pragma unroll
				for (int c = 0; c < WN; ++c) sum += red[c * TM + t];
				double qv = 0.0; // the VG planes in order: plane 0 holds the one N-tile, the others are zero
				qv += sum;
				const double m = msum + mu; // (w == 0 for these threads) the last k-range
				if (a.fin_sdev == nullptr) a.q[m0 + t] = qv, a.mu[m0 + t] = m;
				else if (m0 + t < a.M)
				{
					// predict_finish_real_kernel's lines (kernel.cpp:496-522)
					const double vv = fit_gave_up(a.fin_sdev) ? __builtin_nan("") : a.fin_self - qv;
					const double cf = cutoff_value(m * m, fabs(m), vv);
					if (a.fin_mean) a.fin_mean[m0 + t] = m;
					if (a.fin_var) a.fin_var[m0 + t] = vv;
					if (a.fin_cut) a.fin_cut[m0 + t] = m * cf / *a.fin_sdev;
				}
			}
		}
		__global__ void __launch_bounds__(NTHREADS, 1) predict_fused256_kernel(const PredictArgs a, int ksplit)
		{
			constexpr int KB = 16, TM = 64, WN = 2, NST = 3;
			__shared__ __attribute__((aligned(16))) double lds[2 * KB * (TM + 16) + NST * KB * BS + WN * TM + BN];
			const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
			const int simd = w & 3, half = w >> 2;
			const int wm = simd, wn = half ^ (simd & 1); // a SIMD's two waves on the two column groups (they thin out in opposite phase)
			const int m0 = blockIdx.x * TM;
			if (blockIdx.x == 0 && threadIdx.x == 0 && a.prune_thr > 0.0 && a.prune_stats != nullptr)
			{
				// a pruned predict that landed here contracted every row (cheaper than the statistics pass at this size; same bits either way)
				atomicAdd(a.prune_stats, static_cast<unsigned long long>(a.m_rows / BM));
				atomicAdd(a.prune_stats + 1, static_cast<unsigned long long>(a.m_rows / BM));
			}
			if (wn == 0) fused256_wave<0>(a, ksplit, lds, m0, wm);
			else fused256_wave<1>(a, ksplit, lds, m0, wm);
		}

		// ---- a handful of test points (the one-point predicts of main.cpp:75-101, evolve.cpp:298, mc.cpp:158-172) --------------
		// For M <= FEW_MAX typed rows the tiled paths pad to 128 rows, materialise K* and run a GEMM against all of T: 0.1-0.65 ms
		// per call, almost all of it launches, padding and copies.  Here the contraction is a triangular mat-vec per point with K*
		// generated on the fly: workgroup (bx, by) owns 64 rows of T and every KS-th 64-wide chunk of k up to its last row; k*(k)
		// of every point is computed once per workgroup and chunk into LDS (one exp per thread and point), each thread streams
		// two rows x 8 k of T per chunk (double2 loads, all eight in flight) and the partial products u = T k* go to a small
		// buffer; a second kernel adds the KS partial vectors, squares and sums them.  T is read once (lower triangle).
		constexpr int FEW_MAX = 16, FEW_ROWS = 64, FEW_KC = 64, FEW_KS_MAX = 8;
		__global__ void __launch_bounds__(256) predict_few_kernel(const PredictArgs a, int mt, double* __restrict__ upart, double* __restrict__ mupart)
		{
			__shared__ double ks[FEW_MAX][FEW_KC];
			__shared__ double red[4][FEW_MAX][FEW_ROWS];
			const int t = threadIdx.x, li = t & 31, sl = t >> 5; // row pair inside the block, k-slice
			const int i0 = blockIdx.x * FEW_ROWS;
			const int nchunks = blockIdx.x + 1; // T(i, k) = 0 for k > i
			const int KS = gridDim.y, by = blockIdx.y;
			const bool last = blockIdx.x == gridDim.x - 1;
			double u0[FEW_MAX], u1[FEW_MAX], mu[FEW_MAX];
#pragma unroll
			for (int m = 0; m < FEW_MAX; ++m) u0[m] = 0.0, u1[m] = 0.0, mu[m] = 0.0;
			for (int c = by; c < nchunks; c += KS)
			{
				const int kc = c * FEW_KC;
				// the eight double2 of this thread for the chunk: requested before the k* generation, consumed after it
				d2 tv[8];
				const double* __restrict__ tb = a.T + (i0 + 2 * li) + static_cast<long>(kc + sl) * a.ldt;
#pragma unroll
				for (int j = 0; j < 8; ++j) tv[j] = *reinterpret_cast<const d2*>(tb + static_cast<long>(8 * j) * a.ldt);
				__syncthreads();
				{
					const int k = kc + (t & 63);
					const int type_k = k >= a.n_split;
					const int pk = type_k ? k - a.n_split : k;
					const bool valid = pk < a.N;
					const int pc = valid ? pk : 0;
					const double xk = a.Xt[2 * pc], pkv = a.Xt[2 * pc + 1];
					for (int m = t >> 6; m < mt; m += 4)
					{
						const int type_m = m >= a.m_split_few;
						const int pidx = type_m ? m - a.m_split_few : m;
						const SEParam& p = a.ps.p[type_m + type_k];
						const double xm = a.Xs[2 * pidx], pm = a.Xs[2 * pidx + 1];
						const double d0 = (xm - xk) * p.rl0, d1 = (pm - pkv) * p.rl1;
						const double g = exp_nonpos(-0.5 * (d0 * d0 + d1 * d1));
						const double delta = (type_m == type_k && xm == xk && pm == pkv) ? p.n2 : 0.0;
						ks[m][t & 63] = valid ? p.amp * (g + delta) : 0.0;
					}
				}
				__syncthreads();
#pragma unroll
				for (int j = 0; j < 8; ++j)
				{
					const int kk = sl + 8 * j;
#pragma unroll
					for (int m = 0; m < FEW_MAX; ++m)
						if (m < mt)
						{
							const double kv = ks[m][kk];
							u0[m] = fma(tv[j].x, kv, u0[m]);
							u1[m] = fma(tv[j].y, kv, u1[m]);
						}
				}
				if (last && t < FEW_KC) // the last row block sees every k: it also reduces the mean k* . v
				{
					const double vk = a.v[kc + t];
#pragma unroll
					for (int m = 0; m < FEW_MAX; ++m)
						if (m < mt) mu[m] = fma(ks[m][t], vk, mu[m]);
				}
			}
			// sum over the 8 k-slices: the upper four hand theirs to the lower four through LDS, the lower four publish the sums
			__syncthreads();
			if (sl >= 4)
			{
#pragma unroll
				for (int m = 0; m < FEW_MAX; ++m)
					if (m < mt) red[sl - 4][m][2 * li] = u0[m], red[sl - 4][m][2 * li + 1] = u1[m];
			}
			__syncthreads();
			if (sl < 4)
			{
#pragma unroll
				for (int m = 0; m < FEW_MAX; ++m)
					if (m < mt) u0[m] += red[sl][m][2 * li], u1[m] += red[sl][m][2 * li + 1];
			}
			__syncthreads();
			if (sl < 4)
			{
#pragma unroll
				for (int m = 0; m < FEW_MAX; ++m)
					if (m < mt) red[sl][m][2 * li] = u0[m], red[sl][m][2 * li + 1] = u1[m];
			}
			__syncthreads();
			for (int idx = t; idx < mt * FEW_ROWS; idx += 256)
			{
				const int m = idx >> 6, r = idx & 63;
				upart[(static_cast<long>(by) * FEW_MAX + m) * a.n_total + i0 + r] = (red[0][m][r] + red[1][m][r]) + (red[2][m][r] + red[3][m][r]);
			}
			if (last && t < FEW_KC)
			{
#pragma unroll
				for (int m = 0; m < FEW_MAX; ++m)
					if (m < mt)
					{
						double s = mu[m];
#pragma unroll
						for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
						if (t == 0) mupart[by * FEW_MAX + m] = s;
					}
			}
		}
		// q[row(m)] = sum_i (sum_by upart)^2, mu[row(m)] = sum_by mupart; row(m): compact typed row -> padded layout; one workgroup per m
		__global__ void __launch_bounds__(256) predict_few_reduce_kernel(const double* __restrict__ upart, const double* __restrict__ mupart, int KS, int n_total,
			int m_split_few, int m_split, double* __restrict__ q, double* __restrict__ mu)
		{
			__shared__ double red[4];
			const int m = blockIdx.x;
			double s = 0.0;
			for (int i = threadIdx.x; i < n_total; i += 256)
			{
				double u = 0.0;
				for (int b = 0; b < KS; ++b) u += upart[(static_cast<long>(b) * FEW_MAX + m) * n_total + i];
				s = fma(u, u, s);
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
			if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
			__syncthreads();
			if (threadIdx.x == 0)
			{
				const int row = m >= m_split_few ? m_split + (m - m_split_few) : m;
				q[row] = (red[0] + red[1]) + (red[2] + red[3]);
				double ms = 0.0;
				for (int b = 0; b < KS; ++b) ms += mupart[b * FEW_MAX + m];
				mu[row] = ms;
			}
		}

		// plane blockIdx.y of the partial sums -> out[plane][row]
		__global__ void __launch_bounds__(256) sum_mu_kernel(const double* __restrict__ mu_part, int m_rows, int ksplit, double* __restrict__ out)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i >= m_rows) return;
			const double* __restrict__ p = mu_part + static_cast<long>(blockIdx.y) * ksplit * m_rows;
			double s = 0.0;
			// the planes are added in order; eight loads are in flight at a time (one after the other this loop was a chain of ksplit memory latencies:
			// 17 us for the 64 planes of a few-rows predict — the extra-point sets of the objective at small N)
			int ky = 0;
			for (; ky + 8 <= ksplit; ky += 8)
			{
				double v[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) v[u] = p[static_cast<long>(ky + u) * m_rows + i];
#pragma unroll
				for (int u = 0; u < 8; ++u) s += v[u];
			}
			for (; ky < ksplit; ++ky) s += p[static_cast<long>(ky) * m_rows + i];
			out[static_cast<long>(blockIdx.y) * m_rows + i] = s;
		}
		// q[m] = sum_n Z(n, m)^2 (column m of the n x rows matrix Z = T K*^T): one workgroup per test row
		__global__ void __launch_bounds__(256) colsumsq_kernel(const double* __restrict__ Z, long ldz, int n, double* __restrict__ q)
		{
			__shared__ double red[4];
			const double* __restrict__ z = Z + static_cast<long>(blockIdx.x) * ldz;
			double s = 0.0;
			for (int i = threadIdx.x; i < n; i += 256) s = fma(z[i], z[i], s);
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
			if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
			__syncthreads();
			if (threadIdx.x == 0) q[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
		}

		// Few test rows (the extra-point sets of the objective, opt.cpp:441-482; single-point lookups): rownorm_kernel
		// gives every 128 rows to one workgroup that walks its share of T, at best 1/8 of it (rownorm_split) — with a
		// handful of row blocks that still leaves most CUs idle.  There the contraction is spread over (n/64) x (rows/64)
		// tiles instead: Z = T K*^T by the triangular-K GEMM, then column sums of squares.  Measured crossover
		// (probes/predict_path_crossover.py): about 4096 rows at N = 1024, 2048 rows at N = 4096.
		constexpr int SMALL_M_WORK = 160; // row blocks x N-tiles
		constexpr size_t SMALL_M_Z_DOUBLES = size_t(1) << 26; // 512 MiB of Z at most
		// Tile groups per row block for the streaming kernel.  One workgroup per CU at a time (LDS), so a launch takes
		// ceil(blocks G / 256) rounds of 1/G the length: G is the power of two that minimises that (a little is charged per
		// extra group for the partial sums); every group needs a snake pair of N-tiles.
		constexpr int ROWNORM_SPLIT_MAX = 8;
		int rownorm_split(int m_rows, int n_total)
		{
			const long blocks = m_rows / BM, ntiles = n_total / BN;
			static const int forced = [] {
				const char* e = getenv("GPLE_ROWNORM_SPLIT"); // A/B: groups per row block whatever the model says (must leave every group a snake pair)
				return e ? atoi(e) : 0;
			}();
			if (forced > 0 && (forced == 1 || 4 * forced <= 2 * ntiles) && forced <= ROWNORM_SPLIT_MAX) return forced;
			int best = 1;
			double best_cost = 1e300;
			for (int g = 1; g <= ROWNORM_SPLIT_MAX; g *= 2)
			{
				if (g > 1 && 4 * g > 2 * ntiles) break; // fewer than one snake pair per group
				const double cost = static_cast<double>((blocks * g + 255) / 256) / g * (1.0 + 0.02 * (g - 1));
				if (cost < best_cost - 1e-12) best_cost = cost, best = g;
			}
			return best;
		}
		int device_cu_count()
		{
			static const int n = [] {
				int dev = 0, cus = 256;
				if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
				return cus > 0 ? cus : 256;
			}();
			return n;
		}
		bool small_m(const PredictArgs& a)
		{
			const bool fits = static_cast<size_t>(a.m_rows) * a.n_total <= SMALL_M_Z_DOUBLES;
			const char* force = getenv("GPLE_PREDICT_SMALL_M"); // "0" / "1": A/B runs and the path-against-path parity test
			if (force && (force[0] == '0' || force[0] == '1')) return force[0] == '1' && fits;
			// one N-tile (n <= 256): the streaming kernel has nothing to split, but from 64 row blocks on it fills a quarter of the chip with
		// 16-step workgroups and beats the GEMM + column-sum pair, which moves K* and Z through HBM three times (C1: 47 vs 58 us)
		if (a.n_total / BN == 1 && a.m_rows / BM >= 64) return false;
		return static_cast<long>(a.m_rows / BM) * (a.n_total / BN) <= SMALL_M_WORK && fits;
		}
	} // namespace

	size_t predict_scratch_doubles(const PredictArgs& a, int* chunk_rows, bool* few_rows)
	{
		// rolling K* chunk: as many 128-row tiles as fit PREDICT_SCRATCH_BYTES (at least one), never more than needed
		const size_t per_row = static_cast<size_t>(a.n_total) * sizeof(double);
		size_t rows = PREDICT_SCRATCH_BYTES / per_row / BM * BM;
		if (rows < static_cast<size_t>(BM)) rows = BM;
		if (rows > static_cast<size_t>(a.m_rows)) rows = a.m_rows;
		const bool small = small_m(a); // decided once per call: launch_predict_q gets the same answer
		*few_rows = small;
		if (small) rows = a.m_rows; // one chunk (<= 512 MiB), plus Z of the same size
		*chunk_rows = static_cast<int>(rows);
		return rows * a.n_total + static_cast<size_t>(gen_ksplit(a.m_rows)) * a.m_rows * (a.dv ? (a.complex_deriv ? 15 : 7) : 1)
			+ (small ? rows * a.n_total : 0) + static_cast<size_t>(ROWNORM_SPLIT_MAX) * a.m_rows
			+ (a.prune_thr > 0.0 || a.cut_thr > 0.0 || a.mean_only ? (static_cast<size_t>(gen_ksplit(a.m_rows)) + 1) * a.m_rows + 64 : 0); // + K*^2 sums, list and positions (2 ints per row)
	}

	bool predict_is_few(const PredictArgs& a)
	{
		static const bool off = [] {
			const char* e = getenv("GPLE_PREDICT_FEW");
			return e && e[0] == '0';
		}();
		const int mt = (a.m_split < a.m_rows ? 2 : 1) * a.M; // typed rows
		return !off && a.dv == nullptr && mt <= FEW_MAX;
	}
	static int predict_few_ksplit(int n_total)
	{
		const int nblk = n_total / FEW_ROWS;
		int ks = (256 + nblk - 1) / nblk; // about one workgroup per CU
		return ks < 1 ? 1 : (ks > FEW_KS_MAX ? FEW_KS_MAX : ks);
	}
	size_t predict_few_scratch_doubles(const PredictArgs& a)
	{
		return static_cast<size_t>(FEW_KS_MAX) * FEW_MAX * (static_cast<size_t>(a.n_total) + 1);
	}
	hipError_t launch_predict_few(hipStream_t s, const PredictArgs& a0, double* scratch)
	{
		PredictArgs a = a0;
		const bool cplx = a.m_split < a.m_rows;
		const int mt = (cplx ? 2 : 1) * a.M;
		a.m_split_few = cplx ? a.M : mt;
		const int nblk = a.n_total / FEW_ROWS, KS = predict_few_ksplit(a.n_total);
		double* upart = scratch;
		double* mupart = scratch + static_cast<size_t>(FEW_KS_MAX) * FEW_MAX * a.n_total;
		// a k-slice that owns no chunk of a row block (by >= chunks of the block) still writes its zeros: no stale partials
		hipLaunchKernelGGL(predict_few_kernel, dim3(nblk, KS), dim3(256), 0, s, a, mt, upart, mupart);
		hipLaunchKernelGGL(predict_few_reduce_kernel, dim3(mt), dim3(256), 0, s, upart, mupart, KS, a.n_total, a.m_split_few, a.m_split, a.q, a.mu);
		return hipGetLastError();
	}

	// ---- part of a predict beside the fit it follows ------------------------------------------------------------------------------------------
	// The contraction needs T, the fit's last product but for its first rows: T(0 .. early_rows, .) is final at the first fork of the factorisation
	// (chol_inverse_factor), a third of the way into a fit at n = 4096, and the fit hardly uses the chip from there on (a latency-bound spine, GEMMs
	// on a side stream).  Here, on a stream of the context's own: K* of the rows (store only: the mean needs v, the fit's LAST product) as soon as
	// the points are there, then — behind the fork's event — rownormp_kernel over the N-tiles below early_rows as a work queue of FEWER workgroups
	// than CUs (a contraction workgroup shares its CU with nothing: the CUs it leaves out are the fit's), saving every virtual group's per-lane sums.
	// On the main stream, behind the fit: the means (the generation pass without the store), then the late tiles on all CUs, starting from the saved
	// sums: bit for bit the unsplit contraction (Prune::state).  Buffers of its own (the pooled scratch is ordered by the main stream).
	// GPLE_PREDICT_OVERLAP=1 switches it on; GPLE_PREDICT_OVERLAP_CUS = CUs left to the fit (64).  DEFAULT OFF, because it does not pay: measured on the
	// per-rank proxy (probes/overlap_sweep.sh, N = 4096, an eighth of the 512 x 512 grid) the step is 9.41 ms in turn and 9.46 ... 17 ms overlapped — the
	// one-launch factorisation slows by 0.3 - 1.0 ms beside ANY early work (its GEMMs and tile tasks want the CUs the early part holds), which is
	// more than the early part saves (profiles/r04_notes.md).  Kept as a tested, bit-exact experiment for a fit that tolerates company.
	namespace
	{
		__global__ void set_int2_kernel(int* p, int a, int b) { p[0] = a, p[1] = b; }
		struct OverlapBuffers
		{
			double *Ks, *state, *qpart, *mu_part, *nrm_part, *xs;
			int* counters; // [0] n_live (= rows: every row is live), [1] the work queue's counter
			size_t bytes;
		};
		OverlapBuffers overlap_layout(char* base, const PredictArgs& a, size_t xs_doubles)
		{
			OverlapBuffers b{};
			const size_t ksplit = gen_ksplit(a.m_rows), nblk = a.m_rows / BM;
			size_t off = 0;
			auto take = [&](size_t doubles) {
				double* p = base ? reinterpret_cast<double*>(base + off) : nullptr;
				off += (doubles * sizeof(double) + 255) / 256 * 256;
				return p;
			};
			b.Ks = take(static_cast<size_t>(a.m_rows) * a.n_total);
			b.state = take(nblk * (VG * 8 * 4 * 64));
			b.qpart = take(static_cast<size_t>(VG) * a.m_rows);
			b.mu_part = take(ksplit * a.m_rows);
			b.nrm_part = take(ksplit * a.m_rows);
			b.xs = take(xs_doubles);
			b.counters = reinterpret_cast<int*>(take(32));
			b.bytes = off;
			return b;
		}
	} // namespace
	bool predict_overlap_enabled()
	{
		static const bool on = [] {
			const char* e = getenv("GPLE_PREDICT_OVERLAP");
			return e != nullptr && atoi(e) != 0;
		}();
		return on;
	}
	// can this predict start beside the fit whose first early_rows rows of T are final early?  (one chunk, the full contraction on rownormp_kernel<4,4>,
	// no derivative pass, at least two N-tiles on either side of the split)
	bool predict_overlap_applicable(const Ctx* ctx, const PredictArgs& a, int early_rows)
	{
		if (!predict_overlap_enabled() || early_rows < 2 * BN || a.dv || a.mean_only || a.cut_thr > 0.0 || a.prune_thr > 0.0) return false;
		if (a.m_rows % BM || a.n_total % BN || a.n_total / BN < 8 || a.n_total / BN > TL_MAX || early_rows / BN > a.n_total / BN - 2) return false;
		if (static_cast<size_t>(a.m_rows) * a.n_total * sizeof(double) > PREDICT_SCRATCH_BYTES) return false; // one chunk
		if (ctx->rownorm_pipe == 0) return false;
		return !small_m(a) && a.m_rows / BM >= 16;
	}
	// the stream, events and buffers of the early work (sized for `a` and xs_doubles of test-point storage); *xs: where a caller may put the points
	hipError_t predict_overlap_prepare(Ctx* ctx, const PredictArgs& a, size_t xs_doubles, double** xs)
	{
		hipError_t e;
		if (!ctx->early_stream)
		{
			if ((e = hipStreamCreateWithFlags(&ctx->early_stream, hipStreamNonBlocking)) != hipSuccess) return e;
			for (hipEvent_t* ev : {&ctx->early_points, &ctx->early_done, &ctx->early_free})
				if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return e;
		}
		const OverlapBuffers need = overlap_layout(nullptr, a, xs_doubles);
		if (need.bytes > ctx->early_bytes)
		{
			if ((e = hipDeviceSynchronize()) != hipSuccess) return e; // (rare: the buffers only grow)
			if (ctx->early_buf) (void)hipFree(ctx->early_buf);
			ctx->early_buf = nullptr, ctx->early_bytes = 0, ctx->early_free_pending = false;
			if ((e = hipMalloc(&ctx->early_buf, need.bytes)) != hipSuccess) return e;
			ctx->early_bytes = need.bytes;
		}
		// the early stream may touch the buffers once the previous predict's main-stream part is through with them
		if (ctx->early_free_pending && (e = hipStreamWaitEvent(ctx->early_stream, ctx->early_free, 0)) != hipSuccess) return e;
		ctx->early_free_pending = false;
		*xs = overlap_layout(static_cast<char*>(ctx->early_buf), a, xs_doubles).xs;
		return hipSuccess;
	}
	// a.Xs must be readable on the early stream: now, or — points_ready, an event recorded on it — behind work already enqueued there; fills a.q and a.mu like launch_predict_q
	hipError_t launch_predict_overlapped(Ctx* ctx, hipStream_t s, const PredictArgs& a, size_t xs_doubles, int early_rows, hipEvent_t t_early, hipEvent_t points_ready)
	{
		hipStream_t E = ctx->early_stream;
		const OverlapBuffers b = overlap_layout(static_cast<char*>(ctx->early_buf), a, xs_doubles);
		if (!E || b.bytes > ctx->early_bytes) return hipErrorInvalidValue;
		static const int fit_cus = [] {
			const char* e = getenv("GPLE_PREDICT_OVERLAP_CUS");
			return e ? atoi(e) : 64;
		}();
		// GPLE_PREDICT_OVERLAP_TILES: at most that many N-tiles go early (the early part should end with the fit, not after it);
		// GPLE_PREDICT_OVERLAP_KSTAR_LATE=1: the generation waits for the early rows too (it disturbs the factorisation's first launch otherwise)
		static const int max_tiles = [] {
			const char* e = getenv("GPLE_PREDICT_OVERLAP_TILES");
			return e ? atoi(e) : 1 << 20;
		}();
		static const bool kstar_late = [] {
			const char* e = getenv("GPLE_PREDICT_OVERLAP_KSTAR_LATE");
			return e != nullptr && atoi(e) != 0;
		}();
		const int ksplit = gen_ksplit(a.m_rows), ntiles = a.n_total / BN, jt_split = std::max(2, std::min(early_rows / BN, max_tiles)), nblocks = a.m_rows / BM;
		const int* none = nullptr;
		hipError_t e;
		// early stream: K*, store only
		if (kstar_late && (e = hipStreamWaitEvent(E, t_early, 0)) != hipSuccess) return e;
		hipLaunchKernelGGL((kstar_gen_kernel<0, 3>), dim3(a.m_rows / 128, ksplit), dim3(128), 0, E, a, 0, a.m_rows, b.Ks, static_cast<double*>(nullptr), static_cast<double*>(nullptr), none, none);
		hipLaunchKernelGGL(set_int2_kernel, dim3(1), dim3(1), 0, E, b.counters, a.m_rows, 0);
		if ((e = hipStreamWaitEvent(E, t_early, 0)) != hipSuccess) return e;
		{
			// the tiles below the split: one virtual group per unit (the finest units), fewer resident workgroups than CUs
			int G = 1;
			while (2 * G <= ROWNORM_SPLIT_MAX && 4 * (2 * G) <= 2 * ntiles) G *= 2;
			Prune pr{b.counters + 1, b.counters, 0, nblocks, G};
			pr.jt_lo = 0, pr.jt_hi = jt_split, pr.state = b.state, pr.state_mode = 1;
			const int wgs = std::max(16, std::min(device_cu_count() - fit_cus, nblocks * G));
			hipLaunchKernelGGL((rownormp_kernel<4, 4, true>), dim3(wgs), dim3(NTHREADS), 0, E, b.Ks, a.m_rows, a.T, a.ldt, a.n_total, b.qpart, static_cast<long>(a.m_rows), pr);
		}
		if ((e = hipEventRecord(ctx->early_done, E)) != hipSuccess) return e;
		// main stream (behind the fit): the means, then the late tiles from the saved sums
		if (points_ready && (e = hipStreamWaitEvent(s, points_ready, 0)) != hipSuccess) return e;
		hipLaunchKernelGGL((kstar_gen_kernel<0, 1>), dim3(a.m_rows / 128, ksplit), dim3(128), 0, s, a, 0, a.m_rows, b.Ks, b.mu_part, b.nrm_part, none, none);
		if ((e = hipStreamWaitEvent(s, ctx->early_done, 0)) != hipSuccess) return e;
		{
			Prune pr{};
			pr.jt_lo = jt_split, pr.jt_hi = ntiles, pr.state = b.state, pr.state_mode = 2;
			const int split = rownorm_split(a.m_rows, a.n_total);
			chunk_timer_start(ctx);
			ctx->last_contraction = "rownormp_kernel<4,4,false>";
			hipLaunchKernelGGL((rownormp_kernel<4, 4, false>), dim3(nblocks, split), dim3(NTHREADS), 0, s, b.Ks, a.m_rows, a.T, a.ldt, a.n_total, b.qpart, static_cast<long>(a.m_rows), pr);
			chunk_timer_stop(ctx);
		}
		hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, 1), dim3(256), 0, s, b.qpart, a.m_rows, VG, a.q);
		hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, 1), dim3(256), 0, s, b.mu_part, a.m_rows, ksplit, a.mu);
		if ((e = hipEventRecord(ctx->early_free, s)) != hipSuccess) return e;
		ctx->early_free_pending = true;
		ctx->overlapped_predicts += 1;
		return hipGetLastError();
	}

	hipError_t launch_predict_q(Ctx* ctx, hipStream_t s, const PredictArgs& a, double* scratch, int chunk_rows, bool few_rows, bool* finished)
	{
		if (finished) *finished = false;
		if (a.M <= 0) return hipSuccess;
		if (a.m_rows % BM || a.n_total % BN || a.m_split % BM || a.n_split % BN || chunk_rows % BM || chunk_rows <= 0)
			return hipErrorInvalidValue;
		// 0: 8 waves x (16 rows x 256 columns); 2 / 3: 8 waves x (4 x 4) / (2 x 8) fragments.  Measured at C2 / C4r (TFLOP/s, same
		// box): 60.1 / 65.07, 59.6 / 65.78, 60.4 / 65.75 (two more shapes — 4 waves with BK 8, and 4 waves on a 128 x 128 tile, both
		// with two workgroups per CU — were 61 and 64.0 at C4r and are gone; profiles/r02_notes.md).  Default: 2 from eight N-tiles
		// on (+1.1 % at C4r), else 0.
		static const int forced = [] {
			const char* e = getenv("GPLE_ROWNORM_VARIANT");
			return e ? atoi(e) : -1;
		}();
		// Round 2: with the LDS-DMA staging the 2 x 8 blocking led below eight N-tiles (C2: 64.3 vs 62.9 / 61.2 TFLOP/s).  Round 4, rownormp_kernel: 4 x 4 leads at
		// C2 too (three same-box pairs: 0.835 / 0.841 / 0.846 of the peak against 0.830 / 0.838 / 0.836); one and two N-tiles stay on 2 x 8, the arithmetic of
		// rownorm3_kernel and predict_fused256_kernel (the same rows come out bit for bit whichever of the three a call gets)
		const int variant = forced >= 0 ? forced : (a.n_total / BN >= 3 ? 2 : 3);
		// short factors with few row blocks (C1): 64-row workgroups with three slabs in flight (rownorm3_kernel; same bits as <2,8>).  GPLE_ROWNORM_SHORT=0: A/B
		static const bool short_ok = [] {
			const char* e = getenv("GPLE_ROWNORM_SHORT");
			return e == nullptr || atoi(e) != 0;
		}();
		// the k-steps of a unit as one pipeline (rownormp_kernel; same bits).  GPLE_ROWNORM_PIPE=0: rownorm2_kernel, A/B
		static const bool pipe_env = [] {
			const char* e = getenv("GPLE_ROWNORM_PIPE");
			return (e == nullptr || atoi(e) != 0);
		}();
		const bool pipe = ctx->rownorm_pipe < 0 ? pipe_env : ctx->rownorm_pipe != 0;
		if (a.n_total / BN > TL_MAX) return hipErrorInvalidValue;
		const bool short_factor = short_ok && variant == 3 && a.n_total / BN <= 2 && a.m_rows / BM <= 2 * device_cu_count();
		double* Ks = scratch;
		double* mu_part = scratch + static_cast<size_t>(chunk_rows) * a.n_total;
		const bool small = few_rows && chunk_rows == a.m_rows;
		const int ksplit = gen_ksplit(a.m_rows);
		double* Z = mu_part + static_cast<size_t>(ksplit) * a.m_rows * (a.dv ? (a.complex_deriv ? 15 : 7) : 1);
		// groups per row block: by the number of row blocks (rownorm_split); the value of q does not depend on it (virtual groups)
		int split = small ? 1 : rownorm_split(a.m_rows, a.n_total);
		double* qpart = Z + (small ? static_cast<size_t>(chunk_rows) * a.n_total : 0);
		const int dplanes = a.dv ? (a.complex_deriv ? 15 : 7) : 1;
		if (a.mean_only)
		{
			// only the (uncut) mean is consumed: one generation pass without storing K*, no contraction
			double* nrm_part = qpart + static_cast<size_t>(VG) * a.m_rows;
			const dim3 ggrid(a.m_rows / 128, ksplit);
			const int* none = nullptr;
			if (a.dv && a.complex_deriv) hipLaunchKernelGGL((kstar_gen_kernel<2, 1>), ggrid, dim3(128), 0, s, a, 0, a.m_rows, Ks, mu_part, nrm_part, none, none);
			else if (a.dv) hipLaunchKernelGGL((kstar_gen_kernel<1, 1>), ggrid, dim3(128), 0, s, a, 0, a.m_rows, Ks, mu_part, nrm_part, none, none);
			else hipLaunchKernelGGL((kstar_gen_kernel<0, 1>), ggrid, dim3(128), 0, s, a, 0, a.m_rows, Ks, mu_part, nrm_part, none, none);
			const hipError_t e = hipMemsetAsync(a.q, 0, static_cast<size_t>(a.m_rows) * sizeof(double), s);
			if (e != hipSuccess) return e;
			hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, 1), dim3(256), 0, s, mu_part, a.m_rows, ksplit, a.mu);
			if (a.dv) hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, dplanes), dim3(256), 0, s, mu_part, a.m_rows, ksplit, a.dacc);
			return hipGetLastError();
		}
		// the real GP with one N-tile (N <= 256): generation, contraction and both sums in one launch, whatever the number of rows; a pruned request
		// contracts every row there (same bits, cheaper than the statistics pass).  GPLE_PREDICT_FUSED_SMALL=0: the separate kernels, A/B
		static const bool fused_env = [] {
			const char* e = getenv("GPLE_PREDICT_FUSED_SMALL");
			return e == nullptr || atoi(e) != 0;
		}();
		if ((ctx->fused_small < 0 ? fused_env : ctx->fused_small != 0) && a.n_total == BN && a.m_split == a.m_rows && !a.dv && !(a.cut_thr > 0.0))
		{
			chunk_timer_start(ctx);
			ctx->last_contraction = "predict_fused256_kernel";
			hipLaunchKernelGGL(predict_fused256_kernel, dim3(a.m_rows / 64), dim3(NTHREADS), 0, s, a, ksplit);
			chunk_timer_stop(ctx);
			if (finished) *finished = a.fin_sdev != nullptr;
			return hipGetLastError();
		}
		// the variance only decides the cut-off factor: points whose mean is large enough for factor 1 whatever the variance are not contracted
		const bool cut_only = a.cut_thr > 0.0 && !small && a.prune_stats != nullptr;
		// far-row pruning (Prune): streaming kernels only, no derivative pass
		const bool prune = !cut_only && a.prune_thr > 0.0 && !a.dv && !small && a.prune_stats != nullptr;
		auto launch_rownorm = [&](bool queue_mode, dim3 grid, int rows, double* qdst, const Prune& pr) {
			auto launch = [&](auto full_kernel, auto queue_kernel) {
				if (queue_mode) hipLaunchKernelGGL(queue_kernel, grid, dim3(NTHREADS), 0, s, Ks, rows, a.T, a.ldt, a.n_total, qdst, static_cast<long>(a.m_rows), pr);
				else hipLaunchKernelGGL(full_kernel, grid, dim3(NTHREADS), 0, s, Ks, rows, a.T, a.ldt, a.n_total, qdst, static_cast<long>(a.m_rows), pr);
			};
			if (variant == 3 && pipe) { launch(rownormp_kernel<2, 8, false>, rownormp_kernel<2, 8, true>); ctx->last_contraction = queue_mode ? "rownormp_kernel<2,8,true>" : "rownormp_kernel<2,8,false>"; }
			else if (variant == 2 && pipe) { launch(rownormp_kernel<4, 4, false>, rownormp_kernel<4, 4, true>); ctx->last_contraction = queue_mode ? "rownormp_kernel<4,4,true>" : "rownormp_kernel<4,4,false>"; }
			else if (variant == 3) { launch(rownorm2_kernel<2, 8, false>, rownorm2_kernel<2, 8, true>); ctx->last_contraction = queue_mode ? "rownorm2_kernel<2,8,true>" : "rownorm2_kernel<2,8,false>"; }
			else if (variant == 2) { launch(rownorm2_kernel<4, 4, false>, rownorm2_kernel<4, 4, true>); ctx->last_contraction = queue_mode ? "rownorm2_kernel<4,4,true>" : "rownorm2_kernel<4,4,false>"; }
			else { launch(rownorm_kernel<8, 16, false>, rownorm_kernel<8, 16, true>); ctx->last_contraction = queue_mode ? "rownorm_kernel<8,16,true>" : "rownorm_kernel<8,16,false>"; }
		};
		if (prune || cut_only)
		{
			// 1. partial means and partial sums of K*^2 of every row (no K* stored); 2. the live rows -> list; 3. per chunk OF THE LIST:
			// K* of the listed rows, row norms by a work queue; 4. q back to the rows.  The number of live rows stays on the device:
			// launches that turn out to have nothing to do return at once.
			// cut_only: step 1 also forms the derivative planes, the means are summed at once and step 2 lists the points whose |mu|^2 leaves
			// the cut-off factor open.
			double* nrm_part = qpart + static_cast<size_t>(VG) * a.m_rows;
			int* list = reinterpret_cast<int*>(nrm_part + static_cast<size_t>(ksplit) * a.m_rows);
			int* pos = list + a.m_rows;
			int* n_live = reinterpret_cast<int*>(a.prune_stats + 3);
			int* queue = reinterpret_cast<int*>(a.prune_stats + 2);
			hipError_t e = hipMemsetAsync(n_live, 0, sizeof(int), s);
			if (e != hipSuccess) return e;
			const dim3 ggrid(a.m_rows / 128, ksplit);
			const int* none = nullptr;
			if (a.dv && a.complex_deriv) hipLaunchKernelGGL((kstar_gen_kernel<2, 1>), ggrid, dim3(128), 0, s, a, 0, a.m_rows, Ks, mu_part, nrm_part, none, none);
			else if (a.dv) hipLaunchKernelGGL((kstar_gen_kernel<1, 1>), ggrid, dim3(128), 0, s, a, 0, a.m_rows, Ks, mu_part, nrm_part, none, none);
			else hipLaunchKernelGGL((kstar_gen_kernel<0, 1>), ggrid, dim3(128), 0, s, a, 0, a.m_rows, Ks, mu_part, nrm_part, none, none);
			if (cut_only)
			{
				hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, 1), dim3(256), 0, s, mu_part, a.m_rows, ksplit, a.mu);
				hipLaunchKernelGGL(compact_by_mean_kernel, dim3((a.m_rows + 255) / 256), dim3(256), 0, s, a.mu, a.m_rows, a.m_split, a.M, a.cut_thr, list, pos, n_live);
			}
			else
				hipLaunchKernelGGL(compact_rows_kernel, dim3((a.m_rows + 255) / 256), dim3(256), 0, s, nrm_part, ksplit, a.m_rows, a.prune_thr, list, pos, n_live);
			// the number of live row blocks is only known on the device: the finest units (most groups per block that still hold a snake pair
			// of N-tiles each) keep the last round of the queue short whatever it turns out to be — 93 live blocks in 2 groups fill 186 of 256
			// CUs once, in 8 groups they make 2.9 rounds of an eighth
			static const int queue_split = [] {
				const char* e = getenv("GPLE_PREDICT_QUEUE_SPLIT"); // 0: the split of the full launch (>= 2), as before
				return e ? atoi(e) : ROWNORM_SPLIT_MAX;
			}();
			if (split < 2 && a.n_total / BN >= 4) split = 2;
			for (int g = split * 2; g <= queue_split && 4 * g <= 2 * (a.n_total / BN); g *= 2) split = g;
			for (int c0 = 0; c0 < a.m_rows; c0 += chunk_rows)
			{
				const int rows = a.m_rows - c0 < chunk_rows ? a.m_rows - c0 : chunk_rows;
				hipLaunchKernelGGL((kstar_gen_kernel<0, 2>), dim3(rows / 128, ksplit), dim3(128), 0, s, a, c0, rows, Ks, mu_part, nrm_part, list, n_live);
				if ((e = hipMemsetAsync(queue, 0, sizeof(int), s)) != hipSuccess) return e;
				const int nblocks = rows / BM;
				const Prune pr{queue, n_live, c0, nblocks, split};
				chunk_timer_start(ctx);
				launch_rownorm(true, dim3(std::min(nblocks * split, 2 * device_cu_count())), rows, qpart + c0, pr);
				chunk_timer_stop(ctx);
			}
			hipLaunchKernelGGL(scatter_q_kernel, dim3((a.m_rows + 255) / 256), dim3(256), 0, s, qpart, static_cast<long>(a.m_rows), pos, a.m_rows, a.q, n_live,
				cut_only ? static_cast<unsigned long long*>(nullptr) : a.prune_stats); // (the far-row statistics count far rows only)
		}
		else
		{
			for (int row0 = 0; row0 < a.m_rows; row0 += chunk_rows)
			{
				const int rows = a.m_rows - row0 < chunk_rows ? a.m_rows - row0 : chunk_rows;
				const dim3 ggrid(rows / 128, ksplit);
				const int* none = nullptr;
				if (a.dv && a.complex_deriv) hipLaunchKernelGGL((kstar_gen_kernel<2, 0>), ggrid, dim3(128), 0, s, a, row0, rows, Ks, mu_part, static_cast<double*>(nullptr), none, none);
				else if (a.dv) hipLaunchKernelGGL((kstar_gen_kernel<1, 0>), ggrid, dim3(128), 0, s, a, row0, rows, Ks, mu_part, static_cast<double*>(nullptr), none, none);
				else hipLaunchKernelGGL((kstar_gen_kernel<0, 0>), ggrid, dim3(128), 0, s, a, row0, rows, Ks, mu_part, static_cast<double*>(nullptr), none, none);
				chunk_timer_start(ctx);
				if (small)
				{
					GemmDesc g{};
					g.A = a.T, g.lda = a.ldt, g.B = Ks, g.ldb = rows, g.C = Z, g.ldc = a.n_total;
					g.M = a.n_total, g.N = rows, g.K = a.n_total, g.batch = 1, g.alpha = 1.0, g.beta = 0.0;
					g.krange = K_LE_M; // T(n, k) = 0 for k > n
					g.a_kmajor = false, g.b_kmajor = false, g.c_trans = false;
					const hipError_t e = launch_gemm(s, g, gemm_pick_tile(a.n_total, rows, 1, true));
					if (e != hipSuccess) return e;
					ctx->last_contraction = "gemm_f64 (Z = T K*^T) + colsumsq_kernel";
					hipLaunchKernelGGL(colsumsq_kernel, dim3(rows), dim3(256), 0, s, Z, static_cast<long>(a.n_total), a.n_total, a.q + row0);
				}
				else if (short_factor)
				{
					ctx->last_contraction = "rownorm3_kernel";
					hipLaunchKernelGGL(rownorm3_kernel, dim3(rows / 64, split), dim3(NTHREADS), 0, s, Ks, rows, a.T, a.ldt, a.n_total, qpart + row0, static_cast<long>(a.m_rows));
				}
				else
					launch_rownorm(false, dim3(rows / BM, split), rows, qpart + row0, Prune{});
				chunk_timer_stop(ctx);
			}
			if (!small) hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, 1), dim3(256), 0, s, qpart, a.m_rows, VG, a.q); // the VG planes, in order
		}
		// a.mu receives plane 0 (the mean); with derivatives a.dacc receives all 7 / 15 planes (plane 0 = the mean again)
		if (!cut_only) hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, 1), dim3(256), 0, s, mu_part, a.m_rows, ksplit, a.mu);
		if (a.dv) hipLaunchKernelGGL(sum_mu_kernel, dim3((a.m_rows + 255) / 256, dplanes), dim3(256), 0, s, mu_part, a.m_rows, ksplit, a.dacc);
		return hipGetLastError();
	}
} // namespace gple
