// Standalone timing probe for the Cholesky panel kernel variants (not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 probes/panel_probe.hip -o probes/panel_probe && probes/panel_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <utility>
#include <vector>

constexpr int NB = 64, LR = NB + 2, THREADS = 256;

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_newton(double d)
{
	double r = __builtin_amdgcn_rsq(d);
	r = r * fma(-0.5 * d * r, r, 1.5);
	r = r * fma(-0.5 * d * r, r, 1.5);
	return r;
}
// branch-free time stamps (a branch between the unrolled phases makes hipcc spill): every thread stores, compile-time switch
#define STAMP(i) \
	if constexpr (ST) stamps[(i) * (gridDim.x * blockDim.x) + blockIdx.x * blockDim.x + threadIdx.x] = clock64();

// V0: barrier version (round-1 baseline): wave 0 = diagonal rows, waves 1..3 = 192 panel rows
template <bool ST>
__global__ void __launch_bounds__(THREADS) panel_v0(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double col[NB];
	__shared__ __attribute__((aligned(16))) double pv[NB];
	__shared__ double dump[THREADS];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool diag = w == 0;
	const int row = diag ? lane : NB + blockIdx.x * 192 + (w - 1) * 64 + lane;
	const bool valid = row < m;
	double* const my_pv = diag ? &pv[lane] : &dump[threadIdx.x];
	double* const my_col = diag ? &col[lane] : &dump[threadIdx.x];
	double a[NB];
	const double* __restrict__ src = A + (valid ? row : 0);
	STAMP(0)
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	STAMP(1)
#pragma unroll
	for (int k = 0; k < NB; ++k)
	{
		*my_pv = a[k];
		__syncthreads();
		const double d = pv[k];
		const double r = rsqrt_newton(d);
		double sd = d * r;
		sd = fma(fma(-sd, sd, d), 0.5 * r, sd);
		const double lp = a[k] * r;
		const double l = diag ? (lane == k ? sd : (lane > k ? lp : 0.0)) : lp;
		a[k] = l;
		*my_col = l;
		__syncthreads();
#pragma unroll
		for (int j = k + 1; j < NB; ++j) a[j] = fma(-l, col[j], a[j]);
	}
	STAMP(2)
	if (valid && (!diag || blockIdx.x == 0))
	{
		double* __restrict__ dst = A + row;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
	STAMP(3)
}

// V3: V0 with one barrier per column: wave 0 takes the pivot by readlane, publishes 1/L_kk next to the scaled column
// (double-buffered), the other waves scale their own entry after the barrier.  All waves run the same instruction stream.
template <bool ST, int NT>
__global__ void __launch_bounds__(NT) panel_v3(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double col[2][NB + 2];
	__shared__ double dump[NT];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool diag = w == 0;
	const int row = diag ? lane : NB + blockIdx.x * (NT - 64) + (w - 1) * 64 + lane;
	const bool valid = row < m;
	double a[NB];
	const double* __restrict__ src = A + (valid ? row : 0);
	STAMP(0)
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	STAMP(1)
#pragma unroll
	for (int k = 0; k < NB; ++k)
	{
		double* const cb = &col[k & 1][0];
		double* const my_col = diag ? &cb[lane] : &dump[threadIdx.x];
		double* const my_r = (diag && lane == 0) ? &cb[NB] : &dump[threadIdx.x];
		const double d = readlane_f64(a[k], k); // the pivot in wave 0; a harmless number elsewhere
		const double r0 = rsqrt_newton(d);
		double sd = d * r0;
		sd = fma(fma(-sd, sd, d), 0.5 * r0, sd);
		*my_col = lane == k ? sd : (lane > k ? a[k] * r0 : 0.0);
		*my_r = r0;
		__syncthreads();
		const double r = cb[NB];
		const double lp = a[k] * r;
		const double l = diag ? (lane == k ? sd : (lane > k ? lp : 0.0)) : lp;
		a[k] = l;
#pragma unroll
		for (int j = k + 1; j < NB; ++j) a[j] = fma(-l, cb[j], a[j]);
	}
	STAMP(2)
	if (valid && (!diag || blockIdx.x == 0))
	{
		double* __restrict__ dst = A + row;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
	STAMP(3)
}

// V4: V3 software-pipelined: the pivot chain of column k+1 (readlane -> rsqrt -> scaled column) is started right after the
// first FMA of column k, so its latency runs under the remaining FMAs of column k instead of after them.
template <bool ST>
__global__ void __launch_bounds__(THREADS) panel_v4(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double col[2][NB + 2];
	__shared__ double dump[THREADS];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool diag = w == 0;
	const int row = diag ? lane : NB + blockIdx.x * 192 + (w - 1) * 64 + lane;
	const bool valid = row < m;
	double a[NB];
	const double* __restrict__ src = A + (valid ? row : 0);
	STAMP(0)
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	STAMP(1)
	double* const my_slot0 = diag ? &col[0][lane] : &dump[threadIdx.x];
	double* const my_slot1 = diag ? &col[1][lane] : &dump[threadIdx.x];
	double* const my_r0 = (diag && lane == 0) ? &col[0][NB] : &dump[threadIdx.x];
	double* const my_r1 = (diag && lane == 0) ? &col[1][NB] : &dump[threadIdx.x];
	// prologue: column 0
	double sd;
	{
		const double d = readlane_f64(a[0], 0);
		const double r0 = rsqrt_newton(d);
		sd = d * r0;
		sd = fma(fma(-sd, sd, d), 0.5 * r0, sd);
		*my_slot0 = lane == 0 ? sd : a[0] * r0;
		*my_r0 = r0;
	}
#pragma unroll
	for (int k = 0; k < NB; ++k)
	{
		const double* const cb = &col[k & 1][0];
		__syncthreads();
		const double r = cb[NB];
		const double lp = a[k] * r;
		const double l = diag ? (lane == k ? sd : (lane > k ? lp : 0.0)) : lp;
		a[k] = l;
		if (k + 1 < NB)
		{
			a[k + 1] = fma(-l, cb[k + 1], a[k + 1]);
			const double d = readlane_f64(a[k + 1], k + 1); // the next pivot in wave 0; a harmless number elsewhere
			const double r0 = rsqrt_newton(d);
			sd = d * r0;
			sd = fma(fma(-sd, sd, d), 0.5 * r0, sd);
			*((k & 1) ? my_slot0 : my_slot1) = lane == k + 1 ? sd : (lane > k + 1 ? a[k + 1] * r0 : 0.0);
			*((k & 1) ? my_r0 : my_r1) = r0;
		}
#pragma unroll
		for (int j = k + 2; j < NB; ++j) a[j] = fma(-l, cb[j], a[j]);
	}
	STAMP(2)
	if (valid && (!diag || blockIdx.x == 0))
	{
		double* __restrict__ dst = A + row;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
	STAMP(3)
}

// V6: V3 with hand-pipelined broadcast reads.  hipcc keeps only ~3 ds_read_b128 in flight in the rank-1 update, which
// exposes the LDS latency 16 times per column; here the reads are issued from inline asm in chunks of 8 (16 columns), two
// chunks in flight, with explicit s_waitcnt.
typedef double v2f64 __attribute__((ext_vector_type(2)));
struct Chunk
{
	v2f64 c[8];
};
template <int OFF>
__device__ __forceinline__ void lds_read8(Chunk& q, unsigned addr)
{
	asm volatile("ds_read_b128 %0, %8 offset:%9\n"
				 "ds_read_b128 %1, %8 offset:%9+16\n"
				 "ds_read_b128 %2, %8 offset:%9+32\n"
				 "ds_read_b128 %3, %8 offset:%9+48\n"
				 "ds_read_b128 %4, %8 offset:%9+64\n"
				 "ds_read_b128 %5, %8 offset:%9+80\n"
				 "ds_read_b128 %6, %8 offset:%9+96\n"
				 "ds_read_b128 %7, %8 offset:%9+112\n"
				 : "=&v"(q.c[0]), "=&v"(q.c[1]), "=&v"(q.c[2]), "=&v"(q.c[3]), "=&v"(q.c[4]), "=&v"(q.c[5]), "=&v"(q.c[6]), "=&v"(q.c[7])
				 : "v"(addr), "n"(OFF));
}
// the same request tied to a value the preceding FMAs produced: without a data dependency the instruction selector hoists
// the request above those FMAs, the old chunk stays live and the new one lands in fresh registers (spills)
template <int OFF>
__device__ __forceinline__ void lds_read8_after(Chunk& q, unsigned addr, double dep)
{
	asm volatile("ds_read_b128 %0, %8 offset:%9\n"
				 "ds_read_b128 %1, %8 offset:%9+16\n"
				 "ds_read_b128 %2, %8 offset:%9+32\n"
				 "ds_read_b128 %3, %8 offset:%9+48\n"
				 "ds_read_b128 %4, %8 offset:%9+64\n"
				 "ds_read_b128 %5, %8 offset:%9+80\n"
				 "ds_read_b128 %6, %8 offset:%9+96\n"
				 "ds_read_b128 %7, %8 offset:%9+112\n"
				 : "=&v"(q.c[0]), "=&v"(q.c[1]), "=&v"(q.c[2]), "=&v"(q.c[3]), "=&v"(q.c[4]), "=&v"(q.c[5]), "=&v"(q.c[6]), "=&v"(q.c[7])
				 : "v"(addr), "n"(OFF), "v"(dep));
}
template <int N>
__device__ __forceinline__ void lds_wait(Chunk& q)
{
	asm volatile("s_waitcnt lgkmcnt(%8)"
				 : "+v"(q.c[0]), "+v"(q.c[1]), "+v"(q.c[2]), "+v"(q.c[3]), "+v"(q.c[4]), "+v"(q.c[5]), "+v"(q.c[6]), "+v"(q.c[7])
				 : "n"(N));
}
template <int K, int C>
__device__ __forceinline__ void apply_chunk(double (&a)[NB], const Chunk& q, double l)
{
#pragma unroll
	for (int i = 0; i < 8; ++i)
	{
		constexpr int dummy = 0;
		const int j = 16 * C + 2 * i;
		if (j > K) a[j] = fma(-l, q.c[i].x, a[j]);
		if (j + 1 > K) a[j + 1] = fma(-l, q.c[i].y, a[j + 1]);
		(void)dummy;
	}
}
// rank-1 update of columns K+1..63 with the broadcast column at LDS byte address addr
template <int K>
__device__ __forceinline__ void update_row(double (&a)[NB], unsigned addr, double l)
{
	constexpr int CF = (K + 1) / 16; // first chunk holding a column > K
	if constexpr (K + 1 < NB)
	{
		Chunk q0, q1;
		lds_read8<CF * 128>(q0, addr);
		if constexpr (CF + 1 < 4) lds_read8<(CF + 1) * 128>(q1, addr);
		// chunk CF
		lds_wait<(CF + 1 < 4) ? 8 : 0>(q0);
		apply_chunk<K, CF>(a, q0, l);
		if constexpr (CF + 2 < 4) lds_read8<(CF + 2) * 128>(q0, addr);
		if constexpr (CF + 1 < 4)
		{
			lds_wait<(CF + 2 < 4) ? 8 : 0>(q1);
			apply_chunk<K, CF + 1>(a, q1, l);
			if constexpr (CF + 3 < 4) lds_read8<(CF + 3) * 128>(q1, addr);
		}
		if constexpr (CF + 2 < 4)
		{
			lds_wait<(CF + 3 < 4) ? 8 : 0>(q0);
			apply_chunk<K, CF + 2>(a, q0, l);
		}
		if constexpr (CF + 3 < 4)
		{
			lds_wait<0>(q1);
			apply_chunk<K, CF + 3>(a, q1, l);
		}
	}
}

template <int K, int NT>
__device__ __forceinline__ void column_v6(double (&a)[NB], double (*col)[NB + 2], double* dump, int lane, bool diag)
{
	double* const cb = &col[K & 1][0];
	double* const my_col = diag ? &cb[lane] : &dump[threadIdx.x];
	double* const my_r = (diag && lane == 0) ? &cb[NB] : &dump[threadIdx.x];
	const double d = readlane_f64(a[K], K); // the pivot in wave 0; a harmless number elsewhere
	const double r0 = rsqrt_newton(d);
	double sd = d * r0;
	sd = fma(fma(-sd, sd, d), 0.5 * r0, sd);
	*my_col = lane == K ? sd : (lane > K ? a[K] * r0 : 0.0);
	*my_r = r0;
	__syncthreads();
	const double r = cb[NB];
	const double lp = a[K] * r;
	const double l = diag ? (lane == K ? sd : (lane > K ? lp : 0.0)) : lp;
	a[K] = l;
	const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) double*)cb));
	update_row<K>(a, addr, l);
}
template <int NT, int... Ks>
__device__ __forceinline__ void all_columns_v6(double (&a)[NB], double (*col)[NB + 2], double* dump, int lane, bool diag, std::integer_sequence<int, Ks...>)
{
	(column_v6<Ks, NT>(a, col, dump, lane, diag), ...);
}

template <bool ST, int NT>
__global__ void __launch_bounds__(NT) panel_v6(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double col[2][NB + 2];
	__shared__ double dump[NT];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool diag = w == 0;
	const int row = diag ? lane : NB + blockIdx.x * (NT - 64) + (w - 1) * 64 + lane;
	const bool valid = row < m;
	double a[NB];
	const double* __restrict__ src = A + (valid ? row : 0);
	STAMP(0)
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	STAMP(1)
	all_columns_v6<NT>(a, col, dump, lane, diag, std::make_integer_sequence<int, NB>{});
	STAMP(2)
	if (valid && (!diag || blockIdx.x == 0))
	{
		double* __restrict__ dst = A + row;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
	STAMP(3)
}

// V8: V6 + the pivot chain of column K+1 started right after a one-element fast path, so that it runs while the chunk
// reads of column K are in flight.
template <int KMIN, int C>
__device__ __forceinline__ void apply_chunk_from(double (&a)[NB], const Chunk& q, double l)
{
#pragma unroll
	for (int i = 0; i < 8; ++i)
	{
		const int j = 16 * C + 2 * i;
		if (j >= KMIN) a[j] = fma(-l, q.c[i].x, a[j]);
		if (j + 1 >= KMIN) a[j + 1] = fma(-l, q.c[i].y, a[j + 1]);
	}
}
template <int K, int NT>
__device__ __forceinline__ void column_v8(double (&a)[NB], double (*col)[NB + 2], double* dump, int lane, bool diag, double& sd)
{
	// on entry: column K (scaled, with 1/L_KK) has been written to col[K & 1] by wave 0; sd = L_KK
	double* const cb = &col[K & 1][0];
	double* const nb = &col[(K + 1) & 1][0];
	__syncthreads();
	const double r = cb[NB];
	const double lp = a[K] * r;
	const double l = diag ? (lane == K ? sd : (lane > K ? lp : 0.0)) : lp;
	a[K] = l;
	if constexpr (K + 1 < NB)
	{
		const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) double*)cb));
		constexpr int KMIN = K + 2;          // first column the chunked update still owes
		constexpr int CF = KMIN / 16;        // its chunk
		a[K + 1] = fma(-l, cb[K + 1], a[K + 1]); // fast path: everything the next pivot needs
		Chunk q0, q1;
		if constexpr (KMIN < NB) lds_read8<CF * 128>(q0, addr);
		if constexpr (KMIN < NB && CF + 1 < 4) lds_read8<(CF + 1) * 128>(q1, addr);
		{
			double* const my_col = diag ? &nb[lane] : &dump[threadIdx.x];
			double* const my_r = (diag && lane == 0) ? &nb[NB] : &dump[threadIdx.x];
			const double d = readlane_f64(a[K + 1], K + 1); // the next pivot in wave 0; a harmless number elsewhere
			const double r0 = rsqrt_newton(d);
			sd = d * r0;
			sd = fma(fma(-sd, sd, d), 0.5 * r0, sd);
			*my_col = lane == K + 1 ? sd : (lane > K + 1 ? a[K + 1] * r0 : 0.0);
			*my_r = r0;
		}
		if constexpr (KMIN < NB)
		{
			lds_wait<(CF + 1 < 4) ? 8 : 0>(q0);
			apply_chunk_from<KMIN, CF>(a, q0, l);
			if constexpr (CF + 2 < 4) lds_read8<(CF + 2) * 128>(q0, addr);
			if constexpr (CF + 1 < 4)
			{
				lds_wait<(CF + 2 < 4) ? 8 : 0>(q1);
				apply_chunk_from<KMIN, CF + 1>(a, q1, l);
				if constexpr (CF + 3 < 4) lds_read8<(CF + 3) * 128>(q1, addr);
			}
			if constexpr (CF + 2 < 4)
			{
				lds_wait<(CF + 3 < 4) ? 8 : 0>(q0);
				apply_chunk_from<KMIN, CF + 2>(a, q0, l);
			}
			if constexpr (CF + 3 < 4)
			{
				lds_wait<0>(q1);
				apply_chunk_from<KMIN, CF + 3>(a, q1, l);
			}
		}
	}
}
template <int NT, int... Ks>
__device__ __forceinline__ void all_columns_v8(double (&a)[NB], double (*col)[NB + 2], double* dump, int lane, bool diag, double& sd, std::integer_sequence<int, Ks...>)
{
	(column_v8<Ks, NT>(a, col, dump, lane, diag, sd), ...);
}

template <bool ST, int NT>
__global__ void __launch_bounds__(NT) panel_v8(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double col[2][NB + 2];
	__shared__ double dump[NT];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const bool diag = w == 0;
	const int row = diag ? lane : NB + blockIdx.x * (NT - 64) + (w - 1) * 64 + lane;
	const bool valid = row < m;
	double a[NB];
	const double* __restrict__ src = A + (valid ? row : 0);
	STAMP(0)
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	STAMP(1)
	double sd;
	{
		double* const my_col = diag ? &col[0][lane] : &dump[threadIdx.x];
		double* const my_r = (diag && lane == 0) ? &col[0][NB] : &dump[threadIdx.x];
		const double d = readlane_f64(a[0], 0);
		const double r0 = rsqrt_newton(d);
		sd = d * r0;
		sd = fma(fma(-sd, sd, d), 0.5 * r0, sd);
		*my_col = lane == 0 ? sd : a[0] * r0;
		*my_r = r0;
	}
	all_columns_v8<NT>(a, col, dump, lane, diag, sd, std::make_integer_sequence<int, NB>{});
	STAMP(2)
	if (valid && (!diag || blockIdx.x == 0))
	{
		double* __restrict__ dst = A + row;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
	STAMP(3)
}

// V9: wave-specialised, one barrier per BLK columns.  Wave 0 (diagonal block) never waits for LDS on its pivot chain: the
// next pivot's dependency goes through readlane (fast path), the rest of each rank-1 update is applied one column late from
// its own published column (same-wave LDS write -> read needs no barrier).  The row wave consumes a block of BLK published
// columns per barrier.  Blocks are double-buffered.
constexpr int BLK = 4;
// chunked a[j] -= l * column[j] for j >= KMIN; on entry the first one (two) chunk reads may already be in flight
template <int KMIN, bool PREISSUED>
__device__ __forceinline__ void update_from(double (&a)[NB], unsigned addr, double l, Chunk& q0, Chunk& q1)
{
	if constexpr (KMIN < NB)
	{
		constexpr int CF = KMIN / 16;
		if constexpr (!PREISSUED)
		{
			lds_read8<CF * 128>(q0, addr);
			if constexpr (CF + 1 < 4) lds_read8<(CF + 1) * 128>(q1, addr);
		}
		lds_wait<(CF + 1 < 4) ? 8 : 0>(q0);
		apply_chunk_from<KMIN, CF>(a, q0, l);
		if constexpr (CF + 2 < 4) lds_read8_after<(CF + 2) * 128>(q0, addr, a[16 * CF + 15]);
		if constexpr (CF + 1 < 4)
		{
			lds_wait<(CF + 2 < 4) ? 8 : 0>(q1);
			apply_chunk_from<KMIN, CF + 1>(a, q1, l);
			if constexpr (CF + 3 < 4) lds_read8_after<(CF + 3) * 128>(q1, addr, a[16 * (CF + 1) + 15]);
		}
		if constexpr (CF + 2 < 4)
		{
			lds_wait<(CF + 3 < 4) ? 8 : 0>(q0);
			apply_chunk_from<KMIN, CF + 2>(a, q0, l);
		}
		if constexpr (CF + 3 < 4)
		{
			lds_wait<0>(q1);
			apply_chunk_from<KMIN, CF + 3>(a, q1, l);
		}
	}
}
template <int KMIN>
__device__ __forceinline__ void issue_first(unsigned addr, Chunk& q0, Chunk& q1)
{
	if constexpr (KMIN < NB)
	{
		constexpr int CF = KMIN / 16;
		lds_read8<CF * 128>(q0, addr);
		if constexpr (CF + 1 < 4) lds_read8<(CF + 1) * 128>(q1, addr);
	}
}
__device__ __forceinline__ unsigned lds_addr(const double* p)
{
	return static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const double*)p));
}

// column C in wave 0.  lprev = this lane's entry of column C - 1, whose deferred update (entries >= C + 1) is applied here;
// q0/q1 hold its first chunks, requested one column ago.  The pivot chain of column C and the FMAs of that update are
// independent instruction streams between the two waits, so the scheduler can fill the chain's latency with FMAs.
template <int C>
__device__ __forceinline__ void diag_column(double (&a)[NB], double (*colb)[BLK][NB + 2], int lane, double& lprev, Chunk& q0, Chunk& q1)
{
	double* const cb = &colb[(C / BLK) & 1][C % BLK][0];
	constexpr int KMIN = C + 1;
	constexpr bool DEF = C >= 1 && KMIN < NB; // a deferred update exists
	constexpr int CF = KMIN / 16;
	if constexpr (DEF) lds_wait<(CF + 1 < 4) ? 8 : 0>(q0);
	const double d = readlane_f64(a[C], C);
	const double r = rsqrt_newton(d);
	double sd = d * r;
	sd = fma(fma(-sd, sd, d), 0.5 * r, sd);
	const double l = lane > C ? a[C] * r : 0.0;
	a[C] = lane == C ? sd : l;
	if constexpr (DEF) apply_chunk_from<KMIN, CF>(a, q0, lprev);
	if constexpr (C + 1 < NB) a[C + 1] = fma(-l, readlane_f64(l, C + 1), a[C + 1]); // fast path: all the next pivot needs
	if constexpr (DEF)
	{
		const unsigned addr = lds_addr(&colb[((C - 1) / BLK) & 1][(C - 1) % BLK][0]);
		if constexpr (CF + 2 < 4) lds_read8_after<(CF + 2) * 128>(q0, addr, a[16 * CF + 15]);
		if constexpr (CF + 1 < 4)
		{
			lds_wait<(CF + 2 < 4) ? 8 : 0>(q1);
			apply_chunk_from<KMIN, CF + 1>(a, q1, lprev);
			if constexpr (CF + 3 < 4) lds_read8_after<(CF + 3) * 128>(q1, addr, a[16 * (CF + 1) + 15]);
		}
		if constexpr (CF + 2 < 4)
		{
			lds_wait<(CF + 3 < 4) ? 8 : 0>(q0);
			apply_chunk_from<KMIN, CF + 2>(a, q0, lprev);
		}
		if constexpr (CF + 3 < 4)
		{
			lds_wait<0>(q1);
			apply_chunk_from<KMIN, CF + 3>(a, q1, lprev);
		}
	}
	cb[lane] = lane == C ? r : l; // slot C carries 1/L_CC for the row waves; nobody reads L_CC from here
	if constexpr (C % BLK == BLK - 1 || C == NB - 1) __syncthreads(); // block published
	if constexpr (C + 2 < NB)
	{
		constexpr int CN = (C + 2) / 16;
		lds_read8_after<CN * 128>(q0, lds_addr(cb), a[NB - 1]); // consumed while the next column's chain runs
		if constexpr (CN + 1 < 4) lds_read8_after<(CN + 1) * 128>(q1, lds_addr(cb), a[NB - 17]);
	}
	lprev = l;
}
template <int... Cs>
__device__ __forceinline__ void diag_columns(double (&a)[NB], double (*colb)[BLK][NB + 2], int lane, std::integer_sequence<int, Cs...>)
{
	double lprev = 0.0;
	Chunk q0, q1;
	(diag_column<Cs>(a, colb, lane, lprev, q0, q1), ...);
}
template <int C>
__device__ __forceinline__ void row_column(double (&a)[NB], double (*colb)[BLK][NB + 2])
{
	if constexpr (C % BLK == 0) __syncthreads(); // the block holding column C has been published
	const double* const cb = &colb[(C / BLK) & 1][C % BLK][0];
	const unsigned addr = lds_addr(cb);
	constexpr int CF = (C + 1) / 16;
	constexpr bool ANY = C + 1 < NB;
	double r;
	// tied to the last entries the previous column updated: keeps that column's FMAs above this column's requests
	asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(r) : "v"(addr), "n"(C * 8), "v"(a[NB - 1]), "v"(a[NB - 17]));
	Chunk q0, q1;
	if constexpr (ANY) lds_read8<CF * 128>(q0, addr);
	asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(ANY ? 8 : 0));
	const double l = a[C] * r;
	a[C] = l;
	if constexpr (ANY)
	{
		if constexpr (CF + 1 < 4) lds_read8<(CF + 1) * 128>(q1, addr);
		lds_wait<(CF + 1 < 4) ? 8 : 0>(q0);
		apply_chunk_from<C + 1, CF>(a, q0, l);
		if constexpr (CF + 2 < 4) lds_read8_after<(CF + 2) * 128>(q0, addr, a[16 * CF + 15]);
		if constexpr (CF + 1 < 4)
		{
			lds_wait<(CF + 2 < 4) ? 8 : 0>(q1);
			apply_chunk_from<C + 1, CF + 1>(a, q1, l);
			if constexpr (CF + 3 < 4) lds_read8_after<(CF + 3) * 128>(q1, addr, a[16 * (CF + 1) + 15]);
		}
		if constexpr (CF + 2 < 4)
		{
			lds_wait<(CF + 3 < 4) ? 8 : 0>(q0);
			apply_chunk_from<C + 1, CF + 2>(a, q0, l);
		}
		if constexpr (CF + 3 < 4)
		{
			lds_wait<0>(q1);
			apply_chunk_from<C + 1, CF + 3>(a, q1, l);
		}
	}
}
template <int... Cs>
__device__ __forceinline__ void row_columns(double (&a)[NB], double (*colb)[BLK][NB + 2], std::integer_sequence<int, Cs...>)
{
	(row_column<Cs>(a, colb), ...);
}
__device__ __forceinline__ void diag_wave(double* __restrict__ A, long lda, double (*colb)[BLK][NB + 2], int lane, bool store)
{
	double a[NB];
	const double* __restrict__ src = A + lane;
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	diag_columns(a, colb, lane, std::make_integer_sequence<int, NB>{});
	if (store)
	{
		int lane2 = lane;
		asm volatile("" : "+v"(lane2)); // a fresh value: keeps hipcc from holding the 64 load addresses live for the stores
		double* __restrict__ dst = A + lane2;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
}
__device__ __forceinline__ void row_wave(double* __restrict__ A, long lda, double (*colb)[BLK][NB + 2], int row, bool valid)
{
	double a[NB];
	const double* __restrict__ src = A + row;
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
	row_columns(a, colb, std::make_integer_sequence<int, NB>{});
	if (valid)
	{
		int row2 = row;
		asm volatile("" : "+v"(row2));
		double* __restrict__ dst = A + row2;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
}
template <bool ST>
__global__ void __launch_bounds__(128) panel_v9(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double colb[2][BLK][NB + 2];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int row = NB + blockIdx.x * 64 + lane;
	const bool valid = row < m;
	STAMP(0)
	if (w == 0) diag_wave(A, lda, colb, lane, blockIdx.x == 0);
	else row_wave(A, lda, colb, valid ? row : lane, valid);
	STAMP(2)
}

// V1: barrier-free, two phases, each in its own non-inlined function so that their 64-entry rows are never live together.
// MODE 0: phase-1 broadcasts through LDS; MODE 1: through readlane
template <int MODE>
__device__ __noinline__ void phase1(double* __restrict__ A, long lda, double* Lw, int lane, bool store)
{
	double a[NB];
	const double* __restrict__ src = A + lane;
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
#pragma unroll
	for (int k = 0; k < NB; ++k)
	{
		double* const cw = Lw + k * LR;
		const double d = readlane_f64(a[k], k);
		const double r = rsqrt_newton(d);
		double sd = d * r;
		sd = fma(fma(-sd, sd, d), 0.5 * r, sd);
		const double lp = a[k] * r;
		const double l = lane > k ? lp : 0.0;
		a[k] = lane == k ? sd : l;
		cw[lane] = lane == k ? r : l;
		if (MODE == 0)
		{
			if (k + 1 < NB) a[k + 1] = fma(-l, readlane_f64(l, k + 1), a[k + 1]);
#pragma unroll
			for (int j = k + 2; j < NB; ++j) a[j] = fma(-l, cw[j], a[j]);
		}
		else
		{
#pragma unroll
			for (int j = k + 1; j < NB; ++j) a[j] = fma(-l, readlane_f64(l, j), a[j]);
		}
	}
	if (store)
	{
		double* __restrict__ dst = A + lane;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
}

__device__ __noinline__ void phase2(double* __restrict__ A, long lda, const double* Lw, int row, bool valid)
{
	double a[NB];
	const double* __restrict__ src = A + row;
#pragma unroll
	for (int j = 0; j < NB; ++j) a[j] = src[static_cast<long>(j) * lda];
#pragma unroll
	for (int k = 0; k < NB; ++k)
	{
		const double* const cw = Lw + k * LR;
		const double l = a[k] * cw[k];
		a[k] = l;
#pragma unroll
		for (int j = k + 1; j < NB; ++j) a[j] = fma(-l, cw[j], a[j]);
	}
	if (valid)
	{
		double* __restrict__ dst = A + row;
#pragma unroll
		for (int j = 0; j < NB; ++j) dst[static_cast<long>(j) * lda] = a[j];
	}
}

template <int MODE, bool ST>
__global__ void __launch_bounds__(THREADS) panel_v1(double* __restrict__ A, long lda, int m, long long* stamps)
{
	__shared__ __attribute__((aligned(16))) double Ls[(THREADS / 64) * NB * LR];
	const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int row = NB + (blockIdx.x * (THREADS / 64) + w) * 64 + lane;
	const bool valid = row < m;
	double* const Lw = &Ls[w * (NB * LR)];
	STAMP(0)
	phase1<MODE>(A, lda, Lw, lane, blockIdx.x == 0 && w == 0);
	STAMP(2)
	phase2(A, lda, Lw, valid ? row : lane, valid);
	STAMP(6)
}


#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
	const int m = 1024;
	const long lda = m;
	std::vector<double> h(static_cast<size_t>(m) * NB);
	for (int j = 0; j < NB; ++j)
		for (int i = 0; i < m; ++i) h[i + j * lda] = (i == j ? 70.0 : 0.0) + std::exp(-0.001 * (i - j) * (i - j)) + 0.01 * std::cos(i * 0.37 + j * 1.3) * (i >= NB);
	double* d;
	long long* st;
	CK(hipMalloc(&d, h.size() * 8));
	CK(hipMalloc(&st, 8 * 16 * THREADS * 8));
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	std::vector<double> ref;
	for (int variant = 0; variant < 10; ++variant)
	{
		const int rows_per_wg = (variant == 5 || variant == 6 || variant == 8 || variant == 9) ? 64 : (variant == 0 || variant >= 3) ? 192 : 256;
		const int nthreads = (variant == 5 || variant == 6 || variant == 8 || variant == 9) ? 128 : THREADS;
		const int nwg = (m - NB + rows_per_wg - 1) / rows_per_wg;
		auto launch = [&](long long* s) {
			if (variant == 0 && s) hipLaunchKernelGGL((panel_v0<true>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 1 && s) hipLaunchKernelGGL((panel_v1<0, true>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 2 && s) hipLaunchKernelGGL((panel_v1<1, true>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 3 && s) hipLaunchKernelGGL((panel_v3<true, 256>), dim3(nwg), dim3(256), 0, 0, d, lda, m, s);
			if (variant == 3 && !s) hipLaunchKernelGGL((panel_v3<false, 256>), dim3(nwg), dim3(256), 0, 0, d, lda, m, s);
			if (variant == 6 && s) hipLaunchKernelGGL((panel_v6<true, 128>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 6 && !s) hipLaunchKernelGGL((panel_v6<false, 128>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 7 && s) hipLaunchKernelGGL((panel_v6<true, 256>), dim3(nwg), dim3(256), 0, 0, d, lda, m, s);
			if (variant == 7 && !s) hipLaunchKernelGGL((panel_v6<false, 256>), dim3(nwg), dim3(256), 0, 0, d, lda, m, s);
			if (variant == 8 && s) hipLaunchKernelGGL((panel_v8<true, 128>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 8 && !s) hipLaunchKernelGGL((panel_v8<false, 128>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 9 && s) hipLaunchKernelGGL((panel_v9<true>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 9 && !s) hipLaunchKernelGGL((panel_v9<false>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 5 && s) hipLaunchKernelGGL((panel_v3<true, 128>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 5 && !s) hipLaunchKernelGGL((panel_v3<false, 128>), dim3(nwg), dim3(128), 0, 0, d, lda, m, s);
			if (variant == 4 && s) hipLaunchKernelGGL((panel_v4<true>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 4 && !s) hipLaunchKernelGGL((panel_v4<false>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 0 && !s) hipLaunchKernelGGL((panel_v0<false>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 1 && !s) hipLaunchKernelGGL((panel_v1<0, false>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
			if (variant == 2 && !s) hipLaunchKernelGGL((panel_v1<1, false>), dim3(nwg), dim3(THREADS), 0, 0, d, lda, m, s);
		};
		CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
		CK(hipMemset(st, 0, 8 * 16 * THREADS * 8));
		launch(st);
		CK(hipDeviceSynchronize());
		std::vector<double> out(h.size());
		CK(hipMemcpy(out.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
		std::vector<long long> all(8 * 16 * THREADS);
		CK(hipMemcpy(all.data(), st, all.size() * 8, hipMemcpyDeviceToHost));
		long long hs[8];
		for (int i = 0; i < 8; ++i) hs[i] = all[static_cast<size_t>(i) * nwg * nthreads + ((variant == 0 || variant >= 3) ? 0 : 64)]; // block 0: wave 0 (v0, v3) / wave 1
		double maxdiff = 0;
		if (variant == 0) ref = out;
		else
			for (size_t i = 0; i < out.size(); ++i) maxdiff = std::fmax(maxdiff, std::fabs(out[i] - ref[i]));
		// timing without stamps (the factor of an already factored panel is still the same instruction stream)
		for (int i = 0; i < 5; ++i) launch(nullptr);
		hipEventRecord(e0, 0);
		for (int i = 0; i < 50; ++i) launch(nullptr);
		hipEventRecord(e1, 0);
		CK(hipEventSynchronize(e1));
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("variant %d: %.2f us/launch, maxdiff vs v0 %.3e, stamps(cycles from start):", variant, ms * 1000 / 50, maxdiff);
		for (int i = 1; i < 7; ++i) printf(" %lld", hs[i] ? hs[i] - hs[0] : 0LL);
		if (variant == 9) printf("   v9 end stamps: wave0 %lld  wave1 %lld", all[2 * nwg * nthreads + 0] - all[0], all[2 * nwg * nthreads + 64] - all[64]);
		printf("\n");
	}
	return 0;
}
