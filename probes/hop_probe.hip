// Cost of handing a 64 x 64 fp64 tile from one workgroup to another INSIDE a running kernel (flag + payload through global memory), the
// building block of a factorisation whose panels are not separated by kernel launches.  Two workgroups play ping-pong; the others exit.
//   mode 0: plain stores, __threadfence() (agent-scope release: L2 write-back), flag; consumer polls, acquire fence (L2 invalidate), plain loads
//   mode 1: payload by agent-scope relaxed atomics (sc1: write-through / L2-bypassing loads), s_waitcnt, flag — no whole-cache maintenance
//   mode 2: flag only (no payload)
// Every wait is bounded: a poll loop gives up after 2^22 rounds and the kernel ends.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool wait_flag(const int* f, int want)
{
	for (int i = 0; i < (1 << 22); ++i)
	{
		if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
		__builtin_amdgcn_s_sleep(1);
	}
	return false;
}

template <int MODE>
__global__ void __launch_bounds__(256) pingpong(double* bufA, double* bufB, int* flagA, int* flagB, int iters, long long* out, int partner)
{
	const int b = blockIdx.x;
	if (b != 0 && b != partner) return;
	const bool first = b == 0;
	double* mine = first ? bufA : bufB;
	const double* theirs = first ? bufB : bufA;
	int* myflag = first ? flagA : flagB;
	const int* theirflag = first ? flagB : flagA;
	const int t = threadIdx.x;
	__shared__ int ok;
	double v[16];
	for (int q = 0; q < 16; ++q) v[q] = t + q;
	const long long t0 = wall_clock64();
	for (int it = 1; it <= iters; ++it)
	{
		if (!first || it > 1)
		{
			// receive
			if (t == 0) ok = wait_flag(theirflag, first ? it - 1 : it);
			__syncthreads();
			if (!ok) return;
			if (MODE == 0)
			{
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				for (int q = 0; q < 16; ++q) v[q] += theirs[t + 256 * q];
			}
			else if (MODE == 1)
				for (int q = 0; q < 16; ++q) v[q] += __hip_atomic_load(theirs + t + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		// send
		if (MODE == 0)
		{
			for (int q = 0; q < 16; ++q) mine[t + 256 * q] = v[q];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		}
		else if (MODE == 1)
		{
			for (int q = 0; q < 16; ++q) __hip_atomic_store(mine + t + 256 * q, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		__syncthreads();
		if (t == 0) __hip_atomic_store(myflag, it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (first)
	{
		if (t == 0) ok = wait_flag(theirflag, iters);
		__syncthreads();
		if (t == 0) out[0] = wall_clock64() - t0, out[1] = ok;
		if (t == 1) out[2] = static_cast<long long>(v[3]);
	}
}

// something that keeps every L2 busy and dirty beside the ping-pong: a streaming read-modify-write over a large buffer
__global__ void __launch_bounds__(256) churn(double* x, long n, int rounds)
{
	for (int r = 0; r < rounds; ++r)
		for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) x[i] = x[i] * 1.0000001 + 1.0;
}

int main()
{
	double *a, *b, *big;
	int* flags;
	long long* out;
	const long nbig = 1L << 26; // 512 MB
	hipMalloc(&a, 4096 * 8), hipMalloc(&b, 4096 * 8), hipMalloc(&flags, 64 * 4), hipMalloc(&out, 64), hipMalloc(&big, nbig * 8);
	hipMemset(big, 0, nbig * 8);
	hipStream_t s, s2;
	hipStreamCreateWithFlags(&s, hipStreamNonBlocking), hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	const int iters = 200;
	for (int busy = 0; busy < 2; ++busy)
		for (int mode = 0; mode < 3; ++mode)
			for (int partner : {1, 8, 3})
			{
				hipMemsetAsync(flags, 0, 64 * 4, s), hipMemsetAsync(out, 0, 64, s);
				hipStreamSynchronize(s);
				if (busy) hipLaunchKernelGGL(churn, dim3(1024), dim3(256), 0, s2, big, nbig, 4);
				if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(16), dim3(256), 0, s, a, b, flags, flags + 32, iters, out, partner);
				if (mode == 1) hipLaunchKernelGGL(pingpong<1>, dim3(16), dim3(256), 0, s, a, b, flags, flags + 32, iters, out, partner);
				if (mode == 2) hipLaunchKernelGGL(pingpong<2>, dim3(16), dim3(256), 0, s, a, b, flags, flags + 32, iters, out, partner);
				hipStreamSynchronize(s);
				hipStreamSynchronize(s2);
				long long h[3];
				hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
				printf("busy %d mode %d partner block %d: %.2f us per hop (ok %lld)\n", busy, mode, partner, h[0] * 0.01 / (2.0 * iters), h[1]);
			}
	return 0;
}
