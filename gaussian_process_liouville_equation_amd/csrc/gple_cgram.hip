// gple_cgram.hip — ComplexKernelBase (complex_kernel.h:14-145, complex_kernel.cpp:20-200): the covariance K (real), the
// pseudo-covariance K~ (complex) and their 8 + 8 parameter-derivative matrices between two point sets, materialised.
//
// The reference builds three KernelBase objects (real-part, imaginary-part and correlation sub-kernels, noise 0) and combines
// their Gram / derivative matrices with Eigen expressions.  Here one thread produces all of an entry's outputs from three
// bit-faithful squared exponentials; the combinations follow the reference's tables term by term, including what they omit
// (no s^2 on the sub-kernel and noise derivatives of K: complex_kernel.cpp:39-51).
#include "gple_kernels.h"

namespace gple
{
	namespace
	{
		// exp(-(((a0-b0)/l0)^2 + ((a1-b1)/l1)^2)/2) with the reference's operation order, no contraction (kernel.cpp:46-47)
		__device__ __forceinline__ double se_ref(double a0, double a1, double b0, double b1, double l0, double l1, double* d0sq, double* d1sq)
		{
			const double d0 = __ddiv_rn(__dsub_rn(a0, b0), l0);
			const double d1 = __ddiv_rn(__dsub_rn(a1, b1), l1);
			*d0sq = __dmul_rn(d0, d0), *d1sq = __dmul_rn(d1, d1);
			return exp(__ddiv_rn(-__dadd_rn(*d0sq, *d1sq), 2.0));
		}
		// one noise-free KernelBase entry and its magnitude / length derivatives (kernel.cpp:168-215 with noise = 0)
		struct SubEntry
		{
			double k, ds, dl0, dl1;
		};
		__device__ __forceinline__ SubEntry sub_entry(double a0, double a1, double b0, double b1, double s, double l0, double l1, bool zero_diag)
		{
			double q0, q1;
			const double g = se_ref(a0, a1, b0, b1, l0, l1, &q0, &q1);
			SubEntry e;
			e.k = __dmul_rn(__dmul_rn(s, s), g);
			e.ds = __dmul_rn(e.k, 2.0 / s);                                // :181
			e.dl0 = zero_diag ? 0.0 : __dmul_rn(e.k, __ddiv_rn(q0, l0)); // :109,131, zero diagonal on the training branch
			e.dl1 = zero_diag ? 0.0 : __dmul_rn(e.k, __ddiv_rn(q1, l1));
			return e;
		}

		struct CGramParam
		{
			double s, sn;           // global magnitude, noise
			double sR, lR[2], sI, lI[2];
			double sC, lC[2];       // correlation kernel, complex_kernel.cpp:144-157
		};

		__global__ void __launch_bounds__(256) complex_gram_kernel(const double* __restrict__ L, int R, const double* __restrict__ Rt, int C, int same,
			CGramParam p, double* __restrict__ K, double* __restrict__ Kt, double* __restrict__ dK, double* __restrict__ dKt)
		{
			const int i = blockIdx.x * 64 + (threadIdx.x & 63);
			if (i >= R) return;
			const double a0 = L[2 * i], a1 = L[2 * i + 1];
			const long sz = static_cast<long>(R) * C;
			const double m2 = p.s * p.s, n2 = p.sn * p.sn;
			for (int e = 0; e < 4; ++e)
			{
				const int j = blockIdx.y * 16 + (threadIdx.x >> 6) * 4 + e;
				if (j >= C) continue;
				const double b0 = Rt[2 * j], b1 = Rt[2 * j + 1];
				const double delta = same ? (i == j ? 1.0 : 0.0) : ((a0 == b0 && a1 == b1) ? 1.0 : 0.0); // kernel.cpp:8-31
				const bool zd = same && i == j;
				const SubEntry r = sub_entry(a0, a1, b0, b1, p.sR, p.lR[0], p.lR[1], zd);
				const SubEntry im = sub_entry(a0, a1, b0, b1, p.sI, p.lI[0], p.lI[1], zd);
				const SubEntry c = sub_entry(a0, a1, b0, b1, p.sC, p.lC[0], p.lC[1], zd);
				const long idx = i + static_cast<long>(j) * R;
				const double k = m2 * ((r.k + im.k) + n2 * delta);       // complex_kernel.cpp:163
				const double ktr = m2 * (r.k - im.k), kti = m2 * (2.0 * c.k); // :164
				K[idx] = k;
				if (Kt) Kt[2 * idx] = ktr, Kt[2 * idx + 1] = kti;
				if (dK)
				{
					dK[idx] = 2.0 / p.s * k;                                // :36
					dK[1 * sz + idx] = r.ds, dK[2 * sz + idx] = r.dl0, dK[3 * sz + idx] = r.dl1;    // :39-43 (no s^2: sic)
					dK[4 * sz + idx] = im.ds, dK[5 * sz + idx] = im.dl0, dK[6 * sz + idx] = im.dl1; // :45-49
					dK[7 * sz + idx] = same ? 2.0 * p.sn * delta : 0.0;     // :51-58
				}
				if (dKt)
				{
					auto put = [&](int ip, double re, double imv) { dKt[2 * (ip * sz + idx)] = re, dKt[2 * (ip * sz + idx) + 1] = imv; };
					put(0, 2.0 / p.s * ktr, 2.0 / p.s * kti);                                               // :94
					put(1, r.ds, 2.0 / p.sR * c.k);                                                         // :101
					put(4, -im.ds, 2.0 / p.sI * c.k);                                                       // :117
					const double cdl[2] = {c.dl0, c.dl1}, rdl[2] = {r.dl0, r.dl1}, idl[2] = {im.dl0, im.dl1};
#pragma unroll
					for (int d = 0; d < 2; ++d)
					{
						const double lr = p.lR[d], li = p.lI[d], lc = p.lC[d];
						put(2 + d, rdl[d], 2.0 * (1.0 / lr - lr / (lc * lc)) * c.k + lr / lc * cdl[d]);    // :104-109
						put(5 + d, -idl[d], 2.0 * (1.0 / li - li / (lc * lc)) * c.k + li / lc * cdl[d]);   // :120-125
					}
					put(7, 0.0, 0.0);                                                                       // :129
				}
			}
		}
	} // namespace

	hipError_t launch_complex_gram(hipStream_t s, const double theta[8], const double* L, int R, const double* Rt, int C, int same, double* K,
		double* Kt, double* dK, double* dKt)
	{
		if (R == 0 || C == 0) return hipSuccess;
		CGramParam p;
		p.s = theta[0], p.sn = theta[7];
		p.sR = theta[1], p.lR[0] = theta[2], p.lR[1] = theta[3];
		p.sI = theta[4], p.lI[0] = theta[5], p.lI[1] = theta[6];
		double prod = 1.0;
		for (int d = 0; d < 2; ++d)
		{
			const double ss = p.lR[d] * p.lR[d] + p.lI[d] * p.lI[d];
			prod *= 2.0 * p.lR[d] * p.lI[d] / ss;
			p.lC[d] = std::sqrt(ss / 2.0);
		}
		p.sC = std::sqrt(p.sR * p.sI * prod);
		hipLaunchKernelGGL(complex_gram_kernel, dim3((R + 63) / 64, (C + 15) / 16), dim3(256), 0, s, L, R, Rt, C, same, p, K, Kt, dK, dKt);
		return hipGetLastError();
	}
} // namespace gple
