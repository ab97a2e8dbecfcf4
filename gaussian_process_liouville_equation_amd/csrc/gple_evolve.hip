// gple_evolve.hip — the per-tick step loop around the GP, device resident:
//   Tully's models in the adiabatic representation            pes.cpp:8-189, pes.h:8-41
//   MQCLE point propagation with 3-branch back-propagation    evolve.cpp:28-423  (NumPES = 2, Dim = 1: what the reference instantiates)
//   Metropolis chains on the fitted distribution              mc.cpp:118-165, 287-327
// The reference evaluates its DistributionFunction once per phase-space point from inside these loops (8 one-point predicts per
// sample and tick, one per Metropolis step).  Here one kernel lays out ALL back-propagated points of a tick as three query
// lists (one per density-matrix element), the library predicts each list in one batch, and a second kernel combines the
// predictions into the new density — no host round trip inside a tick.  The Metropolis chains of all walkers advance together:
// per Monte-Carlo step one proposal kernel (counter-based Philox4x32-10 stream: the reference's clock-seeded, thread-shared
// mt19937 of mc.cpp:17 is not reproducible), one batched predict, one accept kernel.
#include "gple_kernels.h"

namespace gple
{
	namespace
	{
		constexpr double HBAR = 1.0; // stdafx.h:107

		__device__ __forceinline__ double sgn(double v) { return static_cast<double>((v > 0.0) - (v < 0.0)); } // pes.h:13-17

		struct Dia
		{
			double v00, v01, v11, f00, f01, f11;
		};
		// pes.cpp:25-69
		__device__ __forceinline__ Dia diabatic(double x, int model)
		{
			Dia d{0, 0, 0, 0, 0, 0};
			if (model == 0) // SAC
			{
				constexpr double A = 0.01, B = 1.6, C = 0.005, D = 1.0;
				const double e = exp(-sgn(x) * B * x);
				d.v00 = sgn(x) * A * (1.0 - e), d.v11 = -d.v00, d.v01 = C * exp(-D * x * x);
				d.f00 = -A * B * e, d.f11 = -d.f00, d.f01 = 2.0 * C * D * x * exp(-D * x * x);
			}
			else if (model == 1) // DAC
			{
				constexpr double A = 0.10, B = 0.28, C = 0.015, D = 0.06, E = 0.05;
				d.v11 = E - A * exp(-B * x * x), d.v01 = C * exp(-D * x * x);
				d.f11 = -2 * A * B * x * exp(-B * x * x), d.f01 = 2 * C * D * x * exp(-D * x * x);
			}
			else // ECR
			{
				constexpr double A = 6e-4, B = 0.10, C = 0.90;
				const double e = exp(-sgn(x) * C * x);
				d.v00 = A, d.v11 = -A, d.v01 = B * (1 - sgn(x) * (e - 1));
				d.f01 = -B * C * e;
			}
			return d;
		}
		struct Adia
		{
			double e0, e1;       // adiabatic_potential, pes.cpp:98-120
			double f00, f10, f11; // adiabatic_force (lower triangle mirrored), pes.cpp:122-135
			double nac01;        // adiabatic_coupling(0, 1) = -F(1,0) / (E1 - E0), pes.cpp:137-155
		};
		__device__ __forceinline__ Adia adiabatic(double x, int model)
		{
			const Dia d = diabatic(x, model);
			Adia a;
			const double diff = d.v00 - d.v11, root = sqrt(diff * diff + (2.0 * d.v01) * (2.0 * d.v01));
			a.e0 = (-root + (d.v00 + d.v11)) / 2.0, a.e1 = (root + (d.v00 + d.v11)) / 2.0;
			// pes.cpp:73-96: columns of the transformation, normalised
			const double root4 = sqrt(diff * diff + 4.0 * d.v01 * d.v01);
			double c00 = (-root4 + diff) / (2.0 * d.v01), c01 = (root4 + diff) / (2.0 * d.v01), c10 = 1.0, c11 = 1.0;
			const double n0 = sqrt(c00 * c00 + c10 * c10), n1 = sqrt(c01 * c01 + c11 * c11);
			c00 /= n0, c10 /= n0, c01 /= n1, c11 /= n1;
			const double m00 = d.f00 * c00 + d.f01 * c10, m01 = d.f00 * c01 + d.f01 * c11;
			const double m10 = d.f01 * c00 + d.f11 * c10, m11 = d.f01 * c01 + d.f11 * c11;
			a.f00 = c00 * m00 + c10 * m10, a.f10 = c01 * m00 + c11 * m10, a.f11 = c01 * m01 + c11 * m11;
			a.nac01 = -(a.f10 / (a.e1 - a.e0));
			return a;
		}
		// evolve.cpp:39-82 with CouplingCriterion = 0 and `>=`: true unless both expressions are NaN
		__device__ __forceinline__ bool is_coupling(double x, double p, double mass, double dt, int model)
		{
			const Adia a = adiabatic(x, model);
			return (fabs(a.nac01 * p / mass) * dt >= 0.0) || (fabs(a.f10 / ((a.f00 + a.f11) / 2.0)) >= 0.0);
		}
		__device__ __forceinline__ double diag_force(const Adia& a, int i) { return i == 0 ? a.f00 : a.f11; }
		// evolve.cpp:103-128
		__device__ __forceinline__ void adiabatic_evolve(double& x, double& p, double mass, double dt, double drc, int row, int col, int model)
		{
			x += drc * dt / 2.0 * (p / mass);
			const Adia a = adiabatic(x, model);
			p += drc * dt / 2.0 * (diag_force(a, row) + diag_force(a, col));
			x += drc * dt / 2.0 * (p / mass);
		}
		// evolve.cpp:137-151 for (RowIndex, ColIndex) = (0, 1)
		__device__ __forceinline__ double omega01(double xa, double xb, double drc, int model)
		{
			const Adia a = adiabatic(xa, model), b = adiabatic(xb, model);
			return drc * (a.e0 - a.e1 + b.e0 - b.e1) / 2.0 / HBAR;
		}
		struct Cplx
		{
			double re, im;
		};
		__device__ __forceinline__ Cplx cmul_phase(Cplx z, double angle)
		{
			const double c = cos(angle), s = sin(angle);
			return Cplx{z.re * c - z.im * s, z.re * s + z.im * c};
		}
		// evolve.cpp:219-235
		__device__ __forceinline__ void offdiagonal_rotation(Cplx (&rho)[3], double x, double p, double mass, double dt, int model)
		{
			const Adia a = adiabatic(x, model);
			const double phi = p / mass * a.nac01 * (is_coupling(x, p, mass, dt, model) ? 1.0 : 0.0);
			const double c = cos(2.0 * phi * dt), s = sin(2.0 * phi * dt);
			const Cplx r0 = rho[0], r1 = rho[1], r2 = rho[2];
			rho[0] = Cplx{(1.0 + c) / 2.0 * r0.re - s * r1.re + (1.0 - c) / 2.0 * r2.re, (1.0 + c) / 2.0 * r0.im + (1.0 - c) / 2.0 * r2.im};
			rho[1] = Cplx{s / 2.0 * r0.re + c * r1.re - s / 2.0 * r2.re, s / 2.0 * r0.im + r1.im - s / 2.0 * r2.im};
			rho[2] = Cplx{(1.0 - c) / 2.0 * r0.re + s * r1.re + (1.0 + c) / 2.0 * r2.re, (1.0 - c) / 2.0 * r0.im + (1.0 + c) / 2.0 * r2.im};
		}

		// the geometry of one back-propagation (evolve.cpp:236-284) from the point r = (x0, p0) of element (row, col)
		struct BackProp
		{
			double x2, p1, p2[3], x4[3][3], p3[3][3]; // [element it comes from: (0,0), (1,0), (1,1)][branch -1, 0, +1]
		};
		__device__ __forceinline__ BackProp back_propagate(double x0, double p0, double mass, double dt, int row, int col, int model)
		{
			constexpr double drc = -1.0;
			BackProp g;
			const double couple = is_coupling(x0, p0, mass, dt, model) ? 1.0 : 0.0;
			g.x2 = x0, g.p1 = p0;
			adiabatic_evolve(g.x2, g.p1, mass, dt / 2.0, drc, row, col, model);
			const double f01 = adiabatic(g.x2, model).f10 * couple;
#pragma unroll
			for (int b = 0; b < 3; ++b)
			{
				const double n = static_cast<double>(b - 1);
				g.p2[b] = g.p1 + dt * drc * n * f01;                        // :244-250
				const double x3 = g.x2 + drc * (dt / 4.0) * g.p2[b] / mass; // :251
				const Adia a = adiabatic(x3, model);
				const int ei[3] = {0, 1, 1}, ej[3] = {0, 0, 1};
#pragma unroll
				for (int e = 0; e < 3; ++e)
				{
					g.p3[e][b] = g.p2[b] + drc * (dt / 2.0) / 2.0 * (diag_force(a, ei[e]) + diag_force(a, ej[e])); // :253-281
					g.x4[e][b] = x3 + drc * (dt / 4.0) * g.p3[e][b] / mass;                                        // :283
				}
			}
			return g;
		}

		// Layout of the query lists.  Source element s (0: (0,0), 1: (1,0), 2: (1,1)) holds n[s] points starting at off[s] in the
		// concatenated point arrays.  Target element e receives from source s three rows per point (branches -1, 0, +1), except
		// from itself: the 0-branch is the exact density (evolve.cpp:309-313), so two rows per point — and the first of the two
		// carries the un-propagated point instead when the point is not coupled (the adiabatic branch, evolve.cpp:411-418).
		// new_points (new_point_predict, evolve.cpp:425-443): the points are NOT propagated, there is no exact density — the element
		// itself is predicted on its 0-branch like every other slot (three rows per point everywhere) — and an uncoupled point gets 0.
		struct EvolveLayout
		{
			int n[3], off[3];
			long qoff[3][3]; // row of the first query of source s inside target e's list: qoff[e][s]
			long qlen[3];
			int new_points;
		};
		__device__ __forceinline__ long query_row(const EvolveLayout& L, int e, int s, int idx, int b)
		{
			if (e == s && !L.new_points) return L.qoff[e][s] + 2L * idx + (b == 0 ? 0 : 1); // b in {0, 2}
			return L.qoff[e][s] + 3L * idx + b;
		}

		// One thread per sample point: forward propagation (two half steps), the nine back-propagated points into the query
		// lists, the new coordinates into r_new.
		__global__ void __launch_bounds__(256) evolve_prepare_kernel(const double* __restrict__ r, EvolveLayout L, double mass, double dt, int model,
			double* __restrict__ r_new, unsigned char* __restrict__ coupled, double* __restrict__ q0, double* __restrict__ q1, double* __restrict__ q2)
		{
			const int t = blockIdx.x * 256 + threadIdx.x;
			const int total = L.n[0] + L.n[1] + L.n[2];
			if (t >= total) return;
			const int s = t >= L.off[2] ? 2 : (t >= L.off[1] ? 1 : 0);
			const int idx = t - L.off[s];
			const int row = s == 0 ? 0 : 1, col = s == 2 ? 1 : 0;
			const double x0 = r[2 * t], p0 = r[2 * t + 1];
			const bool cpl = is_coupling(x0, p0, mass, dt, model);
			double xn = x0, pn = p0;
			if (!L.new_points)
			{
				if (cpl)
				{
					adiabatic_evolve(xn, pn, mass, dt / 2.0, 1.0, row, col, model);
					adiabatic_evolve(xn, pn, mass, dt / 2.0, 1.0, row, col, model);
				}
				else
					adiabatic_evolve(xn, pn, mass, dt, 1.0, row, col, model);
			}
			r_new[2 * t] = xn, r_new[2 * t + 1] = pn;
			coupled[t] = cpl ? 1 : 0;
			const BackProp g = back_propagate(xn, pn, mass, dt, row, col, model);
			double* const q[3] = {q0, q1, q2};
#pragma unroll
			for (int e = 0; e < 3; ++e)
#pragma unroll
				for (int b = 0; b < 3; ++b)
				{
					if (e == s && b == 1 && !L.new_points) continue; // the exact element on the 0-branch is not predicted
					const long qr = query_row(L, e, s, idx, b);
					const bool adiabatic_slot = !cpl && e == s && b == 0;
					q[e][2 * qr] = adiabatic_slot ? x0 : g.x4[e][b];
					q[e][2 * qr + 1] = adiabatic_slot ? p0 : g.p3[e][b];
				}
		}

		// One thread per sample point: the predicted densities at its back-propagated points -> the density at its new position
		// (evolve.cpp:286-372); pred[e]: interleaved (re, im) for the complex element e = 1, plain doubles for e = 0, 2, or
		// nullptr for an element without a kernel (prediction 0, main.cpp:86-88).
		__global__ void __launch_bounds__(256) evolve_combine_kernel(const double* __restrict__ r_old, const double* __restrict__ r_new,
			const double* __restrict__ rho_old, const unsigned char* __restrict__ coupled, EvolveLayout L, double mass, double dt, int model,
			const double* __restrict__ pred0, const double* __restrict__ pred1, const double* __restrict__ pred2, double* __restrict__ rho_new)
		{
			const int t = blockIdx.x * 256 + threadIdx.x;
			const int total = L.n[0] + L.n[1] + L.n[2];
			if (t >= total) return;
			const int s = t >= L.off[2] ? 2 : (t >= L.off[1] ? 1 : 0);
			const int idx = t - L.off[s];
			const int row = s == 0 ? 0 : 1, col = s == 2 ? 1 : 0;
			auto fetch = [&](int e, int b) -> Cplx {
				const long qr = query_row(L, e, s, idx, b);
				if (e == 1) return pred1 ? Cplx{pred1[2 * qr], pred1[2 * qr + 1]} : Cplx{0.0, 0.0};
				const double* __restrict__ p = e == 0 ? pred0 : pred2;
				return Cplx{p ? p[qr] : 0.0, 0.0};
			};
			if (!coupled[t] && L.new_points) // evolve.cpp:439-442
			{
				rho_new[2 * t] = 0.0, rho_new[2 * t + 1] = 0.0;
				return;
			}
			if (!coupled[t])
			{
				// rho = distribution(r_old) * exp(-i omega0 dt), evolve.cpp:415
				const Cplx d = fetch(s, 0);
				const double w = row == col ? 0.0 : 1.0 * ((adiabatic(r_old[2 * t], model).e1 - adiabatic(r_old[2 * t], model).e0)
					+ (adiabatic(r_new[2 * t], model).e1 - adiabatic(r_new[2 * t], model).e0)) / 2.0 / HBAR; // omega0(x0, x2, Forward, 1, 0)
				const Cplx z = cmul_phase(d, -w * dt);
				rho_new[2 * t] = z.re, rho_new[2 * t + 1] = z.im;
				return;
			}
			const double x0 = r_new[2 * t], p0 = r_new[2 * t + 1];
			const BackProp g = back_propagate(x0, p0, mass, dt, row, col, model);
			Cplx comb[3] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
			for (int b = 0; b < 3; ++b)
			{
				Cplx rp[3];
#pragma unroll
				for (int e = 0; e < 3; ++e) rp[e] = (e == s && b == 1 && !L.new_points) ? Cplx{rho_old[2 * t], rho_old[2 * t + 1]} : fetch(e, b);
				rp[1] = cmul_phase(rp[1], omega01(g.x2, g.x4[1][b], 1.0, model) * dt / 2.0); // :327-329
				offdiagonal_rotation(rp, g.x2, g.p2[b], mass, dt / 2.0, model);              // :331-337
				if (b == 0) // branch -1, :341-344: the same value goes to all three elements
				{
					const Cplx v{(rp[0].re + 2.0 * rp[1].re + rp[2].re) / 4.0, (rp[0].im + rp[2].im) / 4.0};
#pragma unroll
					for (int e = 0; e < 3; ++e) comb[e].re += v.re, comb[e].im += v.im;
				}
				else if (b == 1) // branch 0, :345-352
				{
					const Cplx v{(rp[0].re - rp[2].re) / 2.0, (rp[0].im - rp[2].im) / 2.0};
					comb[0].re += v.re, comb[0].im += v.im;
					comb[1].im += rp[1].im;
					comb[2].re -= v.re, comb[2].im -= v.im;
				}
				else // branch +1, :353-361
				{
					const Cplx v{(rp[0].re - 2.0 * rp[1].re + rp[2].re) / 4.0, (rp[0].im + rp[2].im) / 4.0};
					comb[0].re += v.re, comb[0].im += v.im;
					comb[1].re -= v.re, comb[1].im -= v.im;
					comb[2].re += v.re, comb[2].im += v.im;
				}
			}
			offdiagonal_rotation(comb, g.x2, g.p1, mass, dt / 2.0, model); // :369-375
			Cplx res = comb[s];
			if (row != col) res = cmul_phase(res, omega01(x0, g.x2, 1.0, model) * dt / 2.0); // :379-382
			rho_new[2 * t] = res.re, rho_new[2 * t + 1] = res.im;
		}

		// adiabatic quantities at M positions, for the parity tests: out[6 * i + {0..5}] = E0, E1, F00, F10, F11, NAC01
		__global__ void __launch_bounds__(256) pes_kernel(const double* __restrict__ x, int M, int model, double* __restrict__ out)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i >= M) return;
			const Adia a = adiabatic(x[i], model);
			out[6 * i] = a.e0, out[6 * i + 1] = a.e1, out[6 * i + 2] = a.f00, out[6 * i + 3] = a.f10, out[6 * i + 4] = a.f11, out[6 * i + 5] = a.nac01;
		}

		// ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) -------------------
		__device__ __forceinline__ void philox4x32(unsigned (&c)[4], unsigned k0, unsigned k1)
		{
#pragma unroll
			for (int round = 0; round < 10; ++round)
			{
				const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
				const unsigned hi0 = static_cast<unsigned>(p0 >> 32), lo0 = static_cast<unsigned>(p0), hi1 = static_cast<unsigned>(p1 >> 32), lo1 = static_cast<unsigned>(p1);
				const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
				c[0] = n0, c[1] = lo1, c[2] = n2, c[3] = lo0;
				k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
			}
		}
		__device__ __forceinline__ double unit53(unsigned hi, unsigned lo)
		{
			return static_cast<double>(((static_cast<unsigned long long>(hi) << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
		}
		// three uniforms in [0, 1) for (walker, step): counter (walker, step, block, 0), key = seed
		__device__ __forceinline__ void philox_uniform3(unsigned walker, unsigned step, unsigned long long seed, double& u0, double& u1, double& u2)
		{
			unsigned a[4] = {walker, step, 0u, 0u}, b[4] = {walker, step, 1u, 0u};
			philox4x32(a, static_cast<unsigned>(seed), static_cast<unsigned>(seed >> 32));
			philox4x32(b, static_cast<unsigned>(seed), static_cast<unsigned>(seed >> 32));
			u0 = unit53(a[0], a[1]), u1 = unit53(a[2], a[3]), u2 = unit53(b[0], b[1]);
		}
		// mc.cpp:134-142: uniform displacement in [-d, d) per dimension
		__global__ void __launch_bounds__(256) mc_propose_kernel(const double* __restrict__ r, int n, unsigned step, unsigned long long seed, double d,
			double* __restrict__ r_prop)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i >= n) return;
			double u0, u1, u2;
			philox_uniform3(static_cast<unsigned>(i), step, seed, u0, u1, u2);
			r_prop[2 * i] = r[2 * i] + (2.0 * u0 - 1.0) * d;
			r_prop[2 * i + 1] = r[2 * i + 1] + (2.0 * u1 - 1.0) * d;
		}
		// |distribution| of the proposed points: pred holds the cut-off prediction (real, or (re, im) pairs), nullptr = 0
		__device__ __forceinline__ double weight_of(const double* __restrict__ pred, int is_complex, int i)
		{
			if (!pred) return 0.0;
			return is_complex ? hypot(pred[2 * i], pred[2 * i + 1]) : fabs(pred[i]);
		}
		__global__ void __launch_bounds__(256) mc_weight_kernel(const double* __restrict__ pred, int is_complex, int n, double* __restrict__ weight)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i < n) weight[i] = weight_of(pred, is_complex, i);
		}
		// mc.cpp:143-155: accept when the new weight is larger, or with probability new / old
		__global__ void __launch_bounds__(256) mc_accept_kernel(double* __restrict__ r, const double* __restrict__ r_prop, const double* __restrict__ pred,
			int is_complex, int n, unsigned step, unsigned long long seed, double* __restrict__ weight, unsigned* __restrict__ accepted)
		{
			const int i = blockIdx.x * 256 + threadIdx.x;
			if (i >= n) return;
			double u0, u1, u2;
			philox_uniform3(static_cast<unsigned>(i), step, seed, u0, u1, u2);
			const double w_new = weight_of(pred, is_complex, i), w_old = weight[i];
			if (w_new > w_old || w_new / w_old > u2)
			{
				r[2 * i] = r_prop[2 * i], r[2 * i + 1] = r_prop[2 * i + 1];
				weight[i] = w_new;
				accepted[i] += 1u;
			}
		}
	} // namespace

	hipError_t launch_pes(hipStream_t s, const double* x, int M, int model, double* out6)
	{
		if (M == 0) return hipSuccess;
		hipLaunchKernelGGL(pes_kernel, dim3((M + 255) / 256), dim3(256), 0, s, x, M, model, out6);
		return hipGetLastError();
	}
	void evolve_layout(const int n[3], long qoff[3][3], long qlen[3], int off[3], int new_points)
	{
		off[0] = 0, off[1] = n[0], off[2] = n[0] + n[1];
		for (int e = 0; e < 3; ++e)
		{
			long pos = 0;
			for (int s = 0; s < 3; ++s)
			{
				qoff[e][s] = pos;
				pos += static_cast<long>(e == s && !new_points ? 2 : 3) * n[s];
			}
			qlen[e] = pos;
		}
	}
	hipError_t launch_evolve_prepare(hipStream_t s, const double* r, const int n[3], double mass, double dt, int model, double* r_new,
		unsigned char* coupled, double* const q[3], int new_points)
	{
		EvolveLayout L;
		L.new_points = new_points;
		evolve_layout(n, L.qoff, L.qlen, L.off, new_points);
		for (int e = 0; e < 3; ++e) L.n[e] = n[e];
		const int total = n[0] + n[1] + n[2];
		if (total == 0) return hipSuccess;
		hipLaunchKernelGGL(evolve_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, s, r, L, mass, dt, model, r_new, coupled, q[0], q[1], q[2]);
		return hipGetLastError();
	}
	hipError_t launch_evolve_combine(hipStream_t s, const double* r_old, const double* r_new, const double* rho_old, const unsigned char* coupled,
		const int n[3], double mass, double dt, int model, const double* const pred[3], double* rho_new, int new_points)
	{
		EvolveLayout L;
		L.new_points = new_points;
		evolve_layout(n, L.qoff, L.qlen, L.off, new_points);
		for (int e = 0; e < 3; ++e) L.n[e] = n[e];
		const int total = n[0] + n[1] + n[2];
		if (total == 0) return hipSuccess;
		hipLaunchKernelGGL(evolve_combine_kernel, dim3((total + 255) / 256), dim3(256), 0, s, r_old, r_new, rho_old, coupled, L, mass, dt, model, pred[0], pred[1],
			pred[2], rho_new);
		return hipGetLastError();
	}
	hipError_t launch_mc_propose(hipStream_t s, const double* r, int n, unsigned step, unsigned long long seed, double d, double* r_prop)
	{
		hipLaunchKernelGGL(mc_propose_kernel, dim3((n + 255) / 256), dim3(256), 0, s, r, n, step, seed, d, r_prop);
		return hipGetLastError();
	}
	hipError_t launch_mc_weight(hipStream_t s, const double* pred, int is_complex, int n, double* weight)
	{
		hipLaunchKernelGGL(mc_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pred, is_complex, n, weight);
		return hipGetLastError();
	}
	hipError_t launch_mc_accept(hipStream_t s, double* r, const double* r_prop, const double* pred, int is_complex, int n, unsigned step,
		unsigned long long seed, double* weight, unsigned* accepted)
	{
		hipLaunchKernelGGL(mc_accept_kernel, dim3((n + 255) / 256), dim3(256), 0, s, r, r_prop, pred, is_complex, n, step, seed, weight, accepted);
		return hipGetLastError();
	}
} // namespace gple
