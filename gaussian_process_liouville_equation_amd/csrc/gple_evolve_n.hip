// gple_evolve_n.hip — the step loop around the GP for N-level systems ("extend non_adiabatic_evolve_predict
// beyond NumPES == 2"; BASELINE configs[4]: 3-state PES).  The reference asserts at three levels (evolve.cpp:367-371); what is generic
// in its sources is kept as it is — pes.cpp:73-155 (adiabatic states from an eigen-decomposition, F = C^T F_dia C, d_jk = F_jk /
// (E_j - E_k)), adiabatic_evolve, calculate_omega0, evolve() — and the back-propagation is the N-level form of the operator splitting
// its two-level code implements (derivation: DESIGN.md §10; numpy restatement: oracle/evolve_oracle_n.py, which reproduces the
// two-level oracle to rounding at N = 2):
//     exp(L dt) ~ A(dt/2) R(dt/2) J(dt) R(dt/2) A(dt/2)
//     A  classical motion of element (k, l) on the mean surface, phase exp(-i (E_k - E_l) t)
//     R  rho <- O rho O^T, O = exp(-v D t), D = antisymmetric matrix of the non-adiabatic couplings        (evolve.cpp:219-235 at N = 2)
//     J  P_a rho P_b translated in momentum by (lambda_a + lambda_b) / 2 t, (lambda, P) = eigen-system of F_off  (evolve.cpp:244-367 at N = 2)
// N (N + 1) / 2 branches x N (N + 1) / 2 source elements: 9 predicted densities per sample point at N = 2, 36 at N = 3, gathered into
// one query list per element exactly like the two-level kernels of gple_evolve.hip do.
// Potentials: models 0-2 are pes.cpp's diabatic_potential as it compiles for NumPES = 3 (Tully's two surfaces + an uncoupled third
// diabat at V = 0); model 3 (TSAC) is a genuinely three-level model of ours (the reference has none): V00 = A tanh(B x), V11 = 0,
// V22 = -A tanh(B x), V01 = V12 = C sech(D x).  Adiabatic states: eigenvalues ascending, last non-zero component of every eigenvector
// positive (pes.cpp:73-96 at two levels; Eigen's sign for more — unpinned in the reference).
#include "gple_kernels.h"

namespace gple
{
	namespace
	{
		constexpr double HBAR_N = 1.0;
		template <int NP>
		struct Mat
		{
			double a[NP][NP];
		};
		template <int NP>
		struct AdiaN
		{
			double E[NP];
			Mat<NP> F, NAC;
		};
		__device__ __forceinline__ double sgn_n(double v) { return static_cast<double>((v > 0.0) - (v < 0.0)); }

		template <int NP>
		__device__ __forceinline__ void diabatic_n(double x, int model, Mat<NP>& V, Mat<NP>& F)
		{
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j) V.a[i][j] = 0.0, F.a[i][j] = 0.0;
			if (model == 0) // SAC, pes.cpp:40-44, 58-62
			{
				constexpr double A = 0.01, B = 1.6, C = 0.005, D = 1.0;
				const double e = exp(-sgn_n(x) * B * x);
				V.a[0][0] = sgn_n(x) * A * (1.0 - e), V.a[1][1] = -V.a[0][0], V.a[0][1] = V.a[1][0] = C * exp(-D * x * x);
				F.a[0][0] = -A * B * e, F.a[1][1] = -F.a[0][0], F.a[0][1] = F.a[1][0] = 2.0 * C * D * x * exp(-D * x * x);
			}
			else if (model == 1) // DAC
			{
				constexpr double A = 0.10, B = 0.28, C = 0.015, D = 0.06, E = 0.05;
				V.a[1][1] = E - A * exp(-B * x * x), V.a[0][1] = V.a[1][0] = C * exp(-D * x * x);
				F.a[1][1] = -2 * A * B * x * exp(-B * x * x), F.a[0][1] = F.a[1][0] = 2 * C * D * x * exp(-D * x * x);
			}
			else if (model == 2) // ECR
			{
				constexpr double A = 6e-4, B = 0.10, C = 0.90;
				const double e = exp(-sgn_n(x) * C * x);
				V.a[0][0] = A, V.a[1][1] = -A, V.a[0][1] = V.a[1][0] = B * (1 - sgn_n(x) * (e - 1));
				F.a[0][1] = F.a[1][0] = -B * C * e;
			}
			else if constexpr (NP == 3) // TSAC (ours)
			{
				constexpr double A = 0.02, B = 0.8, C = 0.005, D = 0.5;
				const double t = tanh(B * x), g = 1.0 / cosh(D * x), th = tanh(D * x);
				V.a[0][0] = A * t, V.a[2][2] = -A * t;
				V.a[0][1] = V.a[1][0] = V.a[1][2] = V.a[2][1] = C * g;
				F.a[0][0] = -A * B * (1.0 - t * t), F.a[2][2] = A * B * (1.0 - t * t);
				F.a[0][1] = F.a[1][0] = F.a[1][2] = F.a[2][1] = C * D * g * th;
			}
		}

		// Cyclic Jacobi for a symmetric NP x NP matrix (destroyed): eigenvalues ascending in lam, eigenvectors in the columns of W.  A pair
		// whose off-diagonal entry is exactly zero is not rotated, so a decoupled level keeps exact zeros in its row and column.
		template <int NP>
		__device__ __forceinline__ void jacobi_eig(Mat<NP>& A, double (&lam)[NP], Mat<NP>& W)
		{
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j) W.a[i][j] = i == j ? 1.0 : 0.0;
			for (int sweep = 0; sweep < 8; ++sweep)
			{
#pragma unroll
				for (int p = 0; p < NP - 1; ++p)
#pragma unroll
					for (int q = p + 1; q < NP; ++q)
					{
						const double apq = A.a[p][q];
						if (apq == 0.0) continue;
						const double theta = (A.a[q][q] - A.a[p][p]) / (2.0 * apq);
						const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
						const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
						A.a[p][p] -= t * apq, A.a[q][q] += t * apq, A.a[p][q] = A.a[q][p] = 0.0;
#pragma unroll
						for (int r = 0; r < NP; ++r)
						{
							if (r != p && r != q)
							{
								const double arp = A.a[r][p], arq = A.a[r][q];
								A.a[r][p] = A.a[p][r] = c * arp - s * arq;
								A.a[r][q] = A.a[q][r] = s * arp + c * arq;
							}
							const double wrp = W.a[r][p], wrq = W.a[r][q];
							W.a[r][p] = c * wrp - s * wrq, W.a[r][q] = s * wrp + c * wrq;
						}
					}
			}
#pragma unroll
			for (int i = 0; i < NP; ++i) lam[i] = A.a[i][i];
			// ascending (selection sort on NP <= 3 entries, columns move with their eigenvalue)
#pragma unroll
			for (int i = 0; i < NP - 1; ++i)
#pragma unroll
				for (int j = i + 1; j < NP; ++j)
					if (lam[j] < lam[i])
					{
						const double tl = lam[i];
						lam[i] = lam[j], lam[j] = tl;
#pragma unroll
						for (int r = 0; r < NP; ++r)
						{
							const double tw = W.a[r][i];
							W.a[r][i] = W.a[r][j], W.a[r][j] = tw;
						}
					}
		}

		// pes.cpp:73-155 for NP levels
		template <int NP>
		__device__ __forceinline__ void adiabatic_n(double x, int model, AdiaN<NP>& out)
		{
			Mat<NP> V, Fd, C;
			diabatic_n<NP>(x, model, V, Fd);
			jacobi_eig<NP>(V, out.E, C);
#pragma unroll
			for (int k = 0; k < NP; ++k) // last non-zero component positive
			{
				double last = 0.0;
#pragma unroll
				for (int i = 0; i < NP; ++i)
					if (C.a[i][k] != 0.0) last = C.a[i][k];
				if (last < 0.0)
#pragma unroll
					for (int i = 0; i < NP; ++i) C.a[i][k] = -C.a[i][k];
			}
			Mat<NP> M; // M = F_dia C
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int k = 0; k < NP; ++k)
				{
					double acc = 0.0;
#pragma unroll
					for (int j = 0; j < NP; ++j) acc += Fd.a[i][j] * C.a[j][k];
					M.a[i][k] = acc;
				}
#pragma unroll
			for (int k = 0; k < NP; ++k)
#pragma unroll
				for (int l = 0; l <= k; ++l)
				{
					double a1 = 0.0, a2 = 0.0;
#pragma unroll
					for (int i = 0; i < NP; ++i) a1 += C.a[i][k] * M.a[i][l], a2 += C.a[i][l] * M.a[i][k];
					out.F.a[k][l] = out.F.a[l][k] = 0.5 * (a1 + a2);
				}
#pragma unroll
			for (int j = 0; j < NP; ++j)
			{
				out.NAC.a[j][j] = 0.0;
#pragma unroll
				for (int k = 0; k < j; ++k)
				{
					const double d = out.F.a[j][k] == 0.0 ? 0.0 : out.F.a[j][k] / (out.E[j] - out.E[k]);
					out.NAC.a[j][k] = d, out.NAC.a[k][j] = -d;
				}
			}
		}

		template <int NP>
		__device__ __forceinline__ void matmul(const Mat<NP>& A, const Mat<NP>& B, Mat<NP>& C)
		{
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j)
				{
					double acc = 0.0;
#pragma unroll
					for (int k = 0; k < NP; ++k) acc += A.a[i][k] * B.a[k][j];
					C.a[i][j] = acc;
				}
		}
		// O = exp(-vt NAC): Taylor series of the argument scaled below 1/8, then repeated squaring
		template <int NP>
		__device__ __forceinline__ void rotation_n(const Mat<NP>& NAC, double vt, Mat<NP>& O)
		{
			Mat<NP> X;
			double nrm = 0.0;
#pragma unroll
			for (int i = 0; i < NP; ++i)
			{
				double row = 0.0;
#pragma unroll
				for (int j = 0; j < NP; ++j) X.a[i][j] = -vt * NAC.a[i][j], row += fabs(X.a[i][j]);
				nrm = fmax(nrm, row);
			}
			int sq = 0;
			while (nrm > 0.125 && sq < 40) nrm *= 0.5, ++sq;
			const double scale = ldexp(1.0, -sq);
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j) X.a[i][j] *= scale;
			Mat<NP> term, next;
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j) O.a[i][j] = term.a[i][j] = i == j ? 1.0 : 0.0;
			for (int k = 1; k <= 14; ++k)
			{
				matmul<NP>(term, X, next);
#pragma unroll
				for (int i = 0; i < NP; ++i)
#pragma unroll
					for (int j = 0; j < NP; ++j) term.a[i][j] = next.a[i][j] / k, O.a[i][j] += term.a[i][j];
			}
			for (int s = 0; s < sq; ++s)
			{
				matmul<NP>(O, O, next);
				O = next;
			}
		}
		// R <- O R O^T for the real and the imaginary part of a Hermitian matrix
		template <int NP>
		__device__ __forceinline__ void conjugate(const Mat<NP>& O, Mat<NP>& Rre, Mat<NP>& Rim)
		{
			Mat<NP> T, Ot;
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j) Ot.a[i][j] = O.a[j][i];
			matmul<NP>(O, Rre, T), matmul<NP>(T, Ot, Rre);
			matmul<NP>(O, Rim, T), matmul<NP>(T, Ot, Rim);
		}

		template <int NP>
		__device__ __forceinline__ void adiabatic_evolve_n(double& x, double& p, double mass, double dt, double drc, int row, int col, int model)
		{
			x += drc * dt / 2.0 * (p / mass);
			AdiaN<NP> a;
			adiabatic_n<NP>(x, model, a);
			p += drc * dt / 2.0 * (a.F.a[row][row] + a.F.a[col][col]);
			x += drc * dt / 2.0 * (p / mass);
		}

		struct PtrsN // one device pointer per density-matrix element (NE <= 6)
		{
			double* p[6];
		};
		template <int NP>
		struct LayoutN
		{
			static constexpr int NE = NP * (NP + 1) / 2;
			int n[NE], off[NE];
			long qoff[NE][NE]; // row of the first query of the points of element s inside the list of element e: qoff[e][s]
			int new_points;
		};
		template <int NP>
		__device__ __forceinline__ void element_of(const LayoutN<NP>& L, int t, int& s, int& idx, int& row, int& col)
		{
			constexpr int NE = NP * (NP + 1) / 2;
			s = 0;
#pragma unroll
			for (int e = 1; e < NE; ++e)
				if (t >= L.off[e]) s = e;
			idx = t - L.off[s];
			row = 0;
			while ((row + 1) * (row + 2) / 2 <= s) ++row;
			col = s - row * (row + 1) / 2;
		}

		// the geometry of one back-propagation: (x2, p1), the eigen-system of F_off(x2), and per branch (a <= b) the momentum p2
		template <int NP>
		struct BackN
		{
			static constexpr int NE = NP * (NP + 1) / 2;
			double x2, p1, lam[NP], p2[NE];
			Mat<NP> W;
			AdiaN<NP> at2;
		};
		template <int NP>
		__device__ __forceinline__ void back_n(double x0, double p0, double mass, double dt, int row, int col, int model, BackN<NP>& g)
		{
			g.x2 = x0, g.p1 = p0;
			adiabatic_evolve_n<NP>(g.x2, g.p1, mass, dt / 2.0, -1.0, row, col, model);
			adiabatic_n<NP>(g.x2, model, g.at2);
			Mat<NP> Foff = g.at2.F;
#pragma unroll
			for (int k = 0; k < NP; ++k) Foff.a[k][k] = 0.0;
			jacobi_eig<NP>(Foff, g.lam, g.W);
			int br = 0;
#pragma unroll
			for (int a = 0; a < NP; ++a)
#pragma unroll
				for (int b = a; b < NP; ++b) g.p2[br++] = g.p1 + dt * 0.5 * (g.lam[a] + g.lam[b]);
		}

		// One thread per sample point: forward propagation (two half steps), the NE x NE back-propagated points into the query lists
		template <int NP>
		__global__ void __launch_bounds__(128) evolve_prepare_n_kernel(const double* __restrict__ r, LayoutN<NP> L, double mass, double dt, int model,
			double* __restrict__ r_new, PtrsN q)
		{
			constexpr int NE = NP * (NP + 1) / 2;
			const int t = blockIdx.x * 128 + threadIdx.x;
			const int total = L.off[NE - 1] + L.n[NE - 1];
			if (t >= total) return;
			int s, idx, row, col;
			element_of<NP>(L, t, s, idx, row, col);
			double xn = r[2 * t], pn = r[2 * t + 1];
			if (!L.new_points)
			{
				adiabatic_evolve_n<NP>(xn, pn, mass, dt / 2.0, 1.0, row, col, model);
				adiabatic_evolve_n<NP>(xn, pn, mass, dt / 2.0, 1.0, row, col, model);
			}
			r_new[2 * t] = xn, r_new[2 * t + 1] = pn;
			BackN<NP> g;
			back_n<NP>(xn, pn, mass, dt, row, col, model, g);
			for (int b = 0; b < NE; ++b)
			{
				const double x3 = g.x2 - (dt / 4.0) * g.p2[b] / mass;
				AdiaN<NP> a3;
				adiabatic_n<NP>(x3, model, a3);
				int e = 0;
#pragma unroll
				for (int k = 0; k < NP; ++k)
#pragma unroll
					for (int l = 0; l <= k; ++l, ++e)
					{
						const double p3 = g.p2[b] - (dt / 4.0) * (a3.F.a[k][k] + a3.F.a[l][l]);
						const double x4 = x3 - (dt / 4.0) * p3 / mass;
						const long qr = L.qoff[e][s] + static_cast<long>(NE) * idx + b;
						q.p[e][2 * qr] = x4, q.p[e][2 * qr + 1] = p3;
					}
			}
		}

		// One thread per sample point: the predicted densities at its back-propagated points -> the density at its new position.
		// pred[e]: plain doubles for a diagonal element, (re, im) pairs for an off-diagonal one, nullptr for an element without a kernel
		template <int NP>
		__global__ void __launch_bounds__(128) evolve_combine_n_kernel(const double* __restrict__ r_new, const double* __restrict__ rho_old, LayoutN<NP> L,
			double mass, double dt, int model, PtrsN pred, double* __restrict__ rho_new)
		{
			constexpr int NE = NP * (NP + 1) / 2;
			const int t = blockIdx.x * 128 + threadIdx.x;
			const int total = L.off[NE - 1] + L.n[NE - 1];
			if (t >= total) return;
			int s, idx, row, col;
			element_of<NP>(L, t, s, idx, row, col);
			const double x0 = r_new[2 * t], p0 = r_new[2 * t + 1];
			BackN<NP> g;
			back_n<NP>(x0, p0, mass, dt, row, col, model, g);
			double lam_scale = 0.0;
#pragma unroll
			for (int a = 0; a < NP; ++a) lam_scale = fmax(lam_scale, fabs(g.lam[a]));
			Mat<NP> Cre, Cim;
#pragma unroll
			for (int i = 0; i < NP; ++i)
#pragma unroll
				for (int j = 0; j < NP; ++j) Cre.a[i][j] = Cim.a[i][j] = 0.0;
			int b = 0;
			for (int a1 = 0; a1 < NP; ++a1)
				for (int b1 = a1; b1 < NP; ++b1, ++b)
				{
					const double shift = 0.5 * (g.lam[a1] + g.lam[b1]);
					const bool zero_shift = fabs(shift) <= 1e-13 * lam_scale; // the branch that retraces the forward step (evolve.cpp:309-313)
					const double x3 = g.x2 - (dt / 4.0) * g.p2[b] / mass;
					AdiaN<NP> a3;
					adiabatic_n<NP>(x3, model, a3);
					Mat<NP> Rre, Rim;
					int e = 0;
#pragma unroll
					for (int k = 0; k < NP; ++k)
#pragma unroll
						for (int l = 0; l <= k; ++l, ++e)
						{
							const long qr = L.qoff[e][s] + static_cast<long>(NE) * idx + b;
							double vre = 0.0, vim = 0.0;
							if (e == s && zero_shift && !L.new_points) vre = rho_old[2 * t], vim = rho_old[2 * t + 1];
							else if (pred.p[e])
							{
								if (k == l) vre = pred.p[e][qr];
								else vre = pred.p[e][2 * qr], vim = pred.p[e][2 * qr + 1];
							}
							if (k != l)
							{
								// exp(i calculate_omega0(x2, x4, Forward, l, k) dt / 2): (E_l - E_k) averaged over x2 and x4 (evolve.cpp:327-329)
								const double p3 = g.p2[b] - (dt / 4.0) * (a3.F.a[k][k] + a3.F.a[l][l]);
								const double x4 = x3 - (dt / 4.0) * p3 / mass;
								AdiaN<NP> a4;
								adiabatic_n<NP>(x4, model, a4);
								const double w = (g.at2.E[l] - g.at2.E[k] + a4.E[l] - a4.E[k]) / 2.0 / HBAR_N * dt / 2.0;
								const double c = cos(w), sn = sin(w), re = vre * c - vim * sn, im = vre * sn + vim * c;
								vre = re, vim = im;
							}
							Rre.a[k][l] = Rre.a[l][k] = vre;
							Rim.a[k][l] = vim, Rim.a[l][k] = -vim;
							if (k == l) Rim.a[k][k] = vim; // (a diagonal prediction is real; the exact density of a diagonal element may carry an imaginary part)
						}
					Mat<NP> O;
					rotation_n<NP>(g.at2.NAC, g.p2[b] / mass * (dt / 2.0), O);
					conjugate<NP>(O, Rre, Rim);
					// comb += P_a R P_b (+ P_b R P_a when a != b), P_a = u_a u_a^T:  z_ab = u_a^T R u_b, z_ba = u_b^T R u_a (= conj(z_ab) for a Hermitian
					// R; the diagonal of R may carry an imaginary part — the exact density is stored complex — so both are formed)
					double zre = 0.0, zim = 0.0, yre = 0.0, yim = 0.0;
#pragma unroll
					for (int i = 0; i < NP; ++i)
#pragma unroll
						for (int j = 0; j < NP; ++j)
						{
							const double ab = g.W.a[i][a1] * g.W.a[j][b1], ba = g.W.a[i][b1] * g.W.a[j][a1];
							zre += ab * Rre.a[i][j], zim += ab * Rim.a[i][j];
							yre += ba * Rre.a[i][j], yim += ba * Rim.a[i][j];
						}
#pragma unroll
					for (int i = 0; i < NP; ++i)
#pragma unroll
						for (int j = 0; j < NP; ++j)
						{
							const double ab = g.W.a[i][a1] * g.W.a[j][b1];
							Cre.a[i][j] += zre * ab, Cim.a[i][j] += zim * ab;
							if (a1 != b1)
							{
								const double ba = g.W.a[i][b1] * g.W.a[j][a1];
								Cre.a[i][j] += yre * ba, Cim.a[i][j] += yim * ba;
							}
						}
				}
			Mat<NP> O;
			rotation_n<NP>(g.at2.NAC, g.p1 / mass * (dt / 2.0), O);
			conjugate<NP>(O, Cre, Cim);
			double re = Cre.a[row][col], im = Cim.a[row][col];
			if (row != col)
			{
				AdiaN<NP> a0;
				adiabatic_n<NP>(x0, model, a0);
				const double w = (a0.E[col] - a0.E[row] + g.at2.E[col] - g.at2.E[row]) / 2.0 / HBAR_N * dt / 2.0; // evolve.cpp:379-382
				const double c = cos(w), sn = sin(w), r2 = re * c - im * sn, i2 = re * sn + im * c;
				re = r2, im = i2;
			}
			rho_new[2 * t] = re, rho_new[2 * t + 1] = im;
		}

		// adiabatic quantities at M positions: out[(NP + 2 NE) i + ...] = E (NP), F lower-packed (NE), NAC lower-packed (NE: NAC(j, k), j >= k)
		template <int NP>
		__global__ void __launch_bounds__(128) pes_n_kernel(const double* __restrict__ x, int M, int model, double* __restrict__ out)
		{
			constexpr int NE = NP * (NP + 1) / 2, W = NP + 2 * NE;
			const int i = blockIdx.x * 128 + threadIdx.x;
			if (i >= M) return;
			AdiaN<NP> a;
			adiabatic_n<NP>(x[i], model, a);
			double* o = out + static_cast<long>(W) * i;
#pragma unroll
			for (int k = 0; k < NP; ++k) o[k] = a.E[k];
			int e = 0;
#pragma unroll
			for (int k = 0; k < NP; ++k)
#pragma unroll
				for (int l = 0; l <= k; ++l, ++e) o[NP + e] = a.F.a[k][l], o[NP + NE + e] = a.NAC.a[k][l];
		}

		template <int NP>
		void fill_layout(const int* n, int new_points, LayoutN<NP>& L, long* qlen)
		{
			constexpr int NE = NP * (NP + 1) / 2;
			L.new_points = new_points;
			int pos = 0;
			for (int e = 0; e < NE; ++e) L.n[e] = n[e], L.off[e] = pos, pos += n[e];
			for (int e = 0; e < NE; ++e)
			{
				long q = 0;
				for (int s = 0; s < NE; ++s) L.qoff[e][s] = q, q += static_cast<long>(NE) * n[s];
				qlen[e] = q;
			}
		}
	} // namespace

	hipError_t launch_pes_n(hipStream_t s, int num_pes, const double* x, int M, int model, double* out)
	{
		if (M == 0) return hipSuccess;
		if (num_pes == 2) hipLaunchKernelGGL(pes_n_kernel<2>, dim3((M + 127) / 128), dim3(128), 0, s, x, M, model, out);
		else if (num_pes == 3) hipLaunchKernelGGL(pes_n_kernel<3>, dim3((M + 127) / 128), dim3(128), 0, s, x, M, model, out);
		else return hipErrorInvalidValue;
		return hipGetLastError();
	}
	void evolve_layout_n(int num_pes, const int* n, long* qlen)
	{
		const int NE = num_pes * (num_pes + 1) / 2;
		long total = 0;
		for (int s = 0; s < NE; ++s) total += static_cast<long>(NE) * n[s];
		for (int e = 0; e < NE; ++e) qlen[e] = total; // every element is asked at every branch of every point
	}
	hipError_t launch_evolve_prepare_n(hipStream_t s, int num_pes, const double* r, const int* n, double mass, double dt, int model, double* r_new,
		double* const* q, int new_points)
	{
		long qlen[6];
		PtrsN q_dev{};
		for (int e = 0; e < num_pes * (num_pes + 1) / 2 && e < 6; ++e) q_dev.p[e] = q[e];
		if (num_pes == 2)
		{
			LayoutN<2> L;
			fill_layout<2>(n, new_points, L, qlen);
			const int total = L.off[2] + L.n[2];
			if (total) hipLaunchKernelGGL(evolve_prepare_n_kernel<2>, dim3((total + 127) / 128), dim3(128), 0, s, r, L, mass, dt, model, r_new, q_dev);
		}
		else if (num_pes == 3)
		{
			LayoutN<3> L;
			fill_layout<3>(n, new_points, L, qlen);
			const int total = L.off[5] + L.n[5];
			if (total) hipLaunchKernelGGL(evolve_prepare_n_kernel<3>, dim3((total + 127) / 128), dim3(128), 0, s, r, L, mass, dt, model, r_new, q_dev);
		}
		else return hipErrorInvalidValue;
		return hipGetLastError();
	}
	hipError_t launch_evolve_combine_n(hipStream_t s, int num_pes, const double* r_new, const double* rho_old, const int* n, double mass, double dt, int model,
		const double* const* pred, double* rho_new, int new_points)
	{
		long qlen[6];
		PtrsN pred_dev{};
		for (int e = 0; e < num_pes * (num_pes + 1) / 2 && e < 6; ++e) pred_dev.p[e] = const_cast<double*>(pred[e]);
		if (num_pes == 2)
		{
			LayoutN<2> L;
			fill_layout<2>(n, new_points, L, qlen);
			const int total = L.off[2] + L.n[2];
			if (total) hipLaunchKernelGGL(evolve_combine_n_kernel<2>, dim3((total + 127) / 128), dim3(128), 0, s, r_new, rho_old, L, mass, dt, model, pred_dev, rho_new);
		}
		else if (num_pes == 3)
		{
			LayoutN<3> L;
			fill_layout<3>(n, new_points, L, qlen);
			const int total = L.off[5] + L.n[5];
			if (total) hipLaunchKernelGGL(evolve_combine_n_kernel<3>, dim3((total + 127) / 128), dim3(128), 0, s, r_new, rho_old, L, mass, dt, model, pred_dev, rho_new);
		}
		else return hipErrorInvalidValue;
		return hipGetLastError();
	}
} // namespace gple
