// gple_oracle.cpp — CPU ORACLE: plain C++ restatement of the reference's GPR fit + predict algorithm.
//
// TEST INFRASTRUCTURE ONLY (see gple_oracle.h).  Nothing in the product path may call into this file.
// Every function cites the reference file:line it follows; paths are relative to
// /root/reference/gaussian_process_liouville_equation/ unless they start with test/.
//
// Restated third-party semantics (Eigen 3.4, un-vendored, no pinned version => "parity unpinned"):
//   * Eigen::LDLT            = diagonal-pivoting  P^T L D L^H P, pivot = largest |diagonal| of the *not yet
//                              updated* trailing diagonal (Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked);
//                              solve() uses the pseudo-inverse of D with tolerance DBL_MIN.
//   * Eigen::LLT             = unpivoted lower Cholesky.
//   * selfadjointView<Lower> = mirror the lower triangle (conjugated) into the upper one.
// Arithmetic order follows the reference expression by expression; compile with -ffp-contract=off so that no
// FMA contraction sneaks into the Gram-matrix arguments.
#include "gple_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <complex>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
	using cd = std::complex<double>;
	constexpr double pi = 3.141592653589793238462643383279502884;
	constexpr double sqrt2 = 1.414213562373095048801688724209698079;
	constexpr int Dim = 1;                            // stdafx.h:119
	constexpr int PhaseDim = 2;                       // stdafx.h:121
	constexpr double hbar = 1.0;                      // stdafx.h:107
	constexpr double PurityFactor = 2.0 * pi * hbar;  // stdafx.h:125 (Dim = 1)
	constexpr double RescaleMaximum = 10.0;           // kernel.h:37
	constexpr double ConnectingPoint = 2.0;           // kernel.h:16
	static_assert(Dim == 1 && PhaseDim == 2, "the oracle is written for Dim = 1");

	template <typename S>
	struct Matrix
	{
		size_t r = 0, c = 0;
		std::vector<S> a;
		Matrix() = default;
		Matrix(size_t rows, size_t cols, S v = S(0)): r(rows), c(cols), a(rows * cols, v) {}
		S& operator()(size_t i, size_t j) { return a[i + j * r]; }
		const S& operator()(size_t i, size_t j) const { return a[i + j * r]; }
		S* col(size_t j) { return a.data() + j * r; }
		const S* col(size_t j) const { return a.data() + j * r; }
	};
	using Mat = Matrix<double>;
	using CMat = Matrix<cd>;
	using Vec = std::vector<double>;
	using CVec = std::vector<cd>;

	inline double sq(double x) { return x * x; }
	inline double conj_(double x) { return x; }
	inline cd conj_(cd x) { return std::conj(x); }
	inline double real_(double x) { return x; }
	inline double real_(cd x) { return x.real(); }

	// ---------------------------------------------------------------------------------------------------------
	// dense helpers (column-major, OpenMP over result columns like the reference's MKL/TBB threading)
	// ---------------------------------------------------------------------------------------------------------
	template <typename SA, typename SB>
	auto gemm(const Matrix<SA>& A, const Matrix<SB>& B) -> Matrix<decltype(SA() * SB())>
	{
		using S = decltype(SA() * SB());
		Matrix<S> C(A.r, B.c);
		const size_t n = A.r, K = A.c;
#pragma omp parallel for schedule(static)
		for (size_t j = 0; j < B.c; j++)
		{
			S* cj = C.col(j);
			for (size_t k = 0; k < K; k++)
			{
				const SB b = B(k, j);
				const SA* ak = A.col(k);
				for (size_t i = 0; i < n; i++)
				{
					cj[i] += ak[i] * b;
				}
			}
		}
		return C;
	}
	template <typename S>
	Matrix<S> adjoint(const Matrix<S>& A)
	{
		Matrix<S> R(A.c, A.r);
		for (size_t j = 0; j < A.c; j++)
			for (size_t i = 0; i < A.r; i++)
				R(j, i) = conj_(A(i, j));
		return R;
	}
	CMat conjugate(const CMat& A)
	{
		CMat R = A;
		for (cd& z : R.a) z = std::conj(z);
		return R;
	}
	template <typename SA, typename SV>
	auto matvec(const Matrix<SA>& A, const std::vector<SV>& x) -> std::vector<decltype(SA() * SV())>
	{
		using S = decltype(SA() * SV());
		std::vector<S> y(A.r, S(0));
		for (size_t k = 0; k < A.c; k++)
		{
			const SV xk = x[k];
			const SA* ak = A.col(k);
			for (size_t i = 0; i < A.r; i++) y[i] += ak[i] * xk;
		}
		return y;
	}
	// x^T A y without conjugation
	template <typename SX, typename SA, typename SY>
	auto bilinear(const std::vector<SX>& x, const Matrix<SA>& A, const std::vector<SY>& y) -> decltype(SX() * SA() * SY())
	{
		using S = decltype(SX() * SA() * SY());
		S total(0);
		for (size_t j = 0; j < A.c; j++)
		{
			decltype(SX() * SA()) s(0);
			const SA* aj = A.col(j);
			for (size_t i = 0; i < A.r; i++) s += x[i] * aj[i];
			total += s * y[j];
		}
		return total;
	}

	// ---------------------------------------------------------------------------------------------------------
	// Eigen::LDLT restated (Eigen/src/Cholesky/LDLT.h): used by kernel.cpp:281-283, complex_kernel.cpp:264-266,419
	// ---------------------------------------------------------------------------------------------------------
	template <typename S>
	struct LDLT
	{
		size_t n = 0;
		Matrix<S> m;                // strictly lower: L (unit diagonal implied); diagonal: D
		std::vector<size_t> transp; // transpositions
		int info = 0;               // 0 fine, k+1 if pivot k was exactly zero

		explicit LDLT(const Matrix<S>& A): n(A.r), m(A), transp(A.r)
		{
			std::vector<S> temp(n);
			for (size_t k = 0; k < n; k++)
			{
				// largest |diagonal| of the trailing part; first maximum wins (strict >)
				size_t big = k;
				double best = std::abs(m(k, k));
				for (size_t i = k + 1; i < n; i++)
				{
					const double v = std::abs(m(i, i));
					if (v > best)
					{
						best = v;
						big = i;
					}
				}
				transp[k] = big;
				if (big != k)
				{
					// symmetric transposition restricted to the lower triangle
					for (size_t j = 0; j < k; j++) std::swap(m(k, j), m(big, j));
					for (size_t i = big + 1; i < n; i++) std::swap(m(i, k), m(i, big));
					std::swap(m(k, k), m(big, big));
					for (size_t i = k + 1; i < big; i++)
					{
						const S tmp = m(i, k);
						m(i, k) = conj_(m(big, i));
						m(big, i) = conj_(tmp);
					}
					m(big, k) = conj_(m(big, k));
				}
				const size_t rs = n - k - 1;
				if (k > 0)
				{
					// temp = D(0:k) .* A10^H ; A11 -= A10 temp ; A21 -= A20 temp
					for (size_t j = 0; j < k; j++) temp[j] = real_(m(j, j)) * conj_(m(k, j));
					S dot(0);
					for (size_t j = 0; j < k; j++) dot += m(k, j) * temp[j];
					m(k, k) -= dot;
					if (rs > 0)
					{
						S* a21 = &m(k + 1, k);
#pragma omp parallel for schedule(static) if (rs * k > 65536)
						for (size_t blk = 0; blk < (rs + 255) / 256; blk++)
						{
							const size_t i0 = blk * 256, i1 = std::min(rs, i0 + 256);
							for (size_t j = 0; j < k; j++)
							{
								const S t = temp[j];
								const S* a20 = &m(k + 1, j);
								for (size_t i = i0; i < i1; i++) a21[i] -= a20[i] * t;
							}
						}
					}
				}
				const double akk = real_(m(k, k));
				const bool pivot_is_valid = std::abs(akk) > 0.0;
				if (!pivot_is_valid && info == 0) info = static_cast<int>(k) + 1;
				if (k == 0 && !pivot_is_valid)
				{
					for (size_t j = 0; j < n; j++) transp[j] = j;
					return;
				}
				if (rs > 0 && pivot_is_valid)
				{
					for (size_t i = k + 1; i < n; i++) m(i, k) /= akk;
				}
			}
		}

		// solve A X = B in place (B is n x nrhs), LDLT::_solve_impl_transposed
		template <typename SB>
		void solve_inplace(Matrix<SB>& B) const
		{
			const double tolerance = std::numeric_limits<double>::min();
#pragma omp parallel for schedule(dynamic, 8)
			for (size_t c = 0; c < B.c; c++)
			{
				SB* x = B.col(c);
				for (size_t k = 0; k < n; k++)
					if (transp[k] != k) std::swap(x[k], x[transp[k]]);
				for (size_t k = 0; k < n; k++) // unit lower forward
				{
					const SB xk = x[k];
					const S* l = m.col(k);
					for (size_t i = k + 1; i < n; i++) x[i] -= l[i] * xk;
				}
				for (size_t i = 0; i < n; i++)
				{
					const double d = real_(m(i, i));
					if (std::abs(d) > tolerance)
						x[i] /= d;
					else
						x[i] = SB(0);
				}
				for (size_t kk = n; kk-- > 0;) // L^H backward
				{
					const S* l = m.col(kk);
					SB s(0);
					for (size_t i = kk + 1; i < n; i++) s += conj_(l[i]) * x[i];
					x[kk] -= s;
				}
				for (size_t kk = n; kk-- > 0;)
					if (transp[kk] != kk) std::swap(x[kk], x[transp[kk]]);
			}
		}
	};

	template <typename S>
	Matrix<S> identity(size_t n)
	{
		Matrix<S> I(n, n);
		for (size_t i = 0; i < n; i++) I(i, i) = S(1);
		return I;
	}

	// ---------------------------------------------------------------------------------------------------------
	// kernel.cpp:8-242  — delta kernel, Gaussian kernel, derivatives, KernelBase
	// ---------------------------------------------------------------------------------------------------------
	struct KernelParameter // kernel.h:41
	{
		double magnitude;
		double length[PhaseDim];
		double noise;
	};

	// kernel.cpp:8-31.  `same` restates `LeftFeature.data() == RightFeature.data()`.
	Mat delta_kernel(const double* L, size_t R, const double* Rt, size_t C, bool same)
	{
		if (same) return identity<double>(R);
		Mat result(R, C);
#pragma omp parallel for schedule(static)
		for (size_t j = 0; j < C; j++)
			for (size_t i = 0; i < R; i++)
				result(i, j) = static_cast<double>(L[2 * i] == Rt[2 * j] && L[2 * i + 1] == Rt[2 * j + 1]);
		return result;
	}

	// kernel.cpp:38-85 (element: :46-47 — subtract, divide by l, square, sum, negate, /2.0, exp)
	inline double gaussian_element(const double* l, const double* a, const double* b)
	{
		const double d0 = (a[0] - b[0]) / l[0];
		const double d1 = (a[1] - b[1]) / l[1];
		return std::exp(-(d0 * d0 + d1 * d1) / 2.0);
	}
	Mat gaussian_kernel(const double* l, const double* L, size_t R, const double* Rt, size_t C, bool same)
	{
		Mat result(R, C);
		if (same && R == C)
		{
			// the reference evaluates all R*C entries here (member copies have different pointers, kernel.cpp:227),
			// the values are those of the symmetric branch bit for bit: (a-b)^2 == (b-a)^2, exp(-0/2) == 1
#pragma omp parallel for schedule(dynamic, 16)
			for (size_t j = 0; j < C; j++)
			{
				result(j, j) = 1.0;
				for (size_t i = j + 1; i < R; i++) result(i, j) = gaussian_element(l, L + 2 * i, Rt + 2 * j);
			}
			for (size_t j = 0; j < C; j++)
				for (size_t i = 0; i < j; i++) result(i, j) = result(j, i);
		}
		else
		{
#pragma omp parallel for schedule(static)
			for (size_t j = 0; j < C; j++)
				for (size_t i = 0; i < R; i++) result(i, j) = gaussian_element(l, L + 2 * i, Rt + 2 * j);
		}
		return result;
	}

	// kernel.cpp:99-160: result[d](i,j) = G(i,j) * ((x_i,d - x_j,d)/l_d)^2 / l_d ; training: diagonal 0, mirrored lower
	std::array<Mat, PhaseDim> gaussian_derivative_over_char_length(const double* l, const double* L, size_t R,
		const double* Rt, size_t C, const Mat& G, bool same)
	{
		std::array<Mat, PhaseDim> result{G, G};
		const bool training = same && R == C;
#pragma omp parallel for schedule(dynamic, 16)
		for (size_t j = 0; j < C; j++)
		{
			for (size_t i = training ? j + 1 : 0; i < R; i++)
			{
				for (int d = 0; d < PhaseDim; d++)
				{
					const double diff = (L[2 * i + d] - Rt[2 * j + d]) / l[d];
					result[d](i, j) *= diff * diff / l[d];
				}
			}
		}
		if (training)
		{
			for (int d = 0; d < PhaseDim; d++)
				for (size_t j = 0; j < C; j++)
				{
					result[d](j, j) = 0.0;
					for (size_t i = 0; i < j; i++) result[d](i, j) = result[d](j, i);
				}
		}
		return result;
	}

	// kernel.cpp:168-215
	std::array<Mat, 4> calculate_derivative(const KernelParameter& p, const double* L, size_t R, const double* Rt,
		size_t C, const Mat& K, bool same)
	{
		std::array<Mat, 4> result;
		result[0] = K;
		for (double& x : result[0].a) x *= 2.0 / p.magnitude; // :181
		std::array<Mat, PhaseDim> gd;
		if (same)
		{
			const double Noise = p.magnitude * p.noise; // :185
			Mat Gm = K;
			for (size_t i = 0; i < R; i++) Gm(i, i) = K(i, i) - sq(Noise) * 1.0; // :190
			gd = gaussian_derivative_over_char_length(p.length, L, R, Rt, C, Gm, true);
		}
		else
		{
			gd = gaussian_derivative_over_char_length(p.length, L, R, Rt, C, K, false);
		}
		result[1] = std::move(gd[0]);
		result[2] = std::move(gd[1]);
		result[3] = Mat(R, C);
		if (same)
			for (size_t i = 0; i < R; i++) result[3](i, i) = 2.0 * sq(p.magnitude) * p.noise * 1.0; // :207
		return result;
	}

	// kernel.cpp:217-242
	struct KernelBase
	{
		KernelParameter params;
		Mat K;
		std::array<Mat, 4> dK;
		bool has_derivative = false;
		KernelBase() = default;
		KernelBase(const KernelParameter& p, const double* L, size_t R, const double* Rt, size_t C, bool same, bool deriv):
			params(p), has_derivative(deriv)
		{
			K = gaussian_kernel(p.length, L, R, Rt, C, same);
			const Mat delta = delta_kernel(L, R, Rt, C, same);
			const double m2 = sq(p.magnitude), n2 = sq(p.noise);
			for (size_t i = 0; i < K.a.size(); i++) K.a[i] = m2 * (K.a[i] + n2 * delta.a[i]); // :227
			if (deriv) dK = calculate_derivative(p, L, R, Rt, C, K, same);
		}
	};

	// kernel.h:285-294
	KernelParameter construct_purity_auxiliary_kernel_params(const KernelParameter& o)
	{
		KernelParameter r;
		r.magnitude = sq(o.magnitude) * std::sqrt(o.length[0] * o.length[1]);
		for (int d = 0; d < PhaseDim; d++) r.length[d] = sqrt2 * o.length[d];
		r.noise = 0.0;
		return r;
	}

	// kernel.h:301-332
	inline double cutoff_one(double pred_square, double abs_pred, double var)
	{
		if (pred_square >= sq(ConnectingPoint) * var) return 1.0;
		if (pred_square <= var) return 0.0;
		const double a = abs_pred / std::sqrt(var);
		const double c1 = ConnectingPoint - 1.0;
		return (3.0 * ConnectingPoint - 2.0 * a - 1.0) * ((a - 1) * (a - 1)) / (c1 * (c1 * c1));
	}

	double nan_() { return std::numeric_limits<double>::quiet_NaN(); }
	void fill_nan(double* p, size_t n)
	{
		for (size_t i = 0; i < n; i++) p[i] = nan_();
	}
} // namespace

// -------------------------------------------------------------------------------------------------------------
// TrainingKernel  (kernel.cpp:244-479)
// -------------------------------------------------------------------------------------------------------------
struct oracle_real_fit
{
	size_t N = 0;
	KernelParameter params;
	double theta[4];
	std::vector<double> X;
	bool has_err = false, has_avg = false, has_der = false;
	double s = 0; // RescaleFactor
	Vec label;    // rescaled
	KernelBase base;
	Mat W; // Inverse
	Vec v; // InvLbl
	std::array<Mat, 4> dW;
	std::array<Vec, 4> dv;
	gple_real_fit_scalars sc;
};

namespace
{
	KernelParameter unpack_real(const double* theta) // kernel.cpp:253-273
	{
		KernelParameter p;
		p.magnitude = theta[0];
		p.length[0] = theta[1];
		p.length[1] = theta[2];
		p.noise = theta[3];
		return p;
	}

	void real_fit(oracle_real_fit& f, const double* theta, const double* X, const double* y, int y_is_complex, size_t N,
		unsigned flags)
	{
		f.N = N;
		std::memcpy(f.theta, theta, sizeof(f.theta));
		f.params = unpack_real(theta);
		f.X.assign(X, X + 2 * N);
		f.has_err = flags & GPLE_CALC_ERROR;
		f.has_avg = flags & GPLE_CALC_AVERAGE;
		f.has_der = flags & GPLE_CALC_DERIVATIVE;
		const KernelParameter& p = f.params;
		const double* Xp = f.X.data();
		f.base = KernelBase(p, Xp, N, Xp, N, true, f.has_der);
		// :279-280
		double maxabs = 0.0;
		const size_t stride = y_is_complex ? 2 : 1;
		for (size_t i = 0; i < N; i++) maxabs = std::max(maxabs, std::abs(y[i * stride]));
		f.s = RescaleMaximum / maxabs;
		f.label.resize(N);
		for (size_t i = 0; i < N; i++) f.label[i] = y[i * stride] * f.s;
		// :281-283
		const LDLT<double> dec(f.base.K);
		f.W = identity<double>(N);
		dec.solve_inplace(f.W);
		{
			Mat rhs(N, 1);
			rhs.a = f.label;
			dec.solve_inplace(rhs);
			f.v = rhs.a;
		}
		gple_real_fit_scalars& sc = f.sc;
		fill_nan(&sc.rescale_factor, (sizeof(sc) - sizeof(int)) / sizeof(double));
		sc.info = dec.info;
		sc.rescale_factor = f.s;
		{
			// kernel.h:167-179
			double dot = 0.0;
			for (size_t i = 0; i < N; i++) dot += f.label[i] * f.v[i];
			const double within = dot / static_cast<double>(N);
			sc.magnitude = within < 0 ? std::sqrt(-within) : std::sqrt(within);
		}
		const Vec& v = f.v;
		const Mat& W = f.W;
		if (f.has_err) // :285
		{
			double e = 0.0;
			for (size_t i = 0; i < N; i++) e += sq(v[i] / W(i, i));
			sc.error = e;
		}
		KernelBase K1b;
		if (f.has_avg)
		{
			const double GlobalFactor = 2.0 * pi; // power<Dim>(2 pi), :291
			double vsum = 0.0;
			for (size_t i = 0; i < N; i++) vsum += v[i];
			const double lprod = p.length[0] * p.length[1];
			sc.population = GlobalFactor * sq(p.magnitude) * lprod * vsum / f.s; // :293
			for (int d = 0; d < PhaseDim; d++) // :308
			{
				double xv = 0.0;
				for (size_t i = 0; i < N; i++) xv += Xp[2 * i + d] * v[i];
				sc.first_order_average[d] = GlobalFactor * sq(p.magnitude) * lprod * xv / f.s;
			}
			K1b = KernelBase(construct_purity_auxiliary_kernel_params(p), Xp, N, Xp, N, true, f.has_der); // :313-324
			const double PurityGlobal = PurityFactor * pi;                                                // :330
			sc.purity = PurityGlobal * bilinear(v, K1b.K, v) / sq(f.s);                                     // :331
		}
		if (f.has_der)
		{
			// :337-364
			f.dW[0] = W;
			for (double& x : f.dW[0].a) x *= -2.0 / p.magnitude;
			for (int d = 0; d < PhaseDim; d++)
			{
				Mat t = gemm(gemm(W, f.base.dK[1 + d]), W);
				for (double& x : t.a) x = -x;
				f.dW[1 + d] = std::move(t);
			}
			{
				Mat scaled = W;
				for (double& x : scaled.a) x *= -2 * sq(p.magnitude) * p.noise;
				f.dW[3] = gemm(W, scaled); // result[iParam] *= (-2 sf^2 sn * Inverse), :358
			}
			for (int ip = 0; ip < 4; ip++) f.dv[ip] = matvec(f.dW[ip], f.label); // :365-379
			if (f.has_err)                                                        // :381-400
			{
				for (int ip = 0; ip < 4; ip++)
				{
					double t = 0.0;
					for (size_t i = 0; i < N; i++)
					{
						const double invd = W(i, i), diff = v[i] / invd;
						t += diff / invd * (f.dv[ip][i] - diff * f.dW[ip](i, i));
					}
					sc.error_derivative[ip] = 2.0 * t;
				}
			}
			if (f.has_avg)
			{
				// :401-435
				const double GlobalFactor = 2.0 * pi;
				const double ThisTimeFactor = GlobalFactor * sq(p.magnitude) * (p.length[0] * p.length[1]);
				double vsum = 0.0;
				for (size_t i = 0; i < N; i++) vsum += v[i];
				auto sum = [](const Vec& a) { double t = 0.0; for (double x : a) t += x; return t; };
				sc.population_derivative[0] = 0.0;
				for (int d = 0; d < PhaseDim; d++)
					sc.population_derivative[1 + d] = ThisTimeFactor * (vsum / p.length[d] + sum(f.dv[1 + d]));
				sc.population_derivative[3] = ThisTimeFactor * sum(f.dv[3]);
				for (double& d : sc.population_derivative) d /= f.s;
				// :436-477
				const double PurityGlobal = PurityFactor * pi;
				sc.purity_derivative[0] = 0.0;
				for (int d = 0; d < PhaseDim; d++)
				{
					Mat comb = K1b.K;
					for (size_t i = 0; i < comb.a.size(); i++)
						comb.a[i] = K1b.K.a[i] / p.length[d] + sqrt2 * K1b.dK[1 + d].a[i];
					double r = bilinear(v, comb, v) + 2.0 * bilinear(f.dv[1 + d], K1b.K, v);
					r *= PurityGlobal;
					sc.purity_derivative[1 + d] = r;
				}
				sc.purity_derivative[3] = 2.0 * PurityGlobal * bilinear(f.dv[3], K1b.K, v);
				for (double& d : sc.purity_derivative) d /= sq(f.s);
			}
		}
	}

	// PredictiveKernel (kernel.cpp:481-544)
	void real_predict(const oracle_real_fit& f, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* out)
	{
		const size_t N = f.N;
		const bool deriv = flags & GPLE_CALC_DERIVATIVE;
		const KernelParameter& p = f.params;
		const KernelBase kb(p, Xs, M, f.X.data(), N, false, deriv); // :488-493
		const Mat& Ks = kb.K;
		Vec mu = matvec(Ks, f.v); // :495
		Vec var(M);
		// :496-518 — one GEMV + dot per test point, like the reference's per-row loop
		const double self = sq(p.magnitude) * (1.0 + sq(p.noise) * 1.0); // KernelBase(params, col, col).get_kernel().value()
#pragma omp parallel
		{
			Vec row(N), tmp(N);
#pragma omp for schedule(static)
			for (size_t i = 0; i < M; i++)
			{
				for (size_t j = 0; j < N; j++) row[j] = Ks(i, j);
				// row * W  (1 x N times N x N), then * row^T
				for (size_t j = 0; j < N; j++)
				{
					const double* wj = f.W.col(j);
					double s = 0.0;
					for (size_t k = 0; k < N; k++) s += row[k] * wj[k];
					tmp[j] = s;
				}
				double q = 0.0;
				for (size_t j = 0; j < N; j++) q += tmp[j] * row[j];
				var[i] = self - q;
			}
		}
		Vec cut(M); // :519
		for (size_t i = 0; i < M; i++)
			cut[i] = mu[i] * cutoff_one(mu[i] * mu[i], std::abs(mu[i]), var[i]) / f.s;
		if (prediction) std::copy(mu.begin(), mu.end(), prediction);
		if (variance) std::copy(var.begin(), var.end(), variance);
		if (cutoff_prediction) std::copy(cut.begin(), cut.end(), cutoff_prediction);
		if (out)
		{
			fill_nan(&out->error, 9);
			if (labels)
			{
				Vec lab(M); // :521
				for (size_t i = 0; i < M; i++) lab[i] = labels[i] * f.s;
				double e = 0.0;
				for (size_t i = 0; i < M; i++) e += sq(mu[i] - lab[i]); // :522
				out->error = e;
				if (deriv) // :524-542
				{
					Vec diff(M);
					for (size_t i = 0; i < M; i++) diff[i] = cut[i] * f.s - lab[i];
					for (int ip = 0; ip < 4; ip++)
					{
						const Vec a = matvec(kb.dK[ip], f.v), b = matvec(Ks, f.dv[ip]);
						double t = 0.0;
						for (size_t i = 0; i < M; i++) t += diff[i] * (a[i] + b[i]);
						out->error_derivative[ip] = 2.0 * t;
					}
				}
			}
		}
	}
} // namespace

// -------------------------------------------------------------------------------------------------------------
// ComplexKernelBase / TrainingComplexKernel / PredictiveComplexKernel (complex_kernel.cpp)
// -------------------------------------------------------------------------------------------------------------
namespace
{
	struct ComplexParameter // complex_kernel.h:28
	{
		double magnitude;
		double sub_magnitude[2];
		double sub_length[2][PhaseDim];
		double noise;
	};
	ComplexParameter unpack_complex(const double* t) // complex_kernel.cpp:230-256
	{
		ComplexParameter p;
		p.magnitude = t[0];
		for (int k = 0; k < 2; k++)
		{
			p.sub_magnitude[k] = t[1 + 3 * k];
			p.sub_length[k][0] = t[2 + 3 * k];
			p.sub_length[k][1] = t[3 + 3 * k];
		}
		p.noise = t[7];
		return p;
	}

	// complex_kernel.cpp:206-219
	KernelParameter construct_purity_auxiliary_mixed_kernel_params(const KernelParameter& a, const KernelParameter& b)
	{
		KernelParameter r;
		double prod = 1.0;
		for (int d = 0; d < PhaseDim; d++) prod *= 0.5 * (sq(1.0 / a.length[d]) + sq(1.0 / b.length[d]));
		r.magnitude = a.magnitude * b.magnitude / std::sqrt(std::sqrt(prod));
		for (int d = 0; d < PhaseDim; d++) r.length[d] = std::sqrt(sq(a.length[d]) + sq(b.length[d]));
		r.noise = 0.0;
		return r;
	}

	// complex_kernel.cpp:134-200
	struct ComplexKernelBase
	{
		ComplexParameter params;
		KernelParameter RealParams, ImagParams, CorrParams;
		KernelBase RealKernel, ImagKernel, CorrKernel;
		Mat K;
		CMat Kt; // pseudo kernel
		std::array<Mat, 8> dK;
		std::array<CMat, 8> dKt;
		bool has_derivative = false;
		ComplexKernelBase() = default;
		ComplexKernelBase(const ComplexParameter& p, const double* L, size_t R, const double* Rt, size_t C, bool same,
			bool deriv):
			params(p), has_derivative(deriv)
		{
			RealParams = {p.sub_magnitude[0], {p.sub_length[0][0], p.sub_length[0][1]}, 0.0}; // :142
			ImagParams = {p.sub_magnitude[1], {p.sub_length[1][0], p.sub_length[1][1]}, 0.0}; // :143
			{
				// :144-157
				double prod = 1.0, ss[PhaseDim];
				for (int d = 0; d < PhaseDim; d++)
				{
					ss[d] = sq(RealParams.length[d]) + sq(ImagParams.length[d]);
					prod *= 2.0 * RealParams.length[d] * ImagParams.length[d] / ss[d];
				}
				CorrParams.magnitude = std::sqrt(RealParams.magnitude * ImagParams.magnitude * prod);
				for (int d = 0; d < PhaseDim; d++) CorrParams.length[d] = std::sqrt(ss[d] / 2.0);
				CorrParams.noise = 0.0;
			}
			// the sub-kernels see the member copies of the features (different pointers) — with zero noise this only
			// matters through exact duplicates, whose delta entries are multiplied by noise^2 = 0 (:160-162)
			RealKernel = KernelBase(RealParams, L, R, Rt, C, same, deriv);
			ImagKernel = KernelBase(ImagParams, L, R, Rt, C, same, deriv);
			CorrKernel = KernelBase(CorrParams, L, R, Rt, C, same, deriv);
			const Mat delta = delta_kernel(L, R, Rt, C, same);
			const double m2 = sq(p.magnitude), n2 = sq(p.noise);
			K = Mat(R, C);
			Kt = CMat(R, C);
			for (size_t i = 0; i < K.a.size(); i++)
			{
				K.a[i] = m2 * (RealKernel.K.a[i] + ImagKernel.K.a[i] + n2 * delta.a[i]);                 // :163
				Kt.a[i] = m2 * cd(RealKernel.K.a[i] - ImagKernel.K.a[i], 2.0 * CorrKernel.K.a[i]);       // :164
			}
			if (deriv)
			{
				calc_derivative(R, C, same);
				calc_pseudo_derivative(R, C);
			}
		}
		void calc_derivative(size_t R, size_t C, bool same) // :20-59
		{
			dK[0] = K;
			for (double& x : dK[0].a) x *= 2.0 / params.magnitude;
			for (int i = 0; i < 3; i++) dK[1 + i] = RealKernel.dK[i];
			for (int i = 0; i < 3; i++) dK[4 + i] = ImagKernel.dK[i];
			dK[7] = Mat(R, C);
			if (same)
				for (size_t i = 0; i < R; i++) dK[7](i, i) = 2.0 * params.noise * 1.0; // :51
		}
		void calc_pseudo_derivative(size_t R, size_t C) // :74-132
		{
			const cd I(0.0, 1.0);
			const Mat& KC = CorrKernel.K;
			const size_t n = R * C;
			dKt[0] = Kt;
			for (cd& z : dKt[0].a) z *= 2.0 / params.magnitude; // :94
			const KernelParameter* sub[2] = {&RealParams, &ImagParams};
			const KernelBase* subk[2] = {&RealKernel, &ImagKernel};
			for (int k = 0; k < 2; k++)
			{
				const double sign = k == 0 ? 1.0 : -1.0; // imaginary kernel derivatives enter with a minus sign
				const int base = 1 + 3 * k;
				dKt[base] = CMat(R, C);
				for (size_t i = 0; i < n; i++) // :101, :117
					dKt[base].a[i] = sign * subk[k]->dK[0].a[i] + (2.0 * I) / sub[k]->magnitude * KC.a[i];
				for (int d = 0; d < PhaseDim; d++) // :104-109, :120-125
				{
					const double l = sub[k]->length[d], lc = CorrParams.length[d];
					dKt[base + 1 + d] = CMat(R, C);
					for (size_t i = 0; i < n; i++)
						dKt[base + 1 + d].a[i] = sign * subk[k]->dK[1 + d].a[i] + (2.0 * I) * (1.0 / l - l / sq(lc)) * KC.a[i]
							+ (1.0 * I) * l / lc * CorrKernel.dK[1 + d].a[i];
				}
			}
			dKt[7] = CMat(R, C); // :129
		}
	};
} // namespace

struct oracle_complex_fit
{
	size_t N = 0;
	ComplexParameter params;
	double theta[8];
	std::vector<double> X;
	bool has_err = false, has_avg = false, has_der = false;
	double s = 0;
	CVec label;
	ComplexKernelBase base;
	CMat A; // KernelInversePseudoConjugate
	CMat P, Q;
	CVec v;
	std::array<CMat, 8> dP, dQ;
	std::array<CVec, 8> dv;
	gple_complex_fit_scalars sc;
};

namespace
{
	template <typename S>
	Matrix<S> scaled(const Matrix<S>& A, S f)
	{
		Matrix<S> R = A;
		for (S& x : R.a) x *= f;
		return R;
	}
	CMat to_complex(const Mat& A)
	{
		CMat R(A.r, A.c);
		for (size_t i = 0; i < A.a.size(); i++) R.a[i] = A.a[i];
		return R;
	}
	template <typename S>
	void add_inplace(Matrix<S>& A, const Matrix<S>& B)
	{
		for (size_t i = 0; i < A.a.size(); i++) A.a[i] += B.a[i];
	}

	void complex_fit(oracle_complex_fit& f, const double* theta, const double* X, const double* y, size_t N,
		unsigned flags)
	{
		f.N = N;
		std::memcpy(f.theta, theta, sizeof(f.theta));
		f.params = unpack_complex(theta);
		f.X.assign(X, X + 2 * N);
		f.has_err = flags & GPLE_CALC_ERROR;
		f.has_avg = flags & GPLE_CALC_AVERAGE;
		f.has_der = flags & GPLE_CALC_DERIVATIVE;
		const ComplexParameter& p = f.params;
		const double* Xp = f.X.data();
		f.base = ComplexKernelBase(p, Xp, N, Xp, N, true, f.has_der);
		const Mat& K = f.base.K;
		const CMat& Kt = f.base.Kt;
		// :262-263
		double maxabs = 0.0;
		for (size_t i = 0; i < N; i++) maxabs = std::max(maxabs, std::abs(cd(y[2 * i], y[2 * i + 1])));
		f.s = RescaleMaximum / maxabs;
		f.label.resize(N);
		for (size_t i = 0; i < N; i++) f.label[i] = cd(y[2 * i], y[2 * i + 1]) * f.s;
		// :264-268
		const LDLT<cd> dec(to_complex(K));
		f.A = conjugate(Kt);
		dec.solve_inplace(f.A);
		int info = dec.info;
		{
			CMat S = gemm(Kt, f.A);
			for (size_t i = 0; i < S.a.size(); i++) S.a[i] = K.a[i] - S.a[i];
			const LDLT<cd> decS(S); // uses the lower triangle only == selfadjointView<Lower>().ldlt()
			if (info == 0) info = decS.info;
			f.P = identity<cd>(N);
			decS.solve_inplace(f.P);
			for (size_t j = 0; j < N; j++) // .selfadjointView<Lower>() of the solution
				for (size_t i = 0; i < j; i++) f.P(i, j) = std::conj(f.P(j, i));
		}
		f.Q = gemm(f.A, f.P);
		for (cd& z : f.Q.a) z = -z;
		{
			const CVec Py = matvec(f.P, f.label), Qy = matvec(f.Q, f.label);
			f.v.resize(N);
			for (size_t i = 0; i < N; i++) f.v[i] = Py[i] + std::conj(Qy[i]);
		}
		gple_complex_fit_scalars& sc = f.sc;
		fill_nan(&sc.rescale_factor, (sizeof(sc) - sizeof(int)) / sizeof(double));
		sc.info = info;
		sc.rescale_factor = f.s;
		{
			// complex_kernel.h:192-204 : Label.dot(v).real() / N  (dot conjugates its first argument)
			cd dot(0);
			for (size_t i = 0; i < N; i++) dot += std::conj(f.label[i]) * f.v[i];
			const double within = dot.real() / static_cast<double>(N);
			sc.magnitude = within < 0 ? std::sqrt(-within) : std::sqrt(within);
		}
		const CMat &P = f.P, &Q = f.Q;
		const CVec& v = f.v;
		if (f.has_err) // :270-286
		{
			double e = 0.0;
			for (size_t i = 0; i < N; i++)
			{
				const cd pd = P(i, i), qd = Q(i, i);
				const double den = sq(pd.real()) - std::norm(qd);
				const cd diff = (pd * v[i] - std::conj(qd * v[i])) / den;
				e += std::norm(diff);
			}
			sc.error = e;
		}
		KernelBase KRp, KIp, KCp, KRC, KIC;
		KernelParameter pRC{}, pIC{};
		if (f.has_avg) // :287-377
		{
			const ComplexKernelBase& b = f.base;
			pRC = construct_purity_auxiliary_mixed_kernel_params(b.RealParams, b.CorrParams);
			pIC = construct_purity_auxiliary_mixed_kernel_params(b.ImagParams, b.CorrParams);
			KRp = KernelBase(construct_purity_auxiliary_kernel_params(b.RealParams), Xp, N, Xp, N, true, f.has_der);
			KIp = KernelBase(construct_purity_auxiliary_kernel_params(b.ImagParams), Xp, N, Xp, N, true, f.has_der);
			KCp = KernelBase(construct_purity_auxiliary_kernel_params(b.CorrParams), Xp, N, Xp, N, true, f.has_der);
			KRC = KernelBase(pRC, Xp, N, Xp, N, true, f.has_der);
			KIC = KernelBase(pIC, Xp, N, Xp, N, true, f.has_der);
			const double GlobalFactor = PurityFactor * 2.0 * pi;                       // :369
			const double ThisTimeFactor = GlobalFactor * sq(sq(p.magnitude));           // :370
			Mat K1(N, N);
			CMat K2(N, N);
			for (size_t i = 0; i < K1.a.size(); i++)
			{
				K1.a[i] = KRp.K.a[i] + KIp.K.a[i] + 2.0 * KCp.K.a[i];                                   // :371
				K2.a[i] = cd(KRp.K.a[i] - KIp.K.a[i], -2.0 * (KRC.K.a[i] + KIC.K.a[i]));               // :372
			}
			CVec vc(N);
			for (size_t i = 0; i < N; i++) vc[i] = std::conj(v[i]);
			sc.purity = ThisTimeFactor * (bilinear(vc, K1, v).real() + bilinear(v, K2, v).real()) / sq(f.s); // :373
		}
		if (f.has_der)
		{
			const ComplexKernelBase& b = f.base;
			const CMat Qh = adjoint(Q);
			for (int ip = 0; ip < 8; ip++) // :379-401
			{
				const CMat D = to_complex(b.dK[ip]);
				const CMat& Dt = b.dKt[ip];
				CMat r = gemm(gemm(P, D), P);
				add_inplace(r, gemm(gemm(Qh, D), Q));
				add_inplace(r, gemm(gemm(P, Dt), Q));
				add_inplace(r, gemm(gemm(Qh, conjugate(Dt)), P));
				const CMat rh = adjoint(r);
				for (size_t i = 0; i < r.a.size(); i++) r.a[i] = -(r.a[i] + rh.a[i]) / 2.0;
				f.dP[ip] = std::move(r);
			}
			for (int ip = 0; ip < 8; ip++) // :402-425
			{
				const CMat D = to_complex(b.dK[ip]);
				CMat rhs = gemm(D, Q);
				add_inplace(rhs, gemm(conjugate(b.dKt[ip]), P));
				dec.solve_inplace(rhs);
				const CMat AdP = gemm(f.A, f.dP[ip]);
				for (size_t i = 0; i < rhs.a.size(); i++) rhs.a[i] = -rhs.a[i] - AdP.a[i];
				f.dQ[ip] = std::move(rhs);
			}
			for (int ip = 0; ip < 8; ip++) // :426-442
			{
				const CVec a = matvec(f.dP[ip], f.label), c = matvec(f.dQ[ip], f.label);
				f.dv[ip].resize(N);
				for (size_t i = 0; i < N; i++) f.dv[ip][i] = a[i] + std::conj(c[i]);
			}
			if (f.has_err) // :444-474
			{
				for (int ip = 0; ip < 8; ip++)
				{
					double total = 0.0;
					for (size_t i = 0; i < N; i++)
					{
						const cd pd = P(i, i), qd = Q(i, i);
						const double square_diff = sq(pd.real()) - std::norm(qd);
						const cd diff = (pd * v[i] - std::conj(qd * v[i])) / square_diff;
						const cd pdd = f.dP[ip](i, i), qdd = f.dQ[ip](i, i), vd = f.dv[ip][i];
						const cd numerator_deriv = std::conj(diff) * (pdd * v[i] + pd * vd - std::conj(qdd * v[i] + qd * vd));
						const cd denominator_deriv = -2.0 * std::norm(diff) * (pd * pdd - (std::conj(qd) * qdd).real());
						total += ((numerator_deriv + denominator_deriv) / square_diff).real();
					}
					sc.error_derivative[ip] = 2.0 * total;
				}
			}
			if (f.has_avg) // :475-590
			{
				const double GlobalFactor = PurityFactor * 2.0 * pi; // :497 (no magnitude^4: reference quirk, :584)
				const KernelParameter &RP = b.RealParams, &IP = b.ImagParams, &CP = b.CorrParams;
				double ROverC[2], IOverC[2], ROverCSquare[2], IOverCSquare[2], ROverRC[2], ROverIC[2], IOverRC[2], IOverIC[2];
				for (int d = 0; d < PhaseDim; d++)
				{
					ROverC[d] = RP.length[d] / CP.length[d];
					IOverC[d] = IP.length[d] / CP.length[d];
					ROverCSquare[d] = RP.length[d] / sq(CP.length[d]);
					IOverCSquare[d] = IP.length[d] / sq(CP.length[d]);
					ROverRC[d] = RP.length[d] / pRC.length[d];
					ROverIC[d] = RP.length[d] / pIC.length[d];
					IOverRC[d] = IP.length[d] / pRC.length[d];
					IOverIC[d] = IP.length[d] / pIC.length[d];
				}
				const size_t n2 = N * N;
				std::array<Mat, 8> dKRp, dKIp, dKCp, dKRC, dKIC;
				for (int ip = 0; ip < 8; ip++) dKRp[ip] = dKIp[ip] = dKCp[ip] = dKRC[ip] = dKIC[ip] = Mat(N, N);
				// real magnitude (:525-529)
				for (size_t i = 0; i < n2; i++)
				{
					dKRp[1].a[i] = 4.0 / RP.magnitude * KRp.K.a[i];
					dKCp[1].a[i] = 2.0 / RP.magnitude * KCp.K.a[i];
					dKRC[1].a[i] = 3.0 / RP.magnitude * KRC.K.a[i];
					dKIC[1].a[i] = 1.0 / RP.magnitude * KIC.K.a[i];
				}
				for (int d = 0; d < PhaseDim; d++) // :532-542
				{
					const int ip = 2 + d;
					for (size_t i = 0; i < n2; i++)
					{
						dKRp[ip].a[i] = KRp.K.a[i] / RP.length[d] + sqrt2 * KRp.dK[1 + d].a[i];
						dKCp[ip].a[i] = (2.0 / RP.length[d] - 3.0 * ROverCSquare[d] / 2.0) * KCp.K.a[i]
							+ 1.0 / sqrt2 * ROverC[d] * KCp.dK[1 + d].a[i];
						dKRC[ip].a[i] = (2.0 / RP.length[d] - ROverCSquare[d] / 2.0) * KRC.K.a[i]
							+ 1.5 * ROverRC[d] * (KRC.dK[1 + d].a[i] - KRC.K.a[i] / pRC.length[d]);
						dKIC[ip].a[i] = (1.0 / RP.length[d] - ROverCSquare[d] / 2.0) * KIC.K.a[i]
							+ ROverIC[d] / 2.0 * (KIC.dK[1 + d].a[i] - KIC.K.a[i] / pIC.length[d]);
					}
				}
				// imaginary magnitude (:548-552)
				for (size_t i = 0; i < n2; i++)
				{
					dKIp[4].a[i] = 4.0 / IP.magnitude * KIp.K.a[i];
					dKCp[4].a[i] = 2.0 / IP.magnitude * KCp.K.a[i];
					dKRC[4].a[i] = 1.0 / IP.magnitude * KRC.K.a[i];
					dKIC[4].a[i] = 3.0 / IP.magnitude * KIC.K.a[i];
				}
				for (int d = 0; d < PhaseDim; d++) // :555-565
				{
					const int ip = 5 + d;
					for (size_t i = 0; i < n2; i++)
					{
						dKIp[ip].a[i] = KIp.K.a[i] / IP.length[d] + sqrt2 * KIp.dK[1 + d].a[i];
						dKCp[ip].a[i] = (2.0 / IP.length[d] - 3.0 * IOverCSquare[d] / 2.0) * KCp.K.a[i]
							+ 1.0 / sqrt2 * IOverC[d] * KCp.dK[1 + d].a[i];
						dKRC[ip].a[i] = (1.0 / IP.length[d] - IOverCSquare[d] / 2.0) * KRC.K.a[i]
							+ IOverRC[d] / 2.0 * (KRC.dK[1 + d].a[i] - KRC.K.a[i] / pRC.length[d]);
						dKIC[ip].a[i] = (2.0 / IP.length[d] - IOverCSquare[d] / 2.0) * KIC.K.a[i]
							+ 1.5 * IOverIC[d] * (KIC.dK[1 + d].a[i] - KIC.K.a[i] / pIC.length[d]);
					}
				}
				Mat K1(N, N);
				CMat K2(N, N);
				for (size_t i = 0; i < n2; i++)
				{
					K1.a[i] = KRp.K.a[i] + KIp.K.a[i] + 2.0 * KCp.K.a[i];
					K2.a[i] = cd(KRp.K.a[i] - KIp.K.a[i], -2.0 * (KRC.K.a[i] + KIC.K.a[i]));
				}
				CVec vc(N);
				for (size_t i = 0; i < N; i++) vc[i] = std::conj(v[i]);
				for (int ip = 0; ip < 8; ip++) // :576-585
				{
					Mat K1d(N, N);
					CMat K2d(N, N);
					for (size_t i = 0; i < n2; i++)
					{
						K1d.a[i] = dKRp[ip].a[i] + dKIp[ip].a[i] + 2.0 * dKCp[ip].a[i];
						K2d.a[i] = cd(dKRp[ip].a[i] - dKIp[ip].a[i], -2.0 * (dKRC[ip].a[i] + dKIC[ip].a[i]));
					}
					double r = 2.0 * bilinear(vc, K1, f.dv[ip]).real() + bilinear(vc, K1d, v).real()
						+ 2.0 * bilinear(v, K2, f.dv[ip]).real() + bilinear(v, K2d, v).real();
					r *= GlobalFactor / sq(f.s);
					sc.purity_derivative[ip] = r;
				}
			}
		}
	}

	// PredictiveComplexKernel (complex_kernel.cpp:594-670)
	void complex_predict(const oracle_complex_fit& f, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* out)
	{
		const size_t N = f.N;
		const bool deriv = flags & GPLE_CALC_DERIVATIVE;
		const ComplexParameter& p = f.params;
		const ComplexKernelBase kb(p, Xs, M, f.X.data(), N, false, deriv);
		const CVec& v = f.v;
		CVec vc(N);
		for (size_t i = 0; i < N; i++) vc[i] = std::conj(v[i]);
		CVec mu(M); // :608
		{
			const CVec a = matvec(kb.K, v), b = matvec(kb.Kt, vc);
			for (size_t i = 0; i < M; i++) mu[i] = a[i] + b[i];
		}
		// ComplexKernelBase(KernelParams, col, col, false).get_kernel().value(), :632
		const double self = sq(p.magnitude) * (sq(p.sub_magnitude[0]) * (1.0 + 0.0) + sq(p.sub_magnitude[1]) * (1.0 + 0.0)
			+ sq(p.noise) * 1.0);
		Vec var(M);
		const CMat Pc = conjugate(f.P), Qc = conjugate(f.Q);
#pragma omp parallel
		{
			CVec pr(N), pcol(N);
			Vec kr(N);
#pragma omp for schedule(static)
			for (size_t i = 0; i < M; i++) // :620-639
			{
				for (size_t j = 0; j < N; j++)
				{
					kr[j] = kb.K(i, j);
					pr[j] = kb.Kt(i, j);
					pcol[j] = std::conj(pr[j]); // pseudo_row.adjoint()
				}
				cd t1(0), t2(0), t3(0), t4(0);
				for (size_t j = 0; j < N; j++)
				{
					const cd *Pj = f.P.col(j), *Pcj = Pc.col(j), *Qj = f.Q.col(j), *Qcj = Qc.col(j);
					cd a1(0), a2(0), a3(0), a4(0);
					for (size_t k = 0; k < N; k++)
					{
						a1 += kr[k] * Pj[k];  // kernel_row * P
						a2 += pr[k] * Pcj[k]; // pseudo_row * P.conjugate()
						a3 += pr[k] * Qj[k];  // pseudo_row * Q
						a4 += kr[k] * Qcj[k]; // kernel_row * Q.conjugate()
					}
					t1 += a1 * kr[j];
					t2 += a2 * pcol[j];
					t3 += a3 * kr[j];
					t4 += a4 * pcol[j];
				}
				var[i] = (cd(self) - t1 - t2 - t3 - t4).real();
			}
		}
		CVec cut(M); // :643
		for (size_t i = 0; i < M; i++)
			cut[i] = mu[i] * cutoff_one(std::norm(mu[i]), std::abs(mu[i]), var[i]) / f.s;
		if (prediction)
			for (size_t i = 0; i < M; i++) prediction[2 * i] = mu[i].real(), prediction[2 * i + 1] = mu[i].imag();
		if (variance) std::copy(var.begin(), var.end(), variance);
		if (cutoff_prediction)
			for (size_t i = 0; i < M; i++) cutoff_prediction[2 * i] = cut[i].real(), cutoff_prediction[2 * i + 1] = cut[i].imag();
		if (out)
		{
			fill_nan(&out->error, 9);
			if (labels)
			{
				CVec lab(M); // :645
				for (size_t i = 0; i < M; i++) lab[i] = cd(labels[2 * i], labels[2 * i + 1]) * f.s;
				double e = 0.0;
				for (size_t i = 0; i < M; i++) e += std::norm(mu[i] - lab[i]); // :646
				out->error = e;
				if (deriv) // :648-668
				{
					CVec diff(M);
					for (size_t i = 0; i < M; i++) diff[i] = cut[i] * f.s - lab[i];
					for (int ip = 0; ip < 8; ip++)
					{
						CVec dvc(N);
						for (size_t i = 0; i < N; i++) dvc[i] = std::conj(f.dv[ip][i]);
						const CVec a = matvec(kb.dK[ip], v), b = matvec(kb.K, f.dv[ip]), c = matvec(kb.dKt[ip], vc),
								   d = matvec(kb.Kt, dvc);
						cd t(0);
						for (size_t i = 0; i < M; i++) t += std::conj(diff[i]) * (a[i] + b[i] + c[i] + d[i]);
						out->error_derivative[ip] = 2.0 * t.real();
					}
				}
			}
		}
	}

	// opt.cpp:420-431
	inline void make_normal(double& d)
	{
		if (std::isnan(d) || std::isinf(d)) d = std::numeric_limits<double>::max();
	}
} // namespace

// -------------------------------------------------------------------------------------------------------------
// C interface
// -------------------------------------------------------------------------------------------------------------
extern "C"
{
	int oracle_set_num_threads(int n)
	{
#ifdef _OPENMP
		if (n > 0) omp_set_num_threads(n);
#else
		(void)n;
#endif
		return GPLE_OK;
	}

	int oracle_num_threads(void)
	{
#ifdef _OPENMP
		return omp_get_max_threads();
#else
		return 1;
#endif
	}

	int oracle_real_gram(const double theta[4], const double* left, size_t R, const double* right, size_t C,
		int same_features, double* K, double* dK)
	{
		if (!theta || !left || !right || !K) return GPLE_ERR_BAD_ARG;
		const KernelBase kb(unpack_real(theta), left, R, right, C, same_features != 0, dK != nullptr);
		std::copy(kb.K.a.begin(), kb.K.a.end(), K);
		if (dK)
			for (int ip = 0; ip < 4; ip++) std::copy(kb.dK[ip].a.begin(), kb.dK[ip].a.end(), dK + ip * R * C);
		return GPLE_OK;
	}

	// ComplexKernelBase as a whole (complex_kernel.cpp:20-200): K, K~ and the 8 + 8 derivative matrices
	int oracle_complex_gram(const double theta[8], const double* left, size_t R, const double* right, size_t C, int same_features, double* K,
		double* Kt, double* dK, double* dKt)
	{
		if (!theta || !left || !right || !K) return GPLE_ERR_BAD_ARG;
		const ComplexKernelBase kb(unpack_complex(theta), left, R, right, C, same_features != 0, dK != nullptr || dKt != nullptr);
		std::copy(kb.K.a.begin(), kb.K.a.end(), K);
		const size_t rc = R * C;
		if (Kt)
			for (size_t i = 0; i < rc; i++) Kt[2 * i] = kb.Kt.a[i].real(), Kt[2 * i + 1] = kb.Kt.a[i].imag();
		for (int ip = 0; ip < 8; ip++)
		{
			if (dK) std::copy(kb.dK[ip].a.begin(), kb.dK[ip].a.end(), dK + ip * rc);
			if (dKt)
				for (size_t i = 0; i < rc; i++) dKt[2 * (ip * rc + i)] = kb.dKt[ip].a[i].real(), dKt[2 * (ip * rc + i) + 1] = kb.dKt[ip].a[i].imag();
		}
		return GPLE_OK;
	}

	int oracle_cutoff_factor(const double* prediction, int is_complex, const double* variance, size_t M, double* factor)
	{
		if (!prediction || !variance || !factor) return GPLE_ERR_BAD_ARG;
		for (size_t i = 0; i < M; i++)
		{
			if (is_complex)
			{
				const cd z(prediction[2 * i], prediction[2 * i + 1]);
				factor[i] = cutoff_one(std::norm(z), std::abs(z), variance[i]);
			}
			else
			{
				factor[i] = cutoff_one(prediction[i] * prediction[i], std::abs(prediction[i]), variance[i]);
			}
		}
		return GPLE_OK;
	}

	int oracle_real_fit_create(const double theta[4], const double* X, const double* y, int y_is_complex, size_t N,
		unsigned flags, gple_real_fit_scalars* scalars, oracle_real_fit** out)
	{
		if (!theta || !X || !y || !out || N == 0) return GPLE_ERR_BAD_ARG;
		std::unique_ptr<oracle_real_fit> f(new oracle_real_fit);
		real_fit(*f, theta, X, y, y_is_complex, N, flags);
		if (scalars) *scalars = f->sc;
		*out = f.release();
		return GPLE_OK;
	}
	int oracle_real_fit_release(oracle_real_fit* fit)
	{
		delete fit;
		return GPLE_OK;
	}
	int oracle_real_fit_get(oracle_real_fit* f, gple_real_array which, double* dst)
	{
		if (!f || !dst) return GPLE_ERR_BAD_ARG;
		const size_t N = f->N;
		switch (which)
		{
		case GPLE_R_KERNEL: std::copy(f->base.K.a.begin(), f->base.K.a.end(), dst); break;
		case GPLE_R_INVERSE: std::copy(f->W.a.begin(), f->W.a.end(), dst); break;
		case GPLE_R_INVLBL: std::copy(f->v.begin(), f->v.end(), dst); break;
		case GPLE_R_INVLBL_DERIV:
			if (!f->has_der) return GPLE_ERR_STATE;
			for (int ip = 0; ip < 4; ip++) std::copy(f->dv[ip].begin(), f->dv[ip].end(), dst + ip * N);
			break;
		case GPLE_R_LABEL: std::copy(f->label.begin(), f->label.end(), dst); break;
		case GPLE_R_INVERSE_DIAG:
			for (size_t i = 0; i < N; i++) dst[i] = f->W(i, i);
			break;
		default: return GPLE_ERR_BAD_ARG;
		}
		return GPLE_OK;
	}
	int oracle_real_predict(const oracle_real_fit* fit, const double* Xs, size_t M, unsigned flags, const double* labels,
		double* prediction, double* variance, double* cutoff_prediction, gple_predict_scalars* scalars)
	{
		if (!fit || (!Xs && M > 0)) return GPLE_ERR_BAD_ARG;
		if ((flags & GPLE_CALC_DERIVATIVE) && labels && !fit->has_der) return GPLE_ERR_STATE;
		real_predict(*fit, Xs, M, flags, labels, prediction, variance, cutoff_prediction, scalars);
		return GPLE_OK;
	}

	int oracle_complex_fit_create(const double theta[8], const double* X, const double* y, size_t N, unsigned flags,
		gple_complex_fit_scalars* scalars, oracle_complex_fit** out)
	{
		if (!theta || !X || !y || !out || N == 0) return GPLE_ERR_BAD_ARG;
		std::unique_ptr<oracle_complex_fit> f(new oracle_complex_fit);
		complex_fit(*f, theta, X, y, N, flags);
		if (scalars) *scalars = f->sc;
		*out = f.release();
		return GPLE_OK;
	}
	int oracle_complex_fit_release(oracle_complex_fit* fit)
	{
		delete fit;
		return GPLE_OK;
	}
	int oracle_complex_fit_get(oracle_complex_fit* f, gple_complex_array which, double* dst)
	{
		if (!f || !dst) return GPLE_ERR_BAD_ARG;
		const size_t N = f->N;
		auto put = [&](const cd* src, size_t n, double* d) {
			for (size_t i = 0; i < n; i++) d[2 * i] = src[i].real(), d[2 * i + 1] = src[i].imag();
		};
		switch (which)
		{
		case GPLE_C_KERNEL: std::copy(f->base.K.a.begin(), f->base.K.a.end(), dst); break;
		case GPLE_C_PSEUDO: put(f->base.Kt.a.data(), N * N, dst); break;
		case GPLE_C_UPPER_LEFT: put(f->P.a.data(), N * N, dst); break;
		case GPLE_C_LOWER_LEFT: put(f->Q.a.data(), N * N, dst); break;
		case GPLE_C_INVLBL: put(f->v.data(), N, dst); break;
		case GPLE_C_INVLBL_DERIV:
			if (!f->has_der) return GPLE_ERR_STATE;
			for (int ip = 0; ip < 8; ip++) put(f->dv[ip].data(), N, dst + 2 * ip * N);
			break;
		case GPLE_C_LABEL: put(f->label.data(), N, dst); break;
		default: return GPLE_ERR_BAD_ARG;
		}
		return GPLE_OK;
	}
	int oracle_complex_predict(const oracle_complex_fit* fit, const double* Xs, size_t M, unsigned flags,
		const double* labels, double* prediction, double* variance, double* cutoff_prediction,
		gple_predict_scalars* scalars)
	{
		if (!fit || (!Xs && M > 0)) return GPLE_ERR_BAD_ARG;
		if ((flags & GPLE_CALC_DERIVATIVE) && labels && !fit->has_der) return GPLE_ERR_STATE;
		complex_predict(*fit, Xs, M, flags, labels, prediction, variance, cutoff_prediction, scalars);
		return GPLE_OK;
	}

	// loose_function (opt.cpp:441-482)
	int oracle_loose_function(const double* x, size_t n, const double* X, const double* y, size_t N,
		const double* X_extra, const double* y_extra, size_t M_extra, double* value, double* grad)
	{
		if (!x || !X || !y || !value || (n != 4 && n != 8)) return GPLE_ERR_BAD_ARG;
		const unsigned flags = GPLE_CALC_ERROR | (grad ? GPLE_CALC_DERIVATIVE : 0u);
		gple_predict_scalars ps;
		double result = 0.0;
		if (n == 4)
		{
			oracle_real_fit f;
			real_fit(f, x, X, y, 1, N, flags);
			Vec lab(M_extra);
			for (size_t i = 0; i < M_extra; i++) lab[i] = y_extra[2 * i]; // ExtraTrainingLabel.real(), :451
			real_predict(f, X_extra, M_extra, flags, lab.data(), nullptr, nullptr, nullptr, &ps);
			// an empty extra set contributes the empty sum, 0 (Eigen's .sum() over zero rows), not "no label"
			result = f.sc.error + (M_extra ? ps.error : 0.0);
			if (grad)
				for (int i = 0; i < 4; i++) grad[i] = f.sc.error_derivative[i] + (M_extra ? ps.error_derivative[i] : 0.0);
		}
		else
		{
			oracle_complex_fit f;
			complex_fit(f, x, X, y, N, flags);
			complex_predict(f, X_extra, M_extra, flags, y_extra, nullptr, nullptr, nullptr, &ps);
			result = f.sc.error + (M_extra ? ps.error : 0.0);
			if (grad)
				for (int i = 0; i < 8; i++) grad[i] = f.sc.error_derivative[i] + (M_extra ? ps.error_derivative[i] : 0.0);
		}
		make_normal(result);
		if (grad)
			for (size_t i = 0; i < n; i++) make_normal(grad[i]);
		*value = result;
		return GPLE_OK;
	}

	// negative_log_marginal_likelihood (test/gpr.cpp:499-532) with kernels {Diag, GaussianARD(diagonal weights)}
	// Shogun 6.1.4 GaussianARDKernel with vector weights a: k = exp(-|a o (x - x')|^2 / 2)   (test/gpr.cpp:356-367;
	// weights are inverse lengths, test/gpr.cpp:167-173).  Shogun is absent: restated from that formula, parity unpinned.
	// The default build carries the lower-triangular weight matrix W = [[a, 0], [c, b]] (test/gpr.cpp:313-321, hyper-parameter
	// order a, c, b): k = exp(-|W^T (x - x')|^2 / 2); x5 = (w_d, w_g, a, c, b), the NOCROSS build is c = 0.
	static void nlml_params(const double* x, size_t n, double x5[5])
	{
		x5[0] = x[0], x5[1] = x[1], x5[2] = x[2];
		x5[3] = n == 5 ? x[3] : 0.0;
		x5[4] = n == 5 ? x[4] : x[3];
	}
	static Mat nlml_gram(const double* x, const double* L, size_t R, const double* Rt, size_t C, bool training)
	{
		Mat K(R, C);
#pragma omp parallel for schedule(static)
		for (size_t j = 0; j < C; j++)
			for (size_t i = 0; i < R; i++)
			{
				const double e0 = L[2 * i] - Rt[2 * j], e1 = L[2 * i + 1] - Rt[2 * j + 1];
				const double d0 = x[2] * e0 + x[3] * e1, d1 = x[4] * e1;
				double k = x[1] * x[1] * std::exp(-(d0 * d0 + d1 * d1) / 2.0);
				if (training && i == j) k += x[0] * x[0]; // DiagKernel only on the training set (test/gpr.cpp:384-388)
				K(i, j) = k;
			}
		return K;
	}
	struct LLT // Eigen::LLT lower
	{
		Mat L;
		int info = 0;
		explicit LLT(const Mat& A): L(A)
		{
			const size_t n = A.r;
			for (size_t k = 0; k < n; k++)
			{
				double d = L(k, k);
				for (size_t j = 0; j < k; j++) d -= L(k, j) * L(k, j);
				if (!(d > 0.0) && info == 0) info = static_cast<int>(k) + 1;
				d = std::sqrt(d);
				L(k, k) = d;
				for (size_t i = k + 1; i < n; i++)
				{
					double s = L(i, k);
					for (size_t j = 0; j < k; j++) s -= L(i, j) * L(k, j);
					L(i, k) = s / d;
				}
			}
			for (size_t j = 0; j < n; j++)
				for (size_t i = 0; i < j; i++) L(i, j) = 0.0;
		}
		void solve_inplace(Mat& B) const
		{
			const size_t n = L.r;
#pragma omp parallel for schedule(dynamic, 8)
			for (size_t c = 0; c < B.c; c++)
			{
				double* x = B.col(c);
				for (size_t k = 0; k < n; k++)
				{
					x[k] /= L(k, k);
					for (size_t i = k + 1; i < n; i++) x[i] -= L(i, k) * x[k];
				}
				for (size_t kk = n; kk-- > 0;)
				{
					double s = x[kk];
					for (size_t i = kk + 1; i < n; i++) s -= L(i, kk) * x[i];
					x[kk] = s / L(kk, kk);
				}
			}
		}
	};

	static int nlml_impl(const double* xin, size_t n, const double* X, const double* y, size_t N, double* value, double* grad)
	{
		if (!xin || !X || !y || !value) return GPLE_ERR_BAD_ARG;
		double x[5];
		nlml_params(xin, n, x);
		const Mat K = nlml_gram(x, X, N, X, N, true);
		const LLT llt(K);
		Mat KInv = identity<double>(N);
		llt.solve_inplace(KInv);
		Mat b(N, 1);
		std::copy(y, y + N, b.a.begin());
		llt.solve_inplace(b);
		double r = 0.0;
		for (size_t i = 0; i < N; i++) r += y[i] * b.a[i];
		r /= 2.0;
		for (size_t i = 0; i < N; i++) r += std::log(std::abs(llt.L(i, i)));
		*value = r; // test/gpr.cpp:515
		if (grad)
		{
			// dK as the reference builds it (test/gpr.cpp:408-468): weight * K_k for the two weights (sic: not 2 w K_k); for the
			// entries of W: weight^2 / W_jj * dG/dlog(W_jj) on the diagonal (:444), weight^2 * dG/dW_jk off it (:448), with
			// u = W^T e: dG/da = -G u0 e0, dG/dc = -G u0 e1, dG/db = -G u1 e1
			double g5[5];
			for (int ip = 0; ip < 5; ip++)
			{
				double tr = 0.0;
#pragma omp parallel for reduction(+ : tr) schedule(static)
				for (size_t j = 0; j < N; j++)
					for (size_t i = 0; i < N; i++)
					{
						const double e0 = X[2 * i] - X[2 * j], e1 = X[2 * i + 1] - X[2 * j + 1];
						const double u0 = x[2] * e0 + x[3] * e1, u1 = x[4] * e1;
						const double g = std::exp(-(u0 * u0 + u1 * u1) / 2.0);
						double dk;
						if (ip == 0) dk = i == j ? x[0] : 0.0;
						else if (ip == 1) dk = x[1] * g;
						else if (ip == 2) dk = x[1] * x[1] / x[2] * (-g * x[2] * u0 * e0);
						else if (ip == 3) dk = x[1] * x[1] * (-g * u0 * e1);
						else dk = x[1] * x[1] / x[4] * (-g * x[4] * u1 * e1);
						// ((KInv - b b^T) * dK).trace() = sum_ij (KInv - b b^T)(i,j) dK(j,i)
						tr += (KInv(i, j) - b.a[i] * b.a[j]) * dk;
					}
				g5[ip] = tr / 2.0; // test/gpr.cpp:525
			}
			if (n == 5) std::copy(g5, g5 + 5, grad);
			else grad[0] = g5[0], grad[1] = g5[1], grad[2] = g5[2], grad[3] = g5[4];
		}
		return GPLE_OK;
	}
	static int nlml_predict_impl(const double* xin, size_t n, const double* X, const double* y, size_t N, const double* Xs, size_t M, double* mean)
	{
		if (!xin || !X || !y || (!Xs && M) || !mean) return GPLE_ERR_BAD_ARG;
		double x[5];
		nlml_params(xin, n, x);
		const Mat K = nlml_gram(x, X, N, X, N, true);
		const LLT llt(K);
		Mat b(N, 1);
		std::copy(y, y + N, b.a.begin());
		llt.solve_inplace(b); // test/gpr.cpp:692
		const Mat Ks = nlml_gram(x, Xs, M, X, N, false);
		const Vec mu = matvec(Ks, b.a); // test/gpr.cpp:700
		std::copy(mu.begin(), mu.end(), mean);
		return GPLE_OK;
	}
	int oracle_nlml(const double x[4], const double* X, const double* y, size_t N, double* value, double* grad) { return nlml_impl(x, 4, X, y, N, value, grad); }
	int oracle_nlml_predict(const double x[4], const double* X, const double* y, size_t N, const double* Xs, size_t M, double* mean)
	{
		return nlml_predict_impl(x, 4, X, y, N, Xs, M, mean);
	}
	int oracle_nlml_cross(const double x[5], const double* X, const double* y, size_t N, double* value, double* grad) { return nlml_impl(x, 5, X, y, N, value, grad); }
	int oracle_nlml_cross_predict(const double x[5], const double* X, const double* y, size_t N, const double* Xs, size_t M, double* mean)
	{
		return nlml_predict_impl(x, 5, X, y, N, Xs, M, mean);
	}
}
