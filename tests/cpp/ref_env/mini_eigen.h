// mini_eigen.h — TEST SCAFFOLDING.  This image has no Eigen, so the header adapters in
// gaussian_process_liouville_equation_amd/host/ cannot be compiled against the real library here.  This file provides, in
// namespace Eigen, the handful of members the adapters and the reference's call sites for this path use (column-major dense
// storage, data(), rows(), cols(), size(), element access, real()/imag(), value(), Zero, format) so that the adapters are
// compiled and exercised.  It is not an Eigen replacement and is never a way to build the reference.
#ifndef GPLE_TEST_MINI_EIGEN_H
#define GPLE_TEST_MINI_EIGEN_H

#include <cassert>
#include <complex>
#include <cstddef>
#include <memory>
#include <ostream>
#include <type_traits>
#include <vector>

namespace Eigen
{
	constexpr int Dynamic = -1;
	using Index = std::ptrdiff_t;
	enum { StreamPrecision = -1, DontAlignCols = 1 };
	namespace StorageOptions
	{
		enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0 };
	}
	template <typename T>
	using aligned_allocator = std::allocator<T>;

	struct IOFormat
	{
		const char* coeff_sep = " ";
		IOFormat() = default;
		IOFormat(int, int, const char* c, const char*, const char*, const char*, const char*, const char*): coeff_sep(c) {}
	};

	template <typename T, int R, int C, int Opt = 0>
	class Matrix
	{
	public:
		using Scalar = T;
		Matrix(): r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C), a_(static_cast<std::size_t>(r_ * c_)) {}
		explicit Matrix(Index n): r_(C == 1 ? n : (R == Dynamic ? n : R)), c_(C == 1 ? 1 : (R == Dynamic ? 1 : n)), a_(static_cast<std::size_t>(r_ * c_)) {}
		Matrix(Index rows, Index cols): r_(rows), c_(cols), a_(static_cast<std::size_t>(rows * cols))
		{
			assert((R == Dynamic || R == rows) && (C == Dynamic || C == cols));
		}
		// fixed-size column -> dynamic matrix of the same height (a ClassicalPhaseVector passed where PhasePoints is expected, main.cpp:83)
		template <int R2, int C2, int O2, typename = std::enable_if_t<(R2 != R || C2 != C)>>
		Matrix(const Matrix<T, R2, C2, O2>& o): r_(o.rows()), c_(o.cols()), a_(o.data(), o.data() + o.size())
		{
			assert((R == Dynamic || R == r_) && (C == Dynamic || C == c_));
		}
		T* data() { return a_.data(); }
		const T* data() const { return a_.data(); }
		Index rows() const { return r_; }
		Index cols() const { return c_; }
		Index size() const { return r_ * c_; }
		T& operator[](Index i) { return a_[static_cast<std::size_t>(i)]; }
		const T& operator[](Index i) const { return a_[static_cast<std::size_t>(i)]; }
		T& operator()(Index i) { return a_[static_cast<std::size_t>(i)]; }
		const T& operator()(Index i) const { return a_[static_cast<std::size_t>(i)]; }
		T& operator()(Index i, Index j) { return a_[static_cast<std::size_t>(i + j * r_)]; }
		const T& operator()(Index i, Index j) const { return a_[static_cast<std::size_t>(i + j * r_)]; }
		const T& value() const
		{
			assert(size() == 1);
			return a_[0];
		}
		Matrix<T, R, 1> col(Index j) const
		{
			Matrix<T, R, 1> v(r_);
			for (Index i = 0; i < r_; i++) v[i] = (*this)(i, j);
			return v;
		}
		static Matrix Zero(Index n)
		{
			Matrix m(n);
			for (auto& x : m.a_) x = T(0);
			return m;
		}
		static Matrix Zero(Index rows, Index cols)
		{
			Matrix m(rows, cols);
			for (auto& x : m.a_) x = T(0);
			return m;
		}
		auto real() const
		{
			using S = decltype(std::real(T()));
			Matrix<S, R, C> m(r_, c_);
			for (Index i = 0; i < size(); i++) m[i] = std::real(a_[static_cast<std::size_t>(i)]);
			return m;
		}
		auto imag() const
		{
			using S = decltype(std::real(T()));
			Matrix<S, R, C> m(r_, c_);
			for (Index i = 0; i < size(); i++) m[i] = std::imag(a_[static_cast<std::size_t>(i)]);
			return m;
		}
		Matrix operator/(double d) const
		{
			Matrix m(*this);
			for (auto& x : m.a_) x /= d;
			return m;
		}
		Matrix operator*(double d) const
		{
			Matrix m(*this);
			for (auto& x : m.a_) x *= d;
			return m;
		}
		T sum() const
		{
			T s = T(0);
			for (const auto& x : a_) s += x;
			return s;
		}
		struct Formatted
		{
			const Matrix& m;
			const IOFormat& f;
			friend std::ostream& operator<<(std::ostream& os, const Formatted& p)
			{
				for (Index i = 0; i < p.m.size(); i++) os << (i ? p.f.coeff_sep : "") << p.m[i];
				return os;
			}
		};
		Formatted format(const IOFormat& f) const { return Formatted{*this, f}; }
		friend bool operator==(const Matrix& a, const Matrix& b) { return a.r_ == b.r_ && a.c_ == b.c_ && a.a_ == b.a_; }

	private:
		Index r_, c_;
		std::vector<T> a_;
	};
	using VectorXd = Matrix<double, Dynamic, 1>;
	using VectorXcd = Matrix<std::complex<double>, Dynamic, 1>;
	using MatrixXd = Matrix<double, Dynamic, Dynamic>;
	using MatrixXcd = Matrix<std::complex<double>, Dynamic, Dynamic>;
} // namespace Eigen
#endif
