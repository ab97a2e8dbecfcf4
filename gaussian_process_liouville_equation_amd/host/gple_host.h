// gple_host.h — common part of the header-only C++ adapters that re-create the reference's kernel classes on top of
// the C-ABI (include/gple.h).  See INTEGRATION.md.
//
// Types: with Eigen on the include path the adapters use the reference's own typedefs (stdafx.h:133-155), so they slot
// into main.cpp / evolve.cpp / opt.cpp unchanged.  Without Eigen (this image has none) a minimal owning array type with
// the handful of members the adapters and the tests need stands in; it is NOT an Eigen replacement and exists only so
// that the adapters compile and are tested here.
#ifndef GPLE_HOST_H
#define GPLE_HOST_H

#include <array>
#include <cassert>
#include <complex>
#include <cstddef>
#include <cstdlib>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/gple.h"

#if __has_include(<Eigen/Eigen>)
#include <Eigen/Eigen>
namespace gple_host
{
	constexpr std::size_t Dim = 1, PhaseDim = 2; // stdafx.h:119-121
	using ClassicalPhaseVector = Eigen::Matrix<double, PhaseDim, 1>;
	using PhasePoints = Eigen::Matrix<double, PhaseDim, Eigen::Dynamic>;
	using VectorXd = Eigen::VectorXd;
	using VectorXcd = Eigen::VectorXcd;
	using MatrixXd = Eigen::MatrixXd;
	using MatrixXcd = Eigen::MatrixXcd;
	inline std::size_t num_points(const PhasePoints& p) { return static_cast<std::size_t>(p.cols()); }
	inline PhasePoints make_points(std::size_t n) { return PhasePoints(PhaseDim, n); }
} // namespace gple_host
#else
namespace gple_host
{
	constexpr std::size_t Dim = 1, PhaseDim = 2;
	template <typename T>
	struct Array
	{
		std::vector<T> a;
		std::size_t r = 0, c = 0; // column-major r x c
		Array() = default;
		explicit Array(std::size_t n): a(n), r(n), c(1) {}
		Array(std::size_t rows, std::size_t cols): a(rows * cols), r(rows), c(cols) {}
		T* data() { return a.data(); }
		const T* data() const { return a.data(); }
		std::size_t size() const { return a.size(); }
		std::size_t rows() const { return r; }
		std::size_t cols() const { return c; }
		T& operator[](std::size_t i) { return a[i]; }
		const T& operator[](std::size_t i) const { return a[i]; }
		T& operator()(std::size_t i, std::size_t j) { return a[i + j * r]; }
		const T& operator()(std::size_t i, std::size_t j) const { return a[i + j * r]; }
	};
	using ClassicalPhaseVector = std::array<double, PhaseDim>;
	using PhasePoints = Array<double>; // PhaseDim x N, column-major == interleaved [x0,p0,x1,p1,...]
	using VectorXd = Array<double>;
	using VectorXcd = Array<std::complex<double>>;
	using MatrixXd = Array<double>;
	using MatrixXcd = Array<std::complex<double>>;
	inline std::size_t num_points(const PhasePoints& p) { return p.cols(); }
	inline PhasePoints make_points(std::size_t n) { return PhasePoints(PhaseDim, n); }
} // namespace gple_host
#endif

namespace gple_host
{
	inline void check(int status, gple_ctx* ctx)
	{
		if (status != GPLE_OK)
			throw std::runtime_error(std::string("gple: ") + gple_status_string(status) + (ctx ? std::string(": ") + gple_ctx_last_error(ctx) : ""));
	}
	// One context per process (device from GPLE_DEVICE, default 0).  The reference's objects have no notion of a device;
	// constructing a kernel object is the hot path.
	inline gple_ctx* context()
	{
		struct Holder
		{
			gple_ctx* ctx = nullptr;
			Holder()
			{
				const char* d = std::getenv("GPLE_DEVICE");
				check(gple_ctx_create(d ? std::atoi(d) : 0, nullptr, &ctx), nullptr);
			}
			~Holder() { gple_ctx_destroy(ctx); }
		};
		static Holder h;
		return h.ctx;
	}
	inline unsigned flags(bool err, bool avg, bool der)
	{
		return (err ? GPLE_CALC_ERROR : 0u) | (avg ? GPLE_CALC_AVERAGE : 0u) | (der ? GPLE_CALC_DERIVATIVE : 0u);
	}
} // namespace gple_host
#endif // GPLE_HOST_H
