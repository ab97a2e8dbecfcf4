// stdafx.h — TEST SCAFFOLDING standing where the reference's stdafx.h stands on the include path: it defines the same include
// guard and the same global names (constants and typedefs of gaussian_process_liouville_equation/stdafx.h:107-155) on top of
// mini_eigen.h, so that the header adapters meet exactly the declarations they will meet inside the reference's translation
// units (no redefinition, no ambiguity) — without Eigen / xtensor / NLopt / spdlog, which this image lacks.
#ifndef STDAFX_H
#define STDAFX_H

#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <functional>
#include <iostream>
#include <limits>
#include <memory>
#include <numbers>
#include <numeric>
#include <optional>
#include <ranges>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "mini_eigen.h"

template <std::size_t N, typename T>
inline constexpr T power(const T t)
{
	T r = 1;
	for (std::size_t i = 0; i < N; i++) r *= t;
	return r;
}

constexpr double hbar = 1.0;
#ifndef GPLE_TEST_NUM_PES
#define GPLE_TEST_NUM_PES 2
#endif
constexpr std::size_t NumPES = GPLE_TEST_NUM_PES;
constexpr std::size_t NumElements = NumPES * NumPES;
constexpr std::size_t NumOffDiagonalElements = (NumElements - NumPES) / 2;
constexpr std::size_t NumTriangularElements = (NumElements + NumPES) / 2;
constexpr std::size_t Dim = 1;
constexpr std::size_t PhaseDim = Dim * 2;
constexpr double PurityFactor = power<Dim>(2.0 * std::numbers::pi * hbar);

const Eigen::IOFormat VectorFormatter(Eigen::StreamPrecision, Eigen::DontAlignCols, " ", " ", "", "", "", "");
const Eigen::IOFormat MatrixFormatter(Eigen::StreamPrecision, Eigen::DontAlignCols, " ", "\n", "", "", "", "");

template <typename T>
using QuantumMatrix = Eigen::Matrix<T, NumPES, NumPES, Eigen::StorageOptions::RowMajor | Eigen::StorageOptions::AutoAlign>;
template <typename T>
using QuantumVector = Eigen::Matrix<T, NumPES, 1>;
template <typename T>
using ClassicalVector = Eigen::Matrix<T, Dim, 1>;
template <typename T>
using EigenVector = std::vector<T, Eigen::aligned_allocator<T>>;
using ClassicalPhaseVector = Eigen::Matrix<double, PhaseDim, 1>;
using PhasePoints = Eigen::Matrix<double, PhaseDim, Eigen::Dynamic>;
using DistributionFunction = std::function<std::complex<double>(const ClassicalPhaseVector&, std::size_t, std::size_t)>;

#endif // !STDAFX_H
