// evolve.h — adapter for gaussian_process_liouville_equation/evolve.h:16-52.  The reference's three functions keep their
// declarations (they stay defined in the reference's evolve.cpp and keep working through the point-wise DistributionFunction);
// next to each stands an overload that takes the fitted TrainingKernels in place of the per-point callback and runs the whole
// loop on the device: gple_evolve propagates every selected point and rebuilds its density by the three-branch back-propagation
// of evolve.cpp:184-372 with the 8 predicts per point gathered into one batch per density-matrix element.
// main.cpp:140-141 then reads
//     evolve(density, mass, dt, *all_kernels);          // was: evolve(density, mass, dt, predict_distribution);
//     evolve(extra_points, mass, dt, *all_kernels);
//     IsSmall = is_very_small(density, mass, dt, *all_kernels);
// and a tick costs three batched predicts per call instead of 8 N one-point predicts of 36-235 us each.
#ifndef EVOLVE_H
#define EVOLVE_H

#include "stdafx.h"

#include "predict.h"
#include "storage.h"

/// evolve.h:16-21, evolve.cpp:377-423 (unchanged, point-wise)
void evolve(AllPoints& density, const ClassicalVector<double>& mass, const double dt, const DistributionFunction& distribution);
/// evolve.h:33-40, evolve.cpp:425-443
std::complex<double> new_point_predict(const ClassicalPhaseVector& r, const ClassicalVector<double>& mass, const double dt,
	const DistributionFunction& distribution, const std::size_t RowIndex, const std::size_t ColIndex);
/// evolve.h:47-52, evolve.cpp:445-478
QuantumStorage<bool> is_very_small(const AllPoints& density, const ClassicalVector<double>& mass, const double dt, const DistributionFunction& distribution);

namespace gple_host
{
	/// pes.h:27-41: the model the reference was compiled for (TestModel; DAC unless -DTestModel=...).  Inside the reference pes.h is on
	/// the include path of every caller of evolve(); without it (the test scaffolding) GPLE_HOST_PES_MODEL selects, default Tully II.
	inline int pes_model()
	{
#ifdef PES_H
		return static_cast<int>(TestModel); // SAC = 0, DAC = 1, ECR = 2: the values of gple_pes_model
#elif defined(GPLE_HOST_PES_MODEL)
		return GPLE_HOST_PES_MODEL;
#else
		return GPLE_PES_DAC;
#endif
	}
	/// the fits behind the elements of TrainingKernels in the packing order (0,0), (1,0), (1,1), ... (an element without a kernel: {NULL, NULL})
	inline std::vector<gple_element> elements_of(const TrainingKernels& AllKernels)
	{
		std::vector<gple_element> Elements;
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			for (std::size_t jPES = 0; jPES <= iPES; jPES++)
			{
				gple_element e{nullptr, nullptr};
				if (iPES == jPES)
				{
					if (AllKernels(iPES).has_value()) e.real = AllKernels(iPES)->handle();
				}
				else if (AllKernels(iPES, jPES).has_value())
					e.cplx = AllKernels(iPES, jPES)->handle();
				Elements.push_back(e);
			}
		return Elements;
	}
	/// AoS PhaseSpacePoint{r, rho} of every element <-> the interleaved arrays of gple_points
	struct PackedPoints
	{
		std::array<std::vector<double>, NumTriangularElements> r, rho;
		std::array<gple_points, NumTriangularElements> pts;
		explicit PackedPoints(const AllPoints& density)
		{
			std::size_t e = 0;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++, e++)
				{
					const ElementPoints& el = density(iPES, jPES);
					r[e].resize(PhaseDim * el.size()), rho[e].resize(2 * el.size());
					for (std::size_t i = 0; i < el.size(); i++)
					{
						const auto& [ri, rhoi] = el[i];
						for (std::size_t d = 0; d < PhaseDim; d++) r[e][PhaseDim * i + d] = ri[d];
						rho[e][2 * i] = std::real(rhoi), rho[e][2 * i + 1] = std::imag(rhoi);
					}
					pts[e] = gple_points{r[e].data(), rho[e].data(), el.size()};
				}
		}
		void unpack(AllPoints& density) const
		{
			std::size_t e = 0;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++, e++)
				{
					ElementPoints& el = density(iPES, jPES);
					for (std::size_t i = 0; i < el.size(); i++)
					{
						auto& [ri, rhoi] = el[i];
						for (std::size_t d = 0; d < PhaseDim; d++) ri[d] = r[e][PhaseDim * i + d];
						rhoi = std::complex<double>(rho[e][2 * i], rho[e][2 * i + 1]);
					}
				}
		}
	};
	/// the device tick for the system the translation unit is compiled for: two levels = the reference's own three-branch back-propagation
	/// (gple_evolve); three levels = its N-level form (gple_evolve_n), where the reference asserts (evolve.cpp:367-371)
	inline int tick(const gple_element* Elements, const double mass, const double dt, gple_points* pts, const unsigned flags)
	{
		if constexpr (NumPES == 2) return gple_evolve(context(), Elements, pes_model(), mass, dt, pts, flags);
		else return gple_evolve_n(context(), static_cast<int>(NumPES), Elements, pes_model(), mass, dt, pts, flags);
	}
	/// new_point_predict (evolve.cpp:425-443) for all the given points of element (RowIndex, ColIndex) at once
	inline Eigen::VectorXcd new_points_predict(const ElementPoints& points, const ClassicalVector<double>& mass, const double dt,
		const TrainingKernels& AllKernels, const std::size_t RowIndex, const std::size_t ColIndex)
	{
		Eigen::VectorXcd result(static_cast<Eigen::Index>(points.size()));
		if constexpr (NumPES <= 3 && Dim == 1)
		{
			AllPoints probe;
			probe(RowIndex, ColIndex) = points;
			PackedPoints packed(probe);
			const std::vector<gple_element> Elements = elements_of(AllKernels);
			check(tick(Elements.data(), mass[0], dt, packed.pts.data(), GPLE_EVOLVE_NEW_POINTS), context());
			const std::vector<double>& rho = packed.rho[DistributionBatcher::element_index(RowIndex, ColIndex)];
			for (std::size_t i = 0; i < points.size(); i++) result[static_cast<Eigen::Index>(i)] = std::complex<double>(rho[2 * i], rho[2 * i + 1]);
		}
		else
			assert(!"NO INSTANTATION OF MORE THAN THREE LEVEL SYSTEM NOW"); // evolve.cpp:367-371 stops at two; the library goes to three
		return result;
	}
} // namespace gple_host

/// evolve(density, mass, dt, distribution) with distribution = the cut-off prediction of `AllKernels` (main.cpp:75-101), on the device
inline void evolve(AllPoints& density, const ClassicalVector<double>& mass, const double dt, const TrainingKernels& AllKernels)
{
	if constexpr (NumPES <= 3 && Dim == 1)
	{
		gple_host::PackedPoints packed(density);
		const std::vector<gple_element> Elements = gple_host::elements_of(AllKernels);
		gple_host::check(gple_host::tick(Elements.data(), mass[0], dt, packed.pts.data(), 0), gple_host::context());
		packed.unpack(density);
	}
	else
		assert(!"NO INSTANTATION OF MORE THAN THREE LEVEL SYSTEM NOW"); // evolve.cpp:367-371 stops at two
}

/// new_point_predict with the kernels in place of the callback (one point: a batch of one)
inline std::complex<double> new_point_predict(const ClassicalPhaseVector& r, const ClassicalVector<double>& mass, const double dt,
	const TrainingKernels& AllKernels, const std::size_t RowIndex, const std::size_t ColIndex)
{
	ElementPoints one;
	one.emplace_back(r, std::complex<double>(0.0));
	return gple_host::new_points_predict(one, mass, dt, AllKernels, RowIndex, ColIndex)[0];
}

/// is_very_small (evolve.cpp:445-478): an element without points is small if the back-propagated density of every selected point of
/// element (0,0) stays below 1e-5 in modulus there — one batched call per empty element
inline QuantumStorage<bool> is_very_small(const AllPoints& density, const ClassicalVector<double>& mass, const double dt, const TrainingKernels& AllKernels)
{
	static constexpr double epsilon = power<2>(1e-5);
	QuantumStorage<bool> result(false, false);
	for (std::size_t iPES = 0; iPES < NumPES; iPES++)
		for (std::size_t jPES = 0; jPES <= iPES; jPES++)
			if (density(iPES, jPES).empty())
			{
				const Eigen::VectorXcd rho = gple_host::new_points_predict(density(0), mass, dt, AllKernels, iPES, jPES);
				bool all_small = true;
				for (Eigen::Index i = 0; i < rho.size(); i++) all_small = all_small && std::norm(rho[i]) < epsilon;
				result(iPES, jPES) = all_small;
			}
	return result;
}

#endif // !EVOLVE_H
