// mc.h — adapter for gaussian_process_liouville_equation/mc.h:12-121.  The reference's declarations are kept (initial_distribution,
// generate_extra_points, MCParameters, monte_carlo_selection, new_element_point_selection stay defined in the reference's mc.cpp and
// keep working through the point-wise DistributionFunction).  Added: the Metropolis walk itself — mc.cpp's file-local
// generate_markov_chain (mc.cpp:118-165), one predict per step and walker in the reference — for all walkers of an element at once on
// the device (gple_markov_chain: Philox4x32-10 keyed by `seed`; the reference's clock-seeded engine shared between threads,
// mc.cpp:17, has no reproducible stream to match).
#ifndef MC_H
#define MC_H

#include "stdafx.h"

#include "predict.h"
#include "storage.h"

/// mc.h:13
using AutoCorrelations = QuantumStorage<Eigen::VectorXd>;

/// mc.h:24-32, mc.cpp:20-57
std::complex<double> initial_distribution(const ClassicalPhaseVector& r0, const ClassicalPhaseVector& SigmaR0, const ClassicalPhaseVector& r,
	const std::size_t RowIndex, const std::size_t ColIndex, const std::array<double, NumPES>& InitialPopulation = {1.0},
	const std::array<double, NumPES>& InitialPhaseFactor = {0});
/// mc.h:39-43, mc.cpp:96-117
AllPoints generate_extra_points(const AllPoints& density, const std::size_t NumExtraPoints, const DistributionFunction& distribution);

/// mc.h:46-92
class MCParameters final
{
	std::size_t NOMC;
	double displacement;

public:
	static constexpr double AboveMinFactor = 1.1;
	MCParameters(const std::size_t InitialSteps = 200, const double InitialDisplacement = 1.0): NOMC(InitialSteps), displacement(InitialDisplacement) {}
	void set_num_MC_steps(const std::size_t NOMC_) { NOMC = NOMC_; }
	void set_displacement(const double displacement_) { displacement = displacement_; }
	std::size_t get_num_MC_steps(void) const { return NOMC; }
	double get_max_displacement(void) const { return displacement; }
};

/// mc.h:98-102, mc.cpp:376-403
void monte_carlo_selection(AllPoints& density, QuantumStorage<MCParameters>& MCParams, const DistributionFunction& distribution);
/// mc.h:114-121, mc.cpp:405-537
void new_element_point_selection(AllPoints& density, AllPoints& extra_points, const QuantumStorage<bool>& IsSmallOld, const QuantumStorage<bool>& IsSmall,
	QuantumStorage<MCParameters>& MCParams, const DistributionFunction& distribution);

namespace gple_host
{
	/// generate_markov_chain (mc.cpp:118-165) for every start point at once: NumSteps Metropolis steps on |cut-off prediction| of element
	/// (iPES, jPES), uniform displacements in [-MaxDisplacement, MaxDisplacement) per dimension.  Returns the last point of every chain
	/// (WholeChain.back()) and its acceptance ratio.
	inline std::tuple<EigenVector<ClassicalPhaseVector>, std::vector<double>> generate_markov_chain(const std::size_t NumSteps,
		const TrainingKernels& AllKernels, const double MaxDisplacement, const std::size_t iPES, const std::size_t jPES,
		const EigenVector<ClassicalPhaseVector>& r, const unsigned long long seed)
	{
		std::tuple<EigenVector<ClassicalPhaseVector>, std::vector<double>> result(r, std::vector<double>(r.size(), 0.0));
		auto& [last, ratio] = result;
		std::vector<double> flat(PhaseDim * r.size());
		for (std::size_t i = 0; i < r.size(); i++)
			for (std::size_t d = 0; d < PhaseDim; d++) flat[PhaseDim * i + d] = r[i][d];
		gple_element e{nullptr, nullptr};
		if (iPES == jPES)
		{
			if (AllKernels(iPES).has_value()) e.real = AllKernels(iPES)->handle();
		}
		else if (AllKernels(iPES, jPES).has_value())
			e.cplx = AllKernels(iPES, jPES)->handle();
		check(gple_markov_chain(context(), &e, NumSteps, MaxDisplacement, seed, flat.data(), r.size(), ratio.data()), context());
		for (std::size_t i = 0; i < r.size(); i++)
			for (std::size_t d = 0; d < PhaseDim; d++) last[i][d] = flat[PhaseDim * i + d];
		return result;
	}

	/// the selection loop of element_monte_carlo (mc.cpp:349-369) with the step count and displacement of MCParams: every point walks its
	/// chain, ends at the chain's last point and takes the predicted density there — two device calls per element instead of
	/// (NumSteps + 2) one-point predicts per point
	inline void element_monte_carlo_walk(ElementPoints& density, const MCParameters& MCParams, const TrainingKernels& AllKernels, const std::size_t RowIndex,
		const std::size_t ColIndex, const unsigned long long seed)
	{
		EigenVector<ClassicalPhaseVector> start;
		start.reserve(density.size());
		for (const PhaseSpacePoint& psp : density) start.push_back(psp.get<0>());
		const auto [last, ratio] = generate_markov_chain(MCParams.get_num_MC_steps(), AllKernels, MCParams.get_max_displacement(), RowIndex, ColIndex, start, seed);
		PhasePoints pts(PhaseDim, static_cast<Eigen::Index>(last.size()));
		for (std::size_t i = 0; i < last.size(); i++)
			for (std::size_t d = 0; d < PhaseDim; d++) pts(static_cast<Eigen::Index>(d), static_cast<Eigen::Index>(i)) = last[i][d];
		const Eigen::VectorXcd rho = DistributionBatcher(AllKernels)(pts, RowIndex, ColIndex);
		for (std::size_t i = 0; i < density.size(); i++)
		{
			auto& [r, rho_i] = density[i];
			r = last[i];
			rho_i = rho[static_cast<Eigen::Index>(i)];
		}
	}
} // namespace gple_host

#endif // !MC_H
