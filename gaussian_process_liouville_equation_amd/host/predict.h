// predict.h — adapter for gaussian_process_liouville_equation/predict.h:14-143: the training-set packing and TrainingKernels
// with its aggregates (predict.cpp:246-559) on top of the MI355X library.  The Monte-Carlo observables of predict.h:22-62
// (predict.cpp:25-244) are host glue outside the hot path: they stay declared here and defined in the reference's predict.cpp.
#ifndef PREDICT_H
#define PREDICT_H

#include "stdafx.h"

#include "complex_kernel.h"
#include "kernel.h"
#include "storage.h"

#include <mutex>

/// predict.h:14-17
using AllTrainingSets = QuantumStorage<ElementTrainingSet>;
static constexpr std::size_t NumTotalParameters = KernelBase::NumTotalParameters * NumPES + ComplexKernelBase::NumTotalParameters * NumOffDiagonalElements;

/// predict.h:22-62 — defined in the reference's predict.cpp:25-244 (unchanged)
QuantumVector<double> calculate_population_each_surface(const AllPoints& density);
ClassicalPhaseVector calculate_1st_order_average_one_surface(const ElementPoints& density);
ClassicalPhaseVector calculate_standard_deviation_one_surface(const ElementPoints& density);
ClassicalPhaseVector calculate_1st_order_average_all_surface(const AllPoints& density);
double calculate_total_energy_average_one_surface(const ElementPoints& density, const ClassicalVector<double>& mass, const std::size_t PESIndex);
QuantumVector<double> calculate_total_energy_average_each_surface(const AllPoints& density, const ClassicalVector<double>& mass);
double calculate_total_energy_average_all_surface(const AllPoints& density, const ClassicalVector<double>& mass);
QuantumMatrix<double> calculate_purity_each_element(const AllPoints& density);

/// predict.h:67-70
inline double calculate_population_one_surface(const TrainingKernel& kernel)
{
	return kernel.get_population();
}
/// predict.h:75-78
inline ClassicalPhaseVector calculate_1st_order_average_one_surface(const TrainingKernel& kernel)
{
	return kernel.get_1st_order_average() / kernel.get_population();
}

/// predict.h:83, predict.cpp:246-280: AoS PhaseSpacePoint{r, rho} -> PhasePoints (2 x N) + VectorXcd (N) per element
inline AllTrainingSets construct_training_sets(const AllPoints& density)
{
	AllTrainingSets result;
	for (std::size_t iPES = 0; iPES < NumPES; iPES++)
		for (std::size_t jPES = 0; jPES <= iPES; jPES++)
		{
			const ElementPoints& ElementDensity = density(iPES, jPES);
			const std::size_t NumPoints = ElementDensity.size();
			PhasePoints feature(PhaseDim, NumPoints);
			Eigen::VectorXcd label(NumPoints);
			for (std::size_t iPoint = 0; iPoint < NumPoints; iPoint++)
			{
				const auto& [r, rho] = ElementDensity[iPoint];
				for (std::size_t d = 0; d < PhaseDim; d++) feature(d, iPoint) = r[d];
				label[iPoint] = rho;
			}
			result(iPES, jPES) = ElementTrainingSet(std::move(feature), std::move(label));
		}
	return result;
}

/// predict.h:89-143
class TrainingKernels final: public QuantumStorage<std::optional<TrainingKernel>, std::optional<TrainingComplexKernel>>
{
public:
	using BaseType = QuantumStorage<std::optional<TrainingKernel>, std::optional<TrainingComplexKernel>>;

	TrainingKernels(const QuantumStorage<ParameterVector>& ParameterVectors, const AllTrainingSets& TrainingSets, const bool IsToCalculateError,
		const bool IsToCalculateAverage, const bool IsToCalculateDerivative)
	{
		for (std::size_t iPES = 0; iPES < NumPES; iPES++) // predict.cpp:290-318: an empty set gives no kernel
			if (std::get<0>(TrainingSets(iPES)).size() != 0)
				BaseType::operator()(iPES).emplace(ParameterVectors(iPES), TrainingSets(iPES), IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative);
		for (std::size_t iPES = 1; iPES < NumPES; iPES++) // predict.cpp:328-360: so do all-zero parameters off the diagonal
			for (std::size_t jPES = 0; jPES < iPES; jPES++)
			{
				const ParameterVector& p = ParameterVectors(iPES, jPES);
				const bool AllZero = std::all_of(p.cbegin(), p.cend(), [](double d) { return d == 0.0; });
				if (std::get<0>(TrainingSets(iPES, jPES)).size() != 0 && !AllZero)
					BaseType::operator()(iPES, jPES).emplace(p, TrainingSets(iPES, jPES), IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative);
			}
	}
	/// predict.cpp:390-393: what main.cpp builds every tick (error, averages, no derivatives)
	TrainingKernels(const QuantumStorage<ParameterVector>& ParameterVectors, const AllPoints& density):
		TrainingKernels(ParameterVectors, construct_training_sets(density), true, true, false)
	{
	}
	double calculate_population(void) const // predict.cpp:395-406
	{
		double result = 0.0;
		for (const std::optional<TrainingKernel>& k : BaseType::get_diagonal_data())
			if (k.has_value()) result += k->get_population();
		return result;
	}
	ClassicalPhaseVector calculate_1st_order_average(void) const // predict.cpp:408-419
	{
		ClassicalPhaseVector result;
		for (std::size_t d = 0; d < PhaseDim; d++) result[d] = 0.0;
		for (const std::optional<TrainingKernel>& k : BaseType::get_diagonal_data())
			if (k.has_value())
				for (std::size_t d = 0; d < PhaseDim; d++) result[d] += k->get_1st_order_average()[d];
		return result;
	}
	double calculate_total_energy_average(const QuantumVector<double>& Energies) const // predict.cpp:423-436
	{
		double result = 0.0;
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			if (BaseType::operator()(iPES).has_value()) result += BaseType::operator()(iPES)->get_population() * Energies[iPES];
		return result;
	}
	double calculate_purity(void) const // predict.cpp:439-463: weight 1 on the diagonal, 2 off it
	{
		double result = 0.0;
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
		{
			if (BaseType::operator()(iPES).has_value()) result += BaseType::operator()(iPES)->get_purity();
			for (std::size_t jPES = 0; jPES < iPES; jPES++)
				if (BaseType::operator()(iPES, jPES).has_value()) result += 2.0 * BaseType::operator()(iPES, jPES)->get_purity();
		}
		return result;
	}
	ParameterVector population_derivative(void) const // predict.cpp:465-484
	{
		ParameterVector result(NumPES * KernelBase::NumTotalParameters, 0.0);
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			if (BaseType::operator()(iPES).has_value())
			{
				const KernelBase::ParameterArray<double>& g = BaseType::operator()(iPES)->get_population_derivative();
				std::copy(g.cbegin(), g.cend(), result.begin() + iPES * KernelBase::NumTotalParameters);
			}
		return result;
	}
	ParameterVector total_energy_derivative(const QuantumVector<double>& Energies) const // predict.cpp:486-510
	{
		ParameterVector result = population_derivative();
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			for (std::size_t k = 0; k < KernelBase::NumTotalParameters; k++) result[iPES * KernelBase::NumTotalParameters + k] *= Energies[iPES];
		return result;
	}
	ParameterVector purity_derivative(void) const // predict.cpp:512-559
	{
		ParameterVector result(NumTotalParameters, 0.0);
		std::size_t iParam = 0;
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			for (std::size_t jPES = 0; jPES <= iPES; jPES++)
			{
				if (iPES == jPES)
				{
					if (BaseType::operator()(iPES).has_value())
					{
						const KernelBase::ParameterArray<double>& d = BaseType::operator()(iPES)->get_purity_derivative();
						std::copy(d.cbegin(), d.cend(), result.begin() + iParam);
					}
					iParam += KernelBase::NumTotalParameters;
				}
				else
				{
					if (BaseType::operator()(iPES, jPES).has_value())
					{
						const ComplexKernelBase::ParameterArray<double>& d = BaseType::operator()(iPES, jPES)->get_purity_derivative();
						for (std::size_t k = 0; k < d.size(); k++) result[iParam + k] = d[k] * 2;
					}
					iParam += ComplexKernelBase::NumTotalParameters;
				}
			}
		return result;
	}
};

namespace gple_host
{
	/// loose_function of opt.cpp:441-482 evaluated in ONE library call (fit + predict on the extra set + make_normal), with the
	/// NLopt objective ABI double(const std::vector<double>&, std::vector<double>&, void*): opt.cpp's own static loose_function
	/// keeps working through the class adapters; pointing the optimisers at this one instead saves the host round trips.
	using ElementTrainingParameters = std::tuple<const ElementTrainingSet&, const ElementTrainingSet&>;
	inline double loose_function(const ParameterVector& x, ParameterVector& grad, void* params)
	{
		const auto& [TrainingSet, ExtraTrainingSet] = *static_cast<ElementTrainingParameters*>(params);
		const auto& [Feature, Label] = TrainingSet;
		const auto& [ExtraFeature, ExtraLabel] = ExtraTrainingSet;
		double value = 0.0;
		check(gple_loose_function(context(), x.data(), x.size(), Feature.data(), reinterpret_cast<const double*>(Label.data()),
				  static_cast<std::size_t>(Feature.cols()), ExtraFeature.data(), reinterpret_cast<const double*>(ExtraLabel.data()),
				  static_cast<std::size_t>(ExtraFeature.cols()), &value, grad.empty() ? nullptr : grad.data()),
			context());
		return value;
	}

	/// gather - predict - scatter replacement of the per-point DistributionFunction (stdafx.h:155).
	/// main.cpp:75-101 answers every call by constructing a Predictive*Kernel for ONE point; evolve.cpp:298 asks 8 times per
	/// sample and mc.cpp:158-172 once per Metropolis step.  Callers `request` points (from any thread) and keep the ticket,
	/// `flush` runs one predict per density-matrix element (gple_predict_batch), `result(ticket)` hands the values back.
	class DistributionBatcher
	{
	public:
		explicit DistributionBatcher(const TrainingKernels& AllKernels)
		{
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
		}
		/// index of element (Row, Col), Col <= Row, in the reference's packing order (0,0), (1,0), (1,1), ...
		static int element_index(std::size_t RowIndex, std::size_t ColIndex) { return static_cast<int>(RowIndex * (RowIndex + 1) / 2 + ColIndex); }
		std::size_t request(const ClassicalPhaseVector& r, std::size_t RowIndex, std::size_t ColIndex)
		{
			std::lock_guard<std::mutex> lk(Mu);
			Points.push_back(r[0]), Points.push_back(r[1]);
			ElementOfRequest.push_back(element_index(RowIndex, ColIndex));
			return ElementOfRequest.size() - 1;
		}
		void flush(void)
		{
			std::lock_guard<std::mutex> lk(Mu);
			const std::size_t done = Results.size() / 2, n = ElementOfRequest.size() - done;
			if (n == 0) return;
			Results.resize(2 * ElementOfRequest.size());
			check(gple_predict_batch(context(), Elements.data(), Elements.size(), Points.data() + 2 * done, ElementOfRequest.data() + done, n,
					  Results.data() + 2 * done),
				context());
		}
		std::complex<double> result(std::size_t ticket) const { return {Results[2 * ticket], Results[2 * ticket + 1]}; }
		void clear(void)
		{
			std::lock_guard<std::mutex> lk(Mu);
			Points.clear(), ElementOfRequest.clear(), Results.clear();
		}
		/// all points of one element at once: the batched form of main.cpp:75-101's lambda
		Eigen::VectorXcd operator()(const PhasePoints& r, std::size_t RowIndex, std::size_t ColIndex) const
		{
			const std::size_t n = static_cast<std::size_t>(r.cols());
			Eigen::VectorXcd out(r.cols());
			const std::vector<int> which(n, element_index(RowIndex, ColIndex));
			check(gple_predict_batch(context(), Elements.data(), Elements.size(), r.data(), which.data(), n, reinterpret_cast<double*>(out.data())), context());
			return out;
		}
		/// drop-in DistributionFunction for callers that stay point-wise (each call is a batch of one)
		DistributionFunction pointwise(void) const
		{
			return [this](const ClassicalPhaseVector& r, const std::size_t RowIndex, const std::size_t ColIndex) -> std::complex<double>
			{
				const int which = element_index(RowIndex, ColIndex);
				const double pt[2] = {r[0], r[1]};
				double out[2];
				check(gple_predict_batch(context(), Elements.data(), Elements.size(), pt, &which, 1, out), context());
				return {out[0], out[1]};
			};
		}

	private:
		std::vector<gple_element> Elements;
		std::vector<double> Points, Results;
		std::vector<int> ElementOfRequest;
		mutable std::mutex Mu;
	};
} // namespace gple_host

#endif // !PREDICT_H
