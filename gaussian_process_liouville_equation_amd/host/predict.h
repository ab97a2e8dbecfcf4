// predict.h — adapter for the kernel-related part of gaussian_process_liouville_equation/predict.h:14-143
// (TrainingKernels and its aggregates, predict.cpp:290-559) and the objective of opt.cpp:441-482.
// NumPES is the reference's compile-time constant (stdafx.h:111); define GPLE_NUM_PES to change it.
#ifndef PREDICT_H
#define PREDICT_H

#include "complex_kernel.h"

#ifndef GPLE_NUM_PES
#define GPLE_NUM_PES 2
#endif
constexpr std::size_t NumPES = GPLE_NUM_PES;
constexpr std::size_t NumOffDiagonalElements = NumPES * (NumPES - 1) / 2;
/// predict.h:17
static constexpr std::size_t NumTotalParameters = KernelBase::NumTotalParameters * NumPES + TrainingComplexKernel::NumTotalParameters * NumOffDiagonalElements;

/// storage.h:22-26
inline constexpr std::size_t calculate_offdiagonal_index(const std::size_t RowIndex, const std::size_t ColIndex)
{
	return RowIndex * (RowIndex - 1) / 2 + ColIndex;
}

/// storage.h QuantumStorage restricted to what TrainingKernels needs: diagonal and strictly lower elements
template <typename DiagDT, typename OffDiagDT = DiagDT>
class QuantumStorage
{
public:
	QuantumStorage() = default;
	DiagDT& operator()(std::size_t i) { return diag[i]; }
	const DiagDT& operator()(std::size_t i) const { return diag[i]; }
	OffDiagDT& operator()(std::size_t i, std::size_t j)
	{
		assert(j < i);
		return off[calculate_offdiagonal_index(i, j)];
	}
	const OffDiagDT& operator()(std::size_t i, std::size_t j) const
	{
		assert(j < i);
		return off[calculate_offdiagonal_index(i, j)];
	}
	std::array<DiagDT, NumPES>& get_diagonal_data() { return diag; }
	const std::array<DiagDT, NumPES>& get_diagonal_data() const { return diag; }
	std::array<OffDiagDT, NumOffDiagonalElements>& get_offdiagonal_data() { return off; }
	const std::array<OffDiagDT, NumOffDiagonalElements>& get_offdiagonal_data() const { return off; }

private:
	std::array<DiagDT, NumPES> diag;
	std::array<OffDiagDT, NumOffDiagonalElements> off;
};
using AllTrainingSets = QuantumStorage<ElementTrainingSet>;

/// predict.h:89-143
class TrainingKernels final: public QuantumStorage<std::optional<TrainingKernel>, std::optional<TrainingComplexKernel>>
{
public:
	using BaseType = QuantumStorage<std::optional<TrainingKernel>, std::optional<TrainingComplexKernel>>;
	TrainingKernels(const QuantumStorage<ParameterVector>& ParameterVectors, const AllTrainingSets& TrainingSets, const bool IsToCalculateError,
		const bool IsToCalculateAverage, const bool IsToCalculateDerivative)
	{
		for (std::size_t i = 0; i < NumPES; i++) // predict.cpp:290-318
			if (num_points(std::get<0>(TrainingSets(i))) != 0)
				(*this)(i).emplace(ParameterVectors(i), TrainingSets(i), IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative);
		for (std::size_t i = 1; i < NumPES; i++) // predict.cpp:328-360
			for (std::size_t j = 0; j < i; j++)
			{
				const ParameterVector& p = ParameterVectors(i, j);
				bool all_zero = true;
				for (double d : p) all_zero = all_zero && d == 0;
				if (num_points(std::get<0>(TrainingSets(i, j))) != 0 && !all_zero)
					(*this)(i, j).emplace(p, TrainingSets(i, j), IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative);
			}
	}
	double calculate_population(void) const // predict.cpp:395-406
	{
		double result = 0.0;
		for (const auto& k : BaseType::get_diagonal_data())
			if (k.has_value()) result += k->get_population();
		return result;
	}
	ClassicalPhaseVector calculate_1st_order_average(void) const // predict.cpp:408-419
	{
		ClassicalPhaseVector result{0.0, 0.0};
		for (const auto& k : BaseType::get_diagonal_data())
			if (k.has_value())
			{
				const ClassicalPhaseVector r = k->get_1st_order_average();
				result[0] += r[0], result[1] += r[1];
			}
		return result;
	}
	template <typename EnergyVector>
	double calculate_total_energy_average(const EnergyVector& Energies) const // predict.cpp:423-436
	{
		double result = 0.0;
		for (std::size_t i = 0; i < NumPES; i++)
			if ((*this)(i).has_value()) result += (*this)(i)->get_population() * Energies[i];
		return result;
	}
	double calculate_purity(void) const // predict.cpp:439-463
	{
		double result = 0.0;
		for (std::size_t i = 0; i < NumPES; i++)
		{
			if ((*this)(i).has_value()) result += (*this)(i)->get_purity();
			for (std::size_t j = 0; j < i; j++)
				if ((*this)(i, j).has_value()) result += 2.0 * (*this)(i, j)->get_purity();
		}
		return result;
	}
	ParameterVector population_derivative(void) const // predict.cpp:465-484
	{
		ParameterVector result(NumPES * KernelBase::NumTotalParameters, 0.0);
		for (std::size_t i = 0; i < NumPES; i++)
			if ((*this)(i).has_value())
			{
				const auto g = (*this)(i)->get_population_derivative();
				std::copy(g.cbegin(), g.cend(), result.begin() + i * KernelBase::NumTotalParameters);
			}
		return result;
	}
	template <typename EnergyVector>
	ParameterVector total_energy_derivative(const EnergyVector& Energies) const // predict.cpp:486-510
	{
		ParameterVector result = population_derivative();
		for (std::size_t i = 0; i < NumPES; i++)
			for (std::size_t k = 0; k < KernelBase::NumTotalParameters; k++) result[i * KernelBase::NumTotalParameters + k] *= Energies[i];
		return result;
	}
	ParameterVector purity_derivative(void) const // predict.cpp:512-559
	{
		ParameterVector result(NumTotalParameters, 0.0);
		std::size_t iParam = 0;
		for (std::size_t i = 0; i < NumPES; i++)
			for (std::size_t j = 0; j <= i; j++)
			{
				if (i == j)
				{
					if ((*this)(i).has_value())
					{
						const auto d = (*this)(i)->get_purity_derivative();
						std::copy(d.cbegin(), d.cend(), result.begin() + iParam);
					}
					iParam += KernelBase::NumTotalParameters;
				}
				else
				{
					if ((*this)(i, j).has_value())
					{
						const auto d = (*this)(i, j)->get_purity_derivative();
						for (std::size_t k = 0; k < d.size(); k++) result[iParam + k] = d[k] * 2;
					}
					iParam += TrainingComplexKernel::NumTotalParameters;
				}
			}
		return result;
	}
};

/// opt.cpp:441-482 with the NLopt objective ABI double(const std::vector<double>&, std::vector<double>&, void*)
using ElementTrainingParameters = std::tuple<const ElementTrainingSet&, const ElementTrainingSet&>;
inline double loose_function(const ParameterVector& x, ParameterVector& grad, void* params)
{
	const auto& [TrainingSet, ExtraTrainingSet] = *static_cast<ElementTrainingParameters*>(params);
	const auto& [Feature, Label] = TrainingSet;
	const auto& [ExtraFeature, ExtraLabel] = ExtraTrainingSet;
	double value = 0.0;
	check(gple_loose_function(context(), x.data(), x.size(), Feature.data(), reinterpret_cast<const double*>(Label.data()), num_points(Feature),
			  ExtraFeature.data(), reinterpret_cast<const double*>(ExtraLabel.data()), num_points(ExtraFeature), &value, grad.empty() ? nullptr : grad.data()),
		context());
	return value;
}

#endif // !PREDICT_H
