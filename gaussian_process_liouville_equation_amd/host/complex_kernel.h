// complex_kernel.h — adapter for gaussian_process_liouville_equation/complex_kernel.h:150-391 (training and predictive
// complex kernels) on top of include/gple.h.
#ifndef COMPLEX_KERNEL_H
#define COMPLEX_KERNEL_H

#include "kernel.h"

/// complex_kernel.h:150-318
class TrainingComplexKernel final
{
public:
	static constexpr std::size_t NumKernels = 2;
	static constexpr std::size_t NumTotalParameters = 1 + NumKernels * (KernelBase::NumTotalParameters - 1) + 1;
	template <typename T>
	using ParameterArray = std::array<T, NumTotalParameters>;

	TrainingComplexKernel(const ParameterVector& Parameter, const ElementTrainingSet& TrainingSet, const bool IsToCalculateError,
		const bool IsToCalculateAverage, const bool IsToCalculateDerivative):
		Params(Parameter), Feature(std::get<0>(TrainingSet)), Flags(flags(IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative))
	{
		assert(Parameter.size() == NumTotalParameters);
		const VectorXcd& label = std::get<1>(TrainingSet);
		gple_complex_fit* h = nullptr;
		check(gple_complex_fit_create(context(), Parameter.data(), Feature.data(), reinterpret_cast<const double*>(label.data()), num_points(Feature), Flags,
				  &Scalars, &h),
			context());
		Handle = std::shared_ptr<gple_complex_fit>(h, [](gple_complex_fit* p) { gple_complex_fit_release(p); });
	}
	const ParameterVector& get_parameters(void) const { return Params; }
	const PhasePoints& get_left_feature(void) const { return Feature; }
	double get_rescale_factor(void) const { return Scalars.rescale_factor; }
	double get_magnitude(void) const { return Scalars.magnitude; }
	MatrixXd get_kernel(void) const
	{
		const std::size_t N = num_points(Feature);
		MatrixXd m(N, N);
		check(gple_complex_fit_get(Handle.get(), GPLE_C_KERNEL, 0, m.data()), context());
		return m;
	}
	MatrixXcd get_pseudo_kernel(void) const { return cmatrix(GPLE_C_PSEUDO); }
	MatrixXcd get_upper_left_block_of_augmented_inverse(void) const { return cmatrix(GPLE_C_UPPER_LEFT); }
	MatrixXcd get_lower_left_block_of_augmented_inverse(void) const { return cmatrix(GPLE_C_LOWER_LEFT); }
	VectorXcd get_upper_part_of_augmented_inverse_times_label(void) const
	{
		VectorXcd v(num_points(Feature));
		check(gple_complex_fit_get(Handle.get(), GPLE_C_INVLBL, 0, reinterpret_cast<double*>(v.data())), context());
		return v;
	}
	double get_error(void) const
	{
		assert(Flags & GPLE_CALC_ERROR);
		return Scalars.error;
	}
	double get_purity(void) const
	{
		assert(Flags & GPLE_CALC_AVERAGE);
		return Scalars.purity;
	}
	ParameterArray<double> get_error_derivative(void) const
	{
		assert((Flags & GPLE_CALC_ERROR) && (Flags & GPLE_CALC_DERIVATIVE));
		ParameterArray<double> r;
		std::copy(Scalars.error_derivative, Scalars.error_derivative + NumTotalParameters, r.begin());
		return r;
	}
	ParameterArray<double> get_purity_derivative(void) const
	{
		assert((Flags & GPLE_CALC_AVERAGE) && (Flags & GPLE_CALC_DERIVATIVE));
		ParameterArray<double> r;
		std::copy(Scalars.purity_derivative, Scalars.purity_derivative + NumTotalParameters, r.begin());
		return r;
	}
	const gple_complex_fit* handle(void) const { return Handle.get(); }

private:
	MatrixXcd cmatrix(gple_complex_array which) const
	{
		const std::size_t N = num_points(Feature);
		MatrixXcd m(N, N);
		check(gple_complex_fit_get(Handle.get(), which, 0, reinterpret_cast<double*>(m.data())), context());
		return m;
	}
	ParameterVector Params;
	PhasePoints Feature;
	unsigned Flags;
	gple_complex_fit_scalars Scalars;
	std::shared_ptr<gple_complex_fit> Handle;
};

/// complex_kernel.h:323-391
class PredictiveComplexKernel final
{
public:
	static constexpr std::size_t NumTotalParameters = TrainingComplexKernel::NumTotalParameters;
	template <typename T>
	using ParameterArray = std::array<T, NumTotalParameters>;

	PredictiveComplexKernel(const PhasePoints& TestFeature, const TrainingComplexKernel& kernel, const bool IsToCalculateDerivative,
		const std::optional<VectorXcd> TestLabel = std::nullopt):
		Prediction(num_points(TestFeature)), ElementwiseVariance(num_points(TestFeature)), CutoffPrediction(num_points(TestFeature)),
		HasLabel(TestLabel.has_value()), HasDerivative(IsToCalculateDerivative)
	{
		check(gple_complex_predict(context(), kernel.handle(), TestFeature.data(), num_points(TestFeature), IsToCalculateDerivative ? GPLE_CALC_DERIVATIVE : 0u,
				  TestLabel.has_value() ? reinterpret_cast<const double*>(TestLabel->data()) : nullptr, reinterpret_cast<double*>(Prediction.data()),
				  ElementwiseVariance.data(), reinterpret_cast<double*>(CutoffPrediction.data()), &Scalars),
			context());
	}
	const VectorXd& get_variance(void) const { return ElementwiseVariance; }
	const VectorXcd& get_cutoff_prediction(void) const { return CutoffPrediction; }
	double get_error(void) const
	{
		assert(HasLabel);
		return Scalars.error;
	}
	ParameterArray<double> get_error_derivative(void) const
	{
		assert(HasLabel && HasDerivative);
		ParameterArray<double> r;
		std::copy(Scalars.error_derivative, Scalars.error_derivative + NumTotalParameters, r.begin());
		return r;
	}

private:
	VectorXcd Prediction;
	VectorXd ElementwiseVariance;
	VectorXcd CutoffPrediction;
	bool HasLabel, HasDerivative;
	gple_predict_scalars Scalars;
};

#endif // !COMPLEX_KERNEL_H
