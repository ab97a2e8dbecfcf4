// kernel.h — adapter with the reference's names and signatures (gaussian_process_liouville_equation/kernel.h:9-403),
// forwarding to the MI355X library through include/gple.h.  Value semantics are kept: objects are immutable after
// construction and cheap to copy (the fit handle is shared), as predict.cpp:470,522,538 copy-construct optionals of them.
#ifndef KERNEL_H
#define KERNEL_H

#include "gple_host.h"

using namespace gple_host;

/// kernel.h:10-14
using ParameterVector = std::vector<double>;
using ElementTrainingSet = std::tuple<PhasePoints, VectorXcd>;
static constexpr double ConnectingPoint = 2.0; // kernel.h:16

/// kernel.h:22, kernel.cpp:8-31 (pointer identity selects the identity branch, exactly like the reference)
inline MatrixXd delta_kernel(const PhasePoints& LeftFeature, const PhasePoints& RightFeature)
{
	const std::size_t R = num_points(LeftFeature), C = num_points(RightFeature);
	MatrixXd result(R, C);
	for (std::size_t j = 0; j < C; j++)
		for (std::size_t i = 0; i < R; i++)
			result(i, j) = LeftFeature.data() == RightFeature.data()
				? static_cast<double>(i == j)
				: static_cast<double>(LeftFeature.data()[2 * i] == RightFeature.data()[2 * j] && LeftFeature.data()[2 * i + 1] == RightFeature.data()[2 * j + 1]);
	return result;
}

/// kernel.h:29-106
class KernelBase
{
public:
	static constexpr std::size_t NumTotalParameters = 1 + PhaseDim + 1;
	static constexpr double RescaleMaximum = 10.0;
	using KernelParameter = std::tuple<double, ClassicalPhaseVector, double>;
	template <typename T>
	using ParameterArray = std::array<T, NumTotalParameters>;

	KernelBase(const KernelParameter& Parameter, const PhasePoints& left_feature, const PhasePoints& right_feature, const bool IsToCalculateDerivative):
		KernelParams(Parameter), LeftFeature(left_feature), RightFeature(right_feature),
		KernelMatrix(num_points(left_feature), num_points(right_feature))
	{
		const std::size_t R = num_points(left_feature), C = num_points(right_feature);
		const double theta[4] = {std::get<0>(Parameter), std::get<1>(Parameter)[0], std::get<1>(Parameter)[1], std::get<2>(Parameter)};
		std::vector<double> dk(IsToCalculateDerivative ? 4 * R * C : 0);
		check(gple_real_gram(context(), theta, left_feature.data(), R, right_feature.data(), C, left_feature.data() == right_feature.data(), 0,
				  KernelMatrix.data(), IsToCalculateDerivative ? dk.data() : nullptr),
			context());
		if (IsToCalculateDerivative)
		{
			ParameterArray<MatrixXd> d;
			for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
			{
				d[ip] = MatrixXd(R, C);
				std::copy(dk.begin() + ip * R * C, dk.begin() + (ip + 1) * R * C, d[ip].data());
			}
			Derivatives = std::move(d);
		}
	}
	const KernelParameter& get_formatted_parameters(void) const { return KernelParams; }
	const PhasePoints& get_left_feature(void) const { return LeftFeature; }
	const PhasePoints& get_right_feature(void) const { return RightFeature; }
	const MatrixXd& get_kernel(void) const { return KernelMatrix; }
	const ParameterArray<MatrixXd>& get_derivative(void) const
	{
		assert(Derivatives.has_value());
		return Derivatives.value();
	}

private:
	KernelParameter KernelParams;
	PhasePoints LeftFeature, RightFeature;
	MatrixXd KernelMatrix;
	std::optional<ParameterArray<MatrixXd>> Derivatives;
};

/// kernel.h:111-280
class TrainingKernel final
{
public:
	static constexpr std::size_t NumTotalParameters = KernelBase::NumTotalParameters;
	template <typename T>
	using ParameterArray = KernelBase::ParameterArray<T>;

	TrainingKernel(const ParameterVector& Parameter, const ElementTrainingSet& TrainingSet, const bool IsToCalculateError,
		const bool IsToCalculateAverage, const bool IsToCalculateDerivative):
		Params(Parameter), Feature(std::get<0>(TrainingSet)), Flags(flags(IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative))
	{
		assert(Parameter.size() == NumTotalParameters);
		const VectorXcd& label = std::get<1>(TrainingSet);
		gple_real_fit* h = nullptr;
		check(gple_real_fit_create(context(), Parameter.data(), Feature.data(), reinterpret_cast<const double*>(label.data()), 1,
				  num_points(Feature), Flags, &Scalars, &h),
			context());
		Handle = std::shared_ptr<gple_real_fit>(h, [](gple_real_fit* p) { gple_real_fit_release(p); });
	}
	const ParameterVector& get_parameters(void) const { return Params; }
	KernelBase::KernelParameter get_formatted_parameters(void) const { return {Params[0], ClassicalPhaseVector{Params[1], Params[2]}, Params[3]}; }
	const PhasePoints& get_left_feature(void) const { return Feature; }
	const PhasePoints& get_right_feature(void) const { return Feature; }
	double get_rescale_factor(void) const { return Scalars.rescale_factor; }
	MatrixXd get_kernel(void) const { return matrix(GPLE_R_KERNEL); }
	MatrixXd get_inverse(void) const { return matrix(GPLE_R_INVERSE); }
	VectorXd get_inverse_times_label(void) const
	{
		VectorXd v(num_points(Feature));
		check(gple_real_fit_get(Handle.get(), GPLE_R_INVLBL, 0, v.data()), context());
		return v;
	}
	double get_magnitude(void) const { return Scalars.magnitude; }
	double get_error(void) const
	{
		assert(Flags & GPLE_CALC_ERROR);
		return Scalars.error;
	}
	double get_population(void) const
	{
		assert(Flags & GPLE_CALC_AVERAGE);
		return Scalars.population;
	}
	ClassicalPhaseVector get_1st_order_average(void) const
	{
		assert(Flags & GPLE_CALC_AVERAGE);
		return ClassicalPhaseVector{Scalars.first_order_average[0], Scalars.first_order_average[1]};
	}
	double get_purity(void) const
	{
		assert(Flags & GPLE_CALC_AVERAGE);
		return Scalars.purity;
	}
	ParameterArray<VectorXd> get_inverse_times_label_derivative(void) const
	{
		assert(Flags & GPLE_CALC_DERIVATIVE);
		const std::size_t N = num_points(Feature);
		std::vector<double> buf(NumTotalParameters * N);
		check(gple_real_fit_get(Handle.get(), GPLE_R_INVLBL_DERIV, 0, buf.data()), context());
		ParameterArray<VectorXd> r;
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
		{
			r[ip] = VectorXd(N);
			std::copy(buf.begin() + ip * N, buf.begin() + (ip + 1) * N, r[ip].data());
		}
		return r;
	}
	ParameterArray<double> get_error_derivative(void) const
	{
		assert((Flags & GPLE_CALC_ERROR) && (Flags & GPLE_CALC_DERIVATIVE));
		return to_array(Scalars.error_derivative);
	}
	ParameterArray<double> get_population_derivative(void) const
	{
		assert((Flags & GPLE_CALC_AVERAGE) && (Flags & GPLE_CALC_DERIVATIVE));
		return to_array(Scalars.population_derivative);
	}
	ParameterArray<double> get_purity_derivative(void) const
	{
		assert((Flags & GPLE_CALC_AVERAGE) && (Flags & GPLE_CALC_DERIVATIVE));
		return to_array(Scalars.purity_derivative);
	}
	/// for the sibling PredictiveKernel only
	const gple_real_fit* handle(void) const { return Handle.get(); }
	int get_info(void) const { return Scalars.info; }

private:
	static ParameterArray<double> to_array(const double (&a)[4]) { return {a[0], a[1], a[2], a[3]}; }
	MatrixXd matrix(gple_real_array which) const
	{
		const std::size_t N = num_points(Feature);
		MatrixXd m(N, N);
		check(gple_real_fit_get(Handle.get(), which, 0, m.data()), context());
		return m;
	}
	ParameterVector Params;
	PhasePoints Feature;
	unsigned Flags;
	gple_real_fit_scalars Scalars;
	std::shared_ptr<gple_real_fit> Handle;
};

/// kernel.h:285-294
inline KernelBase::KernelParameter construct_purity_auxiliary_kernel_params(const KernelBase::KernelParameter& OriginalParams)
{
	const auto& [OriginalMagnitude, OriginalCharLength, OriginalNoise] = OriginalParams;
	(void)OriginalNoise;
	KernelBase::KernelParameter result;
	auto& [magnitude, char_length, noise] = result;
	magnitude = OriginalMagnitude * OriginalMagnitude * std::sqrt(OriginalCharLength[0] * OriginalCharLength[1]);
	char_length = ClassicalPhaseVector{std::sqrt(2.0) * OriginalCharLength[0], std::sqrt(2.0) * OriginalCharLength[1]};
	noise = 0.0;
	return result;
}

/// kernel.h:301-332
template <typename T>
VectorXd cutoff_factor(const std::conditional_t<std::is_same_v<T, double>, VectorXd, VectorXcd>& Prediction, const VectorXd& Variance)
{
	assert(Prediction.size() == Variance.size());
	VectorXd result(Variance.size());
	check(gple_cutoff_factor(context(), reinterpret_cast<const double*>(Prediction.data()), !std::is_same_v<T, double>, Variance.data(), Variance.size(), 0,
			  result.data()),
		context());
	return result;
}

/// kernel.h:336-403
class PredictiveKernel
{
public:
	static constexpr std::size_t NumTotalParameters = KernelBase::NumTotalParameters;
	template <typename T>
	using ParameterArray = KernelBase::ParameterArray<T>;

	PredictiveKernel(const PhasePoints& TestFeature, const TrainingKernel& kernel, const bool IsToCalculateDerivative,
		const std::optional<VectorXd> TestLabel = std::nullopt):
		RescaleFactor(kernel.get_rescale_factor()), Prediction(num_points(TestFeature)), ElementwiseVariance(num_points(TestFeature)),
		CutoffPrediction(num_points(TestFeature)), HasLabel(TestLabel.has_value()), HasDerivative(IsToCalculateDerivative)
	{
		check(gple_real_predict(context(), kernel.handle(), TestFeature.data(), num_points(TestFeature), IsToCalculateDerivative ? GPLE_CALC_DERIVATIVE : 0u,
				  TestLabel.has_value() ? TestLabel->data() : nullptr, Prediction.data(), ElementwiseVariance.data(), CutoffPrediction.data(), &Scalars),
			context());
	}
	const VectorXd& get_variance(void) const { return ElementwiseVariance; }
	const VectorXd& get_cutoff_prediction(void) const { return CutoffPrediction; }
	double get_error(void) const
	{
		assert(HasLabel);
		return Scalars.error;
	}
	ParameterArray<double> get_error_derivative(void) const
	{
		assert(HasLabel && HasDerivative);
		return {Scalars.error_derivative[0], Scalars.error_derivative[1], Scalars.error_derivative[2], Scalars.error_derivative[3]};
	}

private:
	double RescaleFactor;
	VectorXd Prediction, ElementwiseVariance, CutoffPrediction;
	bool HasLabel, HasDerivative;
	gple_predict_scalars Scalars;
};

#endif // !KERNEL_H
