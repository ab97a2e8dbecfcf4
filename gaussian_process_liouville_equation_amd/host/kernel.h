// kernel.h — adapter with the reference's names and signatures (gaussian_process_liouville_equation/kernel.h:9-403),
// forwarding to the MI355X library through include/gple.h.  Put this directory in front of the reference's on the include path
// (INTEGRATION.md §2): "stdafx.h" below is then the reference's own.  Value semantics are kept: objects are immutable after
// construction and cheap to copy (the fit handle is shared), as predict.cpp:470,522,538 copy-construct optionals of them.
#ifndef KERNEL_H
#define KERNEL_H

#include "stdafx.h"

#include "gple_host.h"

/// kernel.h:10-14
using ParameterVector = std::vector<double>;
using ElementTrainingSet = std::tuple<PhasePoints, Eigen::VectorXcd>;
static constexpr double ConnectingPoint = 2.0; // kernel.h:16

/// kernel.h:22, kernel.cpp:8-31 (pointer identity selects the identity branch, exactly like the reference)
inline Eigen::MatrixXd delta_kernel(const PhasePoints& LeftFeature, const PhasePoints& RightFeature)
{
	const std::size_t R = static_cast<std::size_t>(LeftFeature.cols()), C = static_cast<std::size_t>(RightFeature.cols());
	Eigen::MatrixXd result(R, C);
	const double *l = LeftFeature.data(), *r = RightFeature.data();
	for (std::size_t j = 0; j < C; j++)
		for (std::size_t i = 0; i < R; i++)
			result(i, j) = l == r ? static_cast<double>(i == j) : static_cast<double>(l[2 * i] == r[2 * j] && l[2 * i + 1] == r[2 * j + 1]);
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
		KernelParams(Parameter), LeftFeature(left_feature), RightFeature(right_feature), KernelMatrix(left_feature.cols(), right_feature.cols())
	{
		const std::size_t R = static_cast<std::size_t>(left_feature.cols()), C = static_cast<std::size_t>(right_feature.cols());
		const double theta[4] = {std::get<0>(Parameter), std::get<1>(Parameter)[0], std::get<1>(Parameter)[1], std::get<2>(Parameter)};
		std::vector<double> dk(IsToCalculateDerivative ? 4 * R * C : 0);
		gple_host::check(gple_real_gram(gple_host::context(), theta, left_feature.data(), R, right_feature.data(), C,
							 left_feature.data() == right_feature.data(), 0, KernelMatrix.data(), IsToCalculateDerivative ? dk.data() : nullptr),
			gple_host::context());
		if (IsToCalculateDerivative)
		{
			ParameterArray<Eigen::MatrixXd> d;
			for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
			{
				d[ip] = Eigen::MatrixXd(left_feature.cols(), right_feature.cols());
				std::copy(dk.begin() + ip * R * C, dk.begin() + (ip + 1) * R * C, d[ip].data());
			}
			Derivatives = std::move(d);
		}
	}
	const KernelParameter& get_formatted_parameters(void) const { return KernelParams; }
	const PhasePoints& get_left_feature(void) const { return LeftFeature; }
	const PhasePoints& get_right_feature(void) const { return RightFeature; }
	const Eigen::MatrixXd& get_kernel(void) const { return KernelMatrix; }
	const ParameterArray<Eigen::MatrixXd>& get_derivative(void) const
	{
		assert(Derivatives.has_value());
		return Derivatives.value();
	}

private:
	KernelParameter KernelParams;
	PhasePoints LeftFeature, RightFeature;
	Eigen::MatrixXd KernelMatrix;
	std::optional<ParameterArray<Eigen::MatrixXd>> Derivatives;
};

/// kernel.h:111-280.  Not derived from KernelBase: the N x N Gram stays on the device and get_kernel() / get_inverse()
/// materialise it on request (their only consumers in the reference are the sibling Predictive* constructors).
class TrainingKernel final
{
public:
	static constexpr std::size_t NumTotalParameters = KernelBase::NumTotalParameters;
	template <typename T>
	using ParameterArray = KernelBase::ParameterArray<T>;

	TrainingKernel(const ParameterVector& Parameter, const ElementTrainingSet& TrainingSet, const bool IsToCalculateError,
		const bool IsToCalculateAverage, const bool IsToCalculateDerivative):
		Params(Parameter), Feature(std::get<0>(TrainingSet)),
		Flags(gple_host::flags(IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative))
	{
		assert(Parameter.size() == NumTotalParameters);
		const Eigen::VectorXcd& label = std::get<1>(TrainingSet);
		gple_real_fit* h = nullptr;
		gple_host::check(gple_real_fit_create(gple_host::context(), Parameter.data(), Feature.data(), reinterpret_cast<const double*>(label.data()), 1,
							 static_cast<std::size_t>(Feature.cols()), Flags, &Scalars, &h),
			gple_host::context());
		Handle = std::shared_ptr<gple_real_fit>(h, [](gple_real_fit* p) { gple_real_fit_release(p); });
		FirstOrderAverage[0] = Scalars.first_order_average[0], FirstOrderAverage[1] = Scalars.first_order_average[1];
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
			ErrorDerivatives[ip] = Scalars.error_derivative[ip], PopulationDerivatives[ip] = Scalars.population_derivative[ip],
			PurityDerivatives[ip] = Scalars.purity_derivative[ip];
	}
	const ParameterVector& get_parameters(void) const
	{
		assert(Params.size() == NumTotalParameters);
		return Params;
	}
	KernelBase::KernelParameter get_formatted_parameters(void) const
	{
		ClassicalPhaseVector l;
		l[0] = Params[1], l[1] = Params[2];
		return KernelBase::KernelParameter(Params[0], l, Params[3]);
	}
	const PhasePoints& get_left_feature(void) const { return Feature; }
	const PhasePoints& get_right_feature(void) const { return Feature; }
	double get_rescale_factor(void) const { return Scalars.rescale_factor; }
	Eigen::MatrixXd get_kernel(void) const { return matrix(GPLE_R_KERNEL); }
	Eigen::MatrixXd get_inverse(void) const { return matrix(GPLE_R_INVERSE); }
	Eigen::VectorXd get_inverse_times_label(void) const
	{
		Eigen::VectorXd v(Feature.cols());
		gple_host::check(gple_real_fit_get(Handle.get(), GPLE_R_INVLBL, 0, v.data()), nullptr);
		return v;
	}
	double get_magnitude(void) const { return Scalars.magnitude; } // kernel.h:167-179 (the |.| of a negative square included)
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
	const ClassicalPhaseVector& get_1st_order_average(void) const
	{
		assert(Flags & GPLE_CALC_AVERAGE);
		return FirstOrderAverage;
	}
	double get_purity(void) const
	{
		assert(Flags & GPLE_CALC_AVERAGE);
		return Scalars.purity;
	}
	ParameterArray<Eigen::VectorXd> get_inverse_times_label_derivative(void) const
	{
		assert(Flags & GPLE_CALC_DERIVATIVE);
		const std::size_t N = static_cast<std::size_t>(Feature.cols());
		std::vector<double> buf(NumTotalParameters * N);
		gple_host::check(gple_real_fit_get(Handle.get(), GPLE_R_INVLBL_DERIV, 0, buf.data()), nullptr);
		ParameterArray<Eigen::VectorXd> r;
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
		{
			r[ip] = Eigen::VectorXd(Feature.cols());
			std::copy(buf.begin() + ip * N, buf.begin() + (ip + 1) * N, r[ip].data());
		}
		return r;
	}
	const ParameterArray<double>& get_error_derivative(void) const
	{
		assert((Flags & GPLE_CALC_ERROR) && (Flags & GPLE_CALC_DERIVATIVE));
		return ErrorDerivatives;
	}
	const ParameterArray<double>& get_population_derivative(void) const
	{
		assert((Flags & GPLE_CALC_AVERAGE) && (Flags & GPLE_CALC_DERIVATIVE));
		return PopulationDerivatives;
	}
	const ParameterArray<double>& get_purity_derivative(void) const
	{
		assert((Flags & GPLE_CALC_AVERAGE) && (Flags & GPLE_CALC_DERIVATIVE));
		return PurityDerivatives;
	}
	/// not in the reference: the device fit behind this object (for the sibling PredictiveKernel and the batched predict)
	const gple_real_fit* handle(void) const { return Handle.get(); }
	int get_info(void) const { return Scalars.info; }

private:
	Eigen::MatrixXd matrix(gple_real_array which) const
	{
		Eigen::MatrixXd m(Feature.cols(), Feature.cols());
		gple_host::check(gple_real_fit_get(Handle.get(), which, 0, m.data()), nullptr);
		return m;
	}
	ParameterVector Params;
	PhasePoints Feature;
	unsigned Flags;
	gple_real_fit_scalars Scalars;
	ClassicalPhaseVector FirstOrderAverage;
	ParameterArray<double> ErrorDerivatives, PopulationDerivatives, PurityDerivatives;
	std::shared_ptr<gple_real_fit> Handle;
};

/// kernel.h:285-294
inline KernelBase::KernelParameter construct_purity_auxiliary_kernel_params(const KernelBase::KernelParameter& OriginalParams)
{
	const double OriginalMagnitude = std::get<0>(OriginalParams);
	const ClassicalPhaseVector& OriginalCharLength = std::get<1>(OriginalParams);
	ClassicalPhaseVector char_length;
	char_length[0] = std::numbers::sqrt2 * OriginalCharLength[0], char_length[1] = std::numbers::sqrt2 * OriginalCharLength[1];
	return KernelBase::KernelParameter(OriginalMagnitude * OriginalMagnitude * std::sqrt(OriginalCharLength[0] * OriginalCharLength[1]), char_length, 0.0);
}

/// kernel.h:301-332
template <typename T>
Eigen::VectorXd cutoff_factor(const Eigen::Matrix<T, Eigen::Dynamic, 1>& Prediction, const Eigen::VectorXd& Variance)
{
	assert(Prediction.size() == Variance.size());
	Eigen::VectorXd result(Variance.size());
	gple_host::check(gple_cutoff_factor(gple_host::context(), reinterpret_cast<const double*>(Prediction.data()), !std::is_same_v<T, double>, Variance.data(),
						 static_cast<std::size_t>(Variance.size()), 0, result.data()),
		gple_host::context());
	return result;
}

/// kernel.h:336-403.  Not derived from KernelBase: the M x N Gram K* is never kept (it would be 8.6 GB at the north-star size).
class PredictiveKernel
{
public:
	static constexpr std::size_t NumTotalParameters = KernelBase::NumTotalParameters;
	template <typename T>
	using ParameterArray = KernelBase::ParameterArray<T>;

	PredictiveKernel(const PhasePoints& TestFeature, const TrainingKernel& kernel, const bool IsToCalculateDerivative,
		const std::optional<Eigen::VectorXd> TestLabel = std::nullopt):
		RescaleFactor(kernel.get_rescale_factor()), Prediction(TestFeature.cols()), ElementwiseVariance(TestFeature.cols()),
		CutoffPrediction(TestFeature.cols()), HasLabel(TestLabel.has_value()), HasDerivative(IsToCalculateDerivative)
	{
		gple_host::check(gple_real_predict(gple_host::context(), kernel.handle(), TestFeature.data(), static_cast<std::size_t>(TestFeature.cols()),
							 IsToCalculateDerivative ? GPLE_CALC_DERIVATIVE : 0u, TestLabel.has_value() ? TestLabel->data() : nullptr, Prediction.data(),
							 ElementwiseVariance.data(), CutoffPrediction.data(), &Scalars),
			gple_host::context());
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++) ErrorDerivatives[ip] = Scalars.error_derivative[ip];
	}
	const Eigen::VectorXd& get_variance(void) const { return ElementwiseVariance; }
	const Eigen::VectorXd& get_cutoff_prediction(void) const { return CutoffPrediction; }
	double get_error(void) const
	{
		assert(HasLabel);
		return Scalars.error;
	}
	const ParameterArray<double>& get_error_derivative(void) const
	{
		assert(HasLabel && HasDerivative);
		return ErrorDerivatives;
	}

private:
	double RescaleFactor;
	Eigen::VectorXd Prediction, ElementwiseVariance, CutoffPrediction;
	bool HasLabel, HasDerivative;
	gple_predict_scalars Scalars;
	ParameterArray<double> ErrorDerivatives;
};

#endif // !KERNEL_H
