// complex_kernel.h — adapter for gaussian_process_liouville_equation/complex_kernel.h:14-391 (ComplexKernelBase, the training and
// the predictive complex kernel) on top of include/gple.h.
#ifndef COMPLEX_KERNEL_H
#define COMPLEX_KERNEL_H

#include "stdafx.h"

#include "kernel.h"

/// complex_kernel.h:14-145: covariance K and pseudo-covariance K~ between two point sets and their parameter derivatives,
/// materialised on the host (gple_complex_gram).  opt.cpp uses its statics and ParameterArray (opt.cpp:74-232, 468).
class ComplexKernelBase
{
public:
	static constexpr std::size_t NumKernels = 2;
	static constexpr std::size_t NumTotalParameters = 1 + NumKernels * (KernelBase::NumTotalParameters - 1) + 1;
	static constexpr double RescaleMaximum = KernelBase::RescaleMaximum;
	using KernelParameter = std::tuple<double, std::array<std::tuple<double, ClassicalPhaseVector>, 2>, double>;
	template <typename T>
	using ParameterArray = std::array<T, NumTotalParameters>;

	ComplexKernelBase(const KernelParameter& Parameter, const PhasePoints& left_feature, const PhasePoints& right_feature, const bool IsToCalculateDerivative):
		KernelParams(Parameter), RealParams(sub_kernel(Parameter, 0)), ImagParams(sub_kernel(Parameter, 1)), CorrParams(correlation(RealParams, ImagParams)),
		LeftFeature(left_feature), RightFeature(right_feature), KernelMatrix(left_feature.cols(), right_feature.cols()),
		PseudoKernelMatrix(left_feature.cols(), right_feature.cols())
	{
		const std::size_t R = static_cast<std::size_t>(left_feature.cols()), C = static_cast<std::size_t>(right_feature.cols());
		double theta[NumTotalParameters];
		serialise(Parameter, theta);
		std::vector<double> dk(IsToCalculateDerivative ? NumTotalParameters * R * C : 0), dkt(IsToCalculateDerivative ? 2 * NumTotalParameters * R * C : 0);
		gple_host::check(gple_complex_gram(gple_host::context(), theta, left_feature.data(), R, right_feature.data(), C,
							 left_feature.data() == right_feature.data(), 0, KernelMatrix.data(), reinterpret_cast<double*>(PseudoKernelMatrix.data()),
							 IsToCalculateDerivative ? dk.data() : nullptr, IsToCalculateDerivative ? dkt.data() : nullptr),
			gple_host::context());
		if (IsToCalculateDerivative)
		{
			ParameterArray<Eigen::MatrixXd> d;
			ParameterArray<Eigen::MatrixXcd> dt;
			for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
			{
				d[ip] = Eigen::MatrixXd(left_feature.cols(), right_feature.cols());
				dt[ip] = Eigen::MatrixXcd(left_feature.cols(), right_feature.cols());
				std::copy(dk.begin() + ip * R * C, dk.begin() + (ip + 1) * R * C, d[ip].data());
				std::copy(dkt.begin() + 2 * ip * R * C, dkt.begin() + 2 * (ip + 1) * R * C, reinterpret_cast<double*>(dt[ip].data()));
			}
			Derivatives = std::move(d);
			PseudoDerivatives = std::move(dt);
		}
	}
	const KernelParameter& get_formatted_parameters(void) const { return KernelParams; }
	const KernelBase::KernelParameter& get_real_kernel_parameters(void) const { return RealParams; }
	const KernelBase::KernelParameter& get_imaginary_kernel_parameters(void) const { return ImagParams; }
	const KernelBase::KernelParameter& get_correlation_kernel_parameters(void) const { return CorrParams; }
	const PhasePoints& get_left_feature(void) const { return LeftFeature; }
	const PhasePoints& get_right_feature(void) const { return RightFeature; }
	const Eigen::MatrixXd& get_kernel(void) const { return KernelMatrix; }
	const Eigen::MatrixXcd& get_pseudo_kernel(void) const { return PseudoKernelMatrix; }
	const ParameterArray<Eigen::MatrixXd>& get_derivative(void) const
	{
		assert(Derivatives.has_value());
		return Derivatives.value();
	}
	const ParameterArray<Eigen::MatrixXcd>& get_pseudo_derivative(void) const
	{
		assert(PseudoDerivatives.has_value());
		return PseudoDerivatives.value();
	}
	/// (s, sR, lR_x, lR_p, sI, lI_x, lI_p, sn): the serialised order of complex_kernel.cpp:230-256
	static void serialise(const KernelParameter& p, double* theta)
	{
		theta[0] = std::get<0>(p);
		for (std::size_t k = 0; k < NumKernels; k++)
		{
			theta[1 + 3 * k] = std::get<0>(std::get<1>(p)[k]);
			theta[2 + 3 * k] = std::get<1>(std::get<1>(p)[k])[0];
			theta[3 + 3 * k] = std::get<1>(std::get<1>(p)[k])[1];
		}
		theta[7] = std::get<2>(p);
	}
	static KernelParameter deserialise(const ParameterVector& Parameter)
	{
		assert(Parameter.size() == NumTotalParameters);
		KernelParameter p;
		std::get<0>(p) = Parameter[0];
		for (std::size_t k = 0; k < NumKernels; k++)
		{
			std::get<0>(std::get<1>(p)[k]) = Parameter[1 + 3 * k];
			std::get<1>(std::get<1>(p)[k])[0] = Parameter[2 + 3 * k];
			std::get<1>(std::get<1>(p)[k])[1] = Parameter[3 + 3 * k];
		}
		std::get<2>(p) = Parameter[7];
		return p;
	}

private:
	static KernelBase::KernelParameter sub_kernel(const KernelParameter& p, std::size_t k) // complex_kernel.cpp:142-143: noise 0
	{
		return KernelBase::KernelParameter(std::get<0>(std::get<1>(p)[k]), std::get<1>(std::get<1>(p)[k]), 0.0);
	}
	static KernelBase::KernelParameter correlation(const KernelBase::KernelParameter& re, const KernelBase::KernelParameter& im) // :144-157
	{
		ClassicalPhaseVector l;
		double prod = 1.0;
		for (std::size_t d = 0; d < PhaseDim; d++)
		{
			const double lr = std::get<1>(re)[d], li = std::get<1>(im)[d], ss = lr * lr + li * li;
			prod *= 2.0 * lr * li / ss;
			l[d] = std::sqrt(ss / 2.0);
		}
		return KernelBase::KernelParameter(std::sqrt(std::get<0>(re) * std::get<0>(im) * prod), l, 0.0);
	}
	KernelParameter KernelParams;
	KernelBase::KernelParameter RealParams, ImagParams, CorrParams;
	PhasePoints LeftFeature, RightFeature;
	Eigen::MatrixXd KernelMatrix;
	Eigen::MatrixXcd PseudoKernelMatrix;
	std::optional<ParameterArray<Eigen::MatrixXd>> Derivatives;
	std::optional<ParameterArray<Eigen::MatrixXcd>> PseudoDerivatives;
};

/// complex_kernel.h:150-318.  Not derived from ComplexKernelBase (the N x N matrices stay on the device, getters on request).
class TrainingComplexKernel final
{
public:
	static constexpr std::size_t NumKernels = ComplexKernelBase::NumKernels;
	static constexpr std::size_t NumTotalParameters = ComplexKernelBase::NumTotalParameters;
	template <typename T>
	using ParameterArray = ComplexKernelBase::ParameterArray<T>;

	TrainingComplexKernel(const ParameterVector& Parameter, const ElementTrainingSet& TrainingSet, const bool IsToCalculateError,
		const bool IsToCalculateAverage, const bool IsToCalculateDerivative):
		Params(Parameter), Feature(std::get<0>(TrainingSet)),
		Flags(gple_host::flags(IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative))
	{
		assert(Parameter.size() == NumTotalParameters);
		const Eigen::VectorXcd& label = std::get<1>(TrainingSet);
		gple_complex_fit* h = nullptr;
		gple_host::check(gple_complex_fit_create(gple_host::context(), Parameter.data(), Feature.data(), reinterpret_cast<const double*>(label.data()),
							 static_cast<std::size_t>(Feature.cols()), Flags, &Scalars, &h),
			gple_host::context());
		Handle = std::shared_ptr<gple_complex_fit>(h, [](gple_complex_fit* p) { gple_complex_fit_release(p); });
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
			ErrorDerivatives[ip] = Scalars.error_derivative[ip], PurityDerivatives[ip] = Scalars.purity_derivative[ip];
	}
	const ParameterVector& get_parameters(void) const
	{
		assert(Params.size() == NumTotalParameters);
		return Params;
	}
	ComplexKernelBase::KernelParameter get_formatted_parameters(void) const { return ComplexKernelBase::deserialise(Params); }
	const PhasePoints& get_left_feature(void) const { return Feature; }
	const PhasePoints& get_right_feature(void) const { return Feature; }
	double get_rescale_factor(void) const { return Scalars.rescale_factor; }
	double get_magnitude(void) const { return Scalars.magnitude; } // complex_kernel.h:192-204
	Eigen::MatrixXd get_kernel(void) const
	{
		Eigen::MatrixXd m(Feature.cols(), Feature.cols());
		gple_host::check(gple_complex_fit_get(Handle.get(), GPLE_C_KERNEL, 0, m.data()), nullptr);
		return m;
	}
	Eigen::MatrixXcd get_pseudo_kernel(void) const { return cmatrix(GPLE_C_PSEUDO); }
	Eigen::MatrixXcd get_upper_left_block_of_augmented_inverse(void) const { return cmatrix(GPLE_C_UPPER_LEFT); }
	Eigen::MatrixXcd get_lower_left_block_of_augmented_inverse(void) const { return cmatrix(GPLE_C_LOWER_LEFT); }
	Eigen::VectorXcd get_upper_part_of_augmented_inverse_times_label(void) const
	{
		Eigen::VectorXcd v(Feature.cols());
		gple_host::check(gple_complex_fit_get(Handle.get(), GPLE_C_INVLBL, 0, reinterpret_cast<double*>(v.data())), nullptr);
		return v;
	}
	ParameterArray<Eigen::VectorXcd> get_upper_part_of_augmented_inverse_times_label_derivative(void) const
	{
		assert(Flags & GPLE_CALC_DERIVATIVE);
		const std::size_t N = static_cast<std::size_t>(Feature.cols());
		std::vector<double> buf(2 * NumTotalParameters * N);
		gple_host::check(gple_complex_fit_get(Handle.get(), GPLE_C_INVLBL_DERIV, 0, buf.data()), nullptr);
		ParameterArray<Eigen::VectorXcd> r;
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++)
		{
			r[ip] = Eigen::VectorXcd(Feature.cols());
			std::copy(buf.begin() + 2 * ip * N, buf.begin() + 2 * (ip + 1) * N, reinterpret_cast<double*>(r[ip].data()));
		}
		return r;
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
	const ParameterArray<double>& get_error_derivative(void) const
	{
		assert((Flags & GPLE_CALC_ERROR) && (Flags & GPLE_CALC_DERIVATIVE));
		return ErrorDerivatives;
	}
	const ParameterArray<double>& get_purity_derivative(void) const
	{
		assert((Flags & GPLE_CALC_AVERAGE) && (Flags & GPLE_CALC_DERIVATIVE));
		return PurityDerivatives;
	}
	const gple_complex_fit* handle(void) const { return Handle.get(); }
	int get_info(void) const { return Scalars.info; }

private:
	Eigen::MatrixXcd cmatrix(gple_complex_array which) const
	{
		Eigen::MatrixXcd m(Feature.cols(), Feature.cols());
		gple_host::check(gple_complex_fit_get(Handle.get(), which, 0, reinterpret_cast<double*>(m.data())), nullptr);
		return m;
	}
	ParameterVector Params;
	PhasePoints Feature;
	unsigned Flags;
	gple_complex_fit_scalars Scalars;
	ParameterArray<double> ErrorDerivatives, PurityDerivatives;
	std::shared_ptr<gple_complex_fit> Handle;
};

/// complex_kernel.h:323-391
class PredictiveComplexKernel final
{
public:
	static constexpr std::size_t NumTotalParameters = ComplexKernelBase::NumTotalParameters;
	template <typename T>
	using ParameterArray = ComplexKernelBase::ParameterArray<T>;

	PredictiveComplexKernel(const PhasePoints& TestFeature, const TrainingComplexKernel& kernel, const bool IsToCalculateDerivative,
		const std::optional<Eigen::VectorXcd> TestLabel = std::nullopt):
		RescaleFactor(kernel.get_rescale_factor()), Prediction(TestFeature.cols()), ElementwiseVariance(TestFeature.cols()),
		CutoffPrediction(TestFeature.cols()), HasLabel(TestLabel.has_value()), HasDerivative(IsToCalculateDerivative)
	{
		gple_host::check(gple_complex_predict(gple_host::context(), kernel.handle(), TestFeature.data(), static_cast<std::size_t>(TestFeature.cols()),
							 IsToCalculateDerivative ? GPLE_CALC_DERIVATIVE : 0u, TestLabel.has_value() ? reinterpret_cast<const double*>(TestLabel->data()) : nullptr,
							 reinterpret_cast<double*>(Prediction.data()), ElementwiseVariance.data(), reinterpret_cast<double*>(CutoffPrediction.data()), &Scalars),
			gple_host::context());
		for (std::size_t ip = 0; ip < NumTotalParameters; ip++) ErrorDerivatives[ip] = Scalars.error_derivative[ip];
	}
	const Eigen::VectorXd& get_variance(void) const { return ElementwiseVariance; }
	const Eigen::VectorXcd& get_cutoff_prediction(void) const { return CutoffPrediction; }
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
	Eigen::VectorXcd Prediction;
	Eigen::VectorXd ElementwiseVariance;
	Eigen::VectorXcd CutoffPrediction;
	bool HasLabel, HasDerivative;
	gple_predict_scalars Scalars;
	ParameterArray<double> ErrorDerivatives;
};

#endif // !COMPLEX_KERNEL_H
