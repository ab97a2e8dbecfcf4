// dropin_callers.cpp — the call patterns of the reference's translation units against the header adapters, in the
// reference's include order: stdafx.h first (here the scaffolding of tests/cpp/ref_env/ with the same guard and the same
// global names, this image having no Eigen), then the headers main.cpp / opt.cpp / output.cpp include, which resolve to
// gaussian_process_liouville_equation_amd/host/.  tests/test_host_logic.py compiles this file with -fsyntax-only on the CPU
// (no redefinition of NumPES / Dim / PhaseDim / QuantumStorage / calculate_offdiagonal_index, no ambiguity, every member the
// callers use present); tests/test_gpu_adapters.py links it against libgple_hip.so and runs it on the GPU.
//   opt.cpp:74-232    bounds and reparametrisation helpers on ComplexKernelBase / KernelBase statics
//   opt.cpp:441-482   loose_function (static, same name and signature as in opt.cpp: must not collide with the adapters)
//   opt.cpp:622-719   diagonal_constraints on TrainingKernels aggregates
//   opt.cpp:1179-1195 get_magnitude after optimisation
//   main.cpp:74-101   TrainingKernels(params, density) + the predict_distribution lambda
//   main.cpp:140-143  evolve(density, ...), evolve(extra_points, ...), is_very_small(...) with the kernels in place of the lambda (host/evolve.h)
//   mc.cpp:118-165, 349-369  the Metropolis walk of an element's points (host/mc.h)
//   output.cpp:181-233, 262-290  output_phase over the grid, rescale factors in the log line
#include "stdafx.h"

#include "evolve.h"
#include "mc.h"
#include "predict.h"
#include "storage.h"

#include <chrono>
#include <cstdio>
#include <sstream>
#include <thread>

using Bounds = std::array<ParameterVector, 2>;
using ElementTrainingParameters = std::tuple<const ElementTrainingSet&, const ElementTrainingSet&>;                                      // opt.cpp:16
using AnalyticalConstraintParameters = std::tuple<const AllTrainingSets&, const QuantumVector<double>&, const double, const double>; // opt.cpp:22

// opt.cpp:67-104 in outline: sizes and loops run on the statics of both kernel bases
static Bounds complex_kernel_bounds(const ClassicalPhaseVector& lo, const ClassicalPhaseVector& hi)
{
	Bounds result{ParameterVector(ComplexKernelBase::NumTotalParameters), ParameterVector(ComplexKernelBase::NumTotalParameters)};
	auto& [lb, ub] = result;
	std::size_t iParam = 0;
	lb[iParam] = ub[iParam] = 1.0;
	iParam++;
	for ([[maybe_unused]] const std::size_t iKernel : std::ranges::iota_view{0ul, ComplexKernelBase::NumKernels})
	{
		lb[iParam] = 0.1, ub[iParam] = 10.0;
		iParam++;
		for (const std::size_t iDim : std::ranges::iota_view{0ul, PhaseDim}) lb[iParam + iDim] = lo[iDim], ub[iParam + iDim] = hi[iDim];
		iParam += PhaseDim;
	}
	lb[iParam] = ub[iParam] = 1e-2;
	return result;
}
// opt.cpp:109-144
static ParameterVector local_parameter_to_global(const ParameterVector& param)
{
	assert(param.size() == KernelBase::NumTotalParameters || param.size() == ComplexKernelBase::NumTotalParameters);
	ParameterVector result = param;
	if (param.size() == ComplexKernelBase::NumTotalParameters)
	{
		for (std::size_t iKernel = 0; iKernel < ComplexKernelBase::NumKernels; iKernel++) result[1 + iKernel * (1 + PhaseDim)] = std::log(result[1 + iKernel * (1 + PhaseDim)]);
		result.back() = std::log(result.back());
	}
	else
		result[1 + PhaseDim] = std::log(result[1 + PhaseDim]);
	return result;
}

static inline void make_normal(double& d)
{
	if (std::isnan(d) || std::isinf(d)) d = std::numeric_limits<double>::max();
}

// opt.cpp:441-482
static double loose_function(const ParameterVector& x, ParameterVector& grad, void* params)
{
	const auto& [TrainingSet, ExtraTrainingSet] = *static_cast<ElementTrainingParameters*>(params);
	const auto& [ExtraTrainingFeature, ExtraTrainingLabel] = ExtraTrainingSet;
	double result = 0.0;
	if (x.size() == KernelBase::NumTotalParameters)
	{
		const TrainingKernel kernel(x, TrainingSet, true, false, !grad.empty());
		const PredictiveKernel ExtraKernel(ExtraTrainingFeature, kernel, !grad.empty(), ExtraTrainingLabel.real());
		result = kernel.get_error() + ExtraKernel.get_error();
		if (!grad.empty())
		{
			const KernelBase::ParameterArray<double> trn_deriv = kernel.get_error_derivative(), vld_deriv = ExtraKernel.get_error_derivative();
			for (const std::size_t iParam : std::ranges::iota_view{0ul, KernelBase::NumTotalParameters}) grad[iParam] = trn_deriv[iParam] + vld_deriv[iParam];
		}
	}
	else
	{
		const TrainingComplexKernel kernel(x, TrainingSet, true, false, !grad.empty());
		const PredictiveComplexKernel ExtraKernel(ExtraTrainingFeature, kernel, !grad.empty(), ExtraTrainingLabel);
		result = kernel.get_error() + ExtraKernel.get_error();
		if (!grad.empty())
		{
			const ComplexKernelBase::ParameterArray<double> trn_deriv = kernel.get_error_derivative(), vld_deriv = ExtraKernel.get_error_derivative();
			for (const std::size_t iParam : std::ranges::iota_view{0ul, ComplexKernelBase::NumTotalParameters}) grad[iParam] = trn_deriv[iParam] + vld_deriv[iParam];
		}
	}
	make_normal(result);
	for (double& d : grad) make_normal(d);
	return result;
}

// opt.cpp:622-635
static QuantumStorage<ParameterVector> construct_all_parameters_from_diagonal(const double* x)
{
	QuantumStorage<ParameterVector> result;
	for (const std::size_t iPES : std::ranges::iota_view{0ul, NumPES})
		for (const std::size_t jPES : std::ranges::iota_view{0ul, iPES + 1})
			result(iPES, jPES) = iPES == jPES ? ParameterVector(x + iPES * KernelBase::NumTotalParameters, x + (iPES + 1) * KernelBase::NumTotalParameters)
											   : ParameterVector(ComplexKernelBase::NumTotalParameters, 0.0);
	return result;
}
// opt.cpp:644-719 (NLopt m-constraint ABI)
static void diagonal_constraints(const unsigned NumConstraints, double* result, [[maybe_unused]] const unsigned NumParams, const double* x, double* grad, void* params)
{
	[[maybe_unused]] const auto& [TrainingSets, Energies, TotalEnergy, Purity] = *static_cast<AnalyticalConstraintParameters*>(params);
	const TrainingKernels AllKernels(construct_all_parameters_from_diagonal(x), TrainingSets, false, true, grad != nullptr);
	result[0] = AllKernels.calculate_population() - 1.0;
	result[1] = AllKernels.calculate_total_energy_average(Energies) - TotalEnergy;
	if (NumConstraints == 3) result[2] = AllKernels.calculate_purity() - Purity;
	if (grad != nullptr)
	{
		std::size_t iParam = 0;
		const ParameterVector& PplDeriv = AllKernels.population_derivative();
		std::copy(PplDeriv.cbegin(), PplDeriv.cend(), grad + iParam);
		iParam += NumPES * KernelBase::NumTotalParameters;
		const ParameterVector& EngDeriv = AllKernels.total_energy_derivative(Energies);
		std::copy(EngDeriv.cbegin(), EngDeriv.cend(), grad + iParam);
		iParam += NumPES * KernelBase::NumTotalParameters;
		if (NumConstraints == 3)
		{
			const ParameterVector& PrtDeriv = AllKernels.purity_derivative();
			assert(PrtDeriv.size() == NumTotalParameters);
			std::copy(PrtDeriv.cbegin(), PrtDeriv.cbegin() + KernelBase::NumTotalParameters, grad + iParam);
		}
	}
}

// output.cpp:181-233
static void output_phase(std::ostream& phase, std::ostream& variance, const TrainingKernels& AllKernels, const PhasePoints& PhaseGrids)
{
	const std::size_t NumPoints = PhaseGrids.cols();
	for (const std::size_t iPES : std::ranges::iota_view{0ul, NumPES})
		for (const std::size_t jPES : std::ranges::iota_view{0ul, iPES + 1})
		{
			if (iPES == jPES)
			{
				if (AllKernels(iPES).has_value())
				{
					const PredictiveKernel k(PhaseGrids, AllKernels(iPES).value(), false);
					phase << k.get_cutoff_prediction().format(VectorFormatter) << '\n';
					phase << Eigen::VectorXd::Zero(NumPoints).format(VectorFormatter) << '\n';
					variance << k.get_variance().format(VectorFormatter) << '\n';
				}
			}
			else if (AllKernels(iPES, jPES).has_value())
			{
				const PredictiveComplexKernel ck(PhaseGrids, AllKernels(iPES, jPES).value(), false);
				const Eigen::VectorXcd& pred = ck.get_cutoff_prediction();
				phase << pred.real().format(VectorFormatter) << '\n';
				phase << pred.imag().format(VectorFormatter) << '\n';
				variance << ck.get_variance().format(VectorFormatter) << '\n';
			}
		}
}

int main(int argc, char** argv)
{
	// a small synthetic density: NumPES real elements and their coherences, like mc.cpp's selected points
	const std::size_t N = argc > 1 ? std::atoi(argv[1]) : 60;
	AllPoints density, extra;
	unsigned long long lcg = 12345;
	auto uni = [&lcg]() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return (lcg >> 11) * (1.0 / 9007199254740992.0); };
	auto sample = [&](ElementPoints& pts, std::size_t n, bool cplx)
	{
		for (std::size_t i = 0; i < n; i++)
		{
			ClassicalPhaseVector r;
			r[0] = -10.0 + 2.4 * (uni() - 0.5), r[1] = 14.112 + 2.4 * (uni() - 0.5);
			const double rho = std::exp(-0.5 * (std::pow((r[0] + 10.0) / 0.7086, 2) + std::pow((r[1] - 14.112) / 0.7056, 2))) / (2.0 * std::numbers::pi * 0.7086 * 0.7056);
			pts.emplace_back(r, cplx ? 0.5 * rho * std::exp(std::complex<double>(0.0, 0.5 * (r[0] + 10.0))) : std::complex<double>(rho));
		}
	};
	for (std::size_t iPES = 0; iPES < NumPES; iPES++)
		for (std::size_t jPES = 0; jPES <= iPES; jPES++)
			if (!(iPES == NumPES - 1 && jPES == iPES)) // the last diagonal element stays unpopulated (predict.cpp:308-315)
				sample(density(iPES, jPES), N, iPES != jPES), sample(extra(iPES, jPES), 2 * N, iPES != jPES);
	const ParameterVector theta{1.0, 0.7086, 0.7056, 1e-2}, ctheta{1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05};
	QuantumStorage<ParameterVector> params(theta, ctheta);

	// main.cpp:74-101
	std::unique_ptr<TrainingKernels> all_kernels = std::make_unique<TrainingKernels>(params, density);
	const DistributionFunction& predict_distribution = [&all_kernels](const ClassicalPhaseVector& r, const std::size_t RowIndex, const std::size_t ColIndex) -> std::complex<double>
	{
		if (RowIndex == ColIndex)
			return (*all_kernels)(RowIndex).has_value() ? PredictiveKernel(r, (*all_kernels)(RowIndex).value(), false).get_cutoff_prediction().value() : 0.0;
		return (*all_kernels)(RowIndex, ColIndex).has_value()
			? PredictiveComplexKernel(r, (*all_kernels)(RowIndex, ColIndex).value(), false).get_cutoff_prediction().value()
			: 0.0;
	};
	ClassicalPhaseVector r0;
	r0[0] = -10.1, r0[1] = 14.3;
	const std::complex<double> d00 = predict_distribution(r0, 0, 0), d10 = predict_distribution(r0, 1, 0), d11 = predict_distribution(r0, NumPES - 1, NumPES - 1);
	std::printf("point_00 %.17g %.17g\npoint_10 %.17g %.17g\npoint_last %.17g %.17g\n", d00.real(), d00.imag(), d10.real(), d10.imag(), d11.real(), d11.imag());
	std::printf("all_population %.17g\nall_purity %.17g\n", all_kernels->calculate_population(), all_kernels->calculate_purity());
	QuantumVector<double> Energies;
	for (std::size_t i = 0; i < NumPES; i++) Energies[i] = 0.1 * (i + 1);
	std::printf("all_energy %.17g\n", all_kernels->calculate_total_energy_average(Energies));
	const ClassicalPhaseVector mean_r = calculate_1st_order_average_one_surface((*all_kernels)(0).value());
	std::printf("mean_r %.17g %.17g\npopulation_0 %.17g\n", mean_r[0], mean_r[1], calculate_population_one_surface((*all_kernels)(0).value()));
	// output.cpp:262-290
	std::printf("rescale %.17g %.17g\n", (*all_kernels)(0)->get_rescale_factor(), (*all_kernels)(1, 0)->get_rescale_factor());

	// main.cpp:140-143: the tick's three calls, the kernels standing where predict_distribution stood (the printed lines are checked for the
	// two-level system the reference instantiates, evolve.cpp:367-371; compiled for three levels the same calls go to gple_evolve_n)
	if constexpr (NumPES <= 3)
	{
		ClassicalVector<double> mass;
		mass[0] = 2000.0;
		const double dt = 1.0;
		AllPoints moved = density, moved_extra = extra;
		evolve(moved, mass, dt, *all_kernels);
		evolve(moved_extra, mass, dt, *all_kernels);
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			for (std::size_t jPES = 0; jPES <= iPES; jPES++)
			{
				std::printf("evolve_%zu%zu", iPES, jPES);
				for (const PhaseSpacePoint& psp : moved(iPES, jPES))
				{
					const auto& [r, rho] = psp;
					std::printf(" %.17g %.17g %.17g %.17g", r[0], r[1], rho.real(), rho.imag());
				}
				std::printf("\n");
			}
		std::printf("evolve_extra_sizes %zu %zu %zu\n", moved_extra(0).size(), moved_extra(1, 0).size(), moved_extra(1).size());
		const QuantumStorage<bool> IsSmall = is_very_small(density, mass, dt, *all_kernels);
		std::printf("is_small %d %d %d\n", IsSmall(0) ? 1 : 0, IsSmall(1, 0) ? 1 : 0, IsSmall(1) ? 1 : 0);
		const std::complex<double> np = new_point_predict(r0, mass, dt, *all_kernels, NumPES - 1, NumPES - 1);
		std::printf("new_point_11 %.17g %.17g\n", np.real(), np.imag());
		// mc.cpp:118-165 for all points of rho_00 at once, then the selection body of mc.cpp:349-369
		EigenVector<ClassicalPhaseVector> start;
		for (const PhaseSpacePoint& psp : density(0)) start.push_back(psp.get<0>());
		const auto [last, ratio] = gple_host::generate_markov_chain(25, *all_kernels, 0.3, 0, 0, start, 0xC0FFEE1234ULL);
		std::printf("chain_last");
		for (const ClassicalPhaseVector& r : last) std::printf(" %.17g %.17g", r[0], r[1]);
		std::printf("\nchain_ratio");
		for (double a : ratio) std::printf(" %.17g", a);
		std::printf("\n");
		ElementPoints walked = density(0);
		MCParameters mcp(25, 0.3);
		gple_host::element_monte_carlo_walk(walked, mcp, *all_kernels, 0, 0, 0xC0FFEE1234ULL);
		std::printf("walk_rho");
		for (const PhaseSpacePoint& psp : walked) std::printf(" %.17g", psp.get<1>().real());
		std::printf("\nwalk_same_points %d\n", std::equal(walked.begin(), walked.end(), last.begin(), [](const PhaseSpacePoint& a, const ClassicalPhaseVector& b) { return a.get<0>() == b; }) ? 1 : 0);
	}

	// evolve.cpp:392-420, mc.cpp:214-246: the same lambda from several worker threads at once (std::thread standing in for TBB);
	// every thread must get what the lone call returns, and the calls per second are printed for one and for eight threads
	{
		const std::size_t per_thread = 400;
		auto run = [&](std::size_t nthreads) -> std::pair<double, bool>
		{
			std::vector<std::thread> workers;
			std::vector<char> ok(nthreads, 1);
			const auto t0 = std::chrono::steady_clock::now();
			for (std::size_t w = 0; w < nthreads; w++)
				workers.emplace_back([&, w]() {
					for (std::size_t i = 0; i < per_thread; i++)
					{
						const std::complex<double> a = predict_distribution(r0, 0, 0), b = predict_distribution(r0, 1, 0);
						if (a != d00 || b != d10) ok[w] = 0;
					}
				});
			for (std::thread& t : workers) t.join();
			const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			return {2.0 * per_thread * nthreads / dt, std::all_of(ok.begin(), ok.end(), [](char c) { return c != 0; })};
		};
		const auto [rate1, ok1] = run(1);
		const auto [rate8, ok8] = run(8);
		std::printf("threads_rate %.1f %.1f\nthreads_same %d %d\n", rate1, rate8, ok1 ? 1 : 0, ok8 ? 1 : 0);
	}

	// the batched replacement (N1): same values as the point-wise lambda
	gple_host::DistributionBatcher batcher(*all_kernels);
	const std::size_t t00 = batcher.request(r0, 0, 0), t10 = batcher.request(r0, 1, 0), tl = batcher.request(r0, NumPES - 1, NumPES - 1);
	batcher.flush();
	std::printf("batch_00 %.17g %.17g\nbatch_10 %.17g %.17g\nbatch_last %.17g %.17g\n", batcher.result(t00).real(), batcher.result(t00).imag(),
		batcher.result(t10).real(), batcher.result(t10).imag(), batcher.result(tl).real(), batcher.result(tl).imag());
	const std::complex<double> pw = batcher.pointwise()(r0, 1, 0);
	std::printf("pointwise_10 %.17g %.17g\n", pw.real(), pw.imag());

	// opt.cpp:441-482 through the class adapters, and the one-call form
	const AllTrainingSets TrainingSets = construct_training_sets(density), ExtraTrainingSets = construct_training_sets(extra);
	for (const ParameterVector* x : {&theta, &ctheta})
	{
		const bool cplx = x->size() == ComplexKernelBase::NumTotalParameters;
		ElementTrainingParameters etp = cplx ? std::tie(TrainingSets(1, 0), ExtraTrainingSets(1, 0)) : std::tie(TrainingSets(0), ExtraTrainingSets(0));
		ParameterVector grad(x->size()), grad1(x->size()), none;
		const double v = loose_function(*x, grad, &etp), v0 = loose_function(*x, none, &etp);
		gple_host::ElementTrainingParameters etp1 = etp;
		const double v1 = gple_host::loose_function(*x, grad1, &etp1);
		std::printf("%s_loose %.17g %.17g %.17g\n%s_loose_grad", cplx ? "complex" : "real", v, v0, v1, cplx ? "complex" : "real");
		for (double g : grad) std::printf(" %.17g", g);
		std::printf("\n%s_loose_grad_onecall", cplx ? "complex" : "real");
		for (double g : grad1) std::printf(" %.17g", g);
		std::printf("\n");
	}
	const ParameterVector glob = local_parameter_to_global(ctheta);
	const Bounds b = complex_kernel_bounds(r0, r0);
	std::printf("reparam %.17g %.17g %zu\n", glob[1], glob[7], b[0].size());

	// opt.cpp:644-719
	std::vector<double> xd;
	for (std::size_t i = 0; i < NumPES; i++) xd.insert(xd.end(), theta.begin(), theta.end());
	double res[3];
	std::vector<double> cgrad(3 * NumPES * KernelBase::NumTotalParameters, 0.0);
	AnalyticalConstraintParameters acp(TrainingSets, Energies, 0.25, 1.0);
	diagonal_constraints(3, res, NumPES * KernelBase::NumTotalParameters, xd.data(), cgrad.data(), &acp);
	std::printf("constraints %.17g %.17g %.17g\nconstraints_grad", res[0], res[1], res[2]);
	for (double g : cgrad) std::printf(" %.17g", g);
	std::printf("\n");

	// opt.cpp:1179-1195
	std::printf("magnitude %.17g %.17g\n", TrainingKernel(theta, TrainingSets(0), false, false, false).get_magnitude(),
		TrainingComplexKernel(ctheta, TrainingSets(1, 0), false, false, false).get_magnitude());

	// ComplexKernelBase as a class (complex_kernel.h:39-44): consistent with the training kernel's getters
	{
		const PhasePoints& F = std::get<0>(TrainingSets(1, 0));
		const ComplexKernelBase ckb(ComplexKernelBase::deserialise(ctheta), F, F, true);
		const TrainingComplexKernel tck(ctheta, TrainingSets(1, 0), false, false, false);
		const Eigen::MatrixXd Kt = tck.get_kernel();
		const Eigen::MatrixXcd Pt = tck.get_pseudo_kernel();
		double dk = 0.0, dp = 0.0;
		for (Eigen::Index i = 0; i < Kt.size(); i++) dk = std::max(dk, std::abs(Kt.data()[i] - ckb.get_kernel().data()[i])), dp = std::max(dp, std::abs(Pt.data()[i] - ckb.get_pseudo_kernel().data()[i]));
		std::printf("ckb_vs_training %.3g %.3g\nckb_corr_magnitude %.17g\nckb_dkt_1_0 %.17g %.17g\n", dk, dp, std::get<0>(ckb.get_correlation_kernel_parameters()),
			ckb.get_pseudo_derivative()[1].data()[1].real(), ckb.get_pseudo_derivative()[1].data()[1].imag());
	}

	// output.cpp:181-233 on a small grid
	PhasePoints PhaseGrids(PhaseDim, 12);
	for (std::size_t i = 0; i < 12; i++) PhaseGrids(0, i) = -11.0 + 0.2 * i, PhaseGrids(1, i) = 13.5 + 0.1 * i;
	std::ostringstream phase, variance;
	output_phase(phase, variance, *all_kernels, PhaseGrids);
	std::size_t lines = 0;
	for (char c : phase.str()) lines += c == '\n';
	std::printf("phase_lines %zu\nphase_first %s\n", lines, phase.str().substr(0, phase.str().find(' ')).c_str());
	return 0;
}
