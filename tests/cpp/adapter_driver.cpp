// adapter_driver.cpp — exercises the header adapters (host/kernel.h, complex_kernel.h, predict.h) the way the reference's
// callers use the classes (predict.cpp:390-393, output.cpp:204-222, main.cpp:75-101, opt.cpp:441-482) and prints the
// results as "key value..." lines; tests/test_gpu_adapters.py compares them with the Python path.
// Input file: N M  then N lines "x p re im", then M lines "x p".
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../gaussian_process_liouville_equation_amd/host/predict.h"

int main(int argc, char** argv)
{
	if (argc < 2) return 2;
	std::ifstream in(argv[1]);
	std::size_t N, M;
	in >> N >> M;
	PhasePoints X = make_points(N), Xs = make_points(M);
	VectorXcd y(N);
	for (std::size_t i = 0; i < N; i++)
	{
		double re, im;
		in >> X.data()[2 * i] >> X.data()[2 * i + 1] >> re >> im;
		y.data()[i] = {re, im};
	}
	for (std::size_t i = 0; i < M; i++) in >> Xs.data()[2 * i] >> Xs.data()[2 * i + 1];
	const ParameterVector theta{1.0, 0.7086, 0.7056, 1e-2};
	const ParameterVector ctheta{1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05};
	const ElementTrainingSet set{X, y};
	{
		const TrainingKernel k(theta, set, true, true, true);
		std::printf("real_error %.17g\nreal_population %.17g\nreal_purity %.17g\n", k.get_error(), k.get_population(), k.get_purity());
		const auto ed = k.get_error_derivative();
		std::printf("real_error_derivative %.17g %.17g %.17g %.17g\n", ed[0], ed[1], ed[2], ed[3]);
		const PredictiveKernel p(Xs, k, false);
		double s1 = 0, s2 = 0;
		for (std::size_t i = 0; i < M; i++) s1 += p.get_cutoff_prediction().data()[i], s2 += p.get_variance().data()[i];
		std::printf("real_cut_sum %.17g\nreal_var_sum %.17g\n", s1, s2);
		// one-point predict, as main.cpp:83 does per Monte-Carlo step
		PhasePoints one = make_points(1);
		one.data()[0] = Xs.data()[0], one.data()[1] = Xs.data()[1];
		std::printf("real_one_point %.17g\n", PredictiveKernel(one, k, false).get_cutoff_prediction().data()[0]);
		const TrainingKernel copy = k; // value semantics: copies share the device fit
		std::printf("real_copy_error %.17g\n", copy.get_error());
	}
	{
		const TrainingComplexKernel k(ctheta, set, true, true, false);
		std::printf("complex_error %.17g\ncomplex_purity %.17g\n", k.get_error(), k.get_purity());
		const PredictiveComplexKernel p(Xs, k, false);
		double s1 = 0, s2 = 0;
		for (std::size_t i = 0; i < M; i++) s1 += std::abs(p.get_cutoff_prediction().data()[i]), s2 += p.get_variance().data()[i];
		std::printf("complex_cut_abs_sum %.17g\ncomplex_var_sum %.17g\n", s1, s2);
	}
	{
		QuantumStorage<ParameterVector> params;
		params(0) = theta, params(1) = theta, params(1, 0) = ctheta;
		AllTrainingSets sets;
		sets(0) = set, sets(1, 0) = set, sets(1) = ElementTrainingSet{make_points(0), VectorXcd(0)};
		const TrainingKernels all(params, sets, true, true, false);
		std::printf("all_population %.17g\nall_purity %.17g\nall_has_11 %d\n", all.calculate_population(), all.calculate_purity(), (int)all(1).has_value());
	}
	{
		// loose_function through the NLopt objective ABI
		PhasePoints Xe = make_points(M);
		VectorXcd ye(M);
		for (std::size_t i = 0; i < M; i++) Xe.data()[2 * i] = Xs.data()[2 * i], Xe.data()[2 * i + 1] = Xs.data()[2 * i + 1], ye.data()[i] = {0.01 * (i % 7), 0.0};
		const ElementTrainingSet extra{Xe, ye};
		ElementTrainingParameters etp = std::tie(set, extra);
		ParameterVector grad(4), none;
		const double v = loose_function(theta, grad, &etp);
		std::printf("loose_value %.17g\nloose_grad %.17g %.17g %.17g %.17g\nloose_value_nograd %.17g\n", v, grad[0], grad[1], grad[2], grad[3],
			loose_function(theta, none, &etp));
	}
	return 0;
}
