// dropin_opt.cpp — main.cpp:66-74 and output.cpp:120-132, 236-300 against the Optimization adapter (host/opt.h), in the
// reference's include order with the scaffolding of tests/cpp/ref_env/.  tests/test_host_logic.py compiles it with -fsyntax-only
// on the CPU; tests/test_gpu_adapters.py links it against libgple_hip.so and runs it on the GPU.  The two Monte-Carlo
// observables of predict.cpp that Optimization::optimize calls are host glue the reference keeps (host/predict.h declares
// them); this test defines them from their documentation so that the program links without the reference.
#include "stdafx.h"

#include "opt.h"

#include <cstdio>

/// predict.cpp:104-124: per-coordinate standard deviation of the point positions (unweighted)
ClassicalPhaseVector calculate_standard_deviation_one_surface(const ElementPoints& density)
{
	ClassicalPhaseVector result;
	for (std::size_t d = 0; d < PhaseDim; d++)
	{
		double s = 0.0, s2 = 0.0;
		for (const PhaseSpacePoint& psp : density) s += psp.get<0>()[d], s2 += psp.get<0>()[d] * psp.get<0>()[d];
		result[d] = std::sqrt(s2 / density.size() - (s / density.size()) * (s / density.size()));
	}
	return result;
}
/// predict.cpp:158-206: density-weighted <p^2 / 2m + V_ii(x)> of every populated surface (0 otherwise)
QuantumVector<double> calculate_total_energy_average_each_surface(const AllPoints& density, const ClassicalVector<double>& mass)
{
	QuantumVector<double> result;
	for (std::size_t iPES = 0; iPES < NumPES; iPES++)
	{
		result[iPES] = 0.0;
		const ElementPoints& pts = density(iPES);
		if (pts.empty()) continue;
		std::vector<double> x(pts.size()), pes(6 * pts.size());
		for (std::size_t i = 0; i < pts.size(); i++) x[i] = pts[i].get<0>()[0];
		gple_host::check(gple_pes_adiabatic(gple_host::context(), GPLE_PES_DAC, x.data(), x.size(), 0, pes.data()), gple_host::context());
		double w = 0.0, e = 0.0;
		for (std::size_t i = 0; i < pts.size(); i++)
		{
			const auto& [r, rho] = pts[i];
			w += rho.real(), e += rho.real() * (r[1] * r[1] / (2.0 * mass[0]) + pes[6 * i + (iPES < 2 ? iPES : 1)]);
		}
		result[iPES] = e / w;
	}
	return result;
}

// output.cpp:120-132
static void output_param(std::ostream& os, const Optimization& Optimizer)
{
	const QuantumStorage<ParameterVector> lb = Optimizer.get_lower_bounds(), param = Optimizer.get_parameters(), ub = Optimizer.get_upper_bounds();
	for (const std::size_t iPES : std::ranges::iota_view{0ul, NumPES})
		for (const std::size_t jPES : std::ranges::iota_view{0ul, iPES + 1})
			for (const ParameterVector* v : {&lb(iPES, jPES), &param(iPES, jPES), &ub(iPES, jPES)})
			{
				for (double d : *v) os << ' ' << d;
				os << '\n';
			}
}

int main(int argc, char** argv)
{
	const std::size_t N = argc > 1 ? std::atoi(argv[1]) : 120;
	const bool coherence = argc > 2 && std::atoi(argv[2]) != 0;
	ClassicalVector<double> mass;
	mass[0] = 2000.0;
	ClassicalPhaseVector r0, sigma, rmin, rmax;
	r0[0] = -10.0, r0[1] = 14.112, sigma[0] = 0.7086, sigma[1] = 0.7056;
	for (std::size_t d = 0; d < PhaseDim; d++) rmin[d] = r0[d] - 5.0 * sigma[d], rmax[d] = r0[d] + 5.0 * sigma[d];
	const InitialParameters InitParams(mass, r0, sigma, rmin, rmax);

	// main.cpp:48-66 in outline: points drawn from the initial Gaussian, all population on surface 0
	AllPoints density, extra;
	unsigned long long lcg = 2024;
	auto uni = [&lcg]() { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; return ((lcg >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
	auto normal = [&uni]() { return std::sqrt(-2.0 * std::log(uni())) * std::cos(2.0 * std::numbers::pi * uni()); };
	auto sample = [&](ElementPoints& pts, std::size_t n, double weight, bool cplx)
	{
		for (std::size_t i = 0; i < n; i++)
		{
			ClassicalPhaseVector r;
			for (std::size_t d = 0; d < PhaseDim; d++) r[d] = r0[d] + sigma[d] * normal();
			const double rho = weight * std::exp(-0.5 * (std::pow((r[0] - r0[0]) / sigma[0], 2) + std::pow((r[1] - r0[1]) / sigma[1], 2))) / (2.0 * std::numbers::pi * sigma[0] * sigma[1]);
			pts.emplace_back(r, cplx ? rho * std::exp(std::complex<double>(0.0, 0.5 * (r[0] - r0[0]))) : std::complex<double>(rho));
		}
	};
	sample(density(0), N, coherence ? 0.8 : 1.0, false), sample(extra(0), 2 * N, coherence ? 0.8 : 1.0, false);
	if (coherence)
	{
		sample(density(1, 0), N, 0.4, true), sample(extra(1, 0), 2 * N, 0.4, true);
		sample(density(1), N, 0.2, false), sample(extra(1), 2 * N, 0.2, false);
	}
	// exact values of the synthetic density: population 1, purity of the analytic Gaussians
	const double Purity = coherence ? 0.8 * 0.8 + 0.2 * 0.2 + 2.0 * 0.4 * 0.4 : 1.0;
	const QuantumVector<double> e = calculate_total_energy_average_each_surface(density, mass);
	const double TotalEnergy = coherence ? 0.8 * e[0] + 0.2 * e[1] : e[0];

	Optimization optimizer(InitParams, TotalEnergy, Purity);                            // main.cpp:71
	Optimization::Result opt_result = optimizer.optimize(density, extra);               // main.cpp:73
	const TrainingKernels all_kernels(optimizer.get_parameters(), density);             // main.cpp:74
	const auto& [error, Steps, OptType] = opt_result;                                   // output.cpp:245
	std::printf("opt_error %.17g\nopt_type %d\nopt_steps", error, static_cast<int>(OptType));
	for (std::size_t s : Steps) std::printf(" %zu", s);
	std::printf("\npopulation %.17g\nenergy %.17g %.17g\npurity %.17g %.17g\n", all_kernels.calculate_population(), all_kernels.calculate_total_energy_average(e),
		TotalEnergy, all_kernels.calculate_purity(), Purity);
	output_param(std::cout, optimizer);
	// every parameter inside its bounds
	const QuantumStorage<ParameterVector> lb = optimizer.get_lower_bounds(), ub = optimizer.get_upper_bounds();
	bool inside = true;
	for (std::size_t iPES = 0; iPES < NumPES; iPES++)
		for (std::size_t jPES = 0; jPES <= iPES; jPES++)
			for (std::size_t k = 1; k < lb(iPES, jPES).size(); k++) // [0] is the magnitude, rewritten after the search (opt.cpp:1179-1195)
				inside = inside && optimizer.get_parameters()(iPES, jPES)[k] >= lb(iPES, jPES)[k] && optimizer.get_parameters()(iPES, jPES)[k] <= ub(iPES, jPES)[k];
	std::printf("inside_bounds %d\n", inside ? 1 : 0);
	// a second call starts from the previous parameters (main.cpp:159)
	opt_result = optimizer.optimize(density, extra);
	std::printf("reopt_error %.17g\nreopt_type %d\n", std::get<0>(opt_result), static_cast<int>(std::get<2>(opt_result)));
	return 0;
}
