// opt.h — adapter for gaussian_process_liouville_equation/opt.h:13-105: `Optimization` with the reference's constructor, `optimize`,
// getters and result type, on top of the MI355X library.  The objective and constraint evaluations are the library's
// (gple_objective_*, TrainingKernels aggregates); the searches are the library's own Nelder-Mead and augmented Lagrangian
// (include/gple.h, csrc/gple_opt.hip) behind NLopt's C callback ABIs, so NLopt is no longer needed at link time.  What is kept
// from opt.cpp: the parameter layout, the bounds (:33-104, 394-413, 1027-1047), the log reparametrisation of the global tier
// (:109-232), make_normal (:420-431), the element-wise -> diagonal -> full sequence (:1101-1198), the previous / initial /
// global tiers with check_averages and compare_and_overwrite (:1200-1392).  What differs: the iterates of the searches (any two
// implementations' do).  The global tier runs the library's DIRECT-L (gple_objective_minimize_direct_l) where the reference runs NLopt's
// GN_DIRECT_L.  The algorithm arguments of the reference's constructor are accepted and ignored.
#ifndef OPT_H
#define OPT_H

#include "stdafx.h"

#include "input.h"
#include "predict.h"

/// opt.h:13
static constexpr double AverageTolerance = 0.05;

class Optimization final
{
public:
	/// opt.h:20-30
	enum OptimizationType
	{
		Default,
		LocalPrevious,
		LocalInitial,
		Global
	};
	/// opt.h:36
	using Result = std::tuple<double, std::vector<std::size_t>, OptimizationType>;
	using Bounds = std::array<ParameterVector, 2>;

	/// opt.h:47-56 (the five nlopt::algorithm arguments may follow; the searches are the library's own)
	template <typename... Algorithms>
	Optimization(const InitialParameters& InitParams, const double InitialTotalEnergy, const double InitialPurity, Algorithms...):
		TotalEnergy(InitialTotalEnergy), Purity(InitialPurity), mass(InitParams.get_mass()),
		InitialKernelParameter(initial_real(InitParams.get_sigma_r0())), InitialComplexKernelParameter(initial_complex(InitParams.get_sigma_r0())),
		ParameterVectors(InitialKernelParameter, InitialComplexKernelParameter)
	{
		// opt.cpp:394-413: lengths between 1/100 and the size of the phase-space box until the first optimize() narrows them
		ClassicalPhaseVector lo, hi;
		for (std::size_t d = 0; d < PhaseDim; d++) lo[d] = 1.0 / 100.0, hi[d] = InitParams.get_rmax()[d] - InitParams.get_rmin()[d];
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			for (std::size_t jPES = 0; jPES <= iPES; jPES++) ParameterBounds(iPES, jPES) = iPES == jPES ? kernel_bounds(lo, hi) : complex_kernel_bounds(lo, hi);
	}

	/// opt.h:62-65, opt.cpp:1019-1392
	Result optimize(const AllPoints& density, const AllPoints& extra_points)
	{
		const AllTrainingSets TrainingSets = construct_training_sets(density), ExtraTrainingSets = construct_training_sets(extra_points);
		const QuantumVector<double> Energies = calculate_total_energy_average_each_surface(density, mass);
		for (std::size_t iPES = 0; iPES < NumPES; iPES++) // opt.cpp:1027-1047: lengths between sigma / sqrt(N) and 2 sigma of the points
			for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				if (!density(iPES, jPES).empty())
				{
					const ClassicalPhaseVector StdDev = calculate_standard_deviation_one_surface(density(iPES, jPES));
					ClassicalPhaseVector lo, hi;
					for (std::size_t d = 0; d < PhaseDim; d++) lo[d] = StdDev[d] / std::sqrt(static_cast<double>(density(iPES, jPES).size())), hi[d] = 2.0 * StdDev[d];
					ParameterBounds(iPES, jPES) = iPES == jPES ? kernel_bounds(lo, hi) : complex_kernel_bounds(lo, hi);
				}
		Session session(*this, density, TrainingSets, ExtraTrainingSets, Energies);
		// 1. locally from the previous parameters (opt.cpp:1320-1326)
		Result result = session.do_optimize(ParameterVectors, LocalPrevious);
		std::array<double, 3> check_result = session.check_averages(ParameterVectors);
		if (all_zero(check_result)) return result;
		// 2. locally from the initial parameters (:1327-1343)
		QuantumStorage<ParameterVector> param_vec_initial(InitialKernelParameter, InitialComplexKernelParameter);
		const Result result_initial = session.do_optimize(param_vec_initial, LocalInitial);
		compare_and_overwrite(result, check_result, result_initial, session.check_averages(param_vec_initial), param_vec_initial);
		if (all_zero(check_result)) return result;
		// 3. globally in the log-parameter box, then locally from there (:1344-1386)
		QuantumStorage<ParameterVector> param_vec_global(InitialKernelParameter, InitialComplexKernelParameter);
		session.move_into_bounds(param_vec_global);
		const std::vector<std::size_t> steps_global = session.global_elementwise(param_vec_global);
		Result result_global = session.do_optimize(param_vec_global, Global);
		for (std::size_t i = 0; i < steps_global.size(); i++) std::get<1>(result_global)[i] += steps_global[i];
		compare_and_overwrite(result, check_result, result_global, session.check_averages(param_vec_global), param_vec_global);
		return result;
	}

	/// opt.h:69-72
	const QuantumStorage<ParameterVector>& get_parameters(void) const { return ParameterVectors; }
	/// opt.h:76, 80; opt.cpp:1394-1418
	QuantumStorage<ParameterVector> get_lower_bounds(void) const { return stack_bounds(0); }
	QuantumStorage<ParameterVector> get_upper_bounds(void) const { return stack_bounds(1); }

	static constexpr double InitialMagnitude = 1.0; // opt.cpp:25
	static constexpr double InitialNoise = 1e-2;    // opt.cpp:27

private:
	static ParameterVector initial_real(const ClassicalPhaseVector& rSigma) // opt.cpp:286-303
	{
		return ParameterVector{InitialMagnitude, rSigma[0], rSigma[1], InitialNoise};
	}
	static ParameterVector initial_complex(const ClassicalPhaseVector& rSigma) // opt.cpp:304-330
	{
		return ParameterVector{InitialMagnitude, InitialMagnitude, rSigma[0], rSigma[1], InitialMagnitude, rSigma[0], rSigma[1], InitialNoise};
	}
	static Bounds kernel_bounds(const ClassicalPhaseVector& lo, const ClassicalPhaseVector& hi) // opt.cpp:33-61: only the lengths move
	{
		return Bounds{ParameterVector{InitialMagnitude, lo[0], lo[1], InitialNoise}, ParameterVector{InitialMagnitude, hi[0], hi[1], InitialNoise}};
	}
	static Bounds complex_kernel_bounds(const ClassicalPhaseVector& lo, const ClassicalPhaseVector& hi) // opt.cpp:67-104: sub-kernel weights a decade either way
	{
		return Bounds{ParameterVector{InitialMagnitude, InitialMagnitude / 10.0, lo[0], lo[1], InitialMagnitude / 10.0, lo[0], lo[1], InitialNoise},
			ParameterVector{InitialMagnitude, InitialMagnitude * 10.0, hi[0], hi[1], InitialMagnitude * 10.0, hi[0], hi[1], InitialNoise}};
	}
	QuantumStorage<ParameterVector> stack_bounds(std::size_t which) const
	{
		QuantumStorage<ParameterVector> result;
		for (std::size_t iPES = 0; iPES < NumPES; iPES++)
			for (std::size_t jPES = 0; jPES <= iPES; jPES++) result(iPES, jPES) = ParameterBounds(iPES, jPES)[which];
		return result;
	}
	static bool all_zero(const std::array<double, 3>& c) { return c[0] == 0.0 && c[1] == 0.0 && c[2] == 0.0; }
	static void make_normal(double& d) // opt.cpp:420-431
	{
		if (std::isnan(d) || std::isinf(d)) d = std::numeric_limits<double>::max();
	}
	static std::vector<std::size_t> log_indices(std::size_t n) // opt.cpp:109-144: noise (+ the two sub-kernel weights)
	{
		return n == ComplexKernelBase::NumTotalParameters ? std::vector<std::size_t>{1, 4, 7} : std::vector<std::size_t>{3};
	}

	/// opt.cpp:1258-1318
	void compare_and_overwrite(Result& result, std::array<double, 3>& check_result, const Result& result_new, const std::array<double, 3>& check_new,
		const QuantumStorage<ParameterVector>& param_vec_new)
	{
		auto& [error, steps, type] = result;
		const auto& [error_new, steps_new, type_new] = result_new;
		std::size_t BetterResults = 0, WorseResults = 0;
		double sum_new = 0.0, sum_old = 0.0;
		for (std::size_t i = 0; i < 3; i++)
		{
			BetterResults += check_new[i] < check_result[i] && check_result[i] > 2.0 * AverageTolerance;
			WorseResults += check_new[i] > check_result[i] && check_new[i] > 2.0 * AverageTolerance;
			sum_new += check_new[i], sum_old += check_result[i];
		}
		if (BetterResults > WorseResults || (BetterResults == WorseResults && (sum_new < sum_old || error_new < error)))
		{
			ParameterVectors = param_vec_new;
			error = error_new;
			for (std::size_t i = 0; i < steps_new.size() && i < steps.size(); i++) steps[i] += steps_new[i];
			type = type_new;
			check_result = check_new;
		}
	}

	/// one optimize() call: the data of every element resident on the device, the stages of do_optimize (opt.cpp:1101-1198)
	struct Session
	{
		Optimization& self;
		const AllPoints& density;
		const AllTrainingSets &TrainingSets, &ExtraTrainingSets;
		const QuantumVector<double>& Energies;
		QuantumStorage<gple_objective*> objectives; // loose_function's `void* params`, uploaded once
		bool want_purity = true;
		double purity_target = 0.0;

		Session(Optimization& s, const AllPoints& d, const AllTrainingSets& ts, const AllTrainingSets& ets, const QuantumVector<double>& e):
			self(s), density(d), TrainingSets(ts), ExtraTrainingSets(ets), Energies(e), objectives(nullptr, nullptr)
		{
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					const auto& [X, y] = TrainingSets(iPES, jPES);
					const auto& [Xe, ye] = ExtraTrainingSets(iPES, jPES);
					if (X.size() == 0) continue;
					gple_host::check(gple_objective_create(gple_host::context(), X.data(), reinterpret_cast<const double*>(y.data()), static_cast<std::size_t>(X.cols()),
										 Xe.data(), reinterpret_cast<const double*>(ye.data()), static_cast<std::size_t>(Xe.cols()), &objectives(iPES, jPES)),
						gple_host::context());
				}
		}
		Session(const Session&) = delete;
		~Session()
		{
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++) gple_objective_release(objectives(iPES, jPES));
		}
		static std::size_t width(std::size_t iPES, std::size_t jPES) { return iPES == jPES ? KernelBase::NumTotalParameters : ComplexKernelBase::NumTotalParameters; }
		void move_into_bounds(QuantumStorage<ParameterVector>& pv) const // opt.cpp:1055-1068
		{
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
					for (std::size_t k = 0; k < width(iPES, jPES); k++)
						pv(iPES, jPES)[k] = std::clamp(pv(iPES, jPES)[k], self.ParameterBounds(iPES, jPES)[0][k], self.ParameterBounds(iPES, jPES)[1][k]);
		}
		/// loose_function (opt.cpp:441-482) of one element at x, gradient on request
		double element_loose(std::size_t iPES, std::size_t jPES, const double* x, double* grad) const
		{
			double value = 0.0;
			gple_host::check(gple_objective_eval(objectives(iPES, jPES), x, width(iPES, jPES), &value, grad), gple_host::context());
			return value;
		}
		/// optimize_elementwise (opt.cpp:517-587) with the local search
		Result elementwise(QuantumStorage<ParameterVector>& pv) const
		{
			double total_error = 0.0;
			std::vector<std::size_t> num_steps;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					if (!objectives(iPES, jPES))
					{
						num_steps.push_back(0);
						continue;
					}
					double err = 0.0;
					int n_eval = 0;
					gple_objective* const obj = objectives(iPES, jPES);
					const Bounds& b = self.ParameterBounds(iPES, jPES);
					const int st = gple_objective_minimize_neldermead(&obj, 1, width(iPES, jPES), b[0].data(), b[1].data(), nullptr, pv(iPES, jPES).data(), &err, &n_eval);
					if (st != GPLE_OK) err = 0.0; // opt.cpp:549-562: a failed search keeps what it had
					total_error += err;
					num_steps.push_back(static_cast<std::size_t>(n_eval));
				}
			return Result(total_error, num_steps, Default);
		}
		/// the global tier's element-wise stage (opt.cpp:1344-1365): GN_DIRECT_L (opt.h:54) on the same objective in log-parameters
		/// (loose_function_global_wrapper, :489-497) — the library's DIRECT-L on the resident objective, MaximumEvaluations and the
		/// tolerances of opt.cpp:339-350; pv in, pv out in normal parameters
		std::vector<std::size_t> global_elementwise(QuantumStorage<ParameterVector>& pv) const
		{
			std::vector<std::size_t> num_steps;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					if (!objectives(iPES, jPES))
					{
						num_steps.push_back(0);
						continue;
					}
					const std::size_t n = width(iPES, jPES);
					const std::vector<std::size_t> logs = log_indices(n);
					std::vector<unsigned char> is_log(n, 0);
					for (std::size_t i : logs) is_log[i] = 1;
					auto to_global = [&logs](ParameterVector v) {
						for (std::size_t i : logs) v[i] = std::log(v[i]);
						return v;
					};
					const Bounds& b = self.ParameterBounds(iPES, jPES);
					const ParameterVector lb = to_global(b[0]), ub = to_global(b[1]);
					ParameterVector x = to_global(pv(iPES, jPES));
					gple_objective* obj = objectives(iPES, jPES);
					const gple_opt_options opt{1e-5, 1e-5, 1e-15, 1e-15, 0.5, 100000}; // MaximumEvaluations, opt.cpp:339
					double fx = 0.0;
					int n_eval = 0;
					if (gple_objective_minimize_direct_l(&obj, 1, n, lb.data(), ub.data(), is_log.data(), &opt, x.data(), &fx, &n_eval) == GPLE_OK)
					{
						for (std::size_t i : logs) x[i] = std::exp(x[i]); // global_parameter_to_local, opt.cpp:197-232
						pv(iPES, jPES) = x;
					} // opt.cpp:549-562: a failed search keeps what it had
					num_steps.push_back(static_cast<std::size_t>(n_eval));
				}
			return num_steps;
		}
		/// diagonal_loose / full_loose (opt.cpp:594-617, 844-870) as an nlopt_func over the packed parameter vector
		static double packed_loose(unsigned n, const double* x, double* grad, void* data)
		{
			const Session& s = *static_cast<const Session*>(data);
			const bool diagonal_only = n == NumPES * KernelBase::NumTotalParameters;
			double err = 0.0;
			std::size_t pos = 0;
			if (grad) std::fill(grad, grad + n, 0.0);
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					if (diagonal_only && iPES != jPES) continue;
					const std::size_t w = width(iPES, jPES);
					if (s.objectives(iPES, jPES)) err += s.element_loose(iPES, jPES, x + pos, grad ? grad + pos : nullptr);
					pos += w;
				}
			make_normal(err);
			if (grad)
				for (unsigned i = 0; i < n; i++) make_normal(grad[i]);
			return err;
		}
		/// diagonal_constraints / full_constraints (opt.cpp:644-719, 879-929) as an nlopt_mfunc
		static void packed_constraints(unsigned m, double* result, unsigned n, const double* x, double* grad, void* data)
		{
			const Session& s = *static_cast<const Session*>(data);
			const bool diagonal_only = n == NumPES * KernelBase::NumTotalParameters;
			QuantumStorage<ParameterVector> pv;
			std::size_t pos = 0;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					const std::size_t w = width(iPES, jPES);
					if (diagonal_only && iPES != jPES) pv(iPES, jPES) = ParameterVector(w, 0.0); // opt.cpp:622-635: no off-diagonal kernels
					else pv(iPES, jPES) = ParameterVector(x + pos, x + pos + w), pos += w;
				}
			const TrainingKernels AllKernels(pv, s.TrainingSets, false, true, grad != nullptr);
			result[0] = AllKernels.calculate_population() - 1.0;
			result[1] = AllKernels.calculate_total_energy_average(s.Energies) - s.self.TotalEnergy;
			if (m == 3) result[2] = AllKernels.calculate_purity() - s.purity_target;
			for (unsigned i = 0; i < m; i++) make_normal(result[i]);
			if (!grad) return;
			std::fill(grad, grad + static_cast<std::size_t>(m) * n, 0.0);
			const ParameterVector ppl = AllKernels.population_derivative(), eng = AllKernels.total_energy_derivative(s.Energies), prt = AllKernels.purity_derivative();
			// the diagonal blocks of the population / energy gradients sit at the diagonal elements' slots of the packed vector
			std::size_t packed = 0, full = 0;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					const std::size_t w = width(iPES, jPES);
					if (!(diagonal_only && iPES != jPES))
					{
						for (std::size_t k = 0; k < w; k++)
						{
							if (iPES == jPES) grad[packed + k] = ppl[iPES * KernelBase::NumTotalParameters + k], grad[n + packed + k] = eng[iPES * KernelBase::NumTotalParameters + k];
							if (m == 3) grad[2 * n + packed + k] = prt[full + k];
						}
						packed += w;
					}
					full += w;
				}
			for (std::size_t i = 0; i < static_cast<std::size_t>(m) * n; i++) make_normal(grad[i]);
		}
		/// optimize_diagonal / optimize_full (opt.cpp:730-800, 940-1015)
		Result constrained(QuantumStorage<ParameterVector>& pv, bool diagonal_only, double purity)
		{
			std::vector<double> x, lb, ub;
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
				{
					if (diagonal_only && iPES != jPES) continue;
					const Bounds& b = self.ParameterBounds(iPES, jPES);
					x.insert(x.end(), pv(iPES, jPES).begin(), pv(iPES, jPES).end());
					lb.insert(lb.end(), b[0].begin(), b[0].end());
					ub.insert(ub.end(), b[1].begin(), b[1].end());
				}
			purity_target = purity;
			const unsigned m = std::isnan(purity) ? 2u : 3u; // opt.cpp:1143-1152: a NaN purity drops the purity constraint
			double err = 0.0;
			int n_eval = 0;
			const int st = gple_minimize_auglag_eq(&Session::packed_loose, this, &Session::packed_constraints, this, m, static_cast<unsigned>(x.size()), lb.data(), ub.data(),
				nullptr, x.data(), &err, &n_eval);
			if (st == GPLE_OK)
			{
				std::size_t pos = 0;
				for (std::size_t iPES = 0; iPES < NumPES; iPES++)
					for (std::size_t jPES = 0; jPES <= iPES; jPES++)
					{
						if (diagonal_only && iPES != jPES) continue;
						std::copy(x.begin() + pos, x.begin() + pos + width(iPES, jPES), pv(iPES, jPES).begin());
						pos += width(iPES, jPES);
					}
			}
			return Result(err, std::vector<std::size_t>{static_cast<std::size_t>(n_eval)}, Default);
		}
		/// opt.cpp:1101-1198
		Result do_optimize(QuantumStorage<ParameterVector>& pv, const OptimizationType OptType)
		{
			const bool OffDiagonalOptimization = std::any_of(density.get_offdiagonal_data().cbegin(), density.get_offdiagonal_data().cend(),
				[](const ElementPoints& points) { return !points.empty(); });
			for (std::size_t iPES = 0; iPES < NumPES; iPES++)
				for (std::size_t jPES = 0; jPES <= iPES; jPES++) pv(iPES, jPES)[0] = InitialMagnitude;
			move_into_bounds(pv);
			Result result = elementwise(pv);
			auto& [err, steps, type] = result;
			type = OptType;
			if (OffDiagonalOptimization)
			{
				const Result diag = constrained(pv, true, std::numeric_limits<double>::quiet_NaN());
				const Result full = constrained(pv, false, self.Purity);
				err = std::get<0>(full);
				steps.push_back(std::get<1>(diag)[0]);
				steps.push_back(std::get<1>(full)[0]);
			}
			else
			{
				const Result diag = constrained(pv, true, self.Purity);
				err = std::get<0>(diag);
				steps.push_back(std::get<1>(diag)[0]);
				steps.push_back(0);
			}
			for (std::size_t iPES = 0; iPES < NumPES; iPES++) // opt.cpp:1179-1195: the magnitude is read off the unit-magnitude fit
				for (std::size_t jPES = 0; jPES <= iPES; jPES++)
					if (!density(iPES, jPES).empty())
						pv(iPES, jPES)[0] = iPES == jPES ? TrainingKernel(pv(iPES, jPES), TrainingSets(iPES, jPES), false, false, false).get_magnitude()
														   : TrainingComplexKernel(pv(iPES, jPES), TrainingSets(iPES, jPES), false, false, false).get_magnitude();
			return result;
		}
		/// opt.cpp:1200-1256: relative errors of population, energy and purity beyond the tolerance (0 inside it)
		std::array<double, 3> check_averages(const QuantumStorage<ParameterVector>& pv) const
		{
			const TrainingKernels AllKernels(pv, TrainingSets, false, true, false);
			auto beyond = [](double calc, double ref) {
				const double e = std::abs((calc / ref) - 1.0);
				return e < AverageTolerance ? 0.0 : e;
			};
			return {beyond(AllKernels.calculate_population(), 1.0), beyond(AllKernels.calculate_total_energy_average(Energies), self.TotalEnergy),
				beyond(AllKernels.calculate_purity(), self.Purity)};
		}
	};

	const double TotalEnergy, Purity;
	const ClassicalVector<double> mass;
	const ParameterVector InitialKernelParameter, InitialComplexKernelParameter;
	QuantumStorage<Bounds> ParameterBounds;
	QuantumStorage<ParameterVector> ParameterVectors;
};

#endif // !OPT_H
