// gple_opt.hip — the searches of the hyper-parameter optimisation, host code of the library.
//
// The reference drives NLopt (un-vendored, unpinned): LN_NELDERMEAD for the element-wise stage (opt.cpp:517-587), AUGLAG_EQ
// over a gradient-based local solver for the constrained stages (opt.cpp:730-800, 940-1015), with xtol_rel = ftol_rel = 1e-5,
// xtol_abs = ftol_abs = 1e-15 and an initial step of 0.5 for the derivative-free search (opt.cpp:342-355).  NLopt is not in
// this image; these are own implementations of the published algorithms behind NLopt's C callback ABIs (nlopt_func,
// nlopt_mfunc), so the reference's objective / constraint callbacks plug in unchanged:
//   * Nelder-Mead (Nelder & Mead 1965) inside a box: trial points are clipped to the bounds, coordinates with lb == ub stay fixed;
//   * augmented Lagrangian for equality constraints (Conn, Gould, Toint 1991 / Birgin & Martinez 2008, the scheme NLopt's
//     AUGLAG_EQ follows): minimise f + sum lambda_i h_i + rho / 2 sum h_i^2 in the box, lambda += rho h, rho *= 10 when the
//     infeasibility did not shrink to a quarter; inner solver: projected BFGS with Armijo backtracking.
//   * DIRECT-L (Jones, Perttunen, Stuckman 1993; Gablonsky & Kelley 2001, "A locally-biased form of the DIRECT algorithm") for the
//     global tier (GN_DIRECT_L, opt.h:54, opt.cpp:336, 1344-1365): rectangles measured by their longest side, one rectangle per
//     size on the lower hull, every longest side trisected in the order of the better of its two new values — the variant and
//     the settings NLopt's cdirect runs for GN_DIRECT_L (epsilon = 0; stop when an iteration improves the minimum by less than
//     ftol, when every rectangle divided in an iteration is below xtol, or at maxeval).  All new centres of an iteration are
//     independent: they form ONE batch, which the resident-objective form spreads over its contexts.
// Iterates differ from NLopt's (as any two implementations' do); tolerances, stopping tests and the callback ABIs are kept.
// gple_objective_minimize_neldermead evaluates the simplex vertices CONCURRENTLY on several resident objectives (one context =
// one HIP stream each): the n + 1 start vertices, and per iteration the reflected, expanded and both contracted points
// speculatively — the decisions, hence the result, are those of the sequential algorithm.
#include <algorithm>
#include <cmath>
#include <functional>
#include <future>
#include <limits>
#include <map>
#include <numeric>
#include <set>
#include <vector>

#include "../../include/gple.h"

namespace
{
	struct Box
	{
		std::vector<double> lb, ub;
		std::vector<int> free; // indices with lb < ub
	};
	Box make_box(unsigned n, const double* lb, const double* ub)
	{
		Box b;
		b.lb.assign(n, -std::numeric_limits<double>::infinity());
		b.ub.assign(n, std::numeric_limits<double>::infinity());
		for (unsigned i = 0; i < n; ++i)
		{
			if (lb) b.lb[i] = lb[i];
			if (ub) b.ub[i] = ub[i];
			if (b.ub[i] > b.lb[i]) b.free.push_back(static_cast<int>(i));
		}
		return b;
	}
	double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
	gple_opt_options defaults(const gple_opt_options* o)
	{
		gple_opt_options d{1e-5, 1e-5, 1e-15, 1e-15, 0.5, 0}; // opt.cpp:344-346
		if (o)
		{
			d = *o;
			if (!(d.initial_step > 0)) d.initial_step = 0.5;
		}
		return d;
	}

	// evaluates a batch of points; the sequential form calls f once per point, the concurrent form spreads them over contexts
	using BatchEval = std::function<void(const std::vector<std::vector<double>>&, std::vector<double>&)>;

	int nelder_mead(const BatchEval& eval, unsigned n, const Box& box, const gple_opt_options& opt, double* x, double* fmin, int* n_eval)
	{
		const int nf = static_cast<int>(box.free.size());
		int evals = 0;
		std::vector<double> x0(x, x + n);
		for (unsigned i = 0; i < n; ++i) x0[i] = clip(x0[i], box.lb[i], box.ub[i]);
		auto expand = [&](const std::vector<double>& z) {
			std::vector<double> full = x0;
			for (int k = 0; k < nf; ++k) full[box.free[k]] = z[k];
			return full;
		};
		auto evaluate = [&](const std::vector<std::vector<double>>& zs) {
			std::vector<std::vector<double>> pts;
			for (const auto& z : zs) pts.push_back(expand(z));
			std::vector<double> vals(pts.size());
			eval(pts, vals);
			evals += static_cast<int>(pts.size());
			for (double& v : vals)
				if (std::isnan(v)) v = std::numeric_limits<double>::max();
			return vals;
		};
		if (nf == 0)
		{
			*fmin = evaluate({{}})[0];
			if (n_eval) *n_eval = evals;
			return GPLE_OK;
		}
		std::vector<double> lo(nf), hi(nf);
		for (int k = 0; k < nf; ++k) lo[k] = box.lb[box.free[k]], hi[k] = box.ub[box.free[k]];
		auto project = [&](std::vector<double> z) {
			for (int k = 0; k < nf; ++k) z[k] = clip(z[k], lo[k], hi[k]);
			return z;
		};
		// start simplex: x0 and x0 + h e_k (-h when +h leaves the box, the middle of the box when both do)
		std::vector<std::vector<double>> S(nf + 1, std::vector<double>(nf));
		for (int k = 0; k < nf; ++k) S[0][k] = x0[box.free[k]];
		for (int k = 0; k < nf; ++k)
		{
			S[k + 1] = S[0];
			double v = S[0][k] + opt.initial_step;
			if (v > hi[k]) v = S[0][k] - opt.initial_step;
			if (v < lo[k] || v == S[0][k]) v = 0.5 * (lo[k] + hi[k]);
			S[k + 1][k] = v;
		}
		std::vector<double> F = evaluate(S);
		const int maxeval = opt.max_eval > 0 ? opt.max_eval : 400 * nf;
		std::vector<int> order(nf + 1);
		for (;;)
		{
			std::iota(order.begin(), order.end(), 0);
			std::sort(order.begin(), order.end(), [&](int a, int b) { return F[a] < F[b]; });
			const int best = order[0], worst = order[nf], second = order[nf - 1];
			// stopping tests (NLopt's conventions): function values and simplex extent, relative or absolute
			const double fl = F[best], fh = F[worst];
			bool xconv = true;
			for (int k = 0; k < nf && xconv; ++k)
			{
				double mn = S[0][k], mx = S[0][k];
				for (int v = 1; v <= nf; ++v) mn = std::min(mn, S[v][k]), mx = std::max(mx, S[v][k]);
				if (mx - mn > opt.xtol_abs && mx - mn > opt.xtol_rel * std::max(std::fabs(mn), std::fabs(mx))) xconv = false;
			}
			const bool fconv = std::fabs(fh - fl) <= opt.ftol_abs || std::fabs(fh - fl) <= opt.ftol_rel * 0.5 * (std::fabs(fh) + std::fabs(fl));
			if (xconv || fconv || evals >= maxeval) break;
			std::vector<double> c(nf, 0.0); // centroid of all but the worst
			for (int v = 0; v <= nf; ++v)
				if (v != worst)
					for (int k = 0; k < nf; ++k) c[k] += S[v][k] / nf;
			auto along = [&](double t) {
				std::vector<double> z(nf);
				for (int k = 0; k < nf; ++k) z[k] = c[k] + t * (S[worst][k] - c[k]);
				return project(z);
			};
			// reflection (t = -1), expansion (-2), outside (-1/2) and inside (+1/2) contraction: one batch
			const std::vector<std::vector<double>> trial = {along(-1.0), along(-2.0), along(-0.5), along(0.5)};
			const std::vector<double> ft = evaluate(trial);
			const double fr = ft[0], fe = ft[1], foc = ft[2], fic = ft[3];
			if (fr < F[best])
			{
				if (fe < fr) S[worst] = trial[1], F[worst] = fe;
				else S[worst] = trial[0], F[worst] = fr;
			}
			else if (fr < F[second]) S[worst] = trial[0], F[worst] = fr;
			else
			{
				bool shrink = false;
				if (fr < F[worst])
				{
					if (foc <= fr) S[worst] = trial[2], F[worst] = foc;
					else shrink = true;
				}
				else
				{
					if (fic < F[worst]) S[worst] = trial[3], F[worst] = fic;
					else shrink = true;
				}
				if (shrink)
				{
					std::vector<std::vector<double>> pts;
					std::vector<int> idx;
					for (int v = 0; v <= nf; ++v)
						if (v != best)
						{
							for (int k = 0; k < nf; ++k) S[v][k] = S[best][k] + 0.5 * (S[v][k] - S[best][k]);
							pts.push_back(S[v]);
							idx.push_back(v);
						}
					const std::vector<double> fs = evaluate(pts);
					for (size_t q = 0; q < idx.size(); ++q) F[idx[q]] = fs[q];
				}
			}
		}
		const int best = static_cast<int>(std::min_element(F.begin(), F.end()) - F.begin());
		const std::vector<double> full = expand(S[best]);
		std::copy(full.begin(), full.end(), x);
		*fmin = F[best];
		if (n_eval) *n_eval = evals;
		return GPLE_OK;
	}

	// projected BFGS on phi(z) over the box; returns the number of phi evaluations
	template <class Phi>
	int projected_bfgs(Phi&& phi, const std::vector<double>& lo, const std::vector<double>& hi, std::vector<double>& z, double& fz, int maxiter,
		const gple_opt_options& opt)
	{
		const int n = static_cast<int>(z.size());
		std::vector<double> g(n), gn(n), H(static_cast<size_t>(n) * n, 0.0), d(n), zn(n), s(n), y(n);
		for (int i = 0; i < n; ++i) H[static_cast<size_t>(i) * n + i] = 1.0;
		int evals = 1;
		fz = phi(z, g);
		for (int it = 0; it < maxiter; ++it)
		{
			// projected gradient: components that push against an active bound are dropped
			double pg = 0.0;
			std::vector<char> active(n, 0);
			for (int i = 0; i < n; ++i)
			{
				active[i] = (z[i] <= lo[i] && g[i] > 0) || (z[i] >= hi[i] && g[i] < 0);
				if (!active[i]) pg = std::max(pg, std::fabs(g[i]));
			}
			if (pg == 0.0) break;
			for (int i = 0; i < n; ++i)
			{
				d[i] = 0.0;
				if (active[i]) continue;
				for (int j = 0; j < n; ++j)
					if (!active[j]) d[i] -= H[static_cast<size_t>(i) * n + j] * g[j];
			}
			double slope = 0.0;
			for (int i = 0; i < n; ++i) slope += d[i] * g[i];
			if (!(slope < 0.0)) // not a descent direction: restart from steepest descent
			{
				std::fill(H.begin(), H.end(), 0.0);
				for (int i = 0; i < n; ++i) H[static_cast<size_t>(i) * n + i] = 1.0, d[i] = active[i] ? 0.0 : -g[i];
				slope = 0.0;
				for (int i = 0; i < n; ++i) slope += d[i] * g[i];
			}
			double t = 1.0, fn = fz;
			bool ok = false;
			for (int ls = 0; ls < 40; ++ls, t *= 0.5)
			{
				for (int i = 0; i < n; ++i) zn[i] = clip(z[i] + t * d[i], lo[i], hi[i]);
				double decrease = 0.0;
				for (int i = 0; i < n; ++i) decrease += g[i] * (zn[i] - z[i]);
				fn = phi(zn, gn);
				++evals;
				if (std::isfinite(fn) && fn <= fz + 1e-4 * decrease)
				{
					ok = true;
					break;
				}
			}
			if (!ok) break;
			double sy = 0.0, xmove = 0.0, xscale = 0.0;
			for (int i = 0; i < n; ++i) s[i] = zn[i] - z[i], y[i] = gn[i] - g[i], sy += s[i] * y[i], xmove = std::max(xmove, std::fabs(s[i])), xscale = std::max(xscale, std::fabs(zn[i]));
			const double fdrop = fz - fn;
			z = zn, g = gn;
			const double fprev = fz;
			fz = fn;
			if (sy > 1e-12) // BFGS update of the inverse Hessian
			{
				std::vector<double> Hy(n, 0.0);
				double yHy = 0.0;
				for (int i = 0; i < n; ++i)
					for (int j = 0; j < n; ++j) Hy[i] += H[static_cast<size_t>(i) * n + j] * y[j];
				for (int i = 0; i < n; ++i) yHy += y[i] * Hy[i];
				for (int i = 0; i < n; ++i)
					for (int j = 0; j < n; ++j)
						H[static_cast<size_t>(i) * n + j] += (1.0 + yHy / sy) * s[i] * s[j] / sy - (Hy[i] * s[j] + s[i] * Hy[j]) / sy;
			}
			if (xmove <= opt.xtol_abs || xmove <= opt.xtol_rel * xscale) break;
			if (fdrop <= opt.ftol_abs || fdrop <= opt.ftol_rel * 0.5 * (std::fabs(fprev) + std::fabs(fz))) break;
		}
		return evals;
	}
} // namespace

namespace
{
	// DIRECT-L on the free coordinates of the box, scaled to the unit cube (as NLopt's cdirect wrapper does).
	int direct_l(const BatchEval& eval, unsigned n, const Box& box, const gple_opt_options& opt, double* x, double* fmin, int* n_eval)
	{
		const int nf = static_cast<int>(box.free.size());
		std::vector<double> x0(x, x + n);
		for (unsigned i = 0; i < n; ++i) x0[i] = clip(x0[i], box.lb[i], box.ub[i]);
		for (int k = 0; k < nf; ++k)
			if (!std::isfinite(box.lb[box.free[k]]) || !std::isfinite(box.ub[box.free[k]])) return GPLE_ERR_BAD_ARG; // DIRECT needs a finite box
		int evals = 0;
		auto to_full = [&](const std::vector<double>& u) {
			std::vector<double> full = x0;
			for (int k = 0; k < nf; ++k) full[box.free[k]] = box.lb[box.free[k]] + u[k] * (box.ub[box.free[k]] - box.lb[box.free[k]]);
			return full;
		};
		auto evaluate = [&](const std::vector<std::vector<double>>& us) {
			std::vector<std::vector<double>> pts;
			for (const auto& u : us) pts.push_back(to_full(u));
			std::vector<double> vals(pts.size());
			eval(pts, vals);
			evals += static_cast<int>(pts.size());
			for (double& v : vals)
				if (!std::isfinite(v)) v = std::numeric_limits<double>::max(); // make_normal, opt.cpp:420-431
			return vals;
		};
		if (nf == 0)
		{
			*fmin = evaluate({{}})[0];
			std::copy(x0.begin(), x0.end(), x);
			if (n_eval) *n_eval = evals;
			return GPLE_OK;
		}
		struct Rect
		{
			std::vector<double> c, w; // centre and widths in the unit cube
			double f;
		};
		std::vector<Rect> rects;
		// size class (longest side / 2, rounded to float so that equal sizes compare equal) -> its rectangles by (value, age)
		std::map<float, std::set<std::pair<double, int>>> classes;
		constexpr double EQUAL_SIDE_TOL = 5e-2;
		auto diameter = [&](const Rect& r) { return static_cast<float>(0.5 * *std::max_element(r.w.begin(), r.w.end())); };
		auto insert = [&](Rect r) {
			rects.push_back(std::move(r));
			classes[diameter(rects.back())].insert({rects.back().f, static_cast<int>(rects.size()) - 1});
		};
		std::vector<double> best_u(nf, 0.5);
		{
			Rect r{std::vector<double>(nf, 0.5), std::vector<double>(nf, 1.0), 0.0};
			r.f = evaluate({r.c})[0];
			insert(r);
		}
		double best = rects[0].f;
		const int budget = opt.max_eval > 0 ? opt.max_eval : 100000; // MaximumEvaluations, opt.cpp:339
		// width below which a side counts as converged: xtol_abs is given in the caller's units
		auto is_small = [&](const Rect& r) {
			for (int k = 0; k < nf; ++k)
			{
				const double span = box.ub[box.free[k]] - box.lb[box.free[k]];
				if (!(r.w[k] <= opt.xtol_abs / span || r.w[k] <= opt.xtol_rel)) return false;
			}
			return true;
		};
		int status = GPLE_OK;
		while (evals < budget)
		{
			// 1. potentially optimal rectangles: the best of every size class, then the lower-right convex hull from the class that holds the
			// overall minimum to the largest class (epsilon = 0: every hull point right of the minimum qualifies)
			std::vector<std::pair<float, std::pair<double, int>>> cls;
			for (const auto& [d, members] : classes)
				if (!members.empty()) cls.push_back({d, *members.begin()});
			size_t kmin = 0;
			for (size_t k = 0; k < cls.size(); ++k)
				if (cls[k].second.first <= cls[kmin].second.first) kmin = k; // ties: the larger rectangle
			std::vector<size_t> hull;
			for (size_t k = kmin; k < cls.size(); ++k)
			{
				while (hull.size() >= 2)
				{
					const auto &a = cls[hull[hull.size() - 2]], &b = cls[hull.back()], &c = cls[k];
					// b lies on or above the segment a - c: not on the lower hull
					const double cross = (static_cast<double>(b.first) - a.first) * (c.second.first - a.second.first)
						- (b.second.first - a.second.first) * (static_cast<double>(c.first) - a.first);
					if (cross <= 0.0) hull.pop_back();
					else break;
				}
				hull.push_back(k);
			}
			// 2. every longest side of every selected rectangle gets its two new centres; all of them are one batch
			struct Plan
			{
				int id;
				std::vector<int> sides;
			};
			std::vector<Plan> plans;
			std::vector<std::vector<double>> batch;
			for (size_t h : hull)
			{
				const int id = cls[h].second.second;
				const Rect& r = rects[id];
				const double wmax = *std::max_element(r.w.begin(), r.w.end());
				Plan pl{id, {}};
				for (int k = 0; k < nf; ++k)
					if (wmax - r.w[k] <= wmax * EQUAL_SIDE_TOL) pl.sides.push_back(k);
				for (int k : pl.sides)
					for (int sgn : {+1, -1})
					{
						std::vector<double> u = r.c;
						u[k] += sgn * r.w[k] / 3.0;
						batch.push_back(std::move(u));
					}
				plans.push_back(std::move(pl));
			}
			if (batch.empty()) break;
			const std::vector<double> vals = evaluate(batch);
			// 3. divide: sides in the order of the better of their two values (the best new points end up in the largest new rectangles)
			const double best_before = best;
			bool all_small = true;
			size_t q = 0;
			for (const Plan& pl : plans)
			{
				const size_t base = q;
				q += 2 * pl.sides.size();
				std::vector<size_t> order(pl.sides.size());
				std::iota(order.begin(), order.end(), size_t(0));
				std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
					return std::min(vals[base + 2 * a], vals[base + 2 * a + 1]) < std::min(vals[base + 2 * b], vals[base + 2 * b + 1]);
				});
				Rect parent = rects[pl.id];
				classes[diameter(parent)].erase({parent.f, pl.id});
				for (size_t o : order)
				{
					const int k = pl.sides[o];
					const double third = parent.w[k] / 3.0;
					parent.w[k] = third;
					for (int sgn = 0; sgn < 2; ++sgn)
					{
						Rect child{parent.c, parent.w, vals[base + 2 * o + sgn]};
						child.c[k] += (sgn == 0 ? +1.0 : -1.0) * third;
						if (child.f < best) best = child.f, best_u = child.c;
						insert(std::move(child));
					}
				}
				rects[pl.id] = parent;
				classes[diameter(parent)].insert({parent.f, pl.id});
				all_small = all_small && is_small(parent);
			}
			// 4. NLopt's stopping rules for cdirect: xtol on the rectangles just divided, ftol on an iteration that improved the minimum
			if (all_small) break;
			if (best < best_before)
			{
				const double df = std::fabs(best - best_before);
				if (df < opt.ftol_abs || df < opt.ftol_rel * 0.5 * (std::fabs(best) + std::fabs(best_before))) break;
			}
		}
		const std::vector<double> full = to_full(best_u);
		std::copy(full.begin(), full.end(), x);
		*fmin = best;
		if (n_eval) *n_eval = evals;
		return status;
	}
} // namespace

extern "C"
{
	int gple_minimize_direct_l(gple_objective_fn f, void* data, unsigned n, const double* lb, const double* ub, const gple_opt_options* options, double* x,
		double* fmin, int* n_eval)
	{
		if (!f || !x || !fmin || n == 0 || !lb || !ub) return GPLE_ERR_BAD_ARG;
		const Box box = make_box(n, lb, ub);
		const BatchEval eval = [&](const std::vector<std::vector<double>>& pts, std::vector<double>& vals) {
			for (size_t i = 0; i < pts.size(); ++i) vals[i] = f(n, pts[i].data(), nullptr, data);
		};
		return direct_l(eval, n, box, defaults(options), x, fmin, n_eval);
	}

	int gple_objective_minimize_direct_l(gple_objective* const* objectives, size_t n_objectives, size_t n, const double* lb, const double* ub,
		const unsigned char* is_log, const gple_opt_options* options, double* x, double* fmin, int* n_eval)
	{
		if (!objectives || n_objectives == 0 || !x || !fmin || !lb || !ub || (n != 4 && n != 8)) return GPLE_ERR_BAD_ARG;
		for (size_t i = 0; i < n_objectives; ++i)
			if (!objectives[i]) return GPLE_ERR_BAD_ARG;
		const Box box = make_box(static_cast<unsigned>(n), lb, ub);
		int status = GPLE_OK;
		// loose_function_global_wrapper (opt.cpp:489-497): coordinates flagged in is_log are logarithms of the parameters they stand for
		const BatchEval eval = [&](const std::vector<std::vector<double>>& pts, std::vector<double>& vals) {
			std::vector<std::future<int>> jobs;
			for (size_t w = 0; w < std::min(n_objectives, pts.size()); ++w)
				jobs.push_back(std::async(std::launch::async, [&, w] {
					int st = GPLE_OK;
					std::vector<double> theta(n);
					for (size_t q = w; q < pts.size(); q += n_objectives)
					{
						for (size_t i = 0; i < n; ++i) theta[i] = is_log && is_log[i] ? std::exp(pts[q][i]) : pts[q][i];
						const int s = gple_objective_eval(objectives[w], theta.data(), n, &vals[q], nullptr);
						if (s != GPLE_OK) st = s, vals[q] = std::numeric_limits<double>::max();
					}
					return st;
				}));
			for (auto& j : jobs)
			{
				const int s = j.get();
				if (s != GPLE_OK) status = s;
			}
		};
		const int rc = direct_l(eval, static_cast<unsigned>(n), box, defaults(options), x, fmin, n_eval);
		return rc != GPLE_OK ? rc : status;
	}

	int gple_minimize_neldermead(gple_objective_fn f, void* data, unsigned n, const double* lb, const double* ub, const gple_opt_options* options, double* x,
		double* fmin, int* n_eval)
	{
		if (!f || !x || !fmin || n == 0) return GPLE_ERR_BAD_ARG;
		const Box box = make_box(n, lb, ub);
		const BatchEval eval = [&](const std::vector<std::vector<double>>& pts, std::vector<double>& vals) {
			for (size_t i = 0; i < pts.size(); ++i) vals[i] = f(n, pts[i].data(), nullptr, data);
		};
		return nelder_mead(eval, n, box, defaults(options), x, fmin, n_eval);
	}

	int gple_objective_minimize_neldermead(gple_objective* const* objectives, size_t n_objectives, size_t n, const double* lb, const double* ub,
		const gple_opt_options* options, double* x, double* fmin, int* n_eval)
	{
		if (!objectives || n_objectives == 0 || !x || !fmin || (n != 4 && n != 8)) return GPLE_ERR_BAD_ARG;
		for (size_t i = 0; i < n_objectives; ++i)
			if (!objectives[i]) return GPLE_ERR_BAD_ARG;
		const Box box = make_box(static_cast<unsigned>(n), lb, ub);
		int status = GPLE_OK;
		// point q of a batch goes to objective q mod n_objectives; every objective (= context = HIP stream) has its own host thread
		const BatchEval eval = [&](const std::vector<std::vector<double>>& pts, std::vector<double>& vals) {
			std::vector<std::future<int>> jobs;
			for (size_t w = 0; w < std::min(n_objectives, pts.size()); ++w)
				jobs.push_back(std::async(std::launch::async, [&, w] {
					int st = GPLE_OK;
					for (size_t q = w; q < pts.size(); q += n_objectives)
					{
						const int s = gple_objective_eval(objectives[w], pts[q].data(), n, &vals[q], nullptr);
						if (s != GPLE_OK) st = s, vals[q] = std::numeric_limits<double>::max();
					}
					return st;
				}));
			for (auto& j : jobs)
			{
				const int s = j.get();
				if (s != GPLE_OK) status = s;
			}
		};
		const int rc = nelder_mead(eval, static_cast<unsigned>(n), box, defaults(options), x, fmin, n_eval);
		return rc != GPLE_OK ? rc : status;
	}

	int gple_minimize_auglag_eq(gple_objective_fn f, void* fdata, gple_constraint_fn h, void* hdata, unsigned m, unsigned n, const double* lb,
		const double* ub, const gple_opt_options* options, double* x, double* fmin, int* n_eval)
	{
		if (!f || !x || !fmin || n == 0 || (m && !h)) return GPLE_ERR_BAD_ARG;
		const gple_opt_options opt = defaults(options);
		const Box box = make_box(n, lb, ub);
		const int nf = static_cast<int>(box.free.size());
		std::vector<double> full(x, x + n);
		for (unsigned i = 0; i < n; ++i) full[i] = clip(full[i], box.lb[i], box.ub[i]);
		int evals = 0;
		std::vector<double> gfull(n), hval(m), hgrad(static_cast<size_t>(m) * n);
		if (nf == 0)
		{
			*fmin = f(n, full.data(), nullptr, fdata);
			std::copy(full.begin(), full.end(), x);
			if (n_eval) *n_eval = 1;
			return GPLE_OK;
		}
		std::vector<double> lo(nf), hi(nf), z(nf), lambda(m, 0.0);
		for (int k = 0; k < nf; ++k) lo[k] = box.lb[box.free[k]], hi[k] = box.ub[box.free[k]], z[k] = full[box.free[k]];
		double rho = 1.0, fval = 0.0, infeas_prev = std::numeric_limits<double>::infinity();
		auto set_full = [&](const std::vector<double>& zz) {
			for (int k = 0; k < nf; ++k) full[box.free[k]] = zz[k];
		};
		// start penalty as NLopt does: rho = max(1e-6, min(10, 2 |f| / |h|^2))
		{
			set_full(z);
			const double f0 = f(n, full.data(), nullptr, fdata);
			++evals;
			double h2 = 0.0;
			if (m)
			{
				h(m, hval.data(), n, full.data(), nullptr, hdata);
				for (double v : hval) h2 += v * v;
			}
			if (h2 > 0 && std::isfinite(f0)) rho = std::max(1e-6, std::min(10.0, 2.0 * std::fabs(f0) / h2));
		}
		const int max_outer = 30, budget = opt.max_eval > 0 ? opt.max_eval : 2000;
		for (int outer = 0; outer < max_outer && evals < budget; ++outer)
		{
			auto phi = [&](const std::vector<double>& zz, std::vector<double>& g) {
				set_full(zz);
				double val = f(n, full.data(), gfull.data(), fdata);
				if (m) h(m, hval.data(), n, full.data(), hgrad.data(), hdata);
				for (unsigned i = 0; i < m; ++i)
				{
					val += lambda[i] * hval[i] + 0.5 * rho * hval[i] * hval[i];
					const double w = lambda[i] + rho * hval[i];
					for (unsigned c = 0; c < n; ++c) gfull[c] += w * hgrad[static_cast<size_t>(i) * n + c];
				}
				for (int k = 0; k < nf; ++k) g[k] = gfull[box.free[k]];
				if (!std::isfinite(val)) val = std::numeric_limits<double>::max();
				return val;
			};
			double fz;
			const std::vector<double> z_before = z;
			evals += projected_bfgs(phi, lo, hi, z, fz, 100, opt);
			set_full(z);
			fval = f(n, full.data(), nullptr, fdata);
			++evals;
			double infeas = 0.0;
			if (m)
			{
				h(m, hval.data(), n, full.data(), nullptr, hdata);
				for (unsigned i = 0; i < m; ++i) infeas = std::max(infeas, std::fabs(hval[i])), lambda[i] += rho * hval[i];
			}
			if (infeas > 0.25 * infeas_prev) rho *= 10.0;
			infeas_prev = std::min(infeas_prev, infeas);
			double move = 0.0, scale = 0.0;
			for (int k = 0; k < nf; ++k) move = std::max(move, std::fabs(z[k] - z_before[k])), scale = std::max(scale, std::fabs(z[k]));
			if (outer > 0 && (move <= opt.xtol_abs || move <= opt.xtol_rel * scale) && (m == 0 || infeas <= 1e-8)) break;
			if (m == 0) break;
		}
		std::copy(full.begin(), full.end(), x);
		*fmin = fval;
		if (n_eval) *n_eval = evals;
		return GPLE_OK;
	}
}
