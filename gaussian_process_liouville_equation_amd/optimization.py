"""Host driver of the hyperparameter search — SURVEY.md §8(f) row N2, the caller directly above the hot path.

Mirrors `Optimization` (opt.h:16-131, opt.cpp:272-1392): the same bounds, the same element-wise -> diagonal -> full
sequence, the same three tiers (previous parameters, initial parameters, global) and the same rule for choosing between
their results.  Every objective and constraint evaluation is one call into the HIP library through `kernels.py`; nothing
is computed on the host but the search itself.

The reference drives NLopt 2.7 (LN_NELDERMEAD, AUGLAG_EQ over LD_SLSQP, GN_DIRECT_L; opt.h:51-55), which this image does
not have.  The search algorithms come from SciPy (Nelder-Mead, SLSQP, DIRECT with locally_biased=True) or, searches="native", from the
library itself (csrc/gple_opt.hip: Nelder-Mead, augmented Lagrangian, DIRECT-L behind NLopt's callback ABIs) and the
equality-constrained stage is the augmented-Lagrangian scheme NLopt documents for AUGLAG_EQ (Birgin & Martinez 2008),
written out below.  Iterates therefore differ from NLopt's; what is kept is the reference's control flow, tolerances,
bounds and acceptance logic.
"""
import enum
import math
import warnings

import numpy as np
from scipy import optimize as _sciopt

from . import kernels as K

InitialMagnitude = 1.0  # opt.cpp:25
InitialNoise = 1e-2  # opt.cpp:27
AverageTolerance = 0.05  # opt.h:13
RelativeTolerance = 1e-5  # opt.cpp:344
AbsoluteTolerance = 1e-15  # opt.cpp:345
InitialStepSize = 0.5  # opt.cpp:346
MaximumEvaluations = 100000  # opt.cpp:340 (global optimiser only)
GaussKerMinCharLength = 1.0 / 100.0  # opt.cpp:397


class OptimizationType(enum.IntEnum):  # opt.h:20-30
    Default = 0
    LocalPrevious = 1
    LocalInitial = 2
    Global = 3


def calculate_kernel_bounds(CharLengthLowerBound, CharLengthUpperBound):
    """opt.cpp:33-61: magnitude and noise are pinned, only the characteristic lengths move."""
    lb = [InitialMagnitude, *CharLengthLowerBound, InitialNoise]
    ub = [InitialMagnitude, *CharLengthUpperBound, InitialNoise]
    return [float(v) for v in lb], [float(v) for v in ub]


def calculate_complex_kernel_bounds(CharLengthLowerBound, CharLengthUpperBound):
    """opt.cpp:67-104: the two sub-kernel weights (real, imaginary) may move a decade either way."""
    lb, ub = [InitialMagnitude], [InitialMagnitude]
    for _ in range(2):  # ComplexKernelBase::NumKernels, complex_kernel.h:18
        lb += [InitialMagnitude / 10.0, *CharLengthLowerBound]
        ub += [InitialMagnitude * 10.0, *CharLengthUpperBound]
    lb.append(InitialNoise)
    ub.append(InitialNoise)
    return [float(v) for v in lb], [float(v) for v in ub]


# ---- Monte-Carlo sample statistics the driver needs (predict.cpp:65-190), O(N) host glue -------------------------------
def calculate_standard_deviation_one_surface(points):  # predict.cpp:109-127
    r = np.asarray(points[0], dtype=float)
    return np.sqrt((r * r).sum(axis=0) / len(r) - (r.sum(axis=0) / len(r)) ** 2)


def calculate_1st_order_average_one_surface(points):  # predict.cpp:88-107 (sample form)
    r, rho = np.asarray(points[0], dtype=float), np.asarray(points[1]).real
    return (r * rho[:, None]).sum(axis=0) / rho.sum()


def calculate_total_energy_average_one_surface(points, mass, iPES, potential=None):
    """predict.cpp:148-180: rho-weighted mean of p^2/2m + V_iPES(x).  `potential(x (N,), iPES) -> (N,)` stands for the
    adiabatic surface of pes.cpp, which is outside this path; None means a flat surface."""
    r, rho = np.asarray(points[0], dtype=float), np.asarray(points[1]).real
    e = r[:, 1] ** 2 / (2.0 * float(mass))
    if potential is not None:
        e = e + np.asarray(potential(r[:, 0], iPES), dtype=float)
    return float((e * rho).sum() / rho.sum())


# ---- the searches standing in for NLopt ----------------------------------------------------------------------------------
class _Counted:
    """Objective wrapper over the free (lb < ub) coordinates; counts evaluations like nlopt::opt::get_numevals."""

    def __init__(self, fun, x0, lb, ub):
        self.full = np.array(x0, dtype=float)
        lb, ub = np.asarray(lb, dtype=float), np.asarray(ub, dtype=float)
        self.free = np.flatnonzero(ub > lb)
        self.lb, self.ub = lb[self.free], ub[self.free]
        self.fun, self.numevals = fun, 0

    def expand(self, z):
        x = self.full.copy()
        x[self.free] = z
        return x

    def value(self, z):
        self.numevals += 1
        return self.fun(list(self.expand(z)), [])

    def value_and_grad(self, z):
        self.numevals += 1
        g = [0.0] * len(self.full)
        v = self.fun(list(self.expand(z)), g)
        return v, np.asarray(g, dtype=float)[self.free]


def _nelder_mead(fun, x0, lb, ub, maxeval=None):
    """LN_NELDERMEAD stand-in: SciPy's bounded simplex with the reference's tolerances and initial step (opt.cpp:342-355)."""
    c = _Counted(fun, x0, lb, ub)
    if len(c.free) == 0:
        return list(c.full), c.value(np.zeros(0)), c.numevals
    z0 = np.clip(c.full[c.free], c.lb, c.ub)
    n = len(z0)
    simplex = np.tile(z0, (n + 1, 1))
    for i in range(n):  # nlopt steps +h, or -h when +h would leave the box
        h = InitialStepSize if z0[i] + InitialStepSize <= c.ub[i] else -InitialStepSize
        simplex[i + 1, i] = min(max(z0[i] + h, c.lb[i]), c.ub[i])
        if simplex[i + 1, i] == z0[i]:
            simplex[i + 1, i] = 0.5 * (c.lb[i] + c.ub[i])
    res = _sciopt.minimize(c.value, z0, method="Nelder-Mead", bounds=list(zip(c.lb, c.ub)),
                           options={"initial_simplex": simplex, "xatol": RelativeTolerance * max(1.0, float(np.abs(z0).max())),
                                    "fatol": AbsoluteTolerance, "maxfev": maxeval or 400 * n, "adaptive": False})
    return list(c.expand(res.x)), float(res.fun), c.numevals


def _direct(fun, x0, lb, ub, maxeval=MaximumEvaluations):
    """GN_DIRECT_L stand-in (Gablonsky & Kelley's locally biased DIRECT)."""
    c = _Counted(fun, x0, lb, ub)
    if len(c.free) == 0:
        return list(c.full), c.value(np.zeros(0)), c.numevals
    res = _sciopt.direct(c.value, list(zip(c.lb, c.ub)), locally_biased=True, maxfun=maxeval, f_min_rtol=RelativeTolerance,
                         vol_tol=1e-12)
    return list(c.expand(res.x)), float(res.fun), c.numevals


def _auglag_eq(fun, constraint, m, x0, lb, ub, max_outer=20, inner_maxiter=100):
    """AUGLAG_EQ over LD_SLSQP stand-in.  Minimises f + sum(lambda_i h_i) + rho/2 sum(h_i^2) inside the box with SLSQP,
    then lambda += rho h, and rho *= 10 whenever the infeasibility failed to halve — the published scheme NLopt follows.
    `constraint(x, want_grad) -> (h (m,), grad (m*n,) row-major or None)`."""
    c = _Counted(fun, x0, lb, ub)
    n_full = len(c.full)
    if len(c.free) == 0:
        return list(c.full), c.value(np.zeros(0)), c.numevals
    z = np.clip(c.full[c.free], c.lb, c.ub)
    lam = np.zeros(m)
    f0 = c.value(z)
    h0 = np.asarray(constraint(list(c.expand(z)), False)[0], dtype=float)
    pen = float(h0 @ h0)
    rho = max(1e-6, min(10.0, 2.0 * abs(f0) / pen)) if pen > 0 else 1.0
    prev_infeas = math.inf
    best = (z.copy(), f0)
    for _ in range(max_outer):
        def lagrangian(zz):
            f, gf = c.value_and_grad(zz)
            h, gh = constraint(list(c.expand(zz)), True)
            h = np.asarray(h, dtype=float)
            gh = np.asarray(gh, dtype=float).reshape(m, n_full)[:, c.free]
            return f + lam @ h + 0.5 * rho * (h @ h), gf + (lam + rho * h) @ gh

        with warnings.catch_warnings():  # SLSQP reports every trial point it clips back into the box
            warnings.simplefilter("ignore", RuntimeWarning)
            res = _sciopt.minimize(lagrangian, z, jac=True, method="SLSQP", bounds=list(zip(c.lb, c.ub)),
                                   options={"ftol": RelativeTolerance, "maxiter": inner_maxiter})
        step = float(np.abs(res.x - z).max())
        z = np.clip(res.x, c.lb, c.ub)
        h = np.asarray(constraint(list(c.expand(z)), False)[0], dtype=float)
        infeas = float(np.abs(h).max())
        best = (z.copy(), c.value(z))
        lam = lam + rho * h
        if infeas > 0.5 * prev_infeas:
            rho *= 10.0
        prev_infeas = infeas
        if step <= RelativeTolerance * max(1.0, float(np.abs(z).max())) and infeas <= RelativeTolerance:
            break
    return list(c.expand(best[0])), float(best[1]), c.numevals


def _native_nelder_mead(api, fun, resident, x0, lb, ub, maxeval=None):
    """The library's own Nelder-Mead (gple_objective_minimize_neldermead on the resident objective when there is one: the
    whole search then runs inside the library; otherwise gple_minimize_neldermead calling back into `fun`)."""
    from . import _capi as c
    lib = getattr(api, "lib", None)
    if lib is None or not hasattr(lib, "gple_minimize_neldermead"):  # the oracle binding of the CPU tests has no searches
        return _nelder_mead(fun, x0, lb, ub, maxeval)
    if resident is not None and hasattr(resident, "handle"):
        return c.objective_minimize_neldermead(lib, [resident], x0, lb, ub, maxeval or 0)
    return c.minimize_neldermead(lib, lambda x: fun(list(x), []), x0, lb, ub, maxeval or 0)


def _native_direct(api, fun, resident, x0, lb, ub, maxeval=MaximumEvaluations):
    """The library's own DIRECT-L (GN_DIRECT_L stand-in, csrc/gple_opt.hip) in the log-parameter box of the global tier: on the resident
    objective (with a pool: one handle per context, every iteration's new rectangle centres evaluated concurrently) or calling back into `fun`."""
    from . import _capi as c
    lib = getattr(api, "lib", None)
    if lib is None or not hasattr(lib, "gple_minimize_direct_l"):  # the oracle binding of the CPU tests has no searches
        return _direct(fun, x0, lb, ub, maxeval)
    if resident is not None and hasattr(resident, "handle"):
        flags = [i in K._log_indices(len(x0)) for i in range(len(x0))]
        return c.objective_minimize_direct_l(lib, [resident], x0, lb, ub, flags, maxeval)
    return c.minimize_direct_l(lib, lambda x: fun(list(x), []), x0, lb, ub, maxeval)


def _native_auglag_eq(fun, constraint, m, x0, lb, ub):
    """The library's augmented-Lagrangian search (gple_minimize_auglag_eq) over Python objective / constraint callbacks."""
    from . import _capi as c
    from . import load_library

    def f(x, want_grad):
        g = [0.0] * len(x) if want_grad else []
        return fun(list(x), g), g

    return c.minimize_auglag_eq(load_library(), f, constraint, m, x0, lb, ub)


class Optimization:
    """opt.h:16-131.  `optimize(density, extra_points)` refits every element's hyperparameters for one time step and
    returns (error, steps per stage, OptimizationType); `get_parameters()` then feeds `TrainingKernels` / the predictors.

    The constructor takes the four members of `InitialParameters` the reference reads (input.h: sigma_r0, rmin, rmax, mass)
    instead of that class, whose input-file parsing is outside this path."""

    Result = tuple

    def __init__(self, sigma_r0, rmin, rmax, mass, InitialTotalEnergy, InitialPurity, potential=None, api=None, num_pes=None,
                 local_maxeval=None, searches="scipy"):
        # searches: "scipy" = SciPy's Nelder-Mead / SLSQP behind the reference's control flow; "native" = the library's own
        # Nelder-Mead and augmented Lagrangian (csrc/gple_opt.hip, include/gple.h) — no SciPy in the local stages
        assert searches in ("scipy", "native")
        self.searches = searches
        self.TotalEnergy, self.Purity = float(InitialTotalEnergy), float(InitialPurity)
        self.mass, self.potential = float(np.ravel(mass)[0]), potential
        self.api = api or K.default_api()
        self.num_pes = num_pes or K.NumPES
        self.local_maxeval = local_maxeval
        self.elements = K.element_order(self.num_pes)
        sigma_r0 = [float(v) for v in sigma_r0]
        self.InitialKernelParameter = [InitialMagnitude, *sigma_r0, InitialNoise]  # opt.cpp:286-303
        self.InitialComplexKernelParameter = [InitialMagnitude, *([InitialMagnitude, *sigma_r0] * 2), InitialNoise]  # :304-330
        self.ParameterVectors = self._initial_vectors()
        size = np.asarray(rmax, dtype=float) - np.asarray(rmin, dtype=float)  # opt.cpp:394-413
        ones = np.ones(K.PhaseDim) * GaussKerMinCharLength
        self._bounds = {e: (calculate_kernel_bounds(ones, size) if e[0] == e[1] else calculate_complex_kernel_bounds(ones, size))
                        for e in self.elements}

    # ---- accessors (opt.h:79-92) ----
    def get_parameters(self):
        return self.ParameterVectors

    def get_lower_bounds(self):
        return {e: list(b[0]) for e, b in self._bounds.items()}

    def get_upper_bounds(self):
        return {e: list(b[1]) for e, b in self._bounds.items()}

    def _initial_vectors(self):
        return {e: list(self.InitialKernelParameter if e[0] == e[1] else self.InitialComplexKernelParameter) for e in self.elements}

    def _stack(self, which, diagonal_only):
        return [v for e in self.elements if not diagonal_only or e[0] == e[1] for v in self._bounds[e][which]]

    # ---- stages ----
    def _optimize_elementwise(self, TrainingSets, ExtraTrainingSets, params, is_global):
        """opt.cpp:518-588.  The element searches are independent; with an ApiPool they run concurrently, one context
        (HIP stream) and one host thread each, and return exactly what the sequential loop returns."""

        def search(one_api, e):
            if len(TrainingSets[e][0]) == 0:
                return params[e], 0.0, 0
            # the two sets go to the device once per optimize() call; the hundreds of evaluations move only the parameters
            etp = (TrainingSets[e], ExtraTrainingSets[e], K.resident_objective(self._objectives, one_api, e, TrainingSets, ExtraTrainingSets))
            lb, ub = self._bounds[e]
            try:
                if is_global:
                    obj = lambda x, g: K.loose_function_global_wrapper(x, g, etp, api=one_api)
                    glb, gub = K.local_parameter_to_global(lb), K.local_parameter_to_global(ub)
                    if self.searches == "native":
                        return _native_direct(one_api, obj, etp[2], params[e], glb, gub)
                    return _direct(obj, params[e], glb, gub)
                obj = lambda x, g: K.loose_function(x, g, etp, api=one_api)
                if self.searches == "native":
                    return _native_nelder_mead(one_api, obj, etp[2], params[e], lb, ub, self.local_maxeval)
                return _nelder_mead(obj, params[e], lb, ub, self.local_maxeval)
            except (ArithmeticError, ValueError):  # opt.cpp:555-565: a failed search keeps what it had
                return params[e], 0.0, 0

        if isinstance(self.api, K.ApiPool):
            results = self.api.map(search, self.elements)
        else:
            results = [search(self.api, e) for e in self.elements]
        total_error, num_steps = 0.0, []
        for e, (x, err, n) in zip(self.elements, results):
            params[e] = x
            total_error += err
            num_steps.append(n)
        return total_error, num_steps

    def _optimize_diagonal(self, TrainingSets, ExtraTrainingSets, Energies, Purity, params):
        """opt.cpp:970-1043.  A NaN purity drops the purity constraint (opt.cpp:1143-1152)."""
        n = self.num_pes
        x0 = [v for i in range(n) for v in params[(i, i)]]
        m = 3 if Purity > 0 else 2
        obj = lambda x, g: K.diagonal_loose(x, g, (TrainingSets, ExtraTrainingSets, self._objectives), api=self.api, num_pes=n)
        con = lambda x, want: K.diagonal_constraints(m, x, want, (TrainingSets, Energies, self.TotalEnergy, Purity), api=self.api, num_pes=n)
        x, err, steps = (_native_auglag_eq if self.searches == "native" else _auglag_eq)(obj, con, m, x0, self._stack(0, True), self._stack(1, True))
        for i in range(n):
            params[(i, i)] = list(x[i * K.REAL_NPARAM:(i + 1) * K.REAL_NPARAM])
        return err, [steps]

    def _optimize_full(self, TrainingSets, ExtraTrainingSets, Energies, params):
        """opt.cpp:1045-1118"""
        n = self.num_pes
        x0 = K.construct_combined_parameters(params, n)
        obj = lambda x, g: K.full_loose(x, g, (TrainingSets, ExtraTrainingSets, self._objectives), api=self.api, num_pes=n)
        con = lambda x, want: K.full_constraints(x, want, (TrainingSets, Energies, self.TotalEnergy, self.Purity), api=self.api, num_pes=n)
        x, err, steps = (_native_auglag_eq if self.searches == "native" else _auglag_eq)(obj, con, 3, x0, self._stack(0, False), self._stack(1, False))
        params.update(K.construct_all_parameters(x, n))
        return err, [steps]

    def _one_api(self):
        return self.api.api_for(0) if isinstance(self.api, K.ApiPool) else self.api

    def _kernels(self, params, TrainingSets):
        return K.TrainingKernels(params, TrainingSets, False, True, False, api=self.api, num_pes=self.num_pes)

    def optimize(self, density, extra_points):
        """opt.cpp:1019-1392"""
        TrainingSets = K.construct_training_sets(density, self.num_pes)
        ExtraTrainingSets = K.construct_training_sets(extra_points, self.num_pes)
        for o in getattr(self, "_objectives", {}).values():  # the previous step's resident sets
            if hasattr(o, "release"):
                o.release()
        self._objectives = {}
        n = self.num_pes
        Energies = [0.0 if len(TrainingSets[(i, i)][0]) == 0 else
                    calculate_total_energy_average_one_surface(TrainingSets[(i, i)], self.mass, i, self.potential) for i in range(n)]
        # per-step bounds from the spread of the current samples (opt.cpp:1027-1052); empty elements keep theirs
        for e in self.elements:
            if len(TrainingSets[e][0]) != 0:
                sd = calculate_standard_deviation_one_surface(TrainingSets[e])
                lo, hi = sd / math.sqrt(len(TrainingSets[e][0])), 2.0 * sd
                self._bounds[e] = calculate_kernel_bounds(lo, hi) if e[0] == e[1] else calculate_complex_kernel_bounds(lo, hi)

        def move_into_bounds(params):  # opt.cpp:1054-1067
            for e in self.elements:
                lb, ub = self._bounds[e]
                params[e] = [min(max(p, l), u) for p, l, u in zip(params[e], lb, ub)]

        offdiagonal = any(len(TrainingSets[e][0]) != 0 for e in self.elements if e[0] != e[1])

        def do_optimize(params, opt_type):  # opt.cpp:1098-1199
            for e in self.elements:
                params[e][0] = InitialMagnitude
            move_into_bounds(params)
            err, steps = self._optimize_elementwise(TrainingSets, ExtraTrainingSets, params, False)
            if offdiagonal:
                _, dsteps = self._optimize_diagonal(TrainingSets, ExtraTrainingSets, Energies, math.nan, params)
                err, fsteps = self._optimize_full(TrainingSets, ExtraTrainingSets, Energies, params)
                steps = steps + dsteps + fsteps
            else:
                err, dsteps = self._optimize_diagonal(TrainingSets, ExtraTrainingSets, Energies, self.Purity, params)
                steps = steps + dsteps + [0]
            for e in self.elements:  # afterwards, the magnitude (opt.cpp:1179-1196)
                if len(TrainingSets[e][0]) != 0:
                    cls = K.TrainingKernel if e[0] == e[1] else K.TrainingComplexKernel
                    params[e][0] = cls(params[e], TrainingSets[e], False, False, False, api=self._one_api()).get_magnitude()
            return [err, steps, opt_type]

        def beyond_tolerance_error(calc, ref):  # opt.cpp:1212-1223
            err = abs(calc / ref - 1.0)
            return 0.0 if err < AverageTolerance else err

        def check_averages(params):  # opt.cpp:1201-1274
            ks = self._kernels(params, TrainingSets)
            return np.array([beyond_tolerance_error(ks.calculate_population(), 1.0),
                             beyond_tolerance_error(ks.calculate_total_energy_average(Energies), self.TotalEnergy),
                             beyond_tolerance_error(ks.calculate_purity(), self.Purity)])

        def compare_and_overwrite(result, check_result, result_new, check_new, params_new):  # opt.cpp:1276-1325
            better = int(np.sum((check_new < check_result) & (check_result > 2.0 * AverageTolerance)))
            worse = int(np.sum((check_new > check_result) & (check_new > 2.0 * AverageTolerance)))
            if (better > worse or (better == worse and check_new.sum() < check_result.sum())
                    or (better == worse and result_new[0] < result[0])):
                self.ParameterVectors = params_new
                result[0] = result_new[0]
                for i in range(len(result_new[1])):
                    result[1][i] += result_new[1][i]
                result[2] = result_new[2]
                check_result[:] = check_new

        # 1. local search from the previous step's parameters
        result = do_optimize(self.ParameterVectors, OptimizationType.LocalPrevious)
        check_result = check_averages(self.ParameterVectors)
        if (check_result == 0.0).all():
            return tuple(result)
        # 2. local search from the initial parameters
        param_vec_initial = self._initial_vectors()
        result_initial = do_optimize(param_vec_initial, OptimizationType.LocalInitial)
        compare_and_overwrite(result, check_result, result_initial, check_averages(param_vec_initial), param_vec_initial)
        if (check_result == 0.0).all():
            return tuple(result)
        # 3. global search in log-parameters, then the local sequence from its answer (opt.cpp:1347-1383)
        param_vec_global = self._initial_vectors()
        move_into_bounds(param_vec_global)
        for e in self.elements:
            param_vec_global[e] = K.local_parameter_to_global(param_vec_global[e])
        _, steps_global = self._optimize_elementwise(TrainingSets, ExtraTrainingSets, param_vec_global, True)
        for e in self.elements:
            param_vec_global[e] = K.global_parameter_to_local(param_vec_global[e])
        result_global = do_optimize(param_vec_global, OptimizationType.Global)
        for i in range(len(steps_global)):
            result_global[1][i] += steps_global[i]
        compare_and_overwrite(result, check_result, result_global, check_averages(param_vec_global), param_vec_global)
        return tuple(result)
