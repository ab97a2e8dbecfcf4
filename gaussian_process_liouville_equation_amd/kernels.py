"""Host-side mirror of the reference's kernel interface (same names, argument meaning and error behaviour).

reference (paths relative to /root/reference/gaussian_process_liouville_equation/):
    KernelBase, TrainingKernel, PredictiveKernel                kernel.h:29-403
    ComplexKernelBase, TrainingComplexKernel, PredictiveComplex complex_kernel.h:14-391
    TrainingKernels, construct_training_sets                   predict.h:14-143, predict.cpp:246-559
    loose_function, diagonal_loose, full_loose, *_constraints  opt.cpp:109-232, 420-497, 594-719, 844-929
Every object is immutable after construction, like the reference (all members const, kernel.cpp:244-479): constructing
one IS calling the hot path; the compute happens on the MI355X behind include/gple.h.  Getters that the reference guards
with assert(...has_value()) raise AssertionError here.

`api` is the bound C-ABI (gaussian_process_liouville_equation_amd.open_api()); the tests inject the CPU oracle's binding
through the same parameter to check this host logic without a GPU — the package itself never imports the oracle.
"""
import math

import numpy as np

from . import _capi as c

NumPES = 2  # stdafx.h:111 (compile-time there; a run-time default here: the GP code is generic in it)
PhaseDim = 2  # stdafx.h:121
REAL_NPARAM = 4  # KernelBase::NumTotalParameters, kernel.h:33
COMPLEX_NPARAM = 8  # ComplexKernelBase::NumTotalParameters, complex_kernel.h:22
ConnectingPoint = 2.0  # kernel.h:16

_default_api = None


def default_api():
    """Lazily opened context on device 0 (fails loudly when the HIP library is missing: no CPU fallback)."""
    global _default_api
    if _default_api is None:
        from . import open_api
        _default_api = open_api(0)
    return _default_api


class ApiPool:
    """Several contexts on one device = several HIP streams.  The density-matrix elements are independent GPs (predict.h:89,
    opt.cpp:518-588) and a single fit at N <= 2048 is latency-bound on a few CUs, so fitting them concurrently — element e on
    context e mod n, one host thread each (the ctypes calls drop the GIL) — overlaps them on the GPU.  Results do not depend
    on the pool: every element still runs the same kernels on the same data."""

    def __init__(self, n=3, device=0, apis=None):
        from concurrent.futures import ThreadPoolExecutor
        from . import open_api
        self.apis = list(apis) if apis is not None else [open_api(device) for _ in range(n)]
        self._owned = apis is None
        self._pool = ThreadPoolExecutor(max_workers=len(self.apis))

    def api_for(self, index):
        return self.apis[index % len(self.apis)]

    def map(self, fn, items):
        """fn(api, item) for every item, item i on context i mod n; results in order."""
        futures = [self._pool.submit(fn, self.api_for(i), it) for i, it in enumerate(items)]
        return [f.result() for f in futures]

    def close(self):
        self._pool.shutdown(wait=True)
        if self._owned:
            for a in self.apis:
                a.close()
        self.apis = []


def _flags(err, avg, der):
    return (c.CALC_ERROR if err else 0) | (c.CALC_AVERAGE if avg else 0) | (c.CALC_DERIVATIVE if der else 0)


def delta_kernel(LeftFeature, RightFeature):
    """kernel.cpp:8-31. `is`-identity of the two arrays plays the role of LeftFeature.data() == RightFeature.data()."""
    L, R = np.asarray(LeftFeature), np.asarray(RightFeature)
    if LeftFeature is RightFeature:
        return np.eye(len(L), len(R))
    return (L[:, None, :] == R[None, :, :]).all(axis=2).astype(float)


def cutoff_factor(Prediction, Variance, api=None):
    """kernel.h:301-332 (real or complex predictions)."""
    return (api or default_api()).cutoff_factor(Prediction, Variance)


class KernelBase:
    """kernel.h:29-106: K = sf^2 (G + sn^2 delta) and, on request, its 4 parameter derivatives."""
    NumTotalParameters = REAL_NPARAM

    def __init__(self, Parameter, left_feature, right_feature, IsToCalculateDerivative, api=None):
        api = api or default_api()
        magnitude, char_length, noise = Parameter
        self.KernelParams = (float(magnitude), np.asarray(char_length, dtype=float), float(noise))
        self.LeftFeature, self.RightFeature = np.array(left_feature, dtype=float), np.array(right_feature, dtype=float)
        theta = [magnitude, char_length[0], char_length[1], noise]
        res = api.real_gram(theta, self.LeftFeature, self.RightFeature, left_feature is right_feature, IsToCalculateDerivative)
        self.KernelMatrix, self.Derivatives = res if IsToCalculateDerivative else (res, None)

    def get_formatted_parameters(self):
        return self.KernelParams

    def get_left_feature(self):
        return self.LeftFeature

    def get_right_feature(self):
        return self.RightFeature

    def get_kernel(self):
        return self.KernelMatrix

    def get_derivative(self):
        assert self.Derivatives is not None
        return self.Derivatives


class _TrainingBase:
    def __init__(self, fit, Parameter, TrainingSet, flags, api):
        self._fit, self._api, self._flags = fit, api, flags
        self.Params = list(map(float, Parameter))
        self.LeftFeature = np.array(TrainingSet[0], dtype=float)

    @property
    def _s(self):
        # the scalar members are fetched when a getter first asks (one stream synchronisation); constructing the kernel
        # and predicting from it only enqueue device work
        return self._fit.scalars

    def _need(self, flag):
        assert self._flags & flag, "this quantity was not requested at construction"

    def get_parameters(self):
        return self.Params

    def get_left_feature(self):
        return self.LeftFeature

    get_right_feature = get_left_feature

    def get_rescale_factor(self):
        return self._s["rescale_factor"]

    def get_magnitude(self):
        return self._s["magnitude"]

    def get_error(self):
        self._need(c.CALC_ERROR)
        return self._s["error"]

    def get_purity(self):
        self._need(c.CALC_AVERAGE)
        return self._s["purity"]

    def get_error_derivative(self):
        self._need(c.CALC_ERROR)
        self._need(c.CALC_DERIVATIVE)
        return self._s["error_derivative"]

    def get_purity_derivative(self):
        self._need(c.CALC_AVERAGE)
        self._need(c.CALC_DERIVATIVE)
        return self._s["purity_derivative"]


class TrainingKernel(_TrainingBase):
    """kernel.h:111-280.  TrainingSet = (feature (N,2), label (N,) complex or real; the real part is used)."""
    NumTotalParameters = REAL_NPARAM

    def __init__(self, Parameter, TrainingSet, IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative, api=None):
        api = api or default_api()
        assert len(Parameter) == self.NumTotalParameters
        flags = _flags(IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative)
        fit = api.real_fit(Parameter, TrainingSet[0], TrainingSet[1], flags, defer_scalars=True)
        super().__init__(fit, Parameter, TrainingSet, flags, api)

    def get_formatted_parameters(self):
        p = self.Params
        return (p[0], np.array(p[1:3]), p[3])

    def get_kernel(self):
        return self._fit.get(c.R_KERNEL)

    def get_inverse(self):
        return self._fit.get(c.R_INVERSE)

    def get_inverse_times_label(self):
        return self._fit.get(c.R_INVLBL)

    def get_population(self):
        self._need(c.CALC_AVERAGE)
        return self._s["population"]

    def get_1st_order_average(self):
        self._need(c.CALC_AVERAGE)
        return self._s["first_order_average"]

    def get_inverse_times_label_derivative(self):
        self._need(c.CALC_DERIVATIVE)
        return self._fit.get(c.R_INVLBL_DERIV)

    def get_population_derivative(self):
        self._need(c.CALC_AVERAGE)
        self._need(c.CALC_DERIVATIVE)
        return self._s["population_derivative"]


class TrainingComplexKernel(_TrainingBase):
    """complex_kernel.h:150-318."""
    NumTotalParameters = COMPLEX_NPARAM

    def __init__(self, Parameter, TrainingSet, IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative, api=None):
        api = api or default_api()
        assert len(Parameter) == self.NumTotalParameters
        flags = _flags(IsToCalculateError, IsToCalculateAverage, IsToCalculateDerivative)
        fit = api.complex_fit(Parameter, TrainingSet[0], TrainingSet[1], flags, defer_scalars=True)
        super().__init__(fit, Parameter, TrainingSet, flags, api)

    def get_kernel(self):
        return self._fit.get(c.C_KERNEL)

    def get_pseudo_kernel(self):
        return self._fit.get(c.C_PSEUDO)

    def get_upper_left_block_of_augmented_inverse(self):
        return self._fit.get(c.C_UPPER_LEFT)

    def get_lower_left_block_of_augmented_inverse(self):
        return self._fit.get(c.C_LOWER_LEFT)

    def get_upper_part_of_augmented_inverse_times_label(self):
        return self._fit.get(c.C_INVLBL)

    def get_upper_part_of_augmented_inverse_times_label_derivative(self):
        self._need(c.CALC_DERIVATIVE)
        return self._fit.get(c.C_INVLBL_DERIV)


class _PredictiveBase:
    def __init__(self, res, rescale, has_label, deriv):
        self._r, self.RescaleFactor, self._has_label, self._deriv = res, rescale, has_label, deriv

    def get_prediction(self):
        """the (rescaled, uncut) member `Prediction` (kernel.h:392); the reference keeps it private"""
        return self._r["prediction"]

    def get_variance(self):
        return self._r["variance"]

    def get_cutoff_prediction(self):
        return self._r["cutoff"]

    def get_error(self):
        assert self._has_label
        return self._r["error"]

    def get_error_derivative(self):
        assert self._has_label and self._deriv
        return self._r["error_derivative"]


class PredictiveKernel(_PredictiveBase):
    """kernel.h:336-403: PredictiveKernel(TestFeature, kernel, IsToCalculateDerivative, TestLabel = nullopt)."""

    def __init__(self, TestFeature, kernel, IsToCalculateDerivative, TestLabel=None):
        Xs = np.atleast_2d(np.asarray(TestFeature, dtype=float))
        res = kernel._api.real_predict(kernel._fit, Xs, flags=c.CALC_DERIVATIVE if IsToCalculateDerivative else 0, labels=TestLabel)
        super().__init__(res, kernel.get_rescale_factor(), TestLabel is not None, IsToCalculateDerivative)


class PredictiveComplexKernel(_PredictiveBase):
    """complex_kernel.h:323-391."""

    def __init__(self, TestFeature, kernel, IsToCalculateDerivative, TestLabel=None):
        Xs = np.atleast_2d(np.asarray(TestFeature, dtype=float))
        res = kernel._api.complex_predict(kernel._fit, Xs, flags=c.CALC_DERIVATIVE if IsToCalculateDerivative else 0, labels=TestLabel)
        super().__init__(res, kernel.get_rescale_factor(), TestLabel is not None, IsToCalculateDerivative)


# ---- storage.h: QuantumStorage (lower-triangular element container) ------------------------------------------------
def calculate_offdiagonal_index(RowIndex, ColIndex):
    assert ColIndex < RowIndex  # storage.h:22-26
    return RowIndex * (RowIndex - 1) // 2 + ColIndex


def element_order(num_pes=None):
    """(iPES, jPES) in the order the reference packs parameters: rows, then columns up to the diagonal (opt.cpp:805-837)."""
    n = num_pes or NumPES
    return [(i, j) for i in range(n) for j in range(i + 1)]


def construct_training_sets(density, num_pes=None):
    """predict.cpp:246-280. density: dict {(iPES, jPES): (r (N,2), rho (N,) complex)} or a list of PhaseSpacePoint-like
    (r, rho) pairs per element; returns {(i, j): (feature, label)} with empty sets for absent elements."""
    out = {}
    for (i, j) in element_order(num_pes):
        pts = density.get((i, j)) if hasattr(density, "get") else density[i][j]
        if pts is None or len(pts) == 0:
            out[(i, j)] = (np.zeros((0, 2)), np.zeros(0, dtype=complex))
        elif isinstance(pts, tuple) and len(pts) == 2 and np.ndim(pts[0]) == 2:
            out[(i, j)] = (np.asarray(pts[0], dtype=float), np.asarray(pts[1], dtype=complex))
        else:  # AoS of (r, rho)
            out[(i, j)] = (np.array([p[0] for p in pts], dtype=float), np.array([p[1] for p in pts], dtype=complex))
    return out


class TrainingKernels:
    """predict.h:89-143: one optional TrainingKernel per diagonal element, one optional TrainingComplexKernel per strictly
    lower element.  ParameterVectors / TrainingSets: dict {(iPES, jPES): ...}.  Elements are independent GPs."""

    def __init__(self, ParameterVectors, TrainingSets, IsToCalculateError=True, IsToCalculateAverage=True,
                 IsToCalculateDerivative=False, api=None, num_pes=None):
        self.num_pes = num_pes or NumPES

        def build(one_api, e):
            (i, j) = e
            feature, label = TrainingSets[e]
            params = ParameterVectors[e]
            if len(feature) == 0:
                return None  # predict.cpp:308-315
            if i == j:
                return TrainingKernel(params, (feature, label), IsToCalculateError, IsToCalculateAverage,
                                      IsToCalculateDerivative, api=one_api)
            if all(p == 0 for p in params):
                return None  # predict.cpp:339-357
            return TrainingComplexKernel(params, (feature, label), IsToCalculateError, IsToCalculateAverage,
                                         IsToCalculateDerivative, api=one_api)

        order = element_order(self.num_pes)
        if isinstance(api, ApiPool):  # elements on separate streams, built concurrently
            built = api.map(build, order)
        else:
            built = [build(api, e) for e in order]
        self._k = dict(zip(order, built))

    def __call__(self, iPES, jPES=None):
        return self._k[(iPES, iPES if jPES is None else jPES)]

    def calculate_population(self):  # predict.cpp:395-406
        return sum(self(i).get_population() for i in range(self.num_pes) if self(i) is not None)

    def calculate_1st_order_average(self):  # predict.cpp:408-419
        r = np.zeros(PhaseDim)
        for i in range(self.num_pes):
            if self(i) is not None:
                r = r + self(i).get_1st_order_average()
        return r

    def calculate_total_energy_average(self, Energies):  # predict.cpp:423-436
        return sum(self(i).get_population() * Energies[i] for i in range(self.num_pes) if self(i) is not None)

    def calculate_purity(self):  # predict.cpp:439-463: weight 1 on the diagonal, 2 off the diagonal
        total = 0.0
        for (i, j) in element_order(self.num_pes):
            if self(i, j) is not None:
                total += (1.0 if i == j else 2.0) * self(i, j).get_purity()
        return total

    def population_derivative(self):  # predict.cpp:465-484
        out = np.zeros(self.num_pes * REAL_NPARAM)
        for i in range(self.num_pes):
            if self(i) is not None:
                out[i * REAL_NPARAM:(i + 1) * REAL_NPARAM] = self(i).get_population_derivative()
        return out

    def total_energy_derivative(self, Energies):  # predict.cpp:486-510
        out = np.zeros(self.num_pes * REAL_NPARAM)
        for i in range(self.num_pes):
            if self(i) is not None:
                out[i * REAL_NPARAM:(i + 1) * REAL_NPARAM] = np.asarray(self(i).get_population_derivative()) * Energies[i]
        return out

    def purity_derivative(self):  # predict.cpp:512-559
        out, pos = [], 0
        for (i, j) in element_order(self.num_pes):
            n = REAL_NPARAM if i == j else COMPLEX_NPARAM
            k = self(i, j)
            if k is None:
                out.append(np.zeros(n))
            else:
                out.append(np.asarray(k.get_purity_derivative()) * (1.0 if i == j else 2.0))
            pos += n
        return np.concatenate(out)


NumTotalParameters = REAL_NPARAM * NumPES + COMPLEX_NPARAM * (NumPES * (NumPES - 1) // 2)  # predict.h:17


# ---- opt.cpp: objective / constraint wrappers ------------------------------------------------------------------------
def make_normal(d):
    """opt.cpp:420-431"""
    return float(np.finfo(float).max) if (math.isnan(d) or math.isinf(d)) else d


def _log_indices(n):
    # complex: the two sub-kernel magnitudes and the noise; real: the noise (opt.cpp:109-144)
    return [1, 4, 7] if n == COMPLEX_NPARAM else [3]


def local_parameter_to_global(param):  # opt.cpp:109-144
    out = list(map(float, param))
    for i in _log_indices(len(param)):
        out[i] = math.log(out[i])
    return out


def global_parameter_to_local(param):  # opt.cpp:197-232
    out = list(map(float, param))
    for i in _log_indices(len(param)):
        out[i] = math.exp(out[i])
    return out


def local_gradient_to_global(param, grad):  # opt.cpp:155-192: d/d ln x = x d/dx
    out = list(map(float, grad))
    if len(out) == 0:
        return out
    for i in _log_indices(len(param)):
        out[i] *= param[i]
    return out


def loose_function(x, grad, params, api=None):
    """opt.cpp:441-482.  grad: a list of len(x) to be filled in place, or an empty list for 'no gradient' (the NLopt
    convention).  params = (TrainingSet, ExtraTrainingSet), each (feature, label); an optional third entry is a resident
    objective (`api.objective(...)`: the same two sets already on the device) that is then evaluated instead."""
    api = api or default_api()
    if len(params) > 2 and params[2] is not None:
        value, g = params[2](x, want_grad=len(grad) > 0)
    else:
        (X, y), (Xe, ye) = params[0], params[1]
        value, g = api.loose_function(x, X, np.asarray(y, dtype=complex), Xe, np.asarray(ye, dtype=complex), want_grad=len(grad) > 0)
    if len(grad) > 0:
        grad[:] = list(g)
    return value


def resident_objective(cache, api, element, TrainingSets, ExtraTrainingSets):
    """One resident objective per (context, element) for the lifetime of `cache` (a dict the caller owns, e.g. one per
    Optimization.optimize call): NLopt's `void* params` made device-resident.  None when no cache is given."""
    if cache is None:
        return None
    key = (id(api), element)
    if key not in cache:
        (X, y), (Xe, ye) = TrainingSets[element], ExtraTrainingSets[element]
        cache[key] = api.objective(X, y, Xe, ye)
    return cache[key]


def loose_function_global_wrapper(x, grad, params, api=None):  # opt.cpp:489-497
    grad_local = list(grad)
    x_local = global_parameter_to_local(x)
    result = loose_function(x_local, grad_local, params, api=api)
    grad[:] = local_gradient_to_global(x_local, grad_local)
    return result


def diagonal_loose(x, grad, params, api=None, num_pes=None):
    """opt.cpp:594-617: sum of loose_function over the diagonal elements on parameter slices of 4."""
    TrainingSets, ExtraTrainingSets = params[0], params[1]
    cache = params[2] if len(params) > 2 else None
    n = num_pes or NumPES
    active = [i for i in range(n) if len(TrainingSets[(i, i)][0]) != 0]

    def one(one_api, i):
        g = [0.0] * REAL_NPARAM if len(grad) > 0 else []
        obj = resident_objective(cache, one_api or default_api(), (i, i), TrainingSets, ExtraTrainingSets)
        v = loose_function(x[i * REAL_NPARAM:(i + 1) * REAL_NPARAM], g, (TrainingSets[(i, i)], ExtraTrainingSets[(i, i)], obj), api=one_api)
        return v, g

    results = api.map(one, active) if isinstance(api, ApiPool) else [one(api, i) for i in active]
    err = 0.0
    for i, (v, g) in zip(active, results):  # summed in element order either way
        err += v
        if len(grad) > 0:
            grad[i * REAL_NPARAM:(i + 1) * REAL_NPARAM] = g
    if len(grad) > 0:
        grad[:] = [make_normal(d) for d in grad]
    return make_normal(err)


def construct_all_parameters(x, num_pes=None):  # opt.cpp:805-820
    out, pos = {}, 0
    for (i, j) in element_order(num_pes):
        n = REAL_NPARAM if i == j else COMPLEX_NPARAM
        out[(i, j)] = list(x[pos:pos + n])
        pos += n
    return out


def construct_all_parameters_from_diagonal(x, num_pes=None):  # opt.cpp:622-635
    return {(i, j): (list(x[i * REAL_NPARAM:(i + 1) * REAL_NPARAM]) if i == j else [0.0] * COMPLEX_NPARAM)
            for (i, j) in element_order(num_pes)}


def construct_combined_parameters(x, num_pes=None):  # opt.cpp:825-837
    return [v for e in element_order(num_pes) for v in x[e]]


def full_loose(x, grad, params, api=None, num_pes=None):
    """opt.cpp:844-870"""
    TrainingSets, ExtraTrainingSets = params[0], params[1]
    cache = params[2] if len(params) > 2 else None
    allp = construct_all_parameters(x, num_pes)
    order = element_order(num_pes)

    def one(one_api, e):
        n = REAL_NPARAM if e[0] == e[1] else COMPLEX_NPARAM
        g = [0.0] * n if len(grad) > 0 else []
        if len(TrainingSets[e][0]) == 0:
            return 0.0, g
        obj = resident_objective(cache, one_api or default_api(), e, TrainingSets, ExtraTrainingSets)
        return loose_function(allp[e], g, (TrainingSets[e], ExtraTrainingSets[e], obj), api=one_api), g

    results = api.map(one, order) if isinstance(api, ApiPool) else [one(api, e) for e in order]
    err, grads = 0.0, {}
    for e, (v, g) in zip(order, results):
        err += v
        grads[e] = g
    if len(grad) > 0:
        grad[:] = [make_normal(d) for d in construct_combined_parameters(grads, num_pes)]
    return make_normal(err)


def diagonal_constraints(NumConstraints, x, want_grad, params, api=None, num_pes=None):
    """opt.cpp:644-719: [population - 1, energy - E0, (purity - S0)] and its row-major (m x n) gradient.
    params = (TrainingSets, Energies, TotalEnergy, Purity). Returns (result, grad or None)."""
    TrainingSets, Energies, TotalEnergy, Purity = params
    n = num_pes or NumPES
    ks = TrainingKernels(construct_all_parameters_from_diagonal(x, n), TrainingSets, False, True, want_grad, api=api, num_pes=n)
    result = [ks.calculate_population() - 1.0, ks.calculate_total_energy_average(Energies) - TotalEnergy]
    if NumConstraints == 3:
        result.append(ks.calculate_purity() - Purity)
    grad = None
    if want_grad:
        rows = [ks.population_derivative(), ks.total_energy_derivative(Energies)]
        if NumConstraints == 3:
            pd = construct_all_parameters(ks.purity_derivative(), n)
            rows.append(np.concatenate([pd[(i, i)] for i in range(n)]))
        grad = [make_normal(float(d)) for d in np.concatenate(rows)]
    return [make_normal(r) for r in result], grad


def full_constraints(x, want_grad, params, api=None, num_pes=None):
    """opt.cpp:879-929"""
    TrainingSets, Energies, TotalEnergy, Purity = params
    n = num_pes or NumPES
    ks = TrainingKernels(construct_all_parameters(x, n), TrainingSets, False, True, want_grad, api=api, num_pes=n)
    result = [ks.calculate_population() - 1.0, ks.calculate_total_energy_average(Energies) - TotalEnergy, ks.calculate_purity() - Purity]
    grad = None
    if want_grad:
        widen = lambda diag: construct_combined_parameters(construct_all_parameters_from_diagonal(diag, n), n)
        rows = [widen(ks.population_derivative()), widen(ks.total_energy_derivative(Energies)), list(ks.purity_derivative())]
        grad = [make_normal(float(d)) for r in rows for d in r]
    return [make_normal(r) for r in result], grad


# ---- "next" row N1: batched replacement of the per-point DistributionFunction ---------------------------
def predict_distribution(all_kernels, r, RowIndex, ColIndex):
    """Batched form of main.cpp:75-101's `predict_distribution` lambda: the cut-off prediction of density-matrix element
    (RowIndex, ColIndex) at every phase-space point of r (B, 2) in ONE predict call (the reference constructs one
    PredictiveKernel per point, O(N^2) each, ~10^2 N times per tick: evolve.cpp:298, mc.cpp:158-172).
    Returns a complex array (B,); zeros when the element has no kernel (main.cpp:86-88, 97-99)."""
    r = np.atleast_2d(np.asarray(r, dtype=float))
    k = all_kernels(RowIndex, ColIndex) if RowIndex != ColIndex else all_kernels(RowIndex)
    if k is None or len(r) == 0:
        return np.zeros(len(r), dtype=complex)
    if RowIndex == ColIndex:
        return PredictiveKernel(r, k, False).get_cutoff_prediction().astype(complex)
    return PredictiveComplexKernel(r, k, False).get_cutoff_prediction()


class DistributionBatcher:
    """Gather - predict - scatter queue for the callers that evaluate the distribution point by point (evolve.cpp:184-372
    asks for 8 back-propagated points per sample, mc.cpp:143-188 for one per Metropolis step): callers `request` points and
    keep the returned ticket, `flush` runs one predict per density-matrix element, `result(ticket)` hands the values back."""

    def __init__(self, all_kernels):
        self.all_kernels = all_kernels
        self._pending = {}   # (row, col) -> list of (ticket, points)
        self._results = {}
        self._next = 0

    def request(self, r, RowIndex, ColIndex):
        ticket = self._next
        self._next += 1
        self._pending.setdefault((RowIndex, ColIndex), []).append((ticket, np.atleast_2d(np.asarray(r, dtype=float))))
        return ticket

    def flush(self):
        for (row, col), items in self._pending.items():
            pts = np.concatenate([p for _, p in items], axis=0)
            vals = predict_distribution(self.all_kernels, pts, row, col)
            pos = 0
            for ticket, p in items:
                self._results[ticket] = vals[pos:pos + len(p)]
                pos += len(p)
        self._pending = {}

    def result(self, ticket):
        return self._results.pop(ticket)
