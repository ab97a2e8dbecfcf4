"""Row N2: the Optimization driver above the hot path.  On CPU the objective calls go to the oracle (the
checker stands in for the device so the host logic can be exercised here); the GPU test runs the same search on the HIP
library and must land on the same quality of fit."""
import math

import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import kernels as K
from gaussian_process_liouville_equation_amd import optimization as O

SIGMA = (1.0, 0.5)  # minimum-uncertainty packet, hbar = 1: purity 1
MASS = 2000.0


def _wigner(r):
    return np.exp(-0.5 * (((r - (0.0, 10.0)) / SIGMA) ** 2).sum(axis=1)) / (2 * math.pi * SIGMA[0] * SIGMA[1])


def _density(n, seed=5, offdiag=False):
    """Samples of a Gaussian Wigner function on surface 0; with offdiag, the pure two-level state
    (sqrt(.8), sqrt(.2) e^{0.7i}) x the same packet, so population = 1 and purity = 1 hold exactly."""
    rng = np.random.default_rng(seed)
    draw = lambda: rng.normal(size=(n, 2)) * SIGMA + (0.0, 10.0)
    r = draw()
    if not offdiag:
        return {(0, 0): (r, _wigner(r).astype(complex))}
    r10, r11 = draw(), draw()
    return {(0, 0): (r, (0.8 * _wigner(r)).astype(complex)), (1, 0): (r10, 0.4 * np.exp(0.7j) * _wigner(r10)),
            (1, 1): (r11, (0.2 * _wigner(r11)).astype(complex))}


def _run(api, n, offdiag=False, searches="scipy"):
    density = _density(n, offdiag=offdiag)
    e0 = O.calculate_total_energy_average_one_surface(density[(0, 0)], MASS, 0)
    opt = O.Optimization(SIGMA, (-6.0, 4.0), (6.0, 16.0), MASS, e0, 1.0, api=api, local_maxeval=150, searches=searches)
    start = K.loose_function(opt.InitialKernelParameter, [], (density[(0, 0)], (np.zeros((0, 2)), np.zeros(0, complex))), api=api)
    err, steps, kind = opt.optimize(density, {})
    return opt, density, start, err, steps, kind


def _check(opt, density, start, err, steps, kind, api):
    assert isinstance(kind, O.OptimizationType) and kind != O.OptimizationType.Default
    assert len(steps) == 3 + 2 and steps[0] > 0 and steps[3] > 0  # 3 elements + diagonal + full (opt.cpp:1160-1177)
    p = opt.get_parameters()
    lb, ub = opt.get_lower_bounds(), opt.get_upper_bounds()
    for e in p:  # every parameter but the fitted magnitude stays in the box
        for v, l, u in zip(p[e][1:], lb[e][1:], ub[e][1:]):
            assert l - 1e-12 <= v <= u + 1e-12
    assert p[(0, 0)][3] == O.InitialNoise
    assert math.isfinite(err) and err < 100 * start  # the constrained stage may trade a little error for the averages
    ks = K.TrainingKernels(p, K.construct_training_sets(density), False, True, False, api=api)
    assert abs(ks.calculate_population() - 1.0) < 2 * O.AverageTolerance
    unit = [O.InitialMagnitude, *p[(0, 0)][1:]]  # opt.cpp:1179-1186: the magnitude is read off the unit-magnitude fit
    assert p[(0, 0)][0] == pytest.approx(K.TrainingKernel(unit, density[(0, 0)], False, False, False, api=api).get_magnitude())


def test_bounds_layout():
    lb, ub = O.calculate_kernel_bounds([0.1, 0.2], [1.0, 2.0])
    assert lb == [1.0, 0.1, 0.2, 1e-2] and ub == [1.0, 1.0, 2.0, 1e-2]
    lb, ub = O.calculate_complex_kernel_bounds([0.1, 0.2], [1.0, 2.0])
    assert len(lb) == K.COMPLEX_NPARAM and lb[1] == 0.1 and ub[1] == 10.0 and lb[-1] == ub[-1] == 1e-2
    assert lb[2:4] == [0.1, 0.2] and ub[5:7] == [1.0, 2.0] and lb[4] == 0.1 and ub[4] == 10.0


def test_sample_statistics():
    d = _density(4000, seed=1)[(0, 0)]
    assert np.allclose(O.calculate_standard_deviation_one_surface(d), SIGMA, rtol=0.05)
    # rho-weighted samples of a Gaussian see half its variance
    assert O.calculate_total_energy_average_one_surface(d, MASS, 0) == pytest.approx((100.0 + 0.125) / (2 * MASS), rel=0.02)
    assert O.calculate_total_energy_average_one_surface(d, MASS, 0, potential=lambda x, i: np.full_like(x, 0.25)) == pytest.approx(
        (100.0 + 0.125) / (2 * MASS) + 0.25, rel=0.02)


def test_auglag_on_a_known_problem():
    # min (x-2)^2 + (y-1)^2  s.t. x + y = 1, inside [0,3]^2, third coordinate pinned: answer (1, 0)
    def f(x, g):
        if g:
            g[:] = [2 * (x[0] - 2), 2 * (x[1] - 1), 0.0]
        return (x[0] - 2) ** 2 + (x[1] - 1) ** 2

    def h(x, want):
        return [x[0] + x[1] - 1.0], ([1.0, 1.0, 0.0] if want else None)

    x, v, n = O._auglag_eq(f, h, 1, [0.5, 0.5, 7.0], [0, 0, 7.0], [3, 3, 7.0])
    assert x == pytest.approx([1.0, 0.0, 7.0], abs=1e-3) and v == pytest.approx(2.0, abs=1e-3) and n > 0


def _library():
    import gaussian_process_liouville_equation_amd as pkg
    return pkg.load_library()  # the searches are host code of the library: no GPU needed


def test_direct_l_known_answers():
    """gple_minimize_direct_l (the GN_DIRECT_L stand-in of the global tier, opt.h:54) on the classical test functions of the DIRECT papers
    (Jones et al. 1993, Gablonsky & Kelley 2001), with the reference's tolerances off so that only the budget stops it: the known global
    minima to the accuracy the papers report for these budgets, never worse than SciPy's DIRECT-L with the same budget"""
    import ctypes as C
    from scipy import optimize as so
    from gaussian_process_liouville_equation_amd import _capi as c
    lib = _library()

    def run(f, lb, ub, maxeval, ftol=0.0, x_fixed=None):
        n = len(lb)
        cb = c.OBJECTIVE_FN(lambda nn, xp, gp, d: float(f([xp[i] for i in range(nn)])))
        x = np.array(x_fixed if x_fixed is not None else [0.0] * n, dtype=float)
        fv, ne, dp = C.c_double(), C.c_int(), C.POINTER(C.c_double)
        o = c.OptOptions(1e-9, ftol, 1e-15, 0.0, 0.5, maxeval)
        lib.gple_minimize_direct_l.argtypes = [c.OBJECTIVE_FN, C.c_void_p, C.c_uint, dp, dp, C.POINTER(c.OptOptions), dp, dp, C.POINTER(C.c_int)]
        st = lib.gple_minimize_direct_l(cb, None, n, np.array(lb, float).ctypes.data_as(dp), np.array(ub, float).ctypes.data_as(dp), C.byref(o), x.ctypes.data_as(dp),
                                        C.cast(C.byref(fv), dp), C.byref(ne))
        return st, x, fv.value, ne.value

    def branin(x):
        return (x[1] - 5.1 / (4 * np.pi ** 2) * x[0] ** 2 + 5 / np.pi * x[0] - 6) ** 2 + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x[0]) + 10

    def sixhump(x):
        return (4 - 2.1 * x[0] ** 2 + x[0] ** 4 / 3) * x[0] ** 2 + x[0] * x[1] + (-4 + 4 * x[1] ** 2) * x[1] ** 2

    def shekel10(x):
        A = np.array([[4, 4, 4, 4], [1, 1, 1, 1], [8, 8, 8, 8], [6, 6, 6, 6], [3, 7, 3, 7], [2, 9, 2, 9], [5, 5, 3, 3], [8, 1, 8, 1], [6, 2, 6, 2], [7, 3.6, 7, 3.6]])
        cc = np.array([.1, .2, .2, .4, .4, .6, .3, .7, .5, .5])
        return -np.sum(1 / (np.sum((np.asarray(x) - A) ** 2, axis=1) + cc))

    def hartmann6(x):
        A = np.array([[10, 3, 17, 3.5, 1.7, 8], [.05, 10, 17, .1, 8, 14], [3, 3.5, 1.7, 10, 17, 8], [17, 8, .05, 10, .1, 14]])
        P = 1e-4 * np.array([[1312, 1696, 5569, 124, 8283, 5886], [2329, 4135, 8307, 3736, 1004, 9991], [2348, 1451, 3522, 2883, 3047, 6650], [4047, 8828, 8732, 5743, 1091, 381]])
        return -np.sum(np.array([1, 1.2, 3, 3.2]) * np.exp(-np.sum(A * (np.asarray(x) - P) ** 2, axis=1)))

    for f, lb, ub, fstar, budget in ((branin, [-5, 0], [10, 15], 0.3978874, 400), (sixhump, [-3, -2], [3, 2], -1.0316285, 400),
                                     (shekel10, [0] * 4, [10] * 4, -10.5364, 1000), (hartmann6, [0] * 6, [1] * 6, -3.32237, 1500)):
        st, x, fv, ne = run(f, lb, ub, budget)
        assert st == 0 and budget <= ne <= budget + 40 * len(lb)  # the iteration in flight when the budget runs out is finished
        assert abs(fv - fstar) <= 1e-4 * abs(fstar) + 1e-6, (f.__name__, fv)
        assert fv == pytest.approx(f(list(x)), abs=1e-12) and all(l <= v <= u for v, l, u in zip(x, lb, ub))
        ref = so.direct(f, list(zip(lb, ub)), locally_biased=True, maxfun=budget, f_min_rtol=0, vol_tol=0, len_tol=1e-12)
        assert fv <= ref.fun + 1e-3 * abs(fstar)
    # the reference's tolerances (opt.cpp:344-345) stop the search the way NLopt's cdirect does: at the first iteration that improves
    # the minimum by less than ftol_rel
    st, x, fv, ne = run(branin, [-5, 0], [10, 15], 100000, ftol=1e-5)
    assert st == 0 and ne < 2000 and abs(fv - 0.3978874) < 1e-3
    # a fixed coordinate (lb == ub: sigma_f and the noise in the reference's boxes) is left alone, the others are searched
    st, x, fv, ne = run(lambda v: (v[0] - 1.0) ** 2 + (v[1] - 7.0) ** 2 + (v[2] + 0.5) ** 2, [-2, 7.0, -2], [2, 7.0, 2], 600, x_fixed=[0, 7.0, 0])
    assert st == 0 and x[1] == 7.0 and abs(x[0] - 1.0) < 1e-3 and abs(x[2] + 0.5) < 1e-3
    # an infinite box is refused (DIRECT needs a finite domain), NaN values count as +max (make_normal, opt.cpp:420-431)
    assert run(branin, [-np.inf, 0], [10, 15], 100)[0] != 0
    st, x, fv, ne = run(lambda v: float("nan") if v[0] < 0 else (v[0] - 0.25) ** 2, [-1], [1], 200)
    assert st == 0 and abs(x[0] - 0.25) < 1e-3


def test_optimization_driver_on_oracle(oracle):
    _check(*_run(oracle, 60), oracle)


def test_optimization_driver_offdiagonal_on_oracle(oracle):
    opt, density, start, err, steps, kind = _run(oracle, 40, offdiag=True)
    assert all(s > 0 for s in steps)
    ks = K.TrainingKernels(opt.get_parameters(), K.construct_training_sets(density), False, True, False, api=oracle)
    assert abs(ks.calculate_population() - 1.0) < 2 * O.AverageTolerance
    assert abs(ks.calculate_purity() - 1.0) < 2 * O.AverageTolerance


@pytest.mark.gpu
def test_optimization_driver_on_device(gpu, oracle):
    out = _run(gpu, 300)
    _check(*out, gpu)
    # the device fit and the oracle agree on the objective at the optimum found
    p = out[0].get_parameters()[(0, 0)]
    ts = (out[1][(0, 0)], (np.zeros((0, 2)), np.zeros(0, complex)))
    assert K.loose_function(p, [], ts, api=gpu) == pytest.approx(K.loose_function(p, [], ts, api=oracle), rel=1e-7)


@pytest.mark.gpu
def test_optimization_driver_with_offdiagonal_element(gpu):
    opt, density, start, err, steps, kind = _run(gpu, 200, offdiag=True)
    assert all(s > 0 for s in steps)  # three element searches, then the diagonal and the full constrained stage
    p = opt.get_parameters()
    assert len(p[(1, 0)]) == K.COMPLEX_NPARAM and all(math.isfinite(v) for v in p[(1, 0)])
    assert math.isfinite(err)
    ks = K.TrainingKernels(p, K.construct_training_sets(density), False, True, False, api=gpu)
    assert abs(ks.calculate_population() - 1.0) < 2 * O.AverageTolerance
    assert abs(ks.calculate_purity() - 1.0) < 2 * O.AverageTolerance


@pytest.mark.gpu
def test_optimization_with_the_native_searches(gpu):
    """N2 on the native side: the library's own Nelder-Mead (on the resident objective, inside the library) and augmented
    Lagrangian drive the same three stages; the fit they find is as good as the SciPy-driven one and meets the constraints"""
    opt, density, start, err, steps, kind = _run(gpu, 200, offdiag=True, searches="native")
    _, _, _, err_scipy, _, _ = _run(gpu, 200, offdiag=True, searches="scipy")
    assert all(s > 0 for s in steps) and math.isfinite(err)
    assert err <= 1.5 * err_scipy + 1e-6  # no worse than the stand-in searches, within the slack two local searches differ by
    p = opt.get_parameters()
    lb, ub = opt.get_lower_bounds(), opt.get_upper_bounds()
    for e in p:
        for v, l, u in zip(p[e][1:], lb[e][1:], ub[e][1:]):
            assert l - 1e-12 <= v <= u + 1e-12
    ks = K.TrainingKernels(p, K.construct_training_sets(density), False, True, False, api=gpu)
    assert abs(ks.calculate_population() - 1.0) < 2 * O.AverageTolerance
    assert abs(ks.calculate_purity() - 1.0) < 2 * O.AverageTolerance


@pytest.mark.gpu
def test_direct_l_on_the_resident_objective(gpu):
    """the global tier's search (opt.cpp:1344-1365) inside the library: DIRECT-L in the log-parameter box on the resident objective; with
    three handles on three contexts every iteration's new rectangle centres are evaluated concurrently and the result is the sequential one;
    the Python callback form (gple_minimize_direct_l over loose_function_global_wrapper) walks the same rectangles"""
    import gaussian_process_liouville_equation_amd as pkg
    from gaussian_process_liouville_equation_amd import _capi as c
    from tests import parity
    X, y, _ = parity.synthetic_real(300, 1, 78)
    rng = np.random.default_rng(4)
    Xe = X[np.arange(900) % 300] + rng.normal(0, 0.5, size=(900, 2))
    ye = np.exp(-0.5 * (((Xe[:, 0] + 10.0) / 0.7086) ** 2 + ((Xe[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    apis = [gpu, pkg.open_api(0), pkg.open_api(0)]
    objs = [a.objective(X, y.astype(complex), Xe, ye.astype(complex)) for a in apis]
    lb, ub = [1.0, 0.05, 0.05, 1e-2], [1.0, 6.0, 6.0, 1e-2]
    flags = [i in K._log_indices(4) for i in range(4)]
    glb, gub, x0 = K.local_parameter_to_global(lb), K.local_parameter_to_global(ub), K.local_parameter_to_global([1.0, 1.0, 1.0, 1e-2])
    x1, f1, n1 = c.objective_minimize_direct_l(gpu.lib, objs[:1], x0, glb, gub, flags, 600)
    x3, f3, n3 = c.objective_minimize_direct_l(gpu.lib, objs, x0, glb, gub, flags, 600)
    assert x1 == x3 and f1 == f3 and n1 == n3 and n1 > 20
    etp = ((X, y.astype(complex)), (Xe, ye.astype(complex)), objs[0])
    xc, fc, nc = c.minimize_direct_l(gpu.lib, lambda x: K.loose_function_global_wrapper(list(x), [], etp, api=gpu), x0, glb, gub, 600)
    assert xc == x1 and fc == f1 and nc == n1
    best = K.global_parameter_to_local(x1)
    assert f1 <= objs[0]([1.0, 1.0, 1.0, 1e-2], want_grad=False)[0] and best[0] == 1.0 and abs(best[3] - 1e-2) < 1e-15
    assert 0.3 < best[1] < 1.6 and 0.3 < best[2] < 1.6  # the packet's widths are 0.71: the basin every local search of this case ends in
    for o in objs:
        o.release()
    for a in apis[1:]:
        a.close()


@pytest.mark.gpu
def test_concurrent_vertex_evaluation_gives_the_sequential_result(gpu):
    """gple_objective_minimize_neldermead with three resident objectives (three contexts = three HIP streams): the simplex
    vertices are evaluated concurrently, the decisions are those of the sequential search -> identical minimiser"""
    import gaussian_process_liouville_equation_amd as pkg
    from gaussian_process_liouville_equation_amd import _capi as c
    from tests import parity
    X, y, _ = parity.synthetic_real(400, 1, 77)
    rng = np.random.default_rng(3)
    Xe = X[np.arange(1200) % 400] + rng.normal(0, 0.5, size=(1200, 2))
    ye = np.exp(-0.5 * (((Xe[:, 0] + 10.0) / 0.7086) ** 2 + ((Xe[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    apis = [gpu, pkg.open_api(0), pkg.open_api(0)]
    objs = [a.objective(X, y.astype(complex), Xe, ye.astype(complex)) for a in apis]
    x0, lb, ub = [1.0, 1.5, 0.4, 1e-2], [1.0, 0.01, 0.01, 1e-2], [1.0, 12.0, 12.0, 1e-2]
    x1, f1, n1 = c.objective_minimize_neldermead(gpu.lib, objs[:1], x0, lb, ub, 300)
    x3, f3, n3 = c.objective_minimize_neldermead(gpu.lib, objs, x0, lb, ub, 300)
    assert x1 == x3 and f1 == f3 and n1 == n3
    assert f1 < objs[0](x0, want_grad=False)[0] and lb[1] <= x1[1] <= ub[1] and x1[0] == 1.0 and x1[3] == 1e-2
    for o in objs:
        o.release()
    for a in apis[1:]:
        a.close()


@pytest.mark.gpu
def test_api_pool_gives_the_same_kernels_and_the_same_optimum(gpu):
    """Elements on separate contexts / HIP streams / host threads: same numbers as one after the other."""
    density = _density(160, offdiag=True)
    ts = K.construct_training_sets(density)
    pv = {(0, 0): [1.0, 0.9, 0.45, 1e-2], (1, 0): [1.0, 1.1, 0.9, 0.45, 0.9, 1.0, 0.5, 1e-2], (1, 1): [1.0, 0.8, 0.5, 1e-2]}
    pool = K.ApiPool(3)
    try:
        a = K.TrainingKernels(pv, ts, True, True, True, api=gpu)
        b = K.TrainingKernels(pv, ts, True, True, True, api=pool)
        assert a.calculate_population() == b.calculate_population() and a.calculate_purity() == b.calculate_purity()
        assert np.array_equal(a.purity_derivative(), b.purity_derivative())
        assert np.array_equal(a(1, 0).get_error_derivative(), b(1, 0).get_error_derivative())
        empty = K.construct_training_sets({})
        g1, g2 = [0.0] * 16, [0.0] * 16
        x = K.construct_combined_parameters(pv)
        assert K.full_loose(x, g1, (ts, empty), api=gpu) == K.full_loose(x, g2, (ts, empty), api=pool) and g1 == g2
        e0 = O.calculate_total_energy_average_one_surface(density[(0, 0)], MASS, 0)
        res = []
        for api in (gpu, pool):
            opt = O.Optimization(SIGMA, (-6.0, 4.0), (6.0, 16.0), MASS, e0, 1.0, api=api, local_maxeval=60)
            err, steps, kind = opt.optimize(density, {})
            res.append((err, steps, opt.get_parameters()))
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
        for e in res[0][2]:
            assert list(res[0][2][e]) == list(res[1][2][e])
    finally:
        pool.close()


def test_api_pool_host_logic_on_oracle(oracle):
    """The pool's dispatch (element e -> context e mod n, results in element order) exercised on CPU with two handles of the
    checker standing in for two device contexts."""
    density = _density(40, offdiag=True)
    ts = K.construct_training_sets(density)
    pv = {(0, 0): [1.0, 0.9, 0.45, 1e-2], (1, 0): [1.0, 1.1, 0.9, 0.45, 0.9, 1.0, 0.5, 1e-2], (1, 1): [1.0, 0.8, 0.5, 1e-2]}
    pool = K.ApiPool(apis=[oracle, oracle])
    try:
        assert pool.api_for(0) is oracle and pool.api_for(3) is oracle
        assert pool.map(lambda api, item: (api is oracle, item * 2), [1, 2, 3]) == [(True, 2), (True, 4), (True, 6)]
        a = K.TrainingKernels(pv, ts, True, True, True, api=oracle)
        b = K.TrainingKernels(pv, ts, True, True, True, api=pool)
        assert a.calculate_population() == b.calculate_population() and a.calculate_purity() == b.calculate_purity()
        assert np.array_equal(a.purity_derivative(), b.purity_derivative())
        empty = K.construct_training_sets({})
        x = K.construct_combined_parameters(pv)
        g1, g2 = [0.0] * 16, [0.0] * 16
        assert K.full_loose(x, g1, (ts, empty), api=oracle) == K.full_loose(x, g2, (ts, empty), api=pool) and g1 == g2
        d1, d2 = [0.0] * 8, [0.0] * 8
        xd = pv[(0, 0)] + pv[(1, 1)]
        assert K.diagonal_loose(xd, d1, (ts, empty), api=oracle) == K.diagonal_loose(xd, d2, (ts, empty), api=pool) and d1 == d2
    finally:
        pool.close()  # handles passed in are not closed by the pool
