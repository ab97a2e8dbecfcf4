"""CPU: the Python mirror of the reference interface (kernels.py) driven through the oracle binding — class names, getters,
assert-style errors, TrainingKernels aggregation, parameter packing, log-reparametrisation and make_normal."""
import math
import os
import sys

import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import kernels as K
from tests import parity
from tests.conftest import ROOT, load_golden


def test_training_and_predictive_kernel_getters(oracle):
    g = load_golden("real_a")
    k = K.TrainingKernel(g["theta"], (g["X"], g["y"].astype(complex)), True, True, True, api=oracle)
    assert abs(k.get_error() - g["error"]) < 1e-9 * g["error"]
    assert abs(k.get_population() - g["population"]) < 1e-9
    assert np.allclose(k.get_inverse(), g["W"], rtol=1e-9, atol=1e-9 * np.abs(g["W"]).max())
    assert np.allclose(k.get_inverse_times_label_derivative(), g["dv"], rtol=1e-8, atol=1e-8 * np.abs(g["dv"]).max())
    p = K.PredictiveKernel(g["Xv"], k, True, g["tv"])
    assert abs(p.get_error() - g["v_error"]) < 1e-9 * g["v_error"]
    assert np.allclose(p.get_error_derivative(), g["v_error_derivative"], rtol=1e-7, atol=1e-9)
    one = K.PredictiveKernel(g["Xs"][0], k, False)  # a single 2-vector, like main.cpp:83
    assert one.get_cutoff_prediction().shape == (1,)


def test_getters_assert_like_the_reference(oracle):
    g = load_golden("real_c")
    k = K.TrainingKernel(g["theta"], (g["X"], g["y"]), True, False, False, api=oracle)
    with pytest.raises(AssertionError):
        k.get_population()
    with pytest.raises(AssertionError):
        k.get_error_derivative()
    p = K.PredictiveKernel(g["Xs"], k, False)
    with pytest.raises(AssertionError):
        p.get_error()


def _sets():
    ga, gc = load_golden("real_a"), load_golden("complex_a")
    sets = {(0, 0): (ga["X"], ga["y"].astype(complex)), (1, 0): (gc["X"], gc["y"]), (1, 1): (np.zeros((0, 2)), np.zeros(0, complex))}
    params = {(0, 0): list(ga["theta"]), (1, 0): list(gc["theta"]), (1, 1): [1.0, 0.7, 0.7, 0.01]}
    return ga, gc, sets, params


def test_training_kernels_aggregates(oracle):
    ga, gc, sets, params = _sets()
    ks = K.TrainingKernels(params, sets, False, True, True, api=oracle)
    assert ks(1) is None  # empty element is skipped (predict.cpp:308-315)
    assert abs(ks.calculate_population() - ga["population"]) < 1e-9
    assert abs(ks.calculate_purity() - (ga["purity"] + 2 * gc["purity"])) < 1e-8
    assert abs(ks.calculate_total_energy_average([0.3, 0.7]) - 0.3 * ga["population"]) < 1e-9
    pd = ks.purity_derivative()
    assert len(pd) == K.NumTotalParameters == 16
    assert np.allclose(pd[:4], ga["purity_derivative"], rtol=1e-6, atol=1e-9)
    assert np.allclose(pd[4:12], 2 * gc["purity_derivative"], rtol=1e-6, atol=1e-9)
    assert np.all(pd[12:] == 0)
    # all-zero off-diagonal parameters switch the complex element off (predict.cpp:339-357)
    params0 = dict(params)
    params0[(1, 0)] = [0.0] * 8
    assert K.TrainingKernels(params0, sets, False, True, False, api=oracle)(1, 0) is None


def test_parameter_packing_roundtrip():
    x = list(np.arange(16.0))
    allp = K.construct_all_parameters(x)
    assert allp[(0, 0)] == [0, 1, 2, 3] and allp[(1, 0)] == list(range(4, 12)) and allp[(1, 1)] == [12, 13, 14, 15]
    assert K.construct_combined_parameters(allp) == x
    d = K.construct_all_parameters_from_diagonal(list(np.arange(8.0)))
    assert d[(1, 1)] == [4, 5, 6, 7] and d[(1, 0)] == [0.0] * 8
    assert K.calculate_offdiagonal_index(2, 1) == 2


def test_log_reparametrisation_and_make_normal():
    p4, p8 = [1.0, 0.7, 0.8, 0.01], [1.0, 2.0, 0.7, 0.8, 0.5, 0.6, 0.9, 0.01]
    for p in (p4, p8):
        g = K.local_parameter_to_global(p)
        assert np.allclose(K.global_parameter_to_local(g), p)
    assert K.local_parameter_to_global(p4)[3] == math.log(0.01) and K.local_parameter_to_global(p4)[:3] == p4[:3]
    g8 = K.local_gradient_to_global(p8, [1.0] * 8)
    assert g8 == [1.0, 2.0, 1.0, 1.0, 0.5, 1.0, 1.0, 0.01]
    assert K.local_gradient_to_global(p4, []) == []
    assert K.make_normal(float("nan")) == np.finfo(float).max and K.make_normal(float("inf")) == np.finfo(float).max
    assert K.make_normal(-3.0) == -3.0


def test_objective_wrappers(oracle):
    ga, gc, sets, params = _sets()
    extra = {(0, 0): (ga["Xv"], ga["tv"].astype(complex)), (1, 0): (gc["Xv"], gc["tv"]), (1, 1): (np.zeros((0, 2)), np.zeros(0, complex))}
    x = K.construct_combined_parameters(params)
    grad = [0.0] * 16
    val = K.full_loose(x, grad, (sets, extra), api=oracle)
    assert abs(val - (ga["error"] + ga["v_error"] + gc["error"] + gc["v_error"])) < 1e-8 * val
    assert np.allclose(grad[:4], ga["error_derivative"] + ga["v_error_derivative"], rtol=1e-6, atol=1e-9)
    assert grad[12:] == [0.0] * 4
    assert abs(K.full_loose(x, [], (sets, extra), api=oracle) - val) < 1e-12 * val
    gd = [0.0] * 8
    vd = K.diagonal_loose(x[:4] + params[(1, 1)], gd, (sets, extra), api=oracle)
    assert abs(vd - (ga["error"] + ga["v_error"])) < 1e-9 * vd
    # global wrapper: gradient wrt ln(noise) = noise * d/dnoise
    g1, g2 = [0.0] * 4, [0.0] * 4
    K.loose_function(params[(0, 0)], g1, (sets[(0, 0)], extra[(0, 0)]), api=oracle)
    K.loose_function_global_wrapper(K.local_parameter_to_global(params[(0, 0)]), g2, (sets[(0, 0)], extra[(0, 0)]), api=oracle)
    assert np.allclose(g2[:3], g1[:3]) and abs(g2[3] - g1[3] * params[(0, 0)][3]) < 1e-9 * abs(g2[3])
    res, cg = K.full_constraints(x, True, (sets, [0.3, 0.7], 0.1, 1.0), api=oracle)
    assert abs(res[0] - (ga["population"] - 1.0)) < 1e-9 and len(cg) == 48
    res2, cg2 = K.diagonal_constraints(3, x[:4] + params[(1, 1)], True, (sets, [0.3, 0.7], 0.1, 1.0), api=oracle)
    assert len(res2) == 3 and len(cg2) == 24 and np.allclose(cg2[:4], ga["population_derivative"], rtol=1e-6, atol=1e-9)


def test_batched_distribution_matches_pointwise(oracle):
    """N1: the gather-predict-scatter queue returns what the reference's one-point lambda (main.cpp:75-101) returns"""
    ga, gc, sets, params = _sets()
    ks = K.TrainingKernels(params, sets, True, True, False, api=oracle)
    rng = np.random.default_rng(3)
    pts = ga["Xs"][:12]
    batch = K.predict_distribution(ks, pts, 0, 0)
    single = np.array([K.PredictiveKernel(p, ks(0), False).get_cutoff_prediction()[0] for p in pts])
    assert np.allclose(batch.real, single, rtol=0, atol=1e-15) and np.all(batch.imag == 0)
    assert np.all(K.predict_distribution(ks, pts, 1, 1) == 0)  # element without a kernel -> 0 (main.cpp:86-88)
    q = K.DistributionBatcher(ks)
    t1 = q.request(pts[:5], 0, 0)
    t2 = q.request(gc["Xs"][:4], 1, 0)
    t3 = q.request(pts[5:6], 0, 0)
    q.flush()
    assert np.allclose(q.result(t1), batch[:5]) and np.allclose(q.result(t3), batch[5:6])
    ref = K.PredictiveComplexKernel(gc["Xs"][:4], ks(1, 0), False).get_cutoff_prediction()
    assert np.allclose(q.result(t2), ref, rtol=0, atol=1e-15)


def test_output_writers_layout(oracle):
    """N4: phase.txt / var.txt / param.txt line structure (output.cpp:120-133, 180-232) on a tiny case evaluated by the oracle."""
    import io
    from gaussian_process_liouville_equation_amd import kernels as K, optimization as O, output
    rng = np.random.default_rng(3)
    r = rng.normal(size=(30, 2)) * (1.0, 0.5) + (0.0, 10.0)
    rho = np.exp(-0.5 * (((r - (0.0, 10.0)) / (1.0, 0.5)) ** 2).sum(axis=1)) / math.pi
    ts = K.construct_training_sets({(0, 0): (r, rho.astype(complex)), (1, 0): (r, 0.3j * rho)})
    pv = {(0, 0): [1.0, 1.0, 0.5, 1e-2], (1, 0): [1.0, 1.0, 1.0, 0.5, 1.0, 1.0, 0.5, 1e-2], (1, 1): [1.0, 1.0, 0.5, 1e-2]}
    ks = K.TrainingKernels(pv, ts, False, True, False, api=oracle)
    grid = np.stack(np.meshgrid(np.linspace(-2, 2, 5), np.linspace(8, 12, 4), indexing="ij"), axis=-1).reshape(-1, 2)
    ph, va = io.StringIO(), io.StringIO()
    output.output_phase(ph, va, ks, grid)
    pl, vl = ph.getvalue().split("\n"), va.getvalue().split("\n")
    assert len(pl) == 3 * 2 + 2 and pl[-1] == "" and pl[-2] == "" and len(vl) == 3 + 2
    assert all(len(line.split()) == 20 for line in pl[:6] + vl[:3])
    assert set(pl[1].split()) == {"0"} and set(pl[4].split()) == {"0"} and set(pl[5].split()) == {"0"}  # Im rho00; rho11 absent
    assert any(float(x) != 0 for x in pl[3].split())  # Im rho10
    opt = O.Optimization((1.0, 0.5), (-6.0, 4.0), (6.0, 16.0), 2000.0, 0.025, 1.0, api=oracle)
    f = io.StringIO()
    output.output_param(f, opt)
    lines = f.getvalue().split("\n")
    assert len(lines) == 3 * 3 + 2 and [len(x.split()) for x in lines[:9]] == [4, 4, 4, 8, 8, 8, 4, 4, 4]


@pytest.mark.parametrize("num_pes", [2, 3])
def test_cpp_adapters_compile_in_the_reference_include_order(num_pes):
    """host/kernel.h, complex_kernel.h, predict.h, opt.h behind a stdafx.h / storage.h with the reference's include guards and global
    names (tests/cpp/ref_env, this image has no Eigen): the reference's call patterns (opt.cpp:74-232, 441-482, 622-719,
    1179-1195; main.cpp:74-101; output.cpp:181-290) compile without redefinitions or ambiguities.  Syntax only: no GPU here."""
    import os
    import subprocess
    from tests.conftest import ROOT
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", f"-DGPLE_TEST_NUM_PES={num_pes}",
           "-I" + os.path.join(ROOT, "tests", "cpp", "ref_env"), "-I" + os.path.join(ROOT, "gaussian_process_liouville_equation_amd", "host"),
           ]
    for driver in ("dropin_callers.cpp", "dropin_opt.cpp"):  # main.cpp:66-74 + output.cpp:120-132 on host/opt.h in the second
        r = subprocess.run(cmd + [os.path.join(ROOT, "tests", "cpp", driver)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
    for header in ("kernel.h", "complex_kernel.h", "predict.h", "gple_host.h", "opt.h"):
        text = open(os.path.join(ROOT, "gaussian_process_liouville_equation_amd", "host", header)).read()
        assert "using namespace" not in text
        for name in ("NumPES =", "NumOffDiagonalElements =", "Dim =", "PhaseDim =", "class QuantumStorage", "calculate_offdiagonal_index("):
            assert name not in text, (header, name)  # stdafx.h:107-155 / storage.h stay the reference's


def test_oracle_complex_kernel_base_against_fixture(oracle):
    """oracle_complex_gram (ComplexKernelBase as a whole) against the 50-digit fixtures' K and K~"""
    for name, _ in parity.COMPLEX_FIXTURES:
        g = load_golden(name)
        Ko, Kto = oracle.complex_gram(g["theta"], g["X"], g["X"], True)
        assert parity.rel(Ko, g["K"]) <= 8 * parity.EPS and parity.rel(Kto, g["Kt"]) <= 8 * parity.EPS


def test_native_searches_on_analytic_problems():
    """csrc/gple_opt.hip (N2): the library's own Nelder-Mead and augmented Lagrangian behind NLopt's callback ABIs — host code,
    so it runs without a GPU.  Rosenbrock in a box, a fixed coordinate, an active bound, an equality-constrained quadratic."""
    import gaussian_process_liouville_equation_amd as pkg
    from gaussian_process_liouville_equation_amd import _capi as c
    lib = pkg.load_library()
    rosen = lambda x: 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
    x, f, n = c.minimize_neldermead(lib, rosen, [-1.2, 1.0], [-5, -5], [5, 5], maxeval=2000)
    assert np.allclose(x, [1, 1], atol=1e-4) and f < 1e-9 and 50 < n <= 2000
    x, f, _ = c.minimize_neldermead(lib, rosen, [-1.2, 1.0, 3.0], [-5, -5, 3.0], [0.5, 5, 3.0], maxeval=2000)
    assert abs(x[0] - 0.5) < 1e-6 and abs(x[1] - 0.25) < 1e-3 and x[2] == 3.0  # active bound, fixed coordinate
    fun = lambda x, g: (x[0] ** 2 + 2 * x[1] ** 2, [2 * x[0], 4 * x[1]])
    con = lambda x, g: ([x[0] + x[1] - 1.0], [1.0, 1.0])
    x, f, _ = c.minimize_auglag_eq(lib, fun, con, 1, [3.0, -1.0], [-5, -5], [5, 5])
    assert abs(x[0] + x[1] - 1.0) < 1e-6 and abs(f - 2.0 / 3.0) < 1e-5 and np.allclose(x, [2 / 3, 1 / 3], atol=2e-3)
    # two constraints pin the point: min |x|^2 s.t. x0 + x1 + x2 = 1, x0 - x2 = 0.2
    fun3 = lambda x, g: (sum(v * v for v in x), [2 * v for v in x])
    con3 = lambda x, g: ([x[0] + x[1] + x[2] - 1.0, x[0] - x[2] - 0.2], [1.0, 1.0, 1.0, 1.0, 0.0, -1.0])
    x, f, _ = c.minimize_auglag_eq(lib, fun3, con3, 2, [0.0, 0.0, 0.0], [-2] * 3, [2] * 3)
    assert np.allclose(x, [1 / 3 + 0.1, 1 / 3, 1 / 3 - 0.1], atol=2e-3)


def test_average_point_and_logging_writers(oracle):
    """N4: ave.txt / coord.txt / value.txt / run.log line structure (output.cpp:24-178, 235-302) and the Monte-Carlo observables
    behind them (predict.cpp:65-244) on a small two-surface case evaluated by the oracle"""
    import io
    from gaussian_process_liouville_equation_amd import output
    rng = np.random.default_rng(8)
    mk = lambda n, w: (rng.normal(size=(n, 2)) * (1.0, 0.5) + (0.0, 10.0), None)
    dens, extra = {}, {}
    for e, wgt in (((0, 0), 0.8), ((1, 0), 0.4j), ((1, 1), 0.2)):
        for store, n in ((dens, 30), (extra, 50)):
            r = rng.normal(size=(n, 2)) * (1.0, 0.5) + (0.0, 10.0)
            store[e] = (r, wgt * np.exp(-0.5 * (((r - (0.0, 10.0)) / (1.0, 0.5)) ** 2).sum(axis=1)) / math.pi)
    pv = {(0, 0): [1.0, 1.0, 0.5, 1e-2], (1, 0): [1.0, 1.0, 1.0, 0.5, 1.0, 1.0, 0.5, 1e-2], (1, 1): [1.0, 1.0, 0.5, 1e-2]}
    ks = K.TrainingKernels(pv, K.construct_training_sets(dens), True, True, False, api=oracle)
    pot = lambda x, i: 0.01 * (i + 1) + 0.0 * x
    f = io.StringIO()
    output.output_average(f, ks, dens, 2000.0, 1.0, potential=pot)
    vals = f.getvalue().split()
    assert len(vals) == 2 * 8 + 8 + 2 * (4 + 1) and f.getvalue().startswith(" ") and f.getvalue().endswith("\n")
    v = np.array(vals, dtype=float)
    assert np.isnan(v[3]) and np.isnan(v[11])                       # analytic energy per surface is NaN (output.cpp:52)
    assert abs(v[4] + v[12] - 1.0) < 1e-5                           # Monte-Carlo populations are normalised (predict.cpp:85)
    assert abs(v[16] - (v[0] + v[8])) < 1e-5 * abs(v[16])           # total analytic population = sum over surfaces
    assert abs(v[7] - (((dens[(0, 0)][0][:, 1] ** 2 / 4000.0 + 0.01) * dens[(0, 0)][1].real).sum() / dens[(0, 0)][1].real.sum())) < 1e-5 * abs(v[7])
    prt = v[24:28].reshape(2, 2)
    assert prt[0, 1] == prt[1, 0] and abs(v[28] - (prt[0, 0] + prt[1, 1] + 2 * prt[0, 1])) < 1e-4 * abs(v[28])
    assert abs(v[33] - v[29:33].sum()) < 1e-4 * abs(v[33])
    c, w = io.StringIO(), io.StringIO()
    output.output_point(c, w, dens, extra)
    cl, wl = c.getvalue().split("\n"), w.getvalue().split("\n")
    assert len(cl) == 3 * 2 + 2 and len(wl) == 3 * 2 + 2 and all(len(x.split()) == 80 for x in cl[:6] + wl[:6])
    assert set(wl[1].split()) == {"0"} and any(float(x) != 0 for x in wl[3].split())  # Im rho00 = 0, Im rho10 != 0
    lg = io.StringIO()
    output.output_logging(lg, 12.5, (0.0123, [5, 6, 7, 8, 9], 2), {e: (200, 1.0) for e in dens}, 3.25, ks)
    parts = lg.getvalue().split()
    assert parts[0] == "12.5" and parts[1] == "3.25" and parts[2:5] == ["200"] * 3 and parts[5:8] == ["1"] * 3 and len(parts) >= 2 + 6 + 3 + 1 + 5 + 1 + 2


def test_factorisation_layout_rules():
    """csrc/gple_chol.hip, host logic only (gple_debug_chol_layout makes no device call): the outer blocks of the Cholesky (sized by a tile
    budget per launch) and the fork points of the block-row inverse for every padded size a fit can have — ascending multiples of 64 from 0
    to n, no block or row group narrower than 256 columns; a matrix of one outer block (every n <= 2304) has no fork points — the launch that
    factors it forms the inverse as well, in n^2 doubles of scratch —, larger ones fork, with a workspace large enough for what
    chol_inverse_factor carves out of it (W of the rows below the first fork, two merge trees)"""
    import ctypes
    import gaussian_process_liouville_equation_amd as pkg
    lib = pkg.load_library()
    lib.gple_debug_chol_layout.restype = ctypes.c_int
    cap = 512
    b, f = (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
    nb, nf, wd = ctypes.c_int(), ctypes.c_int(), ctypes.c_ulonglong()
    assert lib.gple_debug_chol_layout(100, cap, b, ctypes.byref(nb), f, ctypes.byref(nf), ctypes.byref(wd)) != 0  # not a multiple of 64
    for n in list(range(256, 8192 + 1, 256)) + [12288, 16384]:
        assert lib.gple_debug_chol_layout(n, cap, b, ctypes.byref(nb), f, ctypes.byref(nf), ctypes.byref(wd)) == 0
        bounds, forks = list(b[:nb.value]), list(f[:nf.value])
        assert bounds[0] == 0 and bounds[-1] == n and all(x % 64 == 0 for x in bounds)
        widths = np.diff(bounds)
        assert np.all(widths >= 256) or len(bounds) == 2
        if n <= 2304:
            assert len(bounds) == 2  # one block: every strip fits the side workgroups of a launch
        if n >= 4096:
            assert len(bounds) > 2 and np.all(np.diff(widths[:-1]) >= 0)  # the blocks widen as the trailing matrix shrinks
        assert (len(forks) == 0) == (len(bounds) == 2)
        edges = [0] + forks + [n]
        assert all(x % 64 == 0 for x in forks) and np.all(np.diff(edges) >= 256)
        # chol_inverse_factor's carve-up of the workspace follows the fork list: W of the widest row-block product, the widest side job's merge tree,
        # the last row block's merge tree (b^2 / 4 doubles for b columns); without forks one merge tree over all n columns
        groups = [int(g) for g in np.diff(edges)]
        if forks:
            need = max(w * e for w, e in zip(groups, edges[:-1])) + max(g * g // 4 for g in groups[:-1]) + groups[-1] ** 2 // 4
            assert wd.value >= need
        else:
            assert wd.value >= n * n  # the transposed tiles of the inverse (potrf_dag_kernel)


def test_factorisation_layout_with_forced_forks():
    """GPLE_CHOL_FORKS may put the forks anywhere (read once per process: own process): a single early fork leaves a last row block of 0.7 n whose
    merge tree needs 0.12 n^2 doubles — the workspace is sized from the fork list in use, not from the default list's bounds"""
    import subprocess
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
import gaussian_process_liouville_equation_amd as pkg
lib = pkg.load_library()
b, f = (ctypes.c_int * 512)(), (ctypes.c_int * 512)()
nb, nf, wd = ctypes.c_int(), ctypes.c_int(), ctypes.c_ulonglong()
for n in (4096, 6144, 8192):
    assert lib.gple_debug_chol_layout(n, 512, b, ctypes.byref(nb), f, ctypes.byref(nf), ctypes.byref(wd)) == 0
    forks = list(f[:nf.value])
    assert len(forks) == 1 and abs(forks[0] - 0.3 * n) <= 64, forks
    last = n - forks[0]
    assert wd.value >= last * forks[0] + forks[0] ** 2 // 4 + last ** 2 // 4, (n, wd.value)
print("ok")
''' % ROOT
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GPLE_CHOL_FORKS="30"), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "ok" in res.stdout, res.stderr[-2000:]
