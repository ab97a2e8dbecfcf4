"""GPU (MI355X): the HIP path, called through the C-ABI, against the 50-digit fixtures, the CPU oracle and
size-independent identities at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", parity.REAL_FIXTURES + parity.REAL_FIXTURES_LARGE)
def test_real_fixture(gpu, name):
    parity.check_real_case(gpu, load_golden(name), deriv=True)


@pytest.mark.parametrize("name,tol", parity.COMPLEX_FIXTURES + parity.COMPLEX_FIXTURES_LARGE)
def test_complex_fixture(gpu, name, tol):
    parity.check_complex_case(gpu, load_golden(name), deriv=True, tol=tol)


@pytest.mark.parametrize("name", parity.REAL_FIXTURES)
def test_gram_and_derivatives_ulp(gpu, oracle, name):
    """KernelBase (kernel.cpp:8-242): Gram entries within (4+|a|) eps of the correctly rounded value, derivative
    matrices likewise; rectangular branch with exact-equality delta against the oracle."""
    g = load_golden(name)
    K, dK = gpu.real_gram(g["theta"], g["X"], g["X"], True, True)
    parity.check_gram_ulp(K, g["K"], g["theta"], g["X"])
    for ip in range(4):
        parity.check_gram_ulp(dK[ip], g["dK"][ip], g["theta"], g["X"])
    Kr, dKr = gpu.real_gram(g["theta"], g["Xs"], g["X"], False, True)
    Ko, dKo = oracle.real_gram(g["theta"], g["Xs"], g["X"], False, True)
    parity.check_gram_ulp(Kr, Ko, g["theta"], g["Xs"], g["X"])
    for ip in range(4):
        parity.check_gram_ulp(dKr[ip], dKo[ip], g["theta"], g["Xs"], g["X"])


def test_cutoff_factor_classes(gpu, oracle):
    """kernel.h:301-332: exact 0 / 1 classes and the cubic in between, real and complex."""
    var = np.array([1.0, 1.0, 1.0, 1.0, 4.0, 0.25, 1.0])
    pred = np.array([0.5, 1.0, 1.5, 2.0, 3.0, -0.9, -2.5])
    f = gpu.cutoff_factor(pred, var)
    assert np.array_equal(f[[0, 1, 3, 6]], [0.0, 0.0, 1.0, 1.0])
    assert np.abs(f - oracle.cutoff_factor(pred, var)).max() <= 4 * parity.EPS
    predc = pred * np.exp(0.7j)
    assert np.abs(gpu.cutoff_factor(predc, var) - oracle.cutoff_factor(predc, var)).max() <= 1e-15


@pytest.mark.parametrize("N,M,seed", [(1, 5, 1), (63, 129, 2), (257, 300, 3), (300, 1000, 4)])
def test_real_against_oracle_ragged_sizes(gpu, oracle, N, M, seed):
    """sizes that are not multiples of any tile (padding paths), theta = the reference's initial parameters"""
    X, y, Xs = parity.synthetic_real(N, M, seed)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    fg, fo = gpu.real_fit(theta, X, y, 3), oracle.real_fit(theta, X, y, 3)
    assert fg.scalars["info"] == 0
    for k in ("population", "purity", "magnitude"):
        assert abs(fg.scalars[k] - fo.scalars[k]) <= 1e-8 * abs(fo.scalars[k]), k
    assert abs(fg.scalars["error"] - fo.scalars["error"]) <= 1e-6 * abs(fo.scalars["error"])
    pg, po = gpu.real_predict(fg, Xs), oracle.real_predict(fo, Xs)
    scale = np.abs(po["prediction"]).max()
    assert np.abs(pg["prediction"] - po["prediction"]).max() <= 1e-10 * scale  # mean abs 1e-10 * s^-1 max|y|
    assert np.abs(pg["variance"] - po["variance"]).max() <= 1e-9                # variance abs 1e-9 * sf^2
    assert np.abs(pg["cutoff"] - po["cutoff"]).max() <= 1e-9 * scale / fo.scalars["rescale_factor"]


def test_empty_and_single_test_sets(gpu):
    X, y, Xs = parity.synthetic_real(40, 1, 9)
    fit = gpu.real_fit([1.0, 0.7, 0.7, 0.1], X, y, 3)
    p0 = gpu.real_predict(fit, np.zeros((0, 2)))
    assert p0["prediction"].size == 0 and np.isnan(p0["error"])
    p1 = gpu.real_predict(fit, Xs)
    assert p1["prediction"].shape == (1,) and np.isfinite(p1["variance"][0])


def test_prediction_at_training_points_uses_delta_kernel(gpu, oracle):
    """test point == training point: the noise delta enters K* and k** (kernel.cpp:26, 512) so var = k** - k W k^T"""
    X, y, _ = parity.synthetic_real(50, 1, 10)
    theta = [1.2, 0.7, 0.6, 0.3]
    fg, fo = gpu.real_fit(theta, X, y, 1), oracle.real_fit(theta, X, y, 1)
    pg, po = gpu.real_predict(fg, X), oracle.real_predict(fo, X)
    assert np.abs(pg["variance"] - po["variance"]).max() <= 1e-11
    assert np.abs(pg["prediction"] - po["prediction"]).max() <= 1e-11 * np.abs(po["prediction"]).max()
    # K v = y  =>  the (uncut) prediction at the training points reproduces the rescaled labels
    assert np.abs(pg["prediction"] - fg.get(c.R_LABEL)).max() <= 1e-10 * 10.0


def test_singular_input_does_not_abort(gpu):
    """exact duplicates with zero noise make K singular: the call must return (no abort, like the reference where
    LDLT::info() is never checked and NaN/Inf are clamped later by make_normal, opt.cpp:420-431); a reported
    breakdown (info > 0) must come with non-finite outputs, a clean factorisation with finite ones."""
    X, y, Xs = parity.synthetic_real(64, 8, 11)
    X[32:] = X[:32]
    fit = gpu.real_fit([1.0, 0.7, 0.7, 0.0], X, y, 3)
    p = gpu.real_predict(fit, Xs)
    finite = np.isfinite(fit.scalars["error"]) and np.all(np.isfinite(p["variance"]))
    assert (fit.scalars["info"] > 0) == (not finite)


@pytest.mark.parametrize("N,M,seed", [(40, 64, 21), (200, 333, 22)])
def test_complex_against_oracle(gpu, oracle, N, M, seed):
    X, yr, Xs = parity.synthetic_real(N, M, seed)
    y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05]
    fg, fo = gpu.complex_fit(theta, X, y, 3), oracle.complex_fit(theta, X, y, 3)
    assert fg.scalars["info"] == 0
    for k in ("error", "purity", "magnitude"):
        assert abs(fg.scalars[k] - fo.scalars[k]) <= 1e-7 * abs(fo.scalars[k]), k
    pg, po = gpu.complex_predict(fg, Xs), oracle.complex_predict(fo, Xs)
    scale = np.abs(po["prediction"]).max()
    assert np.abs(pg["prediction"] - po["prediction"]).max() <= 1e-9 * scale
    assert np.abs(pg["variance"] - po["variance"]).max() <= 1e-8
    assert np.abs(pg["cutoff"] - po["cutoff"]).max() <= 1e-8 * scale / fo.scalars["rescale_factor"]
    assert parity.rel(fg.get(c.C_UPPER_LEFT), fo.get(c.C_UPPER_LEFT)) <= 1e-8
    assert parity.rel(fg.get(c.C_LOWER_LEFT), fo.get(c.C_LOWER_LEFT)) <= 1e-8


def test_full_size_c2_identities(gpu):
    """BASELINE C2 (N=1024, 256x256 grid): size-independent properties instead of an element-wise oracle run:
    K W = I residual, K v = y residual, variance in [-1e-8, k**], mean at far-away grid points -> 0, cut <= mean."""
    N, G = 1024, 256
    X, y, _ = parity.synthetic_real(N, 1, 20240607 + 1)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    fit = gpu.real_fit(theta, X, y, 3)
    assert fit.scalars["info"] == 0
    K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    assert n1(K @ W - np.eye(N)) <= 50 * N * parity.EPS * n1(K) * n1(W)
    assert np.abs(K @ v - ys).max() <= 50 * N * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max())
    dx = 40.0 / G
    xs = -20.0 + dx * np.arange(G)
    ps = (14.112 - np.pi / (2 * dx)) + (np.pi / dx / G) * np.arange(G)
    gx, gp = np.meshgrid(xs, ps, indexing="ij")  # index = ix * G + ip (input.cpp:37-70)
    p = gpu.real_predict(fit, np.stack([gx.ravel(), gp.ravel()], 1))
    kss = theta[0] ** 2 * (1 + theta[3] ** 2)
    assert p["variance"].min() >= -1e-8 and p["variance"].max() <= kss + 1e-12
    far = (np.abs(gx.ravel() + 10) > 8)
    assert np.abs(p["prediction"][far]).max() <= 1e-9 and np.abs(p["variance"][far] - kss).max() <= 1e-9
    assert np.all(np.abs(p["cutoff"]) <= np.abs(p["prediction"]) / fit.scalars["rescale_factor"] + 1e-15)
    # analytic population == grid quadrature of the uncut mean (kernel.cpp:286-297)
    quad = p["prediction"].sum() * dx * (np.pi / dx / G) / fit.scalars["rescale_factor"]
    assert abs(quad - fit.scalars["population"]) <= 1e-6 * abs(fit.scalars["population"])


def test_device_pointer_io_matches_host_io(gpu):
    """GPLE_IO_DEVICE: device-resident inputs/outputs (torch tensors as plain device memory) give the same bits."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    X, y, Xs = parity.synthetic_real(200, 500, 31)
    theta = np.array([1.0, 0.7086, 0.7056, 1e-2])
    fit = gpu.real_fit(theta, X, y, 3)
    ph = gpu.real_predict(fit, Xs)
    dXs = torch.from_numpy(Xs).cuda()
    out = torch.empty(3, len(Xs), dtype=torch.float64, device="cuda")
    dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
    ps = c.PredictScalars()
    st = gpu.lib.gple_real_predict(gpu.ctx, fit.handle, dp(dXs), len(Xs), c.IO_DEVICE, None, dp(out[0]), dp(out[1]), dp(out[2]), C.byref(ps))
    assert st == 0
    gpu.synchronize()
    o = out.cpu().numpy()
    assert np.array_equal(o[0], ph["prediction"]) and np.array_equal(o[1], ph["variance"]) and np.array_equal(o[2], ph["cutoff"])


def test_loose_function_real_matches_fixture_and_oracle(gpu, oracle):
    """opt.cpp:441-482: objective = LOOCV error + validation error, gradient = sum of both derivative sets."""
    g = load_golden("real_a")
    val, grad = gpu.loose_function(g["theta"], g["X"], g["y"].astype(complex), g["Xv"], g["tv"].astype(complex))
    tol = parity.cond_tol(g)
    assert abs(val - (g["error"] + g["v_error"])) <= tol * abs(val)
    ref = g["error_derivative"] + g["v_error_derivative"]
    assert np.abs(grad - ref).max() <= 10 * tol * np.abs(ref).max()
    # a larger, worse conditioned case against the oracle (reference's initial parameters, 5 N validation points)
    X, y, _ = parity.synthetic_real(200, 1, 41)
    rng = np.random.default_rng(42)
    Xe = X[rng.integers(0, len(X), 5 * len(X))] + rng.normal(0, [0.7, 0.7], size=(5 * len(X), 2))
    ye = np.exp(-0.5 * (((Xe[:, 0] + 10.0) / 0.7086) ** 2 + ((Xe[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    vg, gg = gpu.loose_function(theta, X, y.astype(complex), Xe, ye.astype(complex))
    vo, go = oracle.loose_function(theta, X, y.astype(complex), Xe, ye.astype(complex))
    assert abs(vg - vo) <= 1e-6 * abs(vo)
    assert np.abs(gg - go).max() <= 1e-5 * np.abs(go).max()
    # value-only call (grad == NULL) takes the no-derivative path
    v0, _ = gpu.loose_function(theta, X, y.astype(complex), Xe, ye.astype(complex), want_grad=False)
    assert abs(v0 - vg) <= 1e-12 * abs(vg)


def test_loose_function_complex(gpu, oracle):
    g = load_golden("complex_a")
    val, _ = gpu.loose_function(g["theta"], g["X"], g["y"], g["Xv"], g["tv"], want_grad=False)
    assert abs(val - (g["error"] + g["v_error"])) <= 1e-9 * abs(val)
    val2, grad = gpu.loose_function(g["theta"], g["X"], g["y"], g["Xv"], g["tv"])
    ref = g["error_derivative"] + g["v_error_derivative"]
    assert abs(val2 - val) <= 1e-12 * abs(val)
    assert np.abs(grad - ref).max() <= 1e-8 * np.abs(ref).max()
    # larger case against the oracle
    X, yr, _ = parity.synthetic_real(120, 1, 71)
    y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    rng = np.random.default_rng(72)
    Xe = X[rng.integers(0, len(X), 3 * len(X))] + rng.normal(0, [0.5, 0.5], size=(3 * len(X), 2))
    re = np.exp(-0.5 * (((Xe[:, 0] + 10.0) / 0.7086) ** 2 + ((Xe[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    ye = 0.5 * re * np.exp(0.5j * (Xe[:, 0] + 10.0))
    theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05]
    vg, gg = gpu.loose_function(theta, X, y, Xe, ye)
    vo, go = oracle.loose_function(theta, X, y, Xe, ye)
    assert abs(vg - vo) <= 1e-7 * abs(vo)
    assert np.abs(gg - go).max() <= 1e-6 * np.abs(go).max()


def test_nlml_value_gradient_and_prediction(gpu, oracle):
    """test/gpr.cpp:499-532 (NLML + trace gradient, incl. the reference's half-gradient on the two kernel weights) and
    :654-706 (mean-only predict without the noise kernel) against the oracle and a numpy evaluation."""
    X, y, Xs = parity.synthetic_real(150, 300, 61)
    x = np.array([0.1, 1.2, 1.0 / 0.8, 1.0 / 0.7])
    vg, gg = gpu.nlml(x, X, y)
    vo, go = oracle.nlml(x, X, y)
    assert abs(vg - vo) <= 1e-9 * abs(vo)
    assert np.abs(gg - go).max() <= 1e-7 * np.abs(go).max()
    v0, _ = gpu.nlml(x, X, y, want_grad=False)
    assert v0 == vg
    d0 = x[2] * (X[:, None, 0] - X[None, :, 0])
    d1 = x[3] * (X[:, None, 1] - X[None, :, 1])
    Kn = x[1] ** 2 * np.exp(-0.5 * (d0 ** 2 + d1 ** 2)) + x[0] ** 2 * np.eye(len(X))
    ref = 0.5 * y @ np.linalg.solve(Kn, y) + np.log(np.diag(np.linalg.cholesky(Kn))).sum()
    assert abs(vg - ref) <= 1e-9 * abs(ref)
    mg, mo = gpu.nlml_predict(x, X, y, Xs), oracle.nlml_predict(x, X, y, Xs)
    assert np.abs(mg - mo).max() <= 1e-9 * np.abs(mo).max()


@pytest.mark.parametrize("N", [1024, 4096])
def test_nlml_at_baseline_sizes(gpu, oracle, N):
    """negative_log_marginal_likelihood + predict_phase (test/gpr.cpp:499-532, 654-706) at the N of BASELINE configs[1] / configs[3] (the
    reference runs them at N = 200): the value against numpy's Cholesky, the gradient against central differences of the value (ARD weights:
    true derivatives; the two kernel weights: HALF of it — the reference pushes w K instead of 2 w K, test/gpr.cpp:425,432), the mean-only
    prediction on a 256 x 256 grid against the oracle on 64 of its rows and against numpy on 8."""
    X, y, _ = parity.synthetic_real(N, 4, 4400 + N)
    x = np.array([0.05, 1.3, 1.0 / 0.7086, 1.0 / 0.7056])  # (w_d, w_g, a_x, a_p): noise 0.05, ARD weights = inverse lengths (test/gpr.cpp:167-173)

    def gram(xx, A, B):
        d0, d1 = xx[2] * (A[:, None, 0] - B[None, :, 0]), xx[3] * (A[:, None, 1] - B[None, :, 1])
        return xx[1] ** 2 * np.exp(-0.5 * (d0 ** 2 + d1 ** 2))

    def f(xx):
        Kn = gram(xx, X, X) + xx[0] ** 2 * np.eye(N)
        L = np.linalg.cholesky(Kn)
        z = np.linalg.solve(L, y)
        return 0.5 * z @ z + np.log(np.diag(L)).sum()

    val, grad = gpu.nlml(x, X, y)
    ref = f(x)
    cond = np.linalg.cond(gram(x, X, X) + x[0] ** 2 * np.eye(N))
    assert abs(val - ref) <= 50 * cond * parity.EPS * abs(ref) + 1e-9 * abs(ref), (val, ref, cond)
    assert gpu.nlml(x, X, y, want_grad=False)[0] == val  # value alone: the same bits
    for ip in range(4):
        h = 1e-5 * x[ip]
        xp, xm = x.copy(), x.copy()
        xp[ip] += h
        xm[ip] -= h
        fd = (f(xp) - f(xm)) / (2 * h)
        expect = fd / 2 if ip < 2 else fd
        assert abs(grad[ip] - expect) <= 2e-5 * max(1.0, abs(expect)), (ip, grad[ip], expect)
    G = 256
    gx, gp = np.meshgrid(np.linspace(-13.0, -7.0, G), np.linspace(11.0, 17.0, G), indexing="ij")
    grid = np.ascontiguousarray(np.stack([gx.ravel(), gp.ravel()], axis=1))
    mean = gpu.nlml_predict(x, X, y, grid)
    assert mean.shape == (G * G,) and np.all(np.isfinite(mean))
    rows = np.arange(0, G * G, G * G // 64)[:64]
    mo = oracle.nlml_predict(x, X, y, grid[rows])
    scale = np.abs(mean).max()
    assert np.abs(mean[rows] - mo).max() <= 50 * cond * parity.EPS * scale + 1e-9 * scale, np.abs(mean[rows] - mo).max() / scale
    b = np.linalg.solve(gram(x, X, X) + x[0] ** 2 * np.eye(N), y)
    assert np.abs(mean[rows[:8]] - gram(x, grid[rows[:8]], X) @ b).max() <= 50 * cond * parity.EPS * scale + 1e-9 * scale


def test_nlml_cross_term_ard_against_oracle(gpu, oracle):
    """gple_nlml_cross / gple_nlml_cross_predict (default build of test/gpr.cpp: lower-triangular ARD weight matrix, :313-321,
    436-452) against the oracle, and c = 0 against the diagonal-ARD entry points"""
    X, y, Xs = parity.synthetic_real(150, 300, 61)
    x = np.array([0.1, 1.2, 1.0 / 0.8, 0.35, 1.0 / 0.7])
    vg, gg = gpu.nlml(x, X, y)
    vo, go = oracle.nlml(x, X, y)
    assert abs(vg - vo) <= 1e-9 * abs(vo)
    assert gg.shape == (5,) and np.abs(gg - go).max() <= 1e-7 * np.abs(go).max()
    assert gpu.nlml(x, X, y, want_grad=False)[0] == vg
    mg, mo = gpu.nlml_predict(x, X, y, Xs), oracle.nlml_predict(x, X, y, Xs)
    assert np.abs(mg - mo).max() <= 1e-9 * np.abs(mo).max()
    x4, x5 = np.array([0.1, 1.2, 1.25, 1.0 / 0.7]), np.array([0.1, 1.2, 1.25, 0.0, 1.0 / 0.7])
    v4, g4 = gpu.nlml(x4, X, y)
    v5, g5 = gpu.nlml(x5, X, y)
    assert v4 == v5 and np.array_equal(g4, g5[[0, 1, 2, 4]])
    assert np.array_equal(gpu.nlml_predict(x4, X, y, Xs), gpu.nlml_predict(x5, X, y, Xs))


@pytest.mark.parametrize("N", [2100, 4096])
def test_large_fit_identities(gpu, N):
    """Sizes that take the 128-tile MFMA GEMMs and an uneven merge tree (Np = 2304 -> 36 diagonal blocks; 4096 -> the
    north-star size): K W = I and K v = y residuals, LOOCV error from the getters, variance range on scattered points."""
    X, y, Xs = parity.synthetic_real(N, 3000, 20240607 + N)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    fit = gpu.real_fit(theta, X, y, 3)
    assert fit.scalars["info"] == 0
    K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    assert n1(K @ W - np.eye(N)) <= 50 * N * parity.EPS * n1(K) * n1(W)
    assert np.abs(K @ v - ys).max() <= 50 * N * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max())
    assert np.abs(W - W.T).max() <= 1e-9 * np.abs(W).max()
    assert abs(((v / np.diag(W)) ** 2).sum() - fit.scalars["error"]) <= 1e-9 * fit.scalars["error"]
    assert np.abs(np.diag(W) - fit.get(c.R_INVERSE_DIAG)).max() <= 1e-9 * np.abs(np.diag(W)).max()
    p = gpu.real_predict(fit, Xs)
    kss = theta[0] ** 2 * (1 + theta[3] ** 2)
    assert p["variance"].min() >= -1e-7 and p["variance"].max() <= kss + 1e-12
    # mean and variance against numpy in the reference's form k W k^T on a few rows
    Ks = gpu.real_gram(theta, Xs[:64], X, False)
    assert np.abs(Ks @ v - p["prediction"][:64]).max() <= 1e-8 * np.abs(p["prediction"]).max()
    assert np.abs((kss - np.einsum("ij,jk,ik->i", Ks, W, Ks)) - p["variance"][:64]).max() <= 1e-7


def test_large_complex_fit_identities(gpu):
    """complex element with n = 2 Np = 2560 (uneven merge tree in the embedded factorisation)"""
    N = 1100
    X, yr, Xs = parity.synthetic_real(N, 500, 99)
    y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    theta = [1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2]
    fit = gpu.complex_fit(theta, X, y, 3)
    assert fit.scalars["info"] == 0
    K, Kt = fit.get(c.C_KERNEL), fit.get(c.C_PSEUDO)
    P, Q, v, ys = fit.get(c.C_UPPER_LEFT), fit.get(c.C_LOWER_LEFT), fit.get(c.C_INVLBL), fit.get(c.C_LABEL)
    # augmented system: [K Kt; Kt* K] [P; Q] = [I; 0]  and  K v + Kt conj(v) = y
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    scale = 50 * 2 * N * parity.EPS * (n1(K) + n1(Kt)) * (n1(P) + n1(Q))
    assert n1(K @ P + Kt @ Q - np.eye(N)) <= scale
    assert n1(Kt.conj() @ P + K @ Q) <= scale
    assert np.abs(K @ v + Kt @ v.conj() - ys).max() <= 1e-7 * np.abs(ys).max()
    p = gpu.complex_predict(fit, Xs)
    assert np.all(np.isfinite(p["variance"])) and p["variance"].min() >= -1e-7


def test_concurrent_predicts_on_one_fit(gpu):
    """evolve.cpp:392-420 calls the predictors from TBB worker threads: predict must be thread-safe on a shared fit"""
    import threading
    X, y, Xs = parity.synthetic_real(300, 2000, 81)
    fit = gpu.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 1)
    # reference: the same four slices one after the other (the split of the mean's k-sum depends on the row count of a call,
    # so a slice is compared with the same slice, bit for bit)
    ref = [gpu.real_predict(fit, Xs[i * 500:(i + 1) * 500]) for i in range(4)]
    whole = gpu.real_predict(fit, Xs)
    out, errs = {}, []

    def work(i):
        try:
            sl = slice(i * 500, (i + 1) * 500)
            for _ in range(5):
                out[i] = gpu.real_predict(fit, Xs[sl])
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for i in range(4):
        sl = slice(i * 500, (i + 1) * 500)
        assert np.array_equal(out[i]["prediction"], ref[i]["prediction"]) and np.array_equal(out[i]["variance"], ref[i]["variance"])
        assert np.abs(out[i]["prediction"] - whole["prediction"][sl]).max() <= 1e-13 * np.abs(whole["prediction"]).max()
        assert np.abs(out[i]["variance"] - whole["variance"][sl]).max() <= 1e-12


def test_complex_edge_cases(gpu, oracle):
    """single training point, empty test set, prediction at the training points (delta kernel in both typed blocks)"""
    X, yr, Xs = parity.synthetic_real(30, 7, 91)
    y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.2]
    fg, fo = gpu.complex_fit(theta, X, y, 3), oracle.complex_fit(theta, X, y, 3)
    assert gpu.complex_predict(fg, np.zeros((0, 2)))["variance"].size == 0
    pg, po = gpu.complex_predict(fg, X), oracle.complex_predict(fo, X)
    assert np.abs(pg["variance"] - po["variance"]).max() <= 1e-10
    assert np.abs(pg["prediction"] - po["prediction"]).max() <= 1e-10 * np.abs(po["prediction"]).max()
    f1g, f1o = gpu.complex_fit(theta, X[:1], y[:1], 3), oracle.complex_fit(theta, X[:1], y[:1], 3)
    for k in ("error", "purity", "magnitude"):
        assert abs(f1g.scalars[k] - f1o.scalars[k]) <= 1e-12 * abs(f1o.scalars[k])
    p1g, p1o = gpu.complex_predict(f1g, Xs), oracle.complex_predict(f1o, Xs)
    assert np.abs(p1g["cutoff"] - p1o["cutoff"]).max() <= 1e-13 and np.abs(p1g["variance"] - p1o["variance"]).max() <= 1e-13


def test_deferred_scalars_match_immediate(gpu):
    """scalars == NULL at create: nothing synchronises until a getter asks; values equal the immediate path bit for bit,
    and a predict enqueued before the scalars are fetched sees the same fit."""
    X, y, Xs = parity.synthetic_real(700, 900, 31)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    now = gpu.real_fit(theta, X, y, 7)
    later = gpu.real_fit(theta, X, y, 7, defer_scalars=True)
    assert later._scalars is None
    p_later = gpu.real_predict(later, Xs)
    assert later._scalars is None  # the predict did not need them
    p_now = gpu.real_predict(now, Xs)
    for k, v in now.scalars.items():
        assert np.array_equal(np.asarray(v), np.asarray(later.scalars[k]), equal_nan=True), k
    for k in ("prediction", "variance", "cutoff"):
        assert np.array_equal(p_now[k], p_later[k])
    yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0))
    thc = [1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2]
    cn, cl = gpu.complex_fit(thc, X[:300], yc[:300], 7), gpu.complex_fit(thc, X[:300], yc[:300], 7, defer_scalars=True)
    for k, v in cn.scalars.items():
        assert np.array_equal(np.asarray(v), np.asarray(cl.scalars[k]), equal_nan=True), k


def test_small_m_and_streaming_predict_paths_agree(gpu, monkeypatch):
    """The few-rows path (Z = T K*^T by the triangular GEMM + column norms) against the streaming MFMA contraction on the
    same fit and points, real and complex, incl. a row count that is not a tile multiple."""
    X, y, Xs = parity.synthetic_real(900, 1111, 77)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    fit = gpu.real_fit(theta, X, y, 3)
    out = {}
    for force in ("0", "1"):
        monkeypatch.setenv("GPLE_PREDICT_SMALL_M", force)
        out[force] = gpu.real_predict(fit, Xs)
    assert np.array_equal(out["0"]["prediction"], out["1"]["prediction"])  # the mean does not depend on the path
    assert np.abs(out["0"]["variance"] - out["1"]["variance"]).max() <= 1e-11 * (theta[0] ** 2)
    yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0))
    thc = [1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2]
    cfit = gpu.complex_fit(thc, X[:400], yc[:400], 3)
    for force in ("0", "1"):
        monkeypatch.setenv("GPLE_PREDICT_SMALL_M", force)
        out[force] = gpu.complex_predict(cfit, Xs[:333])
    assert np.array_equal(out["0"]["prediction"], out["1"]["prediction"])
    assert np.abs(out["0"]["variance"] - out["1"]["variance"]).max() <= 1e-11 * 4.0


def test_ctx_trim_releases_idle_buffers_only(gpu):
    """the grow-only pool can be emptied between phases; live fits keep their buffers and keep working"""
    X, y, Xs = parity.synthetic_real(200, 4000, 55)
    fit = gpu.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 3)
    before = gpu.real_predict(fit, Xs)
    freed = gpu.trim()
    assert freed > 4000 * 256 * 8  # at least the K* scratch of that predict
    assert gpu.trim() == 0          # nothing idle is left
    after = gpu.real_predict(fit, Xs)
    assert np.array_equal(before["prediction"], after["prediction"]) and np.array_equal(before["variance"], after["variance"])
    assert fit.scalars["info"] == 0


def test_resident_objective_equals_loose_function(gpu):
    """gple_objective_*: the data uploaded once; every evaluation equals gple_loose_function on host arrays bit for bit"""
    X, y, Xs = parity.synthetic_real(333, 120, 17)
    ye = (0.01 + 0.0 * Xs[:, 0]).astype(complex)
    yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0))
    yec = 0.5 * ye * np.exp(0.5j * (Xs[:, 0] + 10.0))
    for theta, yy, yee in (([1.0, 0.7086, 0.7056, 1e-2], y.astype(complex), ye), ([1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2], yc, yec)):
        obj = gpu.objective(X, yy, Xs, yee)
        for scale in (1.0, 0.9, 1.1):
            th = [theta[0]] + [t * scale for t in theta[1:-1]] + [theta[-1]]
            for g in (True, False):
                v1, g1 = obj(th, want_grad=g)
                v2, g2 = gpu.loose_function(th, X, yy, Xs, yee, want_grad=g)
                assert v1 == v2 and (g1 is None) == (g2 is None) and (g1 is None or np.array_equal(g1, g2))
        obj.release()
    empty = gpu.objective(X, y.astype(complex), np.zeros((0, 2)), np.zeros(0, complex))  # no extra set
    v, _ = empty([1.0, 0.7086, 0.7056, 1e-2], want_grad=False)
    assert v == gpu.loose_function([1.0, 0.7086, 0.7056, 1e-2], X, y.astype(complex), np.zeros((0, 2)), np.zeros(0, complex), want_grad=False)[0]


def test_few_points_predict_path(gpu, oracle, monkeypatch):
    """<= 16 typed rows take the one-point path (triangular mat-vec with K* generated on the fly): against the tiled path on the
    same points and against the oracle, real and complex, 1 .. 16 points, incl. a training point (delta kernel) and N not a
    multiple of the row block"""
    X, yr, Xs = parity.synthetic_real(333, 16, 404)
    Xs[3] = X[17]
    yc = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    th, thc = [1.0, 0.7086, 0.7056, 1e-2], [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05]
    fr, fc = gpu.real_fit(th, X, yr, 0), gpu.complex_fit(thc, X, yc, 0)
    fro, fco = oracle.real_fit(th, X, yr, 0), oracle.complex_fit(thc, X, yc, 0)
    for m in (1, 2, 5, 8, 16):
        for fit, fo, pred, po, lim in ((fr, fro, gpu.real_predict, oracle.real_predict, 16), (fc, fco, gpu.complex_predict, oracle.complex_predict, 8)):
            monkeypatch.setenv("GPLE_PREDICT_FEW", "1")
            few = pred(fit, Xs[:m])
            ref = po(fo, Xs[:m])
            scale = np.abs(ref["prediction"]).max()
            assert np.abs(few["prediction"] - ref["prediction"]).max() <= 1e-9 * scale
            assert np.abs(few["variance"] - ref["variance"]).max() <= 1e-8
            assert np.abs(few["cutoff"] - ref["cutoff"]).max() <= 1e-8 * scale / fo.scalars["rescale_factor"]
    # labels still work on few points (error through the same finish kernel)
    lab = yr[:4]
    e_few = gpu.real_predict(fr, X[:4], labels=lab)["error"]
    e_ref = oracle.real_predict(fro, X[:4], labels=lab)["error"]
    assert abs(e_few - e_ref) <= 1e-7 * max(abs(e_ref), 1e-12) + 1e-12


def test_one_point_predicts_from_many_threads(gpu):
    """§8(b) threading: one-point predicts on a shared fit from several host threads (evolve.cpp:392-420 calls the
    DistributionFunction from TBB workers).  Requests that pile up behind a predict in flight are served together by the next
    one (gple_capi.hip, predict_point_combined); every caller gets exactly what a lone call returns."""
    import threading
    X, y, Xs = parity.synthetic_real(300, 256, 77)
    yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0))
    for cplx in (False, True):
        fit = gpu.complex_fit([1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05], X, yc, 0) if cplx else gpu.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 0)
        pred = gpu.complex_predict if cplx else gpu.real_predict
        lone = [pred(fit, Xs[i:i + 1]) for i in range(len(Xs))]
        got = [None] * len(Xs)

        def work(k, nt):
            for i in range(k, len(Xs), nt):
                got[i] = pred(fit, Xs[i:i + 1])

        nt = 12
        ths = [threading.Thread(target=work, args=(k, nt)) for k in range(nt)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for a, b in zip(lone, got):
            for key in ("prediction", "variance", "cutoff"):
                assert np.array_equal(a[key], b[key]), key
        # the batched entry agrees as well
        whole = pred(fit, Xs)
        scale = np.abs(whole["prediction"]).max()
        assert np.abs(np.concatenate([g["prediction"] for g in got]) - whole["prediction"]).max() <= 1e-12 * scale
        fit.release()


def test_far_row_pruning_is_bit_identical(gpu):
    """A real-kernel grid predict skips the variance contraction for 128-row blocks of test points whose K* rows cannot move
    k(x*,x*) - k* K^-1 k*^T by half an ulp (|k*|^2 < 2^-56 sf^2 sn^2 k(x*,x*), lambda_min(K) >= sf^2 sn^2): outputs with and
    without GPLE_PREDICT_FULL agree bit for bit on the reference's grid (input.cpp:37-70: the whole phase-space box, most of it
    far from the density), and most blocks are in fact skipped there; a grid inside the data cloud skips nothing."""
    from tests.test_gpu_configs import config_inputs, THETA_R
    N, G = 1024, 256
    X, y, grid, _ = config_inputs(N, G, 1)
    fit = gpu.real_fit(THETA_R, X, y, 0)
    gpu.prune_stats(reset=True)
    full = gpu.real_predict(fit, grid, flags=c.PREDICT_FULL)
    assert gpu.prune_stats(reset=True) == (0, 0)  # the full contraction does not consult the test at all
    pruned = gpu.real_predict(fit, grid)
    live, seen = gpu.prune_stats(reset=True)
    assert seen == len(grid) // 128 and 0 < live < 0.5 * seen, (live, seen)
    for k in ("prediction", "variance", "cutoff"):
        assert np.array_equal(full[k], pruned[k]), k
    # the skipped rows really are prior-variance rows, the contracted ones are not all
    prior = THETA_R[0] ** 2 * (1 + THETA_R[3] ** 2)
    assert (pruned["variance"] == prior).mean() > 0.5 and (pruned["variance"] < 0.9 * prior).any()
    # test points inside the cloud: nothing to skip
    inside = X[:512] + 0.01
    a, b = gpu.real_predict(fit, inside, flags=c.PREDICT_FULL), gpu.real_predict(fit, inside)
    live, seen = gpu.prune_stats(reset=True)
    assert live == seen or seen == 0  # (few-rows path: not pruned, not counted)
    for k in ("prediction", "variance", "cutoff"):
        assert np.array_equal(a[k], b[k]), k
    # a noise-free kernel has no ridge to bound lambda_min with: threshold 0, nothing skipped
    fit0 = gpu.real_fit([1.0, 0.7086, 0.7056, 0.0], X[:256], y[:256], 0)
    gpu.real_predict(fit0, grid)
    live, seen = gpu.prune_stats(reset=True)
    assert live == seen
    fit0.release()
    fit.release()


def test_far_row_pruning_complex_is_bit_identical(gpu):
    """the same for the complex GP (two typed rows per point in the [Re; Im] embedding, lambda_min >= s^2 sn^2 / 2)"""
    from tests.test_gpu_configs import config_inputs, THETA_C
    X, y, grid, _ = config_inputs(512, 256, 3, cplx=True)
    fit = gpu.complex_fit(THETA_C, X, y, 0)
    full = gpu.complex_predict(fit, grid, flags=c.PREDICT_FULL)
    gpu.prune_stats(reset=True)
    pruned = gpu.complex_predict(fit, grid)
    live, seen = gpu.prune_stats(reset=True)
    assert seen == 2 * len(grid) // 128 and 0 < live < 0.6 * seen, (live, seen)
    for k in ("prediction", "variance", "cutoff"):
        assert np.array_equal(full[k], pruned[k]), k
    fit.release()


def test_far_row_pruning_edge_cases(gpu):
    """the compacted list longer than one K* chunk (all rows live, 135 000 of them at N = 4096: two chunks of the list), and
    the empty list (every row dead): both bit-identical to the full contraction"""
    from tests.test_gpu_configs import config_inputs, THETA_R
    N = 4096
    X, y, _, _ = config_inputs(N, 8, 1)
    fit = gpu.real_fit(THETA_R, X, y, 0)
    rng = np.random.default_rng(11)
    near = X[rng.integers(0, N, 135000)] + rng.normal(0, 0.2, (135000, 2))
    gpu.prune_stats(reset=True)
    a = gpu.real_predict(fit, near, want=("variance", "cutoff"))
    live, seen = gpu.prune_stats(reset=True)
    assert live == seen == (135000 + 127) // 128
    b = gpu.real_predict(fit, near, flags=c.PREDICT_FULL, want=("variance", "cutoff"))
    assert np.array_equal(a["variance"], b["variance"]) and np.array_equal(a["cutoff"], b["cutoff"])
    far = np.stack([np.linspace(30.0, 60.0, 20000), np.linspace(-40.0, -10.0, 20000)], 1)
    a = gpu.real_predict(fit, far)
    live, seen = gpu.prune_stats(reset=True)
    assert live == 0 and seen == (20000 + 127) // 128
    b = gpu.real_predict(fit, far, flags=c.PREDICT_FULL)
    for k in ("prediction", "variance", "cutoff"):
        assert np.array_equal(a[k], b[k]), k
    assert np.all(a["variance"] == THETA_R[0] ** 2 * (1 + THETA_R[3] ** 2))
    fit.release()


@pytest.mark.parametrize("cplx", [False, True])
def test_predict_skips_what_nobody_consumes(gpu, cplx):
    """A predict that is only asked for Error (+ ErrorDerivatives) — the objective of opt.cpp:441-482 — skips the variance contraction where it
    cannot matter: Error uses the uncut mean (kernel.cpp:522: no contraction at all), ErrorDerivatives use the cut mean (:527), and a point with
    |mu|^2 >= 4 k(x*,x*) has cut-off factor 1 whatever its variance (kernel.h:301-332).  Same scalars, bit for bit, as the call that also asks
    for the variance (which contracts every row); sizes on the streaming path (not the few-rows GEMM path)."""
    from tests.test_gpu_configs import config_inputs, extra_set, THETA_C, THETA_R
    N = 2048
    X, y, _, _ = config_inputs(N, 8, 31, cplx=cplx)
    Xe, ye = extra_set(X, 32, cplx)           # 5N points around the packet: about half of them with |mu|^2 >= 4 k(x*,x*)
    theta = list(THETA_C if cplx else THETA_R)
    theta[-1] = 0.05
    fit = (gpu.complex_fit if cplx else gpu.real_fit)(theta, X, y, c.CALC_ERROR | c.CALC_DERIVATIVE)
    pred = gpu.complex_predict if cplx else gpu.real_predict
    lab = ye if cplx else ye.real
    for flags in (0, c.CALC_DERIVATIVE):
        lean = pred(fit, Xe, flags=flags | c.PREDICT_FULL, labels=lab, want=())
        full = pred(fit, Xe, flags=flags | c.PREDICT_FULL, labels=lab, want=("variance", "cutoff"))
        assert lean["error"] == full["error"] and np.isfinite(lean["error"])
        if flags:
            assert np.array_equal(lean["error_derivative"], full["error_derivative"]) and np.all(np.isfinite(lean["error_derivative"]))
            # the skip has something to skip and something to keep: both classes of points are present
            mu2 = np.abs(pred(fit, Xe, want=("prediction",))["prediction"]) ** 2
            kss = theta[0] ** 2 * ((theta[1] ** 2 + theta[4] ** 2 + theta[7] ** 2) if cplx else (1 + theta[3] ** 2))
            assert 0.15 < (mu2 >= 4 * kss).mean() < 0.85
    fit.release()


@pytest.mark.parametrize("N", [200, 256, 400])
def test_short_factor_kernel_has_the_bits_of_the_general_one(gpu, N):
    """n <= 512 with few row blocks (C1) runs on rownorm3_kernel (64-row workgroups, three slabs in flight); the same rows inside a call large
    enough for the general kernel (rownorm2_kernel<2,8>) must come out bit for bit — the per-row arithmetic is the same by construction"""
    from tests.test_gpu_configs import config_inputs, THETA_R
    X, y, grid, _ = config_inputs(N, 128, 5)
    fit = gpu.real_fit(THETA_R, X, y, 0)
    rng = np.random.default_rng(N)
    pts = X[rng.integers(0, N, 16384)] + rng.normal(0, 0.3, (16384, 2))
    a = gpu.real_predict(fit, pts, flags=c.PREDICT_FULL)
    more = np.concatenate([pts, X[rng.integers(0, N, 4 * 16384 + 300)] + rng.normal(0, 0.5, (4 * 16384 + 300, 2))])
    b = gpu.real_predict(fit, more, flags=c.PREDICT_FULL)
    for k in ("prediction", "variance", "cutoff"):
        assert np.array_equal(a[k], b[k][:16384]), k
    assert a["variance"].min() >= -1e-9 and (a["variance"] < 0.5).any()
    fit.release()


@pytest.mark.parametrize("N,M,cplx", [(1024, 40000, False), (700, 9000, True), (2300, 9000, False), (4096, 6100, False), (4096, 40000, False), (1500, 20000, True),
                                          (100, 30000, True), (300, 70000, False), (600, 12000, False),  # one, two and three N-tiles
                                          (4096, 300000, False)])  # three K* chunks in the full predict, a live-row list that crosses a chunk boundary in the pruned one
def test_rownorm_variants_agree_bit_for_bit(gpu, N, M, cplx):
    """rownormp_kernel (the k-steps of a unit as one pipeline: operands of the next step requested right behind the barrier, the DMA issues between
    the MFMAs of the last group, slabs requested across tile boundaries) against rownorm2_kernel (barrier-to-barrier k-steps): per accumulator
    the same MFMAs in the same order and the same sums behind them, so every output must agree bit for bit — full contraction (static grid,
    every split of the N-tiles) and the pruned one (work queue), real and complex (typed rows), 2 x 8 and 4 x 4 blocking"""
    import ctypes
    from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C
    X, y, grid, _ = config_inputs(N, 64, 5, cplx=cplx) if cplx else config_inputs(N, 64, 5)
    rng = np.random.default_rng(N + M)
    near = X[rng.integers(0, N, M // 2)] + rng.normal(0, 0.3, (M // 2, 2))
    far = X[rng.integers(0, N, M - M // 2)] + rng.normal(0, 6.0, (M - M // 2, 2))  # many of these are pruned by the default predict
    pts = np.concatenate([near, far])[rng.permutation(M)]
    fit = gpu.complex_fit(THETA_C, X, y, 0) if cplx else gpu.real_fit(THETA_R, X, y, 0)
    pred = gpu.complex_predict if cplx else gpu.real_predict
    knob = gpu.lib.gple_debug_predict_knobs
    knob.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    out = {}
    try:
        for v in (0, 1):
            assert knob(gpu.ctx, v, -1) == 0
            out[v] = (pred(fit, pts, flags=c.PREDICT_FULL), pred(fit, pts))
    finally:
        knob(gpu.ctx, 2, -1)
    for leg in (0, 1):
        for k in ("prediction", "variance", "cutoff"):
            assert np.array_equal(out[0][leg][k], out[1][leg][k]), (leg, k)
    assert np.isfinite(out[1][0]["variance"]).all() and (out[1][0]["variance"] < 0.5 * out[1][0]["variance"].max()).any()
    fit.release()


@pytest.mark.parametrize("N,M", [(256, 16384), (200, 16384), (256, 1300), (37, 300), (129, 50000)])
def test_fused_small_predict_has_the_bits_of_the_unfused_path(gpu, oracle, N, M):
    """Real fits with N <= 256 (one N-tile of T; C1, the size the reference itself runs) predict in ONE launch (predict_fused256_kernel: K* generated
    inside the contraction, the mean chained over the slab in LDS, both sums in the kernel).  Against the separate kernels (K* generation,
    rownorm3_kernel / rownorm2_kernel<2,8>, two sum kernels) every output must agree bit for bit: same K* expression, same k-ranges of the
    partial means added in the same order, the same fragments and sums in the contraction.  Full and default (pruned) request; against the
    oracle as well, so that the pair cannot be wrong together."""
    import ctypes
    from tests.test_gpu_configs import config_inputs, THETA_R
    X, y, grid, _ = config_inputs(N, 64, 5)
    rng = np.random.default_rng(N + M)
    pts = np.concatenate([X[rng.integers(0, N, M // 2)] + rng.normal(0, 0.3, (M // 2, 2)), X[rng.integers(0, N, M - M // 2)] + rng.normal(0, 5.0, (M - M // 2, 2))])
    pts[:7] = X[:7]  # coincident points: the delta kernel
    fit = gpu.real_fit(THETA_R, X, y, 0)
    knob = gpu.lib.gple_debug_predict_knobs
    knob.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    last = gpu.lib.gple_debug_last_contraction_kernel
    last.argtypes, last.restype = [ctypes.c_void_p], ctypes.c_char_p
    out, ran = {}, {}
    try:
        for v in (0, 1):
            assert knob(gpu.ctx, -1, v) == 0
            out[v] = (gpu.real_predict(fit, pts, flags=c.PREDICT_FULL), gpu.real_predict(fit, pts))
            ran[v] = last(gpu.ctx).decode()
    finally:
        knob(gpu.ctx, -1, 2)
    assert ran[1] == "predict_fused256_kernel" and ran[0] != ran[1], ran
    streaming = ran[0].startswith("rownorm")  # few rows go through Z = T K*^T and column sums when unfused: another order of the same sums
    for leg in (0, 1):
        assert np.array_equal(out[0][leg]["prediction"], out[1][leg]["prediction"]), leg
        for k in ("variance", "cutoff"):
            if streaming:
                assert np.array_equal(out[0][leg][k], out[1][leg][k]), (leg, k, np.abs(out[0][leg][k] - out[1][leg][k]).max())
            else:
                assert np.abs(out[0][leg][k] - out[1][leg][k]).max() < 1e-11, (leg, k)
    assert streaming == (M >= 8192)
    fo = oracle.real_fit(THETA_R, X, y, 0)
    po = oracle.real_predict(fo, pts[:2000])
    assert np.abs(out[1][0]["prediction"][:2000] - po["prediction"]).max() < 1e-9 * max(1.0, np.abs(po["prediction"]).max())
    assert np.abs(out[1][0]["variance"][:2000] - po["variance"]).max() < 1e-9
    fit.release()


def test_batched_derivative_launches_have_the_bits_of_the_launch_per_product_path():
    """Derivative fits of small matrices (n <= 1024) batch their independent matrix-vector products, column dots and GEMMs into a third of the launches
    (csrc/gple_capi.hip, real_fit_derivatives / complex_fit_derivatives); every item keeps its arithmetic, so every derivative member must agree bit for bit
    with the launch-per-product path (GPLE_DERIV_BATCH=0; the switch is read once per process: two child processes)"""
    import os
    import subprocess
    import sys
    import tempfile
    from tests.conftest import ROOT
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C
api = pkg.open_api(0)
out = []
for N in (100, 256, 300, 700):
    for cplx in (False, True):
        X, y, _, _ = config_inputs(N, 8, 11, cplx=cplx)
        fit = (api.complex_fit if cplx else api.real_fit)(THETA_C if cplx else THETA_R, X, y, 7)
        sc = fit.scalars
        out += [np.atleast_1d(np.asarray(sc[k], dtype=float)).ravel() for k in sorted(sc) if "derivative" in k]
        out.append(np.asarray(fit.get(c.C_INVLBL_DERIV if cplx else c.R_INVLBL_DERIV)).view(float).ravel())
        fit.release()
np.save(sys.argv[1], np.concatenate(out))
api.close()
""" % ROOT
    res = []
    with tempfile.TemporaryDirectory() as d:
        for batch in ("0", "1"):
            f = os.path.join(d, f"d{batch}.npy")
            subprocess.run([sys.executable, "-c", code, f], check=True, env=dict(os.environ, GPLE_DERIV_BATCH=batch), cwd=ROOT, timeout=600)
            res.append(np.load(f))
    assert res[0].shape == res[1].shape and len(res[0]) > 1000 and np.isfinite(res[0]).all()
    assert np.array_equal(res[0], res[1]), np.abs(res[0] - res[1]).max()


@pytest.mark.parametrize("N,G,reps", [(1024, 128, 40), (256, 128, 100), (2300, 64, 20)])
def test_repeated_predicts_agree_bit_for_bit(gpu, N, G, reps):
    """the pipelined contraction and the fused small-n predict move their slabs by LDS-DMA into buffers that other waves are reading a step earlier or later:
    a race would show as a run that differs from the first (probes/soak_contraction.py is the long form: 5 000 predicts at the BASELINE sizes, none differed)"""
    from tests.test_gpu_configs import config_inputs, THETA_R
    X, y, grid, _ = config_inputs(N, G, 5)
    fit = gpu.real_fit(THETA_R, X, y, 0)
    for flags in (c.PREDICT_FULL, 0):
        first = gpu.real_predict(fit, grid, flags=flags)
        for _ in range(reps):
            p = gpu.real_predict(fit, grid, flags=flags)
            assert all(np.array_equal(p[k], first[k]) for k in ("prediction", "variance", "cutoff"))
    fit.release()


def test_forced_128_tiles_on_odd_fork_points_stay_off_unwritten_blocks():
    """the block-row inverse runs its triangular GEMMs on sub-matrices whose origin is a fork point — a multiple of 64 only.  A
    128-tile with a triangular k-range starts at its 128-aligned diagonal tile and would take in a block above the diagonal that nobody writes
    when the origin is an odd multiple of 64.  Own process (the knobs are read once): 128-tiles forced for triangular work, one fork at 55 %
    of n = 2304 (column 1216 = 19 x 64), T and the Cholesky workspace poisoned with NaN — the fit's identities must hold."""
    import subprocess
    import sys
    from tests.conftest import ROOT
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity
api = pkg.open_api(0)
for N in (2100, 2304, 3000):
    X, y, Xs = parity.synthetic_real(N, 500, 20240607 + N)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    fit = api.real_fit(theta, X, y, 3)
    assert fit.scalars["info"] == 0 and np.isfinite(fit.scalars["error"])
    K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    assert np.all(np.isfinite(W)) and n1(K @ W - np.eye(N)) <= 50 * N * parity.EPS * n1(K) * n1(W)
    assert np.abs(K @ v - ys).max() <= 50 * N * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max())
    p = api.real_predict(fit, Xs)
    assert np.all(np.isfinite(p["variance"])) and p["variance"].min() >= -1e-7
print("ok")
''' % ROOT
    env = dict(os.environ, GPLE_POISON_T="1", GPLE_CHOL_FORKS="55", GPLE_GEMM_128_MIN_TILES_TRI="1", GPLE_GEMM_SPLITK_MAX_TILES="0")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, (res.stdout[-500:], res.stderr[-2000:])
