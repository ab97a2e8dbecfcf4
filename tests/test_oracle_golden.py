"""CPU: the C++ oracle against the independent 50-digit fixtures (tests/golden, made by oracle/gen_golden.py)
and against identities that do not depend on either restatement."""
import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity
from tests.conftest import load_golden


@pytest.mark.parametrize("name", parity.REAL_FIXTURES + parity.REAL_FIXTURES_LARGE)
def test_real_fixture(oracle, name):
    parity.check_real_case(oracle, load_golden(name), deriv=True)


@pytest.mark.parametrize("name,tol", parity.COMPLEX_FIXTURES + parity.COMPLEX_FIXTURES_LARGE)
def test_complex_fixture(oracle, name, tol):
    parity.check_complex_case(oracle, load_golden(name), deriv=True, tol=tol)


def test_loocv_identity_bruteforce(oracle):
    """Error = sum (v_i / W_ii)^2 equals the squared leave-one-out residuals computed by N refits."""
    X, y, _ = parity.synthetic_real(24, 4, 5)
    theta = [1.0, 0.9, 0.8, 0.1]
    fit = oracle.real_fit(theta, X, y, c.CALC_ERROR)
    s = fit.scalars["rescale_factor"]
    K = oracle.real_gram(theta, X, X, True)
    total = 0.0
    for i in range(len(X)):
        keep = np.arange(len(X)) != i
        mu = K[i, keep] @ np.linalg.solve(K[np.ix_(keep, keep)], y[keep] * s)
        total += (mu - y[i] * s) ** 2
    assert abs(total - fit.scalars["error"]) <= 1e-9 * total


def test_error_gradient_matches_finite_difference(oracle):
    """loose_function gradient vs central differences (the LOOCV part is a true derivative; the validation part uses
    the reference's cut/uncut mix, so use points whose cutoff factor is 1)."""
    X, y, _ = parity.synthetic_real(20, 4, 6)
    theta = np.array([1.1, 0.9, 0.8, 0.15])
    f0 = oracle.real_fit(theta, X, y, c.CALC_ERROR | c.CALC_DERIVATIVE)
    g = f0.scalars["error_derivative"]
    for ip in range(4):
        h = 1e-6 * theta[ip]
        tp, tm = theta.copy(), theta.copy()
        tp[ip] += h
        tm[ip] -= h
        fd = (oracle.real_fit(tp, X, y, c.CALC_ERROR).scalars["error"] - oracle.real_fit(tm, X, y, c.CALC_ERROR).scalars["error"]) / (2 * h)
        assert abs(fd - g[ip]) <= 1e-5 * max(1.0, abs(fd)), (ip, fd, g[ip])


def test_population_is_grid_quadrature_of_the_mean(oracle):
    """Analytic population (kernel.cpp:286-297) == integral of the (uncut) predicted mean over phase space."""
    X, y, _ = parity.synthetic_real(40, 4, 7)
    theta = [1.0, 0.7086, 0.7056, 0.05]
    fit = oracle.real_fit(theta, X, y, c.CALC_AVERAGE)
    G = 160
    xs, ps = np.linspace(-18, -2, G), np.linspace(6, 22, G)
    gx, gp = np.meshgrid(xs, ps, indexing="ij")
    p = oracle.real_predict(fit, np.stack([gx.ravel(), gp.ravel()], 1), want=("prediction",))
    quad = p["prediction"].sum() * (xs[1] - xs[0]) * (ps[1] - ps[0]) / fit.scalars["rescale_factor"]
    assert abs(quad - fit.scalars["population"]) <= 1e-6 * abs(fit.scalars["population"])


def test_loose_function_is_sum_of_parts(oracle):
    g = load_golden("real_a")
    val, grad = oracle.loose_function(g["theta"], g["X"], g["y"].astype(complex), g["Xv"], g["tv"].astype(complex))
    assert abs(val - (g["error"] + g["v_error"])) <= 1e-10 * abs(val)
    assert np.abs(grad - (g["error_derivative"] + g["v_error_derivative"])).max() <= 1e-8 * np.abs(grad).max()
    gc = load_golden("complex_a")
    val, grad = oracle.loose_function(gc["theta"], gc["X"], gc["y"], gc["Xv"], gc["tv"])
    assert abs(val - (gc["error"] + gc["v_error"])) <= 1e-9 * abs(val)
    assert np.abs(grad - (gc["error_derivative"] + gc["v_error_derivative"])).max() <= 1e-7 * np.abs(grad).max()


def test_nlml_gradient_structure(oracle):
    """test/gpr.cpp:499-532: value matches numpy; ARD-weight gradients are true derivatives, the two kernel-weight
    gradients are HALF the true derivative (the reference pushes w*K instead of 2*w*K, test/gpr.cpp:425,432)."""
    X, y, Xs = parity.synthetic_real(30, 10, 8)
    x = np.array([0.1, 1.2, 1.0 / 0.8, 1.0 / 0.7])
    val, grad = oracle.nlml(x, X, y)

    def f(xx):
        d0 = xx[2] * (X[:, None, 0] - X[None, :, 0])
        d1 = xx[3] * (X[:, None, 1] - X[None, :, 1])
        K = xx[1] ** 2 * np.exp(-0.5 * (d0 ** 2 + d1 ** 2)) + xx[0] ** 2 * np.eye(len(X))
        L = np.linalg.cholesky(K)
        return 0.5 * y @ np.linalg.solve(K, y) + np.log(np.diag(L)).sum()

    assert abs(val - f(x)) <= 1e-10 * abs(val)
    for ip in range(4):
        h = 1e-6 * x[ip]
        xp, xm = x.copy(), x.copy()
        xp[ip] += h
        xm[ip] -= h
        fd = (f(xp) - f(xm)) / (2 * h)
        expect = fd / 2 if ip < 2 else fd
        assert abs(grad[ip] - expect) <= 1e-5 * max(1.0, abs(expect)), (ip, grad[ip], expect)
    mean = oracle.nlml_predict(x, X, y, Xs)
    d0 = x[2] * (Xs[:, None, 0] - X[None, :, 0])
    d1 = x[3] * (Xs[:, None, 1] - X[None, :, 1])
    Ks = x[1] ** 2 * np.exp(-0.5 * (d0 ** 2 + d1 ** 2))
    K = x[1] ** 2 * np.exp(-0.5 * ((x[2] * (X[:, None, 0] - X[None, :, 0])) ** 2 + (x[3] * (X[:, None, 1] - X[None, :, 1])) ** 2)) + x[0] ** 2 * np.eye(len(X))
    assert np.abs(mean - Ks @ np.linalg.solve(K, y)).max() <= 1e-9 * np.abs(mean).max()


def test_nlml_cross_term_ard(oracle):
    """The default build of test/gpr.cpp (:99-103, 313-321, 436-452): lower-triangular ARD weight matrix W = [[a, 0], [c, b]],
    k = exp(-(x - x')^T W W^T (x - x') / 2).  Value against numpy; the three weight-matrix gradients are true derivatives (the
    log-domain diagonal convention divided out at :444), the two kernel-weight gradients half of it as in the NOCROSS build;
    c = 0 reproduces the diagonal-ARD entry point exactly.  Parity unpinned (Shogun)."""
    X, y, Xs = parity.synthetic_real(30, 10, 8)
    x = np.array([0.1, 1.2, 1.0 / 0.8, 0.35, 1.0 / 0.7])  # (w_d, w_g, a, c, b)
    val, grad = oracle.nlml(x, X, y)

    def gram(xx, A, B):
        e0, e1 = A[:, None, 0] - B[None, :, 0], A[:, None, 1] - B[None, :, 1]
        W = np.array([[xx[2], 0.0], [xx[3], xx[4]]])
        Mm = W @ W.T
        q = Mm[0, 0] * e0 ** 2 + 2 * Mm[0, 1] * e0 * e1 + Mm[1, 1] * e1 ** 2
        return xx[1] ** 2 * np.exp(-0.5 * q)

    def f(xx):
        K = gram(xx, X, X) + xx[0] ** 2 * np.eye(len(X))
        return 0.5 * y @ np.linalg.solve(K, y) + np.log(np.diag(np.linalg.cholesky(K))).sum()

    assert abs(val - f(x)) <= 1e-10 * abs(val)
    for ip in range(5):
        h = 1e-6 * x[ip]
        xp, xm = x.copy(), x.copy()
        xp[ip] += h
        xm[ip] -= h
        fd = (f(xp) - f(xm)) / (2 * h)
        expect = fd / 2 if ip < 2 else fd
        assert abs(grad[ip] - expect) <= 1e-5 * max(1.0, abs(expect)), (ip, grad[ip], expect)
    mean = oracle.nlml_predict(x, X, y, Xs)
    K = gram(x, X, X) + x[0] ** 2 * np.eye(len(X))
    assert np.abs(mean - gram(x, Xs, X) @ np.linalg.solve(K, y)).max() <= 1e-9 * np.abs(mean).max()
    x4 = np.array([0.1, 1.2, 1.0 / 0.8, 1.0 / 0.7])
    x5 = np.array([0.1, 1.2, 1.0 / 0.8, 0.0, 1.0 / 0.7])
    v4, g4 = oracle.nlml(x4, X, y)
    v5, g5 = oracle.nlml(x5, X, y)
    assert v4 == v5 and np.allclose(g4, g5[[0, 1, 2, 4]], rtol=1e-12, atol=0)  # OpenMP reduction order only
