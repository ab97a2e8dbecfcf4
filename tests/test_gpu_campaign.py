"""Randomised parity campaign: the HIP path against the oracle on random sizes (none a tile multiple), random
hyper-parameters and every flag combination, with tolerances scaled by the conditioning of the case's own kernel matrix.
Covers both predict paths (few rows / streaming) through the row counts drawn."""
import numpy as np
import pytest

from tests import parity
from gaussian_process_liouville_equation_amd import _capi as c

pytestmark = pytest.mark.gpu
EPS = parity.EPS


def _case(seed, complex_case):
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(2, 520))
    M = int(rng.choice([1, 7, 130, 700, 1500]))
    X, y, Xs = parity.synthetic_real(N, M, 500 + seed)
    Xv = X[rng.integers(0, N, max(1, N // 3))] + rng.normal(0, [0.4, 0.4], size=(max(1, N // 3), 2))
    yv = np.exp(-0.5 * (((Xv[:, 0] + 10.0) / 0.7086) ** 2 + ((Xv[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    ls = rng.uniform(0.35, 1.0, size=4)
    sn = float(10 ** rng.uniform(-2.0, -0.5))
    sf = float(rng.uniform(0.6, 1.6))
    if not complex_case:
        return N, M, X, y, Xs, Xv, yv, [sf, ls[0], ls[1], sn]
    ph = lambda Z: np.exp(0.5j * (Z[:, 0] + 10.0))
    theta = [sf, float(rng.uniform(0.7, 1.4)), ls[0], ls[1], float(rng.uniform(0.7, 1.4)), ls[2], ls[3], sn]
    return N, M, X, 0.5 * y * ph(X), Xs, Xv, 0.5 * yv * ph(Xv), theta


@pytest.mark.parametrize("seed", range(14))
def test_random_real_cases(gpu, oracle, seed):
    N, M, X, y, Xs, Xv, yv, theta = _case(seed, False)
    flags = 7
    fg, fo = gpu.real_fit(theta, X, y, flags), oracle.real_fit(theta, X, y, flags)
    K = fo.get(c.R_KERNEL)
    tol = max(1e-11, 100.0 * np.linalg.cond(K) * EPS)
    sg, so = fg.scalars, fo.scalars
    assert sg["info"] == 0 and so["info"] == 0
    for k in ("rescale_factor", "magnitude", "error", "population", "purity"):
        assert abs(sg[k] - so[k]) <= tol * abs(so[k]), (k, sg[k], so[k], tol)
    assert parity.rel(sg["first_order_average"], so["first_order_average"]) <= tol
    for k in ("error_derivative", "population_derivative"):
        assert np.abs(sg[k] - so[k]).max() <= 20 * tol * max(np.abs(so[k]).max(), 1e-300), (k, sg[k], so[k])
    # kernel.cpp:436-477: each entry is purity / l plus two sums of the opposite sign — priced against the purity / l it cancels
    pscale = max(np.abs(so["purity_derivative"]).max(), abs(so["purity"]) / min(theta[1], theta[2]))
    assert np.abs(sg["purity_derivative"] - so["purity_derivative"]).max() <= 20 * tol * pscale
    assert parity.rel(fg.get(c.R_INVLBL), fo.get(c.R_INVLBL)) <= tol
    assert parity.rel(fg.get(c.R_INVLBL_DERIV), fo.get(c.R_INVLBL_DERIV)) <= 20 * tol
    pg, po = gpu.real_predict(fg, Xs), oracle.real_predict(fo, Xs)
    scale = max(np.abs(po["prediction"]).max(), 1e-300)
    assert np.abs(pg["prediction"] - po["prediction"]).max() <= tol * max(scale, np.abs(fo.get(c.R_INVLBL)).max() * theta[0] ** 2)
    assert np.abs(pg["variance"] - po["variance"]).max() <= tol * theta[0] ** 2
    vg = gpu.real_predict(fg, Xv, flags=c.CALC_DERIVATIVE, labels=yv)
    vo = oracle.real_predict(fo, Xv, flags=c.CALC_DERIVATIVE, labels=yv)
    assert abs(vg["error"] - vo["error"]) <= tol * abs(vo["error"])
    assert np.abs(vg["error_derivative"] - vo["error_derivative"]).max() <= 20 * tol * np.abs(vo["error_derivative"]).max()
    lg = gpu.loose_function(theta, X, y.astype(complex), Xv, yv.astype(complex))
    lo = oracle.loose_function(theta, X, y.astype(complex), Xv, yv.astype(complex))
    assert abs(lg[0] - lo[0]) <= tol * abs(lo[0]) and np.abs(lg[1] - lo[1]).max() <= 20 * tol * np.abs(lo[1]).max()


@pytest.mark.parametrize("seed", range(8))
def test_random_complex_cases(gpu, oracle, seed):
    N, M, X, y, Xs, Xv, yv, theta = _case(100 + seed, True)
    N = min(N, 260)  # the literal complex oracle is O(35 N^3) on the CPU
    X, y = X[:N], y[:N]
    fg, fo = gpu.complex_fit(theta, X, y, 7), oracle.complex_fit(theta, X, y, 7)
    Kc = fo.get(c.C_KERNEL)
    tol = max(1e-10, 2000.0 * np.linalg.cond(Kc) * EPS)  # the Schur complement squares part of the conditioning
    sg, so = fg.scalars, fo.scalars
    assert sg["info"] == 0
    for k in ("rescale_factor", "magnitude", "error"):
        assert abs(sg[k] - so[k]) <= tol * abs(so[k]), (k, sg[k], so[k], tol)
    assert abs(sg["purity"] - so["purity"]) <= 100 * tol * abs(so["purity"])
    for k in ("error_derivative", "purity_derivative"):
        assert np.abs(sg[k] - so[k]).max() <= 100 * tol * np.abs(so[k]).max(), (k, sg[k], so[k])
    assert parity.rel(fg.get(c.C_INVLBL), fo.get(c.C_INVLBL)) <= tol
    pg, po = gpu.complex_predict(fg, Xs), oracle.complex_predict(fo, Xs)
    scale = max(np.abs(po["prediction"]).max(), np.abs(fo.get(c.C_INVLBL)).max() * theta[0] ** 2)
    assert np.abs(pg["prediction"] - po["prediction"]).max() <= tol * scale
    assert np.abs(pg["variance"] - po["variance"]).max() <= tol * 4.0 * theta[0] ** 2
    vg = gpu.complex_predict(fg, Xv, flags=c.CALC_DERIVATIVE, labels=yv)
    vo = oracle.complex_predict(fo, Xv, flags=c.CALC_DERIVATIVE, labels=yv)
    assert abs(vg["error"] - vo["error"]) <= tol * abs(vo["error"])
    assert np.abs(vg["error_derivative"] - vo["error_derivative"]).max() <= 100 * tol * np.abs(vo["error_derivative"]).max()
