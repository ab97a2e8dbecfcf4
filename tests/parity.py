"""Shared parity checks: the same assertions drive the CPU oracle (pinned against the mpmath fixtures) and the HIP
path (pinned against fixtures and oracle).  Tolerances follow SURVEY.md §8(d): Gram entries at the ULP model
(4 + |a|) eps; everything downstream of the factorisation scales with cond(K) * eps."""
import numpy as np

from gaussian_process_liouville_equation_amd import _capi as c

EPS = 2.0 ** -53


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def cond_tol(g, factor=200.0, floor=1e-12):
    """relative tolerance for quantities that pass through K^-1: factor * cond * eps (cond from the fixture)."""
    cond = float(g["cond"]) if "cond" in g and np.isfinite(g["cond"]) else 1e6
    return max(floor, factor * cond * EPS)


def check_gram_ulp(K, Kref, theta, X, XR=None, proven=False):
    """|K - Kref| <= (4 + |a|) 2 u |Kref| with a the exponent argument, u = 2^-53: SURVEY.md §8(d)'s model of a Gram entry's rounding.
    proven=True (the fixtures with 10^4 entries and more): the worst case of the reference's operation order instead — subtract, divide by l,
    square, add, negate, halve, exp, add the ridge, multiply by sf^2 (kernel.cpp:46-47, 227): the argument carries at most 6 u relative
    (two roundings in d, five in d^2, one more in the sum), the exponential turns that into 6 |a| u, exp itself (< 1 ulp), the addition
    and the product add 4 u: (4 + 6 |a|) u.  The model above is what a typical entry keeps to — 4 of the 18 496 entries of real_d exceed it, by
    up to 18 %, with the CPU oracle (glibc exp) as well as on the GPU — the proven bound is what no correct implementation may exceed."""
    XR = X if XR is None else XR
    d0 = (X[:, None, 0] - XR[None, :, 0]) / theta[1]
    d1 = (X[:, None, 1] - XR[None, :, 1]) / theta[2]
    a = 0.5 * (d0 ** 2 + d1 ** 2)
    factor = np.maximum((4.0 + a) * 2, 4.0 + 6.0 * a) if proven else (4.0 + a) * 2
    bound = factor * EPS * np.abs(Kref) + 1e-300
    assert np.all(np.abs(K - Kref) <= bound), float((np.abs(K - Kref) / bound).max())
    if proven:  # ... and all but a handful of entries keep to the model
        model = (4.0 + a) * 2 * EPS * np.abs(Kref) + 1e-300
        assert (np.abs(K - Kref) > model).mean() <= 1e-3, float((np.abs(K - Kref) > model).mean())


def check_real_case(api, g, deriv, tight=None):
    tol = tight if tight is not None else cond_tol(g)
    flags = c.CALC_ERROR | c.CALC_AVERAGE | (c.CALC_DERIVATIVE if deriv else 0)
    fit = api.real_fit(g["theta"], g["X"], g["y"], flags)
    s = fit.scalars
    assert s["info"] == 0
    assert abs(s["rescale_factor"] - g["rescale"]) <= 4 * EPS * g["rescale"]
    for k in ("magnitude", "error", "population", "purity"):
        assert abs(s[k] - g[k]) <= tol * abs(g[k]), (k, s[k], float(g[k]))
    assert rel(s["first_order_average"], g["first_order"]) <= tol
    check_gram_ulp(fit.get(c.R_KERNEL), g["K"], g["theta"], g["X"], proven=g["K"].size > 10000)
    assert rel(fit.get(c.R_INVERSE), g["W"]) <= tol
    assert rel(fit.get(c.R_INVERSE_DIAG), np.diag(g["W"])) <= tol
    assert rel(fit.get(c.R_INVLBL), g["v"]) <= tol
    scale = np.abs(g["t_mean"]).max()
    p = api.real_predict(fit, g["Xs"])
    assert np.abs(p["prediction"] - g["t_mean"]).max() <= tol * scale
    assert np.abs(p["variance"] - g["t_var"]).max() <= tol * max(1.0, np.abs(g["t_var"]).max())
    assert np.abs(p["cutoff"] - g["t_cut"]).max() <= tol * scale / g["rescale"] + 1e-300
    pv = api.real_predict(fit, g["Xv"], flags=c.CALC_DERIVATIVE if deriv else 0, labels=g["tv"])
    assert abs(pv["error"] - g["v_error"]) <= tol * abs(g["v_error"])
    if deriv:
        dscale = np.abs(g["error_derivative"]).max()
        assert np.abs(s["error_derivative"] - g["error_derivative"]).max() <= 10 * tol * dscale
        assert np.abs(s["population_derivative"] - g["population_derivative"]).max() <= 10 * tol * np.abs(g["population_derivative"]).max()
        assert np.abs(s["purity_derivative"] - g["purity_derivative"]).max() <= 10 * tol * np.abs(g["purity_derivative"]).max()
        assert rel(fit.get(c.R_INVLBL_DERIV), g["dv"]) <= 10 * tol
        assert np.abs(pv["error_derivative"] - g["v_error_derivative"]).max() <= 10 * tol * np.abs(g["v_error_derivative"]).max()
    fit.release()


def check_complex_case(api, g, deriv, tol):
    flags = c.CALC_ERROR | c.CALC_AVERAGE | (c.CALC_DERIVATIVE if deriv else 0)
    fit = api.complex_fit(g["theta"], g["X"], g["y"], flags)
    s = fit.scalars
    assert s["info"] == 0
    assert abs(s["rescale_factor"] - g["rescale"]) <= 4 * EPS * g["rescale"]
    for k in ("magnitude", "error"):
        assert abs(s[k] - g[k]) <= tol * abs(g[k]), (k, s[k], float(g[k]))
    # the purity of a nearly singular Schur complement cancels heavily: looser
    assert abs(s["purity"] - g["purity"]) <= 1e4 * tol * abs(g["purity"]), (s["purity"], float(g["purity"]))
    assert rel(fit.get(c.C_KERNEL), g["K"]) <= 8 * EPS
    assert rel(fit.get(c.C_PSEUDO), g["Kt"]) <= 8 * EPS
    assert rel(fit.get(c.C_UPPER_LEFT), g["P"]) <= tol
    assert rel(fit.get(c.C_LOWER_LEFT), g["Q"]) <= tol
    assert rel(fit.get(c.C_INVLBL), g["v"]) <= tol
    scale = np.abs(g["t_mean"]).max()
    p = api.complex_predict(fit, g["Xs"])
    assert np.abs(p["prediction"] - g["t_mean"]).max() <= tol * scale
    assert np.abs(p["variance"] - g["t_var"]).max() <= tol * max(1.0, np.abs(g["t_var"]).max())
    assert np.abs(p["cutoff"] - g["t_cut"]).max() <= tol * scale / g["rescale"] + 1e-300
    pv = api.complex_predict(fit, g["Xv"], flags=c.CALC_DERIVATIVE if deriv else 0, labels=g["tv"])
    assert abs(pv["error"] - g["v_error"]) <= tol * abs(g["v_error"])
    if deriv:
        for k in ("error_derivative", "purity_derivative"):
            assert np.abs(s[k] - g[k]).max() <= 1e4 * tol * np.abs(g[k]).max(), k
        assert rel(fit.get(c.C_INVLBL_DERIV), g["dv"]) <= 10 * tol
        assert np.abs(pv["error_derivative"] - g["v_error_derivative"]).max() <= 10 * tol * np.abs(g["v_error_derivative"]).max()
    fit.release()


# fixtures: (name, relative tolerance for the complex cases — cond is not stored there)
REAL_FIXTURES = ["real_a", "real_b", "real_c"]
COMPLEX_FIXTURES = [("complex_a", 1e-10), ("complex_b", 1e-7), ("complex_c", 1e-11)]
# beyond one 64-block of the factorisation / one 128-row block of the contraction (oracle/gen_golden.py: real N = 136, M = 256; complex N = 72,
# i.e. 144 real rows): the multi-panel factorisation, the in-launch inverse and the contraction's tiling against 50-digit values directly
REAL_FIXTURES_LARGE = ["real_d"]
COMPLEX_FIXTURES_LARGE = [("complex_d", 6e-10)]  # 200 cond(K) eps with cond(K) = 2.6e4, the rule of the real cases (the oracle is at 1e-12)


def synthetic_real(N, M, seed):
    """SURVEY.md §8(d) synthetic inputs: Gaussian wave packet samples and a grid-like test set."""
    rng = np.random.Generator(np.random.PCG64(seed))
    X = rng.normal([-10.0, 14.112], [0.7086, 0.7056], size=(N, 2))
    y = np.exp(-0.5 * (((X[:, 0] + 10.0) / 0.7086) ** 2 + ((X[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)
    Xs = rng.normal([-10.0, 14.112], [1.5, 1.5], size=(M, 2))
    return X, y, Xs
