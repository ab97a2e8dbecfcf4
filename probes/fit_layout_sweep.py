"""One fit per padded size n = 256 k (real, k = 1..20; complex n = 512 k, k = 1..8): K W = I, K v = y and the LOOCV error from the getters —
every outer-block layout and fork pattern of chol_block_bounds / chol_fork_points once.  usage: GPLE_POISON_T=1 python probes/fit_layout_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity

api = pkg.open_api(0)
n1 = lambda A: np.abs(A).sum(axis=0).max()
bad = 0
for k in range(1, 21):
    N = 256 * k - 3
    X, y, _ = parity.synthetic_real(N, 8, 1000 + k)
    fit = api.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 3)
    K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
    r1 = n1(K @ W - np.eye(N)) / (N * parity.EPS * n1(K) * n1(W))
    r2 = np.abs(K @ v - ys).max() / (N * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max()))
    r3 = abs(((v / np.diag(W)) ** 2).sum() - fit.scalars["error"]) / fit.scalars["error"]
    ok = fit.scalars["info"] == 0 and r1 <= 50 and r2 <= 50 and r3 <= 1e-9
    bad += not ok
    print(f"real N={N}: |KW-I| {r1:.2f}  |Kv-y| {r2:.2f} (units of N eps scale)  error rel {r3:.1e}  {'ok' if ok else 'FAILED'}", flush=True)
    fit.release()
for k in range(1, 9):
    N = 256 * k - 5
    X, yr, _ = parity.synthetic_real(N, 8, 2000 + k)
    y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    fit = api.complex_fit([1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2], X, y, 3)
    K, Kt, v, ys = fit.get(c.C_KERNEL), fit.get(c.C_PSEUDO), fit.get(c.C_INVLBL), fit.get(c.C_LABEL)
    r = np.abs(K @ v + Kt @ v.conj() - ys).max() / np.abs(ys).max()
    ok = fit.scalars["info"] == 0 and r <= 1e-7
    bad += not ok
    print(f"complex N={N}: |K v + Kt conj(v) - y| / |y| {r:.1e}  {'ok' if ok else 'FAILED'}", flush=True)
    fit.release()
api.close()
sys.exit(1 if bad else 0)
