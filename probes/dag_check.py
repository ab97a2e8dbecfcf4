"""The one-launch factorisation (GPLE_CHOL_SCHEME=dag) against the fit's identities and its own clock.
usage: GPLE_CHOL_SCHEME=dag python probes/dag_check.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity

sizes = [int(a) for a in sys.argv[1:]] or [256, 1024]
api = pkg.open_api(0)
api.enable_timing(True)
for N in sizes:
    X, y, Xs = parity.synthetic_real(N, 64, 20240607 + N)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    t0 = time.time()
    fit = api.real_fit(theta, X, y, 3)
    sc = fit.scalars
    print(f"N={N}: info {sc['info']} error {sc['error']:.6e} ({time.time() - t0:.2f} s)", flush=True)
    if N <= 4096:
        K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
        n1 = lambda A: np.abs(A).sum(axis=0).max()
        print(f"   |K W - I|_1 / (n eps |K| |W|) = {n1(K @ W - np.eye(K.shape[0])) / (K.shape[0] * parity.EPS * n1(K) * n1(W)):.3f}"
              f"   |K v - y| / (n eps ..) = {np.abs(K @ v - ys).max() / (K.shape[0] * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max())):.3f}", flush=True)
    fit.release()
    vals = []
    for _ in range(8):
        f = api.real_fit(theta, X, y, 3); s = f.scalars; f.release()
        vals.append(api.timing(0)[0])
    print(f"   fit {np.median(vals):.4f} ms (min {min(vals):.4f}) info {s['info']}", flush=True)
api.close()
