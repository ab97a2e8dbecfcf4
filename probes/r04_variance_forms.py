"""Round 4 (What's weak 1a): which side of the C5c comparison is off?  tests/test_gpu_configs.py compares the library's variance
(k** - ||T k*||^2, a sum of squares) with the reference's four-term form (complex_kernel.cpp:631-637) evaluated by numpy in fp64 from the
getters P, Q — and had to widen that comparison from 1e-6 to 1e-6 N / 4096 at N = 8192.  Here the reference form is also evaluated in
np.longdouble (x87 80-bit: 64-bit mantissa, 11 more bits than fp64) on the same rows, from the same P, Q:
    |HIP - form_fp64|, |HIP - form_long|, |form_fp64 - form_long|
The last one is numpy's own rounding of the four cancelling N^2 sums.  usage: python probes/r04_variance_forms.py [N ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, complex_rect_kernels, THETA_C, X0, P0


def forms(theta, X, grid, rows, P, Q, dtype, cdtype):
    k, kt = complex_rect_kernels(theta, grid[rows], X)
    k, kt, P, Q = k.astype(dtype), kt.astype(cdtype), P.astype(cdtype), Q.astype(cdtype)
    kss = dtype(theta[0]) ** 2 * (dtype(theta[1]) ** 2 + dtype(theta[4]) ** 2 + dtype(theta[7]) ** 2)
    out = []
    for i in range(len(rows)):  # row by row: no N x N temporaries beyond P, Q themselves
        kr, pr = k[i], kt[i]
        t = (kr @ (P @ kr)) + (pr @ (P.conj() @ pr.conj())) + (pr @ (Q @ kr)) + (kr @ (Q.conj() @ pr.conj()))
        out.append((kss - t).real)
    return np.array(out, dtype=dtype)


def main(sizes):
    api = pkg.open_api(0)
    for N in sizes:
        G = {2048: 256, 4096: 512, 8192: 1024}.get(N, 256)
        X, y, grid, _ = config_inputs(N, G, 20240607 + (2 if N == 2048 else 3 if N == 4096 else 4), cplx=True)
        fit = api.complex_fit(THETA_C, X, y, 3)
        rows = np.argsort(((grid - [X0, P0]) ** 2).sum(axis=1))[:8]
        p = api.complex_predict(fit, grid[rows])
        P, Q = fit.get(c.C_UPPER_LEFT), fit.get(c.C_LOWER_LEFT)
        t0 = time.time()
        v64 = forms(THETA_C, X, grid, rows, P, Q, np.float64, np.complex128)
        vld = forms(THETA_C, X, grid, rows, P, Q, np.longdouble, np.clongdouble)
        hip = p["variance"]
        print(f"N={N}: |HIP - fp64 form| = {np.abs(hip - v64).max():.3e}   |HIP - long-double form| = {np.abs(hip - np.asarray(vld, dtype=np.float64)).max():.3e}   "
              f"|fp64 form - long-double form| = {np.abs(v64 - np.asarray(vld, dtype=np.float64)).max():.3e}   (variance values {hip.min():.3e} .. {hip.max():.3e}; {time.time() - t0:.0f} s)", flush=True)
        fit.release()
    api.close()


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [2048, 4096, 8192])
