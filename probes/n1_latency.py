"""N1 / N3 numbers: one-point predict latency through the C-ABI (what main.cpp:83 pays per call), the batched
form at M = 48 N (the per-tick request volume of evolve.cpp), and one full evolve tick on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C

api = pkg.open_api(0)
for N in (1024, 4096):
    X, y, _, _ = config_inputs(N, 8, 1)
    Xc, yc, _, _ = config_inputs(N, 8, 2, cplx=True)
    fr, fc = api.real_fit(THETA_R, X, y, 0), api.complex_fit(THETA_C, Xc, yc, 0)
    fr.scalars, fc.scalars
    rng = np.random.default_rng(0)
    pts = X[rng.integers(0, N, 48 * N)] + rng.normal(0, 0.3, size=(48 * N, 2))
    for name, fit, pred in (("real", fr, api.real_predict), ("complex", fc, api.complex_predict)):
        for _ in range(5):
            pred(fit, pts[:1], want=("cutoff",))
        t0 = time.perf_counter()
        for i in range(200):
            pred(fit, pts[i:i + 1], want=("cutoff",))
        one = (time.perf_counter() - t0) / 200
        pred(fit, pts, want=("cutoff",))
        t0 = time.perf_counter()
        for _ in range(3):
            pred(fit, pts, want=("cutoff",))
        batch = (time.perf_counter() - t0) / 3
        print(f"N={N} {name}: one-point predict {one * 1e6:.1f} us/call -> {48 * N} calls = {one * 48 * N * 1e3:.1f} ms; batched M=48N: {batch * 1e3:.2f} ms "
              f"({48 * N / batch / 1e6:.2f} Mpoints/s, x{one * 48 * N / batch:.0f})", flush=True)
    # one tick: 3 elements with N points each -> 8 N back-propagated points per element
    dens = {(0, 0): (X, y.astype(complex)), (1, 0): (Xc, yc), (1, 1): (X, (0.5 * y).astype(complex))}
    f11 = api.real_fit(THETA_R, X, 0.5 * y, 0)
    api.evolve([fr, fc, f11], 1, 2000.0, 1.0, dens)
    t0 = time.perf_counter()
    for _ in range(3):
        api.evolve([fr, fc, f11], 1, 2000.0, 1.0, dens)
    print(f"N={N}: evolve tick (3 elements x N points, {8 * 3 * N} batched predicts) {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
api.close()
