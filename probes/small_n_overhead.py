"""Where an objective evaluation at the reference's typical size (N = 300) spends its time: wall clock per call against the
device time the library's own event timers report for the fit and the predict inside it."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
import parity
api = pkg.open_api(0)
for N in (300, 500, 1024):
    X, y, Xs = parity.synthetic_real(N, N // 2, 11 + N)
    ye = np.interp(Xs[:, 0], np.sort(X[:, 0]), y[np.argsort(X[:, 0])]).astype(complex)
    th = [1.0, 0.7086, 0.7056, 1e-2]
    yc = y.astype(complex)
    for g in (False, True):
        for _ in range(5): api.loose_function(th, X, yc, Xs, ye, want_grad=g)
        api.enable_timing(True)
        n = 50
        t = time.perf_counter()
        for _ in range(n): api.loose_function(th, X, yc, Xs, ye, want_grad=g)
        wall = (time.perf_counter() - t) / n * 1e3
        _, ft, fc = api.timing(0); _, pt, pc = api.timing(1)
        print(f"N={N} grad={g}: wall {wall:.3f} ms, fit on device {ft / fc:.3f} ms, predict on device {pt / pc:.3f} ms, host/other {wall - ft / fc - pt / pc:.3f} ms", flush=True)
        api.enable_timing(False)
        obj = api.objective(X, yc, Xs, ye)
        for _ in range(5): obj(th, want_grad=g)
        t = time.perf_counter()
        for _ in range(n): obj(th, want_grad=g)
        print(f"    resident objective: wall {(time.perf_counter() - t) / n * 1e3:.3f} ms", flush=True)
        obj.release()
api.close()
