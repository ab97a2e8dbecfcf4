import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
import parity
api = pkg.open_api(0)
for N in (500, 1024, 2048, 4096):
    X, y, Xs = parity.synthetic_real(N, N // 4, 7 + N)
    ye = np.interp(Xs[:, 0], np.sort(X[:, 0]), y[np.argsort(X[:, 0])])
    th = [1.0, 0.7086, 0.7056, 1e-2]
    thc = [1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2]
    yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0)); yec = 0.5 * ye * np.exp(0.5j * (Xs[:, 0] + 10.0))
    for name, x, yy, yee in (("real", th, y.astype(complex), ye.astype(complex)), ("complex", thc, yc, yec)):
        if name == "complex" and N > 2048: continue
        for g in (False, True):
            api.loose_function(x, X, yy, Xs, yee, want_grad=g)
            t = time.perf_counter(); n = 5
            for _ in range(n): api.loose_function(x, X, yy, Xs, yee, want_grad=g)
            print(f"N={N} {name} grad={g}: {(time.perf_counter()-t)/n*1e3:.2f} ms", flush=True)
api.close()
