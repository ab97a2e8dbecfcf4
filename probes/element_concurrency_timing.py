"""Three density-matrix elements (two real, one complex) fitted one after the other on one context vs concurrently on an
ApiPool (three contexts = three HIP streams, one host thread each).  Run on the GPU box: python probes/element_concurrency_timing.py"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import kernels as K
import parity

gpu = pkg.open_api(0)
pool = K.ApiPool(3)
for N in (500, 1024, 2048):
    X, y, _ = parity.synthetic_real(N, 16, 5 + N)
    yc = 0.5 * y * np.exp(0.5j * (X[:, 0] + 10.0))
    ts = {(0, 0): (X, y.astype(complex)), (1, 0): (X, yc), (1, 1): (X, (0.3 * y).astype(complex))}
    pv = {(0, 0): [1.0, 0.7086, 0.7056, 1e-2], (1, 0): [1.0, 1.0, 0.7086, 0.7056, 1.2, 0.8, 0.6, 1e-2], (1, 1): [1.0, 0.8, 0.6, 1e-2]}
    for deriv in (False, True):
        for name, api in (("one context", gpu), ("pool of 3", pool)):
            def run():
                ks = K.TrainingKernels(pv, ts, True, True, deriv, api=api)
                return ks.calculate_population() + ks.calculate_purity()
            run()
            t = time.perf_counter(); n = 5
            for _ in range(n):
                run()
            print(f"N={N} derivative={deriv} {name}: {(time.perf_counter() - t) / n * 1e3:.2f} ms", flush=True)
pool.close(); gpu.close()
