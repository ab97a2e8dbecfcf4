"""Few-rows predict path (triangular GEMM + column norms) against the streaming rownorm_kernel as a function of the row count,
N = 1024 and 4096, device-resident timing through the library's own event timers."""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
import parity
api = pkg.open_api(0)
api.enable_timing(True)
for N in (1024, 4096):
    X, y, _ = parity.synthetic_real(N, 16, 3 + N)
    fit = api.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 1)
    for M in (2048, 4096, 8192, 12288, 16384, 24576, 32768):
        Xs = np.random.default_rng(M).normal(size=(M, 2)) * 0.7 + (-10.0, 14.0)
        row = []
        for force in ("0", "1"):
            os.environ["GPLE_PREDICT_SMALL_M"] = force
            api.real_predict(fit, Xs, want=("variance",))
            api.enable_timing(True)
            for _ in range(5):
                api.real_predict(fit, Xs, want=("variance",))
            _, tot, cnt = api.timing(1)
            row.append(tot / cnt)
        print(f"N={N} M={M}: streaming {row[0]:.3f} ms, few-rows {row[1]:.3f} ms (device time of the predict call)", flush=True)
api.close()
