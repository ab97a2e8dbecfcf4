"""One evaluation of the resident objective (opt.cpp:441-482: fit + predict of the 5N extra points) with and without gradient, real and complex,
at the sizes of the configs.  GPLE_PREDICT_SKIP=0 restores the full contraction for the A/B.  usage: python probes/objective_eval_timing.py [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, extra_set, THETA_R, THETA_C

api = pkg.open_api(0)
for N in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    for cplx in (False, True):
        X, y, _, _ = config_inputs(N, 8, 3, cplx=cplx)
        Xe, ye = extra_set(X, 77, cplx)
        obj = api.objective(X, np.asarray(y, dtype=complex), Xe, ye)
        th = np.array(THETA_C if cplx else THETA_R)
        for g in (False, True):
            obj(th, want_grad=g)
            t = time.perf_counter(); n = 5
            for _ in range(n):
                v = obj(th, want_grad=g)
            print(f"N={N} {'complex' if cplx else 'real'} grad={g}: {(time.perf_counter() - t) / n * 1e3:.2f} ms  value {v[0]:.6e}", flush=True)
        obj.release()
api.close()
