"""Device time of one fit (GPLE_TIMER_FIT) at the sizes of the BASELINE configs; A/B knobs through the environment
(GPLE_CHOL_OUTER=0|128|256|512, GPLE_CHOL_OVERLAP_MIN_N=...).  usage: python probes/fit_timing.py [real|complex|both] [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C

which = sys.argv[1] if len(sys.argv) > 1 else "both"
sizes = [int(a) for a in sys.argv[2:]] or [256, 1024, 2048, 4096, 8192]
api = pkg.open_api(0)
api.enable_timing(True)
for cplx in ((False, True) if which == "both" else ((which == "complex"),)):
    for N in sizes:
        if cplx and N > 4096:
            continue
        X, y, _, _ = config_inputs(N, 8, 1, cplx=cplx)
        fitf = api.complex_fit if cplx else api.real_fit
        for _ in range(3):
            f = fitf(THETA_C if cplx else THETA_R, X, y, 3); f.scalars; f.release()
        api.enable_timing(True)
        vals = []
        for _ in range(10):
            f = fitf(THETA_C if cplx else THETA_R, X, y, 3); s = f.scalars; f.release()
            vals.append(api.timing(0)[0])
        print(f"{'complex' if cplx else 'real'} N={N}: fit {np.median(vals):.4f} ms (min {min(vals):.4f}) err={s['error']:.6e}", flush=True)
api.close()
