"""Soak of the pipelined contraction (rownormp_kernel) and the fused small-n predict: the same predict many times, every result compared bit for bit with the
first — a race between an LDS-DMA slab and the reads of the buffer it lands in would show as a mismatch.  usage: python probes/soak_contraction.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
api = pkg.open_api(0)
for (N, G, cplx, flags, n) in [(4096, 512, False, c.PREDICT_FULL, reps), (4096, 512, False, 0, reps), (1024, 256, False, c.PREDICT_FULL, 10 * reps), (2048, 256, True, c.PREDICT_FULL, reps),
                               (256, 128, False, c.PREDICT_FULL, 20 * reps), (8192, 512, False, 0, reps // 2)]:
    X, y, grid, _ = config_inputs(N, G, 20240607 + N, cplx=cplx)
    fit = (api.complex_fit if cplx else api.real_fit)(THETA_C if cplx else THETA_R, X, y, 3)
    pred = api.complex_predict if cplx else api.real_predict
    first = pred(fit, grid, flags=flags)
    t0 = time.time(); bad = 0
    for r in range(n):
        p = pred(fit, grid, flags=flags)
        bad += any(not np.array_equal(p[k], first[k]) for k in ("prediction", "variance", "cutoff"))
    print(f"N={N} {'complex' if cplx else 'real'} grid {G}^2 flags {flags:#x}: {n} repeats, {bad} differ from the first ({time.time() - t0:.0f} s)", flush=True)
    fit.release()
api.close()
