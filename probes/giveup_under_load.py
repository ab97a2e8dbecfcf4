"""Do the waiting launches of the factorisation give up when other STREAMS of the same process keep every CU busy?  Three contexts (three streams) in three
host threads: two fit N = 4096 over and over, one contracts the full 512 x 512 grid over and over (one 110 KB workgroup per CU, ~60 ms per predict).
Prints the give-up / recovery counters of the fitting contexts.  usage: python probes/giveup_under_load.py [seconds]"""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, THETA_R

T = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
X, y, grid, _ = config_inputs(4096, 512, 20240607 + 1)
apis = [pkg.open_api(0) for _ in range(3)]
fit_p = apis[2].real_fit(THETA_R, X, y, 3)
stop = time.time() + T
counts = [0, 0, 0]


def fitter(i):
    while time.time() < stop:
        f = apis[i].real_fit(THETA_R, X, y, 3)
        assert np.isfinite(f.scalars["error"])
        f.release()
        counts[i] += 1


def predictor():
    while time.time() < stop:
        apis[2].real_predict(fit_p, grid, flags=c.PREDICT_FULL)
        counts[2] += 1


th = [threading.Thread(target=fitter, args=(0,)), threading.Thread(target=fitter, args=(1,)), threading.Thread(target=predictor)]
[t.start() for t in th]
[t.join() for t in th]
for i in (0, 1):
    g, r = ctypes.c_long(), ctypes.c_long()
    apis[i].lib.gple_debug_chol_knobs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
    apis[i].lib.gple_debug_chol_knobs(apis[i].ctx, -1, -1, -1, ctypes.byref(g), ctypes.byref(r))
    print(f"context {i}: {counts[i]} fits in {T:.0f} s beside {counts[2]} full-grid predicts of another stream: {g.value} give-ups, {r.value} recoveries", flush=True)
