"""One-point predicts from several host threads on one fit (what the reference's TBB workers do, evolve.cpp:392-420): calls per
second with 1, 4, 16 threads, with and without the combining front end (GPLE_POINT_COMBINE=0)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C

api = pkg.open_api(0)
for cplx in (False, True):
    for N in (1024, 4096):
        X, y, _, _ = config_inputs(N, 8, 1, cplx=cplx)
        fit = (api.complex_fit if cplx else api.real_fit)(THETA_C if cplx else THETA_R, X, y, 0)
        pts = X[:2048].copy()
        pred = api.complex_predict if cplx else api.real_predict
        for nt in (1, 4, 16):
            per = 1024 // nt
            def work(k):
                for i in range(per):
                    pred(fit, pts[k * per + i:k * per + i + 1], want=("cutoff",))
            ths = [threading.Thread(target=work, args=(k,)) for k in range(nt)]
            t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            dt = time.perf_counter() - t0
            print(f"{'complex' if cplx else 'real'} N={N} threads={nt}: {nt * per / dt:9.0f} one-point predicts/s ({dt / (nt * per) * 1e6:.1f} us per call)", flush=True)
        fit.release()
api.close()
