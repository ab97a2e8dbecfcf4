"""What would running the contraction beside the replicated fit buy?  Physics first: two contexts (two streams) on one GPU, one fitting
N = 4096, the other predicting one rank's share of the C4r grid (M / P points, full contraction) on an older fit — alone, then started together.
usage: python probes/overlap_probe.py [P ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, THETA_R

N, G = 4096, 512
X, y, grid, _ = config_inputs(N, G, 20240607 + 1)
a, b = pkg.open_api(0), pkg.open_api(0)
fit_b = b.real_fit(THETA_R, X, y, 3)


def fit_once():
    f = a.real_fit(THETA_R, X, y, 3)  # returns after the scalars are back: the fit is done
    f.release()


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


for P in [int(v) for v in sys.argv[1:]] or [8, 4]:
    pts = np.ascontiguousarray(grid[: len(grid) // P])
    pred_once = lambda: b.real_predict(fit_b, pts, flags=c.PREDICT_FULL)
    tf, tp = timed(fit_once, 10), timed(pred_once, 5)
    both = []
    for _ in range(6):
        bar = threading.Barrier(3)
        ends = {}

        def run(name, fn):
            bar.wait()
            fn()
            ends[name] = time.perf_counter()

        th = [threading.Thread(target=run, args=("fit", fit_once)), threading.Thread(target=run, args=("pred", pred_once))]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        both.append(((ends["fit"] - t0) * 1e3, (ends["pred"] - t0) * 1e3))
    both = np.array(both[1:])
    print(f"P={P}: fit alone {tf:.2f} ms, predict of M/{P} points alone {tp:.2f} ms (host pointers: with its copies), sum {tf + tp:.2f}; started together: fit done after "
          f"{both[:, 0].mean():.2f} ms, predict after {both[:, 1].mean():.2f} ms, both after {both.max(axis=1).mean():.2f} ms", flush=True)
a.close(); b.close()
