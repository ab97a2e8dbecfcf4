"""Row-norm kernel time for 1024 blocks of which 384 are live, in different arrangements (which blocks of the launch are live)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R

api = pkg.open_api(0)
N = 4096
X, y, grid, _ = config_inputs(N, 512, 1)
fit = api.real_fit(THETA_R, X, y, 0)
rng = np.random.default_rng(0)
nb = 1024
def run(live_mask, name):
    pts = np.empty((nb * 128, 2))
    for b in range(nb):
        if live_mask[b]:
            pts[b * 128:(b + 1) * 128] = X[rng.integers(0, N, 128)] + rng.normal(0, 0.3, (128, 2))
        else:
            pts[b * 128:(b + 1) * 128] = [15.0, 40.0]
    api.real_predict(fit, pts, want=("variance",))
    api.prune_stats(reset=True)
    api.enable_timing(True)
    for _ in range(3):
        api.real_predict(fit, pts, want=("variance",))
    _, tot, cnt = api.timing(2)
    live, seen = api.prune_stats(reset=True)
    print(f"{name:28s}: rownorm {tot / 3:6.2f} ms per predict ({live // 3} of {seen // 3} blocks live)", flush=True)
idx = np.arange(nb)
run(idx < 384, "first 384 live")
run(idx >= nb - 384, "last 384 live")
run((idx % 8) < 3, "3 of every 8 live")
run(((idx % 4 == 1) | (idx % 4 == 2)) & (idx >= 96) & (idx < 864), "grid-like (2 of 4, middle)")
run(rng.permutation(nb) < 384, "random 384 live")
run((idx % 2) == 0, "every second live (512)")
run(idx < 512, "first 512 live")
api.close()
