"""How long do N_live 128-row blocks of LIVE test points take in the row-norm kernel when nothing else is in the launch?
(diagnosis of the pruned predict: 378 live blocks of the C4r grid take 25 ms inside a launch of 1024 blocks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, THETA_R

api = pkg.open_api(0)
N = 4096
X, y, grid, _ = config_inputs(N, 512, 1)
fit = api.real_fit(THETA_R, X, y, 0)
rng = np.random.default_rng(0)
for nblk in (128, 256, 378, 512, 1024):
    pts = X[rng.integers(0, N, nblk * 128)] + rng.normal(0, 0.3, (nblk * 128, 2))
    for flags, name in ((c.PREDICT_FULL, "full"), (0, "default")):
        api.real_predict(fit, pts, flags=flags, want=("variance",))
        api.enable_timing(True)
        for _ in range(3):
            api.real_predict(fit, pts, flags=flags, want=("variance",))
        _, tot, cnt = api.timing(2)
        print(f"{nblk:5d} live blocks ({name}): rownorm {tot / 3:.2f} ms per predict, {cnt // 3} launches; {nblk * 128 * N * (N + 1) / (tot / 3 * 1e-3) / 1e12:.1f} TFLOP/s", flush=True)
# the same number of rows, far from the data (dead in the default mode, contracted in full mode)
far = np.stack([np.full(378 * 128, 15.0), np.linspace(-5, 30, 378 * 128)], 1)
for flags, name in ((c.PREDICT_FULL, "full"), (0, "default")):
    api.real_predict(fit, far, flags=flags, want=("variance",))
    api.enable_timing(True)
    for _ in range(3):
        api.real_predict(fit, far, flags=flags, want=("variance",))
    _, tot, cnt = api.timing(2)
    print(f"  378 far blocks ({name}): rownorm {tot / 3:.2f} ms per predict", flush=True)
api.close()
