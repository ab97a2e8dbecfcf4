"""Cycle stamps inside rownorm2_kernel (wave 0 of workgroup 0: k-step entry and the point after the step's last MFMA) at C4r: how long a k-step takes
against its 8192 cycles of MFMA time, and where the rest goes.  Needs the instrumented build: `git apply probes/rownorm_stamps.patch && make -C
gaussian_process_liouville_equation_amd/csrc` (adds two __device__ stamp stores per k-step and gple_debug_rownorm_stamps; revert afterwards — the stamp
stores cost ~600 cycles per step themselves, so read differences between variants, not absolute periods)."""
import ctypes, os, sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R
api = pkg.open_api(0)
lib = api.lib
N, G = 4096, 512
X, y, grid, _ = config_inputs(N, G, 1)
fit = api.real_fit(THETA_R, X, y, 3)
from gaussian_process_liouville_equation_amd import _capi as c
st = np.zeros(8192, dtype=np.int64); cnt = ctypes.c_int()
for rep in range(2):
    p = api.real_predict(fit, grid, flags=c.PREDICT_FULL)
    lib.gple_debug_rownorm_stamps(st.ctypes.data_as(ctypes.c_void_p), ctypes.byref(cnt))
n = cnt.value
print("stamps", n)
a = st[:n].reshape(-1, 2)  # (entry, after-mfma) per k-step
entry = a[:, 0]; after = a[:, 1]
step = np.diff(entry)            # k-step period
mf = (after - entry)[:-1]        # entry -> after last MFMA issued
bd = entry[1:] - after[:-1]      # after MFMA -> next entry (store, barrier)
np.save("/root/repo/gpurun_out/rn_stamps.npy", a)
sel = step < 20000
print("k-steps", len(step), "median period", np.median(step[sel]), "median mfma-phase", np.median(mf[sel]), "median boundary", np.median(bd[sel]))
print("period percentiles", np.percentile(step[sel], [5, 25, 50, 75, 95]))
print("first 40 periods", step[:40].tolist())
print("first 40 mfma phases", mf[:40].tolist())
