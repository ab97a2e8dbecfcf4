"""The fit's side stream and HIP's hardware queues (round 3).  Streams are bound to a small set of hardware queues (GPU_MAX_HW_QUEUES, default 4)
in creation order; when the context's side stream lands on the queue of its main stream the block-row inverse no longer runs beside the panels
— and with a low-priority side stream the fit at N = 4096 took 3.85 ms instead of 2.0 ms once an RCCL communicator had been created first.
usage: python probes/hwqueue_fit_probe.py <streams created (and used) before the context> <comm: 0|1>   (env: GPLE_CHOL_SIDE_PRIORITY, GPU_MAX_HW_QUEUES)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R

pre, comm_on = int(sys.argv[1]), int(sys.argv[2])
torch.cuda.set_device(0)
keep = []
if comm_on:
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = Uid(), C.c_void_p()
    os.dup2(2, 1)
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
for _ in range(pre):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        keep.append(torch.zeros(16, device="cuda") + 1)
    keep.append(s)
torch.cuda.synchronize()
out = {}
for N in (1024, 4096):
    api = pkg.open_api(0)
    api.enable_timing(True)
    X, y, _, _ = config_inputs(N, 8, 1)
    for _ in range(3):
        f = api.real_fit(THETA_R, X, y, 3); f.scalars; f.release()
    api.enable_timing(True)
    vals = []
    for _ in range(10):
        f = api.real_fit(THETA_R, X, y, 3); f.scalars; f.release()
        vals.append(api.timing(0)[0])
    out[N] = float(np.median(vals))
    api.close()
sys.stderr.write(f"pre_streams={pre} comm={comm_on} side_prio={os.environ.get('GPLE_CHOL_SIDE_PRIORITY', 'low')} hwq={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: "
                 f"fit N=1024 {out[1024]:.3f} ms, N=4096 {out[4096]:.3f} ms\n")
