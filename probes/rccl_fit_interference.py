"""Does an RCCL communicator in the process change the fit's device time?  (round 3: bench.py --via capi --comm-at-one showed fit_device 3.9 ms
against 2.0 ms without a communicator.)  usage: python probes/rccl_fit_interference.py [none|load|comm]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gaussian_process_liouville_equation_amd as pkg
from tests.test_gpu_configs import config_inputs, THETA_R

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
torch.cuda.set_device(0)
api = pkg.open_api(0)
api.enable_timing(True)
X, y, _, _ = config_inputs(4096, 8, 1)


def timing(tag):
    for _ in range(3):
        f = api.real_fit(THETA_R, X, y, 3); f.scalars; f.release()
    api.enable_timing(True)
    vals = []
    for _ in range(10):
        f = api.real_fit(THETA_R, X, y, 3); f.scalars; f.release()
        vals.append(api.timing(0)[0])
    print(f"{tag}: fit N=4096 {np.median(vals):.4f} ms (min {min(vals):.4f}, max {max(vals):.4f})", flush=True)


timing("before")
if mode in ("load", "comm"):
    rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
    timing("librccl loaded")
if mode == "comm":
    class Uid(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = Uid(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    timing("communicator created")
    a = torch.zeros(1 << 20, dtype=torch.float64, device="cuda"); b = torch.zeros(1 << 20, dtype=torch.float64, device="cuda")
    rccl.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    assert rccl.ncclAllGather(a.data_ptr(), b.data_ptr(), 1 << 20, 8, comm, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    timing("after one all-gather")
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    timing("communicator destroyed")
api.close()
