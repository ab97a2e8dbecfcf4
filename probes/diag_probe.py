"""Stage-by-stage cycle stamps of potrf_diag_kernel (gple_debug_potrf_diag, not a public entry) + correctness against numpy.
Stages between stamps: see `names`."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi

api = pkg.open_api(0)
lib = api.lib
rng = np.random.default_rng(0)
B = rng.standard_normal((64, 96))
A = B @ B.T / 96 + 0.5 * np.eye(64)
Af = np.asfortranarray(A)
T = np.zeros((64, 64), order="F")
stamps = np.zeros(16, dtype=np.int64)
ms = ctypes.c_float()
lib.gple_debug_potrf_diag.restype = ctypes.c_int
rc = lib.gple_debug_potrf_diag(api.ctx, Af.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p), stamps.ctypes.data_as(ctypes.c_void_p), 200, ctypes.byref(ms))
assert rc == 0, rc
ref = np.linalg.inv(np.linalg.cholesky(A))
print("max |T - inv(chol(A))| =", np.abs(T - ref).max(), " scale", np.abs(ref).max())
print("us per launch (back to back):", ms.value * 1e3)
names = "load chain0 upd0 chain1+inv0 upd1 chain2+inv1 upd2+T10 chain3+inv2 inv3+row3 T3x storeT trsm".split()
d = np.diff(stamps[:13])
for n, c in zip(names, d):
    print(f"{n:>18}: {c:7d} cycles")
print("total", stamps[12] - stamps[0], "shader clock cycles (s_memtime)")
