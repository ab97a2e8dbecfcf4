"""Stage-by-stage cycle stamps of potrf_step_kernel (gple_debug_potrf_step, not a public entry) + correctness against numpy:
block column 1 of a (128 + below)-square SPD matrix whose block column 0 is already factored; pend = 1 leaves the update by
block column 0 to the kernel.  usage: python probes/step_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gaussian_process_liouville_equation_amd as pkg

api = pkg.open_api(0)
lib = api.lib
lib.gple_debug_potrf_step.restype = ctypes.c_int
rng = np.random.default_rng(0)


def run(pend, below, reps=200):
    n = 128 + below
    B = rng.standard_normal((n, 2 * n))
    K = B @ B.T / (2 * n) + 0.5 * np.eye(n)
    L = np.linalg.cholesky(K)
    A = K.copy()
    A[:, :64] = L[:, :64]                       # block column 0 factored
    if not pend:
        A[64:, 64:] -= L[64:, :64] @ L[64:, :64].T  # ... and its update applied
    A = np.asfortranarray(A)
    T = np.zeros((n, n), order="F")
    stamps = np.zeros(24, dtype=np.int64)
    ms = ctypes.c_float()
    rc = lib.gple_debug_potrf_step(api.ctx, A.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p), pend, below,
                                   stamps.ctypes.data_as(ctypes.c_void_p), reps, ctypes.byref(ms))
    assert rc == 0, rc
    Tref = np.linalg.inv(L[64:128, 64:128])
    err_t = np.abs(T[64:128, 64:128] - Tref).max()
    err_l = np.abs(A[128:, 64:128] - L[128:, 64:128]).max() if below else 0.0
    return err_t, err_l, ms.value * 1e3, stamps


if __name__ == "__main__":
    for pend in (0, 1):
        for below in (0, 64):
            et, el, us, st = run(pend, below)
            k = int(np.argmax(st)) + 1
            d = np.diff(st[:k])
            print(f"pend {pend} below {below}: |T - ref| {et:.2e}  |L21 - ref| {el:.2e}  {us:.2f} us per launch; stages (cycles): {d.tolist()}  total {st[k - 1] - st[0]}")
