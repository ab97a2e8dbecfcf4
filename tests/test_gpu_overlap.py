"""GPU: part of a predict beside the fit it follows (GPLE_PREDICT_OVERLAP=1; csrc/gple_predict.hip, launch_predict_overlapped): K* and the contraction over the
N-tiles whose rows of T are final at the factorisation's first fork run on a stream of the context's own, with the virtual groups' per-lane sums saved, the late tiles
start from them behind the fit — every output must agree bit for bit with the predict that waits for the fit (the switch is read once per process: child processes)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

CODE = r"""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C
api = pkg.open_api(0)
lib = api.lib
dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
out = []
for (N, cplx, M, shard) in [(4096, False, 32768, None), (4096, False, 262144, (3, 8)), (2100, True, 20000, None), (4096, False, 262144, (0, 2))]:
    X, y, grid, _ = config_inputs(N, 512, 20240607 + N + cplx, cplx=cplx)
    rng = np.random.default_rng(N)
    pts = grid if M == len(grid) else np.ascontiguousarray(X[rng.integers(0, N, M)] + rng.normal(0, 0.4, (M, 2)))
    th = np.array(THETA_C if cplx else THETA_R)
    dX, dpts = torch.tensor(X, device="cuda"), torch.tensor(pts, device="cuda")
    dy = torch.tensor(np.ascontiguousarray(np.asarray(y, dtype=complex)).view(float) if cplx else np.asarray(y, dtype=float), device="cuda")
    ow = 2 if cplx else 1
    o = torch.full((2 * ow + 1, M), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):  # the second round reuses the early buffers behind the first one's late part
        h = C.c_void_p()
        thp = th.ctypes.data_as(C.POINTER(C.c_double))
        if cplx:
            st = lib.gple_complex_fit_create(api.ctx, thp, dp(dX), dp(dy), N, 3 | c.IO_DEVICE, None, C.byref(h))
        else:
            st = lib.gple_real_fit_create(api.ctx, thp, dp(dX), dp(dy), 0, N, 3 | c.IO_DEVICE, None, C.byref(h))
        assert st == 0, lib.gple_ctx_last_error(api.ctx)
        om, ov, oc = o[0:ow].reshape(-1), o[ow], o[ow + 1:].reshape(-1)
        if shard is None:
            fn = lib.gple_complex_predict if cplx else lib.gple_real_predict
            st = fn(api.ctx, h, dp(dpts), C.c_size_t(M), c.IO_DEVICE | c.PREDICT_FULL, None, dp(om), dp(ov), dp(oc), None)
        else:
            lib.gple_set_allgather_function.argtypes = [C.c_void_p]
            lib.gple_set_allgather_function(C.cast(lib.gple_debug_solo_allgather, C.c_void_p))
            fn = lib.gple_real_predict_sharded
            fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_uint, C.c_int, C.c_int, C.c_void_p] + [C.POINTER(C.c_double)] * 3
            st = fn(api.ctx, h, dp(dpts), M, c.IO_DEVICE | c.PREDICT_FULL, shard[0], shard[1], C.c_void_p(1 + shard[0] + 256 * shard[1]), dp(om), dp(ov), dp(oc))
        assert st == 0, lib.gple_ctx_last_error(api.ctx)
        sc = c.ComplexFitScalars() if cplx else c.RealFitScalars()
        st = (lib.gple_complex_fit_get_scalars if cplx else lib.gple_real_fit_get_scalars)(h, C.byref(sc))
        assert st == 0 and sc.info == 0, (st, sc.info)
        (lib.gple_complex_fit_release if cplx else lib.gple_real_fit_release)(h)
        res = o.cpu().numpy().copy()
        if shard is not None:  # the other ranks' blocks are zero-filled by the stand-in transport: keep this rank's
            blocks = np.arange((M + 127) // 128)
            mine = np.repeat(blocks %% shard[1] == shard[0], 128)[:M]
            res = res[:, mine]
        assert np.isfinite(res).all()
        out.append(res.ravel())
lib.gple_debug_overlapped_predicts.restype = C.c_long
lib.gple_debug_overlapped_predicts.argtypes = [C.c_void_p]
print("overlapped predicts:", lib.gple_debug_overlapped_predicts(api.ctx), flush=True)
np.save(sys.argv[1], np.concatenate(out))
api.close()
""" % ROOT


def test_a_predict_beside_its_fit_has_the_bits_of_the_predict_behind_it():
    res = []
    with tempfile.TemporaryDirectory() as d:
        for on in ("0", "1"):
            f = os.path.join(d, f"o{on}.npy")
            run = subprocess.run([sys.executable, "-c", CODE, f], check=True, env=dict(os.environ, GPLE_PREDICT_OVERLAP=on), cwd=ROOT, timeout=900, capture_output=True, text=True)
            count = int(run.stdout.strip().splitlines()[-1].split(":")[1])
            assert count == (8 if on == "1" else 0), (on, run.stdout[-400:], run.stderr[-400:])  # every case of the child, both rounds
            res.append(np.load(f))
    assert res[0].shape == res[1].shape and len(res[0]) > 100000
    assert np.array_equal(res[0], res[1]), (np.abs(res[0] - res[1]).max(), np.flatnonzero(res[0] != res[1])[:10])
