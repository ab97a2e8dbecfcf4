"""GPU: the grid-sharded predict of the C-ABI (gple_*_predict_sharded: block-cyclic share -> predict -> ncclAllGather -> unpack), the native
counterpart of parallel.GridShardedStep for C++ callers (output.cpp:181-233 over several GPUs).
  * real RCCL with a one-rank communicator (a one-GPU box cannot hold two RCCL ranks on one device);
  * world = 2 and 3 as host threads with separate contexts on the one GPU, through an in-process all-gather plugged in with
    gple_set_allgather_function (tests/cpp/fake_allgather.cpp) — every rank must end up with the unsharded result, bit for bit."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import _capi as c
from gaussian_process_liouville_equation_amd import parallel
from tests import parity
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
_dp = C.POINTER(C.c_double)


def _sharded(api, fit, Xs, rank, world, comm, cplx):
    M = len(Xs)
    ow = 2 if cplx else 1
    mean, var, cut = np.empty(ow * M), np.empty(M), np.empty(ow * M)
    fn = api.lib.gple_complex_predict_sharded if cplx else api.lib.gple_real_predict_sharded
    fn.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_size_t, C.c_uint, C.c_int, C.c_int, C.c_void_p, _dp, _dp, _dp]
    Xs = np.ascontiguousarray(Xs)
    st = fn(api.ctx, fit.handle, Xs.ctypes.data_as(_dp), M, 0, rank, world, comm, mean.ctypes.data_as(_dp), var.ctypes.data_as(_dp), cut.ctypes.data_as(_dp))
    assert st == 0, (st, api.lib.gple_ctx_last_error(api.ctx))
    return (mean.view(np.complex128), var, cut.view(np.complex128)) if cplx else (mean, var, cut)


def test_shard_bounds_match_the_python_partition(gpu):
    lo, hi, per = C.c_size_t(), C.c_size_t(), C.c_size_t()
    gpu.lib.gple_shard_bounds.argtypes = [C.c_size_t, C.c_int, C.c_int] + [C.POINTER(C.c_size_t)] * 3
    for M in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert gpu.lib.gple_shard_bounds(M, r, world, C.byref(lo), C.byref(hi), C.byref(per)) == 0
                assert (lo.value, hi.value, per.value) == parallel.shard_bounds(M, r, world)
    assert gpu.lib.gple_shard_bounds(10, 2, 2, None, None, None) == 1


def test_sharded_predict_with_real_rccl_single_rank(gpu):
    path = "/opt/rocm/lib/librccl.so.1"
    if not os.path.exists(path):
        pytest.skip("no system RCCL on this box")
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)  # the library resolves ncclAllGather from the process image first

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    gpu.lib.gple_set_allgather_function(None)
    X, y, Xs = parity.synthetic_real(200, 777, 5)
    fit = gpu.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 0)
    ref = gpu.real_predict(fit, Xs)
    mean, var, cut = _sharded(gpu, fit, Xs, 0, 1, comm, False)
    assert np.array_equal(mean, ref["prediction"]) and np.array_equal(var, ref["variance"]) and np.array_equal(cut, ref["cutoff"])
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("world,cplx", [(2, False), (3, False), (2, True)])
def test_sharded_predict_between_threads(gpu, world, cplx):
    so = os.path.join(ROOT, "tests", "cpp", "libfake_allgather.so")
    if not os.path.exists(so):
        pytest.fail("tests/cpp/libfake_allgather.so missing: run __graft_entry__.build()")
    import gaussian_process_liouville_equation_amd as pkg
    fake = C.CDLL(so)
    fake.fake_group_create.restype = C.c_void_p
    fake.fake_comm_create.restype = C.c_void_p
    fake.fake_comm_create.argtypes = [C.c_void_p, C.c_int]
    fake.fake_comm_destroy.argtypes = [C.c_void_p]
    fake.fake_group_destroy.argtypes = [C.c_void_p]
    gpu.lib.gple_set_allgather_function.argtypes = [C.c_void_p]
    gpu.lib.gple_set_allgather_function(C.cast(fake.fake_allgather, C.c_void_p))
    try:
        M = 1001  # 8 blocks of 128, the last one short: the ranks hold different numbers of points
        X, yr, Xs = parity.synthetic_real(150, M, 9)
        y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0)) if cplx else yr
        theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05] if cplx else [1.0, 0.7086, 0.7056, 1e-2]
        # reference: every rank's share (the 128-point blocks r, r + world, ...) in one unsharded call each (the split of the
        # mean's k-sum depends on the rows of a call), scattered back to grid order
        fit0 = (gpu.complex_fit if cplx else gpu.real_fit)(theta, X, y, 0)
        pred = gpu.complex_predict if cplx else gpu.real_predict
        ref = {"prediction": np.empty(M, dtype=complex if cplx else float), "variance": np.empty(M), "cutoff": np.empty(M, dtype=complex if cplx else float)}
        block = np.arange(M) // 128
        for r in range(world):
            idx = np.nonzero(block % world == r)[0]
            part = pred(fit0, Xs[idx])
            for k in ref:
                ref[k][idx] = part[k]
        group = fake.fake_group_create(world)
        out, errs = {}, []

        def rank_main(r):
            try:
                api = pkg.open_api(0)  # one context (= stream) per rank, the fit replicated like on a real node
                fit = (api.complex_fit if cplx else api.real_fit)(theta, X, y, 0)
                comm = fake.fake_comm_create(group, r)
                out[r] = _sharded(api, fit, Xs, r, world, comm, cplx)
                fake.fake_comm_destroy(comm)
                api.close()
            except Exception as e:  # pragma: no cover
                errs.append(e)

        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(timeout=120) for t in th]
        assert not errs, errs
        fake.fake_group_destroy(group)
        for r in range(world):
            mean, var, cut = out[r]
            assert np.array_equal(mean, ref["prediction"]) and np.array_equal(var, ref["variance"]) and np.array_equal(cut, ref["cutoff"])
    finally:
        gpu.lib.gple_set_allgather_function(None)


def test_sharded_predict_without_rccl_reports_a_collective_error():
    """world > 1, no transport plugged in and no loadable librccl: GPLE_ERR_COLLECTIVE with a message, not a crash (the branch used to
    build its message from a second dlerror() call, which returns NULL).  Own process: the resolved entry point is cached per process."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import gaussian_process_liouville_equation_amd as pkg
from tests import parity
api = pkg.open_api(0)
X, y, Xs = parity.synthetic_real(64, 300, 3)
fit = api.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 0)
dp = C.POINTER(C.c_double)
fn = api.lib.gple_real_predict_sharded
fn.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_size_t, C.c_uint, C.c_int, C.c_int, C.c_void_p, dp, dp, dp]
out = np.empty(300)
st = fn(api.ctx, fit.handle, np.ascontiguousarray(Xs).ctypes.data_as(dp), 300, 0, 0, 2, C.c_void_p(1), out.ctypes.data_as(dp), None, None)
api.lib.gple_ctx_last_error.restype = C.c_char_p
api.lib.gple_ctx_last_error.argtypes = [C.c_void_p]
msg = api.lib.gple_ctx_last_error(api.ctx)
print("STATUS", st, msg.decode())
''' % ROOT
    env = dict(os.environ, GPLE_RCCL_LIBRARY="/nonexistent/librccl.so.1")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("STATUS")][0].split(" ", 2)
    assert int(line[1]) == 5, res.stdout  # GPLE_ERR_COLLECTIVE (include/gple.h)
    assert "ncclAllGather not found" in line[2] and len(line[2]) > len("ncclAllGather not found: ")


@pytest.mark.parametrize("weights,cplx", [([3, 0, 5], False), ([2, 1], True), ([0, 1], False)])
def test_weighted_deal_between_threads(gpu, weights, cplx):
    """gple_*_predict_dealt (the hybrid element x grid plans of DESIGN.md §7): unequal shares, a rank without a share comes without the fit and
    still gets the full grid; every rank's result equals the shares predicted unsharded, bit for bit."""
    so = os.path.join(ROOT, "tests", "cpp", "libfake_allgather.so")
    if not os.path.exists(so):
        pytest.fail("tests/cpp/libfake_allgather.so missing: run __graft_entry__.build()")
    import gaussian_process_liouville_equation_amd as pkg
    world = len(weights)
    fake = C.CDLL(so)
    fake.fake_group_create.restype = C.c_void_p
    fake.fake_comm_create.restype = C.c_void_p
    fake.fake_comm_create.argtypes = [C.c_void_p, C.c_int]
    fake.fake_comm_destroy.argtypes = [C.c_void_p]
    fake.fake_group_destroy.argtypes = [C.c_void_p]
    gpu.lib.gple_set_allgather_function.argtypes = [C.c_void_p]
    gpu.lib.gple_set_allgather_function(C.cast(fake.fake_allgather, C.c_void_p))
    try:
        M = 2000  # 16 blocks, the last one short
        X, yr, Xs = parity.synthetic_real(150, M, 19)
        y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0)) if cplx else yr
        theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05] if cplx else [1.0, 0.7086, 0.7056, 1e-2]
        fit0 = (gpu.complex_fit if cplx else gpu.real_fit)(theta, X, y, 0)
        pred = gpu.complex_predict if cplx else gpu.real_predict
        ref = {"prediction": np.empty(M, dtype=complex if cplx else float), "variance": np.empty(M), "cutoff": np.empty(M, dtype=complex if cplx else float)}
        for r in range(world):
            idx = parallel.dealt_indices(M, r, weights)[0].numpy()
            if len(idx):
                part = pred(fit0, Xs[idx])
                for k in ref:
                    ref[k][idx] = part[k]
        group = fake.fake_group_create(world)
        out, errs = {}, []
        ow = 2 if cplx else 1

        def rank_main(r):
            try:
                api = pkg.open_api(0)
                fit = (api.complex_fit if cplx else api.real_fit)(theta, X, y, 0) if weights[r] else None
                comm = fake.fake_comm_create(group, r)
                mean, var, cut = np.empty(ow * M), np.empty(M), np.empty(ow * M)
                fn = api.lib.gple_complex_predict_dealt if cplx else api.lib.gple_real_predict_dealt
                fn.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_size_t, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, _dp, _dp, _dp]
                st = fn(api.ctx, fit.handle if fit else None, np.ascontiguousarray(Xs).ctypes.data_as(_dp), M, 0, r, world, (C.c_int * world)(*weights), comm,
                        mean.ctypes.data_as(_dp), var.ctypes.data_as(_dp), cut.ctypes.data_as(_dp))
                assert st == 0, (st, api.lib.gple_ctx_last_error(api.ctx))
                out[r] = (mean.view(np.complex128), var, cut.view(np.complex128)) if cplx else (mean, var, cut)
                fake.fake_comm_destroy(comm)
                api.close()
            except Exception as e:  # pragma: no cover
                errs.append(e)

        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(timeout=120) for t in th]
        assert not errs, errs
        fake.fake_group_destroy(group)
        for r in range(world):
            mean, var, cut = out[r]
            assert np.array_equal(mean, ref["prediction"]) and np.array_equal(var, ref["variance"]) and np.array_equal(cut, ref["cutoff"])
        # a rank with a share but without the fit is a caller error, not a hang
        fn = gpu.lib.gple_real_predict_dealt
        fn.argtypes = [C.c_void_p, C.c_void_p, _dp, C.c_size_t, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p, _dp, _dp, _dp]
        assert fn(gpu.ctx, None, np.ascontiguousarray(Xs).ctypes.data_as(_dp), M, 0, 0, 2, (C.c_int * 2)(1, 1), C.c_void_p(1), None, None, None) == 1
    finally:
        gpu.lib.gple_set_allgather_function(None)
