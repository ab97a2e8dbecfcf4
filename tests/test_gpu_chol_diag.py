"""GPU: the diagonal-block kernel of the Cholesky panel step (csrc/gple_chol.hip, potrf_diag_kernel) on its own, through the
library's diagnostic entry gple_debug_potrf_diag (not part of include/gple.h; probes/diag_probe.py prints its stage timings):
T = inv(chol(A)) for one 64 x 64 block against numpy.  The kernel is otherwise covered through every fit (a6, a12)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(gpu, A):
    lib = gpu.lib
    if not hasattr(lib, "gple_debug_potrf_diag"):
        pytest.fail("libgple_hip.so lacks gple_debug_potrf_diag")
    lib.gple_debug_potrf_diag.restype = ctypes.c_int
    Af = np.asfortranarray(A, dtype=np.float64)
    T = np.zeros((64, 64), order="F")
    stamps = np.zeros(16, dtype=np.int64)
    ms = ctypes.c_float()
    rc = lib.gple_debug_potrf_diag(gpu.ctx, Af.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p),
                                   stamps.ctypes.data_as(ctypes.c_void_p), 1, ctypes.byref(ms))
    assert rc == 0
    return T


@pytest.mark.parametrize("seed,ridge", [(0, 0.5), (1, 1e-2), (2, 1e-6)])
def test_diag_block_inverse_factor(gpu, seed, ridge):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((64, 96))
    A = B @ B.T / 96 + ridge * np.eye(64)
    T = _run(gpu, A)
    L = np.linalg.cholesky(A)
    ref = np.linalg.inv(L)
    assert np.all(np.triu(T, 1) == 0.0)  # a full block with exact zeros above the diagonal (merge tree, T^T T read it as such)
    # backward error: T A T^T = I to rounding, scaled by the condition of the block
    res = np.abs(T @ A @ T.T - np.eye(64)).max()
    assert res <= 64 * 2.3e-16 * np.linalg.cond(A) ** 0.5 * 8, res
    assert np.abs(T - ref).max() <= 1e-10 * np.abs(ref).max() * max(1.0, np.linalg.cond(L) * 1e-4)


def test_diag_block_of_a_gram_matrix(gpu):
    """the kind of block a fit hands over: squared-exponential Gram block with the sigma_n^2 ridge"""
    rng = np.random.default_rng(5)
    X = rng.normal(size=(64, 2)) * [0.7, 0.7]
    d = ((X[:, None, :] - X[None, :, :]) / [0.7086, 0.7056]) ** 2
    A = np.exp(-0.5 * d.sum(-1)) + 1e-4 * np.eye(64)
    T = _run(gpu, A)
    res = np.abs(T @ A @ T.T - np.eye(64)).max()
    assert res <= 1e-9, res


@pytest.mark.parametrize("pend", [0, 1])
@pytest.mark.parametrize("below", [0, 64])
@pytest.mark.parametrize("ridge", [0.5, 1e-5])
def test_one_launch_panel_step(gpu, pend, below, ridge):
    """potrf_step_kernel (gple_debug_potrf_step): block column 1 of a (128 + below)-square SPD matrix whose block column 0 is already
    factored.  pend = 1: the update by block column 0 has not been applied and the kernel does it itself (diagonal block before the
    chain, the rows below on the side waves); either way T_11 = inv(L_11) and the rows below come out as numpy's factor has them."""
    lib = gpu.lib
    if not hasattr(lib, "gple_debug_potrf_step"):
        pytest.fail("libgple_hip.so lacks gple_debug_potrf_step")
    lib.gple_debug_potrf_step.restype = ctypes.c_int
    n = 128 + below
    rng = np.random.default_rng(10 * pend + below)
    B = rng.standard_normal((n, 2 * n))
    K = B @ B.T / (2 * n) + ridge * np.eye(n)
    L = np.linalg.cholesky(K)
    A = K.copy()
    A[:, :64] = L[:, :64]
    if not pend:
        A[64:, 64:] -= L[64:, :64] @ L[64:, :64].T
    A = np.asfortranarray(A)
    T = np.zeros((n, n), order="F")
    stamps = np.zeros(24, dtype=np.int64)
    ms = ctypes.c_float()
    rc = lib.gple_debug_potrf_step(gpu.ctx, A.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p), pend, below,
                                   stamps.ctypes.data_as(ctypes.c_void_p), 0, ctypes.byref(ms))
    assert rc == 0
    Tjj, Ljj = T[64:128, 64:128], L[64:128, 64:128]
    assert np.all(np.triu(Tjj, 1) == 0.0)
    scale = np.linalg.cond(Ljj)
    assert np.abs(Tjj @ Ljj - np.eye(64)).max() <= 64 * 2.3e-16 * scale * 8
    if below:
        assert np.abs(A[128:, 64:128] - L[128:, 64:128]).max() <= 64 * 2.3e-16 * scale * 8 * np.abs(L).max()
    # nothing outside block column 1 is touched by the panel workgroups
    assert np.array_equal(A[:, :64], L[:, :64])


@pytest.mark.parametrize("N", [192, 448, 1088, 2496, 4352])
def test_factorisation_identities_across_outer_block_layouts(gpu, N):
    """the whole factorisation at sizes whose outer-block layouts differ (chol_block_bounds: one block up to ~2400 columns, then several
    of growing width; N = 1088 and up also take the two-stream split): K W = I and K v = y to rounding, LOOCV error from the getters"""
    from gaussian_process_liouville_equation_amd import _capi as c
    from tests import parity
    X, y, _ = parity.synthetic_real(N, 8, 777 + N)
    fit = gpu.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 3)
    assert fit.scalars["info"] == 0
    K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    assert n1(K @ W - np.eye(N)) <= 50 * N * parity.EPS * n1(K) * n1(W)
    assert np.abs(K @ v - ys).max() <= 50 * N * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max())
    assert abs(((v / np.diag(W)) ** 2).sum() - fit.scalars["error"]) <= 1e-9 * fit.scalars["error"]
    fit.release()


@pytest.mark.parametrize("pend", [0, 1])
def test_panel_step_reports_a_non_positive_pivot(gpu, pend):
    """reference behaviour (LDLT::info() is never checked, kernel.cpp:281-283): the factorisation does not stop at a non-positive pivot;
    here NaN runs through the block's results and info names the first offending column (1-based, in matrix coordinates)"""
    lib = gpu.lib
    lib.gple_debug_potrf_step.restype = ctypes.c_int
    n, bad = 192, 37
    rng = np.random.default_rng(3 + pend)
    B = rng.standard_normal((n, 2 * n))
    K = B @ B.T / (2 * n) + 0.5 * np.eye(n)
    L = np.linalg.cholesky(K)
    A = K.copy()
    A[:, :64] = L[:, :64]
    S = K[64:, 64:] - L[64:, :64] @ L[64:, :64].T  # what block column 1 factors
    # make the pivot of column `bad` of the diagonal block negative: lower that diagonal entry below what the first `bad` columns take away
    Ld = np.linalg.cholesky(S[:64, :64])
    drop = (Ld[bad, bad] ** 2) * 1.5
    if pend:
        A[64 + bad, 64 + bad] -= drop
    else:
        A[64:, 64:] = S
        A[64 + bad, 64 + bad] -= drop
    A = np.asfortranarray(A)
    T = np.zeros((n, n), order="F")
    stamps = np.zeros(24, dtype=np.int64)
    ms = ctypes.c_float()
    rc = lib.gple_debug_potrf_step(gpu.ctx, A.ctypes.data_as(ctypes.c_void_p), T.ctypes.data_as(ctypes.c_void_p), pend, 64,
                                   stamps.ctypes.data_as(ctypes.c_void_p), 0, ctypes.byref(ms))
    assert rc == 0
    assert stamps[23] == 64 + bad + 1
    assert np.isnan(T[64:128, 64:128]).any() and np.isnan(A[128:, 64:128]).any()
    lo = bad // 16 * 16  # NaN spreads through the 16 x 16 tile products of its own sub-panel; the rows above it never see it
    assert np.all(np.isfinite(T[64:64 + lo, 64:64 + lo]))


@pytest.mark.parametrize("pre", [0, 1, 2, 3, 4, 5])
def test_side_stream_lands_on_its_own_hardware_queue(pre):
    """The fit's side stream (block-row inverse beside the panels, matrices of more than one outer block: n >= 2560) must not share the main stream's hardware queue.  HIP binds
    streams to GPU_MAX_HW_QUEUES (4) queues by use count as they are created, so whether a fresh stream collides depends on how many streams
    the process holds already — here `pre` of them are created (and used) first, and for every count the context must end up with a side stream
    whose probe kernels ran beside the main stream's (csrc/gple_chol.hip, pick_side_stream).  Measured without the probing: N = 4096 fit
    2.0 -> 2.4 / 3.9 ms on a collision (profiles/r03_notes.md)."""
    import torch

    import gaussian_process_liouville_equation_amd as pkg
    from tests.test_gpu_configs import config_inputs, THETA_R
    keep = []
    for _ in range(pre):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            keep.append(torch.zeros(16, device="cuda") + 1)
        keep.append(s)
    torch.cuda.synchronize()
    api = pkg.open_api(0)
    try:
        X, y, _, _ = config_inputs(4096, 8, 1)
        f = api.real_fit(THETA_R, X, y, 3)
        assert f.scalars["info"] == 0
        attempts, overlaps = ctypes.c_int(), ctypes.c_int()
        assert api.lib.gple_debug_side_stream(api.ctx, ctypes.byref(attempts), ctypes.byref(overlaps)) == 0
        assert 1 <= attempts.value <= 8 and overlaps.value == 1, (pre, attempts.value, overlaps.value)
        f.release()
    finally:
        api.close()


def _fit_in_own_process(env, sizes):
    """v, error and the fit's identities from a fit in a process of its own (the factorisation's knobs are read once per process)"""
    import os, subprocess, sys, tempfile
    from tests.conftest import ROOT
    out = tempfile.mktemp(suffix=".npz")
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import gaussian_process_liouville_equation_amd as pkg
from gaussian_process_liouville_equation_amd import _capi as c
from tests import parity
api = pkg.open_api(0)
res = {}
for N in %r:
    X, y, _ = parity.synthetic_real(N, 8, 4242 + N)
    fit = api.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 3)
    assert fit.scalars["info"] == 0
    K, W, v, ys = fit.get(c.R_KERNEL), fit.get(c.R_INVERSE), fit.get(c.R_INVLBL), fit.get(c.R_LABEL)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    assert n1(K @ W - np.eye(N)) <= 50 * N * parity.EPS * n1(K) * n1(W)
    assert np.abs(K @ v - ys).max() <= 50 * N * parity.EPS * (n1(K) * np.abs(v).max() + np.abs(ys).max())
    res["v%%d" %% N], res["e%%d" %% N] = v, fit.scalars["error"]
    fit.release()
np.savez(%r, **res)
print("ok")
''' % (ROOT, list(sizes), out)
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "ok" in res.stdout, (res.stdout[-500:], res.stderr[-2000:])
    data = dict(np.load(out))
    os.remove(out)
    return data


def test_one_launch_per_outer_block_against_one_launch_per_panel():
    """gple_chol.hip: the factorisation's default scheme (potrf_dag_kernel: the panels of an outer block in one launch, tile tasks from a work queue,
    hand-over by flags) and the scheme of rounds 2-3 (GPLE_CHOL_SCHEME=step: a launch per panel) are two summation orders of the same
    factorisation: each satisfies the fit's identities, and v = K^-1 y agrees to what the conditioning allows.  Sizes: one outer block without /
    with a fork of the inverse, several outer blocks (4096)."""
    sizes = (512, 1024, 2304, 4096)
    a = _fit_in_own_process({"GPLE_CHOL_SCHEME": "dag"}, sizes)
    b = _fit_in_own_process({"GPLE_CHOL_SCHEME": "step"}, sizes)
    for N in sizes:
        va, vb = a["v%d" % N], b["v%d" % N]
        assert np.abs(va - vb).max() <= 1e-7 * np.abs(vb).max(), N
        assert abs(a["e%d" % N] - b["e%d" % N]) <= 1e-7 * b["e%d" % N], N


def test_work_queue_makes_progress_with_one_worker_workgroup():
    """the tile tasks of potrf_dag_kernel are handed out in dependency order from one counter, so the launch must complete with ANY number of its
    workgroups running (the situation of a chip shared with other work): here one worker workgroup beside the spine (GPLE_CHOL_DAG_BLOCKS=2) —
    4096 / 64 = 64 panels, four launches, about 60 ms instead of 2"""
    _fit_in_own_process({"GPLE_CHOL_DAG_BLOCKS": "2"}, (1024, 4096))


def test_repeated_fits_agree_bit_for_bit(gpu):
    """every tile of the factorisation is summed in a fixed order whatever the timing of its hand-overs (potrf_dag_kernel), so repeated fits of the same
    inputs must agree BIT FOR BIT — any difference is a race (a tile read before it was final, a stale line).  Sizes interleaved, so that the words of the
    never-cleared flag buffer change their meaning from fit to fit; one- and several-block matrices, real and complex (probes/dag_soak.py: 49 500 fits, alone
    and beside another process, none differed)."""
    from gaussian_process_liouville_equation_amd import _capi as c
    from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C
    cases = [(False, N) for N in (256, 640, 1024, 2048, 3072, 4096)] + [(True, 256), (True, 1024)]
    inputs = {k: config_inputs(k[1], 8, 1, cplx=k[0]) for k in cases}
    ref = {}
    for r in range(12):
        order = list(cases)
        np.random.default_rng(r).shuffle(order)
        for k in order:
            X, y, _, _ = inputs[k]
            f = (gpu.complex_fit if k[0] else gpu.real_fit)(THETA_C if k[0] else THETA_R, X, y, 3)
            sig = (f.scalars["info"], f.scalars["error"], f.get(c.C_INVLBL if k[0] else c.R_INVLBL).tobytes())
            f.release()
            assert sig[0] == 0
            assert ref.setdefault(k, sig) == sig, (k, r)


def test_flag_epochs_start_over():
    """the flags of the one-launch factorisation are never cleared between fits, every fit takes a new epoch; the count starts over (buffer cleared in
    stream order) long before it could run out — here after every third fit (GPLE_CHOL_DAG_EPOCH_LIMIT=3): twelve fits of changing sizes, each
    with the fit's identities"""
    _fit_in_own_process({"GPLE_CHOL_DAG_EPOCH_LIMIT": "3"}, (1024, 4096, 512, 2304, 4096, 1024, 256, 4096, 3072, 1024, 2560, 512))


def test_inverse_in_the_launch_against_the_merge_trees():
    """T = L^-1 below the diagonal blocks comes from tile tasks of the panel launch (one-block matrices: all of it; larger ones: the diagonal part of every row
    block behind the first fork) or from merge trees of GEMMs (GPLE_CHOL_DAG_INVERSE=0, GPLE_CHOL_DAG_BLOCK_INVERSE=0): two summation orders, the same v"""
    sizes = (256, 1024, 3072, 4096, 6144)
    a = _fit_in_own_process({}, sizes)
    b = _fit_in_own_process({"GPLE_CHOL_DAG_INVERSE": "0", "GPLE_CHOL_DAG_BLOCK_INVERSE": "0"}, sizes)
    for N in sizes:
        assert np.abs(a["v%d" % N] - b["v%d" % N]).max() <= 1e-7 * np.abs(b["v%d" % N]).max(), N
        assert abs(a["e%d" % N] - b["e%d" % N]) <= 1e-7 * b["e%d" % N], N
