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


# ---- a give-up of the one-launch factorisation must end in the correct result ---------------------------
def _knobs(api, scheme=-1, poll_limit=-1, dag_blocks=-1):
    """gple_debug_chol_knobs: per-context scheme (0 step, 1 dag, 2 environment), poll limit and workgroup count; returns (giveups, recoveries)"""
    g, r = ctypes.c_long(), ctypes.c_long()
    api.lib.gple_debug_chol_knobs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
    assert api.lib.gple_debug_chol_knobs(api.ctx, scheme, poll_limit, dag_blocks, ctypes.byref(g), ctypes.byref(r)) == 0
    return g.value, r.value


def _fit_signature(api, cplx, theta, X, y, Xs, flags=3, defer=False):
    from gaussian_process_liouville_equation_amd import _capi as c
    f = (api.complex_fit if cplx else api.real_fit)(theta, X, y, flags, defer_scalars=defer)
    p = (api.complex_predict if cplx else api.real_predict)(f, Xs)  # host pointers: drains the stream
    sc = dict(f.scalars)
    sig = {"v": f.get(c.C_INVLBL if cplx else c.R_INVLBL).tobytes(), "mean": p["prediction"].tobytes(), "var": p["variance"].tobytes(), "cut": p["cutoff"].tobytes(),
           "error": sc["error"], "purity": sc["purity"], "info": sc["info"]}
    if flags & 4:
        sig["derr"] = np.asarray(sc["error_derivative"]).tobytes()
        sig["dpur"] = np.asarray(sc["purity_derivative"]).tobytes()
    f.release()
    return sig


@pytest.mark.parametrize("cplx,N,flags", [(False, 600, 3), (False, 1500, 7), (False, 4200, 3), (True, 700, 3), (True, 300, 7)])
def test_a_give_up_of_the_one_launch_factorisation_is_recovered_bit_for_bit(cplx, N, flags):
    """potrf_dag_kernel waits on flags with a bounded number of polls; a wave that gives up sets info = -1 and leaves T unfinished.  The host must
    then end in the CORRECT result: the same fit repeated with one launch per panel (recover_fit).  The give-up is forced deterministically: one worker
    workgroup and a poll limit of 2 — the first tile task waits for the spine's T_0 for microseconds and gives up at once.  The recovered fit must
    equal, bit for bit, the fit of a context that runs the launch-per-panel scheme from the start: scalars, weights, derivative members, and a predict
    (mean, variance, cut-off) on a handful of points.  Sizes: one outer block with the inverse inside the launch (n = 768, 1536), several launches and
    the side stream (n = 4352), complex (n = 1536, 768), with and without derivative members (W and dv are rebuilt from the repeated factor)."""
    import gaussian_process_liouville_equation_amd as pkg
    from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C
    X, y, _, _ = config_inputs(N, 8, 1, cplx=cplx)
    Xs = X[:: max(1, N // 40)] + 0.05
    theta = THETA_C if cplx else THETA_R
    ref_api, bad_api = pkg.open_api(0), pkg.open_api(0)
    try:
        _knobs(ref_api, scheme=0)
        ref = _fit_signature(ref_api, cplx, theta, X, y, Xs, flags)
        assert ref["info"] == 0 and np.isfinite(ref["error"])
        _knobs(bad_api, scheme=1, poll_limit=2, dag_blocks=2)
        # (a) scalars requested at creation: the create call itself notices and recovers
        got = _fit_signature(bad_api, cplx, theta, X, y, Xs, flags)
        g, r = _knobs(bad_api)
        assert g >= 1 and r >= 1, "the give-up was not provoked: the test proves nothing"
        assert got == ref
        # (b) deferred scalars: the first call that drains the stream — here the predict with host pointers — notices, recovers and runs again
        got = _fit_signature(bad_api, cplx, theta, X, y, Xs, flags, defer=True)
        assert _knobs(bad_api)[1] >= r + 1
        assert got == ref
        # the same context with sane knobs again: the one-launch scheme's own result, no further recovery
        _knobs(bad_api, scheme=2, poll_limit=0, dag_blocks=0)
        before = _knobs(bad_api)
        dag = _fit_signature(bad_api, cplx, theta, X, y, Xs, flags)
        assert _knobs(bad_api) == before and dag["info"] == 0
        assert abs(dag["error"] - ref["error"]) <= 1e-7 * abs(ref["error"])
    finally:
        ref_api.close()
        bad_api.close()


def test_work_enqueued_before_the_give_up_was_noticed_is_nan_and_reported():
    """A fit created without a scalars struct and a predict with DEVICE pointers only enqueue work: nobody has looked at the factorisation when the
    predict runs.  If it had given up, the predict's outputs must be NaN (never numbers from the unfinished factor), the next draining call on the fit
    (gple_real_fit_get_scalars) must recover the fit AND say that earlier work has to be repeated (GPLE_ERR_TIMEOUT), and the repeated predict must
    then equal the launch-per-panel result bit for bit."""
    import torch

    import gaussian_process_liouville_equation_amd as pkg
    from gaussian_process_liouville_equation_amd import _capi as c
    from tests.test_gpu_configs import config_inputs, THETA_R
    N = 1500
    X, y, _, _ = config_inputs(N, 8, 1)
    Xs = np.ascontiguousarray(X[::30] + 0.05)
    M = len(Xs)
    ref_api, api = pkg.open_api(0), pkg.open_api(0)
    try:
        _knobs(ref_api, scheme=0)
        ref = _fit_signature(ref_api, False, THETA_R, X, y, Xs)
        _knobs(api, scheme=1, poll_limit=2, dag_blocks=2)
        f = api.real_fit(THETA_R, X, y, 3, defer_scalars=True)
        dXs = torch.from_numpy(Xs).cuda()
        out = torch.zeros(3, M, dtype=torch.float64, device="cuda")
        dp = lambda t: ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_double))
        ps = c.PredictScalars()

        def predict():
            return api.lib.gple_real_predict(api.ctx, f.handle, dp(dXs), M, c.IO_DEVICE, None, dp(out[0]), dp(out[1]), dp(out[2]), ctypes.byref(ps))

        assert predict() == 0
        api.synchronize()
        assert bool(torch.isnan(out).all()), "a predict on a given-up factor returned numbers"
        sc = c.RealFitScalars()
        st = api.lib.gple_real_fit_get_scalars(f.handle, ctypes.byref(sc))
        assert st == c.GPLE_ERR_TIMEOUT, st
        assert b"repeat" in api.lib.gple_ctx_last_error(api.ctx)
        assert api.lib.gple_real_fit_get_scalars(f.handle, ctypes.byref(sc)) == 0  # the fit itself is good now
        assert sc.info == 0 and sc.error == ref["error"] and sc.purity == ref["purity"]
        assert predict() == 0
        api.synchronize()
        got = out.cpu().numpy()
        assert got[0].tobytes() == ref["mean"] and got[1].tobytes() == ref["var"] and got[2].tobytes() == ref["cut"]
        f.release()
    finally:
        ref_api.close()
        api.close()


def test_objective_and_nlml_recover_from_a_give_up():
    """loose_function (fit + predict of the extra points enqueued in one go, one synchronisation) and the NLML path must return the launch-per-panel
    values when the one-launch factorisation gives up under them — not NaN mapped to DBL_MAX by make_normal, not numbers from an unfinished factor."""
    import gaussian_process_liouville_equation_amd as pkg
    from tests import parity
    from tests.test_gpu_configs import config_inputs, THETA_R, THETA_C
    ref_api, api = pkg.open_api(0), pkg.open_api(0)
    try:
        _knobs(ref_api, scheme=0)
        _knobs(api, scheme=1, poll_limit=2, dag_blocks=2)
        for cplx, N in ((False, 900), (True, 400)):
            X, y, _, _ = config_inputs(N, 8, 1, cplx=cplx)
            rng = np.random.default_rng(3)
            Xe = X[np.arange(2 * N) % N] + rng.normal(0, 0.3, size=(2 * N, 2))
            ye = (np.exp(-0.5 * (((Xe[:, 0] + 10) / 0.7086) ** 2 + ((Xe[:, 1] - 14.112) / 0.7056) ** 2)) / (2 * np.pi * 0.7086 * 0.7056)).astype(complex)
            theta = THETA_C if cplx else THETA_R
            r0 = _knobs(api)[1]
            vr, gr = ref_api.loose_function(theta, X, np.asarray(y, dtype=complex), Xe, ye)
            v, g = api.loose_function(theta, X, np.asarray(y, dtype=complex), Xe, ye)
            assert _knobs(api)[1] > r0, "the give-up was not provoked"
            assert np.isfinite(vr) and vr < 1e300
            assert v == vr and g.tobytes() == gr.tobytes()
        X, y, _ = parity.synthetic_real(700, 8, 99)
        x = [0.05, 1.3, 1.0 / 0.7086, 1.0 / 0.7056]
        r0 = _knobs(api)[1]
        vr, gr = ref_api.nlml(x, X, y)
        v, g = api.nlml(x, X, y)
        assert _knobs(api)[1] > r0
        assert v == vr and np.asarray(g).tobytes() == np.asarray(gr).tobytes()
    finally:
        ref_api.close()
        api.close()
