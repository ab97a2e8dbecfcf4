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
