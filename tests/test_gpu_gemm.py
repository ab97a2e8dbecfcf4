"""GPU: the fp64 MFMA GEMM family of the fit (csrc/gple_gemm.hip) on its own, through the library's diagnostic entry
gple_debug_gemm (not part of include/gple.h): every tile kernel (32 = split-k, 64 = 4-slab ring, 128 = double-buffered), every
operand layout it is instantiated for, the triangular k-ranges of the merge tree / T^T T, lower-only results and beta != 0,
against numpy in double precision.  The fits exercise these kernels only at the shapes a fit produces."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K_FULL, K_GE_N, K_LE_M, K_GE_MAX_MN = 0, 1, 2, 3


def _gemm(gpu, A, B, C, ak, bk, ct, alpha, beta, krange, lower_only, tile):
    """A: (M, K) logical, B: (N, K) logical, C: (M, N) logical; stored per the layout flags"""
    lib = gpu.lib
    if not hasattr(lib, "gple_debug_gemm"):
        pytest.fail("libgple_hip.so lacks gple_debug_gemm")
    lib.gple_debug_gemm.restype = ctypes.c_int
    M, K = A.shape
    N = B.shape[0]
    As = np.ascontiguousarray(A) if ak else np.asfortranarray(A)  # kmajor: element (r, k) at k + r * ld
    Bs = np.ascontiguousarray(B) if bk else np.asfortranarray(B)
    Cs = np.ascontiguousarray(C) if ct else np.asfortranarray(C)  # c_trans: C(m, n) at n + m * ldc
    Cs = Cs.copy(order="C" if ct else "F")
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = lib.gple_debug_gemm(gpu.ctx, p(As), ctypes.c_long(K if ak else M), int(ak), p(Bs), ctypes.c_long(K if bk else N), int(bk), p(Cs),
                             ctypes.c_long(N if ct else M), int(ct), M, N, K, ctypes.c_double(alpha), ctypes.c_double(beta), krange, int(lower_only), tile)
    assert rc == 0, rc
    return np.array(Cs)


def _reference(A, B, C, alpha, beta, krange, lower_only, tile):
    M, K = A.shape
    N = B.shape[0]
    Am, Bm = A.copy(), B.copy()
    k = np.arange(K)
    if krange == K_GE_N:  # B(n, k) != 0 only for k >= n
        Bm = Bm * (k[None, :] >= np.arange(N)[:, None])
    elif krange == K_LE_M:  # A(m, k) != 0 only for k <= m
        Am = Am * (k[None, :] <= np.arange(M)[:, None])
    elif krange == K_GE_MAX_MN:
        Am = Am * (k[None, :] >= np.arange(M)[:, None])
        Bm = Bm * (k[None, :] >= np.arange(N)[:, None])
    out = alpha * (Am @ Bm.T) + beta * C
    if lower_only:  # tiles strictly above the block diagonal are not computed: compare the lower tiles only
        bt = 32 if tile == 32 else (128 if tile == 128 else 64)
        mi, ni = np.arange(M)[:, None] // bt, np.arange(N)[None, :] // bt
        return out, ni <= mi
    return out, np.ones((M, N), bool)


CASES = [
    # tile, ak, bk, ct, M, N, K, krange, lower_only, beta
    (32, False, False, False, 192, 128, 64, K_FULL, False, 0.0),     # K = 64 trailing update shape
    (32, False, False, False, 256, 256, 64, K_FULL, True, 1.0),      # ... as the factorisation calls it
    (32, False, True, False, 128, 128, 128, K_GE_N, False, 0.0),     # W = L21 T11
    (32, False, True, False, 256, 256, 256, K_LE_M, False, 0.0),     # T21 = -T22 W
    (32, False, True, False, 96, 160, 48, K_FULL, False, 0.5),       # ragged in 32s, fewer k-groups than waves
    (64, False, False, False, 256, 192, 256, K_FULL, True, 1.0),     # outer update
    (64, False, True, False, 256, 256, 256, K_LE_M, False, 0.0),
    (64, False, True, False, 256, 256, 256, K_GE_N, False, 0.0),
    (64, True, True, False, 256, 256, 256, K_GE_MAX_MN, True, 0.0),  # T^T T
    (64, False, False, True, 128, 192, 80, K_FULL, False, 0.0),
    (64, False, True, True, 128, 128, 128, K_LE_M, False, 0.0),      # few-rows predict
    (128, False, False, False, 256, 384, 144, K_FULL, False, 0.0),   # derivative products
    (128, True, True, False, 256, 256, 256, K_GE_MAX_MN, True, 0.0),
    (128, False, True, False, 256, 256, 256, K_LE_M, False, 2.0),
    # more shapes of the 128-tile kernel (round 4: one and two slabs, k-ranges with row-contiguous operands, lower tiles, the XCD-aware tile order)
    (128, False, False, True, 256, 128, 208, K_FULL, False, 0.0),
    (128, False, False, False, 128, 256, 16, K_FULL, False, 1.5),    # one slab
    (128, False, False, False, 384, 128, 32, K_FULL, False, 0.0),    # two slabs
    (128, False, False, False, 384, 384, 384, K_LE_M, False, 0.0),   # k-range by row block
    (128, False, False, False, 384, 384, 384, K_GE_N, False, 0.0),   # k-range from the column block on
    (128, False, False, False, 512, 512, 96, K_FULL, True, 1.0),     # lower tiles only
    (128, False, False, False, 1024, 1024, 512, K_FULL, False, 0.0), # the XCD-aware tile order (8 | N-tiles)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "t%d_%s%s%s_%dx%dx%d_kr%d_lo%d" % (c[0], "k" if c[1] else "r", "k" if c[2] else "r", "t" if c[3] else "n", c[4], c[5], c[6], c[7], c[8]))
def test_gemm_tile_kernels(gpu, case):
    tile, ak, bk, ct, M, N, K, krange, lower_only, beta = case
    rng = np.random.default_rng(M + 7 * N + 13 * K + tile)
    A, B, C = rng.standard_normal((M, K)), rng.standard_normal((N, K)), rng.standard_normal((M, N))
    if krange in (K_GE_N, K_GE_MAX_MN):
        B = B * (np.arange(K)[None, :] >= np.arange(N)[:, None])  # the operand really is triangular (the kernel may skip or not)
    if krange in (K_LE_M,):
        A = A * (np.arange(K)[None, :] <= np.arange(M)[:, None])
    if krange == K_GE_MAX_MN:
        A = A * (np.arange(K)[None, :] >= np.arange(M)[:, None])
    alpha = -1.0 if lower_only else 0.75
    got = _gemm(gpu, A, B, C, ak, bk, ct, alpha, beta, krange, lower_only, tile)
    ref, mask = _reference(A, B, C, alpha, beta, krange, lower_only, tile)
    scale = np.abs(A) @ np.abs(B).T * abs(alpha) + abs(beta) * np.abs(C)
    err = np.abs(got - ref)[mask] / np.maximum(scale[mask], 1e-300)
    assert err.max() <= 4 * 2.3e-16 * np.sqrt(K), err.max()
    if lower_only:  # tiles that are not computed keep their input
        assert np.array_equal(got[~mask], C[~mask])


def test_gemm_split_k_is_deterministic(gpu):
    """the four waves' partial sums meet in LDS in a fixed order: two runs agree bit for bit"""
    rng = np.random.default_rng(3)
    A, B, C = rng.standard_normal((128, 512)), rng.standard_normal((128, 512)), np.zeros((128, 128))
    r1 = _gemm(gpu, A, B, C, False, True, False, 1.0, 0.0, K_FULL, False, 32)
    r2 = _gemm(gpu, A, B, C, False, True, False, 1.0, 0.0, K_FULL, False, 32)
    assert np.array_equal(r1, r2)

