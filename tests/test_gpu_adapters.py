"""GPU: the header-only C++ adapters (host/kernel.h, complex_kernel.h, predict.h — the reference's class names and
signatures on top of the C-ABI) give the same numbers as the Python path.  The driver is built by __graft_entry__.build()."""
import os
import subprocess

import numpy as np
import pytest

from gaussian_process_liouville_equation_amd import kernels as K
from tests import parity
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cpp_adapters_match_python_path(gpu, tmp_path):
    exe = os.path.join(ROOT, "tests", "cpp", "adapter_driver")
    if not os.path.exists(exe):
        pytest.fail("tests/cpp/adapter_driver missing: run __graft_entry__.build()")
    X, yr, Xs = parity.synthetic_real(150, 77, 51)
    y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0))
    path = tmp_path / "in.txt"
    with open(path, "w") as f:
        f.write(f"{len(X)} {len(Xs)}\n")
        for (a, b), z in zip(X, y):
            f.write("%.17g %.17g %.17g %.17g\n" % (a, b, z.real, z.imag))
        for a, b in Xs:
            f.write("%.17g %.17g\n" % (a, b))
    out = subprocess.run([exe, str(path)], check=True, capture_output=True, text=True, timeout=120).stdout
    got = {l.split()[0]: np.array(list(map(float, l.split()[1:]))) for l in out.strip().splitlines()}
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    ctheta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05]
    k = K.TrainingKernel(theta, (X, y), True, True, True, api=gpu)
    p = K.PredictiveKernel(Xs, k, False)
    close = lambda a, b: np.allclose(a, b, rtol=1e-11, atol=1e-13)
    assert close(got["real_error"], k.get_error()) and close(got["real_copy_error"], k.get_error())
    assert close(got["real_population"], k.get_population()) and close(got["real_purity"], k.get_purity())
    assert np.allclose(got["real_error_derivative"], k.get_error_derivative(), rtol=1e-9, atol=1e-9)
    assert close(got["real_cut_sum"], p.get_cutoff_prediction().sum()) and close(got["real_var_sum"], p.get_variance().sum())
    assert close(got["real_one_point"], p.get_cutoff_prediction()[0])
    ck = K.TrainingComplexKernel(ctheta, (X, y), True, True, False, api=gpu)
    cp = K.PredictiveComplexKernel(Xs, ck, False)
    assert close(got["complex_error"], ck.get_error()) and close(got["complex_purity"], ck.get_purity())
    assert close(got["complex_cut_abs_sum"], np.abs(cp.get_cutoff_prediction()).sum()) and close(got["complex_var_sum"], cp.get_variance().sum())
    assert close(got["all_population"], k.get_population()) and close(got["all_purity"], k.get_purity() + 2 * ck.get_purity())
    assert got["all_has_11"][0] == 0
    ye = np.array([0.01 * (i % 7) for i in range(len(Xs))], dtype=complex)
    grad = [0.0] * 4
    val = K.loose_function(theta, grad, ((X, y), (Xs, ye)), api=gpu)
    assert close(got["loose_value"], val) and close(got["loose_value_nograd"], val)
    assert np.allclose(got["loose_grad"], grad, rtol=1e-9, atol=1e-9)
