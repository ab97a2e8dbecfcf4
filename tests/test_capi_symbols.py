"""CPU: the C-ABI library loads without a GPU and exports every function include/gple.h declares."""
import ctypes
import os
import re

from gaussian_process_liouville_equation_amd import _capi
from tests.conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gple.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gple_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 20
    assert sorted("gple_" + n for n in _capi.GPLE_SYMBOLS) == names


def test_hip_library_exports_every_declared_symbol():
    import gaussian_process_liouville_equation_amd as pkg

    lib = pkg.load_library()  # no HIP call happens at load time, so this works on a CPU-only machine
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.gple_status_string.restype = ctypes.c_char_p
    assert lib.gple_status_string(0) == b"ok"


def test_oracle_exports_the_mirror_interface():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libgple_oracle.so"))
    for name in ["real_gram", "cutoff_factor", "real_fit_create", "real_fit_release", "real_fit_get", "real_predict",
                 "complex_fit_create", "complex_fit_release", "complex_fit_get", "complex_predict", "loose_function", "nlml",
                 "nlml_predict", "nlml_cross", "nlml_cross_predict", "complex_gram"]:
        assert hasattr(lib, "oracle_" + name), name


def test_product_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "gaussian_process_liouville_equation_amd")
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(base, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "libgple_oracle" not in text and "gple_oracle.h" not in text, os.path.join(base, f)
