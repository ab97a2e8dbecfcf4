"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke(), bench.py cpu_baseline)."""
import ctypes
import os

from gaussian_process_liouville_equation_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgple_oracle.so")


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make -C oracle` (or __graft_entry__.build())")
    lib = ctypes.CDLL(LIB_PATH)
    api = _capi.Api(lib, "oracle_", with_ctx=False)
    lib.oracle_num_threads.restype = ctypes.c_int
    api.num_threads = lib.oracle_num_threads()
    return api
