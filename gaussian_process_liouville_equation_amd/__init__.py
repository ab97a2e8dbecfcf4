"""MI355X-native GPR fit + predict hot path of kaigu1997/gaussian_process_liouville_equation.

The compute lives in csrc/ (hand-written HIP for gfx950 behind the C-ABI of include/gple.h); this package is the
thin Python host layer: a ctypes binding (`_capi`) and mirrors of the reference's kernel classes (`kernels`).
There is NO CPU fallback: if libgple_hip.so is missing or cannot be loaded every entry point raises ImportError.
"""
import ctypes
import os

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgple_hip.so")
_lib = None


def load_library():
    """dlopen csrc/libgple_hip.so (built by __graft_entry__.build() / `make -C .../csrc`). Fails loudly."""
    global _lib
    if _lib is None:
        # PyTorch ships its own libamdhip64; whichever HIP runtime is mapped first serves the whole process, and torch
        # cannot see the GPU if the system runtime got in before it.  Import torch first when it is installed so the
        # library, torch tensors (device memory) and torch.distributed (RCCL) share one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: the HIP extension is not built "
                              "(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        try:
            _lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise ImportError(f"cannot load {LIB_PATH}: {e}; there is no CPU fallback") from e
    return _lib


def open_api(device=0, stream=None):
    """Create a context on `device` (optionally on an existing hipStream_t given as int) and return the bound Api."""
    return _capi.Api(load_library(), "gple_", with_ctx=True, device=device, stream=stream)
