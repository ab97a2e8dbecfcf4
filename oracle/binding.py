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
    lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
    lib.oracle_set_num_threads(available_cpus())
    api.num_threads = lib.oracle_num_threads()
    return api


def available_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (and GPLE_ORACLE_THREADS)."""
    n = len(os.sched_getaffinity(0))
    try:  # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    n = min(n, 32)
    if os.environ.get("GPLE_ORACLE_THREADS"):
        n = max(1, int(os.environ["GPLE_ORACLE_THREADS"]))
    return n
