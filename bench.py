#!/usr/bin/env python3
"""bench.py — GP fit + predict step (BASELINE.json metric) on 1..8 MI355X.

One step = what the reference does per output tick for one density-matrix element:
  TrainingKernel(theta, set, error=1, average=1, derivative=0)      (predict.cpp:390-393)
+ PredictiveKernel(grid, kernel, false): mean, variance, cut-off    (output.cpp:204-207)
Default workload C4r = the north-star size of BASELINE.json (N = 4096 samples, 512 x 512 grid, real SE kernel, fp64),
synthetic inputs of SURVEY.md §8(d) (seed 20240607 + 1).  Inputs are resident in HBM before the timed region.
N > 1 GPUs: strong scaling of the same step — every rank fits (replicated, no broadcast needed) and predicts its share of a
block-cyclic deal of the grid (128-point blocks, block b -> rank b mod P: gple_*_predict_sharded); the shares are all-gathered over RCCL
inside the library (the north-star partition).  `python bench.py --gpus N` starts its N ranks itself (launch_ranks) when it is not already
running under torch.distributed.run.
`--workload C4opt` is the optimiser's inner loop instead (opt.cpp:441-482): one step = loose_function value + gradient of
the three density-matrix elements of a 2-state system (2 real + 1 complex GP, N = 4096, 5N extra points each).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # name: (N, G, kernel)
    "C1": (256, 128, "real"),       # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "C2": (1024, 256, "real"),      # configs[1]
    "C3": (2048, 256, "complex"),   # configs[2]: off-diagonal density-matrix element
    "C4r": (4096, 512, "real"),     # one real element of configs[3] = the north-star target size: the default
    "C4c": (4096, 512, "complex"),  # the complex element of C4
    "C5r": (8192, 1024, "real"),    # one real element of C5
    "C4": (4096, 512, "elements2"),  # configs[3] whole: the 3 elements of a 2-state density matrix (real, complex, real), hybrid element x grid plan
    "C5": (8192, 1024, "elements3"), # configs[4]'s GP side: the 6 elements of a 3-state density matrix (3 real + 3 complex)
    "C4opt": (4096, 0, "opt"),      # the opt.cpp loop of configs[3]: 2 real + 1 complex objective evaluations with gradient
    "C2step": (1024, 0, "step"),    # one tick of main.cpp:143-176 at N = 1024: evolve density + 5N extra points, refit 3 elements
    "C5step": (8192, 0, "step"),    # the same at the N of configs[4] (two-level physics: what the reference instantiates)
    "C2step3": (1024, 0, "step3"),  # one tick of a THREE-level system (6 elements, 36 back-propagated predicts per point: gple_evolve_n) at N = 1024
    "C5step3": (8192, 0, "step3"),  # configs[4] as stated: 3-state PES, full step loop, N = 8192
    # the log-marginal-likelihood formulation north_star names (test/gpr.cpp:499-532, 654-706), at the N of configs[1] / configs[3]:
    # one step = NLML value + gradient, then the mean-only prediction on a 256 x 256 grid (each builds and factors K, like the reference's two functions)
    "NLML1024": (1024, 256, "nlml"),
    "NLML4096": (4096, 256, "nlml"),
}
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak; measured 78.4 with v_mfma_f64_16x16x4_f64
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")
# density-matrix elements in the reference's order (lower triangle, row-major: storage.h / predict.cpp:290-360)
ELEMENT_KINDS = {"elements2": ["real", "complex", "real"], "elements3": ["real", "complex", "real", "complex", "complex", "real"]}


def synthetic(N, G, seed, kernel="real"):
    rng = np.random.Generator(np.random.PCG64(seed))
    x0, p0, sx, sp = -10.0, 14.112, 0.7086, 0.7056
    X = rng.normal([x0, p0], [sx, sp], size=(N, 2))
    y = np.exp(-0.5 * (((X[:, 0] - x0) / sx) ** 2 + ((X[:, 1] - p0) / sp) ** 2)) / (2 * np.pi * sx * sp)
    G = max(G, 1)
    dx = 40.0 / G
    xs = -20.0 + dx * np.arange(G)
    ps = (p0 - np.pi / (2 * dx)) + (np.pi / dx / G) * np.arange(G)
    gx, gp = np.meshgrid(xs, ps, indexing="ij")  # point index = ix * G + ip (input.cpp:37-70)
    grid = np.ascontiguousarray(np.stack([gx.ravel(), gp.ravel()], axis=1))
    if kernel == "complex":  # complex labels and the reference's initial complex parameters (opt.cpp:306-332)
        y = 0.5 * y * np.exp(0.5j * (X[:, 0] - x0))
        return X, y, grid, np.array([1.0, 1.0, sx, sp, 1.0, sx, sp, 1e-2])
    return X, y, grid, np.array([1.0, sx, sp, 1e-2])


def extra_points(X, seed, kernel):
    """validation set of the opt loop (mc.cpp:59-94): 5N points r_{i mod N} + N(0, std(r)^2), exact labels"""
    rng = np.random.Generator(np.random.PCG64(seed))
    x0, p0, sx, sp = -10.0, 14.112, 0.7086, 0.7056
    N = len(X)
    Xe = X[np.arange(5 * N) % N] + rng.normal(0.0, X.std(axis=0), size=(5 * N, 2))
    ye = np.exp(-0.5 * (((Xe[:, 0] - x0) / sx) ** 2 + ((Xe[:, 1] - p0) / sp) ** 2)) / (2 * np.pi * sx * sp)
    if kernel == "complex":
        ye = 0.5 * ye * np.exp(0.5j * (Xe[:, 0] - x0))
    return Xe, ye.astype(complex)


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def traffic_for(workload, world):
    """HBM bytes of one rownorm_kernel launch: PMC counters cannot be read from inside the run, so the value committed under
    profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled as the gfx950 guide prescribes)
    is reported when it was taken on this workload at this GPU count; `source` says which commit's kernels it was taken on."""
    try:
        tj = json.load(open(TRAFFIC_FILE))
        e = tj.get(workload)
        if e and world == 1:
            return e["hbm_bytes_per_launch"], {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "git": e.get("git"), "kernel_src_sha16": e.get("kernel_src_sha16")}
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def cpu_baseline(workload, N, G, kernel, X, y, grid, theta, runs=3):
    """The oracle (CPU restatement of the reference's algorithm: materialised K*, pivoted LDLT, explicit inverse, per-row
    row W row^T variance loop) on this box's host cores, bounded sample: full fit + a slice of the grid extrapolated linearly
    in M (rows are independent), median of `runs`; plus the same slice on ONE core."""
    from oracle import binding
    ora = binding.load()
    M = len(grid)
    cplx = kernel == "complex"
    fit_fn = (lambda: ora.complex_fit(theta, X, y, 3)) if cplx else (lambda: ora.real_fit(theta, X, y, 3))
    pred_fn = ora.complex_predict if cplx else ora.real_predict
    # slice sized for ~3 s per run at this N on 16 threads (2 M N^2 flops real, 32 M N^2 complex; the port sustains ~70 GFLOP/s)
    m_s = int(min(M, max(256, 2e11 / ((16.0 if cplx else 1.0) * 2.0 * N * N))))
    tfs, tps, fo = [], [], None
    for _ in range(runs):
        t0 = time.perf_counter()
        fo = fit_fn()
        tfs.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        pred_fn(fo, grid[:m_s])
        tps.append((time.perf_counter() - t0) * (M / m_s))
    tf, tp = statistics.median(tfs), statistics.median(tps)
    # one core: a smaller slice of the predict, the fit only while it stays within seconds
    ora.lib.oracle_set_num_threads(1)
    m_1 = max(64, m_s // 16)
    t0 = time.perf_counter()
    pred_fn(fo, grid[:m_1])
    tp1 = (time.perf_counter() - t0) * (M / m_1)
    tf1 = None
    if N <= (1024 if cplx else 2048):
        t0 = time.perf_counter()
        fit_fn()
        tf1 = time.perf_counter() - t0
    ora.lib.oracle_set_num_threads(ora.num_threads)
    return {"value": round(1e3 * (tf + tp), 1), "unit": "ms/step", "cores": ora.num_threads, "kind": "port",
            "cpu_model": cpu_model(), "runs": runs, "fit_ms_median": round(1e3 * tf, 1), "predict_ms_median_scaled": round(1e3 * tp, 1),
            "one_core_ms": {"predict_scaled": round(1e3 * tp1, 1), "fit": None if tf1 is None else round(1e3 * tf1, 1),
                            "sample": f"predict on {m_1} of {M} grid points scaled to M" + ("" if tf1 is not None else "; fit not timed on one core at this N (minutes)")},
            "sample": f"oracle (CPU restatement of the reference's algorithm, OpenMP, {ora.num_threads} threads), median of {runs}: full fit N={N} "
                      f"({tf:.2f} s) + predict on {m_s} of {M} grid points scaled to M ({tp:.2f} s); workload {workload}"}


class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_RCCL = {}


def _rccl_load(torch, rank):
    """phase 1 of the communicator setup, local to a rank: librccl loaded RTLD_GLOBAL — torch's own copy when it ships one, so that the
    process holds ONE RCCL and libgple_hip.so finds ncclAllGather in the process image, i.e. in the library the communicator belongs to —
    and, on rank 0, the unique id"""
    if os.environ.get("BENCH_FORCE_COMM_FAIL") in ("1", f"rank{rank}"):  # rehearsal of the fallback: on every rank, or on ONE rank only
        raise RuntimeError("BENCH_FORCE_COMM_FAIL is set")
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"]
    path = next((q for q in cands if os.path.exists(q)), "librccl.so.1")
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)
    rccl.ncclGetErrorString.restype = C.c_char_p
    uid = _NcclUniqueId()
    if rank == 0:
        rc = rccl.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId: {rccl.ncclGetErrorString(rc).decode()}")
    return rccl, path, uid


def _all_ok(torch, dist, world, ok):
    """every rank enters this with its own verdict and leaves with the common one (MIN over the control group)"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if world > 1:
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


def try_rccl_comm(torch, dist, rank, world):
    """An ncclComm_t over all ranks for the library's own collective (gple_*_predict_sharded / _dealt), or (None, path, reason) on EVERY rank
    when any rank failed: the caller then falls back to torch.distributed's own RCCL group and says so in the JSON line — a scaling run that dies
    in communicator setup measures nothing.  Two phases, each closed by an agreement over the control group, so that the collectives on that
    group stay matched whichever rank fails where: (1) load librccl + rank 0 draws the unique id — local, under try; agree; (2) only if all ranks
    are ok: the 128 bytes travel by broadcast, every rank joins with ncclCommInitRank on its current device — under try; agree."""
    rccl, path, uid, why = None, None, None, ""
    try:
        rccl, path, uid = _rccl_load(torch, rank)
    except Exception as e:  # noqa: BLE001 - anything: missing library, symbol, ncclGetUniqueId error
        why = f"rank {rank}: {type(e).__name__}: {e}"
    comm = None
    if _all_ok(torch, dist, world, rccl is not None):
        if world > 1:  # every rank is here: the broadcast is matched
            buf = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
            if dist.get_backend() == "nccl":
                buf = buf.cuda()
            dist.broadcast(buf, 0)
            C.memmove(C.byref(uid), bytes(buf.cpu().numpy().tobytes()), 128)
        try:
            if os.environ.get("BENCH_FORCE_COMM_FAIL") == f"init{rank}":
                raise RuntimeError("BENCH_FORCE_COMM_FAIL is set (ncclCommInitRank)")
            comm = C.c_void_p()
            rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
            rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
            if rc != 0:
                comm = None
                raise RuntimeError(f"ncclCommInitRank(rank {rank} of {world}): {rccl.ncclGetErrorString(rc).decode()}")
            _RCCL["lib"] = rccl
        except Exception as e:  # noqa: BLE001
            comm, why = None, f"rank {rank}: {type(e).__name__}: {e}"
        if _all_ok(torch, dist, world, comm is not None):
            return comm, path, ""
        if comm is not None:
            destroy_rccl_comm(comm)
    sys.stderr.write(f"bench.py: the library-side RCCL communicator could not be created ({why or 'on another rank'}); falling back to torch.distributed\n")
    return None, path, why or "failed on another rank"


def destroy_rccl_comm(comm):
    rccl = _RCCL.get("lib")
    if rccl is not None and comm:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def launch_ranks(nproc, script_argv, out=None, timeout=None):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as child processes of THIS process — which has not touched
    the GPU (the call sits in front of `import torch`; nothing is exec'ed over a process that holds a GPU context) — through
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py ...`, i.e. the
    driver's own command line.  The children's stdout is collected: rank 0 prints the result line (the other ranks' stdout is pointed at
    stderr by _keep_stdout_for_the_json_line), anything else that reached stdout (a banner of a library) goes to stderr here, and exactly ONE
    JSON line leaves on stdout.  Returns the exit code: the launcher's if it failed (a rank that dies takes the others down through
    torch.distributed.run), 1 if no result line came, else 0."""
    import socket
    out = out or sys.stdout
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port)] + list(script_argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        sys.stderr.write(f"bench.py: the {nproc} ranks did not finish within {timeout} s\n")
        sys.stderr.write((e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""))
        return 124
    line = None
    for ln in proc.stdout.splitlines():
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            try:
                if "metric" in json.loads(t):
                    line = t
                    continue
            except ValueError:
                pass
        if t:
            sys.stderr.write(ln + "\n")
    if proc.returncode != 0:
        sys.stderr.write(f"bench.py: torch.distributed.run exited with {proc.returncode}\n")
        return proc.returncode
    if line is None:
        sys.stderr.write("bench.py: no result line from rank 0\n")
        return 1
    out.write(line + "\n")
    out.flush()
    return 0


def _keep_stdout_for_the_json_line():
    """RCCL prints a version banner on STDOUT when a communicator is created; the driver parses stdout as ONE JSON line.  File descriptor 1
    is pointed at stderr for the life of the process and the result line goes to a duplicate of the original stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def emit(result):
    OUT.write(json.dumps(result) + "\n")
    OUT.flush()


OUT = sys.stdout


def main():
    global OUT
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C4r", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prune", action="store_true",
                    help="time the library's default predict (far grid rows not contracted) as `value` instead of the full contraction")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1: nccl (= RCCL over xGMI, the default) or gloo (rehearsal of the "
                         "multi-rank path on fewer GPUs than ranks; ranks then share devices and gather through host memory)")
    ap.add_argument("--via", default=None, choices=["capi", "torch"],
                    help="--gpus > 1, who gathers the grid: 'capi' (default with the nccl backend) = the product's own entry point, "
                         "gple_*_predict_sharded / _dealt with an ncclComm_t created here from librccl (what a C++ output_phase calls; "
                         "torch.distributed then only carries the 128-byte unique id, the barriers and the timing reduction, over gloo); "
                         "'torch' = parallel.GridShardedStep with torch.distributed.all_gather_into_tensor (the A/B, and the gloo rehearsal)")
    ap.add_argument("--emulate-rank", default=None, metavar="r/P",
                    help="--gpus 1, grid workloads: time what rank r of P ranks does in the sharded step — the replicated fit, the predict of ITS blocks of the "
                         "block-cyclic deal, the unpack of the full grid — with a stand-in transport that only moves this rank's block "
                         "(gple_debug_solo_allgather): the one-GPU proxy of the per-rank step, everything but the fabric.  Not a judged value.")
    ap.add_argument("--comm-at-one", action="store_true", help="--gpus 1 --via capi: still create a one-rank ncclComm_t and go through the all-gather path")
    ap.add_argument("--plan", default="auto", choices=["auto", "elements", "grid", "hybrid"], help="--workload C4 | C5: force one of the planner's candidates")
    ap.add_argument("--opt-only", type=int, default=None, choices=[0, 1, 2],
                    help="--workload C4opt: evaluate only element i (0, 2: real; 1: complex) — its derivative GEMMs then have the GPU to themselves")
    ap.add_argument("--shard", default="grid", choices=["grid", "elements"],
                    help="--gpus > 1: 'grid' (default) = strong scaling of ONE element's step, the grid prediction split over the "
                         "ranks and all-gathered; 'elements' = weak scaling over the independent density-matrix elements, every "
                         "rank fits and predicts its own element on the whole grid, no data-path collective (SURVEY.md §8e)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: this process has made no GPU call yet (torch is not even imported), so it may start the N
        # ranks as fresh children and relay rank 0's line
        sys.exit(launch_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    OUT = _keep_stdout_for_the_json_line()

    import torch
    import torch.distributed as dist

    import gaussian_process_liouville_equation_amd as pkg
    from gaussian_process_liouville_equation_amd import _capi as c
    from gaussian_process_liouville_equation_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    if world > torch.cuda.device_count():
        # A rehearsal: several ranks on one GPU.  The one-launch factorisation's workgroups wait for one another inside a launch, and a neighbour process
        # whose contraction holds every CU for seconds can starve them past their bounded wait — the library then recovers (a launch per panel) and reports
        # the predicts enqueued before as stale (GPLE_ERR_TIMEOUT), which a multi-rank step cannot repeat without unmatching its collectives (seen once in
        # round 4's rehearsals).  Ranks that share a GPU therefore fit with a launch per panel (no waiting kernels); ranks with a GPU each are not touched.
        os.environ.setdefault("GPLE_CHOL_SCHEME", "step")
    torch.cuda.set_device(dev)
    # Everything (the library's kernels, torch's copies, the collectives) runs on ONE non-blocking stream, not on the legacy null stream:
    # once another library has created blocking streams in the process (RCCL does), every launch on the null stream pays for the implicit
    # synchronisation with them — the fit's 64 dependent panel launches took 3.9 ms instead of 2.0 ms with a communicator in the process
    # (gpurun_out/r03_bench_c4r_capi1.json of round 3; probes/rccl_fit_interference.py shows a context with its own stream is unaffected)
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    multi = WORKLOADS[args.workload][2] in ELEMENT_KINDS
    grid_wl = WORKLOADS[args.workload][2] in ("real", "complex")
    if args.via is None:
        args.via = "capi" if (args.backend == "nccl" and (multi or (grid_wl and args.shard == "grid"))) else "torch"
    if args.via == "capi" and not (multi or grid_wl):
        raise SystemExit("--via capi applies to the grid-predict workloads (C1 ... C5r, C4, C5)")
    if args.via == "capi" and args.backend == "gloo":
        raise SystemExit("--via capi gathers with RCCL inside the library; the gloo rehearsal is --via torch")
    args.control = "gloo" if (args.via == "capi" or args.backend == "gloo") else "nccl"  # who carries barriers and the timing reduction
    if world > 1:
        if args.control == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    if multi:
        return elements_step(args, pkg, c, parallel, torch, dist, rank, world, dev)
    if args.workload == "C4opt":
        return opt_loop(args, pkg, c, parallel, torch, dist, rank, world, dev)
    if WORKLOADS[args.workload][2] in ("step", "step3"):
        return step_loop(args, pkg, torch, dist, rank, world, dev)
    if WORKLOADS[args.workload][2] == "nlml":
        return nlml_step(args, pkg, c, torch, dist, rank, world, dev)

    N, G, kernel = WORKLOADS[args.workload]
    cplx = kernel == "complex"
    M = G * G
    by_element = args.shard == "elements" and world > 1
    X, y, grid, theta = synthetic(N, G, 20240607 + 1 + (rank if by_element else 0), kernel)

    stream = torch.cuda.current_stream()
    api = pkg.open_api(dev, stream=None if os.environ.get("BENCH_OWN_STREAM") else stream.cuda_stream)  # the library runs on torch's current stream
    api.enable_timing(True)
    dX = torch.from_numpy(X).cuda()
    dy = torch.from_numpy(np.ascontiguousarray(y).view(np.float64) if cplx else y).cuda()
    dgrid_all = torch.from_numpy(grid).cuda()
    dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
    th = np.ascontiguousarray(theta)
    sc, ps = (c.ComplexFitScalars() if cplx else c.RealFitScalars()), c.PredictScalars()
    flags = c.CALC_ERROR | c.CALC_AVERAGE | c.IO_DEVICE
    thp = th.ctypes.data_as(C.POINTER(C.c_double))
    # rows: mean, variance, cut-off mean (real: 1 + 1 + 1, complex: 2 + 1 + 2 doubles per point)
    rows = 5 if cplx else 3
    alloc = lambda Cc, per: torch.zeros(Cc, per, dtype=torch.float64, device="cuda")
    # by_element: a rank that owns a whole element predicts the whole grid; nothing is gathered
    # --prune on several GPUs: block-cyclic shares, or one rank would hold all the live rows of the grid
    shard = parallel.GridShardedStep(M, rows, alloc, via_host=args.backend == "gloo", shard=not by_element, cyclic=args.prune and not cplx)
    lo, hi = shard.lo, shard.hi
    dgrid_mine = dgrid_all[shard.idx.to(dgrid_all.device)].contiguous() if shard.cyclic else None
    # --via capi: the product's own collective.  One ncclComm_t per process from the librccl of this process; the library resolves
    # ncclAllGather from the process image, i.e. from the same library.
    emu = None
    if args.emulate_rank:
        er, ep = (int(t) for t in args.emulate_rank.split("/"))
        if world != 1 or not (0 <= er < ep <= 64):
            raise SystemExit("--emulate-rank r/P needs --gpus 1 and 0 <= r < P <= 64")
        emu = (er, ep)
    capi = args.via == "capi" and not by_element and (world > 1 or args.comm_at_one) and not os.environ.get("BENCH_COMM_UNUSED") and emu is None
    if os.environ.get("BENCH_COMM_UNUSED"):  # a communicator in the process that the step does not use (A/B of the hardware-queue hazard)
        _RCCL["unused"] = try_rccl_comm(torch, dist, rank, world)[0]
    comm, via_note = None, ("torch.distributed all_gather_into_tensor" if world > 1 else "no collective (one rank)")
    if capi:
        comm, rccl_path, why = try_rccl_comm(torch, dist, rank, world)
        if comm is None:  # every rank agrees: the A/B path on torch.distributed's own RCCL group, named as a fallback in the line
            capi, args.via = False, "torch"
            group = dist.new_group(backend="nccl") if world > 1 else None
            shard = parallel.GridShardedStep(M, rows, alloc, group=group, shard=not by_element, cyclic=args.prune and not cplx)
            lo, hi = shard.lo, shard.hi
            dgrid_mine = dgrid_all[shard.idx.to(dgrid_all.device)].contiguous() if shard.cyclic else None
            via_note = f"FALLBACK: torch.distributed all_gather_into_tensor over its own RCCL group (the library-side communicator failed: {why})"
    if capi:
        via_note = f"gple_{'complex' if cplx else 'real'}_predict_sharded: ncclAllGather inside the library ({rccl_path}), block-cyclic deal of 128-point blocks"
        full_out = torch.empty(rows, M, dtype=torch.float64, device="cuda")
        pts_mine, _ = parallel.deal_shares(M, [1] * world)
        lo, hi = 0, pts_mine[rank]  # the rank's number of points (for the flop count below)
    sh_rank, sh_world = rank, world
    if emu is not None:  # one rank of a world that is not there: the library's sharded entry point with the stand-in transport
        sh_rank, sh_world = emu
        api.lib.gple_set_allgather_function.argtypes = [C.c_void_p]
        api.lib.gple_set_allgather_function(C.cast(api.lib.gple_debug_solo_allgather, C.c_void_p))
        comm = C.c_void_p(1 + sh_rank + 256 * sh_world)
        capi = True
        via_note = (f"EMULATED rank {sh_rank} of {sh_world}: gple_{'complex' if cplx else 'real'}_predict_sharded with gple_debug_solo_allgather "
                    f"(this rank's blocks only; no fabric)")
        full_out = torch.empty(rows, M, dtype=torch.float64, device="cuda")
        pts_mine, _ = parallel.deal_shares(M, [1] * sh_world)
        lo, hi = 0, pts_mine[sh_rank]

    def fit():
        h = C.c_void_p()
        if cplx:
            st = api.lib.gple_complex_fit_create(api.ctx, thp, dp(dX), dp(dy), N, flags, None, C.byref(h))
        else:
            st = api.lib.gple_real_fit_create(api.ctx, thp, dp(dX), dp(dy), 0, N, flags, None, C.byref(h))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())
        return h

    def predict_slice(h, lo, hi, out):
        # complex outputs are interleaved (re, im) pairs: rows 0-1 / 3-4 of `out` viewed as one buffer of 2 * per doubles
        o_mean, o_var, o_cut = (out[0:2], out[2], out[3:5]) if cplx else (out[0], out[1], out[2])
        fn = api.lib.gple_complex_predict if cplx else api.lib.gple_real_predict
        pts, n = (dgrid_mine, len(dgrid_mine)) if hi is None else (dgrid_all[lo:hi], hi - lo)  # hi is None: cyclic share (lo = its indices)
        st = fn(api.ctx, h, dp(pts), n, c.IO_DEVICE | predict_mode["flag"], None, dp(o_mean), dp(o_var), dp(o_cut), C.byref(ps))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())

    last = {}
    # the timed `value` contracts every grid row (GPLE_PREDICT_FULL) unless --prune is given: the library's default skips 128-row
    # blocks whose K* cannot move the variance (bit-identical output, most of a phase-space grid) — that step is timed separately
    # below and reported as "pruned", so that `value` and `roofline` keep pricing the full N(N+1) flops per grid point
    predict_mode = {"flag": 0 if args.prune else c.PREDICT_FULL}

    def predict_capi(h):
        o_mean, o_var, o_cut = (full_out[0:2], full_out[2], full_out[3:5]) if cplx else (full_out[0], full_out[1], full_out[2])
        fn = api.lib.gple_complex_predict_sharded if cplx else api.lib.gple_real_predict_sharded
        fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_uint, C.c_int, C.c_int, C.c_void_p] + [C.POINTER(C.c_double)] * 3
        st = fn(api.ctx, h, dp(dgrid_all), M, c.IO_DEVICE | predict_mode["flag"], sh_rank, sh_world, comm, dp(o_mean), dp(o_var), dp(o_cut))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())

    def step():
        if capi:
            h = fit()
            predict_capi(h)  # this rank's blocks + ncclAllGather + unpack, all enqueued on the context's stream
            full = full_out
        else:
            h, full = shard.run(fit, predict_slice)  # fit + this rank's slice + all-gather (RCCL on the same stream as the kernels)
        last["full"] = full
        # the fit's scalar members (error, population, <r>, purity): the one host synchronisation of the step; fit and
        # predict above only enqueue
        get = api.lib.gple_complex_fit_get_scalars if cplx else api.lib.gple_real_fit_get_scalars
        st = get(h, C.byref(sc))
        if st == c.GPLE_ERR_TIMEOUT and world == 1:
            # the one-launch factorisation gave up waiting and the getter repeated it with a launch per panel (include/gple.h): the predict
            # enqueued above saw NaN — once more on the good fit, inside the timed region, and the line says so.  (Several ranks: a repeat
            # would leave the collectives unmatched; the step fails instead.)
            last["recovered"] = last.get("recovered", 0) + 1
            if capi:
                predict_capi(h)
            else:
                predict_slice(h, lo, hi if not shard.cyclic else None, shard.local)
            st = get(h, C.byref(sc))
        if st != 0 or not np.isfinite(sc.purity):
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode() or "non-finite fit scalars")
        (api.lib.gple_complex_fit_release if cplx else api.lib.gple_real_fit_release)(h)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    api.enable_timing(True)  # reset the accumulators
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # sanity of the (gathered) grid: every rank must now hold all M points (checked once, outside the timed region)
    full = last["full"]
    if tuple(full.shape) != (rows, M) or not bool(torch.isfinite(full).all()):
        raise RuntimeError("gathered prediction has the wrong shape or contains non-finite values")
    ms_per_step = 1e3 * elapsed / args.steps

    _, fit_total, fit_cnt = api.timing(0)
    _, pred_total, pred_cnt = api.timing(1)
    _, pk_total, pk_cnt = api.timing(2)  # HIP events around every rownorm_kernel launch (the K* chunks of one predict)
    contraction_kernel = api.last_contraction_kernel()  # what the timed steps ran on (asked before the pruned leg below launches its own)
    pk_ms = pk_total / max(1, pk_cnt)    # average duration of ONE launch (what rocprofv3 --stats reports)
    launches_per_step = pk_cnt / max(1, args.steps)
    m_local = hi - lo
    # algorithmic flops of the dominant kernel: triangular contraction ||T k*||^2 = N(N+1) flops per test point
    # (the reference's row * K^-1 * row^T form is 2 N^2 per point), split over the launches
    # complex GP = real GP on [Re; Im]: 2 typed rows per point against the 2N x 2N factor
    nn, trows = (2 * N, 2 * m_local) if cplx else (N, m_local)
    flops = float(trows) * nn * (nn + 1) / max(1.0, launches_per_step)
    achieved = flops / (pk_ms * 1e-3) / 1e12 if pk_ms > 0 else 0.0
    traffic, traffic_src = traffic_for(args.workload, world)
    # SURVEY.md §8(d) per-step models (per element; the complex element in its real [Re; Im] embedding: n = 2N, 2M typed rows)
    t_s = ms_per_step * 1e-3
    nM, TM = (2 * M, 128) if cplx else (M, 128)
    F_fit = float(nn) ** 3 + 4.0 * nn * nn                       # factor + inverse factor + two solves (N^3 + 4 N^2)
    F_contract = float(nM) * nn * (nn + 1)                       # executed (triangular) form of the variance contraction
    F_contract_ref = (32.0 if cplx else 2.0) * M * N * N        # the reference's row W row^T form
    E = (3.0 if cplx else 1.0) * (N * N / 2.0 + N * N / 2.0 + float(M) * N)  # exp evaluations: K, purity K1, K*
    B_fused = 24.0 * N + 16.0 * M + 8.0 * rows * M + 16.0 * nn * nn + np.ceil(nM / TM) * 8.0 * nn * (nn + 1) / 2.0
    B_mat = B_fused + 24.0 * nM * nn                             # + write K* once, read it twice (mean, variance)
    result = {
        "metric": "GP fit+predict ms/step (N samples, M grid pts)",
        "value": round(ms_per_step, 4),
        "unit": "ms/step",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": False,
        "scaling": "weak" if by_element else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: N={N} samples, {G}x{G} grid (M={M}), {kernel} SE kernel, fit(error+average) + grid predict(mean,var,cutoff)",
                   "N": N, "M": M, "parallelism": (f"{world} independent density-matrix elements, one per GPU, no data-path collective" if by_element else
                                   f"grid-sharded x{world}, replicated fit, {'RCCL' if args.backend == 'nccl' else 'gloo (host)'} all-gather") if world > 1 else "single GPU",
                   "via": args.via, "collective": via_note},
        "roofline": {"bound": "mfma", "kernel": contraction_kernel + " (fp64 MFMA triangular contraction ||T k*||^2 over one K* chunk)",
                     "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_PEAK_TFLOPS, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": round(pk_ms, 4), "launches_per_step": launches_per_step,
                     "algorithmic_flops_per_launch": flops,
                     "reference_form_flops_per_launch": (32.0 if cplx else 2.0) * m_local * N * N / max(1.0, launches_per_step)},
        "phases_ms": {"fit_device": round(fit_total / max(1, fit_cnt), 4), "predict_device": round(pred_total / max(1, pred_cnt), 4),
                      "rownorm_kernel_per_step": round(pk_total / max(1, args.steps), 4)},
        # SURVEY.md §8(d) report fields, whole step, whole job (all ranks together process one element's step)
        "mfma_frac_step": round((F_fit + F_contract) / t_s / (FP64_PEAK_TFLOPS * 1e12), 4),
        "mfma_frac_step_reference_form": round((F_fit + F_contract_ref) / t_s / (FP64_PEAK_TFLOPS * 1e12), 4),
        "hbm_gbps_fused": round(B_fused / t_s / 1e9, 2),
        "hbm_gbps_mat": round(B_mat / t_s / 1e9, 2),
        "exp_rate": round(E / t_s, 1),
        "models": {"F_fit": F_fit, "F_contract_triangular": F_contract, "F_contract_reference_form": F_contract_ref, "exp_count": E,
                   "B_fused_bytes": B_fused, "B_mat_bytes": B_mat, "T_M": TM,
                   "note": "B_fused: inputs + outputs + K and T written once + the lower triangle of T re-read per 128-row tile (L2/MALL-resident); "
                           "B_mat adds the materialised K* (8 B written, 16 B read per entry) = the reference-equivalent byte model; "
                           "hbm_gbps_* = model bytes / step time, not counter traffic (that is roofline.traffic)"},
    }
    if args.prune:
        result["config"]["workload"] += " [--prune: far rows not contracted]"
    if emu is not None:
        result["config"]["emulated_rank"] = {"rank": emu[0], "world": emu[1], "points_of_this_rank": hi - lo,
                                             "note": "per-rank step of the sharded predict on ONE GPU (fit + this rank's blocks + unpack; stand-in transport): a proxy, not a judged value"}
    if last.get("recovered"):
        result["recovered_from_give_up"] = last["recovered"]
    if world == 1 and not args.prune and emu is None:
        full_out = last["full"].clone()
        predict_mode["flag"] = 0
        api.prune_stats(reset=True)
        nrep = max(3, min(args.steps, 10))
        step()
        torch.cuda.synchronize()
        api.prune_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(nrep):
            step()
        torch.cuda.synchronize()
        t_pruned = (time.perf_counter() - t0) / nrep
        live, seen = api.prune_stats(reset=True)
        result["pruned"] = {"ms_per_step": round(1e3 * t_pruned, 4), "blocks_contracted": live // nrep, "blocks_seen": seen // nrep,
                            "bit_identical_to_full": bool(torch.equal(full_out, last["full"])),
                            "note": "the library's default predict: grid rows with |k*|^2 < 2^-56 sf^2 sn^2 k(x*,x*) are not contracted "
                                    "(k(x*,x*) - k* K^-1 k*^T rounds to k(x*,x*) either way; blocks_* count live / all rows in units of 128); not the judged value"}
    if rank == 0 and not args.no_cpu_baseline and world == 1 and emu is None:
        result["cpu_baseline"] = cpu_baseline(args.workload, N, G, kernel, X, y, grid, theta)
    if rank == 0:
        emit(result)
    if comm is not None and emu is None:
        destroy_rccl_comm(comm)
    api.close()
    if world > 1:
        dist.destroy_process_group()


def elements_step(args, pkg, c, parallel, torch, dist, rank, world, dev):
    """--workload C4 | C5: one step = fit(error + averages) + full-grid predict of EVERY element of the density matrix (predict.cpp:390-393 +
    output.cpp:181-233: 2 real + 1 complex at NumPES = 2, 3 + 3 at NumPES = 3), the elements spread over the ranks by parallel.plan_elements
    (whole elements, everyone's grid split over everyone, or equal-time stretches) and every element's grid gathered on every rank by the
    library's weighted deal (gple_*_predict_dealt, RCCL inside) or, --via torch, by parallel.gather_dealt."""
    N, G, tag = WORKLOADS[args.workload]
    kinds = ELEMENT_KINDS[tag]
    E, M = len(kinds), G * G
    costs = [parallel.model_costs(k, N, M) for k in kinds]
    plan = parallel.plan_elements(costs, world, M)
    if args.plan != "auto" and world > 1:
        forced = {"elements": None, "grid": [[1] * world] * E, "hybrid": None}
        if args.plan == "grid":
            plan = parallel.Plan("grid", forced["grid"], costs, M)
        elif args.plan == "hybrid":
            plan = parallel.Plan("hybrid", [parallel._quantise(f, parallel.DEAL_CYCLE) for f in parallel._hybrid_fractions(costs, world)], costs, M)
        else:
            w = [[0] * world for _ in range(E)]
            load = [0.0] * world
            for e in sorted(range(E), key=lambda e: -sum(costs[e])):
                r = min(range(world), key=lambda r: load[r])
                load[r] += sum(costs[e])
                w[e][r] = 1
            plan = parallel.Plan("elements", w, costs, M)
    stream = torch.cuda.current_stream()
    api = pkg.open_api(dev, stream=stream.cuda_stream)
    api.enable_timing(True)
    dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
    data = []
    grid = None
    for e, kind in enumerate(kinds):
        X, y, grid, theta = synthetic(N, G, 20240607 + 3 + e, kind)
        cplx = kind == "complex"
        data.append({"cplx": cplx, "X": torch.from_numpy(X).cuda(), "y": torch.from_numpy(np.ascontiguousarray(y).view(np.float64) if cplx else y).cuda(),
                     "theta": np.ascontiguousarray(theta), "out": torch.empty(5 if cplx else 3, M, dtype=torch.float64, device="cuda"),
                     "sc": c.ComplexFitScalars() if cplx else c.RealFitScalars()})
    dgrid = torch.from_numpy(grid).cuda()
    capi = args.via == "capi"
    comm, rccl_path, torch_group, fallback = None, None, None, ""
    if capi and (world > 1 or args.comm_at_one):
        comm, rccl_path, fallback = try_rccl_comm(torch, dist, rank, world)
        if comm is None:  # every rank agrees: gather_dealt over torch.distributed's own RCCL group instead, named as a fallback in the line
            capi, args.via = False, "torch"
            torch_group = dist.new_group(backend="nccl") if world > 1 else None
    flags = c.CALC_ERROR | c.CALC_AVERAGE | c.IO_DEVICE
    predict_flag = 0 if args.prune else c.PREDICT_FULL

    def fit(e):
        d, h = data[e], C.c_void_p()
        thp = d["theta"].ctypes.data_as(C.POINTER(C.c_double))
        if d["cplx"]:
            st = api.lib.gple_complex_fit_create(api.ctx, thp, dp(d["X"]), dp(d["y"]), N, flags, None, C.byref(h))
        else:
            st = api.lib.gple_real_fit_create(api.ctx, thp, dp(d["X"]), dp(d["y"]), 0, N, flags, None, C.byref(h))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())
        return h

    def predict_dealt(e, h, weights):
        d = data[e]
        out = d["out"]
        o_mean, o_var, o_cut = (out[0:2], out[2], out[3:5]) if d["cplx"] else (out[0], out[1], out[2])
        if capi or world == 1:
            fn = api.lib.gple_complex_predict_dealt if d["cplx"] else api.lib.gple_real_predict_dealt
            fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p] + [C.POINTER(C.c_double)] * 3
            st = fn(api.ctx, h, dp(dgrid), M, c.IO_DEVICE | predict_flag, rank, world, (C.c_int * world)(*weights), comm, dp(o_mean), dp(o_var), dp(o_cut))
            if st != 0:
                raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())
            return out
        # --via torch: this rank's share by the plain predict, gathered by torch.distributed (complex results travel de-interleaved: rows
        # Re mean, Im mean, variance, Re cut, Im cut)
        idx, per = parallel.dealt_indices(M, rank, weights)
        n, ow = len(idx), 2 if d["cplx"] else 1
        local = torch.zeros(out.shape[0], per, dtype=torch.float64, device="cuda")
        if h is not None and n:
            pts = dgrid[idx.cuda()].contiguous()
            t_mean, t_var, t_cut = (torch.empty(n * ow, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda"),
                                    torch.empty(n * ow, dtype=torch.float64, device="cuda"))
            fn = api.lib.gple_complex_predict if d["cplx"] else api.lib.gple_real_predict
            ps = c.PredictScalars()
            st = fn(api.ctx, h, dp(pts), n, c.IO_DEVICE | predict_flag, None, dp(t_mean), dp(t_var), dp(t_cut), C.byref(ps))
            if st != 0:
                raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())
            local[0:ow, :n] = t_mean.view(n, ow).t()
            local[ow, :n] = t_var
            local[ow + 1:, :n] = t_cut.view(n, ow).t()
        full = parallel.gather_dealt(local, M, weights, group=torch_group, via_host=args.backend == "gloo")
        out[0:ow].view(-1).view(M, ow).copy_(full[0:ow].t())
        out[ow].copy_(full[ow])
        out[ow + 1:].view(-1).view(M, ow).copy_(full[ow + 1:].t())
        return out

    hybrid = parallel.HybridStep(plan, rank)
    last = {}

    def step():
        handles, outs = hybrid.run(fit, predict_dealt)
        last["outs"] = outs
        vals = {}
        for e, h in enumerate(handles):
            if h is None:
                continue
            d = data[e]
            st = (api.lib.gple_complex_fit_get_scalars if d["cplx"] else api.lib.gple_real_fit_get_scalars)(h, C.byref(d["sc"]))
            if st != 0 or not np.isfinite(d["sc"].purity):
                raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode() or "non-finite fit scalars")
            (api.lib.gple_complex_fit_release if d["cplx"] else api.lib.gple_real_fit_release)(h)
            if plan.owner(e) == rank:
                vals[e] = [d["sc"].error, d["sc"].purity]
        # the elements' scalars (error, purity: what the constraints and ave.txt consume) on every rank
        last["scalars"] = parallel.allgather_element_scalars(vals, E, 2)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    api.enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for e, o in enumerate(last["outs"]):
        if tuple(o.shape) != (5 if data[e]["cplx"] else 3, M) or not bool(torch.isfinite(o).all()):
            raise RuntimeError(f"element {e}: gathered prediction has the wrong shape or contains non-finite values")
    if not bool(torch.isfinite(last["scalars"]).all()):
        raise RuntimeError("non-finite element scalars")
    ms = 1e3 * elapsed / args.steps
    _, fit_total, fit_cnt = api.timing(0)
    _, pk_total, pk_cnt = api.timing(2)
    # contraction flops this rank executed per step: its share of every element (complex: 2 typed rows per point against the 2N x 2N factor)
    flops_rank = 0.0
    for e, kind in enumerate(kinds):
        pts, _ = parallel.deal_shares(M, plan.weights[e])
        n, rows = (2 * N, 2 * pts[rank]) if kind == "complex" else (N, pts[rank])
        flops_rank += float(rows) * n * (n + 1)
    achieved = flops_rank * args.steps / (pk_total * 1e-3) / 1e12 if pk_total > 0 else 0.0
    F_step = sum((float(2 * M) * (2 * N) * (2 * N + 1) + float(2 * N) ** 3) if k == "complex" else (float(M) * N * (N + 1) + float(N) ** 3) for k in kinds)
    result = {
        "metric": "GP fit+predict ms/step (N samples, M grid pts)", "value": round(ms, 4), "unit": "ms/step", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: all {E} elements of a {'2' if E == 3 else '3'}-state density matrix ({kinds.count('real')} real + {kinds.count('complex')} complex GPs), "
                               f"N={N} samples each, {G}x{G} grid (M={M}), fit(error+average) + grid predict(mean,var,cutoff) per element"
                               + (" [--prune: far rows not contracted]" if args.prune else ""),
                   "N": N, "M": M, "elements": kinds, "parallelism": "single GPU, elements in turn" if world == 1 else f"{world} ranks, plan '{plan.name}'",
                   "via": args.via, "collective": (f"gple_*_predict_dealt: ncclAllGather inside the library ({rccl_path}), weighted block deal" if comm is not None
                                                   else ("none (one rank)" if world == 1 else "torch.distributed all_gather_into_tensor"
                                                         + (f" — FALLBACK, the library-side communicator failed: {fallback}" if fallback else ""))),
                   "plan": plan.describe()},
        "roofline": {"bound": "mfma", "kernel": api.last_contraction_kernel() + " (fp64 MFMA triangular contraction ||T k*||^2), all elements of this rank",
                     "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel_ms": round(pk_total / max(1, pk_cnt), 4), "launches_per_step": pk_cnt / max(1, args.steps),
                     "algorithmic_flops_per_step_this_rank": flops_rank},
        "phases_ms": {"fit_device_each": round(fit_total / max(1, fit_cnt), 4), "fits_per_step_this_rank": fit_cnt / max(1, args.steps),
                      "rownorm_kernel_per_step": round(pk_total / max(1, args.steps), 4)},
        "mfma_frac_step": round(F_step / (ms * 1e-3) / (FP64_PEAK_TFLOPS * 1e12) / world, 4),
        "element_scalars": {"error": [float(x) for x in last["scalars"][:, 0]], "purity": [float(x) for x in last["scalars"][:, 1]]},
    }
    if rank == 0:
        emit(result)
    if comm is not None:
        destroy_rccl_comm(comm)
    api.close()
    if world > 1:
        dist.destroy_process_group()


def step_loop(args, pkg, torch, dist, rank, world, dev):
    """--workload C2step / C5step: one tick of the reference's main loop (main.cpp:143-176) per step, all on the device:
    evolve(density) + evolve(extra points, 5N per element) — 8 back-propagated points per sample, predicted in three batches —
    then TrainingKernels(params, density) (three fits with error + averages).  Single GPU (replicas with --gpus N)."""
    from gaussian_process_liouville_equation_amd import kernels as K, steploop
    N = WORKLOADS[args.workload][0]
    num_pes = 3 if WORKLOADS[args.workload][2] == "step3" else 2
    model = steploop.TSAC if num_pes == 3 else steploop.DAC  # three levels: the library's three-state model (the reference has none)
    order = K.element_order(num_pes)
    dens, extra, params = {}, {}, {}
    weight = {2: (0.6, 0.4), 3: (0.5, 0.3, 0.2)}[num_pes]
    for k, e in enumerate(order):
        cplx = e[0] != e[1]
        X, y, _, theta = synthetic(N, 1, 20240607 + 50 + k + 10 * rank, "complex" if cplx else "real")
        X[:, 0] += 8.5 if num_pes == 2 else 9.5  # the packet about to enter the coupling region (Tully II: x ~ -1.5; TSAC: x ~ -0.5)
        Xe, ye = extra_points(X, 20240607 + 60 + k, "complex" if cplx else "real")
        scale = (0.6, 1.0, 0.4)[k] if num_pes == 2 else (weight[e[0]] if not cplx else 0.6 * np.sqrt(weight[e[0]] * weight[e[1]]))
        dens[e] = (X, np.asarray(y, dtype=complex) * scale)
        extra[e] = (Xe, ye * scale)
        # two levels: the reference's initial parameters as in round 2's lines; three levels: distinct sub-kernels for the coherences
        # (tests/test_gpu_step_loop.py::test_tick_at_c5_size explains why the initial complex parameters make a poor fit)
        params[e] = list(theta) if (not cplx or num_pes == 2) else [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 1e-2]
    api = pkg.open_api(dev)
    kernels = K.TrainingKernels(params, K.construct_training_sets(dens, num_pes), True, True, False, api=api, num_pes=num_pes)
    state = {"d": dens, "x": extra, "k": kernels}

    def step():
        state["d"], state["x"], state["k"] = steploop.tick(state["d"], state["x"], params, 2000.0, 1.0, state["k"], model, api)
        return state["k"].calculate_population()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pop = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not np.isfinite(pop):
        raise RuntimeError("non-finite population after the tick")
    ms = 1e3 * elapsed / args.steps
    # contraction flops of the batched predicts of one tick: every element is asked at NE branches of the (N + 5N) points of each of the NE
    # elements (two levels: 8 of the 9 — the exact density stands in for one); complex: 2 typed rows, n = 2N
    NE = len(order)
    rows = (8 if num_pes == 2 else NE * NE) * 6 * N
    n_real, n_cplx = num_pes, NE - num_pes
    F = n_real * float(rows) * N * (N + 1) + n_cplx * float(2 * rows) * (2 * N) * (2 * N + 1)
    result = {
        "metric": "GP fit+predict ms/step (N samples, M grid pts)", "value": round(ms, 3), "unit": "ms/step", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: one tick of main.cpp:143-176, N={N} points per element, {NE} elements ({n_real} real + {n_cplx} complex), 5N extra points each, "
                               f"{'Tully II' if num_pes == 2 else 'three-state model (gple_evolve_n)'}, {rows} back-propagated points per element and tick", "N": N, "M": rows,
                   "num_pes": num_pes,
                   "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
        "roofline": {"bound": "mfma", "kernel": "rownorm kernels of the batched predicts (one per element and evolve call)", "achieved": round(F / (ms * 1e-3) / 1e12, 3), "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(F / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4), "traffic": None,
                     "note": "whole-tick rate: contraction flops of the predicts / tick time (fits, K* generation and the host round trips of the Python mirror included)"},
        "population_after": pop,
    }
    if rank == 0:
        emit(result)
    api.close()
    if world > 1:
        dist.destroy_process_group()


def nlml_step(args, pkg, c, torch, dist, rank, world, dev):
    """--workload NLML1024 | NLML4096: negative_log_marginal_likelihood (value + trace gradient, test/gpr.cpp:499-532) followed by predict_phase
    (mean-only prediction k*(x) K^-1 y on a 256 x 256 grid, :654-706) — both build and factor K, as the reference's two functions do.  Kernel =
    w_d^2 Diag + w_g^2 GaussianARD (NOCROSS build: diagonal weights = inverse lengths).  The dominant work is the factorisation and the inverse
    factor (N^3 / 3 each, twice per step) plus T^T T for the gradient (N^3 / 3): fp64 MFMA; the prediction is M N exponentials: fp64 VALU.
    Single GPU (replicas with --gpus N)."""
    N, G, _ = WORKLOADS[args.workload]
    M = G * G
    X, y, _, _ = synthetic(N, 1, 20240607 + 70 + rank, "real")
    gx, gp = np.meshgrid(np.linspace(-13.0, -7.0, G), np.linspace(11.0, 17.0, G), indexing="ij")
    grid = np.ascontiguousarray(np.stack([gx.ravel(), gp.ravel()], axis=1))
    x = np.array([0.05, 1.3, 1.0 / 0.7086, 1.0 / 0.7056])  # (w_d, w_g, a_x, a_p): initial ARD weight = 1 / sigma (test/gpr.cpp:167-173)
    api = pkg.open_api(dev)
    api.enable_timing(True)
    dgrid = torch.from_numpy(grid).cuda()
    dmean = torch.empty(M, dtype=torch.float64, device="cuda")
    dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
    hp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    Xc, yc = np.ascontiguousarray(X), np.ascontiguousarray(y)
    val, grad = C.c_double(), np.zeros(4)
    api.lib.gple_nlml.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3 + [C.c_size_t] + [C.POINTER(C.c_double)] * 2
    api.lib.gple_nlml_predict.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3 + [C.c_size_t, C.POINTER(C.c_double), C.c_size_t, C.c_uint, C.POINTER(C.c_double)]

    def step():
        st = api.lib.gple_nlml(api.ctx, hp(x), hp(Xc), hp(yc), N, C.cast(C.byref(val), C.POINTER(C.c_double)), hp(grad))
        if st == 0:
            st = api.lib.gple_nlml_predict(api.ctx, hp(x), hp(Xc), hp(yc), N, dp(dgrid), M, c.IO_DEVICE, dp(dmean))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    api.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not (np.isfinite(val.value) and np.all(np.isfinite(grad)) and bool(torch.isfinite(dmean).all())):
        raise RuntimeError("non-finite NLML value, gradient or prediction")
    ms = 1e3 * elapsed / args.steps
    _, f_total, f_cnt = api.timing(0)  # HIP events around chol_inverse_factor: L = chol(K) and T = L^-1 (two per step)
    _, p_total, p_cnt = api.timing(1)  # around the prediction kernels
    n = -(-N // 256) * 256
    f_ms = f_total / max(1, f_cnt)
    fl = 2.0 * float(n) ** 3 / 3.0  # N^3 / 3 (factor) + N^3 / 3 (inverse factor)
    achieved = fl / (f_ms * 1e-3) / 1e12 if f_ms > 0 else 0.0
    p_ms = p_total / max(1, p_cnt)
    result = {
        "metric": "GP fit+predict ms/step (N samples, M grid pts)", "value": round(ms, 4), "unit": "ms/step", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: negative log marginal likelihood value + gradient (test/gpr.cpp:499-532), then mean-only prediction on a {G}x{G} grid "
                               f"(:654-706), N={N}, Diag + GaussianARD kernel (NOCROSS)", "N": N, "M": M,
                   "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
        "roofline": {"bound": "mfma", "kernel": "chol_inverse_factor: potrf_dag_kernel launches + the GEMMs of the block-row inverse (L = chol(K), T = L^-1)",
                     "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_PEAK_TFLOPS, 4), "traffic": None,
                     "kernel_ms": round(f_ms, 4), "launches_per_step": f_cnt / max(1, args.steps), "algorithmic_flops_per_launch": fl,
                     "note": "a latency-bound factorisation (64 dependent panel steps at N = 4096): the fraction of the MFMA peak says how far, not how well tuned"},
        "phases_ms": {"factorisation_each": round(f_ms, 4), "prediction_kernels": round(p_ms, 4)},
        "exp_rate": round(float(M) * N / (p_ms * 1e-3), 1) if p_ms > 0 else None,
        "nlml": {"value": val.value, "gradient": [float(g) for g in grad]},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding
        ora = binding.load()
        t0 = time.perf_counter()
        ora.nlml(x, X, y)
        t_val = time.perf_counter() - t0
        # the prediction re-factors K (as the reference's predict_phase does) and then costs M N exponentials: two sample sizes separate the two parts
        m1, m2 = 1024, 4096
        t0 = time.perf_counter()
        ora.nlml_predict(x, X, y, grid[:m1])
        t1 = time.perf_counter()
        ora.nlml_predict(x, X, y, grid[:m2])
        t2 = time.perf_counter()
        per_point = max(0.0, ((t2 - t1) - (t1 - t0)) / (m2 - m1))
        fixed = max(0.0, (t1 - t0) - per_point * m1)
        t_pred = fixed + per_point * M
        result["cpu_baseline"] = {"value": round(1e3 * (t_val + t_pred), 1), "unit": "ms/step", "cores": ora.num_threads, "kind": "port", "cpu_model": cpu_model(),
                                  "sample": f"oracle (CPU restatement, OpenMP, {ora.num_threads} threads): NLML value + gradient once ({t_val:.2f} s) + prediction: its "
                                            f"factorisation ({fixed:.2f} s) + {per_point * 1e6:.2f} us per grid point from samples of {m1} and {m2} points, scaled to M = {M} ({t_pred:.2f} s); one run"}
    if rank == 0:
        emit(result)
    api.close()
    if world > 1:
        dist.destroy_process_group()


def opt_loop(args, pkg, c, parallel, torch, dist, rank, world, dev):
    """--workload C4opt: the optimiser's inner loop of BASELINE configs[3] (opt.cpp:441-482, 844-870; main.cpp:35).
    One step = full_loose value + gradient = loose_function of 2 real + 1 complex element (N = 4096, 5N extra points each),
    every element on its own context / HIP stream (kernels.ApiPool), data resident (gple_objective_*).  With N ranks the
    elements are dealt out to the ranks (element e -> rank e mod world, parallel.element_owner) and the scalars all-reduced."""
    from gaussian_process_liouville_equation_amd import kernels as K
    N = WORKLOADS["C4opt"][0]
    elems = [("real", 0), ("complex", 1), ("real", 2)]  # reference order (0,0), (1,0), (1,1)
    # one rank: the three elements on three streams.  Several ranks: every rank holds all three objectives and evaluates ITS PART of each
    # (gple_objective_eval_part: the N^3 derivative products of its parameters, its share of the extra points; fits replicated) — dealing
    # whole elements to ranks buys nothing here, the complex element alone is the whole step — and one all-reduce of 3 x 9 doubles adds them up
    split = world > 1 and args.shard == "grid"
    mine = [i for i in range(3) if (split or parallel.element_owner(i, world) == rank) and args.opt_only in (None, i)]
    pool = K.ApiPool(n=max(1, len(mine)), device=dev)
    objs, thetas = {}, {}
    for slot, i in enumerate(mine):
        kernel = elems[i][0]
        X, y, _, theta = synthetic(N, 1, 20240607 + 3 + i, kernel)
        Xe, ye = extra_points(X, 20240607 + 30 + i, kernel)
        api = pool.api_for(slot)
        api.enable_timing(True)
        objs[i] = api.objective(X, np.asarray(y, dtype=complex), Xe, ye)
        thetas[i] = theta

    def step():
        if split:
            res = pool.map(lambda api, i: objs[i].part(thetas[i], rank, world, want_grad=True), mine)
        else:
            res = pool.map(lambda api, i: objs[i](thetas[i], want_grad=True), mine)
        vals = {i: [v] + list(g) + [0.0] * (8 - len(g)) for i, (v, g) in zip(mine, res)}
        if split:  # every rank contributes to every slot: a plain sum (make_normal, opt.cpp:420-431, after it)
            buf = torch.zeros(3, 9, dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
            for i, v in vals.items():
                buf[i] = torch.as_tensor(v, dtype=torch.float64)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            return buf  # raw sums: the sanity check below must see a NaN of any rank's part; make_normal (opt.cpp:420-431) belongs to the value handed to an optimiser
        allv = parallel.allgather_element_scalars(vals, 3, 9, device="cuda" if (world > 1 and args.control == "nccl") else "cpu")
        return allv

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    for slot in range(len(mine)):
        pool.api_for(slot).enable_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allv = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.control == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not bool(torch.isfinite(allv).all()):
        raise RuntimeError("non-finite objective value or gradient")
    ms = 1e3 * elapsed / args.steps
    # derivative GEMMs dK * K^-1 (kernel.cpp:354): HIP events of the library around each launch; 2 n^3 flops each
    gemm_ms, gemm_cnt, gemm_flops, per_elem = 0.0, 0, 0.0, {}
    for slot, i in enumerate(mine):
        api = pool.api_for(slot)
        _, tot, cnt = api.timing(3)
        _, ftot, fcnt = api.timing(0)
        _, ptot, pcnt = api.timing(1)
        n = 2 * 4096 if elems[i][0] == "complex" else 4096
        # real element: one n x n x n product per length parameter; complex element (n = 2N): two N x 2N x N products per sub-kernel
        # parameter inside one timed span — half of the full n^3 product, only the blocks the diagonals need (csrc/gple_capi.hip)
        span_flops = float(n) ** 3 if elems[i][0] == "complex" else 2.0 * float(n) ** 3
        gemm_ms, gemm_cnt, gemm_flops = gemm_ms + tot, gemm_cnt + cnt, gemm_flops + cnt * span_flops
        per_elem[f"element_{i}_{elems[i][0]}"] = {"fit_with_derivatives_ms": round(ftot / max(1, fcnt), 3), "predict_5N_ms": round(ptot / max(1, pcnt), 3),
                                                   "deriv_gemm_ms_each": round(tot / max(1, cnt), 3), "deriv_gemms_per_eval": cnt / max(1, args.steps)}
    achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    # SURVEY.md §8(d) minimum-work gradient model per element: F_grad = 2 D n^3 + 2 n^3 (D = 2 length parameters; real n = N).
    # The complex element in the [Re; Im] embedding runs 6 GEMMs of size 2N.
    F_eval = sum((6 if k == "complex" else 2) * 2.0 * ((2 * N if k == "complex" else N) ** 3) + ((2 * N if k == "complex" else N) ** 3) for k, i in elems
                 if args.opt_only in (None, i))
    result = {
        "metric": "GP fit+predict ms/step (N samples, M grid pts)", "value": round(ms, 4), "unit": "ms/step", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"C4opt: opt.cpp inner loop, full_loose value+gradient of 2 real + 1 complex element, N={N}, 5N={5 * N} extra points each",
                   "N": N, "M": 5 * N, "parallelism": ("3 elements on 3 HIP streams of one GPU" if args.opt_only is None else f"element {args.opt_only} alone") if world == 1 else
                   (f"every element's gradient split over {world} ranks (gple_objective_eval_part: derivative products by parameter, extra points by rows, fits replicated), one all-reduce of 27 doubles"
                    if split else f"elements dealt out over {world} ranks, scalars all-reduced")},
        "roofline": {"bound": "mfma", "kernel": "gemm_f64_kernel<128,128> (dK * K^-1 of the LOOCV gradient, kernel.cpp:354; complex element: the two half-size block products per parameter)", "achieved": round(achieved, 3),
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_PEAK_TFLOPS, 4),
                     "traffic": traffic_for("C4opt" if args.opt_only is None else f"C4opt_only{args.opt_only}", world)[0],
                     "kernel_ms": round(gemm_ms / max(1, gemm_cnt), 4), "launches_per_step": gemm_cnt / max(1, args.steps),
                     "note": ("the three elements run concurrently on one GPU, so a GEMM's event time includes what it shares with the other streams"
                              if args.opt_only is None else f"element {args.opt_only} alone: its GEMMs have the GPU to themselves")},
        "mfma_frac_step": round(F_eval / (ms * 1e-3) / (FP64_PEAK_TFLOPS * 1e12), 4),
        "phases_ms": per_elem,
    }
    if rank == 0:
        emit(result)
    for o in objs.values():
        o.release()
    pool.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
