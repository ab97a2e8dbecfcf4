#!/usr/bin/env python3
"""bench.py — GP fit + predict step (BASELINE.json metric) on 1..8 MI355X.

One step = what the reference does per output tick for one density-matrix element:
  TrainingKernel(theta, set, error=1, average=1, derivative=0)      (predict.cpp:390-393)
+ PredictiveKernel(grid, kernel, false): mean, variance, cut-off    (output.cpp:204-207)
Workload C2 (BASELINE.json configs[1]): N = 1024 samples, 256 x 256 grid, real SE kernel, fp64, synthetic inputs of
SURVEY.md §8(d) (seed 20240607 + 1).  Inputs are resident in HBM before the timed region.
N > 1 GPUs: strong scaling of the same step — every rank fits (replicated, no broadcast needed) and predicts its
contiguous slice of the grid; the slices are all-gathered over RCCL (the north-star partition).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # name: (N, G, kernel)
    "C1": (256, 128, "real"),
    "C2": (1024, 256, "real"),      # BASELINE.json configs[1]: the default
    "C3": (2048, 256, "complex"),   # configs[2]: off-diagonal density-matrix element
    "C4r": (4096, 512, "real"),     # one real element of C4 = the north-star target size
    "C4c": (4096, 512, "complex"),  # the complex element of C4
    "C5r": (8192, 1024, "real"),    # one real element of C5
}
FP64_PEAK_TFLOPS = 78.6  # MI355X fp64 vector = matrix peak; measured 78.4 with v_mfma_f64_16x16x4_f64


def synthetic(N, G, seed, kernel="real"):
    rng = np.random.Generator(np.random.PCG64(seed))
    x0, p0, sx, sp = -10.0, 14.112, 0.7086, 0.7056
    X = rng.normal([x0, p0], [sx, sp], size=(N, 2))
    y = np.exp(-0.5 * (((X[:, 0] - x0) / sx) ** 2 + ((X[:, 1] - p0) / sp) ** 2)) / (2 * np.pi * sx * sp)
    dx = 40.0 / G
    xs = -20.0 + dx * np.arange(G)
    ps = (p0 - np.pi / (2 * dx)) + (np.pi / dx / G) * np.arange(G)
    gx, gp = np.meshgrid(xs, ps, indexing="ij")  # point index = ix * G + ip (input.cpp:37-70)
    grid = np.ascontiguousarray(np.stack([gx.ravel(), gp.ravel()], axis=1))
    if kernel == "complex":  # complex labels and the reference's initial complex parameters (opt.cpp:306-332)
        y = 0.5 * y * np.exp(0.5j * (X[:, 0] - x0))
        return X, y, grid, np.array([1.0, 1.0, sx, sp, 1.0, sx, sp, 1e-2])
    return X, y, grid, np.array([1.0, sx, sp, 1e-2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1: nccl (= RCCL over xGMI, the default) or gloo (rehearsal of the "
                         "multi-rank path on fewer GPUs than ranks; ranks then share devices and gather through host memory)")
    ap.add_argument("--shard", default="grid", choices=["grid", "elements"],
                    help="--gpus > 1: 'grid' (default) = strong scaling of ONE element's step, the grid prediction split over the "
                         "ranks and all-gathered; 'elements' = weak scaling over the independent density-matrix elements, every "
                         "rank fits and predicts its own element on the whole grid, no data-path collective (SURVEY.md §8e)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gaussian_process_liouville_equation_amd as pkg
    from gaussian_process_liouville_equation_amd import _capi as c

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")

    N, G, kernel = WORKLOADS[args.workload]
    cplx = kernel == "complex"
    M = G * G
    by_element = args.shard == "elements" and world > 1
    X, y, grid, theta = synthetic(N, G, 20240607 + 1 + (rank if by_element else 0), kernel)
    # contiguous grid slice of this rank (padded to equal length so that all_gather_into_tensor applies); a rank that owns a
    # whole element predicts the whole grid
    per = M if by_element else (M + world - 1) // world
    lo, hi = (0, M) if by_element else (min(M, rank * per), min(M, (rank + 1) * per))

    stream = torch.cuda.current_stream()
    api = pkg.open_api(dev, stream=stream.cuda_stream)  # the library runs on torch's current stream
    api.enable_timing(True)
    dX = torch.from_numpy(X).cuda()
    dy = torch.from_numpy(np.ascontiguousarray(y).view(np.float64) if cplx else y).cuda()
    dgrid = torch.from_numpy(grid[lo:hi].copy()).cuda()
    # rows: mean, variance, cut-off mean (real: 1 + 1 + 1, complex: 2 + 1 + 2 doubles per point)
    out_local = torch.zeros(5 if cplx else 3, per, dtype=torch.float64, device="cuda")
    out_full = torch.zeros(world * out_local.shape[0], per, dtype=torch.float64, device="cuda") if world > 1 and not by_element else None
    dp = lambda t: C.cast(t.data_ptr(), C.POINTER(C.c_double))
    th = np.ascontiguousarray(theta)
    sc, ps = (c.ComplexFitScalars() if cplx else c.RealFitScalars()), c.PredictScalars()
    flags = c.CALC_ERROR | c.CALC_AVERAGE | c.IO_DEVICE
    thp = th.ctypes.data_as(C.POINTER(C.c_double))
    # complex outputs are interleaved (re, im) pairs: rows 0-1 / 3-4 of out_local viewed as one buffer of 2 * per doubles
    o_mean, o_var, o_cut = (out_local[0:2], out_local[2], out_local[3:5]) if cplx else (out_local[0], out_local[1], out_local[2])

    def step():
        h = C.c_void_p()
        if cplx:
            st = api.lib.gple_complex_fit_create(api.ctx, thp, dp(dX), dp(dy), N, flags, None, C.byref(h))
        else:
            st = api.lib.gple_real_fit_create(api.ctx, thp, dp(dX), dp(dy), 0, N, flags, None, C.byref(h))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())
        fn = api.lib.gple_complex_predict if cplx else api.lib.gple_real_predict
        st = fn(api.ctx, h, dp(dgrid), hi - lo, c.IO_DEVICE, None, dp(o_mean), dp(o_var), dp(o_cut), C.byref(ps))
        if st != 0:
            raise RuntimeError(api.lib.gple_ctx_last_error(api.ctx).decode())
        if world > 1 and not by_element:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(out_full, out_local)  # RCCL, ordered on the same stream as the kernels
            else:
                host = torch.empty(out_full.shape, dtype=out_full.dtype)
                dist.all_gather_into_tensor(host, out_local.cpu())
                out_full.copy_(host)
        # the fit's scalar members (error, population, <r>, purity): the one host synchronisation of the step; fit and
        # predict above only enqueue
        st = (api.lib.gple_complex_fit_get_scalars if cplx else api.lib.gple_real_fit_get_scalars)(h, C.byref(sc))
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # sanity of the gathered grid: every rank must now hold all M points (checked once, outside the timed region)
        full = out_local if by_element else out_full.view(world, out_local.shape[0], per).permute(1, 0, 2).reshape(out_local.shape[0], world * per)[:, :M]
        if not bool(torch.isfinite(full).all()):
            raise RuntimeError("gathered prediction contains non-finite values")
    ms_per_step = 1e3 * elapsed / args.steps

    _, fit_total, fit_cnt = api.timing(0)
    _, pred_total, pred_cnt = api.timing(1)
    _, pk_total, pk_cnt = api.timing(2)  # HIP events around every rownorm_kernel launch (the K* chunks of one predict)
    pk_ms = pk_total / max(1, pk_cnt)    # average duration of ONE launch (what rocprofv3 --stats reports)
    launches_per_step = pk_cnt / max(1, args.steps)
    m_local = hi - lo
    # algorithmic flops of the dominant kernel: triangular contraction ||T k*||^2 = N(N+1) flops per test point
    # (the reference's row * K^-1 * row^T form is 2 N^2 per point), split over the launches
    # complex GP = real GP on [Re; Im]: 2 typed rows per point against the 2N x 2N factor
    nn, rows = (2 * N, 2 * m_local) if cplx else (N, m_local)
    flops = float(rows) * nn * (nn + 1) / max(1.0, launches_per_step)
    achieved = flops / (pk_ms * 1e-3) / 1e12 if pk_ms > 0 else 0.0
    # HBM traffic of one rownorm_kernel launch: PMC counters cannot be read from inside the run; the value committed under
    # profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled as the gfx950 guide
    # prescribes) is reported when it was taken on this workload at this GPU count
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if tj.get("workload") == args.workload and world == 1:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
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
                                   f"grid-sharded x{world}, replicated fit, {'RCCL' if args.backend == 'nccl' else 'gloo (host)'} all-gather") if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": "rownorm_kernel (fp64 MFMA triangular contraction ||T k*||^2 over one K* chunk)",
                     "achieved": round(achieved, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_PEAK_TFLOPS, 4),
                     "traffic": traffic, "kernel_ms": round(pk_ms, 4), "launches_per_step": launches_per_step,
                     "algorithmic_flops_per_launch": flops,
                     "reference_form_flops_per_launch": (32.0 if cplx else 2.0) * m_local * N * N / max(1.0, launches_per_step)},
        "phases_ms": {"fit_device": round(fit_total / max(1, fit_cnt), 4), "predict_device": round(pred_total / max(1, pred_cnt), 4),
                      "rownorm_kernel_per_step": round(pk_total / max(1, args.steps), 4)},
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not cplx:
        from oracle import binding
        ora = binding.load()
        t0 = time.perf_counter()
        fo = ora.real_fit(theta, X, y, 3)
        tf = time.perf_counter() - t0
        # bounded sample: the full fit + a slice of the grid, extrapolated linearly in M (rows are independent)
        m_s = min(M, 16384)
        t0 = time.perf_counter()
        ora.real_predict(fo, grid[:m_s])
        tp = (time.perf_counter() - t0) * (M / m_s)
        result["cpu_baseline"] = {"value": round(1e3 * (tf + tp), 1), "unit": "ms/step", "cores": ora.num_threads, "kind": "port",
                                  "sample": f"oracle (CPU restatement, OpenMP): full fit N={N} ({tf:.2f} s) + predict on {m_s} of {M} grid points scaled to M ({tp:.2f} s)"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    api.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
