"""CPU, world_size 2 (gloo): the N > 1 partitioning logic of gaussian_process_liouville_equation_amd.parallel.
The predictor plugged in here is the CPU oracle; on the GPU box the same host code runs with the HIP predictor and
backend "nccl" (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussian_process_liouville_equation_amd import parallel
from tests import parity


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPLE_ORACLE_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding
    ora = binding.load()
    X, y, Xs = parity.synthetic_real(48, M, 77)
    theta = [1.0, 0.7086, 0.7056, 1e-2]
    fit = ora.real_fit(theta, X, y, 3)  # replicated fit, like bench.py

    def predict_slice(points):
        p = ora.real_predict(fit, points.numpy())
        return torch.from_numpy(np.stack([p["prediction"], p["variance"], p["cutoff"]]))

    full = parallel.sharded_predict(predict_slice, torch.from_numpy(Xs))

    # the step function bench.py --gpus N times (parallel.GridShardedStep), with the oracle plugged in for fit and predict
    def fill(h, lo, hi, out):
        p = ora.real_predict(h, Xs[lo:hi])
        out[:, :hi - lo] = torch.from_numpy(np.stack([p["prediction"], p["variance"], p["cutoff"]]))

    step = parallel.GridShardedStep(M, 3, lambda C, per: torch.zeros(C, per, dtype=torch.float64))
    h, full_step = step.run(lambda: ora.real_fit(theta, X, y, 3), fill)
    assert (step.lo, step.hi, step.per) == parallel.shard_bounds(M, rank, world)
    np.save(os.path.join(out_dir, f"step_{rank}.npy"), full_step.numpy())

    # block-cyclic shares (what the pruned predict wants): every rank predicts its index set, the shares are scattered back
    def fill_idx(h, idx, _, out):
        p = ora.real_predict(h, Xs[idx.numpy()])
        out[:, :len(idx)] = torch.from_numpy(np.stack([p["prediction"], p["variance"], p["cutoff"]]))

    cyc = parallel.GridShardedStep(M, 3, lambda C, per: torch.zeros(C, per, dtype=torch.float64), cyclic=True)
    _, full_cyc = cyc.run(lambda: ora.real_fit(theta, X, y, 3), fill_idx)
    np.save(os.path.join(out_dir, f"cyc_{rank}.npy"), full_cyc.numpy())
    # per-element scalars: element e is owned by rank e % world
    mine = {e: [float(e), fit.scalars["population"] * (e + 1)] for e in range(3) if parallel.element_owner(e, world) == rank}
    scal = parallel.allgather_element_scalars(mine, 3, 2)
    np.save(os.path.join(out_dir, f"full_{rank}.npy"), full.numpy())
    np.save(os.path.join(out_dir, f"scal_{rank}.npy"), scal.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("M", [37, 64, 300])
def test_grid_sharded_predict_matches_unsharded(tmp_path, M):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), M, str(tmp_path)), nprocs=world, join=True)
    from oracle import binding
    ora = binding.load()
    X, y, Xs = parity.synthetic_real(48, M, 77)
    fit = ora.real_fit([1.0, 0.7086, 0.7056, 1e-2], X, y, 3)
    p = ora.real_predict(fit, Xs)
    ref = np.stack([p["prediction"], p["variance"], p["cutoff"]])
    for r in range(world):
        got = np.load(tmp_path / f"full_{r}.npy")
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)  # same code on the same rows: bit-identical, including the padded tail handling
        assert np.array_equal(np.load(tmp_path / f"step_{r}.npy"), ref)  # the bench's step function gives the same grid
        assert np.array_equal(np.load(tmp_path / f"cyc_{r}.npy"), ref)   # and so does the block-cyclic deal
        scal = np.load(tmp_path / f"scal_{r}.npy")
        assert np.allclose(scal[:, 0], [0.0, 1.0, 2.0]) and np.allclose(scal[:, 1], fit.scalars["population"] * np.array([1, 2, 3]))


def test_cyclic_indices_cover_everything_once():
    for M in (1, 127, 128, 129, 1000, 65537):
        for world in (1, 2, 3, 8):
            parts = [parallel.cyclic_indices(M, r, world) for r in range(world)]
            assert sorted(torch.cat([p[0] for p in parts]).tolist()) == list(range(M))
            assert all(len(p[0]) <= p[1] and p[1] == parts[0][1] for p in parts)


def test_shard_bounds_cover_everything_once():
    for M in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi, per = parallel.shard_bounds(M, r, world)
                assert hi - lo <= per
                seen += list(range(lo, hi))
            assert seen == list(range(M))
