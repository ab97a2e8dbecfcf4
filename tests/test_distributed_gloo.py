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


# ---- hybrid element x grid plans ------------------------------------------------------------------------------------------------------------

def test_weighted_deal_host_mirror_matches_the_library():
    """parallel.dealt_indices / deal_shares against gple_deal_share (host arithmetic of the library's weighted deal, no device call)"""
    import ctypes as C
    import gaussian_process_liouville_equation_amd as pkg
    lib = pkg.load_library()
    lib.gple_deal_share.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int)] + [C.POINTER(C.c_size_t)] * 3
    for M in (1, 127, 128, 129, 1000, 8192 + 77, 65537):
        for weights in ([1], [1, 1], [1, 1, 1], [3, 0, 5], [0, 0, 64], [27, 27, 10, 0, 0, 0, 0, 0], [0, 44, 20, 0]):
            world = len(weights)
            w = (C.c_int * world)(*weights)
            seen = []
            pts, per = parallel.deal_shares(M, weights)
            for r in range(world):
                nl, pr = C.c_size_t(), C.c_size_t()
                assert lib.gple_deal_share(M, r, world, w, C.byref(nl), C.byref(pr), None) == 0
                idx = (C.c_size_t * max(1, nl.value))()
                assert lib.gple_deal_share(M, r, world, w, None, None, idx) == 0
                mine, per_py = parallel.dealt_indices(M, r, weights)
                assert list(idx[:nl.value]) == mine.tolist() and pr.value == per_py == per and nl.value == pts[r] <= per
                if weights[r] == 0:
                    assert nl.value == 0
                seen += mine.tolist()
            assert sorted(seen) == list(range(M))
    # the plain deal of gple_*_predict_sharded is the all-ones case
    for world in (2, 3, 8):
        for r in range(world):
            assert torch.equal(parallel.dealt_indices(1000, r, [1] * world)[0], parallel.cyclic_indices(1000, r, world)[0])
    assert lib.gple_deal_share(100, 0, 2, (C.c_int * 2)(0, 0), None, None, None) != 0  # nobody predicts
    assert lib.gple_deal_share(100, 2, 2, (C.c_int * 2)(1, 1), None, None, None) != 0


def test_planner_candidates_and_model():
    kinds5 = ["real", "complex", "real", "complex", "complex", "real"]
    c5 = [parallel.model_costs(k, 8192, 1024 * 1024) for k in kinds5]
    c4 = [parallel.model_costs(k, 4096, 512 * 512) for k in ["real", "complex", "real"]]
    assert 7.5 < c4[1][1] / c4[0][1] < 8.5 and abs(c4[0][1] - 65.0) < 3.0  # complex element = 8 x the real one; C4r contraction + generation as measured
    for costs, world, M in ((c4, 4, 512 * 512), (c5, 8, 1024 * 1024), (c5, 4, 1024 * 1024), (c4, 2, 512 * 512), (c4, 1, 512 * 512), (c5, 3, 1024 * 1024)):
        plan = parallel.plan_elements(costs, world, M)
        total = sum(f + p for f, p in costs)
        assert plan.step_ms >= total / world * 0.999  # never better than perfect balance without replicated fits
        assert plan.step_ms == min(plan.candidates.values()) or plan.step_ms <= 1.02 * min(plan.candidates.values())
        for e, w in enumerate(plan.weights):
            assert len(w) == world and sum(w) > 0 and all(x >= 0 for x in w)
            assert 0 <= plan.owner(e) < world and w[plan.owner(e)] > 0
        if world > 1:
            assert set(plan.candidates) == {"elements", "grid", "hybrid"}
            # whole elements leave the rank with a complex element alone with it — unless elements and ranks pair up (3 + 3 elements on 3 ranks)
            if (len(costs), world) == (6, 3):
                assert plan.name == "elements"
            else:
                assert plan.candidates["elements"] > 1.15 * plan.step_ms
            # modelled time of a rank = its fits + its shares of the predicts + the gathers
            r = 0
            t = len(costs) * parallel.GATHER_MS
            for (f, p), w in zip(costs, plan.weights):
                if w[r]:
                    t += f + p * parallel.deal_shares(M, w)[0][r] / M
            assert abs(t - plan.rank_ms[r]) < 1e-9
    p8 = parallel.plan_elements(c5, 8, 1024 * 1024)
    assert p8.name == "hybrid" and max(len(p8.fits_of(r)) for r in range(8)) <= 4 and p8.step_ms < 1.08 * sum(f + p for f, p in c5) / 8


def _hybrid_worker(rank, world, port, out_dir, weights_by_element):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPLE_ORACLE_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding
    ora = binding.load()
    M = 700
    kinds = ["real", "complex", "real"]
    costs = [parallel.model_costs(k, 4096, M) for k in kinds]
    plan = parallel.Plan("forced", weights_by_element, costs, M) if weights_by_element else parallel.plan_elements(costs, world, M)
    sets = []
    for e, k in enumerate(kinds):
        X, yr, Xs = parity.synthetic_real(40 + 3 * e, M, 90 + e)
        y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0)) if k == "complex" else yr
        theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05] if k == "complex" else [1.0, 0.7086, 0.7056, 1e-2]
        sets.append((k, X, y, theta))
    _, _, Xs = parity.synthetic_real(40, M, 90)
    fits_made = []

    def fit(e):
        k, X, y, theta = sets[e]
        fits_made.append(e)
        return (ora.complex_fit if k == "complex" else ora.real_fit)(theta, X, y, 3)

    def predict_dealt(e, h, weights):
        idx, per = parallel.dealt_indices(M, rank, weights)
        cplx = sets[e][0] == "complex"
        local = torch.zeros(5 if cplx else 3, per, dtype=torch.float64)
        assert (h is None) == (weights[rank] == 0)
        if h is not None and len(idx):
            p = (ora.complex_predict if cplx else ora.real_predict)(h, Xs[idx.numpy()])
            rows = [p["prediction"].real, p["prediction"].imag, p["variance"], p["cutoff"].real, p["cutoff"].imag] if cplx else [p["prediction"], p["variance"], p["cutoff"]]
            local[:, :len(idx)] = torch.from_numpy(np.stack(rows))
        return parallel.gather_dealt(local, M, weights)

    handles, outs = parallel.HybridStep(plan, rank).run(fit, predict_dealt)
    assert fits_made == plan.fits_of(rank)
    vals = {e: [h.scalars["error"], h.scalars["purity"]] for e, h in enumerate(handles) if h is not None and plan.owner(e) == rank}
    scal = parallel.allgather_element_scalars(vals, len(kinds), 2)
    np.savez(os.path.join(out_dir, f"hyb_{rank}.npz"), scal=scal.numpy(), **{f"e{e}": o.numpy() for e, o in enumerate(outs)})
    dist.destroy_process_group()


@pytest.mark.parametrize("weights", [None, [[3, 0], [1, 2], [0, 1]]])
def test_hybrid_step_world2_matches_unsharded(tmp_path, weights):
    """the step bench.py --workload C4 | C5 times (parallel.HybridStep under a Plan), world 2 over gloo with the oracle plugged in: the planner's own
    plan, and a forced one in which each rank skips an element entirely (no fit there, still in every gather)"""
    world = 2
    mp.spawn(_hybrid_worker, args=(world, _free_port(), str(tmp_path), weights), nprocs=world, join=True)
    from oracle import binding
    ora = binding.load()
    M = 700
    _, _, Xs = parity.synthetic_real(40, M, 90)
    got = [np.load(tmp_path / f"hyb_{r}.npz") for r in range(world)]
    for e, k in enumerate(["real", "complex", "real"]):
        X, yr, _ = parity.synthetic_real(40 + 3 * e, M, 90 + e)
        cplx = k == "complex"
        y = 0.5 * yr * np.exp(0.5j * (X[:, 0] + 10.0)) if cplx else yr
        theta = [1.0, 1.1, 0.8, 0.7, 0.9, 0.7, 0.8, 0.05] if cplx else [1.0, 0.7086, 0.7056, 1e-2]
        f = (ora.complex_fit if cplx else ora.real_fit)(theta, X, y, 3)
        p = (ora.complex_predict if cplx else ora.real_predict)(f, Xs)
        ref = np.stack([p["prediction"].real, p["prediction"].imag, p["variance"], p["cutoff"].real, p["cutoff"].imag] if cplx else [p["prediction"], p["variance"], p["cutoff"]])
        for r in range(world):
            # the oracle's predict is row-wise: any split of the rows gives the same bits
            assert np.array_equal(got[r][f"e{e}"], ref)
            assert np.allclose(got[r]["scal"][e], [f.scalars["error"], f.scalars["purity"]], rtol=1e-13)
