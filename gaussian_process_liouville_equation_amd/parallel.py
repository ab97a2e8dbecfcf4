"""Multi-GPU partitioning of the GPR hot path (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The reference is single-process; its two independent axes are what gets partitioned:
  * test points  — rows of K* are independent (kernel.cpp:505-515, output.cpp:204): every rank predicts one contiguous
    slice of the grid and the slices are all-gathered (mean, variance, cut-off mean: 24 bytes per point);
  * density-matrix elements — TrainingKernels builds one independent GP per element (predict.cpp:290-360): element e is
    fitted by rank e % world and only its scalars travel.
The fit itself is replicated on every rank of a grid-sharded predict (a few ms; broadcasting the N x N factor over one
xGMI link would take as long), so there is no collective on the fit path.

Everything here is plain host logic on torch tensors, so it is exercised on CPU with the gloo backend
(tests/test_distributed_gloo.py).
"""
import math

import torch
import torch.distributed as dist


def shard_bounds(M, rank, world):
    """Contiguous slice [lo, hi) of M points for `rank`; `per` is the padded slice length all ranks allocate."""
    per = int(math.ceil(M / world)) if M > 0 else 0
    lo = min(M, rank * per)
    hi = min(M, lo + per)
    return lo, hi, per


def allgather_rows(local, M, group=None):
    """local: (C, per) tensor holding this rank's slice (padded to `per`) -> (C, M) on every rank.
    One all_gather_into_tensor of C * per doubles per rank: latency-bound, far from the per-link xGMI limit."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local[:, :M]
    C, per = local.shape
    out = torch.empty(world * C, per, dtype=local.dtype, device=local.device)  # concatenated form: gloo and RCCL both take it
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out.view(world, C, per).permute(1, 0, 2).reshape(C, world * per)[:, :M]


def sharded_predict(predict_slice, grid, group=None):
    """predict_slice(points (m,2) tensor) -> (C, m) tensor (e.g. rows mean / variance / cut-off) for this rank's slice.
    Returns the full (C, M) result on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    M = grid.shape[0]
    lo, hi, per = shard_bounds(M, rank, world)
    res = predict_slice(grid[lo:hi])
    local = torch.zeros(res.shape[0], per, dtype=res.dtype, device=res.device)
    local[:, :hi - lo] = res
    return allgather_rows(local, M, group)


def cyclic_indices(M, rank, world, block=128):
    """Grid rows of `rank` under the block-cyclic deal (block b of `block` points -> rank b % world) and the padded share length
    every rank allocates.  What the C-ABI's gple_*_predict_sharded uses: with far-row pruning the live blocks of a phase-space
    grid sit in one corner of it, and contiguous slices would hold anything between all and none of them."""
    nblocks = (M + block - 1) // block
    per = ((nblocks + world - 1) // world) * block
    i = torch.arange(M)
    return i[(i // block) % world == rank], per


class GridShardedStep:
    """One fit + grid-predict step with the grid split over the ranks — the step `bench.py --gpus N` times and
    tests/test_distributed_gloo.py drives on CPU (same code, different backends plugged in).

    fit()                         -> handle (replicated on every rank: see the module docstring)
    predict_slice(h, lo, hi, out) -> fills out[:, :hi - lo] (C rows: mean, variance, cut-off mean, ...) for grid rows [lo, hi)
    `local` is this rank's (C, per) buffer, allocated once by `alloc(C, per)`; `via_host` gathers through host memory (gloo
    rehearsal with device tensors)."""

    def __init__(self, M, C, alloc, group=None, via_host=False, shard=True, cyclic=False):
        # shard=False: this rank owns a whole element and predicts the whole grid itself (nothing is gathered)
        # cyclic=True: block-cyclic shares (cyclic_indices) instead of contiguous slices; predict_slice then receives this rank's
        # index tensor instead of (lo, hi): predict_slice(h, idx, None, out) fills out[:, :len(idx)]
        self.world = dist.get_world_size(group) if (shard and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if (shard and dist.is_initialized()) else 0
        self.M, self.group, self.via_host, self.cyclic = M, group, via_host, cyclic and self.world > 1
        self.lo, self.hi, self.per = shard_bounds(M, self.rank, self.world)
        if self.cyclic:
            self.idx, self.per = cyclic_indices(M, self.rank, self.world)
            self.all_idx = [cyclic_indices(M, r, self.world)[0] for r in range(self.world)]
        self.local = alloc(C, self.per)

    def run(self, fit, predict_slice):
        h = fit()
        if self.cyclic:
            predict_slice(h, self.idx, None, self.local)
            loc = self.local.cpu() if self.via_host else self.local
            C, per = loc.shape
            out = torch.empty(self.world * C, per, dtype=loc.dtype, device=loc.device)
            dist.all_gather_into_tensor(out, loc.contiguous(), group=self.group)
            out = out.view(self.world, C, per)
            full = torch.empty(C, self.M, dtype=loc.dtype, device=loc.device)
            for r, idx in enumerate(self.all_idx):
                full[:, idx.to(loc.device)] = out[r, :, :len(idx)]
            return h, full.to(self.local.device)
        predict_slice(h, self.lo, self.hi, self.local)
        if self.world == 1:
            return h, self.local[:, :self.M]
        if self.via_host:
            return h, allgather_rows(self.local.cpu(), self.M, self.group).to(self.local.device)
        return h, allgather_rows(self.local, self.M, self.group)


def element_owner(index, world):
    """rank that fits density-matrix element number `index` (reference order: (0,0), (1,0), (1,1), ...)."""
    return index % world


def allgather_element_scalars(values, n_elements, width, group=None, device="cpu"):
    """values: {element index: 1-D sequence of `width` doubles} for the elements this rank owns -> (n_elements, width)
    tensor with every element's scalars (error, population, purity, gradients...) on every rank (one all_reduce(sum) of
    n_elements * width doubles: each slot is written by exactly one rank)."""
    buf = torch.zeros(n_elements, width, dtype=torch.float64, device=device)
    for e, v in values.items():
        buf[e] = torch.as_tensor(v, dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf
