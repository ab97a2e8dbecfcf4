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


# ---- hybrid element x grid plans (BASELINE configs[3] / [4]: 2 + 1 resp. 3 + 3 elements on 4 resp. 8 GPUs) -------------------------------
# The elements of a density matrix are independent GPs (predict.cpp:290-360) but far from equal: in the [Re; Im] embedding a complex
# element factors a 2N x 2N matrix (8 x the fit) and contracts 2M typed rows against it (8 x the predict).  Whole elements per rank leave
# 2 of 8 ranks idle at C5 and the rank with a complex element 8 x longer at work than the others; the grid of every element split over all
# ranks balances perfectly but makes every rank fit everything.  A plan says, per element, how many 128-point blocks of every cycle each rank
# predicts (the weights of gple_*_predict_dealt); a rank with weight 0 does not fit that element.  Time model of a plan:
#     T_rank = sum over elements with weight > 0 of fit_ms(e) + sum over elements of predict_ms(e) * share(e, rank) + E * gather_ms
# and the step takes max over ranks.  plan_elements builds three candidates and keeps the one with the smallest modelled step.

DEAL_BLOCK = 128    # SHARD_BLOCK of csrc/gple_capi.hip
DEAL_CYCLE = 64     # blocks per cycle of a hybrid plan's weights (8192 grid points: fine against a 512 x 512 grid's 2048 blocks)
# measured on MI355X: fit(error + averages) by padded matrix size n, contraction rate of rownorm2_kernel, K* generation bandwidth
FIT_MS_BY_N = {256: 0.093, 1024: 0.274, 2048: 0.536, 4096: 1.6, 8192: 8.35, 16384: 58.0}
CONTRACT_FLOPS = 69.3e12
KSTAR_GEN_BYTES_PER_S = 5.4e12
GATHER_MS = 0.08    # one all-gather of <= 25 MB in 8 shares: latency-bound


def model_costs(kind, N, M):
    """(fit_ms, predict_ms) of one element on one MI355X: kind "real" | "complex", N training points, M grid points, every row contracted."""
    n, rows = (2 * N, 2 * M) if kind == "complex" else (N, M)
    sizes = sorted(FIT_MS_BY_N)
    if n <= sizes[0]:
        fit = FIT_MS_BY_N[sizes[0]]
    elif n >= sizes[-1]:
        fit = FIT_MS_BY_N[sizes[-1]] * (n / sizes[-1]) ** 3
    else:  # log-log interpolation between the measured sizes
        hi = next(s for s in sizes if s >= n)
        lo = sizes[sizes.index(hi) - 1] if hi != n else hi
        t = 0.0 if hi == lo else math.log(n / lo) / math.log(hi / lo)
        fit = math.exp((1 - t) * math.log(FIT_MS_BY_N[lo]) + t * math.log(FIT_MS_BY_N[hi]))
    predict = 1e3 * (rows * float(n) * (n + 1) / CONTRACT_FLOPS + rows * float(n) * 8.0 / KSTAR_GEN_BYTES_PER_S)
    return fit, predict


def deal_shares(M, weights, block=DEAL_BLOCK):
    """points of every rank under the weighted deal of gple_*_predict_dealt (host arithmetic of deal_counts in csrc/gple_capi.hip)"""
    w = [int(x) for x in weights]
    S = sum(w)
    nblocks = (M + block - 1) // block
    cum = [0]
    for x in w:
        cum.append(cum[-1] + x)
    rem = nblocks % S
    blocks = [(nblocks // S) * w[r] + min(max(rem - cum[r], 0), w[r]) for r in range(len(w))]
    pts = [b * block for b in blocks]
    if nblocks:
        p = (nblocks - 1) % S
        owner = next(r for r in range(len(w)) if cum[r] <= p < cum[r + 1])
        pts[owner] -= nblocks * block - M
    return pts, max(blocks) * block


def dealt_indices(M, rank, weights, block=DEAL_BLOCK):
    """grid rows of `rank` under the weighted deal, in the order the rank predicts them, and the padded share length (gple_deal_share)"""
    w = [int(x) for x in weights]
    S = sum(w)
    lo = sum(w[:rank])
    nblocks = (M + block - 1) // block
    b = torch.arange(nblocks)
    p = b % S
    mine = b[(p >= lo) & (p < lo + w[rank])]
    idx = (mine[:, None] * block + torch.arange(block)[None, :]).reshape(-1)
    return idx[idx < M], deal_shares(M, w, block)[1]


class Plan:
    """weights[e][r]: blocks per cycle of element e that rank r predicts (0: r neither fits nor predicts e)"""

    def __init__(self, name, weights, costs, M):
        self.name, self.weights, self.costs, self.M = name, [list(map(int, w)) for w in weights], list(costs), M
        self.world = len(self.weights[0])
        self.rank_ms = self._model()
        self.step_ms = max(self.rank_ms)
        self.candidates = {}

    def _model(self):
        t = [len(self.weights) * GATHER_MS if self.world > 1 else 0.0] * self.world
        for (fit_ms, pred_ms), w in zip(self.costs, self.weights):
            pts, _ = deal_shares(self.M, w)
            for r in range(self.world):
                if w[r] > 0:
                    t[r] += fit_ms + pred_ms * pts[r] / max(1, self.M)
        return t

    def owner(self, e):
        """the rank whose fit scalars of element e are reported (the first one that fits it)"""
        return next(r for r, w in enumerate(self.weights[e]) if w > 0)

    def fits_of(self, rank):
        return [e for e, w in enumerate(self.weights) if w[rank] > 0]

    def describe(self):
        return {"name": self.name, "weights": self.weights, "modelled_rank_ms": [round(x, 3) for x in self.rank_ms], "modelled_step_ms": round(self.step_ms, 3),
                "candidates_modelled_step_ms": {k: round(v, 3) for k, v in self.candidates.items()}}


def _quantise(fracs, cycle):
    """fractions (summing to 1) -> integer weights summing to `cycle` by largest remainders; slivers below half a block vanish"""
    raw = [f * cycle for f in fracs]
    w = [int(math.floor(x)) for x in raw]
    order = sorted(range(len(raw)), key=lambda i: raw[i] - w[i], reverse=True)
    for i in order[:cycle - sum(w)]:
        w[i] += 1
    return w


def _hybrid_fractions(costs, world):
    """Lay the elements (most expensive first) on a line and cut it into `world` stretches of equal modelled time, a rank paying the fit of every
    element it touches: bisection on the stretch length.  Returns fracs[e][r]."""
    order = sorted(range(len(costs)), key=lambda e: -(costs[e][0] + costs[e][1]))

    def fill(tau):
        fr = [[0.0] * world for _ in costs]
        r, cap = 0, tau
        for e in order:
            fit_ms, pred_ms = costs[e]
            left = 1.0
            while left > 1e-12:
                if r >= world:
                    return None
                room = (cap - fit_ms) / pred_ms if pred_ms > 0 else 1.0
                if room < min(left, 0.02):  # not worth a fit: next rank
                    r, cap = r + 1, tau
                    continue
                take = min(left, room)
                fr[e][r] += take
                left -= take
                cap -= fit_ms + take * pred_ms
        return fr
    lo, hi = 0.0, sum(f + p for f, p in costs) + 1.0
    best = fill(hi)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        got = fill(mid)
        if got is None:
            lo = mid
        else:
            hi, best = mid, got
    return best


def plan_elements(costs, world, M, cycle=DEAL_CYCLE):
    """costs[e] = (fit_ms, predict_ms) of element e alone on one GPU (model_costs or measured) -> the Plan with the smallest modelled step among
       "elements": whole elements dealt to the ranks, longest first (no data-path collective: weights 0 / 1);
       "grid":     every element's grid dealt evenly over all ranks, fits replicated;
       "hybrid":   equal-time stretches of the line of elements (the expensive elements are grid-sharded over a few ranks each, the cheap ones
                   stay whole)."""
    E = len(costs)
    if world == 1:
        p = Plan("single", [[1]] * E, costs, M)
        p.candidates = {"single": p.step_ms}
        return p
    load, elem_w = [0.0] * world, [[0] * world for _ in range(E)]
    for e in sorted(range(E), key=lambda e: -(costs[e][0] + costs[e][1])):  # LPT
        r = min(range(world), key=lambda r: load[r])
        load[r] += costs[e][0] + costs[e][1]
        elem_w[e][r] = 1
    cands = [Plan("elements", elem_w, costs, M), Plan("grid", [[1] * world] * E, costs, M)]
    fr = _hybrid_fractions(costs, world)
    if fr is not None:
        cands.append(Plan("hybrid", [_quantise(f, cycle) for f in fr], costs, M))
    best = min(cands, key=lambda p: p.step_ms * (1.0 + 0.01 * cands.index(p)))  # within 1 %: the simpler plan
    best.candidates = {p.name: p.step_ms for p in cands}
    return best


def gather_dealt(local, M, weights, group=None, via_host=False):
    """torch.distributed transport of the weighted deal (gloo rehearsal and bench.py --via torch; the product's own is ncclAllGather inside
    gple_*_predict_dealt): local (C, per) holds this rank's share in dealt order -> (C, M) on every rank."""
    world = len(weights)
    loc = local.cpu() if via_host else local
    C, per = loc.shape
    out = torch.empty(world * C, per, dtype=loc.dtype, device=loc.device)
    dist.all_gather_into_tensor(out, loc.contiguous(), group=group)
    out = out.view(world, C, per)
    full = torch.empty(C, M, dtype=loc.dtype, device=loc.device)
    for r in range(world):
        if weights[r] > 0:
            idx = dealt_indices(M, r, weights)[0].to(loc.device)
            full[:, idx] = out[r, :, :len(idx)]
    return full.to(local.device)


class HybridStep:
    """One fit + grid-predict step of ALL elements of a density matrix under a Plan — what `bench.py --workload C4 | C5` times and
    tests/test_distributed_gloo.py drives on CPU with the oracle plugged in.
      fit(e)                         -> handle of element e (called only on ranks whose weight for e is positive)
      predict_dealt(e, h, weights)   -> the element's full (C, M) result on every rank; h is None on ranks without a share.  bench.py plugs in
                                        gple_*_predict_dealt (RCCL inside the library), the gloo test the oracle + gather_dealt.
    Every rank walks the elements in the same order: the all-gathers of one communicator must be entered in the same order everywhere."""

    def __init__(self, plan, rank):
        self.plan, self.rank = plan, rank

    def run(self, fit, predict_dealt):
        handles, outs = [], []
        for e, w in enumerate(self.plan.weights):
            h = fit(e) if w[self.rank] > 0 else None
            handles.append(h)
            outs.append(predict_dealt(e, h, w))
        return handles, outs
