"""Multi-GPU sharding of the batched 2-opt: one process per GPU, torch.distributed (RCCL on ROCm).

The path shards by construction: tours are independent (``group_size == 1``) or
independent groups of ``P`` tours (coupled mode); an instance's tours stay on one rank so each rank
only touches its own instances' coordinates.  There is NO per-iteration collective: every rank runs
the persistent kernel on its block of instances, then ONE ``all_gather`` of tours, iteration counts and
lengths ends the call (TSP-500 x 2048: 8.2 MB in total).

Two input forms:
  * replicated (default): every rank holds the whole batch and solves the instances ``assign_units`` gives
    it -- contiguous blocks (default), interleaved, or balanced by a cost proxy (iteration
    counts vary 2-3x between greedy-start tours, and the slowest rank sets the time);
  * ``local_inputs=True``: every rank passes ONLY its own block ("each GPU gets its instances'
    coordinates only"); the whole batch is the concatenation of the blocks in rank order.

The reference has no counterpart (its only distributed mechanism is lightning DDP for training,
models/modelzoo/base/trainer.py:31); this module is the "final gather of tours and lengths" that
BASELINE.json's north_star asks for.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class ShardedResult:
    """solve_sharded(..., want_result=True): the whole batch on every rank + this rank's own block."""
    tours: torch.Tensor         # (T, N+1) int64, whole batch
    iterations: torch.Tensor    # (T,) int32, whole batch
    lengths: torch.Tensor       # (T,) float64, whole batch
    local: Any                  # the solver's result for this rank's block (TwoOptResult)
    block: Tuple[int, int]      # [first, last) tour of this rank's block (contiguous forms; else (-1, -1))
    tour_index: Optional[torch.Tensor] = None   # (T_local,) int64: whole-batch row of every local tour


def partition(num_units: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [start, stop) blocks of units (instances) per rank; the first
    ``num_units % world`` ranks get one extra unit."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(num_units, world)
    out, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


def assign_units(num_units: int, world: int, balance: str = "contiguous",
                 cost: Optional[Sequence[float]] = None) -> List[List[int]]:
    """Units (instances) of every rank, each list ascending; rank sizes differ by at most one.

    ``contiguous``: the blocks of ``partition`` (an instance's neighbours stay together: the default, and the
    only form ``local_inputs`` can have).  ``interleaved``: unit u goes to rank u % world -- batches whose cost
    drifts along the batch (sorted by size or by decoder) spread evenly.  ``proxy``: longest-processing-time
    greedy on ``cost`` (one number per unit, e.g. the start tours' length: the further a start tour is from a
    local optimum the more iterations it takes): units by decreasing cost, each to the least loaded rank that
    still has room."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if balance == "contiguous":
        return [list(range(a, b)) for a, b in partition(num_units, world)]
    if balance == "interleaved":
        return [list(range(r, num_units, world)) for r in range(world)]
    if balance == "proxy":
        if cost is None or len(cost) != num_units:
            raise ValueError("balance='proxy' needs one cost per unit")
        room = [b - a for a, b in partition(num_units, world)]      # same sizes as the contiguous blocks
        load = [0.0] * world
        out: List[List[int]] = [[] for _ in range(world)]
        for u in sorted(range(num_units), key=lambda k: (-float(cost[k]), k)):
            r = min((q for q in range(world) if len(out[q]) < room[q]), key=lambda q: (load[q], q))
            out[r].append(u)
            load[r] += float(cost[u])
        return [sorted(x) for x in out]
    raise ValueError(f"unknown balance {balance!r}")


def start_cost(points: torch.Tensor, tours: torch.Tensor, tours_per_instance: int) -> torch.Tensor:
    """Cost proxy of every instance for ``balance='proxy'``: the summed length of its start tours (float64)."""
    inst, n = points.shape[0], points.shape[1]
    idx = torch.arange(tours.shape[0], device=tours.device) // tours_per_instance
    p = points.to(torch.float64)[idx.unsqueeze(1), tours]               # (T, N+1, 2)
    ln = (p[:, 1:] - p[:, :-1]).norm(dim=-1).sum(dim=1)
    return ln.view(inst, tours_per_instance).sum(dim=1)


def solve_sharded(points: torch.Tensor, tours: torch.Tensor, *, tours_per_instance: int = 1,
                  group_size: int = 1, max_iterations: int = 5000,
                  solver: Optional[Callable] = None, process_group=None, want_result: bool = False,
                  balance: str = "contiguous", cost: Optional[torch.Tensor] = None,
                  local_inputs: bool = False):
    """Solve one batch over all ranks, each rank doing its share of the instances.

    Replicated form: points (I, N, 2), tours (I * tours_per_instance, N+1) -- identical on every rank and on
    the rank's own device; ``balance`` picks the assignment (``assign_units``; for ``proxy`` the cost is
    ``cost`` or the start tours' lengths, taken from rank 0 so that every rank derives the same assignment).
    ``local_inputs=True``: points / tours are THIS rank's block only (any number of instances, also none); the
    whole batch is the blocks in rank order.
    ``solver(points, tours, tours_per_instance=, group_size=, max_iterations=)`` must return an object with
    ``tours``, ``iterations``, ``lengths`` tensors (default: ``two_opt_device``).  Returns ``(tours (T, N+1)
    int64, iterations (T,) int32, lengths (T,) float64)`` for the WHOLE batch on every rank
    (``want_result=True``: a ``ShardedResult`` that also carries this rank's own solver result).
    """
    if solver is None:
        from .two_opt import two_opt_device
        solver = two_opt_device
    if tours_per_instance % group_size != 0:
        raise ValueError("group_size must divide tours_per_instance (a group stays on one rank)")
    on = dist.is_initialized()
    world = dist.get_world_size(process_group) if on else 1
    rank = dist.get_rank(process_group) if on else 0
    inst = points.shape[0]
    T, n1 = tours.shape
    if T != inst * tours_per_instance:
        raise ValueError("tours must hold tours_per_instance rows per instance")
    dev = tours.device
    tpi = tours_per_instance

    if local_inputs:
        # block sizes of all ranks: one tiny all_gather (control plane), then the blocks line up in rank order
        counts = [inst]
        if world > 1:
            c = torch.tensor([inst], dtype=torch.int64, device=dev)
            allc = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allc, c, group=process_group)
            counts = [int(x) for x in allc.tolist()]
        starts = [sum(counts[:r]) for r in range(world)]
        units = [list(range(starts[r], starts[r] + counts[r])) for r in range(world)]
        my_pts, my_tours = points, tours
        contiguous = True
    else:
        if balance == "proxy" and world > 1:
            c = (cost if cost is not None else start_cost(points, tours, tpi)).to(device=dev, dtype=torch.float64)
            c = c.contiguous()
            dist.broadcast(c, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                           group=process_group)
            units = assign_units(inst, world, "proxy", c.tolist())
        elif balance == "proxy":
            units = assign_units(inst, world, "contiguous")
        else:
            units = assign_units(inst, world, balance)
        contiguous = balance == "contiguous" or world == 1
        mine = units[rank]
        if contiguous:
            i0, i1 = (mine[0], mine[-1] + 1) if mine else (0, 0)
            my_pts, my_tours = points[i0:i1], tours[i0 * tpi:i1 * tpi]
        else:
            ui = torch.tensor(mine, dtype=torch.int64, device=dev)
            ti = (ui.unsqueeze(1) * tpi + torch.arange(tpi, device=dev)).reshape(-1)
            my_pts, my_tours = points.index_select(0, ui.to(points.device)), tours.index_select(0, ti)

    res = solver(my_pts, my_tours, tours_per_instance=tpi, group_size=group_size, max_iterations=max_iterations)

    def tour_rows(us: List[int]) -> torch.Tensor:
        u = torch.tensor(us, dtype=torch.int64, device=dev)
        return (u.unsqueeze(1) * tpi + torch.arange(tpi, device=dev)).reshape(-1)

    mine = units[rank]
    blk = (mine[0] * tpi, (mine[-1] + 1) * tpi) if (contiguous and mine) else ((0, 0) if contiguous else (-1, -1))
    my_rows = tour_rows(mine)
    if world == 1:
        if want_result:
            return ShardedResult(res.tours, res.iterations, res.lengths, res, blk, my_rows)
        return res.tours, res.iterations, res.lengths
    # one padded all_gather per output (the ranks' shares may differ in size)
    cap = max(len(u) for u in units) * tpi
    total = sum(len(u) for u in units) * tpi
    order = torch.cat([tour_rows(u) for u in units])          # whole-batch row of every gathered row

    def gather(x: torch.Tensor, fill=0):
        pad = torch.full((cap,) + tuple(x.shape[1:]), fill, dtype=x.dtype, device=dev)
        pad[: x.shape[0]] = x
        out = torch.empty((world * cap,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        dist.all_gather_into_tensor(out, pad, group=process_group)
        parts = torch.cat([out[r * cap: r * cap + len(u) * tpi] for r, u in enumerate(units)], dim=0)
        if contiguous:
            return parts                                       # blocks in rank order ARE the batch order
        whole = torch.empty((total,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        whole[order] = parts
        return whole

    out = (gather_tours(res.tours.to(dev), gather), gather(res.iterations.to(dev)),
           gather(res.lengths.to(dev)))
    if want_result:
        return ShardedResult(out[0], out[1], out[2], res, blk, my_rows)
    return out


def pack_tours(tours: torch.Tensor) -> torch.Tensor:
    """(T, N+1) int64 node indices -> the bytes that travel in the final gather: int16 when every
    index fits (N <= 32767; a quarter of the int64 volume), as a uint8 view so that any backend's
    all_gather takes it."""
    dt = torch.int16 if tours.shape[1] - 1 <= 32767 else torch.int64
    packed = torch.empty(tuple(tours.shape), dtype=dt, device=tours.device)    # fresh, contiguous strides
    packed.copy_(tours)
    return packed.view(torch.uint8)


def unpack_tours(packed: torch.Tensor, n1: int) -> torch.Tensor:
    """Inverse of pack_tours for tours of n1 = N+1 entries."""
    if n1 - 1 <= 32767:
        return packed.view(torch.int16).to(torch.int64)
    return packed.view(torch.int64)


def gather_tours(tours: torch.Tensor, gather: Callable) -> torch.Tensor:
    return unpack_tours(gather(pack_tours(tours)), tours.shape[1])
