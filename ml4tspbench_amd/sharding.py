"""Multi-GPU sharding of the batched 2-opt: one process per GPU, torch.distributed (RCCL on ROCm).

The path shards by construction: tours are independent (``group_size == 1``) or
independent groups of ``P`` tours (coupled mode); an instance's tours stay on one rank so each rank
only touches its own instances' coordinates.  There is NO per-iteration collective: every rank runs
the persistent kernel on its contiguous block of instances, then ONE ``all_gather`` of tours,
iteration counts and lengths ends the call (TSP-500 x 2048: 8.2 MB in total).

The reference has no counterpart (its only distributed mechanism is lightning DDP for training,
models/modelzoo/base/trainer.py:31); this module is the "final gather of tours and lengths" that
BASELINE.json's north_star asks for.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


@dataclass
class ShardedResult:
    """solve_sharded(..., want_result=True): the whole batch on every rank + this rank's own block."""
    tours: torch.Tensor         # (T, N+1) int64, whole batch
    iterations: torch.Tensor    # (T,) int32, whole batch
    lengths: torch.Tensor       # (T,) float64, whole batch
    local: Any                  # the solver's result for this rank's block (TwoOptResult)
    block: Tuple[int, int]      # [first, last) tour of this rank's block


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


def solve_sharded(points: torch.Tensor, tours: torch.Tensor, *, tours_per_instance: int = 1,
                  group_size: int = 1, max_iterations: int = 5000,
                  solver: Optional[Callable] = None, process_group=None, want_result: bool = False):
    """Solve a batch that every rank holds, each rank doing its block of instances.

    points (I, N, 2), tours (I * tours_per_instance, N+1) -- identical on every rank and on the
    rank's own device.  ``solver(points, tours, tours_per_instance=, group_size=,
    max_iterations=)`` must return an object with ``tours``, ``iterations``, ``lengths`` tensors
    (default: ``two_opt_device``).  Returns ``(tours (T, N+1) int64, iterations (T,) int32,
    lengths (T,) float64)`` for the WHOLE batch on every rank (``want_result=True``: a
    ``ShardedResult`` that also carries this rank's own solver result).
    """
    if solver is None:
        from .two_opt import two_opt_device
        solver = two_opt_device
    if tours_per_instance % group_size != 0:
        raise ValueError("group_size must divide tours_per_instance (a group stays on one rank)")
    world = dist.get_world_size(process_group) if dist.is_initialized() else 1
    rank = dist.get_rank(process_group) if dist.is_initialized() else 0
    inst = points.shape[0]
    T, n1 = tours.shape
    if T != inst * tours_per_instance:
        raise ValueError("tours must hold tours_per_instance rows per instance")
    blocks = partition(inst, world)
    i0, i1 = blocks[rank]
    res = solver(points[i0:i1], tours[i0 * tours_per_instance:i1 * tours_per_instance],
                 tours_per_instance=tours_per_instance, group_size=group_size,
                 max_iterations=max_iterations)
    blk = (i0 * tours_per_instance, i1 * tours_per_instance)
    if world == 1:
        if want_result:
            return ShardedResult(res.tours, res.iterations, res.lengths, res, blk)
        return res.tours, res.iterations, res.lengths
    # one padded all_gather per output (blocks differ by at most one instance)
    cap = max(b - a for a, b in blocks) * tours_per_instance
    dev = tours.device

    def gather(x: torch.Tensor, fill=0):
        pad = torch.full((cap,) + tuple(x.shape[1:]), fill, dtype=x.dtype, device=dev)
        pad[: x.shape[0]] = x
        out = torch.empty((world * cap,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)
        dist.all_gather_into_tensor(out, pad, group=process_group)
        parts = [out[r * cap: r * cap + (b - a) * tours_per_instance]
                 for r, (a, b) in enumerate(blocks)]
        return torch.cat(parts, dim=0)

    out = (gather_tours(res.tours.to(dev), gather), gather(res.iterations.to(dev)),
           gather(res.lengths.to(dev)))
    if want_result:
        return ShardedResult(out[0], out[1], out[2], res, blk)
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
