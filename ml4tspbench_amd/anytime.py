"""Anytime (time-sliced) 2-opt: tour lengths at regular checkpoints of one local search.

Reference (read-only): ``docs/gap_time.py:257-327`` ``two_opt_gap_time`` -- the gap-versus-time study runs
the 2-opt loop in ``record_samples`` wall-clock slices of ``limit_time / record_samples`` seconds and
records, after every slice, the elapsed time and the current tour length (``TSPEvaluator.evaluate``).

Best-improvement 2-opt is a deterministic function of the current tour, so stopping after any iteration
and calling again continues the very same trajectory.  A checkpointed search is therefore a sequence of
``tsp2opt_solve`` calls on the same device-resident tours -- each limited by the slice's wall-clock time
(``time_limit_s``, checked once per iteration inside the persistent kernel) or, for a reproducible
trajectory, by a number of iterations -- with the lengths read after every call.  Nothing leaves the GPU
between slices except the (T,) lengths.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from .two_opt import two_opt_device


@dataclass
class AnytimeResult:
    tours: torch.Tensor         # (T, N+1) int64, state after the last slice
    lengths: torch.Tensor       # (T, S) float64: closed-tour length after every slice
    iterations: torch.Tensor    # (T, S) int32: cumulative iterations after every slice
    times: torch.Tensor         # (S,) float64: cumulative wall-clock seconds after every slice (host clock)
    converged: torch.Tensor     # (T,) bool: the tour reached its 2-opt local optimum


def two_opt_anytime(points: torch.Tensor, tours: torch.Tensor, *, record_samples: int = 10,
                    limit_time: Optional[float] = 1.0, slice_iterations: Optional[int] = None,
                    tours_per_instance: int = 1) -> AnytimeResult:
    """Checkpointed 2-opt (independent termination per tour, arithmetic in the dtype of ``points``).

    ``slice_iterations = k``: every slice runs at most ``k`` iterations per tour (deterministic: the
    lengths after slice ``s`` are those of a plain run capped at ``k * (s + 1)`` iterations).
    Otherwise every slice runs for ``limit_time / record_samples`` seconds per tour, like
    ``docs/gap_time.py:264,283-284``.
    """
    if record_samples < 1:
        raise ValueError("record_samples must be >= 1")
    if slice_iterations is None and (limit_time is None or limit_time <= 0):
        raise ValueError("give limit_time > 0 or slice_iterations")
    if tours.dim() == 1:
        tours = tours.unsqueeze(0)
    T = tours.shape[0]
    dev = points.device
    cur = tours.to(torch.int64).clone(memory_format=torch.contiguous_format)
    lengths = torch.empty((T, record_samples), dtype=torch.float64, device=dev)
    iters = torch.zeros((T, record_samples), dtype=torch.int32, device=dev)
    times = torch.zeros(record_samples, dtype=torch.float64)
    done = torch.zeros(T, dtype=torch.bool, device=dev)
    total = torch.zeros(T, dtype=torch.int32, device=dev)
    t0 = time.perf_counter()
    for s in range(record_samples):
        if slice_iterations is not None:
            res = two_opt_device(points, cur, tours_per_instance=tours_per_instance,
                                 max_iterations=int(slice_iterations), inplace=True)
        else:
            res = two_opt_device(points, cur, tours_per_instance=tours_per_instance, max_iterations=(1 << 31) - 1,
                                 time_limit=limit_time / record_samples, inplace=True)
        cur = res.tours
        stopped_early = (res.status & (_lib.ST_CAPPED | _lib.ST_TIME)) != 0
        done |= ~stopped_early
        total = total + res.iterations
        lengths[:, s] = res.lengths
        iters[:, s] = total
        torch.cuda.synchronize(dev)
        times[s] = time.perf_counter() - t0
    return AnytimeResult(cur, lengths, iters, times, done)
