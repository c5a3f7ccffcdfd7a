"""Host-side mirror of the reference's other local searches.

Reference interface (read-only):
  * ``search/relocate.py:40-61``      ``tsp_relocate`` -- one best relocate move per tour
  * ``search/local_search.py:137-154`` ``tsp_ls``      -- two-opt + relocate until no improvement
  * ``search/local_search.py:157-232`` ``tsp_gls``     -- (regret) guided local search, wall-clock limited
  * ``ml4tsp-library/ml4tsp/nar/local_search/guided_ls.py:10-89`` ``ML4TSPNARGuidedLS`` -- the same as a plugin
registered as ``LOCAL_SEARCH_CLASS[('tsp', 'relocate' | 'ls' | 'gls')]`` (``search/__init__.py:45-52``).

Same names, arguments and return values.  The O(N^2) candidate scans run in ``csrc/ls_ops.hip``
(``tsp2opt_ls_step`` / ``tsp2opt_local_search``), all tours of a call at once, in exact double arithmetic
in the reference's operation order, so tours are bit-identical to the reference's for float64 points
(the reference computes in the dtype ``scipy.spatial.distance_matrix`` returns; float32 points are
up-cast here).  The guided search keeps its perturbation bookkeeping (edge utilities, penalties, cost
comparison) on the host exactly like the reference and calls the device operators for the move scans.
There is no CPU path.
"""
from __future__ import annotations

import time
from typing import Optional

import numpy as np
import torch

from . import _lib
from .local_search import ML4TSPNARLocalSearch
from .two_opt import _resolve_device


def _to_dev(x, dev, dtype):
    t = x.detach() if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    return t.to(device=dev, dtype=dtype).contiguous()


def _raise_on_status(status: torch.Tensor) -> None:
    st = status.cpu().numpy()
    bad = np.nonzero(st & _lib.ST_BAD_INDEX)[0]
    if len(bad):
        raise ValueError(f"tour {int(bad[0])} has an index outside [0, N)")


def _check_batch(points: torch.Tensor, tours: torch.Tensor, tours_per_instance: int):
    """Shapes the kernels index with: tour k reads instance k // tours_per_instance, so the batch must hold
    exactly I * tours_per_instance tours (the reference raises IndexError at np_points[i] for a missing
    instance, relocate.py:50; here it is a ValueError before anything is launched)."""
    if points.dim() != 3 or points.shape[-1] != 2 or tours.dim() != 2:
        raise ValueError("expected points (I, N, 2) float64 and tours (T, N+1) int64")
    n, T = points.shape[1], tours.shape[0]
    if tours.shape[1] != n + 1 or points.dtype != torch.float64 or tours.dtype != torch.int64:
        raise ValueError("expected points (I, N, 2) float64 and tours (T, N+1) int64")
    if tours_per_instance < 1 or T != points.shape[0] * tours_per_instance:
        raise ValueError(f"{T} tours need {T} / tours_per_instance instances, got points for {points.shape[0]} "
                         f"(tours_per_instance = {tours_per_instance})")
    if points.device.type != "cuda" or tours.device != points.device:
        raise ValueError("points and tours must live on the same HIP device")
    return n, T


def ls_step_device(points: torch.Tensor, tours: torch.Tensor, op: int, *, tours_per_instance: int = 1,
                   fixed_i: Optional[torch.Tensor] = None, guide_k: Optional[torch.Tensor] = None,
                   penalty: Optional[torch.Tensor] = None, inplace: bool = False):
    """One ``get_two_opt_tour`` (op 0) / ``get_relocate_tour`` (op 1) call for every tour, on the device.
    points (I, N, 2) float64, tours (T, N+1) int64.  Returns ``(tours, delta (T,) float64)``."""
    if points.dim() == 2:
        points = points.unsqueeze(0)
    if tours.dim() == 1:
        tours = tours.unsqueeze(0)
    dev = points.device
    n, T = _check_batch(points, tours, int(tours_per_instance))
    points = points.contiguous()
    out = tours.contiguous() if inplace else tours.clone(memory_format=torch.contiguous_format)
    delta = torch.empty(T, dtype=torch.float64, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    fi = None if fixed_i is None else fixed_i.to(device=dev, dtype=torch.int32).contiguous()
    gk = None if guide_k is None else guide_k.to(device=dev, dtype=torch.float64).contiguous()
    pen = None if penalty is None else penalty.to(device=dev, dtype=torch.float64).contiguous()
    if fi is not None and fi.numel() != T:
        raise ValueError("fixed_i must hold one row index per tour (T,)")
    if (gk is None) != (pen is None):
        raise ValueError("guide_k and penalty go together")
    if gk is not None and gk.numel() != T:
        raise ValueError("guide_k must hold one factor per tour (T,)")
    if pen is not None and pen.numel() != T * n * n:
        raise ValueError("penalty must be (T, N, N)")
    rc = _lib.lib().tsp2opt_ls_step(points.data_ptr(), out.data_ptr(), delta.data_ptr(), status.data_ptr(), T, n,
                                    int(tours_per_instance), int(op),
                                    None if fi is None else fi.data_ptr(), None if gk is None else gk.data_ptr(),
                                    None if pen is None else pen.data_ptr(), dev.index,
                                    torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "tsp2opt_ls_step")
    _raise_on_status(status)
    return out, delta


def local_search_device(points: torch.Tensor, tours: torch.Tensor, *, tours_per_instance: int = 1,
                        max_rounds: int = 1 << 30, inplace: bool = False):
    """``local_search`` (search/local_search.py:125-134) for every tour.  Returns ``(tours, rounds int32)``.
    A tour whose search was stopped by ``max_rounds`` (``TSP2OPT_ST_CAPPED``) has ``rounds == max_rounds``:
    callers that pass a cap check that; the default cap is never reached by a terminating search."""
    if points.dim() == 2:
        points = points.unsqueeze(0)
    if tours.dim() == 1:
        tours = tours.unsqueeze(0)
    dev = points.device
    n, T = _check_batch(points, tours, int(tours_per_instance))
    points = points.contiguous()
    out = tours.contiguous() if inplace else tours.clone(memory_format=torch.contiguous_format)
    rounds = torch.empty(T, dtype=torch.int32, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    rc = _lib.lib().tsp2opt_local_search(points.data_ptr(), out.data_ptr(), rounds.data_ptr(), status.data_ptr(), T,
                                         n, int(tours_per_instance), int(min(max_rounds, (1 << 31) - 1)), dev.index,
                                         torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "tsp2opt_local_search")
    _raise_on_status(status)
    return out, rounds


def tsp_relocate(np_points, tours, adj_mat=None, device="cpu", **kwargs):
    """Drop-in for ``search/relocate.py:40-61``: the best relocate move of every tour, applied once.
    Like the reference, ``adj_mat`` only supplies the number of tours (its leading dimension) and tour
    ``i`` uses ``np_points[i]``; returns a list of tours (lists of N+1 ints)."""
    t = np.array(tours)
    if t.ndim == 1:
        t = t[None]
    p = np.asarray(np_points)
    if p.ndim == 2:
        p = p[None]
    count = t.shape[0] if adj_mat is None else (1 if np.ndim(adj_mat) == 2 else np.shape(adj_mat)[0])
    if p.shape[0] < count or t.shape[0] < count:     # the reference: IndexError at np_points[i] / tours[i]
        raise ValueError(f"{count} tours need {count} coordinate sets, got {p.shape[0]} (and {t.shape[0]} tours)")
    dev = _resolve_device(device)
    out, _ = ls_step_device(_to_dev(p[:count], dev, torch.float64), _to_dev(t[:count], dev, torch.int64), 1)
    return [r.tolist() for r in out.cpu().numpy()]


def tsp_ls(np_points, tours, adj_mat=None, device="cpu", **kwargs):
    """Drop-in for ``search/local_search.py:137-154`` (vanilla local search): every tour shares
    ``np_points`` (N, 2); returns a list of tours.  (The reference's loop over several tours rebinds its
    own argument and only works for one; here every row of ``tours`` is searched.)"""
    dev = _resolve_device(device)
    t = np.array(tours)
    if t.ndim == 1:
        t = t[None]
    out, _ = local_search_device(_to_dev(np.asarray(np_points), dev, torch.float64), _to_dev(t, dev, torch.int64),
                                 tours_per_instance=t.shape[0])
    return [r.tolist() for r in out.cpu().numpy()]


def guided_ls_device(points: torch.Tensor, tours: torch.Tensor, heat: torch.Tensor, *, tours_per_instance: int = 1,
                     tours_per_map: int = 1, perturbation_moves: int = 50, time_limit: float = 1.0,
                     max_rounds: Optional[int] = None) -> torch.Tensor:
    """(Regret) guided local search (search/local_search.py:157-232) for a whole batch, on the device.

    points (I, N, 2) float64, tours (T, N+1) int64, heat (T // tours_per_map, N, N): tour k uses instance
    k // tours_per_instance and heat map k // tours_per_map.  Every search keeps its current tour, best tour,
    penalties and guide factor in device memory; one kernel launch runs whole outer rounds for all tours
    (``tsp2opt_gls_rounds``).  ``max_rounds``: that many rounds in ONE launch (deterministic; tests).  Otherwise
    the reference's wall-clock loop: one launch per round until ``time_limit`` seconds have passed (every tour
    gets the whole limit, as in the reference, where the tours are searched one after another).
    Returns the best tours (T, N+1) int64 on the device."""
    if points.dim() == 2:
        points = points.unsqueeze(0)
    if tours.dim() == 1:
        tours = tours.unsqueeze(0)
    if heat.dim() == 2:
        heat = heat.unsqueeze(0)
    n, T = _check_batch(points, tours, int(tours_per_instance))
    dev = points.device
    if tours_per_map < 1 or heat.shape[0] * tours_per_map != T or tuple(heat.shape[1:]) != (n, n):
        raise ValueError("expected heat (T / tours_per_map, N, N)")
    adj = -1 * heat.to(device=dev, dtype=torch.float64)                # local_search.py:167
    regret = torch.where(adj > 0, adj, torch.zeros_like(adj)).contiguous()     # :177
    points = points.contiguous()
    cur = tours.clone(memory_format=torch.contiguous_format)
    best = torch.empty_like(cur)
    best_cost = torch.empty(T, dtype=torch.float64, device=dev)
    guide_k = torch.empty(T, dtype=torch.float64, device=dev)
    penalty = torch.zeros((T, n, n), dtype=torch.float64, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream

    def launch(rounds: int, init: int) -> None:
        rc = L.tsp2opt_gls_rounds(points.data_ptr(), cur.data_ptr(), best.data_ptr(), best_cost.data_ptr(),
                                  guide_k.data_ptr(), penalty.data_ptr(), regret.data_ptr(), status.data_ptr(), T, n,
                                  int(tours_per_instance), int(tours_per_map), int(perturbation_moves), int(rounds),
                                  int(init), dev.index, stream)
        _lib.check(rc, "tsp2opt_gls_rounds")

    if max_rounds is not None:
        launch(int(max_rounds), 1)
    else:
        t_lim = time.time() + time_limit                                # :185 (before the first local_search here)
        launch(0, 1)
        torch.cuda.synchronize(dev)
        while time.time() < t_lim:                                      # :187
            launch(1, 0)
            torch.cuda.synchronize(dev)
    _raise_on_status(status)
    return best


def tsp_gls(np_points, tours, adj_mat, device="cpu", **kwargs):
    """Drop-in for ``search/local_search.py:157-232`` (regret guided local search).  ``kwargs``:
    ``time_limit`` (seconds, per tour) and ``perturbation_moves`` like the reference; additionally
    ``max_rounds`` replaces the wall clock by a fixed number of outer rounds (deterministic; tests).
    All tours (one per heat map of ``adj_mat``) share ``np_points`` and are searched side by side."""
    dev = _resolve_device(device)
    adj = np.asarray(adj_mat)
    t = np.array(tours)
    if adj.ndim == 2:
        adj = adj[None]
        t = t[None] if t.ndim == 1 else t
    count = adj.shape[0]
    best = guided_ls_device(_to_dev(np.asarray(np_points, dtype=np.float64), dev, torch.float64),
                            _to_dev(np.asarray(t[:count], dtype=np.int64), dev, torch.int64),
                            _to_dev(adj, dev, torch.float64), tours_per_instance=count, tours_per_map=1,
                            perturbation_moves=int(kwargs["perturbation_moves"]),
                            time_limit=float(kwargs.get("time_limit", 1.0)), max_rounds=kwargs.get("max_rounds"))
    return [r.tolist() for r in best.cpu().numpy()]


class ML4TSPNARGuidedLS(ML4TSPNARLocalSearch):
    """(Regret) guided local search plugin (ml4tsp/nar/local_search/guided_ls.py:10-89)."""

    def __init__(self, time_limit: float = 1.0, perturbation_moves: int = 50, max_rounds: Optional[int] = None):
        super().__init__(time_limit=time_limit)
        self.perturbation_moves = perturbation_moves
        self.max_rounds = max_rounds

    def _local_search(self, tour: np.ndarray, points: np.ndarray, heatmap: np.ndarray = None) -> np.ndarray:
        self.nodes_num = points.shape[-2]
        dev = _resolve_device(self.device)
        best = guided_ls_device(_to_dev(np.asarray(points, dtype=np.float64), dev, torch.float64),
                                _to_dev(np.asarray(tour, dtype=np.int64), dev, torch.int64),
                                _to_dev(np.asarray(heatmap), dev, torch.float64),
                                perturbation_moves=int(self.perturbation_moves), time_limit=float(self.time_limit),
                                max_rounds=self.max_rounds)
        return best[0].cpu().numpy()

    def local_search(self, tours, points, heatmap=None, per_tours_num: int = 1, per_heatmap_num: int = 1):
        """The base class's per-tour loop (base.py:58-64) as ONE batched device search: tour idx uses
        points[idx // per_tours_num // per_heatmap_num] and heatmap[idx // per_tours_num]."""
        if heatmap is None:
            raise ValueError("guided local search needs the heat map")
        from .local_search import to_numpy
        self.tours, self.points, self.heatmap = to_numpy(tours), to_numpy(points), to_numpy(heatmap)
        self.check_tours_dim()
        self.check_points_dim()
        self.check_heatmap_dim()
        tours, points, heatmap = self.tours, self.points, self.heatmap
        self.nodes_num = points.shape[-2]
        dev = _resolve_device(self.device)
        T = tours.shape[0]
        if (T != points.shape[0] * per_tours_num * per_heatmap_num or T != heatmap.shape[0] * per_tours_num):
            return super().local_search(tours, points, heatmap, per_tours_num, per_heatmap_num)
        best = guided_ls_device(_to_dev(np.asarray(points, dtype=np.float64), dev, torch.float64),
                                _to_dev(np.asarray(tours, dtype=np.int64), dev, torch.int64),
                                _to_dev(np.asarray(heatmap), dev, torch.float64),
                                tours_per_instance=per_tours_num * per_heatmap_num, tours_per_map=per_tours_num,
                                perturbation_moves=int(self.perturbation_moves), time_limit=float(self.time_limit),
                                max_rounds=self.max_rounds)
        out = best.cpu().numpy()
        self._validate(out)
        return out
