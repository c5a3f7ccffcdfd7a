"""Host-side mirror of the ``ml4tsp`` 2-opt local-search plugins.

Reference interface (read-only):
  * ``ml4tsp-library/ml4tsp/nar/local_search/base.py:7-84``   ``ML4TSPNARLocalSearch``
  * ``ml4tsp-library/ml4tsp/nar/local_search/two_opt.py:9-67`` ``ML4TSPNARTwoOpt``
  * ``ml4tsp-library/ml4tsp/nar/local_search/__init__.py:7-19`` ``get_local_search_by_name``
  * ``ml4tsp-library/ml4tsp/ar/local_search/*`` -- the AR twins (same arithmetic, + a tqdm bar)
  * ``search/__init__.py:45-60`` -- the legacy ``LOCAL_SEARCH_CLASS`` registry

Same class names, constructor arguments, attributes, method signatures, output shapes and
``ValueError`` messages.  The difference is inside: ``local_search`` hands ALL tours of the batch to
one persistent HIP kernel launch instead of looping ``_local_search`` per tour in Python
(base.py:58-64), with independent termination per tour exactly like the per-tour calls.

The reference needs ``ml4co_kit`` only for ``to_numpy`` / ``to_tensor`` / ``check_dim``
(converters without arithmetic); small local equivalents keep this module importable without it.
"""
from __future__ import annotations

from typing import Union

import numpy as np
import torch

from .two_opt import evaluate_tours, raise_on_status, tsp_2opt, two_opt_device, _resolve_device


def to_numpy(x):
    """ml4co_kit.to_numpy equivalent: tensor/array/list/None -> ndarray/None."""
    if x is None:
        return None
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def check_dim(array: np.ndarray, dim: int) -> None:
    """ml4co_kit.check_dim equivalent."""
    if array.ndim != dim:
        raise ValueError(f"Dimension mismatch: {array.ndim} != {dim}.")


class ML4TSPNARLocalSearch:
    """Plugin base class (base.py:7-84): dim normalisation, tour->instance mapping, validation."""

    def __init__(self, time_limit: Union[float, str] = "auto"):
        self.time_limit = time_limit
        self.nodes_num = None
        self.device = None
        self.tours = None
        self.heatmap = None
        self.points = None

    def check_tours_dim(self):
        if self.tours.ndim == 1:
            self.tours = np.expand_dims(self.tours, axis=0)
        check_dim(self.tours, 2)

    def check_heatmap_dim(self):
        if self.heatmap.ndim == 2:
            self.heatmap = np.expand_dims(self.heatmap, axis=0)
        check_dim(self.heatmap, 3)

    def check_points_dim(self):
        if self.points.ndim == 2:
            self.points = np.expand_dims(self.points, axis=0)
        check_dim(self.points, 3)

    def is_valid_tour(self, tour: np.ndarray):
        return sorted(tour[:-1]) == [i for i in range(self.nodes_num)]

    def _validate(self, solved_tours: np.ndarray) -> None:
        # base.py:67-72 -- same checks, vectorised
        n = self.nodes_num
        body = np.sort(solved_tours[:, :-1], axis=1)
        ok = (body == np.arange(n)[None, :]).all(axis=1) & (solved_tours[:, -1] == 0)
        if not ok.all():
            tour = solved_tours[int(np.nonzero(~ok)[0][0])]
            raise ValueError(f"The tour {tour} is not valid!")

    def local_search(
        self,
        tours: Union[np.ndarray, torch.Tensor],
        points: Union[np.ndarray, torch.Tensor],
        heatmap: Union[np.ndarray, torch.Tensor] = None,
        per_tours_num: int = 1,
        per_heatmap_num: int = 1,
    ) -> np.ndarray:
        """Generic per-tour loop of the reference (base.py:34-74) for subclasses that only
        implement ``_local_search``."""
        self.tours = to_numpy(tours)
        self.heatmap = to_numpy(heatmap)
        self.points = to_numpy(points)
        self.check_tours_dim()
        self.check_points_dim()
        use_heatmap = heatmap is not None
        if use_heatmap:
            self.check_heatmap_dim()
        solved_tours = list()
        for idx in range(self.tours.shape[0]):
            solved_tours.append(self._local_search(
                self.tours[idx],
                self.points[idx // per_tours_num // per_heatmap_num],
                self.heatmap[idx // per_tours_num] if use_heatmap else None))
        solved_tours = np.array(solved_tours)
        self.nodes_num = self.points.shape[-2]
        self._validate(solved_tours)
        return solved_tours

    def _local_search(self, tour: np.ndarray, heatmap: np.ndarray, points: np.ndarray) -> np.ndarray:
        raise NotImplementedError("_local_search is required to implemented in subclasses.")


class ML4TSPNARTwoOpt(ML4TSPNARLocalSearch):
    """2-opt plugin (two_opt.py:9-67) backed by libtsp2opt_hip.so."""

    def __init__(self, time_limit: Union[float, str] = "auto", max_iterations_2opt: int = 5000):
        super().__init__(time_limit=time_limit)
        self.max_iterations_2opt = max_iterations_2opt
        self.last_result = None          # TwoOptResult of the latest batched call (device tensors)
        self.last_eval = None            # EvalResult (validity, lengths, best tour per instance)

    def _device(self, *tensors) -> torch.device:
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.device.type == "cuda":
                return t.device
        return _resolve_device(self.device)

    def local_search(
        self,
        tours: Union[np.ndarray, torch.Tensor],
        points: Union[np.ndarray, torch.Tensor],
        heatmap: Union[np.ndarray, torch.Tensor] = None,
        per_tours_num: int = 1,
        per_heatmap_num: int = 1,
    ) -> np.ndarray:
        """One launch for all tours.  ``heatmap`` is ignored, like the reference's 2-opt
        (two_opt.py:19).  Tour ``idx`` uses ``points[idx // per_tours_num // per_heatmap_num]``
        (base.py:61).  Arithmetic runs in the dtype of ``points`` (two_opt.py:30: no cast)."""
        dev = self._device(points, tours)
        t = tours.detach() if isinstance(tours, torch.Tensor) else torch.from_numpy(np.asarray(tours))
        p = points.detach() if isinstance(points, torch.Tensor) else torch.from_numpy(np.asarray(points))
        if t.dim() == 1:
            t = t.unsqueeze(0)
        if t.dim() != 2:
            raise ValueError(f"Dimension mismatch: {t.dim()} != 2.")
        if p.dim() == 2:
            p = p.unsqueeze(0)
        if p.dim() != 3:
            raise ValueError(f"Dimension mismatch: {p.dim()} != 3.")
        if not p.dtype.is_floating_point:
            raise TypeError("points must be floating point")
        self.nodes_num = p.shape[-2]
        res = two_opt_device(p.to(dev), t.to(dev), tours_per_instance=per_tours_num * per_heatmap_num,
                             group_size=1, max_iterations=self.max_iterations_2opt,
                             time_limit=self.time_limit, want_lengths=True)
        self.last_result = res
        # validity (base.py:67-72), lengths and the best tour per instance, still on the device
        ev = evaluate_tours(p.to(dev), res.tours, tours_per_instance=per_tours_num * per_heatmap_num) \
            if t.shape[0] == p.shape[0] * per_tours_num * per_heatmap_num else None
        self.last_eval = ev
        solved_tours = res.tours.cpu().numpy()
        raise_on_status(res.status.cpu().numpy())
        # kept like the reference: numpy copies of the inputs as attributes
        self.tours, self.points, self.heatmap = to_numpy(t), to_numpy(p), to_numpy(heatmap)
        if ev is None:
            self._validate(solved_tours)
        else:
            ok = (ev.valid.cpu().numpy() & 3) == 3          # permutation and ends at node 0
            if not ok.all():
                tour = solved_tours[int(np.nonzero(~ok)[0][0])]
                raise ValueError(f"The tour {tour} is not valid!")
        return solved_tours

    def _local_search(self, tour: np.ndarray, points: np.ndarray, heatmap: np.ndarray = None) -> np.ndarray:
        """Single tour (two_opt.py:18-67): (N+1,) in, (N+1,) out."""
        pts = points if isinstance(points, torch.Tensor) else torch.from_numpy(np.asarray(points))
        self.nodes_num = pts.shape[-2]
        dev = self._device(pts, tour)
        t = tour if isinstance(tour, torch.Tensor) else torch.from_numpy(np.asarray(tour))
        res = two_opt_device(pts.to(dev), t.to(dev).reshape(1, -1), group_size=1,
                             max_iterations=self.max_iterations_2opt, time_limit=self.time_limit,
                             want_lengths=False)
        raise_on_status(res.status.cpu().numpy())
        return res.tours[0].cpu().numpy()


class ML4TSPARLocalSearch(ML4TSPNARLocalSearch):
    """AR twin of the base class (ar/local_search/base.py:8-85); identical behaviour."""


class ML4TSPARTwoOpt(ML4TSPNARTwoOpt, ML4TSPARLocalSearch):
    """AR twin (ar/local_search/two_opt.py:9-67): same kernel, same semantics."""


def get_local_search_by_name(name: str):
    """nar/local_search/__init__.py:7-19; only the 2-opt plugin is in scope here."""
    local_search_dict = {
        "2opt": ML4TSPNARTwoOpt,
    }
    supported_local_search_type = local_search_dict.keys()
    if name not in supported_local_search_type:
        message = (
            f"Cannot find the local_search whose name is {name},"
            f"ML4TSPNARLocalSearch only supports {supported_local_search_type}."
        )
        raise ValueError(message)
    return local_search_dict[name]


# legacy registry (search/__init__.py:45-60); only the entry this package implements
LOCAL_SEARCH_CLASS = {
    ("tsp", "2opt"): tsp_2opt,
    ("tsp", None): None,
}


def get_local_search_func(task="tsp", name="2opt"):
    return LOCAL_SEARCH_CLASS[(task, name)]
