"""Instance / tour text format of the reference datasets.

One instance per line:  ``x0 y0 x1 y1 ... output t0 t1 ... t0`` with 1-based node indices
(writer: ``generate.py:147-153``; reader/writer: ``solvers/solver.py:51-78,115-138``).  Tours come
back 0-based and closed, ready for ``two_opt_device`` / ``tsp_2opt``.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def read_txt(filename: str, load_tours: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
    """Returns ``(points (I, N, 2) float64, tours (I, N+1) int64 0-based or None)``.

    Like ``TSPSolver.from_txt`` (solvers/solver.py:51-78): a line without ``output`` switches tour
    loading off for the whole file.  (The reference additionally keeps a float32 copy of the points,
    solver.py:77; callers that want that route cast themselves.)
    """
    if not filename.endswith(".txt"):
        raise AssertionError("file name error")
    coords, tours = [], []
    with open(filename, "r") as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if "output" in line:
                pts_s, tour_s = line.split(" output ")
                tours.append(np.array([int(t) for t in tour_s.split(" ") if t], dtype=np.int64) - 1)
            else:
                pts_s = line
                load_tours = False
            v = [float(x) for x in pts_s.split(" ") if x]
            coords.append(np.array(v, dtype=np.float64).reshape(-1, 2))
    points = np.array(coords)
    return points, (np.array(tours) if load_tours and tours else None)


def write_txt(filename: str, points: np.ndarray, tours: np.ndarray) -> None:
    """``TSPSolver.to_txt`` (solvers/solver.py:115-138): tours are written as given (pass closed
    tours ``t0 ... t0`` to get the dataset layout of generate.py:147-153), 1-based."""
    points = np.asarray(points)
    tours = np.asarray(tours)
    with open(filename, "w") as f:
        for idx, tour in enumerate(tours):
            f.write(" ".join(str(x) + " " + str(y) for x, y in points[idx]))
            f.write(" output ")
            f.write(" ".join(str(int(node) + 1) for node in tour))
            f.write("\n")
