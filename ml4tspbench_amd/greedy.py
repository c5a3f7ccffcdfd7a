"""Host-side mirror of the reference's greedy edge-merge decoder.

Reference interface (read-only): ``search/greedy_search.py:19-74`` ``tsp_greedy`` (registered as
``DECODING_CLASS[('tsp', 'greedy')]``, ``search/__init__.py:36``) on top of the Cython
``merge_cython`` (``search/cython_merge/sources/cython_merge.pyx:19-120``).  Same name, arguments
and return value (a list of ``parallel_sampling`` tours, each a list of N+1 ints starting and ending
at node 0); the work runs in ``csrc/greedy.hip`` (dense heat maps: merge rounds, no N^2 sort) and
``csrc/greedy_sparse.hip`` (``sparse_graph=True``: the ``(values, edge_index)`` lists of
``greedy_search.py:39-47`` are decoded as lists instead of being densified).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .two_opt import _resolve_device


def greedy_decode_device(heat: torch.Tensor, points: torch.Tensor, *, maps_per_instance: int = 1) -> torch.Tensor:
    """heat (B, N, N) float32/float64 dense, NOT symmetrised; points (I, N, 2); both on one HIP
    device, B == I * maps_per_instance.  Returns tours (B, N+1) int64 on the device (input of
    ``two_opt_device``).  Asynchronous on the current stream."""
    if heat.device.type != "cuda" or points.device != heat.device:
        raise ValueError("heat and points must live on the same HIP device")
    if points.dim() == 2:
        points = points.unsqueeze(0)
    if heat.dim() == 2:
        heat = heat.unsqueeze(0)
    B, n, n2 = heat.shape
    if n != n2 or points.shape[1] != n or B != points.shape[0] * maps_per_instance:
        raise ValueError("expected heat (I * maps_per_instance, N, N) and points (I, N, 2)")
    if heat.dtype not in (torch.float32, torch.float64):
        heat = heat.to(torch.float32)
    heat = heat.contiguous()
    pts = points.to(torch.float64).contiguous()        # greedy_search.py:15 points.astype("double")
    dev = heat.device
    code = _lib.DTYPE_F64 if heat.dtype == torch.float64 else _lib.DTYPE_F32
    L = _lib.lib()
    need = L.tsp2opt_greedy_workspace_bytes(B, n, code)
    _lib.check(need, "tsp2opt_greedy_workspace_bytes")
    ws = torch.empty(int(need), dtype=torch.uint8, device=dev)     # A + A^T; too big to keep cached
    tours = torch.empty((B, n + 1), dtype=torch.int64, device=dev)
    rc = L.tsp2opt_greedy_decode(heat.data_ptr(), pts.data_ptr(), tours.data_ptr(), B, n,
                                 int(maps_per_instance), code, dev.index, ws.data_ptr(), ws.numel(),
                                 torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "tsp2opt_greedy_decode")
    ws.record_stream(torch.cuda.current_stream(dev))
    return tours


def greedy_decode_sparse_device(values: torch.Tensor, edge_index: torch.Tensor, points: torch.Tensor, *,
                                maps_per_instance: int = 1, return_diagnostics: bool = False,
                                workspace_limit_bytes: int = 1 << 30):
    """Sparse heat maps without densifying them: values (B, E) float32/float64, edge_index (I, 2, E) integer
    (one list per instance, shared by its maps_per_instance heat maps; (2, E) for a single instance),
    points (I, N, 2), all on one HIP device, B == I * maps_per_instance.
    Returns tours (B, N+1) int64 on the device.  Raises ``Tsp2optError`` (code -2) for inputs the sparse
    kernels do not take (negative heat sums, a node with more than 7000 listed edges): densify then.
    The batch is decoded in as few library calls as keep each call's workspace (24 B per listed edge at one
    heat map per instance) under ``workspace_limit_bytes``; a single instance is never split."""
    if values.device.type != "cuda" or points.device != values.device:
        raise ValueError("values and points must live on the same HIP device")
    if points.dim() == 2:
        points = points.unsqueeze(0)
    if values.dim() == 1:
        values = values.unsqueeze(0)
    if edge_index.dim() == 2:
        edge_index = edge_index.unsqueeze(0)
    B, E = values.shape
    I, n = points.shape[0], points.shape[1]
    mpi = int(maps_per_instance)
    if tuple(edge_index.shape) != (I, 2, E) or B != I * mpi:
        raise ValueError("expected values (I * maps_per_instance, E), edge_index (I, 2, E) and points (I, N, 2)")
    if values.dtype not in (torch.float32, torch.float64):
        values = values.to(torch.float32)
    values = values.contiguous()
    ei = edge_index.to(device=values.device, dtype=torch.int64).contiguous()
    pts = points.to(torch.float64).contiguous()
    dev = values.device
    code = _lib.DTYPE_F64 if values.dtype == torch.float64 else _lib.DTYPE_F32
    L = _lib.lib()
    one = L.tsp2opt_greedy_sparse_workspace_bytes(mpi, n, E, mpi, code)
    _lib.check(one, "tsp2opt_greedy_sparse_workspace_bytes")
    per_call = max(1, min(I, int(workspace_limit_bytes) // max(int(one), 1), 65535 // mpi))
    tours = torch.empty((B, n + 1), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = None
    for q0 in range(0, I, per_call):
        q1 = min(I, q0 + per_call)
        need = L.tsp2opt_greedy_sparse_workspace_bytes((q1 - q0) * mpi, n, E, mpi, code)
        _lib.check(need, "tsp2opt_greedy_sparse_workspace_bytes")
        if ws is None or ws.numel() < need:
            ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        rc = L.tsp2opt_greedy_decode_sparse(values[q0 * mpi:q1 * mpi].data_ptr(), ei[q0:q1].data_ptr(),
                                            pts[q0:q1].data_ptr(), tours[q0 * mpi:q1 * mpi].data_ptr(),
                                            (q1 - q0) * mpi, n, E, mpi, code, dev.index, ws.data_ptr(), ws.numel(),
                                            stream)
        _lib.check(rc, "tsp2opt_greedy_decode_sparse")
    if return_diagnostics:                                  # of the last call's first heat map
        d = ws[64:128].cpu().numpy().view("uint64")
        names = ("setup_ticks", "listed_ticks", "unlisted_ticks", "walk_ticks", "rounds", "serial_rounds", "reserved",
                 "listed_merges")
        return tours, {k: int(v) for k, v in zip(names, d)}
    return tours


def tsp_greedy(adj_mat, np_points, edge_index_np=None, sparse_graph=False, parallel_sampling=1,
               device="cpu", **kwargs):
    """Drop-in for ``search/greedy_search.py:19-74``.

    ``adj_mat``: dense ``(parallel_sampling, N, N)`` heat maps, or with ``sparse_graph=True`` the
    flat ``(parallel_sampling * E,)`` edge values for ``edge_index_np (2, E)``, densified exactly
    like the reference (``coo_matrix(...).toarray()``, duplicates summed, :41-47).  Returns a list
    of ``parallel_sampling`` tours (lists of N+1 ints).
    """
    dev = _resolve_device(device)
    pts = torch.as_tensor(np.asarray(np_points)) if not isinstance(np_points, torch.Tensor) else np_points
    n = pts.shape[0]
    a = adj_mat if isinstance(adj_mat, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(adj_mat))
    a = a.to(dev)
    if not sparse_graph:
        heat = a.reshape(parallel_sampling, -1, n, n)[:, 0]         # adj_mat[0] of every split (:36-38)
    else:
        if edge_index_np is None:
            raise AssertionError("edge_index_np must be given when graph is sparse")
        ei = torch.as_tensor(np.asarray(edge_index_np)).to(dev).long()
        vals = a.reshape(parallel_sampling, -1)
        try:        # the listed edges stay a list (no N x N arrays); inputs it does not take are densified below
            tours = greedy_decode_sparse_device(vals, ei, pts.to(dev)[None], maps_per_instance=parallel_sampling)
            return [t.tolist() for t in tours.cpu().numpy()]
        except _lib.Tsp2optError as exc:
            if exc.code != _lib.E_UNSUPPORTED:
                raise
        heat = torch.zeros((parallel_sampling, n, n), dtype=vals.dtype, device=dev)
        heat.index_put_((torch.arange(parallel_sampling, device=dev)[:, None].expand(-1, ei.shape[1]),
                         ei[0][None].expand(parallel_sampling, -1), ei[1][None].expand(parallel_sampling, -1)),
                        vals, accumulate=True)
    tours = greedy_decode_device(heat, pts.to(dev)[None], maps_per_instance=parallel_sampling)
    return [t.tolist() for t in tours.cpu().numpy()]


# legacy registry entry (search/__init__.py:35-42); only the decoder this package implements
DECODING_CLASS = {("tsp", "greedy"): tsp_greedy}


def get_decoding_func(task="tsp", name="greedy"):
    return DECODING_CLASS[(task, name)]
