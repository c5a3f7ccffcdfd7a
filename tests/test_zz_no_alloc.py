"""GPU test of tsp2opt_solve_noalloc (include/tsp2opt.h): the allocation-free variant of tsp2opt_solve
must give the same tours, iteration / sweep counts, status and lengths, called through the C ABI,
on two calls in a row (the second on a smaller input: the shared scratch keeps the first call's
contents) -- for one workgroup per tour, coupled groups and the cluster kernels."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(rng, inst, tpi, n, dtype):
    pts = rng.random((inst, n, 2)).astype(dtype)
    tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
    for k in range(inst * tpi):
        tours[k, 1:n] = rng.permutation(np.arange(1, n))
    return pts, tours


def _call(L, noalloc, pts, tours, tpi, group, cap, dev):
    from ml4tspbench_amd import _lib
    d_p = torch.from_numpy(pts).to(dev)
    out = torch.from_numpy(tours).to(dev).clone()
    T, n = out.shape[0], pts.shape[1]
    iters = torch.full((T,), -7, dtype=torch.int32, device=dev)
    sweeps = torch.full((T,), -7, dtype=torch.int32, device=dev)
    status = torch.zeros(T, dtype=torch.int32, device=dev)
    lengths = torch.zeros(T, dtype=torch.float64, device=dev)
    code = _lib.DTYPE_F64 if pts.dtype == np.float64 else _lib.DTYPE_F32
    stream = torch.cuda.current_stream(dev).cuda_stream
    common = (d_p.data_ptr(), out.data_ptr(), iters.data_ptr(), sweeps.data_ptr(), lengths.data_ptr(),
              status.data_ptr(), T, n, tpi, group, cap, 0.0, code, dev.index)
    if noalloc:
        rc = L.tsp2opt_solve_noalloc(*common, stream)
    else:
        need = L.tsp2opt_workspace_bytes(T, n, group)
        ws = torch.empty(int(need), dtype=torch.uint8, device=dev)
        rc = L.tsp2opt_solve(*common, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, L.tsp2opt_last_error()
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in (out, iters, sweeps, status, lengths)]


CASES = [
    # n, (instances, tours per instance) of the first and of the second, smaller call, group, dtype, cap
    (100, (64, 4), (5, 4), 1, np.float64, 5000),
    (60, (12, 8), (3, 8), 8, np.float64, 5000),           # coupled groups (targets and flags in the scratch)
    (50, (10, 8), (2, 8), 8, np.float32, 5000),
    (500, (40, 2), (3, 2), 1, np.float32, 5000),
    (1000, (3, 1), (1, 1), 1, np.float64, 300),           # full-window cluster: barrier area in the scratch
    (3000, (2, 1), (1, 1), 1, np.float64, 40),            # band cluster
]


@pytest.mark.parametrize("n,first,second,group,dtype,cap", CASES)
def test_noalloc_matches_solve_twice_in_a_row(n, first, second, group, dtype, cap):
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    assert L.tsp2opt_init(0) == 0, L.tsp2opt_last_error()
    rng = np.random.default_rng(900 + n + group)
    for inst, tpi in (first, second):
        pts, tours = _batch(rng, inst, tpi, n, dtype)
        ref = _call(L, False, pts, tours, tpi, group, cap, dev)
        got = _call(L, True, pts, tours, tpi, group, cap, dev)
        assert (ref[3] & (15 | 64)).max(initial=0) == 0
        for a, b, nm in zip(ref, got, ("tours", "iterations", "sweeps", "status", "lengths")):
            assert np.array_equal(a, b), (nm, inst, tpi)
        assert (got[1] > 0).all()


def test_noalloc_rejects_oversized_batches_and_allocates_nothing():
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    assert L.tsp2opt_init(0) == 0
    cap_bytes = L.tsp2opt_noalloc_max_bytes()
    big = 200_000
    assert L.tsp2opt_workspace_bytes(big, 50, 1) > cap_bytes
    rc = L.tsp2opt_solve_noalloc(ctypes.c_void_p(256), ctypes.c_void_p(256), None, None, None, None, big, 50,
                                 1, 1, 10, 0.0, 1, 0, None)
    assert rc == -3 and b"tsp2opt_solve_noalloc" in L.tsp2opt_last_error()
    # steady state: repeated calls leave the device's free memory where it was
    rng = np.random.default_rng(5)
    pts, tours = _batch(rng, 32, 2, 80, np.float64)
    d_p = torch.from_numpy(pts).to(dev)
    d_t = torch.from_numpy(tours).to(dev)
    iters = torch.empty(64, dtype=torch.int32, device=dev)
    outs = [torch.empty_like(d_t) for _ in range(5)]        # every output exists BEFORE free memory is read:
    stream = torch.cuda.current_stream(dev).cuda_stream     # inside the loop torch allocates nothing either

    def once(out):
        out.copy_(d_t)
        rc = L.tsp2opt_solve_noalloc(d_p.data_ptr(), out.data_ptr(), iters.data_ptr(), None, None, None, 64, 80,
                                     2, 1, 5000, 0.0, 1, 0, stream)
        assert rc == 0
        torch.cuda.synchronize()

    once(outs[0])
    free0 = torch.cuda.mem_get_info(dev)[0]
    reserved0 = torch.cuda.memory_reserved(dev)
    for o in outs[1:]:
        once(o)
    assert torch.cuda.mem_get_info(dev)[0] == free0 and torch.cuda.memory_reserved(dev) == reserved0
    assert all(torch.equal(outs[0], o) for o in outs[1:])


@pytest.mark.parametrize("n,inst,cap", [(100, 48, 5000), (1000, 3, 200)])
def test_noalloc_under_graph_capture_replays_like_solve(n, inst, cap):
    """include/tsp2opt.h names graph capture as a caller of tsp2opt_solve_noalloc: capture one call (n = 100:
    one workgroup per tour; n = 1000: the full-window cluster, whose launch chain event must stay out of the
    capture), replay it twice on fresh inputs and compare with tsp2opt_solve; an uncaptured cluster call
    afterwards must still work."""
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    assert L.tsp2opt_init(0) == 0, L.tsp2opt_last_error()
    rng = np.random.default_rng(77 + n)
    pts, tours = _batch(rng, inst, 1, n, np.float64)
    ref = _call(L, False, pts, tours, 1, 1, cap, dev)
    d_p = torch.from_numpy(pts).to(dev)
    d_in = torch.from_numpy(tours).to(dev)
    out = d_in.clone()
    iters = torch.zeros(inst, dtype=torch.int32, device=dev)
    lengths = torch.zeros(inst, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out.copy_(d_in)
        rc = L.tsp2opt_solve_noalloc(d_p.data_ptr(), out.data_ptr(), iters.data_ptr(), None, lengths.data_ptr(), None,
                                     inst, n, 1, 1, cap, 0.0, 1, 0, torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, L.tsp2opt_last_error()
    for _ in range(2):
        out.zero_()
        iters.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref[0]) and np.array_equal(iters.cpu().numpy(), ref[1])
        assert np.array_equal(lengths.cpu().numpy(), ref[4])
    again = _call(L, True, pts, tours, 1, 1, cap, dev)      # uncaptured call after the capture
    assert np.array_equal(again[0], ref[0])
