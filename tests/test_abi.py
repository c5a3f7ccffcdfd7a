"""CPU: the C-ABI library loads and exports every symbol include/tsp2opt.h declares; argument
validation that needs no GPU behaves as documented.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "tsp2opt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsp2opt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    fns = _header_functions()
    for must in ("tsp2opt_solve", "tsp2opt_workspace_bytes", "tsp2opt_tour_lengths",
                 "tsp2opt_last_error", "tsp2opt_version"):
        assert must in fns


def test_library_exports_every_declared_symbol():
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    fns = _header_functions()
    for name in fns:
        assert hasattr(L, name), f"{name} declared in include/tsp2opt.h but not exported"
    assert sorted(_lib.SYMBOLS) == fns, "ml4tspbench_amd/_lib.py SYMBOLS out of sync with the header"
    assert L.tsp2opt_version() == 100


def test_argument_validation_without_gpu():
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    assert L.tsp2opt_workspace_bytes(2048, 500, 1) > 2048 * 12
    assert L.tsp2opt_workspace_bytes(-1, 500, 1) < 0
    # bad arguments are rejected before any HIP call
    rc = L.tsp2opt_solve(None, None, None, None, None, None, 4, 0, 1, 1, 10, 0.0, 1, 0, None, 0, None)
    assert rc == -1 and b"n < 1" in L.tsp2opt_last_error()
    rc = L.tsp2opt_solve(None, None, None, None, None, None, 4, 10, 1, 1, 10, 0.0, 1, 0, None, 0, None)
    assert rc == -1
    rc = L.tsp2opt_solve(ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None, None, 4, 70000, 1, 1,
                         10, 0.0, 1, 0, None, 0, None)
    assert rc == -2
    rc = L.tsp2opt_solve(ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None, None, 6, 10, 4, 3,
                         10, 0.0, 1, 0, None, 0, None)
    assert rc == -1 and b"group_size" in L.tsp2opt_last_error()
    rc = L.tsp2opt_solve(ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None, None, 4, 10, 1, 1,
                         10, 0.0, 7, 0, None, 0, None)
    assert rc == -1 and b"dtype" in L.tsp2opt_last_error()
    # the coupled call has no wall-clock limit in the reference: rejected, not silently ignored
    rc = L.tsp2opt_solve(ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None, None, 8, 10, 4, 4,
                         10, 0.5, 1, 0, None, 0, None)
    assert rc == -1 and b"time_limit_s" in L.tsp2opt_last_error()
    # misaligned arrays / workspace would fault inside the kernels: rejected up front
    rc = L.tsp2opt_solve(ctypes.c_void_p(8), ctypes.c_void_p(12), None, None, None, None, 4, 10, 1, 1,
                         10, 0.0, 1, 0, None, 0, None)
    assert rc == -1 and b"misaligned" in L.tsp2opt_last_error()
    rc = L.tsp2opt_solve(ctypes.c_void_p(8), ctypes.c_void_p(8), None, None, None, None, 4, 10, 1, 1,
                         10, 0.0, 1, 0, ctypes.c_void_p(64), 1 << 20, None)
    assert rc == -1 and b"256-byte" in L.tsp2opt_last_error()


def test_plan_is_consistent():
    from ml4tspbench_amd import _lib
    for n in (5, 9, 12, 18, 50, 100, 500, 1000, 2000, 10000, 65535):
        for dt in (_lib.DTYPE_F32, _lib.DTYPE_F64):
            pl = _lib.plan(n, dt, -1)
            assert pl["threads"] % 64 == 0 and 64 <= pl["threads"] <= 1024
            assert pl["lds_bytes"] <= 160 * 1024
            if pl["filter_class"]:
                G, NS, MS = pl["diag_groups"], pl["row_segments"], pl["macro_steps"]
                assert 4 * G + 1 <= n - 2            # every walked diagonal is a valid pair distance
                assert 4 * G + 1 >= n // 2           # ... and all of them are covered
                assert 4 * NS * MS >= n              # all rows are covered
                assert 4 * NS * MS + 4 * G + 8 <= 8 * pl["filter_class"]   # window fits the LDS arrays
    with pytest.raises(_lib.Tsp2optError):
        _lib.plan(0)


def test_filter_class_256_runs_512_threads():
    """k_two_opt<.., 256, ..> is compiled for at most 512 threads (kMaxThreads): no plan may ask for more."""
    from ml4tspbench_amd import _lib
    seen = set()
    for n in range(600, 1500):
        for dt in (_lib.DTYPE_F32, _lib.DTYPE_F64):
            pl = _lib.plan(n, dt, -1)
            seen.add(pl["filter_class"])
            if pl["filter_class"] == 256:
                assert pl["threads"] <= 512, (n, pl)
    assert {128, 256, 512} <= seen


def test_decoder_argument_validation_without_gpu():
    """Host-side checks of the decoder entry points: sizes, dtypes, limits and null pointers are rejected
    before any HIP call; the workspace queries are pure host arithmetic."""
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    F32, F64 = _lib.DTYPE_F32, _lib.DTYPE_F64
    # dense: a fp32 key matrix + lists + neighbour table per heat map
    w = L.tsp2opt_greedy_workspace_bytes(4, 500, F32)
    assert 4 * 500 * 500 * 4 <= w < 4 * 500 * 500 * 4 + 4 * 500 * 64 + 4096
    assert L.tsp2opt_greedy_workspace_bytes(4, 1, F32) < 0 and L.tsp2opt_greedy_workspace_bytes(4, 50, 7) < 0
    assert L.tsp2opt_greedy_decode(None, None, None, 1, 50, 1, F32, 0, None, 0, None) == -1
    assert L.tsp2opt_greedy_decode(None, None, None, 1, 20000, 1, F32, 0, None, 0, None) == -2
    assert L.tsp2opt_greedy_decode(None, None, None, 0, 50, 1, F32, 0, None, 0, None) == 0      # nothing to do
    # sparse: 12 B per listed edge and instance + 12 B per listed edge and heat map (+ per-node arrays)
    n, E = 10000, 500000
    one = L.tsp2opt_greedy_sparse_workspace_bytes(1, n, E, 1, F32)
    assert 24 * E <= one < 24 * E + 64 * n + 8192
    sixteen_maps = L.tsp2opt_greedy_sparse_workspace_bytes(16, n, E, 16, F32)     # one instance, 16 heat maps
    assert 12 * E + 16 * 12 * E <= sixteen_maps < 12 * E + 16 * 12 * E + 16 * 64 * n + 8192
    assert L.tsp2opt_greedy_sparse_workspace_bytes(16, n, E, 1, F64) >= 16 * 24 * E              # 16 instances
    assert L.tsp2opt_greedy_sparse_workspace_bytes(1, n, E, 0, F32) < 0
    assert L.tsp2opt_greedy_sparse_workspace_bytes(1, 1, E, 1, F32) < 0
    bad = L.tsp2opt_greedy_decode_sparse
    assert bad(None, None, None, None, 1, 50, 10, 1, F32, 0, None, 0, None) == -1               # null pointers
    assert bad(None, None, None, None, 1, 20000, 10, 1, F32, 0, None, 0, None) == -2            # n > 10900
    assert bad(None, None, None, None, 70000, 50, 10, 1, F32, 0, None, 0, None) == -2           # > 65535 heat maps
    assert bad(None, None, None, None, 1, 50, 10, 0, F32, 0, None, 0, None) == -1               # maps_per_instance < 1
    assert bad(None, None, None, None, 1, 50, 10, 1, 9, 0, None, 0, None) == -1                 # dtype
    assert bad(None, None, None, None, 0, 50, 10, 1, F32, 0, None, 0, None) == 0
    assert b"" != L.tsp2opt_last_error()
