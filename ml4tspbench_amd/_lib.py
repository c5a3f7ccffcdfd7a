"""ctypes binding of libtsp2opt_hip.so (include/tsp2opt.h).

Follows the reference's own native-extension convention (search/c_mcts/__init__.py:11-29:
ctypes.CDLL + argtypes/restype).  There is NO fallback: if the library or a GPU is missing the
call fails loudly (the CPU oracle under oracle/ is test infrastructure and is never used here).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSP2OPT_LIB: load another build of the same library (the -DTSP2OPT_FINE diagnostic variant)
LIB_PATH = os.environ.get("TSP2OPT_LIB") or os.path.join(_HERE, "libtsp2opt_hip.so")

DTYPE_F32 = 0
DTYPE_F64 = 1

ST_BAD_INDEX = 1
ST_NOT_CLOSED = 2
ST_NONFINITE = 4
ST_RANGE = 8
ST_CAPPED = 16
ST_TIME = 32
ST_INTERNAL = 64

MAX_N = 65535

_lib = None

# every symbol include/tsp2opt.h declares (tests/test_abi.py checks the list against the header)
SYMBOLS = (
    "tsp2opt_solve",
    "tsp2opt_workspace_bytes",
    "tsp2opt_solve_noalloc",
    "tsp2opt_init",
    "tsp2opt_noalloc_max_bytes",
    "tsp2opt_tour_lengths",
    "tsp2opt_evaluate",
    "tsp2opt_greedy_decode",
    "tsp2opt_greedy_workspace_bytes",
    "tsp2opt_greedy_decode_sparse",
    "tsp2opt_greedy_sparse_workspace_bytes",
    "tsp2opt_ls_step",
    "tsp2opt_local_search",
    "tsp2opt_gls_rounds",
    "tsp2opt_sqrt_check",
    "tsp2opt_plan",
    "tsp2opt_debug_greedy_cycles",
    "tsp2opt_profile_enable",
    "tsp2opt_profile_last_ms",
    "tsp2opt_profile_history_ms",
    "tsp2opt_debug_phase_cycles",
    "tsp2opt_debug_sweep_only",
    "tsp2opt_debug_cluster_mode",
    "tsp2opt_debug_cluster_inc",
    "tsp2opt_debug_tune",
    "tsp2opt_cluster_plan",
    "tsp2opt_last_error",
    "tsp2opt_version",
)


class Tsp2optError(RuntimeError):
    """A negative return code of the library; ``code`` is that value (TSP2OPT_E_*), None when the error was
    raised on the Python side."""

    def __init__(self, message: str, code=None):
        super().__init__(message)
        self.code = code


E_ARG, E_UNSUPPORTED, E_WORKSPACE, E_HIP = -1, -2, -3, -4


def lib() -> ctypes.CDLL:
    """Load the HIP library; raises if it has not been built (python -m ml4tspbench_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Tsp2optError(
            f"{LIB_PATH} is missing: build it with `python -m ml4tspbench_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    L.tsp2opt_solve.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f64, i32, i32,
                                vp, i64, vp]
    L.tsp2opt_solve.restype = ctypes.c_int
    L.tsp2opt_solve_noalloc.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f64, i32, i32, vp]
    L.tsp2opt_solve_noalloc.restype = ctypes.c_int
    L.tsp2opt_init.argtypes = [i32]
    L.tsp2opt_init.restype = ctypes.c_int
    L.tsp2opt_noalloc_max_bytes.argtypes = []
    L.tsp2opt_noalloc_max_bytes.restype = i64
    L.tsp2opt_workspace_bytes.argtypes = [i64, i32, i32]
    L.tsp2opt_workspace_bytes.restype = i64
    L.tsp2opt_tour_lengths.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp]
    L.tsp2opt_tour_lengths.restype = ctypes.c_int
    L.tsp2opt_evaluate.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    L.tsp2opt_evaluate.restype = ctypes.c_int
    L.tsp2opt_greedy_decode.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp]
    L.tsp2opt_greedy_decode.restype = ctypes.c_int
    L.tsp2opt_greedy_workspace_bytes.argtypes = [i64, i32, i32]
    L.tsp2opt_greedy_workspace_bytes.restype = i64
    L.tsp2opt_ls_step.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, i32, vp]
    L.tsp2opt_ls_step.restype = ctypes.c_int
    L.tsp2opt_local_search.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    L.tsp2opt_local_search.restype = ctypes.c_int
    L.tsp2opt_gls_rounds.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]
    L.tsp2opt_gls_rounds.restype = ctypes.c_int
    L.tsp2opt_greedy_decode_sparse.argtypes = [vp, vp, vp, vp, i64, i32, i64, i32, i32, i32, vp, i64, vp]
    L.tsp2opt_greedy_decode_sparse.restype = ctypes.c_int
    L.tsp2opt_greedy_sparse_workspace_bytes.argtypes = [i64, i32, i64, i32, i32]
    L.tsp2opt_greedy_sparse_workspace_bytes.restype = i64
    L.tsp2opt_sqrt_check.argtypes = [vp, vp, i64, i32, i32, vp]
    L.tsp2opt_sqrt_check.restype = ctypes.c_int
    L.tsp2opt_plan.argtypes = [i32, i32, i32, ctypes.POINTER(i32)]
    L.tsp2opt_plan.restype = ctypes.c_int
    L.tsp2opt_profile_enable.argtypes = [i32]
    L.tsp2opt_profile_enable.restype = ctypes.c_int
    L.tsp2opt_profile_last_ms.argtypes = [i32, ctypes.POINTER(f64), ctypes.POINTER(f64)]
    L.tsp2opt_profile_last_ms.restype = ctypes.c_int
    L.tsp2opt_profile_history_ms.argtypes = [i32, i32, ctypes.POINTER(f64), ctypes.POINTER(f64)]
    L.tsp2opt_profile_history_ms.restype = ctypes.c_int
    L.tsp2opt_debug_phase_cycles.argtypes = [i32]
    L.tsp2opt_debug_phase_cycles.restype = ctypes.c_int
    L.tsp2opt_debug_sweep_only.argtypes = [i32]
    L.tsp2opt_debug_sweep_only.restype = ctypes.c_int
    L.tsp2opt_debug_cluster_mode.argtypes = [i32]
    L.tsp2opt_debug_cluster_mode.restype = ctypes.c_int
    L.tsp2opt_debug_cluster_inc.argtypes = [i32]
    L.tsp2opt_debug_cluster_inc.restype = ctypes.c_int
    L.tsp2opt_debug_tune.argtypes = [i32, i32]
    L.tsp2opt_debug_tune.restype = ctypes.c_int
    L.tsp2opt_cluster_plan.argtypes = [i32, i32, i64, i32, vp]
    L.tsp2opt_cluster_plan.restype = ctypes.c_int
    L.tsp2opt_last_error.argtypes = []
    L.tsp2opt_last_error.restype = ctypes.c_char_p
    L.tsp2opt_version.argtypes = []
    L.tsp2opt_version.restype = ctypes.c_int
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc < 0:
        msg = lib().tsp2opt_last_error()
        raise Tsp2optError(f"{what} failed ({rc}): {msg.decode() if msg else ''}", code=int(rc))


def plan(n: int, dtype: int = DTYPE_F64, device: int = -1) -> dict:
    out = (ctypes.c_int32 * 7)()
    check(lib().tsp2opt_plan(n, dtype, device, out), "tsp2opt_plan")
    keys = ("threads", "filter_class", "lds_bytes", "row_segments", "diag_groups", "macro_steps",
            "workgroups_per_cu")
    return dict(zip(keys, list(out)))


def cluster_plan(n: int, dtype: int, num_tours: int, cus: int = 256) -> dict:
    """Geometry tsp2opt_solve would use when several workgroups share a tour (0 workgroups per tour:
    one workgroup per tour, see plan()); row_bands == 0 marks the full-window cluster kernel."""
    out = (ctypes.c_int32 * 12)()
    check(lib().tsp2opt_cluster_plan(n, dtype, num_tours, cus, ctypes.cast(out, ctypes.c_void_p)),
          "tsp2opt_cluster_plan")
    keys = ("workgroups_per_tour", "row_bands", "diag_bands", "threads", "lane_groups", "row_segments",
            "macro_steps", "lds_bytes", "incremental", "steps_per_cache_entry", "cache_entries_per_lane",
            "exact_coords_in_lds")
    return dict(zip(keys, list(out)))
