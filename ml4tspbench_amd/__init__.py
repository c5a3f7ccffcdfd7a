"""ml4tspbench_amd -- MI355X-native batched 2-opt local search, drop-in for ML4TSPBench's
``search/two_opt.py`` and the ``ml4tsp`` NAR/AR 2-opt local-search plugins.

Only the hot path lives here: the HIP kernels + C ABI (``csrc/``, ``libtsp2opt_hip.so``) and the
host-side mirror of the reference interfaces.
"""
from .two_opt import (  # noqa: F401
    batched_two_opt_torch, tsp_2opt, two_opt_device, TwoOptResult, evaluate_tours, EvalResult,
)
from .greedy import (  # noqa: F401
    tsp_greedy, greedy_decode_device, greedy_decode_sparse_device, get_decoding_func, DECODING_CLASS,
)
from .local_search import (  # noqa: F401
    ML4TSPNARLocalSearch, ML4TSPNARTwoOpt, ML4TSPARLocalSearch, ML4TSPARTwoOpt,
    get_local_search_by_name, get_local_search_func, LOCAL_SEARCH_CLASS,
)
from .local_ops import (  # noqa: F401,E402
    tsp_relocate, tsp_ls, tsp_gls, ML4TSPNARGuidedLS, ls_step_device, local_search_device, guided_ls_device,
)
from .anytime import two_opt_anytime, AnytimeResult  # noqa: F401,E402

# the legacy registry's other local searches (search/__init__.py:45-52)
LOCAL_SEARCH_CLASS[("tsp", "relocate")] = tsp_relocate
LOCAL_SEARCH_CLASS[("tsp", "ls")] = tsp_ls
LOCAL_SEARCH_CLASS[("tsp", "gls")] = tsp_gls

__all__ = [
    "two_opt_anytime", "AnytimeResult",
    "tsp_relocate", "tsp_ls", "tsp_gls", "ML4TSPNARGuidedLS", "ls_step_device", "local_search_device", "guided_ls_device",
    "batched_two_opt_torch", "tsp_2opt", "two_opt_device", "TwoOptResult", "evaluate_tours", "EvalResult",
    "ML4TSPNARLocalSearch", "ML4TSPNARTwoOpt", "ML4TSPARLocalSearch", "ML4TSPARTwoOpt",
    "get_local_search_by_name", "get_local_search_func", "LOCAL_SEARCH_CLASS",
    "tsp_greedy", "greedy_decode_device", "greedy_decode_sparse_device", "get_decoding_func", "DECODING_CLASS",
]
