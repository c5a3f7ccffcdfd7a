"""pytest configuration: registers the `gpu` marker and common fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks (runs on CPU).
`-m gpu`      : parity tests proper, through the C ABI on a real MI355X.
"""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden_files():
    """2-opt fixtures g<k>_*.npz (oracle/gen_golden.py); the greedy decoder's are greedy_*.npz
    (tests/test_greedy.py), the evaluator's eval_*.npz."""
    return sorted(f for f in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]*.npz")))


def eval_golden_files():
    """Evaluation fixtures (oracle/gen_golden_eval.py): costs from the reference's TSPEvaluator."""
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "eval_*.npz")))


def load_golden(path):
    z = np.load(path, allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["name"] = os.path.splitext(os.path.basename(path))[0]
    d["group_size"] = int(d["group_size"])
    d["tours_per_instance"] = int(d["tours_per_instance"])
    d["max_iterations"] = int(d["max_iterations"])
    d["entry"] = str(d["entry"])
    return d


@pytest.fixture(scope="session")
def oracle():
    from oracle import two_opt_oracle
    two_opt_oracle.build()
    return two_opt_oracle
