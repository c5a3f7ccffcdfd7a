"""The reference's other move operators: relocate, its two-opt variant, their
local_search loop and guided local search (search/relocate.py, search/local_search.py).
CPU: the oracle (oracle/ls_oracle.c/.py) against goldens made by the reference itself
(oracle/gen_golden_ls.py).  GPU: the HIP operators (csrc/ls_ops.hip) against both."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR


def _load(pattern):
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, pattern)))
    assert files, pattern
    return [(os.path.basename(f)[:-4], np.load(f)) for f in files]


@pytest.fixture(scope="module")
def lso():
    from oracle import ls_oracle
    ls_oracle.build()
    return ls_oracle


def test_oracle_relocate_matches_reference(lso):
    for name, z in _load("ls_relocate_*.npz"):
        out, delta = lso.step(z["points"], z["tours_in"], 1)
        assert np.array_equal(out, z["tours_out"]), name
        assert (delta < 0).all()


def test_oracle_single_operator_calls_match_reference(lso):
    for name, z in _load("ls_step_*.npz"):
        for op in (0, 1):
            for g in (0, 1):
                for f in (0, 1):
                    out, delta = lso.step(z["points"], z["tours_in"], op,
                                          fixed_i=z["fixed_i"] if f else None,
                                          guide_k=z["guide_k"] if g else None,
                                          penalty=z["penalty"] if g else None)
                    key = f"op{op}_g{g}_f{f}"
                    assert np.array_equal(out, z["out_" + key]), (name, key)
                    assert np.array_equal(delta, z["delta_" + key]), (name, key)     # bit-exact doubles


def test_oracle_local_search_matches_reference(lso):
    for name, z in _load("ls_loop_*.npz"):
        out, rounds = lso.local_search(z["points"], z["tours_in"])
        assert np.array_equal(out, z["tours_out"]), name
        assert (rounds >= 1).all()
        n = z["points"].shape[1]
        assert (np.sort(out[:, :-1], axis=1) == np.arange(n)).all() and (out[:, -1] == 0).all()


def test_oracle_guided_local_search_matches_reference(lso):
    for name, z in _load("ls_gls_*.npz"):
        for k in range(len(z["tours_in"])):
            out = lso.guided_local_search(z["points"][k], z["tours_in"][k], z["heat"][k],
                                          rounds=int(z["rounds"]), perturbation_moves=int(z["perturbation_moves"]))
            assert np.array_equal(out, z["tours_out"][k]), (name, k)
