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


# ---- GPU: the HIP operators through the C ABI ---------------------------------------------------------
import torch  # noqa: E402


def _dev(x, dtype):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device="cuda", dtype=dtype)


@pytest.mark.gpu
def test_gpu_relocate_matches_reference():
    import ml4tspbench_amd as amd
    for name, z in _load("ls_relocate_*.npz"):
        n = z["points"].shape[1]
        out = amd.tsp_relocate(z["points"], z["tours_in"], adj_mat=np.zeros((len(z["tours_in"]), n, n)), device="cuda")
        assert isinstance(out, list) and isinstance(out[0], list)
        assert np.array_equal(np.array(out), z["tours_out"]), name
        f = amd.get_local_search_func("tsp", "relocate")
        assert np.array_equal(np.array(f(z["points"], z["tours_in"].tolist(), np.zeros((len(z["tours_in"]), n, n)))), z["tours_out"])


@pytest.mark.gpu
def test_gpu_single_operator_calls_match_reference():
    import ml4tspbench_amd as amd
    for name, z in _load("ls_step_*.npz"):
        pts, tin = _dev(z["points"], torch.float64), _dev(z["tours_in"], torch.int64)
        for op in (0, 1):
            for g in (0, 1):
                for f in (0, 1):
                    out, delta = amd.ls_step_device(
                        pts, tin, op, fixed_i=_dev(z["fixed_i"], torch.int32) if f else None,
                        guide_k=_dev(z["guide_k"], torch.float64) if g else None,
                        penalty=_dev(z["penalty"], torch.float64) if g else None)
                    key = f"op{op}_g{g}_f{f}"
                    assert np.array_equal(out.cpu().numpy(), z["out_" + key]), (name, key)
                    assert np.array_equal(delta.cpu().numpy(), z["delta_" + key]), (name, key)   # bit-exact doubles
        assert torch.equal(tin.cpu(), torch.from_numpy(z["tours_in"]))                      # input not mutated


@pytest.mark.gpu
def test_gpu_local_search_matches_reference_and_oracle(lso):
    import ml4tspbench_amd as amd
    for name, z in _load("ls_loop_*.npz"):
        for k in range(len(z["tours_in"])):
            out = amd.tsp_ls(z["points"][k], z["tours_in"][k].tolist(), np.zeros((2, 2)), device="cuda")
            assert np.array_equal(np.array(out[0]), z["tours_out"][k]), (name, k)
    rng = np.random.default_rng(12)
    for n, inst, tpi in ((7, 5, 2), (64, 6, 3), (150, 3, 2), (300, 2, 1)):
        pts = rng.random((inst, n, 2))
        tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
        for k in range(inst * tpi):
            tours[k, 1:n] = rng.permutation(np.arange(1, n))
        ref, ref_rounds = lso.local_search(pts, tours, tours_per_instance=tpi)
        out, rounds = amd.local_search_device(_dev(pts, torch.float64), _dev(tours, torch.int64), tours_per_instance=tpi)
        assert np.array_equal(out.cpu().numpy(), ref), n
        assert np.array_equal(rounds.cpu().numpy(), ref_rounds), n
        # relocate over long distances (chunked shifts) and both operators on random tours
        for op in (0, 1):
            r1, d1 = lso.step(pts, tours, op, tours_per_instance=tpi)
            o1, e1 = amd.ls_step_device(_dev(pts, torch.float64), _dev(tours, torch.int64), op, tours_per_instance=tpi)
            assert np.array_equal(o1.cpu().numpy(), r1) and np.array_equal(e1.cpu().numpy(), d1), (n, op)


@pytest.mark.gpu
def test_gpu_long_relocate_shift_both_directions(lso):
    """A relocate over more than one chunk of 256 positions, towards the front and towards the back."""
    import ml4tspbench_amd as amd
    n = 700
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    pts = np.stack([np.cos(ang), np.sin(ang)], axis=1)[None]
    for src, dst in ((20, 650), (650, 20)):
        t = np.arange(n + 1, dtype=np.int64)
        t[-1] = 0
        node = t[src]
        body = np.delete(t[:-1], src)
        body = np.insert(body, dst, node)               # node `src` sits far away from its ring neighbours
        tour = np.append(body, 0)[None]
        ref, d = lso.step(pts, tour, 1)
        assert d[0] < 0 and not np.array_equal(ref, tour)
        out, e = amd.ls_step_device(_dev(pts, torch.float64), _dev(tour, torch.int64), 1)
        assert np.array_equal(out.cpu().numpy(), ref) and e.cpu().numpy()[0] == d[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n,rounds", [(2000, 3), (12000, 1)])
def test_gpu_large_n_scan_paths_match_oracle(lso, n, rounds):
    """The two scan paths that only large tours reach (csrc/ls_ops.hip ls_best): rows per lane once the flat
    candidate tables no longer fit the LDS (n above about 1555) and coordinates read from global memory
    (n above about 9100).  Both operators for one step, and a capped local_search."""
    import ml4tspbench_amd as amd
    rng = np.random.default_rng(n)
    inst = 2
    pts = rng.random((inst, n, 2))
    tours = np.zeros((inst, n + 1), dtype=np.int64)
    for k in range(inst):
        # a short random walk away from a space-filling order keeps the oracle's O(n^2) loops to a few rounds
        order = np.lexsort((pts[k, :, 1], np.floor(pts[k, :, 0] * 24)))
        order = order[order != 0]
        sw = rng.integers(0, n - 1, size=(40, 2))
        for a, b in sw:
            order[[a, b]] = order[[b, a]]
        tours[k, 1:n] = order
    d_p, d_t = _dev(pts, torch.float64), _dev(tours, torch.int64)
    for op in (0, 1):
        ref, d = lso.step(pts, tours, op)
        out, e = amd.ls_step_device(d_p, d_t, op)
        assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(e.cpu().numpy(), d), (n, op)
    ref, ref_rounds = lso.local_search(pts, tours, max_rounds=rounds)
    out, got_rounds = amd.local_search_device(d_p, d_t, max_rounds=rounds)
    assert np.array_equal(out.cpu().numpy(), ref) and np.array_equal(got_rounds.cpu().numpy(), ref_rounds)


@pytest.mark.gpu
def test_gpu_guided_local_search_matches_reference():
    import ml4tspbench_amd as amd
    for name, z in _load("ls_gls_*.npz"):
        for k in range(len(z["tours_in"])):
            out = amd.tsp_gls(z["points"][k], z["tours_in"][k], z["heat"][k], device="cuda",
                              time_limit=1.0, perturbation_moves=int(z["perturbation_moves"]),
                              max_rounds=int(z["rounds"]))
            assert np.array_equal(np.array(out[0]), z["tours_out"][k]), (name, k)
    z = _load("ls_gls_n12.npz")[0][1]
    plug = amd.ML4TSPNARGuidedLS(time_limit=0.05, perturbation_moves=5)
    res = plug.local_search(z["tours_in"][:2], z["points"][:2], z["heat"][:2])          # wall-clock limited run
    assert res.shape == (2, 13) and (np.sort(res[:, :-1], axis=1) == np.arange(12)).all()


@pytest.mark.gpu
def test_gpu_batched_guided_local_search_matches_oracle(lso):
    """guided_ls_device: all tours of a batch side by side, whole rounds in one launch (tsp2opt_gls_rounds),
    against the oracle's per-tour driver; the same rounds split over several launches (state kept in device
    memory between calls); the plugin's batched local_search with its instance / heat-map mapping."""
    import ml4tspbench_amd as amd
    rng = np.random.default_rng(77)
    for n, inst, per, rounds, pm in ((12, 3, 2, 3, 4), (40, 2, 3, 2, 10), (90, 2, 1, 2, 15)):
        pts = rng.random((inst, n, 2))
        T = inst * per
        tours = np.zeros((T, n + 1), dtype=np.int64)
        for k in range(T):
            tours[k, 1:n] = rng.permutation(np.arange(1, n))
        d = np.linalg.norm(pts[:, :, None] - pts[:, None], axis=-1)
        heat = -np.exp(-np.repeat(d, per, axis=0) / (0.35 / np.sqrt(n))) * (0.6 + 0.8 * rng.random((T, n, n)))
        heat[0, :, 3] = 0.5                              # positive heat: regret 0 after the clamp
        ref = np.array([lso.guided_local_search(pts[k // per], tours[k], heat[k], rounds=rounds,
                                                perturbation_moves=pm) for k in range(T)])
        got = amd.guided_ls_device(_dev(pts, torch.float64), _dev(tours, torch.int64), _dev(heat, torch.float64),
                                   tours_per_instance=per, tours_per_map=1, perturbation_moves=pm, max_rounds=rounds)
        assert np.array_equal(got.cpu().numpy(), ref), n
        plug = amd.ML4TSPNARGuidedLS(perturbation_moves=pm, max_rounds=rounds)
        out = plug.local_search(tours, pts, heat, per_tours_num=1, per_heatmap_num=per)
        assert np.array_equal(out, ref), n
    # one heat map shared by two tours (per_tours_num = 2)
    n, pm = 30, 6
    pts = rng.random((2, n, 2))
    tours = np.zeros((4, n + 1), dtype=np.int64)
    for k in range(4):
        tours[k, 1:n] = rng.permutation(np.arange(1, n))
    heat = -rng.random((2, n, n))
    ref = np.array([lso.guided_local_search(pts[k // 2], tours[k], heat[k // 2], rounds=2, perturbation_moves=pm)
                    for k in range(4)])
    out = amd.ML4TSPNARGuidedLS(perturbation_moves=pm, max_rounds=2).local_search(tours, pts, heat, per_tours_num=2)
    assert np.array_equal(out, ref)


@pytest.mark.gpu
def test_gpu_ls_rejects_bad_tours():
    import ml4tspbench_amd as amd
    pts = np.random.default_rng(0).random((1, 9, 2))
    t = np.array([[0, 1, 2, 3, 44, 5, 6, 7, 8, 0]])
    with pytest.raises(ValueError):
        amd.tsp_relocate(pts, t, adj_mat=np.zeros((1, 9, 9)))
    with pytest.raises(ValueError):
        amd.tsp_ls(pts[0], t[0], np.zeros((9, 9)))
