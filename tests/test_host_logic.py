"""CPU: host-side mirror of the reference interfaces (no GPU, no oracle in the product path)."""
import numpy as np
import pytest
import torch


def test_registries_match_reference_names():
    import ml4tspbench_amd as amd
    assert amd.get_local_search_func("tsp", "2opt") is amd.tsp_2opt
    assert amd.LOCAL_SEARCH_CLASS[("tsp", None)] is None
    assert amd.get_local_search_by_name("2opt") is amd.ML4TSPNARTwoOpt
    with pytest.raises(ValueError, match="Cannot find the local_search"):
        amd.get_local_search_by_name("3opt")
    ls = amd.ML4TSPNARTwoOpt(time_limit=2.0, max_iterations_2opt=7)
    assert (ls.time_limit, ls.max_iterations_2opt) == (2.0, 7)
    for attr in ("nodes_num", "device", "tours", "heatmap", "points"):
        assert getattr(ls, attr) is None
    assert issubclass(amd.ML4TSPARTwoOpt, amd.ML4TSPARLocalSearch)


def test_base_class_loop_mapping_and_validation():
    """The generic per-tour loop of the base class (base.py:58-72) with a stub _local_search."""
    from ml4tspbench_amd import ML4TSPNARLocalSearch

    seen = []

    class Stub(ML4TSPNARLocalSearch):
        def _local_search(self, tour, points, heatmap=None):
            seen.append((points[0, 0], None if heatmap is None else heatmap[0, 0]))
            return tour

    n = 5
    tours = np.tile(np.array([0, 1, 2, 3, 4, 0]), (8, 1))
    points = np.arange(2 * n * 2, dtype=np.float64).reshape(2, n, 2) * 0 + np.array([10.0, 20.0])[:, None, None]
    heat = np.arange(4, dtype=np.float64)[:, None, None] * np.ones((4, n, n))
    out = Stub().local_search(torch.from_numpy(tours), points, heat, per_tours_num=2, per_heatmap_num=2)
    assert out.shape == (8, 6)
    # tour idx -> points[idx // 2 // 2], heatmap[idx // 2]   (base.py:61-62)
    assert [s[0] for s in seen] == [10.0] * 4 + [20.0] * 4
    assert [s[1] for s in seen] == [0.0, 0.0, 1.0, 1.0, 2.0, 2.0, 3.0, 3.0]
    bad = tours.copy()
    bad[3, 2] = 1
    with pytest.raises(ValueError, match="is not valid"):
        Stub().local_search(bad, points, None, per_tours_num=4)
    bad = tours.copy()
    bad[0] = [1, 2, 3, 4, 0, 1]          # permutation but does not end with 0
    with pytest.raises(ValueError, match="is not valid"):
        Stub().local_search(bad, points, None, per_tours_num=4)
    with pytest.raises(NotImplementedError):
        ML4TSPNARLocalSearch().local_search(tours, points)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device every front end raises instead of computing."""
    import ml4tspbench_amd as amd
    from ml4tspbench_amd._lib import Tsp2optError
    pts = np.random.default_rng(0).random((12, 2))
    t = np.append(np.arange(12), 0)
    with pytest.raises(Tsp2optError, match="no HIP device"):
        amd.tsp_2opt(pts, t)
    with pytest.raises(Tsp2optError, match="no HIP device"):
        amd.batched_two_opt_torch(pts, t)
    with pytest.raises(Tsp2optError, match="no HIP device"):
        amd.ML4TSPNARTwoOpt().local_search(t, pts)
    with pytest.raises(ValueError):
        amd.two_opt_device(torch.from_numpy(pts), torch.from_numpy(t))   # CPU tensors are rejected


def test_product_never_imports_the_oracle():
    import os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml4tspbench_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "two_opt_oracle" not in text or f == "_lib.py" and False, f


def test_partition_is_contiguous_and_balanced():
    from ml4tspbench_amd.sharding import partition
    for units in (0, 1, 7, 8, 9, 128, 1000):
        for world in (1, 2, 3, 8):
            blocks = partition(units, world)
            assert len(blocks) == world and blocks[0][0] == 0 and blocks[-1][1] == units
            sizes = [b - a for a, b in blocks]
            assert all(blocks[k][1] == blocks[k + 1][0] for k in range(world - 1))
            assert max(sizes) - min(sizes) <= 1


def test_assign_units_covers_every_unit_once_and_balances():
    from ml4tspbench_amd.sharding import assign_units, partition
    rng = np.random.default_rng(3)
    for units in (0, 1, 7, 8, 9, 128):
        for world in (1, 2, 3, 8):
            cost = rng.random(units) ** 3 * 100
            for balance in ("contiguous", "interleaved", "proxy"):
                got = assign_units(units, world, balance, cost.tolist())
                assert len(got) == world and sorted(u for g in got for u in g) == list(range(units))
                assert all(g == sorted(g) for g in got)
                sizes = [len(g) for g in got]
                assert max(sizes) - min(sizes) <= 1
            assert assign_units(units, world) == [list(range(a, b)) for a, b in partition(units, world)]
    # the proxy assignment evens out a batch whose cost grows along the batch; contiguous blocks do not
    cost = np.arange(64, dtype=float) ** 2
    load = lambda got: max(sum(cost[u] for u in g) for g in got) / (cost.sum() / len(got))
    assert load(assign_units(64, 8, "proxy", cost.tolist())) < 1.05 < load(assign_units(64, 8, "contiguous"))
    assert load(assign_units(64, 8, "interleaved")) < 1.25
    with pytest.raises(ValueError):
        assign_units(4, 2, "proxy")


def test_ls_ops_reject_inconsistent_batches_before_any_launch():
    """T must equal I * tours_per_instance, fixed_i / guide_k one entry per tour -- checked on
    the host (CPU tensors get as far as the device check, which comes last)."""
    import torch
    from ml4tspbench_amd import local_ops
    pts = torch.zeros((1, 6, 2), dtype=torch.float64)
    tours = torch.zeros((3, 7), dtype=torch.int64)
    with pytest.raises(ValueError, match="tours need"):
        local_ops.ls_step_device(pts, tours, 1)                       # 3 tours, one instance, tpi = 1
    with pytest.raises(ValueError, match="tours need"):
        local_ops.local_search_device(pts, tours, tours_per_instance=2)
    with pytest.raises(ValueError, match="HIP device"):
        local_ops.ls_step_device(pts, tours, 1, tours_per_instance=3)  # consistent: only the device is wrong
    with pytest.raises(ValueError, match="coordinate sets"):
        local_ops.tsp_relocate(np.zeros((6, 2)), np.zeros((3, 7), dtype=np.int64))   # one (N, 2) set, 3 tours


def test_txt_format_round_trip(tmp_path):
    """Dataset text format of the reference (generate.py:147-153, solvers/solver.py:51-78,115-138)."""
    from ml4tspbench_amd.io import read_txt, write_txt
    rng = np.random.default_rng(2)
    pts = rng.random((3, 6, 2))
    tours = np.array([np.concatenate([[0], rng.permutation(np.arange(1, 6)), [0]]) for _ in range(3)])
    f = str(tmp_path / "tsp6.txt")
    write_txt(f, pts, tours)
    first = open(f).readline().strip()
    # literal layout: "x0 y0 x1 y1 ... output t0+1 ... t0+1"
    assert first.startswith(f"{pts[0, 0, 0]} {pts[0, 0, 1]} {pts[0, 1, 0]}")
    assert first.split(" output ")[1].split(" ")[0] == "1" and first.endswith(" 1")
    p2, t2 = read_txt(f)
    assert np.array_equal(p2, pts) and np.array_equal(t2, tours)     # repr(float) round-trips exactly
    # a file without tours
    with open(f, "w") as fh:
        fh.write("0.1 0.2 0.3 0.4 0.5 0.6\n")
    p3, t3 = read_txt(f)
    assert p3.shape == (1, 3, 2) and t3 is None
    with pytest.raises(AssertionError):
        read_txt(str(tmp_path / "x.tsp"))


def test_txt_reader_parses_reference_written_lines():
    """Lines in the exact layout of the reference's writers (oracle/gen_golden_eval.py restates
    generate.py:147-153 -- closing node and TRAILING SPACE -- and TSPSolver.to_txt, solvers/solver.py:129-136)."""
    import os, shutil, tempfile
    from conftest import GOLDEN_DIR
    from ml4tspbench_amd.io import read_txt, write_txt
    src = os.path.join(GOLDEN_DIR, "txt_format_lines.txt")
    lines = open(src).read().splitlines(keepends=True)
    assert len(lines) == 4 and lines[0].endswith(" 1 \n") and not lines[3].endswith(" \n")
    z = np.load(os.path.join(GOLDEN_DIR, "txt_format_parsed.npz"))
    with tempfile.TemporaryDirectory() as d:
        for k, line in enumerate(lines):                    # instances differ in N: one file per line
            f = os.path.join(d, f"l{k}.txt")
            open(f, "w").write(line)
            pts, tours = read_txt(f)
            assert pts.shape == (1,) + z[f"p{k}"].shape and np.array_equal(pts[0], z[f"p{k}"])
            assert tours.dtype == np.int64 and np.array_equal(tours[0], z[f"t{k}"])
            assert tours[0, 0] == 0 and tours[0, -1] == 0
        # our writer reproduces TSPSolver.to_txt's line byte for byte
        f = os.path.join(d, "w.txt")
        write_txt(f, z["p3"][None], z["t3"][None])
        assert open(f).read() == lines[3]


def test_oracle_tour_lengths_match_reference_evaluator(oracle):
    """oracle.tour_lengths (the checker other tests use for lengths) against TSPEvaluator costs."""
    from conftest import eval_golden_files
    files = eval_golden_files()
    assert len(files) >= 5
    for path in files:
        z = np.load(path)
        ref = oracle.tour_lengths(z["points"], z["tours"], int(z["tours_per_instance"]))
        np.testing.assert_allclose(ref, z["costs"], rtol=1e-5 if z["points"].dtype == np.float32 else 1e-12)
