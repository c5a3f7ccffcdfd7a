"""GPU tests of the drop-in Python boundary (same call shapes as the reference's
search/two_opt.py and ml4tsp plugin), checked against the golden vectors / oracle."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, eval_golden_files, load_golden

pytestmark = pytest.mark.gpu


def test_tsp_2opt_rank_dtype_and_kwargs():
    from ml4tspbench_amd import tsp_2opt, get_local_search_func
    g = load_golden(f"{GOLDEN_DIR}/g8_n100_f32values_f64.npz")
    pts32 = g["points"][0]                       # float32 values, cast to fp64 inside (two_opt.py:67)
    tour = g["tours_in"][0]
    out = tsp_2opt(pts32, tour.tolist(), None, device="cpu", max_iterations_2opt=5000,
                   parallel_sampling=1, some_other_cli_flag="x")
    assert out.ndim == 1 and out.dtype == np.int64
    assert np.array_equal(out, g["tours_out"][0])
    f = get_local_search_func("tsp", "2opt")
    out2 = f(pts32, tour[None], None, device=torch.device("cuda", 0))
    assert out2.shape == (1, 101) and np.array_equal(out2[0], g["tours_out"][0])
    # int32 tours, torch inputs
    out3 = f(torch.from_numpy(pts32), torch.from_numpy(tour.astype(np.int32)))
    assert np.array_equal(out3, g["tours_out"][0])


def test_batched_two_opt_torch_grouped_semantics():
    from ml4tspbench_amd import batched_two_opt_torch
    g = load_golden(f"{GOLDEN_DIR}/g3_n50_group16_f64.npz")
    tin = g["tours_in"][:16].copy()
    keep = tin.copy()
    out, it = batched_two_opt_torch(g["points"][0], tin, max_iterations=5000, device="cuda")
    assert out.shape == (16, 51) and it == int(g["iterators"][0])
    assert np.array_equal(out, g["tours_out"][:16])
    assert np.array_equal(tin, keep)            # input never mutated (two_opt.py:12)
    g7 = load_golden(f"{GOLDEN_DIR}/g7_n100_group8_f32.npz")   # dtype of points preserved: fp32 math
    out, it = batched_two_opt_torch(g7["points"][0], g7["tours_in"][:8])
    assert np.array_equal(out, g7["tours_out"][:8]) and it == int(g7["iterators"][0])
    out1, it1 = batched_two_opt_torch(g["points"][0], g["tours_in"][3])     # 1-D in, 2-D out
    assert out1.shape == (1, 51)


def test_plugin_batched_equals_per_tour_reference():
    from ml4tspbench_amd import ML4TSPNARTwoOpt, ML4TSPARTwoOpt, get_local_search_by_name
    g = load_golden(f"{GOLDEN_DIR}/g3_n50_group16_as_indep_f64.npz")
    ls = get_local_search_by_name("2opt")()
    assert isinstance(ls, ML4TSPNARTwoOpt) and ls.max_iterations_2opt == 5000 and ls.time_limit == "auto"
    out = ls.local_search(g["tours_in"], g["points"], None, per_tours_num=16)
    assert isinstance(out, np.ndarray) and np.array_equal(out, g["tours_out"])
    assert ls.nodes_num == 50
    # torch tensors on the device, keyword call like narsolver.py:105-108
    dev = torch.device("cuda", 0)
    out = ls.local_search(tours=torch.from_numpy(g["tours_in"]).to(dev),
                          points=torch.from_numpy(g["points"]).to(dev), heatmap=None,
                          per_tours_num=8, per_heatmap_num=2)
    assert np.array_equal(out, g["tours_out"])
    one = ML4TSPARTwoOpt()._local_search(g["tours_in"][5], g["points"][0])
    assert one.shape == (51,) and np.array_equal(one, g["tours_out"][5])


def test_plugin_rejects_invalid_result():
    from ml4tspbench_amd import ML4TSPNARTwoOpt
    rng = np.random.default_rng(3)
    pts = rng.random((1, 12, 2))
    t = np.append(np.append(0, rng.permutation(np.arange(1, 12))), 0)
    t[4] = t[5]                                   # not a permutation
    with pytest.raises(ValueError, match="is not valid"):
        ML4TSPNARTwoOpt().local_search(t, pts)
    t2 = np.arange(13); t2[-1] = 0; t2[3] = 99
    with pytest.raises(ValueError):
        ML4TSPNARTwoOpt().local_search(t2, pts)


def test_evaluate_tours_flags_lengths_best(oracle):
    from ml4tspbench_amd import evaluate_tours
    rng = np.random.default_rng(11)
    inst, tpi, n = 7, 5, 83
    pts = rng.random((inst, n, 2))
    tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
    for k in range(inst * tpi):
        tours[k, 1:n] = rng.permutation(np.arange(1, n))
    tours[3, 10] = tours[3, 11]                 # duplicate -> not a permutation
    tours[8, -1] = 5                            # does not end at node 0
    tours[9] = np.roll(tours[9], 1); tours[9, 0] = tours[9, -1]   # closed, starts elsewhere
    tours[12, 4] = n + 3                        # out of range
    dev = torch.device("cuda", 0)
    ev = evaluate_tours(torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev), tours_per_instance=tpi)
    torch.cuda.synchronize()
    valid = ev.valid.cpu().numpy()
    perm = np.array([sorted(t[:-1]) == list(range(n)) for t in tours])
    assert np.array_equal((valid & 1) == 1, perm)
    assert np.array_equal((valid & 2) == 2, tours[:, -1] == 0)
    assert np.array_equal((valid & 4) == 4, tours[:, -1] == tours[:, 0])
    lengths = ev.lengths.cpu().numpy()
    ok = np.ones(len(tours), dtype=bool); ok[12] = False
    ref = oracle.tour_lengths(pts, np.where(ok[:, None], tours, 0), tpi)
    np.testing.assert_allclose(lengths[ok], ref[ok], rtol=1e-12)
    assert np.isnan(lengths[12])
    # TSPEvaluator-style cost (utils/evaluator.py:5-13) on one tour
    d = np.sqrt(((pts[0][:, None] - pts[0][None]) ** 2).sum(-1))
    cost = sum(d[tours[0][i], tours[0][i + 1]] for i in range(n))
    assert abs(lengths[0] - cost) <= 1e-12 * cost
    best = ev.best_index.cpu().numpy()
    L = np.where(np.isnan(lengths), np.inf, lengths).reshape(inst, tpi)
    assert np.array_equal(best, np.arange(inst) * tpi + L.argmin(1))
    np.testing.assert_array_equal(ev.best_length.cpu().numpy(), L.min(1))


@pytest.mark.parametrize("path", eval_golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_evaluate_matches_reference_evaluator(path):
    """tsp2opt_evaluate against fixtures whose costs come from the reference's TSPEvaluator
    (utils/evaluator.py:5-13, run by oracle/gen_golden_eval.py) and whose validity verdicts restate
    base.py:31-32,67-72.  Flags and best index exact; lengths rtol 1e-12 (a double sum in another order;
    fp32 fixture: the reference accumulates float32 distances, 1e-5)."""
    from ml4tspbench_amd import evaluate_tours
    z = np.load(path)
    tpi = int(z["tours_per_instance"])
    dev = torch.device("cuda", 0)
    ev = evaluate_tours(torch.from_numpy(z["points"]).to(dev), torch.from_numpy(z["tours"]).to(dev),
                        tours_per_instance=tpi)
    torch.cuda.synchronize()
    valid = ev.valid.cpu().numpy()
    assert np.array_equal((valid & 1) == 1, z["is_permutation"])
    assert np.array_equal((valid & 2) == 2, z["ends_at_zero"])
    f32 = z["points"].dtype == np.float32
    np.testing.assert_allclose(ev.lengths.cpu().numpy(), z["costs"], rtol=1e-5 if f32 else 1e-12)
    np.testing.assert_allclose(ev.best_length.cpu().numpy(), z["best_cost"], rtol=1e-5 if f32 else 1e-12)
    if not f32:
        assert np.array_equal(ev.best_index.cpu().numpy(), z["best_index"])


def test_plugin_validity_rule_on_reference_fixtures():
    """The plugin's ValueError fires exactly on the tours the reference's rule rejects."""
    from ml4tspbench_amd import ML4TSPNARLocalSearch
    z = np.load(eval_golden_files()[0])
    ls = ML4TSPNARLocalSearch()
    ls.nodes_num = z["points"].shape[1]
    for t, perm, e0 in zip(z["tours"], z["is_permutation"], z["ends_at_zero"]):
        assert ls.is_valid_tour(t) == bool(perm)
        if perm and e0:
            ls._validate(t[None])
        else:
            with pytest.raises(ValueError, match="is not valid"):
                ls._validate(t[None])


def test_anytime_checkpoints_follow_the_capped_trajectory(oracle):
    """two_opt_anytime (docs/gap_time.py:257-327): with iteration slices the lengths after slice s are those
    of the oracle capped at k (s + 1) iterations, and the final state is the plain run's; with wall-clock
    slices the lengths never increase and converged tours equal the plain run's."""
    import ml4tspbench_amd as amd
    rng = np.random.default_rng(21)
    inst, tpi, n, k, S = 3, 2, 120, 25, 8
    pts = rng.random((inst, n, 2))
    tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
    for t in range(inst * tpi):
        tours[t, 1:n] = rng.permutation(np.arange(1, n))
    dev = torch.device("cuda", 0)
    d_p, d_t = torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev)
    res = amd.two_opt_anytime(d_p, d_t, record_samples=S, slice_iterations=k, tours_per_instance=tpi)
    for s in range(S):
        ref, ref_it, _ = oracle.solve(pts, tours, tours_per_instance=tpi, max_iterations=k * (s + 1), threads=4)
        np.testing.assert_allclose(res.lengths[:, s].cpu().numpy(), oracle.tour_lengths(pts, ref, tpi), rtol=1e-12)
        assert np.array_equal(res.iterations[:, s].cpu().numpy(), ref_it), s
    assert np.array_equal(res.tours.cpu().numpy(), ref)
    assert torch.equal(d_t.cpu(), torch.from_numpy(tours))                      # inputs untouched
    full, full_it, _ = oracle.solve(pts, tours, tours_per_instance=tpi, threads=4)
    conv = res.converged.cpu().numpy()
    assert np.array_equal(conv, full_it <= k * S) or np.array_equal(conv, full_it < k * S)
    timed = amd.two_opt_anytime(d_p, d_t, record_samples=5, limit_time=2.0, tours_per_instance=tpi)
    L = timed.lengths.cpu().numpy()
    assert (np.diff(L, axis=1) <= 1e-12).all() and (np.diff(timed.times.numpy()) > 0).all()
    assert timed.converged.all() and np.array_equal(timed.tours.cpu().numpy(), full)


def test_zero_tour_batches_are_accepted():
    """A rank of a sharded solve may own no instance (1 instance on 2 GPUs): two_opt_device and the C entry
    point take an empty batch and return empty results without launching anything."""
    import ml4tspbench_amd as amd
    from ml4tspbench_amd import _lib
    dev = torch.device("cuda", 0)
    pts = torch.zeros((0, 40, 2), dtype=torch.float64, device=dev)
    tours = torch.zeros((0, 41), dtype=torch.int64, device=dev)
    for group in (1, 4):
        res = amd.two_opt_device(pts, tours, tours_per_instance=4, group_size=group)
        torch.cuda.synchronize()
        assert res.tours.shape == (0, 41) and res.iterations.shape == (0,) and res.lengths.shape == (0,)
    L = _lib.lib()
    rc = L.tsp2opt_solve(None, None, None, None, None, None, 0, 40, 1, 1, 5000, 0.0, 1, 0, None, 0, None)
    assert rc == 0, L.tsp2opt_last_error()
    assert L.tsp2opt_solve_noalloc(None, None, None, None, None, None, 0, 40, 1, 1, 5000, 0.0, 1, 0, None) == 0
    torch.cuda.synchronize()
    ev = amd.evaluate_tours(pts, tours, tours_per_instance=1)
    assert ev.lengths.shape == (0,)


def test_solve_sharded_world1_nccl_with_the_real_solver(oracle):
    """sharding.solve_sharded with two_opt_device under an initialised nccl (RCCL) group of one rank: the code
    path of the N-GPU launch (assignment, solve, final gather) with the real kernel, every input form."""
    import os
    import socket
    import torch.distributed as dist
    from ml4tspbench_amd.sharding import solve_sharded
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        rng = np.random.default_rng(31)
        inst, tpi, n = 6, 4, 60
        pts = rng.random((inst, n, 2))
        tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
        for k in range(inst * tpi):
            tours[k, 1:n] = rng.permutation(np.arange(1, n))
        d_p, d_t = torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev)
        for group in (1, 4):
            ref, ref_it, _ = oracle.solve(pts, tours, tours_per_instance=tpi, group_size=group)
            for kw in ({}, {"balance": "interleaved"}, {"balance": "proxy"}, {"local_inputs": True}):
                r = solve_sharded(d_p, d_t, tours_per_instance=tpi, group_size=group, want_result=True, **kw)
                assert np.array_equal(r.tours.cpu().numpy(), ref), (group, kw)
                assert np.array_equal(r.iterations.cpu().numpy(), ref_it), (group, kw)
                assert torch.equal(r.tours[r.tour_index], r.local.tours)
        # an all_reduce on the same communicator still works afterwards (the solve left the stream usable)
        x = torch.ones(4, device=dev)
        dist.all_reduce(x)
        assert float(x.sum().item()) == 4.0
        e = solve_sharded(d_p[:0], d_t[:0], tours_per_instance=tpi, local_inputs=True)
        assert e[0].shape == (0, n + 1)
    finally:
        dist.destroy_process_group()
