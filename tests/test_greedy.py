"""Greedy edge-merge decoder (row f1): the CPU oracle against the goldens made by the reference's own
tsp_greedy (CPU), and the HIP decoder against both (GPU)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR

FILES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "greedy_*.npz")))
# the sparse fixture needs zero-score edges to join fragments; all of those tie exactly and the
# reference orders ties by numpy's unstable argsort: parity unpinned there (only validity is checked)
TIED = {"greedy_n100_sparse_p2"}


def _load(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["name"] = os.path.basename(path)[:-4]
    return d


def _dense(g):
    from oracle import greedy_oracle
    if "heatmap" in g:
        return g["heatmap"]
    return greedy_oracle.densify(g["values"], g["edge_index"], g["points"].shape[0])


@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_greedy_oracle_matches_reference_golden(path):
    from oracle import greedy_oracle
    g = _load(path)
    t = greedy_oracle.greedy(g["points"], _dense(g))
    n = g["points"].shape[0]
    assert all(sorted(r[:-1]) == list(range(n)) and r[0] == 0 and r[-1] == 0 for r in t)
    if g["name"] not in TIED:
        assert np.array_equal(t, g["tours"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=lambda p: os.path.basename(p)[:-4])
def test_gpu_greedy_matches_reference_golden(path):
    from ml4tspbench_amd import tsp_greedy
    from oracle import greedy_oracle
    g = _load(path)
    p = g["tours"].shape[0]
    if "heatmap" in g:
        tours = tsp_greedy(g["heatmap"], g["points"], parallel_sampling=p, device="cuda", extra_kwarg=1)
    else:
        tours = tsp_greedy(g["values"].reshape(-1), g["points"], edge_index_np=g["edge_index"],
                           sparse_graph=True, parallel_sampling=p)
    assert isinstance(tours, list) and len(tours) == p and isinstance(tours[0], list)
    got = np.array(tours)
    assert np.array_equal(got, greedy_oracle.greedy(g["points"], _dense(g)))   # same tie rule as the oracle
    if g["name"] not in TIED:
        assert np.array_equal(got, g["tours"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,inst,mpi,dtype", [(7, 3, 2, np.float32), (33, 5, 3, np.float64),
                                             (100, 16, 4, np.float32), (257, 3, 2, np.float32),
                                             (1000, 2, 1, np.float32),
                                             # list lengths 1 / 4 / 2 and both rescan widths
                                             (128, 3, 2, np.float32), (130, 4, 2, np.float64),
                                             (513, 2, 1, np.float32), (1800, 1, 1, np.float32),
                                             # no lists in LDS (n > 4500); third group-minimum register (n > 8192)
                                             (4600, 1, 2, np.float32), (8300, 1, 1, np.float32)])
def test_gpu_greedy_matches_oracle_random(n, inst, mpi, dtype):
    from ml4tspbench_amd import greedy_decode_device, two_opt_device, evaluate_tours
    from oracle import greedy_oracle
    rng = np.random.default_rng(90 + n)
    pts = rng.random((inst, n, 2))
    heat = rng.random((inst * mpi, n, n)).astype(dtype)
    dev = torch.device("cuda", 0)
    d_pts = torch.from_numpy(pts).to(dev)
    tours = greedy_decode_device(torch.from_numpy(heat).to(dev), d_pts, maps_per_instance=mpi)
    torch.cuda.synchronize()
    got = tours.cpu().numpy()
    for k in range(inst * mpi):
        ref = greedy_oracle.greedy(pts[k // mpi], heat[k])
        assert np.array_equal(got[k], ref[0]), k
    # the decoded tours feed the 2-opt kernel directly, all on the device
    res = two_opt_device(d_pts, tours, tours_per_instance=mpi)
    ev = evaluate_tours(d_pts, res.tours, tours_per_instance=mpi)
    torch.cuda.synchronize()
    assert ((ev.valid.cpu().numpy() & 7) == 7).all()
    l0 = evaluate_tours(d_pts, tours, tours_per_instance=mpi).lengths
    assert (ev.lengths <= l0 + 1e-9).all()


def _fuzz_maps(rng, n, kind, dtype, inst=2, mpi=2):
    pts = rng.random((inst, n, 2))
    heat = rng.random((inst * mpi, n, n))
    if kind == "sparse":                               # a few nonzero entries per row, the rest exactly 0
        k = min(n - 1, 5)
        heat = np.where(heat >= np.sort(heat, axis=2)[:, :, n - k][:, :, None], heat, 0.0)
    elif kind == "ties":                               # quantised heat: equal values stress the list refills
        heat = np.round(heat * 4) / 4
    elif kind == "grid":                               # lattice points + quantised heat: exactly equal keys
        side = int(np.ceil(np.sqrt(n)))
        g = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2)[:n] / side
        pts = np.stack([g, g[::-1]]).astype(np.float64)
        heat = np.round(heat * 2) / 2
    return pts, heat.astype(dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dtype,eps", [(30, np.float64, 1e-12), (100, np.float64, 1e-10), (128, np.float32, 1e-7),
                                         (100, np.float32, 3e-7), (64, np.float64, 1e-15), (200, np.float64, 1e-12),
                                         (600, np.float32, 1e-7),
                                         # around the decoders' margin (2^-18 = 3.8e-6): differences just above it
                                         # are decided by the fp32 filter keys alone, whose error must stay far below
                                         (100, np.float64, 2e-6), (128, np.float64, 8e-6), (100, np.float32, 4e-6),
                                         (300, np.float64, 4e-6), (90, np.float64, 2e-5)])
def test_gpu_greedy_near_ties_are_decided_exactly(n, dtype, eps):
    """Heat proportional to the distance, so that ALL keys -S/d agree to within eps (far below what the fp32
    filters of the decoders resolve: every choice has to fall through to the exact double comparison), plus
    coincident points (distance 0: keys -inf / +inf / NaN) and exact zeros.  Tours must equal the oracle's."""
    from ml4tspbench_amd import greedy_decode_device
    from oracle import greedy_oracle
    rng = np.random.default_rng(n + int(-np.log10(eps)))
    inst, mpi = 3, 2
    pts = rng.random((inst, n, 2))
    pts[:, 5] = pts[:, 2]                                   # a pair of coincident points per instance
    d = np.sqrt(((pts[:, :, None] - pts[:, None]) ** 2).sum(-1))
    heat = (0.37 * np.repeat(d, mpi, axis=0) * (1.0 + eps * rng.standard_normal((inst * mpi, n, n)))).astype(dtype)
    heat[:, 7, :] = 0                                       # exact zeros: keys -0.0, all tied
    heat[:, :, 7] = 0
    heat[1, 2, 5] = 0.25                                    # S > 0 at distance 0: key -inf (goes first)
    dev = torch.device("cuda", 0)
    got = greedy_decode_device(torch.from_numpy(heat).to(dev), torch.from_numpy(pts).to(dev),
                               maps_per_instance=mpi).cpu().numpy()
    for k in range(inst * mpi):
        ref = greedy_oracle.greedy(pts[k // mpi], heat[k])
        assert np.array_equal(got[k], ref[0]), k


@pytest.mark.gpu
def test_gpu_greedy_small_maps_through_the_general_kernels():
    """Heat maps of <= 128 nodes normally take the on-chip kernel; the diagnostic switch sends them through
    prepare + merge rounds (short rows, 8 rows in flight per wave): same tours."""
    import ctypes
    from ml4tspbench_amd import greedy_decode_device, _lib
    from oracle import greedy_oracle
    L = _lib.lib()
    L.tsp2opt_debug_greedy_cycles.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    try:
        assert L.tsp2opt_debug_greedy_cycles(3, 0, None) == 0
        for n, dtype in ((2, np.float32), (3, np.float64), (50, np.float32), (100, np.float64), (128, np.float32)):
            rng = np.random.default_rng(n)
            pts = rng.random((2, n, 2))
            heat = rng.random((4, n, n)).astype(dtype)
            heat[1][rng.random((n, n)) < 0.7] = 0          # mostly zeros: index-ordered ties
            got = greedy_decode_device(torch.from_numpy(heat).to(dev), torch.from_numpy(pts).to(dev),
                                       maps_per_instance=2).cpu().numpy()
            for k in range(4):
                assert np.array_equal(got[k], greedy_oracle.greedy(pts[k // 2], heat[k])[0]), (n, k)
    finally:
        L.tsp2opt_debug_greedy_cycles(2, 0, None)
    # and the default path on the same kind of input
    n = 100
    rng = np.random.default_rng(n)
    pts = rng.random((2, n, 2))
    heat = rng.random((4, n, n)).astype(np.float64)
    heat[1][rng.random((n, n)) < 0.7] = 0
    got = greedy_decode_device(torch.from_numpy(heat).to(dev), torch.from_numpy(pts).to(dev),
                               maps_per_instance=2).cpu().numpy()
    for k in range(4):
        assert np.array_equal(got[k], greedy_oracle.greedy(pts[k // 2], heat[k])[0]), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "sparse", "ties", "grid"])
def test_gpu_greedy_grid_of_n(kind):
    """A grid of N (list lengths, scan widths, row tails) x map kinds, both dtypes, against the oracle."""
    from ml4tspbench_amd import greedy_decode_device
    from oracle import greedy_oracle
    dev = torch.device("cuda", 0)
    bad = []
    for n in list(range(3, 70)) + list(range(70, 270, 11)) + [511, 777]:
        rng = np.random.default_rng(n)
        for dtype in (np.float32, np.float64):
            pts, heat = _fuzz_maps(rng, n, kind, dtype)
            got = greedy_decode_device(torch.from_numpy(heat).to(dev), torch.from_numpy(pts).to(dev),
                                       maps_per_instance=2)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            for m in range(heat.shape[0]):
                if not np.array_equal(got[m], greedy_oracle.greedy(pts[m // 2], heat[m])[0]):
                    bad.append((n, dtype.__name__, m))
    assert not bad, bad


# ---- sparse heat maps: (values, edge_index) without densifying ----------------------------------------
def _knn_inputs(rng, n, k, maps, dtype, self_loops=True, dup=0):
    """kNN graph like utils/utils.py:168-189 (source-major, the node itself among its neighbours) with
    positive heat values; `dup` extra duplicated edges."""
    pts = rng.random((n, 2))
    d = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    nn = np.argsort(d, axis=1)[:, :k] if self_loops else np.argsort(d, axis=1)[:, 1:k + 1]
    e0 = np.repeat(np.arange(n), nn.shape[1])
    e1 = nn.reshape(-1)
    if dup:
        pick = rng.integers(0, len(e0), size=dup)
        e0, e1 = np.concatenate([e0, e0[pick]]), np.concatenate([e1, e1[pick]])
    ei = np.stack([e0, e1]).astype(np.int64)
    vals = (rng.random((maps, ei.shape[1])) + 0.01).astype(dtype)
    return pts, ei, vals


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,maps,dtype,dup", [(12, 4, 3, np.float32, 0), (60, 6, 2, np.float64, 5),
                                                (200, 10, 3, np.float32, 0), (1000, 50, 2, np.float32, 40),
                                                (3000, 20, 1, np.float32, 0), (700, 3, 2, np.float32, 0)])
def test_gpu_sparse_greedy_matches_oracle_and_dense_path(n, k, maps, dtype, dup):
    """The sparse decoder against the CPU oracle on the densified heat map (the oracle's and the dense
    kernel's tie rule -- lower flat index -- is also the sparse path's rule for the unlisted, all-equal
    pairs, so the tours must be identical, ties included) and against the dense HIP decoder."""
    from ml4tspbench_amd import greedy_decode_device, greedy_decode_sparse_device
    from oracle import greedy_oracle
    rng = np.random.default_rng(700 + n + k)
    pts, ei, vals = _knn_inputs(rng, n, k, maps, dtype, dup=dup)
    dev = torch.device("cuda", 0)
    got = greedy_decode_sparse_device(torch.from_numpy(vals).to(dev), torch.from_numpy(ei).to(dev),
                                      torch.from_numpy(pts).to(dev)[None], maps_per_instance=maps).cpu().numpy()
    dense = greedy_oracle.densify(vals, ei, n)
    ref = greedy_oracle.greedy(pts, dense)
    assert np.array_equal(got, ref)
    d_got = greedy_decode_device(torch.from_numpy(dense).to(dev), torch.from_numpy(pts).to(dev)[None],
                                 maps_per_instance=maps).cpu().numpy()
    assert np.array_equal(d_got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,twins,listed", [(22, 2, 1, True), (22, 2, 1, False), (90, 3, 4, True), (300, 5, 9, False),
                                              (70, 2, 30, False)])
def test_gpu_sparse_greedy_coincident_points_with_zero_sum(n, k, twins, listed):
    """A pair of coincident points without a positive sum has the key -0.0 / 0 = NaN, which numpy sorts after the
    zero keys of all the other unlisted pairs (found by tests/fuzz/fuzz_greedy.py, seed 303 case 1520: the twin
    pair was listed with two zero values).  `listed`: the twin pairs stay in the list with zero values;
    otherwise they are not listed at all."""
    from ml4tspbench_amd import greedy_decode_device, greedy_decode_sparse_device
    from oracle import greedy_oracle
    rng = np.random.default_rng(n + 7 * twins)
    pts = rng.random((n, 2))
    pairs = rng.permutation(n)[:2 * twins].reshape(twins, 2)
    pts[pairs[:, 1]] = pts[pairs[:, 0]]
    d = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    nn = np.argsort(d, axis=1, kind="stable")[:, :k]
    e0, e1 = np.repeat(np.arange(n), k), nn.reshape(-1)
    twin = (d[e0, e1] == 0) & (e0 != e1)
    assert twin.any()
    if not listed:
        e1 = np.where(twin, e0, e1)                         # the twin's slot becomes one more self loop
    ei = np.stack([e0, e1]).astype(np.int64)
    vals = (rng.random((2, ei.shape[1])) + 0.01).astype(np.float32)
    vals[:, twin] = 0
    dev = torch.device("cuda", 0)
    got = greedy_decode_sparse_device(torch.from_numpy(vals).to(dev), torch.from_numpy(ei).to(dev),
                                      torch.from_numpy(pts).to(dev)[None], maps_per_instance=2).cpu().numpy()
    dense = greedy_oracle.densify(vals, ei, n)
    ref = greedy_oracle.greedy(pts, dense)
    assert np.array_equal(got, ref)
    d_got = greedy_decode_device(torch.from_numpy(dense).to(dev), torch.from_numpy(pts).to(dev)[None],
                                 maps_per_instance=2).cpu().numpy()
    assert np.array_equal(d_got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_gpu_sparse_greedy_batch_of_instances(dtype):
    """One call, one edge list per instance (equal lengths), maps_per_instance heat maps each: every map equals
    the single-instance call and the oracle."""
    from ml4tspbench_amd import greedy_decode_sparse_device
    from oracle import greedy_oracle
    rng = np.random.default_rng(91)
    inst, n, k, maps = 5, 130, 7, 3
    data = [_knn_inputs(rng, n, k, maps, dtype, dup=4) for _ in range(inst)]
    dev = torch.device("cuda", 0)
    pts = torch.from_numpy(np.stack([d[0] for d in data])).to(dev)
    ei = torch.from_numpy(np.stack([d[1] for d in data])).to(dev)
    vals = torch.from_numpy(np.concatenate([d[2] for d in data])).to(dev)
    got, diag = greedy_decode_sparse_device(vals, ei, pts, maps_per_instance=maps, return_diagnostics=True)
    got = got.cpu().numpy()
    assert diag["rounds"] > 0 and diag["listed_merges"] <= n - 1
    for q, (p, e, v) in enumerate(data):
        ref = greedy_oracle.greedy(p, greedy_oracle.densify(v, e, n))
        assert np.array_equal(got[q * maps:(q + 1) * maps], ref), q
    with pytest.raises(ValueError):
        greedy_decode_sparse_device(vals, ei[:2], pts, maps_per_instance=maps)


@pytest.mark.gpu
def test_gpu_sparse_greedy_front_end_and_fallbacks():
    from ml4tspbench_amd import tsp_greedy, greedy_decode_sparse_device, _lib
    from oracle import greedy_oracle
    rng = np.random.default_rng(3)
    pts, ei, vals = _knn_inputs(rng, 150, 8, 2, np.float32)
    tours = tsp_greedy(vals.reshape(-1), pts, edge_index_np=ei, sparse_graph=True, parallel_sampling=2)
    ref = greedy_oracle.greedy(pts, greedy_oracle.densify(vals, ei, 150))
    assert isinstance(tours, list) and np.array_equal(np.array(tours), ref)
    # a negative heat sum sorts after the unlisted pairs: the sparse kernels refuse, the front end densifies
    neg = vals.copy()
    neg[:, 5] = -5.0                                   # the reverse listing (<= 1.01) cannot lift the sum above 0
    dev = torch.device("cuda", 0)
    with pytest.raises(_lib.Tsp2optError):
        greedy_decode_sparse_device(torch.from_numpy(neg).to(dev), torch.from_numpy(ei).to(dev),
                                    torch.from_numpy(pts).to(dev)[None], maps_per_instance=2)
    tours = tsp_greedy(neg.reshape(-1), pts, edge_index_np=ei, sparse_graph=True, parallel_sampling=2)
    assert np.array_equal(np.array(tours), greedy_oracle.greedy(pts, greedy_oracle.densify(neg, ei, 150)))
    bad = ei.copy()
    bad[1, 7] = 150
    with pytest.raises(_lib.Tsp2optError):
        greedy_decode_sparse_device(torch.from_numpy(vals).to(dev), torch.from_numpy(bad).to(dev),
                                    torch.from_numpy(pts).to(dev)[None], maps_per_instance=2)
    # no listed edge at all: every pair ties, flat-index order builds the path 0-1-2-...
    empty = greedy_decode_sparse_device(torch.zeros((1, 0), dtype=torch.float32, device=dev),
                                        torch.zeros((2, 0), dtype=torch.int64, device=dev),
                                        torch.from_numpy(pts[:9]).to(dev)[None]).cpu().numpy()
    assert np.array_equal(empty, greedy_oracle.greedy(pts[:9], np.zeros((1, 9, 9), dtype=np.float32)))


@pytest.mark.gpu
def test_gpu_sparse_greedy_tsp10000_k50_full_size():
    """BASELINE.json configs[3]/[4] scale: 16 heat maps of 10 000 nodes with k = 50 listed neighbours stay a
    list (the dense decoder would need 12.8 GB of key matrices): valid tours that 2-opt improves, the workspace
    stays under 200 MB (100 MB in two calls), and two of the maps are checked against the dense HIP decoder."""
    from ml4tspbench_amd import greedy_decode_device, greedy_decode_sparse_device, two_opt_device, evaluate_tours, _lib
    n, k, maps = 10000, 50, 16
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    pts = torch.rand((maps, n, 2), generator=g, device=dev, dtype=torch.float64)
    vals_all, ei_all = [], []
    for m in range(maps):
        d = torch.cdist(pts[m], pts[m])
        dk, nn = d.topk(k, dim=1, largest=False)
        ei_all.append(torch.stack([torch.arange(n, device=dev).repeat_interleave(k), nn.reshape(-1)]))
        vals_all.append((torch.exp(-dk / (0.35 / n ** 0.5)) * (0.6 + 0.8 * torch.rand(dk.shape, generator=g, device=dev,
                                                                                   dtype=torch.float64))).float().reshape(1, -1))
        del d
    tours = greedy_decode_sparse_device(torch.cat(vals_all), torch.stack(ei_all), pts)      # 16 instances, one call
    need = _lib.lib().tsp2opt_greedy_sparse_workspace_bytes(maps, n, n * k, 1, _lib.DTYPE_F32)
    assert need < 200 * 2 ** 20, need                   # 24 B per listed edge (12 sorted rows + 12 lists) vs 12.8 GB
    # the same batch under a 100 MB workspace limit (two calls of 8 instances): identical tours
    halves = greedy_decode_sparse_device(torch.cat(vals_all), torch.stack(ei_all), pts, workspace_limit_bytes=100 * 2 ** 20)
    assert torch.equal(halves, tours)
    assert _lib.lib().tsp2opt_greedy_sparse_workspace_bytes(8, n, n * k, 1, _lib.DTYPE_F32) < 100 * 2 ** 20
    ev0 = evaluate_tours(pts, tours)
    assert ((ev0.valid.cpu().numpy() & 7) == 7).all()
    res = two_opt_device(pts, tours, max_iterations=300)
    ev = evaluate_tours(pts, res.tours)
    assert ((ev.valid.cpu().numpy() & 7) == 7).all() and (ev.lengths < ev0.lengths).all()
    for m in (0, 9):                                     # dense path on the densified map: same tour
        heat = torch.zeros((1, n, n), dtype=torch.float32, device=dev)
        heat[0].index_put_((ei_all[m][0], ei_all[m][1]), vals_all[m][0], accumulate=True)
        dense = greedy_decode_device(heat, pts[m][None])
        assert torch.equal(dense[0], tours[m]), m
        del heat
