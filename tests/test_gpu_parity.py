"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libtsp2opt_hip.so, against (a) the golden vectors made by the reference itself and (b) the CPU
oracle on seeded inputs.  Tours, iteration counts and sweep counts are integers: bit-exact.
Tour lengths are a double-precision sum in a different order: rtol 1e-12.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR as GOLDEN, golden_files, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import ml4tspbench_amd
    from ml4tspbench_amd import _lib
    _lib.lib()                       # fails loudly if the HIP extension is missing
    assert torch.cuda.is_available()
    return ml4tspbench_amd


def _run(amd, pts, tours, tpi, group, max_it, lengths=True):
    dev = torch.device("cuda", 0)
    res = amd.two_opt_device(torch.from_numpy(np.ascontiguousarray(pts)).to(dev),
                             torch.from_numpy(np.ascontiguousarray(tours)).to(dev),
                             tours_per_instance=tpi, group_size=group, max_iterations=max_it,
                             want_lengths=lengths)
    torch.cuda.synchronize()
    return (res.tours.cpu().numpy(), res.iterations.cpu().numpy(), res.sweeps.cpu().numpy(),
            res.lengths.cpu().numpy() if lengths else None, res.status.cpu().numpy())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_sqrt_is_correctly_rounded(amd, dtype):
    """SURVEY hard part (b): the device sqrt used by the exact path must agree with the host's
    IEEE sqrt to 0 ulp (fp64: 4e7 values; fp32: 4e7 values incl. squared-distance-like ranges)."""
    from ml4tspbench_amd import _lib
    rng = np.random.default_rng(7)
    n = 10_000_000
    chunks = [rng.random(n), rng.random(n) * 2.0, rng.random(n) ** 8, rng.random(n) * 1e6,
              np.array([0.0, 0.5, 1.0, 2.0, 4.0, 1e-300, 1e300, 2.0 ** -1060, 3.0, 0.1])]
    x = np.concatenate(chunks).astype(dtype)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    code = _lib.DTYPE_F64 if dtype == np.float64 else _lib.DTYPE_F32
    rc = _lib.lib().tsp2opt_sqrt_check(xd.data_ptr(), yd.data_ptr(), xd.numel(), code, 0,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    ref = np.sqrt(x)
    assert np.array_equal(y, ref), f"{(y != ref).sum()} of {len(x)} differ"


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_matches_reference_golden(amd, path):
    g = load_golden(path)
    pts = g["points"].astype(np.float64) if g["entry"] == "tsp_2opt" else g["points"]
    out, iters, sweeps, lengths, status = _run(amd, pts, g["tours_in"], g["tours_per_instance"],
                                               g["group_size"], g["max_iterations"])
    assert (status & 15).max(initial=0) == 0
    assert np.array_equal(out, g["tours_out"]), g["name"]
    per_call = iters[::g["group_size"]]
    assert np.array_equal(per_call, g["iterators"]), g["name"]
    assert np.array_equal(iters, np.repeat(per_call, g["group_size"]))
    capped = per_call >= g["max_iterations"]
    assert np.array_equal(sweeps[::g["group_size"]], np.where(capped, per_call, per_call + 1))


def _random_batch(rng, inst, tpi, n, dtype):
    pts = rng.random((inst, n, 2)).astype(dtype)
    tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
    for k in range(inst * tpi):
        tours[k, 1:n] = rng.permutation(np.arange(1, n))
    return pts, tours


CASES = [
    # n, instances, tours/instance, group, dtype, cap
    (9, 6, 2, 1, np.float64, 5000), (10, 6, 2, 2, np.float64, 5000), (12, 4, 4, 4, np.float32, 5000),
    (17, 8, 1, 1, np.float64, 5000), (31, 4, 4, 1, np.float32, 5000), (50, 16, 4, 1, np.float64, 5000),
    (50, 8, 8, 8, np.float64, 5000), (64, 4, 2, 1, np.float64, 3), (100, 24, 1, 1, np.float64, 5000),
    (100, 4, 8, 8, np.float32, 5000), (127, 3, 2, 2, np.float64, 5000), (161, 4, 2, 1, np.float64, 5000),
    (200, 4, 4, 4, np.float64, 5000), (321, 2, 2, 1, np.float32, 5000), (500, 2, 8, 1, np.float64, 5000),
    (500, 2, 8, 8, np.float64, 5000), (500, 2, 4, 4, np.float32, 5000), (641, 2, 1, 1, np.float64, 5000),
    (1000, 2, 2, 1, np.float64, 5000), (1000, 1, 4, 4, np.float64, 300), (1500, 1, 2, 1, np.float32, 200),
]


@pytest.mark.parametrize("n,inst,tpi,group,dtype,cap", CASES)
def test_gpu_matches_oracle_random(amd, oracle, n, inst, tpi, group, dtype, cap):
    rng = np.random.default_rng(1000 + n * 7 + group)
    pts, tours = _random_batch(rng, inst, tpi, n, dtype)
    ref_out, ref_it, ref_sw = oracle.solve(pts, tours, tours_per_instance=tpi, group_size=group,
                                           max_iterations=cap, threads=8)
    out, iters, sweeps, lengths, status = _run(amd, pts, tours, tpi, group, cap)
    assert np.array_equal(out, ref_out)
    assert np.array_equal(iters, ref_it)
    assert np.array_equal(sweeps, ref_sw)
    ref_len = oracle.tour_lengths(pts, ref_out, tpi)
    np.testing.assert_allclose(lengths, ref_len, rtol=1e-12)


def test_gpu_inputs_not_mutated_and_status(amd):
    rng = np.random.default_rng(5)
    pts, tours = _random_batch(rng, 4, 1, 40, np.float64)
    dev = torch.device("cuda", 0)
    t_dev = torch.from_numpy(tours).to(dev)
    res = amd.two_opt_device(torch.from_numpy(pts).to(dev), t_dev)
    torch.cuda.synchronize()
    assert torch.equal(t_dev.cpu(), torch.from_numpy(tours))
    assert (res.status.cpu().numpy() == 0).all()
    # bad index / not closed / NaN coordinates are reported per tour, tours left untouched
    bad = tours.copy()
    bad[0, 3] = 40
    bad[1, -1] = 5
    p2 = pts.copy()
    p2[2, 7, 0] = np.nan
    out, iters, sweeps, _, status = _run(amd, p2, bad, 1, 1, 5000)
    from ml4tspbench_amd import _lib
    assert status[0] & _lib.ST_BAD_INDEX and status[1] & _lib.ST_NOT_CLOSED and status[2] & _lib.ST_NONFINITE
    assert status[3] == 0
    assert np.array_equal(out[:3], bad[:3]) and (iters[:3] == 0).all()


def test_gpu_time_limit_stops_early(amd):
    rng = np.random.default_rng(6)
    pts, tours = _random_batch(rng, 2, 1, 1000, np.float64)
    dev = torch.device("cuda", 0)
    res = amd.two_opt_device(torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev),
                             time_limit=1e-4)
    torch.cuda.synchronize()
    from ml4tspbench_amd import _lib
    st = res.status.cpu().numpy()
    assert (st & _lib.ST_TIME).all() and (res.iterations.cpu().numpy() < 600).all()
    out = res.tours.cpu().numpy()
    assert (np.sort(out[:, :-1], axis=1) == np.arange(1000)).all()


# ---- several workgroups per tour (cluster kernels) -----------------------------------------------
# variant 0: automatic (full-window cluster when the fast path holds the whole window, else bands),
# 1: band kernel (k_two_opt_cluster), 2: full-window kernel (k_two_opt<.., true>) or nothing
CLUSTER_CASES = [
    # n, tours, dtype, cap, forced workgroups per tour (0 = automatic choice), variant
    (64, 3, np.float64, 5000, 2, 0), (100, 5, np.float64, 5000, 4, 0), (200, 2, np.float32, 5000, 8, 0),
    (500, 3, np.float64, 5000, 16, 1), (500, 2, np.float64, 5000, 2, 1), (777, 2, np.float64, 5000, 0, 1),
    (1000, 4, np.float64, 5000, 0, 1), (1000, 1, np.float32, 5000, 0, 1), (1500, 1, np.float64, 300, 32, 1),
    (3000, 2, np.float64, 60, 0, 0), (5000, 1, np.float64, 40, 0, 0),
    # coarse cache entries (K > 1 macro steps per entry: LDS-limited geometries)
    (2000, 2, np.float64, 150, 2, 1), (2500, 1, np.float32, 120, 3, 1), (6000, 1, np.float64, 30, 8, 0),
    # full-window cluster: lane groups interleaved over the workgroups
    (400, 3, np.float64, 5000, 4, 2), (500, 3, np.float64, 5000, 16, 2), (500, 2, np.float32, 5000, 3, 2),
    (777, 2, np.float64, 5000, 0, 0), (1000, 4, np.float64, 5000, 0, 0), (1000, 1, np.float32, 5000, 0, 0),
    (1000, 9, np.float64, 5000, 2, 2), (1500, 1, np.float64, 300, 32, 2), (2000, 2, np.float64, 150, 5, 2),
    (2500, 1, np.float32, 120, 0, 0),
    # band kernel, shared rows, few long row segments: the walked columns used to run past twice the wrap point
    # (found by tests/fuzz/fuzz_two_opt.py at n = 180 .. 192); now the extended window is planned there
    (180, 4, np.float64, 5000, 2, 1), (188, 3, np.float64, 5000, 4, 1), (192, 2, np.float32, 5000, 2, 1),
    (184, 2, np.float64, 5000, 8, 1), (352, 2, np.float64, 5000, 2, 1),
]


@pytest.mark.parametrize("n,tours_n,dtype,cap,force,variant", CLUSTER_CASES)
def test_gpu_cluster_kernel_matches_oracle(amd, oracle, n, tours_n, dtype, cap, force, variant):
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(7000 + n + force)
    pts, tours = _random_batch(rng, tours_n, 1, n, dtype)
    ref_out, ref_it, ref_sw = oracle.solve(pts, tours, max_iterations=cap, threads=8)
    L.tsp2opt_debug_cluster_mode(force if force else -1)
    L.tsp2opt_debug_tune(3, variant)
    try:
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        plan = _lib.cluster_plan(n, _lib.DTYPE_F64 if dtype == np.float64 else _lib.DTYPE_F32, tours_n, cus)
        assert plan["workgroups_per_tour"] >= 2, plan          # the case really runs a cluster kernel
        if variant:
            assert (plan["row_bands"] == 0) == (variant == 2), plan
        out, iters, sweeps, lengths, status = _run(amd, pts, tours, 1, 1, cap)
    finally:
        L.tsp2opt_debug_cluster_mode(-1)
        L.tsp2opt_debug_tune(3, 0)
    assert (status & (15 | 64)).max(initial=0) == 0
    assert np.array_equal(out, ref_out)
    assert np.array_equal(iters, ref_it)
    assert np.array_equal(sweeps, ref_sw)


@pytest.mark.parametrize("variant", [0, 1])
def test_gpu_cluster_kernel_goldens_and_ties(amd, variant):
    """Goldens through the cluster kernels, including the tie-heavy grids (exact sweep path)."""
    from ml4tspbench_amd import _lib
    names = ["g2_n100_indep_f64", "g9_grid12_indep_f64", "g9_grid8_indep_f32", "g9_n120_clustered_f64",
             "g9_n80_offset_scaled_f64", "g7_n500_indep_f32", "g4_n1000_indep_f64", "g5_n10000_cap32_f64",
             "g6_n50_cap7_f64", "g9_n40_duplicates_f64"]
    _lib.lib().tsp2opt_debug_cluster_mode(4)
    _lib.lib().tsp2opt_debug_tune(3, variant)
    try:
        for nm in names:
            g = load_golden(f"{GOLDEN}/{nm}.npz")
            pts = g["points"].astype(np.float64) if g["entry"] == "tsp_2opt" else g["points"]
            out, iters, _, _, status = _run(amd, pts, g["tours_in"], 1, 1, g["max_iterations"])
            assert (status & (15 | 64)).max(initial=0) == 0, nm
            assert np.array_equal(out, g["tours_out"]), nm
            assert np.array_equal(iters, g["iterators"]), nm
    finally:
        _lib.lib().tsp2opt_debug_cluster_mode(-1)
        _lib.lib().tsp2opt_debug_tune(3, 0)


def test_gpu_cluster_time_limit(amd):
    """The wall-clock limit travels through the cluster exchange (workgroup 0 decides for all)."""
    from ml4tspbench_amd import _lib
    rng = np.random.default_rng(66)
    pts, tours = _random_batch(rng, 1, 1, 1200, np.float64)
    dev = torch.device("cuda", 0)
    for variant in (1, 2):
        _lib.lib().tsp2opt_debug_tune(3, variant)
        try:
            res = amd.two_opt_device(torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev),
                                     time_limit=1e-4)
            torch.cuda.synchronize()
        finally:
            _lib.lib().tsp2opt_debug_tune(3, 0)
        st = res.status.cpu().numpy()
        assert (st & _lib.ST_TIME).all() and (res.iterations.cpu().numpy() < 800).all()
        out = res.tours.cpu().numpy()
        assert (np.sort(out[:, :-1], axis=1) == np.arange(1200)).all()


@pytest.mark.parametrize("side,dtype", [(20, np.float64), (23, np.float64), (20, np.float32)])
def test_gpu_tie_heavy_grids_all_kernels(amd, oracle, side, dtype):
    """Regular grids: thousands of exactly tied deltas, so nearly every iteration takes the ambiguous
    path (candidates from the tile cache, list overflow -> exact sweep).  Same tours from the
    one-workgroup kernel, the full-window cluster and the band cluster as from the oracle."""
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    n = side * side
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(n, 2)
    pts = (g / float(side - 1)).astype(dtype)[None]
    rng = np.random.default_rng(500 + side)
    tours = np.zeros((2, n + 1), dtype=np.int64)
    for k in range(2):
        tours[k, 1:n] = rng.permutation(np.arange(1, n))
    ref_out, ref_it, ref_sw = oracle.solve(pts, tours, tours_per_instance=2, max_iterations=5000, threads=2)
    for mode, variant in ((0, 0), (4, 2), (4, 1)):
        L.tsp2opt_debug_cluster_mode(mode)
        L.tsp2opt_debug_tune(3, variant)
        try:
            out, iters, sweeps, _, status = _run(amd, pts, tours, 2, 1, 5000)
        finally:
            L.tsp2opt_debug_cluster_mode(-1)
            L.tsp2opt_debug_tune(3, 0)
        assert (status & (15 | 64)).max(initial=0) == 0, (mode, variant)
        assert np.array_equal(out, ref_out), (mode, variant)
        assert np.array_equal(iters, ref_it) and np.array_equal(sweeps, ref_sw), (mode, variant)


@pytest.mark.parametrize("variant", [1, 2])
def test_gpu_cluster_invalid_inputs_reported(amd, oracle, variant):
    """Several workgroups per tour: every workgroup of a cluster must reach the same verdict on a bad
    tour (index out of range, not closed, NaN coordinate), leave it untouched, and go on to the next."""
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    n = 800
    rng = np.random.default_rng(321)
    pts, tours = _random_batch(rng, 5, 1, n, np.float64)
    bad = tours.copy()
    bad[0, 17] = n + 3
    bad[1, -1] = 9
    p2 = pts.copy()
    p2[2, 5, 1] = np.nan
    L.tsp2opt_debug_cluster_mode(4)
    L.tsp2opt_debug_tune(3, variant)
    try:
        out, iters, sweeps, _, status = _run(amd, p2, bad, 1, 1, 400)
    finally:
        L.tsp2opt_debug_cluster_mode(-1)
        L.tsp2opt_debug_tune(3, 0)
    assert status[0] & _lib.ST_BAD_INDEX and status[1] & _lib.ST_NOT_CLOSED and status[2] & _lib.ST_NONFINITE
    assert np.array_equal(out[:3], bad[:3]) and (iters[:3] == 0).all() and (sweeps[:3] == 0).all()
    ref_out, ref_it, _ = oracle.solve(pts[3:], tours[3:], max_iterations=400, threads=2)
    assert (status[3:] & (15 | 64)).max() == 0
    assert np.array_equal(out[3:], ref_out) and np.array_equal(iters[3:], ref_it)


def test_gpu_profile_history(amd):
    """tsp2opt_profile_history_ms: the HIP-event durations of the last calls stay readable afterwards."""
    import ctypes
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    pts = torch.from_numpy(rng.random((4, 60, 2))).cuda()
    tours = torch.from_numpy(np.stack([np.append(rng.permutation(60), 0) for _ in range(4)]).astype(np.int64))
    tours = torch.where(tours == 0, tours[:, :1], torch.where(tours == tours[:, :1], 0, tours)).cuda()
    L.tsp2opt_profile_enable(1)
    try:
        for _ in range(3):
            amd.two_opt_device(pts, tours)
        torch.cuda.synchronize()
        m, f = ctypes.c_double(), ctypes.c_double()
        for back in range(3):
            assert L.tsp2opt_profile_history_ms(0, back, ctypes.byref(m), ctypes.byref(f)) == 0
            assert 0.0 < m.value < 1000.0
        assert L.tsp2opt_profile_history_ms(0, 64, ctypes.byref(m), ctypes.byref(f)) != 0
        assert L.tsp2opt_profile_last_ms(0, ctypes.byref(m), ctypes.byref(f)) == 0
    finally:
        L.tsp2opt_profile_enable(0)


@pytest.mark.parametrize("lo,hi,step,force,wide", [(4, 200, 1, 0, 1), (201, 700, 13, 0, 1), (18, 330, 8, 3, 1),
                                                   (161, 660, 7, 0, 0)])
def test_gpu_every_n_matches_oracle(amd, oracle, lo, hi, step, force, wide):
    """Geometry boundaries (threads per workgroup, filter class, tile cache on/off, cluster split):
    a grid of N through the automatic kernel choice and through forced clusters, both dtypes.  wide = 0: the
    geometry of large batches (tsp2opt_debug_tune(9, 0): a batch of at most one tour per CU otherwise gets twice
    the threads per tour, and these three-tour batches would never run the geometry the benchmarks run)."""
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    L.tsp2opt_debug_cluster_mode(force if force else -1)
    L.tsp2opt_debug_tune(9, wide)
    bad = []
    try:
        for n in range(lo, hi + 1, step):
            for dtype in (np.float64, np.float32):
                rng = np.random.default_rng(n * 3 + (dtype == np.float32))
                pts, tours = _random_batch(rng, 3, 1, n, dtype)
                ref_out, ref_it, ref_sw = oracle.solve(pts, tours, max_iterations=5000, threads=8)
                out, iters, sweeps, _, status = _run(amd, pts, tours, 1, 1, 5000)
                if not (np.array_equal(out, ref_out) and np.array_equal(iters, ref_it)
                        and np.array_equal(sweeps, ref_sw) and (status & (15 | 64)).max(initial=0) == 0):
                    bad.append((n, dtype.__name__))
    finally:
        L.tsp2opt_debug_cluster_mode(-1)
        L.tsp2opt_debug_tune(9, 1)
    assert not bad, bad


# ---- coupled mode: k_group_lockstep's active path -------------------------------------------------
def _group_goldens():
    return [f for f in golden_files() if load_golden(f)["group_size"] > 1]


@pytest.mark.parametrize("path", _group_goldens(), ids=lambda p: p.split("/")[-1][:-4])
def test_gpu_forced_lockstep_matches_reference_golden(amd, path):
    """tsp2opt_debug_tune(6, 1): the coupled call is replayed by k_group_lockstep from iteration 0
    (phase 1 only validates, no forced continuation), so its sweep / argmin / global-memory reversal /
    cap handling runs on every grouped fixture -- bit-exact against the reference's outputs."""
    from ml4tspbench_amd import _lib
    g = load_golden(path)
    pts = g["points"].astype(np.float64) if g["entry"] == "tsp_2opt" else g["points"]
    _lib.lib().tsp2opt_debug_tune(6, 1)
    try:
        out, iters, sweeps, lengths, status = _run(amd, pts, g["tours_in"], g["tours_per_instance"],
                                                   g["group_size"], g["max_iterations"])
    finally:
        _lib.lib().tsp2opt_debug_tune(6, 0)
    assert (status & (15 | 64)).max(initial=0) == 0
    assert np.array_equal(out, g["tours_out"]), g["name"]
    per_call = iters[::g["group_size"]]
    assert np.array_equal(per_call, g["iterators"]), g["name"]
    capped = per_call >= g["max_iterations"]
    assert np.array_equal(sweeps[::g["group_size"]], np.where(capped, per_call, per_call + 1))
    assert np.array_equal((status & 16) != 0, np.repeat(capped, g["group_size"]))


@pytest.mark.parametrize("n,inst,tpi,scale,dtype,cap,forced", [
    (30, 40, 8, 1.0e-5, np.float64, 5000, 0), (64, 24, 4, 1.0e-4, np.float64, 5000, 0),
    (100, 12, 8, 3.0e-5, np.float32, 5000, 0), (200, 6, 8, 1.0e-4, np.float64, 5000, 0),
    (500, 2, 8, 1.0e-4, np.float64, 5000, 0), (64, 24, 4, 1.0e-4, np.float64, 37, 0),
    (50, 8, 8, 1.0, np.float64, 5000, 1), (127, 3, 2, 1.0, np.float64, 5000, 1),
    (200, 2, 4, 1.0, np.float32, 5000, 1), (300, 1, 4, 1.0e-4, np.float64, 5000, 1),
    (64, 4, 4, 1.0e-4, np.float64, 21, 1)])
def test_gpu_reactivating_groups_match_oracle(amd, oracle, n, inst, tpi, scale, dtype, cap, forced):
    """Small coordinate scales make many real moves gain less than the 1e-6 threshold: converged tours
    keep applying them while their siblings run and are often active again at the group's target, which
    is exactly when k_group_lockstep takes over (natural path, forced = 0) -- and the forced replay."""
    from ml4tspbench_amd import _lib
    rng = np.random.default_rng(4000 + n + tpi)
    pts, tours = _random_batch(rng, inst, tpi, n, np.float64)
    pts = (pts * scale).astype(dtype)
    ref_out, ref_it, ref_sw = oracle.solve(pts, tours, tours_per_instance=tpi, group_size=tpi,
                                           max_iterations=cap, threads=8)
    if scale < 1.0 and cap == 5000:
        _, own_it, _ = oracle.solve(pts, tours, tours_per_instance=tpi, group_size=1, threads=8)
        assert (ref_it.reshape(-1, tpi)[:, 0] > own_it.reshape(-1, tpi).max(axis=1)).any(), \
            "case does not re-activate: it would not reach the lock-step kernel"
    _lib.lib().tsp2opt_debug_tune(6, forced)
    try:
        out, iters, sweeps, _, status = _run(amd, pts, tours, tpi, tpi, cap)
    finally:
        _lib.lib().tsp2opt_debug_tune(6, 0)
    assert (status & (15 | 64)).max(initial=0) == 0
    assert np.array_equal(out, ref_out)
    assert np.array_equal(iters, ref_it) and np.array_equal(sweeps, ref_sw)


def test_gpu_forced_lockstep_reports_invalid_tours(amd):
    from ml4tspbench_amd import _lib
    rng = np.random.default_rng(77)
    pts, tours = _random_batch(rng, 2, 4, 40, np.float64)
    bad = tours.copy()
    bad[1, 5] = 99
    _lib.lib().tsp2opt_debug_tune(6, 1)
    try:
        out, iters, sweeps, _, status = _run(amd, pts, bad, 4, 4, 5000)
    finally:
        _lib.lib().tsp2opt_debug_tune(6, 0)
    assert status[1] & _lib.ST_BAD_INDEX and np.array_equal(out[:4], bad[:4])     # whole group untouched
    assert (status[4:] & 15).max() == 0 and (iters[4:] > 0).all()


# ---- co-residency of cluster kernels ----------------------------------------------------------------
@pytest.mark.parametrize("n,variant", [(1000, 0), (3000, 0), (1000, 1)])
def test_gpu_two_cluster_solves_on_two_streams(amd, oracle, n, variant):
    """Two cluster-kernel solves issued back to back on two streams: each needs all its workgroups
    resident at once, so the library chains cluster launches of one device through an event.  Both
    results must be bit-exact (no barrier timeout, no TSP2OPT_ST_INTERNAL)."""
    from ml4tspbench_amd import _lib
    cap = 400 if n <= 1000 else 40
    rng = np.random.default_rng(31 + n)
    dev = torch.device("cuda", 0)
    batches = [_random_batch(rng, 3, 1, n, np.float64) for _ in range(2)]
    refs = [oracle.solve(p, t, max_iterations=cap, threads=8) for p, t in batches]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    _lib.lib().tsp2opt_debug_tune(3, variant)
    try:
        assert _lib.cluster_plan(n, _lib.DTYPE_F64, 3, cus)["workgroups_per_tour"] >= 2
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        d_in = [(torch.from_numpy(p).to(dev), torch.from_numpy(t).to(dev)) for p, t in batches]
        torch.cuda.synchronize()
        res = []
        for rep in range(3):                         # interleaved: A B A B A B, no host synchronisation
            for s, (dp, dt) in zip(streams, d_in):
                with torch.cuda.stream(s):
                    r = amd.two_opt_device(dp, dt, max_iterations=cap, want_lengths=False)
                if rep == 2:
                    res.append(r)
        torch.cuda.synchronize()
    finally:
        _lib.lib().tsp2opt_debug_tune(3, 0)
    for r, (ref_out, ref_it, _) in zip(res, refs):
        st = r.status.cpu().numpy()
        assert (st & (15 | _lib.ST_INTERNAL)).max(initial=0) == 0
        assert np.array_equal(r.tours.cpu().numpy(), ref_out)
        assert np.array_equal(r.iterations.cpu().numpy(), ref_it)


def test_gpu_single_workgroup_geometry_boundaries(amd, oracle):
    """One workgroup per tour (cluster choice off) across the filter-class / thread-count switches above
    N = 700: class 256 compiled for 512 threads, the re-plan near N = 1281..1340, class 512 beyond."""
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    ns = list(range(701, 1400, 37)) + [1279, 1280, 1281, 1290, 1336, 1340, 1341, 1345, 1500, 1900, 2600]
    L.tsp2opt_debug_cluster_mode(0)
    bad = []
    try:
        for n in ns:
            for dtype in (np.float64, np.float32):
                rng = np.random.default_rng(n * 5 + (dtype == np.float32))
                pts, tours = _random_batch(rng, 2, 1, n, dtype)
                cap = 120
                ref_out, ref_it, ref_sw = oracle.solve(pts, tours, max_iterations=cap, threads=8)
                out, iters, sweeps, _, status = _run(amd, pts, tours, 1, 1, cap, lengths=False)
                if not (np.array_equal(out, ref_out) and np.array_equal(iters, ref_it)
                        and np.array_equal(sweeps, ref_sw) and (status & (15 | 64)).max(initial=0) == 0):
                    bad.append((n, dtype.__name__))
    finally:
        L.tsp2opt_debug_cluster_mode(-1)
        L.tsp2opt_debug_tune(9, 1)
    assert not bad, bad
