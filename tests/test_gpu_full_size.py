"""GPU property tests at BASELINE.json's full sizes (run with -m gpu on an MI355X).

The oracle cannot solve these batches in seconds, so the checks are the size-independent properties
of a best-improvement 2-opt: every result is a closed permutation, no tour gets longer, converged
tours are fixed points (a second call does 0 iterations and returns the same array), the run is
deterministic, a group shares one iteration count, and a sample of results is 2-optimal according
to the CPU oracle (which must also do 0 iterations on them).
"""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

#            config      mode           cap    oracle sample (tours)
FULL = [("tsp50", "independent", 5000, 8), ("tsp100", "independent", 5000, 8),
        ("tsp500", "independent", 5000, 4), ("tsp500", "grouped", 5000, 0),
        ("tsp1000", "independent", 5000, 2), ("tsp10000", "independent", 60, 0)]


@pytest.mark.parametrize("config,mode,cap,sample", FULL, ids=[f"{c}-{m}" for c, m, _, _ in FULL])
def test_full_size_properties(oracle, config, mode, cap, sample):
    import ml4tspbench_amd as amd
    from ml4tspbench_amd import _lib
    pts, tours, tpi = bench.make_batch(config, 1234, np.float64)
    n = pts.shape[1]
    group = tpi if mode == "grouped" else 1
    dev = torch.device("cuda", 0)
    d_pts, d_tours = torch.from_numpy(pts).to(dev), torch.from_numpy(tours).to(dev)
    start = amd.evaluate_tours(d_pts, d_tours, tours_per_instance=tpi)
    res = amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi, group_size=group, max_iterations=cap)
    ev = amd.evaluate_tours(d_pts, res.tours, tours_per_instance=tpi)
    again = amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi, group_size=group, max_iterations=cap)
    torch.cuda.synchronize()
    status = res.status.cpu().numpy()
    iters = res.iterations.cpu().numpy()
    assert (status & (15 | _lib.ST_INTERNAL)).max(initial=0) == 0
    assert torch.equal(d_tours.cpu(), torch.from_numpy(tours))                  # inputs untouched
    assert ((ev.valid.cpu().numpy() & 7) == 7).all()                           # closed permutations ending at 0
    assert (ev.lengths <= start.lengths + 1e-9).all()                          # never longer
    np.testing.assert_allclose(res.lengths.cpu().numpy(), ev.lengths.cpu().numpy(), rtol=1e-12)
    assert torch.equal(res.tours, again.tours) and torch.equal(res.iterations, again.iterations)   # deterministic
    capped = (status & _lib.ST_CAPPED) != 0
    assert (iters[capped] == cap).all() and (iters[~capped] < cap).all()
    sweeps = res.sweeps.cpu().numpy()
    assert np.array_equal(sweeps, np.where(capped, iters, iters + 1))
    if group > 1:
        assert (iters.reshape(-1, group) == iters.reshape(-1, group)[:, :1]).all()
    if config == "tsp10000":
        assert capped.all()                                                     # 60 iterations do not converge
        return
    assert not capped.any()
    # fixed point: an independent second pass over the results changes nothing
    fix = amd.two_opt_device(d_pts, res.tours, tours_per_instance=tpi, group_size=1, max_iterations=cap)
    torch.cuda.synchronize()
    if group == 1:
        assert torch.equal(fix.tours, res.tours) and int(fix.iterations.max().item()) == 0
    else:
        # coupled results may sit one fp-noise move away from their fixed point: lengths must agree
        np.testing.assert_allclose(fix.lengths.cpu().numpy(), res.lengths.cpu().numpy(), rtol=1e-12)
    if sample:
        out = res.tours[:sample].cpu().numpy()
        ref, ref_it, _ = oracle.solve(pts[: (sample + tpi - 1) // tpi], out, tours_per_instance=tpi,
                                      max_iterations=cap, threads=8)
        assert (ref_it == 0).all() and np.array_equal(ref, out)                # 2-optimal for the oracle too
