#!/usr/bin/env python3
"""bench.py -- throughput of the batched 2-opt hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config tsp500|tsp100|tsp1000|tsp50|tsp10000]
                    [--mode independent|grouped] [--dtype f64|f32] [--no-cpu] [--no-configs]

One "step" = one pass of the hot path over one batch of synthetic input: ALL tours of the batch are
driven to their 2-opt local optimum by one tsp2opt_solve call (default workload: BASELINE.json
configs[2], TSP-500 uniform, 128 instances x 16 tours = 2048 tours per GPU, random start tours as in
ml4tsp/nar/decoder/random.py:27-29, max_iterations 5000, plugin (independent) termination).
Inputs are resident in HBM before the timed region.  Multi-GPU: one process per GPU (torch
distributed, RCCL), every rank solves its own batch of the same shape (weak scaling, no data-path
collective) and the timed region ends with one all_gather of tours + lengths.

metric: pair-deltas/s = sum over tours of sweeps * (N-1)(N-2)/2 / wall time, sweeps = iterator+1
when converged else the cap.  Prints ONE JSON line on rank 0.

The default invocation (no --config) times the headline exactly as above (W warm-up steps, K timed
steps, top-level keys) and then, outside that timed region, adds to the SAME line:
  "configs": the other four BASELINE.json configurations, 3 timed steps each, with their own roofline
             objects, plus two more start-tour variants of configs[2] (heat-map-weighted sampled starts,
             ml4tsp/nar/decoder/sampling.py:21-43, and greedy-decoded starts) with tours/s;
  "strong":  (--gpus N > 1, or under torch.distributed.run) BASELINE.json configs[2] as it is worded --
             ONE batch of 128 instances x 16 tours sharded over the N ranks by sharding.solve_sharded,
             final all_gather included;
  "cpu_baseline": the CPU port on the box's cores (N = 1 only), with a "single_core" leg.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (N, instances, tours per instance)
    "tsp50": (50, 1280, 1),
    "tsp100": (100, 10000, 1),
    "tsp500": (500, 128, 16),
    "tsp1000": (1000, 128, 1),
    "tsp10000": (10000, 16, 1),
}
FLOP_PER_PAIR = 15.0          # 2 x (2 sub + 2 mul + add + sqrt) + 3 add/sub
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk/CU
PEAK_FP64_VALU_TFLOPS = 78.6   # SURVEY.md 8(d3) / BASELINE.md: the fp64 vector peak
# DESIGN.md 4.2 (tools/valu_chain_microbench.hip): SIMD cycles per 64 evaluated pairs of the fully packed
# per-pair chain (1 v_sqrt_f32, packed sub / mul / fma / add, min) -- the instruction floor of the sweep
INSN_FLOOR_CYCLES_PER_64_PAIRS = 25.4
SIMDS, CLOCK_HZ = 1024, 2.4e9


def lib_sha256() -> str:
    import hashlib
    from ml4tspbench_amd import _lib
    with open(_lib.LIB_PATH, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def pmc_for_this_build(config: str, mode: str, dtype: str):
    """Counter-derived figures of the headline kernel (rocprofv3 --pmc cannot run in-process): the
    newest committed profiles/*_tsp500_pmc_summary.json whose `lib_sha256` is THIS library's, else None
    -- a stale number is never quoted for a changed kernel."""
    if not (config == "tsp500" and mode == "independent" and dtype == "f64"):
        return None, "PMC passes exist for the default workload only"
    import glob
    sha = lib_sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_tsp500_pmc_summary.json")), reverse=True):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except Exception:
            continue
        if d.get("lib_sha256") == sha:
            d["_file"] = os.path.relpath(path, ROOT)
            return d, None
    return None, f"no PMC summary under profiles/ was measured on this build (lib sha256 {sha[:12]})"


def make_batch(name: str, seed: int, dtype):
    n, inst, tpi = CONFIGS[name]
    rng = np.random.default_rng(seed)
    pts = rng.random((inst, n, 2))                    # generate.py:22-23 uniform U[0,1)^2
    tours = np.zeros((inst * tpi, n + 1), dtype=np.int64)
    body = np.arange(1, n)
    for k in range(inst * tpi):
        tours[k, 1:n] = rng.permutation(body)         # decoder/random.py:27-29: 0, perm, 0
    return pts.astype(dtype), tours, tpi


def make_heat(pts: torch.Tensor, per: int, knn: int, seed: int) -> torch.Tensor:
    """Synthetic heat maps for greedy-decoded start tours (no checkpoints exist offline): (I,N,2) float64
    on the device -> (I*per, N, N) float32, exp(-d_ij / tau) times per-sample multiplicative noise, i.e.
    "nearest edges are hot", which decodes to a reasonable but unoptimised tour -- the kind of start 2-opt
    sees in the reference (BASELINE.json configs[0..3]).  knn > 0 zeroes everything outside each node's
    knn nearest neighbours (configs[3]'s sparse k=50 heat map, densified as greedy_search.py:41-47 does)."""
    inst, n, _ = pts.shape
    g = torch.Generator(device=pts.device)
    g.manual_seed(seed)
    out = torch.empty((inst * per, n, n), dtype=torch.float32, device=pts.device)
    tau = 0.35 / np.sqrt(n)
    step = max(1, (1 << 26) // (n * n * per))
    for a in range(0, inst, step):
        p = pts[a:a + step]
        d = torch.cdist(p, p)                                    # (c,N,N) float64
        h = torch.exp(-d / tau).to(torch.float32)
        if knn:
            kth = d.topk(knn + 1, dim=-1, largest=False).values[..., -1:]
            h = torch.where(d <= kth, h, torch.zeros_like(h))
        h = h.repeat_interleave(per, dim=0)
        noise = 0.6 + 0.8 * torch.rand(h.shape, generator=g, device=pts.device, dtype=torch.float32)
        h = h * noise
        h.diagonal(dim1=-2, dim2=-1).zero_()
        out[a * per:(a + step) * per] = h
    return out


def sampled_start_tours(heat: torch.Tensor, seed: int) -> torch.Tensor:
    """Heat-map-weighted sequential sampling of start tours, batched on the device: the procedure of
    ml4tsp/nar/decoder/sampling.py:21-43 (start at node 0; the next node is drawn with probability
    proportional to the current node's heat-map row over the nodes not yet visited).  heat (B, N, N) ->
    (B, N+1) int64.  Input generator of the bench, not part of the timed path."""
    B, n, _ = heat.shape
    g = torch.Generator(device=heat.device)
    g.manual_seed(seed)
    tours = torch.zeros((B, n + 1), dtype=torch.int64, device=heat.device)
    open_ = torch.ones((B, n), dtype=torch.float32, device=heat.device)
    open_[:, 0] = 0
    cur = torch.zeros(B, dtype=torch.int64, device=heat.device)
    rows = torch.arange(B, device=heat.device)
    for k in range(1, n):
        w = heat[rows, cur] * open_ + 1e-30 * open_            # a row without heat left: uniform over the open nodes
        cur = torch.multinomial(w, 1, generator=g).squeeze(1)
        tours[:, k] = cur
        open_[rows, cur] = 0
    return tours


def cpu_model() -> str:
    """Model string of the host CPU as the OS reports it, with family / model / stepping (virtualised
    boxes often mask the SKU in `model name`; the numeric ids still identify the part)."""
    name, fam, model, step = "unknown", "", "", ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "model name" and name == "unknown":
                    name = val
                elif key == "cpu family" and not fam:
                    fam = val
                elif key == "model" and not model:
                    model = val
                elif key == "stepping" and not step:
                    step = val
                elif key == "processor" and val != "0":
                    break
    except OSError:
        pass
    ids = f"family {fam} model {model} stepping {step}" if fam else ""
    # AMD family 25 = Zen 3 / Zen 4 (model 1: EPYC 7003 Milan, 17: EPYC 9004 Genoa, 160: 97x4 Bergamo),
    # family 26 = Zen 5 (model 2: EPYC 9005 Turin, 17: Turin Dense)
    part = {("25", "1"): "Zen 3, EPYC 7003 'Milan'", ("25", "17"): "Zen 4, EPYC 9004 'Genoa'",
            ("25", "160"): "Zen 4c, EPYC 97x4 'Bergamo'", ("26", "2"): "Zen 5, EPYC 9005 'Turin'",
            ("26", "17"): "Zen 5c, EPYC 9005 'Turin Dense'"}.get((fam, model))
    if part and "AMD" in name:
        ids += f": {part}"
    return f"{name} ({ids})" if ids else name


def cpu_baseline(name: str, pts: np.ndarray, tours: np.ndarray, tpi: int, group: int, cap: int,
                 max_threads: int):
    """The oracle (oracle/two_opt_oracle.c, a port of the reference algorithm) timed on this box's
    host cores on a bounded sample of the same workload (about 20 s of CPU work): the first `want`
    tours, and for sizes where even one tour per core exceeds the budget (TSP-10000) also a lower
    iteration cap.  BASELINE.md 2 / SURVEY 8 d4 ask for two legs: OpenMP over tours on the box's cores
    (the dict itself) and one core (`single_core`, about 2 s of the same sample).
    Returns the baseline dict, the oracle's tours/iterations and the cap it ran with."""
    from oracle import two_opt_oracle as orc
    n = pts.shape[1]
    # default 16 = the CPU share of a one-GPU box in this pool (its os.cpu_count() shows the whole machine)
    threads = max(1, min(max_threads, os.cpu_count() or 1, orc.max_threads()))
    pairs = (n - 1) * (n - 2) // 2
    budget = 20.0 * 3.0e8 * threads                  # ~3e8 pair evaluations per core-second
    est_sweeps = min(cap, int(n * 1.3) + 1)
    want = int(max(threads, min(len(tours), budget / (pairs * est_sweeps))))
    want = max(tpi, (want // tpi) * tpi)
    want = min(want, len(tours))
    cpu_cap = cap
    rounds = (want + threads - 1) // threads         # tours each core solves one after another
    if pairs * est_sweeps * rounds * threads > 1.5 * budget:
        cpu_cap = max(1, int(budget / (pairs * rounds * threads)))
    sub_t = tours[:want]
    sub_p = pts[: (want + tpi - 1) // tpi]
    t0 = time.perf_counter()
    out, iters, sweeps = orc.solve(sub_p, sub_t, tours_per_instance=tpi, group_size=group,
                                   max_iterations=cpu_cap, threads=threads)
    dt = time.perf_counter() - t0
    capnote = "" if cpu_cap == cap else f", iteration cap lowered to {cpu_cap} for the sample"
    # single core: whole groups from the front of the same sample, about 2 s of work
    one = int(max(1, min(want, 2.0 * 5.0e8 / (pairs * min(cpu_cap, est_sweeps)))))
    one = min(want, max(group, (one // group) * group))
    t1 = time.perf_counter()
    out1, it1, sw1 = orc.solve(pts[: (one + tpi - 1) // tpi], tours[:one], tours_per_instance=tpi,
                               group_size=group, max_iterations=cpu_cap, threads=1)
    dt1 = time.perf_counter() - t1
    single = {"value": float(sw1.astype(np.int64).sum()) * pairs / dt1, "unit": "pair-deltas/s", "cores": 1,
              "sample": f"first {one} tours of the same batch{capnote}, {dt1:.1f} s",
              "tours_per_s": one / dt1 if cpu_cap == cap else None,
              "matches_threaded_run": bool(np.array_equal(out1, out[:one]) and np.array_equal(it1, iters[:one]))}
    return {
        "value": float(sweeps.astype(np.int64).sum()) * pairs / dt,
        "unit": "pair-deltas/s",
        "cores": threads,
        "cores_total": os.cpu_count(),
        "kind": "port",
        "sample": f"first {want} of {len(tours)} tours of the same batch{capnote}, {dt:.1f} s, "
                  f"{want / dt:.2f} tours/s, OpenMP over tours",
        "tours_per_s": want / dt if cpu_cap == cap else None,
        "single_core": single,
    }, out, iters, cpu_cap


def dense_torch_baseline(pts: np.ndarray, tours: np.ndarray, tpi: int, budget_s: float = 4.0):
    """The reference's COST STRUCTURE on the host (reported baseline only): a dense
    torch formulation of the same best-improvement step -- all N x N deltas of a (P, N+1) batch as
    tensors, argmin, host read-back, reversal -- written from the algorithm's definition,
    fp64, torch CPU threads.  Timed per iteration on the first instance's tours for a bounded time."""
    n = pts.shape[1]
    P = torch.from_numpy(pts[0])
    t = torch.from_numpy(tours[:tpi].copy())
    torch_threads = torch.get_num_threads()
    it, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and it < 200:
        a, b = P[t[:, :-1]], P[t[:, 1:]]                            # (B, N, 2) edge tails and heads
        d_aa = torch.cdist(a, a)                                    # D(t_i, t_j)
        d_bb = torch.cdist(b, b)                                    # D(t_i+1, t_j+1)
        e = (a - b).norm(dim=-1)                                    # D(t_i, t_i+1)
        delta = torch.triu(d_aa + d_bb - e[:, :, None] - e[:, None, :], diagonal=2)
        flat = delta.reshape(len(t), -1)
        best, arg = flat.min(dim=1)
        if float(best.min()) >= -1e-6:
            break
        for k in range(len(t)):
            i, j = divmod(int(arg[k]), n)
            if j > i:
                t[k, i + 1:j + 1] = t[k, i + 1:j + 1].flip(0)
        it += 1
    dt = time.perf_counter() - t0
    pairs = (n - 1) * (n - 2) // 2
    return {"value": it * len(t) * pairs / dt if it else None, "unit": "pair-deltas/s",
            "kind": "dense-torch formulation (cost structure of the reference, not its code)",
            "torch_threads": torch_threads,
            "sample": f"{len(t)} tours of one instance, {it} iterations, {dt:.1f} s"}


class Harness:
    """Timing of one batch shape: W untimed + K timed steps back to back, bracketed by a device
    synchronisation (and a barrier across ranks), MAX over ranks, HIP-event kernel times read afterwards."""

    def __init__(self, dev, local_rank, dist, world, rank):
        self.dev, self.local_rank, self.dist, self.world, self.rank = dev, local_rank, dist, world, rank
        from ml4tspbench_amd import _lib
        self.L = _lib.lib()

    def fence(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            torch.cuda.synchronize()

    def time_steps(self, step, steps: int, warmup: int):
        res = None
        for _ in range(warmup):
            res = step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):      # back to back, no host synchronisation inside the timed region
            res = step()
        self.fence()
        dt = time.perf_counter() - t0
        kern_ms = []
        for back in range(min(steps, 64)):      # HIP events of the timed steps, read afterwards
            m, f = ctypes.c_double(), ctypes.c_double()
            if self.L.tsp2opt_profile_history_ms(self.local_rank, back, ctypes.byref(m), ctypes.byref(f)) == 0:
                kern_ms.append(m.value + f.value)
        if self.dist is not None:
            tdt = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(tdt, op=self.dist.ReduceOp.MAX)
            dt = float(tdt.item())
        return res, dt, (float(np.mean(kern_ms)) if kern_ms else None)

    def totals(self, sweeps: torch.Tensor, iters: torch.Tensor, pairs: int, local_T: int):
        """Whole-job sums over the ranks: pair-deltas per step, tours, iterations (sum, max)."""
        v = torch.tensor([float(sweeps.to(torch.int64).sum().item()) * pairs, float(local_T),
                          float(iters.sum().item()), float(iters.max().item()) if local_T else 0.0],
                         dtype=torch.float64, device=self.dev)
        if self.dist is not None:
            mx = v[3:].clone()
            self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
            self.dist.all_reduce(mx, op=self.dist.ReduceOp.MAX)
            v[3] = mx[0]
        return [float(x) for x in v.tolist()]


def kernel_of(n: int, T: int, dtype: str, device: int = 0) -> str:
    """Name of the kernel variant the library picks for this shape (tsp2opt_plan / tsp2opt_cluster_plan)."""
    from ml4tspbench_amd import _lib
    code = _lib.DTYPE_F64 if dtype == "f64" else _lib.DTYPE_F32
    cp = _lib.cluster_plan(n, code, T)
    if cp["workgroups_per_tour"] >= 2:
        kind = "full-window cluster" if cp["row_bands"] == 0 else "band cluster"
        return f"{'k_two_opt' if cp['row_bands'] == 0 else 'k_two_opt_cluster'} ({cp['workgroups_per_tour']} workgroups of {cp['threads']} threads per tour, {kind})"
    p = _lib.plan(n, code, device)
    return f"k_two_opt (one workgroup of {p['threads']} threads per tour, {p['workgroups_per_cu']} per CU)"


def simple_roofline(pair_deltas_rank0: float, kernel_ms, kernel: str):
    if not kernel_ms:
        return None
    ach = pair_deltas_rank0 * FLOP_PER_PAIR / (kernel_ms * 1e-3) / 1e12
    return {"bound": "valu", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_FP32_VALU_TFLOPS, "frac_fp64_peak": ach / PEAK_FP64_VALU_TFLOPS,
            "kernel": kernel, "kernel_ms": kernel_ms, "traffic": None,
            "note": "reference-equivalent work (see the headline's roofline.note); no PMC pass for this shape"}


def run_extra_config(h: Harness, name: str, steps: int, seed: int, cap: int, start: str = "random"):
    """One of the other BASELINE.json configurations (or another start-tour variant of configs[2]):
    1 warm-up + `steps` timed steps, weak scaling when several ranks run it.  Returns the sub-object."""
    import ml4tspbench_amd as amd
    n, inst, tpi = CONFIGS[name]
    pairs = (n - 1) * (n - 2) // 2
    pts, tours, tpi = make_batch(name, seed + h.rank, np.float64)
    d_pts = torch.from_numpy(pts).to(h.dev)
    extra = {}
    if start == "random":
        d_tours = torch.from_numpy(tours).to(h.dev)
        what = "random start tours"
    else:
        heat = make_heat(d_pts, tpi, 0, seed + h.rank)
        if start == "sampled":
            d_tours = sampled_start_tours(heat, seed + h.rank)
            what = "heat-map-weighted sampled start tours (ml4tsp/nar/decoder/sampling.py:21-43, synthetic heat maps)"
        else:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            amd.greedy_decode_device(heat, d_pts, maps_per_instance=tpi)
            ev[0].record()
            d_tours = amd.greedy_decode_device(heat, d_pts, maps_per_instance=tpi)
            ev[1].record()
            torch.cuda.synchronize()
            extra["greedy_decode_ms"] = ev[0].elapsed_time(ev[1])
            what = "greedy edge-merge decoded start tours (search/greedy_search.py:19-74, synthetic heat maps)"
        del heat
    T = d_tours.shape[0]

    def step():
        return amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi, group_size=1,
                                  max_iterations=cap, want_lengths=True)

    warm = 2 if n <= 1000 else 1        # short steps (TSP-50 / TSP-100) measure 5-10 % slow right after one warm-up
    if n <= 1000:
        steps = max(steps, 5)
    res, dt, kern_ms = h.time_steps(step, steps, warm)
    status = res.status.cpu().numpy()
    assert (status & (15 | 64)).max(initial=0) == 0, f"{name}: input rejected"
    my_pd = float(res.sweeps.to(torch.int64).sum().item()) * pairs
    pd, tours_total, it_sum, it_max = h.totals(res.sweeps, res.iterations, pairs, T)
    ev = amd.evaluate_tours(d_pts, res.tours, tours_per_instance=tpi)
    ok = bool(((ev.valid & 3) == 3).all().item())
    out = {"name": name if start == "random" else f"{name}_{start}_start",
           "workload": f"TSP-{n} uniform, {inst} instances x {tpi} {what} per GPU, 2-opt to local optimum, "
                       f"max_iterations {cap}, independent termination",
           "steps": steps, "warmup": warm, "ms_per_step": dt / steps * 1e3, "value": pd * steps / dt,
           "unit": "pair-deltas/s", "tours_per_s": tours_total * steps / dt,
           "mean_iterations": it_sum / tours_total, "max_iterations_seen": it_max,
           "all_tours_valid": ok, "mean_length": float(res.lengths.mean().item()),
           "kernel": kernel_of(n, T, "f64", h.local_rank), "kernel_ms": kern_ms,
           "roofline": simple_roofline(my_pd, kern_ms, "k_two_opt" if n < 5000 else "k_two_opt_cluster")}
    out.update(extra)
    return out


def main():
    # stdout carries exactly ONE line, the JSON: everything else that libraries print there (RCCL's
    # version banner at communicator creation, for one) is sent to stderr until the line is due
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="time this configuration only (default: tsp500 as the headline + the others in `configs`)")
    ap.add_argument("--mode", default="independent", choices=["independent", "grouped"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--max-iterations", type=int, default=5000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="headline only: skip the `configs` / `strong` sub-objects")
    ap.add_argument("--config-steps", type=int, default=3, help="timed steps of every entry of `configs`")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads of the CPU baseline's OpenMP leg (default 16 = the CPU share of a one-GPU box)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank solves its own batch; strong: ONE batch (BASELINE.json configs[2]: "
                         "128 instances x 16 tours) sharded over the ranks by sharding.solve_sharded")
    ap.add_argument("--instances", type=int, default=0, help="diagnostic: override the number of instances")
    ap.add_argument("--tune", action="append", default=[], help="diagnostic: tsp2opt_debug_tune what=value")
    ap.add_argument("--cluster-mode", type=int, default=-1,
                    help="diagnostic: -1 automatic, 0 one workgroup per tour, k>=2 up to k workgroups per tour")
    args = ap.parse_args()
    diagnostic = bool(args.tune) or args.cluster_mode != -1 or args.instances > 0
    all_configs = (args.config is None and not args.no_configs and not diagnostic
                   and args.mode == "independent" and args.dtype == "f64" and args.scaling == "weak")
    if args.config is None:
        args.config = "tsp500"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # launched by torch.distributed.run (RANK set): the distributed path runs even at world size 1,
    # so a 1-GPU box rehearses exactly the code the N-GPU launch executes
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ml4tspbench_amd as amd
    from ml4tspbench_amd import _lib
    L = _lib.lib()
    L.tsp2opt_profile_enable(1)
    L.tsp2opt_debug_cluster_mode(args.cluster_mode)
    for kv in args.tune:                             # diagnostic knobs, e.g. --tune 0=1024
        k_, v_ = kv.split("=")
        L.tsp2opt_debug_tune(int(k_), int(v_))
    h = Harness(dev, local_rank, dist if use_dist else None, world, rank)

    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    if args.instances > 0:
        CONFIGS[args.config] = (CONFIGS[args.config][0], args.instances, CONFIGS[args.config][2])
    n, inst, tpi = CONFIGS[args.config]
    strong = args.scaling == "strong"
    # seed 1234 (utils/args.py:27); strong scaling: every rank holds the same batch
    pts, tours, tpi = make_batch(args.config, 1234 + (0 if strong else rank), np_dtype)
    group = tpi if args.mode == "grouped" and tpi > 1 else 1
    T = len(tours)
    pairs = (n - 1) * (n - 2) // 2
    d_pts = torch.from_numpy(pts).to(dev)
    d_tours = torch.from_numpy(tours).to(dev)
    gather_t = gather_l = None
    if use_dist:
        from ml4tspbench_amd.sharding import pack_tours, unpack_tours
        gather_t = torch.empty((world * T, pack_tours(d_tours[:1]).shape[1]), dtype=torch.uint8, device=dev)
        gather_l = torch.empty(world * T, dtype=torch.float64, device=dev)

    def step_strong(p=None, t=None):
        # the tested function is the timed one: contiguous instance blocks per rank, one final all_gather
        from ml4tspbench_amd.sharding import solve_sharded
        return solve_sharded(d_pts if p is None else p, d_tours if t is None else t, tours_per_instance=tpi,
                             group_size=group, max_iterations=args.max_iterations, want_result=True)

    def step():
        if strong:
            return step_strong()
        res = amd.two_opt_device(d_pts, d_tours, tours_per_instance=tpi, group_size=group,
                                 max_iterations=args.max_iterations, want_lengths=True)
        if use_dist:        # the only exchange of the path: final gather of tours (as int16) and lengths
            dist.all_gather_into_tensor(gather_t, pack_tours(res.tours))
            dist.all_gather_into_tensor(gather_l, res.lengths)
            res.all_tours = unpack_tours(gather_t, n + 1)      # (world * T, N+1) int64 on every rank
        return res

    res, dt, kernel_ms = h.time_steps(step, args.steps, args.warmup)

    if strong:
        whole = res                                  # ShardedResult: whole batch on every rank
        res = whole.local                            # this rank's block (TwoOptResult)
        b0, b1 = whole.block
        assert whole.tours.shape[0] == T and torch.equal(whole.tours[b0:b1], res.tours), "final gather corrupted the tours"
        local_T = b1 - b0
    else:
        local_T = T
        if use_dist:        # the gathered copy holds this rank's tours at its own offset
            assert torch.equal(res.all_tours[rank * T:(rank + 1) * T], res.tours), "final gather corrupted the tours"
    sweeps = res.sweeps.to(torch.int64)
    pd_per_step, tours_total, it_sum, it_max = h.totals(res.sweeps, res.iterations, pairs, local_T)
    value = pd_per_step * args.steps / dt
    status = res.status.cpu().numpy()
    assert (status & (15 | 64)).max(initial=0) == 0, "input rejected"

    gap = None
    if args.dtype == "f32" and rank == 0 and not strong:
        # north_star: "the fp32 path must give tour lengths within 1e-6 relative, with the optimality
        # gap reported": the same start tours solved in fp64 (untimed), both lengths summed in double
        r64 = amd.two_opt_device(d_pts.double(), d_tours, tours_per_instance=tpi, group_size=group,
                                 max_iterations=args.max_iterations, want_lengths=True)
        l32 = amd.evaluate_tours(d_pts.double(), res.tours, tours_per_instance=tpi).lengths
        rel = (l32 - r64.lengths) / r64.lengths
        gap = {"mean_rel_length_gap_fp32_vs_fp64": float(rel.mean().item()),
               "max_abs_rel_length_gap": float(rel.abs().max().item()),
               "identical_tours_fraction": float((res.tours == r64.tours).all(dim=1).double().mean().item())}

    # ---- outside the headline's timed region: the other configurations, strong scaling of configs[2] --------
    extra_configs, strong_obj = [], None
    if all_configs:
        for name in ("tsp50", "tsp100", "tsp1000", "tsp10000"):
            extra_configs.append(run_extra_config(h, name, args.config_steps, 1234, args.max_iterations))
        for start in ("sampled", "greedy"):
            extra_configs.append(run_extra_config(h, "tsp500", args.config_steps, 1234, args.max_iterations, start))
        if use_dist:
            # BASELINE.json configs[2] as worded: ONE batch of 128 x 16 tours sharded over the ranks
            sp, st_, _ = make_batch("tsp500", 1234, np.float64)         # the same batch on every rank
            s_pts, s_tours = torch.from_numpy(sp).to(dev), torch.from_numpy(st_).to(dev)
            sres, sdt, skern = h.time_steps(lambda: step_strong(s_pts, s_tours), args.steps, 1)
            b0, b1 = sres.block
            assert torch.equal(sres.tours[b0:b1], sres.local.tours), "final gather corrupted the tours"
            spd, stours, sit, _ = h.totals(sres.local.sweeps, sres.local.iterations, pairs, b1 - b0)
            strong_obj = {"workload": f"TSP-500 uniform, ONE batch of 128 instances x 16 random start tours sharded "
                                      f"over {world} GPU(s) by sharding.solve_sharded (contiguous instance blocks), "
                                      "final all_gather of tours / iterations / lengths inside the timed region",
                          "scaling": "strong", "n_gpus": world, "steps": args.steps, "warmup": 1,
                          "ms_per_step": sdt / args.steps * 1e3, "value": spd * args.steps / sdt,
                          "unit": "pair-deltas/s", "tours_per_s": stours * args.steps / sdt,
                          "tours_per_gpu": b1 - b0, "kernel_ms_rank0": skern}

    out = None
    if rank == 0:
        my_pairs = float(sweeps.sum().item()) * pairs
        roof = None
        if kernel_ms:
            ach = my_pairs * FLOP_PER_PAIR / (kernel_ms * 1e-3) / 1e12
            # counter-derived figures: only for the library they were measured on (else null)
            pmc, why = pmc_for_this_build(args.config, args.mode, args.dtype) if world == 1 else (None, "N = 1 only")
            exe = busy = bank = evald = traffic = cyc64 = None
            if pmc:
                dv = pmc["derived"]
                pick = lambda pre: next((v for k, v in dv.items() if k.startswith(pre)), None)
                exe_flop, busy, bank = pick("fp32_flop_executed"), pick("valu_busy_frac"), pick("lds_bank_conflict_rate")
                traffic, sq = pick("hbm_bytes"), pick("sqrt_lanes")
                exe = exe_flop / (kernel_ms * 1e-3) / 1e12 if exe_flop else None
                evald = sq / my_pairs if sq and my_pairs else None
                # SIMD cycles the whole chip spent per 64 pairs the kernel really evaluated (one v_sqrt_f32
                # wave-instruction = 64 pairs): everything the kernel does -- marking, lists, selection,
                # reversal, idle issue slots -- is charged to the sweep arithmetic it exists for
                cyc64 = kernel_ms * 1e-3 * CLOCK_HZ * SIMDS / (sq / 64.0) if sq else None
            roof = {"bound": "valu", "achieved": ach, "peak": PEAK_FP32_VALU_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_FP32_VALU_TFLOPS,
                    "frac_fp32_peak": ach / PEAK_FP32_VALU_TFLOPS, "frac_fp64_peak": ach / PEAK_FP64_VALU_TFLOPS,
                    "executed_tflops": exe, "executed_frac": exe / PEAK_FP32_VALU_TFLOPS if exe else None,
                    "simd_cycles_per_64_evaluated_pairs": cyc64,
                    "insn_floor_cycles_per_64_pairs": INSN_FLOOR_CYCLES_PER_64_PAIRS,
                    "insn_floor_frac": INSN_FLOOR_CYCLES_PER_64_PAIRS / cyc64 if cyc64 else None,
                    "hardware_fraction": "insn_floor_frac (how close the VALUs run to the measured instruction floor "
                                         "of the evaluated pairs) and executed_frac (executed FP32 FLOP / vector peak) "
                                         "are hardware utilisation; frac is reference-equivalent work",
                    "valu_busy": busy, "lds_bank_conflict_rate": bank, "pairs_evaluated_frac": evald,
                    "traffic": traffic, "pmc_source": pmc["_file"] if pmc else why,
                    "kernel": "k_two_opt", "kernel_ms": kernel_ms,
                    "note": "achieved/frac are REFERENCE-EQUIVALENT work, not executed FLOP: the reference "
                            "algorithm's pair-deltas (sweeps x (N-1)(N-2)/2) x 15 FLOP / HIP-event duration of the "
                            "persistent kernel (rank 0), priced against the FP32 vector peak 157.3 TFLOP/s because "
                            "the kernel computes in fp32 (exact fp64 only for near-ties) although results are "
                            "bit-exact fp64 (dtype); against the fp64 vector peak 78.6 the same figure is "
                            "frac_fp64_peak and may exceed 1.  The kernel re-evaluates only pairs_evaluated_frac of the "
                            "pairs per sweep (tile-minimum cache, reversals folded onto the shorter side of the cycle) "
                            "at one distance per pair: executed_*, insn_floor_frac and valu_busy are what the VALUs "
                            "really did (rocprofv3 PMC passes of this build; a packed instruction counts once, so "
                            "executed_tflops is a lower bound).  HBM traffic is O(N) per tour."}
        if strong:
            work = (f"TSP-{n} uniform, ONE batch of {inst} instances x {tpi} random start tours sharded over "
                    f"{world} GPU(s) by sharding.solve_sharded (contiguous instance blocks)")
        else:
            work = f"TSP-{n} uniform, {inst} instances x {tpi} random start tours per GPU"
        out = {
            "metric": "2-opt pair-deltas/sec", "value": value, "unit": "pair-deltas/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{work}, 2-opt to local optimum, max_iterations "
                                   f"{args.max_iterations}, {args.mode} termination",
                       "tours_per_gpu": local_T if strong else T, "nodes": n, "mode": args.mode,
                       "parallelism": f"tours sharded over {world} GPU(s), final all_gather only"},
            "tours_per_s": tours_total * args.steps / dt,
            "mean_iterations": it_sum / tours_total, "max_iterations_seen": it_max,
            "roofline": roof,
        }
        if gap is not None:
            out["fp32_vs_fp64"] = gap
        if extra_configs:
            out["configs"] = extra_configs
        if strong_obj is not None:
            out["strong"] = strong_obj
        if not args.no_cpu and world == 1:       # the host baseline is timed at N = 1 only
            cb, ref_out, ref_it, cpu_cap = cpu_baseline(args.config, pts, tours, tpi, group,
                                                        args.max_iterations, args.cpu_threads)
            k = len(ref_out)
            chk = res
            if cpu_cap != args.max_iterations:      # untimed GPU run of the sample at the sample's cap
                chk = amd.two_opt_device(d_pts[: (k + tpi - 1) // tpi], d_tours[:k], tours_per_instance=tpi,
                                         group_size=group, max_iterations=cpu_cap, want_lengths=False)
            same = bool(np.array_equal(chk.tours[:k].cpu().numpy(), ref_out)
                        and np.array_equal(chk.iterations[:k].cpu().numpy(), ref_it))
            cb["gpu_matches_oracle_on_sample"] = same
            cb["cpu_model"] = cpu_model()
            out["cpu_baseline"] = cb
            if n <= 1000:
                out["cpu_dense_torch"] = dense_torch_baseline(pts.astype(np.float64), tours, tpi)
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
