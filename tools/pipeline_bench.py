#!/usr/bin/env python3
"""tools/pipeline_bench.py -- decode -> 2-opt -> validate/select, all on one MI355X.

The reference's test-time pipeline for one batch (models/modelzoo/gnn/tsp_gnn.py:190-221): greedy
edge-merge decode of every heat map (CPU, one core), 2-opt on the decoded tours, validity check,
cost of every tour, best of the P samples.  Here the three stages are three library calls on device
tensors (tsp2opt_greedy_decode, tsp2opt_solve, tsp2opt_evaluate); this tool times each with HIP
events on the current stream and checks a sample against the CPU oracles.

Heat maps are synthetic (no checkpoints here): exp(-d_ij / tau) times per-sample multiplicative
noise, i.e. a "nearest edges are hot" map that decodes to a reasonable but unoptimised tour, the
kind of start 2-opt sees in the reference (BASELINE.json configs[0..3]).  --knn K zeroes everything
outside each node's K nearest neighbours (configs[3]'s sparse k=50 heat map, densified as
greedy_search.py:41-47 does).

    python tools/pipeline_bench.py --config tsp100 [--steps 3] [--knn 0] [--check 2]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {            # N, instances, heat maps per instance  (BASELINE.json configs)
    "tsp50": (50, 1280, 1),
    "tsp100": (100, 10000, 1),
    "tsp500": (500, 128, 16),
    "tsp1000": (1000, 128, 1),
    "tsp10000": (10000, 16, 1),
}


from bench import make_heat  # noqa: E402  (shared with tests/test_gpu_full_size.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tsp100", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--knn", type=int, default=0)
    ap.add_argument("--sparse", action="store_true",
                    help="with --knn: also decode from (values, edge_index) lists (one list per instance, one call "
                         "for the batch) and use those tours downstream")
    ap.add_argument("--check", type=int, default=2, help="instances checked against the CPU oracles")
    ap.add_argument("--max-iterations", type=int, default=5000)
    ap.add_argument("--cluster-mode", type=int, default=-1, help="diagnostic, see bench.py")
    ap.add_argument("--phases", action="store_true", help="diagnostic: phase cycles of one extra 2-opt run")
    ap.add_argument("--tune", action="append", default=[], help="diagnostic: tsp2opt_debug_tune what=value")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    dev = torch.device("cuda", 0)
    import ml4tspbench_amd as amd
    from ml4tspbench_amd import _lib
    _lib.lib().tsp2opt_debug_cluster_mode(args.cluster_mode)
    for kv in args.tune:
        k_, v_ = kv.split("=")
        _lib.lib().tsp2opt_debug_tune(int(k_), int(v_))

    n, inst, per = CONFIGS[args.config]
    rng = np.random.default_rng(1234)
    pts_np = rng.random((inst, n, 2))
    pts = torch.from_numpy(pts_np).to(dev)
    heat = make_heat(pts, per, args.knn, 1234)
    lists = None
    if args.sparse:
        if not args.knn:
            raise SystemExit("--sparse needs --knn")
        lists = []
        for k in range(inst):                   # the list form of the same heat maps: listed = nonzero in map 0
            ei = (heat[k * per] != 0).nonzero().T.contiguous()
            lists.append((torch.stack([heat[k * per + m][ei[0], ei[1]] for m in range(per)]).contiguous(), ei))
        if len({int(ei.shape[1]) for _, ei in lists}) != 1:
            raise SystemExit("instances list different numbers of edges")
        sp_vals = torch.cat([v for v, _ in lists]).contiguous()          # (inst*per, E)
        sp_ei = torch.stack([ei for _, ei in lists]).contiguous()        # (inst, 2, E)
    torch.cuda.synchronize()

    def run():
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        if lists is None:
            tours = amd.greedy_decode_device(heat, pts, maps_per_instance=per)
        else:
            tours = amd.greedy_decode_sparse_device(sp_vals, sp_ei, pts, maps_per_instance=per)
        ev[1].record()
        res = amd.two_opt_device(pts, tours, tours_per_instance=per, group_size=1,
                                 max_iterations=args.max_iterations, want_lengths=False)
        ev[2].record()
        sel = amd.evaluate_tours(pts, res.tours, tours_per_instance=per)
        ev[3].record()
        torch.cuda.synchronize()
        return tours, res, sel, [ev[k].elapsed_time(ev[k + 1]) for k in range(3)]

    run()
    times = []
    for _ in range(args.steps):
        tours, res, sel, t = run()
        times.append(t)
    t = np.mean(np.array(times), axis=0)
    start_len = amd.evaluate_tours(pts, tours, tours_per_instance=per)
    valid = sel.valid.cpu().numpy()
    pairs = (n - 1) * (n - 2) // 2
    sweeps = int(res.sweeps.to(torch.int64).sum().item())
    out = {
        "config": args.config, "nodes": n, "instances": inst, "maps_per_instance": per, "knn": args.knn,
        "ms": {"greedy_decode": float(t[0]), "two_opt": float(t[1]), "evaluate": float(t[2]),
               "total": float(t.sum())},
        "instances_per_s": inst / (float(t.sum()) * 1e-3),
        "tours_per_s": inst * per / (float(t.sum()) * 1e-3),
        "two_opt_pair_deltas_per_s": sweeps * pairs / (float(t[1]) * 1e-3),
        "mean_2opt_iterations": float(res.iterations.double().mean().item()),
        "all_valid": bool(((valid & 3) == 3).all()),
        "mean_length_decoded": float(start_len.lengths.mean().item()),
        "mean_length_after_2opt": float(sel.lengths.mean().item()),
        "mean_length_best_of_p": float(sel.best_length.mean().item()),
    }
    if lists is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        dense_tours = amd.greedy_decode_device(heat, pts, maps_per_instance=per)
        ev[1].record()
        torch.cuda.synchronize()
        _, diag = amd.greedy_decode_sparse_device(sp_vals, sp_ei, pts, maps_per_instance=per, return_diagnostics=True)
        out["sparse"] = {"map0_diagnostics": diag, "listed_edges_per_instance": int(lists[0][1].shape[1]),
                         "workspace_bytes": int(_lib.lib().tsp2opt_greedy_sparse_workspace_bytes(
                             inst * per, n, int(lists[0][1].shape[1]), per, _lib.DTYPE_F32)),
                         "dense_decode_ms": float(ev[0].elapsed_time(ev[1])),
                         "dense_key_bytes": int(inst * per * n * n * 8),
                         "same_tours_as_dense_decode": bool(torch.equal(dense_tours, tours))}
    if args.phases:
        from ml4tspbench_amd import two_opt
        L = _lib.lib()
        L.tsp2opt_debug_phase_cycles(1)
        r2 = amd.two_opt_device(pts, tours, tours_per_instance=per, group_size=1,
                                max_iterations=args.max_iterations, want_lengths=False)
        torch.cuda.synchronize()
        L.tsp2opt_debug_phase_cycles(0)
        ws = list(two_opt._workspaces.values())[0]
        st = ws[64:256].cpu().numpy().view(np.uint64)
        sw = float(r2.sweeps.sum().item())
        names = ["sweep", "select", "apply", "stage/exchange", "tour_total", "amb events", "exact-stage runs", "cand cycles",
                 "sweep: mark+compact", "sweep: barrier+changed tiles", "sweep: barrier+cached minima", "sweep: full",
                 "select: wg filter min", "select: fp32 candidates", "select: fp32 reduction", "select: exact stage",
                 "apply: positions+coords", "apply: interior edges", "apply: barrier", "cluster exchange"]
        out["phase_cycles_per_sweep"] = {nm: round(float(x) / sw, 2) for nm, x in zip(names, st) if x}
    if args.check:
        from oracle import greedy_oracle, two_opt_oracle
        c = min(args.check, inst)
        h_np = heat[:c * per].cpu().numpy()
        ok_g = ok_t = True
        for k in range(c):
            ref_t = greedy_oracle.greedy(pts_np[k], h_np[k * per:(k + 1) * per])
            ok_g &= bool(np.array_equal(ref_t, tours[k * per:(k + 1) * per].cpu().numpy()))
            if n > 3000:            # minutes of single-core oracle time per tour: the decoder only
                ok_t = None
                continue
            ref_o, ref_it, _ = two_opt_oracle.solve(pts_np[k:k + 1], ref_t, tours_per_instance=per, group_size=1,
                                                    max_iterations=args.max_iterations, threads=1)
            ok_t &= bool(np.array_equal(ref_o, res.tours[k * per:(k + 1) * per].cpu().numpy()))
        out["sample_matches_oracles"] = {"instances": c, "greedy": ok_g, "two_opt": ok_t}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
