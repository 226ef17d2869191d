#!/usr/bin/env python3
"""The device-resident Adaptive-Metropolis sampler at the reference's own run length
(data/configuration/mcmc_settings.txt: 100 000 iterations, burn-in 5 000, adaptation_period 100, thinning 100) on the
headline problem: ms per iteration over the whole run against the bare evaluation step, peak device memory, and
-- in segments of the run -- that the iteration does not slow down as the chain grows (the covariance refresh is
O(P^2) from running co-moments, not a walk over the history).  Under `rocprofv3 --kernel-trace --stats` the kernel
summary gives the split evaluation / moment catch-up + refresh / Cholesky."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=4096)
    ap.add_argument("--iterations", type=int, nargs="+", default=[100000],
                    help="several lengths: the slope between them is the late-run cost per iteration")
    ap.add_argument("--burn-in", type=int, default=5000)
    ap.add_argument("--adaptation-period", type=int, default=100)
    ap.add_argument("--thinning", type=int, default=100)
    ap.add_argument("--arith", default="fma")
    ap.add_argument("--two-pass", action="store_true", help="the reference's literal refresh over the whole history")
    ap.add_argument("--workload", default="c1")
    ap.add_argument("--host-streams", action="store_true", help="draw the chains' mt19937 streams on the host (rounds 1-2) instead of the device")
    args = ap.parse_args()
    import torch
    mm = mmid_amd_loader.load()
    pb = mm.workloads.build(args.workload, os.path.join(ROOT, "tests", "golden"))
    pb = pb.with_(arith=mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT, constraint_mode=1)
    C = args.chains
    x0 = mm.draws.jitter_draws(pb, 1, C)
    hip = mm.HipObjective(pb)
    d_theta = torch.from_numpy(x0).cuda()
    d_ll = torch.empty(C, dtype=torch.float64, device="cuda")
    for _ in range(5):
        hip.eval_batch_device(d_theta, d_ll, B=C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        hip.eval_batch_device(d_theta, d_ll, B=C)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / 50 * 1e3
    del hip
    host = mm.HostObjective(pb)
    host.metropolis_hastings(x0[:16], 1, 4, 1, device_state=True)
    free0, total = torch.cuda.mem_get_info()
    runs = []
    for n in args.iterations:
        low = [free0]
        stop = threading.Event()

        def watch():
            while not stop.wait(0.25):
                low[0] = min(low[0], torch.cuda.mem_get_info()[0])
        th = threading.Thread(target=watch, daemon=True)
        th.start()
        t0 = time.perf_counter()
        r = host.metropolis_hastings(x0, 1, n, min(args.burn_in, n // 3), adaptation_period=args.adaptation_period,
                                     thinning=args.thinning, device_state=True, two_pass_covariance=args.two_pass,
                                     want_trace=False, device_streams=not args.host_streams)
        wall = time.perf_counter() - t0
        stop.set()
        th.join()
        runs.append({"iterations": n, "loop_seconds": r["loop_seconds"], "wall_seconds": wall,
                     "ms_per_iteration": r["loop_seconds"] / (n - 1) * 1e3,
                     "proposals_per_s": C * (n - 1) / r["loop_seconds"],
                     "vs_bare_step": r["loop_seconds"] / (n - 1) * 1e3 / step_ms,
                     "acceptance": float(r["accepted"].mean() / (n - 1)),
                     "final_scale_median": float(np.median(r["final_scale"])),
                     "samples_per_chain": int(r["samples"].shape[1]),
                     "peak_device_memory_gb": (free0 - low[0]) / 1e9,
                     "best_value_max": float(r["best_value"].max())})
        print(json.dumps(runs[-1]), flush=True)
    out = {"chains": C, "workload": args.workload, "arith": args.arith, "burn_in": args.burn_in,
           "adaptation_period": args.adaptation_period, "thinning": args.thinning,
           "covariance": "two-pass over the whole history" if args.two_pass else "running co-moments",
           "streams": "host (libstdc++)" if args.host_streams else "device (csrc/sepaihrd_rng.inc)",
           "bare_evaluation_step_ms": step_ms, "device_memory_total_gb": total / 1e9, "runs": runs}
    if len(runs) >= 2:
        a, b = runs[-2], runs[-1]
        out["late_ms_per_iteration"] = (b["loop_seconds"] - a["loop_seconds"]) / (b["iterations"] - a["iterations"]) * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
