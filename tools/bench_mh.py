#!/usr/bin/env python3
"""End-to-end throughput of the multi-chain Adaptive Metropolis sampler (C++ host loop + one
device launch per iteration): proposals per second at C chains, next to the bare evaluation rate."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, nargs="+", default=[256, 4096])
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--burn-in", type=int, default=100)
    ap.add_argument("--adaptation-period", type=int, default=100)
    ap.add_argument("--arith", default="fma")
    ap.add_argument("--state", choices=["host", "device", "both"], default="both")
    ap.add_argument("--groups", type=int, nargs="*", default=[], help="also time the device-resident sampler split into G groups")
    args = ap.parse_args()
    mm = mmid_amd_loader.load()
    pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden"))
    pb = pb.with_(arith=mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT, constraint_mode=1)
    for C in args.chains:
        x0 = mm.draws.jitter_draws(pb, 1, C)
        runs = {}
        for state in (["host", "device"] if args.state == "both" else [args.state]):
            host = mm.HostObjective(pb)
            host.metropolis_hastings(x0[:8], 1, 5, 2, device_state=(state == "device"))  # warm-up
            t0 = time.perf_counter()
            r = host.metropolis_hastings(x0, 1, args.iterations, args.burn_in, adaptation_period=args.adaptation_period,
                                         thinning=10, device_state=(state == "device"))
            runs[state] = (time.perf_counter() - t0, r)
        dt, r = runs["device" if "device" in runs else "host"]
        grouped = {}
        for G in args.groups:
            objs = [mm.HostObjective(pb) for _ in range(G)]
            mm.hostabi.metropolis_hastings_groups(objs, x0[:8 * G], 1, 5, 2)
            t0 = time.perf_counter()
            rg = mm.hostabi.metropolis_hastings_groups(objs, x0, 1, args.iterations, args.burn_in,
                                                       adaptation_period=args.adaptation_period, thinning=10)
            dtg = time.perf_counter() - t0
            grouped[G] = {"ms_per_iteration": dtg / (args.iterations - 1) * 1e3, "proposals_per_s": C * (args.iterations - 1) / dtg,
                          "same_accept_traces": bool(np.array_equal(rg["accept_trace"], r["accept_trace"]))}
        hip = mm.HipObjective(pb)
        hip.eval_batch(x0)
        t1 = time.perf_counter()
        for _ in range(10):
            hip.eval_batch(x0)
        ev = (time.perf_counter() - t1) / 10
        print(json.dumps({"chains": C, "iterations": args.iterations, "seconds": dt,
                          "proposals_per_s": C * (args.iterations - 1) / dt, "ms_per_iteration": dt / (args.iterations - 1) * 1e3,
                          "eval_batch_host_pointer_ms": ev * 1e3, "acceptance": float(r["accepted"].mean() / (args.iterations - 1)),
                          "groups": grouped, "ms_per_iteration_by_state": {k: v[0] / (args.iterations - 1) * 1e3 for k, v in runs.items()},
                          "same_accept_traces": (np.array_equal(runs["host"][1]["accept_trace"], runs["device"][1]["accept_trace"])
                                                 if len(runs) == 2 else None)}))


if __name__ == "__main__":
    main()
