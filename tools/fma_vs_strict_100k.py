#!/usr/bin/env python3
"""Which arithmetic carries north_star's acceptance contract at the reference's run length?

The device-resident Adaptive-Metropolis sampler on the headline problem (BASELINE configs[1]: 4096 chains, Dopri5,
400 days) at the reference's own settings (data/configuration/mcmc_settings.txt: 100 000 iterations, burn-in 5 000,
adaptation_period 100, thinning 100), once in `fma` arithmetic and once in `strict`, same seeds, accept traces kept.
A chain whose accept decision differs ONCE is a different chain from there on (state, stream position and covariance
all follow the decision), so the figure that matters is the number of chains whose traces differ anywhere and the
iteration of each chain's FIRST differing decision; decisions after it are not comparable and are reported only as a
raw count.  The rule this run settles: any flip => `strict` is bench.py's `value` and the default of the
drop-in constructors; none => `fma` is.

    python tools/fma_vs_strict_100k.py --out gpurun_out/r04_fma_vs_strict_100k.json
"""
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
    ap.add_argument("--chains", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=100000)
    ap.add_argument("--burn-in", type=int, default=5000)
    ap.add_argument("--adaptation-period", type=int, default=100)
    ap.add_argument("--thinning", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--workload", default="c1")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    mm = mmid_amd_loader.load()
    pb = mm.workloads.build(args.workload, os.path.join(ROOT, "tests", "golden"))
    C, n = args.chains, args.iterations
    x0 = mm.draws.jitter_draws(pb, 1, C)
    runs = {}
    for name, arith in (("fma", mm.ARITH_FMA), ("strict", mm.ARITH_STRICT)):
        host = mm.HostObjective(pb.with_(arith=arith, constraint_mode=mm.CONSTRAINT_REFLECT))
        host.metropolis_hastings(x0[:16], args.seed, 4, 1, device_state=True)
        t0 = time.perf_counter()
        r = host.metropolis_hastings(x0, args.seed, n, min(args.burn_in, n // 3), adaptation_period=args.adaptation_period,
                                     thinning=args.thinning, device_state=True, want_trace=True)
        wall = time.perf_counter() - t0
        runs[name] = r
        print(json.dumps({"arith": name, "loop_seconds": r["loop_seconds"], "wall_seconds": wall,
                          "ms_per_iteration": r["loop_seconds"] / (n - 1) * 1e3,
                          "acceptance": float(r["accepted"].mean() / (n - 1))}), flush=True)
        del host
    a, b = runs["fma"]["accept_trace"], runs["strict"]["accept_trace"]
    diff = a != b
    per_chain = diff.any(axis=1)
    first = np.where(per_chain, diff.argmax(axis=1) + 1, 0)  # iteration t of the first differing accept test (1-based)
    chains_hit = np.flatnonzero(per_chain)
    decisions = int(diff.size)
    # decisions made while the two runs of a chain were still the same chain: everything up to and including the first flip
    comparable = int(np.where(per_chain, first, n - 1).sum())
    out = {
        "workload": args.workload, "chains": C, "iterations": n, "burn_in": min(args.burn_in, n // 3),
        "adaptation_period": args.adaptation_period, "thinning": args.thinning, "seed": args.seed,
        "decisions_per_run": decisions,
        "decisions_compared_on_identical_chains": comparable,
        "chains_with_a_flip": int(per_chain.sum()),
        "flips_per_comparable_decision": (float(per_chain.sum()) / comparable) if comparable else None,
        "first_flip_iteration_by_chain": {int(c): int(first[c]) for c in chains_hit[:256]},
        "first_flip_iteration_quartiles": [int(q) for q in np.percentile(first[per_chain], [0, 25, 50, 75, 100])] if per_chain.any() else None,
        "raw_trace_mismatches": int(diff.sum()),
        "accepted_counts_equal_on_unflipped_chains": bool(np.array_equal(runs["fma"]["accepted"][~per_chain], runs["strict"]["accepted"][~per_chain])),
        "samples_max_abs_diff_on_unflipped_chains": float(np.abs(runs["fma"]["samples"][~per_chain] - runs["strict"]["samples"][~per_chain]).max()) if (~per_chain).any() else None,
        "ms_per_iteration": {k: runs[k]["loop_seconds"] / (n - 1) * 1e3 for k in runs},
        "proposals_per_s": {k: C * (n - 1) / runs[k]["loop_seconds"] for k in runs},
        "acceptance": {k: float(runs[k]["accepted"].mean() / (n - 1)) for k in runs},
        "rule": "any flip => strict is bench.py's value and the constructors' default; none => fma is",
        "verdict": "strict" if per_chain.any() else "fma",
    }
    line = json.dumps(out)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(json.dumps(out, indent=1) + "\n")


if __name__ == "__main__":
    main()
