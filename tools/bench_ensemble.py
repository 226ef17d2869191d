#!/usr/bin/env python3
"""Throughput of the posterior-ensemble summaries (sepaihrd_ensemble_quantiles) on one GPU.
Host-pointer entry point: the time includes the theta upload, the integration of S samples from
the fixed initial state (trajectory output on when seroprevalence is requested), the series pass,
the LDS sort of 6*T_pos*n + T segments and the download of the quantiles."""
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
    ap.add_argument("--samples", type=int, nargs="+", default=[1024, 4096, 16384])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--arith", default="fma")
    args = ap.parse_args()
    mm = mmid_amd_loader.load()
    pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden"))
    pb = pb.with_(arith=mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT)
    probs = [0.025, 0.05, 0.5, 0.95, 0.975]
    hip = mm.HipObjective(pb)
    hip.set_initial_state_mode(1)
    for S in args.samples:
        theta = mm.draws.jitter_draws(pb, 1, S)
        for sero, rt in ((False, False), (True, False), (True, True)):
            hip.ensemble_quantiles(theta[:64], probs, want_sero=sero, want_rt=rt)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                r = hip.ensemble_quantiles(theta, probs, want_sero=sero, want_rt=rt)
            dt = (time.perf_counter() - t0) / args.reps
            print(json.dumps({"samples": S, "seroprevalence": sero, "rt": rt, "ms": dt * 1e3, "samples_per_s": S / dt,
                              "n_valid": r["n_valid"], "segments": 6 * r["ppc"].shape[2] * pb.n + (pb.n_times if sero else 0)}))


if __name__ == "__main__":
    main()
