#!/usr/bin/env python3
"""BASELINE configs[4]: SEPAIHRD with 16 age groups, 1000 days, Dopri5 -- tolerance sweep
abs = rel in {1e-3, 1e-4, 1e-5, 1e-6}, fp32 state vs fp64 state, one table.
Per tolerance and number type: throughput of the HIP path at --chains chains, steps per evaluation, and the accuracy
of states and log-likelihood of --sample chains against a tight (1e-11, fp64, strict) run of the same path -- plus,
for the fp32 arm, its distance from the fp64 arm AT THE SAME TOLERANCE (what the number type alone costs)."""
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
    ap.add_argument("--chains", type=int, default=32768)
    ap.add_argument("--sample", type=int, default=32)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--arith", default="fma", help="arithmetic of the fp64 arm")
    ap.add_argument("--workload", default="c5")
    args = ap.parse_args()
    mm = mmid_amd_loader.load()
    arith = mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT
    pb0 = mm.workloads.build(args.workload, os.path.join(ROOT, "tests", "golden"), hip_factory=lambda p: mm.HipObjective(p))
    theta = mm.draws.jitter_draws(pb0, 1, args.chains)
    import torch
    d_theta = torch.from_numpy(theta).cuda()
    d_ll = torch.empty(args.chains, dtype=torch.float64, device="cuda")
    tight = mm.HipObjective(pb0.with_(abs_err=1e-11, rel_err=1e-11, arith=mm.ARITH_STRICT)).eval_batch(theta[:args.sample], want_traj=True)

    def rel_state(a, b):
        return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))

    for tol in (1e-3, 1e-4, 1e-5, 1e-6):
        ref64 = None
        for dtype, prec in (("f64", mm.PRECISION_F64), ("f32", mm.PRECISION_F32)):
            pb = pb0.with_(abs_err=tol, rel_err=tol, arith=arith, precision=prec)
            hip = mm.HipObjective(pb)
            s = hip.eval_batch(theta[:args.sample], want_traj=True)
            hip.eval_batch_device(d_theta, d_ll)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                hip.eval_batch_device(d_theta, d_ll)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.reps
            info = hip.kernel_info(args.chains)
            row = {"abs_rel_tol": tol, "dtype": dtype, "evals_per_s": args.chains / dt, "ms_per_step": dt * 1e3,
                   "accepted_mean": float(s["n_accept"].mean()), "rejected_mean": float(s["n_reject"].mean()),
                   "status_ok": int(np.sum(s["status"] == 0)),
                   "max_rel_state_err_vs_1e-11": rel_state(s["traj"], tight["traj"]),
                   "max_rel_loglik_err_vs_1e-11": float(np.max(np.abs(s["loglik"] - tight["loglik"]) / np.abs(tight["loglik"]))),
                   "max_abs_loglik_err_vs_1e-11": float(np.max(np.abs(s["loglik"] - tight["loglik"]))),
                   "kernel": info["kernel_name"], "vgprs": info["vgprs"], "scratch": info["scratch_bytes"],
                   "max_blocks_per_cu": info["max_blocks_per_cu"]}
            if dtype == "f64":
                ref64 = s
                row["arith"] = args.arith
            else:
                row["max_rel_state_err_vs_f64_same_tol"] = rel_state(s["traj"], ref64["traj"])
                row["max_abs_loglik_diff_vs_f64_same_tol"] = float(np.max(np.abs(s["loglik"] - ref64["loglik"])))
                row["median_abs_loglik_diff_vs_f64_same_tol"] = float(np.median(np.abs(s["loglik"] - ref64["loglik"])))
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
