#!/usr/bin/env python3
"""BASELINE configs[4]: SEPAIHRD with 16 age groups, 1000 days, Dopri5 -- tolerance sweep
abs = rel in {1e-3, 1e-4, 1e-5, 1e-6} (fp64 arm; the fp32 arm is not built).
Per tolerance: throughput of the HIP path, steps per evaluation, and the accuracy of states and
log-likelihood against a tight (1e-11) run of the same path on a sample of chains."""
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
    ap.add_argument("--arith", default="fma")
    args = ap.parse_args()
    mm = mmid_amd_loader.load()
    arith = mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT
    pb0 = mm.workloads.build("c5", os.path.join(ROOT, "tests", "golden"), hip_factory=lambda p: mm.HipObjective(p))
    theta = mm.draws.jitter_draws(pb0, 1, args.chains)
    import torch
    d_theta = torch.from_numpy(theta).cuda()
    d_ll = torch.empty(args.chains, dtype=torch.float64, device="cuda")
    tight = mm.HipObjective(pb0.with_(abs_err=1e-11, rel_err=1e-11, arith=mm.ARITH_STRICT)).eval_batch(theta[:args.sample], want_traj=True)
    for tol in (1e-3, 1e-4, 1e-5, 1e-6):
        pb = pb0.with_(abs_err=tol, rel_err=tol, arith=arith)
        hip = mm.HipObjective(pb)
        s = hip.eval_batch(theta[:args.sample], want_traj=True)
        err = np.max(np.abs(s["traj"] - tight["traj"]) / np.maximum(np.abs(tight["traj"]), 1.0))
        ll_err = np.max(np.abs(s["loglik"] - tight["loglik"]) / np.abs(tight["loglik"]))
        hip.eval_batch_device(d_theta, d_ll)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            hip.eval_batch_device(d_theta, d_ll)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        print(json.dumps({"abs_rel_tol": tol, "evals_per_s": args.chains / dt, "ms_per_step": dt * 1e3,
                          "accepted_mean": float(s["n_accept"].mean()), "rejected_mean": float(s["n_reject"].mean()),
                          "max_rel_state_err_vs_1e-11": float(err), "max_rel_loglik_err_vs_1e-11": float(ll_err),
                          "arith": args.arith, "dtype": "f64"}))


if __name__ == "__main__":
    main()
