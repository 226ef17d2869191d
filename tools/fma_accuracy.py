#!/usr/bin/env python3
"""How far the production (fma) arithmetic is from the CPU oracle on the headline workload:
max relative state error, log-likelihood error and the share of chains with identical step counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mmid_amd_loader, oracle_py
mm = mmid_amd_loader.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for solver in (0, 1):
    pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden")).with_(arith=mm.ARITH_FMA, solver=solver)
    theta = mm.draws.jitter_draws(pb, 1, B)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    err = np.abs(got["traj"] - ref["traj"]) / np.maximum(np.abs(ref["traj"]), 1.0)
    same = np.mean((got["n_accept"] == ref["n_accept"]) & (got["n_reject"] == ref["n_reject"]))
    print(f"solver {solver}: chains {B}  max rel state err {err.max():.3e}  (99.9th pct {np.quantile(err, 0.999):.3e})  "
          f"max rel loglik err {np.max(np.abs(got['loglik'] - ref['loglik']) / np.abs(ref['loglik'])):.3e}  "
          f"identical step counts {100 * same:.1f} %  status equal {np.array_equal(got['status'], ref['status'])}")
