#!/usr/bin/env python3
"""One-off long parity runs (too long for the test-suite): strict arithmetic against the oracle on the whole
headline batch, and the device-resident sampler against the oracle's sampler over thousands of iterations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mmid_amd_loader, oracle_py
mm = mmid_amd_loader.load()
golden = os.path.join(ROOT, "tests", "golden")

for solver in (0, 1):
    pb = mm.workloads.build("c1", golden).with_(arith=mm.ARITH_STRICT, solver=solver)
    theta = mm.draws.jitter_draws(pb, 1, 4096)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    err = np.abs(got["traj"] - ref["traj"]) / np.maximum(np.abs(ref["traj"]), 1.0)
    print(f"strict solver {solver}: 4096 chains, step counts identical in "
          f"{int(np.sum((got['n_accept'] == ref['n_accept']) & (got['n_reject'] == ref['n_reject'])))} chains, "
          f"max rel state err {err.max():.2e}, max rel loglik err "
          f"{np.max(np.abs(got['loglik'] - ref['loglik']) / np.abs(ref['loglik'])):.2e}")

pb = mm.SEPAIHRDProblem.load(os.path.join(golden, "shipped_problem.json")).with_(arith=mm.ARITH_STRICT, constraint_mode=1)
C, iters, burn, ap = 8, 3000, 500, 100
x0 = oracle_py.Oracle(pb).jitter_draws(pb.base_theta, 3, C, mode=1)
t0 = time.time()
dev = mm.HostObjective(pb).metropolis_hastings(x0, seed=5, iterations=iters, burn_in=burn, adaptation_period=ap, thinning=10,
                                                device_state=True)
t1 = time.time()
same = 0
for c in range(C):
    ref = oracle_py.Oracle(pb).metropolis_hastings(x0[c], 5 + c, iters, burn, adaptation_period=ap, thinning=10)
    ok = np.array_equal(dev["accept_trace"][c], ref["accept_trace"]) and np.array_equal(dev["samples"][c], ref["samples"])
    same += int(ok)
print(f"sampler: {C} chains x {iters} iterations (25 covariance refreshes): accept traces and samples bit-identical to the "
      f"oracle's sampler in {same} of {C} chains; acceptance {dev['accept_trace'].mean():.3f}; device run {t1 - t0:.1f} s")
