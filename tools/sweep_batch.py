#!/usr/bin/env python3
"""Throughput of the hot path against the number of chains per launch (theta resident in HBM)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
import torch
for solver, name in ((0, "dopri5"), (1, "cashkarp")):
    pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden")).with_(arith=mm.ARITH_FMA, solver=solver, constraint_mode=1)
    hip = mm.HipObjective(pb)
    base = mm.draws.jitter_draws(pb, 1, 4096)
    sizes = [int(v) for v in os.environ.get("SWEEP_SIZES", "16,256,1024,4096,16384,65536,262144").split(",")]
    for B in sizes:
        theta = np.tile(base, ((B + 4095) // 4096, 1))[:B]
        d_t = torch.from_numpy(theta).cuda()
        d_l = torch.empty(B, dtype=torch.float64, device="cuda")
        hip.reserve(B)
        hip.eval_batch_device(d_t, d_l); torch.cuda.synchronize()
        reps = 20 if B <= 16384 else 4
        t0 = time.perf_counter()
        for _ in range(reps):
            hip.eval_batch_device(d_t, d_l)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"solver": name, "chains": B, "ms": round(dt * 1e3, 4), "evals_per_s": round(B / dt)}))
