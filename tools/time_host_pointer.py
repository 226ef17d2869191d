#!/usr/bin/env python3
"""Host-pointer entry point (sepaihrd_eval_batch: theta upload, launch, wait, results back) against the device-resident
call, per batch size.  The difference is what a host caller of the reference's calculate() / batched optimisers pays."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
import torch
pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden")).with_(arith=mm.ARITH_FMA, solver=0, constraint_mode=1)
hip = mm.HipObjective(pb)
base = mm.draws.jitter_draws(pb, 1, 4096)
for B in (1, 63, 256, 4096, 16384):
    theta = np.ascontiguousarray(np.tile(base, ((B + 4095) // 4096, 1))[:B])
    d_t = torch.from_numpy(theta).cuda()
    d_l = torch.empty(B, dtype=torch.float64, device="cuda")
    hip.reserve(B)
    hip.eval_batch(theta); hip.eval_batch_device(d_t, d_l); torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.eval_batch_device(d_t, d_l)
        torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.eval_batch(theta)
    host = (time.perf_counter() - t0) / reps
    print(json.dumps({"chains": B, "device_resident_ms": round(dev * 1e3, 4), "host_pointer_ms": round(host * 1e3, 4),
                      "overhead_us": round((host - dev) * 1e6, 1)}))
