#!/usr/bin/env python3
"""Diagnostic: do evaluations of small batches issued on different streams (one ctx each) overlap?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
import torch
pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden")).with_(arith=mm.ARITH_FMA)
for G, per in ((1, 4096), (1, 1024), (2, 1024), (4, 1024), (4, 256), (8, 512)):
    theta = mm.draws.jitter_draws(pb, 1, per)
    objs = [mm.HipObjective(pb) for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    d_t = [torch.from_numpy(theta).cuda() for _ in range(G)]
    d_l = [torch.empty(per, dtype=torch.float64, device="cuda") for _ in range(G)]
    for o in objs: o.reserve(per)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(20):
            for g in range(G):
                objs[g].eval_batch_device(d_t[g], d_l[g], stream=streams[g].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f"G={G} chains/group={per}: {dt*1e3:.3f} ms per round of {G} evaluations ({G*per/dt/1e6:.2f} M evals/s)")
