import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
pb = mm.workloads.build("c1", os.path.join(ROOT, "tests", "golden")).with_(arith=mm.ARITH_FMA, constraint_mode=1)
B = 4096
pools = [torch.from_numpy(mm.draws.jitter_draws(pb, 1 + s * B, B)).cuda() for s in range(4)]
d_ll = torch.empty(B, dtype=torch.float64, device="cuda")
d_st = torch.empty(B, dtype=torch.int32, device="cuda")
d_a = torch.empty(B, dtype=torch.int32, device="cuda"); d_r = torch.empty(B, dtype=torch.int32, device="cuda")
hip = mm.HipObjective(pb)
stream = torch.cuda.current_stream()
def step(i, s=None):
    hip.eval_batch_device(pools[i % 4], d_ll, d_status=d_st, d_n_accept=d_a, d_n_reject=d_r, stream=(s or stream).cuda_stream, B=B)
hip.reserve(B)
for i in range(5): step(i)
torch.cuda.synchronize()
def timed(K, timing):
    hip.set_timing(timing)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if timing: hip.get_timing()
    hip.set_timing(False)
    return dt / K * 1e3
for rep in range(3):
    print("plain launches, events off: %.4f ms/step; events on: %.4f" % (timed(200, False), timed(200, True)))
side = torch.cuda.Stream(); side.wait_stream(stream)
with torch.cuda.stream(side): step(0, side)
stream.wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    for i in range(20): step(i, torch.cuda.current_stream())
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph of 20 steps: %.4f ms/step" % ((time.perf_counter() - t0) / 200 * 1e3))
