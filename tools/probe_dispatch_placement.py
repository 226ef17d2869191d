"""Does a kernel that ran before it change how long the evaluation of a one-wave-per-SIMD batch takes?  (Round 4: yes -- 16 384
chains, 0.91 ms in a loop of evaluations, 1.46 ms behind any other kernel: the dispatcher had put two of its waves on some SIMDs.
csrc/sepaihrd_kernels.hip launch_lds_bytes caps a CU at four of those workgroups since.)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
from mmid_amd import draws, workloads
dev = torch.device("cuda", 0)
pb = workloads.build(os.environ.get("PROBE_WORKLOAD", "c1"), os.path.join(ROOT, "tests", "golden"), hip_factory=lambda q: mm.HipObjective(q))
pb.arith = mm.ARITH_STRICT if os.environ.get("PROBE_ARITH") == "strict" else mm.ARITH_FMA; pb.constraint_mode = mm.CONSTRAINT_REFLECT
if os.environ.get("PROBE_SOLVER") == "cashkarp": pb.solver = mm.SOLVER_CASH_KARP54
if os.environ.get("PROBE_PRECISION") == "f32": pb.precision = mm.PRECISION_F32
B = int(os.environ.get("PROBE_CHAINS", "16384"))
theta = torch.from_numpy(draws.jitter_draws(pb, 1, B)).to(dev)
d_ll = torch.empty(B, dtype=torch.float64, device=dev); d_st = torch.empty(B, dtype=torch.int32, device=dev)
hip = mm.HipObjective(pb); hip.reserve(B)
s = torch.cuda.current_stream(dev)
big = torch.zeros(64 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)     # 64 MB
huge = torch.zeros(1024 * 1024 * 1024 // 8, dtype=torch.float64, device=dev)  # 1 GB
def run(name, before):
    for _ in range(10):
        before(); hip.eval_batch_device(theta, d_ll, d_status=d_st, stream=s.cuda_stream, B=B)
    torch.cuda.synchronize(dev)
    hip.set_timing(1)
    for _ in range(20):
        before(); hip.eval_batch_device(theta, d_ll, d_status=d_st, stream=s.cuda_stream, B=B)
    torch.cuda.synchronize(dev)
    tm = hip.get_timing(); hip.set_timing(False)
    print("%-44s integrator %.4f ms  ll %.4f ms" % (name, tm["integrator_ms"] / tm["launches"], tm["likelihood_ms"] / tm["launches"]), flush=True)
tiny = torch.zeros(64, dtype=torch.float64, device=dev)
mid = torch.zeros(1024 * 1024 // 8, dtype=torch.float64, device=dev)   # 1 MB
run("nothing between evaluations", lambda: None)
if os.environ.get("PROBE_SHORT") != "1":
    run("a 512-B elementwise kernel before each", lambda: tiny.add_(1.0))
    run("a 1-MB elementwise kernel before each", lambda: mid.add_(1.0))
run("a 64-MB elementwise kernel before each", lambda: big.add_(1.0))
if os.environ.get("PROBE_SHORT") != "1":
    run("a 1-GB elementwise kernel before each", lambda: huge.add_(1.0))
    run("a 1-GB memset before each", lambda: huge.zero_())
k64 = torch.zeros(65536, dtype=torch.float64, device=dev)
def both(*fs):
    def f():
        for g in fs: g()
    return f
run("64 MB, then a 512-B kernel", both(lambda: big.add_(1.0), lambda: tiny.add_(1.0)))
run("64 MB, then three 512-B kernels", both(lambda: big.add_(1.0), lambda: tiny.add_(1.0), lambda: tiny.add_(1.0), lambda: tiny.add_(1.0)))
run("64 MB, then a 512-KB kernel (256 workgroups)", both(lambda: big.add_(1.0), lambda: k64.add_(1.0)))
run("64 MB, then a host synchronize", both(lambda: big.add_(1.0), lambda: torch.cuda.synchronize(dev)))
run("nothing between evaluations (again)", lambda: None)
