"""Runs the diagnostic stamps build (tools/libsepaihrd_stamps.so) and prints where a wave's cycles go.
Diagnostic only: its run time is not a benchmark number."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
from mmid_amd import draws, hipabi
arith = sys.argv[1] if len(sys.argv) > 1 else "strict"
hipabi._lib = hipabi.load_library(os.path.join(ROOT, "tools", "libsepaihrd_stamps.so"))
pb = mm.SEPAIHRDProblem.load(os.path.join(ROOT, "tests", "golden", "synth_400d_n4.json"))
pb.arith = mm.ARITH_FMA if arith == "fma" else mm.ARITH_STRICT
n_chains = int(os.environ.get("STAMP_CHAINS", "4096"))
pb.solver = int(os.environ.get("STAMP_SOLVER", "0"))
theta = draws.jitter_draws(pb, 1, n_chains)
hip = mm.HipObjective(pb)
hip.eval_batch(theta)
r = hip.eval_batch(theta)
quad = os.environ.get("SEPAIHRD_LANE_SPLIT", "") != "0" and n_chains <= 4096  # the 16-lane form unless switched off
if quad:
    parts = r["ll_parts"].reshape(-1, 4, 3)   # per wave: 4 chains
    head, body, err = parts[:, 0, 0], parts[:, 0, 1], parts[:, 0, 2]
    tail, att = parts[:, 1, 0], parts[:, 1, 1]
    errA, tailA = err * 0, tail * 0             # sub-sections are stamped in the 4-lane kernel only
else:
    parts = r["ll_parts"].reshape(-1, 16, 3)  # per wave: 16 chains
    head, body, err = parts[:, 0, 0], parts[:, 0, 1], parts[:, 0, 2]
    tail, att, errA = parts[:, 1, 0], parts[:, 1, 1], parts[:, 1, 2]
    tailA = parts[:, 2, 0]
tot = head + body + err + tail
print("arith", arith, "chains", n_chains, "waves", len(tot), "attempts/wave mean %.1f" % att.mean())
for name, v in (("head(stage times, schedule)", head), ("RK body (6 RHS + stage sums + xerr)", body),
                ("error norm + controller", err), ("accept/observe/likelihood tail", tail)):
    print("%-40s %8.0f cycles/attempt  %5.1f %%" % (name, (v / att).mean(), 100 * v.sum() / tot.sum()))
print("   of error section: scale/compare/divisions %.0f, controller pow paths %.0f" % ((errA / att).mean(), ((err - errA) / att).mean()))
print("   of tail: reject/accept state update %.0f, observe+likelihood+prefetch %.0f" % ((tailA / att).mean(), ((tail - tailA) / att).mean()))
print("total stamped cycles/attempt %.0f (s_memtime ticks)" % (tot / att).mean())
