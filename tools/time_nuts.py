"""Diagnostic: wall time of the No-U-Turn sampler per gradient evaluation (reference test fixture + two multipliers)."""
import sys, time, os
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
pb = mm.SEPAIHRDProblem.load(os.path.join(ROOT, "tests/golden/reference_test_fixture.json"))
names = list(pb.param_names) + ["E0_multiplier", "I0_multiplier"]
sig = dict(pb.sigmas); sig.update(E0_multiplier=0.05, I0_multiplier=0.05)
bnd = dict(pb.bounds); bnd.update(E0_multiplier=(0.5, 1.2), I0_multiplier=(0.1, 3.0))
theta = np.concatenate([np.asarray(pb.base_theta), [1.0, 0.8]])
pb = pb.with_(param_names=names, sigmas=sig, bounds=bnd, base_theta=theta, arith=mm.ARITH_FMA, constraint_mode=1)
h = mm.HostObjective(pb)
h.nuts(theta, 3, iterations=3, adaptation_window=2, max_tree_depth=2)
t = time.perf_counter()
r = h.nuts(theta, 3, iterations=40, adaptation_window=10, max_tree_depth=4)
dt = time.perf_counter() - t
print("nuts: %.3f s, gradient calls %d, launches %d, %.3f ms per gradient launch" % (dt, r["gradient_calls"], r["gradient_launches"], 1e3 * dt / r["gradient_launches"]))
# a longer grid (observations tiled): the two evaluations of a gradient are each ~0.5 ms of kernel here
T = 300
reps = -(-T // len(pb.times))
pl = pb.with_(times=np.arange(float(T)), obs_H=np.tile(pb.obs_H, (reps, 1))[:T], obs_ICU=np.tile(pb.obs_ICU, (reps, 1))[:T],
              obs_D=np.tile(pb.obs_D, (reps, 1))[:T])
hl = mm.HostObjective(pl)
hl.nuts(theta, 3, iterations=2, adaptation_window=1, max_tree_depth=2)
t = time.perf_counter()
r = hl.nuts(theta, 3, iterations=12, adaptation_window=4, max_tree_depth=3)
dt = time.perf_counter() - t
print("nuts, %d-day grid: %.3f s, gradient calls %d, launches %d, %.3f ms per gradient launch" % (T, dt, r["gradient_calls"], r["gradient_launches"], 1e3 * dt / r["gradient_launches"]))
