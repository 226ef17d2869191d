import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mmid_amd_loader
mm = mmid_amd_loader.load()
pb0 = mm.workloads.build("c5", os.path.join(ROOT, "tests", "golden"), hip_factory=lambda p: mm.HipObjective(p))
theta = mm.draws.jitter_draws(pb0, 1, 8)
for tol in (1e-3, 1e-6):
    a = mm.HipObjective(pb0.with_(abs_err=tol, rel_err=tol, arith=mm.ARITH_FMA)).eval_batch(theta, want_traj=True)
    b = mm.HipObjective(pb0.with_(abs_err=tol, rel_err=tol, precision=mm.PRECISION_F32)).eval_batch(theta, want_traj=True)
    n = pb0.n
    ta, tb = a["traj"].reshape(8, -1, 11, n), b["traj"].reshape(8, -1, 11, n)
    rel = np.abs(tb - ta) / np.maximum(np.abs(ta), 1.0)
    print("tol", tol, "steps", a["n_accept"][:4], b["n_accept"][:4])
    for c, name in enumerate("S E P A I H ICU R D CumH CumICU".split()):
        r = rel[:, :, c, :]
        idx = np.unravel_index(np.argmax(r), r.shape)
        print(f"  {name:7s} max rel {r.max():.2e} at chain {idx[0]} t-index {idx[1]} age {idx[2]} value {ta[idx[0], idx[1], c, idx[2]]:.4g}; median over (t,age) of max-over-chains {np.median(r.max(axis=0)):.2e}; at T/2 {r[:, r.shape[1]//2].max():.2e}; final {r[:, -1].max():.2e}")
    inc_a = np.diff(ta[:, :, 9, :], axis=1); inc_b = np.diff(tb[:, :, 9, :], axis=1)
    ri = np.abs(inc_b - inc_a) / np.maximum(np.abs(inc_a), 1e-3)
    print("  daily CumH increments: max rel", ri.max(), "median", np.median(ri), "ll diff", (b["loglik"] - a["loglik"])[:8])
