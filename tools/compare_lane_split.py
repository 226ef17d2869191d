#!/usr/bin/env python3
"""Diagnostic: the same chains through the 4-lane and the 16-lane kernel (two processes: the switch is read once).
usage: compare_lane_split.py [--solver 0|1] [--chains N]"""
import argparse, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(args):
    sys.path.insert(0, ROOT)
    import mmid_amd_loader
    mm = mmid_amd_loader.load()
    pb = mm.SEPAIHRDProblem.load(os.path.join(ROOT, "tests", "golden", args.problem)).with_(
        solver=args.solver, arith=mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT)
    if args.one_step:
        times = np.array([0.0, args.one_step])
        pb = pb.with_(times=times, obs_H=pb.obs_H[:2], obs_ICU=pb.obs_ICU[:2], obs_D=pb.obs_D[:2])
    rng = np.random.default_rng(1)
    theta = pb.base_theta[None, :] * (1 + 0.02 * rng.standard_normal((args.chains, pb.n_params)))
    r = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    np.savez(args.out, **{k: v for k, v in r.items() if isinstance(v, np.ndarray)})

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", type=int, default=0)
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--problem", default="synth_400d_n4.json")
    ap.add_argument("--arith", choices=["fma", "strict"], default="fma")
    ap.add_argument("--one-step", type=float, default=0.0)
    a = ap.parse_args()
    if a.out:
        child(a)
        sys.exit(0)
    outs = []
    for mode in ("0", "1"):
        out = f"/tmp/lane_split_{mode}.npz"
        env = dict(os.environ, SEPAIHRD_LANE_SPLIT=mode)
        subprocess.run([sys.executable, __file__, "--solver", str(a.solver), "--chains", str(a.chains), "--out", out, "--problem", a.problem, "--arith", a.arith,
                        "--one-step", str(a.one_step)],
                       env=env, check=True)
        outs.append(np.load(out))
    all_same = True
    for k in outs[0].files:
        x, y = outs[0][k], outs[1][k]
        same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
        all_same &= bool(same)
        msg = f"{k}: identical={same}"
        if not same and x.dtype.kind == "f":
            rel = np.abs(x - y) / np.maximum(np.abs(x), 1e-300)
            msg += f" differing={int((x != y).sum())}/{x.size} max_rel={rel.max():.3e}"
        elif not same:
            msg += f" differing={int((x != y).sum())}/{x.size} max_abs={np.abs(x - y).max()}"
        print(msg)
        if k == "traj" and not same:
            d = (x != y)
            tfirst = np.argmax(d.any(axis=(0, 2)))
            print("first differing output index:", tfirst, "components (c*n+age):", np.unique(np.nonzero(d[:, tfirst, :])[1])[:44])
            ch = np.nonzero(d[:, tfirst, :])[0][0]
            n = x.shape[2] // 11
            for c in range(11):
                dd = d[:, tfirst, c * n:(c + 1) * n]
                print("  comp", c, "differing", int(dd.sum()), "of", dd.size, "n_accept", outs[0]["n_accept"][:4])
            print("chain", ch, "values", x[ch, tfirst, d[ch, tfirst]][:6], y[ch, tfirst, d[ch, tfirst]][:6])
    sys.exit(0 if all_same else 1)
