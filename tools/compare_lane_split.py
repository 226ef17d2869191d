#!/usr/bin/env python3
"""The same chains through both forms of the fp64 integrator (sepaihrd_set_integrator_form), every output array compared bit for bit;
with tools/libsepaihrd_hip_experiments.so present also through its one-wavefront-per-chain kernel.
usage: compare_lane_split.py [--solver 0|1] [--chains N]"""
import argparse, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def evaluate(mm, args, form, fma_lib_wave_chain=False):
    """the batch (or the fuzz cases) through one form of the integrator; returns {name: array}"""
    pb = mm.SEPAIHRDProblem.load(os.path.join(ROOT, "tests", "golden", args.problem)).with_(
        solver=args.solver, arith=mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT)
    if args.one_step:
        times = np.array([0.0, args.one_step])
        pb = pb.with_(times=times, obs_H=pb.obs_H[:2], obs_ICU=pb.obs_ICU[:2], obs_D=pb.obs_D[:2])

    def run(q, th):
        hip = mm.HipObjective(q)
        try:
            if form is not None:
                hip.set_integrator_form(form)
            return hip.eval_batch(th, want_traj=True)
        finally:
            hip.close()  # not left to the interpreter's shutdown: two dozen contexts torn down in arbitrary order beside a second process on the card

    rng = np.random.default_rng(1)
    if args.fuzz:
        # random variants of the problem, the same for every form: age classes 3 or 4 (3 pads a lane), output grids
        # of random length and stride (steps that stop at odd places), tolerances, attempt budgets that cut chains
        # short, both constraint modes, wide parameter draws (some invalid), ragged batch sizes
        res = {}
        frng = np.random.default_rng(args.fuzz)
        for v in range(args.fuzz_cases):
            q = pb
            if frng.random() < 0.4:
                q = mm.restrict_age_classes(q, [0, 1, 2])
            T = int(frng.integers(2, 90))
            stride = int(frng.integers(1, 4))
            first = int(np.searchsorted(np.asarray(q.times), 0.0))
            idx = np.concatenate([np.arange(first), first + stride * np.arange(T)])
            idx = idx[idx < len(q.times)]
            times = np.asarray(q.times)[idx]
            nobs = int(np.sum(times >= 0))
            q = q.with_(times=times, obs_H=q.obs_H[:nobs], obs_ICU=q.obs_ICU[:nobs], obs_D=q.obs_D[:nobs],
                        abs_err=float(10.0 ** frng.uniform(-8, -4)), rel_err=float(10.0 ** frng.uniform(-8, -4)),
                        constraint_mode=int(frng.integers(0, 2)))
            if frng.random() < 0.3:
                q = q.with_(max_attempts=int(frng.integers(20, 200)))
            B = int(frng.integers(1, 70))
            lo, hi, _ = q.bounds_arrays()
            wide = frng.random() < 0.5
            th = (lo + (hi - lo) * frng.uniform(-0.1, 1.1, (B, q.n_params))) if wide else \
                np.asarray(q.base_theta)[None, :] * (1 + 0.05 * frng.standard_normal((B, q.n_params)))
            r = run(q, th)
            for k, a in r.items():
                if isinstance(a, np.ndarray):
                    res[f"case{v}_{k}"] = a
        return res
    theta = pb.base_theta[None, :] * (1 + 0.02 * rng.standard_normal((args.chains, pb.n_params)))
    r = run(pb, theta)
    return {k: v for k, v in r.items() if isinstance(v, np.ndarray)}


def child(args):
    """the experiment build's one-wavefront-per-chain kernel (own process: SEPAIHRD_HIP_LIB and the switch are read once)"""
    sys.path.insert(0, ROOT)
    import mmid_amd_loader
    mm = mmid_amd_loader.load()
    np.savez(args.out, **evaluate(mm, args, None))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", type=int, default=0)
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--problem", default="synth_400d_n4.json")
    ap.add_argument("--arith", choices=["fma", "strict"], default="fma")
    ap.add_argument("--fuzz", type=int, default=0, help="seed of a run over random problem variants (0: off)")
    ap.add_argument("--fuzz-cases", type=int, default=24)
    ap.add_argument("--one-step", type=float, default=0.0)
    a = ap.parse_args()
    if a.out:
        child(a)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)  # the arrays are on disk and every context is closed: nothing for the runtime's exit handlers to race over
    sys.path.insert(0, ROOT)
    import mmid_amd_loader
    mm = mmid_amd_loader.load()
    # one lane per (chain, age) and the sixteen-lanes-per-chain form of the shipped library, forced through the C ABI
    outs = [evaluate(mm, a, mm.hipabi.FORM_LANE_PER_AGE), evaluate(mm, a, mm.hipabi.FORM_QUAD)]
    # the one-wavefront-per-chain kernel exists in the experiment build only (csrc/Makefile `experiments`)
    exp_lib = os.path.join(ROOT, "tools", "libsepaihrd_hip_experiments.so")
    if a.arith == "fma" and os.path.exists(exp_lib):
        out = "/tmp/lane_split_wave_chain.npz"
        env = dict(os.environ, SEPAIHRD_HIP_LIB=exp_lib, SEPAIHRD_WAVE_CHAIN="2")
        subprocess.run([sys.executable, __file__, "--solver", str(a.solver), "--chains", str(a.chains), "--out", out, "--problem", a.problem, "--arith", a.arith, "--fuzz", str(a.fuzz), "--fuzz-cases", str(a.fuzz_cases),
                        "--one-step", str(a.one_step)],
                       env=env, check=True)
        loaded = np.load(out)
        outs.append({k: loaded[k] for k in loaded.files})
    all_same = True
    for other in range(1, len(outs)):
        tag = "16-lane" if other == 1 else "wave-per-chain"
        for k in outs[0]:
            x, y = outs[0][k], outs[other][k]
            if k.endswith("traj"):  # rows after the point where a chain stopped (status != 0) are never written
                ok = outs[0][k[:-4] + "status"] == 0
                x, y = x[ok], y[ok]
            same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
            all_same &= bool(same)
            msg = f"{tag} {k}: identical={same}"
            if not same and x.dtype.kind == "f":
                rel = np.abs(x - y) / np.maximum(np.abs(x), 1e-300)
                msg += f" differing={int((x != y).sum())}/{x.size} max_rel={rel.max():.3e}"
            elif not same:
                msg += f" differing={int((x != y).sum())}/{x.size} max_abs={np.abs(x - y).max()}"
            if not (a.fuzz and same):
                print(msg)
            if k == "traj" and not same and not a.fuzz:
                d = (x != y)
                tfirst = np.argmax(d.any(axis=(0, 2)))
                print("first differing output index:", tfirst, "components (c*n+age):", np.unique(np.nonzero(d[:, tfirst, :])[1])[:44])
                ch = np.nonzero(d[:, tfirst, :])[0][0]
                n = x.shape[2] // 11
                for c in range(11):
                    dd = d[:, tfirst, c * n:(c + 1) * n]
                    print("  comp", c, "differing", int(dd.sum()), "of", dd.size, "n_accept", outs[0]["n_accept"][:4])
                print("chain", ch, "values", x[ch, tfirst, d[ch, tfirst]][:6], y[ch, tfirst, d[ch, tfirst]][:6])
    if a.fuzz:
        st = np.concatenate([outs[0][k] for k in outs[0] if k.endswith("_status")])
        print(f"fuzz seed {a.fuzz}: {a.fuzz_cases} problem variants, {st.size} chains, status counts {np.bincount(st, minlength=4).tolist()}, "
              f"all arrays identical={all_same}")
    sys.exit(0 if all_same else 1)
