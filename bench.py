#!/usr/bin/env python3
"""bench.py -- throughput of the SEPAIHRD likelihood hot path on MI355X.

A "step" is one pass of the hot path (theta -> ODE solve -> Poisson log-likelihood) over one
batch of synthetic parameter draws that is already resident in HBM.  Default workload is
BASELINE.json configs[1]: SEPAIHRD, 4 age groups, Dopri5 (abs=rel=1e-6), 400-day daily grid
(t=-20..380, T=401), 4096 independent chains per GPU, fp64.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run, one rank per GPU (RCCL).  Chains are independent, so ranks shard the
chains with no data-path collective ("scaling": "weak"); the only collective is the barrier /
max-reduction of the elapsed time, plus the optional post-run all-gather of per-chain summaries
(reported separately, outside the timed region).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# vector FP64 peak: MI355X_MICROARCH.md gives the FP32 vector peak (157.3 TFLOP/s, SIMD-32: a wave64 instruction in
# 2 cycles); FP64 vector instructions take 4 cycles -> half of it (78.6, AMD's public figure)
FP64_VALU_PEAK_TFLOPS = 78.6
PRE_WARM_SECONDS = 0.25     # the same step, untimed, before the W warm-up steps (see main)
LOG_FLOP_EQUIV = 20         # flop-equivalents per log


# which arithmetic carries which contract (tests named are in the -m gpu suite)
PARITY_NOTES = {
    "strict": "no contraction, IEEE division, the CPU build's operation sequence: step counts identical to the oracle "
              "for every chain of this batch, states <= 1e-9, MH accept traces bit-identical "
              "(test_headline_batch_matches_oracle_chain_by_chain, test_multichain_mh_matches_oracle_sampler)",
    "fma": "the product arithmetic (default of bench.py AND of the drop-in constructors; SEPAIHRD_ARITH=strict opts out): "
           "contraction + folded constants + hardware log2/exp2 in the step-size factor: states <= 1e-6 (north-star "
           "tolerance; measured 1e-11), log-likelihood <= 1e-7; MH accept decisions: 0 of 409 595 904 differ from strict over "
           "4096 chains x the reference's 100 000 iterations (profiles/r04_fma_vs_strict_100k.json), 0 of 1.2 M against the strict "
           "ORACLE (test_headline_batch_accept_traces_in_production_arithmetic), this run's own count in "
           "sampler_pipeline.accept_trace_mismatches_vs_strict -- measured, not proven",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c5"], default="c1",
                    help="BASELINE.json configs: c1 = configs[1] (default), c2 = Cash-Karp 65 536 chains, "
                         "c3 = 32 768 chains/GPU, c5 = 16 age groups x 1 000 days")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU per step (0 = the workload's)")
    ap.add_argument("--solver", choices=["dopri5", "cashkarp", "workload"], default="workload")
    ap.add_argument("--arith", choices=["strict", "fma"], default="fma",
                    help="fma: mul+add contraction on (production mode, parity-tested to the 1e-6 north-star "
                         "tolerance); strict: the CPU build's operation sequence (bit-level parity mode)")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64",
                    help="number type of the ODE state: f64 = the reference's arithmetic (headline); f32 = the fp32-state arm "
                         "of BASELINE configs[4]'s sweep (fp64 likelihood), 3..16 age classes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--allgather", action="store_true", help="time the RCCL all-gather of chain summaries")
    ap.add_argument("--other-workloads", type=int, default=1,
                    help="1 (default, with --workload c1 on one GPU): a few steps of every other BASELINE config after the timed region "
                         "(other_workloads in the line; about 25 s, most of it generating their draws); 0 = skip")
    ap.add_argument("--sampler-iterations", type=int, default=400,
                    help="informational Adaptive-Metropolis run around the kernel after the timed region (0 = skip)")
    ap.add_argument("--sampler-long-iterations", type=int, default=100000,
                    help="one more such run at the reference's own settings (data/configuration/mcmc_settings.txt: 100 000 "
                         "iterations, burn-in 5 000, adaptation_period 100, thinning 100): about a minute (0 = skip)")
    return ap.parse_args()


def algorithmic_flops_per_eval(n, accepted_steps, t_obs):
    rhs = 2 * n * n + 43 * n + 1
    return accepted_steps * (6 * rhs + 59 * 11 * n) + rhs + 3 * t_obs * n * (LOG_FLOP_EQUIV + 4)


def problem_bytes(pb):
    """Shared read-only problem data read once per launch (amortised over the batch)."""
    n, T = pb.n, pb.n_times
    return 8 * (T + 3 * pb.n_obs * n + n * n + 11 * n + 10 * n + len(pb.beta_values) * 2 +
                len(pb.kappa_values) * 2 + 3 * pb.n_params)


def host_cpu_share(omp_max):
    """Threads the CPU baseline may use: the process' CPU affinity, capped by a cgroup CPU quota when one
    is set (a 1-GPU box grants a 16-CPU share of a 128-thread host)."""
    n = min(omp_max, len(os.sched_getaffinity(0)))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    env_cap = os.environ.get("SEPAIHRD_CPU_THREADS")
    return max(1, min(n, int(env_cap)) if env_cap else n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pb, theta, budget_s):
    """Oracle (CPU restatement of the reference path, kind "port") on a bounded sample of the
    same draws, OpenMP over chains on all host cores.  Checker only: never the measured path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    orc = oracle_py.Oracle(pb)
    cores = host_cpu_share(oracle_py.load().oracle_num_threads())
    # single thread first: ~2 s
    t0 = time.perf_counter()
    orc.eval_batch(theta[:8], nthreads=1)
    per_eval_1t = (time.perf_counter() - t0) / 8
    n1 = int(max(8, min(len(theta), 2.0 / per_eval_1t)))
    t0 = time.perf_counter()
    orc.eval_batch(theta[:n1], nthreads=1)
    dt1 = time.perf_counter() - t0
    # all cores: repeat the batch until ~budget_s of wall time has been spent
    orc.eval_batch(theta[:min(len(theta), 4 * cores)], nthreads=cores)  # thread-pool warm-up
    n_eval, dt = 0, 0.0
    while dt < budget_s and n_eval < 400 * len(theta):
        t0 = time.perf_counter()
        orc.eval_batch(theta, nthreads=cores)
        dt += time.perf_counter() - t0
        n_eval += len(theta)
    native = None
    try:  # the clearly-labelled second figure of SURVEY.md 8(d): the same port built -O3 -march=native (FMA contraction on)
        import subprocess
        odir = os.path.join(ROOT, "oracle")
        # -B: rebuilt on THIS host (a copy built elsewhere may use instructions this CPU lacks)
        subprocess.run(["make", "-B", "-C", odir, "liboracle_native.so"], check=True, capture_output=True, timeout=300)
        nat = oracle_py.Oracle(pb, lib_path=os.path.join(odir, "liboracle_native.so"))
        nat.eval_batch(theta[:min(len(theta), 4 * cores)], nthreads=cores)
        n_nat, dt_nat = 0, 0.0
        while dt_nat < max(2.0, budget_s / 4) and n_nat < 400 * len(theta):
            t0 = time.perf_counter()
            nat.eval_batch(theta, nthreads=cores)
            dt_nat += time.perf_counter() - t0
            n_nat += len(theta)
        native = {"value": n_nat / dt_nat, "unit": "evals/s", "cores": cores,
                  "build": "oracle/liboracle_native.so: g++ -O3 -DNDEBUG -march=native (not the reference's flags; "
                           "results differ from the strict oracle in the last digits)"}
    except Exception as e:  # the native build is optional
        native = {"error": str(e)[:160]}
    return {
        "value": n_eval / dt, "unit": "evals/s", "cores": cores, "kind": "port", "march_native": native,
        "sample": f"{n_eval} evaluations ({n_eval // len(theta)} passes over the step's {len(theta)} jittered "
                  f"draws, {dt:.1f} s), oracle/liboracle.so (g++ -O3 -DNDEBUG, no -march=native, no FMA), "
                  f"OpenMP over chains on {cores} threads; single thread: {n1 / dt1:.1f} evals/s",
        "single_thread_value": n1 / dt1,
        "cpu_model": cpu_model(), "nproc": os.cpu_count(), "omp_max_threads": oracle_py.load().oracle_num_threads(),
    }


def sampler_pipeline(mm, pb, theta, iterations=None, long_iterations=0, step_ms=None):
    """Informational, outside the timed region: the whole Adaptive-Metropolis iteration around the kernel
    (random streams, proposal / adaptation state, accept test and scale adaptation all on the device, one evaluation per
    chain and iteration; the host queues iterations), proposals per second for the step's chains.
      * a short run (`iterations`): the iteration loop as the host library times it, MEDIAN of three runs, ONE chain group (the default; two
        groups are timed beside it); the same run in strict arithmetic gives accept_trace_mismatches_vs_strict -- the
        acceptance contract of the arithmetic `value` is measured in, counted on this run's own chains;
      * a run at the reference's own settings (`long_iterations`, burn-in 5 000, adaptation_period 100, thinning 100):
        the covariance refresh is O(P^2) from running co-moments, so the iteration must not slow down as the chain
        grows -- reported against the bare evaluation step.
    None if the host library is absent."""
    try:
        iters = int(iterations or 120)
        host = mm.HostObjective(pb)
        host.metropolis_hastings(theta[:16], 1, 4, 1, device_state=True)
        kw = dict(burn_in=iters // 3, adaptation_period=max(10, iters // 4), thinning=iters)

        def median_of_three(run):
            runs = sorted((run(iters) for _ in range(3)), key=lambda p: p[1]["loop_seconds"])
            dt, r = runs[1]
            return r["loop_seconds"] / (iters - 1), dt, r

        def run_one(n):
            t0 = time.perf_counter()
            r = host.metropolis_hastings(theta, 1, n, device_state=True, **kw)
            return time.perf_counter() - t0, r
        # two groups of chains, each with its own context, stream and host thread: while one group's evaluation runs
        # the other group's accept test and draws are made (same chains, same results: chain c draws from mt19937(1 + c))
        pair = [mm.HostObjective(pb), mm.HostObjective(pb)]
        mm.hostabi.metropolis_hastings_groups(pair, theta[:32], 1, 4, 1)

        def run_two(n):
            t0 = time.perf_counter()
            r = mm.hostabi.metropolis_hastings_groups(pair, theta, 1, n, **kw)
            return time.perf_counter() - t0, r
        s1, dt1, r1 = median_of_three(run_one)
        s2, dt2, r2 = median_of_three(run_two)
        same = bool(np.array_equal(r1["accept_trace"], r2["accept_trace"]))
        # ONE chain group is what the sampler runs by default and what this line reports; two groups (two contexts, streams and
        # host threads) beside it for information (no "better of")
        steady, dt, r, groups = s1, dt1, r1, 1
        out = {"proposals_per_s": theta.shape[0] / steady, "ms_per_iteration": steady * 1e3, "chain_groups": groups,
               "ms_per_iteration_by_groups": {"1": s1 * 1e3, "2": s2 * 1e3}, "groups_give_identical_accept_traces": same,
               "ms_per_iteration_incl_setup": dt / (iters - 1) * 1e3, "setup_and_readback_ms": max(0.0, (dt - steady * (iters - 1)) * 1e3),
               "chains": int(theta.shape[0]), "iterations": iters, "covariance_refreshes_in_run": int(sum(1 for t in range(kw["burn_in"] + 1, iters) if t % kw["adaptation_period"] == 0)),
               "acceptance": float(r["accepted"].mean() / (iters - 1)),
               "note": "ms_per_iteration = the iteration loop of a %d-iteration run timed inside the host library (median of three runs; the run's "
                       "set-up and read-back are reported separately), one chain group; sampler state resident in HBM, "
                       "as are the chains' mt19937 streams, the accept test and the scale adaptation: the host only queues iterations (DESIGN.md 6c)" % iters}
        if pb.arith == mm.ARITH_FMA:
            strict = mm.HostObjective(pb.with_(arith=mm.ARITH_STRICT)).metropolis_hastings(theta, 1, iters, device_state=True, **kw)
            diff = strict["accept_trace"] != r1["accept_trace"]
            out["accept_trace_mismatches_vs_strict"] = int(diff.sum())
            out["accept_trace_decisions_compared"] = int(diff.size)
            out["chains_with_a_mismatch"] = int(diff.any(axis=1).sum())
        if long_iterations and long_iterations > 1:
            n = int(long_iterations)
            t0 = time.perf_counter()
            rl = host.metropolis_hastings(theta, 1, n, min(5000, n // 3), adaptation_period=100, thinning=100, device_state=True,
                                          want_trace=False)
            wall = time.perf_counter() - t0
            ms = rl["loop_seconds"] / (n - 1) * 1e3
            out["long_run"] = {"iterations": n, "burn_in": min(5000, n // 3), "adaptation_period": 100, "thinning": 100,
                               "ms_per_iteration": ms, "proposals_per_s": theta.shape[0] * (n - 1) / rl["loop_seconds"],
                               "loop_seconds": rl["loop_seconds"], "wall_seconds": wall,
                               "vs_ms_per_step": (ms / step_ms) if step_ms else None,
                               "samples_per_chain": int(rl["samples"].shape[1]),
                               "acceptance": float(rl["accepted"].mean() / (n - 1)),
                               "note": "the reference's own run length (data/configuration/mcmc_settings.txt); covariance refresh from "
                                       "running co-moments, O(P^2) per refresh whatever the chain length; one chain group"}
            try:  # the acceptance contract at THIS run length: fma against strict, same seeds (replayed from the committed run)
                with open(os.path.join(ROOT, "profiles", "r04_fma_vs_strict_100k.json")) as fh:
                    fv = json.load(fh)
                out["long_run"]["accept_trace_mismatches_vs_strict"] = {
                    "chains_with_a_flip": fv["chains_with_a_flip"], "decisions_compared": fv["decisions_compared_on_identical_chains"],
                    "chains": fv["chains"], "iterations": fv["iterations"],
                    "source": "profiles/r04_fma_vs_strict_100k.json (tools/fma_vs_strict_100k.py: this workload, these settings, "
                              "once in each arithmetic; replayed, NOT measured in this run -- the strict run takes 85 s)"}
            except Exception:
                pass
        return out
    except Exception as e:  # informational only
        return {"error": str(e)[:200]}


OTHER_WORKLOADS = (("c2", "f64", 10), ("c3", "f64", 10), ("c5", "f64", 5), ("c5_f32", "f32", 5))


def measure_other_workload(mm, torch, key, precision, steps, arith, dev, local_rank, golden_dir, traffic_table):
    """One more BASELINE config in the driver's own line, after the timed region of the headline:
    `steps` steps of the workload's own batch (configs[2] 65 536 chains Cash-Karp, configs[3]'s 32 768-chain per-GPU share,
    configs[4] 16 ages x 1000 days in fp64 and with fp32 state), theta resident, two warm-up steps, HIP events on the stream
    around the pass (after the same quarter-second pre-warm as the headline) and, in the same pass, around every launch's integrator
    kernel and likelihood pass (kernel_ms + likelihood_pass_ms <= ms_per_step by construction)."""
    from mmid_amd import draws, workloads
    name = key.split("_")[0]
    pb = workloads.build(name, golden_dir, hip_factory=lambda q: mm.HipObjective(q, device=local_rank))
    pb.arith = arith
    pb.constraint_mode = mm.CONSTRAINT_REFLECT
    pb.precision = mm.PRECISION_F32 if precision == "f32" else mm.PRECISION_F64
    B = workloads.DEFAULT_CHAINS[name]
    theta = torch.from_numpy(draws.jitter_draws(pb, 1, B)).to(dev)
    d_ll = torch.empty(B, dtype=torch.float64, device=dev)
    d_status = torch.empty(B, dtype=torch.int32, device=dev)
    d_acc = torch.empty(B, dtype=torch.int32, device=dev)
    d_rej = torch.empty(B, dtype=torch.int32, device=dev)
    hip = mm.HipObjective(pb, device=local_rank)
    stream = torch.cuda.current_stream(dev)

    def step():
        hip.eval_batch_device(theta, d_ll, d_status=d_status, d_n_accept=d_acc, d_n_reject=d_rej, stream=stream.cuda_stream, B=B)

    hip.reserve(B)
    # the same untimed pre-warm as the headline (PRE_WARM_SECONDS of the step, at least two): a context's first launches run
    # 4-8 % slow (measured: c3 2.17 ms in the first ten steps, 1.99 in the next ten)
    t_pre, n_pre = time.perf_counter(), 0
    while n_pre < 2 or time.perf_counter() - t_pre < PRE_WARM_SECONDS:
        step()
        n_pre += 1
        torch.cuda.synchronize(dev)

    def timed_pass():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(steps):
            step()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3 / steps, e0.elapsed_time(e1) / steps

    # ONE pass gives the step and the kernel's share of it: the per-launch event records cost ~15 us of stream time, nothing
    # against steps of 2-16 ms (the headline's 0.5-ms step is the case that needs a second pass), and kernel_ms +
    # likelihood_pass_ms <= ms_per_step holds by construction
    hip.set_timing(1)
    wall_ms, stream_ms = timed_pass()
    tm = hip.get_timing()
    hip.set_timing(False)
    kernel_ms = tm["integrator_ms"] / max(tm["launches"], 1)
    ll_ms = tm["likelihood_ms"] / max(tm["launches"], 1)
    info = hip.kernel_info(B)
    acc = d_acc.cpu().numpy().astype(np.float64)
    rej = d_rej.cpu().numpy().astype(np.float64)
    status = d_status.cpu().numpy()
    flops_eval = algorithmic_flops_per_eval(pb.n, float(acc.mean()), pb.n_obs)
    peak = FP64_VALU_PEAK_TFLOPS if precision == "f64" else 2 * FP64_VALU_PEAK_TFLOPS
    solver_name = "dopri5" if pb.solver == mm.SOLVER_DOPRI5 else "cashkarp"
    bytes_eval = 8 * pb.n_params + 16 + problem_bytes(pb) / B
    split_ll = info.get("likelihood_form", 0) == 1 and precision == "f64"
    tkey = f"{name}_{solver_name}_{('fma' if arith == mm.ARITH_FMA else 'strict') if precision == 'f64' else 'f32'}_B{B}"
    out = {
        "workload": f"BASELINE {name}: SEPAIHRD {pb.n} age groups, {solver_name}, {int(pb.times[-1] - pb.times[0])} days (T={pb.n_times}), "
                    f"{B} chains/GPU, {'fp64' if precision == 'f64' else 'fp32 state, fp64 likelihood / time / theta'}",
        "steps": steps, "ms_per_step": wall_ms, "step_ms_on_stream": stream_ms, "evals_per_s": B / (wall_ms * 1e-3),
        "kernel": info["kernel_name"], "kernel_ms": kernel_ms, "likelihood_pass_ms": ll_ms,
        "kernel_ms_source": "HIP events around every launch of the SAME pass ms_per_step is timed over",
        "likelihood_form": {0: "inline", 1: "separate pass over parked increments", 2: "consumer waves of the integrator's workgroup"}.get(info.get("likelihood_form", 0)),
        "roofline": {"bound": "fp64_valu" if precision == "f64" else "fp32_valu", "peak": peak, "unit": "TFLOP/s",
                     "achieved": flops_eval * B / (kernel_ms * 1e-3) / 1e12, "frac": flops_eval * B / (kernel_ms * 1e-3) / 1e12 / peak,
                     "frac_of_step": flops_eval * B / (wall_ms * 1e-3) / 1e12 / peak, "algorithmic_flops_per_eval": flops_eval,
                     "hbm_algorithmic_bytes_per_eval": bytes_eval,
                     "hbm_workspace_bytes_per_eval": pb.n_times * 3 * pb.n * 8 if split_ll else 0,
                     "hbm_frac_algorithmic": bytes_eval * B / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "traffic": traffic_table.get(tkey, {}).get("hbm_bytes_per_launch"),
                     "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc, replayed)" if tkey in traffic_table else None},
        "kernel_info": {k: info[k] for k in ("lanes_per_chain", "chains_per_wave", "vgprs", "lds_bytes", "scratch_bytes", "max_blocks_per_cu",
                                             "phase_pass_applied")},
        "steps_per_eval": {"accepted_mean": float(acc.mean()), "rejected_mean": float(rej.mean()), "attempts_max": float((acc + rej).max())},
        "status_counts": np.bincount(status, minlength=5).tolist(),
    }
    hip.close()
    theta_host = theta.cpu().numpy() if name == "c2" and precision == "f64" else None
    del theta, d_ll, d_status, d_acc, d_rej
    torch.cuda.empty_cache()
    if theta_host is not None:
        # configs[2] is the reference's largest single-process run: the whole Adaptive-Metropolis iteration at its size (streams,
        # accept test, scale and covariance adaptation on the device; one short run, one chain group)
        try:
            iters = 200
            host = mm.HostObjective(pb)
            host.metropolis_hastings(theta_host[:16], 1, 4, 1, device_state=True)
            r = host.metropolis_hastings(theta_host, 1, iters, iters // 3, adaptation_period=50, thinning=iters, device_state=True)
            ms = r["loop_seconds"] / (iters - 1) * 1e3
            out["sampler_iteration"] = {"iterations": iters, "ms_per_iteration": ms, "proposals_per_s": B / (ms * 1e-3),
                                        "vs_ms_per_step": ms / wall_ms, "acceptance": float(r["accepted"].mean() / (iters - 1)),
                                        "note": "draws of the next test queue behind the evaluation at this size (DESIGN.md 6)"}
            del host
        except Exception as e:  # the host library is optional
            out["sampler_iteration"] = {"error": repr(e)}
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # SEPAIHRD_BENCH_REHEARSAL=1: every rank on device 0 with the gloo backend -- exercises the multi-rank
    # code path (sharded draws, barriers, max-reduction, rank-0 line) on a one-GPU box; not a measurement
    rehearsal = os.environ.get("SEPAIHRD_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import mmid_amd_loader
    mm = mmid_amd_loader.load()
    from mmid_amd import draws

    from mmid_amd import workloads
    pb = workloads.build(args.workload, os.path.join(ROOT, "tests", "golden"),
                         hip_factory=lambda q: mm.HipObjective(q, device=local_rank))
    if args.solver != "workload":
        pb.solver = mm.SOLVER_DOPRI5 if args.solver == "dopri5" else mm.SOLVER_CASH_KARP54
    solver_name = "dopri5" if pb.solver == mm.SOLVER_DOPRI5 else "cashkarp"
    pb.arith = mm.ARITH_STRICT if args.arith == "strict" else mm.ARITH_FMA
    pb.constraint_mode = mm.CONSTRAINT_REFLECT
    pb.precision = mm.PRECISION_F32 if args.precision == "f32" else mm.PRECISION_F64
    B, K, W = (args.chains or workloads.DEFAULT_CHAINS[args.workload]), args.steps, args.warmup
    P = pb.n_params

    # synthetic draws: chain b of rank r in pool slot s owns mt19937(1 + (s*world + r)*B + b)
    n_pool = min(max(K, 1), 4)
    pools_host = [draws.jitter_draws(pb, 1 + (s * world + rank) * B, B) for s in range(n_pool)]
    dev = torch.device("cuda", local_rank)
    pools = [torch.from_numpy(p).to(dev) for p in pools_host]
    d_ll = torch.empty(B, dtype=torch.float64, device=dev)
    d_status = torch.empty(B, dtype=torch.int32, device=dev)
    d_acc = torch.empty(B, dtype=torch.int32, device=dev)
    d_rej = torch.empty(B, dtype=torch.int32, device=dev)

    hip = mm.HipObjective(pb, device=local_rank)
    stream = torch.cuda.current_stream(dev)

    def step(i):
        hip.eval_batch_device(pools[i % n_pool], d_ll, d_status=d_status, d_n_accept=d_acc, d_n_reject=d_rej,
                              stream=stream.cuda_stream, B=B)

    hip.reserve(B)
    # Before the W warm-up steps: the device is brought out of idle (clocks, code objects, launch path) by running the same
    # step for PRE_WARM_SECONDS -- a step is half a millisecond here, so the driver's 5 warm-up steps alone end before the
    # card has left its idle state and the first timed launches pay for it (ms_per_step read 7 % above the kernel's own
    # time).  Untimed, reported as config.pre_warmup_steps.
    pre_warm_steps = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < PRE_WARM_SECONDS or pre_warm_steps < 8:
        step(pre_warm_steps)
        pre_warm_steps += 1
        if pre_warm_steps % 8 == 0:
            torch.cuda.synchronize(dev)
    torch.cuda.synchronize(dev)
    for i in range(W):
        step(i)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(K):
        step(i)
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    step_ms = ev0.elapsed_time(ev1) / max(K, 1)
    # The dominant kernel's own duration: a SECOND pass of the same K steps, outside the timed region, with HIP events
    # on the launch stream around the integrator kernel and the likelihood pass of EVERY launch (the records cost ~15 us of
    # stream time per launch, which is why they are not in the timed pass: the kernel time is what they bracket, not
    # what they cost).  roofline.achieved divides by this; roofline.frac_of_step divides by ms_per_step instead.
    hip.set_timing(1)
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record(stream)
    for i in range(K):
        step(i)
    k1.record(stream)
    torch.cuda.synchronize(dev)
    kernel_pass_step_ms = k0.elapsed_time(k1) / max(K, 1)   # that pass's OWN step: kernel_ms is part of it by construction
    tm = hip.get_timing()
    hip.set_timing(False)
    kernel_ms = tm["integrator_ms"] / max(tm["launches"], 1)      # dominant kernel: sepaihrd_eval_kernel
    ll_pass_ms = tm["likelihood_ms"] / max(tm["launches"], 1)     # ll_terms + ll_reduce (0 when inline)

    el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed_max = float(el.item())

    status = d_status.cpu().numpy()
    acc = d_acc.cpu().numpy().astype(np.float64)
    rej = d_rej.cpu().numpy().astype(np.float64)

    # the other arithmetic mode, same draws, reported next to the headline (not part of `value`)
    other = "strict" if args.arith == "fma" else "fma"
    if args.precision == "f32":  # next to the fp32 arm: the fp64 production arithmetic on the same draws
        other = "fma"
        hip.set_precision(mm.PRECISION_F64)
    hip.set_arith(mm.ARITH_STRICT if other == "strict" else mm.ARITH_FMA)
    for i in range(max(1, W)):
        step(i)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_other = time.perf_counter()
    e0.record(stream)
    n_other = max(1, K)  # the SAME number of steps as the headline
    for i in range(n_other):
        step(i)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    other_wall_ms = (time.perf_counter() - t_other) * 1e3 / n_other
    other_ms = e0.elapsed_time(e1) / n_other
    strict_wall_ms = other_wall_ms if other == "strict" else (elapsed / max(K, 1) * 1e3 if args.arith == "strict" and args.precision == "f64" else None)
    hip.set_arith(pb.arith)
    hip.set_precision(pb.precision)

    allgather_ms = None
    allgather_check = None
    if args.allgather and world > 1:
        # post-calibration ensemble summary record per chain: [P means | P variances |
        # best logpost | accept count] -- here filled with the chain's theta and log-likelihood.
        # RCCL over xGMI on GPUs; in the rehearsal (every rank on one GPU) the same call over gloo on host copies
        rec = torch.zeros(B, 2 * P + 2, dtype=torch.float64, device=dev)
        rec[:, :P] = pools[(K - 1) % n_pool]
        rec[:, 2 * P] = d_ll
        rec[:, 2 * P + 1] = float(rank)
        if rehearsal:
            rec = rec.cpu()
        gathered = torch.empty(world * B, 2 * P + 2, dtype=torch.float64, device=rec.device)
        dist.all_gather_into_tensor(gathered, rec)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(gathered, rec)
        torch.cuda.synchronize(dev)
        allgather_ms = (time.perf_counter() - t1) * 1e3
        # every rank's block sits at its rank offset and this rank's own block came back unchanged
        owners = gathered[:, 2 * P + 1].reshape(world, B)
        allgather_check = bool(torch.equal(owners, torch.arange(world, dtype=torch.float64, device=rec.device)[:, None].expand(world, B))
                               and torch.equal(gathered[rank * B:(rank + 1) * B], rec))

    if rank == 0:
        evals_total = world * B * K
        value = evals_total / elapsed_max
        info = hip.kernel_info(B)
        bytes_eval = 8 * P + 16 + problem_bytes(pb) / B        # theta in, loglik + counters out
        split_ll = info.get("likelihood_form", 0) == 1 and args.precision == "f64"  # SEPAIHRD_LL_SEPARATE_PASS: increments parked in HBM
        ws_bytes_eval = pb.n_times * 3 * pb.n * 8 if split_ll else 0
        bytes_launch = bytes_eval * B
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        flops_eval = algorithmic_flops_per_eval(pb.n, float(acc.mean()), pb.n_obs)
        fp64_tflops = flops_eval * B / (kernel_ms * 1e-3) / 1e12
        traffic = None
        traffic_src = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(traffic_src):
            try:
                with open(traffic_src) as fh:
                    tj = json.load(fh)
                key = f"{args.workload}_{solver_name}_{args.arith if args.precision == 'f64' else 'f32'}_B{B}"
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "ODE-solve+likelihood evals/sec (SEPAIHRD %d-age, %dd)" % (pb.n, int(pb.times[-1] - pb.times[0])),
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed_max / max(K, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.precision == "f64" else "f32", "data": "synthetic",
            # the bit-exact-contract arithmetic on the same draws over the same number of steps (config.parity_mode)
            "value_strict": (world * B / (strict_wall_ms * 1e-3)) if strict_wall_ms else None,
            "config": {
                "workload": f"BASELINE {args.workload}: SEPAIHRD {pb.n} age groups, {solver_name}, "
                            f"{int(pb.times[-1] - pb.times[0])} days (T={pb.n_times}), {B} chains/GPU, fp64",
                "chains_per_gpu": B, "n_params": P, "abs_err": pb.abs_err, "rel_err": pb.rel_err,
                "pre_warmup_steps": pre_warm_steps,  # untimed, before the W warm-up steps: PRE_WARM_SECONDS of the same step
                "arith": args.arith if args.precision == "f64" else "f32 state (fma), fp64 likelihood / time / theta",
                "other_arith": {"mode": other if args.precision == "f64" else "f64 " + other, "steps": n_other, "ms_per_step": other_wall_ms,
                                "ms_per_step_on_stream": other_ms,
                                "evals_per_s_per_gpu": B / (other_wall_ms * 1e-3)},
                "parity_mode": PARITY_NOTES,
                "draws": "reflect(base + sigma*N(0,1)), mt19937(1+chain), libstdc++ order",
                "parallelism": f"chains sharded over {world} GPU(s), no data-path collective",
            },
            # the binding resource first: ~4000 flop per byte puts the path on the FP64 vector ALU, not on
            # HBM and not on MFMA (a 4x4 contact contraction is far below a tile); the HBM figures BASELINE.json asks
            # for follow in "hbm"
            "roofline": {
                "bound": "fp64_valu" if args.precision == "f64" else "fp32_valu", "achieved": fp64_tflops,
                "peak": FP64_VALU_PEAK_TFLOPS if args.precision == "f64" else 2 * FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fp64_tflops / (FP64_VALU_PEAK_TFLOPS if args.precision == "f64" else 2 * FP64_VALU_PEAK_TFLOPS),
                "frac_of_step": flops_eval * B / (elapsed_max / max(K, 1)) / 1e12 / (FP64_VALU_PEAK_TFLOPS if args.precision == "f64" else 2 * FP64_VALU_PEAK_TFLOPS),
                "kernel_ms_source": "HIP events around every launch of a second, untimed pass of the same %d steps" % K,
                "likelihood_form": {0: "inline", 1: "separate pass over parked increments", 2: "consumer waves of the integrator's workgroup"}.get(info.get("likelihood_form", 0)),
                "algorithmic_flops_per_eval": flops_eval,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": ("profile (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of this "
                                   "workload, gfx950 corrections applied; replayed, NOT measured in this run)"
                                   if traffic is not None else None),
                "kernel": info["kernel_name"], "kernel_ms": kernel_ms, "likelihood_pass_ms": ll_pass_ms,
                "step_ms_on_stream": step_ms,
                # the pass kernel_ms comes from, timed as a whole on the same stream: kernel_ms <= kernel_pass_step_ms always;
                # against ms_per_step of the timed pass (no event records between its launches) it can read either way by the
                # box's clock of the moment
                "kernel_pass_step_ms": kernel_pass_step_ms,
                "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_eval": bytes_eval,
                        "likelihood_workspace_bytes_per_eval": ws_bytes_eval,
                        "achieved_incl_workspace": (bytes_eval + ws_bytes_eval) * B / (kernel_ms * 1e-3) / 1e9,
                        "note": "BASELINE.json asks for the HBM fraction: reported, never padded -- the path moves "
                                "~0.5 KB per 2 MFLOP evaluation and cannot approach the HBM roof"},
            },
            "kernel_info": {k: info[k] for k in ("lanes_per_chain", "chains_per_wave", "vgprs", "lds_bytes",
                                                 "scratch_bytes", "max_blocks_per_cu", "num_cus",
                                                 "device_name", "phase_pass_applied")},
            "steps_per_eval": {"accepted_mean": float(acc.mean()), "rejected_mean": float(rej.mean()),
                               "attempts_max": float((acc + rej).max())},
            "status_counts": np.bincount(status, minlength=4).tolist(),
        }
        if allgather_ms is not None:
            out["allgather_ms"] = allgather_ms
            out["allgather"] = {"ms": allgather_ms, "bytes_per_rank": B * (2 * P + 2) * 8, "ranks": world,
                                "backend": "gloo (rehearsal on one GPU)" if rehearsal else "nccl (RCCL)",
                                "blocks_in_rank_order": allgather_check}
        if rehearsal:
            out["rehearsal"] = True  # not a measurement: every rank shares device 0
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(pb, pools_host[0], args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        out["other_workloads"] = None
        if world == 1 and args.workload == "c1" and args.other_workloads and args.precision == "f64":
            tt = {}
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                    tt = json.load(fh)
            except Exception:
                pass
            out["other_workloads"] = {}
            for key, precision, steps in OTHER_WORKLOADS:
                try:
                    out["other_workloads"][key] = measure_other_workload(mm, torch, key, precision, steps, pb.arith, dev, local_rank,
                                                                         os.path.join(ROOT, "tests", "golden"), tt)
                except Exception as e:  # informational: never costs the headline its line
                    out["other_workloads"][key] = {"error": str(e)[:200]}
        out["sampler_pipeline"] = (sampler_pipeline(mm, pb, pools_host[0], args.sampler_iterations,
                                                    args.sampler_long_iterations if args.workload == "c1" else 0, out["ms_per_step"])
                                   if world == 1 and args.sampler_iterations > 1 else None)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
