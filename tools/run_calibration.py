#!/usr/bin/env python3
"""End-to-end calibration run on the device path, shaped like the reference's `main --algorithm hill` / `--algorithm
pso` (src/model/main.cpp:383-560): Hill-Climbing or particle-swarm phase -> conditioned covariance -> Adaptive-Metropolis chains ->
posterior trace files in the sampler's CSV format (MetropolisHastingsSampler.cpp:414-438) -> post-calibration
ensemble (posterior predictive quantiles, seroprevalence and Rt trajectories, per-sample metric table).

    python tools/run_calibration.py --problem tests/golden/shipped_problem.json --out gpurun_out/calibration

The problem is a fixture JSON (tests/golden/make_fixtures.py builds them from a reference-format configuration
directory through config_io.py).  Settings use the reference's keys and defaults
(data/configuration/{hill_climbing,mcmc}_settings.txt)."""
import argparse
import csv
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mmid_amd_loader  # noqa: E402

PROBS = [0.025, 0.05, 0.5, 0.95, 0.975]
SERIES = ["daily_hospitalizations", "daily_icu_admissions", "daily_deaths", "cumulative_hospitalizations",
          "cumulative_icu_admissions", "cumulative_deaths"]
METRICS = ["R0", "overall_IFR", "overall_attack_rate", "peak_hospital", "peak_ICU", "time_to_peak_hospital",
           "time_to_peak_ICU", "total_deaths", "max_Rt", "min_Rt", "final_Rt", "seroprevalence_day64"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default=os.path.join(ROOT, "tests", "golden", "shipped_problem.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "calibration"))
    ap.add_argument("--algorithm", choices=["hill", "pso"], default="hill")
    ap.add_argument("--pso-iterations", type=int, default=100)
    ap.add_argument("--swarm-size", type=int, default=256)
    ap.add_argument("--pso-variant", type=int, default=0)
    ap.add_argument("--pso-topology", type=int, default=0)
    ap.add_argument("--chains", type=int, default=16)
    ap.add_argument("--hc-iterations", type=int, default=60)
    ap.add_argument("--hc-threads", type=int, default=16)
    ap.add_argument("--cloud-size-multiplier", type=int, default=8)
    ap.add_argument("--mcmc-iterations", type=int, default=2000)
    ap.add_argument("--burn-in", type=int, default=500)
    ap.add_argument("--adaptation-period", type=int, default=100)
    ap.add_argument("--thinning", type=int, default=10)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--arith", choices=["fma", "strict"], default="fma")
    args = ap.parse_args()

    mm = mmid_amd_loader.load()
    pb = mm.SEPAIHRDProblem.load(args.problem)
    pb = pb.with_(arith=mm.ARITH_FMA if args.arith == "fma" else mm.ARITH_STRICT, constraint_mode=0)
    os.makedirs(args.out, exist_ok=True)
    host = mm.HostObjective(pb)

    t0 = time.perf_counter()
    if args.algorithm == "pso":
        cal = host.calibrate_pso(dict(iterations=args.pso_iterations, swarm_size=args.swarm_size, variant=args.pso_variant,
                                      topology=args.pso_topology, seed=args.seed),
                                 mh_seed=args.seed + 1, mh_iterations=args.mcmc_iterations, burn_in=args.burn_in,
                                 adaptation_period=args.adaptation_period, thinning=args.thinning, chains=args.chains)
    else:
        cal = host.calibrate(hc_seed=args.seed, mh_seed=args.seed + 1, hc_iterations=args.hc_iterations,
                           mh_iterations=args.mcmc_iterations, burn_in=args.burn_in,
                           cloud_size_multiplier=args.cloud_size_multiplier, threads=args.hc_threads,
                           adaptation_period=args.adaptation_period, thinning=args.thinning, chains=args.chains)
    t_cal = time.perf_counter() - t0
    for c in range(args.chains):
        mm.config_io.write_posterior_trace_csv(os.path.join(args.out, f"posterior_trace_chain{c}.csv"), cal["samples"][c],
                                               cal["sample_values"][c], list(pb.param_names))
    with open(os.path.join(args.out, "calibrated_parameters_final.txt"), "w") as fh:
        fh.write(f"# best objective value: {cal['best_value']:.8e}\n")
        for name, v in zip(pb.param_names, cal["best"]):
            fh.write(f"{name} {v:.10g}\n")

    # post-calibration ensemble over the pooled post-burn-in samples of every chain, from the initial state
    # of the problem as given (SimulationRunner::runSimulation)
    first = args.burn_in // max(1, args.thinning) + 1
    pooled = cal["samples"][:, first:, :].reshape(-1, pb.n_params)
    pooled = pooled[:16384]
    t0 = time.perf_counter()
    hip = mm.HipObjective(pb.with_(constraint_mode=1))
    hip.set_initial_state_mode(1)
    ens = hip.ensemble_quantiles(pooled, PROBS, want_sero=True, want_rt=True, want_metrics=True)
    t_ens = time.perf_counter() - t0
    times = np.asarray(pb.times)
    pos = times[times >= 0]
    # the reference's post-calibration output tree (file names, headers and number formats of AnalysisWriter.cpp; what
    # scripts/model/PostCalibrationAnalysis.py loads): posterior_predictive/, parameter_posteriors/, rt_trajectories/,
    # seroprevalence/, mcmc_batches/, mcmc_aggregated/
    observed = {"daily_hospitalizations": pb.obs_H, "daily_icu_admissions": pb.obs_ICU, "daily_deaths": pb.obs_D,
                "cumulative_hospitalizations": np.cumsum(pb.obs_H, axis=0), "cumulative_icu_admissions": np.cumsum(pb.obs_ICU, axis=0),
                "cumulative_deaths": np.cumsum(pb.obs_D, axis=0)}
    mm.config_io.write_post_calibration_tree(args.out, times, ens, pooled, list(pb.param_names), pb.n, observed=observed)
    assert len(pos) == ens["ppc"].shape[2]

    evals = args.chains * args.mcmc_iterations
    summary = {"initial_value": cal["initial_value"], "phase1_best_value": cal["phase1_best_value"],
               "best_value": cal["best_value"], "algorithm": args.algorithm, "chains": args.chains, "mcmc_iterations": args.mcmc_iterations,
               "acceptance_rate_mean": float(cal["accept_trace"].mean()), "calibration_seconds": t_cal,
               "proposals_per_s": evals / t_cal, "ensemble_samples": int(len(pooled)), "ensemble_valid": int(ens["n_valid"]),
               "ensemble_seconds": t_ens, "median_R0": float(np.nanmedian(ens["metrics"][:, 0])), "out": args.out}
    with open(os.path.join(args.out, "run_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
