"""Readers for the on-disk formats either side of the likelihood path.

These follow the reference's host-side file formats so that a calibration set up for
the reference (``data/configuration/*.txt``, ``data/contacts.csv``,
``data/processed/processed_data.csv``) can drive this build unchanged.  Host plumbing
only -- nothing here is on the hot path.

Reference behaviour followed (paths under /root/reference):
  * ``key value...`` text files with ``#`` comments:
    src/utils/ReadCalibrationConfiguration.cpp:164-271 (readSEPAIHRDParameters; ``beta_k`` /
    ``kappa_k`` are 1-based indexed scalars, the age vectors sit on one line),
    :273-305 (readParamBounds), :307-339 (readProposalSigmas), :341-… (readParamsToCalibrate)
  * contact matrix CSV: src/utils/ReadContactMatrix.cpp:8-83 (line i, field j -> M(i,j))
  * calibration CSV: src/utils/GetCalibrationData.cpp:236-401 (date-window filter by string
    comparison, the 4 fixed age bands, population from the first in-window row)
"""
from __future__ import annotations

import csv
import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

AGE_VECTORS = ("a", "h_infec", "p", "h", "icu", "d_H", "d_ICU", "d_community")
SCALARS = (
    "beta", "theta", "sigma", "gamma_p", "gamma_A", "gamma_I", "gamma_H", "gamma_ICU",
    "E0_multiplier", "P0_multiplier", "A0_multiplier", "I0_multiplier", "H0_multiplier",
    "ICU0_multiplier", "R0_multiplier", "D0_multiplier", "runup_days", "seed_exposed",
)
AGE_BANDS = ("0_30", "30_60", "60_80", "80_plus")


def _config_lines(path: str):
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            yield line.split()


def read_sepaihrd_parameters(path: str, num_age_classes: int) -> dict:
    """initial_guess.txt -> dict of scalars / vectors (readSEPAIHRDParameters)."""
    out: dict = {name: np.zeros(num_age_classes) for name in AGE_VECTORS}
    beta_map: Dict[int, float] = {}
    kappa_map: Dict[int, float] = {}
    for tok in _config_lines(path):
        name, vals = tok[0], [float(v) for v in tok[1:]]
        if not vals:
            continue
        if name.startswith("beta_") and name != "beta_end_times":
            beta_map[int(name[5:])] = vals[0]
        elif name.startswith("kappa_") and name != "kappa_end_times":
            kappa_map[int(name[6:])] = vals[0]
        elif name in SCALARS:
            out[name] = vals[0]
        elif name in ("beta_end_times", "kappa_end_times"):
            out[name] = np.array(vals)
        elif name in AGE_VECTORS:
            if len(vals) != num_age_classes:
                raise ValueError(f"{name}: expected {num_age_classes} values, got {len(vals)}")
            out[name] = np.array(vals)
    for key, mp in (("beta_values", beta_map), ("kappa_values", kappa_map)):
        if mp:
            arr = np.zeros(max(mp))
            for k, v in mp.items():
                arr[k - 1] = v
            out[key] = arr
        else:
            out[key] = np.zeros(0)
    out.setdefault("beta_end_times", np.zeros(0))
    out.setdefault("kappa_end_times", np.zeros(0))
    out.setdefault("beta", 0.0)
    return out


def read_param_bounds(path: str) -> Dict[str, Tuple[float, float]]:
    return {tok[0]: (float(tok[1]), float(tok[2])) for tok in _config_lines(path) if len(tok) >= 3}


def read_proposal_sigmas(path: str) -> Dict[str, float]:
    return {tok[0]: float(tok[1]) for tok in _config_lines(path) if len(tok) >= 2}


def read_params_to_calibrate(path: str) -> List[str]:
    return [tok[0] for tok in _config_lines(path)]


def read_settings(path: str) -> Dict[str, float]:
    """mcmc_settings.txt & friends -> {key: double}."""
    return {tok[0]: float(tok[1]) for tok in _config_lines(path) if len(tok) >= 2}


def read_matrix_csv(path: str, rows: int, cols: int) -> np.ndarray:
    with open(path, "r") as fh:
        data = [[float(v) for v in line.strip().split(",")] for line in fh if line.strip()]
    m = np.array(data, dtype=np.float64)
    if m.shape != (rows, cols):
        raise ValueError(f"contact matrix shape {m.shape}, expected {(rows, cols)}")
    return m


@dataclass
class CalibrationData:
    """The matrices CalibrationData holds after readCSVData (T_obs x 4, row-major)."""
    dates: List[str]
    new_confirmed: np.ndarray
    new_deaths: np.ndarray
    new_hospitalizations: np.ndarray
    new_icu: np.ndarray
    cumulative_confirmed: np.ndarray
    cumulative_deaths: np.ndarray
    cumulative_hospitalizations: np.ndarray
    cumulative_icu: np.ndarray
    population: np.ndarray

    @property
    def num_data_points(self) -> int:
        return len(self.dates)


def read_calibration_csv(path: str, start_date: str, end_date: str) -> CalibrationData:
    cols = {
        "new_confirmed": "new_confirmed_", "new_deaths": "new_deceased_",
        "new_hospitalizations": "new_hospitalized_patients_", "new_icu": "new_intensive_care_patients_",
        "cumulative_confirmed": "cumulative_confirmed_", "cumulative_deaths": "cumulative_deceased_",
        "cumulative_hospitalizations": "cumulative_hospitalized_patients_",
        "cumulative_icu": "cumulative_intensive_care_patients_",
    }
    acc: Dict[str, list] = {k: [] for k in cols}
    dates: List[str] = []
    population = None
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            d = row["date"]
            if start_date and d < start_date:
                continue
            if end_date and d > end_date:
                continue
            dates.append(d)
            for key, prefix in cols.items():
                acc[key].append([float(row[prefix + band]) for band in AGE_BANDS])
            if population is None:
                population = np.array([float(row["population_" + band]) for band in AGE_BANDS])
    if not dates:
        raise ValueError("no data points in the requested date range")
    return CalibrationData(dates=dates, population=population,
                           **{k: np.array(v, dtype=np.float64) for k, v in acc.items()})


def initial_sepaihrd_state(data: CalibrationData, sigma: float, gamma_p: float, gamma_a: float,
                           gamma_i: float, p_asym: np.ndarray, h_hosp: np.ndarray) -> np.ndarray:
    """CalibrationData::getInitialSEPAIHRDState (src/utils/GetCalibrationData.cpp:107-234).

    Heuristic 11n-vector anchored on the cumulative observations of day 0; it matters only
    in the objective's *multiplier* branch (run-up disabled).
    """
    n = len(data.population)
    N = data.population
    D0 = np.maximum(data.cumulative_deaths[0], 0.0)
    H0 = np.maximum(data.cumulative_hospitalizations[0], 0.0)
    ICU0 = np.maximum(data.cumulative_icu[0], 0.0)
    CumH0, CumICU0 = H0.copy(), ICU0.copy()
    I0 = np.maximum(data.cumulative_confirmed[0] - D0, 0.0)
    E0, P0, A0, R0 = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    for i in range(n):
        p_i = min(max(p_asym[i], 0.0), 1.0)
        omp = 1.0 - p_i
        P0[i] = I0[i] * gamma_i / (omp * gamma_p) if (gamma_p > 1e-9 and omp > 1e-9) else I0[i]
        A0[i] = P0[i] * p_i * gamma_p / gamma_a if gamma_a > 1e-9 else P0[i] * p_i
        E0[i] = P0[i] * gamma_p / sigma if sigma > 1e-9 else P0[i]
    E0, P0, A0 = np.maximum(E0, 0.0), np.maximum(P0, 0.0), np.maximum(A0, 0.0)
    for i in range(n):
        D0[i] = min(D0[i], N[i])
        ICU0[i] = min(ICU0[i], max(0.0, N[i] - D0[i]))
        H0[i] = min(H0[i], max(0.0, N[i] - D0[i] - ICU0[i]))
        I0[i] = min(I0[i], max(0.0, N[i] - D0[i] - ICU0[i] - H0[i]))
        R0[i] = min(R0[i], max(0.0, N[i] - D0[i] - ICU0[i] - H0[i] - I0[i]))
    for i in range(n):
        set_sum = I0[i] + H0[i] + ICU0[i] + R0[i] + D0[i]
        inferred = E0[i] + P0[i] + A0[i]
        avail = max(N[i] - set_sum, 0.0)
        if inferred > avail:
            scale = avail / inferred if inferred > 1e-9 else 0.0
            E0[i] *= scale
            P0[i] *= scale
            A0[i] *= scale
    state = np.zeros(11 * n)
    for c, v in ((1, E0), (2, P0), (3, A0), (4, I0), (5, H0), (6, ICU0), (7, R0), (8, D0),
                 (9, CumH0), (10, CumICU0)):
        state[c * n:(c + 1) * n] = v
    for i in range(n):
        s = 0.0
        for j in range(1, 9):
            s += state[j * n + i]
        state[i] = max(0.0, N[i] - s)
    return state


def write_posterior_trace_csv(path: str, samples: np.ndarray, values: np.ndarray, names: List[str]) -> None:
    """posterior_trace*.csv as MetropolisHastingsSampler::saveSamplesToCSV writes it
    (src/sir_age_structured/optimizers/MetropolisHastingsSampler.cpp:414-438):
    ``iter,log_posterior,<names...>`` with ``std::scientific << setprecision(6)``."""
    with open(path, "w") as fh:
        fh.write("iter,log_posterior" + "".join("," + n for n in names) + "\n")
        for i in range(len(values)):
            fh.write(str(i) + "," + "%.6e" % values[i] + "".join(",%.6e" % v for v in samples[i]) + "\n")


# ---------------------------------------------------------------------------------------------------------------
# Post-calibration output tree, in the layout and column names the reference writes and its plotting script reads
#   writer    src/model/AnalysisWriter.cpp:201-283 (parameter posteriors), :285-345 (posterior predictive),
#             :349-398 (batch metrics), :400-437 (metrics summary), :512-540 (aggregated trajectories)
#   layout    src/model/PostCalibrationAnalyser.cpp:64-90,253,283-343
#   consumer  scripts/model/PostCalibrationAnalysis.py:98-133,169-170,216,271,322,358
# ---------------------------------------------------------------------------------------------------------------
PPC_SERIES = ["daily_hospitalizations", "daily_icu_admissions", "daily_deaths", "cumulative_hospitalizations",
              "cumulative_icu_admissions", "cumulative_deaths"]
PPC_PROBS = [0.025, 0.05, 0.5, 0.95, 0.975]           # lower95, lower90, median, upper90, upper95
_PPC_SUFFIX = ["lower95", "lower90", "median", "upper90", "upper95"]
ESSENTIAL_METRICS = ["R0", "overall_IFR", "overall_attack_rate", "peak_hospital", "peak_ICU", "time_to_peak_hospital",
                     "time_to_peak_ICU", "total_deaths", "max_Rt", "min_Rt", "final_Rt", "seroprevalence_day64"]


def _cxx_default(v: float) -> str:
    """operator<<(double) with the stream defaults (%g, 6 significant digits): how the reference prints times."""
    return "%g" % v


def _quantile_sorted(values: np.ndarray, q: float) -> float:
    pos = q * (len(values) - 1)                       # PostCalibrationAnalyser.cpp:316-326
    idx = int(pos)
    frac = pos - idx
    return float(values[idx] * (1.0 - frac) + values[idx + 1] * frac) if idx + 1 < len(values) else float(values[idx])


def write_posterior_predictive(out_dir: str, times_pos, ppc: np.ndarray, observed: Dict[str, np.ndarray]) -> List[str]:
    """posterior_predictive/<series>_{median,lower90,upper90,lower95,upper95,observed}.csv: ``time,age_0,...`` and
    ``std::fixed << setprecision(6)`` values.  ppc: [6 series][5 PPC_PROBS][T_pos][n]; observed: series -> [T_pos][n]
    (series without an entry get no _observed file)."""
    os.makedirs(out_dir, exist_ok=True)
    written = []
    n = ppc.shape[3]
    header = "time" + "".join(f",age_{a}" for a in range(n)) + "\n"

    def dump(path, mat):
        with open(path, "w") as fh:
            fh.write(header)
            for ti, t in enumerate(times_pos):
                fh.write(_cxx_default(t) + "".join(",%.6f" % mat[ti, a] for a in range(n)) + "\n")
        written.append(path)

    for si, name in enumerate(PPC_SERIES):
        for pi, suffix in enumerate(_PPC_SUFFIX):
            dump(os.path.join(out_dir, f"{name}_{suffix}.csv"), ppc[si, pi])
        if name in observed:
            dump(os.path.join(out_dir, f"{name}_observed.csv"), np.asarray(observed[name])[:len(times_pos)])
    return written


def write_aggregated_trajectory(path: str, times, quantiles: np.ndarray) -> None:
    """rt_trajectories/Rt_aggregated_with_uncertainty.csv, seroprevalence/seroprevalence_trajectory.csv:
    ``time,median,q025,q975,q05,q95`` fixed 6 digits.  quantiles: [5 PPC_PROBS][T]."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    q025, q05, med, q95, q975 = quantiles
    with open(path, "w") as fh:
        fh.write("time,median,q025,q975,q05,q95\n")
        for ti, t in enumerate(times):
            fh.write("%.6f,%.6f,%.6f,%.6f,%.6f,%.6f\n" % (t, med[ti], q025[ti], q975[ti], q05[ti], q95[ti]))


def write_parameter_posteriors(out_dir: str, samples: np.ndarray, names: List[str], burn_in: int = 0, thinning: int = 1) -> None:
    """parameter_posteriors/posterior_samples.csv (``sample_index,<names>``, scientific 8 digits, rows burn_in,
    burn_in + thinning, ...) and posterior_summary.csv (``parameter,mean,median,std_dev,lower_95_ci,upper_95_ci``,
    fixed 8 digits; median = sorted[n/2], bounds = sorted[int(q n)], population std: the writer's own rules)."""
    os.makedirs(out_dir, exist_ok=True)
    kept = np.asarray(samples)[burn_in::max(1, thinning)]
    with open(os.path.join(out_dir, "posterior_samples.csv"), "w") as fh:
        fh.write("sample_index" + "".join("," + nm for nm in names) + "\n")
        for i, row in enumerate(kept):
            fh.write(str(i) + "".join(",%.8e" % v for v in row) + "\n")
    with open(os.path.join(out_dir, "posterior_summary.csv"), "w") as fh:
        fh.write("parameter,mean,median,std_dev,lower_95_ci,upper_95_ci\n")
        for p, nm in enumerate(names):
            v = np.sort(kept[:, p])
            if len(v) == 0:
                continue
            mean = float(np.sum(v) / len(v))
            std = float(np.sqrt(np.sum((v - mean) ** 2) / len(v)))
            fh.write("%s,%.8f,%.8f,%.8f,%.8f,%.8f\n" % (nm, mean, v[len(v) // 2], std, v[int(0.025 * len(v))],
                                                        v[min(int(0.975 * len(v)), len(v) - 1)]))


def essential_metric_columns(n_age: int) -> List[str]:
    return ESSENTIAL_METRICS + [f"{m}_age_{a}" for a in range(n_age) for m in ("IFR", "IHR", "IICUR", "AttackRate")]


def write_batch_metrics(path: str, metrics: np.ndarray, n_age: int, kappa: Dict[str, np.ndarray] | None = None) -> None:
    """mcmc_batches/batch_k.csv: ``sample_idx,R0,...,seroprevalence_day64,IFR_age_0,IHR_age_0,IICUR_age_0,
    AttackRate_age_0,...[,kappa_1,...]``, values with the stream defaults.  metrics: [S][12 + 4 n] as
    sepaihrd_ensemble_quantiles returns them; NaN rows (skipped samples) are left out like the reference's
    ``if (!sim_result.isValid()) continue``."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    cols = essential_metric_columns(n_age)
    kappa = kappa or {}
    with open(path, "w") as fh:
        fh.write("sample_idx," + ",".join(cols + list(kappa)) + "\n")
        i = 0
        for s, row in enumerate(np.asarray(metrics)):
            if not np.all(np.isfinite(row[:1])):
                continue
            fh.write(str(i) + "".join("," + _cxx_default(v) for v in row) +
                     "".join("," + _cxx_default(kappa[k][s]) for k in kappa) + "\n")
            i += 1


def write_metrics_summary(path: str, metrics: np.ndarray, n_age: int) -> None:
    """mcmc_aggregated/metrics_summary.csv: ``metric,mean,median,std_dev,q025,q975`` fixed 8 digits, one row per
    metric in map (alphabetical) order -- the index the plotting script looks ``IFR_age_j`` up in.  Exact
    sort-based quantiles over the valid rows (the reference pools per-batch P-square estimates)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    m = np.asarray(metrics)
    m = m[np.isfinite(m[:, 0])]
    cols = essential_metric_columns(n_age)
    with open(path, "w") as fh:
        fh.write("metric,mean,median,std_dev,q025,q975\n")
        for name in sorted(cols):
            v = np.sort(m[:, cols.index(name)])
            if len(v) == 0:
                continue
            mean = float(np.mean(v))
            fh.write("%s,%.8f,%.8f,%.8f,%.8f,%.8f\n" % (name, mean, _quantile_sorted(v, 0.5), float(np.sqrt(np.mean((v - mean) ** 2))),
                                                        _quantile_sorted(v, 0.025), _quantile_sorted(v, 0.975)))


def write_post_calibration_tree(out_base: str, times, ensemble: dict, samples: np.ndarray, names: List[str], n_age: int,
                                observed: Dict[str, np.ndarray] | None = None, burn_in: int = 0, thinning: int = 1) -> None:
    """Everything PostCalibrationAnalysis.py loads (except the scenario comparison, which is not on this path), under
    out_base, from one sepaihrd_ensemble_quantiles result (keys ppc, sero, rt, metrics; quantiles at PPC_PROBS)."""
    times = np.asarray(times, dtype=np.float64)
    write_posterior_predictive(os.path.join(out_base, "posterior_predictive"), times[times >= 0], ensemble["ppc"], observed or {})
    write_parameter_posteriors(os.path.join(out_base, "parameter_posteriors"), samples, names, burn_in, thinning)
    if ensemble.get("rt") is not None:
        write_aggregated_trajectory(os.path.join(out_base, "rt_trajectories", "Rt_aggregated_with_uncertainty.csv"), times, ensemble["rt"])
    if ensemble.get("sero") is not None:
        write_aggregated_trajectory(os.path.join(out_base, "seroprevalence", "seroprevalence_trajectory.csv"), times, ensemble["sero"])
    if ensemble.get("metrics") is not None:
        write_batch_metrics(os.path.join(out_base, "mcmc_batches", "batch_0.csv"), ensemble["metrics"], n_age)
        write_metrics_summary(os.path.join(out_base, "mcmc_aggregated", "metrics_summary.csv"), ensemble["metrics"], n_age)
