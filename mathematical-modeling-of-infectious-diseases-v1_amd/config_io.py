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
import re
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


class CSVReadError(ValueError):
    """epidemic::CSVReadException (include/exceptions/CSVReadException.hpp): `kind` is the name of its ErrorType."""

    def __init__(self, kind: str, message: str):
        super().__init__(f"{kind}: {message}")
        self.kind = kind


class FileIOError(OSError):
    """epidemic::FileIOException."""


class DataFormatError(ValueError):
    """epidemic::DataFormatException."""


_STOD_PREFIX = re.compile(r"\s*[+-]?(?:(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?|inf(?:inity)?|nan)", re.IGNORECASE)


def _stod(cell: str) -> float:
    """std::stod: leading whitespace, then the longest numeric prefix; ValueError when there is none."""
    m = _STOD_PREFIX.match(cell)
    if not m:
        raise ValueError(cell)
    return float(m.group(0))


def read_matrix_csv(path: str, rows: int, cols: int) -> np.ndarray:
    """readMatrixFromCSV (src/utils/ReadContactMatrix.cpp:8-83): leading ``//`` lines are skipped, the next line is row 1
    (line i, field j -> M(i, j)), empty lines between later rows are skipped, fields and rows beyond rows x cols are
    not looked at.  Errors carry the reference's CSVReadException::ErrorType as `kind`: FileOpenError, NotEnoughRows,
    NotEnoughColumns, InvalidNumberFormat (tests/utils/ReadContactMatrixTests.cpp:57-137)."""
    try:
        with open(path, "rb") as fh:
            lines = fh.read().decode("utf-8", "replace").split("\n")
    except OSError:
        raise CSVReadError("FileOpenError", path) from None
    if lines and lines[-1] == "":
        lines.pop()  # std::getline does not deliver an empty last line after the final newline
    k = 0
    while k < len(lines) and lines[k] != "" and lines[k][:2] == "//":
        k += 1
    if k >= len(lines):
        raise CSVReadError("NotEnoughRows", "No data rows found in file: " + path)
    mat = np.zeros((rows, cols))

    def parse_row(i: int, line: str) -> None:
        cells = line.split(",") if line != "" else []
        if line.endswith(","):
            cells.pop()  # std::getline(ss, cell, ',') does not deliver an empty field behind the last comma
        for j in range(cols):
            if j >= len(cells):
                raise CSVReadError("NotEnoughColumns", f"row {i + 1} in {path}")
            try:
                mat[i, j] = _stod(cells[j])
            except ValueError:
                raise CSVReadError("InvalidNumberFormat", f"row {i + 1}, column {j + 1}: '{cells[j]}' in {path}") from None

    parse_row(0, lines[k])
    k += 1
    i = 1
    while i < rows:
        if k >= len(lines):
            raise CSVReadError("NotEnoughRows", f"expected {rows} rows, found {i} in {path}")
        line = lines[k]
        k += 1
        if line == "":
            continue
        parse_row(i, line)
        i += 1
    return mat


def join_paths(path1: str, path2: str) -> str:
    """FileUtils::joinPaths (src/utils/FileUtils.cpp:62-72): one leading '/' of path2 dropped, then
    (path1 / path2).lexically_normal()."""
    if path2 == "":
        return path1
    rel = path2[1:] if path2.startswith("/") else path2
    if path1 == "":
        joined = rel
    elif path1.endswith("/"):
        joined = path1 + rel
    else:
        joined = path1 + "/" + rel
    if joined == "":
        return ""
    out = os.path.normpath(joined)
    # lexically_normal keeps a trailing separator as "dir/"; normpath drops it, and turns "" into "."
    if joined.endswith("/") and out != "/":
        out += "/"
    return out


FILEUTILS_AGE_VECTORS = ("p", "h", "icu", "d_H", "d_ICU")
FILEUTILS_SCALARS = ("beta", "theta", "sigma", "gamma_p", "gamma_A", "gamma_I", "gamma_H", "gamma_ICU",
                     "contact_matrix_scaling_factor")
_ISTREAM_DOUBLE = re.compile(r"[+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?")


class _IStream:
    """The part of std::istringstream these readers use: `>> word` and `>> double` (whitespace skipped, the longest
    numeric prefix taken, the rest left in the stream)."""

    def __init__(self, text: str):
        self.s, self.pos = text, 0

    def _skip(self) -> None:
        while self.pos < len(self.s) and self.s[self.pos] in " \t\n\r\f\v":
            self.pos += 1

    def at_end(self) -> bool:
        self._skip()
        return self.pos >= len(self.s)

    def word(self):
        self._skip()
        start = self.pos
        while self.pos < len(self.s) and self.s[self.pos] not in " \t\n\r\f\v":
            self.pos += 1
        return self.s[start:self.pos] or None

    def double(self):
        self._skip()
        m = _ISTREAM_DOUBLE.match(self.s, self.pos)
        if not m:
            return None
        self.pos = m.end()
        return float(m.group(0))


def read_sepaihrd_parameters_fileutils(path: str, num_age_classes: int) -> dict:
    """FileUtils::readSEPAIHRDParameters (src/utils/FileUtils.cpp:74-196), the older of the reference's two parameter
    readers and the one its tests pin (tests/utils/FileUtilsTests.cpp:146-320): ``name value...`` lines, `#` starts a
    comment line, the age vectors p / h / icu / d_H / d_ICU take exactly num_age_classes values, kappa_end_times /
    kappa_values take any number, everything else exactly one; the last occurrence of a name wins; unknown scalar names
    are ignored with a warning.  Errors: FileIOError (file cannot be opened) and DataFormatError with the reference's
    message texts."""
    where = "FileUtils::readSEPAIHRDParameters"
    try:
        fh = open(path, "r")
    except OSError:
        raise FileIOError(f"{where}: Unable to open parameters file: {path}") from None
    out: dict = {name: np.zeros(num_age_classes) for name in FILEUTILS_AGE_VECTORS}
    out["kappa_end_times"], out["kappa_values"] = [], []
    with fh:
        for line_number, raw in enumerate(fh, start=1):
            line = raw.strip(" \t\n\r\f\v")
            if not line or line[0] == "#":
                continue
            iss = _IStream(line)
            name = iss.word()
            if name in FILEUTILS_AGE_VECTORS:
                for i in range(num_age_classes):
                    v = iss.double()
                    if v is None:
                        raise DataFormatError(f"{where}: Error reading value for age class {i} of parameter '{name}' on line {line_number}")
                    out[name][i] = v
                if iss.double() is not None:
                    raise DataFormatError(f"{where}: Too many values provided for age-specific parameter '{name}' on line "
                                          f"{line_number}. Expected {num_age_classes} values.")
            elif name in ("kappa_end_times", "kappa_values"):
                vals = []
                while True:
                    v = iss.double()
                    if v is None:
                        break
                    vals.append(v)
                if not iss.at_end():  # iss.fail() && !iss.eof()
                    raise DataFormatError(f"{where}: Invalid non-numeric data found for '{name}' on line {line_number}")
                out[name] = vals
            else:
                v = iss.double()
                if v is None:
                    found = iss.word()
                    if found is None:
                        raise DataFormatError(f"{where}: Missing scalar value for parameter '{name}' on line {line_number}")
                    raise DataFormatError(f"{where}: Error reading scalar value for parameter '{name}' on line {line_number}. Found: '{found}'")
                if iss.double() is not None:
                    raise DataFormatError(f"{where}: Too many values provided for scalar parameter '{name}' on line {line_number}. Expected 1 value.")
                if name in FILEUTILS_SCALARS:
                    out[name] = v
    if out["kappa_end_times"] and out["kappa_values"] and len(out["kappa_end_times"]) != len(out["kappa_values"]):
        raise DataFormatError(f"{where}: Mismatch between number of kappa_end_times ({len(out['kappa_end_times'])}) and "
                              f"kappa_values ({len(out['kappa_values'])}) read from file: {path}")
    return out


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
    num_age_classes: int = 4

    @property
    def num_data_points(self) -> int:
        return len(self.dates)

    @classmethod
    def from_matrices(cls, new_confirmed, new_hospitalizations, new_icu, new_deaths, population, cum_confirmed0, cum_deaths0,
                      cum_hospitalizations0, cum_icu0, num_age_classes: int) -> "CalibrationData":
        """The reference's matrix constructor (src/utils/GetCalibrationData.cpp:24-89; argument order as there): the
        cumulative matrices start at the given day-0 rows and add the PREVIOUS day's increments, dates are
        ``mock_date_i``; size mismatches are std::invalid_argument (ValueError)."""
        n = int(num_age_classes)
        mats = [np.asarray(m, dtype=np.float64).reshape(-1, n) if np.asarray(m).size else np.zeros((0, n))
                for m in (new_confirmed, new_hospitalizations, new_icu, new_deaths)] if n > 0 else []
        if n <= 0:
            raise ValueError("Number of age classes must be positive.")
        pop = np.asarray(population, dtype=np.float64).ravel()
        if pop.size != n:
            raise ValueError("Population vector size mismatch with num_age_classes.")
        for m in (new_confirmed, new_hospitalizations, new_icu, new_deaths):
            m = np.asarray(m)
            if m.ndim != 2 or m.shape[1] != n:
                raise ValueError("Incidence data matrix column count mismatch with num_age_classes.")
        row0 = [np.asarray(v, dtype=np.float64).ravel() for v in (cum_confirmed0, cum_deaths0, cum_hospitalizations0, cum_icu0)]
        if any(v.size != n for v in row0):
            raise ValueError("Initial cumulative data vector size mismatch with num_age_classes.")
        new_c, new_h, new_i, new_d = mats
        T = new_c.shape[0]

        def cumulative(new, first):
            out = np.zeros((T, n))
            for i in range(T):
                if i == 0:
                    out[0] = first
                elif i - 1 < new.shape[0]:
                    out[i] = out[i - 1] + new[i - 1]
                else:
                    out[i] = out[i - 1]
            return out

        return cls(dates=[f"mock_date_{i}" for i in range(T)], new_confirmed=new_c, new_deaths=new_d,
                   new_hospitalizations=new_h, new_icu=new_i, cumulative_confirmed=cumulative(new_c, row0[0]),
                   cumulative_deaths=cumulative(new_d, row0[1]), cumulative_hospitalizations=cumulative(new_h, row0[2]),
                   cumulative_icu=cumulative(new_i, row0[3]), population=pop, num_age_classes=n)

    def initial_active_cases(self) -> np.ndarray:
        """getInitialActiveCases (:101-106): row 0 of the cumulative confirmed cases."""
        if self.cumulative_confirmed.shape[0] == 0:
            raise RuntimeError("Cannot get initial active cases: cumulative_confirmed_cases data is empty.")
        return self.cumulative_confirmed[0].copy()


CSV_COLUMN_PREFIXES = {
    "new_confirmed": "new_confirmed_", "new_deaths": "new_deceased_",
    "new_hospitalizations": "new_hospitalized_patients_", "new_icu": "new_intensive_care_patients_",
    "cumulative_confirmed": "cumulative_confirmed_", "cumulative_deaths": "cumulative_deceased_",
    "cumulative_hospitalizations": "cumulative_hospitalized_patients_",
    "cumulative_icu": "cumulative_intensive_care_patients_",
}


def read_calibration_csv(path: str, start_date: str = "", end_date: str = "") -> CalibrationData:
    """CalibrationData(filename, start, end) (src/utils/GetCalibrationData.cpp:15-22,236-401): columns found by NAME in the
    header (a missing one is a runtime_error naming it), rows kept when start <= date <= end as strings (an empty bound is
    open), population from the first kept row; an unreadable file, a header-only file or an empty date window is the
    constructor's runtime_error "Failed to initialize CalibrationData from file"."""
    fail = RuntimeError("Failed to initialize CalibrationData from file: " + path)
    try:
        fh = open(path, newline="")
    except OSError:
        raise fail from None
    with fh:
        header = fh.readline()
        if header == "":
            raise fail
        index = {}
        for k, name in enumerate(header.rstrip("\r\n").split(",")):
            index[name] = k  # a repeated name: the last one wins (std::map assignment)
        def col(name):
            if name not in index:
                raise RuntimeError("Missing required column: " + name)
            return index[name]
        date_idx = col("date")
        idx = {key: [col(prefix + band) for band in AGE_BANDS] for key, prefix in CSV_COLUMN_PREFIXES.items()}
        pop_idx = [col("population_" + band) for band in AGE_BANDS]
        need = max([date_idx] + pop_idx + [i for v in idx.values() for i in v]) + 1
        acc: Dict[str, list] = {k: [] for k in idx}
        dates: List[str] = []
        population = None
        for raw in fh:
            line = raw.rstrip("\r\n")
            if line == "":
                continue
            row = line.split(",")
            if line.endswith(","):
                row.pop()
            d = row[date_idx] if date_idx < len(row) else row[-1]
            if start_date and d < start_date:
                continue
            if end_date and d > end_date:
                continue
            if len(row) < need:
                raise fail  # "Insufficient columns in data row": readCSVData returns false
            dates.append(d)
            try:
                for key in idx:
                    acc[key].append([float(row[i]) for i in idx[key]])
                if population is None:
                    population = np.array([float(row[i]) for i in pop_idx])
            except ValueError as e:
                raise RuntimeError("Failed to parse value: " + str(e)) from None
    if not dates:
        raise fail
    return CalibrationData(dates=dates, population=population,
                           **{k: np.array(v, dtype=np.float64) for k, v in acc.items()})


def initial_sepaihrd_state(data: CalibrationData, sigma: float, gamma_p: float, gamma_a: float,
                           gamma_i: float, p_asym: np.ndarray, h_hosp: np.ndarray) -> np.ndarray:
    """CalibrationData::getInitialSEPAIHRDState (src/utils/GetCalibrationData.cpp:107-234).

    Heuristic 11n-vector anchored on the cumulative observations of day 0; it matters only
    in the objective's *multiplier* branch (run-up disabled).  The reference's checks (:115-127) are runtime_errors.
    """
    n = len(data.population)
    if data.num_data_points == 0:
        raise RuntimeError("Cannot get initial SEPAIHRD state: No data points loaded.")
    if len(p_asym) != n:
        raise RuntimeError("p_asymptomatic vector size mismatch with num_age_classes.")
    if len(h_hosp) != n:
        raise RuntimeError("h_hospitalization vector size mismatch with num_age_classes.")
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
