"""Problem description for the SEPAIHRD likelihood path (host-side plumbing).

``SEPAIHRDProblem`` carries exactly what ``SEPAIHRDObjectiveFunction`` is constructed from
in the reference (src/model/objectives/SEPAIHRDObjectiveFunction.cpp:22-50): the model's
baseline ``SEPAIHRDParameters`` (include/model/parameters/SEPAIHRDParameters.hpp:20-124),
the parameter manager's names / sigmas / bounds
(src/model/parameters/SEPAIHRDParameterManager.cpp:13-89), the output grid, the initial
state and the three observed matrices.

``resolve_param_name`` is the one-off, host-side replacement for the per-evaluation string
dispatch of SEPAIHRDParameterManager::updateModelParameters (:197-267): names become
(field code, index) pairs of include/sepaihrd_hip.h.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field, replace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

NUM_COMPARTMENTS = 11
SOLVER_DOPRI5, SOLVER_CASH_KARP54 = 0, 1
CONSTRAINT_CLAMP, CONSTRAINT_REFLECT = 0, 1
ARITH_STRICT, ARITH_FMA = 0, 1
PRECISION_F64, PRECISION_F32 = 0, 1  # number type of the ODE state (include/sepaihrd_hip.h)

# enum sepaihrd_field (include/sepaihrd_hip.h)
F_NONE = -1
F_SCALARS = {
    "beta": 0, "theta": 1, "sigma": 2, "gamma_p": 3, "gamma_A": 4, "gamma_I": 5, "gamma_H": 6,
    "gamma_ICU": 7, "E0_multiplier": 8, "P0_multiplier": 9, "A0_multiplier": 10, "I0_multiplier": 11,
    "H0_multiplier": 12, "ICU0_multiplier": 13, "R0_multiplier": 14, "D0_multiplier": 15,
    "runup_days": 16, "seed_exposed": 17,
}
F_BETA_VALUE, F_KAPPA_VALUE = 18, 19
# prefix dispatch ORDER matters ("h_infec_" before "h_"): PM.cpp:221-228
F_AGE_PREFIXES = (("a_", 20), ("h_infec_", 21), ("p_", 22), ("h_", 23), ("icu_", 24), ("d_H_", 25),
                  ("d_ICU_", 26), ("d_community_", 27))


def resolve_param_name(name: str, npi_names: Sequence[str], n_age: int, n_beta: int) -> Tuple[int, int]:
    """name -> (field, index), in the reference's if/else-if order (PM.cpp:201-266)."""
    if name == "beta":
        return F_SCALARS["beta"], 0
    if name.startswith("beta_"):
        idx = int(name[5:]) - 1  # std::stoul(name.substr(5)) - 1
        if not 0 <= idx < n_beta:
            raise ValueError(f"Beta index out of range for name: {name}")
        return F_BETA_VALUE, idx
    for key in ("theta", "sigma", "gamma_p", "gamma_A", "gamma_I", "gamma_H", "gamma_ICU"):
        if name == key:
            return F_SCALARS[key], 0
    for prefix, code in F_AGE_PREFIXES:
        if name.startswith(prefix):
            idx = int(name[len(prefix):])
            if code == 27 and idx >= n_age:  # d_community_: silently ignored when out of range
                return F_NONE, 0
            if not 0 <= idx < n_age:
                raise ValueError(f"Invalid age index for parameter {name}")
            return code, idx
    for key in ("seed_exposed", "runup_days", "E0_multiplier", "P0_multiplier", "A0_multiplier",
                "I0_multiplier", "H0_multiplier", "ICU0_multiplier", "R0_multiplier", "D0_multiplier"):
        if name == key:
            return F_SCALARS[key], 0
    if name.startswith("kappa_"):
        for k, nm in enumerate(npi_names):  # first match among the strategy's calibratable names
            if nm == name:
                return F_KAPPA_VALUE, k + 1
        return F_NONE, 0  # "[Warning] NPI param not found as calibratable"
    return F_NONE, 0  # "[Warning] Unknown parameter name"


def _arr(x, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


@dataclass
class SEPAIHRDProblem:
    # model (SEPAIHRDParameters)
    N: np.ndarray
    M: np.ndarray  # n x n, M[i, j] = contacts of i with j (row i of contacts.csv)
    a: np.ndarray
    h_infec: np.ndarray
    p: np.ndarray
    h: np.ndarray
    icu: np.ndarray
    d_H: np.ndarray
    d_ICU: np.ndarray
    d_community: np.ndarray
    theta: float
    sigma: float
    gamma_p: float
    gamma_A: float
    gamma_I: float
    gamma_H: float
    gamma_ICU: float
    kappa_end_times: np.ndarray  # baseline period first
    kappa_values: np.ndarray
    beta: float = 0.0
    beta_end_times: np.ndarray = field(default_factory=lambda: np.zeros(0))
    beta_values: np.ndarray = field(default_factory=lambda: np.zeros(0))
    multipliers: np.ndarray = field(default_factory=lambda: np.ones(8))  # E0,P0,A0,I0,H0,ICU0,R0,D0
    runup_days: float = 30.0
    seed_exposed: float = 10.0
    # objective
    times: np.ndarray = field(default_factory=lambda: np.zeros(0))
    initial_state: np.ndarray = field(default_factory=lambda: np.zeros(0))
    obs_H: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    obs_ICU: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    obs_D: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))
    # parameter manager
    param_names: List[str] = field(default_factory=list)
    sigmas: Dict[str, float] = field(default_factory=dict)
    bounds: Dict[str, Tuple[float, float]] = field(default_factory=dict)
    npi_names: Optional[List[str]] = None  # names of kappa_values[1:], default kappa_2..
    constraint_mode: int = CONSTRAINT_REFLECT
    # solver
    solver: int = SOLVER_DOPRI5
    abs_err: float = 1e-6
    rel_err: float = 1e-6
    dt_hint: float = 1.0
    arith: int = ARITH_STRICT
    precision: int = PRECISION_F64
    max_attempts: int = 0  # build-side guard on RK step attempts per chain, 0 = default (1e6)
    # calibrated starting point (getCurrentParameters of the shipped model)
    base_theta: Optional[np.ndarray] = None

    def __post_init__(self):
        self.N = _arr(self.N)
        n = self.n
        self.M = _arr(self.M, (n, n))
        for name in ("a", "h_infec", "p", "h", "icu", "d_H", "d_ICU", "d_community"):
            setattr(self, name, _arr(getattr(self, name), (n,)))
        for name in ("kappa_end_times", "kappa_values", "beta_end_times", "beta_values", "times"):
            setattr(self, name, _arr(getattr(self, name)))
        self.multipliers = _arr(self.multipliers, (8,))
        self.initial_state = _arr(self.initial_state, (NUM_COMPARTMENTS * n,))
        for name in ("obs_H", "obs_ICU", "obs_D"):
            setattr(self, name, _arr(getattr(self, name)).reshape(-1, n))
        if self.npi_names is None:
            self.npi_names = [f"kappa_{k + 2}" for k in range(len(self.kappa_values) - 1)]
        if len(self.npi_names) != len(self.kappa_values) - 1:
            raise ValueError("npi_names must name kappa_values[1:]")
        if self.base_theta is not None:
            self.base_theta = _arr(self.base_theta, (len(self.param_names),))

    # ---- sizes
    @property
    def n(self) -> int:
        return int(self.N.shape[0])

    @property
    def n_params(self) -> int:
        return len(self.param_names)

    @property
    def n_times(self) -> int:
        return int(self.times.shape[0])

    @property
    def n_obs(self) -> int:
        return int(self.obs_H.shape[0])

    @property
    def runup_offset(self) -> int:
        idx = np.nonzero(self.times >= 0.0)[0]
        return int(idx[0]) if idx.size else 0

    # ---- parameter-manager views
    def field_map(self) -> Tuple[np.ndarray, np.ndarray]:
        codes, idxs = [], []
        for nm in self.param_names:
            c, i = resolve_param_name(nm, self.npi_names, self.n, len(self.beta_values))
            codes.append(c)
            idxs.append(i)
        return np.array(codes, dtype=np.int32), np.array(idxs, dtype=np.int32)

    def bounds_arrays(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        lo = np.zeros(self.n_params)
        hi = np.zeros(self.n_params)
        has = np.zeros(self.n_params, dtype=np.uint8)
        for k, nm in enumerate(self.param_names):
            if nm in self.bounds:
                lo[k], hi[k] = self.bounds[nm]
                has[k] = 1
        return lo, hi, has

    def sigma_array(self) -> np.ndarray:
        return np.array([self.sigmas.get(nm, 0.0) for nm in self.param_names], dtype=np.float64)

    def current_parameters(self) -> np.ndarray:
        """SEPAIHRDParameterManager::getCurrentParameters (PM.cpp:91-158)."""
        codes, idxs = self.field_map()
        scal = {0: self.beta, 1: self.theta, 2: self.sigma, 3: self.gamma_p, 4: self.gamma_A,
                5: self.gamma_I, 6: self.gamma_H, 7: self.gamma_ICU, 16: self.runup_days,
                17: self.seed_exposed}
        vecs = {20: self.a, 21: self.h_infec, 22: self.p, 23: self.h, 24: self.icu, 25: self.d_H,
                26: self.d_ICU, 27: self.d_community}
        out = np.zeros(self.n_params)
        for k, (c, i) in enumerate(zip(codes, idxs)):
            if c in scal:
                out[k] = scal[c]
            elif 8 <= c <= 15:
                out[k] = self.multipliers[c - 8]
            elif c == F_BETA_VALUE:
                out[k] = self.beta_values[i]
            elif c == F_KAPPA_VALUE:
                out[k] = self.kappa_values[i]
            elif c in vecs:
                out[k] = vecs[c][i]
            else:
                raise ValueError(f"Unknown parameter name: {self.param_names[k]}")
        return out

    def with_(self, **kw) -> "SEPAIHRDProblem":
        return replace(self, **kw)

    # ---- (de)serialisation of fixtures
    def to_json_dict(self) -> dict:
        d = {}
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                d[k] = v.tolist()
            elif isinstance(v, dict):
                d[k] = {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in v.items()}
            else:
                d[k] = v
        return d

    @staticmethod
    def from_json_dict(d: dict) -> "SEPAIHRDProblem":
        d = dict(d)
        if "bounds" in d:
            d["bounds"] = {k: (float(v[0]), float(v[1])) for k, v in d["bounds"].items()}
        return SEPAIHRDProblem(**d)

    @staticmethod
    def load(path: str) -> "SEPAIHRDProblem":
        with open(path, "r") as fh:
            return SEPAIHRDProblem.from_json_dict(json.load(fh))

    def save(self, path: str) -> None:
        with open(path, "w") as fh:
            json.dump(self.to_json_dict(), fh)


def widen_age_classes(pb: SEPAIHRDProblem, factor: int) -> SEPAIHRDProblem:
    """Synthetic n*factor-age problem of SURVEY.md section 8(d) config 5: every band split evenly,
    M'(i,j) = M(i/f, j/f)/f, age vectors replicated, per-age parameter names re-indexed.
    Observations are split evenly too (callers normally replace them with synthetic draws)."""
    n, f = pb.n, factor
    rep = lambda v: np.repeat(np.asarray(v, dtype=np.float64), f)
    M = np.repeat(np.repeat(pb.M, f, axis=0), f, axis=1) / f
    init = np.concatenate([rep(pb.initial_state[c * n:(c + 1) * n]) / f for c in range(NUM_COMPARTMENTS)])
    names, sig, bnd, base = [], {}, {}, []
    age_prefixes = [p for p, _ in F_AGE_PREFIXES]
    for k, nm in enumerate(pb.param_names):
        pref = next((p for p in sorted(age_prefixes, key=len, reverse=True)
                     if nm.startswith(p) and nm[len(p):].isdigit()), None)
        if pref is None:
            new = [nm]
        else:
            i = int(nm[len(pref):])
            new = [f"{pref}{i * f + r}" for r in range(f)]
        for nn in new:
            names.append(nn)
            if nm in pb.sigmas:
                sig[nn] = pb.sigmas[nm]
            if nm in pb.bounds:
                bnd[nn] = pb.bounds[nm]
            if pb.base_theta is not None:
                base.append(pb.base_theta[k])
    spread = lambda o: np.repeat(o, f, axis=1) / f
    return pb.with_(N=rep(pb.N) / f, M=M, a=rep(pb.a), h_infec=rep(pb.h_infec), p=rep(pb.p), h=rep(pb.h),
                    icu=rep(pb.icu), d_H=rep(pb.d_H), d_ICU=rep(pb.d_ICU), d_community=rep(pb.d_community),
                    initial_state=init, obs_H=spread(pb.obs_H), obs_ICU=spread(pb.obs_ICU),
                    obs_D=spread(pb.obs_D), param_names=names, sigmas=sig, bounds=bnd,
                    base_theta=np.array(base) if base else None)


def restrict_age_classes(pb: SEPAIHRDProblem, keep: Sequence[int]) -> SEPAIHRDProblem:
    """Sub-problem on a subset of age classes (test helper: n = 1, 2, 3 exercise the narrow and the
    padded lane layouts).  Age-indexed parameter names are re-indexed; others are kept."""
    keep = list(keep)
    n = pb.n
    sel = np.array(keep)
    init = np.concatenate([pb.initial_state[c * n:(c + 1) * n][sel] for c in range(NUM_COMPARTMENTS)])
    age_prefixes = sorted([p for p, _ in F_AGE_PREFIXES], key=len, reverse=True)
    names, sig, bnd, base = [], {}, {}, []
    for k, nm in enumerate(pb.param_names):
        pref = next((p for p in age_prefixes if nm.startswith(p) and nm[len(p):].isdigit()), None)
        if pref is not None:
            i = int(nm[len(pref):])
            if i not in keep:
                continue
            new = f"{pref}{keep.index(i)}"
        else:
            new = nm
        names.append(new)
        if nm in pb.sigmas:
            sig[new] = pb.sigmas[nm]
        if nm in pb.bounds:
            bnd[new] = pb.bounds[nm]
        if pb.base_theta is not None:
            base.append(pb.base_theta[k])
    return pb.with_(N=pb.N[sel], M=pb.M[np.ix_(sel, sel)], a=pb.a[sel], h_infec=pb.h_infec[sel], p=pb.p[sel],
                    h=pb.h[sel], icu=pb.icu[sel], d_H=pb.d_H[sel], d_ICU=pb.d_ICU[sel],
                    d_community=pb.d_community[sel], initial_state=init, obs_H=pb.obs_H[:, sel],
                    obs_ICU=pb.obs_ICU[:, sel], obs_D=pb.obs_D[:, sel], param_names=names, sigmas=sig, bounds=bnd,
                    base_theta=np.array(base) if base else None)
