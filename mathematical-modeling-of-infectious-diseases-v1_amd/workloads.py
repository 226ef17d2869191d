"""BASELINE.json workloads as problem builders (host plumbing for bench.py and the full-size tests).

  c1  SEPAIHRD n=4, Dopri5, 400 days (t=-20..380), 4 096 chains / GPU          (configs[1], bench default)
  c2  same model, Cash-Karp, 65 536 chains / GPU                                (configs[2])
  c3  same model, Dopri5, 32 768 chains / GPU (262 144 over 8 GPUs)             (configs[3])
  c5  n=16 synthetic (bands split 4 ways, M16 = M4/4), Dopri5, 1 000 days, 32 768 chains / GPU   (configs[4])

Observations of c5 are drawn once from the base-theta trajectory computed by the HIP path itself
(trajectory mode) with numpy's RandomState(12345), i.e. SURVEY.md 8(d)'s recipe without touching the oracle.
"""
from __future__ import annotations

import os

import numpy as np

from .problem import SEPAIHRDProblem, widen_age_classes, SOLVER_DOPRI5, SOLVER_CASH_KARP54

DEFAULT_CHAINS = {"c1": 4096, "c2": 65536, "c3": 32768, "c5": 32768}


def _incidence(traj: np.ndarray, n: int, comp: int, runup_offset: int) -> np.ndarray:
    cum = traj[:, comp * n:(comp + 1) * n]
    inc = np.maximum(np.diff(cum, axis=0, prepend=cum[:1]), 0.0)
    return inc[runup_offset:]


def with_synthetic_observations(pb: SEPAIHRDProblem, hip_factory, seed: int = 12345) -> SEPAIHRDProblem:
    """obs = Poisson(model incidence at base theta); the trajectory comes from the device."""
    T_obs = pb.n_times - pb.runup_offset
    blank = pb.with_(obs_H=np.zeros((T_obs, pb.n)), obs_ICU=np.zeros((T_obs, pb.n)), obs_D=np.zeros((T_obs, pb.n)))
    blank.base_theta = pb.base_theta
    hip = hip_factory(blank)
    traj = hip.eval_batch(pb.base_theta[None, :], want_traj=True)["traj"][0]
    hip.close()
    rs = np.random.RandomState(seed)
    obs = {name: rs.poisson(_incidence(traj, pb.n, comp, pb.runup_offset)).astype(np.float64)
           for name, comp in (("obs_H", 9), ("obs_ICU", 10), ("obs_D", 8))}
    out = blank.with_(**obs)
    out.base_theta = pb.base_theta
    return out


def build(name: str, golden_dir: str, hip_factory=None) -> SEPAIHRDProblem:
    if name in ("c1", "c2", "c3"):
        pb = SEPAIHRDProblem.load(os.path.join(golden_dir, "synth_400d_n4.json"))
        pb.solver = SOLVER_CASH_KARP54 if name == "c2" else SOLVER_DOPRI5
        return pb
    if name == "c5":
        base = SEPAIHRDProblem.load(os.path.join(golden_dir, "shipped_problem.json"))
        wide = widen_age_classes(base, 4)
        wide = wide.with_(times=np.arange(-20, 981, dtype=np.float64))
        wide.base_theta = widen_age_classes(base, 4).base_theta
        wide.solver = SOLVER_DOPRI5
        if hip_factory is None:
            raise ValueError("workload c5 needs a device to draw its synthetic observations")
        return with_synthetic_observations(wide, hip_factory)
    raise ValueError(f"unknown workload {name}")
