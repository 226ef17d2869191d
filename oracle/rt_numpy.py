"""oracle/rt_numpy.py -- TEST INFRASTRUCTURE.  numpy restatement of the reference's effective reproduction
number, following src/model/ReproductionNumberCalculator.cpp line by line:
  buildFMatrixForRt (:55-92), buildVMatrix (:95-123), calculateRt (:158-171): the largest eigenvalue
  magnitude of F V^-1 over the 4 n states (E, P, A, I) x age, with LAPACK where the reference calls
  Eigen (V.inverse(), EigenSolver).
Schedules: beta(t) = PiecewiseConstantParameterStrategy::getValue (first period with t <= end, past the last
end the last value; no schedule -> constant beta), kappa(t) = PiecewiseConstantNpiStrategy::getReductionFactor
(baseline for t < 0 or t <= baseline end)."""
import numpy as np


def _piecewise(ends, values, t):
    idx = int(np.sum(np.asarray(ends) < t))
    return values[min(idx, len(values) - 1)]


def rt_value(S, t, mp, pb):
    n = pb.n
    N = np.asarray(pb.N, dtype=np.float64)
    M = np.asarray(pb.M, dtype=np.float64).reshape(n, n)
    kappa = mp["kappa_values"][0] if t < 0 else _piecewise(pb.kappa_end_times, mp["kappa_values"], t)
    beta = _piecewise(pb.beta_end_times, mp["beta_values"], t) if len(mp["beta_values"]) else mp["beta"]
    F = np.zeros((4 * n, 4 * n))
    for i in range(n):
        for j in range(n):
            if N[j] < 1e-9:
                continue
            term = beta * kappa * M[i, j] * mp["a"][i] * mp["h_infec"][j] * (S[i] / N[j])
            term = max(0.0, term)
            F[i, n + j] = term
            F[i, 2 * n + j] = term
            F[i, 3 * n + j] = mp["theta"] * term
    V = np.zeros((4 * n, 4 * n))
    for a in range(n):
        e, p_, a_, i_ = a, n + a, 2 * n + a, 3 * n + a
        V[e, e] = mp["sigma"]
        V[p_, e] = -mp["sigma"]
        V[p_, p_] = mp["gamma_p"]
        V[a_, p_] = -mp["p"][a] * mp["gamma_p"]
        V[i_, p_] = -(1.0 - mp["p"][a]) * mp["gamma_p"]
        V[a_, a_] = mp["gamma_A"]
        V[i_, i_] = mp["gamma_I"] + mp["h"][a]
    K = F @ np.linalg.inv(V)
    return float(np.max(np.abs(np.linalg.eigvals(K))))


def rt_trajectory(traj, mp, pb):
    """traj: [T][11 n] of one sample -> Rt at every output time (MetricsCalculator::calculateRtTrajectory)."""
    return np.array([rt_value(traj[k, :pb.n], pb.times[k], mp, pb) for k in range(len(pb.times))])


def sorted_quantile(v, q):
    """PostCalibrationAnalyser.cpp:316-326"""
    v = np.sort(np.asarray(v, dtype=np.float64))
    pos = q * (len(v) - 1)
    idx = int(pos)
    frac = pos - idx
    if idx + 1 < len(v):
        return v[idx] * (1.0 - frac) + v[idx + 1] * frac
    return v[idx]
