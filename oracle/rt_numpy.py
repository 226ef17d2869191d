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


def r0_value(mp, pb):
    """ReproductionNumberCalculator::calculateR0 (:22-52,141-155): F with beta(0), kappa(0), N_i / N_j, not clipped."""
    n = pb.n
    N = np.asarray(pb.N, dtype=np.float64)
    M = np.asarray(pb.M, dtype=np.float64).reshape(n, n)
    kappa = _piecewise(pb.kappa_end_times, mp["kappa_values"], 0.0)
    beta = _piecewise(pb.beta_end_times, mp["beta_values"], 0.0) if len(mp["beta_values"]) else mp["beta"]
    F = np.zeros((4 * n, 4 * n))
    for i in range(n):
        for j in range(n):
            if N[j] < 1e-9:
                continue
            term = beta * kappa * M[i, j] * mp["a"][i] * mp["h_infec"][j] * (N[i] / N[j])
            F[i, n + j] = term
            F[i, 2 * n + j] = term
            F[i, 3 * n + j] = mp["theta"] * term
    V = np.zeros((4 * n, 4 * n))
    for a in range(n):
        e, p_, a_, i_ = a, n + a, 2 * n + a, 3 * n + a
        V[e, e] = mp["sigma"]; V[p_, e] = -mp["sigma"]; V[p_, p_] = mp["gamma_p"]
        V[a_, p_] = -mp["p"][a] * mp["gamma_p"]; V[i_, p_] = -(1.0 - mp["p"][a]) * mp["gamma_p"]
        V[a_, a_] = mp["gamma_A"]; V[i_, i_] = mp["gamma_I"] + mp["h"][a]
    return float(np.max(np.abs(np.linalg.eigvals(F @ np.linalg.inv(V)))))


def essential_metrics(traj, mp, pb):
    """MetricsCalculator::calculateEssentialMetrics (src/model/MetricsCalculator.cpp:8-170) for one sample:
    [R0, overall_IFR, overall_attack_rate, peak_hospital, peak_ICU, time_to_peak_hospital, time_to_peak_ICU,
     total_deaths, max_Rt, min_Rt, final_Rt, seroprevalence_day64] + per age [IFR, IHR, IICUR, attack rate]."""
    n = pb.n
    N = np.asarray(pb.N, dtype=np.float64)
    M = np.asarray(pb.M, dtype=np.float64).reshape(n, n)
    x0 = np.asarray(pb.initial_state, dtype=np.float64).reshape(11, n)
    times = np.asarray(pb.times, dtype=np.float64)
    cum_inf = x0[1:8].sum(axis=0).copy()  # E0 + P0 + A0 + I0 + H0 + ICU0 + R0
    total_pop = N.sum()
    target = int(np.argmin(np.abs(times - 64.0)))  # first minimum, like the strict "<" scan
    peak_h = peak_icu = t_h = t_icu = 0.0
    max_rt, min_rt, final_rt, sero = 0.0, 1e6, 0.0, 0.0
    for k, t in enumerate(times):
        row = traj[k].reshape(11, n)
        S, P, A, I, H, ICU = row[0], row[2], row[3], row[4], row[5], row[6]
        dt = (t - times[k - 1]) if k > 0 else 1.0
        rt = rt_value(S, t, mp, pb)
        max_rt, min_rt = max(max_rt, rt), min(min_rt, rt)
        if k == len(times) - 1:
            final_rt = rt
        if H.sum() > peak_h:
            peak_h, t_h = H.sum(), t
        if ICU.sum() > peak_icu:
            peak_icu, t_icu = ICU.sum(), t
        kappa = mp["kappa_values"][0] if t < 0 else _piecewise(pb.kappa_end_times, mp["kappa_values"], t)
        load = np.where(N > 1e-9, (P + A + mp["theta"] * I) / N, 0.0)
        lam = mp["beta"] * kappa * (M @ load)
        cum_inf += lam * S * dt
        if k == target:
            sero = (total_pop - S.sum()) / total_pop
    last = traj[-1].reshape(11, n)
    deaths, hosp, icu = last[8] - x0[8], last[9] - x0[9], last[10] - x0[10]
    out = [r0_value(mp, pb), deaths.sum() / cum_inf.sum() if cum_inf.sum() > 1e-9 else 0.0, cum_inf.sum() / total_pop,
           peak_h, peak_icu, t_h, t_icu, deaths.sum(), max_rt, min_rt, final_rt, sero]
    for a in range(n):
        if cum_inf[a] > 1.0:
            ratios = [max(0.0, min(v[a] / cum_inf[a], 1.0)) for v in (deaths, hosp, icu)]
        else:
            ratios = [0.0, 0.0, 0.0]
        out += ratios + [cum_inf[a] / N[a] if N[a] > 0 else 0.0]
    return np.array(out)
