#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ (run in the BUILD container only).

Inputs are *data files* of the reference (never its sources):
  /root/reference/data/configuration/{initial_guess,param_bounds,proposal_sigmas,params_to_calibrate}.txt
  /root/reference/data/contacts.csv, /root/reference/data/processed/processed_data.csv
and the data-generation recipe of the reference's test fixture
(tests/model/SEPAIHRDObjectivefunctionTest.cpp:63-91,136-242: populations, contact matrix,
rates, kappa schedule, Poisson(10)+Gaussian-bump observations from std::mt19937(42)).

Outputs (all small, committed):
  shipped_problem.json        the shipped n=4 / 326-point calibration problem
  synth_400d_n4.json          BASELINE configs 2-4: grid t=-20..380, synthetic Poisson observations
  reference_test_fixture.json the reference test-fixture problem (multiplier branch, 30 points)
  golden_highprec.json        independent high-precision answers (SciPy DOP853 rtol=1e-13,
                              restarted at every beta/kappa breakpoint; mpmath RHS spot values;
                              closed-form Poisson log-likelihood)
Usage: python tests/golden/make_fixtures.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mmid_amd_loader  # noqa: E402

mm = mmid_amd_loader.load()
cio = mm.config_io
REF = "/root/reference/data"


# ------------------------------------------------------------------ problems
def shipped_problem():
    n = 4
    par = cio.read_sepaihrd_parameters(f"{REF}/configuration/initial_guess.txt", n)
    bounds = cio.read_param_bounds(f"{REF}/configuration/param_bounds.txt")
    sigmas = cio.read_proposal_sigmas(f"{REF}/configuration/proposal_sigmas.txt")
    names = cio.read_params_to_calibrate(f"{REF}/configuration/params_to_calibrate.txt")
    data = cio.read_calibration_csv(f"{REF}/processed/processed_data.csv", "2020-03-01", "2020-12-31")
    M = cio.read_matrix_csv(f"{REF}/contacts.csv", n, n)
    runup = par["runup_days"]
    times = np.arange(-int(runup), data.num_data_points, dtype=np.float64)  # main.cpp:247-253
    init = cio.initial_sepaihrd_state(data, par["sigma"], par["gamma_p"], par["gamma_A"], par["gamma_I"],
                                      par["p"], par["h"])
    # main.cpp / benchmark :332-364 seed the run-up state; the objective overwrites it again
    N = data.population
    if runup > 0 and par["seed_exposed"] > 0:
        frac = N / N.sum()
        init[n:2 * n] = par["seed_exposed"] * frac
        init[2 * n:] = 0.0
    for i in range(n):
        s = sum(init[j * n + i] for j in range(1, 9))
        init[i] = 0.0 if s > N[i] else N[i] - s
    pb = mm.SEPAIHRDProblem(
        N=N, M=M, a=par["a"], h_infec=par["h_infec"], p=par["p"], h=par["h"], icu=par["icu"],
        d_H=par["d_H"], d_ICU=par["d_ICU"], d_community=par["d_community"], theta=par["theta"],
        sigma=par["sigma"], gamma_p=par["gamma_p"], gamma_A=par["gamma_A"], gamma_I=par["gamma_I"],
        gamma_H=par["gamma_H"], gamma_ICU=par["gamma_ICU"], kappa_end_times=par["kappa_end_times"],
        kappa_values=par["kappa_values"], beta=0.0, beta_end_times=par["beta_end_times"],
        beta_values=par["beta_values"],
        multipliers=[par[k] for k in ("E0_multiplier", "P0_multiplier", "A0_multiplier", "I0_multiplier",
                                      "H0_multiplier", "ICU0_multiplier", "R0_multiplier", "D0_multiplier")],
        runup_days=runup, seed_exposed=par["seed_exposed"], times=times, initial_state=init,
        obs_H=data.new_hospitalizations, obs_ICU=data.new_icu, obs_D=data.new_deaths,
        param_names=names, sigmas={k: sigmas[k] for k in names}, bounds={k: bounds[k] for k in names},
        constraint_mode=mm.CONSTRAINT_REFLECT)
    pb.base_theta = pb.current_parameters()
    return pb


def synth_400d(shipped, oracle_mod):
    """Grid t=-20..380 (T=401, T_obs=381); obs = Poisson(model incidence at base theta), drawn once
    with numpy's legacy MT19937 RandomState(12345)."""
    times = np.arange(-20, 381, dtype=np.float64)
    T_obs = 381
    tmp = shipped.with_(times=times, obs_H=np.zeros((T_obs, 4)), obs_ICU=np.zeros((T_obs, 4)),
                        obs_D=np.zeros((T_obs, 4)))
    tmp.base_theta = shipped.base_theta
    orc = oracle_mod.Oracle(tmp)
    tr = orc.eval_batch(tmp.base_theta, want_traj=True, nthreads=1)["traj"][0].reshape(401, 11, 4)
    rs = np.random.RandomState(12345)
    obs = {}
    for name, comp in (("obs_H", 9), ("obs_ICU", 10), ("obs_D", 8)):
        inc = np.maximum(np.diff(tr[:, comp, :], axis=0, prepend=tr[:1, comp, :]), 0.0)[20:]
        obs[name] = rs.poisson(inc).astype(np.float64)
    out = tmp.with_(**obs)
    out.base_theta = shipped.base_theta
    return out


POISSON_HELPER = r"""
#include <cmath>
#include <cstdio>
#include <random>
#include <algorithm>
int main() {
    std::mt19937 rng; rng.seed(42);
    std::poisson_distribution<int> pois(10);
    const int T = 30, A = 4;
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < A; ++a) {
            double base_rate = 100.0 * std::exp(-0.5 * std::pow((t - 15.0) / 10.0, 2));
            double age_factor = (a + 1) * 0.5;
            int hosp = std::max(0, pois(rng) + static_cast<int>(base_rate * age_factor * 0.1));
            int icu = std::max(0, pois(rng) + static_cast<int>(base_rate * age_factor * 0.03));
            int deaths = std::max(0, static_cast<int>(base_rate * age_factor * 0.01));
            int cases = std::max(0, pois(rng) + static_cast<int>(base_rate * age_factor));
            std::printf("%d %d %d %d\n", hosp, icu, deaths, cases);
        }
    return 0;
}
"""


def reference_test_fixture():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "gen.cpp")
        with open(src, "w") as fh:
            fh.write(POISSON_HELPER)
        exe = os.path.join(td, "gen")
        subprocess.run(["g++", "-O1", "-o", exe, src], check=True)
        rows = np.array([[int(v) for v in ln.split()] for ln in
                         subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()],
                        dtype=np.float64).reshape(30, 4, 4)
    hosp, icu, deaths, cases = (rows[:, :, k] for k in range(4))
    N = np.array([3e6, 4e6, 2e6, 1e6])
    data = cio.CalibrationData(
        dates=[f"mock_date_{i}" for i in range(30)], new_confirmed=cases, new_deaths=deaths,
        new_hospitalizations=hosp, new_icu=icu,
        # CalibrationData matrix ctor (GetCalibrationData.cpp:60-83): row0 = given initial row,
        # row i = row i-1 + new[i-1]; the fixture passes cumsum(new)[0] = new[0] as the initial row
        cumulative_confirmed=np.vstack([cases[:1], cases[:1] + np.cumsum(cases[:-1], axis=0)]),
        cumulative_deaths=np.vstack([deaths[:1], deaths[:1] + np.cumsum(deaths[:-1], axis=0)]),
        cumulative_hospitalizations=np.vstack([hosp[:1], hosp[:1] + np.cumsum(hosp[:-1], axis=0)]),
        cumulative_icu=np.vstack([icu[:1], icu[:1] + np.cumsum(icu[:-1], axis=0)]), population=N)
    p = np.array([0.4, 0.3, 0.2, 0.1])
    h = np.array([0.01, 0.03, 0.08, 0.15])
    sigma, gamma_p, gamma_A, gamma_I = 1 / 3.0, 1 / 2.0, 1 / 5.0, 1 / 5.0
    init = cio.initial_sepaihrd_state(data, sigma, gamma_p, gamma_A, gamma_I, p, h)
    names = ["beta", "theta", "kappa_1", "kappa_2", "kappa_3"]
    lo, hi, sg = [0.01, 0.1, 0.1, 0.1, 0.1], [1.0, 1.0, 1.5, 1.5, 1.5], [0.01, 0.01, 0.05, 0.05, 0.05]
    pb = mm.SEPAIHRDProblem(
        N=N, M=[[7, 5, 2, 1], [5, 8, 3, 1.5], [2, 3, 4, 2], [1, 1.5, 2, 3]], a=np.ones(4), h_infec=np.ones(4),
        p=p, h=h, icu=[0.05, 0.10, 0.25, 0.40], d_H=[0.01, 0.02, 0.05, 0.10], d_ICU=[0.20, 0.30, 0.40, 0.50],
        d_community=np.zeros(4), theta=0.5, sigma=sigma, gamma_p=gamma_p, gamma_A=gamma_A, gamma_I=gamma_I,
        gamma_H=1 / 10.0, gamma_ICU=1 / 14.0, kappa_end_times=[13.0, 63.0, 111.0, 305.0],
        kappa_values=[1.0, 0.5, 0.7, 0.9], beta=0.05, multipliers=np.ones(8), runup_days=0.0, seed_exposed=0.0,
        times=np.arange(30, dtype=np.float64), initial_state=init, obs_H=hosp, obs_ICU=icu, obs_D=deaths,
        param_names=names, sigmas=dict(zip(names, sg)), bounds={k: (a, b) for k, a, b in zip(names, lo, hi)},
        npi_names=["kappa_1", "kappa_2", "kappa_3"], constraint_mode=mm.CONSTRAINT_CLAMP)
    pb.base_theta = pb.current_parameters()
    return pb


# ------------------------------------------------------------------ independent high-precision answers
def model_from(pb, theta=None):
    """Plain-numpy model parameters after theta has been written in (mirror written from the
    model equations, not from the oracle)."""
    m = {k: np.array(getattr(pb, k), dtype=np.float64) for k in
         ("N", "M", "a", "h_infec", "p", "h", "icu", "d_H", "d_ICU", "d_community", "kappa_end_times",
          "kappa_values", "beta_end_times", "beta_values", "multipliers")}
    for k in ("beta", "theta", "sigma", "gamma_p", "gamma_A", "gamma_I", "gamma_H", "gamma_ICU", "runup_days",
              "seed_exposed"):
        m[k] = float(getattr(pb, k))
    if theta is not None:
        codes, idxs = pb.field_map()
        scal = {0: "beta", 1: "theta", 2: "sigma", 3: "gamma_p", 4: "gamma_A", 5: "gamma_I", 6: "gamma_H",
                7: "gamma_ICU", 16: "runup_days", 17: "seed_exposed"}
        vec = {20: "a", 21: "h_infec", 22: "p", 23: "h", 24: "icu", 25: "d_H", 26: "d_ICU", 27: "d_community"}
        for v, c, i in zip(theta, codes, idxs):
            if c in scal:
                m[scal[c]] = float(v)
            elif 8 <= c <= 15:
                m["multipliers"][c - 8] = v
            elif c == 18:
                m["beta_values"][i] = v
            elif c == 19:
                m["kappa_values"][i] = v
            elif c in vec:
                m[vec[c]][i] = v
    return m


def sched(ends, vals, t):
    for e, v in zip(ends, vals):
        if t <= e:
            return v
    return vals[-1]


def rhs_numpy(m, t, x, bk=None):
    n = len(m["N"])
    S, E, P, A, I, H, ICU = (x[c * n:(c + 1) * n] for c in range(7))
    if bk is None:
        beta = sched(m["beta_end_times"], m["beta_values"], t) if len(m["beta_values"]) else m["beta"]
        bk = beta * sched(m["kappa_end_times"], m["kappa_values"], t)
    pi = (P + A + m["theta"] * I) * m["h_infec"] / m["N"]
    lam = np.maximum(bk * m["a"] * (m["M"] @ pi), 0.0)
    out = np.zeros(11 * n)
    out[0:n] = -lam * S
    out[n:2 * n] = lam * S - m["sigma"] * E
    out[2 * n:3 * n] = m["sigma"] * E - m["gamma_p"] * P
    out[3 * n:4 * n] = m["p"] * m["gamma_p"] * P - m["gamma_A"] * A
    out[4 * n:5 * n] = (1 - m["p"]) * m["gamma_p"] * P - (m["gamma_I"] + m["h"] + m["d_community"]) * I
    out[5 * n:6 * n] = m["h"] * I - (m["gamma_H"] + m["d_H"] + m["icu"]) * H
    out[6 * n:7 * n] = m["icu"] * H - (m["gamma_ICU"] + m["d_ICU"]) * ICU
    out[7 * n:8 * n] = m["gamma_A"] * A + m["gamma_I"] * I + m["gamma_H"] * H + m["gamma_ICU"] * ICU
    out[8 * n:9 * n] = m["d_H"] * H + m["d_ICU"] * ICU + m["d_community"] * I
    out[9 * n:10 * n] = m["h"] * I
    out[10 * n:11 * n] = m["icu"] * H
    return out


def initial_state_for(pb, m):
    n = pb.n
    x = np.array(pb.initial_state, dtype=np.float64)
    if m["runup_days"] > 0 and m["seed_exposed"] > 0:
        x[n:2 * n] = m["seed_exposed"] * m["N"] / m["N"].sum()
        x[2 * n:] = 0.0
    else:
        for c in range(1, 9):
            x[c * n:(c + 1) * n] *= m["multipliers"][c - 1]
    x[0:n] = m["N"] - sum(x[c * n:(c + 1) * n] for c in range(1, 9))
    return x


def highprec_trajectory(pb, theta):
    """DOP853, rtol=1e-13/atol=1e-9, restarted at every schedule breakpoint so that no step
    straddles a discontinuity of beta(t)kappa(t)."""
    from scipy.integrate import solve_ivp
    m = model_from(pb, theta)
    x = initial_state_for(pb, m)
    times = np.array(pb.times)
    brk = sorted(set(float(e) for e in list(m["kappa_end_times"]) + list(m["beta_end_times"])
                     if times[0] < e < times[-1]))
    edges = [times[0]] + brk + [times[-1]]
    sol = np.zeros((len(times), len(x)))
    sol[0] = x
    for lo, hi in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (lo + hi)
        beta = sched(m["beta_end_times"], m["beta_values"], mid) if len(m["beta_values"]) else m["beta"]
        bk = beta * sched(m["kappa_end_times"], m["kappa_values"], mid)
        mask = (times > lo) & (times <= hi)
        r = solve_ivp(lambda t, y: rhs_numpy(m, t, y, bk), (lo, hi), x, method="DOP853", rtol=1e-13,
                      atol=1e-9, t_eval=np.append(times[mask], hi) if not mask.any() or times[mask][-1] != hi
                      else times[mask])
        assert r.success
        sol[mask] = r.y.T[:mask.sum()]
        x = r.y[:, -1]
    return sol


def loglik_from_traj(pb, sol):
    n = pb.n
    m0 = pb.runup_offset
    init = sol[0]
    total = 0.0
    parts = []
    for comp, obs in ((9, pb.obs_H), (10, pb.obs_ICU), (8, pb.obs_D)):
        cum = sol[:, comp * n:(comp + 1) * n]
        inc = np.maximum(np.diff(cum, axis=0, prepend=init[None, comp * n:(comp + 1) * n]), 0.0)[m0:]
        ok = np.isfinite(obs) & (obs >= 0)
        s = inc + 1e-10
        ll = float(np.sum(np.where(ok, obs * np.log(s) - s, 0.0)))
        parts.append(ll)
        total += ll
    return total, parts


def mpmath_rhs_spots(pb, theta):
    import mpmath as mp
    mp.mp.dps = 40
    m = model_from(pb, theta)
    n = pb.n
    rs = np.random.RandomState(7)
    spots = []
    for t in (-3.0, 13.0, 13.5, 63.0, 111.25, 200.0):
        x = np.concatenate([m["N"] * 0.9] + [m["N"] * f * rs.uniform(0.5, 1.5, n)
                                             for f in (1e-3, 8e-4, 6e-4, 5e-4, 1e-4, 2e-5, 5e-2, 1e-3, 2e-3, 4e-4)])
        X = [mp.mpf(float(v)) for v in x]
        g = lambda k, i=None: mp.mpf(float(m[k])) if i is None else mp.mpf(float(m[k][i]))
        beta = sched(m["beta_end_times"], m["beta_values"], t) if len(m["beta_values"]) else m["beta"]
        bk = mp.mpf(float(beta)) * mp.mpf(float(sched(m["kappa_end_times"], m["kappa_values"], t)))
        pi = [(X[2 * n + j] + X[3 * n + j] + g("theta") * X[4 * n + j]) * g("h_infec", j) / g("N", j)
              for j in range(n)]
        dx = [mp.mpf(0)] * (11 * n)
        for i in range(n):
            lam = bk * g("a", i) * sum(mp.mpf(float(m["M"][i, j])) * pi[j] for j in range(n))
            lam = max(lam, mp.mpf(0))
            S, E, P, A, I, H, ICU = (X[c * n + i] for c in range(7))
            dx[i] = -lam * S
            dx[n + i] = lam * S - g("sigma") * E
            dx[2 * n + i] = g("sigma") * E - g("gamma_p") * P
            dx[3 * n + i] = g("p", i) * g("gamma_p") * P - g("gamma_A") * A
            dx[4 * n + i] = (1 - g("p", i)) * g("gamma_p") * P - (g("gamma_I") + g("h", i) + g("d_community", i)) * I
            dx[5 * n + i] = g("h", i) * I - (g("gamma_H") + g("d_H", i) + g("icu", i)) * H
            dx[6 * n + i] = g("icu", i) * H - (g("gamma_ICU") + g("d_ICU", i)) * ICU
            dx[7 * n + i] = g("gamma_A") * A + g("gamma_I") * I + g("gamma_H") * H + g("gamma_ICU") * ICU
            dx[8 * n + i] = g("d_H", i) * H + g("d_ICU", i) * ICU + g("d_community", i) * I
            dx[9 * n + i] = g("h", i) * I
            dx[10 * n + i] = g("icu", i) * H
        spots.append({"t": t, "x": x.tolist(), "dxdt": [float(v) for v in dx]})
    return spots


def main():
    import oracle_py
    ship = shipped_problem()
    ship.save(os.path.join(HERE, "shipped_problem.json"))
    syn = synth_400d(ship, oracle_py)
    syn.save(os.path.join(HERE, "synth_400d_n4.json"))
    fix = reference_test_fixture()
    fix.save(os.path.join(HERE, "reference_test_fixture.json"))

    gold = {}
    sel = [0, 1, 5, 20, 33, 34, 60, 83, 84, 131, 200, 257, 325]
    for key, pb in (("shipped", ship), ("reference_test_fixture", fix)):
        theta = pb.base_theta
        sol = highprec_trajectory(pb, theta)
        ll, parts = loglik_from_traj(pb, sol)
        idx = [k for k in sel if k < pb.n_times]
        gold[key] = {"theta": theta.tolist(), "loglik": ll, "ll_parts": parts, "time_index": idx,
                     "states": sol[idx].tolist()}
    # a perturbed theta on the shipped problem (off-bound values)
    rs = np.random.RandomState(3)
    lo, hi, _ = ship.bounds_arrays()
    th2 = lo + (hi - lo) * rs.uniform(0.2, 0.8, ship.n_params)
    sol = highprec_trajectory(ship, th2)
    ll, parts = loglik_from_traj(ship, sol)
    idx = [k for k in sel if k < ship.n_times]
    gold["shipped_perturbed"] = {"theta": th2.tolist(), "loglik": ll, "ll_parts": parts, "time_index": idx,
                                 "states": sol[idx].tolist()}
    gold["rhs_spots_shipped"] = mpmath_rhs_spots(ship, ship.base_theta)
    # closed-form Poisson log-likelihood of the reference's ManualPoissonLikelihoodTest matrices
    obs = np.array([[5, 3], [2, 7], [4, 1], [6, 0], [3, 5]], dtype=np.float64)
    sim = np.array([[4.8, 3.2], [2.1, 6.9], [3.9, 1.1], [5.8, 0.2], [3.1, 4.9]])
    import mpmath as mp
    mp.mp.dps = 40
    ll = sum(mp.mpf(float(o)) * mp.log(mp.mpf(float(s)) + mp.mpf("1e-10")) - (mp.mpf(float(s)) + mp.mpf("1e-10"))
             for o, s in zip(obs.ravel(), sim.ravel()))
    gold["manual_poisson"] = {"obs": obs.tolist(), "sim": sim.tolist(), "loglik": float(ll)}
    with open(os.path.join(HERE, "golden_highprec.json"), "w") as fh:
        json.dump(gold, fh)
    for k, v in gold.items():
        if isinstance(v, dict) and "loglik" in v:
            print(k, "loglik", v["loglik"])


if __name__ == "__main__":
    main()
