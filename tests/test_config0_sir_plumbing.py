"""BASELINE configs[0]: "single-chain deterministic AgeSIRModel, 3 age groups, ... 200 days on CPU
reference (plumbing, no GPU)".  CPU only by definition.  The reference has no fixed-step RK4 strategy
(SURVEY section 0), so the run goes through the same controlled-Dopri5 / integrate_times restatement as the
SEPAIHRD path; the right-hand side is pinned by the reference's OWN known-answer vectors
(tests/sir_age_structured/AgeSIRModelTest.cpp:109-165, committed as data in tests/golden/)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sir_reference_vectors.json")


def test_sir_rhs_reference_known_answers(oracle_py):
    g = json.load(open(GOLDEN))
    for case in g["cases"]:
        d = oracle_py.sir_rhs(g["N"], g["C"], g["gamma"], g["q"], g["scale_C"], case["state"])
        exp = np.array(case["expected"])
        start = 0
        if case.get("first_entry_is_lower_bound"):
            assert d[0] >= exp[0]  # EXPECT_GE(derivatives[0], 0.0)
            start = 1
        assert np.all(np.abs(d[start:] - exp[start:]) <= case["tol"]), (case["name"], d)


def test_config0_three_age_groups_200_days(oracle_py):
    N = np.array([5.0e5, 1.2e6, 3.0e5])
    Cm = np.array([[8.0, 3.0, 1.0], [3.0, 6.0, 2.0], [1.0, 2.0, 3.0]])
    gamma = np.array([0.2, 0.2, 0.15])
    q, scale = 0.03, 1.0
    I0 = np.array([10.0, 20.0, 5.0])
    init = np.concatenate([N - I0, I0, np.zeros(3)])
    times = np.arange(0.0, 201.0)
    r = oracle_py.sir_simulate(N, Cm, gamma, q, scale, init, times)
    traj = r["traj"]
    assert traj.shape == (201, 9) and r["n_accept"] >= 200  # never steps across an output time
    assert np.array_equal(traj[0], init)
    tot = traj[:, 0:3] + traj[:, 3:6] + traj[:, 6:9]
    assert np.max(np.abs(tot - N) / N) < 1e-12          # dS + dI + dR = 0 row by row
    assert np.all(np.diff(traj[:, 0:3], axis=0) <= 0) and np.all(np.diff(traj[:, 6:9], axis=0) >= 0)
    assert traj[-1, 6:9].sum() > 0.5 * N.sum()            # the epidemic ran (R0 > 1 for these values)

    # independent high-precision answer
    from scipy.integrate import solve_ivp

    def f(t, x):
        lam = q * (Cm * scale) @ (x[3:6] / N)
        return np.concatenate([-lam * x[0:3], lam * x[0:3] - gamma * x[3:6], gamma * x[3:6]])
    ref = solve_ivp(f, (0.0, 200.0), init, method="DOP853", t_eval=times, rtol=1e-12, atol=1e-9).y.T
    assert np.max(np.abs(traj - ref) / np.maximum(np.abs(ref), 1.0)) < 2e-4   # solver-tolerance limited at 1e-6
    tight = oracle_py.sir_simulate(N, Cm, gamma, q, scale, init, times, 1e-11, 1e-11)["traj"]
    assert np.max(np.abs(tight - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-8
