"""Independent pins of the integrator restatement (the reference holds no vector for it).

The controlled steppers of the oracle are restated from Boost.Odeint's published algorithm.  These
tests check their stage arithmetic and their error norm against code that shares nothing with that
restatement:
  * Dormand-Prince 5(4): SciPy's RK45 tableau and `rk_step` (scipy.integrate._ivp.rk);
  * Cash-Karp 5(4): the tableau of Cash & Karp (1990) as exact fractions, applied by a generic
    explicit Runge-Kutta step written here;
  * error norm / accept rule of odeint's default_error_checker:
    max_i |err_i| / (eps_abs + eps_rel (|x_i| + dt |dxdt_i|)) <= 1, located by bisection on the
    tolerance at which the oracle's FIRST attempt flips from accepted to rejected.
A one-interval output grid [0, h] with dt_hint >= h makes the first attempt exactly one step of size h.
"""
from fractions import Fraction as F

import numpy as np
import pytest

H = 0.37


def _one_step(oracle_py, pb, solver, tol, h=H):
    q = pb.with_(times=np.array([0.0, h]), solver=solver, abs_err=tol, rel_err=tol, dt_hint=1.0,
                 obs_H=pb.obs_H[:2], obs_ICU=pb.obs_ICU[:2], obs_D=pb.obs_D[:2])
    orc = oracle_py.Oracle(q)
    r = orc.eval_batch(np.asarray(q.base_theta)[None, :], want_traj=True, nthreads=1)
    return orc, q, r


def _rk_step(f, t, y, h, A, B, C):
    K = []
    for s in range(len(C)):
        ys = y.copy()
        for j in range(s):
            if A[s][j] != 0:
                ys = ys + h * float(A[s][j]) * K[j]
        K.append(f(t + float(C[s]) * h, ys))
    y_new = y.copy()
    for j, b in enumerate(B):
        if b != 0:
            y_new = y_new + h * float(b) * K[j]
    return y_new, K


# Cash & Karp, ACM TOMS 16 (1990) 201-222, table of the 5(4) pair
CK_C = [F(0), F(1, 5), F(3, 10), F(3, 5), F(1), F(7, 8)]
CK_A = [[],
        [F(1, 5)],
        [F(3, 40), F(9, 40)],
        [F(3, 10), F(-9, 10), F(6, 5)],
        [F(-11, 54), F(5, 2), F(-70, 27), F(35, 27)],
        [F(1631, 55296), F(175, 512), F(575, 13824), F(44275, 110592), F(253, 4096)]]
CK_B5 = [F(37, 378), F(0), F(250, 621), F(125, 594), F(0), F(512, 1771)]
CK_B4 = [F(2825, 27648), F(0), F(18575, 48384), F(13525, 55296), F(277, 14336), F(1, 4)]


def test_dopri5_step_matches_scipy_rk45(oracle_py, ref_fixture):
    from scipy.integrate._ivp import rk
    orc, q, r = _one_step(oracle_py, ref_fixture, 0, 1e6)
    assert r["n_accept"][0] == 1 and r["n_reject"][0] == 0
    y0, y1 = r["traj"][0, 0], r["traj"][0, 1]
    f = lambda t, y: orc.rhs(y, t, q.base_theta)
    K = np.empty((rk.RK45.n_stages + 1, y0.size))
    y_new, _ = rk.rk_step(f, 0.0, y0, f(0.0, y0), H, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
    assert np.max(np.abs(y1 - y_new) / np.maximum(np.abs(y_new), 1.0)) < 5e-15


def test_cash_karp_step_matches_published_tableau(oracle_py, ref_fixture):
    assert sum(CK_B5) == 1 and sum(CK_B4) == 1 and all(sum(a) == c for a, c in zip(CK_A, CK_C))
    orc, q, r = _one_step(oracle_py, ref_fixture, 1, 1e6)
    assert r["n_accept"][0] == 1 and r["n_reject"][0] == 0
    y0, y1 = r["traj"][0, 0], r["traj"][0, 1]
    f = lambda t, y: orc.rhs(y, t, q.base_theta)
    y_new, _ = _rk_step(f, 0.0, y0, H, CK_A, CK_B5, CK_C)
    assert np.max(np.abs(y1 - y_new) / np.maximum(np.abs(y_new), 1.0)) < 5e-15


@pytest.mark.parametrize("solver", [0, 1])
def test_error_norm_flip_tolerance(oracle_py, ref_fixture, solver):
    """The first attempt is accepted iff max |err_i| / (tol (1 + |x_i| + h |f_i|)) <= 1 with the
    embedded error taken from the independent tableau."""
    orc, q, r = _one_step(oracle_py, ref_fixture, solver, 1e6)
    y0 = r["traj"][0, 0]
    f = lambda t, y: orc.rhs(y, t, q.base_theta)
    f0 = f(0.0, y0)
    if solver == 0:
        from scipy.integrate._ivp import rk
        K = np.empty((rk.RK45.n_stages + 1, y0.size))
        rk.rk_step(f, 0.0, y0, f0, H, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
        err = (K.T @ rk.RK45.E) * H
    else:
        _, K = _rk_step(f, 0.0, y0, H, CK_A, CK_B5, CK_C)
        err = H * sum(float(b5 - b4) * k for b5, b4, k in zip(CK_B5, CK_B4, K))
    tol_star = np.max(np.abs(err) / (1.0 + np.abs(y0) + H * np.abs(f0)))
    assert tol_star > 0

    def first_attempt_accepted(tol):
        rr = _one_step(oracle_py, ref_fixture, solver, tol)[2]
        return rr["n_reject"][0] == 0 and rr["n_accept"][0] == 1

    assert first_attempt_accepted(tol_star * (1 + 1e-9)) and not first_attempt_accepted(tol_star * (1 - 1e-9))


def _independent_err(oracle_py, pb, solver, h):
    """max_i |err_i| / (1 + |x_i| + h |f_i|) of one step of size h from the objective's initial state,
    from the independent tableaus: odeint's error value is this divided by tol (eps_abs = eps_rel = tol)."""
    orc, q, r = _one_step(oracle_py, pb, solver, 1e6, h)
    y0 = r["traj"][0, 0]
    f = lambda t, y: orc.rhs(y, t, q.base_theta)
    f0 = f(0.0, y0)
    if solver == 0:
        from scipy.integrate._ivp import rk
        K = np.empty((rk.RK45.n_stages + 1, y0.size))
        rk.rk_step(f, 0.0, y0, f0, h, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
        err = (K.T @ rk.RK45.E) * h
    else:
        _, K = _rk_step(f, 0.0, y0, h, CK_A, CK_B5, CK_C)
        err = h * sum(float(b5 - b4) * k for b5, b4, k in zip(CK_B5, CK_B4, K))
    return np.max(np.abs(err) / (1.0 + np.abs(y0) + h * np.abs(f0)))


def _accepts(oracle_py, pb, solver, tol, times, dt_hint):
    q = pb.with_(times=np.asarray(times, dtype=np.float64), solver=solver, abs_err=tol, rel_err=tol, dt_hint=dt_hint,
                 obs_H=pb.obs_H[:len(times)], obs_ICU=pb.obs_ICU[:len(times)], obs_D=pb.obs_D[:len(times)])
    r = oracle_py.Oracle(q).eval_batch(np.asarray(q.base_theta)[None, :], nthreads=1)
    return int(r["n_accept"][0]), int(r["n_reject"][0])


@pytest.mark.parametrize("solver", [0, 1])
def test_step_increase_rule(oracle_py, ref_fixture, solver):
    """default_step_adjuster after an accepted step with err < 0.5: dt <- dt 0.9 err^(-1/5) (order 5).
    The second output interval is covered by ONE step iff it is no longer than the new dt."""
    h, err1 = 0.2, 0.1
    tol = _independent_err(oracle_py, ref_fixture, solver, h) / err1
    dt_new = h * 0.9 * err1 ** (-1.0 / 5.0)
    for delta, want in ((dt_new * (1 - 1e-9), 2), (dt_new * (1 + 1e-9), 3)):
        acc, rej = _accepts(oracle_py, ref_fixture, solver, tol, [0.0, h, h + delta], dt_hint=h)
        assert (acc, rej) == (want, 0), (delta, acc, rej)


@pytest.mark.parametrize("solver", [0, 1])
def test_step_decrease_rule(oracle_py, ref_fixture, solver):
    """after a rejected step: dt <- dt max(0.9 err^(-1/3), 0.2) (error order 4); with err just above 1
    the retry is accepted with 0.5 <= err < 1 (no growth), the rest of the interval is a short step
    whose grown size stays below dt, so the NEXT interval starts with exactly the reduced dt."""
    h, err1 = 0.2, 1.1
    tol = _independent_err(oracle_py, ref_fixture, solver, h) / err1
    dt_red = h * max(0.9 * err1 ** (-1.0 / 3.0), 0.2)
    err2 = _independent_err(oracle_py, ref_fixture, solver, dt_red) / tol
    assert 0.5 <= err2 < 1.0  # premise of the construction
    for delta, want in ((dt_red * (1 - 1e-9), 3), (dt_red * (1 + 1e-9), 4)):
        acc, rej = _accepts(oracle_py, ref_fixture, solver, tol, [0.0, h, h + delta], dt_hint=h)
        assert (acc, rej) == (want, 1), (delta, acc, rej)
