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


# ---------------------------------------------------------------------------------------------------------------
# integrate_times itself -- the loop around the controlled stepper -- written a second time, here, sharing no code
# with oracle/: Boost.Odeint's integrate_times for a controlled stepper
#   loop: t = *times++; observer(x, t); while (t_next - t > eps): cur = min(dt, t_next - t);
#         try_step(x, t, cur) == success ? dt = max(dt, cur) : dt = cur          (cur is updated by try_step)
# i.e. no step crosses an output time, dt never shrinks on success, the time restarts from the grid value at every
# output, the FSAL derivative of Dormand-Prince is carried over outputs AND over beta / kappa breakpoints (the derivative
# at t = breakpoint is evaluated with the period that ends there), Cash-Karp evaluates f(x, t) afresh at every attempt.
# Stage arithmetic: SciPy's rk_step / the generic step above (a different association than odeint's, so values agree to
# rounding and accept decisions agree unless an error value sits within ~1e-13 of 1 -- none does here).
# ---------------------------------------------------------------------------------------------------------------
def _integrate_times_second_author(f, y0, times, dt, tol, solver):
    from scipy.integrate._ivp import rk
    eps = np.finfo(np.float64).eps
    y = np.array(y0, dtype=np.float64)
    states, counts, sizes = [y.copy()], [], []
    dxdt = None                      # FSAL derivative (Dormand-Prince only), initialised at the first attempt
    for k in range(len(times) - 1):
        t, t_next = float(times[k]), float(times[k + 1])
        acc = rej = 0
        while t_next - t > eps:
            cur = min(dt, t_next - t)
            if solver == 0:
                if dxdt is None:
                    dxdt = f(t, y)
                K = np.empty((rk.RK45.n_stages + 1, y.size))
                y_new, f_new = rk.rk_step(f, t, y, dxdt, cur, rk.RK45.A, rk.RK45.B, rk.RK45.C, K)
                err = (K.T @ rk.RK45.E) * cur
                d0 = dxdt
            else:
                d0 = f(t, y)
                y_new, K = _rk_step(f, t, y, cur, CK_A, CK_B5, CK_C)
                err = cur * sum(float(b5 - b4) * kk for b5, b4, kk in zip(CK_B5, CK_B4, K))
            value = np.max(np.abs(err) / (tol + tol * (np.abs(y) + cur * np.abs(d0))))
            if value > 1.0:
                rej += 1
                dt = cur * max(0.9 * value ** (-1.0 / 3.0), 0.2)
                continue
            t += cur
            y = y_new
            if solver == 0:
                dxdt = f_new
            acc += 1
            sizes.append(cur)
            grown = cur
            if value < 0.5:
                grown = cur * 0.9 * max(5.0 ** -5, value) ** (-1.0 / 5.0)
            dt = max(dt, grown)
        states.append(y.copy())
        counts.append((acc, rej))
    return np.array(states), counts, sizes


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("tol,stride", [(1e-6, 1.0), (1e-9, 1.0), (1e-7, 2.5)])
def test_integrate_times_loop_against_a_second_implementation(oracle_py, ref_fixture, solver, tol, stride):
    """30 output intervals across the kappa breakpoint at t = 13 (and, with stride 2.5, the one at 63): accepted and
    rejected attempts PER INTERVAL (the oracle run on every prefix of the grid) and the states at every output against
    the second implementation above, both steppers, tolerances that make the controller reject (1e-9) and grow steps
    across outputs, and a non-integer stride (the remainder of an interval after a rejected step, the time restarting
    from the grid value)."""
    times = stride * np.arange(31.0)
    n_obs = len(times)
    obs = np.zeros((n_obs, ref_fixture.n))
    q = ref_fixture.with_(times=times, solver=solver, abs_err=tol, rel_err=tol, dt_hint=1.0, obs_H=obs, obs_ICU=obs, obs_D=obs)
    theta = np.asarray(q.base_theta)
    orc = oracle_py.Oracle(q)
    full = orc.eval_batch(theta[None, :], want_traj=True, nthreads=1)
    assert full["status"][0] == 0
    f = lambda t, y: orc.rhs(y, t, theta)
    states, counts, sizes = _integrate_times_second_author(f, full["traj"][0, 0], times, 1.0, tol, solver)
    # per-interval counts of the oracle: the integration up to output k does not depend on later outputs
    cum = [(0, 0)]
    for k in range(1, len(times)):
        qk = q.with_(times=times[:k + 1], obs_H=obs[:k + 1], obs_ICU=obs[:k + 1], obs_D=obs[:k + 1])
        r = oracle_py.Oracle(qk).eval_batch(theta[None, :], nthreads=1)
        cum.append((int(r["n_accept"][0]), int(r["n_reject"][0])))
    per_interval = [(a1 - a0, r1 - r0) for (a0, r0), (a1, r1) in zip(cum[:-1], cum[1:])]
    assert per_interval == counts
    assert cum[-1] == (int(full["n_accept"][0]), int(full["n_reject"][0]))
    assert sum(r for _, r in counts) > 0 or tol >= 1e-6          # the tight tolerance makes the controller reject
    assert max(a for a, _ in counts) > 1                         # some interval needs more than one step
    rel = np.abs(full["traj"][0] - states) / np.maximum(np.abs(states), 1.0)
    assert rel.max() < 1e-11, rel.max()
    assert len(sizes) == cum[-1][0]
