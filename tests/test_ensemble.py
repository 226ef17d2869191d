"""Posterior-ensemble summaries: CPU restatement checks (-m "not gpu") and
HIP-vs-oracle parity (-m gpu).

Reference behaviour: ResultAggregator.cpp:297-345 (daily / cumulative incidence on t >= 0),
MetricsCalculator.cpp:199-226 (seroprevalence), PostCalibrationAnalyser.cpp:303-340 (exact-sort
quantile rule), SimulationRunner.cpp:24-104 (simulation from the given initial state).
"""
import numpy as np
import pytest

PROBS = [0.025, 0.05, 0.5, 0.95, 0.975]  # ResultAggregator.cpp:224, PostCalibrationAnalyser.cpp:306


def _draws(oracle_py, pb, S, seed0=11):
    return oracle_py.Oracle(pb).jitter_draws(pb.base_theta, seed0, S, mode=1)


def test_oracle_single_sample_is_its_own_quantile(oracle_py, shipped):
    orc = oracle_py.Oracle(shipped)
    theta = np.array(shipped.base_theta)[None, :]
    r = orc.ensemble_quantiles(theta, PROBS, nthreads=1)
    assert r["n_valid"] == 1 and r["status"][0] == 0
    ppc, sero = r["ppc"], r["sero"]
    for p in range(1, len(PROBS)):
        assert np.array_equal(ppc[:, p], ppc[:, 0]) and np.array_equal(sero[p], sero[0])
    daily, cumulative = ppc[0:3, 0], ppc[3:6, 0]
    assert (daily >= 0).all()
    # cumulative = running sums of the daily values in time order (ResultAggregator.cpp:324-335)
    run = np.zeros_like(daily[:, 0])
    for t in range(daily.shape[1]):
        run = daily[:, t] if t == 0 else run + daily[:, t]
        assert np.array_equal(cumulative[:, t], run)
    assert (sero[0] > 0).all() and (sero[0] < 1).all()
    assert (np.diff(sero[0]) > -1e-12).all()  # S only decreases in this model


def test_oracle_quantile_rule_is_linear_interpolation(oracle_py, shipped):
    """pos = q (n - 1), linear interpolation == numpy's default ('linear') definition."""
    S = 23
    theta = _draws(oracle_py, shipped, S)
    orc = oracle_py.Oracle(shipped)
    r = orc.ensemble_quantiles(theta, PROBS)
    assert r["n_valid"] == S
    n, T = shipped.n, shipped.n_times
    times = np.asarray(shipped.times)
    pos = np.nonzero(times >= 0)[0]
    # rebuild the per-sample series from trajectories of the same fixed-state runs is not exposed;
    # check order statistics instead: quantiles are non-decreasing in p and bracketed by min / max
    ext = orc.ensemble_quantiles(theta, [0.0, 1.0])
    assert (np.diff(r["ppc"], axis=1) >= 0).all() and (np.diff(r["sero"], axis=0) >= 0).all()
    assert (r["ppc"] >= ext["ppc"][:, :1]).all() and (r["ppc"] <= ext["ppc"][:, 1:]).all()
    assert r["ppc"].shape == (6, len(PROBS), len(pos), n) and r["sero"].shape == (len(PROBS), T)
    # a permutation of the samples changes nothing (exact sort, unlike the reference's P^2 estimator)
    perm = np.random.RandomState(0).permutation(S)
    r2 = orc.ensemble_quantiles(theta[perm], PROBS)
    assert np.array_equal(r2["ppc"], r["ppc"]) and np.array_equal(r2["sero"], r["sero"])


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("S", [1, 37, 200])
def test_hip_ensemble_matches_oracle(mm, oracle_py, shipped, S, solver):
    pb = shipped
    pb.solver = solver
    pb.arith = mm.ARITH_STRICT
    theta = _draws(oracle_py, pb, S)
    ref = oracle_py.Oracle(pb).ensemble_quantiles(theta, PROBS)
    hip = mm.HipObjective(pb)
    hip.set_initial_state_mode(1)
    got = hip.ensemble_quantiles(theta, PROBS)
    assert np.array_equal(got["status"], ref["status"]) and got["n_valid"] == ref["n_valid"] == S
    # strict arithmetic: trajectories agree to ~5e-14 relative, so do their order statistics
    np.testing.assert_allclose(got["ppc"], ref["ppc"], rtol=1e-9, atol=1e-9)
    # (sum N - sum S) / sum N cancels at early times: the bar is absolute on the S / N scale
    np.testing.assert_allclose(got["sero"], ref["sero"], rtol=1e-9, atol=1e-12)
    # the objective path is untouched by the ensemble mode of ANOTHER ctx
    plain = mm.HipObjective(pb).eval_batch(theta[:4])
    ref_ll = oracle_py.Oracle(pb).eval_batch(theta[:4])
    np.testing.assert_allclose(plain["loglik"], ref_ll["loglik"], rtol=1e-10)


@pytest.mark.gpu
def test_hip_ensemble_skips_failed_samples(mm, oracle_py, shipped):
    """Samples whose integration fails (here: the attempt budget, status 3) are skipped on both sides
    like the reference's `if (!sim_result.isValid()) continue`, and the quantile positions use the
    count of valid samples."""
    S = 50
    theta = _draws(oracle_py, shipped, S)
    probe = mm.HipObjective(shipped.with_(arith=mm.ARITH_STRICT))
    probe.set_initial_state_mode(1)
    r = probe.eval_batch(theta)
    attempts = r["n_accept"] + r["n_reject"]
    budget = int(np.sort(attempts)[S // 2])  # about half of the samples run out of attempts
    pb = shipped.with_(arith=mm.ARITH_STRICT, max_attempts=budget)
    orc = oracle_py.Oracle(pb)
    orc.set_max_attempts(budget)
    ref = orc.ensemble_quantiles(theta, PROBS)
    hip = mm.HipObjective(pb)
    hip.set_initial_state_mode(1)
    got = hip.ensemble_quantiles(theta, PROBS)
    assert 0 < ref["n_valid"] < S and got["n_valid"] == ref["n_valid"]
    assert np.array_equal(got["status"], ref["status"])
    np.testing.assert_allclose(got["ppc"], ref["ppc"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(got["sero"], ref["sero"], rtol=1e-9, atol=1e-12)


@pytest.mark.gpu
def test_hip_ensemble_argument_checks(mm, shipped):
    hip = mm.HipObjective(shipped)
    hip.set_initial_state_mode(1)
    theta = np.tile(np.array(shipped.base_theta), (3, 1))
    with pytest.raises(RuntimeError):
        hip.ensemble_quantiles(theta, [0.5, 1.5])
    with pytest.raises(RuntimeError):
        hip.set_initial_state_mode(7)


@pytest.mark.gpu
def test_host_posterior_ensemble_mirror(mm, oracle_py, shipped):
    """C++ HipPosteriorEnsemble (ResultAggregator::aggregatePosteriorPredictives shape): same sample
    selection as the reference's mt19937 + uniform_int_distribution, same numbers as the oracle."""
    pb = shipped.with_(arith=mm.ARITH_STRICT)
    n_samples, num_for_ppc, seed = 120, 40, 2024
    samples = _draws(oracle_py, pb, n_samples, seed0=77)
    host = mm.HostObjective(pb)
    got = host.posterior_ensemble(samples, num_for_ppc, seed, burn_in=20, thinning=3, want_rt=True)
    sel = oracle_py.ppc_select(n_samples, num_for_ppc, seed)
    assert np.array_equal(got["selected"], sel) and len(sel) == num_for_ppc
    orc = oracle_py.Oracle(pb)
    ref = orc.ensemble_quantiles(samples[sel], PROBS)
    assert got["samples_used"] == ref["n_valid"] == num_for_ppc
    np.testing.assert_allclose(got["ppc"], ref["ppc"], rtol=1e-9, atol=1e-9)
    ref_sero = orc.ensemble_quantiles(samples[20::3], PROBS)["sero"]
    np.testing.assert_allclose(got["sero"], ref_sero, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got["rt"], _rt_reference(oracle_py, pb, samples[20::3])[1], rtol=1e-9)
    # num_for_ppc >= size: every sample once, in order (ResultAggregator.cpp:263-266)
    allsel = host.posterior_ensemble(samples[:10], 50, seed, want_sero=False)["selected"]
    assert np.array_equal(allsel, np.arange(10))


def test_ppc_sample_selection_rule(oracle_py):
    sel = oracle_py.ppc_select(1000, 25, 12345)
    assert sel.shape == (25,) and sel.min() >= 0 and sel.max() < 1000
    assert np.array_equal(sel, oracle_py.ppc_select(1000, 25, 12345))
    assert not np.array_equal(sel, oracle_py.ppc_select(1000, 25, 12346))
    assert np.array_equal(oracle_py.ppc_select(7, 0, 1), np.arange(7))
    assert np.array_equal(oracle_py.ppc_select(7, 7, 1), np.arange(7))


def _rt_reference(oracle_py, pb, theta):
    import rt_numpy
    orc = oracle_py.Oracle(pb)
    sim = orc.simulate_samples(theta)
    assert np.all(sim["status"] == 0)
    rts = np.array([rt_numpy.rt_trajectory(sim["traj"][s], orc.model_parameters(theta[s]), pb) for s in range(len(theta))])
    q = np.array([[rt_numpy.sorted_quantile(rts[:, k], p) for k in range(rts.shape[1])] for p in PROBS])
    return rts, q


def test_rt_spectrum_lives_in_the_exposed_block(oracle_py, shipped):
    """The next-generation matrix F V^-1 (4n x 4n, ReproductionNumberCalculator.cpp:55-171) has non-zero
    rows only for E: its spectral radius is the Perron root of the n x n block
    T_ij (1/gamma_p + p_j/gamma_A + theta (1 - p_j)/(gamma_I + h_j)) the device path iterates on."""
    import rt_numpy
    pb = shipped
    orc = oracle_py.Oracle(pb)
    theta = _draws(oracle_py, pb, 3)
    sim = orc.simulate_samples(theta)
    n = pb.n
    N, M = np.asarray(pb.N), np.asarray(pb.M).reshape(n, n)
    for s in range(3):
        mp = orc.model_parameters(theta[s])
        for k in (0, 25, 140, pb.n_times - 1):
            t, S = pb.times[k], sim["traj"][s, k, :n]
            full = rt_numpy.rt_value(S, t, mp, pb)
            kappa = mp["kappa_values"][0] if t < 0 else rt_numpy._piecewise(pb.kappa_end_times, mp["kappa_values"], t)
            beta = rt_numpy._piecewise(pb.beta_end_times, mp["beta_values"], t) if len(mp["beta_values"]) else mp["beta"]
            T = np.maximum(0.0, beta * kappa * M * mp["a"][:, None] * mp["h_infec"][None, :] * S[:, None] / N[None, :])
            dwell = 1.0 / mp["gamma_p"] + mp["p"] / mp["gamma_A"] + mp["theta"] * (1.0 - mp["p"]) / (mp["gamma_I"] + mp["h"])
            block = np.max(np.abs(np.linalg.eigvals(T * dwell[None, :])))
            assert abs(block - full) <= 1e-12 * full, (s, k, block, full)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [0, 1])
def test_hip_rt_trajectory_quantiles(mm, oracle_py, shipped, solver):
    """Rt quantiles of the device ensemble against the numpy restatement of ReproductionNumberCalculator
    (F, V, inverse, eigenvalues through LAPACK) on the oracle's trajectories."""
    pb = shipped.with_(solver=solver, arith=mm.ARITH_STRICT)
    S = 25
    theta = _draws(oracle_py, pb, S)
    rts, q = _rt_reference(oracle_py, pb, theta)
    assert 0.05 < rts.min() < 1.0 < rts.max() < 10.0  # an epidemic that grows and is brought under control
    hip = mm.HipObjective(pb)
    hip.set_initial_state_mode(1)
    got = hip.ensemble_quantiles(theta, PROBS, want_sero=True, want_rt=True)
    assert got["n_valid"] == S
    np.testing.assert_allclose(got["rt"], q, rtol=1e-9)
    # the other summaries are unchanged by asking for Rt
    plain = hip.ensemble_quantiles(theta, PROBS, want_sero=True)
    assert np.array_equal(plain["ppc"], got["ppc"]) and np.array_equal(plain["sero"], got["sero"])
    only_rt = hip.ensemble_quantiles(theta, PROBS, want_sero=False, want_rt=True)
    assert np.array_equal(only_rt["rt"], got["rt"])


@pytest.mark.gpu
def test_hip_essential_metrics_table(mm, oracle_py, shipped):
    """Per-sample EssentialMetrics rows of the device ensemble against the numpy restatement of
    MetricsCalculator::calculateEssentialMetrics on the oracle's trajectories."""
    import rt_numpy
    pb = shipped.with_(arith=mm.ARITH_STRICT)
    S = 12
    theta = _draws(oracle_py, pb, S)
    orc = oracle_py.Oracle(pb)
    sim = orc.simulate_samples(theta)
    ref = np.array([rt_numpy.essential_metrics(sim["traj"][s], orc.model_parameters(theta[s]), pb) for s in range(S)])
    hip = mm.HipObjective(pb)
    hip.set_initial_state_mode(1)
    got = hip.ensemble_quantiles(theta, PROBS, want_sero=True, want_rt=True, want_metrics=True)["metrics"]
    assert got.shape == ref.shape == (S, 12 + 4 * pb.n)
    # columns 5, 6 are output times (exact); the rest are sums over 326 output rows of 1e-13-accurate states
    assert np.array_equal(got[:, 5:7], ref[:, 5:7])
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12)
    assert np.all(ref[:, 0] > 1.0) and np.all((ref[:, 11] > 0) & (ref[:, 11] < 0.2))  # R0 > 1, plausible seroprevalence
    only = hip.ensemble_quantiles(theta, PROBS, want_sero=False, want_rt=False, want_metrics=True)["metrics"]
    assert np.array_equal(only, got)


@pytest.mark.gpu
def test_hip_ensemble_larger_than_the_lds_sort(mm, oracle_py, ref_fixture):
    """More than 16 384 samples: segments are sorted in global memory (library segmented radix sort) instead of
    LDS; same quantile rule, same numbers as the oracle."""
    pb = ref_fixture.with_(arith=mm.ARITH_STRICT)
    S = 16384 + 3000 + 7   # not a multiple of 64
    base = _draws(oracle_py, pb, 4096, seed0=5)
    theta = np.tile(base, (S // 4096 + 1, 1))[:S]
    theta[:, 0] *= 1.0 + 1e-3 * np.arange(S) / S   # make the tiled samples distinct
    ref = oracle_py.Oracle(pb).ensemble_quantiles(theta, PROBS)
    hip = mm.HipObjective(pb)
    hip.set_initial_state_mode(1)
    got = hip.ensemble_quantiles(theta, PROBS, want_sero=True, want_rt=True)
    assert got["n_valid"] == ref["n_valid"] == S
    np.testing.assert_allclose(got["ppc"], ref["ppc"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(got["sero"], ref["sero"], rtol=1e-9, atol=1e-12)
    assert np.all(np.diff(got["rt"], axis=0) >= 0)
