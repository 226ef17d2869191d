"""CPU oracle (oracle/) pinned against everything that can pin it in this container:

  * the reference's only closed-form known answer, ManualPoissonLikelihoodTest
    (tests/model/SEPAIHRDObjectivefunctionTest.cpp:688-752, tolerance 1e-8);
  * independent high-precision answers committed under tests/golden/ (SciPy DOP853 at
    rtol=1e-13 restarted at every schedule breakpoint, mpmath RHS values) -- these bound the
    restated Boost.Odeint integrator, for which the reference holds no pinned value;
  * the structural properties the reference tests for calculate() (same file :334-685).
"""
import numpy as np
import pytest


def test_manual_poisson_likelihood(oracle_py, golden):
    g = golden["manual_poisson"]
    obs, sim = np.array(g["obs"]), np.array(g["sim"])
    manual = 0.0
    for i in range(obs.shape[0]):
        for j in range(obs.shape[1]):
            s = sim[i, j] + 1e-10
            manual += obs[i, j] * np.log(s) - s
    got = oracle_py.poisson_loglik(sim, obs)
    assert abs(got - manual) < 1e-8          # the reference's own assertion
    assert abs(got - g["loglik"]) < 1e-12    # 40-digit mpmath value


def test_poisson_skips_nan_and_negative_observations(oracle_py):
    sim = np.array([[1.0, 2.0], [3.0, 4.0]])
    obs = np.array([[1.0, np.nan], [-1.0, 2.0]])
    want = (1.0 * np.log(1.0 + 1e-10) - (1.0 + 1e-10)) + (2.0 * np.log(4.0 + 1e-10) - (4.0 + 1e-10))
    assert abs(oracle_py.poisson_loglik(sim, obs) - want) < 1e-12
    # negative simulated values are clamped to 0 before the epsilon
    assert abs(oracle_py.poisson_loglik(np.array([[-5.0]]), np.array([[2.0]])) -
               (2.0 * np.log(1e-10) - 1e-10)) < 1e-9


def test_rhs_against_mpmath_spot_values(oracle_py, shipped, golden):
    orc = oracle_py.Oracle(shipped)
    for spot in golden["rhs_spots_shipped"]:
        dx = orc.rhs(np.array(spot["x"]), spot["t"], theta=shipped.base_theta)
        ref = np.array(spot["dxdt"])
        scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
        assert (np.abs(dx - ref) / scale).max() < 1e-11, spot["t"]


def test_rhs_population_conservation(oracle_py, shipped):
    orc = oracle_py.Oracle(shipped)
    n = shipped.n
    rs = np.random.RandomState(0)
    x = np.concatenate([shipped.N * 0.9] + [shipped.N * 1e-3 * rs.uniform(0.1, 1, n) for _ in range(10)])
    dx = orc.rhs(x, 40.0, theta=shipped.base_theta).reshape(11, n)
    # S..D sum to zero per age class; CumH / CumICU are bookkeeping only
    assert np.abs(dx[:9].sum(axis=0)).max() < 1e-9 * np.abs(dx[:9]).max()
    assert np.all(dx[9] >= 0) and np.all(dx[10] >= 0)


def test_schedule_boundary_rule(oracle_py, shipped):
    """t <= end keeps the OLD value at a breakpoint; t < 0 is the kappa baseline; past the last
    end the last value holds (PiecewiseConstantParameterStrategy.cpp:37-74, NPI.cpp:86-127)."""
    orc = oracle_py.Oracle(shipped)
    th = shipped.base_theta
    bv, kv = shipped.beta_values, shipped.kappa_values
    assert orc.beta_kappa(13.0, th) == (bv[0], kv[0])
    assert orc.beta_kappa(np.nextafter(13.0, 14.0), th) == (bv[1], kv[1])
    assert orc.beta_kappa(-5.0, th) == (bv[0], kv[0])
    assert orc.beta_kappa(63.0, th) == (bv[1], kv[1])
    assert orc.beta_kappa(305.0, th) == (bv[6], kv[6])
    assert orc.beta_kappa(400.0, th) == (bv[6], kv[6])


@pytest.mark.parametrize("key,fixture_name", [("shipped", "shipped"), ("reference_test_fixture", "ref_fixture"),
                                              ("shipped_perturbed", "shipped")])
@pytest.mark.parametrize("solver", [0, 1])
def test_integrator_against_high_precision(oracle_py, golden, request, key, fixture_name, solver):
    """abs = rel = 1e-6 controlled steppers vs a 1e-13 reference: agreement is limited by the
    solver tolerance (states ~1e-4 relative, log-likelihood ~1e-5..2e-3 relative where it is
    hugely negative)."""
    pb = request.getfixturevalue(fixture_name)
    pb.solver = solver
    g = golden[key]
    r = oracle_py.Oracle(pb).eval_batch(np.array(g["theta"]), want_traj=True, nthreads=1)
    assert r["status"][0] == 0
    st = r["traj"][0][g["time_index"]]
    gs = np.array(g["states"])
    assert (np.abs(st - gs) / (np.abs(gs) + 1e-3)).max() < 5e-4
    assert abs(r["loglik"][0] - g["loglik"]) / abs(g["loglik"]) < 5e-3
    if key != "shipped_perturbed":
        assert abs(r["loglik"][0] - g["loglik"]) / abs(g["loglik"]) < 2e-4
    # every daily interval costs at least one accepted step; dt_hint = 1 day
    assert r["n_accept"][0] >= pb.n_times - 1


def test_tighter_tolerance_converges_to_high_precision(oracle_py, golden, shipped):
    g = golden["shipped"]
    errs = []
    for tol in (1e-6, 1e-8, 1e-10):
        shipped.abs_err = shipped.rel_err = tol
        r = oracle_py.Oracle(shipped).eval_batch(np.array(g["theta"]), nthreads=1)
        errs.append(abs(r["loglik"][0] - g["loglik"]) / abs(g["loglik"]))
    assert errs[1] < errs[0] and errs[2] < 1e-8, errs


# ---- the reference's structural tests of calculate(), SEPAIHRDObjectivefunctionTest.cpp
def test_calculate_is_finite_repeatable_and_sensitive(oracle_py, ref_fixture):
    orc = oracle_py.Oracle(ref_fixture)
    th = ref_fixture.base_theta
    v = orc.calculate(th)
    assert np.isfinite(v)                                   # :334
    assert all(orc.calculate(th) == v for _ in range(5))    # :492 repeatability
    th2 = th.copy()
    th2[0] *= 1.5
    assert orc.calculate(th2) != v                          # :368 sensitivity


def test_zero_data_and_dense_grid(oracle_py, ref_fixture):
    pb = ref_fixture
    zero = pb.with_(obs_H=np.zeros_like(pb.obs_H), obs_ICU=np.zeros_like(pb.obs_ICU),
                    obs_D=np.zeros_like(pb.obs_D))
    v0 = oracle_py.Oracle(zero).calculate(pb.base_theta)
    assert np.isfinite(v0) and v0 <= 0.0                    # :384 (only -sim terms remain)
    t = np.arange(0, 291) * 0.1                             # :414 0.1-day grid
    dense = pb.with_(times=t, obs_H=np.ones((291, 4)), obs_ICU=np.ones((291, 4)), obs_D=np.ones((291, 4)))
    assert np.isfinite(oracle_py.Oracle(dense).calculate(pb.base_theta))


def test_streams_are_additive(oracle_py, ref_fixture):
    pb = ref_fixture
    r = oracle_py.Oracle(pb).eval_batch(pb.base_theta, nthreads=1)
    assert r["loglik"][0] == (r["ll_parts"][0, 0] + r["ll_parts"][0, 1]) + r["ll_parts"][0, 2]   # :222-225


def test_failure_sentinels(oracle_py, shipped, mm):
    pb = shipped
    pb.bounds = dict(pb.bounds)
    pb.bounds["seed_exposed"] = (5.0, 1e9)
    th = pb.base_theta.copy()
    th[pb.param_names.index("seed_exposed")] = 9e8          # E0 > N  ->  lowest()
    r = oracle_py.Oracle(pb).eval_batch(th, nthreads=1)
    assert r["status"][0] == 1 and r["loglik"][0] == mm.LOWEST
    # observation rows != number of t >= 0 points  ->  lowest()  (:176-178)
    bad = pb.with_(obs_H=pb.obs_H[:-1], obs_ICU=pb.obs_ICU[:-1], obs_D=pb.obs_D[:-1])
    r = oracle_py.Oracle(bad).eval_batch(pb.base_theta, nthreads=1)
    assert r["status"][0] == 1 and r["loglik"][0] == mm.LOWEST


def test_runup_branch_ignores_multipliers_and_runup_days(oracle_py, shipped):
    """in the shipped configuration theta[48..55] and theta[57] are inert."""
    orc = oracle_py.Oracle(shipped)
    th = shipped.base_theta.copy()
    v = orc.calculate(th)
    for name in ("E0_multiplier", "D0_multiplier", "runup_days"):
        th2 = th.copy()
        lo, hi = shipped.bounds[name]
        th2[shipped.param_names.index(name)] = 0.5 * (lo + hi)
        assert orc.calculate(th2) == v


def test_step_budget_guard(oracle_py, ref_fixture):
    """Degenerate tolerance: the trial step underflows to 0 and zero-length steps are accepted for
    ever (odeint's 500-rejection check never fires because every rejection shrinks dt by >= 5x).
    The build-side attempt budget ends the evaluation with status 3 and lowest()."""
    pb = ref_fixture.with_(abs_err=0.0, rel_err=1e-300)
    orc = oracle_py.Oracle(pb)
    orc.set_max_attempts(5000)
    r = orc.eval_batch(pb.base_theta, nthreads=1)
    assert r["status"][0] == 3 and r["n_accept"][0] + r["n_reject"][0] >= 5000


def test_cache_hash_quantisation(oracle_py):
    """SimulationCache::computeHash: theta quantised to 1e-8 (SimulationCache.cpp:35-52)."""
    a = np.array([0.123456781, 2.5])
    b = a + 3e-9
    c = a + 2e-8
    assert oracle_py.cache_hash(a) == oracle_py.cache_hash(b)
    assert oracle_py.cache_hash(a) != oracle_py.cache_hash(c)


def test_metropolis_hastings_restated(oracle_py, ref_fixture):
    orc = oracle_py.Oracle(ref_fixture)
    r1 = orc.metropolis_hastings(ref_fixture.base_theta, seed=7, iterations=300, burn_in=100,
                                 adaptation_period=50, thinning=10)
    r2 = orc.metropolis_hastings(ref_fixture.base_theta, seed=7, iterations=300, burn_in=100,
                                 adaptation_period=50, thinning=10)
    assert r1["accepted"] == r2["accepted"] and np.array_equal(r1["accept_trace"], r2["accept_trace"])
    assert len(r1["accept_trace"]) == 299 and r1["accepted"] == int(r1["accept_trace"].sum())
    assert r1["samples"].shape == (30, 5)          # t = 0 plus every 10th of 1..299
    assert r1["best_value"] >= r1["sample_values"][0]
    lo, hi, _ = ref_fixture.bounds_arrays()
    assert np.all(r1["samples"] >= lo - 1e-12) and np.all(r1["samples"] <= hi + 1e-12)   # reflect keeps bounds
    r3 = orc.metropolis_hastings(ref_fixture.base_theta, seed=8, iterations=300, burn_in=100,
                                 adaptation_period=50, thinning=10)
    assert not np.array_equal(r1["accept_trace"], r3["accept_trace"])


def test_phase1_covariance_conditioning_matches_symmetric_eigendecomposition(oracle_py, shipped):
    """ModelCalibrator.cpp:93-131 restated with a Jacobi iteration == the same formula through LAPACK."""
    P = shipped.n_params
    rs = np.random.RandomState(3)
    A = rs.randn(P, P) * 1e-3
    cov = A @ A.T * 0.5 + np.diag(np.abs(rs.randn(P)) * 1e-8)
    cov[0, 1] += 1e-9  # not exactly symmetric: the reference symmetrises first
    got = oracle_py.Oracle(shipped).condition_covariance(cov)
    w, Q = np.linalg.eigh(0.5 * (cov + cov.T))
    w = np.maximum(w, (0.1 * shipped.sigma_array()) ** 2)  # i-th smallest eigenvalue <-> parameter i
    ref = (Q * w) @ Q.T * 4.0
    ref += 1e-8 * np.trace(ref) / P * np.eye(P)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.all(np.linalg.eigvalsh(got) > 0)


def test_hill_climbing_restatement_properties(oracle_py, shipped):
    orc = oracle_py.Oracle(shipped)
    a = orc.hill_climbing(shipped.base_theta, 7, 12, cloud_size_multiplier=4, threads=2)
    b = orc.hill_climbing(shipped.base_theta, 7, 12, cloud_size_multiplier=4, threads=2)
    c = orc.hill_climbing(shipped.base_theta, 8, 12, cloud_size_multiplier=4, threads=2)
    assert np.array_equal(a["trace"], b["trace"]) and not np.array_equal(a["trace"], c["trace"])
    assert np.all(np.diff(a["trace"]) >= 0) and a["best_value"] == a["trace"][-1]
    assert np.allclose(a["final_cov"], a["final_cov"].T, rtol=0, atol=0)
    lo, hi, has = shipped.bounds_arrays()
    assert np.all(a["best"][has.astype(bool)] >= lo[has.astype(bool)]) and np.all(a["best"][has.astype(bool)] <= hi[has.astype(bool)])


def test_particle_swarm_restatement_properties(oracle_py, shipped):
    """oracle::particle_swarm (ParticleSwarmOptimizer.cpp:105-948): determinism, monotone global best, bounds,
    and that with the GLOBAL_BEST topology the serial order and the deferred order are the same search."""
    orc = oracle_py.Oracle(shipped)
    kw = dict(iterations=6, swarm_size=8)
    a = orc.particle_swarm(shipped.base_theta, 3, **kw)
    b = orc.particle_swarm(shipped.base_theta, 3, **kw)
    c = orc.particle_swarm(shipped.base_theta, 4, **kw)
    assert np.array_equal(a["trace"], b["trace"]) and not np.array_equal(a["best"], c["best"])
    assert np.all(np.diff(a["trace"]) >= 0) and a["best_value"] == a["trace"][-1]
    assert a["evaluations"] == 8 * (1 + 6)
    lo, hi, _ = shipped.bounds_arrays()
    assert np.all(a["best"] >= lo) and np.all(a["best"] <= hi)
    assert np.array_equal(a["final_cov"], a["final_cov"].T) and np.all(np.diag(a["final_cov"]) >= 1e-6)
    # warm start: particle 0 is the given vector, so the first global best is at least its value
    assert a["trace"][0] >= orc.calculate(shipped.base_theta)
    for variant in (0, 1, 3, 4):
        s = orc.particle_swarm(shipped.base_theta, 5, variant=variant, **kw)
        d = orc.particle_swarm(shipped.base_theta, 5, variant=variant, deferred_personal_bests=1, **kw)
        assert np.array_equal(s["trace"], d["trace"]) and np.array_equal(s["final_cov"], d["final_cov"]), variant
    # a ring topology reads neighbours' personal bests inside the loop: the two orders are different searches
    s = orc.particle_swarm(shipped.base_theta, 5, topology=1, iterations=12, swarm_size=8)
    d = orc.particle_swarm(shipped.base_theta, 5, topology=1, iterations=12, swarm_size=8, deferred_personal_bests=1)
    assert not np.array_equal(s["final_cov"], d["final_cov"])
    # adaptive variant: elitist learning adds up to three evaluations on iterations 0 and 5
    e = orc.particle_swarm(shipped.base_theta, 5, variant=2, use_adaptive_parameters=1, **kw)
    assert 8 * 7 + 2 <= e["evaluations"] <= 8 * 7 + 6
    # stagnation restart keeps three elite particles and re-draws the rest (one extra evaluation each)
    r = orc.particle_swarm(shipped.base_theta, 5, iterations=6, swarm_size=8, max_stagnation=1, restart_threshold=1e300)
    assert r["evaluations"] == 8 * 7 + 2 * 5 and np.all(np.diff(r["trace"]) >= 0)
    # opposition learning re-evaluates the selected swarm once
    o = orc.particle_swarm(shipped.base_theta, 5, use_opposition_learning=1, **kw)
    assert o["evaluations"] == 8 * 8


def _nuts_fixture(mm, ref_fixture):
    names = list(ref_fixture.param_names) + ["E0_multiplier", "I0_multiplier"]
    sig = dict(ref_fixture.sigmas); sig.update(E0_multiplier=0.05, I0_multiplier=0.05)
    bnd = dict(ref_fixture.bounds); bnd.update(E0_multiplier=(0.5, 1.2), I0_multiplier=(0.1, 3.0))
    theta = np.concatenate([np.asarray(ref_fixture.base_theta), [1.0, 0.8]])
    return ref_fixture.with_(param_names=names, sigmas=sig, bounds=bnd, base_theta=theta, arith=mm.ARITH_STRICT,
                             constraint_mode=1)


def test_nuts_restatement_properties(mm, oracle_py, ref_fixture):
    """oracle::nuts (NUTSSampler.cpp:41-428): determinism, bookkeeping of the reference's gradient calls (three per
    tree leaf + one per iteration + the step-size search), dual averaging only inside the adaptation window, samples
    inside the bounds, best value = max of the stored values."""
    pb = _nuts_fixture(mm, ref_fixture)
    orc = oracle_py.Oracle(pb)
    kw = dict(iterations=10, adaptation_window=4, max_tree_depth=3)
    a = orc.nuts(pb.base_theta, 3, **kw)
    b = orc.nuts(pb.base_theta, 3, **kw)
    c = orc.nuts(pb.base_theta, 4, **kw)
    assert np.array_equal(a["samples"], b["samples"]) and not np.array_equal(a["samples"], c["samples"])
    assert len(a["samples"]) == 10 and np.all((a["depth_trace"] >= 0) & (a["depth_trace"] <= 3))
    assert np.all(a["epsilon_trace"][4:] == a["epsilon_trace"][4]) and len(set(a["epsilon_trace"][:4])) == 4
    lo, hi, _ = pb.bounds_arrays()
    assert np.all(a["samples"] >= lo) and np.all(a["samples"] <= hi)
    assert a["best_value"] == a["sample_values"].max() and np.any(np.diff(a["sample_values"]) != 0)
    # a tree of depth j that runs to completion has 2^j leaves; the main loop doubles j = 0, 1, ... so an iteration
    # that reached depth d evaluated at most 2^(d+1) - 1 leaves (one more subtree may have been built and refused)
    leaves_max = sum(2 ** (int(d) + 1) - 1 + 2 ** (int(d) + 1) for d in a["depth_trace"])
    assert 10 + 3 <= a["gradient_calls"] <= 10 + 3 * leaves_max + 3 * 6 + 1
    # the stored value is the objective at the stored (constrained) sample
    np.testing.assert_allclose(a["sample_values"][-1], orc.calculate(a["samples"][-1]), rtol=1e-12)


def test_running_comoment_covariance_against_the_two_pass_form(mm, oracle_py, shipped):
    """oracle::RunningMoments (the recurrence host library and device kernels follow) against the literal restatement
    of recomputeFullCovariance (MetropolisHastingsSampler.cpp:168-199) inside the same sampler: covariances equal to
    1e-12 of the largest entry after six refreshes, accept traces equal, samples equal to rounding."""
    pb = shipped.with_(constraint_mode=1)
    orc = oracle_py.Oracle(pb)
    x0 = orc.jitter_draws(pb.base_theta, 3, 1, mode=1)[0]
    kw = dict(adaptation_period=50, thinning=5)
    a = orc.metropolis_hastings(x0, 17, 420, 100, **kw)
    b = orc.metropolis_hastings(x0, 17, 420, 100, two_pass_covariance=True, **kw)
    assert 0 < a["accepted"] < 419
    assert np.array_equal(a["accept_trace"], b["accept_trace"])
    assert np.abs(a["final_cov"] - b["final_cov"]).max() <= 1e-12 * np.abs(b["final_cov"]).max()
    np.testing.assert_allclose(a["samples"], b["samples"], rtol=1e-10, atol=1e-13)
    assert np.array_equal(a["final_cov"], a["final_cov"].T)
