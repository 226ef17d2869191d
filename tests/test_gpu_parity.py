"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Tolerances (fp64):
  * strict arithmetic: the kernel performs the oracle's operation sequence; the only
    differences are libm (pow in the step-size controller after a rejected step, log in
    the likelihood).  Accepted/rejected step counts must be identical, trajectories agree
    to 1e-9 relative (north-star bar: 1e-6), log-likelihood to 1e-10 relative.
  * fma arithmetic: contraction changes roundings; north-star bar 1e-6 relative on states.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL_STATE_BAR = 1e-6  # BASELINE.json north_star: trajectory states within 1e-6 relative


def rel_state_err(a, b, pb):
    """relative error per state entry, scaled by max(|ref|, abs_tol-level floor of 1 person)"""
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


@pytest.fixture(scope="module")
def draws(mm, oracle_py):
    def make(pb, B, seed0=1):
        return oracle_py.Oracle(pb).jitter_draws(pb.base_theta, seed0, B, mode=1)
    return make


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("fixture_name", ["shipped", "ref_fixture", "synth400"])
def test_strict_matches_oracle(mm, oracle_py, request, fixture_name, solver, draws):
    pb = request.getfixturevalue(fixture_name)
    pb.solver = solver
    pb.arith = mm.ARITH_STRICT
    B = 37 if fixture_name != "ref_fixture" else 21  # ragged: not a multiple of 16 chains/wave
    if fixture_name == "ref_fixture":
        rs = np.random.RandomState(5)
        lo, hi, _ = pb.bounds_arrays()
        theta = lo + (hi - lo) * rs.uniform(0, 1, (B, pb.n_params))
        theta[0] = pb.base_theta
    else:
        theta = draws(pb, B)
        theta[0] = pb.base_theta
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    hip = mm.HipObjective(pb)
    got = hip.eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["n_accept"], ref["n_accept"]), (got["n_accept"], ref["n_accept"])
    assert np.array_equal(got["n_reject"], ref["n_reject"])
    err = rel_state_err(got["traj"], ref["traj"], pb).max()
    assert err < 1e-9, err
    assert err < REL_STATE_BAR
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)
    np.testing.assert_allclose(got["ll_parts"], ref["ll_parts"], rtol=1e-10)
    # likelihood-only launch gives the same numbers as the trajectory launch
    again = hip.eval_batch(theta)
    assert np.array_equal(again["loglik"], got["loglik"])


@pytest.mark.parametrize("solver", [0, 1])
def test_fma_within_north_star_tolerance(mm, oracle_py, synth400, solver, draws):
    pb = synth400
    pb.solver = solver
    pb.arith = mm.ARITH_FMA
    theta = draws(pb, 64)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"])
    err = rel_state_err(got["traj"], ref["traj"], pb).max()
    assert err < REL_STATE_BAR, err
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-7)
    same_steps = np.mean((got["n_accept"] == ref["n_accept"]) & (got["n_reject"] == ref["n_reject"]))
    assert same_steps > 0.9


def test_golden_highprec(mm, golden, shipped):
    """HIP path against the independent high-precision answers (solver-tolerance limited)."""
    g = golden["shipped"]
    got = mm.HipObjective(shipped).eval_batch(np.array(g["theta"])[None, :], want_traj=True)
    assert abs(got["loglik"][0] - g["loglik"]) / abs(g["loglik"]) < 5e-5
    st = got["traj"][0][g["time_index"]]
    gs = np.array(g["states"])
    assert (np.abs(st - gs) / (np.abs(gs) + 1.0)).max() < 1e-3


def test_failure_sentinels(mm, oracle_py, shipped):
    """calculate() returns lowest() for a seed larger than the population (S < 0 branch,
    SEPAIHRDObjectiveFunction.cpp:155-163) -- same chain indices flagged as the oracle."""
    pb = shipped
    pb.bounds = dict(pb.bounds)
    pb.bounds["seed_exposed"] = (5.0, 1e9)
    theta = np.tile(pb.base_theta, (5, 1))
    k = pb.param_names.index("seed_exposed")
    theta[1, k] = 9e8
    theta[3, k] = 2e8
    ref = oracle_py.Oracle(pb).eval_batch(theta)
    got = mm.HipObjective(pb).eval_batch(theta)
    assert np.array_equal(got["status"], ref["status"])
    assert got["status"].tolist() == [0, 1, 0, 1, 0]
    assert got["loglik"][1] == mm.LOWEST and got["loglik"][3] == mm.LOWEST
    np.testing.assert_allclose(got["loglik"][[0, 2, 4]], ref["loglik"][[0, 2, 4]], rtol=1e-10)


def test_nan_observations_skipped_and_clamp_mode(mm, oracle_py, ref_fixture):
    pb = ref_fixture
    pb.obs_H = pb.obs_H.copy()
    pb.obs_H[3, 1] = np.nan
    pb.obs_H[7, 2] = -1.0
    rs = np.random.RandomState(11)
    lo, hi, _ = pb.bounds_arrays()
    theta = lo + (hi - lo) * rs.uniform(-0.3, 1.3, (16, pb.n_params))  # outside bounds: clamp acts
    ref = oracle_py.Oracle(pb).eval_batch(theta)
    got = mm.HipObjective(pb).eval_batch(theta)
    assert np.all(np.isfinite(got["loglik"]))
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)


def test_step_budget_guard_matches_oracle(mm, oracle_py, ref_fixture):
    """Degenerate tolerance -> zero-length steps for ever; both sides stop at the attempt budget."""
    pb = ref_fixture.with_(abs_err=0.0, rel_err=1e-300, max_attempts=3000)
    theta = np.tile(pb.base_theta, (3, 1))
    orc = oracle_py.Oracle(pb)
    orc.set_max_attempts(3000)
    ref = orc.eval_batch(theta)
    got = mm.HipObjective(pb).eval_batch(theta)
    assert got["status"].tolist() == [3, 3, 3] == ref["status"].tolist()
    assert np.all(got["loglik"] == mm.LOWEST)
    assert np.array_equal(got["n_accept"] + got["n_reject"], ref["n_accept"] + ref["n_reject"])


def test_device_pointer_entry_point(mm, oracle_py, synth400, draws):
    import torch
    pb = synth400
    theta = draws(pb, 100)
    hip = mm.HipObjective(pb)
    host = hip.eval_batch(theta)
    d_theta = torch.from_numpy(theta).cuda()
    d_ll = torch.empty(100, dtype=torch.float64, device="cuda")
    d_st = torch.empty(100, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    hip.eval_batch_device(d_theta, d_ll, d_status=d_st, stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_ll.cpu().numpy(), host["loglik"])
    assert np.array_equal(d_st.cpu().numpy(), host["status"])


def test_full_size_properties(mm, synth400, draws):
    """BASELINE config 2 size (4096 chains, 400 days): size-independent properties --
    duplicated chains give bit-identical results wherever they sit in the batch, every
    chain finishes, step counts are >= the 400 daily intervals."""
    pb = synth400
    base = draws(pb, 512)
    theta = np.tile(base, (8, 1))
    got = mm.HipObjective(pb).eval_batch(theta)
    ll = got["loglik"].reshape(8, 512)
    assert np.all(got["status"] == 0)
    assert np.all(ll == ll[0:1])
    assert np.all(got["n_accept"] >= 400)


@pytest.mark.parametrize("n_age", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("solver", [0, 1])
def test_other_age_class_counts(mm, oracle_py, shipped, n_age, solver):
    """Lane layouts other than 4 lanes per chain: n = 1, 2 (narrow groups), n = 3 (padded to 4 lanes),
    n = 8, 16 (wave-shuffle contraction; 16 is BASELINE config 5's shape)."""
    from mmid_amd import draws
    if n_age <= 3:
        pb = mm.restrict_age_classes(shipped, list(range(n_age)))
    else:
        pb = mm.widen_age_classes(shipped, n_age // 4)
    pb.solver = solver
    pb.arith = mm.ARITH_STRICT
    pb.times = pb.times[:120]
    pb = pb.with_(obs_H=pb.obs_H[:100], obs_ICU=pb.obs_ICU[:100], obs_D=pb.obs_D[:100])
    B = 19
    theta = draws.jitter_draws(pb, 3, B)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"]) and np.all(ref["status"] == 0)
    assert np.array_equal(got["n_accept"], ref["n_accept"]) and np.array_equal(got["n_reject"], ref["n_reject"])
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < 1e-9
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)


@pytest.mark.parametrize("n_age", [1, 2, 3, 8, 16])
@pytest.mark.parametrize("solver", [0, 1])
def test_other_age_class_counts_tolerance_arithmetic(mm, oracle_py, shipped, n_age, solver):
    """The same layouts in the production (fma) arithmetic, which has its own code paths there: n = 3 runs the
    16-lanes-per-chain small-batch kernel with a padded age class, n = 8 and 16 form the contact sum with
    v_fmac_f64_dpp row broadcasts (bank-masked for two 8-lane chains per row).  North-star bar 1e-6 relative on the
    states; measured ~1e-11."""
    from mmid_amd import draws
    if n_age <= 3:
        pb = mm.restrict_age_classes(shipped, list(range(n_age)))
    else:
        pb = mm.widen_age_classes(shipped, n_age // 4)
    pb.solver = solver
    pb.arith = mm.ARITH_FMA
    pb.times = pb.times[:120]
    pb = pb.with_(obs_H=pb.obs_H[:100], obs_ICU=pb.obs_ICU[:100], obs_D=pb.obs_D[:100])
    B = 19
    theta = draws.jitter_draws(pb, 3, B)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"]) and np.all(ref["status"] == 0)
    same = (got["n_accept"] == ref["n_accept"]) & (got["n_reject"] == ref["n_reject"])
    assert same.mean() >= 0.9
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < 1e-8
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-8)


@pytest.mark.parametrize("arith", ["strict", "fma"])
@pytest.mark.parametrize("solver,B", [(0, 20000), (0, 32768 + 37), (1, 20000), (1, 32768 + 37)])
def test_saturating_batches_use_the_same_arithmetic(mm, oracle_py, synth400, draws, solver, B, arith):
    """BASELINE config 3 / 4 sizes per GPU (and config 3's Cash-Karp): batches of more than 1024 waves run
    the inline-likelihood kernel (Cash-Karp from 2048 waves on: its two-waves-per-SIMD build).  The same
    chains evaluated in chunks of 4096 (separate likelihood pass, one wave per SIMD) must give the SAME
    bits -- log-likelihood, status and step counts -- and a sample of them is checked against the oracle."""
    pb = synth400.with_(solver=solver, arith=mm.ARITH_STRICT if arith == "strict" else mm.ARITH_FMA)
    base = draws(pb, 4096)
    theta = np.tile(base, ((B + 4095) // 4096, 1))[:B]
    hip = mm.HipObjective(pb)
    big = hip.eval_batch(theta)
    assert np.all(big["status"] == 0)
    for off in range(0, B, 4096):
        part = hip.eval_batch(theta[off:off + 4096])
        for k in ("loglik", "status", "n_accept", "n_reject"):
            assert np.array_equal(part[k], big[k][off:off + 4096]), (k, off)
    if arith == "strict":
        idx = np.linspace(0, B - 1, 24).astype(int)
        ref = oracle_py.Oracle(pb).eval_batch(theta[idx])
        assert np.array_equal(big["n_accept"][idx], ref["n_accept"]) and np.array_equal(big["n_reject"][idx], ref["n_reject"])
        np.testing.assert_allclose(big["loglik"][idx], ref["loglik"], rtol=1e-10)


def test_sixteen_age_classes_saturating_batch(mm, oracle_py, shipped):
    """BASELINE config 5's lane layout (16 lanes per chain, 4 chains per wave) in a batch of more than
    1024 waves: inline-likelihood kernel == chunks through the separate pass == oracle on a sample."""
    from mmid_amd import draws as dr
    pb = mm.widen_age_classes(shipped, 4)
    pb.arith = mm.ARITH_STRICT
    pb.times = pb.times[:80]
    pb = pb.with_(obs_H=pb.obs_H[:60], obs_ICU=pb.obs_ICU[:60], obs_D=pb.obs_D[:60])
    B = 4 * 1100 + 3
    theta = np.tile(dr.jitter_draws(pb, 5, 512), (B // 512 + 1, 1))[:B]
    hip = mm.HipObjective(pb)
    big = hip.eval_batch(theta)
    assert np.all(big["status"] == 0)
    for off in range(0, B, 2048):
        part = hip.eval_batch(theta[off:off + 2048])
        for k in ("loglik", "n_accept", "n_reject"):
            assert np.array_equal(part[k], big[k][off:off + 2048]), (k, off)
    idx = np.linspace(0, B - 1, 12).astype(int)
    ref = oracle_py.Oracle(pb).eval_batch(theta[idx])
    assert np.array_equal(big["n_accept"][idx], ref["n_accept"])
    np.testing.assert_allclose(big["loglik"][idx], ref["loglik"], rtol=1e-10)


def test_workspace_budget_chunks_give_the_same_bits(mm, synth400, draws, monkeypatch):
    """A batch whose likelihood workspace exceeds the budget is evaluated in chunks of chains on the same
    stream (default budget 24 GiB; SEPAIHRD_WORKSPACE_MB at context creation overrides it)."""
    pb = synth400.with_(arith=mm.ARITH_FMA)
    theta = draws(pb, 3000)                      # 3000 x 9.6 KB = 29 MB of workspace
    whole = mm.HipObjective(pb).eval_batch(theta)
    monkeypatch.setenv("SEPAIHRD_WORKSPACE_MB", "4")   # ~430 chains per chunk -> 7 chunks, the last one ragged
    chunked = mm.HipObjective(pb)
    monkeypatch.delenv("SEPAIHRD_WORKSPACE_MB")
    got = chunked.eval_batch(theta)
    for k in ("loglik", "status", "n_accept", "n_reject", "ll_parts"):
        assert np.array_equal(got[k], whole[k]), k
    traj = chunked.eval_batch(theta[:900], want_traj=True)["traj"]
    assert np.array_equal(traj, mm.HipObjective(pb).eval_batch(theta[:900], want_traj=True)["traj"])


def test_device_entry_point_is_graph_capturable(mm, synth400, draws):
    """sepaihrd_eval_batch_device neither synchronises nor allocates once sepaihrd_reserve has covered the batch:
    its three launches can be captured into a HIP graph and replayed on new inputs in the same buffers."""
    import torch
    pb = synth400.with_(arith=mm.ARITH_FMA)
    B = 512
    th_a, th_b = draws(pb, B, seed0=1), draws(pb, B, seed0=5000)
    hip = mm.HipObjective(pb)
    want_a, want_b = hip.eval_batch(th_a)["loglik"], hip.eval_batch(th_b)["loglik"]
    hip.reserve(B)
    d_theta = torch.from_numpy(th_a).cuda()
    d_ll = torch.zeros(B, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip.eval_batch_device(d_theta, d_ll, stream=side.cuda_stream)  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hip.eval_batch_device(d_theta, d_ll, stream=torch.cuda.current_stream().cuda_stream)
    d_ll.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(d_ll.cpu().numpy(), want_a)
    d_theta.copy_(torch.from_numpy(th_b).cuda())
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(d_ll.cpu().numpy(), want_b)


@pytest.mark.parametrize("T", [1, 2, 3])
@pytest.mark.parametrize("solver", [0, 1])
def test_shortest_grids(mm, oracle_py, ref_fixture, T, solver):
    """One, two and three output times (the observer fires once at t0; one interval = the first attempt is one
    step of the whole interval): same status, counters, states and likelihood as the oracle."""
    times = np.array([0.0, 0.37, 1.0])[:T]
    pb = ref_fixture.with_(times=times, solver=solver, arith=mm.ARITH_STRICT, obs_H=ref_fixture.obs_H[:T],
                           obs_ICU=ref_fixture.obs_ICU[:T], obs_D=ref_fixture.obs_D[:T])
    rs = np.random.RandomState(T)
    lo, hi, _ = pb.bounds_arrays()
    theta = lo + (hi - lo) * rs.uniform(0, 1, (9, pb.n_params))
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["n_accept"], ref["n_accept"]) and np.array_equal(got["n_reject"], ref["n_reject"])
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < 1e-12
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)


def test_minimal_problem_one_parameter_no_schedule(mm, oracle_py, ref_fixture):
    """One calibrated parameter, a single kappa period (no breakpoint inside the run), one age class."""
    pb = mm.restrict_age_classes(ref_fixture, [1])
    pb = pb.with_(param_names=["beta"], sigmas={"beta": 0.01}, bounds={"beta": (0.01, 1.0)}, base_theta=np.array([0.06]),
                  kappa_end_times=np.array([1000.0]), kappa_values=np.array([0.8]), npi_names=[], arith=mm.ARITH_STRICT)
    theta = np.linspace(0.02, 0.4, 23)[:, None]
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"]) and np.all(ref["status"] == 0)
    assert np.array_equal(got["n_accept"], ref["n_accept"]) and np.array_equal(got["n_reject"], ref["n_reject"])
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < 1e-9
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)
    assert np.ptp(got["loglik"]) > 0  # the one parameter matters


@pytest.mark.gpu
@pytest.mark.parametrize("arith", ["fma", "strict"])
@pytest.mark.parametrize("problem,solver,chains", [("synth_400d_n4.json", 0, 1021), ("synth_400d_n4.json", 1, 1021),
                                                   ("shipped_problem.json", 0, 37)])
def test_small_batch_kernel_gives_the_same_bits(problem, solver, chains, arith):
    """Up to 4096 chains a 4-age problem runs the 16-lanes-per-chain form of the integrator
    (csrc/sepaihrd_lane_split.inc), in both arithmetic builds; sepaihrd_set_integrator_form forces either form.  Same
    chains through both (and, when the experiment build exists, through its one-wavefront-per-chain kernel of
    csrc/sepaihrd_wave_chain.inc): log-likelihood, status, step counts and every
    trajectory state are the same bits -- a chain's result does not depend on the batch it was evaluated in.
    1021 and 37 chains leave a ragged last wave in both layouts."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "compare_lane_split.py")
    r = subprocess.run([sys.executable, tool, "--problem", problem, "--solver", str(solver), "--chains", str(chains),
                        "--arith", arith], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "16-lane traj: identical=True" in r.stdout and "16-lane n_accept: identical=True" in r.stdout
    if arith == "fma" and os.path.exists(os.path.join(os.path.dirname(tool), "libsepaihrd_hip_experiments.so")):
        assert "wave-per-chain traj: identical=True" in r.stdout and "wave-per-chain loglik: identical=True" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("arith,solver,seed", [("fma", 0, 7), ("fma", 1, 8), ("strict", 0, 9), ("strict", 1, 10)])
def test_small_batch_kernel_fuzz(arith, solver, seed):
    """The two forms of the integrator against each other on 24 random variants of the shipped problem: 3 or 4 age
    classes (3 pads a lane), output grids of random length and stride, tolerances 1e-8 .. 1e-4, attempt budgets
    that cut chains short (status 3), both constraint modes, draws beyond the bounds, 1 .. 69 chains.  Every output
    array, trajectories included, must be the same bits."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "compare_lane_split.py")
    r = subprocess.run([sys.executable, tool, "--problem", "shipped_problem.json", "--arith", arith, "--solver", str(solver),
                        "--fuzz", str(seed)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all arrays identical=True" in r.stdout, r.stdout + r.stderr


# ---------------------------------------------------------------------------------------------------------------
# BASELINE-size parity (round 2): the whole headline batch and the configs[4] workload against the oracle
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("solver", [0, 1])
def test_headline_batch_matches_oracle_chain_by_chain(mm, oracle_py, solver):
    """BASELINE configs[1] exactly as bench.py runs it (4096 jittered chains, 400 days, T = 401; solver 1 = the
    Cash-Karp stepper of configs[2] on the same batch), strict arithmetic: EVERY chain has the oracle's accepted /
    rejected step counts and status, every trajectory state agrees to 1e-9 relative (north-star bar 1e-6), every
    log-likelihood to 1e-8 relative (the daily increments of the cumulative compartments amplify the 1e-13 state
    differences; measured 1.3e-9)."""
    from mmid_amd import draws as dr
    pb = mm.workloads.build("c1", os.path.join(os.path.dirname(__file__), "golden")).with_(arith=mm.ARITH_STRICT, solver=solver)
    theta = dr.jitter_draws(pb, 1, 4096)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"]) and np.all(ref["status"] == 0)
    assert np.array_equal(got["n_accept"], ref["n_accept"])
    assert np.array_equal(got["n_reject"], ref["n_reject"])
    worst = 0.0
    for lo in range(0, 4096, 512):  # in slices: the difference array of the whole batch is another 0.6 GB
        worst = max(worst, rel_state_err(got["traj"][lo:lo + 512], ref["traj"][lo:lo + 512], pb).max())
    assert worst < 1e-9, worst
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-8)


@pytest.fixture(scope="module")
def c5_problem(mm):
    """BASELINE configs[4]: 16 age groups, t = -20 .. 980 (T = 1001), synthetic Poisson observations drawn from the
    base-theta trajectory (workloads.build("c5"), what bench.py --workload c5 runs)."""
    return mm.workloads.build("c5", os.path.join(os.path.dirname(__file__), "golden"),
                              hip_factory=lambda q: mm.HipObjective(q))


@pytest.mark.parametrize("solver", [0, 1])
def test_config5_workload_strict_matches_oracle(mm, oracle_py, c5_problem, solver):
    """configs[4] at its own size in time and age (T = 1001, n = 16, 158 parameters), 70 chains (a ragged last wave
    of the 4-chains-per-wave layout), strict arithmetic: identical step counts, states 1e-9, likelihood 1e-9."""
    from mmid_amd import draws as dr
    pb = c5_problem.with_(arith=mm.ARITH_STRICT, solver=solver)
    assert pb.n == 16 and pb.n_times == 1001
    theta = dr.jitter_draws(pb, 1, 70)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"]) and np.all(ref["status"] == 0)
    assert np.array_equal(got["n_accept"], ref["n_accept"]) and np.array_equal(got["n_reject"], ref["n_reject"])
    assert np.all(got["n_accept"] >= 1000)
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < 1e-9
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-9)


def test_config5_workload_production_arithmetic(mm, oracle_py, c5_problem):
    """The same workload in the fma arithmetic bench.py --workload c5 times: states within the north-star 1e-6
    (measured ~1e-10), likelihood 1e-7, step counts identical for at least 90 % of the chains."""
    from mmid_amd import draws as dr
    pb = c5_problem.with_(arith=mm.ARITH_FMA, solver=0)
    theta = dr.jitter_draws(pb, 1, 70)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    assert np.array_equal(got["status"], ref["status"])
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < REL_STATE_BAR
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-7)
    same = (got["n_accept"] == ref["n_accept"]) & (got["n_reject"] == ref["n_reject"])
    assert same.mean() >= 0.9


def test_forced_small_batch_form_sizes_its_own_workspace(mm, synth400):
    """sepaihrd_set_integrator_form(QUAD) sends a strict batch of more than 16 384 chains to the 16-lane integrator, which
    parks its increments in the workspace; the C ABI must size it from the SAME decision the launch code takes (no prior
    sepaihrd_reserve), and the results are the default path's bits."""
    pb = synth400.with_(arith=mm.ARITH_STRICT)
    pb.times = pb.times[:60]
    pb = pb.with_(obs_H=pb.obs_H[:40], obs_ICU=pb.obs_ICU[:40], obs_D=pb.obs_D[:40])
    from mmid_amd import draws
    theta = np.tile(draws.jitter_draws(pb, 1, 1024), (17, 1))[:16384 + 777]
    want = mm.HipObjective(pb).eval_batch(theta)
    assert np.all(want["status"] == 0)
    hip = mm.HipObjective(pb)
    hip.set_integrator_form(mm.hipabi.FORM_QUAD)
    got = hip.eval_batch(theta)
    for k in ("loglik", "status", "n_accept", "n_reject"):
        assert np.array_equal(got[k], want[k]), k


# ---------------------------------------------------------------------------------------------------------------
# fp32-state arm of BASELINE configs[4] (csrc/sepaihrd_kernels_f32.hip).  No bit contract: the reference has no fp32
# path.  Its test is the distance from the fp64 kernel at the same tolerance, with the bars set from what a 24-bit
# coefficient can carry: a growth rate off by ~5e-8 per day is ~1e-4 relative after a thousand days.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("solver", [0, 1])
def test_fp32_arm_on_config5_workload(mm, c5_problem, solver):
    from mmid_amd import draws as dr
    pb = c5_problem.with_(abs_err=1e-4, rel_err=1e-4, solver=solver, arith=mm.ARITH_FMA)
    theta = dr.jitter_draws(pb, 1, 70)
    f64 = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    f32 = mm.HipObjective(pb.with_(precision=mm.PRECISION_F32)).eval_batch(theta, want_traj=True)
    assert np.array_equal(f32["status"], f64["status"]) and np.all(f64["status"] == 0)
    # same controller on nearly the same error estimates: the step counts differ by a handful at most
    assert np.all(np.abs((f32["n_accept"] + f32["n_reject"]) - (f64["n_accept"] + f64["n_reject"])) <= 0.01 * f64["n_accept"])
    err = rel_state_err(f32["traj"], f64["traj"], pb)
    assert err.max() < 5e-4, err.max()                      # measured 8e-5
    assert np.median(err.max(axis=(1, 2))) < 2e-4
    # likelihood: fp64 terms and sums over fp32 increments; measured max |diff| 17 on values of ~1.5e4
    assert np.max(np.abs(f32["loglik"] - f64["loglik"]) / np.abs(f64["loglik"])) < 5e-3
    np.testing.assert_allclose(f32["ll_parts"].sum(axis=1), f32["loglik"], rtol=1e-12)
    # the cumulative compartments come out of the fp64 totals: monotone, and their daily increments are what the
    # likelihood saw (no 24-bit staircase on a compartment of 1e5 people)
    n = pb.n
    cum = f32["traj"][:, :, 9 * n:10 * n]
    assert np.all(np.diff(cum, axis=1) >= 0)
    inc32, inc64 = np.diff(cum, axis=1), np.diff(f64["traj"][:, :, 9 * n:10 * n], axis=1)
    big = inc64 > 1.0
    assert np.max(np.abs(inc32[big] - inc64[big]) / inc64[big]) < 1e-3
    # a likelihood-only launch gives the same numbers as the trajectory launch
    again = mm.HipObjective(pb.with_(precision=mm.PRECISION_F32)).eval_batch(theta)
    assert np.array_equal(again["loglik"], f32["loglik"])


@pytest.mark.parametrize("n_age", [3, 4, 8])
def test_fp32_arm_other_lane_layouts(mm, shipped, n_age):
    """4 lanes per chain (n = 3 pads one, n = 4) and 8 lanes per chain (two chains per DPP row) on the shipped
    326-point problem at its own tolerance 1e-6, ragged batch."""
    from mmid_amd import draws as dr
    pb = mm.restrict_age_classes(shipped, [0, 1, 2]) if n_age == 3 else (shipped if n_age == 4 else mm.widen_age_classes(shipped, 2))
    pb = pb.with_(arith=mm.ARITH_FMA)
    theta = dr.jitter_draws(pb, 3, 37)
    f64 = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    f32 = mm.HipObjective(pb.with_(precision=mm.PRECISION_F32)).eval_batch(theta, want_traj=True)
    assert np.array_equal(f32["status"], f64["status"]) and np.all(f64["status"] == 0)
    assert rel_state_err(f32["traj"], f64["traj"], pb).max() < 5e-4
    assert np.max(np.abs(f32["loglik"] - f64["loglik"]) / np.abs(f64["loglik"])) < 2e-3


def test_fp32_arm_limits_are_errors_not_fallbacks(mm, shipped):
    two = mm.restrict_age_classes(shipped, [0, 1])
    with pytest.raises(RuntimeError, match="fp32"):
        mm.HipObjective(two.with_(precision=mm.PRECISION_F32))
    hip = mm.HipObjective(shipped.with_(precision=mm.PRECISION_F32))
    hip.set_initial_state_mode(1)
    with pytest.raises(RuntimeError, match="F64"):
        hip.ensemble_quantiles(np.asarray(shipped.base_theta)[None, :], [0.5])
    hip.set_precision(mm.PRECISION_F64)                       # the same context switches back
    hip.set_initial_state_mode(0)
    ref = mm.HipObjective(shipped).eval_batch(np.asarray(shipped.base_theta)[None, :])
    assert np.array_equal(hip.eval_batch(np.asarray(shipped.base_theta)[None, :])["loglik"], ref["loglik"])


# ---------------------------------------------------------------------------------------------------------------
# The cases of the reference's own test file for this path (tests/model/SEPAIHRDObjectivefunctionTest.cpp), on the
# reference's fixture (n = 4, kappa schedule, multiplier branch, 30 daily points), each against the oracle
# ---------------------------------------------------------------------------------------------------------------
def _poisson_obs(oracle_py, pb, seed):
    """Observations the way the reference's fixture makes them (:217-242): Poisson draws around the base trajectory."""
    base = oracle_py.Oracle(pb).eval_batch(pb.base_theta[None, :], want_traj=True)["traj"][0]
    n = len(pb.N)
    rs = np.random.RandomState(seed)
    inc = lambda c: np.maximum(np.diff(base[:, c * n:(c + 1) * n], axis=0, prepend=base[:1, c * n:(c + 1) * n]), 0.0)
    return (rs.poisson(inc(9)).astype(float), rs.poisson(inc(10)).astype(float), rs.poisson(inc(8)).astype(float))


def test_reference_suite_zero_data_and_repeatability(mm, oracle_py, ref_fixture):
    """ZeroDataTest (:384): all observations 0 -> a finite value, the oracle's; ParallelConsistencyTest (:492): five
    evaluations of one vector give one value; SensitivityTest (:368): another beta gives another value."""
    pb = ref_fixture.with_(arith=mm.ARITH_STRICT)
    zero = pb.with_(obs_H=np.zeros_like(pb.obs_H), obs_ICU=np.zeros_like(pb.obs_ICU), obs_D=np.zeros_like(pb.obs_D))
    got = mm.HipObjective(zero).eval_batch(zero.base_theta[None, :])
    ref = oracle_py.Oracle(zero).eval_batch(zero.base_theta[None, :])
    assert got["status"][0] == 0 and np.isfinite(got["loglik"][0]) and got["loglik"][0] < 0
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)
    hip = mm.HipObjective(pb)
    five = [hip.calculate(pb.base_theta) for _ in range(5)]
    assert len(set(five)) == 1
    th = pb.base_theta.copy()
    th[pb.param_names.index("beta")] *= 1.1
    assert hip.calculate(th) != five[0]


@pytest.mark.parametrize("scale", [1e-3, 1.0, 40.0])
def test_reference_suite_extreme_beta(mm, oracle_py, ref_fixture, scale):
    """ExtremeParameterTest (:510): a transmission rate far below / far above the fitted one still gives the oracle's
    value (or both sides' failure sentinel)."""
    pb = ref_fixture.with_(arith=mm.ARITH_STRICT, bounds={})
    th = pb.base_theta.copy()
    th[pb.param_names.index("beta")] *= scale
    got = mm.HipObjective(pb).eval_batch(th[None, :])
    ref = oracle_py.Oracle(pb).eval_batch(th[None, :])
    assert np.array_equal(got["status"], ref["status"])
    if ref["status"][0] == 0:
        np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)
        assert np.array_equal(got["n_accept"], ref["n_accept"]) and np.array_equal(got["n_reject"], ref["n_reject"])
    else:
        assert got["loglik"][0] == mm.LOWEST


@pytest.mark.parametrize("step,days", [(0.1, 30.0), (1.0, 365.0)])
def test_reference_suite_dense_and_long_grids(mm, oracle_py, ref_fixture, step, days):
    """DenseTimeGridTest (:414, 0.1-day outputs) and LongSimulationTest (:566, a 365-day grid): more outputs than RK steps
    in the first, the whole kappa schedule in the second; values and step counts are the oracle's."""
    times = np.round(np.arange(0.0, days + 0.5 * step, step), 10)
    T = len(times)
    n = len(ref_fixture.N)
    pb = ref_fixture.with_(arith=mm.ARITH_STRICT, times=times, obs_H=np.zeros((T, n)), obs_ICU=np.zeros((T, n)), obs_D=np.zeros((T, n)))
    oH, oI, oD = _poisson_obs(oracle_py, pb, 42)
    pb = pb.with_(obs_H=oH, obs_ICU=oI, obs_D=oD)
    rs = np.random.RandomState(5)
    theta = pb.base_theta[None, :] * (1 + 0.05 * rs.standard_normal((6, pb.n_params)))
    got = mm.HipObjective(pb).eval_batch(theta, want_traj=True)
    ref = oracle_py.Oracle(pb).eval_batch(theta, want_traj=True)
    assert np.all(got["status"] == 0) and np.array_equal(got["status"], ref["status"])
    assert np.array_equal(got["n_accept"], ref["n_accept"]) and np.array_equal(got["n_reject"], ref["n_reject"])
    np.testing.assert_allclose(got["loglik"], ref["loglik"], rtol=1e-10)
    assert rel_state_err(got["traj"], ref["traj"], pb).max() < 1e-9


def test_reference_suite_streams_enter_separately(mm, oracle_py, ref_fixture):
    """IndividualLikelihoodComponentsTest / the per-stream tests (:528, :604): changing one stream's observations moves
    exactly that stream's part of the log-likelihood."""
    pb = ref_fixture.with_(arith=mm.ARITH_STRICT)
    base = mm.HipObjective(pb).eval_batch(pb.base_theta[None, :])["ll_parts"][0]
    for k, name in enumerate(("obs_H", "obs_ICU", "obs_D")):
        other = pb.with_(**{name: getattr(pb, name) + 1.0})
        parts = mm.HipObjective(other).eval_batch(other.base_theta[None, :])["ll_parts"][0]
        ref = oracle_py.Oracle(other).eval_batch(other.base_theta[None, :])["ll_parts"][0]
        np.testing.assert_allclose(parts, ref, rtol=1e-10)
        for j in range(3):
            assert (parts[j] != base[j]) == (j == k)


def test_poisson_term_log_is_the_hosts_log(mm, shipped):
    """The log of the Poisson term (the reference calls std::log, SEPAIHRDObjectiveFunction.cpp:264-276) is the table path of
    glibc's log on the constants extracted from this image's libm (csrc/sepaihrd_dev_common.inc log_pos): outside glibc's
    near-one window it has std::log's bits, inside it stays within 1e-17 absolute of the true log (2e-17 of the host's
    rounded one) -- a term obs log(sim) of a sum near 1e5
    needs no relative accuracy of a result near zero."""
    import math
    rng = np.random.default_rng(5)
    lo_edge, hi_edge = 1.0 - 2.0 ** -4, 1.0 + float.fromhex("0x1.09p-4")
    x = np.concatenate([
        10.0 ** rng.uniform(-10.0, 7.0, 60000),                 # sim + 1e-10 of every size the likelihood sees
        rng.uniform(lo_edge, hi_edge, 20000),                   # the window glibc treats with its log1p polynomial
        np.nextafter(lo_edge, 0.0) * np.ones(1), np.array([lo_edge, hi_edge, np.nextafter(hi_edge, 0.0)]),
        np.array([1e-10, 1.0 + 1e-10, 1.0, 0.5, 2.0, 1.5, np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0), 1e300, 2.3e-308]),
        2.0 ** rng.integers(-30, 30, 200).astype(float),        # exact powers of two: z = 1, r at a table boundary
    ])
    got = mm.HipObjective(shipped).device_log_values(x)
    ref = np.array([math.log(v) for v in x])
    inside = (x >= lo_edge) & (x < hi_edge)
    assert inside.sum() > 20000 and (~inside).sum() > 60000
    assert np.array_equal(got[~inside].view(np.uint64), ref[~inside].view(np.uint64))
    assert np.abs(got[inside] - ref[inside]).max() < 2e-17
