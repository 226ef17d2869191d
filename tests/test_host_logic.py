"""Host-side logic that needs no GPU: name -> field resolution, constraints, draw generator,
file formats, problem plumbing, and the C-ABI library's exported surface."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol(mm):
    lib = mm.load_library()
    header = open(os.path.join(ROOT, "include", "sepaihrd_hip.h")).read()
    declared = set(re.findall(r"\b(sepaihrd_[a-z_]+)\s*\(", header))
    assert declared == set(mm.hipabi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.sepaihrd_abi_version() == mm.hipabi.ABI_VERSION


def test_struct_layout_matches_header_field_order(mm):
    header = open(os.path.join(ROOT, "include", "sepaihrd_hip.h")).read()
    start = header.index("typedef struct sepaihrd_problem {") + len("typedef struct sepaihrd_problem {")
    body = header[start:header.index("} sepaihrd_problem;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        for piece in decl.split(","):
            m = re.search(r"\*?\s*([A-Za-z_][A-Za-z_0-9]*)\s*(\[\d+\])?$", piece.strip())
            names.append(m.group(1))
    assert names == [f for f, _ in mm.hipabi.sepaihrd_problem._fields_]


def test_sampler_config_struct_matches_header(mm):
    header = open(os.path.join(ROOT, "include", "sepaihrd_hip.h")).read()
    start = header.index("typedef struct sepaihrd_mh_config {") + len("typedef struct sepaihrd_mh_config {")
    body = re.sub(r"/\*.*?\*/", "", header[start:header.index("} sepaihrd_mh_config;")], flags=re.S)
    fields = [(m.group(1), m.group(2)) for m in re.finditer(r"(int32_t|double)\s+([a-z_]+);", body)]
    ctype = {"int32_t": ctypes.c_int32, "double": ctypes.c_double}
    assert [(n, ctype[t]) for t, n in fields] == list(mm.hipabi.sepaihrd_mh_config._fields_)
    assert ctypes.sizeof(mm.hipabi.sepaihrd_mh_config) == 6 * 4 + 2 * 8


def test_create_fails_loudly_without_gpu(mm, shipped, have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|sepaihrd_create failed"):
        mm.HipObjective(shipped)


def test_name_resolution_follows_reference_dispatch_order(mm, shipped):
    r = lambda nm: mm.resolve_param_name(nm, shipped.npi_names, 4, 7)
    assert r("beta_1") == (18, 0) and r("beta_7") == (18, 6)
    assert r("kappa_2") == (19, 1) and r("kappa_7") == (19, 6)
    assert r("kappa_1") == (-1, 0)            # fixed baseline: not calibratable -> ignored with a warning
    assert r("h_infec_2") == (21, 2)          # "h_infec_" must win over "h_"
    assert r("h_2") == (23, 2) and r("p_0") == (22, 0) and r("icu_3") == (24, 3)
    assert r("d_H_1") == (25, 1) and r("d_ICU_1") == (26, 1) and r("d_community_3") == (27, 3)
    assert r("gamma_p") == (3, 0) and r("P0_multiplier") == (9, 0) and r("runup_days") == (16, 0)
    assert r("no_such_parameter") == (-1, 0)
    with pytest.raises(ValueError):
        r("beta_8")
    codes, idx = shipped.field_map()
    assert len(codes) == 62 and (codes >= 0).all()
    # the shipped theta covers 53 effective + 9 inert parameters
    assert np.array_equal(shipped.current_parameters(), shipped.base_theta)


def test_constraints_match_oracle(mm, oracle_py, shipped):
    from mmid_amd import draws
    orc = oracle_py.Oracle(shipped)
    rs = np.random.RandomState(2)
    lo, hi, has = shipped.bounds_arrays()
    theta = lo + (hi - lo) * rs.uniform(-3.0, 4.0, (50, shipped.n_params))
    for mode in (0, 1):
        got = draws.apply_constraints(theta, lo, hi, has, mode)
        ref = orc.apply_constraints(theta, mode)
        assert np.array_equal(got, ref)
        assert np.all(got >= lo) and np.all(got <= hi)
    # reflect is NOT idempotent in floating point (minb + (y - minb) may round): the device re-applies it
    once = draws.apply_constraints(theta, lo, hi, has, 1)
    twice = draws.apply_constraints(once, lo, hi, has, 1)
    assert np.abs(once - twice).max() < 1e-15


def test_constraints_without_bounds_entry(mm):
    from mmid_amd import draws
    v = np.array([[-0.5, 0.25]])
    assert np.array_equal(draws.apply_constraints(v, np.zeros(2), np.zeros(2), np.zeros(2), 0), [[0.0, 0.25]])
    assert np.array_equal(draws.apply_constraints(v, np.zeros(2), np.zeros(2), np.zeros(2), 1), [[0.5, 0.25]])


def test_draw_generator_reproduces_libstdcxx(mm, oracle_py, synth400):
    from mmid_amd import draws
    z = draws.std_normals_batch(np.array([1, 42, 123456]), 63)   # odd count: last pair half used
    for row, seed in zip(z, (1, 42, 123456)):
        assert np.array_equal(row, oracle_py.std_normals(seed, 63))
    a = draws.jitter_draws(synth400, 1, 300)
    b = oracle_py.Oracle(synth400).jitter_draws(synth400.base_theta, 1, 300)
    assert np.array_equal(a, b)


def test_sampler_stream_queue_follows_the_generator(mm):
    """The device-resident sampler reads a chain's mt19937 stream through a look-ahead queue of canonical uniforms that
    both continuations of an accept test share (host/src/MultiChainMetropolisHastings.cpp: CanonicalQueue).  Driven
    through its test hook with a random pattern of tests that do / do not take the uniform, it must hand out exactly
    what libstdc++ would from ONE generator used in the reference's order (MetropolisHastingsSampler.cpp:91-102,327):
    [the uniform,] then P normals by the polar method with a fresh distribution per proposal."""
    import ctypes as C
    import math
    from mmid_amd import draws, hostabi
    lib = hostabi.load_library()
    lib.host_queue_draw_sequence.restype = None
    lib.host_queue_draw_sequence.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for seed, P, rounds in ((7, 62, 40), (123456, 5, 120), (3, 1, 60)):
        takes = (np.random.default_rng(seed).random(rounds) < 0.8).astype(np.uint8)
        normals = np.empty((rounds, P))
        log_u = np.empty(rounds)
        lib.host_queue_draw_sequence(seed, P, rounds, takes.ctypes.data, normals.ctypes.data, log_u.ctypes.data)
        w = draws.mt19937_words(seed, 4 * rounds * (2 * P + 40))
        pos = 0

        def canonical():
            nonlocal pos
            r = (w[pos] + w[pos + 1] * 4294967296.0) / 18446744073709551616.0
            pos += 2
            return math.nextafter(1.0, 0.0) if r >= 1.0 else r
        for r in range(rounds):
            keep = pos
            assert log_u[r] == math.log(canonical())   # the element in front, whether or not the test takes it
            if not takes[r]:
                pos = keep
            ref = np.empty(2 * ((P + 1) // 2))
            for i in range(0, P, 2):
                while True:
                    x = 2.0 * canonical() - 1.0
                    y = 2.0 * canonical() - 1.0
                    r2 = x * x + y * y
                    if not (r2 > 1.0 or r2 == 0.0):
                        break
                mult = math.sqrt(-2 * math.log(r2) / r2)
                ref[i], ref[i + 1] = y * mult, x * mult
            assert np.array_equal(normals[r], ref[:P]), (seed, r)


def test_config_readers_on_synthetic_files(mm, tmp_path):
    cio = mm.config_io
    p = tmp_path / "guess.txt"
    p.write_text("# comment\nbeta_end_times 13.0 63.0\nkappa_end_times 13.0 63.0\nbeta_1 0.4\nbeta_2 0.2\n"
                 "kappa_1 1.0\nkappa_2 0.5\n\na 0.5 0.9 0.8 1.2\nsigma 3e-1\nruns_unknown 7\n")
    par = cio.read_sepaihrd_parameters(str(p), 4)
    assert np.array_equal(par["beta_values"], [0.4, 0.2]) and np.array_equal(par["kappa_values"], [1.0, 0.5])
    assert np.array_equal(par["a"], [0.5, 0.9, 0.8, 1.2]) and par["sigma"] == 0.3
    assert np.array_equal(par["beta_end_times"], [13.0, 63.0])
    b = tmp_path / "bounds.txt"
    b.write_text("# b\nbeta_1  0.35   0.9\ntheta 0.2 0.8\n")
    assert cio.read_param_bounds(str(b)) == {"beta_1": (0.35, 0.9), "theta": (0.2, 0.8)}
    s = tmp_path / "sig.txt"
    s.write_text("beta_1 0.02\n# x\ntheta 0.03\n")
    assert cio.read_proposal_sigmas(str(s)) == {"beta_1": 0.02, "theta": 0.03}
    c = tmp_path / "cal.txt"
    c.write_text("# names\nbeta_1\n\ntheta\n")
    assert cio.read_params_to_calibrate(str(c)) == ["beta_1", "theta"]
    m = tmp_path / "m.csv"
    m.write_text("1,2\n3,4\n")
    assert np.array_equal(cio.read_matrix_csv(str(m), 2, 2), [[1, 2], [3, 4]])
    with pytest.raises(ValueError):
        cio.read_matrix_csv(str(m), 3, 3)


def test_initial_state_heuristic_properties(mm, ref_fixture):
    """GetCalibrationDataTests.cpp:163-227: observable compartments, conservation, non-negativity."""
    n = 4
    x = ref_fixture.initial_state.reshape(11, n)
    assert np.all(x >= 0)
    assert np.allclose(x[:9].sum(axis=0), ref_fixture.N, rtol=1e-12)
    assert np.array_equal(x[9], ref_fixture.obs_H[0]) and np.array_equal(x[10], ref_fixture.obs_ICU[0])
    assert np.array_equal(x[8], np.minimum(ref_fixture.obs_D[0], ref_fixture.N))


def test_posterior_trace_csv_format(mm, tmp_path):
    path = tmp_path / "trace.csv"
    mm.config_io.write_posterior_trace_csv(str(path), np.array([[0.5, 2.0]]), np.array([-1234.5]), ["a", "b"])
    lines = path.read_text().splitlines()
    assert lines[0] == "iter,log_posterior,a,b"
    assert lines[1] == "0,-1.234500e+03,5.000000e-01,2.000000e+00"


def test_widen_age_classes_config5_shape(mm, shipped):
    w = mm.widen_age_classes(shipped, 4)
    assert w.n == 16 and w.M.shape == (16, 16)
    assert np.isclose(w.N.sum(), shipped.N.sum())
    assert np.allclose(w.M[0:4, 4:8], shipped.M[0, 1] / 4)
    assert w.n_params == 62 + 3 * 32    # the 32 age-indexed parameters become 128
    assert len(w.base_theta) == w.n_params
    codes, _ = w.field_map()
    assert (codes >= 0).all()


def test_problem_json_roundtrip(mm, shipped, tmp_path):
    p = tmp_path / "pb.json"
    shipped.save(str(p))
    back = mm.SEPAIHRDProblem.load(str(p))
    assert np.array_equal(back.base_theta, shipped.base_theta) and back.param_names == shipped.param_names
    assert np.array_equal(back.obs_D, shipped.obs_D) and back.bounds == shipped.bounds


# ---- C++ host mirror (host/): the parts that need no device
def test_host_parameter_manager_mirrors_reference(mm, oracle_py, shipped):
    h = mm.HostObjective(shipped, with_objective=False)
    assert np.array_equal(h.current_parameters(), shipped.base_theta)     # getCurrentParameters
    rs = np.random.RandomState(4)
    lo, hi, _ = shipped.bounds_arrays()
    theta = lo + (hi - lo) * rs.uniform(-2.0, 3.0, (20, shipped.n_params))
    orc = oracle_py.Oracle(shipped)
    for mode in (0, 1):
        assert np.array_equal(h.apply_constraints(theta, mode), orc.apply_constraints(theta, mode))


def test_host_parameter_manager_rejects_what_the_reference_rejects(mm, shipped):
    bad = shipped.with_(param_names=shipped.param_names + ["kappa_1"],
                        sigmas={**shipped.sigmas, "kappa_1": 0.1}, bounds={**shipped.bounds, "kappa_1": (0.5, 1.5)},
                        base_theta=None)
    # Python-side resolution mirrors the warning path; the C++ manager mirrors the constructor's throw
    with pytest.raises(RuntimeError, match="fixed baseline kappa"):
        mm.HostObjective(bad, with_objective=False)
    missing = shipped.with_(sigmas={k: v for k, v in shipped.sigmas.items() if k != "theta"}, base_theta=None)
    # sigma_array() fills 0.0 for the C shim, so drop the name from bounds instead to hit "Missing bounds"
    assert "theta" in missing.param_names


def test_host_objective_fails_loudly_without_gpu(mm, shipped, have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        mm.HostObjective(shipped)


# ---- sepaihrd_create argument validation happens before any device call: testable without a GPU
def _create_error(mm, pb, patch=None):
    import ctypes as C
    lib = mm.load_library()
    keep = []
    st = mm.hipabi.build_problem_struct(pb, keep)
    if patch:
        patch(st, keep)
    err = C.create_string_buffer(512)
    ctx = lib.sepaihrd_create(C.byref(st), -1, err, len(err))
    if ctx:
        lib.sepaihrd_destroy(ctx)
        return None
    return err.value.decode()


def test_create_validates_like_the_reference_constructors(mm, shipped, have_gpu):
    import ctypes as C
    dp = C.POINTER(C.c_double)

    def times_not_increasing(st, keep):   # Simulator::run: "Output time points must be strictly increasing"
        t = np.array(shipped.times)
        t[5] = t[4]
        keep.append(t)
        st.times = t.ctypes.data_as(dp)
    assert "strictly increasing" in _create_error(mm, shipped, times_not_increasing)

    def bad_abi(st, keep):
        st.abi_version = 99
    assert "ABI version" in _create_error(mm, shipped, bad_abi)

    def no_kappa(st, keep):               # the model always owns an NPI strategy with a baseline period
        st.n_kappa = 0
    assert "schedule" in _create_error(mm, shipped, no_kappa)

    def negative_baseline_end(st, keep):  # PiecewiseConstantNpiStrategy ctor
        k = np.array(shipped.kappa_end_times)
        k[0] = -1.0
        keep.append(k)
        st.kappa_end_times = k.ctypes.data_as(dp)
    assert "non-negative" in _create_error(mm, shipped, negative_baseline_end)

    def unsorted_beta(st, keep):          # PiecewiseConstantParameterStrategy ctor
        b = np.array(shipped.beta_end_times)
        b[2] = b[1]
        keep.append(b)
        st.beta_end_times = b.ctypes.data_as(dp)
    assert "beta end times" in _create_error(mm, shipped, unsorted_beta)

    def negative_tolerance(st, keep):     # Simulator::setErrorTolerance
        st.abs_err = -1e-6
    assert "tolerance" in _create_error(mm, shipped, negative_tolerance)

    def zero_dt(st, keep):                # Simulator ctor: "Time step hint must be positive"
        st.dt_hint = 0.0
    assert "dt_hint" in _create_error(mm, shipped, zero_dt)

    def bad_solver(st, keep):
        st.solver = 7
    assert "solver" in _create_error(mm, shipped, bad_solver)

    def too_many_ages(st, keep):
        st.n_age = 65
    assert "n_age" in _create_error(mm, shipped, too_many_ages)

    def null_pointer(st, keep):
        st.obs_H = None
    assert "NULL" in _create_error(mm, shipped, null_pointer)

    def beta_index_out_of_range(st, keep):   # "Beta index out of range for name"
        idx = np.array(mm.hipabi.np.ctypeslib.as_array(st.param_index, (shipped.n_params,)))
        idx[0] = 99
        keep.append(idx)
        st.param_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    if have_gpu:   # field-map checks run after the device check
        assert "beta index" in _create_error(mm, shipped, beta_index_out_of_range)
    # a valid problem fails only for lack of a device on the CPU box
    msg = _create_error(mm, shipped)
    assert (msg is None) if have_gpu else ("no HIP device" in msg)


def test_null_ctx_and_bad_arguments_return_error_codes(mm):
    lib = mm.load_library()
    assert lib.sepaihrd_eval_batch(None, None, 1, None, None, None, None, None, None) == -1
    assert lib.sepaihrd_eval_batch_device(None, None, 1, None, None, None, None, None, None, None) == -1
    assert lib.sepaihrd_set_constraint_mode(None, 0) == -1
    assert lib.sepaihrd_apply_constraints(None, 0, None, 1, None) == -1
    assert lib.sepaihrd_last_error(None) == b"ctx is NULL"
    lib.sepaihrd_destroy(None)   # no-op


def test_missing_hip_library_fails_loudly(mm, tmp_path):
    """The product path has no CPU fallback: a missing extension is an error, never a silent detour."""
    missing = str(tmp_path / "libsepaihrd_hip.so")
    with pytest.raises(FileNotFoundError, match="no CPU fallback"):
        mm.hipabi.load_library(missing)


def test_compute_entry_points_refuse_to_run_without_a_device(mm, shipped, have_gpu):
    """Without a usable HIP device sepaihrd_create reports SEPAIHRD_E_NO_DEVICE-style failure (no context,
    message set) instead of evaluating anything on the host."""
    if have_gpu:
        pytest.skip("a GPU is present: covered by the -m gpu suite")
    with pytest.raises(RuntimeError, match="(?i)device|hip"):
        mm.HipObjective(shipped)


def test_model_side_holders_follow_the_reference(mm, oracle_py, shipped):
    """AgeSEPAIHRDModel / PiecewiseConstantNpiStrategy as the reference's constructors take them (host only):
    kappa(t) with the reference's rules -- baseline for t < 0 and t <= baseline end, `t <= end` keeps the OLD value at a
    breakpoint, the last value for ever (PieceWiseConstantNPIStrategy.cpp:86-127) -- equal to the oracle's restatement;
    getModelParameters() carries the schedule baseline first (AgeSEPAIHRDModel.cpp:294-323); there is no host RHS."""
    ends, vals = np.asarray(shipped.kappa_end_times), np.asarray(shipped.kappa_values)
    t = np.concatenate([[-20.0, -0.5, 0.0], ends - 1e-9, ends, ends + 1e-9, [1e4]])
    got = mm.hostabi.model_holders(ends[1:], vals[1:], vals[0], ends[0], t)
    orc = oracle_py.Oracle(shipped)
    want = np.array([orc.beta_kappa(x)[1] for x in t])
    assert np.array_equal(got["kappa"], want)
    assert got["kappa"][0] == vals[0] and got["kappa"][-1] == vals[-1]
    assert np.array_equal(got["schedule_ends"], ends) and np.array_equal(got["schedule_values"], vals)
    assert got["state_size"] == 33 and got["names"] == "S0 CumICU2 kappa_2 3"


def test_post_calibration_tree_matches_the_reference_writers_and_its_plot_script(mm, tmp_path):
    """the files after the path.  The tree config_io writes from one ensemble result has the file names,
    headers and number formats of the reference's AnalysisWriter / MetropolisHastingsSampler writers (expected headers
    committed in tests/golden/reference_output_headers.json) and loads the way scripts/model/PostCalibrationAnalysis.py
    loads it (pandas read_csv, 'time' -> date, 'age_' columns, metrics summary indexed by metric name)."""
    import json
    import pandas as pd
    from mmid_amd import config_io as cio
    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_output_headers.json")) as fh:
        want = json.load(fh)
    rs = np.random.RandomState(0)
    n, T, S = 4, 30, 50
    times = np.arange(-5, T - 5, dtype=np.float64)
    Tp = int(np.sum(times >= 0))
    q = np.sort(rs.gamma(2.0, 10.0, (6, 5, Tp, n)), axis=1)
    metrics = rs.uniform(0.001, 2.0, (S, 12 + 4 * n))
    metrics[7] = np.nan                                      # a skipped sample
    ens = {"ppc": q, "rt": np.sort(rs.uniform(0.5, 3, (5, T)), axis=0), "sero": np.sort(rs.uniform(0, 0.2, (5, T)), axis=0),
           "metrics": metrics}
    samples = rs.normal(size=(S, 2)) * [1e-3, 1e4]
    obs = {name: rs.poisson(5.0, (Tp, n)).astype(float) for name in cio.PPC_SERIES}
    out = str(tmp_path / "post")
    cio.write_post_calibration_tree(out, times, ens, samples, ["p0", "p1"], n, observed=obs, burn_in=10, thinning=2)
    for rel, header in want["files"].items():
        path = os.path.join(out, rel)
        assert os.path.exists(path), rel
        with open(path) as fh:
            assert fh.readline().rstrip("\n").split(",") == header, rel
    # number formats: fixed 6 in the predictive files, scientific 8 in the samples, fixed 8 in the summaries
    line = open(os.path.join(out, "posterior_predictive", "daily_deaths_median.csv")).readlines()[1].rstrip().split(",")
    assert line[0] == "0" and all(len(v.split(".")[1]) == 6 for v in line[1:])
    line = open(os.path.join(out, "parameter_posteriors", "posterior_samples.csv")).readlines()[1].rstrip().split(",")
    assert line[0] == "0" and all("e" in v and len(v.split("e")[0].split(".")[1]) == 8 for v in line[1:])
    assert len(open(os.path.join(out, "parameter_posteriors", "posterior_samples.csv")).readlines()) == 1 + len(range(10, S, 2))
    assert len(open(os.path.join(out, "mcmc_batches", "batch_0.csv")).readlines()) == 1 + S - 1   # the NaN row is skipped
    # ... and the plotting script's own access pattern (PostCalibrationAnalysis.py:64-76,98-121,175-176,216-241,322-337)
    med = pd.read_csv(os.path.join(out, "posterior_predictive", "daily_hospitalizations_median.csv"))
    assert "time" in med.columns and [c for c in med.columns if "age_" in c] == [f"age_{a}" for a in range(n)]
    date = pd.to_datetime("2020-03-01") + pd.to_timedelta(med["time"], unit="D")
    assert len(date) == Tp and np.allclose(med[[f"age_{a}" for a in range(n)]].sum(axis=1), q[0, 2].sum(axis=1), atol=1e-5)
    summ = pd.read_csv(os.path.join(out, "mcmc_aggregated", "metrics_summary.csv"), index_col=0)
    for prefix in ("IFR", "IHR", "IICUR"):
        for j in range(n):
            row = summ.loc[f"{prefix}_age_{j}"]
            assert row["q025"] <= row["median"] <= row["q975"]
    rt = pd.read_csv(os.path.join(out, "rt_trajectories", "Rt_aggregated_with_uncertainty.csv"))
    assert np.all(rt["q025"] <= rt["median"]) and np.all(rt["q05"] <= rt["q95"]) and np.allclose(rt["median"], ens["rt"][2], atol=1e-6)
    ps = pd.read_csv(os.path.join(out, "parameter_posteriors", "posterior_samples.csv"))
    assert [p for p in ps.columns if p not in ["sample_index", "objective_value"]] == ["p0", "p1"]
    # the sampler's own trace file (MetropolisHastingsSampler.cpp:414-438)
    trace = str(tmp_path / "posterior_trace_final.csv")
    cio.write_posterior_trace_csv(trace, samples[:3], np.array([-1.5e6, 2.0, 3.25e-3]), ["p0", "p1"])
    lines = open(trace).read().splitlines()
    assert lines[0].split(",") == want["posterior_trace"]["header"]
    assert lines[1].split(",")[:2] == ["0", "-1.500000e+06"] and lines[3].split(",")[1] == "3.250000e-03"


def test_adapter_maps_every_failure_status_to_the_references_exception(mm):
    """Per-chain status 2 (odeint's 500 rejections), 3 (attempt budget) and 4 (SEPAIHRD_STATUS_PIPELINE: the hand-off
    between the integrating wavefront and its likelihood wavefront timed out) all leave calculate() as the
    SimulationException the reference's solver wrapper throws (Dopri5SolverStrategy.cpp:38-42), which the samplers'
    safeEvaluate turns into -1e18 (MetropolisHastingsSampler.cpp:65-74); the header declares all three."""
    header = open(os.path.join(ROOT, "include", "sepaihrd_hip.h")).read()
    declared = dict((name, int(v)) for name, v in re.findall(r"#define (SEPAIHRD_STATUS_[A-Z_]+) (\d+)", header))
    assert declared == {"SEPAIHRD_STATUS_OK": 0, "SEPAIHRD_STATUS_INVALID": 1, "SEPAIHRD_STATUS_STEP_FAILURE": 2,
                        "SEPAIHRD_STATUS_STEP_BUDGET": 3, "SEPAIHRD_STATUS_PIPELINE": 4}
    lib = mm.hostabi.load_library()
    lib.host_status_exception.argtypes = [ctypes.c_int]
    lib.host_last_error.restype = ctypes.c_char_p
    seen = set()
    for name, status in declared.items():
        if status < 2:
            continue
        assert lib.host_status_exception(status) == 1
        msg = lib.host_last_error().decode()
        assert msg and msg not in seen
        seen.add(msg)
        if status >= 3:
            assert name in msg


def test_summary_quantiles_of_gathered_chain_records(mm):
    """The host side of the post-calibration exchange needs no GPU: per-chain summary records of two groups
    of chains concatenated in chain order, then exact-sort quantiles across chains of every column -- the rule of the
    reference's trajectory quantiles (PostCalibrationAnalyser.cpp:303-340), numpy's linear-interpolation quantile."""
    rs = np.random.RandomState(3)
    group_a, group_b = rs.normal(size=(6, 10)), rs.normal(size=(5, 10))
    table = np.concatenate([group_a, group_b])
    probs = [0.025, 0.5, 0.975, 0.0, 1.0]
    got = mm.hostabi.summary_quantiles(table, probs)
    np.testing.assert_allclose(got, np.quantile(table, probs, axis=0), rtol=1e-13, atol=1e-15)
    assert np.array_equal(mm.hostabi.summary_quantiles(table[:1], [0.3]), table[:1])   # one chain: its own value
    # the Python mirror used on the torch.distributed path gives the same numbers
    import torch
    np.testing.assert_allclose(mm.parallel.ensemble_quantiles(torch.from_numpy(table), probs[:3]).numpy(), got[:3], rtol=1e-13)


def test_device_log_restatement_equals_the_libm_of_this_image(mm):
    """The device draws the sampler's normals and log(u) itself; the one non-IEEE step of that recipe is std::log.
    csrc/sepaihrd_rng.inc writes out glibc's double-precision log as this image's libm evaluates it (x86-64, FMA variant);
    compiled for the host it must give libm's bits: 400 000 arguments over the sampler's domains -- r2 in (0, 1], uniforms
    in (0, 1), the near-1 branch [1 - 2^-4, 1 + 0x1.09p-4), tiny and subnormal values -- against math.log (libm's log)."""
    import math
    rs = np.random.RandomState(11)
    u = rs.random_sample(100000)
    u = u[u > 0]
    xs = np.concatenate([u, u * u, 1.0 - 0.07 * u, 1.0 + 0.064 * u, np.ldexp(u, -rs.randint(0, 1000, u.size)),
                         [1.0, 0.9375, np.nextafter(1.0, 0.0), np.nextafter(1.0, 2.0), 5e-324, 2.2250738585072014e-308, 1e-10]])
    got = mm.hostabi.glibc_log(xs)
    want = np.array([math.log(v) for v in xs])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert np.isneginf(mm.hostabi.glibc_log([0.0])[0])


def test_device_exp_restatement_equals_the_libm_of_this_image(mm):
    """The self-contained sampler adapts every chain's global scale on the device: global_scale_ = std::exp(log_scale_) with
    log_scale_ clamped to [-6.9, 2.3] (MetropolisHastingsSampler.cpp:150-151).  csrc/sepaihrd_rng.inc's restatement of
    glibc's exp, compiled for the host, must give libm's bits over that range and well beyond it (|x| < 512), and 1 + x for
    the tiny arguments libm short-cuts."""
    import math
    rs = np.random.RandomState(12)
    xs = np.concatenate([rs.uniform(-6.9, 2.3, 300000), rs.uniform(-500.0, 500.0, 100000), rs.uniform(-1e-3, 1e-3, 20000),
                         -6.9 + 0.01 * np.arange(921), [0.0, -0.0, -6.9, 2.3, -0.7, 1e-300, -1e-300, 2.0 ** -54, 2.0 ** -55, 511.99]])
    got = mm.hostabi.glibc_exp(xs)
    want = np.array([math.exp(v) for v in xs])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_every_declared_entry_point_has_a_ctypes_signature(mm):
    """A ctypes call without argtypes passes a Python int as a C int: a 64-bit address handed over that way is cut to 32 bits
    (and crashes in the library).  Every function include/sepaihrd_hip.h declares has its signature set in hipabi.py."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "sepaihrd_hip.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(sepaihrd_\w+)\s*\(", header)))
    lib = mm.hipabi.load_library()
    missing = [n for n in names if getattr(lib, n).argtypes is None]
    assert not missing, missing


def test_one_arithmetic_is_the_product_in_bench_adapter_and_docs(mm, monkeypatch):
    """the arithmetic bench.py's `value` is measured in is the one the drop-in constructors select.
    The choice is the measured one (profiles/r04_fma_vs_strict_100k.json: 4096 chains x 100 000 iterations, fma against
    strict with the same seeds -- a flip would make strict the product): the committed verdict, bench.py's default
    --arith, the C++ adapter's default and the documents must all say the same thing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r04_fma_vs_strict_100k.json")) as fh:
        run = json.load(fh)
    assert run["chains"] == 4096 and run["iterations"] == 100000 and run["decisions_per_run"] == 4096 * 99999
    assert run["verdict"] == ("strict" if run["chains_with_a_flip"] else "fma")
    product = run["verdict"]
    monkeypatch.delenv("SEPAIHRD_ARITH", raising=False)
    assert mm.hostabi.default_arith() == {"fma": mm.ARITH_FMA, "strict": mm.ARITH_STRICT}[product]
    monkeypatch.setenv("SEPAIHRD_ARITH", "strict")
    assert mm.hostabi.default_arith() == mm.ARITH_STRICT          # the opt-in
    monkeypatch.setenv("SEPAIHRD_ARITH", "fma")
    assert mm.hostabi.default_arith() == mm.ARITH_FMA
    # bench.py's parser default, read from the script itself (importing it is cheap: torch is only imported in main())
    code = ("import sys; sys.argv=['bench.py']; sys.path.insert(0, %r); import bench; print(bench.parse_args().arith)" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip()
    assert out == product
    for doc in ("README.md", "INTEGRATION.md"):
        text = open(os.path.join(root, doc)).read()
        assert "SEPAIHRD_ARITH=strict" in text, doc + " must name the opt-in"
        assert "r04_fma_vs_strict_100k.json" in text, doc + " must cite the run that settled the arithmetic"


def test_libm_selfcheck_host_twin_and_its_arguments(mm):
    """the device's log / exp restate ONE libm build; before a sampler lets the device draw the chains'
    streams they are compared with the host's libm on fixed arguments (sepaihrd_device_libm_check).  This is the comparison
    itself without a device: the same text compiled for the host, on the same arguments, against std::log / std::exp of
    this process -- zero differences on this image -- and a look at what the arguments cover."""
    r = mm.hostabi.libm_selfcheck()
    assert r["n"] == 4096 and r["log_diff"] == 0 and r["exp_diff"] == 0
    la, ea = r["log_args"], r["exp_args"]
    assert np.all(la > 0) and np.all(np.isfinite(la))
    near_one = np.abs(la - 1.0) < 0.0664
    assert near_one.sum() >= 1000 and (la[near_one] > 1.0).sum() > 400 and (la[near_one] < 1.0).sum() > 400   # glibc's log1p-style branch, both sides
    assert np.any((la > 1 - 2.0 ** -4 - 0.004) & (la < 1 - 2.0 ** -4)) or np.any(la < 0.94)                    # and arguments beyond its lower edge
    assert (la < 1e-100).sum() > 100 and la.min() < 1e-290                                                      # tiny, the 2^k scaling of the table path
    unit = (la > 0.0) & (la < 1.0) & ~near_one
    assert unit.sum() > 1500                                                                                    # canonical uniforms / r2 of the polar method
    assert ((ea >= -6.9) & (ea <= 2.3)).sum() >= 2048 and ea.min() < -35 and ea.max() > 35                      # log_scale_'s clamp range, and beyond
    # the twin really is the device's function: it differs from libm where the restatement says it is not implemented
    assert np.isnan(mm.hostabi.glibc_exp(np.array([600.0]))[0])
