"""Row a12 against the vectors the reference's OWN tests hold (tests/golden/reference_input_vectors.json, extracted by
tests/golden/make_reference_input_vectors.py): one mirrored test per reference test file.

    tests/utils/ReadContactMatrixTests.cpp:57-137   -> config_io.read_matrix_csv
    tests/utils/FileUtilsTests.cpp:89-320           -> config_io.join_paths, read_sepaihrd_parameters_fileutils
    tests/utils/GetCalibrationDataTests.cpp:89-364  -> config_io.CalibrationData, initial_sepaihrd_state, read_calibration_csv
"""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(GOLDEN, "reference_input_vectors.json")) as fh:
        return json.load(fh)


def _materialise(tmp_path, files):
    for name, text in files.items():
        (tmp_path / name).write_bytes(text.encode())


def test_read_contact_matrix_cases_of_the_reference(mm, vectors, tmp_path):
    """ReadContactMatrixTests.cpp: ReadValidMatrix, FileOpenError, InvalidNumberFormat, NotEnoughRows, NotEnoughColumns,
    EmptyFile -- the same files, the same dimensions, the reference's CSVReadException::ErrorType."""
    cio = mm.config_io
    block = vectors["contact_matrix"]
    _materialise(tmp_path, block["files"])
    seen = set()
    for case in block["cases"]:
        path = str(tmp_path / case["file"])
        assert os.path.exists(path) == case["file_exists"]
        if "error" in case:
            with pytest.raises(cio.CSVReadError) as e:
                cio.read_matrix_csv(path, case["rows"], case["cols"])
            assert e.value.kind == case["error"], case["test"]
        else:
            got = cio.read_matrix_csv(path, case["rows"], case["cols"])
            assert got.shape == (case["rows"], case["cols"])
            assert np.array_equal(got, np.array(case["expected"])), case["test"]  # isApprox there; the parse is exact
        seen.add(case["test"])
    assert seen == {"ReadValidMatrix", "FileOpenError", "InvalidNumberFormat", "NotEnoughRows", "NotEnoughColumns", "EmptyFile"}


def test_file_utils_cases_of_the_reference(mm, vectors, tmp_path):
    """FileUtilsTests.cpp: JoinPaths and every ReadSEPAIHRDParameters_* case (values, sizes, exception class, message text)."""
    cio = mm.config_io
    block = vectors["file_utils"]
    assert len(block["join_paths"]) == 8
    for a, b, expected in block["join_paths"]:
        assert cio.join_paths(a, b) == expected, (a, b)
    _materialise(tmp_path, block["files"])
    errors = {"FileIOException": cio.FileIOError, "DataFormatException": cio.DataFormatError}
    n_value_cases = n_error_cases = 0
    for case in block["parameter_cases"]:
        path = str(tmp_path / case["file"])
        assert os.path.exists(path) == case["file_exists"]
        if "error" in case:
            with pytest.raises(errors[case["error"]]) as e:
                cio.read_sepaihrd_parameters_fileutils(path, case["num_age_classes"])
            if "message_contains" in case:
                assert case["message_contains"] in str(e.value), case["test"]
            n_error_cases += 1
            continue
        par = cio.read_sepaihrd_parameters_fileutils(path, case["num_age_classes"])
        for name, value in case["expect"]["scalars"].items():
            assert par[name] == value, (case["test"], name)          # EXPECT_DOUBLE_EQ on parsed literals: exact
        for name, elems in case["expect"]["elements"].items():
            for idx, value in elems.items():
                assert par[name][int(idx)] == value, (case["test"], name, idx)
        for name, size in case["expect"]["sizes"].items():
            assert len(par[name]) == size, (case["test"], name)
        n_value_cases += 1
    assert (n_value_cases, n_error_cases) == (3, 8)


def _cd_from_inputs(cio, v, n):
    """CalibrationData as the reference tests build it: (new_c, new_h, new_i, new_d, pop, cum_c0, cum_d0, cum_h0, cum_i0, n);
    matrices the test leaves uninitialised (new_h / new_i / new_d dummies) are zeros here -- nothing reads them."""
    row = lambda key: np.array([v[key]]) if key in v else np.zeros((1, n))
    return cio.CalibrationData.from_matrices(row("new_c"), row("new_h"), row("new_i"), row("new_d"), v["pop"], v["cum_c0"],
                                             v["cum_d0"], v["cum_h0"], v["cum_i0"], n)


def test_calibration_data_cases_of_the_reference(mm, vectors, tmp_path):
    """GetCalibrationDataTests.cpp: the matrix constructor, getInitialActiveCases, getInitialSEPAIHRDState (the hand values
    of :163-227 -- I0 = 5, H0 = 2, ICU0 = 1, D0 = 0, CumH0 = 2, CumICU0 = 1 --, the clamping case :296, the zero-rate case
    :346) and every *_Throws* case with its exception class (runtime_error -> RuntimeError, invalid_argument -> ValueError)."""
    cio = mm.config_io
    block = vectors["calibration_data"]
    n = block["num_age_classes"]
    cases = {c["test"]: c for c in block["cases"]}
    assert len(cases) == 10

    c = cases["ConstructorWithMatrices_PopulatesDataCorrectly"]
    cd = cio.CalibrationData.from_matrices(c["new_c"], c["new_h"], c["new_i"], c["new_d"], c["pop"], c["cum_c0"], c["cum_d0"],
                                           c["cum_h0"], c["cum_i0"], n)
    assert cd.num_data_points == c["expect"]["num_data_points"] and cd.num_age_classes == n
    assert len(cd.dates) == c["expect"]["num_data_points"] and cd.dates[0] == c["expect"]["date0"]
    assert np.array_equal(cd.new_confirmed, c["new_c"]) and np.array_equal(cd.new_hospitalizations, c["new_h"])
    assert np.array_equal(cd.new_icu, c["new_i"]) and np.array_equal(cd.new_deaths, c["new_d"])
    assert np.array_equal(cd.population, c["pop"])
    for cum, first, new in ((cd.cumulative_confirmed, "cum_c0", "new_c"), (cd.cumulative_deaths, "cum_d0", "new_d"),
                            (cd.cumulative_hospitalizations, "cum_h0", "new_h"), (cd.cumulative_icu, "cum_i0", "new_i")):
        assert np.array_equal(cum[0], c[first])
        assert np.array_equal(cum[1], np.array(c[first]) + np.array(c[new][0]))

    c = cases["GetInitialActiveCases_ReturnsFirstRowCumulativeConfirmed"]
    ones = np.ones((c["n_points"], n))
    cd = cio.CalibrationData.from_matrices(ones, ones, ones, ones, c["pop"], c["cum_c0"], c["cum_c0"], c["cum_c0"], c["cum_c0"], n)
    assert np.array_equal(cd.initial_active_cases(), c["expect"])

    for name in ("GetInitialSEPAIHRDState_CorrectlyCalculates", "GetInitialSEPAIHRDState_HandlesLargeInitialValuesClampingCorrectly"):
        c = cases[name]
        v = c["inputs"]
        cd = _cd_from_inputs(cio, v, n)
        x = cio.initial_sepaihrd_state(cd, v["sigma_rate"], v["gamma_p_rate"], v["gamma_a_rate"], v["gamma_i_rate"],
                                       np.array(v["p_asymptomatic"]), np.array(v["h_hospitalization"]))
        e = c["expect"]
        assert x.size == e["size"]
        for item in e["equal"]:
            assert x[item["compartment"] * n + item["age"]] == item["value"], (name, item)
        for item in e["at_most"]:
            assert x[item["compartment"] * n + item["age"]] <= item["value"], (name, item)
        for item in e["at_least"]:
            assert x[item["compartment"] * n + item["age"]] >= item["value"], (name, item)
        tol = e["population_conserved_over_compartments_0_to_8_within"]
        assert np.all(np.abs(x.reshape(11, n)[:9].sum(axis=0) - np.array(v["pop"])) <= tol), name
        if e["all_non_negative"]:
            assert np.all(x >= 0.0)

    c = cases["GetInitialSEPAIHRDState_HandlesInvalidRates"]
    v = c["inputs"]
    m, vec = np.array(v["valid_matrix"]), v["valid_vector"]
    cd = cio.CalibrationData.from_matrices(m, m, m, m, vec, vec, vec, vec, vec, n)
    x = cio.initial_sepaihrd_state(cd, v["sigma_rate"], v["gamma_p_rate"], v["gamma_a_rate"], v["gamma_i_rate"],
                                   np.array(v["p_asymptomatic"]), np.array(v["h_hospitalization"]))  # EXPECT_NO_THROW
    assert v["sigma_rate"] == 0.0 and np.all(np.isfinite(x))

    py_error = {"runtime_error": RuntimeError, "invalid_argument": ValueError}
    empty, zero = np.zeros((0, n)), np.zeros(n)
    c = cases["GetInitialActiveCases_ThrowsIfDataEmpty"]
    cd = cio.CalibrationData.from_matrices(empty, empty, empty, empty, c["inputs"]["valid_pop"], zero, zero, zero, zero, n)
    with pytest.raises(py_error[c["error"]]):
        cd.initial_active_cases()
    c = cases["GetInitialSEPAIHRDState_ThrowsIfNoDataPoints"]
    v = c["inputs"]
    cd = cio.CalibrationData.from_matrices(empty, empty, empty, empty, zero, zero, zero, zero, zero, n)
    with pytest.raises(py_error[c["error"]]):
        cio.initial_sepaihrd_state(cd, v["sigma_rate"], v["gamma_p_rate"], v["gamma_a_rate"], v["gamma_i_rate"],
                                   np.array(v["p_asymptomatic"]), np.array(v["h_hospitalization"]))
    c = cases["GetInitialSEPAIHRDState_ThrowsIfPopMismatch"]
    m = np.array(c["inputs"]["valid_matrix"])
    with pytest.raises(py_error[c["error"]]):
        cio.CalibrationData.from_matrices(m, m, m, m, c["inputs"]["wrong_pop"], np.ones(n), np.ones(n), np.ones(n), np.ones(n), n)
    c = cases["GetInitialSEPAIHRDState_ThrowsIfParameterSizeMismatch"]
    v = c["inputs"]
    m, vec = np.array(v["valid_matrix"]), v["valid_vector"]
    cd = cio.CalibrationData.from_matrices(m, m, m, m, vec, vec, vec, vec, vec, n)
    for p, h in ((v["p_mismatch"], v["h_valid"]), (v["p_valid"], v["h_mismatch"])):
        with pytest.raises(py_error[c["error"]]):
            cio.initial_sepaihrd_state(cd, v["sigma_rate"], v["gamma_p_rate"], v["gamma_a_rate"], v["gamma_i_rate"],
                                       np.array(p), np.array(h))
    # ..._ThrowsIfRequiredMatricesEmpty: a CSV that holds the reference fixture's header line and no data row
    c = cases["GetInitialSEPAIHRDState_ThrowsIfRequiredMatricesEmpty"]
    header_only = tmp_path / "header_only.csv"
    header_only.write_text(block["csv_header"] + "\n")
    assert block["csv_header"].split(",")[0] == "date" and len(block["csv_header"].split(",")) == 37
    with pytest.raises(py_error[c["error"]]):
        cio.read_calibration_csv(str(header_only))
    # the same header with one data row in the shape of the fixture's getValidDataRowLines reads back by column NAME
    cols = block["csv_header"].split(",")
    row = ["2020-03-01"] + [str(10 * k) for k in range(1, len(cols))]
    one = tmp_path / "one_row.csv"
    one.write_text(block["csv_header"] + "\n" + ",".join(row) + "\n")
    cd = cio.read_calibration_csv(str(one))
    assert cd.num_data_points == 1 and cd.dates == ["2020-03-01"]
    assert np.array_equal(cd.population, [float(row[cols.index("population_" + b)]) for b in cio.AGE_BANDS])
    assert np.array_equal(cd.cumulative_icu[0], [float(row[cols.index("cumulative_intensive_care_patients_" + b)]) for b in cio.AGE_BANDS])
