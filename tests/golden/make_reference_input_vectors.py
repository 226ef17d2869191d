#!/usr/bin/env python3
"""Extracts the input-side vectors the reference's own tests hold into
tests/golden/reference_input_vectors.json: the literal numbers, file contents and expected values / error kinds of

    tests/utils/GetCalibrationDataTests.cpp   (matrix constructor, getInitialActiveCases, getInitialSEPAIHRDState)
    tests/utils/ReadContactMatrixTests.cpp    (readMatrixFromCSV: one good file, five failure kinds)
    tests/utils/FileUtilsTests.cpp            (joinPaths, FileUtils::readSEPAIHRDParameters)

as DATA -- the test source itself is read here as text and not kept.  Run in the build container (needs
/root/reference); the GPU box only sees the JSON.  tests/test_reference_input_vectors.py replays the cases against
config_io.py.

    python tests/golden/make_reference_input_vectors.py
"""
import json
import os
import re
import sys

REF_TESTS = os.environ.get("MMID_REFERENCE", "/root/reference") + "/tests/utils"
HERE = os.path.dirname(os.path.abspath(__file__))
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def read(name):
    with open(os.path.join(REF_TESTS, name)) as fh:
        return fh.read()


def c_string(lit: str) -> str:
    """Value of a C string literal's inside (the escapes these files use)."""
    return (lit.replace("\\n", "\n").replace("\\t", "\t").replace('\\"', '"').replace("\\\\", "\\"))


def block_after(src: str, start: int) -> str:
    """The brace block that opens at or after `start` (contents without the outer braces)."""
    i = src.index("{", start)
    depth, j = 0, i
    in_str = False
    while j < len(src):
        ch = src[j]
        if in_str:
            if ch == "\\":
                j += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
        elif ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 0:
                return src[i + 1:j]
        j += 1
    raise ValueError("unbalanced block")


def test_bodies(src: str) -> dict:
    out = {}
    for m in re.finditer(r"TEST(?:_F)?\(\s*(\w+)\s*,\s*(\w+)\s*\)", src):
        out[m.group(2)] = (block_after(src, m.end()), src.count("\n", 0, m.start()) + 1)
    return out


def files_written(body: str) -> dict:
    """{stream variable: (path expression, contents)} for every `std::ofstream v(path); v << "...";` in body."""
    out = {}
    for m in re.finditer(r"std::ofstream\s+(\w+)\(([^)]*)\)\s*;", body):
        var = m.group(1)
        text = "".join(c_string(x.group(1)) for x in re.finditer(r"\b%s\s*<<\s*\"((?:[^\"\\]|\\.)*)\"\s*;" % re.escape(var), body[m.end():]))
        out[var] = (m.group(2).strip(), text)
    return out


def numbers(text: str):
    return [float(x) for x in re.findall(NUM, text)]


def contact_matrix():
    src = read("ReadContactMatrixTests.cpp")
    fixture = block_after(src, src.index("class ReadContactMatrixTest"))
    setup = block_after(fixture, fixture.index("void SetUp()"))
    path_names = dict(re.findall(r"(\w+)\s*=\s*FileUtils::joinPaths\(testDir,\s*\"([^\"]+)\"\)", setup))
    path_names.update(re.findall(r"std::string\s+(\w+)\s*=\s*\"([^\"]+)\"\s*;", fixture))  # nonExistentPath
    files = {path_names[path]: text for path, text in files_written(setup).values()}
    cases = []
    for name, (body, line) in test_bodies(src).items():
        call = re.search(r"readMatrixFromCSV\((\w+),\s*(\w+),\s*(\w+)\)", body)
        ints = dict(re.findall(r"int\s+(\w+)\s*=\s*(\d+)\s*;", body))
        rows, cols = (int(ints.get(v, v)) for v in call.group(2, 3))
        case = {"test": name, "line": line, "file": path_names[call.group(1)], "rows": rows, "cols": cols,
                "file_exists": path_names[call.group(1)] in files}
        kind = re.search(r"ErrorType::(\w+)", body)
        if kind:
            case["error"] = kind.group(1)
        else:
            vals = numbers(re.search(r"expected\s*<<([^;]*);", body).group(1))
            case["expected"] = [vals[r * cols:(r + 1) * cols] for r in range(rows)]
        cases.append(case)
    return {"source": "tests/utils/ReadContactMatrixTests.cpp", "files": files, "cases": cases}


def file_utils():
    src = read("FileUtilsTests.cpp")
    fixture = block_after(src, src.index("class FileUtilsFixture"))
    names = dict(re.findall(r"std::string\s+(\w+)\s*=\s*\"([^\"]+)\"\s*;", fixture))
    setup = block_after(fixture, fixture.index("void SetUp()"))
    files = {names[path]: text for path, text in files_written(setup).values()}
    bodies = test_bodies(src)
    join = [[c_string(a), c_string(b), c_string(c)] for a, b, c in
            re.findall(r"EXPECT_EQ\(FileUtils::joinPaths\(\"([^\"]*)\",\s*\"([^\"]*)\"\),\s*\"([^\"]*)\"\)", bodies["JoinPaths"][0])]
    cases = []
    for name, (body, line) in bodies.items():
        if not name.startswith("ReadSEPAIHRDParameters_"):
            continue
        local = dict(re.findall(r"std::string\s+(\w+)\s*=\s*\"([^\"]+)\"\s*;", body))
        for path, text in files_written(body).values():
            files[local.get(path, names.get(path, path))] = text
        call = re.search(r"readSEPAIHRDParameters\((\w+),\s*(\w+)\)", body)
        ints = dict(re.findall(r"int\s+(\w+)\s*=\s*(\d+)\s*;", body))
        fname = local.get(call.group(1)) or names[call.group(1)]
        case = {"test": name, "line": line, "file": fname, "file_exists": fname in files,
                "num_age_classes": int(ints.get(call.group(2), call.group(2)))}
        throws = re.search(r"epidemic::(\w+Exception)\s*\)\s*;", body)
        if "EXPECT_THROW" in body and throws:
            case["error"] = throws.group(1)
            msg = re.search(r"\.find\(\"([^\"]+)\"\)", body)
            if msg:
                case["message_contains"] = msg.group(1)
        else:
            scalars, elems, sizes = {}, {}, {}
            for field, idx_paren, idx_brack, val in re.findall(r"EXPECT_DOUBLE_EQ\(params\.(\w+)(?:\((\d+)\)|\[(\d+)\])?,\s*(%s)\)" % NUM, body):
                if idx_paren or idx_brack:
                    elems.setdefault(field, {})[idx_paren or idx_brack] = float(val)
                else:
                    scalars[field] = float(val)
            for field, size in re.findall(r"ASSERT_EQ\(params\.(\w+)\.size\(\),\s*(\w+)\)", body):
                sizes[field] = int(ints.get(size, size)) if (size.isdigit() or size in ints) else case["num_age_classes"]
            case["expect"] = {"scalars": scalars, "elements": elems, "sizes": sizes}
        cases.append(case)
    return {"source": "tests/utils/FileUtilsTests.cpp", "join_paths": join, "files": files, "parameter_cases": cases}


def eigen_inits(body: str, n: int) -> dict:
    """Literal Eigen initialisations of a test body: `v << a, b, c, d;`, `m.row(0) << ...;`, Constant / Ones / Zero."""
    out = {}
    for var, vals in re.findall(r"\b(\w+)(?:\.row\(0\))?\s*<<\s*((?:%s\s*,\s*)+%s)\s*;" % (NUM, NUM), body):
        out[var] = numbers(vals)
    for var, size, val in re.findall(r"(\w+)\s*=\s*Eigen::VectorXd::Constant\((NUM_AGE_CLASSES(?:\s*-\s*1)?),\s*(%s)\)" % NUM, body):
        out[var] = [float(val)] * (n - 1 if "-" in size else n)
    for var, rows in re.findall(r"(\w+)\s*=\s*Eigen::MatrixXd::Ones\((\d+),\s*NUM_AGE_CLASSES\)", body):
        out[var] = [[1.0] * n for _ in range(int(rows))]
    for var, size in re.findall(r"(\w+)\s*=\s*Eigen::VectorXd::(?:Zero|Ones)\((NUM_AGE_CLASSES(?:\s*-\s*1)?)\)", body):
        fill = 1.0 if re.search(r"%s\s*=\s*Eigen::VectorXd::Ones" % var, body) else 0.0
        out[var] = [fill] * (n - 1 if "-" in size else n)
    for var, expr in re.findall(r"double\s+(\w+)\s*=\s*([^;]+);", body):
        q = re.fullmatch(r"\s*(%s)\s*/\s*(%s)\s*" % (NUM, NUM), expr)
        if q:
            out[var] = float(q.group(1)) / float(q.group(2))
        elif re.fullmatch(r"\s*%s\s*" % NUM, expr):
            out[var] = float(expr)
    return out


def calibration_data():
    src = read("GetCalibrationDataTests.cpp")
    n = int(re.search(r"const int NUM_AGE_CLASSES\s*=\s*(\d+)", src).group(1))
    bodies = test_bodies(src)
    fixture = block_after(src, src.index("class CalibrationDataTest"))
    header = "".join(c_string(x) for x in re.findall(r"\"((?:[^\"\\]|\\.)*)\"", block_after(fixture, fixture.index("getValidHeaderLines()"))))
    # the argument order of the matrix constructor as the tests call it:
    # (new_c, new_h, new_i, new_d, pop, cum_c0, cum_d0, cum_h0, cum_i0, n)
    cases = []

    body, line = bodies["ConstructorWithMatrices_PopulatesDataCorrectly"]
    n_points = int(re.search(r"int n_points\s*=\s*(\d+)", body).group(1))
    mult = dict(re.findall(r"(\w+)\(r, c\)\s*=\s*\(r \+ 1\) \* (\d+) \+ c;", body))
    mats = {k: [[(r + 1) * int(m) + c for c in range(n)] for r in range(n_points)] for k, m in mult.items()}
    pop_scale = int(re.search(r"pop\(c\)\s*=\s*(\d+) \* \(c \+ 1\);", body).group(1))
    offs = dict(re.findall(r"(cum_\w+)\(c\)\s*=\s*(\d+) \+ c;", body))
    cases.append({"test": "ConstructorWithMatrices_PopulatesDataCorrectly", "line": line, "kind": "constructor",
                  "new_c": mats["new_c"], "new_h": mats["new_h"], "new_i": mats["new_i"], "new_d": mats["new_d"],
                  "pop": [pop_scale * (c + 1) for c in range(n)],
                  **{k: [int(v) + c for c in range(n)] for k, v in offs.items()},
                  "expect": {"num_data_points": n_points, "date0": re.search(r"getDates\(\)\[0\],\s*\"([^\"]+)\"", body).group(1),
                             "cumulative_row1_is_row0_plus_new_row0": True}})

    body, line = bodies["GetInitialActiveCases_ReturnsFirstRowCumulativeConfirmed"]
    v = eigen_inits(body, n)
    cases.append({"test": "GetInitialActiveCases_ReturnsFirstRowCumulativeConfirmed", "line": line, "kind": "initial_active_cases",
                  "n_points": 2, "pop": [float(re.search(r"Constant\(NUM_AGE_CLASSES,\s*(\d+)\)", body).group(1))] * n,
                  "cum_c0": v["cum_c0"], "expect": v["cum_c0"]})

    def state_case(name, kind="initial_state"):
        body, line = bodies[name]
        v = eigen_inits(body, n)
        case = {"test": name, "line": line, "kind": kind, "inputs": v}
        exact = [{"compartment": int(c), "age": int(a), "value": float(val)} for c, a, val in
                 re.findall(r"EXPECT_DOUBLE_EQ\(initial_state\((\d+) \* NUM_AGE_CLASSES \+ (\d+)\),\s*(%s)\)" % NUM, body)]
        le = [{"compartment": int(c), "age": int(a), "value": float(val)} for c, a, val in
              re.findall(r"EXPECT_LE\(initial_state\((\d+) \* NUM_AGE_CLASSES \+ (\d+)\),\s*(%s)\)" % NUM, body)]
        ge = [{"compartment": int(c), "age": int(a), "value": float(val)} for c, a, val in
              re.findall(r"EXPECT_GE\(initial_state\((\d+) \* NUM_AGE_CLASSES \+ (\d+)\),\s*(%s)\)" % NUM, body)]
        near = re.search(r"EXPECT_NEAR\(sum_comps,\s*pop\(age\),\s*(%s)\)" % NUM, body)
        case["expect"] = {"size": 11 * n, "equal": exact, "at_most": le, "at_least": ge,
                          "population_conserved_over_compartments_0_to_8_within": float(near.group(1)) if near else None,
                          "all_non_negative": "EXPECT_GE(initial_state(i), 0.0)" in body}
        return case

    cases.append(state_case("GetInitialSEPAIHRDState_CorrectlyCalculates"))
    cases.append(state_case("GetInitialSEPAIHRDState_HandlesLargeInitialValuesClampingCorrectly"))
    c = state_case("GetInitialSEPAIHRDState_HandlesInvalidRates")
    c["expect"] = {"no_throw": True}
    cases.append(c)
    for name, err in (("GetInitialActiveCases_ThrowsIfDataEmpty", "runtime_error"),
                      ("GetInitialSEPAIHRDState_ThrowsIfNoDataPoints", "runtime_error"),
                      ("GetInitialSEPAIHRDState_ThrowsIfPopMismatch", "invalid_argument"),
                      ("GetInitialSEPAIHRDState_ThrowsIfParameterSizeMismatch", "runtime_error"),
                      ("GetInitialSEPAIHRDState_ThrowsIfRequiredMatricesEmpty", "runtime_error")):
        body, line = bodies[name]
        assert "std::" + err in body, (name, err)
        cases.append({"test": name, "line": line, "kind": "throws", "error": err, "inputs": eigen_inits(body, n)})
    return {"source": "tests/utils/GetCalibrationDataTests.cpp", "num_age_classes": n,
            "csv_header": header, "cases": cases}


def main():
    out = {"generated_by": "tests/golden/make_reference_input_vectors.py from the reference's tests/utils/*.cpp (inputs and expected values only)",
           "contact_matrix": contact_matrix(), "file_utils": file_utils(), "calibration_data": calibration_data()}
    path = os.path.join(HERE, "reference_input_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print("wrote", path, {k: len(v.get("cases", v.get("parameter_cases", []))) for k, v in out.items() if isinstance(v, dict)})


if __name__ == "__main__":
    sys.exit(main())
