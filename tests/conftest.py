import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import mmid_amd_loader  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """Contexts the tests left to the garbage collector are destroyed HERE, while the HIP runtime is still whole, instead of in
    whatever order the interpreter tears modules down at exit (a runtime that has begun to unload can crash the process after
    the last test has passed and turn a green run into a non-zero exit code)."""
    import gc
    gc.collect()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mm():
    return mmid_amd_loader.load()


@pytest.fixture(scope="session")
def oracle_py():
    import oracle_py as op  # test infrastructure: the CPU checker
    op.load()
    return op


def _load(mm, name):
    return mm.SEPAIHRDProblem.load(os.path.join(GOLDEN, name))


@pytest.fixture()
def shipped(mm):
    return _load(mm, "shipped_problem.json")


@pytest.fixture()
def synth400(mm):
    return _load(mm, "synth_400d_n4.json")


@pytest.fixture()
def ref_fixture(mm):
    return _load(mm, "reference_test_fixture.json")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(GOLDEN, "golden_highprec.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
