import csv
import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blob():
    from desirna_amd import params
    return params.load_blob()


@pytest.fixture(scope="session")
def oracle(blob):
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle.Oracle(blob)


@pytest.fixture(scope="session")
def traj_golden():
    with gzip.open(os.path.join(GOLDEN, "traj_golden.csv.gz"), "rt") as fh:
        return list(csv.DictReader(fh))


@pytest.fixture(scope="session")
def example_inputs():
    return json.load(open(os.path.join(GOLDEN, "example_inputs.json")))


@pytest.fixture(scope="session")
def eterna_solutions():
    return list(csv.DictReader(open(os.path.join(GOLDEN, "eterna_v1_solutions.csv"))))


@pytest.fixture(scope="session")
def eterna_targets():
    return {r["name"]: r["structure"] for r in csv.DictReader(open(os.path.join(GOLDEN, "eterna_v1_targets.csv")))}
