import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_full():
    with open(os.path.join(GOLDEN, "ntt_full_vectors.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_digests():
    with open(os.path.join(GOLDEN, "ntt_digests.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def golden_field():
    with open(os.path.join(GOLDEN, "field_constants.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def port():
    import oracle
    return oracle.port()
