import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx():
    """One engine context on cuda:0 for the whole GPU session (fails loudly without a GPU)."""
    import pairec_amd as pa
    c = pa.Context(0)
    yield c
    c.close()
