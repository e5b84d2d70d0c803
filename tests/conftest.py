import gzip
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_fixture(name):
    """Reference test inputs (twitter.json, ...), stored gzip-compressed under tests/golden/data."""
    with gzip.open(os.path.join(ROOT, "tests", "golden", "data", name + ".gz"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def twitter():
    return load_fixture("twitter.json")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import oracle
    oracle.build()


def number_vectors():
    """The reference's NumberParsingTest literal vectors (tests/golden/number_vectors.json, made by
    tests/golden/make_number_vectors.py): dicts with input, optional length, and double_bits | long | message."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "number_vectors.json")) as f:
        return json.load(f)
