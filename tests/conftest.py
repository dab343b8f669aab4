import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as _o
    return _o.load()


@pytest.fixture(scope="session")
def reflib():
    import oracle as _o
    r = _o.load_ref()
    if r is None:
        pytest.skip("oracle/_ref/libkmpref.so not built (reference absent)")
    return r


@pytest.fixture(scope="session")
def fixture_counts():
    with open(os.path.join(GOLDEN, "fixture_counts.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def tokens(fixture_counts):
    return [t.encode() for t in fixture_counts["tokens"]]


@pytest.fixture(scope="session")
def kat_matcher():
    with open(os.path.join(GOLDEN, "kat_matcher.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat_extract():
    with open(os.path.join(GOLDEN, "kat_extract.json")) as f:
        return json.load(f)
