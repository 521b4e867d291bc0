import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build libacmi.so / liboracle.so when sources are newer (hipcc cross-compiles without a GPU)."""
    from php_aho_corasick_amd import build
    try:
        build.build()
    except Exception as e:  # on the GPU box hipcc exists too; a prebuilt .so is fine otherwise
        if not os.path.exists(build.LIB):
            raise
        print("libacmi.so: using prebuilt library:", e)
    from oracle import pyoracle
    if not os.path.exists(pyoracle.PORT_SO):
        pyoracle.build()
    yield


@pytest.fixture(scope="session")
def golden():
    from helpers import load_json
    return {(c["dataset"], c["find_all"]): c for c in load_json("golden.json")["cases"]}


@pytest.fixture(scope="session")
def kat():
    from helpers import load_json
    return load_json("kat.json")["cases"]
