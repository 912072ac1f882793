import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timing: wall-clock bounds on a shared device (GPU box, `-m timing`; not part of `-m gpu`)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    # the oracle is test infrastructure; build it once per session if missing/stale
    import oracle_lib
    oracle_lib.build_oracle()
