import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native libraries are git-ignored build products: (re)build whatever is missing or stale before collecting
    import __graft_entry__ as entry
    entry.build_hip()
    entry.build_topo()
    entry.build_oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
