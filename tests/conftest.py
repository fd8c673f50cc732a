import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_BUILD_ERROR = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native libraries are git-ignored build products: (re)build whatever is missing or stale before collecting.  A build
    # that fails must not turn into "no tests collected" (VERDICT r3): the error is kept and EVERY test then fails with it.
    try:
        import __graft_entry__ as entry
        entry.build_hip()
        entry.build_topo()
        entry.build_h5()
        entry.build_oracle()
    except Exception as err:                         # compiler missing / compile error / ...
        _BUILD_ERROR.append(f"{type(err).__name__}: {err}")


@pytest.fixture(autouse=True, scope="session")
def native_libraries_built():
    if _BUILD_ERROR:
        pytest.fail("the native libraries could not be built (python -c 'import __graft_entry__ as g; g.build()'): " + _BUILD_ERROR[0],
                    pytrace=False)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
