"""
The reference's OWN unit-test files, unmodified and where they lie (/root/reference/sparseSpatialSampling/tests/test_*.py: read and
executed, never copied), run against THIS package (development container only; own process):

    python tests/golden/run_reference_tests.py [pytest arguments]

How: the import name ``sparseSpatialSampling`` is this repository's alias package (compat/), whose submodules ``geometry``, ``s_cube``,
``data`` ... ARE the modules of ``sparsespatialsampling_amd``; the reference's package directory is appended to the alias' ``__path__`` so
that the one subpackage the alias does not have -- ``sparseSpatialSampling.tests`` -- is found there.  The tests' relative imports
(``from ..geometry import CubeGeometry``, ``from ..s_cube import SamplingTree``, ``from ..data import Dataloader``) therefore bind to this
build's classes, which the runner verifies before it starts pytest.  Without a GPU the tree's device backend is replaced by the
oracle-backed one (test tooling, as tests/test_tree_host_logic.py does); everything else is the product.  Nothing is written into
/root/reference (no bytecode, no pytest cache).  ``test_geometry_STL.py`` is not collected: the STL geometry is out of scope.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "compat"), ROOT, HERE]

import sparseSpatialSampling  # noqa: E402  (the alias)
import sparsespatialsampling_amd  # noqa: E402

assert sparseSpatialSampling.__file__.startswith(os.path.join(ROOT, "compat")), sparseSpatialSampling.__file__
sparseSpatialSampling.__path__.append(os.path.join(REF, "sparseSpatialSampling"))
for name in ("geometry", "s_cube", "data", "geometry.geometry_base"):
    mod = sys.modules[f"sparseSpatialSampling.{name}"]
    assert mod.__file__.startswith(os.path.join(ROOT, "sparsespatialsampling_amd")), (name, mod.__file__)

import torch as pt  # noqa: E402

if not pt.cuda.is_available():
    import sparsespatialsampling_amd.s_cube as s_cube
    from tests.oracle_backend import OracleTreeBackend
    s_cube._make_backend = lambda v, t, k: OracleTreeBackend(v, t, k)

import pytest  # noqa: E402

os.chdir(REF)                                       # (test_s_cube_dataloader.py opens "sparseSpatialSampling/tests/<file>")
tests = os.path.join(REF, "sparseSpatialSampling", "tests")
files = sorted(os.path.join(tests, f) for f in os.listdir(tests) if f.startswith("test_") and f.endswith(".py") and "STL" not in f)
sys.exit(pytest.main(["-q", "-p", "no:cacheprovider", "--rootdir", "/tmp", "-c", os.devnull, *sys.argv[1:], *files]))
