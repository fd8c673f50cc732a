"""the ``sparseSpatialSampling`` import-name alias (compat/): a script written against the reference imports the MI355X
classes without an edit (reference examples/s3_for_cylinder2D_Re100.py:13-17 import lines)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
from sparseSpatialSampling.export import ExportData
from sparseSpatialSampling.geometry import CubeGeometry, SphereGeometry, CylinderGeometry3D, GeometryCoordinates2D
from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling
from sparseSpatialSampling.s_cube import SamplingTree
from sparseSpatialSampling.data import Dataloader, Datawriter, XDMFWriter
from sparseSpatialSampling.geometry.triangle_geometry import TriangleGeometry
from sparseSpatialSampling.const import GRID
import sparseSpatialSampling, sparsespatialsampling_amd.export, sparsespatialsampling_amd.s_cube
assert ExportData is sparsespatialsampling_amd.export.ExportData and SamplingTree is sparsespatialsampling_amd.s_cube.SamplingTree
assert sparseSpatialSampling.__file__.endswith("compat/sparseSpatialSampling/__init__.py") and GRID == "grid"
g = CubeGeometry("domain", True, [0, 0], [1, 1])
assert g.name == "domain" and sparseSpatialSampling.__version__
print("alias ok")
"""


def test_reference_import_names_resolve_to_this_package():
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    run = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert run.returncode == 0 and "alias ok" in run.stdout, run.stderr[-2000:]
