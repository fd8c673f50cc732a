"""
Import-name alias: put this directory (``<repo>/compat``) and the repository root on ``PYTHONPATH`` and scripts written
against the reference -- ``from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling``,
``from sparseSpatialSampling.export import ExportData``, ``from sparseSpatialSampling.geometry import CubeGeometry`` ...
(reference sparseSpatialSampling/__init__.py, examples/*.py) -- run unmodified on the MI355X implementation.

Pure re-exports: every module name of the reference's package that this build covers is registered as an alias of the
corresponding ``sparsespatialsampling_amd`` module (same objects, nothing is wrapped).  It lives outside the repository
root on purpose: the test tooling imports the REAL reference under the same name from /root/reference.
"""
import importlib
import sys

from sparsespatialsampling_amd.version import __version__  # noqa: F401

_MODULES = {
    "s_cube": "s_cube", "export": "export", "data": "data", "const": "const", "version": "version",
    "sparse_spatial_sampling": "sparse_spatial_sampling", "utils": "utils",
    "geometry": "geometry", "geometry.geometry_base": "geometry.geometry_base",
    "geometry.cube_geometry": "geometry.cube_geometry", "geometry.sphere_geometry": "geometry.sphere_geometry",
    "geometry.cylinder_geometry": "geometry.cylinder_geometry", "geometry.coordinates_2d": "geometry.coordinates_2d",
    # the four polytope bodies live in one module here
    "geometry.triangle_geometry": "geometry.polytope_geometry", "geometry.prism_geometry": "geometry.polytope_geometry",
    "geometry.tetrahedron_geometry": "geometry.polytope_geometry", "geometry.pyramid_geometry": "geometry.polytope_geometry",
}

for _alias, _target in _MODULES.items():
    _module = importlib.import_module(f"sparsespatialsampling_amd.{_target}")
    sys.modules[f"{__name__}.{_alias}"] = _module
    if "." not in _alias:
        globals()[_alias] = _module
