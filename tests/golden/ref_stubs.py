"""
Import shims that let the *real* reference package (read-only at /root/reference) be imported in the
development container, where several of its third-party dependencies are not installed.

This file contains NO reference code.  It only pre-seeds ``sys.modules`` with small stand-ins for
packages the reference imports at module level:

* ``numba``      -> ``njit`` is an identity decorator, ``prange`` is ``range``
* ``flowtorch``  -> ``flowtorch.data.mask_box`` / ``mask_sphere`` with the semantics the reference's own
                    unit tests pin (inclusive bounds; tests/test_cube_geometry.py:46-78,
                    tests/test_sphere_geometry.py:43-75), dummy ``FOAMDataloader`` and ``SVD``
* ``shapely``    -> minimal ``Point`` / ``Polygon`` (strict-interior ``within``; bounds; boundary.is_closed)
* ``pyvista``, ``pymeshfix`` -> import-only dummies (never exercised by the golden generator)
* ``h5py``       -> ``h5py_standin.py`` (next to this file): ctypes on the HDF5 C library, independent of the product's
                    ``libs3h5.so``; lets the reference's real ``ExportData.export`` / ``Datawriter`` / ``XDMFWriter`` /
                    ``Dataloader`` run here.  Validated by the reference's own tests/test_s_cube_dataloader.py passing.

It is used ONLY by ``tests/golden/gen_golden.py`` (fixture generation in the dev container).  Nothing in the
product, the GPU tests, ``smoke()`` or ``bench.py`` imports it, and /root/reference never travels to the GPU box.
"""
import sys
import types

import numpy as np
import torch as pt

REFERENCE_ROOT = "/root/reference"


def _njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def deco(fn):
        return fn

    return deco


def _mask_box(vertices, lower, upper):
    mask = pt.ones(vertices.shape[0], dtype=pt.bool)
    for i in range(len(lower)):
        mask &= (vertices[:, i] >= lower[i]) & (vertices[:, i] <= upper[i])
    return mask


def _mask_sphere(vertices, center, radius):
    c = pt.as_tensor(center, dtype=vertices.dtype)
    return (vertices - c).norm(dim=1) <= radius


class _Boundary:
    is_closed = True


class _Polygon:
    """Simple polygon; ``contains_strict`` = point strictly inside (boundary excluded) by crossing number."""

    def __init__(self, coordinates):
        xy = np.asarray(coordinates, dtype=np.float64)
        if np.all(xy[0] == xy[-1]):
            xy = xy[:-1]
        self.xy = xy
        self.bounds = (xy[:, 0].min(), xy[:, 1].min(), xy[:, 0].max(), xy[:, 1].max())
        self.boundary = _Boundary()

    def contains_strict(self, px, py):
        x, y = self.xy[:, 0], self.xy[:, 1]
        n = len(x)
        inside = False
        for i in range(n):
            j = (i + 1) % n
            xi, yi, xj, yj = x[i], y[i], x[j], y[j]
            # on-edge -> not strictly inside
            cross = (xj - xi) * (py - yi) - (yj - yi) * (px - xi)
            if cross == 0 and min(xi, xj) <= px <= max(xi, xj) and min(yi, yj) <= py <= max(yi, yj):
                return False
            if (yi > py) != (yj > py):
                xint = xi + (py - yi) * (xj - xi) / (yj - yi)
                if px < xint:
                    inside = not inside
        return inside


class _Point:
    def __init__(self, xy):
        self.x, self.y = float(xy[0]), float(xy[1])

    def within(self, poly):
        return poly.contains_strict(self.x, self.y)


def install():
    """Seed ``sys.modules`` and put the reference on ``sys.path``.  Idempotent."""
    sys.dont_write_bytecode = True       # /root/reference is read-only by contract: importing it must not leave __pycache__ behind
    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")
        numba.njit = _njit
        numba.prange = range
        sys.modules["numba"] = numba

    if "flowtorch" not in sys.modules:
        ft = types.ModuleType("flowtorch")
        ft_data = types.ModuleType("flowtorch.data")
        ft_data.mask_box = _mask_box
        ft_data.mask_sphere = _mask_sphere
        ft_data.FOAMDataloader = type("FOAMDataloader", (), {})
        ft_an = types.ModuleType("flowtorch.analysis")
        ft_an.SVD = type("SVD", (), {})
        ft.data, ft.analysis = ft_data, ft_an
        sys.modules.update({"flowtorch": ft, "flowtorch.data": ft_data, "flowtorch.analysis": ft_an})

    if "shapely" not in sys.modules:
        sh = types.ModuleType("shapely")
        sh.Point, sh.Polygon = _Point, _Polygon
        sys.modules["shapely"] = sh

    if "pyvista" not in sys.modules:
        pv = types.ModuleType("pyvista")
        pv.PolyData = type("PolyData", (), {})
        pv.read = lambda *a, **k: None
        sys.modules["pyvista"] = pv

    if "pymeshfix" not in sys.modules:
        pm = types.ModuleType("pymeshfix")
        pm.MeshFix = type("MeshFix", (), {})
        sys.modules["pymeshfix"] = pm

    try:
        import h5py  # noqa: F401
    except ModuleNotFoundError:
        import importlib.util as _ilu
        import os as _os
        spec = _ilu.spec_from_file_location("h5py", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "h5py_standin.py"))
        h5 = _ilu.module_from_spec(spec)
        sys.modules["h5py"] = h5
        spec.loader.exec_module(h5)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the name ``sparseSpatialSampling`` must resolve to the REAL reference here, never to the import-name alias of this
    # repository (compat/sparseSpatialSampling): a generator or fuzzer that compared the package with itself would pin nothing
    import importlib.util
    import os
    spec = importlib.util.find_spec("sparseSpatialSampling")
    if not os.path.isdir(REFERENCE_ROOT):
        return                       # no reference on this machine (GPU box): nothing to import, nothing to confuse
    if spec is None or not str(spec.origin).startswith(REFERENCE_ROOT + "/"):
        raise ImportError(f"'sparseSpatialSampling' resolves to {getattr(spec, 'origin', None)}, not to the reference under "
                          f"{REFERENCE_ROOT}")


install()
