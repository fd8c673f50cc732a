"""
ctypes/numpy front-end of the CPU oracle (oracle/s3_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- as the checker or
as the reported CPU baseline, never by the product package.  Parity of every function with the real reference is
pinned by tests/test_oracle_vs_golden.py against tests/golden/*.npz.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libs3oracle.so")
_lib = None

DIRS = {
    2: np.array([[-1, -1], [-1, 1], [1, 1], [1, -1]], dtype=np.float64),
    3: np.array([[-1, -1, 1], [-1, 1, 1], [1, 1, 1], [1, -1, 1],
                 [-1, -1, -1], [-1, 1, -1], [1, 1, -1], [1, -1, -1]], dtype=np.float64),
}


def build(force=False):
    src = os.path.join(_HERE, "s3_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libs3oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.s3o_sumsq.restype = C.c_double
        _lib.s3o_torch_inner_sum.restype = C.c_double
        _lib.s3o_numpy_pairwise_sum.restype = C.c_double
        _lib.s3o_grid_create.restype = C.c_void_p
        _lib.s3o_grid_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double]
        _lib.s3o_grid_destroy.argtypes = [C.c_void_p]
        _lib.s3o_grid_destroy.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_threads():
    return lib().s3o_num_threads()


def set_num_threads(n):
    lib().s3o_set_num_threads(int(n))


def knn(pts, q, k):
    pts, q = _f64(pts), _f64(q)
    n, d = pts.shape
    idx = np.empty((len(q), k), dtype=np.int64)
    dist = np.empty((len(q), k), dtype=np.float64)
    rc = lib().s3o_knn(_p(pts), C.c_int64(n), d, _p(q), C.c_int64(len(q)), k, _p(idx), _p(dist))
    assert rc == 0, rc
    return idx, dist


def idw_predict(pts, y, q, k):
    pts, y, q = _f64(pts), _f64(y), _f64(q)
    n, d = pts.shape
    out = np.empty(len(q), dtype=np.float64)
    rc = lib().s3o_idw_predict(_p(pts), C.c_int64(n), d, _p(y), _p(q), C.c_int64(len(q)), k, _p(out))
    assert rc == 0, rc
    return out


class GridIndex:
    """the exact k-nearest query of ``knn`` / ``idw_predict`` / ``child_gain`` through a bucket grid (the oracle's stand-in
    for the reference's kd-tree, s_cube.py:161-163): identical results, usable at the reference's problem sizes"""

    def __init__(self, pts, occupancy=3.0):
        self.pts = _f64(pts)
        self.n, self.dim = self.pts.shape
        self._h = C.c_void_p(lib().s3o_grid_create(_p(self.pts), C.c_int64(self.n), self.dim, C.c_double(occupancy)))
        assert self._h.value, "s3o_grid_create failed"

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().s3o_grid_destroy(self._h)
            self._h = C.c_void_p(0)

    __del__ = close

    def knn(self, q, k):
        q = _f64(q)
        idx = np.empty((len(q), k), dtype=np.int64)
        dist = np.empty((len(q), k), dtype=np.float64)
        rc = lib().s3o_grid_knn(self._h, _p(q), C.c_int64(len(q)), k, _p(idx), _p(dist))
        assert rc == 0, rc
        return idx, dist

    def idw_predict(self, y, q, k):
        y, q = _f64(y), _f64(q)
        out = np.empty(len(q), dtype=np.float64)
        rc = lib().s3o_grid_idw_predict(self._h, _p(y), _p(q), C.c_int64(len(q)), k, _p(out))
        assert rc == 0, rc
        return out

    def child_gain(self, y, k, centers, level, width, gain0):
        y, centers = _f64(y), _f64(centers)
        level = np.ascontiguousarray(level, dtype=np.int32)
        n, d = centers.shape
        tab = level_factor_table(width, d)
        metric = np.empty((n, 2 ** d + 1), dtype=np.float64)
        gain = np.empty(n, dtype=np.float64)
        rc = lib().s3o_grid_child_gain(self._h, _p(y), k, _p(centers), _p(level), C.c_int64(n), C.c_double(float(width)),
                                       _p(tab), C.c_double(float(gain0)), _p(metric), _p(gain))
        assert rc == 0, rc
        return metric, gain


def idw_weights(dist):
    dist = _f64(dist)
    w = np.empty_like(dist)
    rc = lib().s3o_idw_weights(_p(dist), C.c_int64(dist.shape[0]), dist.shape[1], _p(w))
    assert rc == 0, rc
    return w


def interp(w, idx, data):
    """data: [N, ...] float32/float64; returns float64 [Nc, ...]."""
    w = _f64(w)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    data = np.ascontiguousarray(data)
    assert data.dtype in (np.float32, np.float64)
    row_len = int(np.prod(data.shape[1:]))
    out = np.empty((w.shape[0],) + data.shape[1:], dtype=np.float64)
    rc = lib().s3o_interp(_p(w), _p(idx), C.c_int64(w.shape[0]), w.shape[1], _p(data),
                          0 if data.dtype == np.float32 else 1, C.c_int64(data.shape[0]), C.c_int64(row_len), _p(out))
    assert rc == 0, rc
    return out


def level_factor_table(width, n_dims, max_level=64):
    """1/2^d * (width/2^level)^d exactly as the reference evaluates it in Python (s_cube.py:1859)."""
    return np.array([1 / (2 ** n_dims) * ((width / (2 ** lv)) ** n_dims) for lv in range(max_level)], dtype=np.float64)


def child_gain(pts, y, k, centers, level, width, gain0):
    pts, y, centers = _f64(pts), _f64(y), _f64(centers)
    level = np.ascontiguousarray(level, dtype=np.int32)
    n, d = centers.shape
    tab = level_factor_table(width, d)
    metric = np.empty((n, 2 ** d + 1), dtype=np.float64)
    gain = np.empty(n, dtype=np.float64)
    rc = lib().s3o_child_gain(_p(pts), C.c_int64(len(pts)), d, _p(y), k, _p(centers), _p(level), C.c_int64(n),
                              C.c_double(float(width)), _p(tab), C.c_double(float(gain0)), _p(metric), _p(gain))
    assert rc == 0, rc
    return metric, gain


def sumsq(m):
    m = _f64(m)
    return lib().s3o_sumsq(_p(m), C.c_int64(len(m)))


def topn(gain, ids, n_top):
    gain = _f64(gain)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    n_top = min(int(n_top), len(ids))
    out = np.empty(n_top, dtype=np.int64)
    rc = lib().s3o_topn(_p(gain), _p(ids), C.c_int64(len(ids)), C.c_int64(n_top), _p(out))
    assert rc == 0, rc
    return out


def _cells(centers, level):
    centers = _f64(centers)
    level = np.ascontiguousarray(level, dtype=np.int32)
    return centers, level, np.empty(len(centers), dtype=np.uint8)


def mask_box(centers, level, width, lo, hi, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    lo, hi = _f64(lo), _f64(hi)
    lib().s3o_mask_box(_p(centers), _p(level), C.c_int64(len(centers)), centers.shape[1], C.c_double(float(width)),
                       _p(lo), _p(hi), int(refine_mode), int(keep_inside), _p(inv))
    return inv.astype(bool)


def mask_sphere(centers, level, width, pos, radius, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    pos = _f64(pos)
    lib().s3o_mask_sphere(_p(centers), _p(level), C.c_int64(len(centers)), centers.shape[1], C.c_double(float(width)),
                          _p(pos), C.c_double(float(radius)), int(refine_mode), int(keep_inside), _p(inv))
    return inv.astype(bool)


def cylinder_params(position, radius):
    """Parameter packing that mirrors cylinder_geometry.py:51-56 (positions rounded through float32)."""
    p = np.asarray(position, dtype=np.float32)
    axis = (p[1] - p[0]).astype(np.float64)
    norm = float(np.sqrt((axis * axis).sum()))          # torch Tensor.norm() of 3 doubles
    if isinstance(radius, (int, float)):
        r0 = r1 = float(radius)
        cone = 0
    else:
        r0, r1 = float(radius[0]), float(radius[1])
        cone = 1
    return p[0].astype(np.float64), axis, norm, r0, r1, cone


def mask_cylinder(centers, level, width, position, radius, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    p0, axis, norm, r0, r1, cone = cylinder_params(position, radius)
    lib().s3o_mask_cylinder(_p(centers), _p(level), C.c_int64(len(centers)), C.c_double(float(width)), _p(p0),
                            _p(axis), C.c_double(norm), C.c_double(r0), C.c_double(r1), cone, int(refine_mode),
                            int(keep_inside), _p(inv))
    return inv.astype(bool)


def mask_triangle(centers, level, width, points, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    pts = _f64(points).reshape(3, 2)
    lib().s3o_mask_triangle(_p(centers), _p(level), C.c_int64(len(centers)), C.c_double(float(width)), _p(pts),
                            int(refine_mode), int(keep_inside), _p(inv))
    return inv.astype(bool)


def mask_prism(centers, level, width, origin, axis, norm, dims, triangle, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    origin, axis, tri = _f64(origin), _f64(axis), _f64(triangle).reshape(3, 2)
    dims = np.ascontiguousarray(dims, dtype=np.int32)
    lib().s3o_mask_prism(_p(centers), _p(level), C.c_int64(len(centers)), C.c_double(float(width)), _p(origin),
                         _p(axis), C.c_double(float(norm)), _p(dims), _p(tri), int(refine_mode), int(keep_inside),
                         _p(inv))
    return inv.astype(bool)


def mask_tetrahedra(centers, level, width, positions, normals, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    pos, nrm = _f64(positions), _f64(normals)
    n_tets = pos.shape[0]
    assert pos.shape == (n_tets, 4, 3) and nrm.shape == (n_tets, 3, 4)
    lib().s3o_mask_tetrahedra(_p(centers), _p(level), C.c_int64(len(centers)), C.c_double(float(width)), _p(pos),
                              _p(nrm), n_tets, int(refine_mode), int(keep_inside), _p(inv))
    return inv.astype(bool)


def mask_polygon(centers, level, width, poly, refine_mode, keep_inside):
    centers, level, inv = _cells(centers, level)
    poly = _f64(poly)
    if np.all(poly[0] == poly[-1]):
        poly = np.ascontiguousarray(poly[:-1])
    lib().s3o_mask_polygon(_p(centers), _p(level), C.c_int64(len(centers)), C.c_double(float(width)), _p(poly),
                           len(poly), int(refine_mode), int(keep_inside), _p(inv))
    return inv.astype(bool)
