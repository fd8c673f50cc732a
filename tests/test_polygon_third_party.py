"""
The polygon predicate against THIRD-PARTY point-in-polygon implementations (the reference's own, shapely / GEOS, is not installed
here -- VERDICT r4 "missing 3"; the predicate was so far pinned by the builder's own exact-arithmetic restatement only):

* ``sympy.geometry.Polygon.encloses_point`` -- exact rational arithmetic; "being on the border is considered False", i.e. the
  semantics of ``shapely.Point.within(Polygon)`` (reference geometry/coordinates_2d.py:70) INCLUDING the boundary cases: points on
  vertices, on edges, on the extension of an edge;
* ``matplotlib.path.Path.contains_points`` (Agg's crossing-number code) -- on many random points of random non-convex outlines,
  away from the boundary (where Agg makes no promise).

Host predicate (``_Outline.strictly_inside``) on the CPU; the device kernel is held against the host predicate bit for bit by
tests/test_polygon_predicate.py::test_device_kernel_* and the refine goldens.
"""
import numpy as np
import pytest

from sparsespatialsampling_amd.geometry.coordinates_2d import _Outline

sympy = pytest.importorskip("sympy")

S = 1.0 / 16.0
OUTLINES = {
    "square": [(4, 4), (12, 4), (12, 12), (4, 12)],
    "l_shape": [(2, 2), (14, 2), (14, 6), (6, 6), (6, 14), (2, 14)],
    "collinear_edges": [(2, 2), (8, 2), (14, 2), (14, 8), (14, 14), (8, 14), (2, 14), (2, 8)],
    "diamond": [(8, 1), (15, 8), (8, 15), (1, 8)],
    "comb": [(1, 1), (15, 1), (15, 13), (12, 13), (12, 5), (9, 5), (9, 13), (6, 13), (6, 5), (3, 5), (3, 13), (1, 13)],
    "star": [(8, 15), (10, 10), (15, 8), (10, 6), (8, 1), (6, 6), (1, 8), (6, 10)],
    "steps_clockwise": [(2, 12), (14, 12), (14, 6), (10, 6), (10, 4), (6, 4), (6, 2), (2, 2)],
}


def _on_border(verts, i, j):
    """(i / 32, j / 32) on an edge of the outline with vertices in sixteenths?  integer arithmetic in units of 1 / 32"""
    n = len(verts)
    for a in range(n):
        (x0, y0), (x1, y1) = verts[a], verts[(a + 1) % n]
        x0, y0, x1, y1 = 2 * x0, 2 * y0, 2 * x1, 2 * y1
        if (x1 - x0) * (j - y0) == (i - x0) * (y1 - y0) and min(x0, x1) <= i <= max(x0, x1) and min(y0, y1) <= j <= max(y0, y1):
            return True
    return False


@pytest.mark.parametrize("name", ["collinear_edges", "comb", "star"])          # (sympy takes ~17 s per outline)
def test_strict_interior_equals_sympy_encloses_point(name):
    """every point of a half-pitch lattice over the outline's box -- vertices, edge points, edge mid-points, points on the extension
    of edges, interior and exterior points -- classified exactly as sympy classifies it (exact rationals; border = outside)"""
    from sympy import Point2D, Polygon, Rational
    verts = OUTLINES[name]
    poly = Polygon(*[Point2D(Rational(x, 16), Rational(y, 16)) for x, y in verts])
    ours = _Outline(np.array(verts, dtype=np.float64) * S)
    n_in = n_border = 0
    for i in range(0, 33):
        for j in range(0, 33):
            px, py = Rational(i, 32), Rational(j, 32)
            want = bool(poly.encloses_point(Point2D(px, py)))
            got = ours.strictly_inside(float(px), float(py))
            assert got == want, (name, float(px), float(py), got, want)
            n_in += want
            n_border += _on_border(verts, i, j)
    assert n_in > 20 and n_border >= 8                       # the lattice does hit the interior and the boundary


def test_strict_interior_equals_matplotlib_away_from_the_boundary():
    """random star-shaped, non-convex outlines (40 vertices, radius between 0.3 and 1): 20 000 random points each, the ones closer
    than 1e-9 to an edge left out -- same verdict as matplotlib's ``Path.contains_points``"""
    path_mod = pytest.importorskip("matplotlib.path")
    rng = np.random.default_rng(5)
    for case in range(6):
        ang = np.sort(rng.random(40)) * 2 * np.pi
        rad = 0.3 + 0.7 * rng.random(40)
        verts = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
        if case % 2:
            verts = verts[::-1].copy()                       # clockwise outlines too
        pts = rng.random((20000, 2)) * 2.2 - 1.1
        a, b = verts, np.roll(verts, -1, axis=0)
        d = b - a
        t = np.clip(((pts[:, None, :] - a[None]) * d[None]).sum(-1) / (d * d).sum(-1)[None], 0, 1)
        dist = np.linalg.norm(pts[:, None, :] - (a[None] + t[..., None] * d[None]), axis=-1).min(1)
        keep = dist > 1e-9
        want = path_mod.Path(np.vstack([verts, verts[:1]]), closed=True).contains_points(pts[keep])
        ours = _Outline(verts)
        got = np.fromiter((ours.strictly_inside(float(x), float(y)) for x, y in pts[keep]), dtype=bool, count=int(keep.sum()))
        assert np.array_equal(got, want), (case, int((got != want).sum()))
        assert 0.2 < want.mean() < 0.8
