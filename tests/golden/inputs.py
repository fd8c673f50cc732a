"""
Seeded synthetic inputs shared by the golden generator (which feeds them to the REAL reference) and by the tests
(which feed them to the oracle / the HIP path).  Pure numpy; no reference code, no product code.
"""
import hashlib

import numpy as np


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def cloud(seed, n, lo, hi):
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    return lo + rng.random((n, len(lo))) * (hi - lo)


def wake_metric(xyz, centre, decay=6.0):
    """Smooth, strictly positive synthetic 'std-of-pressure' style metric peaking behind ``centre``."""
    d = xyz[:, :2] - np.asarray(centre)[None, :2]
    r = np.sqrt((d ** 2).sum(1))
    wake = np.exp(-((d[:, 1]) / 0.08) ** 2) * np.where(d[:, 0] > 0, np.exp(-d[:, 0]), 0.0)
    m = 0.05 + np.exp(-decay * r) + 0.8 * wake * (1 + 0.3 * np.sin(9.0 * d[:, 0]))
    if xyz.shape[1] == 3:
        m = m * (1.0 + 0.2 * np.cos(5.0 * xyz[:, 2]))
    return m


REFINE_CASES = {
    # 2-D cylinder-like: cube domain + refined sphere body, metric stopping
    "refine_2d_metric": dict(d=2, seed=71, n=6000, lo=[0.0, 0.0], hi=[2.2, 0.41], body="sphere",
                             kw=dict(uniform_level=4, min_metric=0.6)),
    # 2-D, n_cells_max stopping + n_cells_iter ramp arguments
    "refine_2d_ncells": dict(d=2, seed=72, n=5000, lo=[0.0, 0.0], hi=[2.2, 0.41], body="sphere_norefine",
                             kw=dict(uniform_level=3, n_cells=900, n_cells_iter_start=12, n_cells_iter_end=4)),
    # 2-D with 2:1 balance (max_delta_level=True) -> host-only neighbour walks
    "refine_2d_delta": dict(d=2, seed=73, n=4000, lo=[0.0, 0.0], hi=[2.2, 0.41], body="sphere",
                            kw=dict(uniform_level=3, min_metric=0.5, max_delta_level=True)),
    # 3-D: cube + cylinder body (refined), metric stopping
    "refine_3d_metric": dict(d=3, seed=74, n=20000, lo=[0.0, 0.0, 0.0], hi=[2.4, 2.0, 0.3], body="cylinder",
                             kw=dict(uniform_level=3, min_metric=0.45)),
    # 3-D with 2:1 balance: exercises the 26-slot neighbour walks of _check_nb / _check_constraint
    "refine_3d_delta": dict(d=3, seed=75, n=9000, lo=[0.0, 0.0, 0.0], hi=[1.0, 1.0, 1.0], body="sphere3d",
                            kw=dict(uniform_level=2, min_metric=0.4, max_delta_level=True)),
    # 3-D, n_cells_max stopping, cone body (two radii) refined to a fixed level
    "refine_3d_ncells_cone": dict(d=3, seed=76, n=12000, lo=[0.0, 0.0, 0.0], hi=[2.0, 1.0, 1.0], body="cone",
                                  kw=dict(uniform_level=3, n_cells=3000, n_cells_iter_start=20, n_cells_iter_end=5)),
    # 2-D: triangular body refined to a fixed level
    "refine_2d_triangle": dict(d=2, seed=77, n=5000, lo=[0.0, 0.0], hi=[2.2, 0.41], body="triangle",
                               kw=dict(uniform_level=3, min_metric=0.5)),
    # 3-D: prism + tetrahedron + (refined) pyramid bodies in one domain
    "refine_3d_polytopes": dict(d=3, seed=78, n=9000, lo=[0.0, 0.0, 0.0], hi=[1.0, 1.0, 1.0], body="polytopes",
                                kw=dict(uniform_level=2, min_metric=0.4)),
}


def refine_inputs(name, geometry):
    """``geometry`` = module exposing CubeGeometry / SphereGeometry / CylinderGeometry3D (reference or product)."""
    case = REFINE_CASES[name]
    d = case["d"]
    x = cloud(case["seed"], case["n"], case["lo"], case["hi"])
    if case["body"] == "triangle":
        centre = [0.4, 0.2]
        body = geometry.TriangleGeometry("wedge", False, [(0.3, 0.1), (0.6, 0.2), (0.3, 0.3)], refine=True,
                                         min_refinement_level=6)
        y = wake_metric(x, centre)
    elif case["body"] == "polytopes":
        centre = [0.4, 0.5, 0.5]
        body = [geometry.PrismGeometry3D("prism", False, [[(0.1, 0.1, 0.1), (0.3, 0.1, 0.1), (0.1, 0.35, 0.1)],
                                                          [(0.1, 0.1, 0.4), (0.3, 0.1, 0.4), (0.1, 0.35, 0.4)]]),
                geometry.TetrahedronGeometry3D("tet", False, [[0.6, 0.1, 0.1], [0.95, 0.15, 0.1], [0.7, 0.45, 0.15],
                                                              [0.75, 0.2, 0.5]]),
                geometry.PyramidGeometry3D("pyramid", False, [[0.3, 0.55, 0.2], [0.7, 0.55, 0.2], [0.7, 0.9, 0.2],
                                                              [0.3, 0.9, 0.2], [0.5, 0.7, 0.7]], refine=True)]
        y = wake_metric(x, centre, decay=4.0)
    elif d == 2:
        centre, rad = [0.2, 0.2], 0.05
        keep = ((x - np.asarray(centre)) ** 2).sum(1) > rad ** 2
        x = np.ascontiguousarray(x[keep])
        if case["body"] == "sphere":
            body = geometry.SphereGeometry("cylinder", False, centre, rad, refine=True, min_refinement_level=6)
        else:
            body = geometry.SphereGeometry("cylinder", False, centre, rad)
        y = wake_metric(x, centre)
    elif case["body"] == "sphere3d":
        centre, rad = [0.4, 0.5, 0.5], 0.12
        keep = ((x - np.asarray(centre)) ** 2).sum(1) > rad ** 2
        x = np.ascontiguousarray(x[keep])
        body = geometry.SphereGeometry("ball", False, centre, rad, refine=True)
        y = wake_metric(x, centre, decay=4.0)
    elif case["body"] == "cone":
        centre = [0.6, 0.5, 0.5]
        body = geometry.CylinderGeometry3D("cone", False, [(0.4, 0.5, 0.5), (0.9, 0.5, 0.5)], [0.2, 0.05], refine=True,
                                           min_refinement_level=5)
        y = wake_metric(x, centre, decay=3.0)
    else:
        centre, rad = [0.8, 1.0, 0.0], 0.15
        keep = ((x[:, :2] - np.asarray(centre[:2])) ** 2).sum(1) > rad ** 2
        x = np.ascontiguousarray(x[keep])
        body = geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], rad, refine=True)
        y = wake_metric(x, centre, decay=2.5)
    geos = [geometry.CubeGeometry("domain", True, case["lo"], case["hi"])] + (body if isinstance(body, list) else [body])
    return x, y, geos, case["kw"]


def mask_cells(d, n, rng):
    """Cell stream of the ``masks`` fixture: centre + half-width per cell."""
    c = rng.random((n, d)) * 2.0 - 0.5
    h = rng.random(n) * 0.4 + 0.01
    return c, h


# flat-faced bodies of the ``masks_polytopes`` fixture: constructor arguments per class; "*_dyadic" bodies have corners on
# the 1/8 lattice, so that lattice cells put nodes exactly on faces, edges and corners (ties of the sign tests)
POLYTOPES = {
    "tri_generic": ("TriangleGeometry", dict(points=[(0.1, 0.1), (0.9, 0.2), (0.4, 1.1)])),
    "tri_dyadic": ("TriangleGeometry", dict(points=[(-0.25, -0.125), (0.25, 1.0), (1.5, -0.125)])),
    "prism_generic": ("PrismGeometry3D", dict(positions=[[(0.1, 0.1, 0.2), (0.1, 0.9, 0.3), (0.1, 0.4, 1.0)],
                                                         [(0.8, 0.1, 0.2), (0.8, 0.9, 0.3), (0.8, 0.4, 1.0)]])),
    "prism_dyadic": ("PrismGeometry3D", dict(positions=[[(-0.25, -0.125, 0.0), (0.25, 1.0, 0.0), (1.5, -0.125, 0.0)],
                                                        [(-0.25, -0.125, 0.75), (0.25, 1.0, 0.75), (1.5, -0.125, 0.75)]])),
    "tet_generic": ("TetrahedronGeometry3D", dict(positions=[[-0.3, 0.1, -0.1], [1.1, -0.2, 0.0], [0.9, 1.2, 0.1],
                                                             [0.4, 0.5, 1.3]])),
    "tet_dyadic": ("TetrahedronGeometry3D", dict(positions=[[-0.5, -0.5, -0.5], [1.5, -0.5, -0.5], [-0.5, 1.5, -0.5],
                                                            [-0.5, -0.5, 1.5]])),
    "pyr_generic": ("PyramidGeometry3D", dict(nodes=[[0.0, 0.1, 0.05], [1.0, 0.0, 0.05], [1.1, 0.9, 0.05],
                                                     [0.1, 1.0, 0.05], [0.5, 0.5, 1.2]])),
    "pyr_dyadic": ("PyramidGeometry3D", dict(nodes=[[0.375, 0.375, 1.25], [-0.25, -0.25, -0.25], [1.0, -0.25, -0.25],
                                                    [1.0, 1.0, -0.25], [-0.25, 1.0, -0.25]])),
}


def polytope(geometry, key, keep_inside):
    """instance of POLYTOPES[key] from the given geometry module (reference or this package); arguments are copied
    because the constructors convert list entries in place"""
    import copy
    cls, kw = POLYTOPES[key]
    return getattr(geometry, cls)("g", keep_inside, **copy.deepcopy(kw))


def polytope_cells(d, rng):
    """300 random cells + 300 cells on the 1/8 lattice (half width 1/8 or 1/4): centre + half width per cell"""
    c, h = mask_cells(d, 300, rng)
    cl = rng.integers(-4, 13, size=(300, d)) / 8.0
    hl = np.where(rng.random(300) < 0.5, 0.125, 0.25)
    return np.concatenate([c, cl]), np.concatenate([h, hl])


def c1_cylinder2d(geometry):
    """BASELINE config C1 realised synthetically: ~14 000 points in the cylinder2D channel, wake-like metric,
    Cube domain + Sphere body refined to level 9 (reference examples/s3_for_cylinder2D_Re100.py:43-52)"""
    rng = np.random.default_rng(0)
    x = rng.random((14500, 2)) * [2.2, 0.41]
    x = np.ascontiguousarray(x[((x - [0.2, 0.2]) ** 2).sum(1) > 0.05 ** 2])
    m = (0.02 + np.exp(-((x[:, 1] - 0.2) / 0.08) ** 2) * np.where(x[:, 0] > 0.2, np.exp(-(x[:, 0] - 0.2)), 0)
         + np.exp(-20 * np.hypot(x[:, 0] - 0.2, x[:, 1] - 0.2)))
    geos = [geometry.CubeGeometry("domain", True, [0, 0], [2.2, 0.41]),
            geometry.SphereGeometry("cylinder", False, [0.2, 0.2], 0.05, refine=True, min_refinement_level=9)]
    return x, m, geos, dict(uniform_level=5, min_metric=0.75)
