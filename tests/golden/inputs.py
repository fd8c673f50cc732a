"""
Seeded synthetic inputs shared by the golden generator (which feeds them to the REAL reference) and by the tests
(which feed them to the oracle / the HIP path).  Pure numpy; no reference code, no product code.
"""
import copy
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
    # 2-D, OAT15-like (BASELINE config C2 reduced): clustered cloud, NACA outline as GeometryCoordinates2D (refined),
    # n_cells_max stopping.  The reference evaluates the outline through shapely; the generator's stand-in
    # (ref_stubs.py) is the strict-interior rule the reference's own tests pin
    "refine_2d_polygon": dict(d=2, seed=79, n=40000, lo=[-0.2, -0.5], hi=[1.2, 0.5], body="polygon",
                              kw=dict(uniform_level=4, n_cells=6000)),
    # 3-D: prism + tetrahedron + (refined) pyramid bodies in one domain
    "refine_3d_polytopes": dict(d=3, seed=78, n=9000, lo=[0.0, 0.0, 0.0], hi=[1.0, 1.0, 1.0], body="polytopes",
                                kw=dict(uniform_level=2, min_metric=0.4)),
}


def naca_outline(n=120, chord=1.0, t=0.12):
    """closed NACA-00xx outline (the OAT15-like body of BASELINE config C2)"""
    xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, n // 2)))
    yt = 5 * t * (0.2969 * np.sqrt(xs) - 0.126 * xs - 0.3516 * xs ** 2 + 0.2843 * xs ** 3 - 0.1036 * xs ** 4)
    upper = np.stack([xs, yt], 1)
    lower = np.stack([xs[::-1], -yt[::-1]], 1)[1:-1]
    return np.concatenate([upper, lower]) * chord


def refine_inputs(name, geometry):
    """``geometry`` = module exposing CubeGeometry / SphereGeometry / CylinderGeometry3D (reference or product)."""
    case = REFINE_CASES[name]
    d = case["d"]
    x = cloud(case["seed"], case["n"], case["lo"], case["hi"])
    if case["body"] == "polygon":
        rng = np.random.default_rng(case["seed"])
        poly = naca_outline()
        x = np.concatenate([rng.random((case["n"] * 3 // 4, 2)) * [1.4, 1.0] + [-0.2, -0.5],
                            poly[rng.integers(0, len(poly), case["n"] // 4)] + 0.03 * rng.standard_normal((case["n"] // 4, 2))])
        y = 0.05 + np.exp(-8 * np.abs(x[:, 1])) * (1 + np.sin(6 * x[:, 0]) ** 2)
        geos = [geometry.CubeGeometry("domain", True, case["lo"], case["hi"]),
                geometry.GeometryCoordinates2D("airfoil", False, poly, refine=True, min_refinement_level=8)]
        return x, y, geos, case["kw"]
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


# ---- randomly drawn refine configurations (tools/fuzz_refine_vs_reference.py and the ``refine_random_*`` fixtures) ----
def random_bodies(rng, d):
    """[(class name, kwargs)] for 0-2 bodies inside the unit domain"""
    out = []
    for _ in range(int(rng.integers(0, 3))):
        refine = bool(rng.random() < 0.5)
        extra = dict(refine=refine)
        if refine and rng.random() < 0.5:
            extra["min_refinement_level"] = int(rng.integers(3, 6))
        c = rng.random(d) * 0.6 + 0.2
        r = float(rng.random() * 0.15 + 0.05)
        if d == 2:
            kind = rng.choice(["sphere", "triangle", "cube"])
            if kind == "sphere":
                out.append(("SphereGeometry", dict(position=c.tolist(), radius=r, **extra)))
            elif kind == "cube":
                out.append(("CubeGeometry", dict(lower_bound=(c - r).tolist(), upper_bound=(c + r).tolist(), **extra)))
            else:
                p = [(c + r * np.array([np.cos(a), np.sin(a)])).tolist() for a in rng.random() * 6.28 + np.array([0, 2.1, 4.2])]
                out.append(("TriangleGeometry", dict(points=[tuple(v) for v in p], **extra)))
        else:
            kind = rng.choice(["sphere", "cylinder", "cone", "cube", "prism", "tet", "pyramid"])
            if kind == "sphere":
                out.append(("SphereGeometry", dict(position=c.tolist(), radius=r, **extra)))
            elif kind == "cube":
                out.append(("CubeGeometry", dict(lower_bound=(c - r).tolist(), upper_bound=(c + r).tolist(), **extra)))
            elif kind in ("cylinder", "cone"):
                ax = rng.standard_normal(3); ax *= 0.3 / np.linalg.norm(ax)
                rad = r if kind == "cylinder" else [r, r * 0.4]
                out.append(("CylinderGeometry3D", dict(position=[tuple((c - ax).tolist()), tuple((c + ax).tolist())], radius=rad, **extra)))
            elif kind == "prism":
                a = int(rng.integers(0, 3))
                dims = [j for j in range(3) if j != a]
                tri = c[dims][None] + r * np.array([[-1.0, -0.8], [1.2, -0.5], [0.1, 1.1]])
                lo, hi = [], []
                for v in tri:
                    p0 = np.zeros(3); p0[dims] = v; p0[a] = c[a] - r
                    p1 = p0.copy(); p1[a] = c[a] + r
                    lo.append(tuple(p0.tolist())); hi.append(tuple(p1.tolist()))
                out.append(("PrismGeometry3D", dict(positions=[lo, hi], **extra)))
            elif kind == "tet":
                p = c[None] + r * 1.5 * np.array([[-1, -1, -1], [1.2, -0.7, -0.9], [0.1, 1.1, -0.8], [0.0, 0.1, 1.2]])
                out.append(("TetrahedronGeometry3D", dict(positions=p.tolist(), **extra)))
            else:
                z0 = float(c[2] - r)
                base = [[c[0] - r, c[1] - r, z0], [c[0] + r, c[1] - r, z0], [c[0] + r, c[1] + r, z0], [c[0] - r, c[1] + r, z0]]
                out.append(("PyramidGeometry3D", dict(nodes=[[float(v) for v in b] for b in base] + [[float(c[0]), float(c[1]), float(c[2] + 1.5 * r)]], **extra)))
    return out


def build_geometries(geometry, d, spec):
    geos = [geometry.CubeGeometry("domain", True, [0.0] * d, [1.0] * d)]
    for i, (cls, kw) in enumerate(spec):
        kw = copy.deepcopy(kw)
        first = kw.pop(next(iter(kw)))                   # the positional geometry argument
        geos.append(getattr(geometry, cls)(f"body{i}", False, first, **kw))
    return geos




def random_refine_case(seed):
    """(x, y, body spec, SamplingTree keyword arguments, d) of the randomly drawn configuration ``seed``: dimension, cloud,
    metric, stopping rule, cell ramp, 2:1 balance, pre_select, 0-2 bodies.  N >= 2100 and ramps >= 2 keep the reference
    away from its single-cell-iteration IndexError."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.integers(2, 4))
    n = int(rng.integers(2100, 5000))
    x = rng.random((n, d))
    c0 = rng.random(d)
    y = 0.05 + np.exp(-rng.uniform(2, 12) * np.linalg.norm(x - c0, axis=1)) * (1 + 0.3 * np.sin(9 * x[:, 0]))
    kw = dict(uniform_level=int(rng.integers(1, 4 if d == 2 else 3)))
    if rng.random() < 0.6:
        kw["min_metric"] = float(rng.uniform(0.3, 0.9))
    else:
        kw["n_cells"] = int(rng.integers(100, 1500))
    if rng.random() < 0.3:
        kw["max_delta_level"] = True
    if rng.random() < 0.3:
        kw["n_cells_iter_start"], kw["n_cells_iter_end"] = int(rng.integers(2, 30)), int(rng.integers(2, 10))
    if rng.random() < 0.3:
        kw["relTol"] = float(10.0 ** rng.uniform(-4, -1.5))
    if rng.random() < 0.3:
        kw["reach_at_least"] = float(rng.uniform(0.3, 0.95))
    if rng.random() < 0.2:
        kw["pre_select"] = True
    return x, y, random_bodies(rng, d), kw, d


# ---- BASELINE config C2 (OAT15-like) at full size ---------------------------------------------------------------------
def naca_outline_c2(n=200, thickness=0.12):
    """closed NACA-00xx outline with ``n`` distinct vertices, chord [0, 1] (upper side leading -> trailing edge, then back)"""
    xs = 0.5 * (1 - np.cos(np.linspace(0, np.pi, n // 2 + 1)))
    yt = 5 * thickness * (0.2969 * np.sqrt(xs) - 0.126 * xs - 0.3516 * xs ** 2 + 0.2843 * xs ** 3 - 0.1036 * xs ** 4)
    upper = np.stack([xs, yt], 1)
    lower = np.stack([xs[::-1], -yt[::-1]], 1)[1:-1]
    return np.ascontiguousarray(np.concatenate([upper, lower]))


def c2_cloud():
    """3*10^5 points in [-0.2, 1.2] x [-0.5, 0.5]: half uniform, half clustered around the airfoil outline (seed 1)"""
    rng = np.random.default_rng(1)
    poly = naca_outline_c2()
    far = rng.random((150000, 2)) * [1.4, 1.0] + [-0.2, -0.5]
    near = poly[rng.integers(0, len(poly), 150000)] + 0.02 * rng.standard_normal((150000, 2))
    x = np.concatenate([far, near])
    keep = (x[:, 0] >= -0.2) & (x[:, 0] <= 1.2) & (x[:, 1] >= -0.5) & (x[:, 1] <= 0.5)
    return np.ascontiguousarray(x[keep]), poly


def c2_fields(x, t0, t1):
    """snapshots t0..t1-1 of a buffet-like synthetic flow: p [N, 1, T], U [N, 2, T] float32 (a shock region oscillating on
    the suction side + a wake shedding behind the trailing edge)"""
    t = np.arange(t0, t1, dtype=np.float64)[None, :]
    xx, yy = x[:, :1], x[:, 1:2]
    shock = np.exp(-((xx - 0.45 - 0.08 * np.sin(2 * np.pi * t / 80.0)) / 0.05) ** 2) * np.exp(-((yy - 0.12) / 0.18) ** 2)
    wake = np.where(xx > 1.0, np.exp(-(xx - 1.0) / 0.5), 0.0) * np.exp(-(yy / 0.06) ** 2) * np.sin(2 * np.pi * (t / 25.0 - 3 * xx))
    p = 1.0 + 0.4 * shock + 0.15 * wake
    u = 1.0 - 0.5 * shock + 0.2 * wake
    v = 0.3 * wake * np.cos(2 * np.pi * t / 25.0) + 0.1 * shock
    return p.astype(np.float32)[:, None, :], np.stack([u, v], 1).astype(np.float32)


def c2_oat15(geometry, metric):
    """(x, metric, geometries, SamplingTree kwargs) of the full-size C2 run; ``metric`` comes from the fixture (computed
    once by gen_golden.py with torch on the CPU, rounded to float16 so that 3*10^5 values stay a small file)"""
    x, poly = c2_cloud()
    geos = [geometry.CubeGeometry("domain", True, [-0.2, -0.5], [1.2, 0.5]),
            geometry.GeometryCoordinates2D("airfoil", False, poly, refine=True)]
    return x, np.asarray(metric, dtype=np.float64), geos, dict(uniform_level=5, n_cells=25000)


# ----------------------------------------------------------------------------------------------------------------------
# export cases (fixture group ``export_*``): the same script of ExportData calls is run on the REAL reference (generator,
# h5py through h5py_standin.py) and on the product (tests).  Pure orchestration of the public API -- no reference code.
# ----------------------------------------------------------------------------------------------------------------------
EXPORT_CASES = {
    # scalar field in batches 7 + 7 + 3 with n_snapshots_total, then a 2-component field at once; one file
    "export_2d_batches": dict(refine="refine_2d_metric", n_t=17, times="str", script="batches"),
    # 3-D, interpolation at the cell vertices too, write times given as floats (dataset names = str(float))
    "export_3d_vertices": dict(refine="refine_3d_delta", n_t=5, times="float", script="vertices"),
    # one file per field
    "export_2d_newfile": dict(refine="refine_2d_metric", n_t=6, times="str", script="newfile"),
    # a second ExportData appends fields to the file of the first (append_existing=True); one field handed over as [N, T]
    "export_2d_append": dict(refine="refine_2d_triangle", n_t=6, times="int", script="append"),
}


def export_fields(x, n_t, seed):
    """p [N, 1, T], u [N, d, T] float32: smooth in space and time plus noise"""
    rng = np.random.default_rng(seed)
    t = np.arange(n_t, dtype=np.float64)
    phase = 2.0 * np.pi * (x[:, :1] / 0.4 - t[None, :] / 7.0)
    p = np.sin(phase) * np.exp(-x[:, 1:2]) + 1e-2 * rng.standard_normal((len(x), n_t))
    u = np.stack([np.cos(phase + 0.3 * j) * (1.0 + x[:, j:j + 1]) for j in range(x.shape[1])], 1)
    u = u + 1e-2 * rng.standard_normal(u.shape)
    return p[:, None, :].astype(np.float32), u.astype(np.float32)


def export_times(kind, n_t):
    if kind == "str":
        return [f"{0.05 * i:.2f}" for i in range(n_t)]
    if kind == "float":
        return [0.1 * i for i in range(n_t)]             # 0.30000000000000004: the names are str(t), whatever that is
    return list(range(n_t))


def run_export_case(name, s_cube, export_cls, to_tensor, x, y):
    """drive ``export_cls`` (the reference's or the product's ExportData) through the calls of case ``name``.  ``s_cube`` is
    the finished SparseSpatialSampling-like object of the same implementation, ``to_tensor`` turns a numpy array into the
    tensor type the implementation takes.  Returns {"error_second_file": exception class name or ""}."""
    case = EXPORT_CASES[name]
    n_t, times = case["n_t"], export_times(case["times"], case["n_t"])
    p, u = export_fields(x, n_t, seed=sum(map(ord, name)))
    xt = to_tensor(x)
    info = {"error_second_file": ""}
    if case["script"] == "batches":
        ex = export_cls(s_cube, write_times=times)
        for a, b in ((0, 7), (7, 14), (14, 17)):
            ex.export(xt, to_tensor(p[:, :, a:b]), "p", n_snapshots_total=n_t)
        ex.export(xt, to_tensor(u), "U")
    elif case["script"] == "vertices":
        ex = export_cls(s_cube, interpolate_at_vertices=True, write_times=times)
        for a, b in ((0, 3), (3, 5)):
            ex.export(xt, to_tensor(p[:, :, a:b]), "p", n_snapshots_total=n_t)
        ex.export(xt, to_tensor(u), "U", n_snapshots_total=n_t)
    elif case["script"] == "newfile":
        ex = export_cls(s_cube, write_new_file_for_each_field=True, write_times=times)
        for a, b in ((0, 4), (4, 6)):
            ex.export(xt, to_tensor(p[:, :, a:b]), "p", n_snapshots_total=n_t)
        try:
            ex.export(xt, to_tensor(u), "U")
        except Exception as err:                           # the reference cannot write a second file (see the fixture)
            info["error_second_file"] = type(err).__name__
    elif case["script"] == "append":
        ex = export_cls(s_cube, write_times=times)
        ex.export(xt, to_tensor(p), "p")
        ex2 = export_cls(s_cube, write_times=times, append_existing=True)
        for a, b in ((0, 2), (2, 6)):
            ex2.export(xt, to_tensor(u[:, :, a:b]), "U", n_snapshots_total=n_t)
        ex2.export(xt, to_tensor(np.ascontiguousarray(p[:, 0, :] * 2.0)), "q")          # scalar as [N, T]
    return info


def datawriter_script(dataloader_cls, datawriter_cls, directory, src_name, dst_name):
    """a user's script against ``data.Datawriter`` (the same calls for the reference's class and the product's): the grid copied from
    an existing S^3 file through a loader (``write_grid`` -> the writer knows the number of cells), temporal fields under plain names
    (the writer appends ``_center`` / ``_vertices`` by the leading size, data.py:388-391), names that carry their suffix already, an
    int and a float time step, the warning path of a missing time step (-> ``data/0``), a duplicate (logged and skipped,
    data.py:403-407), constants incl. a scalar, then the XDMF file"""
    import torch as pt
    rng = np.random.default_rng(17)
    loader = dataloader_cls(directory, src_name)
    n_c, n_v, d = loader.vertices.shape[0], loader.nodes.shape[0], loader.vertices.shape[1]
    w = datawriter_cls(directory, dst_name)
    w.write_grid(loader)
    assert w.n_cells == n_c
    w.write_data("levels", group="constant", data=loader.levels.unsqueeze(-1))
    w.write_data("scale", group="constant", data=3.25)
    w.write_data("flag", group="constant", data=pt.from_numpy(rng.integers(0, 2, n_v).astype(np.int32)))
    for t in (0.5, 2, "10"):
        w.write_data("p", group="data", time_step=t, data=pt.from_numpy(rng.random(n_c)))                 # -> p_center
        w.write_data("U", group="data", time_step=t, data=pt.from_numpy(rng.random((n_v, d))))            # -> U_vertices
        w.write_data("T_center", group="data", time_step=t, data=pt.from_numpy(rng.random((n_c, 1))))      # keeps its name
    w.write_data("p", group="data", time_step=2, data=pt.from_numpy(rng.random(n_c)))                     # exists: skipped
    w.write_data("q", group="data", data=pt.from_numpy(rng.random(n_c)))                                  # no time step -> data/0
    w.write_data("k", time_step="0.5", data=pt.from_numpy(rng.random(n_c).astype(np.float32)))            # a time step alone means "data"
    w.write_xdmf_file()


def random_export_case(seed):
    """a randomly drawn export: dimension, cloud, snapshot count, batch cuts, flags, kind of write times, scalar handed over as
    [N, T] or [N, 1, T] -- the script both implementations run is ``run_random_export``"""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.integers(2, 4))
    n_t = int(rng.integers(1, 10))
    cuts = sorted(set(int(c) for c in rng.integers(1, n_t, size=int(rng.integers(0, 3))))) if n_t > 1 else []
    return dict(d=d, n=int(rng.integers(900, 2200)), n_t=n_t, cuts=[0] + cuts + [n_t], vertices=bool(rng.integers(0, 2)),
                times=["str", "float", "int"][int(rng.integers(0, 3))], flat_scalar=bool(rng.integers(0, 2)),
                append=bool(rng.integers(0, 2)), new_file=bool(rng.integers(0, 2)), seed=int(seed),
                uniform=int(rng.integers(2, 4)), min_metric=float(rng.uniform(0.3, 0.6)))


def random_export_cloud(case, geometry):
    rng = np.random.default_rng(5000 + case["seed"])
    d = case["d"]
    hi = [1.0, 0.6, 0.5][:d]
    x = rng.random((case["n"], d)) * hi
    y = wake_metric(np.concatenate([x, np.zeros((len(x), 3 - d))], 1)[:, :3] if d == 2 else x, [0.3, 0.3, 0.2][:d] + [0.0] * (3 - d), decay=4.0)
    geos = [geometry.CubeGeometry("domain", True, [0.0] * d, hi)]
    # (a constant six cells per refinement step: a step that selects exactly ONE cell is an IndexError in the reference, s_cube.py:883,
    # and its ramp, s_cube.py:287-315, ends at one)
    return x, y[:len(x)], geos, dict(uniform_levels=case["uniform"], min_metric=case["min_metric"], n_cells_iter_start=6, n_cells_iter_end=6)


def run_random_export(case, s_cube, export_cls, to_tensor, x):
    n_t, times = case["n_t"], export_times(case["times"], case["n_t"])
    p, u = export_fields(x, n_t, seed=case["seed"])
    xt = to_tensor(x)
    ex = export_cls(s_cube, write_new_file_for_each_field=case["new_file"], interpolate_at_vertices=case["vertices"], write_times=times)
    for a, b in zip(case["cuts"][:-1], case["cuts"][1:]):
        piece = np.ascontiguousarray(p[:, 0, a:b]) if case["flat_scalar"] else p[:, :, a:b]
        ex.export(xt, to_tensor(piece), "p", n_snapshots_total=n_t)
    if case["new_file"]:
        return                                   # (a second file is what the reference cannot write: export_2d_newfile pins that)
    target = export_cls(s_cube, interpolate_at_vertices=case["vertices"], write_times=times, append_existing=True) if case["append"] else ex
    target.export(xt, to_tensor(u), "U")


def describe_facade(s3, directory):
    """what ``SparseSpatialSampling.execute_grid_generation()`` leaves behind, as JSON-able facts (the same function describes the
    reference's object and the product's): public attributes (type, dtype, shape, checksum), the saved ``mesh_info_<name>.pt``
    (keys in order, value types, the values that are not wall-clock times) and the pickled ``s_cube_<name>.pt`` loaded back"""
    import os
    import torch as pt

    def fact(v):
        if isinstance(v, pt.Tensor):
            return ["tensor", str(v.dtype), list(v.shape), sha(v.detach().cpu().contiguous().numpy())]
        if isinstance(v, (list, tuple)):
            return [type(v).__name__, len(v), [float(e) for e in v]]
        return [type(v).__name__, v if isinstance(v, (int, float, str, bool, type(None))) else repr(v)]
    public = ("n_jobs", "save_path", "save_name", "grid_name", "n_dimensions", "size_initial_cell", "centers", "vertices", "faces",
              "levels", "coordinates", "metric")
    info = pt.load(os.path.join(directory, f"mesh_info_{s3.save_name}.pt"), weights_only=False)
    again = pt.load(os.path.join(directory, f"s_cube_{s3.save_name}.pt"), weights_only=False)
    return {"attributes": {k: fact(getattr(s3, k)) for k in public},
            "mesh_info_keys": list(info), "mesh_info": {k: fact(v) for k, v in info.items() if not k.startswith("t_")},
            "mesh_info_time_types": {k: type(v).__name__ for k, v in info.items() if k.startswith("t_")},
            "pickled": {k: fact(getattr(again, k)) for k in public}, "sampling_dropped": getattr(again, "_sampling", "missing") is None,
            "files": sorted(f for f in os.listdir(directory) if f.endswith(".pt"))}


def invalid_calls(geometry, s3_cls, tree_cls):
    """[(label, thunk)]: calls the reference rejects (geometry constructors `_check_geometry`, `SparseSpatialSampling._check_input`,
    `SamplingTree` dimension checks) and a few it accepts -- the same list is run on the reference's classes and on the product's, the
    outcomes (exception TYPE, or "ok") are compared"""
    import torch as pt
    g = geometry
    x2, m2 = pt.rand((50, 2), generator=pt.Generator().manual_seed(0)).double(), pt.ones(50, dtype=pt.float64)
    dom2 = lambda: g.CubeGeometry("domain", True, [0, 0], [1, 1])
    tet = [[0.1, 0.1, 0.1], [0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9]]
    pyr = [[0.3, 0.3, 0.2], [0.7, 0.3, 0.2], [0.7, 0.7, 0.2], [0.3, 0.7, 0.2], [0.5, 0.5, 0.7]]
    tri3 = lambda z: [(0.1, 0.1, z), (0.3, 0.1, z), (0.1, 0.35, z)]
    calls = [
        ("cube ok", lambda: g.CubeGeometry("c", False, [0, 0], [1, 1])),
        ("cube empty name", lambda: g.CubeGeometry("", False, [0, 0], [1, 1])),
        ("cube keep_inside not bool", lambda: g.CubeGeometry("c", 1, [0, 0], [1, 1])),
        ("cube empty lower", lambda: g.CubeGeometry("c", False, [], [1, 1])),
        ("cube empty upper", lambda: g.CubeGeometry("c", False, [0, 0], [])),
        ("cube bounds of different length", lambda: g.CubeGeometry("c", False, [0, 0], [1, 1, 1])),
        ("cube lower >= upper", lambda: g.CubeGeometry("c", False, [0, 1], [1, 1])),
        ("cube min_refinement_level 0", lambda: g.CubeGeometry("c", False, [0, 0], [1, 1], refine=True, min_refinement_level=0)),
        ("sphere ok", lambda: g.SphereGeometry("s", False, [0.5, 0.5], 0.1)),
        ("sphere empty position", lambda: g.SphereGeometry("s", False, [], 0.1)),
        ("sphere radius 0", lambda: g.SphereGeometry("s", False, [0.5, 0.5], 0.0)),
        ("sphere radius list", lambda: g.SphereGeometry("s", False, [0.5, 0.5], [0.1])),
        ("cylinder ok", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], 0.1)),
        ("cone ok", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], [0.2, 0.0])),
        ("cylinder empty position", lambda: g.CylinderGeometry3D("c", False, [], 0.1)),
        ("cylinder one position", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0)], 0.1)),
        ("cylinder zero length", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 0)], 0.1)),
        ("cylinder radius str", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], "1")),
        ("cylinder radius 0", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], 0)),
        ("cylinder three radii", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], [0.1, 0.2, 0.3])),
        ("cylinder negative radius", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], [0.1, -0.2])),
        ("cylinder both radii 0", lambda: g.CylinderGeometry3D("c", False, [(0, 0, 0), (0, 0, 1)], [0.0, 0.0])),
        ("triangle ok", lambda: g.TriangleGeometry("t", False, [(0.3, 0.1), (0.6, 0.2), (0.3, 0.3)])),
        ("triangle two points", lambda: g.TriangleGeometry("t", False, [(0.3, 0.1), (0.6, 0.2)])),
        ("triangle 3-d points", lambda: g.TriangleGeometry("t", False, [(0.3, 0.1, 0), (0.6, 0.2, 0), (0.3, 0.3, 0)])),
        ("triangle zero area", lambda: g.TriangleGeometry("t", False, [(0.1, 0.1), (0.2, 0.2), (0.3, 0.3)])),
        ("triangle points str", lambda: g.TriangleGeometry("t", False, "abc")),
        ("prism ok", lambda: g.PrismGeometry3D("p", False, [tri3(0.1), tri3(0.4)])),
        ("prism empty", lambda: g.PrismGeometry3D("p", False, [])),
        ("prism one triangle", lambda: g.PrismGeometry3D("p", False, [tri3(0.1)])),
        ("prism triangle of two points", lambda: g.PrismGeometry3D("p", False, [tri3(0.1)[:2], tri3(0.4)[:2]])),
        ("tetrahedron ok", lambda: g.TetrahedronGeometry3D("t", False, tet)),
        ("tetrahedron three points", lambda: g.TetrahedronGeometry3D("t", False, tet[:3])),
        ("tetrahedron 2-d points", lambda: g.TetrahedronGeometry3D("t", False, [p[:2] for p in tet])),
        ("tetrahedron flat", lambda: g.TetrahedronGeometry3D("t", False, [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]])),
        ("tetrahedron empty", lambda: g.TetrahedronGeometry3D("t", False, [])),
        ("pyramid ok", lambda: g.PyramidGeometry3D("p", False, pyr)),
        ("pyramid four vertices", lambda: g.PyramidGeometry3D("p", False, pyr[:4])),
        ("pyramid vertex of two components", lambda: g.PyramidGeometry3D("p", False, [v[:2] for v in pyr])),
        ("facade metric 2-d", lambda: s3_cls(x2, m2.reshape(-1, 1), [dom2()], "/tmp", "x")),
        ("facade no geometries", lambda: s3_cls(x2, m2, [], "/tmp", "x")),
        ("facade no domain", lambda: s3_cls(x2, m2, [g.SphereGeometry("s", False, [0.5, 0.5], 0.1)], "/tmp", "x")),
        ("facade ok", lambda: s3_cls(x2, m2, [dom2()], "/tmp", "x", uniform_levels=0, min_metric=1.5)),
        ("tree geometry of another dimension", lambda: tree_cls(x2, m2, [dom2(), g.SphereGeometry("s", False, [0.5, 0.5, 0.5], 0.1)])),
        ("tree no domain", lambda: tree_cls(x2, m2, [g.SphereGeometry("s", False, [0.5, 0.5], 0.1)])),
    ]
    return calls


def outcomes_of(calls):
    out = {}
    for label, thunk in calls:
        try:
            thunk()
            out[label] = "ok"
        except BaseException as err:                 # noqa: BLE001 -- the TYPE is what is compared
            out[label] = type(err).__name__
    return out


def logged_run(geometry, s3_cls, export_cls, directory):
    """a small grid generation + export (scalar handed over as [N, T] in two batches, then a vector field; write times set late) with
    every ``logging`` record captured: [(level, message)] -- run on the reference's classes and on the product's"""
    import logging
    import torch as pt
    records = []

    class Capture(logging.Handler):
        def emit(self, r):
            records.append([r.levelname, r.getMessage()])
    root = logging.getLogger()
    saved, level = root.handlers[:], root.level
    root.handlers, _ = [Capture()], root.setLevel(logging.INFO)
    try:
        x, y, geos, kw = refine_inputs("refine_2d_metric", geometry)
        kw = {{"uniform_level": "uniform_levels", "n_cells": "n_cells_max"}.get(k, k): v for k, v in kw.items()}
        s3 = s3_cls(pt.from_numpy(x), pt.from_numpy(y), geos, directory, "case", n_jobs=1, **kw)
        s3.execute_grid_generation()
        p, u = export_fields(x, 4, 1)
        ex = export_cls(s3)
        ex.write_times = ["0", "1", "2", "3"]
        ex.export(pt.from_numpy(x), pt.from_numpy(np.ascontiguousarray(p[:, 0, :2])), "p", n_snapshots_total=4)
        ex.export(pt.from_numpy(x), pt.from_numpy(p[:, :, 2:]), "p", n_snapshots_total=4)
        ex.export(pt.from_numpy(x), pt.from_numpy(u), "U")
    finally:
        root.handlers = saved
        root.setLevel(level)
    return records
