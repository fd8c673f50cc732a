"""
``GeometryCoordinates2D`` against expectation TABLES written out by hand from GEOS semantics -- for the host predicate
(``check_cell``) and for the device kernel (``s3_mask_polygon``) alike.

The reference asks shapely ``Point(node).within(Polygon(outline))`` per cell node (reference geometry/coordinates_2d.py:54-94).
``within`` is the DE-9IM pattern ``T*F**F***``: the point's interior must meet the polygon's INTERIOR and nothing of the point
may lie in the polygon's exterior.  A point has no boundary, so: strictly inside -> True; on an edge or on a vertex (the
polygon's boundary) -> False; outside -> False.  Collinear consecutive edges and duplicate closing points do not change the
point set of the polygon, hence not the answer.  shapely / GEOS are not installed here, so every expected value below is argued
from that definition by hand (never computed by a point-in-polygon routine): rows 1-3 are the reference's own six expectations
(tests/test_coordinates_2d_geometry.py:39-49 with the ``DummyCells`` squares of tests/const.py), the others put nodes on
vertices, on edges, on a vertex shared by collinear edges, and on the height of a horizontal edge -- where crossing-number code
classically miscounts.

Cell verdicts follow from the node flags by ``GeometryObject._apply_mask`` (reference geometry_base.py:40-76): body
(``keep_inside=False``): removed iff ALL nodes are within, selected for refinement iff ANY is; domain (``keep_inside=True``):
removed iff NO node is within, selected iff NOT ALL are.
"""
import numpy as np
import pytest
import torch as pt

from sparsespatialsampling_amd.geometry import GeometryCoordinates2D

T, F = True, False
# node order of a cell (reference s_cube.py:188-194): (-x,-y), (-x,+y), (+x,+y), (+x,-y)
REF_SQUARE = [(-1, -1), (-1, 1.25), (1.25, 1.25), (1.25, -1)]                     # the reference's test polygon
BOX2 = [(0, 0), (2, 0), (2, 2), (0, 2)]
# collinear bottom edges (0,0)-(1,0)-(2,0), a notch with the horizontal edge (2,1)-(3,1) at node height, closed explicitly
STEP = [(0, 0), (1, 0), (2, 0), (2, 1), (3, 1), (3, 2), (0, 2), (0, 0)]

# (polygon, cell centre, cell width = 2 * half edge, within-flag per node, why)
CASES = [
    (REF_SQUARE, (0.5, 0.5), 1.0, [T, T, T, T], "reference: cell_inside_2D, unit square well inside"),
    (REF_SQUARE, (5.5, 5.5), 1.0, [F, F, F, F], "reference: cell_outside_2D"),
    (REF_SQUARE, (1.0, 1.0), 1.0, [T, F, F, F], "reference: cell_partially_2D, only (0.5, 0.5) lies inside 1.25"),
    (BOX2, (0.5, 0.5), 1.0, [F, F, T, F], "(0,0) is a vertex, (0,1) and (1,0) lie on edges: boundary, not within; (1,1) interior"),
    (BOX2, (1.0, 1.0), 2.0, [F, F, F, F], "the cell IS the polygon: all four nodes are its vertices"),
    (BOX2, (1.0, 1.0), 1.0, [T, T, T, T], "nodes (0.5..1.5)^2: interior"),
    (BOX2, (2.0, 1.0), 2.0, [F, F, F, F], "(1,0), (1,2) on the bottom / top edge; (3,2), (3,0) outside"),
    (STEP, (2.5, 1.5), 1.0, [F, F, F, F], "(2,1), (3,2), (3,1) are vertices, (2,2) lies on the top edge"),
    (STEP, (1.5, 0.5), 1.0, [F, T, F, F], "(1,0): vertex between collinear edges; (1,1): interior although the ray towards +x runs "
                                          "along the horizontal edge (2,1)-(3,1); (2,1), (2,0): vertices"),
    (STEP, (0.5, 1.0), 1.0, [F, F, T, T], "(0,0.5), (0,1.5) on the left edge; (1,1.5), (1,0.5) interior"),
    (STEP, (-0.5, 0.5), 1.0, [F, F, F, F], "(-1,0), (-1,1) outside (their rays pass along the bottom edges / the notch edge and through "
                                           "vertices); (0,1) on the left edge; (0,0) vertex"),
    (STEP, (2.5, 0.5), 0.5, [F, F, F, F], "inside the notch x in (2,3), y in (0,1): exterior of the polygon"),
    (STEP, (2.5, 0.5), 1.0, [F, F, F, F], "(2,0), (2,1), (3,1): vertices; (3,0): outside"),
    (STEP, (1.0, 1.0), 2.0, [F, F, F, F], "(0,0), (0,2): vertices; (2,2): on the top edge; (2,0): vertex"),
    (STEP, (1.5, 1.5), 1.0, [T, F, F, F], "(1,1) interior; (1,2), (2,2) on the top edge; (2,1) vertex of the notch"),
]


def verdicts(flags):
    """{(keep_inside, refine_geometry): verdict} by the truth table of geometry_base.py:40-76"""
    n_in, n = sum(flags), len(flags)
    return {(False, False): n_in == n, (False, True): n_in > 0, (True, False): n_in == 0, (True, True): n_in != n}


def nodes_of(center, width):
    h = width / 2.0
    return np.array([[center[0] - h, center[1] - h], [center[0] - h, center[1] + h],
                     [center[0] + h, center[1] + h], [center[0] + h, center[1] - h]])


@pytest.mark.parametrize("case", range(len(CASES)))
def test_host_predicate_against_the_table(case):
    poly, center, width, flags, why = CASES[case]
    nodes = pt.from_numpy(nodes_of(center, width))
    for keep_inside in (False, True):
        g = GeometryCoordinates2D("outline", keep_inside, poly, refine=True)
        # node by node first (a one-node "cell": removed as a body <=> that node is within)
        got = [GeometryCoordinates2D("outline", False, poly).check_cell(nodes[i:i + 1]) for i in range(4)]
        assert got == flags, (why, got)
        for refine_mode in (False, True):
            assert g.check_cell(nodes, refine_geometry=refine_mode) is verdicts(flags)[(keep_inside, refine_mode)], (why, keep_inside, refine_mode)


@pytest.mark.gpu
@pytest.mark.parametrize("poly_name", ["REF_SQUARE", "BOX2", "STEP"])
def test_device_kernel_against_the_table(poly_name):
    """``s3_mask_polygon`` on the cells of the table: centre + level + root width in, one verdict per cell out (the kernel
    forms the nodes as centre + direction * 0.5 * width / 2^level like the refine loop does)"""
    from sparsespatialsampling_amd import hipops
    from sparsespatialsampling_amd.geometry.coordinates_2d import _Outline
    poly = globals()[poly_name]
    rows = [c for c in CASES if c[0] is poly]
    root_width = 2.0                                                     # a cell of width w sits on level log2(2 / w)
    centers = np.array([c[1] for c in rows], dtype=np.float64)
    levels = np.array([int(round(np.log2(root_width / c[2]))) for c in rows], dtype=np.int32)
    assert all(root_width / 2.0 ** lv == c[2] for lv, c in zip(levels, rows))
    d_center, d_level = hipops.to_device(centers), hipops.to_device(levels)
    poly_dev = hipops.to_device(np.ascontiguousarray(_Outline(np.array(poly, dtype=np.float64)).xy))
    for keep_inside in (False, True):
        for refine_mode in (False, True):
            out = pt.zeros(len(rows), dtype=pt.uint8, device="cuda")
            hipops.mask_polygon(d_center, d_level, None, 0, len(rows), root_width, poly_dev, int(refine_mode), int(keep_inside), out)
            want = [verdicts(c[3])[(keep_inside, refine_mode)] for c in rows]
            assert out.cpu().numpy().astype(bool).tolist() == want, (poly_name, keep_inside, refine_mode)
