"""
The reference's own hot-path unit tests, restated as expectation tables (values taken from the assertions of
sparseSpatialSampling/tests/test_assignment_neighbors.py:11-212 and test_assignment_nodes.py:11-198) plus the
geometry truth tables of tests/test_{geometry_base,cube,sphere,cylinder,coordinates_2d}_geometry.py with the
``DummyCells`` fixture (tests/const.py:9-71).  CPU run: numerical kernels from the oracle backend (test-only injection);
GPU run (``-m gpu``): the HIP backend.
"""
import numpy as np
import pytest
import torch as pt

import sparsespatialsampling_amd.s_cube as s_cube
from sparsespatialsampling_amd import geometry
from tests.oracle_backend import OracleTreeBackend

N = None   # "no neighbour"

# cell id -> (position in list(_leaf_cells), expected nb ids per slot)   [2-D, uniform_level=2]
NB_2D = {
    5: (0, [N, N, 6, 7, 8, N, N, N]),
    7: (2, [6, 9, 12, 13, 18, 17, 8, 5]),
    13: (8, [12, 11, 14, 15, 16, 19, 18, 7]),
    10: (5, [N, N, N, N, 11, 12, 9, N]),
    15: (10, [14, N, N, N, N, N, 16, 13]),
    20: (-1, [17, 18, 19, N, N, N, N, N]),
}
# 3-D, uniform_level=2: slots 8..25 (lower plane, below, upper plane, above)
NB_3D = {
    9: (0, {8: N, 9: N, 10: 14, 11: 15, 12: 16, 13: N, 14: N, 15: N, 16: 13, 17: N, 18: N, 19: N, 20: N, 21: N, 22: N,
            23: N, 24: N, 25: N}),
    43: (34, {8: 46, 9: 53, 10: 56, 11: 61, 12: 70, 13: 69, 14: 48, 15: 45, 16: 47, 17: 14, 18: 21, 19: 24, 20: 29,
              21: 38, 22: 37, 23: 16, 24: 13, 25: 15}),
    32: (23, {8: 57, 9: 58, 10: 59, 11: N, 12: N, 13: N, 14: 67, 15: 66, 16: 60, 17: 25, 18: 26, 19: 27, 20: N, 21: N,
              22: N, 23: 35, 24: 34, 25: 28}),
}
NODES_2D = {5: (0, [0, 9, 10, 11]), 7: (2, [10, 12, 5, 13]), 12: (7, [12, 15, 17, 5]), 13: (8, [5, 17, 18, 19]),
            15: (10, [18, 20, 2, 21])}
NODES_3D_L1 = {1: [0, 8, 9, 10, 11, 12, 13, 14], 2: [8, 1, 15, 9, 12, 16, 17, 13], 3: [9, 15, 2, 18, 13, 17, 19, 20],
               4: [10, 9, 18, 3, 14, 13, 20, 21], 5: [11, 12, 13, 14, 4, 22, 23, 24], 6: [12, 16, 17, 13, 22, 5, 25, 23],
               7: [13, 17, 19, 20, 23, 25, 6, 26], 8: [14, 13, 20, 21, 24, 23, 26, 7]}


def _tree(d, n_pts, level, use_gpu, monkeypatch):
    if not use_gpu:
        monkeypatch.setattr(s_cube, "_make_backend", lambda v, t, k: OracleTreeBackend(v, t, k))
    xy = pt.randint(0, 11, (n_pts, d))                       # integer points, metric = 1 (as the reference tests)
    tree = s_cube.SamplingTree(xy, pt.ones(xy.size(0)), uniform_level=level,
                               geometry_obj=[geometry.CubeGeometry("domain", True, [0] * d, [10] * d)])
    tree._refine_uniform()
    return tree


BACKENDS = [pytest.param(False, id="oracle-backend"), pytest.param(True, id="hip", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("use_gpu", BACKENDS)
def test_assignment_nb_uniform_grid_2d(use_gpu, monkeypatch):
    tree = _tree(2, 25, 2, use_gpu, monkeypatch)
    leaves = list(tree._leaf_cells)
    for idx, (pos, expect) in NB_2D.items():
        cell = tree._cells[leaves[pos]]
        assert cell.index == idx
        assert [None if n is None else n.index for n in cell.nb] == expect


@pytest.mark.parametrize("use_gpu", BACKENDS)
def test_assignment_nb_uniform_grid_3d(use_gpu, monkeypatch):
    tree = _tree(3, 50, 2, use_gpu, monkeypatch)
    leaves = list(tree._leaf_cells)
    for idx, (pos, expect) in NB_3D.items():
        cell = tree._cells[leaves[pos]]
        assert cell.index == idx
        nb = cell.nb
        for slot, want in expect.items():
            assert (None if nb[slot] is None else nb[slot].index) == want, (idx, slot)


@pytest.mark.parametrize("use_gpu", BACKENDS)
def test_assignment_nodes_uniform_grid_2d(use_gpu, monkeypatch):
    tree = _tree(2, 25, 2, use_gpu, monkeypatch)
    leaves = list(tree._leaf_cells)
    for idx, (pos, expect) in NODES_2D.items():
        cell = tree._cells[leaves[pos]]
        assert cell.index == idx and cell.node_idx == expect


@pytest.mark.parametrize("use_gpu", BACKENDS)
def test_assignment_nodes_uniform_grid_3d_single_level(use_gpu, monkeypatch):
    tree = _tree(3, 50, 1, use_gpu, monkeypatch)
    assert len(tree.all_nodes) == 27
    leaves = list(tree._leaf_cells)
    for pos, (idx, expect) in enumerate(NODES_3D_L1.items()):
        cell = tree._cells[leaves[pos]]
        assert cell.index == idx and cell.node_idx == expect


# ---- geometry predicates with the DummyCells fixture -----------------------------------------------------------
CELLS_2D = {"inside": [[0, 0], [0, 1], [1, 1], [1, 0]], "outside": [[5, 5], [6, 5], [6, 6], [5, 6]],
            "partially": [[0.5, 0.5], [0.5, 1.5], [1.5, 1.5], [1.5, 0.5]]}
CELLS_3D = {"inside": [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]],
            "outside": [[5, 5, 5], [6, 5, 5], [6, 6, 5], [5, 6, 5], [5, 5, 6], [6, 5, 6], [6, 6, 6], [5, 6, 6]],
            "partially": [[0.5, 0.5, 0.5], [1.5, 0.5, 0.5], [1.5, 1.5, 0.5], [0.5, 1.5, 0.5],
                          [0.5, 0.5, 1.5], [1.5, 0.5, 1.5], [1.5, 1.5, 1.5], [0.5, 1.5, 1.5]]}


def cell(table, key):
    return pt.tensor(table[key], dtype=pt.float32)


# (keep_inside, cell) -> (remove?, refine-mode verdict): truth table of GeometryObject._apply_mask for a geometry that
# contains the unit square/cube (inclusive faces), reference tests/test_geometry_base.py:124-169
TRUTH = {(False, "inside"): (True, True), (False, "outside"): (False, False), (False, "partially"): (False, True),
         (True, "inside"): (False, False), (True, "outside"): (True, True), (True, "partially"): (False, True)}


@pytest.mark.parametrize("keep_inside", [False, True])
@pytest.mark.parametrize("where", ["inside", "outside", "partially"])
def test_geometry_truth_tables(keep_inside, where):
    remove, refine = TRUTH[(keep_inside, where)]
    geos2 = [geometry.CubeGeometry("g", keep_inside, [0, 0], [1, 1]),
             geometry.SphereGeometry("g", keep_inside, [0.5, 0.5], 0.75),
             geometry.GeometryCoordinates2D("g", keep_inside, [(-1, -1), (-1, 1.25), (1.25, 1.25), (1.25, -1)])]
    geos3 = [geometry.CubeGeometry("g", keep_inside, [0, 0, 0], [1, 1, 1]),
             geometry.SphereGeometry("g", keep_inside, [0.5, 0.5, 0.5], 0.9),
             geometry.CylinderGeometry3D("g", keep_inside, [(0.5, 0.5, -0.25), (0.5, 0.5, 1.25)], 0.75)]
    for g in geos2:
        assert g.check_cell(cell(CELLS_2D, where)) is remove, g.type
        assert g.check_cell(cell(CELLS_2D, where), refine_geometry=True) is refine, g.type
    for g in geos3:
        assert g.check_cell(cell(CELLS_3D, where)) is remove, g.type
        assert g.check_cell(cell(CELLS_3D, where), refine_geometry=True) is refine, g.type


# the flat-faced bodies of the reference's tests/test_{triangle,prism,tetrahedron,pyramid}_geometry.py: the partially
# overlapping cell is neither removed in body mode nor in domain mode
POLY_TRUTH = {(False, "inside"): True, (False, "outside"): False, (False, "partially"): False,
              (True, "inside"): False, (True, "outside"): True, (True, "partially"): False}


@pytest.mark.parametrize("keep_inside", [False, True])
@pytest.mark.parametrize("where", ["inside", "outside", "partially"])
def test_polytope_truth_tables(keep_inside, where):
    tri = geometry.TriangleGeometry("triangle", keep_inside, [(-1, -0.5), (0.25, 4), (1.5, -0.5)])
    prism = geometry.PrismGeometry3D("prism", keep_inside, [[(-1, -0.5, -0.5), (0.25, 4, -0.5), (1.5, -0.5, -0.5)],
                                                            [(-1, -0.5, 1.25), (0.25, 4, 1.25), (1.5, -0.5, 1.25)]])
    tet = geometry.TetrahedronGeometry3D("tetra", keep_inside, [[-1.5, 0.5, -0.1], [1.5, -1.5, -0.1], [1.5, 2.5, -0.1],
                                                                [0.5, 0.5, 3]])
    pyr = geometry.PyramidGeometry3D("pyramid", keep_inside, [[-1, -1, -0.25], [2, -1, -0.25], [2, 2, -0.25],
                                                              [-1, 2, -0.25], [0.5, 0.5, 3]])
    assert tri.check_cell(cell(CELLS_2D, where)) is POLY_TRUTH[(keep_inside, where)]
    for g in (prism, tet, pyr):
        assert g.check_cell(cell(CELLS_3D, where)) is POLY_TRUTH[(keep_inside, where)], g.type
    assert (tri.type, prism.type, tet.type, pyr.type) == ("triangle", "prism", "tetrahedron", "pyramid")


def test_polytope_argument_checks():
    T, P, H, Y = (geometry.TriangleGeometry, geometry.PrismGeometry3D, geometry.TetrahedronGeometry3D,
                  geometry.PyramidGeometry3D)
    T("triangle", False, [(0, 0), (1, 0), (0, 1)])
    assert T("triangle", False, [pt.tensor([0.0, 0.0]), pt.tensor([1.0, 0.0]), pt.tensor([0.0, 1.0])]).type == "triangle"
    with pytest.raises(AssertionError, match="Expected 3 points"):
        T("triangle", False, [(0, 0), (1, 0)])
    with pytest.raises(AssertionError, match="Expected 3 points"):
        T("triangle", False, [(0, 0), (1, 0), (0, 1), (1, 1)])
    with pytest.raises(AssertionError, match="have to contain exactly 2 entries"):
        T("triangle", False, [(0, 0), (1, 1, 5), (0, 1)])
    with pytest.raises(AssertionError, match="area of the triangle has to be larger than zero"):
        T("triangle", False, [(0, 0), (1, 1), (2, 2)])
    with pytest.raises(AssertionError):
        P("bad_prism", True, positions=[[[0, 0, 0], [1, 0, 0], [0, 1, 0]]])
    with pytest.raises(AssertionError):
        P("bad_prism2", True, positions=[[[0, 0, 0], [1, 0, 0]], [[0, 0, 1], [1, 0, 1], [0, 1, 1]]])
    for bad in ([], [[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5]]):
        with pytest.raises(AssertionError):
            H("bad_tetra", True, positions=bad)
    for bad in ([], [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 1]],
                [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 1], [0.5, 0.5]]):
        with pytest.raises(AssertionError):
            Y("bad_pyramid", True, nodes=bad)
    # derived attributes follow the reference's definitions
    t = T("t", True, [(0, 0), (2, 0), (0, 1)])
    assert t.main_width == 2.0 and pt.allclose(t.center, pt.tensor([2 / 3, 1 / 3], dtype=pt.float64))
    p = P("p", True, [[(0, 0, 0), (2, 0, 0), (0, 1, 0)], [(0, 0, 3), (2, 0, 3), (0, 1, 3)]])
    assert p.main_width == 3.0 and pt.allclose(p.center, pt.tensor([2 / 3, 1 / 3, 1.5], dtype=pt.float64))
    y = Y("y", True, [[-1, -1, -0.25], [2, -1, -0.25], [2, 2, -0.25], [-1, 2, -0.25], [0.5, 0.5, 3]])
    assert y._apex_idx == 4 and y.main_width == 3.25


def test_geometry_properties_and_argument_checks():
    c = geometry.CubeGeometry("domain", True, [0, 0], [2.2, 0.41])
    assert c.main_width == 2.2 and pt.allclose(c.center, pt.tensor([1.1, 0.205])) and c.type == "cube"
    s = geometry.SphereGeometry("s", False, [0.2, 0.2], 0.05, min_refinement_level=4)
    assert s.refine is True and s.min_refinement_level == 4 and s.main_width == 0.05      # refine switched on
    cy = geometry.CylinderGeometry3D("c", False, [(0.8, 1.0, -1), (0.8, 1.0, 1)], 0.05, refine=True)
    assert cy.main_width == 2.0 and pt.allclose(cy.center, pt.tensor([0.8, 1.0, 0.0]).float())
    p = geometry.GeometryCoordinates2D("p", False, [(0, 0), (2, 0), (2, 1), (0, 1)])
    assert p.main_width == 2 and p.pre_check_cell(pt.tensor([[0.5, 0.5]])) is True
    assert p.pre_check_cell(pt.tensor([[5.0, 5.0]])) is False
    with pytest.raises(AssertionError):
        geometry.CubeGeometry("", True, [0, 0], [1, 1])
    with pytest.raises(AssertionError):
        geometry.CubeGeometry("x", True, [0, 2], [1, 1])
    with pytest.raises(AssertionError):
        geometry.SphereGeometry("x", True, [0, 0], -1.0)
    with pytest.raises(AssertionError):
        geometry.CylinderGeometry3D("x", True, [(0, 0, 0), (0, 0, 0)], 1.0)
    with pytest.raises(AssertionError):
        geometry.CubeGeometry("x", True, [0, 0], [1, 1], refine=True, min_refinement_level=0)
